import os, sys
import numpy as np
os.environ.setdefault("SANN_NO_TORCH", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package(); lib = pkg.load_library()
n_t, n_c = 40_000, 500
ix = pkg.ClusterTweetIndex.synthetic(n_t, n_c, index_cap=2500, n_partitions=4)
cnt, cl, sc = ix.tweet_embeddings(0, 3000)
bad = 0
for i in range(0, 3000, 7):
    tid = lib.sann_synth_tweet_id(i, n_t, 1_700_000_000_000, 24)
    for j in range(cnt[i]):
        c, s = int(cl[i, j]), float(sc[i, j])
        t, v, _ = ix.get_list(c)
        d = dict(zip(t.tolist(), v.tolist()))
        if tid not in d and len(v) < 2500:
            bad += 1
            if bad <= 5:
                same_score = [int(x) for x, y in d.items() if y == s]
                print("tweet", i, "id", tid, "cluster", c, "score", s, "list len", len(v), "ids with same score", same_score[:3],
                      "n clusters of tweet", cnt[i], "j", j)
print("bad", bad)
cnt, cl, sc = ix.tweet_embeddings(0, n_t)
mask = np.arange(64)[None, :] < cnt[:, None]
per_cluster = np.bincount(cl[mask], minlength=n_c + 1)
for c in (262, 284, 23, 67, 1, 2, 100):
    t, v, _ = ix.get_list(c)
    print("cluster", c, "tweets whose embedding has it", per_cluster[c], "list len", len(t))
print("total embedding entries", int(mask.sum()), "index postings", ix.info().n_postings_total)
