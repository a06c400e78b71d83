"""What cluster-range sharding with a per-shard top-k would lose ("measure the loss rather than assume it
is zero").  north_star names cluster-id ranges; this build shards by tweet hash instead, because a
candidate's score is a SUM over the query's clusters: with cluster ranges every GPU holds only a partial
(dot_g, nsq_g) of a candidate, and a top-k' taken per shard before the partials are added drops candidates whose
parts are individually unremarkable.  This script measures that on the benchmark's synthetic corpus, in numpy, with no
library kernel involved: for each query it forms the exact answer (all partials added, cosine, top-400) and the
answer of G cluster-range shards that each deliver their top-k' by their own partial score, the owner adding whatever
partials arrive.  Reported: recall@400 against the exact answer, and how many of the returned 400 carry a wrong
(incomplete) score.

usage: cluster_range_loss.py [tweets [queries]]      (host corpus generator; 1M tweets by default)
       cluster_range_loss.py --device 100000000 32   (lists exported from the device-built 100M index; needs a GPU)
"""
import os
import sys

import numpy as np

os.environ.setdefault("SANN_NO_TORCH", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
args = [a for a in sys.argv[1:] if not a.startswith("--")]
device = "--device" in sys.argv
T = int(args[0]) if args else 1_000_000
NQ = int(args[1]) if len(args) > 1 else 64
K, M = 400, 800
C = pkg.corpus.N_CLUSTERS

offs, cids, scs = pkg.corpus.make_queries(1024)
if device:
    index = pkg.ClusterTweetIndex.synthetic(T, C, seed=pkg.corpus.CORPUS_SEED, index_cap=2000, now_ms=pkg.corpus.NOW_MS)
    l_cids, l_offs, l_t, l_s = index.export_lists(cids[:offs[NQ]])
else:
    co = pkg.corpus.make_corpus(T)
    l_cids, l_offs, l_t, l_s = co.cluster_ids, co.list_offsets, co.tweet_ids, co.scores
row_of = {int(c): i for i, c in enumerate(l_cids)}


def top(ids, score, k):
    order = np.lexsort((ids, -score))[:k]  # score descending, id ascending: the build's tie order
    return ids[order]


def one_query(q, G, kp, rank_by):
    qc, qw = cids[offs[q]:offs[q + 1]], scs[offs[q]:offs[q + 1]]
    l2 = np.sqrt((qw * qw).sum())
    t_all, d_all, n_all, g_all = [], [], [], []
    for c, w in zip(qc, qw):
        r = row_of.get(int(c))
        if r is None:
            continue
        a, b = l_offs[r], min(l_offs[r + 1], l_offs[r] + M)
        t_all.append(l_t[a:b]); d_all.append(l_s[a:b] * w); n_all.append(l_s[a:b] ** 2)
        g_all.append(np.full(b - a, int(c) * G // C))  # equal-width cluster-id ranges (hot clusters are spread over ids)
    t, d, n, g = (np.concatenate(x) for x in (t_all, d_all, n_all, g_all))
    ids, inv = np.unique(t, return_inverse=True)
    dot = np.bincount(inv, d, len(ids)); nsq = np.bincount(inv, n, len(ids))
    exact_score = dot / l2 / np.sqrt(nsq)
    exact = top(ids, exact_score, K)
    # shards: partial sums per (candidate, shard); each shard delivers its top-k' by its own partial score
    got_dot = np.zeros(len(ids)); got_nsq = np.zeros(len(ids)); got_any = np.zeros(len(ids), bool)
    for s in range(G):
        m = g == s
        if not m.any():
            continue
        pd = np.bincount(inv[m], d[m], len(ids)); pn = np.bincount(inv[m], n[m], len(ids))
        have = pn > 0
        part = np.where(have, pd / l2 / np.sqrt(np.where(have, pn, 1.0)) if rank_by == "cosine" else pd, -np.inf)
        idx = np.flatnonzero(have)
        sel = idx[np.lexsort((ids[idx], -part[idx]))[:kp]]
        got_dot[sel] += pd[sel]; got_nsq[sel] += pn[sel]; got_any[sel] = True
    cand = np.flatnonzero(got_any)
    merged_score = got_dot[cand] / l2 / np.sqrt(got_nsq[cand])
    ans_pos = cand[np.lexsort((ids[cand], -merged_score))[:K]]
    ans = ids[ans_pos]
    wrong = int((got_nsq[ans_pos] != nsq[ans_pos]).sum())  # returned with an incomplete sum
    return len(np.intersect1d(ans, exact)) / max(len(exact), 1), wrong / max(len(ans), 1)


print(f"corpus: {T} tweets ({'device' if device else 'host'} generator), {NQ} queries, N=50 M={M} k={K}, cosine")
print("shards  k' per shard  shard ranks by  recall@400   returned with incomplete score")
for G in (2, 4, 8):
    for kp in (400, 800, 1600, 4000):
        for rank_by in ("cosine", "dot"):
            r = [one_query(q, G, kp, rank_by) for q in range(NQ)]
            print(f"{G:6d}  {kp:12d}  {rank_by:>14s}  {np.mean([x[0] for x in r]):10.4f}   {np.mean([x[1] for x in r]):.4f}")
