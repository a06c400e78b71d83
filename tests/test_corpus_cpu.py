"""Synthetic corpus generator invariants and oracle properties on it.  CPU only."""
import numpy as np


def test_corpus_invariants(pkg):
    co = pkg.corpus.make_corpus(20000, 2000, seed=11)
    assert np.all(np.diff(co.cluster_ids) > 0)
    assert len(np.unique(co.tweet_id_of)) == 20000
    for i in range(len(co.cluster_ids)):
        b, e = co.list_offsets[i], co.list_offsets[i + 1]
        s = co.scores[b:e]
        t = co.tweet_ids[b:e]
        assert e - b <= 2000
        assert np.all(s > 0)
        assert np.all(np.diff(s) <= 0), "lists are sorted by score descending"
        assert len(np.unique(t)) == len(t), "tweet ids unique inside a list"
    # snowflake ids fall in the 24 h window before now
    ms = (co.tweet_id_of >> 22) + 1288834974657
    assert ms.max() < co.now_ms and ms.min() >= co.now_ms - 24 * 3600_000
    # mean clusters per tweet ~ 25 * (1 - (24/25)^50) less duplicate draws
    n_t = np.diff(co.tweet_emb_offsets)
    assert 15 < n_t.mean() < 25 and n_t.max() <= 50


def test_queries(pkg):
    offs, cids, scs = pkg.corpus.make_queries(8, 2000, seed=5)
    assert list(np.diff(offs)) == [50] * 8
    for q in range(8):
        c = cids[offs[q]:offs[q + 1]]
        assert len(set(c.tolist())) == 50
    assert np.all(scs > 0)


def test_oracle_variants_agree_and_order_is_total(pkg, oracle):
    co = pkg.corpus.make_corpus(5000, 300, seed=3)
    offs, cids, scs = pkg.corpus.make_queries(6, 300, seed=4, clusters_per_user=20)
    cfg = pkg.SimClustersANNConfig(maxNumResults=50, maxTopTweetsPerCluster=100, maxScanClusters=10)
    for q in range(6):
        e_ids, e_sc = cids[offs[q]:offs[q + 1]], scs[offs[q]:offs[q + 1]]
        res = [oracle.sann_query(e_ids, e_sc, None, cfg, co.now_ms, co.cluster_ids, co.list_offsets, co.tweet_ids,
                                 co.scores, variant=v) for v in (0, 1, 2)]
        for r in res[1:]:
            assert np.array_equal(r[0], res[0][0]) and np.array_equal(r[1], res[0][1]) and r[2] == res[0][2]
        ids, sc, _ = res[0]
        assert len(ids) == len(set(ids.tolist()))
        # sorted by (score desc, id asc)
        for i in range(1, len(ids)):
            assert sc[i - 1] > sc[i] or (sc[i - 1] == sc[i] and ids[i - 1] < ids[i])
