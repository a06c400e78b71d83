"""Device-side corpus generator + index builder (sann_index_build_synthetic): invariants of the
lists it builds, determinism, losslessness of the score-threshold shortcut, and query parity
against the oracle on the lists exported back from the device."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

T, C, CAP = 300_000, 3000, 300


@pytest.fixture(scope="module")
def synth(pkg):
    ix = pkg.ClusterTweetIndex.synthetic(T, C, index_cap=CAP, n_partitions=16)
    yield ix
    ix.close()


def test_list_invariants(pkg, synth):
    info = synth.info()
    assert info.n_clusters == C and info.n_postings == info.n_postings_total and info.max_list_len <= CAP
    total = 0
    lo = 1_700_000_000_000 - 24 * 3600_000
    for c in list(range(1, 60)) + list(range(C - 40, C + 1)):
        t, s, r = synth.get_list(c)
        total += len(t)
        assert len(t) <= CAP
        assert np.array_equal(r, np.arange(len(t))), "ranks are the positions in the sorted, capped list"
        assert np.all(s >= 0.001)
        assert np.all(np.diff(s) <= 0), "score descending"
        ties = np.nonzero(np.diff(s) == 0)[0]
        assert np.all(t[ties] < t[ties + 1]), "ties by tweet id ascending"
        assert len(np.unique(t)) == len(t)
        ms = (t >> 22) + 1288834974657
        assert ms.min() >= lo and ms.max() < 1_700_000_000_000
    assert total > 0
    # nearly every cluster of a 300k-tweet / 3000-cluster corpus is longer than the cap
    assert 0.97 * C * CAP < info.n_postings_total <= C * CAP


def test_tweet_ids_are_unique_and_in_window(pkg):
    lib = pkg.load_library()
    n = 200_000
    ids = np.array([lib.sann_synth_tweet_id(t, n, 1_700_000_000_000, 24) for t in range(0, n, 7)], np.int64)
    assert len(np.unique(ids)) == len(ids)
    ms = (ids >> 22) + 1288834974657
    assert ms.min() >= 1_700_000_000_000 - 24 * 3600_000 and ms.max() < 1_700_000_000_000


def test_deterministic_and_independent_of_partitioning(pkg, synth):
    other = pkg.ClusterTweetIndex.synthetic(T, C, index_cap=CAP, n_partitions=4)
    for c in (1, 2, 17, 500, 2999):
        a, b = synth.get_list(c), other.get_list(c)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    other.close()
    # shards split every list by tweet hash and keep the global ranks
    shards = [pkg.ClusterTweetIndex.synthetic(T, C, index_cap=CAP, n_partitions=8, shard_id=s, n_shards=3) for s in range(3)]
    assert sum(s.info().n_postings for s in shards) == shards[0].info().n_postings_total
    for c in (1, 42, 2500):
        full_t, full_s, _ = synth.get_list(c)
        parts = [s.get_list(c) for s in shards]
        t = np.concatenate([p[0] for p in parts]); r = np.concatenate([p[2] for p in parts])
        order = np.argsort(r)
        assert np.array_equal(r[order], np.arange(len(full_t))) and np.array_equal(t[order], full_t)
    for s in shards:
        s.close()


def test_score_threshold_shortcut_is_lossless(pkg, synth):
    """Lists built with cap 300 (thresholds active on every cluster) are exactly the first 300
    entries of the lists built with cap 2500 from the same seed."""
    big = pkg.ClusterTweetIndex.synthetic(T, C, index_cap=2500, n_partitions=16)
    for c in (1, 3, 77, 1500, 3000):
        a, b = synth.get_list(c), big.get_list(c)
        n = len(a[0])
        assert n == min(CAP, len(b[0])) and len(b[0]) >= n
        assert np.array_equal(a[0], b[0][:n]) and np.array_equal(a[1], b[1][:n])
    big.close()


def test_queries_on_device_built_index_match_oracle(pkg, oracle, synth):
    offs, cids, scs = pkg.corpus.make_queries(32, C, seed=77, clusters_per_user=50)
    now = 1_700_000_000_000
    lists = synth.export_lists(cids)
    for alg in (pkg.ScoringAlgorithm.CosineSimilarity, pkg.ScoringAlgorithm.DotProduct, pkg.ScoringAlgorithm.LogCosineSimilarity):
        cfg = pkg.SimClustersANNConfig(maxNumResults=400, maxTopTweetsPerCluster=200, annAlgorithm=alg)
        qb = pkg.QueryBatch(synth, offs, cids, scs, cfg, now_ms=now)
        qb.run(); qb.finish()
        ids, scores, counts, msz = qb.results()
        qb.close()
        for q in range(32):
            o_ids, o_sc, o_msz = oracle.sann_query(cids[offs[q]:offs[q + 1]], scs[offs[q]:offs[q + 1]], None, cfg, now, *lists)
            assert counts[q] == len(o_ids) and msz[q] == o_msz
            assert np.array_equal(ids[q, :counts[q]], o_ids)
            assert np.array_equal(scores[q, :counts[q]].view(np.int64), o_sc.view(np.int64))


def test_tweet_embeddings_agree_with_the_index(pkg):
    """The full embeddings the quality metric regenerates are the ones the index was built from:
    every (tweet, cluster, score) whose score reaches the cluster's list is in that list, bit for bit."""
    n_t, n_c = 40_000, 500
    ix = pkg.ClusterTweetIndex.synthetic(n_t, n_c, index_cap=2500, n_partitions=4)
    lib = pkg.load_library()
    cnt, cl, sc = ix.tweet_embeddings(0, 3000)
    assert cnt.min() >= 1 and cnt.max() <= 50
    lists = {}
    checked = present = 0
    for i in range(0, 3000, 7):
        tid = lib.sann_synth_tweet_id(i, n_t, 1_700_000_000_000, 24)
        assert len(set(cl[i, :cnt[i]].tolist())) == cnt[i], "clusters of a tweet are distinct"
        for j in range(cnt[i]):
            c, s = int(cl[i, j]), float(sc[i, j])
            if c not in lists:
                t, v, _ = ix.get_list(c)
                lists[c] = (dict(zip(t.tolist(), v.tolist())), float(v.min()) if len(v) else 0.0, len(v))
            d, vmin, n = lists[c]
            checked += 1
            if tid in d:
                assert d[tid] == s
                present += 1
            else:
                assert n == 2500 and s <= vmin, "a posting missing from a list must have lost to the cap"
    assert checked > 1000 and present > 0.5 * checked
    ix.close()


def test_exact_cosine_topk_matches_numpy_brute_force(pkg):
    n_t, n_c, k = 20_000, 400, 50
    ix = pkg.ClusterTweetIndex.synthetic(n_t, n_c, index_cap=300, n_partitions=4)
    lib = pkg.load_library()
    cnt, cl, sc = ix.tweet_embeddings(0, n_t)
    offs, cids, scs = pkg.corpus.make_queries(6, n_c, seed=3, clusters_per_user=20)
    ids, cos, got = ix.exact_cosine_topk(offs, cids, scs, k)
    tids = np.array([lib.sann_synth_tweet_id(t, n_t, 1_700_000_000_000, 24) for t in range(n_t)], np.int64)
    mask = np.arange(64)[None, :] < cnt[:, None]
    tn = np.sqrt((np.where(mask, sc, 0.0) ** 2).sum(1))
    for q in range(6):
        w = np.zeros(n_c + 1)
        w[cids[offs[q]:offs[q + 1]]] = scs[offs[q]:offs[q + 1]]
        dot = (np.where(mask, sc, 0.0) * w[np.where(mask, cl, 0)]).sum(1)
        ref = dot / (np.linalg.norm(w) * tn)
        order = np.lexsort((tids, -ref))[:k]
        assert got[q] == k
        # summation order differs from numpy's by an ulp, so near-ties may swap places: compare the
        # cosines position by position and the id sets away from the k-th boundary
        assert np.allclose(cos[q], ref[order], rtol=1e-12, atol=0)
        safe = ref[order] > ref[order][-1] * (1 + 1e-9)
        assert set(tids[order][safe].tolist()) <= set(ids[q].tolist())
    ix.close()
