"""BASELINE.json's configurations at their stated sizes (C2, C3, C4), on the device-generated corpus
of SURVEY 8(d).  The oracle answers a sample of the queries from the posting lists copied back from the device
(export_lists); every query of the batch is checked through size-independent properties.

  C2  simclusters-ann top-400, 1M tweets x 144,428 clusters, SINGLE-query batches, all three algorithms
  C3  batched simclusters-ann, 1024 concurrent user queries, 100M tweets, one GPU: 64 queries against the oracle
  C4  ann/ dense d = 256 fp16, brute-force search, k = 200, 1024 queries -- 5M vectors (the oracle is a float64 scan
      on the host; 50M is bench.py --workload dense's size)
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N_CLUSTERS = 144_428


def _cfg(pkg, alg, k=400):
    # cr-mixer's default config with maxNumResults = 400
    return pkg.SimClustersANNConfig(maxNumResults=k, minScore=0.0, maxTopTweetsPerCluster=800, maxScanClusters=50,
                                    maxTweetCandidateAgeHours=24, minTweetCandidateAgeHours=0, annAlgorithm=alg)


def _check_properties(ids, scores, counts, msz, k):
    """What must hold for every query whatever its size: at most k results, ordered by (score desc, tweet id asc),
    no tweet twice, no more results than accumulated candidates."""
    for q in range(len(counts)):
        n = counts[q]
        assert 0 <= n <= k and n <= msz[q]
        s, t = scores[q, :n], ids[q, :n]
        assert np.all(s[:-1] >= s[1:])
        tie = s[:-1] == s[1:]
        assert np.all(t[:-1][tie] < t[1:][tie])
        assert len(np.unique(t)) == n
        assert np.all(np.isfinite(s)) and np.all(s >= 0.0)


def _check_oracle(oracle, lists, offs, cids, scs, cfg, now_ms, ids, scores, counts, msz, queries):
    for q in queries:
        o_ids, o_sc, o_msz = oracle.sann_query(cids[offs[q]:offs[q + 1]], scs[offs[q]:offs[q + 1]], None, cfg, now_ms, *lists)
        assert counts[q] == len(o_ids) and msz[q] == o_msz, (q, counts[q], len(o_ids), msz[q], o_msz)
        assert np.array_equal(ids[q, :counts[q]], o_ids), f"query {q}: ids differ"
        assert np.array_equal(scores[q, :counts[q]].view(np.int64), o_sc.view(np.int64)), f"query {q}: scores differ"


@pytest.fixture(scope="module")
def corpus_1m(pkg):
    ix = pkg.ClusterTweetIndex.synthetic(1_000_000, N_CLUSTERS, seed=pkg.corpus.CORPUS_SEED, index_cap=2000,
                                         now_ms=pkg.corpus.NOW_MS)
    yield ix
    ix.close()


@pytest.mark.parametrize("alg", ["CosineSimilarity", "LogCosineSimilarity", "DotProduct"])
def test_c2_single_query_1m_tweets(pkg, oracle, corpus_1m, alg):
    """configs[1]: one query per call (nq = 1), both through a prepared batch and through the one-call boundary."""
    index = corpus_1m
    cfg = _cfg(pkg, getattr(pkg.ScoringAlgorithm, alg))
    offs, cids, scs = pkg.corpus.make_queries(12)
    lists = index.export_lists(cids)
    now = pkg.corpus.NOW_MS
    for q in range(12):
        one = (offs[q:q + 2] - offs[q], cids[offs[q]:offs[q + 1]], scs[offs[q]:offs[q + 1]])
        qb = pkg.QueryBatch(index, *one, cfg, now_ms=now)
        qb.run()
        qb.finish()
        ids, scores, counts, msz = qb.results()
        assert qb.stats().n_fallback_units == 0
        qb.close()
        _check_properties(ids, scores, counts, msz, 400)
        _check_oracle(oracle, lists, *one, cfg, now, ids, scores, counts, msz, [0])
        e = pkg.simclusters_ann.get_tweet_candidates(index, *one, cfg, now_ms=now)
        assert np.array_equal(e[2], counts) and np.array_equal(e[0][0, :counts[0]], ids[0, :counts[0]])
        assert np.array_equal(e[1][0, :counts[0]].view(np.int64), scores[0, :counts[0]].view(np.int64))


@pytest.fixture(scope="module")
def corpus_100m(pkg):
    ix = pkg.ClusterTweetIndex.synthetic(100_000_000, N_CLUSTERS, seed=pkg.corpus.CORPUS_SEED, index_cap=2000,
                                         now_ms=pkg.corpus.NOW_MS)
    yield ix
    ix.close()


def test_c3_1024_queries_100m_tweets(pkg, oracle, corpus_100m):
    """configs[2], the benchmark's own workload: 1024 queries in one batch; 64 of them (spread over the batch) against
    the oracle bit for bit, all of them through the properties; no unit may leave the fast path."""
    index = corpus_100m
    info = index.info()
    assert info.n_clusters == N_CLUSTERS and info.n_postings_total > 200_000_000
    offs, cids, scs = pkg.corpus.make_queries(1024)
    now = pkg.corpus.NOW_MS
    checked = list(range(0, 1024, 16))
    sel = np.concatenate([cids[offs[q]:offs[q + 1]] for q in checked])
    lists = index.export_lists(sel)
    for alg, n_oracle in (("CosineSimilarity", 64), ("LogCosineSimilarity", 16), ("DotProduct", 16)):
        cfg = _cfg(pkg, getattr(pkg.ScoringAlgorithm, alg))
        qb = pkg.QueryBatch(index, offs, cids, scs, cfg, now_ms=now)
        qb.run()
        qb.finish()
        ids, scores, counts, msz = qb.results()
        st = qb.stats()
        qb.close()
        assert st.n_fallback_units == 0 and st.n_requeried == 0
        assert st.postings_scanned > 30_000_000  # ~40k postings per query
        assert int(counts.min()) == 400
        _check_properties(ids, scores, counts, msz, 400)
        _check_oracle(oracle, lists, offs, cids, scs, cfg, now, ids, scores, counts, msz, checked[:n_oracle])
    # the boundary call gives the same batch the same answer
    e = pkg.simclusters_ann.get_tweet_candidates(index, offs, cids, scs, cfg, now_ms=now)
    assert np.array_equal(e[2], counts) and np.array_equal(e[0], ids) and np.array_equal(e[1].view(np.int64), scores.view(np.int64))


def test_c4_dense_5m_vectors_k200(pkg):
    """configs[3]'s exhaustive leg at k = 200 (the production k, cr-mixer HnswANNSimilarityEngine.scala:52-53): 5M x 256
    fp16 vectors, 1024 queries; 8 queries against a float64 scan of the stored vectors on the host (distances within
    1e-5, ids equal wherever the reference distances are separated by more than that)."""
    da = pkg.dense_ann
    n, d, k, nq = 5_000_000, 256, 200, 1024
    ix = da.BruteForceIndex.synthetic(da.DistanceMetric.Cosine, n, d, seed=3)
    rng = np.random.default_rng(4)
    queries = rng.standard_normal((nq, d)).astype(np.float32)
    ids, dist, cnt = ix.search(queries, k)
    assert np.all(cnt == k)
    assert np.all(np.diff(dist, axis=1) >= 0)
    for q in range(0, nq, 97):
        assert len(np.unique(ids[q])) == k
    # host scan of 8 queries, chunked (250k vectors = 256 MB of float32 at a time)
    qs = list(range(0, nq, 128))
    pq = queries[qs].astype(np.float64)
    pq /= np.sqrt((pq ** 2).sum(axis=1))[:, None]
    pq = pq.astype(np.float32).astype(np.float16).astype(np.float64)  # the query as the index sees it
    best_d = np.full((len(qs), 0), np.inf)
    best_i = np.zeros((len(qs), 0), np.int64)
    step = 250_000
    for i0 in range(0, n, step):
        x = ix.stored_vectors(i0, min(step, n - i0)).astype(np.float64)
        dd = 1.0 - pq @ x.T
        ii = np.broadcast_to(np.arange(i0, i0 + x.shape[0], dtype=np.int64), dd.shape)
        best_d = np.concatenate([best_d, dd], axis=1)
        best_i = np.concatenate([best_i, ii], axis=1)
        order = np.lexsort((best_i, best_d), axis=1)[:, :k + 1]
        best_d = np.take_along_axis(best_d, order, axis=1)
        best_i = np.take_along_axis(best_i, order, axis=1)
    for j, q in enumerate(qs):
        r_d, r_i = best_d[j, :k], best_i[j, :k]
        np.testing.assert_allclose(dist[q], r_d, rtol=1e-5, atol=1e-5)
        tol = 1e-5 + 1e-5 * np.abs(best_d[j])
        gap = np.diff(best_d[j])  # k gaps over k+1 sorted reference distances
        clear = np.ones(k, bool)
        clear &= gap > 2 * tol[:k]          # separated from the next one (the (k+1)-th for the last place)
        clear[1:] &= gap[:-1] > 2 * tol[1:k]  # and from the previous one
        assert clear.sum() > k // 2
        assert np.array_equal(ids[q][clear], r_i[clear])
    ix.close()


def _float64_reference_topk(ix, pq, n, k, step):
    """Exact top-(k+1) of `pq` (float64 queries as the index sees them) over the index's stored vectors: a float32 BLAS
    pass per chunk shortlists 4k positions per query (its 1e-6 error cannot move a true top-(k+1) member out of a 4k
    shortlist), the shortlist is re-scored in float64."""
    nqs = pq.shape[0]
    short = 4 * k
    cand_s = np.full((nqs, 0), -np.inf, np.float32)
    cand_i = np.zeros((nqs, 0), np.int64)
    pq32 = pq.astype(np.float32)
    for i0 in range(0, n, step):
        x = ix.stored_vectors(i0, min(step, n - i0))
        sc = pq32 @ x.T
        part = np.argpartition(-sc, short, axis=1)[:, :short]
        cand_s = np.concatenate([cand_s, np.take_along_axis(sc, part, axis=1)], axis=1)
        cand_i = np.concatenate([cand_i, part.astype(np.int64) + i0], axis=1)
        if cand_s.shape[1] > short:
            keep = np.argpartition(-cand_s, short, axis=1)[:, :short]
            cand_s = np.take_along_axis(cand_s, keep, axis=1)
            cand_i = np.take_along_axis(cand_i, keep, axis=1)
        del x, sc
    best_d = np.empty((nqs, k + 1))
    best_i = np.empty((nqs, k + 1), np.int64)
    for j in range(nqs):
        rows = np.stack([ix.stored_vectors(int(i), 1)[0] for i in cand_i[j]]).astype(np.float64)
        dd = 1.0 - rows @ pq[j]
        order = np.lexsort((cand_i[j], dd))[:k + 1]
        best_d[j], best_i[j] = dd[order], cand_i[j][order]
    return best_d, best_i


def test_c4_dense_50m_vectors_k200_at_size(pkg):
    """BASELINE configs[3] at its stated size: 50M x d = 256 fp16 vectors, 1024 queries, k = 200 (the dense
    leg stopped at 5M).  Size-independent properties for every query (k results, ascending distances, distinct ids), and
    4 queries against a float64 scan of ALL 50M stored vectors (distances within 1e-5; ids equal wherever the reference
    distances are separated by more than the tolerance) -- BruteForceIndex.scala:66-91 restated on the host."""
    da = pkg.dense_ann
    n, d, k, nq = 50_000_000, 256, 200, 1024
    ix = da.BruteForceIndex.synthetic(da.DistanceMetric.Cosine, n, d, seed=3)
    rng = np.random.default_rng(5)
    queries = rng.standard_normal((nq, d)).astype(np.float32)
    ids, dist, cnt = ix.search(queries, k)
    assert np.all(cnt == k)
    assert np.all(np.diff(dist, axis=1) >= 0)
    assert ids.min() >= 0 and ids.max() < n
    for q in range(0, nq, 61):
        assert len(np.unique(ids[q])) == k
    qs = [0, 341, 682, 1023]
    pq = queries[qs].astype(np.float64)
    pq /= np.sqrt((pq ** 2).sum(axis=1))[:, None]
    pq = pq.astype(np.float32).astype(np.float16).astype(np.float64)  # the query as the index sees it
    best_d, best_i = _float64_reference_topk(ix, pq, n, k, 2_000_000)
    for j, q in enumerate(qs):
        np.testing.assert_allclose(dist[q], best_d[j, :k], rtol=1e-5, atol=1e-5)
        tol = 1e-5 + 1e-5 * np.abs(best_d[j])
        gap = np.diff(best_d[j])
        clear = np.ones(k, bool)
        clear &= gap > 2 * tol[:k]
        clear[1:] &= gap[:-1] > 2 * tol[1:k]
        assert clear.sum() > k // 4  # (50M i.i.d. neighbours crowd together: fewer gaps clear the tolerance than at 5M)
        assert np.array_equal(ids[q][clear], best_i[j, :k][clear])
        # and as sets, up to members within the tolerance of the k-th distance
        kth = best_d[j, k - 1]
        sure = best_i[j, :k][best_d[j, :k] < kth - 2 * tol[k - 1]]
        assert set(sure.tolist()) <= set(ids[q].tolist())
    ix.close()


def test_c4_hnsw_1m_vectors_walk_is_the_oracles(pkg, oracle):
    """BASELINE configs[3]'s HNSW leg at size: a 1M x d = 256 graph (maxM 16, efConstruction 200, i.i.d. N(0,1) as SURVEY
    8(d) says; built by the device builder), 64 queries at the production (k, ef) = (200, 800) and at (10, 100): ids, order
    and float distance bits equal the oracle's walk (HnswIndex.java:538-623 restated in oracle/hnsw_oracle.c) over the
    exported graph and the stored fp16-rounded vectors."""
    da, hn = pkg.dense_ann, pkg.hnsw_ann
    n, d = 1_000_000, 256
    rng = np.random.default_rng(11)
    x = rng.standard_normal((n, d)).astype(np.float32)
    m = da.DistanceMetric.Cosine
    ix = hn.Hnsw.build(m, x, max_m=16, ef_construction=200, seed=1, n_threads=16, gpu=True)
    del x
    graph, stored = ix.graph(), ix.stored_vectors()
    lv, it, off, nb, entry, max_level = graph
    assert (lv == 0).sum() == n and max_level >= 3
    q = rng.standard_normal((64, d)).astype(np.float32)
    pq = oracle.dense_prepare(int(m), q)
    for k, ef in ((200, 800), (10, 100)):
        ids, dist, cnt = ix.search(q, k, ef)
        assert np.all(cnt == k)
        for qi in range(len(q)):
            o_items, o_dist, _ = oracle.hnsw_search(int(m), stored, graph, pq[qi], k, ef)
            assert np.array_equal(ids[qi, :cnt[qi]], o_items), f"k={k} ef={ef} query {qi}: neighbours differ"
            assert np.array_equal(dist[qi, :cnt[qi]].view(np.int32), o_dist.view(np.int32)), f"k={k} ef={ef} query {qi}: distance bits differ"
    ix.close()
