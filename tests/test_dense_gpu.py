"""Exhaustive dense search (include/dense_ann.h) against the float64 restatement in oracle/oracle.py
on the same fp16-rounded inputs.  Tolerance: 1e-5 relative (+1e-5 absolute) on distances -- the device
accumulates fp16 products in fp32 on the matrix cores; ids must agree wherever the oracle's distances
are separated by more than that tolerance.  PARITY UNPINNED against the JVM (see oracle.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-5, 1e-5


def _check(pkg, oracle, ix, metric, queries, k, ids=None, stored=None):
    got_ids, got_dist, cnt = ix.search(queries, k)
    stored = ix.stored_vectors() if stored is None else stored
    pq = oracle.dense_prepare(int(metric), queries)
    ref = oracle.dense_bruteforce(int(metric), stored, ids, pq, k)
    n = len(stored)
    for q, (r_ids, r_dist) in enumerate(ref):
        m = min(k, n)
        assert cnt[q] == m
        assert np.all(np.diff(got_dist[q, :m]) >= 0), "ascending by distance"
        np.testing.assert_allclose(got_dist[q, :m], r_dist, rtol=RTOL, atol=ATOL)
        # the same neighbours, up to swaps among distances closer than the tolerance
        tol = ATOL + RTOL * np.abs(r_dist)
        clear = np.ones(m, bool)
        clear[:-1] &= np.diff(r_dist) > 2 * tol[:-1]
        clear[1:] &= np.diff(r_dist) > 2 * tol[1:]
        # the last place is only "clear" if the (k+1)-th is separated too: recompute it
        assert np.array_equal(got_ids[q, :m][clear][:-1], r_ids[clear][:-1])
        assert len(set(got_ids[q, :m].tolist())) == m
    return got_ids, got_dist


@pytest.mark.parametrize("metric", ["Cosine", "InnerProduct", "L2"])
@pytest.mark.parametrize("d", [256, 64, 100, 512])
def test_matches_oracle(pkg, oracle, metric, d):
    m = getattr(pkg.dense_ann.DistanceMetric, metric)
    rng = np.random.default_rng(d)
    x = rng.standard_normal((5000, d)).astype(np.float32)
    ix = pkg.dense_ann.BruteForceIndex.build(m, x)
    _check(pkg, oracle, ix, m, rng.standard_normal((70, d)).astype(np.float32), 10)
    ix.close()


def test_ids_are_returned_and_order_ties(pkg, oracle):
    m = pkg.dense_ann.DistanceMetric.InnerProduct
    rng = np.random.default_rng(1)
    base = rng.standard_normal((300, 128)).astype(np.float16).astype(np.float32)
    x = np.concatenate([base, base, base])          # every vector three times: exact ties
    ids = rng.permutation(900).astype(np.int64) * 7 + 3
    ix = pkg.dense_ann.BruteForceIndex.build(m, x, ids)
    q = rng.standard_normal((9, 128)).astype(np.float32)
    got_ids, got_dist, cnt = ix.search(q, 30)
    order = np.argsort(ids, kind="stable")
    assert np.array_equal(ix.stored_vectors(), x[order]), "positions follow id order"
    ref = oracle.dense_bruteforce(int(m), x[order], ids[order], oracle.dense_prepare(int(m), q), 30)
    for qi, (r_ids, r_dist) in enumerate(ref):
        np.testing.assert_allclose(got_dist[qi], r_dist, rtol=RTOL, atol=ATOL)
        # triples of equal distance come out id-ascending, and whole triples agree with the oracle
        assert np.array_equal(got_ids[qi], r_ids)
    ix.close()


@pytest.mark.parametrize("n,k", [(1, 1), (31, 40), (513, 1024), (20000, 1000), (3000, 1)])
def test_small_indexes_and_large_k(pkg, oracle, n, k):
    """Too few tiles for pass A, k > n, and the buffer-overflow refinement (20000 x k=1000)."""
    m = pkg.dense_ann.DistanceMetric.L2
    rng = np.random.default_rng(n)
    x = rng.standard_normal((n, 48)).astype(np.float32)
    ix = pkg.dense_ann.BruteForceIndex.build(m, x)
    _check(pkg, oracle, ix, m, rng.standard_normal((33, 48)).astype(np.float32), k)
    ix.close()


def test_synthetic_index_and_reference_style_calls(pkg, oracle):
    m = pkg.dense_ann.DistanceMetric.Cosine
    ix = pkg.dense_ann.BruteForceIndex.synthetic(m, 200_000, 256, seed=9)
    stored = ix.stored_vectors()
    assert abs(float(np.linalg.norm(stored, axis=1).mean()) - 1.0) < 1e-3
    assert abs(float(stored.mean())) < 1e-3
    rng = np.random.default_rng(2)
    q = rng.standard_normal((40, 256)).astype(np.float32)
    _check(pkg, oracle, ix, m, q, 10, stored=stored)
    # a stored vector is its own nearest neighbour at distance ~0
    nn = ix.queryWithDistance(stored[12345], 3)
    assert nn[0][0] == 12345 and abs(nn[0][1]) < 1e-3
    assert ix.query(stored[777], 1) == [777]
    ix.close()


def test_sharded_search_composes_exactly(pkg):
    """ComposedQueryable (ShardApi.scala:71-87): per-shard top-k, concat, sort, take k == unsharded."""
    m = pkg.dense_ann.DistanceMetric.Cosine
    rng = np.random.default_rng(4)
    x = rng.standard_normal((6000, 64)).astype(np.float32)
    ids = np.arange(6000, dtype=np.int64) * 3
    q = rng.standard_normal((20, 64)).astype(np.float32)
    full = pkg.dense_ann.BruteForceIndex.build(m, x, ids)
    f_ids, f_dist, f_cnt = full.search(q, 25)
    parts = []
    for s in range(3):
        sel = np.arange(s, 6000, 3)
        sh = pkg.dense_ann.BruteForceIndex.build(m, x[sel], ids[sel])
        parts.append(sh.search(q, 25))
        sh.close()
    c_ids, c_dist, c_cnt = pkg.dense_ann.compose(parts, 25)
    assert np.array_equal(c_ids, f_ids) and np.array_equal(c_dist, f_dist) and np.array_equal(c_cnt, f_cnt)
    full.close()


def test_argument_errors(pkg):
    m = pkg.dense_ann.DistanceMetric.L2
    x = np.zeros((10, 16), np.float32)
    ix = pkg.dense_ann.BruteForceIndex.build(m, x)
    with pytest.raises(pkg.dense_ann.DannError):
        ix.search(np.zeros((1, 16), np.float32), 0)
    with pytest.raises(pkg.dense_ann.DannError):
        ix.search(np.zeros((1, 16), np.float32), 2000)
    with pytest.raises(ValueError):
        ix.search(np.zeros((1, 8), np.float32), 1)
    with pytest.raises(pkg.dense_ann.DannError):
        pkg.dense_ann.BruteForceIndex.build(m, np.zeros((4, 600), np.float32))
    ix.close()


@pytest.mark.parametrize("n,d,nq,k", [(6000, 64, 20, 25), (6000, 16, 1, 5), (40000, 128, 70, 10), (30000, 256, 200, 10)])
def test_repeated_searches_are_identical_and_exact(pkg, oracle, n, d, nq, k):
    """Race screen for the LDS-DMA query ring: short k-loops (small d, one or two query blocks) leave
    the least slack between a stage being issued and being read."""
    m = pkg.dense_ann.DistanceMetric.InnerProduct
    rng = np.random.default_rng(n + d)
    x = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal((nq, d)).astype(np.float32)
    ix = pkg.dense_ann.BruteForceIndex.build(m, x)
    first = _check(pkg, oracle, ix, m, q, k)
    for _ in range(15):
        ids, dist, _ = ix.search(q, k)
        assert np.array_equal(ids, first[0]) and np.array_equal(dist, first[1])
    ix.close()
