"""Dense-search oracle (float64 restatement, oracle/oracle.py) on hand-checkable cases, and the
host-side shard composition.  PARITY UNPINNED: the reference ships no fixture for this path."""
import numpy as np
import pytest


def test_metric_definitions(oracle):
    x = np.array([[3.0, 4.0], [1.0, 0.0], [0.0, 2.0]], np.float32)
    q = np.array([1.0, 0.0], np.float32)
    # L2: plain Euclidean distance (Metric.scala:88-97)
    assert np.allclose(oracle.dense_distances(0, x, q), [np.sqrt(4 + 16), 0.0, np.sqrt(1 + 4)])
    # InnerProduct: 1 - dot (Metric.scala:150-158)
    assert np.allclose(oracle.dense_distances(2, x, q), [1 - 3, 1 - 1, 1 - 0])
    # Cosine: vectors normalised at insert and query time, then 1 - dot (DistanceFunctionGenerator.scala:12-30)
    xs, qs = oracle.dense_prepare(1, x), oracle.dense_prepare(1, q[None, :])[0]
    assert np.allclose(oracle.dense_distances(1, xs, qs), [1 - 0.6, 0.0, 1.0], atol=1e-3)


def test_bruteforce_orders_ascending_and_breaks_ties_by_id(oracle):
    x = np.array([[1.0, 0.0], [0.0, 1.0], [1.0, 0.0], [-1.0, 0.0]], np.float32)
    ids = np.array([40, 10, 20, 30], np.int64)
    order = np.argsort(ids)
    (got_ids, got_dist), = oracle.dense_bruteforce(2, x[order], ids[order], np.array([[1.0, 0.0]], np.float32), 3)
    assert got_ids.tolist() == [20, 40, 10] and np.allclose(got_dist, [0.0, 0.0, 1.0])
    (got_ids, _), = oracle.dense_bruteforce(2, x[order], ids[order], np.array([[1.0, 0.0]], np.float32), 10)
    assert got_ids.tolist() == [20, 40, 10, 30]


def test_compose_is_concat_sort_take(pkg):
    a = (np.array([[1, 5, 0]]), np.array([[0.1, 0.5, 0.0]], np.float32), np.array([2], np.int32))
    b = (np.array([[2, 3, 9]]), np.array([[0.1, 0.3, 0.9]], np.float32), np.array([3], np.int32))
    ids, dist, cnt = pkg.dense_ann.compose([a, b], 4)
    assert cnt.tolist() == [4] and ids[0].tolist() == [1, 2, 3, 5]
    assert np.allclose(dist[0], [0.1, 0.1, 0.3, 0.5])


def test_native_compose_equals_the_numpy_restatement_and_refuses_bad_counts(pkg):
    """dann_compose_shards (ComposedQueryable.queryWithDistance, ShardApi.scala:71-87) against concat + lexsort + take on ragged
    shard answers with ties across shards; host arithmetic, no GPU."""
    rng = np.random.default_rng(4)
    nq, k = 9, 7
    parts = []
    for k_in in (5, 7, 3):
        ids = rng.integers(0, 40, (nq, k_in)).astype(np.int64)
        dist = (rng.integers(0, 6, (nq, k_in)) / 4).astype(np.float32)  # few distinct values: ties across shards
        order = np.lexsort((ids, dist), axis=1) if False else np.argsort(dist, axis=1, kind="stable")
        ids, dist = np.take_along_axis(ids, order, 1), np.take_along_axis(dist, order, 1)
        parts.append((ids, dist, rng.integers(0, k_in + 1, nq).astype(np.int32)))
    got_ids, got_dist, got_cnt = pkg.dense_ann.compose(parts, k)
    for q in range(nq):
        i = np.concatenate([p[0][q, :p[2][q]] for p in parts])
        d = np.concatenate([p[1][q, :p[2][q]] for p in parts])
        o = np.lexsort((i, d))[:k]
        assert got_cnt[q] == len(o) and got_ids[q, :len(o)].tolist() == i[o].tolist() and got_dist[q, :len(o)].tolist() == d[o].tolist()
    bad = (parts[0][0], parts[0][1], np.full(nq, 99, np.int32))
    with pytest.raises(pkg.dense_ann.DannError):
        pkg.dense_ann.compose([bad], k)
