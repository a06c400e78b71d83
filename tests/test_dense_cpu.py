"""Dense-search oracle (float64 restatement, oracle/oracle.py) on hand-checkable cases, and the
host-side shard composition.  PARITY UNPINNED: the reference ships no fixture for this path."""
import numpy as np


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
