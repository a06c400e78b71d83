"""HNSW walk oracle (oracle/hnsw_oracle.c) on graphs small enough to trace by hand.
PARITY UNPINNED in the float distances (no fixture in the reference); the walk itself is checked against
the reference's control flow (HnswIndex.java:447-475,538-623)."""
import numpy as np


def _graph(entries, entry_point, max_level):
    lv = np.array([e[0] for e in entries], np.int32)
    it = np.array([e[1] for e in entries], np.int64)
    off = np.cumsum([0] + [len(e[2]) for e in entries]).astype(np.int64)
    nb = np.array([x for e in entries for x in e[2]], np.int64)
    return lv, it, off, nb, entry_point, max_level


def test_chain_walk_finds_the_far_end(oracle):
    # points on a line, each linked to its neighbours only: the beam has to walk the chain
    x = np.arange(10, dtype=np.float32)[:, None] * np.array([[1.0, 0.0]], np.float32)
    entries = [(0, i, [j for j in (i - 1, i + 1) if 0 <= j < 10]) for i in range(10)]
    g = _graph(entries, 0, 0)
    q = np.array([8.2, 0.0], np.float32)
    items, dist, evals = oracle.hnsw_search(0, x, g, q, 3, 3)
    assert items.tolist() == [8, 9, 7]
    assert np.allclose(dist, [0.2, 0.8, 1.2], atol=1e-3)
    # ef = 1: the beam keeps one result; the walk still follows strictly improving candidates
    items, _, _ = oracle.hnsw_search(0, x, g, q, 1, 1)
    assert items.tolist() == [8]


def test_upper_layers_descend_greedily_and_k_gt_found(oracle):
    x = np.array([[0, 0], [10, 0], [10, 1], [0, 1]], np.float32)
    # level 1: 0 <-> 1 ; level 0: 1 <-> 2 only (0 and 3 unreachable at level 0 from 1)
    g = _graph([(1, 0, [1]), (1, 1, [0]), (0, 1, [2]), (0, 2, [1])], 0, 1)
    items, dist, evals = oracle.hnsw_search(0, x, g, np.array([10, 0.9], np.float32), 5, 10)
    assert items.tolist() == [2, 1]          # only what the level-0 component holds, ascending
    assert evals == 1 + 1 + 1 + 1 + 1        # entry, its level-1 neighbour, (second pass) node 0 again, level-0 entry, node 2


def test_empty_index_and_ties_follow_the_java_heap(oracle):
    x = np.zeros((0, 4), np.float32)
    g = _graph([], -1, 0)
    items, dist, _ = oracle.hnsw_search(2, x, g, np.zeros(4, np.float32), 3, 3)
    assert len(items) == 0
    # five identical vectors: every distance ties; the order is whatever java.util.PriorityQueue yields
    x = np.ones((5, 4), np.float32)
    g = _graph([(0, i, [j for j in range(5) if j != i]) for i in range(5)], 0, 0)
    items, dist, _ = oracle.hnsw_search(2, x, g, np.ones(4, np.float32), 5, 5)
    assert sorted(items.tolist()) == [0, 1, 2, 3, 4] and np.all(dist == dist[0])
    # traced by hand: w-queue offers 0,1,2,3,4 (all equal, no sift), polls yield 0,4,3,2,1; reversed
    assert items.tolist() == [1, 2, 3, 4, 0]
