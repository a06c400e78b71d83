"""HNSW search on the device (include/hnsw_ann.h) against oracle/hnsw_oracle.c walking the SAME graph over the
SAME fp16-rounded vectors: ids, order and float distance bits must be identical (the walk is exact; the
distance arithmetic is specified to the last bit).  Graphs come from the library's host-side builder, which
follows HnswIndex.insert; its quality is checked by recall against the exhaustive search."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _compare(pkg, oracle, ix, metric, queries, k, ef, graph=None, stored=None):
    graph = ix.graph() if graph is None else graph
    stored = ix.stored_vectors() if stored is None else stored
    ids, dist, cnt = ix.search(queries, k, ef)
    pq = oracle.dense_prepare(int(metric), queries)
    for q in range(len(queries)):
        o_items, o_dist, _ = oracle.hnsw_search(int(metric), stored, graph, pq[q], k, ef)
        assert cnt[q] == len(o_items), (q, cnt[q], len(o_items))
        assert np.array_equal(ids[q, :cnt[q]], o_items), f"query {q}: neighbours differ"
        assert np.array_equal(dist[q, :cnt[q]].view(np.int32), o_dist.view(np.int32)), f"query {q}: distance bits differ"
    return ids, dist, cnt


@pytest.fixture(scope="module")
def small(pkg):
    rng = np.random.default_rng(3)
    x = rng.standard_normal((4000, 64)).astype(np.float32)
    out = {}
    for name in ("InnerProduct", "Cosine", "L2"):
        m = getattr(pkg.dense_ann.DistanceMetric, name)
        out[name] = (m, pkg.hnsw_ann.Hnsw.build(m, x, max_m=8, ef_construction=40, seed=5))
    yield x, out
    for _, ix in out.values():
        ix.close()


@pytest.mark.parametrize("metric", ["InnerProduct", "Cosine", "L2"])
@pytest.mark.parametrize("k,ef", [(10, 50), (1, 1), (50, 20), (200, 800)])
def test_walk_is_bit_exact(pkg, oracle, small, metric, k, ef):
    x, ixs = small
    m, ix = ixs[metric]
    rng = np.random.default_rng(k * 1000 + ef)
    _compare(pkg, oracle, ix, m, rng.standard_normal((24, 64)).astype(np.float32), k, ef)


def test_graph_structure_and_recall(pkg, small):
    x, ixs = small
    m, ix = ixs["Cosine"]
    lv, it, off, nb, entry, max_level = ix.graph()
    assert max_level >= 1 and lv.max() == max_level and 0 <= entry < len(x)
    sizes = np.diff(off)
    assert sizes[lv == 0].max() <= 16 and sizes[lv > 0].max() <= 8, "maxM0 = 2 maxM at level 0, maxM above"
    assert (lv == 0).sum() == len(x), "every inserted item has a level-0 entry"
    for e in range(0, len(lv), 97):
        assert it[e] not in nb[off[e]:off[e + 1]], "no self loops"
    # level draw (int)(-ln U / ln maxM): about 1/8 of the items reach level 1
    assert 0.06 < (lv == 1).sum() / len(x) < 0.2
    rng = np.random.default_rng(8)
    q = rng.standard_normal((100, 64)).astype(np.float32)
    ids, _, cnt = ix.search(q, 10, 120)
    bf = pkg.dense_ann.BruteForceIndex.build(m, x)
    t_ids, _, _ = bf.search(q, 10)
    bf.close()
    recall = np.mean([len(set(ids[i, :cnt[i]]) & set(t_ids[i])) / 10 for i in range(100)])
    st = ix.last_stats()
    assert st["distance_evals"] > 100 * 120 and st["expansions"] >= 100 and st["spilled_queries"] == 0
    ids2, _, cnt2 = ix.search(q, 10, 600)
    recall2 = np.mean([len(set(ids2[i, :cnt2[i]]) & set(t_ids[i])) / 10 for i in range(100)])
    # 64-d isotropic Gaussians are the hard case for a maxM = 8 graph; the point is that the walk reaches
    # the true neighbours and that a wider beam reaches more of them
    assert recall > 0.8 and recall2 > 0.95 and recall2 >= recall, (recall, recall2)


def test_duplicate_vectors_tie_like_the_jvm_heap(pkg, oracle):
    """Every vector four times: most comparisons in the walk are exact ties, resolved by java.util.PriorityQueue's
    sift order on both sides."""
    m = pkg.dense_ann.DistanceMetric.InnerProduct
    rng = np.random.default_rng(21)
    base = rng.standard_normal((300, 32)).astype(np.float32)
    x = np.concatenate([base] * 4)[rng.permutation(1200)]
    ix = pkg.hnsw_ann.Hnsw.build(m, x, max_m=6, ef_construction=30, seed=2)
    _compare(pkg, oracle, ix, m, rng.standard_normal((30, 32)).astype(np.float32), 40, 64)
    _compare(pkg, oracle, ix, m, x[:20], 8, 8)
    ix.close()


def test_large_beam_and_the_global_queue_path(pkg, oracle):
    m = pkg.dense_ann.DistanceMetric.L2
    rng = np.random.default_rng(30)
    x = rng.standard_normal((12000, 16)).astype(np.float32)
    ix = pkg.hnsw_ann.Hnsw.build(m, x, max_m=16, ef_construction=60, seed=4)
    q = rng.standard_normal((12, 16)).astype(np.float32)
    _compare(pkg, oracle, ix, m, q, 100, 1024)
    assert ix.last_stats()["spilled_queries"] == 0
    # the candidate queue's top is in LDS and its tail in global memory.  Shrink both (40 + 40 entries) to force the
    # re-run with the large global part, and demand the same answers; then shrink only the LDS part, so that nearly
    # every heap step crosses the LDS / global boundary
    import os
    os.environ["HNSW_DEBUG_CCAP"] = "40"
    try:
        _compare(pkg, oracle, ix, m, q, 100, 1024)
        assert ix.last_stats()["spilled_queries"] == len(q)
        for v in ("-40", "-1", "-7"):
            os.environ["HNSW_DEBUG_CCAP"] = v
            _compare(pkg, oracle, ix, m, q, 100, 1024)
            assert ix.last_stats()["spilled_queries"] == 0
    finally:
        del os.environ["HNSW_DEBUG_CCAP"]
    with pytest.raises(pkg.hnsw_ann.HnswError):
        ix.search(q, 10, 2000)
    ix.close()


def test_loaded_graph_ids_and_odd_dimensions(pkg, oracle):
    m = pkg.dense_ann.DistanceMetric.Cosine
    rng = np.random.default_rng(40)
    x = rng.standard_normal((1500, 100)).astype(np.float32)     # d = 100 -> padded to 128
    built = pkg.hnsw_ann.Hnsw.build(m, x, max_m=10, ef_construction=40, seed=9)
    graph = built.graph()
    ids = rng.permutation(1500).astype(np.int64) * 11 + 7
    loaded = pkg.hnsw_ann.Hnsw.from_graph(m, x, graph, ids, max_m=10)
    g2 = loaded.graph()
    assert all(np.array_equal(a, b) for a, b in zip(graph[:4], g2[:4])) and graph[4:] == g2[4:]
    q = rng.standard_normal((16, 100)).astype(np.float32)
    a_ids, a_dist, a_cnt = built.search(q, 20, 60)
    b_ids, b_dist, b_cnt = loaded.search(q, 20, 60)
    assert np.array_equal(ids[a_ids], b_ids) and np.array_equal(a_dist, b_dist) and np.array_equal(a_cnt, b_cnt)
    _compare(pkg, oracle, built, m, q, 20, 60)
    # reference-style calls
    nn = loaded.queryWithDistance(x[5], 3, pkg.hnsw_ann.HnswParams(ef=50))
    assert nn[0][0] == ids[5] and abs(nn[0][1]) < 2e-3
    assert loaded.query(x[9], 1, pkg.hnsw_ann.HnswParams(ef=50)) == [int(ids[9])]
    built.close(); loaded.close()


def test_degenerate_indexes(pkg, oracle):
    m = pkg.dense_ann.DistanceMetric.L2
    empty = pkg.hnsw_ann.Hnsw.from_graph(m, np.zeros((0, 8), np.float32), (np.zeros(0, np.int32), np.zeros(0, np.int64),
                                         np.zeros(1, np.int64), np.zeros(0, np.int64), -1, 0), max_m=4)
    ids, dist, cnt = empty.search(np.zeros((3, 8), np.float32), 5, 10)
    assert cnt.tolist() == [0, 0, 0]
    empty.close()
    one = pkg.hnsw_ann.Hnsw.build(m, np.ones((1, 8), np.float32), max_m=4, ef_construction=10)
    ids, dist, cnt = one.search(np.zeros((2, 8), np.float32), 5, 10)
    assert cnt.tolist() == [1, 1] and ids[:, 0].tolist() == [0, 0] and np.allclose(dist[:, 0], np.sqrt(8))
    one.close()
    with pytest.raises(pkg.hnsw_ann.HnswError):
        pkg.hnsw_ann.Hnsw.build(m, np.ones((4, 8), np.float32), max_m=64)


def test_concurrent_builder_gives_a_searchable_graph(pkg, oracle):
    """n_threads > 1: the graph depends on the interleaving, the walk over it is still the reference's walk."""
    m = pkg.dense_ann.DistanceMetric.Cosine
    rng = np.random.default_rng(77)
    x = rng.standard_normal((6000, 48)).astype(np.float32)
    ix = pkg.hnsw_ann.Hnsw.build(m, x, max_m=12, ef_construction=60, seed=3, n_threads=6)
    lv, it, off, nb, entry, max_level = ix.graph()
    assert (lv == 0).sum() == len(x) and np.diff(off)[lv == 0].max() <= 24
    q = rng.standard_normal((20, 48)).astype(np.float32)
    ids, _, cnt = _compare(pkg, oracle, ix, m, q, 10, 200)
    bf = pkg.dense_ann.BruteForceIndex.build(m, x)
    t_ids, _, _ = bf.search(q, 10)
    bf.close()
    assert np.mean([len(set(ids[i, :cnt[i]]) & set(t_ids[i])) / 10 for i in range(20)]) > 0.85
    one = pkg.hnsw_ann.Hnsw.build(m, x[:1500], max_m=12, ef_construction=60, seed=3, n_threads=1)
    two = pkg.hnsw_ann.Hnsw.build(m, x[:1500], max_m=12, ef_construction=60, seed=3, n_threads=1)
    assert all(np.array_equal(a, b) for a, b in zip(one.graph()[:4], two.graph()[:4])), "one thread: deterministic"
    ix.close(); one.close(); two.close()
