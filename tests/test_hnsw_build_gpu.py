"""HNSW index construction: the library's host-side builder against the oracle's independent
restatement of HnswIndex.insert / mutuallyConnectNewElement / selectNearestNeighboursByHeuristic
(ann/src/main/java/com/twitter/ann/hnsw/HnswIndex.java:137-200,384-440,479-526) on the same fp16-rounded vectors
and the same per-item levels: the two graphs must be IDENTICAL -- same HnswNode keys, same neighbour lists in the
same order, same entry point and maximum level.  (Needs the GPU only because creating an index uploads it.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _as_dict(graph):
    lv, it, off, nb, entry, max_level = graph
    return {(int(lv[e]), int(it[e])): tuple(int(v) for v in nb[off[e]:off[e + 1]]) for e in range(len(lv))}, entry, max_level


def _levels(n, max_m, seed):
    rng = np.random.default_rng(seed)
    return np.minimum(60, (-np.log(1.0 - rng.random(n)) / np.log(max_m)).astype(np.int32)).astype(np.int32)


@pytest.mark.parametrize("metric", ["InnerProduct", "Cosine", "L2"])
@pytest.mark.parametrize("n,d,max_m,efc", [(1500, 64, 8, 40), (600, 100, 4, 10), (900, 32, 16, 200)])
def test_built_graph_equals_the_oracle(pkg, oracle, metric, n, d, max_m, efc):
    m = getattr(pkg.dense_ann.DistanceMetric, metric)
    rng = np.random.default_rng(n + d)
    x = rng.standard_normal((n, d)).astype(np.float32)
    levels = _levels(n, max_m, n)
    ix = pkg.hnsw_ann.Hnsw.build(m, x, max_m=max_m, ef_construction=efc, levels=levels)
    try:
        got, g_entry, g_max = _as_dict(ix.graph())
        want, w_entry, w_max = _as_dict(oracle.hnsw_build(int(m), ix.stored_vectors(), levels, max_m, efc))
    finally:
        ix.close()
    assert (g_entry, g_max) == (w_entry, w_max)
    assert got.keys() == want.keys()
    bad = [k for k in got if got[k] != want[k]]
    assert not bad, f"{len(bad)} of {len(got)} neighbour lists differ, first {bad[0]}: {got[bad[0]]} vs {want[bad[0]]}"


def test_duplicate_vectors_and_small_graphs(pkg, oracle):
    """Exact distance ties everywhere (every vector four times): the queues' sift order decides the lists."""
    m = pkg.dense_ann.DistanceMetric.L2
    rng = np.random.default_rng(4)
    base = rng.integers(-2, 3, (120, 16)).astype(np.float32)
    x = np.concatenate([base] * 4)
    for n in (1, 2, 3, 17, len(x)):
        levels = _levels(n, 4, n + 1)
        ix = pkg.hnsw_ann.Hnsw.build(m, x[:n], max_m=4, ef_construction=12, levels=levels)
        try:
            got = _as_dict(ix.graph())
            want = _as_dict(oracle.hnsw_build(int(m), ix.stored_vectors(), levels, 4, 12))
        finally:
            ix.close()
        assert got == want, n


def test_levels_are_validated(pkg):
    m = pkg.dense_ann.DistanceMetric.L2
    x = np.zeros((4, 8), np.float32)
    with pytest.raises(pkg.hnsw_ann.HnswError):
        pkg.hnsw_ann.Hnsw.build(m, x, max_m=4, ef_construction=4, levels=[0, 1, 61, 0])
    with pytest.raises(ValueError):
        pkg.hnsw_ann.Hnsw.build(m, x, max_m=4, ef_construction=4, levels=[0, 1])
