"""The device-side batched HNSW builder (hnsw_index_build_insert_gpu).  It is the reference's multi-writer insertion mode
taken wide (HnswIndex.java:150-200,376-380), so its graph is NOT the sequential one and cannot be compared entry for entry;
what can be checked: it is a well-formed HNSW graph, searches on it are the oracle's walk on the same graph bit for bit, and
its recall is that of the host-built (sequential, reference-identical) graph."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _recall(ix, bf_ids, q, k, ef):
    ids, _, cnt = ix.search(q, k, ef)
    return float(np.mean([len(set(ids[i, :cnt[i]]) & set(bf_ids[i])) / k for i in range(len(q))]))


@pytest.mark.parametrize("metric", ["Cosine", "L2", "InnerProduct"])
def test_gpu_built_graph_is_well_formed_and_recalls_like_the_host_built_one(pkg, oracle, metric):
    m = getattr(pkg.dense_ann.DistanceMetric, metric)
    rng = np.random.default_rng(5)
    centres = rng.standard_normal((200, 48)).astype(np.float32) * 2.0
    x = (centres[rng.integers(0, 200, 40_000)] + rng.standard_normal((40_000, 48)).astype(np.float32) * 0.6).astype(np.float32)
    q = (centres[rng.integers(0, 200, 200)] + rng.standard_normal((200, 48)).astype(np.float32) * 0.6).astype(np.float32)
    bf = pkg.dense_ann.BruteForceIndex.build(m, x)
    t_ids, _, _ = bf.search(q, 10)
    bf.close()
    keys = (np.arange(len(x), dtype=np.int64) * 7 + 3)
    gpu = pkg.hnsw_ann.Hnsw.build(m, x, ids=keys, max_m=12, ef_construction=100, seed=4, gpu=True, batch=2048)
    host = pkg.hnsw_ann.Hnsw.build(m, x, ids=keys, max_m=12, ef_construction=100, seed=4, n_threads=8)
    try:
        lv, it, off, nb, entry, max_level = gpu.graph()
        sizes = np.diff(off)
        assert (lv == 0).sum() == len(x), "every item has a layer-0 entry"
        assert sizes[lv == 0].max() <= 24 and sizes[lv > 0].max() <= 12
        assert sizes[lv == 0].min() >= 1, "no isolated node"
        for e in range(0, len(lv), 211):
            row = nb[off[e]:off[e + 1]]
            assert it[e] not in row and len(set(row.tolist())) == len(row), "no self loops, no repeated neighbour"
        r_gpu = _recall(gpu, keys[t_ids], q, 10, 120)
        r_host = _recall(host, keys[t_ids], q, 10, 120)
        assert r_gpu > 0.9 and r_gpu > r_host - 0.03, (r_gpu, r_host)
        # searches on the device-built graph are still the reference's walk: oracle on the same graph, bit for bit
        stored, graph = gpu.stored_vectors(), gpu.graph()
        ids, dist, cnt = gpu.search(q[:16], 10, 60)
        pq = oracle.dense_prepare(int(m), q[:16])
        for i in range(16):
            o_items, o_dist, _ = oracle.hnsw_search(int(m), stored, graph, pq[i], 10, 60)
            assert np.array_equal(ids[i, :cnt[i]], keys[o_items]) and np.array_equal(dist[i, :cnt[i]].view(np.int32), o_dist.view(np.int32))
    finally:
        gpu.close(); host.close()


def test_small_and_degenerate_inputs(pkg):
    m = pkg.dense_ann.DistanceMetric.L2
    rng = np.random.default_rng(1)
    for n in (1, 5, 900, 3000):
        x = rng.standard_normal((n, 16)).astype(np.float32)
        ix = pkg.hnsw_ann.Hnsw.build(m, x, max_m=4, ef_construction=16, seed=2, gpu=True, batch=256)
        try:
            # (the batched builder is not deterministic and maxM = 4 makes a poor graph: with 8 queries and a bar of 7
            # hits the check failed once in a dozen runs)
            nqs = min(n, 64)
            ids, _, cnt = ix.search(x[:nqs] + 1e-3, 1, 64)
            assert all(cnt[i] == 1 for i in range(len(cnt)))
            assert np.mean(ids[:, 0] == np.arange(nqs)) >= 0.7
        finally:
            ix.close()
    with pytest.raises(pkg.hnsw_ann.HnswError):
        pkg.hnsw_ann.Hnsw.build(m, x, max_m=4, ef_construction=300, gpu=True)
