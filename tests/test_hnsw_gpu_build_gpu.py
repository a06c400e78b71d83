"""The device-side batched HNSW builder (hnsw_index_build_insert_gpu).  It is the reference's multi-writer insertion mode
taken wide (HnswIndex.java:150-200,376-380) with the interleaving fixed, so its graph is NOT the sequential one -- but it is ONE
graph: oracle/hnsw_oracle.c restates the batched insertion (oracle_hnsw_build_batched) and the device's graph must equal it
entry for entry; two builds of one input must be identical; it is a well-formed HNSW graph, searches on it are the oracle's
walk on the same graph bit for bit, and its recall is that of the host-built (sequential, reference-identical) graph."""
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


def _levels_of(graph, n):
    lv, it = graph[0], graph[1]
    out = np.zeros(n, np.int32)
    np.maximum.at(out, it, lv)
    out[graph[4]] = graph[5]  # the entry point carries maxLevel, but has a key only on the layers where a back link reached it
    return out


def _same_graph(a, b):
    return all(np.array_equal(x, y) for x, y in zip(a[:4], b[:4])) and a[4] == b[4] and a[5] == b[5]


@pytest.mark.parametrize("metric,n,d,max_m,efc,batch", [("L2", 3000, 24, 6, 40, 256), ("Cosine", 2500, 70, 8, 64, 128),
                                                        ("InnerProduct", 2000, 16, 4, 16, 512), ("L2", 1500, 8, 2, 256, 64)])
def test_device_built_graph_is_the_batched_oracles_graph(pkg, oracle, metric, n, d, max_m, efc, batch):
    m = getattr(pkg.dense_ann.DistanceMetric, metric)
    rng = np.random.default_rng(n + d)
    x = rng.standard_normal((n, d)).astype(np.float32)
    x[n // 2] = x[n // 3]  # equal rows: equal distances must order the same way on both sides
    x[n // 2 + 1] = x[n // 3]
    gpu = pkg.hnsw_ann.Hnsw.build(m, x, max_m=max_m, ef_construction=efc, seed=11, gpu=True, batch=batch)
    try:
        g = gpu.graph()
        levels = _levels_of(g, n)
        assert levels.max() >= 2, "the upper layers are part of what is compared"
        want = oracle.hnsw_build_batched(int(m), gpu.stored_vectors(), levels, max_m, efc, batch)
        assert g[4] == want[4] and g[5] == want[5], "entry point / max level"
        assert np.array_equal(g[0], want[0]) and np.array_equal(g[1], want[1]), "the same HnswNode(level, item) keys"
        assert np.array_equal(g[2], want[2]) and np.array_equal(g[3], want[3]), "the same lists, in the same order"
        rounds, unseen, prunes, dropped = gpu.build_stats()
        assert rounds > n // batch and unseen == 0 and dropped == 0
        # the levels-given entry point builds the same graph
        again = pkg.hnsw_ann.Hnsw.build(m, x, max_m=max_m, ef_construction=efc, levels=levels, gpu=True, batch=batch)
        try:
            assert _same_graph(again.graph(), g)
        finally:
            again.close()
    finally:
        gpu.close()


def test_a_small_candidate_queue_prunes_like_the_oracle(pkg, oracle, monkeypatch):
    """With beams of <= 256 the 1024-entry candidate queue of a construction walk practically never fills (no prune in any
    build tried, tools/hnsw_build_probe.py), so the test shrinks it (HNSW_BUILD_CCAP, read per build): the prune path runs,
    some candidates are even dropped because a pruned queue is still full, and the graph is still the oracle's, which
    prunes and drops by the same rule."""
    m = pkg.dense_ann.DistanceMetric.L2
    rng = np.random.default_rng(77)
    n, d = 5000, 12
    x = rng.standard_normal((n, d)).astype(np.float32)
    for ccap in (40, 150):
        monkeypatch.setenv("HNSW_BUILD_CCAP", str(ccap))
        gpu = pkg.hnsw_ann.Hnsw.build(m, x, max_m=16, ef_construction=128, seed=3, gpu=True, batch=512)
        try:
            g = gpu.graph()
            rounds, unseen, prunes, dropped = gpu.build_stats()
            want = oracle.hnsw_build_batched(int(m), gpu.stored_vectors(), _levels_of(g, n), 16, 128, 512, ccap=ccap)
            assert _same_graph(g, want)
            assert prunes > 0 and (ccap > 128 or dropped > 0), (prunes, dropped)
        finally:
            gpu.close()


def test_two_device_builds_of_one_input_are_one_graph(pkg):
    m = pkg.dense_ann.DistanceMetric.Cosine
    rng = np.random.default_rng(9)
    x = rng.standard_normal((60_000, 64)).astype(np.float32)
    a = pkg.hnsw_ann.Hnsw.build(m, x, max_m=16, ef_construction=100, seed=5, gpu=True)
    b = pkg.hnsw_ann.Hnsw.build(m, x, max_m=16, ef_construction=100, seed=5, gpu=True)
    try:
        assert _same_graph(a.graph(), b.graph())
    finally:
        a.close(); b.close()


def test_small_and_degenerate_inputs(pkg):
    """Tiny inputs, and graphs with few links.  The build is deterministic, so every number below is ONE number per
    (n, parameters), not a distribution.  maxM = 4 / efConstruction = 16 makes a poor graph whoever builds it: the host's
    sequential, reference-identical build finds 0.77 of the 64 probes at n = 3000 (tools/hnsw_build_probe.py), so at those
    parameters the bar is the sequential graph's own number; at maxM = 8 / efConstruction = 40 it is 0.85 outright."""
    m = pkg.dense_ann.DistanceMetric.L2
    rng = np.random.default_rng(1)

    def hits(ix, x, nqs):
        ids, _, cnt = ix.search(x[:nqs] + 1e-3, 1, 64)
        assert all(cnt[i] == 1 for i in range(len(cnt)))
        return float(np.mean(ids[:, 0] == np.arange(nqs)))

    for n in (1, 2, 5, 900, 3000):
        x = rng.standard_normal((n, 16)).astype(np.float32)
        nqs = min(n, 64)
        ix = pkg.hnsw_ann.Hnsw.build(m, x, max_m=4, ef_construction=16, seed=2, gpu=True, batch=256)
        host = pkg.hnsw_ann.Hnsw.build(m, x, max_m=4, ef_construction=16, seed=2)
        try:
            assert hits(ix, x, nqs) >= hits(host, x, nqs) - 0.02, n
        finally:
            ix.close(); host.close()
        ix = pkg.hnsw_ann.Hnsw.build(m, x, max_m=8, ef_construction=40, seed=2, gpu=True, batch=256)
        try:
            assert hits(ix, x, nqs) >= 0.85, n
        finally:
            ix.close()
    with pytest.raises(pkg.hnsw_ann.HnswError):
        pkg.hnsw_ann.Hnsw.build(m, x, max_m=4, ef_construction=300, gpu=True)


def test_the_visited_undo_log_of_large_indexes_on_a_small_one(pkg, oracle, monkeypatch):
    """From 8M vectors up a walk logs the nodes it marks visited and clears exactly those bitmap words when it ends (wiping
    6.25 MB per walk at 50M was a third of the build and 5 ms of every search batch).  HNSW_DEBUG_VLOG forces that path here:
    the device-built graph is still the oracle's, searches are still the oracle's walk, and a SECOND search on the same
    index -- whose bitmaps were cleaned by the walks, not by a memset -- gives the same answers."""
    monkeypatch.setenv("HNSW_DEBUG_VLOG", "1")
    m = pkg.dense_ann.DistanceMetric.L2
    rng = np.random.default_rng(123)
    n, d = 4000, 20
    x = rng.standard_normal((n, d)).astype(np.float32)
    gpu = pkg.hnsw_ann.Hnsw.build(m, x, max_m=8, ef_construction=64, seed=9, gpu=True, batch=256)
    try:
        g = gpu.graph()
        want = oracle.hnsw_build_batched(int(m), gpu.stored_vectors(), _levels_of(g, n), 8, 64, 256)
        assert _same_graph(g, want)
        q = rng.standard_normal((40, d)).astype(np.float32)
        stored = gpu.stored_vectors()
        pq = oracle.dense_prepare(int(m), q)
        first = gpu.search(q, 20, 100)
        second = gpu.search(q, 20, 100)
        third = gpu.search(q[::-1].copy(), 20, 100)
        assert all(np.array_equal(a, b) for a, b in zip(first, second))
        assert np.array_equal(third[0][::-1], first[0])
        for i in range(0, 40, 3):
            o_items, o_dist, _ = oracle.hnsw_search(int(m), stored, g, pq[i], 20, 100)
            assert np.array_equal(first[0][i, :first[2][i]], o_items) and np.array_equal(first[1][i, :first[2][i]].view(np.int32), o_dist.view(np.int32))
    finally:
        gpu.close()
