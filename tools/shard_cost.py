"""What one GPU of an N-GPU run does per step, measured on one GPU: shard 0 of N, 1024*N queries, partitions 32/N.
Prints the step time and the descriptor / unit / per-shard-merge kernel times.  usage: shard_cost.py N [P [tweets [k]]]  (k = per-shard list length, default 400)"""
import os
import sys

os.environ.setdefault("SANN_NO_TORCH", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
import time

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
P = int(sys.argv[2]) if len(sys.argv) > 2 else max(1, 32 // N)
T = int(sys.argv[3]) if len(sys.argv) > 3 else 100_000_000
K = int(sys.argv[4]) if len(sys.argv) > 4 else 400
nq = 1024 * N
offs, cids, scs = pkg.corpus.make_queries(nq)
index = pkg.ClusterTweetIndex.synthetic(T, n_partitions=P, shard_id=0, n_shards=N)
cfg = pkg.SimClustersANNConfig(maxNumResults=K, annAlgorithm=pkg.ScoringAlgorithm.CosineSimilarity)
qb = pkg.QueryBatch(index, offs, cids, scs, cfg, now_ms=pkg.corpus.NOW_MS)
for _once in (0,):
    for _ in range(3):
        qb.run(); qb.finish()
    qb.set_profiling(True)
    fb0 = qb.stats().n_fallback_units
    t0 = time.perf_counter()
    for _ in range(10):
        qb.run(); qb.finish()
    wall = (time.perf_counter() - t0) / 10
    u, m, n = qb.kernel_times()
    d = qb.desc_time()
    qb.set_profiling(False)
    st = qb.stats()
    # the same with two batches in flight (bench.py's default): unit kernels chained across two streams
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    streams = []
    for _ in range(2):
        h = C.c_void_p()
        assert hip.hipStreamCreateWithFlags(C.byref(h), 1) == 0
        streams.append(h.value)
    qb2 = pkg.QueryBatch(index, offs, cids, scs, cfg, now_ms=pkg.corpus.NOW_MS)
    pair = [qb, qb2]
    for rep in range(2):
        t0 = time.perf_counter()
        for i in range(20):
            j = i & 1
            pair[j].run_after(streams[j], pair[1 - j], after_merge=False)
            if i:
                pair[1 - j].finish(streams[1 - j])
        pair[1].finish(streams[1])
        wall2 = (time.perf_counter() - t0) / 20
    print(f"N={N} P={P} k={K}: two batches in flight: step {wall2 * 1e3:.3f} ms")
    print(f"N={N} P={P} k={K}: step {wall * 1e3:.3f} ms, fallback/step {(st.n_fallback_units - fb0) / 10:.0f}, desc {d / n * 1e3:.0f} us, unit {u / n * 1e3:.0f} us, merge {m / n * 1e3:.0f} us, "
          f"fallback units {st.n_fallback_units}, units {st.n_units}")
