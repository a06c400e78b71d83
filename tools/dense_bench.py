"""Exhaustive dense search throughput (BASELINE configs[3]: d=256 fp16, 50M vectors).  Prints one JSON
line: queries/s, the pass timings (HIP events inside dann_search) and the MFMA roofline fraction of
the full-index GEMM pass (2*N*nq*d_padded flop / pass-B time / 2.5 PFLOP/s dense fp16)."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--vectors", type=int, default=50_000_000)
    ap.add_argument("--dim", type=int, default=256)
    ap.add_argument("--queries", type=int, default=1024)
    ap.add_argument("--k", type=int, default=200, help="production k (cr-mixer HnswANNSimilarityEngine.scala:52-53)")
    ap.add_argument("--metric", default="Cosine")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--cpu-vectors", type=int, default=1_000_000, help="index sample of the CPU leg (0 = skip)")
    a = ap.parse_args()
    pkg = load_package()
    m = getattr(pkg.dense_ann.DistanceMetric, a.metric)
    t0 = time.time()
    ix = pkg.dense_ann.BruteForceIndex.synthetic(m, a.vectors, a.dim, seed=3)
    build_s = time.time() - t0
    q = np.random.default_rng(0).standard_normal((a.queries, a.dim)).astype(np.float32)
    for _ in range(a.warmup):
        ix.search(q, a.k)
    ta = tb = ts = 0.0
    t0 = time.time()
    for _ in range(a.steps):
        ids, dist, cnt = ix.search(q, a.k)
        x, y, z = ix.last_timing_ms()
        ta += x; tb += y; ts += z
    wall = (time.time() - t0) / a.steps
    dpad = 64
    while dpad < a.dim:
        dpad *= 2
    flop = 2.0 * a.vectors * a.queries * dpad
    tb_ms = tb / a.steps
    # CPU leg: the reference's BruteForceIndex is a linear scan with a size-k heap per query (BruteForceIndex.scala:66-91);
    # timed here as what a host does best -- one float32 GEMM (numpy / BLAS, all cores) over a SAMPLE of the stored
    # vectors plus a per-query partial selection -- and scaled linearly to the full index (the scan is O(N)).
    cpu = None
    if a.cpu_vectors > 0:
        ns = min(a.cpu_vectors, a.vectors)
        xs = ix.stored_vectors(0, ns)
        qn = q / np.linalg.norm(q, axis=1, keepdims=True) if a.metric == "Cosine" else q
        # one core budget for every cpu_baseline of this repo: 16 threads, the CPU share of a 1-GPU box (cgroup quota; bench.py's leg
        # and tools/hnsw_bench.py's use the same)
        from threadpoolctl import threadpool_limits
        CORES = 16
        with threadpool_limits(limits=CORES):
            t0 = time.time()
            sc = qn @ xs.T
            gemm_s = time.time() - t0
        kk = min(a.k, ns - 1)
        from concurrent.futures import ThreadPoolExecutor
        t0 = time.time()
        with ThreadPoolExecutor(CORES) as ex:  # (argpartition releases the GIL)
            list(ex.map(lambda r: np.argpartition(-sc[r:r + 64], kk, axis=1)[:, :kk], range(0, len(sc), 64)))
        cpu_s = gemm_s + time.time() - t0
        del sc, xs
        cpu = {"value": a.queries / (cpu_s * a.vectors / ns), "unit": "queries/s", "cores": CORES, "kind": "port",
               "sample": f"float32 BLAS GEMM + argpartition top-{kk} over the first {ns} stored vectors x {a.queries} queries "
                         f"({cpu_s:.2f} s, GEMM {gemm_s:.2f} s), scaled by {a.vectors / ns:.0f} to the full index; {CORES} threads"}
    print(json.dumps({
        "metric": "exhaustive dense queries/sec", "value": a.queries / wall, "unit": "queries/s",
        "config": {"workload": f"{a.vectors} x d={a.dim} fp16 {a.metric}, {a.queries} queries, k={a.k}"},
        "ms_per_batch": wall * 1e3, "pass_a_ms": ta / a.steps, "pass_b_ms": tb_ms, "select_ms": ts / a.steps,
        "build_s": build_s,
        "roofline": {"bound": "mfma", "achieved": flop / (tb_ms * 1e-3) / 1e12, "peak": 2500.0, "unit": "TFLOP/s",
                     "frac": flop / (tb_ms * 1e-3) / 2.5e15},
        "cpu_baseline": cpu,
        "index_gb_per_s_pass_b": a.vectors * dpad * 2 / (tb_ms * 1e-3) / 1e9,
        "sample_dist": [float(v) for v in dist[0, :3]]}))


if __name__ == "__main__":
    main()
