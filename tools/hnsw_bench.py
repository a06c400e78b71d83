"""HNSW walk throughput (BASELINE configs[3]: d=256 fp16 vectors; prod k=200 / ef=800 per SURVEY 8 row D3).
Builds the graph with the library's host builder (HnswIndex.insert semantics, multi-threaded), searches a batch
of queries, and reports queries/s, distance evaluations/s, the bytes those gathers move, and recall@k against
the exhaustive search of the same vectors.  One JSON line per (k, ef)."""
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
    ap.add_argument("--vectors", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=256)
    ap.add_argument("--queries", type=int, default=4096)
    ap.add_argument("--max-m", type=int, default=16)
    ap.add_argument("--ef-construction", type=int, default=200)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--build", choices=["host", "gpu"], default="gpu", help="host: the reference's insertion, 16 threads; gpu: the batched device builder")
    ap.add_argument("--configs", default="10:100,200:800")
    ap.add_argument("--clusters", type=int, default=0, help="synthetic data: 0 = i.i.d. N(0,1) (SURVEY 8d), n = mixture of n Gaussians")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--cpu-queries", type=int, default=256, help="queries of the CPU leg (0 = skip)")
    ap.add_argument("--f32-gen", action="store_true", help="draw the synthetic vectors as float32 directly, in slabs (half the host memory and time; a different stream than the default)")
    a = ap.parse_args()
    t_start = time.time()
    pkg = load_package()
    m = pkg.dense_ann.DistanceMetric.Cosine
    rng = np.random.default_rng(0)
    if a.clusters > 0:
        centres = rng.standard_normal((a.clusters, a.dim)).astype(np.float32)
        x = centres[rng.integers(0, a.clusters, a.vectors)] + 0.6 * rng.standard_normal((a.vectors, a.dim)).astype(np.float32)
        q = centres[rng.integers(0, a.clusters, a.queries)] + 0.6 * rng.standard_normal((a.queries, a.dim)).astype(np.float32)
    elif a.f32_gen:
        x = np.empty((a.vectors, a.dim), np.float32)
        for r0 in range(0, a.vectors, 1 << 20):
            x[r0:r0 + (1 << 20)] = rng.standard_normal((min(1 << 20, a.vectors - r0), a.dim), dtype=np.float32)
        q = rng.standard_normal((a.queries, a.dim), dtype=np.float32)
    else:
        x = rng.standard_normal((a.vectors, a.dim)).astype(np.float32)
        q = rng.standard_normal((a.queries, a.dim)).astype(np.float32)
    print(f"vectors drawn, {time.time() - t_start:.1f} s", file=sys.stderr, flush=True)
    t0 = time.time()
    ix = pkg.hnsw_ann.Hnsw.build(m, x, max_m=a.max_m, ef_construction=a.ef_construction, seed=1, n_threads=a.threads, gpu=a.build == "gpu")
    build_s = time.time() - t0
    print(f"graph built in {build_s:.1f} s", ix.build_stats() if a.build == "gpu" else "", file=sys.stderr, flush=True)
    bf = pkg.dense_ann.BruteForceIndex.build(m, x)
    # CPU leg: the reference's walk (HnswIndex.searchKnn, restated in oracle/hnsw_oracle.c) over the SAME graph, one
    # thread, a bounded sample of the queries; also checks that the device results of those queries are the oracle's
    from __graft_entry__ import load_oracle
    oracle = load_oracle()
    graph = stored = None
    for cfg in a.configs.split(","):
        k, ef = (int(v) for v in cfg.split(":"))
        ix.search(q[:64], k, ef)
        t0 = time.time()
        for _ in range(a.steps):
            ids, dist, cnt = ix.search(q, k, ef)
        wall = (time.time() - t0) / a.steps
        st = ix.last_stats()
        nt = min(256, a.queries)
        t_ids, _, _ = bf.search(q[:nt], k)
        recall = float(np.mean([len(set(ids[i, :cnt[i]].tolist()) & set(t_ids[i].tolist())) / k for i in range(nt)]))
        n_cpu = min(a.cpu_queries, a.queries)
        cpu = None
        if n_cpu > 0:
            if graph is None:
                graph, stored = ix.graph(), ix.stored_vectors()
            pq = oracle.dense_prepare(int(m), q[:n_cpu])
            # one core budget for every cpu_baseline of this repo: 16 threads, the CPU share of a 1-GPU box (queries are independent;
            # the C walk releases the GIL)
            from concurrent.futures import ThreadPoolExecutor
            CORES = 16

            def one(qi):
                o_items, o_dist, _ = oracle.hnsw_search(int(m), stored, graph, pq[qi], k, ef)
                return bool(np.array_equal(o_items, ids[qi, :cnt[qi]]))

            t0 = time.time()
            with ThreadPoolExecutor(CORES) as ex:
                same = all(ex.map(one, range(n_cpu)))
            cpu_s = time.time() - t0
            cpu = {"value": n_cpu / cpu_s, "unit": "queries/s", "cores": CORES, "kind": "port",
                   "sample": f"the first {n_cpu} queries through the C restatement of HnswIndex.searchKnn on the same graph, {CORES} threads, "
                             f"{cpu_s:.2f} s; device results identical: {same}"}
        row_bytes = ((a.dim + 63) // 64 * 64) * 2
        print(json.dumps({
            "metric": "hnsw queries/sec", "value": a.queries / wall, "unit": "queries/s",
            "config": {"workload": f"{a.vectors} x d={a.dim} fp16 Cosine HNSW maxM={a.max_m} efConstruction={a.ef_construction}, "
                                   f"{a.queries} queries, k={k}, ef={ef}"},
            "ms_per_batch": wall * 1e3, "kernel_ms": st["kernel_ms"], "recall_at_k": recall,
            "distance_evals_per_query": st["distance_evals"] / a.queries, "expansions_per_query": st["expansions"] / a.queries,
            "admissions_per_query": st["admissions"] / a.queries, "largest_candidate_queue": st["largest_candidate_queue"],
            "distance_evals_per_sec": st["distance_evals"] / (st["kernel_ms"] * 1e-3),
            "roofline": {"bound": "hbm", "achieved": st["distance_evals"] * row_bytes / (st["kernel_ms"] * 1e-3) / 1e9, "peak": 8000.0,
                         "unit": "GB/s", "frac": st["distance_evals"] * row_bytes / (st["kernel_ms"] * 1e-3) / 8e12,
                         "note": "random 512-B row gathers; latency-bound walk"},
            "cpu_baseline": cpu,
            "spilled_queries": st["spilled_queries"], "build_s": build_s, "build": a.build, "build_threads": a.threads,
            "device_build_counters": dict(zip(("rounds", "unseen_additions", "queue_prunes", "dropped_candidates"), ix.build_stats())) if a.build == "gpu" else None}), flush=True)
    ix.close(); bf.close()


if __name__ == "__main__":
    main()
