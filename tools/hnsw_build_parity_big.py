"""One-off evidence beyond tests/: the device HNSW builder against the oracle's restatement of the batched insertion on larger
graphs than the test-suite affords (the oracle is single-threaded C)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import _pkg  # noqa: E402
import oracle  # noqa: E402

pkg = _pkg.load_package()
for metric, n, d, max_m, efc, batch in (("Cosine", 100_000, 32, 16, 100, 4096), ("L2", 60_000, 96, 12, 200, 2048), ("InnerProduct", 200_000, 16, 8, 64, 4096)):
    m = getattr(pkg.dense_ann.DistanceMetric, metric)
    rng = np.random.default_rng(n)
    x = rng.standard_normal((n, d)).astype(np.float32)
    t0 = time.time()
    gpu = pkg.hnsw_ann.Hnsw.build(m, x, max_m=max_m, ef_construction=efc, seed=17, gpu=True, batch=batch)
    t_gpu = time.time() - t0
    g = gpu.graph()
    lv = np.zeros(n, np.int32)
    np.maximum.at(lv, g[1], g[0])
    lv[g[4]] = g[5]
    t0 = time.time()
    want = oracle.hnsw_build_batched(int(m), gpu.stored_vectors(), lv, max_m, efc, batch)
    t_cpu = time.time() - t0
    same = all(np.array_equal(a, b) for a, b in zip(g[:4], want[:4])) and g[4] == want[4] and g[5] == want[5]
    print(f"{metric} n={n} d={d} maxM={max_m} efC={efc} batch={batch}: device {t_gpu:.1f} s, oracle {t_cpu:.1f} s, entries {len(g[0])}, "
          f"neighbours {len(g[3])}, max level {g[5]}, build counters {gpu.build_stats()}, graphs identical: {same}", flush=True)
    gpu.close()
    assert same
