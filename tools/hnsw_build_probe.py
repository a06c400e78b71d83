"""Recall of device-built vs host-built (sequential) graphs on the degenerate test's shape, by batch size."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from _pkg import load_package
pkg = load_package()
m = pkg.dense_ann.DistanceMetric.L2
rng = np.random.default_rng(1)
for n in (1, 5, 900, 3000):
    x = rng.standard_normal((n, 16)).astype(np.float32)
    if n < 900:
        continue
    nqs = 64
    def rec(ix):
        ids, _, cnt = ix.search(x[:nqs] + 1e-3, 1, 64)
        return float(np.mean(ids[:, 0] == np.arange(nqs)))
    h = pkg.hnsw_ann.Hnsw.build(m, x, max_m=4, ef_construction=16, seed=2)
    print(n, "host sequential", rec(h)); h.close()
    for b in (256, 64, 16, 1):
        g = pkg.hnsw_ann.Hnsw.build(m, x, max_m=4, ef_construction=16, seed=2, gpu=True, batch=b)
        print(n, "gpu batch", b, rec(g), g.build_stats()); g.close()
