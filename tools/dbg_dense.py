import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
pkg = load_package()
m = pkg.dense_ann.DistanceMetric.Cosine
n, d, nq, k = 6000, 64, 20, 25
rng = np.random.default_rng(4)
x = rng.standard_normal((n, d)).astype(np.float32)
q = rng.standard_normal((nq, d)).astype(np.float32)
full = pkg.dense_ann.BruteForceIndex.build(m, x)
st0 = full.stored_vectors().astype(np.float64)
res = [full.search(q, k) for _ in range(8)]
st = full.stored_vectors().astype(np.float64)
print("stored vectors changed:", not np.array_equal(st, st0))
qn = q / np.linalg.norm(q, axis=1, keepdims=True)
qn = qn.astype(np.float16).astype(np.float64)
dist = 1 - qn @ st.T
truth = np.argsort(dist, axis=1)[:, :k]
for run in (0, 3, 7):
    ids, dd, cnt = res[run]
    for qi in range(nq):
        if set(ids[qi].tolist()) != set(truth[qi].tolist()):
            extra = sorted(set(ids[qi].tolist()) - set(truth[qi].tolist()))
            miss = sorted(set(truth[qi].tolist()) - set(ids[qi].tolist()))
            print("run", run, "q", qi, "cnt", cnt[qi], "extra", [(e, round(dist[qi, e], 5)) for e in extra], "missing", [(e, round(dist[qi, e], 5)) for e in miss],
                  "kth", round(dist[qi, truth[qi, -1]], 5), "dups", k - len(set(ids[qi].tolist())))
