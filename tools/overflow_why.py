"""Debug: why do fast units overflow / which queries are inexact, on the bench workload."""
import ctypes as C
import os
import sys

os.environ.setdefault("SANN_NO_TORCH", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
lib = pkg.load_library()
T = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
P = int(sys.argv[2]) if len(sys.argv) > 2 else 32
index = pkg.ClusterTweetIndex.synthetic(T, n_partitions=P)
offs, cids, scs = pkg.corpus.make_queries(1024)
cfg = pkg.SimClustersANNConfig(maxNumResults=400, annAlgorithm=pkg.ScoringAlgorithm.CosineSimilarity)
qb = pkg.QueryBatch(index, offs, cids, scs, cfg, now_ms=pkg.corpus.NOW_MS)
qb.run()
assert lib.sann_device_synchronize(0) == 0
cnt = (C.c_int32 * 8)()
ninx = C.c_int32()
assert lib.sann_debug_overflow_reasons(qb._h, cnt, C.byref(ninx)) == 0
print("T", T, "P", P, "overflow reasons [_, nscan, postings, multi, range/clash, ties]:", list(cnt), "inexact queries:", ninx.value)
qb.finish()
st = qb.stats()
print("postings scanned", st.postings_scanned, "fallback units", st.n_fallback_units, "requeried", st.n_requeried)
