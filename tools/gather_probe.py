"""Measurement: the unit kernel's gather on its own (sann_debug_gather_probe) on the bench workload."""
import ctypes as C
import os
import sys

os.environ.setdefault("SANN_NO_TORCH", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
lib = pkg.load_library()
P = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
offs, cids, scs = pkg.corpus.make_queries(1024)
index = pkg.ClusterTweetIndex.synthetic(T, n_partitions=P)
cfg = pkg.SimClustersANNConfig(maxNumResults=400, annAlgorithm=pkg.ScoringAlgorithm.CosineSimilarity)
qb = pkg.QueryBatch(index, offs, cids, scs, cfg, now_ms=pkg.corpus.NOW_MS)
for _ in range(3):
    qb.run(); qb.finish()
st = qb.stats()
print("P", P, "postings", st.postings_scanned, "bytes", st.algorithmic_bytes, "max unit", st.max_unit_postings)
ms, cs = C.c_double(), C.c_uint64()
for mode, wgs in [(0, 1), (1, 6), (16, 1), (17, 1), (11, 1), (12, 1), (13, 1), (14, 1), (18, 1), (19, 1), (15, 1), (10, 1)]:
    rc = lib.sann_debug_gather_probe(qb._h, mode, wgs, 10, C.byref(ms), C.byref(cs))
    assert rc == 0, lib.sann_last_error()
    print(f"mode {mode} wgs/cu {wgs}: {ms.value*1e3:8.1f} us  {st.algorithmic_bytes/ms.value/1e9:7.2f} TB/s  checksum {cs.value:016x}")
