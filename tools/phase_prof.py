"""Debug: average shader clocks per phase of the fast unit kernel on the bench workload."""
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
ALG = {"cosine": "CosineSimilarity", "logcosine": "LogCosineSimilarity", "dot": "DotProduct"}[sys.argv[3] if len(sys.argv) > 3 else "cosine"]
N = int(sys.argv[4]) if len(sys.argv) > 4 else 1      # shard 0 of N (1024 N queries), as tools/shard_cost.py
K = int(sys.argv[5]) if len(sys.argv) > 5 else 400    # per-shard list length
offs, cids, scs = pkg.corpus.make_queries(1024 * N)
index = pkg.ClusterTweetIndex.synthetic(T, n_partitions=P, shard_id=0, n_shards=N)
cfg = pkg.SimClustersANNConfig(maxNumResults=K, annAlgorithm=getattr(pkg.ScoringAlgorithm, ALG))
qb = pkg.QueryBatch(index, offs, cids, scs, cfg, now_ms=pkg.corpus.NOW_MS)
for _ in range(3):
    qb.run(); qb.finish()
assert lib.sann_debug_phase_cycles(qb._h, 1, None) == 0
qb.run(); qb.finish()
avg = (C.c_double * 16)()
assert lib.sann_debug_phase_cycles(qb._h, 0, avg) == 0
names = ["total", "desc+scan", "issue loads", "wait loads+filter+bloom", "dup resolve", "approx+minmax", "threshold",
         "compact", "exact+emit"]
print("P", P, "units counted", int(avg[15]))
for i, n in enumerate(names):
    print(f"  {n:28s} {avg[i]:10.0f} clk")
for i, n in enumerate(["offsets", "stage", "radix cut", "compact", "sort", "write+proof"]):
    print(f"  merge: {n:21s} {avg[9 + i]:10.0f} clk")
