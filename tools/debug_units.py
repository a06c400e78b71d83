"""Debug: per-unit arrays of one batch (the P = 16 per-query-config test shape) under the library SANN_LIB_PATH names."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
lib = pkg.load_library()
co = pkg.corpus.make_corpus(30000, 1500, seed=21, index_cap=400)
offs, cids, scs = pkg.corpus.make_queries(24, 1500, seed=22, clusters_per_user=50)
index = pkg.ClusterTweetIndex(co.cluster_ids, co.list_offsets, co.tweet_ids, co.scores, n_partitions=16)
rng = np.random.default_rng(5)
nq = 24
cfgs, sources = [], []
for q in range(nq):
    cfgs.append(pkg.SimClustersANNConfig(
        maxNumResults=int(rng.choice([1, 10, 400, 1000, 5000])), minScore=float(rng.choice([0.0, 0.05, 0.3])),
        maxTopTweetsPerCluster=int(rng.choice([1, 50, 400, 10000])), maxScanClusters=int(rng.choice([1, 5, 50, 200])),
        maxTweetCandidateAgeHours=int(rng.choice([6, 12, 24, 175200])), minTweetCandidateAgeHours=int(rng.choice([0, 1, 3])),
        annAlgorithm=pkg.ScoringAlgorithm(int(rng.integers(1, 5)))))
    sources.append(int(co.tweet_ids[rng.integers(0, len(co.tweet_ids))]) if q % 2 else None)
src = np.array([0 if s is None else s for s in sources], np.int64)
has = np.array([0 if s is None else 1 for s in sources], np.uint8)
out = {}
for variant in (0, 1, 2):
    qb = pkg.QueryBatch(index, offs, cids, scs, cfgs, now_ms=co.now_ms, variant=pkg.Variant(variant), source_tweet_ids=src, has_source_tweet=has)
    qb.run(); qb.finish()
    n = nq * 16
    uu, cc, ff, tt = (np.zeros(n, np.int32) for _ in range(4))
    assert lib.sann_debug_unit_arrays(qb._h, uu.ctypes.data_as(C.c_void_p), cc.ctypes.data_as(C.c_void_p), ff.ctypes.data_as(C.c_void_p), tt.ctypes.data_as(C.c_void_p)) == 0
    out[variant] = (uu, cc, ff, tt, qb.results()[3].copy())
    qb.close()
np.savez(sys.argv[1], **{f"v{v}_{k}": a for v, arrs in out.items() for k, a in zip("uctfm", [arrs[0], arrs[1], arrs[3], arrs[2], arrs[4]])})
print("saved", sys.argv[1])
