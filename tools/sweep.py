"""Robustness sweep: algorithms x k x M x minScore on a device-built corpus; timing, fallback counts
and a bit-exact spot check of a few queries against the oracle for every configuration."""
import os
import sys
import time

import numpy as np

os.environ.setdefault("SANN_NO_TORCH", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
oracle = ge.load_oracle()
lib = pkg.load_library()
T = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
nq = 512
index = pkg.ClusterTweetIndex.synthetic(T, n_partitions=32)
offs, cids, scs = pkg.corpus.make_queries(nq)
lists = index.export_lists(cids[:offs[4]])
now = pkg.corpus.NOW_MS
SA = pkg.ScoringAlgorithm
rows = []
for alg in (SA.CosineSimilarity, SA.LogCosineSimilarity, SA.DotProduct, SA.CosineSimilarityNoSourceEmbeddingNormalization):
    for k, M, ms, age in ((400, 800, 0.0, 24), (10, 800, 0.0, 24), (1000, 2000, 0.0, 24), (400, 200, 0.0, 24),
                          (400, 800, 0.05, 24), (400, 800, 0.0, 6), (400, 800, 0.0, 175200)):
        cfg = pkg.SimClustersANNConfig(maxNumResults=k, maxTopTweetsPerCluster=M, minScore=ms, maxTweetCandidateAgeHours=age,
                                       annAlgorithm=alg)
        qb = pkg.QueryBatch(index, offs, cids, scs, cfg, now_ms=now)
        qb.run(); qb.finish()
        t0 = time.perf_counter()
        for _ in range(5):
            qb.run(); qb.finish()
        assert lib.sann_device_synchronize(0) == 0
        dt = (time.perf_counter() - t0) / 5
        ids, sc, cnt, msz = qb.results()
        st = qb.stats()
        ok = True
        for q in range(4):
            o_i, o_s, o_m = oracle.sann_query(cids[offs[q]:offs[q + 1]], scs[offs[q]:offs[q + 1]], None, cfg, now, *lists)
            ok &= cnt[q] == len(o_i) and msz[q] == o_m and np.array_equal(ids[q, :cnt[q]], o_i) and \
                np.array_equal(sc[q, :cnt[q]].view(np.int64), o_s.view(np.int64))
        print(f"{alg.name[:18]:18s} k={k:4d} M={M:4d} minScore={ms:4.2f} age={age:6d}  {dt * 1e3:7.3f} ms/batch  "
              f"fallback/run={st.n_fallback_units / 6:7.1f} requeried/run={st.n_requeried / 6:5.1f} exact={ok}", flush=True)
        qb.close()
