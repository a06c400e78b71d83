"""Debug helper: run one golden case through the C ABI on the GPU in both modes and print stats."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
KAT = json.load(open(os.path.join(ROOT, "tests", "golden", "sann_kat.json")))
name = sys.argv[1] if len(sys.argv) > 1 else "age_window"
case = [c for c in KAT["sann"] if c["name"] == name][0]
lists = {int(c): [(t, s) for t, s in v] for c, v in case["lists"].items()}
c = case["config"]
cfg = pkg.SimClustersANNConfig(**{**c, "annAlgorithm": pkg.ScoringAlgorithm(c["annAlgorithm"])})
emb = case["emb"]
offs = np.array([0, len(emb)], np.int64)
cids = np.array([e[0] for e in emb], np.int32)
scs = np.array([e[1] for e in emb], np.float64)
for force in ("0", "1"):
    os.environ["SANN_FORCE_GENERAL"] = force
    for P in (1, 4):
        index = pkg.ClusterTweetIndex.from_map(lists, n_partitions=P)
        qb = pkg.QueryBatch(index, offs, cids, scs, cfg, now_ms=case["now_ms"], variant=pkg.Variant(case["variant"]))
        qb.run()
        qb.finish()
        ids, scores, counts, msz = qb.results()
        st = qb.stats()
        print("force_general", force, "P", P, "count", counts[0], "msz", msz[0], "fallback", st.n_fallback_units,
              "ids", ids[0, :counts[0]].tolist(), "scores", scores[0, :counts[0]].tolist())
        qb.close()
        index.close()
print("expect", case["expect"], "map_size", case["map_size"])
