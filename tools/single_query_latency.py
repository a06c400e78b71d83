"""BASELINE configs[1] (C2): ONE query against the 1M-tweet (or any) corpus -- the latency of the boundary call
sann_get_tweet_candidates with nq = 1, host buffers in and out, and of the oracle's C restatement on one host core."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import _pkg  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tweets", type=int, default=1_000_000)
    ap.add_argument("--calls", type=int, default=300)
    a = ap.parse_args()
    pkg = _pkg.load_package()
    index = pkg.ClusterTweetIndex.synthetic(a.tweets)
    offs, cids, scs = pkg.corpus.make_queries(64)
    cfg = pkg.SimClustersANNConfig(maxNumResults=400)
    sa = pkg.simclusters_ann
    out = {"workload": f"single query, {a.tweets} tweets x 144428 clusters, top-400, N=50 M=800 cosine", "calls": a.calls}
    # pageable numpy outputs (the runtime's staged copies), then pinned outputs kept across calls (sann_host_alloc, what
    # INTEGRATION.md tells a shim to do: a lone small call's merge kernel then writes the answer straight into them)
    pinned = (sa.pinned_array((1, 400), np.int64), sa.pinned_array((1, 400), np.float64), sa.pinned_array((1,), np.int32), sa.pinned_array((1,), np.int32))
    for name, outs in (("pageable_outputs", None), ("pinned_outputs", pinned)):
        lat = []
        for i in range(a.calls + 20):
            q = i % 64
            o = np.array([0, offs[q + 1] - offs[q]], np.int64)
            t0 = time.perf_counter()
            sa.get_tweet_candidates(index, o, cids[offs[q]:offs[q + 1]], scs[offs[q]:offs[q + 1]], [cfg], now_ms=pkg.corpus.NOW_MS, out=outs)
            if i >= 20:
                lat.append(time.perf_counter() - t0)
        lat = np.array(lat) * 1e3
        out[name] = {"latency_ms_median": float(np.median(lat)), "latency_ms_p99": float(np.percentile(lat, 99)), "latency_ms_min": float(lat.min())}
    try:
        import oracle
        lists = index.export_lists(np.unique(cids[offs[0]:offs[8]]))
        t0 = time.perf_counter()
        for q in range(8):
            oracle.sann_query(cids[offs[q]:offs[q + 1]], scs[offs[q]:offs[q + 1]], None, cfg, pkg.corpus.NOW_MS, *lists)
        out["cpu_oracle_ms_per_query_one_core"] = (time.perf_counter() - t0) / 8 * 1e3
    except Exception as e:  # the CPU leg is a courtesy
        out["cpu_oracle_error"] = str(e)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
