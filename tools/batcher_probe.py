"""Micro-batching queue under native load (tools/micro/batcher_load.c): requests/s and latency against caller threads."""
import ctypes
import os
import sys

import numpy as np

os.environ.setdefault("SANN_NO_TORCH", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
lib = pkg.load_library()
tweets = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
offs, cids, scs = pkg.corpus.make_queries(1024)
index = pkg.ClusterTweetIndex.synthetic(tweets)
cfg = pkg.SimClustersANNConfig(maxNumResults=400)
cfg_c = cfg.to_c()
load = ctypes.CDLL(os.path.join(ROOT, "tools", "micro", "libbatcher_load.so"))
load.batcher_load_run.restype = ctypes.c_int
load.batcher_load_run.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p,
                                  ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_double)]
o, c, s = np.ascontiguousarray(offs, np.int64), np.ascontiguousarray(cids, np.int32), np.ascontiguousarray(scs, np.float64)
res = (ctypes.c_double * 6)()
for threads, window, batch, wait, disp in [(int(x) for x in a.split(":")) for a in (sys.argv[2:] or ["8:64:512:200:3", "8:128:1024:200:3"])]:
    mb = pkg.MicroBatcher(index, max_batch=batch, max_wait_us=wait, n_dispatchers=disp)
    for n_req in (2 * window, max(4 * window, int(os.environ.get("PROBE_REQUESTS", 131072)) // threads)):  # warm-up, then the measured run (requests per thread)
        rc = load.batcher_load_run(mb._h, threads, window, n_req, 1024, o.ctypes.data, c.ctypes.data, s.ctypes.data, ctypes.byref(cfg_c), pkg.corpus.NOW_MS, res)
        assert rc == 0, lib.sann_last_error()
    st = mb.stats()
    mb.close()
    print(f"threads {threads} x window {window} max_batch {batch} wait {wait} us dispatchers {disp}: {res[1] / res[0]:.0f} req/s, {res[2] / res[0]:.3e} candidates/s, "
          f"latency p50 {res[3]:.0f} p99 {res[4]:.0f} max {res[5]:.0f} us, mean batch {st.n_requests / max(st.n_batches, 1):.0f}, "
          f"full {st.n_closed_full} deadline {st.n_closed_by_deadline}", flush=True)
