"""The batched boundary call under 1 / 2 / 4 / 8 native caller threads, with the per-stage wall times of SANN_TRACE_CALLS."""
import ctypes
import os
import sys

import numpy as np

os.environ.setdefault("SANN_NO_TORCH", "1")
os.environ["SANN_TRACE_CALLS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
lib = pkg.load_library()
tweets = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
n_sets, nq = 8, int(os.environ.get("PROBE_NQ", 1024))  # queries per call
o_all, c_all, s_all = pkg.corpus.make_queries(nq * n_sets)
keep = []
for i in range(n_sets):
    lo, hi = o_all[i * nq], o_all[(i + 1) * nq]
    keep.append((np.ascontiguousarray(o_all[i * nq:(i + 1) * nq + 1] - lo, np.int64), np.ascontiguousarray(c_all[lo:hi], np.int32),
                 np.ascontiguousarray(s_all[lo:hi], np.float64)))
index = pkg.ClusterTweetIndex.synthetic(tweets)
cfg_c = pkg.SimClustersANNConfig(maxNumResults=400).to_c()
ld = ctypes.CDLL(os.path.join(ROOT, "tools", "micro", "libbatcher_load.so"))
PP = ctypes.c_void_p * n_sets
a_o, a_c, a_s = PP(*[k[0].ctypes.data for k in keep]), PP(*[k[1].ctypes.data for k in keep]), PP(*[k[2].ctypes.data for k in keep])
ld.e2e_load_run.restype = ctypes.c_int
ld.e2e_load_run.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                            ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_double)]
r3, us4, calls = (ctypes.c_double * 3)(), (ctypes.c_double * 4)(), ctypes.c_int64()
for thr in [int(a) for a in (sys.argv[2:] or ["1", "2", "4", "8"])]:
    for n_calls in (2 * thr, 64):
        lib.sann_debug_call_trace(us4, ctypes.byref(calls))
        assert ld.e2e_load_run(index.handle, thr, n_calls, nq, n_sets, a_o, a_c, a_s, ctypes.byref(cfg_c), pkg.corpus.NOW_MS, r3) == 0, lib.sann_last_error()
    lib.sann_debug_call_trace(us4, ctypes.byref(calls))
    n = max(calls.value, 1)
    print(f"{thr} callers: {r3[0] / r3[1] * 1e3:.3f} ms per call overall; per call: reset+H2D+prep launch{' + kernel launches' if os.environ.get('SANN_ENGINE') != '0' else ''} {us4[0] / n:.0f} us, "
          f"{'enqueue copies' if os.environ.get('SANN_ENGINE') != '0' else 'kernel launches'} {us4[1] / n:.0f} us, "
          f"wait kernels {us4[2] / n:.0f} us, {'wait copies' if os.environ.get('SANN_ENGINE') != '0' else 'copy answer + wait'} {us4[3] / n:.0f} us", flush=True)
