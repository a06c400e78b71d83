import os, sys, time
os.environ.setdefault("SANN_NO_TORCH", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
T = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
t = time.time()
try:
    ix = pkg.ClusterTweetIndex.synthetic(T, n_partitions=32)
    i = ix.info()
    print("built", T, "tweets in", round(time.time() - t, 2), "s; postings", i.n_postings_total, "bytes", i.device_bytes, "maxlen", i.max_list_len)
except Exception as e:
    print("FAILED after", round(time.time() - t, 2), "s:", e)
