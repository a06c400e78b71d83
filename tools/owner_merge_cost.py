"""Owner-side merge of an N-GPU run, measured on one GPU: 1024 queries x N per-shard lists.
Times sann_merge_shards on full-length lists (k entries each) against sann_merge_shards_cut on lists cut at
shard_k = k/N + 6 sigma + 8.  Synthetic sorted lists (distinct ids, random scores).  usage: owner_merge_cost.py N"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
lib = pkg.load_library()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
K, nq = 400, 1024
share = K / N
shard_k = min(K, int(-(-(share + 6.0 * (share * (1.0 - 1.0 / N)) ** 0.5 + 8.0) // 8) * 8))
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(3)


def lists(L):
    sc = torch.rand((N, nq, L), dtype=torch.float64, device=dev, generator=g).sort(dim=2, descending=True).values
    ids = (torch.arange(N * nq * L, dtype=torch.int64, device=dev).reshape(N, nq, L) * 7919) % (1 << 40)
    cnt = torch.full((N, nq), L, dtype=torch.int32, device=dev)
    return ids.contiguous(), sc.contiguous(), cnt, cnt.clone()


def timed(fn):
    for _ in range(5):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / 50 * 1e3


o_ids = torch.zeros((nq, K), dtype=torch.int64, device=dev)
o_sc = torch.zeros((nq, K), dtype=torch.float64, device=dev)
o_cnt = torch.zeros(nq, dtype=torch.int32, device=dev)
o_msz = torch.zeros(nq, dtype=torch.int32, device=dev)
bad = torch.zeros(1, dtype=torch.int32, device=dev)
d_k = torch.full((nq,), K, dtype=torch.int32, device=dev)
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
f_ids, f_sc, f_cnt, f_msz = lists(K)
c_ids, c_sc, c_cnt, c_msz = lists(shard_k)


def full():
    assert lib.sann_merge_shards(0, stream, N, nq, K, 0, f_ids.data_ptr(), f_sc.data_ptr(), f_cnt.data_ptr(), f_msz.data_ptr(),
                                 d_k.data_ptr(), o_ids.data_ptr(), o_sc.data_ptr(), o_cnt.data_ptr(), o_msz.data_ptr()) == 0


def cut():
    assert lib.sann_merge_shards_cut(0, stream, N, nq, shard_k, 0, shard_k, K, K, c_ids.data_ptr(), c_sc.data_ptr(),
                                     c_cnt.data_ptr(), c_msz.data_ptr(), o_ids.data_ptr(), o_sc.data_ptr(), o_cnt.data_ptr(),
                                     o_msz.data_ptr(), bad.data_ptr()) == 0


print(f"N={N}: owner merge of {nq} queries, full lists ({N}x{K}) {timed(full):.0f} us, cut lists ({N}x{shard_k}) {timed(cut):.0f} us, "
      f"exchange bytes/GPU {nq * N * K * 16 * (N - 1) // N} -> {nq * N * shard_k * 16 * (N - 1) // N}")
