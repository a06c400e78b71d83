"""The library's own RCCL exchange (csrc/sann_comm.hip) as far as a 1-GPU box can take it: a communicator of one
rank, outputs bound in owner chunks, sann_exchange_to_owners (a send to itself through RCCL), the owner's merge -- the
call sequence a non-Python worker makes per batch.  The N > 1 run is the driver's."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_exchange_and_owner_merge_with_one_rank(pkg):
    import torch

    lib = pkg.load_library()
    co = pkg.corpus.make_corpus(30000, 1500, seed=21, index_cap=400)
    offs, cids, scs = pkg.corpus.make_queries(24, 1500, seed=22, clusters_per_user=50)
    index = pkg.ClusterTweetIndex(co.cluster_ids, co.list_offsets, co.tweet_ids, co.scores, n_partitions=8)
    K = 200
    cfg = pkg.SimClustersANNConfig(maxNumResults=K, maxTopTweetsPerCluster=300)
    nq = len(offs) - 1

    uid = C.create_string_buffer(128)
    assert lib.sann_comm_unique_id(uid) == 0, lib.sann_last_error()
    assert any(uid.raw)
    comm = C.c_void_p()
    assert lib.sann_comm_create(0, 0, 1, uid, C.byref(comm)) == 0, lib.sann_last_error()
    r, w = C.c_int32(-1), C.c_int32(-1)
    assert lib.sann_comm_info(comm, C.byref(r), C.byref(w)) == 0 and (r.value, w.value) == (0, 1)

    chunk, o_sc, o_cnt, o_msz = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
    assert lib.sann_owner_message_layout(nq, K, C.byref(chunk), C.byref(o_sc), C.byref(o_cnt), C.byref(o_msz)) == 0
    send = torch.zeros(chunk.value, dtype=torch.uint8, device="cuda")
    recv = torch.full((chunk.value,), 0xEE, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    qb = pkg.QueryBatch(index, offs, cids, scs, cfg, now_ms=co.now_ms)
    sp = send.data_ptr()
    qb.bind_outputs_chunked(sp, sp + o_sc.value, sp + o_cnt.value, sp + o_msz.value, nq, chunk.value)
    stream = torch.cuda.Stream()
    qb.run(stream.cuda_stream)
    qb.finish(stream.cuda_stream)
    assert lib.sann_exchange_to_owners(comm, C.c_void_p(stream.cuda_stream), C.c_void_p(sp), C.c_void_p(recv.data_ptr()), chunk.value) == 0, lib.sann_last_error()
    out_ids = torch.zeros((nq, K), dtype=torch.int64, device="cuda")
    out_sc = torch.zeros((nq, K), dtype=torch.float64, device="cuda")
    out_cnt = torch.zeros(nq, dtype=torch.int32, device="cuda")
    out_msz = torch.zeros(nq, dtype=torch.int32, device="cuda")
    rp = recv.data_ptr()
    with torch.cuda.stream(stream):
        pass
    rc = lib.sann_merge_shards(0, C.c_void_p(stream.cuda_stream), 1, nq, K, chunk.value, rp, rp + o_sc.value, rp + o_cnt.value, rp + o_msz.value,
                               qb.device_k(), out_ids.data_ptr(), out_sc.data_ptr(), out_cnt.data_ptr(), out_msz.data_ptr())
    assert rc == 0, lib.sann_last_error()
    stream.synchronize()
    assert torch.equal(send, recv)  # the message arrived byte for byte
    qb.close()
    # the owner's merged answer is the unsharded answer
    ref = pkg.QueryBatch(index, offs, cids, scs, cfg, now_ms=co.now_ms)
    ref.run(); ref.finish()
    ids, scores, counts, msz = ref.results()
    ref.close()
    g_cnt = out_cnt.cpu().numpy()
    assert np.array_equal(g_cnt, counts) and np.array_equal(out_msz.cpu().numpy(), msz)
    g_ids, g_sc = out_ids.cpu().numpy(), out_sc.cpu().numpy()
    for q in range(nq):
        assert np.array_equal(g_ids[q, :counts[q]], ids[q, :counts[q]])
        assert np.array_equal(g_sc[q, :counts[q]].view(np.int64), scores[q, :counts[q]].view(np.int64))
    assert lib.sann_comm_destroy(comm) == 0
    index.close()


def test_exchange_argument_errors(pkg):
    lib = pkg.load_library()
    assert lib.sann_comm_create(0, 2, 2, C.create_string_buffer(128), C.byref(C.c_void_p())) == 1  # rank out of range
    assert lib.sann_exchange_to_owners(None, None, None, None, 8) == 1
    assert lib.sann_comm_unique_id(None) == 1


@pytest.mark.parametrize("sharding", ["tweet-hash", "cluster-range"])
def test_bench_sharded_modes_with_one_rank(sharding):
    """bench.py's N > 1 code paths with a single rank (--exercise-exchange): process group, the library's RCCL communicator,
    the exchange(s) and the owner's merge; the answers must equal the unsharded index's.  Runs in a child process (it
    initialises torch.distributed and redirects its stdout)."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29600 + os.getpid() % 300))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--exercise-exchange", "--sharding", sharding, "--tweets", "2000000",
                        "--queries-per-gpu", "128", "--steps", "3", "--warmup", "1", "--rotate", "2", "--no-cpu-baseline", "--e2e-steps", "0"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    if sharding == "cluster-range":
        assert line["sharded_equals_unsharded"] is True
    else:  # one tweet-hash shard IS the whole index: what went through the exchange and the owner's merge is held to the oracle
        assert line["recall_at_400_parity"] == 1.0 and line["parity_checked_queries"] >= 16
    assert line["config"]["sharding"] == (sharding if sharding == "cluster-range" else "none") and line["value"] > 0  # (one rank: "none")
    assert line["config"].get("queries_not_proven_by_cut_lists", 0) == 0
