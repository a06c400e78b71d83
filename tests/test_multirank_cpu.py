"""N > 1 path on CPU: two gloo ranks, each owning one tweet-hash shard of a small corpus and half of the queries.
Every rank answers the whole batch on its shard (here with the oracle standing in for the GPU kernels) with lists
cut at `sharding.shard_list_length(k, world)`, packs one message per owner with the layout bench.py binds the merge
kernel to, the messages are exchanged (all-gather + slice: gloo has no all-to-all; bench.py's rehearsal path does
the same), and each owner merges the lists of its queries with the numpy restatement of `sann_merge_shards_cut`:
every answer must be proven by the cut lists and equal the unsharded oracle bit for bit -- i.e. tweet-hash sharding
+ ComposedQueryable-style merge is exact.  The shard of a tweet comes from the library's own
sann_tweet_shard (a host function; no GPU needed)."""
import dataclasses
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch
    import torch.distributed as dist
    from _pkg import load_package
    import oracle

    pkg = load_package()
    lib = pkg.load_library()
    sh_ = pkg.sharding
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        co = pkg.corpus.make_corpus(20000, 800, seed=5, index_cap=300)
        nq, k, M = 12, 120, 200
        nql = nq // world
        offs, cids, scs = pkg.corpus.make_queries(nq, 800, seed=6, clusters_per_user=30)
        cfg = pkg.SimClustersANNConfig(maxNumResults=k, maxTopTweetsPerCluster=M, maxScanClusters=30)
        shard_k = sh_.shard_list_length(k, world)
        assert shard_k < k  # the lists really are cut
        cfg_shard = dataclasses.replace(cfg, maxNumResults=shard_k)
        # this rank's shard: the first M postings of every list (global ranks!), then my tweets only
        t_l, s_l, off = [], [], [0]
        for i in range(len(co.cluster_ids)):
            b, e = co.list_offsets[i], min(co.list_offsets[i + 1], co.list_offsets[i] + M)
            t, s = co.tweet_ids[b:e], co.scores[b:e]
            mine = np.array([lib.sann_tweet_shard(int(x), world) == rank for x in t], bool)
            t_l.append(t[mine]); s_l.append(s[mine]); off.append(off[-1] + int(mine.sum()))
        sh = (co.cluster_ids, np.array(off, np.int64), np.concatenate(t_l), np.concatenate(s_l))
        ids = np.zeros((nq, shard_k), np.int64); sc = np.zeros((nq, shard_k)); cnt = np.zeros(nq, np.int32); msz = np.zeros(nq, np.int32)
        for q in range(nq):
            i, s, m = oracle.sann_query(cids[offs[q]:offs[q + 1]], scs[offs[q]:offs[q + 1]], None, cfg_shard, co.now_ms, *sh)
            ids[q, :len(i)] = i; sc[q, :len(i)] = s; cnt[q] = len(i); msz[q] = m
        send = torch.from_numpy(sh_.pack_for_owners(ids, sc, cnt, msz, world))
        parts = [torch.zeros_like(send) for _ in range(world)]
        dist.all_gather(parts, send)
        size = send.numel() // world
        recv = torch.cat([p[rank * size:(rank + 1) * size] for p in parts]).numpy()  # message r of every shard
        g_ids, g_sc, g_cnt, g_msz = sh_.unpack_from_shards(recv, world, nql, shard_k)
        ok = True
        for ql in range(nql):
            q = rank * nql + ql
            m_ids, m_sc, proven = sh_.merge_cut_lists(g_ids[:, ql], g_sc[:, ql], g_cnt[:, ql], k, shard_k)
            o_i, o_s, o_m = oracle.sann_query(cids[offs[q]:offs[q + 1]], scs[offs[q]:offs[q + 1]], None, cfg, co.now_ms,
                                              co.cluster_ids, co.list_offsets, co.tweet_ids, co.scores)
            ok &= proven and m_ids.tolist() == o_i.tolist() and m_sc.tolist() == o_s.tolist() and int(g_msz[:, ql].sum()) == o_m
        out[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_shard_merge_is_exact():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert out.get(0) is True and out.get(1) is True


def _range_worker(rank, world, port, out):
    """The cluster-id-range deployment's message flow between two gloo ranks: rank g holds WHOLE lists of the clusters of its id
    range; per batch it counts the scanned prefixes by destination (sann_tweet_shard), the counts cross, the (id, score) records
    cross packed destination-major exactly as sann_index_export_prefixes_device lays them out (clusters ascending inside a
    destination, ranks ascending inside a cluster), the receiver turns source-major arrival order into its temporary CSR (which
    is what sharding.RangeShard.build_received hands the library), the oracle answers the batch on it, and the owners merge."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch
    import torch.distributed as dist
    from _pkg import load_package
    import oracle

    pkg = load_package()
    lib = pkg.load_library()
    sh_ = pkg.sharding
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def all_to_all_bytes(blocks):
        """blocks[r] = bytes for rank r (any lengths): what arrives, by source.  gloo: all-gather of everything, then slice."""
        lens = torch.tensor([len(b) for b in blocks], dtype=torch.int64)
        all_lens = [torch.zeros_like(lens) for _ in range(world)]
        dist.all_gather(all_lens, lens)
        width = int(max(int(l.sum()) for l in all_lens))
        mine = torch.zeros(max(width, 1), dtype=torch.uint8)
        flat = np.concatenate([np.frombuffer(b, np.uint8) for b in blocks]) if width else np.zeros(0, np.uint8)
        mine[:len(flat)] = torch.from_numpy(flat.copy())
        parts = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        got = []
        for s in range(world):
            o = int(all_lens[s][:rank].sum())
            got.append(parts[s][o:o + int(all_lens[s][rank])].numpy().tobytes())
        return got

    try:
        co = pkg.corpus.make_corpus(20000, 800, seed=5, index_cap=300)
        nq, k, M = 12, 120, 200
        nql = nq // world
        offs, cids, scs = pkg.corpus.make_queries(nq, 800, seed=6, clusters_per_user=30)
        cfg = pkg.SimClustersANNConfig(maxNumResults=k, maxTopTweetsPerCluster=M, maxScanClusters=30)
        shard_k = sh_.shard_list_length(k, world)
        cfg_shard = dataclasses.replace(cfg, maxNumResults=shard_k)
        bounds = sh_.cluster_range_bounds(np.diff(co.list_offsets), world)  # list positions [bounds[g], bounds[g+1]) are rank g's
        lo, hi = int(bounds[rank]), int(bounds[rank + 1])
        my_clusters = co.cluster_ids[lo:hi]
        need = sh_.scanned_clusters(offs, cids, scs, cfg.maxScanClusters)
        mine = np.isin(need, my_clusters)
        # 1. counts[cluster][dest] of my range's scanned prefixes, and the records, destination-major
        counts = np.zeros((len(need), world), np.int32)
        rec = [[] for _ in range(world)]
        rec_t = np.dtype([("id", "<i8"), ("score", "<f8")])
        for ci in np.nonzero(mine)[0]:
            li = int(np.searchsorted(co.cluster_ids, need[ci]))
            b, e = co.list_offsets[li], min(co.list_offsets[li + 1], co.list_offsets[li] + M)
            dest = np.array([lib.sann_tweet_shard(int(x), world) for x in co.tweet_ids[b:e]], np.int64)
            for d in range(world):
                sel = dest == d
                counts[ci, d] = int(sel.sum())
                r = np.zeros(int(sel.sum()), rec_t)
                r["id"], r["score"] = co.tweet_ids[b:e][sel], co.scores[b:e][sel]
                rec[d].append(r)
        # 2. the counts cross (block r of my message = counts[:, r]) ...
        got_counts = all_to_all_bytes([np.ascontiguousarray(counts[:, d]).tobytes() for d in range(world)])
        from_src = np.stack([np.frombuffer(g, np.int32) for g in got_counts]).astype(np.int64)  # [source][cluster]
        assert ((from_src > 0).sum(axis=0) <= 1).all()  # a cluster has ONE source
        # 3. ... then the postings
        got = all_to_all_bytes([(np.concatenate(r) if r else np.zeros(0, rec_t)).tobytes() for r in rec])
        for s in range(world):
            assert len(got[s]) == 16 * int(from_src[s].sum())
        arrived = np.concatenate([np.frombuffer(g, rec_t) for g in got])
        # 4. source-major arrival order IS ascending cluster order (ranges ascend with the rank): the temporary CSR
        to_me = from_src.sum(axis=0)
        keep = to_me > 0
        t_cl = need[keep]
        t_off = np.concatenate([[0], np.cumsum(to_me[keep])]).astype(np.int64)
        assert t_off[-1] == len(arrived)
        for i, c in enumerate(t_cl):  # every list: my tweets of the global top-M prefix, in rank order
            li = int(np.searchsorted(co.cluster_ids, c))
            b, e = co.list_offsets[li], min(co.list_offsets[li + 1], co.list_offsets[li] + M)
            want = [int(x) for x in co.tweet_ids[b:e] if lib.sann_tweet_shard(int(x), world) == rank]
            assert arrived["id"][t_off[i]:t_off[i + 1]].tolist() == want
        sh = (t_cl, t_off, arrived["id"].copy(), arrived["score"].copy())
        # 5. the ordinary batch on the temporary index, then the owners' exchange and proving merge (as the tweet-hash test)
        ids = np.zeros((nq, shard_k), np.int64); sc = np.zeros((nq, shard_k)); cnt = np.zeros(nq, np.int32); msz = np.zeros(nq, np.int32)
        for q in range(nq):
            i, s, m = oracle.sann_query(cids[offs[q]:offs[q + 1]], scs[offs[q]:offs[q + 1]], None, cfg_shard, co.now_ms, *sh)
            ids[q, :len(i)] = i; sc[q, :len(i)] = s; cnt[q] = len(i); msz[q] = m
        send = torch.from_numpy(sh_.pack_for_owners(ids, sc, cnt, msz, world))
        parts = [torch.zeros_like(send) for _ in range(world)]
        dist.all_gather(parts, send)
        size = send.numel() // world
        recv = torch.cat([p[rank * size:(rank + 1) * size] for p in parts]).numpy()
        g_ids, g_sc, g_cnt, g_msz = sh_.unpack_from_shards(recv, world, nql, shard_k)
        ok = True
        for ql in range(nql):
            q = rank * nql + ql
            m_ids, m_sc, proven = sh_.merge_cut_lists(g_ids[:, ql], g_sc[:, ql], g_cnt[:, ql], k, shard_k)
            o_i, o_s, o_m = oracle.sann_query(cids[offs[q]:offs[q + 1]], scs[offs[q]:offs[q + 1]], None, cfg, co.now_ms,
                                              co.cluster_ids, co.list_offsets, co.tweet_ids, co.scores)
            ok &= proven and m_ids.tolist() == o_i.tolist() and m_sc.tolist() == o_s.tolist() and int(g_msz[:, ql].sum()) == o_m
        out[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_cluster_range_exchange_is_exact():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    port = 31700 + os.getpid() % 2000
    procs = [ctx.Process(target=_range_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert out.get(0) is True and out.get(1) is True


def test_shard_and_partition_hash_are_stable(pkg):
    lib = pkg.load_library()
    ids = [0, 1, -1, 2**40 + 12345, 1724188722590646272]
    assert [lib.sann_tweet_shard(i, 1) for i in ids] == [0] * 5
    s8 = [lib.sann_tweet_shard(i, 8) for i in ids]
    p32 = [lib.sann_tweet_partition(i, 32) for i in ids]
    assert all(0 <= x < 8 for x in s8) and all(0 <= x < 32 for x in p32)
    # pinned values: changing the hash silently would re-shard every deployed index
    assert (s8, p32) == ([0, 4, 2, 7, 2], [0, 12, 1, 10, 27])
    rng = np.random.default_rng(0)
    xs = rng.integers(0, 2**62, 4000)
    counts = np.bincount([lib.sann_tweet_shard(int(x), 8) for x in xs], minlength=8)
    assert counts.min() > 400 and counts.max() < 600


def test_shard_list_length_and_owner_messages(pkg):
    sh = pkg.sharding
    assert [sh.shard_list_length(400, n) for n in (1, 2, 4, 8)] == [400, 272, 160, 104]
    assert sh.shard_list_length(10, 8) == 10 and sh.shard_list_length(1000, 8) == 200
    rng = np.random.default_rng(3)
    world, nql, stride = 3, 4, 5
    nq = world * nql
    ids = rng.integers(-2**62, 2**62, (nq, stride))
    sc = rng.normal(size=(nq, stride))
    cnt = rng.integers(0, stride + 1, nq).astype(np.int32)
    msz = rng.integers(0, 1000, nq).astype(np.int32)
    size, offsets = sh.owner_message_layout(nql, stride)
    assert size % 8 == 0 and offsets == (0, nql * stride * 8, 2 * nql * stride * 8, 2 * nql * stride * 8 + 4 * nql)
    sent = [sh.pack_for_owners(ids + s, sc, cnt, msz + s, world) for s in range(world)]  # three "shards"
    for r in range(world):  # owner r receives message r of every shard
        recv = np.concatenate([b[r * size:(r + 1) * size] for b in sent])
        g_ids, g_sc, g_cnt, g_msz = sh.unpack_from_shards(recv, world, nql, stride)
        for s in range(world):
            assert np.array_equal(g_ids[s], ids[r * nql:(r + 1) * nql] + s) and np.array_equal(g_sc[s], sc[r * nql:(r + 1) * nql])
            assert np.array_equal(g_cnt[s], cnt[r * nql:(r + 1) * nql]) and np.array_equal(g_msz[s], msz[r * nql:(r + 1) * nql] + s)


def test_cut_list_merge_proof_rule(pkg):
    """The owner's proof: a list that arrived full may hide candidates below its last entry."""
    merge = pkg.sharding.merge_cut_lists
    ids = np.array([[1, 2, 3], [4, 5, 6]], np.int64)
    sc = np.array([[9.0, 8.0, 7.0], [6.0, 5.0, 4.0]])
    # shard 0 arrived full (3 of shard_k = 3) and its last entry (7.0) ranks above the merged 4th (6.0): unproven
    assert merge(ids, sc, np.array([3, 3]), 4, 3)[2] is False
    # k = 3: the merged 3rd entry IS shard 0's last one; shard 1's last entry lies below it: proven
    got = merge(ids, sc, np.array([3, 3]), 3, 3)
    assert got[0].tolist() == [1, 2, 3] and got[1].tolist() == [9.0, 8.0, 7.0] and got[2] is True
    # fewer merged entries than k and a full list: unproven; no full list: proven
    assert merge(ids, sc, np.array([3, 1]), 10, 3)[2] is False
    assert merge(ids, sc, np.array([2, 1]), 10, 3)[2] is True
    # equal scores: the smaller tweet id ranks first
    tie = merge(np.array([[7], [5]], np.int64), np.array([[1.0], [1.0]]), np.array([1, 1]), 1, 4)
    assert tie[0].tolist() == [5] and tie[2] is True


def test_c_message_layout_is_the_layout_the_exchange_tests_use(pkg):
    """sann_owner_message_layout (what a non-Python worker calls to bind its outputs and size the exchange) against the
    numpy layout the 2-rank gloo test above packs and merges with."""
    import ctypes as C
    lib = pkg.load_library()
    for nql, stride in ((1, 1), (3, 7), (512, 104), (1024, 400), (1000, 1000)):
        chunk, o_sc, o_cnt, o_msz = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        assert lib.sann_owner_message_layout(nql, stride, C.byref(chunk), C.byref(o_sc), C.byref(o_cnt), C.byref(o_msz)) == 0
        size, (p_ids, p_sc, p_cnt, p_msz) = pkg.sharding.owner_message_layout(nql, stride)
        assert (chunk.value, o_sc.value, o_cnt.value, o_msz.value) == (size, p_sc, p_cnt, p_msz) and p_ids == 0
        assert chunk.value % 8 == 0
    assert lib.sann_owner_message_layout(-1, 4, None, None, None, None) == 1  # SANN_EINVAL
