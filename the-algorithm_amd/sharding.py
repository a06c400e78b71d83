"""Multi-GPU plumbing shared by bench.py and the tests: the list length a shard delivers,
the packed message a shard sends to the owner of a block of queries, and a numpy restatement of the owner's merge
with its exactness proof (what `sann_merge_shards_cut` does on the device) for the CPU tests.

Message to one owner (`nql` = queries per owner, `stride` = entries per list), in bytes:
    [0, nql*stride*8)                  tweet ids          int64[nql][stride]
    [nql*stride*8, 2*nql*stride*8)     scores             fp64 bit patterns [nql][stride]
    then int32 counts[nql], then int32 map sizes[nql]
A shard's send buffer is `world` such messages back to back (message r = the queries rank r owns); after the
all-to-all the owner holds `world` messages, one per shard, and merges them in place
(`shard_pitch_bytes` = the message size).  On the device the merge kernel writes the send buffer directly
(`sann_batch_bind_outputs_chunked`); `pack_for_owners` is the same layout from host arrays.
"""
from __future__ import annotations

import math
from typing import Tuple

import numpy as np


def shard_list_length(k: int, world: int) -> int:
    """Entries a shard delivers per query: its share of the final top-k is Binomial(k, 1/world), so
    k/world + 6 sigma + 8 (rounded up to 8, at most k) is enough except with vanishing probability -- and the
    owner's merge proves every answer, so "vanishing" is checked, not trusted."""
    if world <= 1:
        return k
    share = k / world
    need = share + 6.0 * math.sqrt(share * (1.0 - 1.0 / world)) + 8.0
    return min(k, int(-(-need // 8) * 8))


def owner_message_layout(nql: int, stride: int) -> Tuple[int, Tuple[int, int, int, int]]:
    """(message bytes, byte offsets of ids / scores / counts / map sizes inside a message)."""
    arr = nql * stride * 8
    return 2 * arr + 8 * nql, (0, arr, 2 * arr, 2 * arr + 4 * nql)


def pack_for_owners(ids: np.ndarray, scores: np.ndarray, counts: np.ndarray, map_sizes: np.ndarray, world: int) -> np.ndarray:
    """Host version of what the merge kernel writes: results of all `world * nql` queries -> uint8[world * message]."""
    nq, stride = ids.shape
    nql = nq // world
    size, (o_ids, o_sc, o_cnt, o_msz) = owner_message_layout(nql, stride)
    buf = np.zeros(world * size, np.uint8)
    for r in range(world):
        m = buf[r * size:(r + 1) * size]
        q0, q1 = r * nql, (r + 1) * nql
        m[o_ids:o_sc].view(np.int64)[:] = ids[q0:q1].reshape(-1)
        m[o_sc:o_cnt].view(np.int64)[:] = np.ascontiguousarray(scores[q0:q1]).reshape(-1).view(np.int64)
        m[o_cnt:o_msz].view(np.int32)[:] = counts[q0:q1]
        m[o_msz:size].view(np.int32)[:] = map_sizes[q0:q1]
    return buf


def unpack_from_shards(buf: np.ndarray, world: int, nql: int, stride: int):
    """Received buffer (one message per shard) -> ids [world][nql][stride], scores, counts [world][nql], map sizes."""
    size, (o_ids, o_sc, o_cnt, o_msz) = owner_message_layout(nql, stride)
    ids = np.zeros((world, nql, stride), np.int64)
    scores = np.zeros((world, nql, stride), np.float64)
    counts = np.zeros((world, nql), np.int32)
    msz = np.zeros((world, nql), np.int32)
    for s in range(world):
        m = buf[s * size:(s + 1) * size]
        ids[s] = m[o_ids:o_sc].view(np.int64).reshape(nql, stride)
        scores[s] = m[o_sc:o_cnt].view(np.float64).reshape(nql, stride)
        counts[s] = m[o_cnt:o_msz].view(np.int32)
        msz[s] = m[o_msz:size].view(np.int32)
    return ids, scores, counts, msz


def merge_cut_lists(ids: np.ndarray, scores: np.ndarray, counts: np.ndarray, k: int, shard_k: int):
    """One query: per-shard lists (each sorted, cut at shard_k) -> (ids, scores, proven).  Order: score descending,
    tweet id ascending.  `proven` is the device's rule: a list that arrived full (count >= shard_k) may hide
    candidates, all below its last entry; the merged top-k is exact iff no such last entry ranks above the merged
    k-th entry (and, with fewer than k merged entries, iff no list arrived full)."""
    ent = sorted(((-float(scores[s, j]), int(ids[s, j])) for s in range(ids.shape[0]) for j in range(counts[s])))
    kth = ent[k - 1] if len(ent) >= k and k > 0 else None
    proven = True
    for s in range(ids.shape[0]):
        c = int(counts[s])
        if k > 0 and c >= shard_k:
            last = (-float(scores[s, c - 1]), int(ids[s, c - 1]))
            if kth is None or last < kth:
                proven = False
    top = ent[:k]
    return np.array([e[1] for e in top], np.int64), np.array([-e[0] for e in top], np.float64), proven


# ---------------------------------------------------------------------------------------------------------------------
# The partitioning north_star names: CLUSTER-ID RANGES, with an exact merge.
#
# GPU g holds the FULL posting lists of the clusters in its id range; a query's <= 50 clusters scatter over the GPUs, so a
# candidate's score is a sum whose terms live on different GPUs.  A per-shard top-k before the terms meet is lossy
# (tools/cluster_range_loss.py measures how much); exchanging per-shard partial SUMS is exact in the set but not bit for
# bit (fp64 addition is not associative across shard boundaries).  What is exact to the last bit is to bring the TERMS
# together before anything is added: every shard sends, for the clusters the batch scans, the top-M prefix of each of its
# lists to the GPU that hash(tweetId) % N names -- postings are query independent, so a list scanned by many queries
# travels once -- and the receiving GPU, which now holds all postings of its tweets for this batch, runs the ordinary
# pipeline (same kernels, same accumulation order) on that temporary tweet-hash shard; the per-shard answers are merged
# exactly as in the tweet-hash deployment (ComposedQueryable, ann/.../common/ShardApi.scala:71-87).
#
# RangeShard / ClusterRangeDeployment below drive the library's entry points for it (include/simclusters_ann.h:
# sann_index_export_prefix_counts, sann_index_export_prefixes_device, sann_exchange_postings_by_tweet_hash,
# sann_index_build_from_device_postings): everything that touches postings runs on the device.  It is the reference point beside
# the tweet-hash deployment, not a competitor: it re-partitions ~N x M x 16 B per distinct scanned cluster on EVERY batch.
# ---------------------------------------------------------------------------------------------------------------------
def score_key(scores: np.ndarray) -> np.ndarray:
    """Monotone map double -> uint64 in java.lang.Double.compare order (csrc/sann_math.h score_key)."""
    b = np.ascontiguousarray(scores, np.float64).view(np.uint64)
    return np.where((b >> np.uint64(63)).astype(bool), ~b, b | np.uint64(1 << 63))


def cluster_range_bounds(list_lengths: np.ndarray, n_shards: int) -> np.ndarray:
    """Split the (ascending) clusters into n_shards contiguous ranges of about equal POSTING MASS (Zipf: balancing by
    id count would give the first range most of the work).  Returns n_shards + 1 indices into the cluster array."""
    mass = np.concatenate([[0], np.cumsum(np.asarray(list_lengths, np.int64))])
    cuts = [0]
    for g in range(1, n_shards):
        cuts.append(int(np.searchsorted(mass, mass[-1] * g / n_shards, side="left")))
    cuts.append(len(list_lengths))
    return np.maximum.accumulate(np.array(cuts, np.int64))


def scanned_clusters(offs, cids, scs, max_scan: int) -> np.ndarray:
    """Union over the batch of the clusters a query scans: SimClustersEmbedding constructor (score > 0, by score
    descending then cluster id) + truncate(maxScanClusters) (SimClustersEmbedding.scala:490-509,377-392)."""
    out = []
    for q in range(len(offs) - 1):
        c, s = np.asarray(cids[offs[q]:offs[q + 1]]), np.asarray(scs[offs[q]:offs[q + 1]])
        keep = s > 0
        c, s = c[keep], s[keep]
        order = np.lexsort((c, -s))[:max(max_scan, 0)]
        out.append(c[order])
    return np.unique(np.concatenate(out)) if out else np.empty(0, np.int32)


def _mix64(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint64)
    with np.errstate(over="ignore"):
        x ^= x >> np.uint64(33)
        x *= np.uint64(0xff51afd7ed558ccd)
        x ^= x >> np.uint64(33)
        x *= np.uint64(0xc4ceb9fe1a85ec53)
        x ^= x >> np.uint64(33)
    return x


class RangeShard:
    """One rank's part of the cluster-id-range deployment, through the C ABI (include/simclusters_ann.h): the sender side
    (prefix counts, packed export) and the receiver side (temporary index over the postings that arrived, the ordinary batch on
    it).  `exchange` is whatever moves device bytes between ranks: RCCL (sann_exchange_postings_by_tweet_hash) between
    processes, a device-to-device copy between logical shards of one process."""

    def __init__(self, pkg, index, rank: int, world: int, *, device: int = 0, n_partitions: int = 0):
        self.pkg, self.index, self.rank, self.world, self.device, self.n_partitions = pkg, index, rank, world, device, n_partitions
        self.lib = pkg.load_library()
        self._bufs = []

    def _alloc(self, n_bytes: int) -> int:
        import ctypes as C
        p = C.c_void_p()
        rc = self.lib.sann_device_alloc(self.device, max(int(n_bytes), 16), C.byref(p))
        assert rc == 0, self.lib.sann_last_error()
        self._bufs.append(p.value)
        return p.value

    def free(self):
        for p in self._bufs:
            self.lib.sann_device_free(self.device, p)
        self._bufs = []

    def export_counts(self, clusters: np.ndarray, M: int) -> np.ndarray:
        """counts[cluster][dest] of the top-M prefixes of `clusters` (the batch's scanned clusters, ascending; the ones this
        shard does not hold count zero), computed on the device."""
        import ctypes as C
        c = np.ascontiguousarray(clusters, np.int32)
        out = np.zeros((len(c), self.world), np.int32)
        rc = self.lib.sann_index_export_prefix_counts(self.index.handle, None, len(c), c.ctypes.data_as(C.c_void_p), int(M), self.world,
                                                      out.ctypes.data_as(C.c_void_p))
        assert rc == 0, self.lib.sann_last_error()
        return out

    def export(self, clusters: np.ndarray, M: int, counts: np.ndarray):
        """The postings, packed destination-major (clusters ascending inside a destination, ranks ascending inside a cluster)
        in one device buffer: (device pointer, send_counts[world] in postings)."""
        import ctypes as C
        c = np.ascontiguousarray(clusters, np.int32)
        per_dest = counts.sum(axis=0).astype(np.int64)
        dest_base = np.concatenate([[0], np.cumsum(per_dest)])
        seg = (dest_base[:-1][None, :] + np.concatenate([np.zeros((1, self.world), np.int64), np.cumsum(counts, axis=0)[:-1]])).astype(np.int64)
        seg = np.ascontiguousarray(seg)
        d_send = self._alloc(int(dest_base[-1]) * 16)
        rc = self.lib.sann_index_export_prefixes_device(self.index.handle, None, len(c), c.ctypes.data_as(C.c_void_p), int(M), self.world,
                                                        seg.ctypes.data_as(C.c_void_p), C.c_void_p(d_send))
        assert rc == 0, self.lib.sann_last_error()
        return d_send, per_dest

    def build_received(self, clusters: np.ndarray, counts_to_me: np.ndarray, d_recv: int):
        """The temporary index of this rank's tweets: clusters ascending (every cluster has ONE source rank, and the sources'
        ranges ascend with the rank, so source-major arrival order IS ascending cluster order), lists in rank order."""
        import ctypes as C
        sa = self.pkg.simclusters_ann
        keep = counts_to_me > 0
        cl = np.ascontiguousarray(np.asarray(clusters, np.int32)[keep])
        offs = np.ascontiguousarray(np.concatenate([[0], np.cumsum(counts_to_me[keep].astype(np.int64))]))
        opts = sa.sann_index_options_t(self.device, self.n_partitions, 0, 1)
        h = C.c_void_p()
        rc = self.lib.sann_index_build_from_device_postings(C.byref(opts), len(cl), cl.ctypes.data_as(C.c_void_p), offs.ctypes.data_as(C.c_void_p),
                                                            C.c_void_p(d_recv), C.byref(h))
        assert rc == 0, self.lib.sann_last_error()
        ix = sa.ClusterTweetIndex.__new__(sa.ClusterTweetIndex)
        ix._h, ix.device, ix.cluster_ids, ix.list_offsets = h, self.device, cl, offs
        return ix


class ClusterRangeDeployment:
    """N cluster-range shards as LOGICAL shards on one GPU, every step through the library's entry points (the same calls
    bench.py --sharding cluster-range makes with one process per GPU): device-side export of the scanned lists' top-M prefixes,
    the exchange (here a device-to-device regrouping; between processes sann_exchange_postings_by_tweet_hash over RCCL), the
    device-side re-partition into a temporary index, the ordinary pipeline, and the owner's device merge (sann_merge_shards)."""

    def __init__(self, pkg, cluster_ids, list_offsets, tweet_ids, scores, n_shards: int, *, device: int = 0, n_partitions: int = 0):
        self.pkg, self.n_shards, self.device, self.n_partitions = pkg, n_shards, device, n_partitions
        cluster_ids = np.asarray(cluster_ids, np.int32)
        list_offsets = np.asarray(list_offsets, np.int64)
        self.bounds = cluster_range_bounds(np.diff(list_offsets), n_shards)
        self.shards = []
        for g in range(n_shards):
            lo, hi = int(self.bounds[g]), int(self.bounds[g + 1])
            o = list_offsets[lo:hi + 1] - list_offsets[lo]
            ix = pkg.ClusterTweetIndex(cluster_ids[lo:hi], o, tweet_ids[list_offsets[lo]:list_offsets[hi]],
                                       scores[list_offsets[lo]:list_offsets[hi]], device=device, n_partitions=n_partitions)  # every tweet, some clusters
            self.shards.append(RangeShard(pkg, ix, g, n_shards, device=device, n_partitions=n_partitions))
        self.cluster_ids = cluster_ids

    def close(self):
        for s in self.shards:
            s.free()
            s.index.close()

    def get_tweet_candidates(self, offs, cids, scs, cfg, *, now_ms: int, variant=None):
        """One batch, every query under `cfg`.  Returns (ids [nq, k], scores, counts, map_sizes, stats)."""
        import ctypes as C
        pkg, N = self.pkg, self.n_shards
        lib = pkg.load_library()
        nq, k = len(offs) - 1, min(int(cfg.maxNumResults), 1000)
        need = scanned_clusters(offs, cids, scs, int(cfg.maxScanClusters))
        M = max(int(cfg.maxTopTweetsPerCluster), 0)
        # ---- every shard, on the device: counts[cluster][dest], then the packed prefixes -----------------------------------
        counts = [sh.export_counts(need, M) for sh in self.shards]          # [source][cluster][dest]
        sends = [sh.export(need, M, counts[g]) for g, sh in enumerate(self.shards)]
        # ---- the exchange (what sann_exchange_postings_by_tweet_hash does between processes): destination r receives, source by
        # source, the segment each source packed for it
        per = []
        moved = total = 0
        for r, sh in enumerate(self.shards):
            recv_counts = np.array([int(counts[g][:, r].sum()) for g in range(N)], np.int64)
            d_recv = sh._alloc(int(recv_counts.sum()) * 16)
            o = 0
            for g in range(N):
                src_off = int(sends[g][1][:r].sum())
                assert lib.sann_device_copy(self.device, C.c_void_p(d_recv + o * 16), C.c_void_p(sends[g][0] + src_off * 16), int(recv_counts[g]) * 16) == 0
                o += int(recv_counts[g])
                total += int(recv_counts[g])
                moved += int(recv_counts[g]) if g != r else 0
            to_me = np.zeros(len(need), np.int64)
            for g in range(N):
                to_me += counts[g][:, r]
            # ---- its tweets' postings as a temporary index, the ordinary pipeline on it -----------------------------------
            ix = sh.build_received(need, to_me, d_recv)
            qb = pkg.QueryBatch(ix, offs, cids, scs, cfg, now_ms=now_ms) if variant is None else \
                pkg.QueryBatch(ix, offs, cids, scs, cfg, now_ms=now_ms, variant=variant)
            qb.run(); qb.finish()
            per.append((qb, ix))
        # ---- owner merge on the device: exact top-k of the N per-shard lists (sann_merge_shards; here every query has one owner)
        stride = per[0][0].stride
        arr = nq * stride * 8
        buf = self.shards[0]._alloc(N * (2 * arr + 8 * nq))
        for r, (qb, _ix) in enumerate(per):
            (d_ids, d_sc, d_cnt, d_msz), _ = qb.device_results()
            base = buf + r * (2 * arr + 8 * nq)
            for dst, src, n in ((base, d_ids, arr), (base + arr, d_sc, arr), (base + 2 * arr, d_cnt, 4 * nq), (base + 2 * arr + 4 * nq, d_msz, 4 * nq)):
                assert lib.sann_device_copy(self.device, C.c_void_p(dst), C.c_void_p(src), n) == 0
        out = self.shards[0]._alloc(2 * arr + 8 * nq)
        rc = lib.sann_merge_shards(self.device, None, N, nq, stride, 2 * arr + 8 * nq, C.c_void_p(buf), C.c_void_p(buf + arr), C.c_void_p(buf + 2 * arr),
                                   C.c_void_p(buf + 2 * arr + 4 * nq), C.c_void_p(per[0][0].device_k()), C.c_void_p(out), C.c_void_p(out + arr),
                                   C.c_void_p(out + 2 * arr), C.c_void_p(out + 2 * arr + 4 * nq))
        assert rc == 0, lib.sann_last_error()
        assert lib.sann_device_synchronize(self.device) == 0
        ids = np.zeros((nq, stride), np.int64); sc = np.zeros((nq, stride)); cnt = np.zeros(nq, np.int32); msz = np.zeros(nq, np.int32)
        for dst, src in ((ids, out), (sc, out + arr), (cnt, out + 2 * arr), (msz, out + 2 * arr + 4 * nq)):
            assert lib.sann_device_copy(self.device, dst.ctypes.data_as(C.c_void_p), C.c_void_p(src), dst.nbytes) == 0
        for qb, ix in per:
            qb.close(); ix.close()
        for sh in self.shards:
            sh.free()
        return ids[:, :k], sc[:, :k], cnt, msz, {"scanned_clusters": int(len(need)), "postings_regrouped": total, "bytes_moved": moved * 16,
                                                 "bytes_moved_per_gpu": moved * 16 // max(N, 1)}


class ClusterRangeRank:
    """ONE rank of the cluster-id-range deployment with one process per GPU (bench.py --sharding cluster-range): every step
    through the library -- export counts, counts to the destinations (sann_exchange_to_owners), packed prefixes
    (sann_index_export_prefixes_device), postings to the GPU their tweet hashes to (sann_exchange_postings_by_tweet_hash over
    RCCL), a temporary index of this GPU's tweets (sann_index_build_from_device_postings), the ordinary batch on it with its
    results bound owner-chunked, results to the owners (sann_exchange_to_owners) and the owner's proving merge
    (sann_merge_shards_cut).  `index` may hold more clusters than the rank's range [first_cluster, end_cluster): only the
    range is ever exported (the synthetic generator has no range filter, so every rank of the bench generates the corpus)."""

    def __init__(self, pkg, index, comm, rank: int, world: int, first_cluster: int, end_cluster: int, *, device: int, n_partitions: int):
        import ctypes as C
        self.C = C
        self.pkg, self.comm, self.rank, self.world, self.device = pkg, comm, rank, world, device
        self.lib = pkg.load_library()
        self.shard = RangeShard(pkg, index, rank, world, device=device, n_partitions=n_partitions)
        self.first, self.end = first_cluster, end_cluster
        self._grow = {}

    def _buf(self, name: str, n_bytes: int) -> int:
        """A named device buffer that only grows."""
        cur = self._grow.get(name)
        if cur is None or cur[1] < n_bytes:
            if cur is not None:
                self.lib.sann_device_free(self.device, cur[0])
            p = self.C.c_void_p()
            want = int(n_bytes * 1.25) + 4096
            assert self.lib.sann_device_alloc(self.device, want, self.C.byref(p)) == 0, self.lib.sann_last_error()
            cur = (p.value, want)
            self._grow[name] = cur
        return cur[0]

    def close(self):
        for p, _ in self._grow.values():
            self.lib.sann_device_free(self.device, p)
        self._grow = {}
        self.shard.free()

    def step(self, need: np.ndarray, queries, cfg_run, nql: int, K: int, shard_k: int, now_ms: int, out, d_bad: int, stream: int = 0):
        """One batch.  need = the batch's scanned clusters (ascending); queries = (offs, cids, scs) of ALL world * nql queries;
        out = device pointers (ids, scores, counts, map sizes) of this rank's nql merged results."""
        C, lib, N, me = self.C, self.lib, self.world, self.rank
        M = max(int(cfg_run.maxTopTweetsPerCluster), 0)
        st = C.c_void_p(stream)
        # 1. counts[cluster][dest] of the clusters of MY range
        mine = (need >= self.first) & (need < self.end)
        counts = np.zeros((len(need), N), np.int32)
        if mine.any():
            counts[mine] = self.shard.export_counts(need[mine], M)
        # 2. every destination learns what each source will send it: block r of my message = counts[:, r]
        blk = (len(need) * 4 + 7) // 8 * 8
        d_cs, d_cr = self._buf("cs", N * blk), self._buf("cr", N * blk)
        h = np.zeros((N, blk // 4), np.int32)
        h[:, :len(need)] = counts.T
        assert lib.sann_device_copy(self.device, C.c_void_p(d_cs), h.ctypes.data_as(C.c_void_p), h.nbytes) == 0
        assert lib.sann_exchange_to_owners(self.comm, st, C.c_void_p(d_cs), C.c_void_p(d_cr), blk) == 0, lib.sann_last_error()
        assert lib.sann_device_synchronize(self.device) == 0
        got = np.zeros((N, blk // 4), np.int32)
        assert lib.sann_device_copy(self.device, got.ctypes.data_as(C.c_void_p), C.c_void_p(d_cr), got.nbytes) == 0
        from_src = got[:, :len(need)].astype(np.int64)  # [source][cluster] postings for me
        # 3. the packed prefixes, destination-major
        per_dest = counts.sum(axis=0).astype(np.int64)
        d_send = self._buf("send", int(per_dest.sum()) * 16)
        if mine.any():
            own = counts[mine]
            dest_base = np.concatenate([[0], np.cumsum(per_dest)])
            seg = np.ascontiguousarray(dest_base[:-1][None, :] + np.concatenate([np.zeros((1, N), np.int64), np.cumsum(own, axis=0)[:-1]]), np.int64)
            cl = np.ascontiguousarray(need[mine], np.int32)
            assert lib.sann_index_export_prefixes_device(self.shard.index.handle, st, len(cl), cl.ctypes.data_as(C.c_void_p), M, N,
                                                         seg.ctypes.data_as(C.c_void_p), C.c_void_p(d_send)) == 0, lib.sann_last_error()
        # 4. postings to the GPU their tweet hashes to
        recv_counts = np.ascontiguousarray(from_src.sum(axis=1), np.int64)
        d_recv = self._buf("recv", int(recv_counts.sum()) * 16)
        sc_ = np.ascontiguousarray(per_dest, np.int64)
        assert lib.sann_exchange_postings_by_tweet_hash(self.comm, st, C.c_void_p(d_send), sc_.ctypes.data_as(C.c_void_p), C.c_void_p(d_recv),
                                                        recv_counts.ctypes.data_as(C.c_void_p)) == 0, lib.sann_last_error()
        assert lib.sann_device_synchronize(self.device) == 0
        # 5. my tweets' postings as a temporary index (sources' ranges ascend with the rank: arrival order = ascending clusters)
        ix = self.shard.build_received(need, from_src.sum(axis=0), d_recv)
        # 6. the ordinary batch on it, results bound owner-chunked
        offs, cids, scs = queries
        qb = self.pkg.QueryBatch(ix, offs, cids, scs, cfg_run, now_ms=now_ms)
        stride = qb.stride
        chunk, _ = owner_message_layout(nql, stride)
        arr = nql * stride * 8
        d_msg, d_got = self._buf("msg", N * chunk), self._buf("got", N * chunk)
        qb.bind_outputs_chunked(d_msg, d_msg + arr, d_msg + 2 * arr, d_msg + 2 * arr + 4 * nql, nql, chunk)
        qb.run(stream)
        qb.finish(stream)
        unit_ms = 0.0
        # 7. results to the owners, proving merge
        assert lib.sann_exchange_to_owners(self.comm, st, C.c_void_p(d_msg), C.c_void_p(d_got), chunk) == 0, lib.sann_last_error()
        o_ids, o_sc, o_cnt, o_msz = out
        if shard_k < K:
            rc = lib.sann_merge_shards_cut(self.device, st, N, nql, stride, chunk, shard_k, K, K, C.c_void_p(d_got), C.c_void_p(d_got + arr),
                                           C.c_void_p(d_got + 2 * arr), C.c_void_p(d_got + 2 * arr + 4 * nql), C.c_void_p(o_ids), C.c_void_p(o_sc),
                                           C.c_void_p(o_cnt), C.c_void_p(o_msz), C.c_void_p(d_bad))
        else:
            rc = lib.sann_merge_shards(self.device, st, N, nql, stride, chunk, C.c_void_p(d_got), C.c_void_p(d_got + arr), C.c_void_p(d_got + 2 * arr),
                                       C.c_void_p(d_got + 2 * arr + 4 * nql), C.c_void_p(qb.device_k() + self.rank * nql * 4), C.c_void_p(o_ids),
                                       C.c_void_p(o_sc), C.c_void_p(o_cnt), C.c_void_p(o_msz))
        assert rc == 0, lib.sann_last_error()
        assert lib.sann_device_synchronize(self.device) == 0
        stats = qb.stats()
        qb.close()
        ix.close()
        return {"postings_sent": int(per_dest.sum() - per_dest[me]), "postings_received": int(recv_counts.sum() - recv_counts[me]),
                "postings_scanned": int(stats.postings_scanned), "fallback_units": int(stats.n_fallback_units), "unit_ms": unit_ms}
