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
# This module carries that out with logical shards on one GPU through the C ABI (index builds, batches); the exchange is
# a host-side regrouping whose bytes are counted.  It is the reference point beside the tweet-hash deployment, not a
# competitor: it re-partitions ~N x M x 16 B per distinct scanned cluster on EVERY batch.
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


class ClusterRangeDeployment:
    """N logical cluster-range shards on one GPU (see the block comment above)."""

    def __init__(self, pkg, cluster_ids, list_offsets, tweet_ids, scores, n_shards: int, *, device: int = 0, n_partitions: int = 0):
        self.pkg, self.n_shards, self.device, self.n_partitions = pkg, n_shards, device, n_partitions
        cluster_ids = np.asarray(cluster_ids, np.int32)
        list_offsets = np.asarray(list_offsets, np.int64)
        self.bounds = cluster_range_bounds(np.diff(list_offsets), n_shards)
        self.range_first = [int(cluster_ids[self.bounds[g]]) if self.bounds[g] < len(cluster_ids) else 1 << 31 for g in range(n_shards)]
        self.shards = []
        for g in range(n_shards):
            lo, hi = int(self.bounds[g]), int(self.bounds[g + 1])
            o = list_offsets[lo:hi + 1] - list_offsets[lo]
            self.shards.append(pkg.ClusterTweetIndex(cluster_ids[lo:hi], o, tweet_ids[list_offsets[lo]:list_offsets[hi]],
                                                     scores[list_offsets[lo]:list_offsets[hi]], device=device,
                                                     n_partitions=n_partitions))  # every tweet, some clusters
        self.cluster_ids = cluster_ids

    def close(self):
        for s in self.shards:
            s.close()

    def get_tweet_candidates(self, offs, cids, scs, cfg, *, now_ms: int, variant=None):
        """One batch, every query under `cfg`.  Returns (ids [nq, k], scores, counts, map_sizes, stats)."""
        pkg, N = self.pkg, self.n_shards
        nq, k = len(offs) - 1, min(int(cfg.maxNumResults), 1000)
        need = scanned_clusters(offs, cids, scs, int(cfg.maxScanClusters))
        M = max(int(cfg.maxTopTweetsPerCluster), 0)
        # ---- every shard: the top-M prefix of each scanned list it owns ------------------------------------------------
        l_c, l_t, l_s, l_src = [], [], [], []
        for g, ix in enumerate(self.shards):
            lo, hi = int(self.bounds[g]), int(self.bounds[g + 1])
            mine = need[np.isin(need, self.cluster_ids[lo:hi])]
            for c in mine:
                t, s, _r = ix.get_list(int(c))
                l_c.append(int(c)); l_t.append(t[:M]); l_s.append(s[:M]); l_src.append(g)
        order = np.argsort(np.array(l_c, np.int64), kind="stable")  # the regrouped lists, by ascending cluster id
        lens = np.array([len(l_t[i]) for i in order], np.int64)
        g_off = np.concatenate([[0], np.cumsum(lens)])
        g_c = np.array([l_c[i] for i in order], np.int32)
        g_t = np.concatenate([l_t[i] for i in order]) if len(order) else np.empty(0, np.int64)
        g_s = np.concatenate([l_s[i] for i in order]) if len(order) else np.empty(0, np.float64)
        src = np.repeat(np.array([l_src[i] for i in order], np.int64), lens)
        # ---- the exchange: posting -> GPU hash(tweetId) % N (csrc/sann_device.h tweet_shard; checked against the library)
        lib = pkg.load_library()
        dst = ((_mix64(g_t.view(np.uint64)) >> np.uint64(40)) % np.uint64(N)).astype(np.int64) if N > 1 else np.zeros(len(g_t), np.int64)
        for i in range(0, len(g_t), max(1, len(g_t) // 8)):
            assert dst[i] == lib.sann_tweet_shard(int(g_t[i]), N)
        moved = int((src != dst).sum()) * 16
        # ---- every GPU: its tweets' postings as a temporary tweet-hash shard, the ordinary pipeline on it ---------------
        per = []
        for r in range(N):
            ix = pkg.ClusterTweetIndex(g_c, g_off, g_t, g_s, device=self.device, n_partitions=self.n_partitions, shard_id=r, n_shards=N)
            qb = pkg.QueryBatch(ix, offs, cids, scs, cfg, now_ms=now_ms) if variant is None else \
                pkg.QueryBatch(ix, offs, cids, scs, cfg, now_ms=now_ms, variant=variant)
            qb.run(); qb.finish()
            per.append(qb.results())
            qb.close(); ix.close()
        # ---- owner merge: exact top-k of the N per-shard lists, (score desc by Double.compare, tweet id asc) -------------
        ids = np.zeros((nq, k), np.int64); sc = np.zeros((nq, k)); cnt = np.zeros(nq, np.int32); msz = np.zeros(nq, np.int32)
        for q in range(nq):
            ti = np.concatenate([p[0][q, :p[2][q]] for p in per])
            ts = np.concatenate([p[1][q, :p[2][q]] for p in per])
            o = np.lexsort((ti, ~score_key(ts)))[:k]
            cnt[q] = len(o); ids[q, :len(o)] = ti[o]; sc[q, :len(o)] = ts[o]
            msz[q] = sum(int(p[3][q]) for p in per)
        return ids, sc, cnt, msz, {"scanned_clusters": int(len(need)), "postings_regrouped": int(len(g_t)), "bytes_moved": moved,
                                  "bytes_moved_per_gpu": moved // max(N, 1)}
