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
