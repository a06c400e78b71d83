"""Multi-GPU plumbing shared by bench.py and the tests: the packed per-rank result buffer that one
all-gather moves.

Layout, in int64 words, for nq queries and row stride `stride`:
    [0, nq*stride)              tweet ids
    [nq*stride, 2*nq*stride)    scores (fp64 bit patterns)
    [2*nq*stride, +nq/2...)     counts int32[nq] then map sizes int32[nq]   (nq int64 words)
"""
from __future__ import annotations

import numpy as np


def packed_words(nq: int, stride: int) -> int:
    return 2 * nq * stride + nq


def packed_offsets(nq: int, stride: int):
    """Byte offsets of (ids, scores, counts, map_sizes) inside one rank's packed buffer."""
    return 0, nq * stride * 8, 2 * nq * stride * 8, 2 * nq * stride * 8 + nq * 4


def pack(ids: np.ndarray, scores: np.ndarray, counts: np.ndarray, map_sizes: np.ndarray) -> np.ndarray:
    nq, stride = ids.shape
    buf = np.zeros(packed_words(nq, stride), np.int64)
    buf[:nq * stride] = ids.reshape(-1)
    buf[nq * stride:2 * nq * stride] = scores.reshape(-1).view(np.int64)
    tail = buf[2 * nq * stride:].view(np.int32)
    tail[:nq] = counts
    tail[nq:2 * nq] = map_sizes
    return buf


def unpack(buf: np.ndarray, nq: int, stride: int):
    ids = buf[:nq * stride].reshape(nq, stride)
    scores = buf[nq * stride:2 * nq * stride].view(np.float64).reshape(nq, stride)
    tail = buf[2 * nq * stride:].view(np.int32)
    return ids, scores, tail[:nq], tail[nq:2 * nq]
