"""ctypes binding of include/representation_scorer.h and a mirror of the reference's pair scorer.

Reference (paths relative to /root/reference/):
  src/scala/com/twitter/simclusters_v2/score/SimClustersEmbeddingPairScoreStore.scala:39-199
  src/thrift/com/twitter/simclusters_v2/score.thrift:14-22   ScoringAlgorithm ids
"""
from __future__ import annotations

import ctypes as C
import enum
from typing import Sequence, Tuple

import numpy as np

from .simclusters_ann import load_library


class ScoringAlgorithm(enum.IntEnum):
    """score.thrift:14-22 (the pairwise block)."""

    PairEmbeddingDotProduct = 1
    PairEmbeddingCosineSimilarity = 2
    PairEmbeddingJaccardSimilarity = 3
    PairEmbeddingEuclideanDistance = 4
    PairEmbeddingManhattanDistance = 5
    PairEmbeddingLogCosineSimilarity = 6
    PairEmbeddingExpScaledCosineSimilarity = 7


PROTOS = {
    "rsx_last_error": (C.c_char_p, []),
    "rsx_pair_scores": (C.c_int, [C.c_int32, C.c_int32, C.c_int32] + [C.c_void_p] * 6 + [C.c_int32, C.c_void_p]),
    "rsx_pair_scores_device": (C.c_int, [C.c_int32, C.c_void_p, C.c_int32, C.c_int32] + [C.c_void_p] * 7),
}


def _lib():
    lib = load_library()
    if not getattr(lib, "_rsx_ready", False):
        for name, (res, args) in PROTOS.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        lib._rsx_ready = True
    return lib


def simclusters_embedding(pairs: Sequence[Tuple[int, float]]):
    """The SimClustersEmbedding constructor's view (SimClustersEmbedding.scala:490-509): drop score <= 0,
    return (sortedClusterIds int32, sortedScores float64) ascending by cluster id."""
    kept = sorted((int(c), float(s)) for c, s in pairs if s > 0.0)
    return np.array([c for c, _ in kept], np.int32), np.array([s for _, s in kept], np.float64)


def pair_scores(algorithm: ScoringAlgorithm, a_offsets, a_ids, a_scores, b_offsets, b_ids, b_scores, *, device: int = 0,
                validate: bool = True) -> np.ndarray:
    """Scores of n pairs; side A and side B as CSR over already-sorted embeddings."""
    lib = _lib()
    ao = np.ascontiguousarray(a_offsets, np.int64); bo = np.ascontiguousarray(b_offsets, np.int64)
    ai = np.ascontiguousarray(a_ids, np.int32); bi = np.ascontiguousarray(b_ids, np.int32)
    asx = np.ascontiguousarray(a_scores, np.float64); bsx = np.ascontiguousarray(b_scores, np.float64)
    n = len(ao) - 1
    out = np.zeros(n, np.float64)
    p = lambda x: x.ctypes.data_as(C.c_void_p)
    rc = lib.rsx_pair_scores(device, int(algorithm), n, p(ao), p(ai), p(asx), p(bo), p(bi), p(bsx), 1 if validate else 0, p(out))
    if rc != 0:
        raise RuntimeError(f"representation_scorer error {rc}: {lib.rsx_last_error().decode()}")
    return out
