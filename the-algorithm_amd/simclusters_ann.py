"""ctypes binding of include/simclusters_ann.h plus a thin mirror of the reference's operator
interface, so that tests read like the Scala call sites.

Reference interface mirrored (paths relative to /root/reference/):
  simclusters-ann/server/src/main/scala/com/twitter/simclustersann/candidate_source/
    ApproximateCosineSimilarity.scala:26-36        trait ApproximateCosineSimilarity.apply
    SimClustersANNCandidateSource.scala:66-95      fetchCandidates (cluster choice + call)
  simclusters-ann/thrift/src/main/thrift/simClustersAnn.thrift:18-37   SimClustersANNConfig, ScoringAlgorithm

Everything here goes through libsimclusters_amd.so; there is no CPU fallback.  Loading fails
loudly when the library has not been built (`make -C the-algorithm_amd/csrc`).
"""
from __future__ import annotations

import ctypes as C
import dataclasses
import enum
import os
import sys
from typing import Callable, Dict, Iterable, List, Mapping, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SANN_LIB_PATH") or os.path.join(_HERE, "libsimclusters_amd.so")  # (override: A/B builds)


class ScoringAlgorithm(enum.IntEnum):
    """simClustersAnn.thrift:32-37"""

    DotProduct = 1
    CosineSimilarity = 2
    LogCosineSimilarity = 3
    CosineSimilarityNoSourceEmbeddingNormalization = 4
    # the offline all-users job's scores (scio/bq_generation/sql/tweets_ann.sql:44-52): not thrift values
    OfflineLogCosineSimilarity = 5
    OfflineCosineSimilarity = 6


class Variant(enum.IntEnum):
    """Flag `approximate_cosine_similarity` (SimClustersANNCandidateSourceModule.scala:19-38)."""

    original = 0
    optimized = 1
    experimental = 2
    legacy = 3  # simclusters_v2/candidate_source/SimClustersANNCandidateSource.scala:107-181 (not a flag value there)


class sann_config_t(C.Structure):
    _fields_ = [
        ("max_num_results", C.c_int32),
        ("candidate_embedding_type", C.c_int32),
        ("min_score", C.c_double),
        ("max_top_tweets_per_cluster", C.c_int32),
        ("max_scan_clusters", C.c_int32),
        ("max_tweet_candidate_age_hours", C.c_int32),
        ("min_tweet_candidate_age_hours", C.c_int32),
        ("ann_algorithm", C.c_int32),
        ("reserved", C.c_int32),
    ]


class sann_index_options_t(C.Structure):
    _fields_ = [("device", C.c_int32), ("n_partitions", C.c_int32), ("shard_id", C.c_int32), ("n_shards", C.c_int32)]


class sann_index_info_t(C.Structure):
    _fields_ = [
        ("n_clusters", C.c_int64),
        ("n_postings", C.c_int64),
        ("n_postings_total", C.c_int64),
        ("device_bytes", C.c_int64),
        ("n_partitions", C.c_int32),
        ("shard_id", C.c_int32),
        ("n_shards", C.c_int32),
        ("max_list_len", C.c_int32),
    ]


class sann_synth_params_t(C.Structure):
    _fields_ = [
        ("n_tweets", C.c_int64),
        ("now_ms", C.c_int64),
        ("seed", C.c_uint64),
        ("n_clusters", C.c_int32),
        ("index_cap", C.c_int32),
        ("window_hours", C.c_int32),
        ("max_clusters_per_tweet", C.c_int32),
        ("mean_clusters", C.c_float),
        ("reserved", C.c_int32),
    ]


class sann_batch_stats_t(C.Structure):
    _fields_ = [
        ("postings_scanned", C.c_int64),
        ("algorithmic_bytes", C.c_int64),
        ("n_units", C.c_int32),
        ("n_fallback_units", C.c_int32),
        ("n_requeried", C.c_int32),
        ("max_unit_postings", C.c_int32),
    ]


@dataclasses.dataclass
class SimClustersANNConfig:
    """simClustersAnn.thrift:18-27.  Defaults = cr-mixer's DefaultConfig
    (cr-mixer/server/src/main/scala/com/twitter/cr_mixer/config/SimClustersANNConfig.scala:33-42)."""

    maxNumResults: int = 200
    minScore: float = 0.0
    candidateEmbeddingType: int = 0
    maxTopTweetsPerCluster: int = 800
    maxScanClusters: int = 50
    maxTweetCandidateAgeHours: int = 24
    minTweetCandidateAgeHours: int = 0
    annAlgorithm: ScoringAlgorithm = ScoringAlgorithm.CosineSimilarity

    def to_c(self) -> sann_config_t:
        return sann_config_t(
            int(self.maxNumResults),
            int(self.candidateEmbeddingType),
            float(self.minScore),
            int(self.maxTopTweetsPerCluster),
            int(self.maxScanClusters),
            int(self.maxTweetCandidateAgeHours),
            int(self.minTweetCandidateAgeHours),
            int(self.annAlgorithm),
            0,
        )

    @classmethod
    def from_c(cls, c: sann_config_t) -> "SimClustersANNConfig":
        try:
            alg = ScoringAlgorithm(c.ann_algorithm)
        except ValueError:
            alg = c.ann_algorithm  # an id this build does not know: carried as the wire has it
        return cls(c.max_num_results, c.min_score, c.candidate_embedding_type, c.max_top_tweets_per_cluster, c.max_scan_clusters,
                   c.max_tweet_candidate_age_hours, c.min_tweet_candidate_age_hours, alg)


class SannError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"simclusters_amd error {code}: {msg}")
        self.code = code


_lib = None

_PROTOS = {
    "sann_last_error": (C.c_char_p, []),
    "sann_version": (C.c_char_p, []),
    "sann_runtime_advice": (C.c_char_p, []),
    "sann_index_build": (C.c_int, [C.POINTER(sann_index_options_t), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "sann_index_build_with_norms": (C.c_int, [C.POINTER(sann_index_options_t), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "sann_index_build_from_postings": (C.c_int, [C.POINTER(sann_index_options_t), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.POINTER(C.c_void_p)]),
    "sann_topk_merge": (C.c_int, [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_int32, C.c_double, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sann_index_build_synthetic": (C.c_int, [C.POINTER(sann_index_options_t), C.POINTER(sann_synth_params_t), C.POINTER(C.c_void_p)]),
    "sann_synth_tweet_embeddings": (C.c_int, [C.c_int32, C.POINTER(sann_synth_params_t), C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sann_synth_exact_cosine_topk": (C.c_int, [C.c_int32, C.POINTER(sann_synth_params_t), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sann_synth_tweet_id": (C.c_int64, [C.c_int64, C.c_int64, C.c_int64, C.c_int32]),
    "sann_index_info": (C.c_int, [C.c_void_p, C.POINTER(sann_index_info_t)]),
    "sann_index_get_list": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]),
    "sann_index_destroy": (C.c_int, [C.c_void_p]),
    "sann_batch_create": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "sann_batch_reset": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "sann_host_alloc": (C.c_int, [C.c_int64, C.POINTER(C.c_void_p)]),
    "sann_host_free": (C.c_int, [C.c_void_p]),
    "sann_batch_run": (C.c_int, [C.c_void_p, C.c_void_p]),
    "sann_batch_run_after": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]),
    "sann_batch_finish": (C.c_int, [C.c_void_p, C.c_void_p]),
    "sann_batch_results": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "sann_batch_device_results": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int32)]),
    "sann_batch_device_k": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "sann_batch_stats": (C.c_int, [C.c_void_p, C.POINTER(sann_batch_stats_t)]),
    "sann_batch_set_profiling": (C.c_int, [C.c_void_p, C.c_int32]),
    "sann_batch_kernel_times": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int32)]),
    "sann_batch_desc_time": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "sann_tweet_shard": (C.c_int32, [C.c_int64, C.c_int32]),
    "sann_tweet_partition": (C.c_int32, [C.c_int64, C.c_int32]),
    "sann_device_synchronize": (C.c_int, [C.c_int32]),
    "sann_debug_overflow_reasons": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "sann_debug_plan_slow_tail": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "sann_debug_gather_probe": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    "sann_debug_unit_arrays": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sann_debug_phase_cycles": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_double)]),
    "sann_batch_destroy": (C.c_int, [C.c_void_p]),
    "sann_get_tweet_candidates": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "sann_get_tweet_candidates_at": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "sann_batcher_create": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "sann_batcher_destroy": (C.c_int, [C.c_void_p]),
    "sann_submit": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]),
    "sann_wait": (C.c_int, [C.c_void_p, C.c_int64]),
    "sann_poll": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(C.c_int32)]),
    "sann_batcher_get_tweet_candidates": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sann_batcher_stats": (C.c_int, [C.c_void_p, C.c_void_p]),
    "sann_heavy_rank": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "sann_index_export_prefix_counts": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "sann_index_export_prefixes_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "sann_exchange_postings_by_tweet_hash": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sann_index_build_from_device_postings": (C.c_int, [C.POINTER(sann_index_options_t), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "sann_device_alloc": (C.c_int, [C.c_int32, C.c_int64, C.POINTER(C.c_void_p)]),
    "sann_device_free": (C.c_int, [C.c_int32, C.c_void_p]),
    "sann_device_copy": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_int64]),
    "sann_debug_call_trace": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "sann_debug_normalise": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_void_p]),
    "sann_debug_wave_sort": (C.c_int, [C.c_int32, C.c_int32, C.c_void_p]),
    "sann_debug_approx": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]),
    "sann_comm_unique_id": (C.c_int, [C.c_void_p]),
    "sann_comm_create": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.POINTER(C.c_void_p)]),
    "sann_comm_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "sann_comm_destroy": (C.c_int, [C.c_void_p]),
    "sann_exchange_to_owners": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]),
    "sann_owner_message_layout": (C.c_int, [C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "sann_batch_bind_outputs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sann_merge_shards": (C.c_int, [C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sann_merge_shards_cut": (C.c_int, [C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_int32] + [C.c_void_p] * 9),
    "sann_batch_bind_outputs_chunked": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64]),
}


def exported_symbols() -> List[str]:
    """Every symbol include/simclusters_ann.h declares."""
    return list(_PROTOS)


def load_library(path: Optional[str] = None) -> C.CDLL:
    """dlopen libsimclusters_amd.so and attach prototypes.  Raises if it is not built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    # The launcher's part of the contract: batches in flight use several HIP streams, and the
    # runtime reads GPU_MAX_HW_QUEUES once, when it initialises -- so it is set here, by the process that hosts the
    # library and before anything touches HIP, never by the library (sann_runtime_advice() reports a smaller value).
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    # One HIP runtime per process.  torch bundles its own libamdhip64.so.7 and always loads it
    # (RPATH $ORIGIN); ours is found by soname, so it binds to whichever copy is loaded first.
    # If torch will be used in this process it therefore has to be imported BEFORE the dlopen
    # below, or torch later fails with "No HIP GPUs are available" (two ROCr instances).
    # Torch-free processes (bench.py at N=1, a JVM) set SANN_NO_TORCH=1 and get the system runtime.
    if os.environ.get("SANN_NO_TORCH") != "1" and "torch" not in sys.modules:
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    if not os.path.exists(p):
        raise FileNotFoundError(
            f"{p} is missing: build the HIP extension first (make -C the-algorithm_amd/csrc, or "
            "python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback."
        )
    lib = C.CDLL(p)
    for name, (res, args) in _PROTOS.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


def _check(rc: int) -> None:
    if rc != 0:
        raise SannError(rc, load_library().sann_last_error().decode())


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def topk_merge(a, b, *, top_k: int = 1600, threshold: float = 0.001, oldest_tweet_id: int = -(1 << 63), device: int = 0):
    """TopKTweetsWithScoresMonoid.plus for a batch of clusters (sann_topk_merge; Monoids.scala:131-158,378-450).
    a, b: (offsets int64[n+1], tweet_ids, values, scaled_times) CSR sides.  Returns the merged side in the same form,
    every list ordered by (value desc, tweet id asc)."""
    lib = load_library()
    ao, ai, av, at = (np.ascontiguousarray(x, dtype=d) for x, d in zip(a, (np.int64, np.int64, np.float64, np.float64)))
    bo, bi, bv, bt = (np.ascontiguousarray(x, dtype=d) for x, d in zip(b, (np.int64, np.int64, np.float64, np.float64)))
    if len(ao) != len(bo) or len(ao) < 1:
        raise ValueError("both sides need offsets for the same number of lists")
    n = len(ao) - 1
    cap = int(len(ai) + len(bi))
    oo = np.zeros(n + 1, np.int64)
    oi = np.zeros(max(cap, 1), np.int64); ov = np.zeros(max(cap, 1)); ot = np.zeros(max(cap, 1))
    _check(lib.sann_topk_merge(device, n, _ptr(ao), _ptr(ai), _ptr(av), _ptr(at), _ptr(bo), _ptr(bi), _ptr(bv), _ptr(bt),
                               int(top_k), float(threshold), int(oldest_tweet_id), cap, _ptr(oo), _ptr(oi), _ptr(ov), _ptr(ot)))
    m = int(oo[n])
    return oo, oi[:m].copy(), ov[:m].copy(), ot[:m].copy()


class ClusterTweetIndex:
    """Device-resident `ReadableStore[ClusterId, Seq[(TweetId, Double)]]`
    (ClusterTweetIndexProviderModule.scala:34-94): lists already filtered > 0, sorted by score
    descending and capped, exactly as the reference store returns them."""

    def __init__(self, cluster_ids, list_offsets, tweet_ids, scores, *, device: int = 0, n_partitions: int = 0,
                 shard_id: int = 0, n_shards: int = 1, tweet_norms=None):
        """tweet_norms (optional, one per posting): the tweet's full-embedding sum of squares, for the offline job's
        scores (sann_index_build_with_norms)."""
        lib = load_library()
        self.cluster_ids = np.ascontiguousarray(cluster_ids, dtype=np.int32)
        self.list_offsets = np.ascontiguousarray(list_offsets, dtype=np.int64)
        tweet_ids = np.ascontiguousarray(tweet_ids, dtype=np.int64)
        scores = np.ascontiguousarray(scores, dtype=np.float64)
        opts = sann_index_options_t(device, n_partitions, shard_id, n_shards)
        h = C.c_void_p()
        if tweet_norms is None:
            _check(lib.sann_index_build(C.byref(opts), len(self.cluster_ids), _ptr(self.cluster_ids), _ptr(self.list_offsets),
                                        _ptr(tweet_ids), _ptr(scores), C.byref(h)))
        else:
            norms = np.ascontiguousarray(tweet_norms, dtype=np.float64)
            assert norms.shape == scores.shape
            _check(lib.sann_index_build_with_norms(C.byref(opts), len(self.cluster_ids), _ptr(self.cluster_ids),
                                                   _ptr(self.list_offsets), _ptr(tweet_ids), _ptr(scores), _ptr(norms), C.byref(h)))
        self._h = h
        self.device = device

    @classmethod
    def from_raw_postings(cls, cluster_ids, list_offsets, tweet_ids, values, scaled_times, *, now_ms: int,
                          half_life_ms: int = 8 * 3600 * 1000, max_results: int = 2000, device: int = 0, n_partitions: int = 0,
                          shard_id: int = 0, n_shards: int = 1) -> "ClusterTweetIndex":
        """sann_index_build_from_postings: the store's raw (tweet, value, scaledTime) entries -> decay to now, keep > 0,
        sort descending, take(max_results), partition -- on the device."""
        lib = load_library()
        self = cls.__new__(cls)
        self.cluster_ids = np.ascontiguousarray(cluster_ids, dtype=np.int32)
        self.list_offsets = np.ascontiguousarray(list_offsets, dtype=np.int64)
        t = np.ascontiguousarray(tweet_ids, dtype=np.int64)
        v = np.ascontiguousarray(values, dtype=np.float64)
        st = None if scaled_times is None else np.ascontiguousarray(scaled_times, dtype=np.float64)
        opts = sann_index_options_t(device, n_partitions, shard_id, n_shards)
        h = C.c_void_p()
        _check(lib.sann_index_build_from_postings(C.byref(opts), len(self.cluster_ids), _ptr(self.cluster_ids), _ptr(self.list_offsets),
                                                  _ptr(t), _ptr(v), _ptr(st), int(now_ms), int(half_life_ms), int(max_results), C.byref(h)))
        self._h = h
        self.device = device
        return self

    @classmethod
    def synthetic(cls, n_tweets: int, n_clusters: int = 144_428, *, seed: int = 20260104, index_cap: int = 2000,
                  now_ms: int = 1_700_000_000_000, window_hours: int = 24, mean_clusters: float = 25.0,
                  max_clusters_per_tweet: int = 50, device: int = 0, n_partitions: int = 0, shard_id: int = 0,
                  n_shards: int = 1) -> "ClusterTweetIndex":
        """Synthetic corpus generated and indexed on the device."""
        lib = load_library()
        self = cls.__new__(cls)
        opts = sann_index_options_t(device, n_partitions, shard_id, n_shards)
        sp = sann_synth_params_t(n_tweets, now_ms, seed, n_clusters, index_cap, window_hours, max_clusters_per_tweet,
                                 mean_clusters, 0)
        h = C.c_void_p()
        _check(lib.sann_index_build_synthetic(C.byref(opts), C.byref(sp), C.byref(h)))
        self._h = h
        self.device = device
        self.cluster_ids = np.arange(1, n_clusters + 1, dtype=np.int32)
        self.list_offsets = None
        self.now_ms = now_ms
        self.synth_params = sp
        return self

    def exact_cosine_topk(self, emb_offsets, emb_cluster_ids, emb_scores, k: int):
        """Quality truth for a synthetic corpus: exact full-embedding cosine top-k (device brute force)."""
        lib = load_library()
        eo = np.ascontiguousarray(emb_offsets, np.int64)
        ec = np.ascontiguousarray(emb_cluster_ids, np.int32)
        es = np.ascontiguousarray(emb_scores, np.float64)
        nq = len(eo) - 1
        ids = np.zeros((nq, k), np.int64)
        cos = np.zeros((nq, k), np.float64)
        cnt = np.zeros(nq, np.int32)
        _check(lib.sann_synth_exact_cosine_topk(self.device, C.byref(self.synth_params), nq, _ptr(eo), _ptr(ec), _ptr(es), k,
                                                _ptr(ids), _ptr(cos), _ptr(cnt)))
        return ids, cos, cnt

    def tweet_embeddings(self, t0: int, n: int):
        """Full embeddings of synthetic tweets [t0, t0+n): (counts[n], cluster_ids[n,64], scores[n,64])."""
        lib = load_library()
        cnt = np.zeros(n, np.int32)
        cl = np.zeros((n, 64), np.int32)
        sc = np.zeros((n, 64), np.float64)
        _check(lib.sann_synth_tweet_embeddings(self.device, C.byref(self.synth_params), t0, n, _ptr(cnt), _ptr(cl), _ptr(sc)))
        return cnt, cl, sc

    def export_lists(self, cluster_ids):
        """CSR (cluster_ids, list_offsets, tweet_ids, scores) of the given clusters, copied back from
        the device in list order: what the oracle needs to answer a query on this index."""
        cids = np.unique(np.asarray(cluster_ids, np.int32))
        offs = [0]
        ts, ss = [], []
        for c in cids:
            t, s, _r = self.get_list(int(c))
            ts.append(t)
            ss.append(s)
            offs.append(offs[-1] + len(t))
        return (cids, np.array(offs, np.int64), np.concatenate(ts) if ts else np.empty(0, np.int64),
                np.concatenate(ss) if ss else np.empty(0, np.float64))

    @classmethod
    def from_map(cls, cluster_tweets: Mapping[int, Sequence[Tuple[int, float]]], **kw) -> "ClusterTweetIndex":
        cids = sorted(cluster_tweets)
        offs = [0]
        tids: List[int] = []
        scs: List[float] = []
        for c in cids:
            for t, s in cluster_tweets[c]:
                tids.append(t)
                scs.append(s)
            offs.append(len(tids))
        return cls(np.array(cids, np.int32), np.array(offs, np.int64), np.array(tids, np.int64), np.array(scs, np.float64), **kw)

    @property
    def handle(self) -> C.c_void_p:
        return self._h

    def info(self) -> sann_index_info_t:
        i = sann_index_info_t()
        _check(load_library().sann_index_info(self._h, C.byref(i)))
        return i

    def get_list(self, cluster_id: int):
        lib = load_library()
        n = C.c_int64()
        _check(lib.sann_index_get_list(self._h, cluster_id, 0, None, None, None, C.byref(n)))
        t = np.empty(n.value, np.int64)
        s = np.empty(n.value, np.float64)
        r = np.empty(n.value, np.int32)
        if n.value:
            _check(lib.sann_index_get_list(self._h, cluster_id, n.value, _ptr(t), _ptr(s), _ptr(r), C.byref(n)))
        return t, s, r

    def close(self):
        if self._h:
            load_library().sann_index_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _csr(rows: Sequence[Sequence], dtype) -> Tuple[np.ndarray, np.ndarray]:
    offs = np.zeros(len(rows) + 1, np.int64)
    for i, r in enumerate(rows):
        offs[i + 1] = offs[i] + len(r)
    flat = np.empty(int(offs[-1]), dtype)
    for i, r in enumerate(rows):
        flat[offs[i]:offs[i + 1]] = r
    return offs, flat


class QueryBatch:
    """A prepared batch (sann_batch_t): create once, run many times."""

    def __init__(self, index: ClusterTweetIndex, emb_offsets, emb_cluster_ids, emb_scores, configs, *, now_ms: int,
                 variant: Variant = Variant.original, source_tweet_ids=None, has_source_tweet=None,
                 scan_offsets=None, scan_cluster_ids=None):
        self.index = index
        self.variant = variant
        self._h = C.c_void_p()
        args = self._pack(emb_offsets, emb_cluster_ids, emb_scores, configs, source_tweet_ids, has_source_tweet,
                          scan_offsets, scan_cluster_ids)
        h = C.c_void_p()
        _check(load_library().sann_batch_create(index.handle, int(variant), int(now_ms), *args, C.byref(h)))
        self._h = h

    def _pack(self, emb_offsets, emb_cluster_ids, emb_scores, configs, source_tweet_ids, has_source_tweet, scan_offsets,
              scan_cluster_ids):
        self.nq = len(emb_offsets) - 1
        self._keep = [
            np.ascontiguousarray(emb_offsets, np.int64),
            np.ascontiguousarray(emb_cluster_ids, np.int32),
            np.ascontiguousarray(emb_scores, np.float64),
            None if source_tweet_ids is None else np.ascontiguousarray(source_tweet_ids, np.int64),
            None if has_source_tweet is None else np.ascontiguousarray(has_source_tweet, np.uint8),
            None if scan_offsets is None else np.ascontiguousarray(scan_offsets, np.int64),
            None if scan_cluster_ids is None else np.ascontiguousarray(scan_cluster_ids, np.int32),
        ]
        if isinstance(configs, SimClustersANNConfig):
            configs = [configs]
        self.configs = list(configs)
        carr = (sann_config_t * len(self.configs))(*[c.to_c() for c in self.configs])
        self.stride = max(1, max(min(max(c.maxNumResults, 0), 1000) for c in self.configs))
        k = self._keep
        return (self.nq, _ptr(k[0]), _ptr(k[1]), _ptr(k[2]), _ptr(k[3]), _ptr(k[4]), C.cast(carr, C.c_void_p),
                len(self.configs), _ptr(k[5]), _ptr(k[6]))

    def reset(self, emb_offsets, emb_cluster_ids, emb_scores, configs, *, now_ms: int, stream: int = 0,
              source_tweet_ids=None, has_source_tweet=None, scan_offsets=None, scan_cluster_ids=None):
        """sann_batch_reset: new queries into the same batch object (buffers kept); run() must follow on `stream`."""
        args = self._pack(emb_offsets, emb_cluster_ids, emb_scores, configs, source_tweet_ids, has_source_tweet,
                          scan_offsets, scan_cluster_ids)
        _check(load_library().sann_batch_reset(self._h, C.c_void_p(stream), int(now_ms), *args))

    def run(self, stream: int = 0):
        _check(load_library().sann_batch_run(self._h, C.c_void_p(stream)))

    def run_after(self, stream: int, after: "Optional[QueryBatch]", after_merge: bool = True):
        """sann_batch_run_after: this batch's unit kernel waits (on the GPU) for `after`'s merge (or unit) kernel."""
        _check(load_library().sann_batch_run_after(self._h, C.c_void_p(stream), after._h if after is not None else None,
                                                   1 if after_merge else 0))

    def finish(self, stream: int = 0):
        _check(load_library().sann_batch_finish(self._h, C.c_void_p(stream)))

    def results(self):
        ids = np.zeros((self.nq, self.stride), np.int64)
        scores = np.zeros((self.nq, self.stride), np.float64)
        counts = np.zeros(self.nq, np.int32)
        msz = np.zeros(self.nq, np.int32)
        _check(load_library().sann_batch_results(self._h, _ptr(ids), _ptr(scores), self.stride, _ptr(counts), _ptr(msz)))
        return ids, scores, counts, msz

    def device_results(self):
        p = [C.c_void_p() for _ in range(4)]
        st = C.c_int32()
        _check(load_library().sann_batch_device_results(self._h, *[C.byref(x) for x in p], C.byref(st)))
        return [x.value for x in p], st.value

    def bind_outputs(self, d_ids: int, d_scores: int, d_counts: int, d_map_sizes: int):
        _check(load_library().sann_batch_bind_outputs(self._h, C.c_void_p(d_ids), C.c_void_p(d_scores),
                                                      C.c_void_p(d_counts), C.c_void_p(d_map_sizes)))

    def bind_outputs_chunked(self, d_ids: int, d_scores: int, d_counts: int, d_map_sizes: int, queries_per_chunk: int,
                             chunk_pitch_bytes: int):
        """Outputs grouped by owner: query q goes to chunk q // queries_per_chunk (sann_batch_bind_outputs_chunked)."""
        _check(load_library().sann_batch_bind_outputs_chunked(self._h, C.c_void_p(d_ids), C.c_void_p(d_scores),
                                                              C.c_void_p(d_counts), C.c_void_p(d_map_sizes),
                                                              queries_per_chunk, chunk_pitch_bytes))

    def device_k(self) -> int:
        p = C.c_void_p()
        _check(load_library().sann_batch_device_k(self._h, C.byref(p)))
        return p.value

    def set_profiling(self, enable=True):
        """False / 0: off; 1: events around the unit kernel only; True / 2: descriptor, unit and merge kernels."""
        level = 2 if enable is True else int(enable)
        _check(load_library().sann_batch_set_profiling(self._h, level))

    def kernel_times(self):
        """(unit kernel ms total, merge kernel ms total, timed runs) since set_profiling."""
        a, b, n = C.c_double(), C.c_double(), C.c_int32()
        _check(load_library().sann_batch_kernel_times(self._h, C.byref(a), C.byref(b), C.byref(n)))
        return a.value, b.value, n.value

    def desc_time(self) -> float:
        d = C.c_double()
        _check(load_library().sann_batch_desc_time(self._h, C.byref(d)))
        return d.value

    def stats(self) -> sann_batch_stats_t:
        s = sann_batch_stats_t()
        _check(load_library().sann_batch_stats(self._h, C.byref(s)))
        return s

    def close(self):
        if self._h:
            load_library().sann_batch_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def pinned_array(shape, dtype) -> np.ndarray:
    """A numpy array over pinned host memory (sann_host_alloc): request / response buffers a front end keeps across
    calls.  The memory is released when the array (and every view of it) is garbage-collected."""
    lib = load_library()
    dt = np.dtype(dtype)
    n = int(np.prod(shape)) * dt.itemsize
    p = C.c_void_p()
    _check(lib.sann_host_alloc(max(n, 1), C.byref(p)))

    class _Owner:
        def __init__(self, ptr):
            self.ptr = ptr

        def __del__(self):
            try:
                lib.sann_host_free(C.c_void_p(self.ptr))
            except Exception:
                pass

    buf = (C.c_char * max(n, 1)).from_address(p.value)
    buf._owner = _Owner(p.value)  # the ctypes object keeps the allocation alive; numpy keeps the ctypes object
    return np.frombuffer(buf, dtype=dt, count=int(np.prod(shape))).reshape(shape)


class sann_legacy_config_t(C.Structure):
    _fields_ = [("max_num_results", C.c_int32), ("max_tweet_candidate_age_hours", C.c_int32), ("min_tweet_candidate_age_hours", C.c_int32),
                ("candidate_embedding_type", C.c_int32), ("min_score", C.c_double), ("enable_partial_normalization", C.c_int32),
                ("enable_heavy_ranking", C.c_int32), ("ranking_algorithm", C.c_int32), ("max_reranking_candidates", C.c_int32),
                ("max_top_tweets_per_cluster", C.c_int32), ("max_scan_clusters", C.c_int32)]


class sann_batcher_options_t(C.Structure):
    _fields_ = [("variant", C.c_int32), ("max_batch", C.c_int32), ("max_wait_us", C.c_int32), ("n_dispatchers", C.c_int32)]


class sann_batcher_stats_t(C.Structure):
    _fields_ = [("n_requests", C.c_int64), ("n_batches", C.c_int64), ("n_closed_full", C.c_int64),
                ("n_closed_by_deadline", C.c_int64), ("max_batch", C.c_int64)]


class MicroBatcher:
    """sann_batcher_t: the native micro-batching queue.  One request per call, from any number of threads -- the reference's
    calling pattern (SimClustersANNCandidateSource.scala:77-94) -- folded into batches for the GPU."""

    def __init__(self, index: ClusterTweetIndex, *, variant: Variant = Variant.original, max_batch: int = 0, max_wait_us: int = 0,
                 n_dispatchers: int = 0):
        self.index = index
        o = sann_batcher_options_t(int(variant), max_batch, max_wait_us, n_dispatchers)
        h = C.c_void_p()
        _check(load_library().sann_batcher_create(index.handle, C.byref(o), C.byref(h)))
        self._h = h

    def submit(self, cluster_ids, scores, config: "SimClustersANNConfig", *, now_ms: int, source_tweet_id: Optional[int] = None):
        """-> (ticket, (ids, scores, count, map_size)): the arrays are filled when wait(ticket) has returned."""
        cc = np.ascontiguousarray(cluster_ids, np.int32)
        ss = np.ascontiguousarray(scores, np.float64)
        cap = max(1, min(max(config.maxNumResults, 0), 1000))
        out = (np.zeros(cap, np.int64), np.zeros(cap, np.float64), np.zeros(1, np.int32), np.zeros(1, np.int32))
        cfg = config.to_c()
        t = C.c_int64()
        _check(load_library().sann_submit(self._h, int(now_ms), len(cc), _ptr(cc), _ptr(ss), int(source_tweet_id or 0),
                                          0 if source_tweet_id is None else 1, C.byref(cfg), cap, _ptr(out[0]), _ptr(out[1]),
                                          _ptr(out[2]), _ptr(out[3]), C.byref(t)))
        return t.value, out

    def wait(self, ticket: int):
        _check(load_library().sann_wait(self._h, ticket))

    def poll(self, ticket: int) -> bool:
        d = C.c_int32()
        _check(load_library().sann_poll(self._h, ticket, C.byref(d)))
        return bool(d.value)

    def get_tweet_candidates(self, cluster_ids, scores, config, *, now_ms: int, source_tweet_id: Optional[int] = None):
        """One request, blocking: (ids, scores, map_size) as ApproximateCosineSimilarity.apply would return them."""
        t, out = self.submit(cluster_ids, scores, config, now_ms=now_ms, source_tweet_id=source_tweet_id)
        self.wait(t)
        n = int(out[2][0])
        return out[0][:n], out[1][:n], int(out[3][0])

    def stats(self) -> sann_batcher_stats_t:
        st = sann_batcher_stats_t()
        _check(load_library().sann_batcher_stats(self._h, C.byref(st)))
        return st

    def close(self):
        if self._h:
            load_library().sann_batcher_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def get_tweet_candidates(index: ClusterTweetIndex, emb_offsets, emb_cluster_ids, emb_scores, configs, *, now_ms,
                         variant: Variant = Variant.original, source_tweet_ids=None, has_source_tweet=None,
                         scan_offsets=None, scan_cluster_ids=None, out=None):
    """sann_get_tweet_candidates: host arrays in, host arrays out, one call (what the JNI stub of INTEGRATION.md
    binds).  `out` = (ids[nq, stride], scores[nq, stride], counts[nq], map_sizes[nq]) to reuse response buffers
    (e.g. pinned_array); returns that tuple."""
    lib = load_library()
    if isinstance(configs, SimClustersANNConfig):
        configs = [configs]
    configs = list(configs)
    carr = (sann_config_t * len(configs))(*[c.to_c() for c in configs])
    eo = np.ascontiguousarray(emb_offsets, np.int64)
    ec = np.ascontiguousarray(emb_cluster_ids, np.int32)
    es = np.ascontiguousarray(emb_scores, np.float64)
    src = None if source_tweet_ids is None else np.ascontiguousarray(source_tweet_ids, np.int64)
    has = None if has_source_tweet is None else np.ascontiguousarray(has_source_tweet, np.uint8)
    so = None if scan_offsets is None else np.ascontiguousarray(scan_offsets, np.int64)
    sc = None if scan_cluster_ids is None else np.ascontiguousarray(scan_cluster_ids, np.int32)
    nq = len(eo) - 1
    stride = max(1, max(min(max(c.maxNumResults, 0), 1000) for c in configs))
    if out is None:
        out = (np.zeros((nq, stride), np.int64), np.zeros((nq, stride), np.float64), np.zeros(nq, np.int32), np.zeros(nq, np.int32))
    ids, scores, counts, msz = out
    assert ids.shape[0] >= nq and ids.shape[1] >= stride and ids.shape == scores.shape
    if np.ndim(now_ms) == 0:
        _check(lib.sann_get_tweet_candidates(index.handle, int(variant), int(now_ms), nq, _ptr(eo), _ptr(ec), _ptr(es), _ptr(src),
                                             _ptr(has), C.cast(carr, C.c_void_p), len(configs), _ptr(so), _ptr(sc), _ptr(ids),
                                             _ptr(scores), ids.shape[1], _ptr(counts), _ptr(msz)))
    else:  # one Time.now per query
        nows = np.ascontiguousarray(now_ms, np.int64)
        assert len(nows) == nq
        _check(lib.sann_get_tweet_candidates_at(index.handle, int(variant), _ptr(nows), nq, _ptr(eo), _ptr(ec), _ptr(es), _ptr(src),
                                                _ptr(has), C.cast(carr, C.c_void_p), len(configs), _ptr(so), _ptr(sc), _ptr(ids),
                                                _ptr(scores), ids.shape[1], _ptr(counts), _ptr(msz)))
    return out


class ApproximateCosineSimilarity:
    """The `gpu` value of flag approximate_cosine_similarity: same signature as
    `trait ApproximateCosineSimilarity.apply` (ApproximateCosineSimilarity.scala:26-36), with the
    cluster -> tweets map held on the device by handle instead of passed by value.

    sourceEmbedding         iterable of (clusterId, score)
    sourceEmbeddingId       tweet id when the source is InternalId.TweetId, else None
    clusterTweetsMapKeys    optional explicit key order of clusterTweetsMap (None = fetchCandidates)
    """

    def __init__(self, index: ClusterTweetIndex, variant: Variant = Variant.original, now_ms: Optional[int] = None):
        self.index = index
        self.variant = variant
        self.now_ms = now_ms

    def apply(self, sourceEmbedding: Iterable[Tuple[int, float]], sourceEmbeddingId: Optional[int],
              config: SimClustersANNConfig, candidateScoresStat: Callable[[int], None] = lambda _n: None,
              clusterTweetsMapKeys: Optional[Sequence[int]] = None, now_ms: Optional[int] = None) -> List[Tuple[int, float]]:
        res = self.apply_batch([list(sourceEmbedding)], [sourceEmbeddingId], config, candidateScoresStat,
                               None if clusterTweetsMapKeys is None else [clusterTweetsMapKeys], now_ms)
        return res[0]

    def apply_batch(self, embeddings, source_ids, config, candidateScoresStat=lambda _n: None, scan_keys=None,
                    now_ms: Optional[int] = None) -> List[List[Tuple[int, float]]]:
        import time

        now = now_ms if now_ms is not None else (self.now_ms if self.now_ms is not None else int(time.time() * 1000))
        eo, ec = _csr([[c for c, _ in e] for e in embeddings], np.int32)
        _, es = _csr([[s for _, s in e] for e in embeddings], np.float64)
        src = np.array([0 if s is None else s for s in source_ids], np.int64)
        has = np.array([0 if s is None else 1 for s in source_ids], np.uint8)
        so = sc = None
        if scan_keys is not None:
            so, sc = _csr(scan_keys, np.int32)
        qb = QueryBatch(self.index, eo, ec, es, config, now_ms=now, variant=self.variant, source_tweet_ids=src,
                        has_source_tweet=has, scan_offsets=so, scan_cluster_ids=sc)
        try:
            qb.run()
            qb.finish()
            ids, scores, counts, msz = qb.results()
        finally:
            qb.close()
        out = []
        for q in range(len(embeddings)):
            candidateScoresStat(int(msz[q]))
            out.append([(int(ids[q, i]), float(scores[q, i])) for i in range(counts[q])])
        return out


@dataclasses.dataclass
class LegacySimClustersANNConfig:
    """simclusters_v2/candidate_source/SimClustersANNCandidateSource.scala:214-260 (`SimClustersANNConfig`
    case class of the legacy in-process source).  Ages in hours; rankingAlgorithm is a
    representation_scorer.ScoringAlgorithm (score.thrift pair ids)."""

    maxNumResults: int = 200
    maxTweetCandidateAgeHours: int = 24
    minTweetCandidateAgeHours: int = 0
    minScore: float = 0.0
    candidateEmbeddingType: int = 0
    enablePartialNormalization: bool = True
    enableHeavyRanking: bool = False
    rankingAlgorithm: int = 2  # PairEmbeddingCosineSimilarity
    maxReRankingCandidates: int = 400
    maxTopTweetsPerCluster: int = 200
    maxScanClusters: int = 50


class LegacySimClustersANNCandidateSource:
    """fetchCandidates + reranking of the legacy source (SimClustersANNCandidateSource.scala:107-200) on the
    device: light ranking through the LEGACY variant of the C ABI, the optional heavy re-rank
    (HeavyRanker.UniformScoreStoreRanker, HeavyRanker.scala:32-77) through the resident pair scorer.
    Orders left open by the reference (HashMap / Map iteration under a stable sort) are fixed as
    everywhere else: score descending, tweet id ascending."""

    LOG_COSINE = 6  # ScoringAlgorithm.PairEmbeddingLogCosineSimilarity

    def __init__(self, index: ClusterTweetIndex, source_store=None, tweet_store=None, now_ms: Optional[int] = None):
        self.index, self.source_store, self.tweet_store, self.now_ms = index, source_store, tweet_store, now_ms

    def get(self, sourceEmbedding: Iterable[Tuple[int, float]], sourceEmbeddingId: Optional[int], config: LegacySimClustersANNConfig,
            source_internal_id: Optional[int] = None, now_ms: Optional[int] = None) -> List[Tuple[int, float]]:
        """sourceEmbeddingId: tweet id when the source is a tweet (parseTweetId), else None.
        source_internal_id: the id the heavy ranker looks the source embedding up under."""
        emb = list(sourceEmbedding)
        offs = np.array([0, len(emb)], np.int64)
        sid = source_internal_id if source_internal_id is not None else sourceEmbeddingId
        ids, sc, cnt = self.get_batch(offs, np.array([c for c, _ in emb], np.int32), np.array([s for _, s in emb], np.float64), config,
                                      source_tweet_ids=None if sourceEmbeddingId is None else [sourceEmbeddingId],
                                      source_internal_ids=[sid if sid is not None else 0], now_ms=now_ms)
        return list(zip(ids[0, :cnt[0]].tolist(), sc[0, :cnt[0]].tolist()))

    def get_batch(self, emb_offsets, emb_cluster_ids, emb_scores, config: LegacySimClustersANNConfig, *, source_tweet_ids=None,
                  source_internal_ids=None, now_ms: Optional[int] = None):
        """sann_heavy_rank: the whole legacy source for a batch of queries in one call -- light rank and (config.enableHeavyRanking)
        the heavy rank fused behind it on the device.  source_tweet_ids: per query a tweet id or None.  Returns
        (ids [nq, k], scores [nq, k], counts [nq])."""
        if config.enableHeavyRanking and (self.source_store is None or self.tweet_store is None):
            raise ValueError("heavy ranking needs the source and tweet embedding stores")
        lib = load_library()
        eo = np.ascontiguousarray(emb_offsets, np.int64)
        ec = np.ascontiguousarray(emb_cluster_ids, np.int32)
        es = np.ascontiguousarray(emb_scores, np.float64)
        nq = len(eo) - 1
        src = has = None
        if source_tweet_ids is not None:
            src = np.array([0 if t is None else t for t in source_tweet_ids], np.int64)
            has = np.array([0 if t is None else 1 for t in source_tweet_ids], np.uint8)
        sid = np.ascontiguousarray(source_internal_ids if source_internal_ids is not None else np.zeros(nq), np.int64)
        c = sann_legacy_config_t(config.maxNumResults, config.maxTweetCandidateAgeHours, config.minTweetCandidateAgeHours,
                                 config.candidateEmbeddingType, config.minScore, int(config.enablePartialNormalization),
                                 int(config.enableHeavyRanking), int(config.rankingAlgorithm), config.maxReRankingCandidates,
                                 config.maxTopTweetsPerCluster, config.maxScanClusters)
        k = max(1, config.maxNumResults)
        ids, sc, cnt = np.zeros((nq, k), np.int64), np.zeros((nq, k), np.float64), np.zeros(nq, np.int32)
        now = self.now_ms if now_ms is None else now_ms
        if now is None:
            raise ValueError("now_ms is required (Time.now is an explicit input)")
        _check(lib.sann_heavy_rank(self.index.handle, None if self.source_store is None else self.source_store._h,
                                   None if self.tweet_store is None else self.tweet_store._h, int(now), nq, _ptr(eo), _ptr(ec), _ptr(es),
                                   _ptr(src), _ptr(has), _ptr(sid), C.byref(c), _ptr(ids), _ptr(sc), k, _ptr(cnt)))
        return ids, sc, cnt
