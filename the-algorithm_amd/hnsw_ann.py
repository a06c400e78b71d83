"""ctypes binding of include/hnsw_ann.h and a host-side mirror of the reference's HNSW queryable.

Reference (paths relative to /root/reference/ann/src/main/):
  scala/com/twitter/ann/hnsw/Hnsw.scala:95-147                  queryWithDistance(embedding, k, HnswParams(ef))
  scala/com/twitter/ann/hnsw/HnswCommon.scala / HnswParams      runtime params: ef
  java/com/twitter/ann/hnsw/HnswIndex.java:538-553               searchKnn
  java/com/twitter/ann/hnsw/HnswIndexIOUtil.java                 graph entries (HnswNode -> neighbours) + HnswMeta
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

from .dense_ann import DistanceMetric
from .simclusters_ann import load_library

PROTOS = {
    "hnsw_last_error": (C.c_char_p, []),
    "hnsw_index_build": (C.c_int, [C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64,
                                   C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "hnsw_index_build_insert": (C.c_int, [C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                          C.c_uint64, C.c_int32, C.POINTER(C.c_void_p)]),
    "hnsw_index_build_insert_levels": (C.c_int, [C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32,
                                                 C.c_int32, C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]),
    "hnsw_index_build_insert_gpu": (C.c_int, [C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                              C.c_uint64, C.c_int32, C.POINTER(C.c_void_p)]),
    "hnsw_index_build_insert_gpu_levels": (C.c_int, [C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                                     C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]),
    "hnsw_index_build_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "hnsw_index_graph_size": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    "hnsw_index_graph": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hnsw_index_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "hnsw_index_get_ids": (C.c_int, [C.c_void_p, C.c_void_p]),
    "hnsw_index_get_vectors": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]),
    "hnsw_index_destroy": (C.c_int, [C.c_void_p]),
    "hnsw_search": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hnsw_last_walk_counters": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "hnsw_last_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_float)]),
}


class HnswError(RuntimeError):
    pass


def _lib():
    lib = load_library()
    if not getattr(lib, "_hnsw_ready", False):
        for name, (res, args) in PROTOS.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        lib._hnsw_ready = True
    return lib


def _check(lib, rc: int) -> None:
    if rc != 0:
        raise HnswError(f"hnsw_ann error {rc}: {lib.hnsw_last_error().decode()}")


def _p(a):
    return None if a is None else a.ctypes.data


@dataclass
class HnswParams:
    """Runtime params of the reference's Hnsw queryable: ef (HnswCommon.scala)."""
    ef: int


class Hnsw:
    """Device-resident HNSW index (graph + vectors)."""

    def __init__(self, handle, metric, n, d, max_m):
        self._h, self.metric, self.n, self.d, self.max_m = handle, DistanceMetric(metric), n, d, max_m

    @classmethod
    def build(cls, metric: DistanceMetric, vectors: np.ndarray, ids: Optional[Sequence[int]] = None, *, max_m: int = 16,
              ef_construction: int = 200, seed: int = 1, n_threads: int = 1, device: int = 0,
              levels: Optional[Sequence[int]] = None, gpu: bool = False, batch: int = 0) -> "Hnsw":
        """Insert the vectors one by one with the reference's algorithm (HnswIndex.insert), on the host.
        levels: every item's level, instead of the seeded draw (int)(-ln U / ln maxM)."""
        lib = _lib()
        v = np.ascontiguousarray(vectors, np.float32)
        i = None if ids is None else np.ascontiguousarray(ids, np.int64)
        h = C.c_void_p()
        if gpu and levels is not None:
            lv = np.ascontiguousarray(levels, np.int32)
            if lv.shape != (v.shape[0],):
                raise ValueError("levels must hold one level per vector")
            _check(lib, lib.hnsw_index_build_insert_gpu_levels(device, int(metric), v.shape[0], v.shape[1], _p(v), _p(i), max_m,
                                                               ef_construction, _p(lv), batch, C.byref(h)))
            return cls(h, metric, v.shape[0], v.shape[1], max_m)
        if gpu:  # batched construction on the device (hnsw_index_build_insert_gpu)
            _check(lib, lib.hnsw_index_build_insert_gpu(device, int(metric), v.shape[0], v.shape[1], _p(v), _p(i), max_m,
                                                        ef_construction, seed, batch, C.byref(h)))
            return cls(h, metric, v.shape[0], v.shape[1], max_m)
        if levels is not None:
            lv = np.ascontiguousarray(levels, np.int32)
            if lv.shape != (v.shape[0],):
                raise ValueError("levels must hold one level per vector")
            _check(lib, lib.hnsw_index_build_insert_levels(device, int(metric), v.shape[0], v.shape[1], _p(v), _p(i), max_m,
                                                           ef_construction, _p(lv), n_threads, C.byref(h)))
            return cls(h, metric, v.shape[0], v.shape[1], max_m)
        _check(lib, lib.hnsw_index_build_insert(device, int(metric), v.shape[0], v.shape[1], _p(v), _p(i), max_m, ef_construction,
                                                seed, n_threads, C.byref(h)))
        return cls(h, metric, v.shape[0], v.shape[1], max_m)

    @classmethod
    def from_graph(cls, metric: DistanceMetric, vectors: np.ndarray, graph, ids: Optional[Sequence[int]] = None, *, max_m: int = 16,
                   device: int = 0) -> "Hnsw":
        """Load a graph: (entry_level, entry_item, entry_offsets, entry_neighbours, entry_point, max_level)."""
        lib = _lib()
        v = np.ascontiguousarray(vectors, np.float32)
        i = None if ids is None else np.ascontiguousarray(ids, np.int64)
        lv, it, off, nb, entry, max_level = graph
        lv = np.ascontiguousarray(lv, np.int32); it = np.ascontiguousarray(it, np.int64)
        off = np.ascontiguousarray(off, np.int64); nb = np.ascontiguousarray(nb, np.int64)
        h = C.c_void_p()
        _check(lib, lib.hnsw_index_build(device, int(metric), v.shape[0], v.shape[1] if v.ndim == 2 else 1, _p(v), _p(i), max_m,
                                         int(entry), int(max_level), len(lv), _p(lv), _p(it), _p(off), _p(nb), C.byref(h)))
        return cls(h, metric, v.shape[0], v.shape[1], max_m)

    def graph(self):
        lib = _lib()
        ne, nn, entry, ml = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int32()
        _check(lib, lib.hnsw_index_graph_size(self._h, C.byref(ne), C.byref(nn), C.byref(entry), C.byref(ml)))
        lv = np.zeros(ne.value, np.int32); it = np.zeros(ne.value, np.int64)
        off = np.zeros(ne.value + 1, np.int64); nb = np.zeros(max(nn.value, 1), np.int64)
        _check(lib, lib.hnsw_index_graph(self._h, _p(lv), _p(it), _p(off), _p(nb)))
        return lv, it, off, nb[:nn.value], entry.value, ml.value

    def stored_vectors(self) -> np.ndarray:
        out = np.empty((self.n, self.d), np.float32)
        lib = _lib()
        _check(lib, lib.hnsw_index_get_vectors(self._h, 0, self.n, _p(out)))
        return out

    def search(self, queries: np.ndarray, k: int, ef: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        lib = _lib()
        q = np.ascontiguousarray(queries, np.float32)
        if q.ndim == 1:
            q = q[None, :]
        if q.shape[1] != self.d:
            raise ValueError(f"query dimension {q.shape[1]} != index dimension {self.d}")
        nq = q.shape[0]
        dist = np.zeros((nq, k), np.float32); ids = np.zeros((nq, k), np.int64); cnt = np.zeros(nq, np.int32)
        _check(lib, lib.hnsw_search(self._h, nq, _p(q), k, ef, _p(dist), _p(ids), _p(cnt)))
        return ids, dist, cnt

    def queryWithDistance(self, embedding: np.ndarray, numOfNeighbors: int, runtimeParams: HnswParams) -> List[Tuple[int, float]]:
        """Hnsw.scala:118-147."""
        ids, dist, cnt = self.search(embedding, numOfNeighbors, runtimeParams.ef)
        return list(zip(ids[0, :cnt[0]].tolist(), dist[0, :cnt[0]].tolist()))

    def query(self, embedding: np.ndarray, numOfNeighbors: int, runtimeParams: HnswParams) -> List[int]:
        """Hnsw.scala:95-116."""
        return [i for i, _ in self.queryWithDistance(embedding, numOfNeighbors, runtimeParams)]

    def build_stats(self):
        """Counters of the device build: (rounds, additions a re-selection did not see, candidate-queue prunes, dropped candidates)."""
        v = [C.c_int64() for _ in range(4)]
        _check(_lib(), _lib().hnsw_index_build_stats(self._h, *[C.byref(x) for x in v]))
        return tuple(x.value for x in v)

    def last_stats(self):
        lib = _lib()
        a, b, c, d = C.c_int64(), C.c_int64(), C.c_int32(), C.c_float()
        _check(lib, lib.hnsw_last_stats(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        e, f = C.c_int64(), C.c_int64()
        _check(lib, lib.hnsw_last_walk_counters(self._h, C.byref(e), C.byref(f)))
        return dict(distance_evals=a.value, expansions=b.value, spilled_queries=c.value, kernel_ms=d.value, admissions=e.value,
                    largest_candidate_queue=f.value)

    def close(self):
        if self._h:
            _lib().hnsw_index_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
