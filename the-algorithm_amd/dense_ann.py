"""ctypes binding of include/dense_ann.h and a host-side mirror of the reference's exhaustive index.

Reference (paths relative to /root/reference/ann/src/main/):
  scala/com/twitter/ann/brute_force/BruteForceIndex.scala:21-91   BruteForceIndex(metric), append, query,
                                                                  queryWithDistance
  scala/com/twitter/ann/common/Api.scala:24-51                    Queryable.query / queryWithDistance
  scala/com/twitter/ann/common/Api.scala:14-22                    NeighborWithDistance(neighbor, distance)
  scala/com/twitter/ann/common/ShardApi.scala:71-87               ComposedQueryable: concat shard answers,
                                                                  sort by distance, take k
  thrift/com/twitter/ann/common/ann_common.thrift:16-19           DistanceMetric
"""
from __future__ import annotations

import ctypes as C
import enum
from typing import List, Optional, Sequence, Tuple

import numpy as np

from .simclusters_ann import load_library


class DistanceMetric(enum.IntEnum):
    """ann_common.thrift:16-19."""

    L2 = 0
    Cosine = 1
    InnerProduct = 2


PROTOS = {
    "dann_last_error": (C.c_char_p, []),
    "dann_index_build": (C.c_int, [C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "dann_index_build_exact": (C.c_int, [C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "dann_index_build_synthetic": (C.c_int, [C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_uint64, C.POINTER(C.c_void_p)]),
    "dann_index_get_vectors": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]),
    "dann_index_destroy": (C.c_int, [C.c_void_p]),
    "dann_search": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dann_last_rounds": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "dann_last_timing": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "dann_compose_shards": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                      C.c_void_p]),
}


class DannError(RuntimeError):
    pass


def _lib():
    lib = load_library()
    if not getattr(lib, "_dann_ready", False):
        for name, (res, args) in PROTOS.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        lib._dann_ready = True
    return lib


def _check(lib, rc: int) -> None:
    if rc != 0:
        raise DannError(f"dense_ann error {rc}: {lib.dann_last_error().decode()}")


class BruteForceIndex:
    """Exhaustive index resident in HBM.  Unlike the reference's appendable in-memory queue
    (BruteForceIndex.scala:48-64) it is built in one call: the device layout is immutable."""

    def __init__(self, handle, metric: DistanceMetric, n: int, d: int):
        self._h, self.metric, self.n, self.d = handle, DistanceMetric(metric), n, d

    @classmethod
    def build(cls, metric: DistanceMetric, vectors: np.ndarray, ids: Optional[Sequence[int]] = None, *, device: int = 0,
              exact: bool = False):
        """exact=True keeps the fp32 rows and re-ranks every search's survivors in fp32 (dann_index_build_exact)."""
        lib = _lib()
        v = np.ascontiguousarray(vectors, np.float32)
        if v.ndim != 2:
            raise ValueError("vectors must be [n, d]")
        idp = None
        if ids is not None:
            idp = np.ascontiguousarray(ids, np.int64)
            if idp.shape != (v.shape[0],):
                raise ValueError("one id per vector")
        h = C.c_void_p()
        fn = lib.dann_index_build_exact if exact else lib.dann_index_build
        _check(lib, fn(device, int(metric), v.shape[0], v.shape[1], v.ctypes.data,
                       idp.ctypes.data if idp is not None else None, C.byref(h)))
        return cls(h, metric, v.shape[0], v.shape[1])

    @classmethod
    def synthetic(cls, metric: DistanceMetric, n: int, d: int, *, seed: int = 1, device: int = 0):
        lib = _lib()
        h = C.c_void_p()
        _check(lib, lib.dann_index_build_synthetic(device, int(metric), n, d, seed, C.byref(h)))
        return cls(h, metric, n, d)

    def stored_vectors(self, i0: int = 0, n: Optional[int] = None) -> np.ndarray:
        """The vectors as stored (fp16-rounded; unit length for Cosine), in position (= id) order."""
        n = self.n - i0 if n is None else n
        out = np.empty((n, self.d), np.float32)
        lib = _lib()
        _check(lib, lib.dann_index_get_vectors(self._h, i0, n, out.ctypes.data))
        return out

    def last_rounds(self) -> int:
        r = C.c_int32()
        lib = _lib()
        _check(lib, lib.dann_last_rounds(self._h, C.byref(r)))
        return r.value

    def search(self, queries: np.ndarray, k: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """Batched queryWithDistance: (ids [nq, k], distances [nq, k], counts [nq]), ascending by distance."""
        lib = _lib()
        q = np.ascontiguousarray(queries, np.float32)
        if q.ndim == 1:
            q = q[None, :]
        if q.shape[1] != self.d:
            raise ValueError(f"query dimension {q.shape[1]} != index dimension {self.d}")
        nq = q.shape[0]
        dist = np.empty((nq, k), np.float32)
        ids = np.empty((nq, k), np.int64)
        cnt = np.empty(nq, np.int32)
        _check(lib, lib.dann_search(self._h, nq, q.ctypes.data, k, dist.ctypes.data, ids.ctypes.data, cnt.ctypes.data))
        return ids, dist, cnt

    def query(self, embedding: np.ndarray, numOfNeighbors: int) -> List[int]:
        """Queryable.query (Api.scala:33-37)."""
        ids, _, cnt = self.search(embedding, numOfNeighbors)
        return ids[0, :cnt[0]].tolist()

    def queryWithDistance(self, embedding: np.ndarray, numOfNeighbors: int) -> List[Tuple[int, float]]:
        """Queryable.queryWithDistance (Api.scala:45-50; BruteForceIndex.scala:66-91)."""
        ids, dist, cnt = self.search(embedding, numOfNeighbors)
        return list(zip(ids[0, :cnt[0]].tolist(), dist[0, :cnt[0]].tolist()))

    def last_timing_ms(self) -> Tuple[float, float, float]:
        a, b, s = C.c_float(), C.c_float(), C.c_float()
        lib = _lib()
        _check(lib, lib.dann_last_timing(self._h, C.byref(a), C.byref(b), C.byref(s)))
        return a.value, b.value, s.value

    def close(self) -> None:
        if self._h:
            _lib().dann_index_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def compose(shard_results: Sequence[Tuple[np.ndarray, np.ndarray, np.ndarray]], k: int):
    """ComposedQueryable.queryWithDistance (ShardApi.scala:71-87) for batched answers: concatenate every
    shard's (ids, distances, counts), order by (distance, id), keep k -- dann_compose_shards in the library."""
    lib = _lib()
    nq = shard_results[0][0].shape[0]
    k_in = max(r[0].shape[1] for r in shard_results)

    def pad(a, dtype):
        out = np.zeros((nq, k_in), dtype)
        out[:, :a.shape[1]] = a
        return out

    s_ids = np.ascontiguousarray(np.stack([pad(r[0], np.int64) for r in shard_results]))
    s_dist = np.ascontiguousarray(np.stack([pad(r[1], np.float32) for r in shard_results]))
    s_cnt = np.ascontiguousarray(np.stack([np.asarray(r[2], np.int32) for r in shard_results]))
    ids = np.zeros((nq, k), np.int64)
    dist = np.zeros((nq, k), np.float32)
    cnt = np.zeros(nq, np.int32)
    def ptr(a):
        return a.ctypes.data_as(C.c_void_p)

    _check(lib, lib.dann_compose_shards(len(shard_results), nq, k_in, ptr(s_ids), ptr(s_dist), ptr(s_cnt), k, ptr(ids), ptr(dist), ptr(cnt)))
    return ids, dist, cnt
