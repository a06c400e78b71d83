"""ctypes mirror of include/ann_codec.h: the reference's Thrift (TBinaryProtocol) wire and on-disk formats either side of
the hot path -- simClustersAnn.thrift's Query / candidates / getTweetCandidates messages, ann_common.thrift's HNSW index
files and NearestNeighborResult.  Host only; nothing here touches the GPU except load_directory (it builds an index)."""
import ctypes as C
from dataclasses import dataclass
from typing import Optional, Sequence, Tuple

import numpy as np

from .simclusters_ann import SimClustersANNConfig, load_library, sann_config_t


class sann_wire_query_t(C.Structure):
    _fields_ = [("embedding_type", C.c_int32), ("model_version", C.c_int32), ("internal_id_kind", C.c_int32),
                ("internal_id_type", C.c_int32), ("internal_id_value", C.c_int64), ("internal_id_raw", C.c_void_p),
                ("internal_id_raw_len", C.c_int64), ("config", sann_config_t)]


class hnsw_internal_metadata_t(C.Structure):
    _fields_ = [("max_level", C.c_int32), ("has_entry_point", C.c_int32), ("entry_point", C.c_int64),
                ("ef_construction", C.c_int32), ("max_m", C.c_int32), ("num_elements", C.c_int32)]


_VP, _I64P, _I32P = C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int32)
PROTOS = {
    "ann_codec_last_error": (C.c_char_p, []),
    "sann_wire_encode_query": (C.c_int, [C.POINTER(sann_wire_query_t), _VP, C.c_int64, _I64P]),
    "sann_wire_decode_query": (C.c_int, [_VP, C.c_int64, C.POINTER(sann_wire_query_t), _I64P]),
    "sann_wire_encode_candidates": (C.c_int, [C.c_int32, _VP, _VP, _VP, C.c_int64, _I64P]),
    "sann_wire_decode_candidates": (C.c_int, [_VP, C.c_int64, C.c_int32, _VP, _VP, _I32P, _I64P]),
    "sann_wire_encode_call": (C.c_int, [C.c_int32, C.POINTER(sann_wire_query_t), _VP, C.c_int64, _I64P]),
    "sann_wire_decode_call": (C.c_int, [_VP, C.c_int64, _I32P, C.POINTER(sann_wire_query_t), _I64P]),
    "sann_wire_encode_reply": (C.c_int, [C.c_int32, C.c_int32, _VP, _VP, _VP, C.c_int64, _I64P]),
    "sann_wire_decode_reply": (C.c_int, [_VP, C.c_int64, _I32P, C.c_int32, _VP, _VP, _I32P, _I64P]),
    "hnsw_codec_encode_internal_metadata": (C.c_int, [C.POINTER(hnsw_internal_metadata_t), _VP, C.c_int64, _I64P]),
    "hnsw_codec_decode_internal_metadata": (C.c_int, [_VP, C.c_int64, C.POINTER(hnsw_internal_metadata_t)]),
    "hnsw_codec_encode_index_metadata": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _VP, C.c_int64, _I64P]),
    "hnsw_codec_decode_index_metadata": (C.c_int, [_VP, C.c_int64, _I32P, _I32P, _I32P]),
    "hnsw_codec_encode_graph": (C.c_int, [C.c_int64, _VP, _VP, _VP, _VP, _VP, C.c_int64, _I64P]),
    "hnsw_codec_decode_graph": (C.c_int, [_VP, C.c_int64, C.c_int64, C.c_int64, _VP, _VP, _VP, _VP, _I64P, _I64P]),
    "ann_wire_encode_neighbor_result": (C.c_int, [C.c_int32, C.c_int32, _VP, _VP, C.c_int32, _VP, C.c_int64, _I64P]),
    "ann_wire_decode_neighbor_result": (C.c_int, [_VP, C.c_int64, C.c_int32, _VP, _VP, _VP, _I32P, _I64P]),
    "hnsw_index_save_directory": (C.c_int, [_VP, C.c_int32, C.c_char_p]),
    "hnsw_index_load_directory": (C.c_int, [C.c_int32, C.c_int32, C.c_int64, C.c_int32, _VP, _VP, C.c_char_p, C.POINTER(_VP)]),
}

ANNC_ESPACE = -4


class CodecError(ValueError):
    def __init__(self, code: int, text: str):
        super().__init__(f"ann_codec error {code}: {text}")
        self.code = code


def _lib():
    lib = load_library()
    if not getattr(lib, "_codec_ready", False):
        for name, (res, args) in PROTOS.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        lib._codec_ready = True
    return lib


def _check(lib, rc: int) -> None:
    if rc != 0:
        raise CodecError(rc, lib.ann_codec_last_error().decode())


def _p(a):
    return None if a is None else a.ctypes.data


def _encode(call) -> bytes:
    """Two-pass: size with a NULL buffer, then write."""
    lib = _lib()
    n = C.c_int64()
    rc = call(None, 0, C.byref(n))
    if rc not in (0, ANNC_ESPACE):
        _check(lib, rc)
    buf = (C.c_uint8 * max(n.value, 1))()
    _check(lib, call(C.cast(buf, C.c_void_p), n.value, C.byref(n)))
    return bytes(buf[:n.value])


def _in(data: bytes):
    arr = np.frombuffer(data, np.uint8) if len(data) else np.zeros(1, np.uint8)
    return arr, (arr.ctypes.data if len(data) else arr.ctypes.data), len(data)


@dataclass
class InternalId:
    """identifier.thrift's InternalId union: `kind` = the field id (1 tweetId, 2 userId, 3 entityId, 5 clusterId hold
    `value`; the string / struct variants keep their encoded value in `raw`)."""
    kind: int
    value: int = 0
    thrift_type: int = 10
    raw: bytes = b""


@dataclass
class Query:
    """simClustersAnn.thrift Query = SimClustersEmbeddingId(embeddingType, modelVersion, internalId) + config."""
    embedding_type: int
    model_version: int
    internal_id: InternalId
    config: SimClustersANNConfig


def _to_c(q: Query):
    keep = np.frombuffer(q.internal_id.raw, np.uint8) if q.internal_id.raw else None
    c = sann_wire_query_t(q.embedding_type, q.model_version, q.internal_id.kind, q.internal_id.thrift_type, q.internal_id.value,
                          _p(keep), len(q.internal_id.raw), q.config.to_c())
    return c, keep


def _from_c(c: sann_wire_query_t) -> Query:
    raw = C.string_at(c.internal_id_raw, c.internal_id_raw_len) if c.internal_id_raw else b""
    return Query(c.embedding_type, c.model_version, InternalId(c.internal_id_kind, c.internal_id_value, c.internal_id_type, raw),
                 SimClustersANNConfig.from_c(c.config))


def encode_query(q: Query) -> bytes:
    lib = _lib()
    c, _keep = _to_c(q)
    return _encode(lambda b, cap, n: lib.sann_wire_encode_query(C.byref(c), b, cap, n))


def decode_query(data: bytes) -> Tuple[Query, int]:
    lib = _lib()
    arr, ptr, n = _in(data)
    c = sann_wire_query_t()
    used = C.c_int64()
    _check(lib, lib.sann_wire_decode_query(ptr, n, C.byref(c), C.byref(used)))
    return _from_c(c), used.value


def encode_call(seqid: int, q: Query) -> bytes:
    lib = _lib()
    c, _keep = _to_c(q)
    return _encode(lambda b, cap, n: lib.sann_wire_encode_call(seqid, C.byref(c), b, cap, n))


def decode_call(data: bytes) -> Tuple[int, Query, int]:
    lib = _lib()
    arr, ptr, n = _in(data)
    c = sann_wire_query_t()
    used = C.c_int64()
    seq = C.c_int32()
    _check(lib, lib.sann_wire_decode_call(ptr, n, C.byref(seq), C.byref(c), C.byref(used)))
    return seq.value, _from_c(c), used.value


def _cand_arrays(ids, scores):
    i = np.ascontiguousarray(ids, np.int64)
    s = np.ascontiguousarray(scores, np.float64)
    if i.shape != s.shape or i.ndim != 1:
        raise ValueError("ids and scores must be 1-d arrays of one length")
    return i, s


def encode_candidates(ids: Sequence[int], scores: Sequence[float]) -> bytes:
    lib = _lib()
    i, s = _cand_arrays(ids, scores)
    return _encode(lambda b, cap, n: lib.sann_wire_encode_candidates(len(i), _p(i), _p(s), b, cap, n))


def decode_candidates(data: bytes):
    lib = _lib()
    arr, ptr, n = _in(data)
    cnt = C.c_int32()
    used = C.c_int64()
    _check(lib, lib.sann_wire_decode_candidates(ptr, n, 0, None, None, C.byref(cnt), C.byref(used)))
    ids = np.zeros(max(cnt.value, 1), np.int64); sc = np.zeros(max(cnt.value, 1))
    _check(lib, lib.sann_wire_decode_candidates(ptr, n, cnt.value, _p(ids), _p(sc), C.byref(cnt), C.byref(used)))
    return ids[:cnt.value], sc[:cnt.value], used.value


def encode_reply(seqid: int, ids: Sequence[int], scores: Sequence[float]) -> bytes:
    lib = _lib()
    i, s = _cand_arrays(ids, scores)
    return _encode(lambda b, cap, n: lib.sann_wire_encode_reply(seqid, len(i), _p(i), _p(s), b, cap, n))


def decode_reply(data: bytes):
    lib = _lib()
    arr, ptr, n = _in(data)
    cnt = C.c_int32(); used = C.c_int64(); seq = C.c_int32()
    _check(lib, lib.sann_wire_decode_reply(ptr, n, C.byref(seq), 0, None, None, C.byref(cnt), C.byref(used)))
    ids = np.zeros(max(cnt.value, 1), np.int64); sc = np.zeros(max(cnt.value, 1))
    _check(lib, lib.sann_wire_decode_reply(ptr, n, C.byref(seq), cnt.value, _p(ids), _p(sc), C.byref(cnt), C.byref(used)))
    return seq.value, ids[:cnt.value], sc[:cnt.value], used.value


# ---- HNSW index files -------------------------------------------------------------------------------------------------
@dataclass
class HnswInternalIndexMetadata:
    max_level: int
    entry_point: Optional[int]
    ef_construction: int
    max_m: int
    num_elements: int


def encode_internal_metadata(m: HnswInternalIndexMetadata) -> bytes:
    lib = _lib()
    c = hnsw_internal_metadata_t(m.max_level, int(m.entry_point is not None), m.entry_point or 0, m.ef_construction, m.max_m, m.num_elements)
    return _encode(lambda b, cap, n: lib.hnsw_codec_encode_internal_metadata(C.byref(c), b, cap, n))


def decode_internal_metadata(data: bytes) -> HnswInternalIndexMetadata:
    lib = _lib()
    arr, ptr, n = _in(data)
    c = hnsw_internal_metadata_t()
    _check(lib, lib.hnsw_codec_decode_internal_metadata(ptr, n, C.byref(c)))
    return HnswInternalIndexMetadata(c.max_level, c.entry_point if c.has_entry_point else None, c.ef_construction, c.max_m, c.num_elements)


def encode_index_metadata(dimension: int, distance_metric: int, num_elements: int) -> bytes:
    lib = _lib()
    return _encode(lambda b, cap, n: lib.hnsw_codec_encode_index_metadata(dimension, int(distance_metric), num_elements, b, cap, n))


def decode_index_metadata(data: bytes) -> Tuple[int, int, int]:
    lib = _lib()
    arr, ptr, n = _in(data)
    d, m, e = C.c_int32(), C.c_int32(), C.c_int32()
    _check(lib, lib.hnsw_codec_decode_index_metadata(ptr, n, C.byref(d), C.byref(m), C.byref(e)))
    return d.value, m.value, e.value


def encode_graph(entry_level, entry_key, entry_offsets, entry_neighbours) -> bytes:
    lib = _lib()
    lv = np.ascontiguousarray(entry_level, np.int32); k = np.ascontiguousarray(entry_key, np.int64)
    off = np.ascontiguousarray(entry_offsets, np.int64); nb = np.ascontiguousarray(entry_neighbours, np.int64)
    if len(off) != len(lv) + 1 or len(k) != len(lv):
        raise ValueError("graph arrays disagree in length")
    return _encode(lambda b, cap, n: lib.hnsw_codec_encode_graph(len(lv), _p(lv), _p(k), _p(off), _p(nb) if len(nb) else None, b, cap, n))


def decode_graph(data: bytes):
    lib = _lib()
    arr, ptr, n = _in(data)
    ne, nn = C.c_int64(), C.c_int64()
    _check(lib, lib.hnsw_codec_decode_graph(ptr, n, 0, 0, None, None, None, None, C.byref(ne), C.byref(nn)))
    lv = np.zeros(max(ne.value, 1), np.int32); k = np.zeros(max(ne.value, 1), np.int64)
    off = np.zeros(ne.value + 1, np.int64); nb = np.zeros(max(nn.value, 1), np.int64)
    _check(lib, lib.hnsw_codec_decode_graph(ptr, n, ne.value, nn.value, _p(lv), _p(k), _p(off), _p(nb), C.byref(ne), C.byref(nn)))
    return lv[:ne.value], k[:ne.value], off, nb[:nn.value]


def encode_neighbor_result(distance_metric: int, ids, distances=None) -> bytes:
    lib = _lib()
    i = np.ascontiguousarray(ids, np.int64)
    d = None if distances is None else np.ascontiguousarray(distances, np.float32)
    return _encode(lambda b, cap, n: lib.ann_wire_encode_neighbor_result(int(distance_metric), len(i), _p(i), _p(d), int(d is not None), b, cap, n))


def decode_neighbor_result(data: bytes):
    lib = _lib()
    arr, ptr, n = _in(data)
    cnt = C.c_int32(); used = C.c_int64()
    _check(lib, lib.ann_wire_decode_neighbor_result(ptr, n, 0, None, None, None, C.byref(cnt), C.byref(used)))
    ids = np.zeros(max(cnt.value, 1), np.int64); dist = np.zeros(max(cnt.value, 1)); arms = np.zeros(max(cnt.value, 1), np.int32)
    _check(lib, lib.ann_wire_decode_neighbor_result(ptr, n, cnt.value, _p(ids), _p(dist), _p(arms), C.byref(cnt), C.byref(used)))
    return ids[:cnt.value], dist[:cnt.value], arms[:cnt.value], used.value


def save_directory(index, ef_construction: int, directory: str) -> None:
    """Hnsw -> <directory>/{hnsw_index_metadata, hnsw_internal_index/{hnsw_internal_metadata, hnsw_internal_graph}, _SUCCESS}"""
    lib = _lib()
    _check(lib, lib.hnsw_index_save_directory(index._h, ef_construction, directory.encode()))


def load_directory(metric, vectors: np.ndarray, directory: str, ids=None, *, device: int = 0):
    """The reverse: a searchable Hnsw over `vectors` from the directory's graph (keys = ids, or positions)."""
    from .hnsw_ann import Hnsw
    lib = _lib()
    v = np.ascontiguousarray(vectors, np.float32)
    i = None if ids is None else np.ascontiguousarray(ids, np.int64)
    h = C.c_void_p()
    _check(lib, lib.hnsw_index_load_directory(device, int(metric), v.shape[0], v.shape[1], _p(v), _p(i), directory.encode(), C.byref(h)))
    n, d, m, mm = C.c_int64(), C.c_int32(), C.c_int32(), C.c_int32()
    lib.hnsw_index_info.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.hnsw_index_info(h, C.byref(n), C.byref(d), C.byref(m), C.byref(mm))
    return Hnsw(h, metric, v.shape[0], v.shape[1], mm.value)
