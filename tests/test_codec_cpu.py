"""include/ann_codec.h on the CPU.  Expected bytes are built HERE, field by field from the IDL texts with
struct.pack under Thrift's binary protocol (type byte, big-endian i16 field id, value; 0 ends a struct; binary = i32
length + bytes; list = element type, i32 size) -- a second, independent statement of the encoding:
  simclusters-ann/thrift/src/main/thrift/simClustersAnn.thrift:8-27,49-57, src/thrift/com/twitter/simclusters_v2/identifier.thrift
  ann/src/main/thrift/com/twitter/ann/common/ann_common.thrift:65-83,118-144
PARITY UNPINNED: the reference ships no serialised fixture and org.apache.thrift is not in its tree."""
import struct

import numpy as np
import pytest

BOOL, BYTE, DOUBLE, I16, I32, I64, STRING, STRUCT, MAP, SET, LIST = 2, 3, 4, 6, 8, 10, 11, 12, 13, 14, 15


def f(t, fid, payload):
    return struct.pack(">bh", t, fid) + payload


def i32(v): return struct.pack(">i", v)
def i64(v): return struct.pack(">q", v)
def dbl(v): return struct.pack(">d", v)
def binary(b): return i32(len(b)) + b
def st(*fields): return b"".join(fields) + b"\x00"


def config_bytes(c):
    return st(f(I32, 1, i32(c.maxNumResults)), f(DOUBLE, 2, dbl(c.minScore)), f(I32, 3, i32(c.candidateEmbeddingType)),
              f(I32, 4, i32(c.maxTopTweetsPerCluster)), f(I32, 5, i32(c.maxScanClusters)), f(I32, 6, i32(c.maxTweetCandidateAgeHours)),
              f(I32, 7, i32(c.minTweetCandidateAgeHours)), f(I32, 8, i32(int(c.annAlgorithm))))


def query_bytes(emb_type, model, internal_field, c):
    return st(f(STRUCT, 1, st(f(I32, 1, i32(emb_type)), f(I32, 2, i32(model)), f(STRUCT, 3, st(internal_field)))), f(STRUCT, 2, config_bytes(c)))


def test_query_encoding_and_round_trip(pkg):
    ac, sa = pkg.ann_codec, pkg.simclusters_ann
    cfg = sa.SimClustersANNConfig(maxNumResults=400, minScore=0.125, candidateEmbeddingType=3, maxTopTweetsPerCluster=800,
                                  maxScanClusters=50, maxTweetCandidateAgeHours=24, minTweetCandidateAgeHours=0,
                                  annAlgorithm=sa.ScoringAlgorithm.LogCosineSimilarity)
    # InternalId.userId = field 2, i64; EmbeddingType.FavBasedUserInterestedIn = 301; ModelVersion 20M_145K_2020 = 3
    q = ac.Query(301, 3, ac.InternalId(2, 1234567890123), cfg)
    want = query_bytes(301, 3, f(I64, 2, i64(1234567890123)), cfg)
    got = ac.encode_query(q)
    assert got == want
    back, used = ac.decode_query(got + b"trailing")
    assert used == len(want) and back == q
    # a tweet id source (field 1), a cluster id (field 5, i32) and a hashtag (field 4, string: kept raw)
    for internal, field in [(ac.InternalId(1, (1 << 62) + 5), f(I64, 1, i64((1 << 62) + 5))),
                            (ac.InternalId(5, 144428, I32), f(I32, 5, i32(144428))),
                            (ac.InternalId(4, 0, STRING, binary(b"#gfx950")), f(STRING, 4, binary(b"#gfx950"))),
                            # TopicId{1: entityId, 2: optional language} = field 8, a struct
                            (ac.InternalId(8, 0, STRUCT, st(f(I64, 1, i64(77)), f(STRING, 2, binary(b"en")))),
                             f(STRUCT, 8, st(f(I64, 1, i64(77)), f(STRING, 2, binary(b"en")))))]:
        qq = ac.Query(3, 3, internal, cfg)
        enc = ac.encode_query(qq)
        assert enc == query_bytes(3, 3, field, cfg)
        assert ac.decode_query(enc)[0] == qq


def test_decoder_skips_unknown_fields_and_reports_bad_input(pkg):
    ac, sa = pkg.ann_codec, pkg.simclusters_ann
    cfg = sa.SimClustersANNConfig()
    good = query_bytes(301, 3, f(I64, 2, i64(9)), cfg)
    # fields a newer IDL might add, of every container kind, in front of and between the known ones
    extra = (f(MAP, 40, struct.pack(">bbi", I32, STRING, 2) + i32(1) + binary(b"a") + i32(2) + binary(b"bc")) +
             f(LIST, 41, struct.pack(">bi", STRUCT, 1) + st(f(BOOL, 1, b"\x01"), f(I16, 2, struct.pack(">h", -3)))) +
             f(SET, 42, struct.pack(">bi", I64, 2) + i64(1) + i64(2)) + f(BYTE, 43, b"\x7f") + f(DOUBLE, 44, dbl(1.5)))
    sid = st(f(I32, 1, i32(301)), f(I32, 2, i32(3)), f(STRUCT, 3, st(f(I64, 2, i64(9)))))
    padded = extra + f(STRUCT, 1, sid) + extra + f(STRUCT, 2, config_bytes(cfg)[:-1] + f(I32, 99, i32(7)) + b"\x00") + b"\x00"
    assert ac.decode_query(padded)[0] == ac.decode_query(good)[0]
    # truncation anywhere is ANNC_ETRUNC (-2), never a crash or a silent success
    for cut in range(len(good)):
        with pytest.raises(ac.CodecError) as e:
            ac.decode_query(good[:cut])
        assert e.value.code == -2, cut
    # a required field missing, a known field with the wrong type, a union with two arms
    with pytest.raises(ac.CodecError) as e:
        ac.decode_query(st(f(STRUCT, 1, sid)))
    assert e.value.code == -3
    with pytest.raises(ac.CodecError) as e:
        ac.decode_query(st(f(STRUCT, 1, sid), f(STRUCT, 2, st(f(I64, 1, i64(400))))))
    assert e.value.code == -3
    two = st(f(I32, 1, i32(301)), f(I32, 2, i32(3)), f(STRUCT, 3, st(f(I64, 1, i64(1)), f(I64, 2, i64(2)))))
    with pytest.raises(ac.CodecError):
        ac.decode_query(st(f(STRUCT, 1, two), f(STRUCT, 2, config_bytes(cfg))))


def test_candidates_and_service_messages(pkg):
    ac, sa = pkg.ann_codec, pkg.simclusters_ann
    ids = np.array([1700000000000000001, -5, 0], np.int64)
    sc = np.array([0.75, -0.0, float("inf")])
    body = struct.pack(">bi", STRUCT, 3) + b"".join(st(f(I64, 1, i64(int(a))), f(DOUBLE, 2, dbl(float(b)))) for a, b in zip(ids, sc))
    assert ac.encode_candidates(ids, sc) == body
    gi, gs, used = ac.decode_candidates(body)
    assert used == len(body) and np.array_equal(gi, ids) and np.array_equal(gs.view(np.int64), sc.view(np.int64))
    assert ac.encode_candidates([], []) == struct.pack(">bi", STRUCT, 0)
    # strict TBinaryProtocol message: i32 (0x80010000 | type), method name, seqid; CALL args {1: Query}, REPLY result {0: list}
    cfg = sa.SimClustersANNConfig(maxNumResults=10)
    q = ac.Query(301, 3, ac.InternalId(2, 42), cfg)
    name = binary(b"getTweetCandidates")
    call = struct.pack(">I", 0x80010001) + name + i32(77) + st(f(STRUCT, 1, query_bytes(301, 3, f(I64, 2, i64(42)), cfg)))
    assert ac.encode_call(77, q) == call
    assert ac.decode_call(call) == (77, q, len(call))
    reply = struct.pack(">I", 0x80010002) + name + i32(77) + st(f(LIST, 0, body))
    assert ac.encode_reply(77, ids, sc) == reply
    seq, ri, rs, used = ac.decode_reply(reply)
    assert (seq, used) == (77, len(reply)) and np.array_equal(ri, ids)
    # the old (non-strict) header a default TBinaryProtocol reader must also accept: name, type byte, seqid
    old = name + b"\x01" + i32(5) + call[len(struct.pack(">I", 0) + name + i32(0)):]
    assert ac.decode_call(old)[0] == 5
    # a reply that carries a declared exception instead of field 0, another method, a reply where a call is expected
    exc = struct.pack(">I", 0x80010002) + name + i32(1) + st(f(STRUCT, 1, st(f(I32, 1, i32(2)))))
    for bad in (exc,):
        with pytest.raises(ac.CodecError):
            ac.decode_reply(bad)
    with pytest.raises(ac.CodecError):
        ac.decode_call(struct.pack(">I", 0x80010001) + binary(b"somethingElse") + i32(1) + b"\x00")
    with pytest.raises(ac.CodecError):
        ac.decode_call(reply)


def test_hnsw_index_files(pkg):
    ac = pkg.ann_codec
    key = lambda v: binary(i64(v))  # AnnInjections.LongInjection: 8 bytes big-endian
    m = ac.HnswInternalIndexMetadata(max_level=2, entry_point=0x0102030405060708, ef_construction=200, max_m=16, num_elements=3)
    want = st(f(I32, 1, i32(2)), f(STRING, 2, binary(bytes(range(1, 9)))), f(I32, 3, i32(200)), f(I32, 4, i32(16)), f(I32, 5, i32(3)))
    assert ac.encode_internal_metadata(m) == want and ac.decode_internal_metadata(want) == m
    empty = ac.HnswInternalIndexMetadata(-1, None, 200, 16, 0)  # no entry point: optional field 2 absent
    assert ac.encode_internal_metadata(empty) == st(f(I32, 1, i32(-1)), f(I32, 3, i32(200)), f(I32, 4, i32(16)), f(I32, 5, i32(0)))
    assert ac.decode_internal_metadata(ac.encode_internal_metadata(empty)) == empty
    # HnswIndexMetadata{dimension, distanceMetric (enum as i32: Cosine = 1), numElements}
    assert ac.encode_index_metadata(256, 1, 50_000_000) == st(f(I32, 1, i32(256)), f(I32, 2, i32(1)), f(I32, 3, i32(50_000_000)))
    assert ac.decode_index_metadata(ac.encode_index_metadata(64, 2, 7)) == (64, 2, 7)
    # graph file: HnswGraphEntry{level, key, neighbours} back to back, no count, no terminator
    lv = np.array([0, 0, 1, 0], np.int32); k = np.array([10, -3, 10, 1 << 40], np.int64)
    off = np.array([0, 2, 3, 3, 5], np.int64); nb = np.array([-3, 1 << 40, 10, 10, -3], np.int64)
    want = b"".join(st(f(I32, 1, i32(int(lv[e]))), f(STRING, 2, key(int(k[e]))),
                       f(LIST, 3, struct.pack(">bi", STRING, int(off[e + 1] - off[e])) + b"".join(key(int(x)) for x in nb[off[e]:off[e + 1]])))
                    for e in range(4))
    assert ac.encode_graph(lv, k, off, nb) == want
    glv, gk, goff, gnb = ac.decode_graph(want)
    assert np.array_equal(glv, lv) and np.array_equal(gk, k) and np.array_equal(goff, off) and np.array_equal(gnb, nb)
    assert [len(x) for x in ac.decode_graph(b"")] == [0, 0, 1, 0]
    with pytest.raises(ac.CodecError) as e:
        ac.decode_graph(want[:-7])  # the stream ends inside an entry: an error, not END_OF_FILE
    assert e.value.code == -2
    with pytest.raises(ac.CodecError):
        ac.decode_graph(st(f(I32, 1, i32(0)), f(STRING, 2, binary(b"short"))))  # not a long key


def test_nearest_neighbor_result(pkg):
    ac = pkg.ann_codec
    ids = np.array([5, 1 << 50], np.int64); d = np.array([0.25, 1.5], np.float32)
    nn = lambda i, arm, v: st(f(STRING, 1, binary(i64(i))), f(STRUCT, 2, st(f(STRUCT, arm, st(f(DOUBLE, 1, dbl(v)))))))
    for metric, arm in ((1, 1), (0, 2), (2, 3)):  # Cosine -> cosineDistance, L2 -> l2Distance, InnerProduct -> innerProductDistance
        want = st(f(LIST, 1, struct.pack(">bi", STRUCT, 2) + nn(5, arm, 0.25) + nn(1 << 50, arm, 1.5)))
        assert ac.encode_neighbor_result(metric, ids, d) == want
        gi, gd, ga, used = ac.decode_neighbor_result(want)
        assert used == len(want) and np.array_equal(gi, ids) and np.array_equal(gd, d.astype(np.float64)) and list(ga) == [arm, arm]
    bare = st(f(LIST, 1, struct.pack(">bi", STRUCT, 1) + st(f(STRING, 1, binary(i64(9))))))
    assert ac.encode_neighbor_result(1, [9]) == bare
    gi, gd, ga, _ = ac.decode_neighbor_result(bare)
    assert list(gi) == [9] and list(ga) == [0]


def test_random_round_trips(pkg):
    """Random values through every codec and back: what was encoded is what decodes, byte counts agree."""
    ac, sa = pkg.ann_codec, pkg.simclusters_ann
    rng = np.random.default_rng(2)
    for _ in range(200):
        cfg = sa.SimClustersANNConfig(maxNumResults=int(rng.integers(-5, 2000)), minScore=float(rng.normal()),
                                      candidateEmbeddingType=int(rng.integers(0, 400)), maxTopTweetsPerCluster=int(rng.integers(0, 5000)),
                                      maxScanClusters=int(rng.integers(0, 200)), maxTweetCandidateAgeHours=int(rng.integers(0, 200000)),
                                      minTweetCandidateAgeHours=int(rng.integers(0, 50)), annAlgorithm=sa.ScoringAlgorithm(int(rng.integers(1, 5))))
        kind = int(rng.choice([1, 2, 3, 5, 4, 10]))
        if kind in (1, 2, 3):
            iid = ac.InternalId(kind, int(rng.integers(-(1 << 62), 1 << 62)), I64)
        elif kind == 5:
            iid = ac.InternalId(5, int(rng.integers(0, 1 << 30)), I32)
        else:
            raw = bytes(rng.integers(0, 256, int(rng.integers(0, 40)), dtype=np.uint8))
            iid = ac.InternalId(kind, 0, STRING, binary(raw))
        q = ac.Query(int(rng.integers(1, 11000)), int(rng.integers(1, 7)), iid, cfg)
        seq = int(rng.integers(-(1 << 31), 1 << 31))
        enc = ac.encode_call(seq, q)
        assert ac.decode_call(enc) == (seq, q, len(enc))
        n = int(rng.integers(0, 50))
        ids = rng.integers(-(1 << 62), 1 << 62, n)
        sc = rng.normal(size=n)
        rep = ac.encode_reply(seq, ids, sc)
        s2, i2, c2, used = ac.decode_reply(rep)
        assert (s2, used) == (seq, len(rep)) and np.array_equal(i2, ids) and np.array_equal(c2.view(np.int64), sc.view(np.int64))
        ne = int(rng.integers(0, 30))
        lv = rng.integers(0, 5, ne).astype(np.int32)
        keys = rng.integers(-(1 << 62), 1 << 62, ne)
        lens = rng.integers(0, 33, ne)
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        nb = rng.integers(-(1 << 62), 1 << 62, int(off[-1]))
        g = ac.decode_graph(ac.encode_graph(lv, keys, off, nb))
        assert np.array_equal(g[0], lv) and np.array_equal(g[1], keys) and np.array_equal(g[2], off) and np.array_equal(g[3], nb)
