"""An HNSW index directory in the reference's layout (SerializableHnsw.scala:170-190, HnswCommon.scala:16-20,42-50,
HnswIndexIOUtil.java:36-132): written from a built index, read back into a searchable one."""
import os
import struct

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_save_then_load_directory(pkg, tmp_path):
    ac = pkg.ann_codec
    m = pkg.dense_ann.DistanceMetric.Cosine
    rng = np.random.default_rng(21)
    x = rng.standard_normal((3000, 48)).astype(np.float32)
    keys = rng.permutation(np.arange(10_000, 10_000 + 3 * len(x), 3)).astype(np.int64)[:len(x)]
    ix = pkg.hnsw_ann.Hnsw.build(m, x, ids=keys, max_m=8, ef_construction=60, seed=9)
    d = str(tmp_path / "index")
    try:
        ac.save_directory(ix, 60, d)
        lv, it, off, nb, entry, max_level = ix.graph()
        # the layout HnswCommon.isValidHnswIndex checks
        for rel in ("_SUCCESS", "hnsw_index_metadata", "hnsw_internal_index/hnsw_internal_metadata", "hnsw_internal_index/hnsw_internal_graph"):
            assert os.path.exists(os.path.join(d, rel)), rel
        assert ac.decode_index_metadata(open(os.path.join(d, "hnsw_index_metadata"), "rb").read()) == (48, int(m), len(x))
        im = ac.decode_internal_metadata(open(os.path.join(d, "hnsw_internal_index/hnsw_internal_metadata"), "rb").read())
        assert im == ac.HnswInternalIndexMetadata(max_level, int(keys[entry]), 60, 8, len(lv))
        raw = open(os.path.join(d, "hnsw_internal_index/hnsw_internal_graph"), "rb").read()
        glv, gk, goff, gnb = ac.decode_graph(raw)
        assert np.array_equal(glv, lv) and np.array_equal(gk, keys[it]) and np.array_equal(goff, off) and np.array_equal(gnb, keys[nb])
        # first entry byte for byte: level, 8-byte big-endian key, list<binary>
        n0 = int(off[1] - off[0])
        first = (struct.pack(">bhi", 8, 1, int(lv[0])) + struct.pack(">bhiq", 11, 2, 8, int(keys[it[0]])) + struct.pack(">bhbi", 15, 3, 11, n0) +
                 b"".join(struct.pack(">iq", 8, int(keys[v])) for v in nb[:n0]) + b"\x00")
        assert raw[:len(first)] == first
        q = rng.standard_normal((40, 48)).astype(np.float32)
        want = ix.search(q, 20, 100)
        # rows handed over in another order than they were inserted in: keys, not positions, tie them to the graph
        perm = rng.permutation(len(x))
        back = ac.load_directory(m, x[perm], d, ids=keys[perm])
        try:
            got = back.search(q, 20, 100)
            assert np.array_equal(got[2], want[2]) and np.array_equal(got[0], want[0])
            assert np.array_equal(got[1].view(np.int32), want[1].view(np.int32))
        finally:
            back.close()
        with pytest.raises(ac.CodecError, match="Dimensions do not match"):
            ac.load_directory(m, x[:, :40], d, ids=keys)
        with pytest.raises(ac.CodecError, match="DistanceMetric do not match"):
            ac.load_directory(pkg.dense_ann.DistanceMetric.L2, x, d, ids=keys)
        with pytest.raises(ac.CodecError, match="has no vector"):
            ac.load_directory(m, x[:100], d, ids=keys[:100])
    finally:
        ix.close()


def test_empty_and_positional_indexes(pkg, tmp_path):
    ac = pkg.ann_codec
    m = pkg.dense_ann.DistanceMetric.L2
    x = np.random.default_rng(2).standard_normal((300, 16)).astype(np.float32)
    ix = pkg.hnsw_ann.Hnsw.build(m, x, max_m=4, ef_construction=20, seed=3)  # no ids: keys are positions
    d = str(tmp_path / "pos")
    try:
        ac.save_directory(ix, 20, d)
        back = ac.load_directory(m, x, d)
        try:
            q = x[:10] + 0.01
            a, b = ix.search(q, 5, 30), back.search(q, 5, 30)
            assert all(np.array_equal(u, v) for u, v in zip(a, b))
        finally:
            back.close()
    finally:
        ix.close()
    with pytest.raises(ac.CodecError):
        ac.load_directory(m, x, str(tmp_path / "missing"))
