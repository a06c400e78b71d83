"""Whole calls through the JNI glue (the-algorithm_amd/jni/*.c executed with the hand-made JNIEnv of tests/jni_harness.c):
what a JVM shim would get back equals the oracle / the direct C-ABI call, for all three services' surfaces."""
import ctypes as C

import numpy as np
import pytest

import _jni
from _jni import ANN, RSX, SANN

pytestmark = pytest.mark.gpu


def test_sann_through_the_glue_equals_the_oracle(pkg, oracle):
    co = pkg.corpus.make_corpus(20_000, 800, seed=31, index_cap=500)
    offs, cids, scs = pkg.corpus.make_queries(6, 800, seed=32)
    e = _jni.Env()
    h, msg, _ = e.call(SANN, "indexBuild", C.c_int64, 0, 32, 0, 1, e.array(co.cluster_ids.astype(np.int32)), e.array(co.list_offsets.astype(np.int64)),
                       e.array(co.tweet_ids.astype(np.int64)), e.array(co.scores.astype(np.float64)))
    assert msg is None and h != 0
    cfg = pkg.SimClustersANNConfig(maxNumResults=300, annAlgorithm=pkg.ScoringAlgorithm.CosineSimilarity)
    cbuf = np.frombuffer(bytes(cfg.to_c()), np.uint8).copy()
    nq, k = 6, 300
    ids, sc = np.zeros((nq, k), np.int64), np.zeros((nq, k))
    cnt, msz = np.zeros(nq, np.int32), np.zeros(nq, np.int32)
    rc, msg, _ = e.call(SANN, "getTweetCandidates0", C.c_int32, C.c_int64(h), 0, C.c_int64(co.now_ms), nq, 1, e.buffer(offs.astype(np.int64)),
                        e.buffer(cids.astype(np.int32)), e.buffer(scs.astype(np.float64)), None, None, e.buffer(cbuf), None, None,
                        e.buffer(ids), e.buffer(sc), k, e.buffer(cnt), e.buffer(msz))
    assert rc == 0 and msg is None
    for q in range(nq):
        o_ids, o_sc, o_msz = oracle.sann_query(cids[offs[q]:offs[q + 1]], scs[offs[q]:offs[q + 1]], None, cfg, co.now_ms, co.cluster_ids,
                                               co.list_offsets, co.tweet_ids, co.scores)
        assert cnt[q] == len(o_ids) and msz[q] == o_msz
        assert np.array_equal(ids[q, :cnt[q]], o_ids) and np.array_equal(sc[q, :cnt[q]].view(np.int64), o_sc.view(np.int64))
    # one request at a time through the native micro-batching queue (what a Finagle worker thread calls): the same answers
    mb, msg, _ = e.call(SANN, "batcherCreate", C.c_int64, C.c_int64(h), 0, 0, 200, 2)
    assert msg is None and mb != 0
    for q in range(nq):
        r_ids, r_sc, r_cm = np.zeros(k, np.int64), np.zeros(k), np.zeros(2, np.int32)
        rc, msg, _ = e.call(SANN, "request0", C.c_int32, C.c_int64(mb), C.c_int64(co.now_ms), e.array(cids[offs[q]:offs[q + 1]].astype(np.int32)),
                            e.array(scs[offs[q]:offs[q + 1]].astype(np.float64)), C.c_int64(0), C.c_uint8(0), e.buffer(cbuf), e.array(r_ids),
                            e.array(r_sc), e.array(r_cm))
        assert rc == 0 and msg is None
        assert r_cm[0] == cnt[q] and r_cm[1] == msz[q]
        assert np.array_equal(r_ids[:cnt[q]], ids[q, :cnt[q]]) and np.array_equal(r_sc[:cnt[q]].view(np.int64), sc[q, :cnt[q]].view(np.int64))
    rc, msg, _ = e.call(SANN, "request0", C.c_int32, C.c_int64(mb), C.c_int64(co.now_ms), e.array(cids[:3].astype(np.int32)),
                        e.array(scs[:3].astype(np.float64)), C.c_int64(0), C.c_uint8(0), e.buffer(cbuf), e.array(np.zeros(5, np.int64)),
                        e.array(np.zeros(5)), e.array(np.zeros(2, np.int32)))
    assert rc != 0 and "shorter" in msg
    e.call(SANN, "batcherDestroy", None, C.c_int64(mb))
    # pinned buffers through the glue
    b, msg, _ = e.call(SANN, "hostAlloc", C.c_void_p, C.c_int64(4096))
    assert b and msg is None
    e.call(SANN, "hostFree", None, C.c_void_p(b))
    e.call(SANN, "indexDestroy", None, C.c_int64(h))


def test_rsx_through_the_glue_equals_the_oracle(pkg, oracle):
    rng = np.random.default_rng(5)
    n = 200
    ids = np.sort(rng.choice(10_000, n, replace=False)).astype(np.int64)
    lens = rng.integers(1, 40, n)
    offs = np.zeros(n + 1, np.int64)
    offs[1:] = np.cumsum(lens)
    cl = np.concatenate([np.sort(rng.choice(500, l, replace=False)) for l in lens]).astype(np.int32)
    sc = np.exp(rng.normal(0, 1, offs[-1]))
    e = _jni.Env()
    st, msg, _ = e.call(RSX, "storeBuild", C.c_int64, 0, e.array(ids), e.array(offs), e.array(cl), e.array(sc))
    assert msg is None and st != 0
    a = rng.choice(ids, 64)
    b = rng.choice(ids, 64)
    a[3] = 10_001  # not in the store: None
    out, pres = np.zeros(64), np.zeros(64, np.int8)
    _, msg, _ = e.call(RSX, "pairScores", None, C.c_int64(st), C.c_int64(st), 2, e.array(a), e.array(b), e.array(out), e.array(pres))
    assert msg is None
    pos = {int(v): i for i, v in enumerate(ids)}
    for i in range(64):
        if i == 3:
            assert pres[i] == 0
            continue
        ia, ib = pos[int(a[i])], pos[int(b[i])]
        want = oracle.pair_score(2, cl[offs[ia]:offs[ia + 1]], sc[offs[ia]:offs[ia + 1]], cl[offs[ib]:offs[ib + 1]], sc[offs[ib]:offs[ib + 1]])
        assert pres[i] == 1 and np.float64(out[i]).view(np.int64) == np.float64(want).view(np.int64)
    out2, pres2 = np.zeros(64), np.zeros(64, np.int8)
    _, msg, _ = e.call(RSX, "listScores", None, C.c_int64(st), C.c_int64(st), 1, C.c_int64(int(ids[7])), e.array(b), e.array(out2), e.array(pres2))
    assert msg is None and pres2.all()
    for i in range(0, 64, 9):
        ib = pos[int(b[i])]
        want = oracle.pair_score(1, cl[offs[7]:offs[8]], sc[offs[7]:offs[8]], cl[offs[ib]:offs[ib + 1]], sc[offs[ib]:offs[ib + 1]])
        assert np.float64(out2[i]).view(np.int64) == np.float64(want).view(np.int64)
    e.call(RSX, "storeDestroy", None, C.c_int64(st))


def test_ann_through_the_glue_equals_the_direct_calls(pkg, tmp_path):
    rng = np.random.default_rng(6)
    n, d, nq, k = 3000, 64, 8, 10
    x = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal((nq, d)).astype(np.float32)
    m = pkg.dense_ann.DistanceMetric.Cosine
    e = _jni.Env()
    # exhaustive
    h, msg, _ = e.call(ANN, "denseIndexBuild", C.c_int64, 0, int(m), C.c_int64(n), d, e.buffer(x), None, C.c_uint8(0))
    assert msg is None and h
    dist, lab, cnt = np.zeros((nq, k), np.float32), np.zeros((nq, k), np.int64), np.zeros(nq, np.int32)
    _, msg, _ = e.call(ANN, "denseSearch", None, C.c_int64(h), nq, d, e.buffer(q), k, e.buffer(dist), e.buffer(lab), e.buffer(cnt))
    assert msg is None
    bf = pkg.dense_ann.BruteForceIndex.build(m, x)
    r_ids, r_dist, r_cnt = bf.search(q, k)
    assert np.array_equal(lab, r_ids) and np.array_equal(dist.view(np.int32), r_dist.view(np.int32)) and np.array_equal(cnt, r_cnt)
    bf.close()
    e.call(ANN, "denseIndexDestroy", None, C.c_int64(h))
    # HNSW: built through the glue, saved by the library, loaded again through the glue
    h, msg, _ = e.call(ANN, "hnswIndexBuildInsert", C.c_int64, 0, int(m), C.c_int64(n), d, e.buffer(x), None, 8, 40, C.c_int64(5), 1)
    assert msg is None and h
    _, msg, _ = e.call(ANN, "hnswSearch", None, C.c_int64(h), nq, d, e.buffer(q), k, 50, e.buffer(dist), e.buffer(lab), e.buffer(cnt))
    assert msg is None and (cnt == k).all()
    ix = pkg.hnsw_ann.Hnsw.build(m, x, max_m=8, ef_construction=40, seed=5)
    r_ids, r_dist, r_cnt = ix.search(q, k, 50)
    assert np.array_equal(lab, r_ids) and np.array_equal(dist.view(np.int32), r_dist.view(np.int32))
    # nThreads = 0: the deterministic device builder -- the graph the direct call builds, hence the same answers
    hg, msg, _ = e.call(ANN, "hnswIndexBuildInsert", C.c_int64, 0, int(m), C.c_int64(n), d, e.buffer(x), None, 8, 40, C.c_int64(5), 0)
    assert msg is None and hg
    lab_g, dist_g, cnt_g = np.zeros_like(lab), np.zeros_like(dist), np.zeros_like(cnt)
    _, msg, _ = e.call(ANN, "hnswSearch", None, C.c_int64(hg), nq, d, e.buffer(q), k, 50, e.buffer(dist_g), e.buffer(lab_g), e.buffer(cnt_g))
    assert msg is None
    ig = pkg.hnsw_ann.Hnsw.build(m, x, max_m=8, ef_construction=40, seed=5, gpu=True)
    g_ids, g_dist, g_cnt = ig.search(q, k, 50)
    ig.close()
    assert np.array_equal(lab_g, g_ids) and np.array_equal(dist_g.view(np.int32), g_dist.view(np.int32)) and np.array_equal(cnt_g, g_cnt)
    e.call(ANN, "hnswIndexDestroy", None, C.c_int64(hg))
    d_dir = str(tmp_path / "idx")
    pkg.ann_codec.save_directory(ix, 40, d_dir)
    ix.close()
    h2, msg, _ = e.call(ANN, "hnswIndexLoadDirectory", C.c_int64, 0, int(m), C.c_int64(n), d, e.buffer(x), None, e.string(d_dir))
    assert msg is None and h2
    lab2, dist2 = np.zeros_like(lab), np.zeros_like(dist)
    _, msg, _ = e.call(ANN, "hnswSearch", None, C.c_int64(h2), nq, d, e.buffer(q), k, 50, e.buffer(dist2), e.buffer(lab2), e.buffer(cnt))
    assert msg is None and np.array_equal(lab2, lab) and np.array_equal(dist2.view(np.int32), dist.view(np.int32))
    _, msg, _ = e.call(ANN, "hnswSearch", None, C.c_int64(h2), nq, 32, e.buffer(q), k, 50, e.buffer(dist2), e.buffer(lab2), e.buffer(cnt))
    assert "dimension" in msg
    e.call(ANN, "hnswIndexDestroy", None, C.c_int64(h))
    e.call(ANN, "hnswIndexDestroy", None, C.c_int64(h2))
