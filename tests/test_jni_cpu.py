"""The JNI glue of the-algorithm_amd/jni/ EXECUTED on the CPU through a hand-made JNIEnv (tests/jni_harness.c): every size
the glue is responsible for is checked before the library reads through a pinned Java array or a direct buffer (ADVICE round 2:
the glue trusted sizes it never checked, and was only ever compiled).  No device call is reached here."""
import ctypes as C

import numpy as np

import _jni
from _jni import ANN, RSX, SANN


def test_sann_index_build_checks_the_arrays_it_pins(pkg):
    pkg.load_library()
    e = _jni.Env()
    cids = e.array(np.array([1, 2], np.int32))
    offs = e.array(np.array([0, 2, 5], np.int64))
    # tweetIds / scores shorter than listOffsets[n] = 5: a native over-read of a Java heap object, were it not refused
    r, msg, cls = e.call(SANN, "indexBuild", C.c_int64, 0, 32, 0, 1, cids, offs, e.array(np.arange(3, dtype=np.int64)), e.array(np.ones(5)))
    assert r == 0 and "shorter" in msg and "RuntimeException" in cls
    r, msg, _ = e.call(SANN, "indexBuild", C.c_int64, 0, 32, 0, 1, cids, offs, e.array(np.arange(5, dtype=np.int64)), e.array(np.ones(4)))
    assert r == 0 and "shorter" in msg
    r, msg, _ = e.call(SANN, "indexBuild", C.c_int64, 0, 32, 0, 1, cids, e.array(np.array([0, 2], np.int64)), e.array(np.arange(5, dtype=np.int64)), e.array(np.ones(5)))
    assert r == 0 and "listOffsets" in msg
    r, msg, _ = e.call(SANN, "indexBuild", C.c_int64, 0, 32, 0, 1, cids, e.array(np.array([-1, 2, 5], np.int64)), e.array(np.arange(5, dtype=np.int64)), e.array(np.ones(5)))
    assert r == 0 and "shorter" in msg
    r, msg, _ = e.call(SANN, "indexBuild", C.c_int64, 0, 32, 0, 1, None, offs, None, None)
    assert r == 0 and "null" in msg
    # sizes are fine, the library refuses the arguments itself (3 partitions): its message reaches the exception
    r, msg, _ = e.call(SANN, "indexBuild", C.c_int64, 0, 3, 0, 1, cids, offs, e.array(np.arange(5, dtype=np.int64)), e.array(np.ones(5)))
    assert r == 0 and "power of two" in msg


def _gtc(e, nq, stride, n_cfg=1, *, emb_offsets=None, emb_c=None, emb_s=None, src=None, has=None, scan_o=None, scan_c=None, out_q=None,
         cfg_bytes=40):
    out_q = nq if out_q is None else out_q
    eo = np.zeros(max(nq, 0) + 1, np.int64) if emb_offsets is None else emb_offsets
    return e.call(SANN, "getTweetCandidates0", C.c_int32, C.c_int64(0), 0, C.c_int64(0), nq, n_cfg, e.buffer(eo), e.buffer(emb_c), e.buffer(emb_s),
                  e.buffer(src), e.buffer(has), e.buffer(np.zeros(cfg_bytes, np.uint8)), e.buffer(scan_o), e.buffer(scan_c),
                  e.buffer(np.zeros(max(out_q, 1) * max(stride, 1), np.int64)), e.buffer(np.zeros(max(out_q, 1) * max(stride, 1))), stride,
                  e.buffer(np.zeros(max(out_q, 1), np.int32)), e.buffer(np.zeros(max(out_q, 1), np.int32)))


def test_sann_get_tweet_candidates_checks_every_buffer(pkg):
    pkg.load_library()
    e = _jni.Env()
    rc, msg, _ = _gtc(e, -1, 400)
    assert rc == 1 and "nq >= 0" in msg  # (a negative nq used to make every capacity check pass)
    rc, msg, _ = _gtc(e, 4, 0)
    assert rc == 1 and "outStride" in msg
    rc, msg, _ = _gtc(e, 4, 400, n_cfg=3)
    assert rc == 1 and "nConfigs" in msg
    rc, msg, _ = _gtc(e, 4, 400, out_q=2)
    assert rc == 1 and "smaller than the batch needs" in msg
    rc, msg, _ = _gtc(e, 4, 400, emb_offsets=np.zeros(3, np.int64))  # (nq + 1) offsets needed
    assert rc == 1 and "smaller than the batch needs" in msg
    rc, msg, _ = _gtc(e, 4, 400, cfg_bytes=39)
    assert rc == 1 and "smaller than the batch needs" in msg
    eo = np.array([0, 3, 6, 9, 12], np.int64)
    rc, msg, _ = _gtc(e, 4, 400, emb_offsets=eo, emb_c=np.zeros(11, np.int32), emb_s=np.zeros(12))
    assert rc == 1 and "embOffsets" in msg
    rc, msg, _ = _gtc(e, 4, 400, emb_offsets=eo, emb_c=np.zeros(12, np.int32), emb_s=None)
    assert rc == 1 and "embOffsets" in msg
    rc, msg, _ = _gtc(e, 4, 400, emb_offsets=eo, emb_c=np.zeros(12, np.int32), emb_s=np.zeros(12), src=np.zeros(4, np.int64))
    assert rc == 1 and "pairs" in msg
    rc, msg, _ = _gtc(e, 4, 400, emb_offsets=eo, emb_c=np.zeros(12, np.int32), emb_s=np.zeros(12), src=np.zeros(3, np.int64), has=np.zeros(4, np.uint8))
    assert rc == 1 and "smaller than the batch needs" in msg
    rc, msg, _ = _gtc(e, 4, 400, emb_offsets=eo, emb_c=np.zeros(12, np.int32), emb_s=np.zeros(12), scan_o=np.array([0, 1, 2, 3, 9], np.int64),
                      scan_c=np.zeros(8, np.int32))
    assert rc == 1 and "scanOffsets" in msg
    # everything in order: the call reaches the library, which refuses the NULL index
    rc, msg, _ = _gtc(e, 4, 400, emb_offsets=eo, emb_c=np.zeros(12, np.int32), emb_s=np.zeros(12))
    assert rc == 1 and "index is NULL" in msg


def test_rsx_glue_checks_sizes_and_the_algorithm(pkg):
    pkg.load_library()
    e = _jni.Env()
    ids = e.array(np.array([5, 9], np.int64))
    r, msg, _ = e.call(RSX, "storeBuild", C.c_int64, 0, ids, e.array(np.array([0, 2], np.int64)), e.array(np.zeros(4, np.int32)), e.array(np.zeros(4)))
    assert r == 0 and "offsets must have" in msg
    r, msg, _ = e.call(RSX, "storeBuild", C.c_int64, 0, ids, e.array(np.array([0, 2, 4], np.int64)), e.array(np.zeros(3, np.int32)), e.array(np.zeros(4)))
    assert r == 0 and "shorter" in msg
    a = e.array(np.array([1, 2, 3], np.int64))
    # unknown algorithm: IllegalArgumentException, as ScoreFacadeStore throws (ScoreFacadeStore.scala:25-51)
    _, msg, cls = e.call(RSX, "pairScores", None, C.c_int64(1), C.c_int64(1), 9, a, a, e.array(np.zeros(3)), e.array(np.zeros(3, np.int8)))
    assert "unknown" in msg and "IllegalArgumentException" in cls
    _, msg, cls = e.call(RSX, "pairScores", None, C.c_int64(1), C.c_int64(1), 2, a, e.array(np.zeros(2, np.int64)), e.array(np.zeros(3)), e.array(np.zeros(3, np.int8)))
    assert "aIds.length" in msg and "RuntimeException" in cls
    _, msg, _ = e.call(RSX, "pairScores", None, C.c_int64(1), C.c_int64(1), 2, a, a, e.array(np.zeros(2)), e.array(np.zeros(3, np.int8)))
    assert "aIds.length" in msg
    _, msg, _ = e.call(RSX, "pairScores", None, C.c_int64(0), C.c_int64(1), 2, a, a, e.array(np.zeros(3)), e.array(np.zeros(3, np.int8)))
    assert "null" in msg
    _, msg, _ = e.call(RSX, "listScores", None, C.c_int64(1), C.c_int64(1), 2, C.c_int64(7), a, e.array(np.zeros(3)), e.array(np.zeros(2, np.int8)))
    assert "candidateIds.length" in msg


def test_ann_glue_checks_capacities(pkg):
    pkg.load_library()
    e = _jni.Env()
    x = np.zeros((4, 64), np.float32)
    r, msg, _ = e.call(ANN, "denseIndexBuild", C.c_int64, 0, 1, C.c_int64(5), 64, e.buffer(x), None, C.c_uint8(0))
    assert r == 0 and "n x d floats" in msg
    r, msg, _ = e.call(ANN, "denseIndexBuild", C.c_int64, 0, 1, C.c_int64(4), 64, e.buffer(x), e.buffer(np.zeros(3, np.int64)), C.c_uint8(0))
    assert r == 0 and "n x d floats" in msg
    q = np.zeros((2, 64), np.float32)
    args = lambda k, dist, lab, cnt: (C.c_int64(1), 2, 64, e.buffer(q), k, e.buffer(dist), e.buffer(lab), e.buffer(cnt))  # noqa: E731
    _, msg, _ = e.call(ANN, "denseSearch", None, *args(10, np.zeros(19, np.float32), np.zeros(20, np.int64), np.zeros(2, np.int32)))
    assert "smaller than" in msg
    _, msg, _ = e.call(ANN, "denseSearch", None, *args(10, np.zeros(20, np.float32), np.zeros(20, np.int64), np.zeros(1, np.int32)))
    assert "smaller than" in msg
    _, msg, _ = e.call(ANN, "denseSearch", None, C.c_int64(0), 2, 64, e.buffer(q), 10, e.buffer(np.zeros(20, np.float32)), e.buffer(np.zeros(20, np.int64)),
                       e.buffer(np.zeros(2, np.int32)))
    assert "index" in msg
    _, msg, _ = e.call(ANN, "hnswSearch", None, C.c_int64(1), 2, 64, e.buffer(q), 10, 50, e.buffer(np.zeros(20, np.float32)), e.buffer(np.zeros(19, np.int64)),
                       e.buffer(np.zeros(2, np.int32)))
    assert "smaller than" in msg
    r, msg, _ = e.call(ANN, "hnswIndexLoadDirectory", C.c_int64, 0, 1, C.c_int64(4), 64, e.buffer(x), None, None)
    assert r == 0 and "directory" in msg
    r, msg, _ = e.call(ANN, "hnswIndexLoadDirectory", C.c_int64, 0, 1, C.c_int64(4), 64, e.buffer(x), None, e.string("/nonexistent/index/dir"))
    assert r == 0 and msg  # the codec's message (hnsw_index_metadata could not be read)


def test_java_binding_classes_match_the_compiled_glue():
    """the-algorithm_amd/jni/java/**/*.java are the classes a JVM shim loads (no JDK here: they are not compiled).  Every `native`
    method must have its JNI symbol in the glue with the same parameter list, type for type, and every JNIEXPORT function must be
    declared by a class: the two sides cannot drift apart unnoticed."""
    import glob
    import os
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    jmap = {"int": "jint", "long": "jlong", "boolean": "jboolean", "double": "jdouble", "float": "jfloat", "void": "void",
            "int[]": "jintArray", "long[]": "jlongArray", "double[]": "jdoubleArray", "float[]": "jfloatArray", "byte[]": "jbyteArray",
            "ByteBuffer": "jobject", "String": "jstring"}
    java = {}
    for path in glob.glob(os.path.join(root, "the-algorithm_amd", "jni", "java", "**", "*.java"), recursive=True):
        src = open(path).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        pkg = re.search(r"package\s+([\w.]+);", src).group(1)
        cls = re.search(r"public final class (\w+)", src).group(1)
        assert path.endswith(os.path.join(*pkg.split("."), cls + ".java")), "the file sits where its package says"
        for m in re.finditer(r"public static native\s+([\w\[\]]+)\s+(\w+)\s*\(([^)]*)\)\s*;", src, re.S):
            ret, name, args = m.groups()
            types = [" ".join(a.split()[:-1]) for a in args.split(",") if a.strip()]
            java["Java_" + pkg.replace(".", "_") + "_" + cls + "_" + name] = (jmap[ret], [jmap[t] for t in types])
    glue = {}
    for path in glob.glob(os.path.join(root, "the-algorithm_amd", "jni", "*.c")):
        src = re.sub(r"/\*.*?\*/", "", open(path).read(), flags=re.S)
        for m in re.finditer(r"JNIEXPORT\s+(\w+)\s+JNICALL\s+(Java_\w+)\s*\(([^)]*)\)", src, re.S):
            ret, name, args = m.groups()
            types = [a.split()[0] for a in args.split(",")]
            assert types[:2] == ["JNIEnv", "jclass"], name
            glue[name] = (ret, types[2:])
    assert len(java) >= 20 and set(java) == set(glue), sorted(set(java) ^ set(glue))
    for name in java:
        want = (java[name][0] if java[name][0] != "jobject" or glue[name][0] == "jobject" else "jobject", java[name][1])
        assert glue[name] == want, (name, glue[name], java[name])
    # and the symbols are really exported by the compiled glue (the harness links it)
    lib = _jni.load()
    for name in java:
        assert hasattr(lib, name), name
