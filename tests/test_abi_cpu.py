"""The C-ABI library loads on a box without a GPU and exports every symbol the header declares.
No compute calls here (argument validation only, which returns before any HIP call)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg.load_library()
    header = open(os.path.join(ROOT, "include", "simclusters_ann.h")).read()
    declared = set(re.findall(r"\b(sann_[a-z_0-9]+)\s*\(", header))
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/simclusters_ann.h but not exported"
    # the Python binding covers the whole header too
    assert declared == set(pkg.simclusters_ann.exported_symbols())
    rsx = open(os.path.join(ROOT, "include", "representation_scorer.h")).read()
    declared_rsx = set(re.findall(r"\b(rsx_[a-z_0-9]+)\s*\(", rsx))
    assert declared_rsx == set(pkg.representation_scorer.PROTOS)
    for name in sorted(declared_rsx):
        assert hasattr(lib, name), f"{name} declared in include/representation_scorer.h but not exported"
    hn = open(os.path.join(ROOT, "include", "hnsw_ann.h")).read()
    declared_hn = set(re.findall(r"\b(hnsw_[a-z_0-9]+)\s*\(", hn))
    assert declared_hn == set(pkg.hnsw_ann.PROTOS)
    for name in sorted(declared_hn):
        assert hasattr(lib, name), f"{name} declared in include/hnsw_ann.h but not exported"
    codec = open(os.path.join(ROOT, "include", "ann_codec.h")).read()
    declared_codec = set(re.findall(r"\b((?:sann_wire|hnsw_codec|ann_wire|ann_codec|hnsw_index)_[a-z_0-9]+)\s*\(", codec))
    assert declared_codec == set(pkg.ann_codec.PROTOS)
    for name in sorted(declared_codec):
        assert hasattr(lib, name), f"{name} declared in include/ann_codec.h but not exported"
    dann = open(os.path.join(ROOT, "include", "dense_ann.h")).read()
    declared_dann = set(re.findall(r"\b(dann_[a-z_0-9]+)\s*\(", dann))
    assert declared_dann == set(pkg.dense_ann.PROTOS)
    for name in sorted(declared_dann):
        assert hasattr(lib, name), f"{name} declared in include/dense_ann.h but not exported"


def test_version_and_error_paths(pkg):
    lib = pkg.load_library()
    assert b"gfx950" in lib.sann_version()
    h = C.c_void_p()
    assert lib.sann_index_build(None, 0, None, None, None, None, C.byref(h)) == 1  # SANN_EINVAL
    assert b"opts" in lib.sann_last_error()
    opts = pkg.simclusters_ann.sann_index_options_t(0, 3, 0, 1)  # 3 partitions: not a power of two
    assert lib.sann_index_build(C.byref(opts), 0, None, None, None, None, C.byref(h)) == 1
    assert lib.sann_batch_run(None, None) == 1
    assert lib.sann_batch_destroy(None) == 0
    assert lib.sann_index_destroy(None) == 0
    rl = pkg.representation_scorer._lib()
    assert rl.rsx_store_pair_scores(None, None, 2, 1, None, None, None, None) == 1  # RSX_EINVAL before any HIP call
    assert rl.rsx_store_destroy(None) == 0
    hl = pkg.hnsw_ann._lib()
    assert hl.hnsw_search(None, 1, None, 1, 1, None, None, None) == 1  # HNSW_EINVAL before any HIP call
    assert hl.hnsw_index_destroy(None) == 0
    dl = pkg.dense_ann._lib()
    assert dl.dann_index_build(0, 0, 10, 16, None, None, C.byref(h)) == 1  # DANN_EINVAL before any HIP call
    assert dl.dann_search(None, 1, None, 1, None, None, None) == 1
    assert b"null" in dl.dann_last_error()
    assert dl.dann_index_destroy(None) == 0


def test_config_struct_layout_matches_header(pkg):
    # 4+4+8+5*4+4 = 40 bytes, min_score at offset 8
    c = pkg.simclusters_ann.sann_config_t
    assert C.sizeof(c) == 40
    assert c.min_score.offset == 8
    assert c.ann_algorithm.offset == 32


def test_missing_library_fails_loudly(pkg, tmp_path):
    with pytest.raises(FileNotFoundError):
        pkg.load_library(str(tmp_path / "nope.so"))


def test_jni_glue_compiles_against_the_minimal_jni_header():
    """the JNI glue of INTEGRATION.md is real C, compile-checked against a hand-declared JNI
    subset (no JDK in this image) and link-checked against the library: every C-ABI symbol it calls exists.  All three
    services' glue files (tests/test_jni_cpu.py and test_jni_gpu.py also EXECUTE them through a hand-made JNIEnv)."""
    import subprocess, tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    jni = os.path.join(root, "the-algorithm_amd", "jni")
    want = {
        "simclusters_ann_jni.c": ("com_twitter_simclustersann_gpu_SannJni", ("indexBuild", "indexDestroy", "hostAlloc", "hostFree", "getTweetCandidates0", "heavyRank0", "batcherCreate", "batcherDestroy", "request0")),
        "representation_scorer_jni.c": ("com_twitter_representationscorer_gpu_RsxJni", ("storeBuild", "storeDestroy", "pairScores", "listScores")),
        "ann_jni.c": ("com_twitter_ann_gpu_AnnJni", ("denseIndexBuild", "denseIndexDestroy", "denseSearch", "hnswIndexBuildInsert",
                                                     "hnswIndexLoadDirectory", "hnswIndexDestroy", "hnswSearch", "composeShards")),
    }
    for fname, (cls, names) in want.items():
        src = os.path.join(jni, fname)
        subprocess.run(["gcc", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-DSANN_JNI_MINIMAL", "-I", jni, src], check=True)
        with tempfile.TemporaryDirectory() as d:
            out = os.path.join(d, "lib" + fname[:-2] + ".so")
            subprocess.run(["gcc", "-shared", "-fPIC", "-DSANN_JNI_MINIMAL", "-I", jni, src, "-o", out, "-L", os.path.join(root, "the-algorithm_amd"),
                            "-lsimclusters_amd", "-Wl,--no-undefined", "-Wl,--allow-shlib-undefined"], check=True)
            syms = subprocess.run(["nm", "-D", "--defined-only", out], check=True, capture_output=True, text=True).stdout
            for name in names:
                assert f"Java_{cls}_{name}" in syms


def test_the_process_not_the_library_sets_the_hardware_queue_count(pkg):
    """The library has no load-time side effects on the environment (it ran setenv in a constructor until round 3); the
    Python mirror, being the launcher here, exports the variable before it loads the library."""
    import subprocess, sys
    lib_path = pkg.simclusters_ann.LIB_PATH
    code = ("import ctypes, os; os.environ.pop('GPU_MAX_HW_QUEUES', None); ctypes.CDLL(%r); "
            "print(os.environ.get('GPU_MAX_HW_QUEUES'))" % lib_path)
    out = subprocess.run([sys.executable, "-c", code], check=True, capture_output=True, text=True).stdout.strip()
    assert out == "None"
    pkg.load_library()
    assert os.environ.get("GPU_MAX_HW_QUEUES") is not None


def test_finish_rejects_a_corrupted_status_block(pkg):
    """sann_batch_finish sizes host vectors and indexes host arrays from counts and ids that KERNELS wrote (status[0..1],
    the overflow-unit list, the inexact-query list).  They are range-checked first: garbage is an error code -- not a
    std::length_error / bad_alloc through an extern "C" frame (= abort() in the caller's JVM), not a wild host write.
    sann_debug_plan_slow_tail is that decision as pure host code."""
    import numpy as np
    lib = pkg.load_library()
    nu, nqr = C.c_int32(), C.c_int32()

    def plan(nq, P, over, inexact, n_over=None, n_inexact=None):
        o = np.asarray(over, np.int32)
        i = np.asarray(inexact, np.int32)
        return lib.sann_debug_plan_slow_tail(nq, P, len(o) if n_over is None else n_over, o.ctypes.data_as(C.c_void_p),
                                             len(i) if n_inexact is None else n_inexact, i.ctypes.data_as(C.c_void_p),
                                             C.byref(nu), C.byref(nqr))

    # a sane block: units 5 and 9 of query 0 / 1 (P = 8) overflowed, query 3 is unproven -> 2 + 8 units, 3 queries
    assert plan(4, 8, [5, 9], [3]) == 0 and (nu.value, nqr.value) == (10, 3)
    # an overflowed unit of an unproven query is not run twice; duplicates in the lists are tolerated
    assert plan(4, 8, [24, 24, 25], [3, 3]) == 0 and (nu.value, nqr.value) == (8, 1)
    assert plan(0, 8, [], []) == 0 and (nu.value, nqr.value) == (0, 0)
    # counts outside the batch's shape (what an uninitialised or overwritten status block looks like)
    for n_over, n_inexact in ((-1, 0), (0, -7), (33, 0), (0, 5), (0x7fffffff, 0), (-0x80000000, 0), (0, 0x7fffffff)):
        assert plan(4, 8, [], [], n_over=n_over, n_inexact=n_inexact) == 5  # SANN_EINTERNAL
        assert b"status block" in lib.sann_last_error()
    # ids outside the batch
    for over, inexact in (([32], []), ([-1], []), ([], [4]), ([], [-3]), ([0x7fffffff], []), ([], [-0x80000000])):
        assert plan(4, 8, over, inexact) == 5
        assert b"outside the batch" in lib.sann_last_error()


def test_batch_arguments_are_validated_before_any_device_call(pkg):
    """scan_cluster_ids without scan_offsets was accepted silently (and ignored)."""
    import numpy as np
    lib = pkg.load_library()
    sa = pkg.simclusters_ann
    # sann_batch_reset validates before it touches the device, but it needs a batch object, and that needs an index
    # (device memory): on a box without a GPU only the NULL-argument paths are reachable
    assert lib.sann_batch_reset(None, None, 0, 0, None, None, None, None, None, None, 1, None, None) == 1
    assert lib.sann_debug_plan_slow_tail(-1, 8, 0, None, 0, None, None, None) == 1
    assert lib.sann_debug_plan_slow_tail(4, 0, 0, None, 0, None, None, None) == 1
    assert lib.sann_debug_plan_slow_tail(4, 8, 1, None, 0, None, None, None) == 1
