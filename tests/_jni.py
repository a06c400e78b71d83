"""Builds tests/jni_harness.c together with the JNI glue of the-algorithm_amd/jni/ into one shared object and wraps the
fake JNIEnv (see jni_harness.c) for the tests."""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JNI = os.path.join(ROOT, "the-algorithm_amd", "jni")
GLUE = ["simclusters_ann_jni.c", "representation_scorer_jni.c", "ann_jni.c"]
_state = {}


def load():
    if "lib" in _state:
        return _state["lib"]
    d = tempfile.mkdtemp(prefix="jnih_")
    out = os.path.join(d, "libjni_harness.so")
    subprocess.run(["gcc", "-shared", "-fPIC", "-O1", "-Wall", "-Wextra", "-Werror", "-DSANN_JNI_MINIMAL", "-I", JNI,
                    os.path.join(ROOT, "tests", "jni_harness.c")] + [os.path.join(JNI, g) for g in GLUE] +
                   ["-o", out, "-L", os.path.join(ROOT, "the-algorithm_amd"), "-lsimclusters_amd",
                    "-Wl,-rpath," + os.path.join(ROOT, "the-algorithm_amd"), "-Wl,--no-undefined", "-Wl,--allow-shlib-undefined"], check=True)
    lib = C.CDLL(out)
    for name, res in (("jh_env", C.c_void_p), ("jh_array", C.c_void_p), ("jh_buffer", C.c_void_p), ("jh_string", C.c_void_p),
                      ("jh_exception", C.c_char_p), ("jh_exception_class", C.c_char_p), ("jh_buffer_address", C.c_void_p)):
        getattr(lib, name).restype = res
    lib.jh_array.argtypes = [C.c_void_p, C.c_longlong]
    lib.jh_buffer.argtypes = [C.c_void_p, C.c_longlong]
    lib.jh_string.argtypes = [C.c_char_p]
    lib.jh_buffer_address.argtypes = [C.c_void_p]
    _state["lib"] = lib
    return lib


class Env:
    """Keeps the numpy arrays behind the fake Java objects alive and gives typed access to the glue functions."""

    def __init__(self):
        self.lib = load()
        self.env = C.c_void_p(self.lib.jh_env())
        self.keep = []

    def array(self, a):  # a Java primitive array
        if a is None:
            return None
        a = np.ascontiguousarray(a)
        self.keep.append(a)
        return C.c_void_p(self.lib.jh_array(a.ctypes.data_as(C.c_void_p), a.shape[0]))

    def buffer(self, a, capacity_bytes=None):  # a direct ByteBuffer over a numpy array
        if a is None:
            return None
        a = np.ascontiguousarray(a)
        self.keep.append(a)
        return C.c_void_p(self.lib.jh_buffer(a.ctypes.data_as(C.c_void_p), a.nbytes if capacity_bytes is None else capacity_bytes))

    def string(self, s):
        b = s.encode()
        self.keep.append(b)
        return C.c_void_p(self.lib.jh_string(b))

    def call(self, cls, name, restype, *args):
        """Java_<cls>_<name>(env, NULL, args...); returns (result, exception message or None, exception class)."""
        fn = getattr(self.lib, f"Java_{cls}_{name}")
        fn.restype = restype
        self.lib.jh_clear()
        r = fn(self.env, None, *args)
        msg = self.lib.jh_exception().decode()
        return r, (msg or None), self.lib.jh_exception_class().decode()


SANN = "com_twitter_simclustersann_gpu_SannJni"
RSX = "com_twitter_representationscorer_gpu_RsxJni"
ANN = "com_twitter_ann_gpu_AnnJni"
