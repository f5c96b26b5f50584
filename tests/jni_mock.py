"""bindings/jni/fmx_jni.c RUN without a JVM: the glue and tests/jni_stub/mock_jnienv.c (a mock JNIEnv with copy-always array
semantics) built into tests/libfmx_jni_mock.so against the test-local jni.h, linked to the product's libfmx.so, and its Java_*
entry points called the way a JVM calls them.  Test infrastructure only; what a JVM would add (class loading, the Java class
itself) stays untested here — there is no JDK in the image."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
_PREFIX = "Java_com_dynatrace_fm_gpu_GpuFmIndex_"
BYTES, CHARS, INTS, LONGS = 2, 3, 4, 5
_NP = {BYTES: np.int8, CHARS: np.uint16, INTS: np.int32, LONGS: np.int64}

jlong, jint, jboolean, jchar, ref = C.c_int64, C.c_int32, C.c_uint8, C.c_uint16, C.c_void_p

# name -> (return type, argument types after (JNIEnv *, jclass)) — the `native` declarations of GpuFmIndex.java
SIGNATURES = {
    "nativeLoad": (jlong, [ref, jint]),
    "nativeSave": (ref, [jlong, jboolean]),
    "nativeSavedOrderModelled": (jboolean, [jlong]),
    "nativeBuild": (jlong, [ref, jint, jboolean, jint, jboolean]),
    "nativeFree": (None, [jlong]),
    "nativeInputLength": (jint, [jlong]),
    "nativeAlphabetLength": (jint, [jlong]),
    "nativeCountBatch": (None, [jlong, ref, ref, jint, ref, ref]),
    "nativeLocateBatch": (None, [jlong, ref, ref, jint, jint, ref, jint, ref, ref]),
    "nativeExtractBatch": (None, [jlong, ref, ref, jint, ref, jint, jint, ref, ref]),
    "nativeExtractBoundaryBatch": (None, [jlong, ref, jint, jchar, jint, ref, jint, jint, ref, ref, ref]),
    "nativeLocatePipeline": (None, [jlong, ref, ref, jint, jint, jint, jchar, jint, ref, ref, ref, ref, ref, ref, ref]),
    "nativeCountSegments": (None, [ref, ref, ref, jint, ref, ref]),
    "nativeReplicate": (ref, [jlong, ref]),
    "nativeDeviceOf": (jint, [jlong]),
    "nativeCountBatchMulti": (None, [ref, ref, ref, jint, ref, ref]),
    "nativeLocateBatchMulti": (None, [ref, ref, ref, jint, jint, ref, jint, ref, ref]),
    "nativeExtractBatchMulti": (None, [ref, ref, ref, jint, ref, jint, jint, ref, ref]),
    "nativeExtractBoundaryBatchMulti": (None, [ref, ref, jint, jchar, jint, ref, jint, jint, ref, ref, ref]),
    "nativeCountLocateSegmentsMulti": (None, [ref, jint, jint, ref, ref, ref, jint, jint, ref, ref, ref, ref]),
}


class JavaException(Exception):
    """what the glue left pending when a native method returned"""

    def __init__(self, cls, message):
        super().__init__("%s: %s" % (cls, message))
        self.cls, self.message = cls, message


def _build():
    import index4j_amd._lib as il

    so = os.path.join(_HERE, "libfmx_jni_mock.so")
    srcs = [os.path.join(_ROOT, "bindings", "jni", "fmx_jni.c"), os.path.join(_HERE, "jni_stub", "mock_jnienv.c"),
            os.path.join(_HERE, "jni_stub", "jni.h"), os.path.join(_ROOT, "include", "fmx.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        libdir, libname = os.path.split(os.path.abspath(il.LIB_PATH))
        subprocess.check_call(["gcc", "-O1", "-g", "-shared", "-fPIC", "-Wall", "-Werror=implicit-function-declaration",
                               "-I", os.path.join(_HERE, "jni_stub"), "-I", os.path.join(_ROOT, "include"), srcs[0], srcs[1],
                               "-L", libdir, "-l:" + libname, "-Wl,-rpath," + libdir, "-o", so])
    return so


class MockJvm:
    """one per process (the mock keeps its pending exception and its counters in globals)"""
    _instance = None

    def __new__(cls):
        if cls._instance is None:
            cls._instance = super().__new__(cls)
            cls._instance._init()
        return cls._instance

    def _init(self):
        import index4j_amd  # the product library first: the glue's libfmx.so must be the one the package loaded

        self.ia = index4j_amd
        L = self.L = C.CDLL(_build())
        L.mock_env.restype = ref
        L.mock_new_array.restype = ref
        L.mock_new_array.argtypes = [C.c_int, jint]
        L.mock_array_data.restype = ref
        L.mock_array_data.argtypes = [ref]
        L.mock_array_length.argtypes = [ref]
        L.mock_array_kind.argtypes = [ref]
        L.mock_free_array.argtypes = [ref]
        L.mock_free_array.restype = None
        L.mock_exception_class.restype = C.c_char_p
        L.mock_exception_message.restype = C.c_char_p
        L.mock_first_violation.restype = C.c_char_p
        L.mock_exception_clear.restype = None
        L.mock_reset.restype = None
        L.mock_counters.argtypes = [C.POINTER(C.c_int)] * 3
        L.mock_counters.restype = None
        self.env = L.mock_env()
        for name, (res, args) in SIGNATURES.items():
            f = getattr(L, _PREFIX + name)
            f.restype = res
            f.argtypes = [ref, ref] + args
        self.exported = [n for n in SIGNATURES]

    # ---- "Java heap" ----
    def new(self, kind, length):
        a = self.L.mock_new_array(kind, length)
        assert a, "mock_new_array"
        return a

    def array(self, kind, values):
        """a Java array holding `values`"""
        v = np.ascontiguousarray(values, dtype=_NP[kind])
        a = self.new(kind, v.size)
        self.view(a)[:] = v.ravel()
        return a

    def chars(self, values):
        return self.array(CHARS, values)

    def ints(self, values):
        return self.array(INTS, values)

    def longs(self, values):
        return self.array(LONGS, values)

    def bytes_(self, data):
        return self.array(BYTES, np.frombuffer(bytes(data), np.int8))

    def view(self, a):
        """the array's own storage (what Java code reads after the call), as numpy"""
        kind, n = self.L.mock_array_kind(a), self.L.mock_array_length(a)
        dt = np.dtype(_NP[kind])
        if n == 0:
            return np.zeros(0, dt)
        buf = (C.c_char * (n * dt.itemsize)).from_address(self.L.mock_array_data(a))
        return np.frombuffer(buf, dt)

    def free(self, *arrays):
        for a in arrays:
            self.L.mock_free_array(a)

    # ---- calls ----
    def call(self, name, *args):
        """the native method; raises JavaException if it left one pending.  Also holds the glue to the JNI contract: every
        Get<T>ArrayElements released, nothing but the exception calls made while one is pending."""
        self.L.mock_reset()
        before = self.L.mock_outstanding()
        out = getattr(self.L, _PREFIX + name)(self.env, None, *args)
        assert self.L.mock_violations() == 0, self.L.mock_first_violation().decode()
        assert self.L.mock_outstanding() == before, "%s left %d array(s) unreleased" % (name, self.L.mock_outstanding() - before)
        if self.L.mock_exception_pending():
            exc = JavaException(self.L.mock_exception_class().decode(), self.L.mock_exception_message().decode())
            self.L.mock_exception_clear()
            raise exc
        return out

    def counters(self):
        g, c, a = C.c_int(), C.c_int(), C.c_int()
        self.L.mock_counters(C.byref(g), C.byref(c), C.byref(a))
        return {"gets": g.value, "copy_backs": c.value, "aborts": a.value}

    def patterns(self, pats16):
        """(char[] chars, int[] offsets, n) of a list of uint16 arrays — GpuFmIndex.java's flattening of char[][]"""
        offs = np.zeros(len(pats16) + 1, np.int32)
        for i, p in enumerate(pats16):
            offs[i + 1] = offs[i] + len(p)
        flat = np.concatenate([np.asarray(p, np.uint16) for p in pats16]) if len(pats16) and offs[-1] else np.zeros(0, np.uint16)
        return self.chars(flat), self.ints(offs), len(pats16)
