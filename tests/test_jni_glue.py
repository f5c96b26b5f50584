"""bindings/jni/fmx_jni.c RUN on the CPU against a mock JNIEnv (tests/jni_stub/mock_jnienv.c, tests/jni_mock.py): what the glue
does before and after it reaches libfmx — argument checks, the exception a failure becomes (class and message, as GpuFmIndex.java
documents them), arrays released exactly once, results copied back — plus the three-way agreement of the `native` declarations
(Java), the entry points (C) and the table the tests call through.  Queries need a GPU: tests/test_gpu_jni_glue.py.

No JVM here (SURVEY §8 row f2 stays blocked by the image); the mock is the strictest legal JNI: Get<T>ArrayElements always
copies, so a result released with JNI_ABORT would be lost and show up as a zeroed array."""
import os
import re

import numpy as np
import pytest

import index4j_amd as ia
import orc
from common import hdfs_text
from jni_mock import BYTES, CHARS, INTS, LONGS, SIGNATURES, JavaException, MockJvm, jboolean, jchar, jint, jlong, ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JAVA = os.path.join(ROOT, "bindings", "java", "com", "dynatrace", "fm", "gpu", "GpuFmIndex.java")
GLUE = os.path.join(ROOT, "bindings", "jni", "fmx_jni.c")


@pytest.fixture(scope="module")
def jvm():
    return MockJvm()


def _split(params):
    return [p.strip() for p in params.replace("\n", " ").split(",") if p.strip()]


def test_java_declarations_glue_entries_and_test_table_agree_type_by_type():
    java = open(JAVA).read()
    glue = open(GLUE).read()
    java_c = {"long": "jlong", "int": "jint", "boolean": "jboolean", "char": "jchar", "void": "void", "byte[]": "jbyteArray",
              "char[]": "jcharArray", "int[]": "jintArray", "long[]": "jlongArray"}
    ctype = {"jlong": jlong, "jint": jint, "jboolean": jboolean, "jchar": jchar, "void": None, "jbyteArray": ref, "jcharArray": ref,
             "jintArray": ref, "jlongArray": ref}
    decl = {m.group(2): (m.group(1), [p.rsplit(None, 1)[0] for p in _split(m.group(3))])
            for m in re.finditer(r"\bnative\s+([\w\[\]]+)\s+(native\w+)\s*\(([^)]*)\)", java)}
    entry = {m.group(2): (m.group(1), [p.rsplit(None, 1)[0].strip() for p in _split(m.group(3))])
             for m in re.finditer(r"JNIEXPORT\s+(\w+)\s+JNICALL\s+Java_com_dynatrace_fm_gpu_GpuFmIndex_(native\w+)\s*\(([^)]*)\)", glue)}
    assert set(decl) == set(entry) == set(SIGNATURES) and len(decl) >= 20
    for name, (jres, jargs) in decl.items():
        cres, cargs = entry[name]
        assert cargs[:2] == ["JNIEnv *env" .rsplit(None, 1)[0], "jclass"], (name, cargs[:2])  # static natives: (JNIEnv *, jclass, ...)
        assert [java_c[t] for t in jargs] == cargs[2:], (name, jargs, cargs[2:])
        assert java_c[jres] == cres, (name, jres, cres)
        tres, targs = SIGNATURES[name]
        assert tres is ctype[cres] and targs == [ctype[t] for t in cargs[2:]], name


def test_exceptions_the_glue_raises_before_it_reaches_the_library(jvm):
    """raw pointers only cross into libfmx after the Java arrays have been measured: handle 0 is never dereferenced here"""
    chars, offs, n = jvm.patterns([ia.as_chars("abc"), ia.as_chars("de")])
    counts, status = jvm.new(INTS, 2), jvm.new(INTS, 2)
    short = jvm.new(INTS, 1)
    bad_offs = jvm.ints([0, 4, 3])
    far_offs = jvm.ints([0, 3, 6])  # ends beyond chars
    cases = [
        ("nativeCountBatch", (0, chars, offs, -1, counts, status), "negative batch size"),
        ("nativeCountBatch", (0, None, offs, n, counts, status), "chars is null"),
        ("nativeCountBatch", (0, chars, short, n, counts, status), "offsets shorter than n + 1"),
        ("nativeCountBatch", (0, chars, bad_offs, n, counts, status), "pattern offsets are not a partition of chars"),
        ("nativeCountBatch", (0, chars, far_offs, n, counts, status), "pattern offsets are not a partition of chars"),
        ("nativeCountBatch", (0, chars, offs, n, short, status), "counts shorter than n"),
        ("nativeCountBatch", (0, chars, offs, n, counts, None), "status shorter than n"),
        ("nativeLocateBatch", (0, chars, offs, n, 4, jvm.new(INTS, 7), 4, counts, status), "locations shorter than n * locCap"),
        ("nativeLocateBatch", (0, chars, offs, n, 4, jvm.new(INTS, 8), -1, counts, status), None),  # locCap < 0: returns silently (Java checks first)
        ("nativeExtractBatch", (0, counts, short, 2, jvm.new(CHARS, 8), 4, 0, counts, status), "stop shorter than n"),
        ("nativeExtractBatch", (0, counts, counts, 2, jvm.new(CHARS, 7), 4, 0, counts, status), "dst shorter than n * dstLen"),
        ("nativeExtractBoundaryBatch", (0, counts, 2, 10, 0, jvm.new(CHARS, 8), 4, 0, counts, status, short), "aux shorter than n"),
        ("nativeLocatePipeline", (0, chars, offs, n, 2, -1, 0, 3, jvm.new(INTS, 4), counts, jvm.new(CHARS, 11), jvm.new(INTS, 4), status,
                                  jvm.new(INTS, 4), jvm.new(INTS, 4)), "rows shorter than n * maxMatches * rowLength"),
        ("nativeCountSegments", (jvm.new(LONGS, 0), chars, offs, n, jvm.new(LONGS, 2), status), "no segments"),
        ("nativeReplicate", (0, None), "no devices"),
        ("nativeReplicate", (0, jvm.new(INTS, 0)), "no devices"),
        ("nativeCountBatchMulti", (None, chars, offs, n, counts, status), "no replicas, or not replicas x segments handles"),
        ("nativeCountLocateSegmentsMulti", (jvm.longs([1, 2, 3]), 2, 2, jvm.longs([0, 9]), chars, offs, n, 4, jvm.new(LONGS, 2), jvm.new(LONGS, 8),
                                            counts, status), "no replicas, or not replicas x segments handles"),
        ("nativeCountLocateSegmentsMulti", (jvm.longs([1, 2, 3, 4]), 2, 2, jvm.longs([0]), chars, offs, n, 4, jvm.new(LONGS, 2), jvm.new(LONGS, 8),
                                            counts, status), "segmentBase shorter than the number of segments"),
    ]
    for name, args, message in cases:
        if message is None:
            jvm.call(name, *args)
            continue
        with pytest.raises(JavaException) as e:
            jvm.call(name, *args)
        assert e.value.cls == "java/lang/IllegalArgumentException" and e.value.message == message, (name, str(e.value))
    assert (jvm.view(counts) == 0).all() and (jvm.view(status) == 0).all()  # nothing was written
    for name, args, what in (("nativeLoad", (None, 0), "serialized"), ("nativeBuild", (None, 32, 1, 0, 0), "text")):
        with pytest.raises(JavaException) as e:
            jvm.call(name, *args)
        assert e.value.cls == "java/lang/NullPointerException" and e.value.message == what


def test_library_failures_become_the_exceptions_the_java_class_documents(jvm):
    # a stream that is no index: Serialization.readFromByteArray's IOException (SER:46-56)
    with pytest.raises(JavaException) as e:
        jvm.call("nativeLoad", jvm.bytes_(b"\x00\x00\x00\x07garbage-not-an-index"), 0)
    assert e.value.cls == "java/io/IOException" and e.value.message
    fm = ia.FmIndex("abracadabra", 2, True, device=None)
    good = bytearray(fm.write(True))
    # more than 32,767 different symbols: FM:423-426's IllegalArgumentException, message and all
    text = np.arange(1, 32770, dtype=np.uint16)
    with pytest.raises(JavaException) as e:
        jvm.call("nativeBuild", jvm.chars(text), 32, 1, 0, 0)
    assert e.value.cls == "java/lang/IllegalArgumentException" and e.value.message == "Input has more than 32767 different symbols"
    if ia.lib.fmx_device_count() <= 0:
        # no GPU: the build succeeds on the host, making it resident fails — a RuntimeException carrying fmx_last_error, no leak
        with pytest.raises(JavaException) as e:
            jvm.call("nativeBuild", jvm.chars(ia.as_chars("abracadabra")), 2, 1, 0, 0)
        assert e.value.cls == "java/lang/RuntimeException" and "device" in e.value.message.lower()
        with pytest.raises(JavaException) as e:
            jvm.call("nativeLoad", jvm.bytes_(bytes(good)), 0)
        assert e.value.cls == "java/lang/RuntimeException"
    # a query on an index that is not resident: the library's error, not a crash
    chars, offs, n = jvm.patterns([ia.as_chars("abra")])
    counts, status = jvm.new(INTS, 1), jvm.new(INTS, 1)
    with pytest.raises(JavaException) as e:
        jvm.call("nativeCountBatch", fm.handle.value, chars, offs, n, counts, status)
    assert e.value.cls == "java/lang/RuntimeException"
    assert jvm.counters() == {"gets": 5, "copy_backs": 2, "aborts": 3}  # offsets twice (the check, the call); inputs aborted, outputs copied back


def test_accessors_and_save_through_the_glue_equal_the_library_and_the_oracle(jvm):
    text = hdfs_text()[:40_000]
    fm = ia.FmIndex(text, 8, True, device=None)
    o = orc.OracleFmIndex(text, 8, True)
    h = fm.handle.value
    assert jvm.call("nativeInputLength", h) == fm.getInputLength() == len(ia.as_chars(text)) + 1  # (FM:131: the sentinel counts)
    assert jvm.call("nativeAlphabetLength", h) == fm.getAlphabetLength()
    assert jvm.call("nativeSavedOrderModelled", h) == 1
    assert jvm.call("nativeDeviceOf", h) == -1
    for framed in (0, 1):
        arr = jvm.call("nativeSave", h, framed)
        got = jvm.view(arr).view(np.uint8).tobytes()
        assert got == fm.write(bool(framed)) == o.write(bool(framed))
        jvm.free(arr)
    jvm.call("nativeFree", 0)  # close() of an index that never opened
