"""bindings/jni/fmx_jni.c on the GPU without a JVM: every `native` method of GpuFmIndex.java called through its JNI entry point
against a mock JNIEnv (tests/jni_stub/mock_jnienv.c — copy-always arrays, every Get matched by one Release, no JNI call with
an exception pending; tests/jni_mock.py), the arrays a Java caller would read afterwards compared with the oracle's, entry by
entry and whole destination rows.  What row f2 of SURVEY §8 still lacks is the Java class under a JVM (no JDK in the image); the
C half of the binding — what turns Java arrays into the C ABI's pointers and back — runs here.  `-m gpu`."""
import random

import numpy as np
import pytest

import index4j_amd as ia
import orc
from common import hdfs_text
from jni_mock import CHARS, INTS, LONGS, JavaException, MockJvm

pytestmark = pytest.mark.gpu
HD = hdfs_text()


@pytest.fixture(scope="module")
def jvm():
    return MockJvm()


def _queries(t16, rnd, n):
    L = len(t16)
    pats = [t16[s:s + rnd.randrange(1, 24)] for s in (rnd.randrange(max(1, L - 24)) for _ in range(n - 4))]
    pats += [ia.as_chars("zzzzqq"), t16[:1], ia.as_chars("INFO"), ia.as_chars("\n")]
    pats.append(t16[:0])  # an EMPTY pattern (FM:456-457: ArrayIndexOutOfBoundsException -> status 9)
    return pats


def _filled(jvm, kind, n, value):
    a = jvm.new(kind, n)
    jvm.view(a)[:] = value
    return a


@pytest.mark.parametrize("build_on_gpu", [0, 1])
def test_every_native_query_method_equals_the_oracle(jvm, build_on_gpu):
    rnd = random.Random(7 + build_on_gpu)
    text = HD[:150_001]
    t16 = ia.as_chars(text)
    o = orc.OracleFmIndex(text, 16, True)
    h = jvm.call("nativeBuild", jvm.chars(t16), 16, 1, 0, build_on_gpu)
    assert h
    try:
        assert jvm.call("nativeDeviceOf", h) == 0
        assert jvm.call("nativeInputLength", h) == len(t16) + 1
        arr = jvm.call("nativeSave", h, 0)
        assert jvm.view(arr).view(np.uint8).tobytes() == o.write(False)  # the builder's bytes == the oracle's
        jvm.free(arr)

        pats = _queries(t16, rnd, 700)
        chars, offs, n = jvm.patterns(pats)
        ch, off = jvm.view(chars).copy(), jvm.view(offs).copy()
        # count
        counts, status = _filled(jvm, INTS, n, -3), _filled(jvm, INTS, n, -3)
        jvm.call("nativeCountBatch", h, chars, offs, n, counts, status)
        oc, ost = o.count_batch(ch, off)
        assert (jvm.view(counts) == oc).all() and (jvm.view(status) == ost).all() and jvm.view(status)[-1] == 9
        assert (jvm.view(chars) == ch).all() and (jvm.view(offs) == off).all()  # inputs are inputs
        # locate: whole rows, untouched slots too
        for mm, cap in ((16, 16), (-1, 40), (8, 3)):
            locs, found, st = _filled(jvm, INTS, n * cap, -7), _filled(jvm, INTS, n, -3), _filled(jvm, INTS, n, -3)
            jvm.call("nativeLocateBatch", h, chars, offs, n, mm, locs, cap, found, st)
            ol, of, os_ = o.locate_batch(ch, off, mm, cap, fill=-7)
            ok = os_ == 0
            assert (jvm.view(st) == os_).all()
            assert (jvm.view(found)[ok] == of[ok]).all() and (jvm.view(locs).reshape(n, cap)[ok] == ol[ok]).all()
            jvm.free(locs, found, st)
        # extract
        L = len(t16) + 1
        k = 400
        a = np.array([rnd.randrange(L) for _ in range(k)], np.int32)
        b = np.minimum(a + np.array([rnd.randrange(60) for _ in range(k)], np.int32), L)
        a[:3], b[:3] = (-5, 3, 10), (10, L + 7, 5)
        dst, out_len, st = _filled(jvm, CHARS, k * 70, 9), _filled(jvm, INTS, k, -3), _filled(jvm, INTS, k, -3)
        jvm.call("nativeExtractBatch", h, jvm.ints(a), jvm.ints(b), k, dst, 70, 2, out_len, st)
        od, on, oe = o.extract_batch(a, b, 70, 2, fill=9)
        ok = oe == 0
        assert (jvm.view(st) == oe).all() and (jvm.view(dst).reshape(k, 70)[ok] == od[ok]).all() and (jvm.view(out_len)[ok] == on[ok]).all()
        jvm.free(dst, out_len, st)
        # extractUntilBoundary{,Left,Right}
        fr = np.array([rnd.randrange(L) for _ in range(k)], np.int32)
        fr[:2] = (-1, L + 3)
        froms = jvm.ints(fr)
        for mode in (0, 1, 2):
            for cap in (1 << 10, 40):
                dst, out_len = _filled(jvm, CHARS, k * cap, 5), _filled(jvm, INTS, k, -3)
                st, aux = _filled(jvm, INTS, k, -3), _filled(jvm, INTS, k, -3)
                jvm.call("nativeExtractBoundaryBatch", h, froms, k, 10, mode, dst, cap, 0, out_len, st, aux)
                od, on, oe, oa = o.extract_until_boundary_batch(mode, fr, "\n", cap, fill=5)
                assert (jvm.view(st) == oe).all() and (jvm.view(dst).reshape(k, cap) == od).all(), (mode, cap)
                assert (jvm.view(out_len)[oe == 0] == on[oe == 0]).all() and (jvm.view(aux)[oe == 8] == oa[oe == 8]).all()
                jvm.free(dst, out_len, st, aux)
        # the fused pipelines: locate -> extract and locate -> extractUntilBoundary, the hit table kept on the device
        mm, row = 4, 48
        slots = n * mm
        fm = ia.FmIndex(text, 16, True, device=0)
        try:
            for mode in (-1, 0):
                locs, found = _filled(jvm, INTS, slots, -1), _filled(jvm, INTS, n, 0)  # (the Python mirror's initial values)
                rows, out_len = _filled(jvm, CHARS, slots * row, 0), _filled(jvm, INTS, slots, -1)
                st, hst, hax = _filled(jvm, INTS, n, 0), _filled(jvm, INTS, slots, 0), _filled(jvm, INTS, slots, 0)
                jvm.call("nativeLocatePipeline", h, chars, offs, n, mm, mode, 10, row, locs, found, rows, out_len, st, hst, hax)
                ref = fm.locate_extract_batch(ch, off, mm, row) if mode < 0 else fm.locate_lines_batch(ch, off, mm, "\n", row, 0)
                assert (jvm.view(locs).reshape(n, mm) == ref["locs"]).all() and (jvm.view(found) == ref["found"]).all()
                assert (jvm.view(rows).reshape(n, mm, row) == ref["dst"]).all() and (jvm.view(out_len).reshape(n, mm) == ref["out_len"]).all()
                assert (jvm.view(st) == ref["status"]).all() and (jvm.view(hst).reshape(n, mm) == ref["hit_status"]).all()
                # ... and against the oracle, hit by hit, for the patterns that have hits
                fnd, lo = jvm.view(found), jvm.view(locs).reshape(n, mm)
                R = jvm.view(rows).reshape(n, mm, row)
                for i in range(0, n, 9):
                    for j in range(int(max(fnd[i], 0))):
                        if mode < 0:
                            e_n, e_d = o.extract(int(lo[i, j]), min(L, int(lo[i, j]) + row), dest_len=row, offset=0)
                            assert (R[i, j] == e_d).all()
                        elif jvm.view(hst).reshape(n, mm)[i, j] == 0:
                            e_n, e_d = o.extract_until_boundary(0, int(lo[i, j]), row, 0, "\n")
                            assert (R[i, j] == e_d).all() and jvm.view(out_len).reshape(n, mm)[i, j] == e_n
                jvm.free(locs, found, rows, out_len, st, hst, hax)
        finally:
            fm.close()
    finally:
        jvm.call("nativeFree", h)


def test_load_replicate_and_the_multi_methods(jvm):
    """fromSerialized -> replicate(int[]) -> the *Multi natives: two replicas on the one GPU of a box, a batch the replica count does
    not divide, against the oracle; configs[4]'s shape (segments x replicas) through nativeCountLocateSegmentsMulti"""
    rnd = random.Random(23)
    text = HD[:120_000]
    t16 = ia.as_chars(text)
    o = orc.OracleFmIndex(text, 8, True)
    h = jvm.call("nativeLoad", jvm.bytes_(o.write(True)), 0)  # the ORACLE's stream (framed like Serialization.writeToByteArray)
    reps = jvm.call("nativeReplicate", h, jvm.ints([0, 0, 0]))
    handles = jvm.view(reps).copy()
    assert len(handles) == 3 and len(set(handles.tolist()) | {h}) == 4
    try:
        assert all(jvm.call("nativeDeviceOf", int(x)) == 0 for x in handles)
        pats = _queries(t16, rnd, 1000)
        chars, offs, n = jvm.patterns(pats)
        ch, off = jvm.view(chars).copy(), jvm.view(offs).copy()
        counts, status = _filled(jvm, INTS, n, -3), _filled(jvm, INTS, n, -3)
        jvm.call("nativeCountBatchMulti", reps, chars, offs, n, counts, status)
        oc, ost = o.count_batch(ch, off)
        assert (jvm.view(counts) == oc).all() and (jvm.view(status) == ost).all()
        mm, cap = 5, 5
        locs, found, st = _filled(jvm, INTS, n * cap, -7), _filled(jvm, INTS, n, -3), _filled(jvm, INTS, n, -3)
        jvm.call("nativeLocateBatchMulti", reps, chars, offs, n, mm, locs, cap, found, st)
        ol, of, os_ = o.locate_batch(ch, off, mm, cap, fill=-7)
        ok = os_ == 0
        assert (jvm.view(st) == os_).all() and (jvm.view(found)[ok] == of[ok]).all() and (jvm.view(locs).reshape(n, cap)[ok] == ol[ok]).all()
        L = len(t16) + 1
        k = 301
        a = np.array([rnd.randrange(L) for _ in range(k)], np.int32)
        b = np.minimum(a + np.array([rnd.randrange(50) for _ in range(k)], np.int32), L)
        dst, out_len, st2 = _filled(jvm, CHARS, k * 60, 9), _filled(jvm, INTS, k, -3), _filled(jvm, INTS, k, -3)
        jvm.call("nativeExtractBatchMulti", reps, jvm.ints(a), jvm.ints(b), k, dst, 60, 1, out_len, st2)
        od, on, oe = o.extract_batch(a, b, 60, 1, fill=9)
        assert (jvm.view(st2) == oe).all() and (jvm.view(dst).reshape(k, 60)[oe == 0] == od[oe == 0]).all()
        assert (jvm.view(out_len)[oe == 0] == on[oe == 0]).all()
        fr = np.array([rnd.randrange(L) for _ in range(k)], np.int32)
        for mode in (0, 1, 2):
            dst, out_len = _filled(jvm, CHARS, k * 256, 5), _filled(jvm, INTS, k, -3)
            st3, aux = _filled(jvm, INTS, k, -3), _filled(jvm, INTS, k, -3)
            jvm.call("nativeExtractBoundaryBatchMulti", reps, jvm.ints(fr), k, 10, mode, dst, 256, 0, out_len, st3, aux)
            od, on, oe, oa = o.extract_until_boundary_batch(mode, fr, "\n", 256, fill=5)
            assert (jvm.view(st3) == oe).all() and (jvm.view(dst).reshape(k, 256) == od).all()
            assert (jvm.view(out_len)[oe == 0] == on[oe == 0]).all()
        # a replica set whose handle table is stale fails as an exception, not as a crash
        with pytest.raises(JavaException) as e:
            jvm.call("nativeCountBatchMulti", jvm.longs([int(handles[0]), 0]), chars, offs, n, counts, status)
        assert e.value.cls == "java/lang/RuntimeException"
    finally:
        for x in handles:
            jvm.call("nativeFree", int(x))
        jvm.call("nativeFree", h)


def test_segment_sets_through_the_glue(jvm):
    """nativeCountSegments and nativeCountLocateSegmentsMulti (BASELINE configs[4] from a Java host) at a small size: 4 segment
    indexes x 2 replicas, counts summed over the segments' oracles, hits base-shifted and equal to the Python mirror's"""
    rnd = random.Random(31)
    text = HD[:130_000]
    t16 = ia.as_chars(text)
    bases = [0, 33_000, 66_000, 99_000]
    ends = bases[1:] + [len(t16)]
    oracles = [orc.OracleFmIndex(t16[a:b], 16, True) for a, b in zip(bases, ends)]
    segs = [jvm.call("nativeBuild", jvm.chars(t16[a:b]), 16, 1, 0, 1) for a, b in zip(bases, ends)]
    made = []
    try:
        pats = _queries(t16, rnd, 500)
        chars, offs, n = jvm.patterns(pats)
        ch, off = jvm.view(chars).copy(), jvm.view(offs).copy()
        exp = np.zeros(n, np.int64)
        for oo in oracles:
            exp += oo.count_batch(ch, off)[0]
        counts, status = _filled(jvm, LONGS, n, -3), _filled(jvm, INTS, n, -3)
        jvm.call("nativeCountSegments", jvm.longs(segs), chars, offs, n, counts, status)
        assert (jvm.view(counts) == exp).all() and jvm.view(status)[-1] == 9
        # replicas of every segment on devices {0, 0}; handles replica-major
        per_seg = []
        for s in segs:
            r = jvm.call("nativeReplicate", s, jvm.ints([0, 0]))
            per_seg.append(jvm.view(r).copy())
            made += per_seg[-1].tolist()
        table = [int(per_seg[s][r]) for r in range(2) for s in range(len(segs))]
        mm = 6
        counts2, locs = _filled(jvm, LONGS, n, -3), _filled(jvm, LONGS, n * mm, -1)
        found, st = _filled(jvm, INTS, n, -3), _filled(jvm, INTS, n, -3)
        jvm.call("nativeCountLocateSegmentsMulti", jvm.longs(table), 2, len(segs), jvm.longs(bases), chars, offs, n, mm, counts2, locs,
                 found, st)
        assert (jvm.view(counts2) == exp).all() and (jvm.view(st) == jvm.view(status)).all()
        # hits: the first `found` slots hold positions of the WHOLE text at which the pattern stands
        fnd, lo = jvm.view(found), jvm.view(locs).reshape(n, mm)
        assert (fnd[:-1] == np.minimum(exp, mm)[:-1]).all()  # (the last pattern is the empty one: status 9, its row unspecified)
        for i in range(n - 1):
            p = pats[i]
            for j in range(int(fnd[i])):
                at = int(lo[i, j])
                assert (t16[at:at + len(p)] == p).all(), (i, j)
            assert (lo[i, int(fnd[i]):] == -1).all()
    finally:
        for x in made + segs:
            jvm.call("nativeFree", int(x))
