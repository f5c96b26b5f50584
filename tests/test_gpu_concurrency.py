"""Thread-safety and plan-lifetime contract of the C ABI on the GPU (include/fmx.h, "Threading"):

* the host-buffer entry points may be called from any number of host threads on ONE index at once — the
  reference's FmIndex is immutable and @ThreadSafe (FM:82), JMH shares nothing but could (J-FMS:30);
* a plan left by fmx_count_plan_dev is invalidated by anything else that plans on the same stream: a stale perm
  still counts correctly (it just carries no code words);
* fmx_attach_device_blob (the receive side of the RCCL broadcast) validates the image and answers like the
  index it was copied from.
Run with `-m gpu` on an MI355X."""
import ctypes as C
import random
import threading

import numpy as np
import pytest

import index4j_amd as ia
import orc
from common import hdfs_text
from parity_checks import check_all

pytestmark = pytest.mark.gpu
HD = hdfs_text()


def test_host_entry_points_from_many_threads_on_one_index():
    t = ia.synth_log(1 << 22)
    fm = ia.FmIndex(t, 16, True, device=0)
    o = orc.OracleFmIndex.read(fm.write(False))
    sizes = [70_000, 20_000, 33_333, 16_384, 50_001, 1000]  # planned (>= sort_min) and unplanned batches, all different
    work = []
    for k, n in enumerate(sizes):
        pat, off, pos = ia.synth_patterns(t, 8, n, seed=100 + k)
        oc, ost = o.count_batch(pat, off, threads=8)
        ol, of, _ = o.locate_batch(pat[: 4000 * 8], off[:4001], 16, threads=8)
        od, olen, ostx, oaux = o.extract_until_boundary_batch(0, pos[:3000], "\n", 512, threads=8)
        work.append((pat, off, pos, oc, ost, ol, of, od, olen))
    errors = []
    barrier = threading.Barrier(len(sizes))

    def run(k):
        try:
            pat, off, pos, oc, ost, ol, of, od, olen = work[k]
            barrier.wait()
            for _ in range(6):
                c, s = fm.count_batch(pat, off)
                assert (c == oc).all() and (s == ost).all(), "count, thread %d" % k
                locs, found, st = fm.locate_batch(pat[: 4000 * 8], off[:4001], 16)
                live = np.arange(16)[None, :] < found[:, None]
                assert (found == of).all() and (locs[live] == ol[live]).all(), "locate, thread %d" % k
                dst, ln, st2, aux = fm.extract_boundary_batch(pos[:3000], "\n", 0, 512)
                assert (st2 == 0).all() and (ln == olen).all() and (dst == od).all(), "boundary, thread %d" % k
        except BaseException as e:  # noqa: BLE001 - reported to the main thread
            errors.append(e)

    threads = [threading.Thread(target=run, args=(k,)) for k in range(len(sizes))]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[0]
    fm.close()


def test_a_stale_plan_is_harmless():
    """fmx_count_plan_dev -> (another call plans on the same stream, or the plan scratch grows) ->
    fmx_count_ordered_dev with the old perm: correct counts of the NEW batch order semantics aside — results are
    written at the original index, so any valid permutation of 0..n-1 gives the right answer"""
    import torch

    t = ia.synth_log(1 << 21)
    fm = ia.FmIndex(t, 32, True, device=0)
    o = orc.OracleFmIndex.read(fm.write(False))
    dev = torch.device("cuda", 0)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    n = 40_000
    pa, oa, _ = ia.synth_patterns(t, 8, n, seed=7)
    pb, ob, _ = ia.synth_patterns(t, 8, n, seed=8)
    d_pa = torch.from_numpy(pa.view(np.int16)).to(dev)
    d_pb = torch.from_numpy(pb.view(np.int16)).to(dev)
    d_off = torch.from_numpy(oa).to(dev)
    d_cnt = torch.zeros(n, dtype=torch.int32, device=dev)
    d_found = torch.zeros(n, dtype=torch.int32, device=dev)
    d_locs = torch.zeros(n * 4, dtype=torch.int32, device=dev)
    d_rng = torch.zeros(n * 2, dtype=torch.int32, device=dev)
    perm = C.c_void_p()
    assert ia.lib.fmx_count_plan_dev(fm.handle, d_pa.data_ptr(), d_off.data_ptr(), n, C.byref(perm), st) == 0
    assert perm.value
    # another batch is planned into the same per-stream scratch by locate: the code words there now belong to B
    assert ia.lib.fmx_locate_batch_dev(fm.handle, d_pb.data_ptr(), d_off.data_ptr(), n, 4, d_locs.data_ptr(), 4,
                                       d_found.data_ptr(), None, None, d_rng.data_ptr(), st) == 0
    assert ia.lib.fmx_count_ordered_dev(fm.handle, d_pa.data_ptr(), d_off.data_ptr(), perm, n, d_cnt.data_ptr(), None,
                                        None, st) == 0
    torch.cuda.synchronize()
    oc, _ = o.count_batch(pa, oa, threads=8)
    # the perm now is B's order (still a permutation of 0..n-1); A's counts must come out right regardless
    assert (d_cnt.cpu().numpy() == oc).all()
    # growth of the plan scratch (a larger batch) frees the block the old perm pointed into: the library must
    # not hand out code words for it; a fresh plan works
    big = 300_000
    pc, oc_off, _ = ia.synth_patterns(t, 8, big, seed=9)
    d_pc = torch.from_numpy(pc.view(np.int16)).to(dev)
    d_offc = torch.from_numpy(oc_off).to(dev)
    d_cntc = torch.zeros(big, dtype=torch.int32, device=dev)
    assert ia.lib.fmx_count_batch_dev(fm.handle, d_pc.data_ptr(), d_offc.data_ptr(), big, d_cntc.data_ptr(), None, None,
                                      st) == 0
    torch.cuda.synchronize()
    occ, _ = o.count_batch(pc, oc_off, threads=8)
    assert (d_cntc.cpu().numpy() == occ).all()
    fm.close()


def test_attach_device_blob_answers_like_the_source_index():
    """the receive path of the multi-GPU broadcast: blob -> torch.uint8 device tensor -> fmx_attach_device_blob"""
    import torch

    dev = torch.device("cuda", 0)

    class Attached:
        def __init__(self, text, sr):
            self.src = ia.FmIndex(text, sr, True, device=None)
            self.buf = torch.from_numpy(np.array(self.src.blob(), copy=True)).to(dev)
            self.fm = _Both(self.src, ia.FmIndex.attach_device_blob(self.buf.data_ptr(), self.buf.numel(), 0))

    class _Both:
        """queries go to the attached index, persistence to the source (an attached handle has no model)"""

        def __init__(self, src, att):
            self._src, self._att = src, att

        def write(self, framed=True):
            return self._src.write(framed)

        def __getattr__(self, name):
            return getattr(self._att, name)

    from parity_checks import GpuEngine

    def make(text, sr):
        a = Attached(text, sr)
        e = GpuEngine.__new__(GpuEngine)
        e.fm = a.fm
        e._keep = a
        return e

    check_all(make, HD[:150_000], 32, random.Random(1), n_q=100)
    check_all(make, HD[:60_000], 4, random.Random(2), n_q=60)
    a = Attached(HD[:50_000], 8)
    assert a.fm.getInputLength() == 50_001 and str(a.fm._att) == "FMIndex-sampleRate:8-extract:true"


def test_attach_rejects_damaged_images():
    import torch

    dev = torch.device("cuda", 0)
    src = ia.FmIndex(HD[:80_000], 16, True, device=None)
    good = np.array(src.blob(), copy=True)

    def attach(arr):
        buf = torch.from_numpy(arr).to(dev)
        h = C.c_void_p()
        rc = ia.lib.fmx_attach_device_blob(C.c_void_p(buf.data_ptr()), buf.numel(), 0, C.byref(h))
        if rc == 0:
            ia.lib.fmx_free(h)
        return rc

    assert attach(good) == 0
    assert attach(good[:-64].copy()) != 0  # truncated
    rnd = random.Random(6)
    for _ in range(20):  # a flipped bit anywhere in the body fails the checksum
        bad = good.copy()
        i = rnd.randrange(256, len(bad))
        bad[i] ^= 1 << rnd.randrange(8)
        assert attach(bad) != 0
    for off in (8, 16, 40, 64, 96, 100):  # header fields: sizes, counts, section offsets
        bad = good.copy()
        bad[off:off + 4] = np.frombuffer(np.uint32(0x7fffff00).tobytes(), np.uint8)
        assert attach(bad) != 0


def test_pipelined_host_buffer_count_from_threads_with_plain_and_registered_arrays():
    """fmx_count_batch above host_pipeline_min patterns travels in chunks (feeder thread, three streams, results through
    pinned staging or straight into registered arrays) — or, with every array registered, not at all: one launch reads and writes
    the caller's mapped arrays: same counts, statuses and LF-steps as the oracle — from several
    host threads at once, patterns of 8..31 characters (offsets shipped) and of one length (offsets made on the device),
    pageable and registered (fmx_host_register) arrays; offsets that run backwards are refused before anything is launched"""
    from index4j_amd import workload

    t = workload.reference_text(22)
    fm = ia.FmIndex(t, 32, True, device=0)
    o = orc.OracleFmIndex.read(fm.write(False))
    n = 300_000  # two chunks of the pipeline
    work = []
    pat, off, _ = workload.reference_queries(t, n, seed=5)
    work.append((pat, off))
    pat8, off8, _ = ia.synth_patterns(t, 8, n, seed=6)
    work.append((pat8, off8))
    expect = []
    for p, f in work:
        orc.counters_reset()
        oc, ost = o.count_batch(p, f, threads=8)
        expect.append((oc, ost, orc.counters()["lf_steps"]))
    errors = []

    def run(k, registered):
        try:
            p, f = work[k % 2]
            cnt = np.full(n, -1, np.int32)
            st = np.full(n, -1, np.int32)
            lf = np.full(n, -1, np.int32)
            arrays = (p, f, cnt, st, lf) if registered else ()
            for a in arrays:
                assert ia.lib.fmx_host_register(a.ctypes.data, a.nbytes) == 0
            try:
                for _ in range(3):
                    rc = ia.lib.fmx_count_batch(fm.handle, p.ctypes.data, f.ctypes.data, n, cnt.ctypes.data, lf.ctypes.data, st.ctypes.data)
                    assert rc == 0, ia.lib.fmx_last_error()
                    oc, ost, steps = expect[k % 2]
                    assert (cnt == oc).all() and (st == ost).all() and int(lf.astype(np.int64).sum()) == steps
            finally:
                for a in arrays:
                    ia.lib.fmx_host_unregister(a.ctypes.data)
        except Exception as e:  # noqa: BLE001
            errors.append((k, registered, repr(e)))

    # registered arrays: one launch over the mapped arrays (option host_mapped, the default), then the chunk pipeline with the
    # kernels storing straight into the registered result arrays (host_direct_stores), then with result copies
    try:
        for mapped, direct in ((1, 1), (0, 1), (0, 0)):
            assert ia.lib.fmx_set_option(b"host_mapped", mapped) == 0 and ia.lib.fmx_set_option(b"host_direct_stores", direct) == 0
            threads = [threading.Thread(target=run, args=(k, k >= 2)) for k in range(4)]
            for th in threads:
                th.start()
            for th in threads:
                th.join()
            assert not errors, (mapped, direct, errors)
    finally:
        ia.lib.fmx_set_option(b"host_mapped", 1)
        ia.lib.fmx_set_option(b"host_direct_stores", 1)
    # the unpipelined path (option) gives the same
    ia.lib.fmx_set_option(b"host_pipeline_min", 0)
    try:
        c2, s2 = fm.count_batch(work[0][0], work[0][1])
        assert (c2 == expect[0][0]).all() and (s2 == expect[0][1]).all()
    finally:
        ia.lib.fmx_set_option(b"host_pipeline_min", 131072)
    # offsets that decrease: FMX_E_ARG from both paths, nothing launched
    bad = work[1][1].copy()
    bad[200_000] = bad[199_999] - 3
    cnt = np.zeros(n, np.int32)
    for nn in (n, 1000):
        b = bad if nn == n else np.array([0, 8, 4] + [8] * (nn - 2), np.int32)
        rc = ia.lib.fmx_count_batch(fm.handle, work[1][0].ctypes.data, b.ctypes.data, nn, cnt.ctypes.data, None, None)
        assert rc == -1 and b"offsets" in ia.lib.fmx_last_error()
    # a pair of offsets whose int32 differences wrap to non-negative values (8, 2e9, -2e9, 16: the middle difference is
    # +294,967,296 in 32 bits) with both chunk ends in range: rejected by the pipelined path too, nothing launched
    wrap = work[1][1].copy()
    wrap[280_001] = 2_000_000_000
    wrap[280_002] = -2_000_000_000
    rc = ia.lib.fmx_count_batch(fm.handle, work[1][0].ctypes.data, wrap.ctypes.data, n, cnt.ctypes.data, None, None)
    assert rc == -1 and b"offsets" in ia.lib.fmx_last_error()
    # ... and by the mapped path (every array registered): refused before the launch reads a character
    arrays = (work[1][0], wrap, bad, cnt)
    for a in arrays:
        assert ia.lib.fmx_host_register(a.ctypes.data, a.nbytes) == 0
    try:
        for b in (wrap, bad):
            rc = ia.lib.fmx_count_batch(fm.handle, work[1][0].ctypes.data, b.ctypes.data, n, cnt.ctypes.data, None, None)
            assert rc == -1 and b"offsets" in ia.lib.fmx_last_error()
    finally:
        for a in arrays:
            ia.lib.fmx_host_unregister(a.ctypes.data)
    fm.close()


def test_host_entry_points_with_registered_arrays_give_the_same_rows():
    """Round 4: fmx_locate_batch stores its hits straight into a registered `locations` array (only the slots it fills travel) and
    reads registered patterns in place; fmx_extract_batch / fmx_extract_boundary_batch move registered destination rows in chunks
    over three streams.  Same results as with plain arrays and as the oracle, entries the walks do not write keep the caller's
    values (the arrays are in/out)."""
    text = ia.synth_log(1 << 21)
    fm = ia.FmIndex(text, 16, True, device=0)
    o = orc.OracleFmIndex.read(fm.write(False))
    L = ia.lib
    n, M, m = 40_000, 5, 6
    pat, off, pos = ia.synth_patterns(text, m, n, seed=8)
    pat = pat.copy()
    pat[7 * m] = 7  # a pattern without hits
    want_l, want_f, want_s = o.locate_batch(pat, off, M, threads=8, fill=-7)

    def with_registered(arrays, fn):
        for a in arrays:
            assert L.fmx_host_register(a.ctypes.data, a.nbytes) == 0
        try:
            fn()
        finally:
            for a in arrays:
                L.fmx_host_unregister(a.ctypes.data)

    for registered in (False, True):
        locs = np.full((n, M), -7, np.int32)
        found = np.zeros(n, np.int32)
        st = np.zeros(n, np.int32)
        lf = np.zeros(n, np.int32)

        def locate():
            assert L.fmx_locate_batch(fm.handle, pat.ctypes.data, off.ctypes.data, n, M, locs.ctypes.data, M, found.ctypes.data,
                                      lf.ctypes.data, st.ctypes.data) == 0

        if registered:
            with_registered((pat, off, locs, found, st, lf), locate)
        else:
            locate()
        assert (found == want_f).all() and (st == want_s).all() and (locs == want_l).all(), registered
    # an array registered only in part (its first rows): no kernel is pointed at it (mapped arrays are checked at both ends); the HIP
    # runtime refuses to copy a range that is registered in part, so the call fails with a status — and the next one works
    locs = np.full((n, M), -7, np.int32)
    found = np.zeros(n, np.int32)
    st = np.zeros(n, np.int32)
    half = locs[: n // 2]
    assert L.fmx_host_register(half.ctypes.data, half.nbytes) == 0
    try:
        rc = L.fmx_locate_batch(fm.handle, pat.ctypes.data, off.ctypes.data, n, M, locs.ctypes.data, M, found.ctypes.data, None,
                                st.ctypes.data)
        assert rc in (0, ia._lib.E_HIP)
        if rc == 0:
            assert (found == want_f).all() and (locs == want_l).all()
    finally:
        L.fmx_host_unregister(half.ctypes.data)
    assert L.fmx_locate_batch(fm.handle, pat.ctypes.data, off.ctypes.data, n, M, locs.ctypes.data, M, found.ctypes.data, None,
                              st.ctypes.data) == 0
    assert (found == want_f).all() and (st == want_s).all() and (locs == want_l).all()
    # extract: rows of 300 chars (five chunks of the pipeline), windows of 1..200 chars, some past the text's end
    rnd = np.random.default_rng(3)
    cap = 300
    starts = rnd.integers(0, len(text) - 10, n).astype(np.int32)
    stops = (starts + rnd.integers(1, 200, n)).astype(np.int32)
    want_d, want_n, want_s = o.extract_batch(starts, stops, cap, offset=3, threads=8, fill=0xABCD)
    froms = np.ascontiguousarray(pos[:n]).astype(np.int32)
    want_bd, want_bn, want_bs, want_ba = o.extract_until_boundary_batch(0, froms, "\n", cap, offset=2, threads=8, fill=0xABCD)
    for registered in (False, True):
        dst = np.full((n, cap), 0xABCD, np.uint16)
        ol = np.zeros(n, np.int32)
        st = np.zeros(n, np.int32)
        aux = np.zeros(n, np.int32)

        def extract():
            assert L.fmx_extract_batch(fm.handle, starts.ctypes.data, stops.ctypes.data, n, dst.ctypes.data, cap, 3, ol.ctypes.data, None,
                                       st.ctypes.data) == 0

        if registered:
            with_registered((starts, stops, dst, ol, st), extract)
        else:
            extract()
        assert (ol == want_n).all() and (st == want_s).all() and (dst == want_d).all(), registered
        dst[:] = 0xABCD

        def boundary():
            assert L.fmx_extract_boundary_batch(fm.handle, froms.ctypes.data, n, 10, 0, dst.ctypes.data, cap, 2, ol.ctypes.data, None,
                                                st.ctypes.data, aux.ctypes.data) == 0

        if registered:
            with_registered((froms, dst, ol, st, aux), boundary)
        else:
            boundary()
        assert (ol == want_bn).all() and (st == want_bs).all() and (aux == want_ba).all() and (dst == want_bd).all(), registered
    fm.close()
