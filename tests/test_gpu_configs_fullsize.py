"""BASELINE.json's configs at their STATED sizes, HIP kernels (through the C ABI) against the oracle:

  configs[1]  count() of all 1,048,576 8-char patterns on the 256 MiB log index, sampleRate 32
  configs[2]  locate() of 100,000 patterns, maxMatches 16, same index — every position, SA order
  configs[3]  extractUntilBoundary('\\n') of 100,000 hit locations on the sampleRate-64 index — whole rows
  configs[4]  per-GPU share of the 8M-pattern batch over the 2 GiB text as 8 segment indexes

The 256 MiB indexes are built once per module with the suffix-array stage on the GPU (fmx_build_on_device);
the oracle loads the very bytes (FmIndex.write) — its own builder is compared byte for byte at sizes it
finishes in seconds (tests/test_builder_parity.py, test_gpu_parity.py) and, here, anchored by brute-force
scans of the text.  257 superblocks (the LDS superblock cache near its 320 limit), 29-bit suffix samples and
the planned 1 M-pattern path only exist at this size.  Run with `-m gpu` on an MI355X."""
import os

import numpy as np
import pytest

import index4j_amd as ia
import orc
from index4j_amd import workload

pytestmark = [pytest.mark.gpu, pytest.mark.plan_policy]  # the configs as shipped: the library decides which batches it plans
CORES = os.cpu_count() or 1
M = 8


class Built:
    def __init__(self, text, sample_rate):
        self.text = text
        self.fm = ia.FmIndex(text, sample_rate, True, device=0, build_device=0)
        self.ser = self.fm.write(False)
        self.oracle = orc.OracleFmIndex.read(self.ser)


@pytest.fixture(scope="module")
def text256():
    return workload.log_text(28)


@pytest.fixture(scope="module")
def batch(text256):
    return workload.count_batch_patterns(text256, 1 << 20, M)


@pytest.fixture(scope="module")
def idx32(text256):
    b = Built(text256, 32)
    yield b
    b.fm.close()


def brute_count(hay, needle):
    n, i = 0, hay.find(needle)
    while i >= 0:
        n += 1
        i = hay.find(needle, i + 1)
    return n


def test_index_shape_is_the_stated_one(idx32):
    fm = idx32.fm
    assert fm.getInputLength() == (1 << 28) + 1  # includes the sentinel (FM:929)
    assert str(fm) == "FMIndex-sampleRate:32-extract:true"
    assert idx32.oracle.getInputLength() == fm.getInputLength()
    assert idx32.oracle.getAlphabetLength() == fm.getAlphabetLength()
    assert idx32.oracle.write(False) == idx32.ser  # parse -> emit of the oracle reproduces the bytes


def test_config1_all_1m_counts_statuses_and_lf_steps(idx32, batch, text256):
    pat, off, pos = batch
    n = len(off) - 1
    assert n == 1 << 20
    cnt, st, lf = idx32.fm.count_batch(pat, off, want_steps=True)
    orc.counters_reset()
    oc, ost = idx32.oracle.count_batch(pat, off, threads=CORES)
    c = orc.counters()
    assert (cnt == oc).all() and (st == ost).all() and (st == 0).all()
    assert int(lf.astype(np.int64).sum()) == c["lf_steps"]
    assert (cnt >= 1).all()  # every pattern is a substring of the text
    # anchor of the chain: brute-force scans of the 256 MiB text for a few patterns
    hay = text256.astype(np.uint8).tobytes()
    for i in range(0, n, n // 12):
        assert cnt[i] == brute_count(hay, hay[pos[i]: pos[i] + M]), i


def test_config1_unplanned_and_planned_paths_agree(idx32, batch):
    """the 1 M batch under the library's own policy (the host-buffer path counts it in chunks of 262,144: the caller's order),
    the same batch with the plan stage forced (ordered by estimated SA row, by the table's SA row, by trailing codes), and
    small slices — same answers"""
    pat, off, _ = batch
    cnt, st = idx32.fm.count_batch(pat, off)
    try:
        for sa_key in (2, 1, 0):
            assert ia.lib.fmx_set_option(b"plan_sa_min", 0) == 0 and ia.lib.fmx_set_option(b"plan_min_per_string", 0) == 0
            assert ia.lib.fmx_set_option(b"plan_sa_key", sa_key) == 0
            forced, st_f = idx32.fm.count_batch(pat, off)
            assert (forced == cnt).all() and (st_f == st).all(), sa_key
    finally:
        ia.lib.fmx_set_option(b"plan_min_per_string", 16)
        ia.lib.fmx_set_option(b"plan_sa_min", 786432)
        ia.lib.fmx_set_option(b"plan_sa_key", 2)
    k = 5000  # below sort_min: processed in the caller's order
    for lo in (0, 400_000, (1 << 20) - k):
        c2, s2 = idx32.fm.count_batch(pat[lo * M:(lo + k) * M], off[: k + 1])
        assert (c2 == cnt[lo:lo + k]).all() and (s2 == 0).all()



def test_config1_over_two_replicas_through_the_c_abi(idx32, batch):
    """the 1,048,576-pattern batch sharded over a replica set by the library (fmx_replicate + fmx_count_batch_multi
    / fmx_locate_batch_multi): two replicas on the one GPU of the box = the code path of two GPUs (own images, tables, worker
    threads), equal to the single-device call and to the oracle, entry by entry"""
    pat, off, _ = batch
    rs = ia.ReplicaSet(idx32.fm, [0, 0])
    try:
        res = rs.resident_bytes()
        assert res[0] == res[1] and res[0][0] == idx32.fm.device_blob()[1] and res[0][1] > 0 and res[0][2] > 0
        cnt, st, lf = rs.count_batch(pat, off, want_steps=True)
        c1, s1, lf1 = idx32.fm.count_batch(pat, off, want_steps=True)
        oc, ost = idx32.oracle.count_batch(pat, off, threads=CORES)
        assert (cnt == c1).all() and (st == s1).all() and (lf == lf1).all()
        assert (cnt == oc).all() and (st == ost).all()
        K = 100_001  # configs[2] + 1: an odd batch
        locs, found, st2 = rs.locate_batch(pat[: K * M], off[: K + 1], 16)
        olocs, ofound, ost2 = idx32.oracle.locate_batch(pat[: K * M], off[: K + 1], 16, threads=CORES)
        assert (found == ofound).all() and (st2 == ost2).all() and (locs == olocs).all()
    finally:
        rs.close()


def test_host_builder_and_device_builder_give_the_same_bytes_at_2_to_28(idx32, text256):
    """at 256 MiB the oracle reads the DEVICE builder's index.  The host builder (SA-IS, fmx_build) is an
    independent construction — its serialized bytes equal the device builder's at this size too (byte identity of both with the
    oracle's own builder is asserted up to 16 MiB in test_gpu_parity.py)"""
    import hashlib

    host = ia.FmIndex(text256, 32, True, device=None)
    try:
        assert hashlib.sha256(host.write(False)).hexdigest() == hashlib.sha256(idx32.ser).hexdigest()
    finally:
        host.close()


def test_config2_locate_100k_every_position_in_sa_order(idx32, batch, text256):
    pat, off, _ = batch
    K = 100_000
    locs, found, st, lf = idx32.fm.locate_batch(pat[: K * M], off[: K + 1], 16, want_steps=True)
    orc.counters_reset()
    olocs, ofound, ost = idx32.oracle.locate_batch(pat[: K * M], off[: K + 1], 16, threads=CORES)
    c = orc.counters()
    assert (st == ost).all() and (st == 0).all() and (found == ofound).all()
    live = np.arange(16)[None, :] < found[:, None]
    assert (locs[live] == olocs[live]).all()
    assert (locs[~live] == 0).all()  # slots beyond `found` keep the caller's values
    assert int(lf.astype(np.int64).sum()) == c["lf_steps"]
    # every located position holds its pattern
    P = pat[: K * M].reshape(K, M)
    for k in range(16):
        sel = found > k
        assert (text256[locs[sel, k][:, None] + np.arange(M)[None, :]] == P[sel]).all()


def test_locate_of_the_whole_1m_batch_planned_by_sa_row_and_walked_in_range_order(idx32, batch):
    """the sizes where both orders of round 4 are live under the library's own policy: the batch is planned (>= 786,432 patterns:
    ordered by the estimated SA row of the tabulated suffix) and its hits are walked by the first row of their ranges
    (>= 32,768 patterns) — found, every position, statuses, LF-step total against the oracle for all 1,048,576 patterns"""
    pat, off, _ = batch
    n = len(off) - 1
    assert ia.lib.fmx_count_batch_is_planned(idx32.fm.handle, n) == 1
    locs, found, st, lf = idx32.fm.locate_batch(pat, off, 16, want_steps=True)
    orc.counters_reset()
    olocs, ofound, ost = idx32.oracle.locate_batch(pat, off, 16, threads=CORES)
    c = orc.counters()
    assert (st == ost).all() and (st == 0).all() and (found == ofound).all()
    live = np.arange(16)[None, :] < found[:, None]
    assert (locs[live] == olocs[live]).all()
    assert int(lf.astype(np.int64).sum()) == c["lf_steps"]


def test_count_of_a_4m_batch_on_the_device_entry_point(idx32, text256):
    """a batch four times the headline's through fmx_count_batch_dev (one plan, one k_count launch of 4,194,304 lane pairs)"""
    import ctypes as C

    import torch

    n = 1 << 22
    pat, off, _ = workload.count_batch_patterns(text256, n, M, seed=workload.PATTERN_SEED + 17)
    dev = torch.device("cuda", 0)
    sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    d_pat = torch.from_numpy(pat.view(np.int16)).to(dev)
    d_off = torch.from_numpy(off).to(dev)
    d_cnt = torch.zeros(n, dtype=torch.int32, device=dev)
    d_lf = torch.zeros(n, dtype=torch.int32, device=dev)
    d_st = torch.zeros(n, dtype=torch.int32, device=dev)
    assert ia.lib.fmx_count_batch_is_planned(idx32.fm.handle, n) == 1
    assert ia.lib.fmx_count_batch_dev(idx32.fm.handle, d_pat.data_ptr(), d_off.data_ptr(), n, d_cnt.data_ptr(), d_lf.data_ptr(),
                                      d_st.data_ptr(), sp) == 0
    torch.cuda.synchronize()
    orc.counters_reset()
    oc, ost = idx32.oracle.count_batch(pat, off, threads=CORES)
    c = orc.counters()
    assert (d_cnt.cpu().numpy() == oc).all() and (d_st.cpu().numpy() == ost).all()
    assert int(d_lf.cpu().numpy().astype(np.int64).sum()) == c["lf_steps"]


def test_config3_extract_until_boundary_100k_on_sample_rate_64(text256, batch):
    pat, off, _ = batch
    K = 100_000
    b = Built(text256, 64)
    try:
        locs, found, st = b.fm.locate_batch(pat[: K * M], off[: K + 1], 16)
        assert (st == 0).all() and (found >= 1).all()
        fr = np.ascontiguousarray(locs[:, 0])
        olocs, ofound, _ = b.oracle.locate_batch(pat[: K * M], off[: K + 1], 16, threads=CORES)
        assert (ofound == found).all() and (olocs[:, 0] == fr).all()
        cap = 1024
        dst, ol, st2, aux, lf = b.fm.extract_boundary_batch(fr, "\n", 0, cap, want_steps=True)
        orc.counters_reset()
        odst, ool, ost, oaux = b.oracle.extract_until_boundary_batch(0, fr, "\n", cap, threads=CORES)
        c = orc.counters()
        assert (st2 == ost).all() and (st2 == 0).all() and (ol == ool).all()
        assert (dst == odst).all()  # whole destination rows, untouched tails included
        assert int(lf.astype(np.int64).sum()) > 0 and c["lf_steps"] > 0
        # the rows are the lines of the text around each hit
        nl = np.flatnonzero(text256 == 10)
        for i in range(0, K, 97):
            p = int(fr[i])
            j = int(np.searchsorted(nl, p))
            if j < len(nl) and text256[p] != 10:
                lo = int(nl[j - 1]) + 1 if j > 0 else 0
                assert ol[i] == nl[j] - lo and (dst[i, : ol[i]] == text256[lo: nl[j]]).all(), i
        # the Left / Right variants and a destination too small for most lines (exception + "Currently extracted: N")
        for mode, cap2 in ((1, 1024), (2, 1024), (0, 64)):
            d1, l1, s1, a1 = b.fm.extract_boundary_batch(fr[:20_000], "\n", mode, cap2)
            d2, l2, s2, a2 = b.oracle.extract_until_boundary_batch(mode, fr[:20_000], "\n", cap2, threads=CORES)
            assert (s1 == s2).all() and (l1 == l2).all() and (d1 == d2).all()
            assert (a1[s1 == 8] == a2[s1 == 8]).all()
            if cap2 == 64:
                assert (s1 == 8).sum() > 1000
    finally:
        b.fm.close()


def test_config4_share_over_8_segments_vs_8_oracle_indexes():
    """2 GiB as 8 segment indexes, all resident on this GPU (SURVEY §8e scheme (i)): counts summed over the
    segments and base-shifted hits of the whole per-GPU share of the 8M batch (1,048,576 patterns) against 8 oracle
    indexes; segments 1..7 look their own suffix tables up with translated code words.  Then the same share over two replicas
    of the segment set through the C ABI, and configs[4] held by ONE GPU (the whole 8,388,608-pattern batch in one call)."""
    K = 8
    texts = workload.segment_texts(K, 28)
    sf = workload.build_segment_set(texts, 32, device=0, build_device=0)
    try:
        assert sum(len(t) for t in texts) > (1 << 31) - (1 << 20)  # a text one FmIndex cannot hold
        n = 1 << 20
        pat, off = workload.segment_patterns(texts, n, M)
        # the 8M batch of configs[4] and the sample of it the oracles answer (every 419th pattern + the last one)
        n8, mm = 1 << 23, 16
        pat8, off8 = workload.segment_patterns(texts, n8, M, seed=workload.PATTERN_SEED + 1)
        sample = np.unique(np.concatenate([np.arange(0, n8, 419), [n8 - 1]]))
        assert len(sample) >= 20_000
        spat = np.ascontiguousarray(pat8.reshape(n8, M)[sample]).reshape(-1)
        soff = (np.arange(len(sample) + 1, dtype=np.int64) * M).astype(np.int32)
        assert all(f.suffix_table_info()[0] >= 2 for f in sf.segments)
        cnt, st, lf = sf.count_batch(pat, off, want_steps=True)
        locs, found, st2 = sf.locate_batch(pat, off, mm)

        class Expect:
            def __init__(self, k):
                self.c, self.l, self.f = np.zeros(k, np.int64), np.full((k, mm), -1, np.int64), np.zeros(k, np.int32)

            def add(self, o, p, po, base):
                oc, ost = o.count_batch(p, po, threads=CORES)
                assert (ost == 0).all()
                self.c += oc
                # the caller's loop: segment s looks for the maxMatches - taken hits still missing
                ol, of, _ = o.locate_batch(p, po, mm, threads=CORES)
                for k in range(mm):
                    sel = np.flatnonzero((of > k) & (self.f < mm))
                    self.l[sel, self.f[sel]] = ol[sel, k].astype(np.int64) + base
                    self.f[sel] += 1

        exp, exp8 = Expect(n), Expect(len(sample))
        steps = 0
        for s in range(K):
            o = orc.OracleFmIndex.read(sf.segments[s].write(False))
            orc.counters_reset()
            o.count_batch(pat, off, threads=CORES)
            steps += orc.counters()["lf_steps"]
            exp.add(o, pat, off, int(sf.bases[s]))
            exp8.add(o, spat, soff, int(sf.bases[s]))
            del o
        assert (st == 0).all() and (cnt == exp.c).all() and int(lf.sum()) == steps
        assert (st2 == 0).all() and (found == exp.f).all() and (locs == exp.l).all()
        assert (found == np.minimum(cnt, mm)).all()
        # the same share over TWO replicas of the whole segment set through the C ABI (fmx_count_locate_segments_multi): what a
        # Java host runs on a node's GPUs — here two replicas of all 8 images on the one device
        srs = ia.SegmentReplicaSet(sf, [0, 0])
        try:
            c2, l2, f2, s2, lf2 = srs.count_locate_batch(pat, off, mm)
            assert (s2 == 0).all() and (c2 == exp.c).all() and (f2 == exp.f).all() and (l2 == exp.l).all() and int(lf2.sum()) == steps
        finally:
            srs.close()
        # configs[4] held by ONE GPU: all 8,388,608 patterns in one call (count + locate in one pass over the 8 images): the sample
        # against the oracles, every pattern through what the domain offers (found = min(count, 16); a hit holds its pattern)
        one = SegmentSetOnOneDevice(sf)
        c8, l8, f8, s8 = one.count_locate_batch(pat8, off8, mm)
        assert (s8 == 0).all() and (c8 >= 1).all() and (f8 == np.minimum(c8, mm)).all()
        assert (c8[sample] == exp8.c).all() and (f8[sample] == exp8.f).all() and (l8[sample] == exp8.l).all()
        bases = np.asarray(sf.bases, np.int64)
        P8 = pat8.reshape(n8, M)
        for k in (0, 7, 15):
            sel = np.flatnonzero(f8 > k)[:: 64]
            seg = np.searchsorted(bases, l8[sel, k], side="right") - 1
            for sgm in range(K):
                q = sel[seg == sgm]
                if len(q):
                    at = (l8[q, k] - bases[sgm])[:, None] + np.arange(M)[None, :]
                    assert (texts[sgm][at] == P8[q]).all(), (k, sgm)
    finally:
        for f in sf.segments:
            f.close()


class SegmentSetOnOneDevice:
    """fmx_count_locate_segments (host buffers): count() and locate() of a batch over all segment indexes in one pass"""

    def __init__(self, sf):
        self.sf = sf

    def count_locate_batch(self, pat, off, mm):
        import ctypes as C  # noqa: F401

        n = len(off) - 1
        cnt = np.zeros(n, np.int64)
        locs = np.full((n, mm), -1, np.int64)
        found = np.zeros(n, np.int32)
        st = np.zeros(n, np.int32)
        rc = ia.lib.fmx_count_locate_segments(self.sf.handles, len(self.sf), self.sf.base_array.ctypes.data, pat.ctypes.data, off.ctypes.data,
                                              n, mm, cnt.ctypes.data, None, locs.ctypes.data, found.ctypes.data, st.ctypes.data)
        assert rc == 0, ia.lib.fmx_last_error()
        return cnt, locs, found, st
