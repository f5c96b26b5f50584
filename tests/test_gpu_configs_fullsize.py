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
    indexes; segments 1..7 look their own suffix tables up with translated code words"""
    K = 8
    texts = workload.segment_texts(K, 28)
    sf = workload.build_segment_set(texts, 32, device=0, build_device=0)
    try:
        assert sum(len(t) for t in texts) > (1 << 31) - (1 << 20)  # a text one FmIndex cannot hold
        n = 1 << 20
        pat, off = workload.segment_patterns(texts, n, M)
        assert all(f.suffix_table_info()[0] >= 2 for f in sf.segments)
        cnt, st, lf = sf.count_batch(pat, off, want_steps=True)
        locs, found, st2 = sf.locate_batch(pat, off, 16)
        exp_c = np.zeros(n, np.int64)
        exp_l = np.full((n, 16), -1, np.int64)
        exp_f = np.zeros(n, np.int32)
        steps = 0
        for s in range(K):
            o = orc.OracleFmIndex.read(sf.segments[s].write(False))
            orc.counters_reset()
            oc, ost = o.count_batch(pat, off, threads=CORES)
            steps += orc.counters()["lf_steps"]
            assert (ost == 0).all()
            exp_c += oc
            # the caller's loop: segment s looks for the maxMatches - taken hits still missing
            ol, of, _ = o.locate_batch(pat, off, 16, threads=CORES)
            for k in range(16):
                sel = np.flatnonzero((of > k) & (exp_f < 16))
                exp_l[sel, exp_f[sel]] = ol[sel, k].astype(np.int64) + int(sf.bases[s])
                exp_f[sel] += 1
            del o
        assert (st == 0).all() and (cnt == exp_c).all() and int(lf.sum()) == steps
        assert (st2 == 0).all() and (found == exp_f).all() and (locs == exp_l).all()
        assert (found == np.minimum(cnt, 16)).all()
    finally:
        for f in sf.segments:
            f.close()
