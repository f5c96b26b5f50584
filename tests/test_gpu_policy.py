"""The library's batch policies under its SHIPPED defaults.  tests/conftest.py forces the planned paths on
for the rest of the GPU suite; here nothing is forced: the thresholds themselves are asserted — a wrong constant fails a test
instead of only changing timings — and results on both sides of every threshold are compared with the oracle."""
import numpy as np
import pytest

import index4j_amd as ia
import orc

pytestmark = [pytest.mark.gpu, pytest.mark.plan_policy]

PLAN_SA_MIN = 786_432      # fmx_api.cpp g_plan_sa_min
SORT_MIN = 16_384          # fmx_kernels.hip g_sort_min
WALK_ORDER_MIN = 32_768    # g_walk_order_min
BOUNDARY_ORDER_MIN = 32_768  # g_boundary_order_min
PLAN_MIN_PER_STRING = 16   # g_plan_min_per_string


@pytest.fixture(scope="module")
def index16():
    text = ia.synth_log(1 << 24)  # 16 MiB
    o = orc.OracleFmIndex(text, 32, True)
    fm = ia.FmIndex.read(o.write(False), device=0)
    yield text, fm, o
    fm.close()


def policy(fm, kind, n):
    return ia.lib.fmx_batch_policy(fm.handle, kind, n)


def test_thresholds_are_the_documented_ones(index16):
    _text, fm, _o = index16
    chars, _bytes = fm.suffix_table_info()
    assert chars >= 2, "the 16 MiB index has a suffix table"
    # count(): with a suffix table and the SA-row key, planned from plan_sa_min on — exactly
    assert [policy(fm, 0, n) for n in (SORT_MIN, PLAN_SA_MIN - 1, PLAN_SA_MIN, PLAN_SA_MIN + 1, 1 << 22)] == [0, 0, 1, 1, 1]
    assert ia.lib.fmx_count_batch_is_planned(fm.handle, PLAN_SA_MIN) == 1 and ia.lib.fmx_count_batch_is_planned(fm.handle, PLAN_SA_MIN - 1) == 0
    # locate(): the walk order from walk_order_min on; extractUntilBoundary: text-position order from boundary_order_min on
    assert [policy(fm, 1, n) for n in (1, WALK_ORDER_MIN - 1, WALK_ORDER_MIN, 1 << 20)] == [0, 0, 1, 1]
    assert [policy(fm, 2, n) for n in (1, BOUNDARY_ORDER_MIN - 1, BOUNDARY_ORDER_MIN, 1 << 20)] == [0, 0, 1, 1]
    L = ia.lib
    try:
        # launches told to ignore the table: every batch of sort_min patterns or more is planned (the code key), none below
        assert L.fmx_set_option(b"suffix_table", 0) == 0
        assert [policy(fm, 0, n) for n in (SORT_MIN - 1, SORT_MIN, PLAN_SA_MIN - 1)] == [0, 1, 1]
        assert L.fmx_set_option(b"suffix_table", 1) == 0
        # the code key WITH a table (plan_sa_key 0): planned iff >= plan_min_per_string patterns per string of the deepest level —
        # the flip is at a multiple of 16, above sort_min, and moves with the option
        assert L.fmx_set_option(b"plan_sa_key", 0) == 0
        lo, hi = SORT_MIN, 1 << 30
        assert policy(fm, 0, lo - 1) == 0 and policy(fm, 0, hi) == 1
        while lo < hi:  # first n that is planned
            mid = (lo + hi) // 2
            if policy(fm, 0, mid):
                hi = mid
            else:
                lo = mid + 1
        flip = lo
        assert flip == SORT_MIN or flip % PLAN_MIN_PER_STRING == 0
        strings = flip // PLAN_MIN_PER_STRING
        assert L.fmx_set_option(b"plan_min_per_string", 4) == 0
        if flip > SORT_MIN:
            assert policy(fm, 0, max(SORT_MIN, 4 * strings)) == 1 and (4 * strings - 1 < SORT_MIN or policy(fm, 0, 4 * strings - 1) == 0)
    finally:
        L.fmx_set_option(b"suffix_table", 1)
        L.fmx_set_option(b"plan_sa_key", 2)
        L.fmx_set_option(b"plan_min_per_string", PLAN_MIN_PER_STRING)
    # options move the thresholds, and exactly
    try:
        assert L.fmx_set_option(b"plan_sa_min", 50_000) == 0 and L.fmx_set_option(b"walk_order_min", 1000) == 0
        assert L.fmx_set_option(b"boundary_order_min", 777) == 0
        assert [policy(fm, 0, 49_999), policy(fm, 0, 50_000)] == [0, 1]
        assert [policy(fm, 1, 999), policy(fm, 1, 1000)] == [0, 1]
        assert [policy(fm, 2, 776), policy(fm, 2, 777)] == [0, 1]
    finally:
        L.fmx_set_option(b"plan_sa_min", PLAN_SA_MIN)
        L.fmx_set_option(b"walk_order_min", WALK_ORDER_MIN)
        L.fmx_set_option(b"boundary_order_min", BOUNDARY_ORDER_MIN)


@pytest.mark.parametrize("n", [WALK_ORDER_MIN - 1, WALK_ORDER_MIN])
def test_locate_and_boundary_answers_on_both_sides_of_their_thresholds(index16, n):
    text, fm, o = index16
    pat, off, _pos = ia.synth_patterns(text, 8, n, seed=11)
    assert policy(fm, 1, n) == (1 if n >= WALK_ORDER_MIN else 0) == policy(fm, 2, n)
    locs, found, st, lf = fm.locate_batch(pat, off, 4, 4, want_steps=True)
    olocs, ofound, ost = o.locate_batch(pat, off, 4, threads=8)
    live = np.arange(4)[None, :] < ofound[:, None]
    assert (found == ofound).all() and (st == ost).all() and (locs[live] == olocs[live]).all()
    froms = np.ascontiguousarray(locs[:, 0]).astype(np.int32)
    dst, ol, st2, aux, lf2 = fm.extract_boundary_batch(froms, "\n", 0, 512, 0, want_steps=True)
    odst, olen, ost2, oaux = o.extract_until_boundary_batch(0, froms, "\n", 512, threads=8)
    assert (ol == olen).all() and (st2 == ost2).all() and (dst == odst).all()


@pytest.mark.parametrize("n", [PLAN_SA_MIN - 1, PLAN_SA_MIN])
def test_count_answers_on_both_sides_of_plan_sa_min(index16, n):
    text, fm, o = index16
    pat, off, _pos = ia.synth_patterns(text, 8, n, seed=12)
    assert policy(fm, 0, n) == (1 if n >= PLAN_SA_MIN else 0)
    # the device-pointer entry point (the host-buffer one cuts a batch of this size into chunks, each below the threshold)
    import ctypes as C

    import torch

    dev = torch.device("cuda", 0)
    d_pat = torch.from_numpy(pat.view(np.int16)).to(dev)
    d_off = torch.from_numpy(off).to(dev)
    d_cnt = torch.zeros(n, dtype=torch.int32, device=dev)
    d_lf = torch.zeros(n, dtype=torch.int32, device=dev)
    d_st = torch.zeros(n, dtype=torch.int32, device=dev)
    sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = ia.lib.fmx_count_batch_dev(fm.handle, d_pat.data_ptr(), d_off.data_ptr(), n, d_cnt.data_ptr(), d_lf.data_ptr(), d_st.data_ptr(), sp)
    assert rc == 0, ia.lib.fmx_last_error()
    torch.cuda.synchronize()
    orc.counters_reset()
    oc, ost = o.count_batch(pat, off, threads=8)
    assert (d_cnt.cpu().numpy() == oc).all() and (d_st.cpu().numpy() == ost).all()
    assert int(d_lf.sum(dtype=torch.int64).item()) == orc.counters()["lf_steps"]


@pytest.mark.parametrize("sr", [1, 4, 32, 64])
def test_every_query_kind_under_the_shipped_policy(sr):
    """parity_checks.check_all (every query kind, error statuses, whole destination rows, vs the oracle) with NOTHING forced: small
    batches take k_count in the caller's order with the code word made in the kernel, k_locate_walk without a walk order and
    extractUntilBoundary in the caller's order — the paths a typical caller's small batches run (the rest of the GPU
    suite forces the planned ones)."""
    import random

    from common import hdfs_text
    from parity_checks import GpuEngine, check_all

    hd = hdfs_text()
    check_all(lambda text, s: GpuEngine(text, s), hd, sr, random.Random(900 + sr))
    check_all(lambda text, s: GpuEngine(text, s), "".join(ch if ord(ch) < 128 else "?" for ch in hd[:150_000]), sr, random.Random(910 + sr), n_q=80)
