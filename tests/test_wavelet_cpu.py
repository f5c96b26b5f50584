"""Stand-alone wavelet tree (fmx_wavelet_build + device code on the host simulation) vs the oracle, incl. the
sequence that drives WaveletFixedBlockBoosting.rank into its run-block quirk (CPU only)."""
import numpy as np

import hostsim
import index4j_amd as ia
import orc
from wavelet_cases import probes, quirk_sequence


def oracle_ranks(o, pos, sym):
    exp = np.zeros(len(pos), np.int64)
    est = np.zeros(len(pos), np.int32)
    st = orc.C.c_int(0)
    for i, (p, s) in enumerate(zip(pos, sym)):
        st.value = 0
        exp[i] = orc.lib().orc_wfbb_rank(o.h, int(p), int(s), orc.C.byref(st))
        est[i] = st.value
    return exp, est


def check(seq, sampling):
    w = ia.WaveletFixedBlockBoosting(seq, sampling, device=None)
    o = orc.Wfbb(seq, sampling)
    h = hostsim.HostSim(w)
    rng = np.random.default_rng(5)
    pos, sym = probes(seq, rng)
    got, st = h.wt_rank_batch(pos, sym)
    orc.counters_reset()
    exp, est = oracle_ranks(o, pos, sym)
    assert (got == exp).all() and (st == est).all()  # incl. Q3: rank(size) with size % 2^20 == 0 -> AIOOBE
    for p in rng.integers(0, len(seq), 500):
        c, r = h.wt_inverse_select(int(p))
        t = o.inverse_select(int(p))
        assert c == (t & 0xFFFF) and (p == 0 or r == (t >> 32))
    return orc.counters(), got, pos, sym


def test_quirk_sequence_reproduces_the_reference_garbage():
    seq = quirk_sequence()
    cnt, got, pos, sym = check(seq, 32)
    assert cnt["quirk_runblock_right"] > 100  # the quirk path really ran
    truth = np.array([(seq[:p] == s).sum() for p, s in zip(np.minimum(pos, len(seq)), sym)])
    assert (got != truth).sum() > 100  # ... and the reference's answers there are not the true ranks


def test_plain_sequences():
    rng = np.random.default_rng(2)
    check(rng.integers(0, 3, 50_000).astype(np.int16), 64)
    check(np.full(30_000, 3, np.int16), 64)
    check(rng.integers(0, 2000, 200_000).astype(np.int16), 16)
