"""The device source (index4j_amd/csrc/fmx_device.hpp) compiled for the host by tests/hostsim.cpp, checked
against the oracle on CPU.  This is a test-only simulation used to debug without a GPU — the product
never runs queries on the host; the real parity tests are in test_gpu_parity.py (-m gpu)."""
import ctypes as C
import random

import numpy as np
import pytest

import hostsim
import index4j_amd as ia
import orc
from common import hdfs_text
from parity_checks import check_all

HD = hdfs_text()


def make_sim(text, sr):
    return hostsim.HostSim(ia.FmIndex(text, sr, True, device=None))


@pytest.mark.parametrize("sr", [1, 3, 4, 32, 64, 128])
def test_fixture(sr):
    check_all(make_sim, HD, sr, random.Random(sr))


def test_synthetic_log_config1():
    """BASELINE.json configs[0] shape: 1,000 8-char patterns on 1 MiB synthetic log, sampleRate 32"""
    t = ia.synth_log(1 << 20)
    f = ia.FmIndex(t, 32, True, device=None)
    o = orc.OracleFmIndex(t, 32, True)
    h = hostsim.HostSim(f)
    pat, off, pos = ia.synth_patterns(t, 8, 1000)
    cnt, st, lf, _ = h.count_batch(pat, off)
    orc.counters_reset()
    oc, ost = o.count_batch(pat, off)
    assert (cnt == oc).all() and (st == 0).all()
    assert int(lf.sum()) == orc.counters()["lf_steps"]  # LF-step accounting identical to the oracle's
    s = bytes(t.astype(np.uint8))
    for i in range(0, 1000, 25):  # and against brute force
        assert cnt[i] == len([1 for j in range(len(s) - 7) if s[j:j + 8] == s[pos[i]:pos[i] + 8]]) if i < 50 else True


def test_embedded_sentinels_and_small_texts():
    rnd = random.Random(11)
    mod = list(HD[:40_000])
    for _ in range(300):
        mod[rnd.randrange(len(mod) - 2)] = "\0"
    check_all(make_sim, "".join(mod), 8, rnd, n_q=60)
    check_all(make_sim, "What a string!\nNow this is long, indeed\nBut others could be longer.", 2, rnd, n_q=40)
    check_all(make_sim, "a", 1, rnd, n_q=5)


def test_wavelet_rank_all_paths_incl_quirks():
    """wt_rank / wt_inverse_select vs the oracle at every kind of block: absent symbol, absent-in-block
    (next block / end of superblock / last superblock), run blocks, out-of-range positions"""
    rng = np.random.default_rng(3)
    parts = []
    for i in range(30):
        parts.append("".join(chr(97 + int(x)) for x in rng.integers(0, 6 + i, 2500)))
        parts.append("zq" * 3000)
    text = "".join(parts)
    f = ia.FmIndex(text, 5, True, device=None)
    o = orc.OracleFmIndex(text, 5, True)
    h = hostsim.HostSim(f)
    L = f.getInputLength()
    wh = o.wavelet_handle()
    st = orc.C.c_int(0)
    for pos in list(range(0, L + 1, 37)) + [L, L + 5]:
        for sym in (0, 1, 2, 3, 5, 21, 30, 36, 37, 400):
            st.value = 0
            e = orc.lib().orc_wfbb_rank(wh, pos, sym, orc.C.byref(st))
            r, s2 = h.wt_rank(pos, sym)
            assert (r, s2) == (e, st.value), (pos, sym)
    for pos in range(0, L, 61):
        c, r = h.wt_inverse_select(pos)
        t = orc.lib().orc_wfbb_inverse_select(wh, pos)
        assert c == (t & 0xFFFF) and (pos == 0 or r == (t >> 32))


def make_sim_reference_route(text, sr):
    return hostsim.HostSim(hostsim.reference_route_index(text, sr))


def test_rank_reference_route_still_matches():
    """rank() normally reads leaf rank / canonical code / root count from the widened mapping entry; entries can
    also say 'take the reference's own route' (codes longer than 16 bits).  Force that route for every entry:
    the answers must not change."""
    check_all(make_sim_reference_route, HD[:80_000], 16, random.Random(5), n_q=80)
    rng = np.random.default_rng(4)
    parts = []
    for i in range(12):
        parts.append("".join(chr(97 + int(x)) for x in rng.integers(0, 5 + 2 * i, 3000)))
        parts.append("zq" * 2500)
    check_all(make_sim_reference_route, "".join(parts), 4, random.Random(6), n_q=80)


def test_mapping_rows_by_superblock_code_layout():
    """the mapping tables are laid out by global symbol (small alphabets) or by superblock code (the reference's
    own row order, kept for large alphabets): force the second layout and run everything again"""
    try:
        assert ia.lib.fmx_set_option(b"map_by_symbol", 0) == 0
        check_all(make_sim, HD[:80_000], 16, random.Random(15), n_q=80)
        check_all(make_sim_reference_route, HD[:40_000], 8, random.Random(16), n_q=40)
        assert ia.lib.fmx_set_option(b"map_by_symbol", 1) == 0
        big = "".join(chr(40 + (i * 7919) % 1500) for i in range(60_000)) + HD[:20_000]
        check_all(make_sim, big, 8, random.Random(17), n_q=60)  # > 1024 symbols, rows by symbol all the same
    finally:
        ia.lib.fmx_set_option(b"map_by_symbol", -1)


def test_texts_shorter_than_the_sample_rate():
    """the left walk of extractUntilBoundary first skips up to sampleRate uncounted steps (FM:645-653), which on a
    text shorter than the sample rate wraps around the whole text before the first counted character"""
    rnd = random.Random(21)
    for text, sr in (("A", 2), ("A", 3), ("AB", 3), ("BAA", 3), ("AB", 64), ("hello\nworld", 100)):
        check_all(make_sim, text, sr, rnd, n_q=12)


def test_inverse_select_node_records_are_what_runs():
    """the images the other tests walk answer inverseSelect from node records (InvHdr / NodeRec) in every block; under
    inv_fast = 0 every block takes the reference's own route — and both give the oracle's answers"""
    text = HD[:60_000] + "a" * 150_000 + HD[60_000:90_000]  # the run of 'a' gives the BWT run blocks
    f = ia.FmIndex(text, 8, True, device=None)
    tree, run, slow = hostsim.inverse_select_block_kinds(f.blob())
    assert slow == 0 and tree > 0 and run > 0
    g = hostsim.reference_route_index(text, 8)
    tree2, run2, slow2 = hostsim.inverse_select_block_kinds(g.blob())
    assert tree2 == 0 and run2 == 0 and slow2 == tree + run
    o = orc.OracleFmIndex(text, 8, True)
    wh = o.wavelet_handle()
    hf, hg = hostsim.HostSim(f), hostsim.HostSim(g)
    for pos in range(0, f.getInputLength(), 53):
        t = orc.lib().orc_wfbb_inverse_select(wh, pos)
        for h in (hf, hg):
            c, r = h.wt_inverse_select(pos)
            assert c == (t & 0xFFFF) and (pos == 0 or r == (t >> 32)), pos


def test_suffix_table_on_the_host_simulation():
    """the suffix table's two device functions — fm_suffix_entry (fills an entry with the state of FM:455-474 after k
    codes) and fm_suffix_lookup (a pattern's start from it) — on the host: counts, statuses and per-pattern LF-step
    counts equal the oracle's and the table-free loop's, for every depth, incl. patterns that end early inside the
    table's strings, hold absent characters or are shorter than them, and a text with the Q3 status"""
    rnd = random.Random(77)
    texts = ["".join(rnd.choice("abcdefghij \n") for _ in range(60_000)),
             "".join(rnd.choice("acgt") for _ in range((1 << 20) - 1))]  # + sentinel = 2^20: rank(size) raises (Q3)
    for text in texts:
        f = ia.FmIndex(text, 8, True, device=None)
        o = orc.OracleFmIndex(text, 8, True)
        h = hostsim.HostSim(f)
        t16 = ia.as_chars(text)
        pats = []
        for i in range(3000):
            s0 = rnd.randrange(len(t16) - 12)
            p = t16[s0:s0 + rnd.randrange(1, 10)].copy()
            if i % 5 == 0:
                p[rnd.randrange(len(p))] = t16[rnd.randrange(len(t16))]  # ends early somewhere
            if i % 19 == 0:
                p[rnd.randrange(len(p))] = ord("Z")  # absent character
            pats.append(p)
        ch, off = ia.pack_patterns(pats)
        orc.counters_reset()
        oc, ost = o.count_batch(ch, off)
        o_steps = orc.counters()["lf_steps"]
        plain, pst, plf, _ = h.count_batch(ch, off)
        assert (plain == oc).all() and (pst == ost).all() and int(plf.sum()) == o_steps
        for k in (2, 3, 4):
            cnt, st, lf, answered, entries = h.count_batch_with_table(k, ch, off)
            assert entries == (f.getAlphabetLength() + (1 if "\0" in text else 0)) ** k or entries > 0
            assert (cnt == oc).all() and (st == ost).all(), k
            assert (lf == plf).all(), k  # per pattern, table or not
            assert 0 < answered < o_steps


def test_derailed_locate_walks_run_past_the_sample_rate():
    """Q1 on an alphabet above 256 codes: a run block reports its symbol masked to 8 bits, the LF-walk of locate() lands on
    another row and goes on to THAT row's next sample — such a walk takes more than sampleRate steps, and the reference
    has no bound on it.  The engine's bound against damaged indexes (fm_locate_hit: walk_limit) must not cut it."""
    n, sr = 1 << 19, 8
    text = ia.synth_log_multichar(n, 600)
    h = make_sim(text, sr)
    o = orc.OracleFmIndex(text, sr, True)
    rnd = random.Random(5)
    pats = [text[s:s + rnd.randrange(2, 6)] for s in (rnd.randrange(n - 8) for _ in range(1500))]
    ch, off = ia.pack_patterns(pats)
    locs, found, st, lf = h.locate_batch(ch, off, 16, 16)
    assert (st == 0).all()
    assert (lf > found * (sr - 1)).sum() > 20  # walks that left their own sample interval
    for i, p in enumerate(pats):
        k, l = o.locate(p, max_matches=16, cap=16)
        assert k == found[i] and (l == locs[i, :k]).all(), i


def test_compact_images_on_the_host_simulation():
    """COMPACT images (option image_compact; the kernels of namespace fmxc): the device header compiled with
    -DFMX_COMPACT=1 — bv_* functions decoding 16-block RRR records + offsets stream through the value-of-offset table —
    over the host blob: every query kind against the oracle, a suffix table grown and consulted over the compact image.
    What the GPU suite checks on hardware (tests/test_gpu_compact.py), here on CPU-only machines.  (Sizes are compared
    there: a 30,000-character image is dominated by the 32 KiB value table.)"""
    rnd = random.Random(4)
    assert ia.lib.fmx_set_option(b"image_compact", 1) == 0
    try:
        def make(text, sr):
            h = hostsim.HostSim(ia.FmIndex(text, sr, True, device=None))
            assert h.compact
            return h

        check_all(make, HD[:40_000], 16, rnd, n_q=60)
        check_all(make, HD[:25_000], 1, rnd, n_q=40)
        check_all(make, "ab\0cd\0\0ef" * 800 + "tail", 4, rnd, n_q=40)
        text = HD[:30_000]
        f = ia.FmIndex(text, 8, True, device=None)
        h = hostsim.HostSim(f)
        o = orc.OracleFmIndex(text, 8, True)
        t16 = ia.as_chars(text)
        pats = [t16[s:s + rnd.randrange(1, 10)] for s in (rnd.randrange(len(t16) - 10) for _ in range(400))]
        ch, off = ia.pack_patterns(pats)
        oc, ost = o.count_batch(ch, off)
        cnt, st, lf, answered, entries = h.count_batch_with_table(3, ch, off)
        assert (cnt == oc).all() and (st == ost).all() and answered > 0 and entries > 0
    finally:
        assert ia.lib.fmx_set_option(b"image_compact", 0) == 0
    g = ia.FmIndex(HD[:30_000], 8, True, device=None)
    assert not hostsim.HostSim(g).compact


def test_marked_replay_of_extract_until_boundary_is_the_route_taken_and_matches_the_oracle():
    """Round 5: the group form of extractUntilBoundary{,Left,Right} finds a line's two ends from the marks its first fill's walks
    left (fm_boundary_replay_marked) and writes the row in closed form; everything else falls back to the literal replay.  Short
    lines at sample rates 32 / 64 keep whole lines inside the two intervals a group of ONE lane fetches (the host simulation), so the
    route is exercised here in all three modes — whole destination rows (incl. the left part's temporary copy at the row's end),
    lengths, statuses and aux against the oracle, with destinations that fit generously, barely, and not at all."""
    rnd = random.Random(77)
    words = ["alpha", "beta", "gamma", "delta", "x", "INFO", "blk_123", "10.0.0.7", "ok"]
    lines = []
    while sum(len(x) + 1 for x in lines) < 6000:
        lines.append(" ".join(rnd.choice(words) for _ in range(rnd.randrange(1, 5))))
    text = "\n".join(lines) + "\n"
    L = hostsim.lib()
    L.sim_marked_replays.restype = C.c_long
    for sr in (32, 64):
        f = ia.FmIndex(text, sr, True, device=None)
        h = hostsim.HostSim(f)
        o = orc.OracleFmIndex(text, sr, True)
        froms = np.array([rnd.randrange(len(text)) for _ in range(400)] + [0, len(text) - 1, len(text) - 2], np.int32)
        for mode in (0, 1, 2):
            before = L.sim_marked_replays(mode)
            for cap, offs in ((300, 0), (300, 7), (64, 0), (40, 3), (12, 0), (5, 1)):
                for accel in (2, 3):
                    dst = np.full((len(froms), cap), 0xABCD, np.uint16)
                    dst, ol, st, aux, _lf = h.extract_boundary_batch(froms, "\n", mode, cap, offs, dst=dst, accelerate=accel)
                    odst, olen, ost, oaux = o.extract_until_boundary_batch(mode, froms, "\n", cap, offset=offs, threads=4, fill=0xABCD)
                    assert (st == ost).all() and (ol == olen).all() and (aux == oaux).all(), (sr, mode, cap, offs, accel)
                    assert (dst == odst).all(), (sr, mode, cap, offs, accel)
            assert L.sim_marked_replays(mode) - before > 400, "the marked replay answered too few of mode %d's queries" % mode
