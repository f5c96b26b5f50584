"""The window directory (index4j_amd/csrc/fmx_device.hpp "window directory": what fmx_to_device grows beside a resident image)
on the host simulation: the cells and entries are made by the very functions k_win_build / k_win_other run (win_build_cell,
win_build_other) and the LF-walks of the simulation then take their steps from them, as the kernels do (the instantiation the
launchers pick: kWinAlways over a complete directory, kWinAsk over one that left entries to the tree walk) — results, statuses and LF-step counts must still equal the oracle's, on
the reference's fixture and on the quirk-heavy inputs of the other parity tests.  CPU only; the GPU suite runs the same
checks on resident indexes (tests/test_gpu_parity.py with the library's default option window_cells = 2)."""
import random

import numpy as np
import pytest

import hostsim
import index4j_amd as ia
import orc
from common import hdfs_text
from parity_checks import check_all

HD = hdfs_text()
COVER = {}


ENTRY_BYTES = [0]  # the form of the directory's entries make_sim_windows asks for (0: by the alphabet, as fmx_to_device)


def make_sim_windows(text, sr):
    h = hostsim.HostSim(ia.FmIndex(text, sr, True, device=None))
    got, positions, classes, by_entry, unclean = h.attach_windows(ENTRY_BYTES[0])
    assert got + by_entry == positions  # every position's step is in the directory
    assert (h.window_slots >= 0) == (ENTRY_BYTES[0] in (4, -1) or (ENTRY_BYTES[0] == 0 and h.fm.getAlphabetLength() + 2 <= 2050))
    # four-byte entries: an entry points at an eight-byte slot at least where its step carries a status or `suspect`
    assert h.window_slots < 0 or unclean <= h.window_slots <= by_entry
    COVER[(len(text), sr)] = got / max(1, positions)
    SLOTS[(len(text), sr)] = (h.window_slots, by_entry)
    return h


SLOTS = {}


@pytest.fixture(params=[4, 6, -1])
def entry_bytes(request):
    """both forms of the directory's entries, whatever the alphabet (option window_entry_bytes of the library), and the FLAT form
    of the directory (-1: a word per position instead of cells and entries, option window_cells = 3)"""
    ENTRY_BYTES[0] = request.param
    yield request.param
    ENTRY_BYTES[0] = 0


@pytest.mark.parametrize("sr", [1, 4, 32, 64])
def test_fixture_through_the_windows(sr):
    check_all(make_sim_windows, HD, sr, random.Random(100 + sr))
    # log text: the three most frequent symbols of a 120-position stretch of the BWT hold most of it
    assert COVER[(len(HD), sr)] > 0.5


@pytest.mark.parametrize("sr", [2, 32])
def test_fixture_through_both_forms_of_the_entries(sr, entry_bytes):
    """the fixture has 2,061 symbols: past what the four-byte form is offered for by default, so it is forced here (the symbol of an
    entry then comes out of a search over cumulativeCounts where they lie) beside the six-byte form"""
    check_all(make_sim_windows, HD, sr, random.Random(300 + sr))
    slots, by_entry = SLOTS[(len(HD), sr)]
    if entry_bytes == 4:
        assert 0 <= slots < by_entry // 50  # nearly every entry is the row alone


def test_quirk_texts_through_both_forms_of_the_entries(entry_bytes):
    check_all(make_sim_windows, quirk_text(), 8, random.Random(41), n_q=80)
    check_all(make_sim_windows, "ab" * 55 + "c", 4, random.Random(42), n_q=20)
    rnd = random.Random(43)
    mod = list(HD[:30_000])
    for _ in range(200):
        mod[rnd.randrange(len(mod) - 2)] = "\0"
    check_all(make_sim_windows, "".join(mod), 8, rnd, n_q=60)


def test_sentinels_small_texts_and_texts_shorter_than_a_window():
    rnd = random.Random(12)
    mod = list(HD[:40_000])
    for _ in range(300):
        mod[rnd.randrange(len(mod) - 2)] = "\0"
    check_all(make_sim_windows, "".join(mod), 8, rnd, n_q=60)
    check_all(make_sim_windows, "What a string!\nNow this is long, indeed\nBut others could be longer.", 2, rnd, n_q=40)
    check_all(make_sim_windows, "a", 1, rnd, n_q=5)
    check_all(make_sim_windows, "ab" * 55 + "c", 4, rnd, n_q=20)   # wt_size = 112: rank(wt_size) has a cell of its own
    check_all(make_sim_windows, "ab" * 56, 4, rnd, n_q=20)
    check_all(make_sim_windows, "ab" * 60, 4, rnd, n_q=20)


def quirk_text():
    rng = np.random.default_rng(3)
    parts = []
    for i in range(30):
        parts.append("".join(chr(97 + int(x)) for x in rng.integers(0, 6 + i, 2500)))
        parts.append("zq" * 3000)
    return "".join(parts)


def test_rank_and_inverse_select_at_every_kind_of_block():
    """wt_rank / wt_inverse_select through the windows vs the oracle: absent symbols, next-block paths, run blocks (whose
    symbol the reference masks to 8 bits: never a class), out-of-range positions — at EVERY position for the symbols around"""
    text = quirk_text()
    f = ia.FmIndex(text, 5, True, device=None)
    o = orc.OracleFmIndex(text, 5, True)
    h = hostsim.HostSim(f)
    got, positions, classes, by_entry, unclean = h.attach_windows()
    assert 0 < got <= positions and classes > 0 and by_entry > 0 and got + by_entry == positions
    L = f.getInputLength()
    wh = o.wavelet_handle()
    st = orc.C.c_int(0)
    for pos in list(range(0, L + 1, 7)) + [L, L + 5]:
        for sym in (0, 1, 2, 3, 5, 21, 30, 36, 37, 400):
            st.value = 0
            e = orc.lib().orc_wfbb_rank(wh, pos, sym, orc.C.byref(st))
            r, s2 = h.wt_rank(pos, sym)
            assert (r, s2) == (e, st.value), (pos, sym)
    for pos in range(0, L, 3):
        c, r = h.wt_inverse_select(pos)
        t = orc.lib().orc_wfbb_inverse_select(wh, pos)
        assert c == (t & 0xFFFF) and (pos == 0 or r == (t >> 32))
    for row in range(1, L + 1):  # EVERY step through the directory beside the reference's two calls (tests/test_fused_lf.py)
        out = h.lf_step_both(row)
        assert (out[0], out[1], out[4]) == (out[2], out[3], out[5]), row


def test_large_alphabet_with_run_blocks_of_wide_symbols():
    """symbols >= 256 in run blocks (Q1: inverseSelect reports them masked) and a 1,000-symbol alphabet: the directory must leave
    every such position to the tree walk"""
    rng = np.random.default_rng(9)
    parts = []
    for i in range(12):
        parts.append("".join(chr(0x4E00 + int(x) * 7) for x in rng.integers(0, 900, 1500)))
        parts.append(chr(0x30A1 + i) * 70_000)  # long runs of one wide symbol: run blocks
        parts.append("log line %d\n" % i * 50)
    text = "".join(parts)
    h = hostsim.HostSim(ia.FmIndex(text, 16, True, device=None))
    got, positions, classes, by_entry, unclean = h.attach_windows()
    assert unclean > 0 and got + by_entry == positions  # steps with `suspect` (Q1): kept in their entries as they are
    check_all(make_sim_windows, text, 16, random.Random(5), n_q=60)


def test_compact_image_through_the_windows():
    assert ia.lib.fmx_set_option(b"image_compact", 1) == 0
    try:
        check_all(make_sim_windows, HD[:60_000], 32, random.Random(77), n_q=60)
    finally:
        ia.lib.fmx_set_option(b"image_compact", 0)


def test_reference_route_image_through_the_windows():
    """an image whose every block takes the reference's own routes: the directory is grown from THOSE answers"""
    def make(text, sr):
        h = hostsim.HostSim(hostsim.reference_route_index(text, sr))
        h.attach_windows()
        return h

    check_all(make, HD[:50_000], 8, random.Random(31), n_q=50)


def test_locate_without_instalments_over_the_directory_as_well():
    """option walk_pack = 0: k_locate_walk<kWinAlways> (fm_locate_hit, one walk to its end) instead of k_locate_walk_c (the walk in
    instalments between the workgroup's packings: fm_locate_steps_win + fm_locate_finish_win, what every other test of this file runs
    at sample rates >= 8)"""
    for compact in (False, True):
        hostsim.lib(compact).sim_set_pack(0)
    try:
        check_all(make_sim_windows, HD, 32, random.Random(501))
        check_all(make_sim_windows, quirk_text(), 8, random.Random(502), n_q=80)
    finally:
        for compact in (False, True):
            hostsim.lib(compact).sim_set_pack(1)
