"""The occurrence directory (index4j_amd/csrc/fmx_device.hpp "occurrence directory": what fmx_to_device grows beside a resident
image of at most 256 codes) on the host simulation: made by the very function k_occ_build runs (occ_build_window), attached only
if every window reproduced rank() at its checkpoints, and then what count() takes its ranks from — counts, ranges, statuses and
LF-step counts must still equal the oracle's.  CPU only; the GPU suite runs the same checks on resident indexes under both values
of option occ_cells (tests/test_gpu_parity.py)."""
import random

import numpy as np
import pytest

import hostsim
import index4j_amd as ia
import orc
from common import hdfs_text
from parity_checks import check_all

HD = hdfs_text()
ASCII = "".join(ch if ord(ch) < 128 else "?" for ch in HD)  # the fixture with its multi-byte characters folded: < 256 codes
ATTACHED = []


def make_sim_occ(text, sr):
    h = hostsim.HostSim(ia.FmIndex(text, sr, True, device=None))
    ATTACHED.append(h.attach_occ())
    return h


@pytest.mark.parametrize("sr", [1, 4, 32])
def test_ascii_fixture_through_the_occurrence_directory(sr):
    del ATTACHED[:]
    check_all(make_sim_occ, ASCII, sr, random.Random(300 + sr))
    assert ATTACHED == [1]


def test_every_rank_the_directory_can_be_asked():
    """occ_rank_folded vs the oracle's rank at EVERY position for every symbol of a text with absent symbols, next-block paths and
    run blocks (the quirk paths of the other parity tests)"""
    rng = np.random.default_rng(3)
    parts = []
    for i in range(12):
        parts.append("".join(chr(97 + int(x)) for x in rng.integers(0, 6 + i, 2500)))
        parts.append("zq" * 3000)
    text = "".join(parts)
    f = ia.FmIndex(text, 5, True, device=None)
    o = orc.OracleFmIndex(text, 5, True)
    h = hostsim.HostSim(f)
    attached = h.attach_occ()
    L = f.getInputLength()
    wh = o.wavelet_handle()
    st = orc.C.c_int(0)
    # whatever was decided, counts through the simulation equal the oracle's
    pats = [ia.as_chars(text[s:s + 7]) for s in range(0, L - 8, 97)] + [ia.as_chars("zqzqzq"), ia.as_chars("qqq")]
    ch, off = ia.pack_patterns(pats)
    cnt, stt, lf, rngs = h.count_batch(ch, off)
    oc, ost = o.count_batch(ch, off)
    assert (cnt == oc).all() and (stt == ost).all()
    assert attached == 1  # (this text's quirk paths are values that do not depend on the position: the table holds them)
    sigma = int(np.frombuffer(h.blob, np.uint8)[48:52].view(np.int32)[0])  # BlobHeader.wt_sigma
    for pos in list(range(0, L + 1, 3)) + [L, L + 5]:  # the table itself, symbol by symbol, against the oracle's rank()
        for sym in list(range(1, sigma)) + [sigma, sigma + 3]:
            st.value = 0
            e = orc.lib().orc_wfbb_rank(wh, pos, sym, orc.C.byref(st))
            assert st.value == 0 and h.occ_rank(pos, sym) == e, (pos, sym)


def test_small_texts_sentinels_and_texts_shorter_than_a_window():
    rnd = random.Random(14)
    mod = list(ASCII[:40_000])
    for _ in range(300):
        mod[rnd.randrange(len(mod) - 2)] = "\0"
    check_all(make_sim_occ, "".join(mod), 8, rnd, n_q=60)
    check_all(make_sim_occ, "What a string!\nNow this is long, indeed\nBut others could be longer.", 2, rnd, n_q=40)
    check_all(make_sim_occ, "a", 1, rnd, n_q=5)
    check_all(make_sim_occ, "ab" * 31 + "c", 4, rnd, n_q=20)  # wt_size = 64: rank(wt_size) has a window of its own
    check_all(make_sim_occ, "ab" * 32, 4, rnd, n_q=20)


def test_alphabets_beyond_256_codes_get_no_directory():
    h = hostsim.HostSim(ia.FmIndex(HD[:30_000], 8, True, device=None))
    assert h.attach_occ() == -1


def test_compact_image_through_the_occurrence_directory():
    assert ia.lib.fmx_set_option(b"image_compact", 1) == 0
    try:
        check_all(make_sim_occ, ASCII[:60_000], 32, random.Random(79), n_q=60)
    finally:
        ia.lib.fmx_set_option(b"image_compact", 0)
