"""The oracle pinned on the reference's own known-answer tests and fixture (CPU only).

Each test restates a JUnit test of the reference (file:line in the docstring) with literal inputs
and expected values, or checks the oracle against the reference's definitional test oracles
(util/Util.java) re-implemented independently in common.py."""
import hashlib
import json
import os
import random

import numpy as np
import pytest

import orc
from common import GOLDEN, JavaRandom, hdfs_text, naive_sa_order_hits, occurrences, until_boundary, until_boundary_left, until_boundary_right

HD = hdfs_text()


def s_of(arr, n):
    return "".join(map(chr, arr[:n]))


def test_rrr_tables_match_reference_digests():
    """RRR:488-16900 literal tables vs the generated ones (digests minted by tools/check_rrr_tables.py)"""
    g = json.load(open(os.path.join(GOLDEN, "rrr_tables.json")))
    L = orc.lib()
    off = np.ctypeslib.as_array(L.orc_rrr_table_offset_of_value(), shape=(32768,))
    inv = np.ctypeslib.as_array(L.orc_rrr_table_value_of_offset(), shape=(32768,))
    card = np.ctypeslib.as_array(L.orc_rrr_table_cardinality_offsets(), shape=(16,))
    assert hashlib.sha256(off.astype("<u2").tobytes()).hexdigest() == g["offset_of_value_sha256"]
    assert hashlib.sha256(inv.astype("<u2").tobytes()).hexdigest() == g["value_of_offset_sha256"]
    assert card.tolist() == g["cardinality_offsets"]
    bits = np.ctypeslib.as_array(L.orc_rrr_table_bits_needed(), shape=(16,))
    assert bits.tolist() == [1, 4, 7, 9, 11, 12, 13, 13, 13, 13, 12, 11, 9, 7, 4, 1]


def test_rrr_small_bitvector():
    """RrrVectorTest.testEncodeSmallBitVectorAndAnswerQueries (T-RRR:70-100)"""
    bits = np.zeros(1024, np.uint8)
    for i in (0, 2, 11, 18, 19, 20, 199, 512):
        bits[i] = 1
    r = orc.Rrr(bits=bits, sample=32)
    assert [r.access(i) for i in (0, 1, 2, 15, 19, 199, 512)] == [True, False, True, False, True, True, True]
    assert [r.rank_ones(i) for i in range(4)] == [0, 1, 1, 2]
    assert [r.rank_zeroes(i) for i in range(4)] == [0, 0, 1, 1]


def test_rrr_corner_cases():
    """RrrVectorTest.testCornerCases / testBitSequenceRrrAndRankIt / testOutOfBoundsAccess (T-RRR:102-173)"""
    r = orc.Rrr(ints=[5, 1], sample=32)
    assert (r.rank_zeroes(0), r.rank_ones(0), r.rank_zeroes(1), r.rank_ones(1)) == (0, 0, 0, 1)
    assert (r.rank_zeroes(64), r.rank_ones(64)) == (61, 3)
    assert (r.rank_zeroes(-1), r.rank_ones(-1)) == (0, 0)
    bits = [(5 >> i) & 1 for i in range(32)] + [(1 >> i) & 1 for i in range(32)]
    assert [r.access(i) for i in range(64)] == [bool(b) for b in bits]
    r = orc.Rrr(ints=[5], sample=32)
    for bad in (9999, -1):
        with pytest.raises(ValueError):
            r.access(bad)


@pytest.mark.parametrize("sample", [1, 2, 4, 8, 16, 32, 64, 256, 7])
def test_rrr_random_vs_prefix_sums(sample):
    """RrrVectorTest.testEncodeLargeBitVector / ...LargeScaleWithSampleRate (T-RRR:175-249): rank/access vs a plain scan"""
    rng = np.random.default_rng(42 + sample)
    for n, dens in ((1, 0.5), (15, 0.5), (16, 1.0), (1000, 0.25), (100_003, 0.05), (50_000, 0.9)):
        bits = (rng.random(n) < dens).astype(np.uint8)
        r = orc.Rrr(bits=bits, sample=sample)
        ps = np.concatenate([[0], np.cumsum(bits)])
        idx = np.unique(np.concatenate([rng.integers(0, n, 400), [0, n - 1]]))
        for i in idx:
            assert r.rank_ones(int(i)) == ps[i] and r.rank_zeroes(int(i)) == i - ps[i] and r.access(int(i)) == bool(bits[i])
        assert r.rank_ones(n) == ps[n] and r.rank_ones(n + 100) == ps[n]


def test_wfbb_small_text():
    """WaveletFixedBlockBoostingTest.testWaveletRankFromSmallText / ...NonExistingCharacters (T-WFBB:57-68, 79-84)"""
    t = "aloha what a string this is string is eh"
    w = orc.Wfbb(orc.u16(t).astype(np.int16))
    assert w.rank(6, ord("a")) == 2 and w.rank(len(t), ord("a")) == 4
    assert w.rank(len(t), ord("h")) == 4 and w.rank(19, ord("i")) == 1
    assert w.rank(22, ord("Z")) == 0
    w = orc.Wfbb(np.array([ord("a")], np.int16))
    assert w.rank(1, ord("a")) == 1 and w.rank(1, ord("b")) == 0  # testSingleSymbolWavelet T-WFBB:50-55


def test_wfbb_corner_cases():
    """T-WFBB:86-131: all-ones run, out-of-bounds rank, 3M-symbol runs crossing superblocks"""
    w = orc.Wfbb(np.full(100, 1, np.int16))
    assert (w.inverse_select(0) & 0xFFFF) == 1 and (w.inverse_select(5) & 0xFFFF) == 1
    s = np.full(30_000, 3, np.int16)
    s[28_000] = 2
    assert orc.Wfbb(s).rank(90_000, 2) == 1
    s = np.full(3_000_000, 0, np.int16)  # 'b' -> 0, 'a' -> 1 (first-appearance codes, T-UTIL:305-315)
    s[2_800_000] = 1
    assert orc.Wfbb(s).rank(6_900_000, 1) == 1
    s = np.full(3_000_000, 0, np.int16)
    s[100] = 1
    assert orc.Wfbb(s).rank(1_000_000, 1) == 1


def test_wfbb_random_vs_scan():
    """T-WFBB:133-165, 194-235: rank / inverseSelect vs a linear scan, incl. a >700-symbol alphabet (the fixture)"""
    codes = {}
    seq = np.array([codes.setdefault(c, len(codes)) for c in HD], np.int16)
    w = orc.Wfbb(seq)
    rnd = random.Random(42)
    for _ in range(1000):
        pos, sp = rnd.randrange(len(seq)), rnd.randrange(len(seq))
        sym = int(seq[sp])
        assert w.rank(pos, sym) == int((seq[:pos] == sym).sum())
        t = w.inverse_select(sp)
        assert (t & 0xFFFF) == sym and (sp == 0 or (t >> 32) == int((seq[:sp] == sym).sum()))


def test_fm_small_kats():
    """FmIndexTest.shouldCount / ...MultipleSentinels / ...PartialString / ...Sliced / ...NonExisting (T-FM:43-128)"""
    t = "This is a long string\0"
    f = orc.OracleFmIndex(t, 32, False)
    assert f.count("is") == 2
    assert f.count("is a long", 0, 2) == 2 and f.count("is a long", 2, 1) == len(occurrences(t, " "))
    assert f.count("baaa") == 0 and f.locate("baaa", cap=1)[0] == 0
    for p in ("does not exist here", "never seen"):
        assert f.count(p) == 0 and f.locate(p, cap=1)[0] == 0
    t2 = "This \0is a \0long string\0"
    f = orc.OracleFmIndex(t2, 4, True)
    assert f.count("is") == len(occurrences(t2, "is")) and f.count("\0") == 3
    with pytest.raises(ValueError, match="Input has more than 32767 different symbols"):  # T-FM:165-179
        orc.OracleFmIndex(np.arange(32768, dtype=np.uint16), 32, True)


def test_fm_convenience_and_max_matches():
    """T-FM:195-200 (locate INFO cap 100 -> 100; the fixture holds 1,920), T-FM:564-578 (length, alphabet)"""
    f = orc.OracleFmIndex(HD, 32, True)
    assert f.getInputLength() == len(HD) + 1 == 315119
    assert f.getAlphabetLength() == len(set(HD)) + 1 == 763
    assert f.locate("INFO", max_matches=100, cap=100)[0] == 100
    assert f.count("INFO") == 1920


@pytest.mark.parametrize("sr", [1, 2, 4, 8, 16])
def test_fm_count_locate_extract_from_log(sr):
    """T-FM:104-115, 181-193, 360-374 replayed with java.util.Random(42): count == overlapping occurrences,
    sorted(locate) == all occurrence positions, extract == substring"""
    f = orc.OracleFmIndex(HD, sr, True)
    r = JavaRandom(42)
    for _ in range(100):
        start = r.next_int(len(HD) - 32)
        sub = HD[start:start + r.next_int(1, 32)]
        assert f.count(sub) == len(occurrences(HD, sub))
    r = JavaRandom(42)
    for _ in range(100):
        start = r.next_int(0, len(HD) - 32)
        sub = HD[start:start + r.next_int(16, 32)]
        n, locs = f.locate(sub, max_matches=10_000, cap=10_000)
        assert sorted(locs.tolist()) == occurrences(HD, sub)
    r = JavaRandom(42)
    for _ in range(100):
        a = r.next_int(len(HD) - 100)
        b = a + r.next_int(100)
        n, d = f.extract(a, b, dest_len=100)
        assert n == b - a and s_of(d, n) == HD[a:b]
    n, d = f.extract(0, len(HD), dest_len=len(HD))  # T-FM:350-358
    assert s_of(d, n) == HD


@pytest.mark.parametrize("sr", [1, 2, 4, 8, 16])
def test_fm_locate_with_multiple_sentinels(sr):
    """T-FM:202-217: 1,000 embedded '\\0' (alphabet code 1 for '\\0', FM:406-409)"""
    mod = list(HD)
    rnd = random.Random(42)
    for _ in range(1000):
        mod[rnd.randrange(len(mod) - 2)] = "\0"
    mod = "".join(mod)
    f = orc.OracleFmIndex(mod, sr, True)
    for _ in range(100):
        st = rnd.randrange(len(mod) - 32)
        sub = mod[st:st + rnd.randrange(1, 32)]
        n, locs = f.locate(sub, max_matches=-1, cap=100_000)
        assert sorted(locs.tolist()) == occurrences(mod, sub)


def test_fm_truncated_locate_is_suffix_array_order():
    """FM:527-547: a capped locate returns SA rows start+1.. in order — pinned by an independent naive SA"""
    t = (HD[:3000] + "the cat sat on the mat; the cat sat on the hat; ") * 3
    t16 = orc.u16(t)
    code_of = {0: 0}
    for c in t16:
        code_of.setdefault(int(c), len(code_of))
    f = orc.OracleFmIndex(t, 4, True)
    for pat in ("the ", "at", " ", "cat sat", "INFO"):
        n, locs = f.locate(pat, max_matches=7, cap=7)
        assert locs.tolist() == naive_sa_order_hits(t16, orc.u16(pat), code_of, 7)


@pytest.mark.parametrize("seed", [0, 1, 14, 66])
def test_fm_extract_until_boundary_corner_cases(seed):
    """T-FM:376-400 over sampleRate 1..256"""
    s = "What a string!\nNow this is long, indeed\nBut others could be longer."
    sr = 1
    while sr <= 256:
        f = orc.OracleFmIndex(s, sr, True)
        for mode, fn in ((0, until_boundary), (1, until_boundary_left), (2, until_boundary_right)):
            n, d = f.extract_until_boundary(mode, seed, 100, 0, "\n")
            assert s_of(d, n) == fn(s, seed, "\n"), (sr, seed, mode)
        sr <<= 1


@pytest.mark.parametrize("sr", [1, 2, 4, 8, 16])
def test_fm_extract_until_boundary_from_log(sr):
    """T-FM:498-542"""
    f = orc.OracleFmIndex(HD, sr, True)
    r = JavaRandom(42)
    for _ in range(100):
        seed = r.next_int(len(HD) - 100)
        for mode, fn in ((0, until_boundary), (1, until_boundary_left), (2, until_boundary_right)):
            n, d = f.extract_until_boundary(mode, seed, 1 << 15, 0, "\n")
            assert s_of(d, n) == fn(HD, seed, "\n")


def test_fm_error_contract():
    """T-FM:284-348, 402-475: exception types and messages, incl. the pinned 13 / 10 / 11"""
    f = orc.OracleFmIndex(HD, 32, False)
    with pytest.raises(RuntimeError, match="Text recovery not enabled at build time"):
        f.extract(50, 100, dest_len=50)
    with pytest.raises(RuntimeError, match="Text recovery not enabled at build time"):
        f.extract_until_boundary(0, 50, 50, 0, "\n")
    f = orc.OracleFmIndex(HD, 32, True)
    with pytest.raises(RuntimeError, match="Requested position less than 0"):
        f.extract(-5, 100, dest_len=50)
    with pytest.raises(RuntimeError, match="Stop position longer than index string"):
        f.extract(len(HD) + 1, len(HD) + 51, dest_len=50)
    with pytest.raises(RuntimeError, match="Supplied destination is not large enough"):
        f.extract(50, 100, dest_len=10)
    with pytest.raises(RuntimeError, match="Requested position less than 0"):
        f.extract_until_boundary(0, -5, 50, 0, "\n")
    with pytest.raises(RuntimeError, match="Requested position longer than index string"):
        f.extract_until_boundary(0, len(HD) + 1, 50, 0, "\n")
    for mode in (0, 1, 2):
        with pytest.raises(ValueError, match="Boundary does not exist"):
            f.extract_until_boundary(mode, 50, 50, 0, "이")
    with pytest.raises(ValueError, match="Supplied destination for extraction has size zero"):
        f.extract_until_boundary(0, 50, 0, 0, "\n")
    for mode, n in ((0, 13), (1, 10), (2, 11)):
        with pytest.raises(RuntimeError, match="Currently extracted: %d$" % n):
            f.extract_until_boundary(mode, 50, 10, 0, "\n")


def test_fm_two_first_log_lines():
    """T-FM:477-496"""
    f = orc.OracleFmIndex(HD, 32, True)
    d = np.zeros(300, np.uint16)
    n, _ = f.extract_until_boundary(0, 5, 300, 0, "\n", dest=d)
    d[n] = 10
    n += 1
    m, _ = f.extract_until_boundary(0, n + 2, 300, n, "\n", dest=d)
    n += m
    exp = ("081109 203533 44 INFO root: this file should have 2061 unique characters, including 3 and 4 byte UTF8 encoded"
           "\n081109 203615 148 INFO dfs.DataNode$PacketResponder: PacketResponder 1 for block "
           "blk_38865049064139660 由电画留當疾療発 terminating")
    assert s_of(d, n) == exp


@pytest.mark.parametrize("sr", [1, 8])
def test_fm_serialize_round_trip(sr):
    """T-FM:219-242, 544-562: write -> read -> identical bytes and identical answers (framed and raw)"""
    f = orc.OracleFmIndex(HD, sr, True)
    framed, raw = f.write(True), f.write(False)
    assert framed[:4] == b"\xac\xed\x00\x05" and framed[4] == 0x7A
    for blob in (framed, raw):
        g = orc.OracleFmIndex.read(blob)
        assert g.write(False) == raw and g.count("INFO") == 1920
    with pytest.raises(IOError):
        orc.OracleFmIndex.read(b"\x01" + raw[1:])  # SER:46-56 version check


def test_convert_byte_pattern():
    """T-FM:130-163"""
    dest = np.zeros(3, np.uint16)
    bad = orc.C.c_int(0)
    p = np.array([ord("a"), 0b11110000, 0b10000000, 0b10000000, 0b10000000, ord("c")], np.uint8)
    assert orc.lib().orc_convert_byte_pattern(p.ctypes.data, 0, 6, dest.ctypes.data, orc.C.byref(bad)) == 3
    p = np.array([ord("a"), 0b11110111, 0b10111000, 0b10111000, 0b10111000, ord("c")], np.uint8)
    assert orc.lib().orc_convert_byte_pattern(p.ctypes.data, 0, 6, dest.ctypes.data, orc.C.byref(bad)) == -1
    assert bad.value == 2068024
    s = "héllo 由电 wörld"
    b = np.frombuffer(s.encode("utf-8"), np.uint8)
    dest = np.zeros(32, np.uint16)
    n = orc.lib().orc_convert_byte_pattern(b.ctypes.data, 0, len(b), dest.ctypes.data, orc.C.byref(bad))
    assert s_of(dest, n) == s
