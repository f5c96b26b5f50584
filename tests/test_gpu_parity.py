"""Parity tests proper: the HIP kernels, called through the C ABI (libfmx.so), against the oracle, the
committed fixture, brute force, and — at BASELINE.json's sizes — size-independent properties.
Run with `-m gpu` on an MI355X."""
import ctypes as C
import os
import random

import numpy as np
import pytest

import index4j_amd as ia
from index4j_amd import workload
import orc
from common import JavaRandom, hdfs_text, occurrences, until_boundary, until_boundary_left, until_boundary_right
from parity_checks import GpuEngine, check_all

pytestmark = pytest.mark.gpu
HD = hdfs_text()


def make_gpu(text, sr):
    return GpuEngine(text, sr)


def test_native_library_is_the_one_answering():
    assert ia.lib.fmx_device_count() >= 1
    assert os.path.basename(ia.LIB_PATH) == "libfmx.so" and os.path.exists(ia.LIB_PATH)
    f = ia.FmIndex("abracadabra", 2, True, device=0)
    p, n = f.device_blob()
    assert p and n == len(f.blob())
    assert f.count("abra") == 2


@pytest.mark.parametrize("sr", [1, 3, 4, 32, 64, 128])
def test_fixture_all_queries_vs_oracle(sr):
    check_all(make_gpu, HD, sr, random.Random(sr))


def test_literal_right_walk_gives_the_same_answers():
    """extractUntilBoundary's accelerated right walk (one walk per sample interval) vs the literal +4-chunk
    form of FM:692-758, both against the oracle"""
    try:
        assert ia.lib.fmx_set_option(b"boundary_accel", 0) == 0
        check_all(make_gpu, HD[:120_000], 32, random.Random(77), n_q=80)
        check_all(make_gpu, HD[:60_000], 64, random.Random(78), n_q=60)
    finally:
        ia.lib.fmx_set_option(b"boundary_accel", 1)


@pytest.mark.parametrize("group", [0, 1, 2, 4, 8, 16])
def test_group_cooperative_extract_gives_the_same_answers(group):
    """extractUntilBoundary with G lanes per query (each lane walks a different sample interval) vs the oracle"""
    try:
        assert ia.lib.fmx_set_option(b"boundary_group", group) == 0
        check_all(make_gpu, HD[:150_000], 32, random.Random(90 + group), n_q=100)
        check_all(make_gpu, HD[:80_000], 64, random.Random(91 + group), n_q=80)
        check_all(make_gpu, HD[:50_000], 3, random.Random(92 + group), n_q=60)
    finally:
        ia.lib.fmx_set_option(b"boundary_group", 4)


def test_embedded_sentinels_and_small_texts():
    rnd = random.Random(11)
    mod = list(HD[:40_000])
    for _ in range(300):
        mod[rnd.randrange(len(mod) - 2)] = "\0"
    check_all(make_gpu, "".join(mod), 8, rnd, n_q=60)
    check_all(make_gpu, "What a string!\nNow this is long, indeed\nBut others could be longer.", 2, rnd, n_q=40)
    check_all(make_gpu, "a", 1, rnd, n_q=5)


def test_config1_count_1000_patterns_on_1mib_log():
    """BASELINE.json configs[0]: GPU == oracle == brute force, LF-step accounting identical"""
    t = ia.synth_log(1 << 20)
    fm = ia.FmIndexBuilder().setSampleRate(32).build(t, device=0)
    o = orc.OracleFmIndex(t, 32, True)
    assert fm.write() == o.write()
    pat, off, pos = ia.synth_patterns(t, 8, 1000)
    cnt, st, lf = fm.count_batch(pat, off, want_steps=True)
    orc.counters_reset()
    oc, ost = o.count_batch(pat, off)
    assert (cnt == oc).all() and (st == 0).all() and int(lf.sum()) == orc.counters()["lf_steps"]
    s = bytes(t.astype(np.uint8))
    for i in range(0, 1000, 10):
        assert cnt[i] == len(occurrences(s, s[pos[i]:pos[i] + 8]))


def test_scalar_api_reads_like_the_reference_tests():
    """FmIndexTest shouldCount / shouldLocateMaxNumberOfMatches / shouldExtractTwoFirstLogLines /
    shouldAttemptExtraction... (T-FM:43-102, 195-200, 284-348, 402-496) against the GPU engine"""
    text = "This is a long string\0"
    fmi = ia.FmIndexBuilder().setEnableExtraction(False).build(text)
    assert fmi.count("is") == 2
    assert fmi.count("is a long", 0, 2) == 2 and fmi.count("is a long", 2, 1) == 4
    assert fmi.count("baaa") == 0 and fmi.locate("baaa", np.zeros(1, np.int32)) == 0
    with pytest.raises(RuntimeError, match="Text recovery not enabled at build time"):
        fmi.extract(5, 10, np.zeros(50, np.uint16), 0)
    fmi = ia.FmIndex("This \0is a \0long string\0", 4)
    assert fmi.count("is") == 2 and fmi.count("\0") == 3
    fmi = ia.FmIndexBuilder().build(HD)
    assert fmi.getInputLength() == len(HD) + 1 and fmi.getAlphabetLength() == len(set(HD)) + 1
    assert fmi.toString() == "FMIndex-sampleRate:32-extract:true"
    assert fmi.locate("INFO", np.zeros(100, np.int32), 0, 4, 100) == 100
    with pytest.raises(RuntimeError, match="Requested position less than 0"):
        fmi.extract(-5, 100, np.zeros(50, np.uint16), 0)
    with pytest.raises(RuntimeError, match="Stop position longer than index string"):
        fmi.extract(len(HD) + 1, len(HD) + 51, np.zeros(50, np.uint16), 0)
    with pytest.raises(RuntimeError, match="Supplied destination is not large enough"):
        fmi.extract(50, 100, np.zeros(10, np.uint16), 0)
    with pytest.raises(RuntimeError, match="Requested position longer than index string"):
        fmi.extractUntilBoundary(len(HD) + 1, np.zeros(50, np.uint16), 0, "\n")
    for fn in (fmi.extractUntilBoundary, fmi.extractUntilBoundaryLeft, fmi.extractUntilBoundaryRight):
        with pytest.raises(ValueError, match="Boundary does not exist"):
            fn(50, np.zeros(50, np.uint16), 0, "이")
    with pytest.raises(ValueError, match="Supplied destination for extraction has size zero"):
        fmi.extractUntilBoundary(50, np.zeros(0, np.uint16), 0, "\n")
    for fn, n in ((fmi.extractUntilBoundary, 13), (fmi.extractUntilBoundaryLeft, 10), (fmi.extractUntilBoundaryRight, 11)):
        with pytest.raises(RuntimeError, match="Currently extracted: %d$" % n):
            fn(50, np.zeros(10, np.uint16), 0, "\n")
    dest = np.zeros(300, np.uint16)
    n = fmi.extractUntilBoundary(5, dest, 0, "\n")
    dest[n] = 10
    n += 1
    n += fmi.extractUntilBoundary(n + 2, dest, n, "\n")
    assert ia.chars_to_str(dest[:n]) == (
        "081109 203533 44 INFO root: this file should have 2061 unique characters, including 3 and 4 byte UTF8 encoded\n"
        "081109 203615 148 INFO dfs.DataNode$PacketResponder: PacketResponder 1 for block blk_38865049064139660 "
        "由电画留當疾療発 terminating")


@pytest.mark.parametrize("sr", [1, 2, 4, 8, 16])
def test_reference_random_loops_vs_definitional_oracles(sr):
    """T-FM:104-115, 181-193, 360-374, 498-542 replayed (java.util.Random(42)) against brute force"""
    fmi = ia.FmIndexBuilder().setSampleRate(sr).build(HD)
    r = JavaRandom(42)
    subs = []
    for _ in range(100):
        start = r.next_int(len(HD) - 32)
        subs.append(HD[start:start + r.next_int(1, 32)])
    ch, off = ia.pack_patterns(subs)
    cnt, st = fmi.count_batch(ch, off)
    assert cnt.tolist() == [len(occurrences(HD, s)) for s in subs]
    locs, found, st = fmi.locate_batch(ch, off, -1, 100_000)  # T-FM:211: int[100_000], maxMatches -1
    for i, s in enumerate(subs):
        assert st[i] == 0 and sorted(locs[i, :found[i]].tolist()) == occurrences(HD, s)
    locs, found, st = fmi.locate_batch(ch, off, -1, 50)  # overflowing `locations`: Java raises AIOOBE
    for i, s in enumerate(subs):
        n_occ = len(occurrences(HD, s))
        assert (st[i] == 9 and found[i] == 50) if n_occ > 50 else (st[i] == 0 and found[i] == n_occ)
    r = JavaRandom(42)
    a = np.zeros(100, np.int32)
    b = np.zeros(100, np.int32)
    for i in range(100):
        a[i] = r.next_int(len(HD) - 100)
        b[i] = a[i] + r.next_int(100)
    dst, ol, st = fmi.extract_batch(a, b, 100)
    for i in range(100):
        assert ol[i] == b[i] - a[i] and ia.chars_to_str(dst[i, :ol[i]]) == HD[a[i]:b[i]]
    r = JavaRandom(42)
    seeds = np.array([r.next_int(len(HD) - 100) for _ in range(100)], np.int32)
    for mode, fn in ((0, until_boundary), (1, until_boundary_left), (2, until_boundary_right)):
        dst, ol, st, aux = fmi.extract_boundary_batch(seeds, "\n", mode, 1 << 15)
        for i in range(100):
            assert st[i] == 0 and ia.chars_to_str(dst[i, :ol[i]]) == fn(HD, int(seeds[i]), "\n")
    dest = np.zeros(len(HD), np.uint16)
    assert fmi.extract(0, len(HD), dest, 0) == len(HD) and ia.chars_to_str(dest) == HD  # T-FM:350-358


def test_serialized_round_trip_then_query():
    """T-FM:219-242, 544-562: an index4j stream (here minted by the oracle's writer) loaded and queried on the GPU"""
    o = orc.OracleFmIndex(HD, 8, False)
    fmi = ia.FmIndex.read(o.write(True))
    assert fmi.write(True) == o.write(True)
    rnd = random.Random(3)
    subs = [HD[s:s + rnd.randrange(1, 32)] for s in (rnd.randrange(len(HD) - 32) for _ in range(100))]
    ch, off = ia.pack_patterns(subs)
    locs, found, st = fmi.locate_batch(ch, off, -1, 100_000)
    for i, s in enumerate(subs):
        assert sorted(locs[i, :found[i]].tolist()) == occurrences(HD, s)


def test_wavelet_quirk_paths_through_count():
    """texts whose BWT holds run blocks and rare symbols: every absent-symbol / next-block path of WFBB.rank"""
    rng = np.random.default_rng(3)
    parts = []
    for i in range(30):
        parts.append("".join(chr(97 + int(x)) for x in rng.integers(0, 6 + i, 2500)))
        parts.append("zq" * 3000)
    text = "".join(parts)
    check_all(make_gpu, text, 5, random.Random(5), n_q=150)


def test_large_batch_properties_16mib():
    """size-independent properties on a 16 MiB index with a 262,144-pattern batch: every sampled substring
    occurs (count >= 1), count == located hits when under the cap, every located position really holds
    the pattern, extracting a located hit returns the pattern, LF-steps == 2*(m-1) for surviving patterns"""
    n = 1 << 24
    t = ia.synth_log(n)
    fm = ia.FmIndex(t, 32, True, device=0)
    m, N = 8, 1 << 18
    pat, off, pos = ia.synth_patterns(t, m, N)
    cnt, st, lf = fm.count_batch(pat, off, want_steps=True)
    assert (st == 0).all() and (cnt >= 1).all() and (lf == 2 * (m - 1)).all()
    K = 20000
    locs, found, st2 = fm.locate_batch(pat[: K * m], off[: K + 1], 16)
    assert (found == np.minimum(cnt[:K], 16)).all()
    P = pat.reshape(N, m)
    for k in range(16):
        sel = found > k
        idx = locs[sel, k]
        got = t[idx[:, None] + np.arange(m)[None, :]]
        assert (got == P[:K][sel]).all()
    dst, ol, st3 = fm.extract_batch(locs[:, 0], locs[:, 0] + m, m)
    assert (st3 == 0).all() and (dst == P[:K]).all()
    # checksum of checksums against the oracle on a sample, and determinism across launches
    o = orc.OracleFmIndex.read(fm.write(False))
    oc, _ = o.count_batch(pat[: 4000 * m], off[:4001])
    assert (oc == cnt[:4000]).all()
    cnt2, _ = fm.count_batch(pat, off)
    assert (cnt2 == cnt).all()
    # extractUntilBoundary round trip: the line around each hit, vs a numpy scan for '\\n'.  Inside the
    # unterminated LAST line the reference itself returns a length one short (FM:745-752, docs/DESIGN_HISTORY.md Q12),
    # so those seeds are compared with the oracle instead of the scan.
    fr = np.concatenate([locs[:2000, 0], np.arange(n - 40, n, 3)]).astype(np.int32)
    dst, ol, st4, aux = fm.extract_boundary_batch(fr, "\n", 0, 1024)
    nl = np.flatnonzero(t == 10)
    for i in range(len(fr)):
        p = int(fr[i])
        j = np.searchsorted(nl, p)
        if j < len(nl):
            lo = nl[j - 1] + 1 if j > 0 else 0
            exp = t[lo:nl[j]] if t[p] != 10 else t[0:0]
            assert st4[i] == 0 and ol[i] == len(exp) and (dst[i, :ol[i]] == exp).all(), i
        else:
            en, ed = o.extract_until_boundary(0, p, 1024, 0, "\n")
            assert st4[i] == 0 and ol[i] == en and (dst[i] == ed).all(), i


def test_standalone_wavelet_kats_and_quirk_sequence():
    """WaveletFixedBlockBoostingTest on the GPU (T-WFBB:50-131) + the sequence that drives rank() into its
    run-block quirk (WFBB:1081): the kernels must return the reference's garbage bit for bit"""
    from wavelet_cases import probes, quirk_sequence
    from test_wavelet_cpu import oracle_ranks

    t = "aloha what a string this is string is eh"
    w = ia.WaveletFixedBlockBoosting(t)
    assert w.rank(6, "a") == 2 and w.rank(len(t), "a") == 4 and w.rank(len(t), "h") == 4 and w.rank(19, "i") == 1
    assert w.rank(22, "Z") == 0
    w = ia.WaveletFixedBlockBoosting("a")
    assert w.rank(1, "a") == 1 and w.rank(1, "b") == 0
    with pytest.raises(ValueError, match="Input length must be > 0"):
        ia.WaveletFixedBlockBoosting("")
    w = ia.WaveletFixedBlockBoosting(np.full(100, 1, np.int16))
    assert w.inverseSelect(0) == 1 and (w.inverseSelect(5) & 0xFFFF) == 1
    s = np.full(30_000, 3, np.int16)
    s[28_000] = 2
    assert ia.WaveletFixedBlockBoosting(s).rank(90_000, 2) == 1
    s = np.full(3_000_000, 0, np.int16)
    s[2_800_000] = 1
    assert ia.WaveletFixedBlockBoosting(s).rank(6_900_000, 1) == 1
    s = np.full(3_000_000, 0, np.int16)
    s[100] = 1
    assert ia.WaveletFixedBlockBoosting(s).rank(1_000_000, 1) == 1
    for seq, sampling in ((quirk_sequence(), 32), (np.random.default_rng(2).integers(0, 2000, 200_000).astype(np.int16), 16)):
        w = ia.WaveletFixedBlockBoosting(seq, sampling)
        o = orc.Wfbb(seq, sampling)
        pos, sym = probes(seq, np.random.default_rng(5))
        got, st = w.rank_batch(pos, sym)
        orc.counters_reset()
        exp, est = oracle_ranks(o, pos, sym)
        assert (got == exp).all() and (st == est).all()
        p2 = np.random.default_rng(6).integers(0, len(seq), 4000)
        packed, st2 = w.inverse_select_batch(p2)
        assert (st2 == 0).all() and packed.tolist() == [o.inverse_select(int(p)) for p in p2]
    assert orc.counters()["rank_calls"] > 0


_ORC_CODE = {(exc, msg.split("%")[0]): code for code, (exc, msg) in orc.MESSAGES.items()}


def _orc_status(fn):
    """run an oracle call; return (status code, result) in the ABI's terms"""
    try:
        return 0, fn()
    except Exception as e:  # noqa: BLE001 - the oracle raises the reference's exception kinds
        for (exc, prefix), code in _ORC_CODE.items():
            if type(e) is exc and str(e).startswith(prefix):
                return code, None
        raise


@pytest.mark.parametrize("sr,group", [(32, 4), (4, 0), (64, 8)])
def test_locate_extract_pipeline_vs_oracle(sr, group):
    """locate -> extract and locate -> extractUntilBoundary{,Left,Right} fused on the device, against the
    oracle composing the same scalar calls the reference's locateAndExtractBenchmark makes (J-FM:231-249)"""
    rnd = random.Random(1000 + sr)
    text = HD[:100_000]
    fm = ia.FmIndex(text, sr, True, device=0)
    o = orc.OracleFmIndex(text, sr, True)
    t16 = ia.as_chars(text)
    L = len(t16)
    pats = [t16[s:s + rnd.randrange(1, 20)] for s in (rnd.randrange(L - 20) for _ in range(60))]
    pats += [t16[L - 5:], t16[L - 30:L - 20], ia.as_chars("zzzzqq"), ia.as_chars("\n"), ia.as_chars("INFO")]
    ch, off = ia.pack_patterns(pats)
    off = np.concatenate([off, [off[-1]]]).astype(np.int32)  # plus one EMPTY pattern -> AIOOBE in locate
    inlen = o.getInputLength()
    try:
        assert ia.lib.fmx_set_option(b"boundary_group", group) == 0
        for mm, xlen in ((5, 64), (1, 7), (12, 0)):
            r = fm.locate_extract_batch(ch, off, mm, xlen, fill=0xBEEF)
            assert r["status"][-1] == 9 and r["found"][-1] == 0
            for i, p in enumerate(pats):
                n, locs = o.locate(p, max_matches=mm, cap=mm)
                assert r["status"][i] == 0 and r["found"][i] == n and (r["locs"][i, :n] == locs).all()
                for k in range(mm):
                    if k >= n:  # slot without a hit: untouched
                        assert r["locs"][i, k] == -1 and r["out_len"][i, k] == -1 and (r["dst"][i, k] == 0xBEEF).all()
                        continue
                    exp = np.full(xlen, 0xBEEF, np.uint16)
                    st, res = _orc_status(lambda: o.extract(int(locs[k]), min(inlen, int(locs[k]) + xlen), dest=exp))
                    assert r["hit_status"][i, k] == st, (i, k, st)
                    assert (r["dst"][i, k] == exp).all()
                    if st == 0:
                        assert r["out_len"][i, k] == res[0]
        for mode, mm, dlen in ((0, 4, 512), (1, 3, 300), (2, 3, 300), (0, 2, 40)):
            r = fm.locate_lines_batch(ch, off, mm, "\n", dlen, mode=mode, fill=0xBEEF)
            for i, p in enumerate(pats):
                n, locs = o.locate(p, max_matches=mm, cap=mm)
                assert r["status"][i] == 0 and r["found"][i] == n and (r["locs"][i, :n] == locs).all()
                for k in range(mm):
                    if k >= n:
                        assert r["out_len"][i, k] == -1 and (r["dst"][i, k] == 0xBEEF).all()
                        continue
                    exp = np.full(dlen, 0xBEEF, np.uint16)
                    aux = [0]

                    def call():
                        try:
                            return o.extract_until_boundary(mode, int(locs[k]), dlen, 0, "\n", dest=exp)
                        except RuntimeError as e:
                            if "Currently extracted" in str(e):
                                aux[0] = int(str(e).rsplit(" ", 1)[1])
                            raise
                    st, res = _orc_status(call)
                    assert r["hit_status"][i, k] == st, (mode, i, k, st, r["hit_status"][i, k])
                    assert (r["dst"][i, k] == exp).all(), (mode, i, k)
                    if st == 0:
                        assert r["out_len"][i, k] == res[0]
                    if st == 8:
                        assert r["hit_aux"][i, k] == aux[0]
    finally:
        ia.lib.fmx_set_option(b"boundary_group", 4)


def test_locate_lines_pipeline_properties_16mib():
    """grep on a 16 MiB index: every returned line contains the pattern at the located column, lines of
    different hits of one pattern differ in position, and the fused result equals the two-call result"""
    n = 1 << 24
    t = ia.synth_log(n)
    fm = ia.FmIndex(t, 32, True, device=0)
    m, N, mm = 12, 20000, 4
    pat, off, pos = ia.synth_patterns(t, m, N)
    r = fm.locate_lines_batch(pat, off, mm, "\n", 512)
    locs2, found2, st2 = fm.locate_batch(pat, off, mm)
    assert (r["found"] == found2).all() and (r["status"] == 0).all()
    nl = np.flatnonzero(t == 10)
    P = pat.reshape(N, m)
    for k in range(mm):
        sel = np.flatnonzero(r["found"] > k)
        assert (r["locs"][sel, k] == locs2[sel, k]).all()
        loc = r["locs"][sel, k].astype(np.int64)
        j = np.searchsorted(nl, loc)
        ok = (j < len(nl)) & (t[loc] != 10)  # hits inside the unterminated last line: Q12, covered elsewhere
        lo = np.where(j > 0, nl[np.maximum(j - 1, 0)] + 1, 0)
        hi = nl[np.minimum(j, len(nl) - 1)]
        sel, loc, lo, hi = sel[ok], loc[ok], lo[ok], hi[ok]
        assert (r["hit_status"][sel, k] == 0).all()
        assert (r["out_len"][sel, k] == hi - lo).all()
        col = loc - lo
        rows = r["dst"][sel, k]
        got = rows[np.arange(len(sel))[:, None], col[:, None] + np.arange(m)[None, :]]
        fits = col + m <= hi - lo  # pattern may straddle the newline
        assert (got[fits] == P[sel][fits]).all()
    # separate extract call on the located positions gives the same rows
    fr = r["locs"][:, 0][r["found"] > 0]
    dst, ol, st4, aux = fm.extract_boundary_batch(fr, "\n", 0, 512)
    assert (dst == r["dst"][r["found"] > 0, 0]).all() and (ol == r["out_len"][r["found"] > 0, 0]).all()


def test_segment_set_count_and_locate_vs_oracle_and_brute_force():
    """a text as K segment indexes (BASELINE configs[4]'s shape): summed counts and base-shifted
    hits against K oracle indexes queried one by one, and against a brute-force scan of the whole text for
    patterns that cannot span a cut (no '\\n' inside)"""
    rnd = random.Random(4242)
    text = HD[:200_000]
    t16 = ia.as_chars(text)
    sf = ia.SegmentedFmIndex(text, 16, True, device=0, segment_chars=30_000)
    K = len(sf)
    assert K >= 7 and sf.bases[0] == 0
    ends = sf.bases[1:] + [len(t16)]
    for a, b in zip(sf.bases, ends):
        assert 0 < b - a <= 30_000 and (b == len(t16) or t16[b - 1] == 10)
    oracles = [orc.OracleFmIndex(t16[a:b], 16, True) for a, b in zip(sf.bases, ends)]
    for s, o in zip(sf.segments, oracles):
        assert s.write(False) == o.write(False)
    pats = [t16[s:s + rnd.randrange(1, 16)] for s in (rnd.randrange(len(t16) - 16) for _ in range(300))]
    pats += [ia.as_chars("INFO"), ia.as_chars("\n"), ia.as_chars("zzqq"), ia.as_chars("e")]
    ch, off = ia.pack_patterns(pats)
    off = np.concatenate([off, [off[-1]]]).astype(np.int32)  # an EMPTY pattern: AIOOBE in every segment
    cnt, st, lf = sf.count_batch(ch, off, want_steps=True)
    assert st[-1] == 9 and (st[:-1] == 0).all()
    for i, p in enumerate(pats):
        per = [o.count(p) for o in oracles]
        assert cnt[i] == sum(per), i
        if 10 not in p:
            assert cnt[i] == len(occurrences(text, ia.chars_to_str(p))), i
    for mm in (1, 5, 40):
        locs, found, st2 = sf.locate_batch(ch, off, mm)
        assert st2[-1] == 9 and found[-1] == 0 and (locs[-1] == -1).all()
        for i, p in enumerate(pats):
            exp = []
            for o, base in zip(oracles, sf.bases):
                k, l = o.locate(p, max_matches=mm, cap=mm)
                exp.extend(int(x) + base for x in l)
            exp = exp[:mm]
            assert found[i] == len(exp) and list(locs[i, :found[i]]) == exp, (mm, i)
            assert (locs[i, found[i]:] == -1).all()
            for x in exp:
                assert (t16[x:x + len(p)] == p).all()
    # count() and locate() of the batch in ONE pass over the segments (fmx_count_locate_segments_dev): the outputs of the two calls
    import torch

    dev = torch.device("cuda", 0)
    n, mm = len(off) - 1, 5
    d_pat = torch.from_numpy(np.ascontiguousarray(ch).view(np.int16)).to(dev)
    d_off = torch.from_numpy(off).to(dev)
    d_cnt = torch.zeros(n, dtype=torch.int64, device=dev)
    d_lf = torch.zeros(n, dtype=torch.int64, device=dev)
    d_locs = torch.full((n * mm,), -1, dtype=torch.int64, device=dev)
    d_found = torch.zeros(n, dtype=torch.int32, device=dev)
    d_st = torch.zeros(n, dtype=torch.int32, device=dev)
    d_tmp = torch.zeros(n * (4 + mm), dtype=torch.int32, device=dev)
    sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = ia.lib.fmx_count_locate_segments_dev(sf.handles, K, sf.base_array.ctypes.data, d_pat.data_ptr(), d_off.data_ptr(), n, mm,
                                              d_cnt.data_ptr(), d_lf.data_ptr(), d_locs.data_ptr(), d_found.data_ptr(), d_st.data_ptr(),
                                              d_tmp.data_ptr(), sp)
    assert rc == 0, ia.lib.fmx_last_error()
    torch.cuda.synchronize()
    locs, found, st2 = sf.locate_batch(ch, off, mm)
    assert (d_cnt.cpu().numpy() == cnt).all() and (d_lf.cpu().numpy() == lf).all() and (d_st.cpu().numpy() == st).all()
    got = d_locs.cpu().numpy().reshape(n, mm)
    live = np.arange(mm)[None, :] < found[:, None]
    assert (d_found.cpu().numpy() == found).all() and (got[live] == locs[live]).all()


def test_segment_set_with_mixed_alphabet_sizes_long_patterns():
    """segment 0 is plain ASCII (8-bit code words in its plan), segment 1 holds more than 256 symbols, segment 2 is ASCII
    again: the batch's plan is made with segment 0 and reused by the others, so a segment whose codes need 16 bits must not
    take the 8-bit translated route (codes >= 256 were OR-ed into 8-bit fields for patterns longer than the
    record's code word).  Patterns of 9..31 chars, a batch large enough for the planned path; counts and located hits
    against one oracle per segment"""
    rnd = random.Random(777)
    n = 60_000
    ascii_a = np.array([rnd.randrange(97, 123) if rnd.random() < 0.93 else 10 for _ in range(n)], dtype=np.uint16)
    wide = np.array([rnd.randrange(0x400, 0x400 + 600) if rnd.random() < 0.5 else rnd.randrange(97, 123) for _ in range(n)],
                    dtype=np.uint16)
    wide[::50] = 10
    ascii_b = np.array([rnd.randrange(97, 110) if rnd.random() < 0.9 else 10 for _ in range(n)], dtype=np.uint16)
    parts = [ascii_a, wide, ascii_b]
    segs = [ia.FmIndex(t, 16, True, device=0) for t in parts]
    assert segs[0].getAlphabetLength() <= 256 < segs[1].getAlphabetLength() and segs[2].getAlphabetLength() <= 256
    bases = np.cumsum([0] + [len(t) for t in parts[:-1]])
    sf = ia.SegmentedFmIndex.from_segments(segs, bases)
    oracles = [orc.OracleFmIndex(t, 16, True) for t in parts]
    N = 30_000
    pats = []
    for i in range(N):
        t = parts[i % 3]
        s0 = rnd.randrange(len(t) - 32)
        p = t[s0:s0 + rnd.randrange(9, 32)].copy()
        if i % 7 == 0:  # a tail from another segment's alphabet: long, mostly absent
            o = parts[(i + 1) % 3]
            k = rnd.randrange(1, 6)
            p[-k:] = o[s0:s0 + k]
        pats.append(p)
    ch, off = ia.pack_patterns(pats)
    cnt, st, lf = sf.count_batch(ch, off, want_steps=True)
    exp = np.zeros(N, np.int64)
    for o in oracles:
        oc, ost = o.count_batch(ch, off, threads=8)
        assert int(ost.max()) == 0
        exp += oc
    assert (st == 0).all()
    bad = np.flatnonzero(cnt != exp)
    assert len(bad) == 0, (len(bad), bad[:5], cnt[bad[:5]], exp[bad[:5]])
    assert int(exp.sum()) >= N // 2  # the batch really matches
    locs, found, st2 = sf.locate_batch(ch, off, 3)
    for i in range(0, N, 29):
        e = []
        for o, base in zip(oracles, bases):
            k, l = o.locate(pats[i], max_matches=3, cap=3)
            e.extend(int(x) + int(base) for x in l)
        e = e[:3]
        assert found[i] == len(e) and list(locs[i, :found[i]]) == e, i


@pytest.mark.parametrize("sigma", [40, 255, 256, 257, 700, 1100, 4090, 4096, 4097, 4110])
def test_planned_batches_vs_oracle_across_alphabet_sizes(sigma):
    """batches large enough to take the planned path (suffix order + per-pattern code words: 8 codes of 8 bits, 5 of 12 bits —
    alphabets of 257 .. 4,096 codes, round 6 — or 4 of 16 bits, by alphabet size; sizes on both sides of both thresholds), with
    patterns shorter and longer than the planned codes, absent characters and empty patterns — counts, statuses, LF-steps and
    located hits against the oracle; with the default suffix table and with one grown as deep as its keys allow (5 characters of
    12 bits); and, for the 12-bit alphabets, the same index under option code_bits_12 = 0 (16-bit words and keys)"""
    rnd = random.Random(sigma)
    n = 150_000
    arr = np.array([rnd.randrange(1, sigma) if rnd.random() < 0.9 else 10 for _ in range(n)], dtype=np.uint16) + 32
    o = orc.OracleFmIndex(arr, 16, True)
    N = 20_000  # > sort_min
    pats = []
    for i in range(N):
        s = rnd.randrange(n - 16)
        p = arr[s:s + rnd.randrange(1, 15)].copy()
        if i % 11 == 0:
            p[rnd.randrange(len(p))] = 7  # a character the text does not have
        if i % 13 == 0:
            p[0] = arr[rnd.randrange(n)]  # mostly zero matches
        pats.append(p)
    ch, off = ia.pack_patterns(pats)
    off = np.concatenate([off, [off[-1]]]).astype(np.int32)  # + an EMPTY pattern
    orc.counters_reset()
    oc, ost = o.count_batch(ch, off, threads=8)
    steps = orc.counters()["lf_steps"]
    wide = 258 <= o.getAlphabetLength() <= 4000  # (safely inside the 12-bit range)
    variants = [(8, 1)] + ([(0, 1)] if sigma > 256 or wide else []) + ([(0, 0)] if wide else [])  # (table's image fraction, code_bits_12)
    depths = {}
    try:
        for frac, bits12 in variants:
            assert ia.lib.fmx_set_option(b"suffix_table_image_fraction", frac) == 0 and ia.lib.fmx_set_option(b"code_bits_12", bits12) == 0
            fm = ia.FmIndex(arr, 16, True, device=0)
            assert fm.getAlphabetLength() == o.getAlphabetLength()
            # the default rule's form of the window directory: flat (a word per position) while the symbol search runs over
            # cumulativeCounts in LDS (up to 2,048 symbols), cells with six-byte entries beyond
            per_position = _window_bytes(fm) / (len(arr) + 1)
            assert (3.99 < per_position < 4.1) if sigma + 2 <= 2050 else (0.57 < per_position < 6.6 and not 3.99 < per_position < 4.1), (sigma, per_position)
            depths[(frac, bits12)] = fm.suffix_table_info()[0]
            cnt, st, lf = fm.count_batch(ch, off, want_steps=True)
            assert (st == ost).all() and (cnt == oc).all() and st[-1] == 9, (sigma, frac, bits12)
            assert int(lf.astype(np.int64).sum()) == steps, (sigma, frac, bits12)
            locs, found, st2 = fm.locate_batch(ch, off, 4)
            for i in range(0, N, 37):
                k, l = o.locate(pats[i], max_matches=4, cap=4)
                assert found[i] == k and (locs[i, :k] == l).all(), (sigma, frac, bits12, i)
            fm.close()
    finally:
        ia.lib.fmx_set_option(b"suffix_table_image_fraction", 8)
        ia.lib.fmx_set_option(b"code_bits_12", 1)
    if wide:  # five characters fit a 12-bit key, four a 16-bit one
        assert depths[(0, 1)] == 5 and depths[(0, 0)] == 4, depths


def test_suffix_table_changes_nothing_but_the_time():
    """the suffix table (SA interval of every string of k codes that occurs, grown and hashed when the index becomes resident)
    lets count / locate batches skip their first 2 * (k - 1) rank evaluations: counts, statuses, PER-PATTERN LF-step counts
    and located hits must be what they are without it, and what the oracle says — for every table depth the budget
    yields, on patterns that end early at every depth, hold absent characters, are shorter than the table's strings,
    and on a text whose length is a multiple of 2^20 (rank(size) raises in the JVM: Q3 -> entries that say 'ask')"""
    rnd = random.Random(4242)
    texts = [ia.synth_log(1 << 18), np.array([rnd.randrange(33, 33 + 12) for _ in range((1 << 20) - 1)], dtype=np.uint16)]
    try:
        for text in texts:
            o = orc.OracleFmIndex(text, 16, True)
            N = 24_000
            pats = []
            for i in range(N):
                s0 = rnd.randrange(len(text) - 16)
                p = np.array(text[s0:s0 + rnd.randrange(1, 12)], dtype=np.uint16)
                if i % 7 == 0:
                    p[rnd.randrange(len(p))] = text[rnd.randrange(len(text))]  # ends early somewhere
                if i % 17 == 0:
                    p[rnd.randrange(len(p))] = 7  # absent character
                pats.append(p)
            ch, off = ia.pack_patterns(pats)
            orc.counters_reset()
            oc, ost = o.count_batch(ch, off, threads=1)
            o_steps = orc.counters()["lf_steps"]
            seen = set()
            ref = None
            # (mb, depth, image fraction): the size limit is the smaller of the budget and image / fraction (0: budget alone)
            for mb, depth, frac in ((0, 4, 0), (256, 2, 0), (256, 3, 0), (256, 4, 0), (256, 6, 0), (1, 8, 0), (256, 8, 8)):
                assert ia.lib.fmx_set_option(b"suffix_table_mb", mb) == 0
                assert ia.lib.fmx_set_option(b"suffix_table_chars", depth) == 0
                assert ia.lib.fmx_set_option(b"suffix_table_image_fraction", frac) == 0
                fm = ia.FmIndex.read(o.write(False), device=0)
                k, nbytes = fm.suffix_table_info()
                # the depth asked for, unless the size limit stops the growth earlier (1 MB: 45,875 strings at 0.7 load)
                assert (k == 0) == (mb == 0) and k <= depth and (k == 0 or nbytes >= 16 * 1024)
                assert mb != 256 or frac or k == depth
                assert k == 0 or nbytes <= (mb << 20)
                if frac:  # the default policy: an eighth of the image (64 KiB at least)
                    assert nbytes <= max(fm.device_blob()[1] // frac, 64 << 10) and k >= 2
                seen.add(k)
                for use in ((1,) if k == 0 else (1, 0)):
                    assert ia.lib.fmx_set_option(b"suffix_table", use) == 0
                    cnt, st, lf = fm.count_batch(ch, off, want_steps=True)
                    assert (cnt == oc).all() and (st == ost).all(), (mb, use)
                    assert int(lf.sum()) == o_steps, (mb, use)
                    if ref is None:
                        ref = lf.copy()
                    assert (lf == ref).all(), (mb, use)  # per pattern, table or not
                    locs, found, st2 = fm.locate_batch(ch, off, 3)
                    for i in range(0, N, 97):
                        try:
                            kk, ll = o.locate(pats[i], max_matches=3, cap=3)
                            assert st2[i] == 0 and found[i] == kk and (locs[i, :kk] == ll).all(), (mb, use, i)
                        except IndexError:  # Q3
                            assert st2[i] == 9
            assert len(seen) >= 5  # no table, depths 2, 3, 4, 6 (and what a 1 MB budget allows)
    finally:
        ia.lib.fmx_set_option(b"suffix_table_mb", 256)
        ia.lib.fmx_set_option(b"suffix_table_chars", 8)
        ia.lib.fmx_set_option(b"suffix_table_image_fraction", 8)
        ia.lib.fmx_set_option(b"suffix_table", 1)


def test_an_unknown_character_at_the_top_of_the_table_key_does_not_find_the_shorter_string():
    """The table holds every string of 2 .. k codes; a key whose top code is 0 (a character the text does not have, or
    one outside the alphabet) spells the tabulated string that is one character shorter.  Such a pattern ends at that
    character with count 0 (FM:466-468) — it must not come back with the shorter string's interval (found by
    tools/fuzz_gpu.py seed 41 case 336 in round 4: [7, c, c, c] counted like [c, c, c] — one pattern of 20,000: the two
    keys are equal but hash to different homes, so the lookup only meets the shorter string where their probe paths cross;
    the fix never looks such a key up).  Planned batches, 8-bit and
    16-bit code words, the unknown character at every distance from the pattern's end, patterns shorter and longer than
    the table; counts, statuses, LF-steps and located hits against the oracle."""
    rnd = random.Random(336)
    wide = np.array([rnd.randrange(0x400, 0x400 + 500) if rnd.random() < 0.6 else rnd.randrange(97, 110) for _ in range(200_000)],
                    dtype=np.uint16)
    for text in (ia.synth_log(1 << 18), wide):
        o = orc.OracleFmIndex(text, 32, True)
        assert ia.lib.fmx_set_option(b"suffix_table_image_fraction", 0) == 0  # (a small index: the default policy stops at 2 characters)
        assert ia.lib.fmx_set_option(b"suffix_table_chars", 4) == 0
        try:
            fm = ia.FmIndex.read(o.write(False), device=0)
        finally:
            ia.lib.fmx_set_option(b"suffix_table_image_fraction", 8)
            ia.lib.fmx_set_option(b"suffix_table_chars", 8)
        k = fm.suffix_table_info()[0]
        assert k == 4
        pats = []
        for i in range(24_000):
            m = 2 + i % 11
            s0 = rnd.randrange(len(text) - 16)
            p = np.array(text[s0:s0 + m], dtype=np.uint16)
            if i % 2 == 0:
                p[m - 1 - (i // 2) % m] = 7 if i % 4 == 0 else 0xFFF0  # every distance from the end, incl. the key's top
            pats.append(p)
        ch, off = ia.pack_patterns(pats)
        orc.counters_reset()
        oc, ost = o.count_batch(ch, off, threads=8)
        steps = orc.counters()["lf_steps"]
        cnt, st, lf = fm.count_batch(ch, off, want_steps=True)
        bad = np.flatnonzero((cnt != oc) | (st != ost))
        assert len(bad) == 0, (k, len(bad), [(pats[i].tolist(), int(cnt[i]), int(oc[i])) for i in bad[:4]])
        assert int(lf.astype(np.int64).sum()) == steps
        locs, found, st2 = fm.locate_batch(ch, off, 2)
        for i in range(0, 24_000, 53):
            kk, ll = o.locate(pats[i], max_matches=2, cap=2)
            assert st2[i] == 0 and found[i] == kk and (locs[i, :kk] == ll).all(), i
        fm.close()


@pytest.mark.plan_policy
def test_the_plan_stage_is_skipped_where_it_does_not_pay_and_nothing_else_changes():
    """fmx_count_batch_is_planned.  Round 4: with the deeper suffix table the plan stage only pays for large batches —
    ordered by SA row (an estimate from the table's two-character strings; option plan_sa_key 2, 1 = the table's own answer)
    from plan_sa_min = 786,432 patterns on; ordered by the trailing characters' codes (plan_sa_key 0) while a batch holds at
    least plan_min_per_string = 16 patterns per string of the table's deepest level.  Counts, statuses, LF-steps and
    located hits are the same under every order; fmx_count_plan_dev still plans when asked."""
    import torch

    text = ia.synth_log(1 << 22)
    o = orc.OracleFmIndex(text, 32, True)
    assert ia.lib.fmx_set_option(b"suffix_table_image_fraction", 0) == 0
    assert ia.lib.fmx_set_option(b"suffix_table_chars", 4) == 0
    try:
        fm = ia.FmIndex.read(o.write(False), device=0)  # 4 characters: ~26,000 strings at the deepest level
    finally:
        ia.lib.fmx_set_option(b"suffix_table_image_fraction", 8)
        ia.lib.fmx_set_option(b"suffix_table_chars", 8)
    assert fm.suffix_table_info()[0] == 4
    is_planned = lambda n: ia.lib.fmx_count_batch_is_planned(fm.handle, n)
    assert is_planned(1000) == 0 and is_planned(60_000) == 0 and is_planned(700_000) == 0 and is_planned(1 << 20) == 1
    assert ia.lib.fmx_set_option(b"plan_sa_key", 0) == 0  # the code-key order: 2.3 and 40 patterns per string
    assert is_planned(1000) == 0 and is_planned(60_000) == 0 and is_planned(1 << 20) == 1 and is_planned(500_000) == 1
    assert ia.lib.fmx_set_option(b"plan_sa_key", 2) == 0
    n = 60_000
    pat, off, _ = ia.synth_patterns(text, 8, n, seed=11)
    orc.counters_reset()
    oc, ost = o.count_batch(pat, off, threads=8)
    steps = orc.counters()["lf_steps"]
    results = []
    try:
        # the policy (caller's order), then the plan stage forced under each of its three orders
        for sa_min, sa_key, planned in ((786432, 2, 0), (0, 2, 1), (0, 1, 1), (0, 0, 0)):
            assert ia.lib.fmx_set_option(b"plan_sa_min", sa_min) == 0 and ia.lib.fmx_set_option(b"plan_sa_key", sa_key) == 0
            if sa_key == 0:
                assert ia.lib.fmx_set_option(b"plan_min_per_string", 0) == 0
                planned = 1
            assert is_planned(n) == planned
            cnt, st, lf = fm.count_batch(pat, off, want_steps=True)
            assert (cnt == oc).all() and (st == ost).all() and int(lf.astype(np.int64).sum()) == steps
            locs, found, st2 = fm.locate_batch(pat, off, 4)
            results.append((locs.copy(), found.copy()))
    finally:
        ia.lib.fmx_set_option(b"plan_min_per_string", 16)
        ia.lib.fmx_set_option(b"plan_sa_min", 786432)
        ia.lib.fmx_set_option(b"plan_sa_key", 2)
    live = np.arange(4)[None, :] < results[0][1][:, None]
    for r in results[1:]:
        assert (results[0][1] == r[1]).all() and (results[0][0][live] == r[0][live]).all()
    # an explicit plan request is honoured whatever the policy says
    dev = torch.device("cuda", 0)
    d_pat = torch.from_numpy(pat.view(np.int16)).to(dev)
    d_off = torch.from_numpy(off).to(dev)
    d_cnt = torch.zeros(n, dtype=torch.int32, device=dev)
    plan = C.c_void_p()
    sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert ia.lib.fmx_count_plan_dev(fm.handle, d_pat.data_ptr(), d_off.data_ptr(), n, C.byref(plan), sp) == 0 and plan.value
    assert ia.lib.fmx_count_ordered_dev(fm.handle, d_pat.data_ptr(), d_off.data_ptr(), plan, n, d_cnt.data_ptr(), None, None, sp) == 0
    torch.cuda.synchronize()
    assert (d_cnt.cpu().numpy() == oc).all()
    fm.close()


def test_device_construction_is_byte_identical():
    """fmx_build_on_device — suffix array by prefix doubling (FM:329-394) AND the wavelet tree with its RRR vectors
    (FM:173; WFBB:130-154, 362-535, 570-991; RRR:225-286) encoded in HBM — against the ORACLE's builder and the host
    builder: the serialized indexes must be the same bytes.
    Texts: the fixture (763 codes), long repeats (many doubling rounds), a single repeated character (run blocks
    only: empty bit vectors), embedded sentinels, tiny inputs, DNA, alphabets of 1,960 and 1,100 codes (encoded in HBM
    over superblock-local codes since round 4), a superblock with more distinct symbols than a wave's LDS holds (host
    fallback), extraction on and off."""
    rnd = random.Random(99)
    texts = [
        (HD[:120_000], 32), (HD[:50_000], 1), (HD[:33_333], 7),
        ("zq" * 20_000 + "x" + "zq" * 20_000, 16),
        ("a" * 70_000, 8),
        ("ab\0cd\0\0ef" * 3000 + "tail", 4),
        ("", 4), ("a", 1), ("ab", 2), ("\0", 3), ("abracadabra", 2),
        ("".join(rnd.choice("ACGT") for _ in range(200_000)), 64),
        ("".join(chr(rnd.randrange(40, 900)) for _ in range(90_000)), 32),
        ("".join(chr(rnd.randrange(40, 2000)) for _ in range(90_000)), 32),   # 1,960 codes, all in one superblock: host encoder
        # 1,960 codes, at most ~1,000 of them in any one superblock (the BWT groups by context): device encoder
        ("".join(chr(40 + (i >> 11) % 40 * 49 + rnd.randrange(49)) for i in range(2_400_000)), 32),
        (workload.reference_text(21), 32),  # the reference's data-set shape: ~1,100 symbols
        # characters above the alphabet pass's direct LDS tables (hashed slots, collisions to the global tables)
        ("".join(chr(0x4E00 + int(rnd.expovariate(0.004)) % 5000) if rnd.random() < 0.8 else " " for _ in range(120_000)), 16),
        ("".join(chr(97 + min(25, int(rnd.expovariate(0.9)))) for _ in range(1_200_000)), 32),  # skewed, two superblocks
    ]
    for text, sr in texts:
        for extract in (True, False):
            dev = ia.FmIndex(text, sr, extract, device=None, build_device=0)
            expect = orc.OracleFmIndex(text, sr, extract).write(False)
            assert dev.write(False) == expect, (len(text), sr, extract)
            on_device = dev.build_stats["wavelet_device_seconds"] > 0
            codes = len(set(text)) + 1  # the terminator's code 0 + one per distinct character (FM:396-435)
            # the device encoder takes every text whose superblocks each hold at most 1,536 distinct symbols
            if codes <= 1536:
                assert on_device, (len(text), codes)
            if len(text) <= (1 << 20) and codes > 1537:
                assert not on_device, (len(text), codes)
    try:  # the host encoder behind the device suffix-array stage (option): same bytes
        assert ia.lib.fmx_set_option(b"wavelet_on_device", 0) == 0
        dev = ia.FmIndex(HD[:120_000], 32, True, device=None, build_device=0)
        assert dev.build_stats["wavelet_device_seconds"] == 0
        assert dev.write(False) == ia.FmIndex(HD[:120_000], 32, True, device=None).write(False)
    finally:
        ia.lib.fmx_set_option(b"wavelet_on_device", 1)
    dev = ia.FmIndex("a" * 70_000, 8, True, device=0, build_device=0)
    assert dev.build_stats["doubling_rounds"] >= 12  # LCP ~ n: log2(70000 / 16) rounds (16 one-bit codes in the first key)
    assert dev.count("aaaa") == 70_000 - 3
    # at size: 16 MiB of synthetic log
    t = ia.synth_log(1 << 24)
    dev = ia.FmIndex(t, 32, True, device=None, build_device=0)
    assert dev.build_stats["wavelet_device_seconds"] > 0
    assert dev.write(False) == orc.OracleFmIndex(t, 32, True).write(False)


def test_standalone_rrr_vector_kats_and_random_vs_oracle():
    """RrrVectorTest on the GPU (T-RRR:70-249): the compressed RrrVector (records + offset stream + value table in
    LDS) as a stand-alone structure — known answers, corner cases, and random vectors at every sample size
    against the oracle and a plain prefix sum"""
    bits = np.zeros(1024, np.uint8)
    for i in (0, 2, 11, 18, 19, 20, 199, 512):
        bits[i] = 1
    r = ia.RrrVector(bits, 32)  # T-RRR:70-100
    assert [r.access(i) for i in (0, 1, 2, 15, 19, 199, 512)] == [True, False, True, False, True, True, True]
    assert [r.rankOnes(i) for i in range(4)] == [0, 1, 1, 2]
    assert [r.rankZeroes(i) for i in range(4)] == [0, 0, 1, 1]
    b64 = np.array([(5 >> i) & 1 for i in range(32)] + [(1 >> i) & 1 for i in range(32)], np.uint8)
    r = ia.RrrVector(b64, 32)  # T-RRR:102-173
    assert (r.rankZeroes(0), r.rankOnes(0), r.rankZeroes(1), r.rankOnes(1)) == (0, 0, 0, 1)
    assert (r.rankZeroes(64), r.rankOnes(64)) == (61, 3) and (r.rankZeroes(-1), r.rankOnes(-1)) == (0, 0)
    for bad in (9999, -1):
        with pytest.raises(ValueError):
            r.access(bad)
    with pytest.raises(ia.FmxError):  # an RrrVector handle is not an FM-index
        ia.lib.fmx_input_length  # (attribute exists)
        ia._lib.check(ia.lib.fmx_count_batch(r._h, None, None, 0, None, None, None), "fmx_count_batch")
    for sample in (1, 2, 7, 32, 256):  # T-RRR:175-249
        rng = np.random.default_rng(42 + sample)
        for n, dens in ((1, 0.5), (15, 0.5), (16, 1.0), (1000, 0.25), (100_003, 0.05), (50_000, 0.9), (300_000, 0.5)):
            b = (rng.random(n) < dens).astype(np.uint8)
            g = ia.RrrVector(b, sample)
            o = orc.Rrr(bits=b, sample=sample)
            ps = np.concatenate([[0], np.cumsum(b)])
            pos = np.unique(np.concatenate([rng.integers(0, n, 3000), [0, n - 1, n, n + 100, -1]])).astype(np.int32)
            ranks = g.rank_ones_batch(pos)
            acc, st = g.access_batch(pos)
            inside = (pos >= 0) & (pos < n)
            assert (ranks[inside] == ps[pos[inside]]).all() and (acc[inside] == b[pos[inside]]).all()
            assert (st[inside] == 0).all() and (st[~inside] == 9).all()
            for p_ in pos[:50].tolist() + pos[-5:].tolist():
                assert ranks[list(pos).index(p_)] == o.rank_ones(int(p_))


def test_layout_variants_give_the_same_answers():
    """the image has two mapping layouts (rows by global symbol / by superblock code) and the kernels two ways
    to reach a superblock's header (LDS cache / HBM), inverseSelect two routes (node records / the reference's own
    walk over the block headers): combinations against the oracle"""
    try:
        for by_symbol, cache, inv_fast in ((0, 320, 1), (1, 0, 0), (0, 0, 1), (1, 320, 0)):
            assert ia.lib.fmx_set_option(b"map_by_symbol", by_symbol) == 0
            assert ia.lib.fmx_set_option(b"sb_cache_limit", cache) == 0
            assert ia.lib.fmx_set_option(b"inv_fast", inv_fast) == 0
            check_all(make_gpu, HD[:90_000], 16, random.Random(300 + by_symbol + cache), n_q=80)
        t = ia.synth_log(1 << 21)  # three superblocks, planned batch
        fm = ia.FmIndex(t, 32, True, device=0)
        o = orc.OracleFmIndex.read(fm.write(False))
        pat, off, pos = ia.synth_patterns(t, 8, 20000)
        cnt, st = fm.count_batch(pat, off)
        oc, _ = o.count_batch(pat, off, threads=8)
        assert (cnt == oc).all() and (st == 0).all()
    finally:
        ia.lib.fmx_set_option(b"map_by_symbol", -1)
        ia.lib.fmx_set_option(b"sb_cache_limit", 320)
        ia.lib.fmx_set_option(b"inv_fast", 1)


@pytest.mark.plan_policy
def test_locate_walks_the_hits_by_the_first_row_of_the_ranges_and_nothing_else_changes():
    """Round 4: k_locate_walk takes the patterns of a large batch by the first row of their SA ranges (k_walk_hist, k_plan_scatter
    from the ranges, k_plan_fine; option walk_order_min, default 32,768 patterns): found, positions, statuses (the reference's
    AIOOBE where `locations` is shorter than the hits wanted) and LF-steps are those of the caller's order and of the oracle.
    Host-buffer entry (a per-call workspace), device entry (the stream's), a segment set (hits taken by earlier segments)."""
    import torch

    text = ia.synth_log(1 << 21)
    o = orc.OracleFmIndex(text, 16, True)
    fm = ia.FmIndex.read(o.write(False), device=0)
    t16 = ia.as_chars(text)
    r = random.Random(5)
    n = 40_000
    pats = [t16[a:a + r.choice([0, 1, 2, 3, 5, 8, 13])] for a in (r.randrange(len(t16) - 16) for _ in range(n))]
    for k in range(0, n, 37):  # patterns without hits
        pats[k] = np.concatenate([pats[k], np.array([7], np.uint16)])
    ch, off = ia.pack_patterns(pats)
    L = ia.lib
    try:
        for mm, cap in ((5, 5), (6, 3), (-1, 4)):
            want = o.locate_batch(ch, off, mm, loc_cap=cap, threads=8)
            got = {}
            for walk_min, fine in ((0, 1), (32768, 1), (1, 1), (1, 0)):
                assert L.fmx_set_option(b"walk_order_min", walk_min) == 0 and L.fmx_set_option(b"walk_fine", fine) == 0
                got[(walk_min, fine)] = fm.locate_batch(ch, off, mm, loc_cap=cap, want_steps=True)
            base = got[(0, 1)]
            live = np.arange(cap)[None, :] < base[1][:, None]
            assert (base[1] == want[1]).all() and (base[2] == want[2]).all() and (base[0][live] == want[0][live]).all()
            for g in got.values():
                assert (g[1] == base[1]).all() and (g[2] == base[2]).all() and (g[3] == base[3]).all()
                assert (g[0][live] == base[0][live]).all()
        # device entry points: the stream's workspace serves batches of different sizes one after the other
        dev = torch.device("cuda", 0)
        sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        d_pat = torch.from_numpy(ch.view(np.int16)).to(dev)
        d_off = torch.from_numpy(off).to(dev)
        assert L.fmx_set_option(b"walk_order_min", 1) == 0 and L.fmx_set_option(b"walk_fine", 1) == 0
        want = o.locate_batch(ch, off, 4, threads=8)
        for m in (n, 1000, 17, n, 1):
            d_locs = torch.full((m * 4,), -1, dtype=torch.int32, device=dev)
            d_found = torch.zeros(m, dtype=torch.int32, device=dev)
            d_st = torch.zeros(m, dtype=torch.int32, device=dev)
            d_rng = torch.zeros(2 * m, dtype=torch.int32, device=dev)
            assert L.fmx_locate_batch_dev(fm.handle, d_pat.data_ptr(), d_off.data_ptr(), m, 4, d_locs.data_ptr(), 4, d_found.data_ptr(),
                                          None, d_st.data_ptr(), d_rng.data_ptr(), sp) == 0
            torch.cuda.synchronize()
            f = d_found.cpu().numpy()
            live = np.arange(4)[None, :] < f[:, None]
            assert (f == want[1][:m]).all() and (d_st.cpu().numpy() == want[2][:m]).all()
            assert (d_locs.cpu().numpy().reshape(m, 4)[live] == want[0][:m][live]).all()
        # a segment set: each segment walks its own ranges in its own order, limits shrink by what earlier segments gave
        seg = ia.SegmentedFmIndex(text, 16, True, device=0, segment_chars=600_000)
        assert len(seg.segments) >= 3
        res = []
        for walk_min in (0, 1):
            assert L.fmx_set_option(b"walk_order_min", walk_min) == 0
            res.append(seg.locate_batch(ch, off, 6))
        assert (res[0][1] == res[1][1]).all() and (res[0][2] == res[1][2]).all() and (res[0][0] == res[1][0]).all()
        assert int(res[0][1].sum()) > n
    finally:
        L.fmx_set_option(b"walk_order_min", 32768)
        L.fmx_set_option(b"walk_fine", 1)
    fm.close()


def test_workgroups_regrouped_by_pattern_length_give_the_same_answers():
    """Round 4: a k_count workgroup in which some wave holds patterns of different lengths hands its records out again by length
    (option regroup_by_length; planned batches of ONE length skip the vote: CountPlan.mixed).  Lengths 0 (the reference's AIOOBE),
    1..70 (beyond the 63 of the sort's last bin, beyond the code word), absent characters, a batch that does not fill its last
    workgroup; planned and in the caller's order; counts, statuses, LF-steps and located hits against the oracle."""
    text = ia.synth_log(1 << 20)
    o = orc.OracleFmIndex(text, 16, True)
    fm = ia.FmIndex.read(o.write(False), device=0)
    t16 = ia.as_chars(text)
    r = random.Random(9)
    L = ia.lib
    try:
        for n, lens in ((20_001, list(range(0, 71))), (20_001, [8]), (777, [3, 9, 30]), (16_500, [8, 8, 8, 8, 9])):
            pats = [t16[a:a + r.choice(lens)] for a in (r.randrange(len(t16) - 80) for _ in range(n))]
            for k in range(0, n, 53):
                if len(pats[k]):
                    pats[k] = pats[k].copy()
                    pats[k][r.randrange(len(pats[k]))] = 7
            ch, off = ia.pack_patterns(pats)
            oc, ost = o.count_batch(ch, off, threads=8)
            want = o.locate_batch(ch, off, 3, threads=8)
            for regroup in (1, 0):
                for sort_min in (16384, 1 << 30):  # planned (the suite forces plans from sort_min on) / the caller's order
                    assert L.fmx_set_option(b"regroup_by_length", regroup) == 0 and L.fmx_set_option(b"sort_min", sort_min) == 0
                    cnt, st, lf = fm.count_batch(ch, off, want_steps=True)
                    assert (cnt == oc).all() and (st == ost).all(), (n, lens, regroup, sort_min)
                    if regroup == 1 and sort_min == 16384:
                        lf_ref = lf
                    assert (lf == lf_ref).all()
                    locs, found, st2 = fm.locate_batch(ch, off, 3)
                    live = np.arange(3)[None, :] < want[1][:, None]
                    assert (found == want[1]).all() and (st2 == want[2]).all() and (locs[live] == want[0][live]).all()
    finally:
        L.fmx_set_option(b"regroup_by_length", 1)
        L.fmx_set_option(b"sort_min", 16384)
    fm.close()


def test_the_plan_stage_as_one_launch_gives_the_same_answers_also_when_its_barrier_gives_up():
    """Round 5: k_plan_fused (records in registers, tickets from the histogram's atomic adds, a bounded grid barrier) against
    k_plan_codes + k_plan_scatter (option plan_fused = 0), on 8-bit and 16-bit code words, one length and mixed lengths, a batch
    that does not fill its last tile — and with the barrier's patience set to nothing (plan_spin_limit = 0): workgroups abort
    the order and write their records at their own indices, a valid plan in the caller's order.  Counts, statuses, LF-steps and
    located hits against the oracle every time; the workspace's head must come back zeroed (the next plan works)."""
    L = ia.lib
    r = random.Random(5)
    texts = [ia.synth_log(1 << 20), ia.synth_log_multichar(1 << 20, 600)]
    try:
        for text in texts:
            o = orc.OracleFmIndex(text, 16, True)
            fm = ia.FmIndex.read(o.write(False), device=0)
            t16 = ia.as_chars(text)
            for n, lens in ((40_000, [8]), (70_001, [1, 5, 8, 13, 31]), (20_000, list(range(0, 40)))):
                pats = [t16[a:a + r.choice(lens)] for a in (r.randrange(len(t16) - 80) for _ in range(n))]
                for k in range(0, n, 97):
                    if len(pats[k]):
                        pats[k] = pats[k].copy()
                        pats[k][r.randrange(len(pats[k]))] = 7
                ch, off = ia.pack_patterns(pats)
                oc, ost = o.count_batch(ch, off, threads=8)
                want = o.locate_batch(ch, off, 3, threads=8)
                lf_ref = None
                for fused, spin in ((1, 4096), (0, 4096), (1, 0), (1, 4096)):
                    assert L.fmx_set_option(b"plan_fused", fused) == 0 and L.fmx_set_option(b"plan_spin_limit", spin) == 0
                    assert L.fmx_count_batch_is_planned(fm.handle, n) == 1
                    for _ in range(2):  # (twice: the second plan finds the head as the first left it)
                        cnt, st, lf = fm.count_batch(ch, off, want_steps=True)
                        assert (cnt == oc).all() and (st == ost).all(), (n, lens, fused, spin)
                        lf_ref = lf if lf_ref is None else lf_ref
                        assert (lf == lf_ref).all()
                    locs, found, st2 = fm.locate_batch(ch, off, 3)
                    live = np.arange(3)[None, :] < want[1][:, None]
                    assert (found == want[1]).all() and (st2 == want[2]).all() and (locs[live] == want[0][live]).all()
            fm.close()
    finally:
        L.fmx_set_option(b"plan_fused", 1)
        L.fmx_set_option(b"plan_spin_limit", 4096)


def test_api_edge_cases():
    """empty batches, zero-capacity buffers and bad arguments through the C ABI: no kernel launch with bad shapes,
    library-level error codes instead"""
    import ctypes as C

    fm = ia.FmIndex(HD[:5000], 8, True, device=0)
    e = np.zeros(0, np.uint16)
    cnt, st = fm.count_batch(e, np.zeros(1, np.int32))
    assert len(cnt) == 0
    ch, off = ia.pack_patterns(["INFO", "a"])
    locs, found, st = fm.locate_batch(ch, off, 5, loc_cap=0)  # no room at all: the JVM would raise AIOOBE on the first hit
    assert locs.shape == (2, 0) and (found == 0).all() and (st == 9).all()
    dst, ol, st = fm.extract_batch([3], [10], 0)
    assert st[0] == 4  # "Supplied destination is not large enough"
    dst, ol, st, aux = fm.extract_boundary_batch([3], "\n", 0, 0)
    assert st[0] == 6  # "Supplied destination for extraction has size zero"
    L = ia.lib
    z = np.zeros(4, np.int32)
    assert L.fmx_locate_extract_batch(fm.handle, ch.ctypes.data, off.ctypes.data, 2, 0, 8, z.ctypes.data, z.ctypes.data,
                                      None, z.ctypes.data, None, None, None) == ia._lib.E_ARG  # max_matches must be >= 1
    assert L.fmx_count_segments(None, 0, ch.ctypes.data, off.ctypes.data, 2, z.ctypes.data, None, None) == ia._lib.E_ARG
    assert L.fmx_count_batch(None, ch.ctypes.data, off.ctypes.data, 2, z.ctypes.data, None, None) == ia._lib.E_ARG
    host_only = ia.FmIndex("abc", 2, True, device=None)
    assert L.fmx_count_batch(host_only.handle, ch.ctypes.data, off.ctypes.data, 2, z.ctypes.data, None, None) == ia._lib.E_NO_DEVICE
    bad = np.frombuffer(b"\x07garbage-not-an-index", dtype=np.uint8)
    h = C.c_void_p()
    assert L.fmx_load(bad.ctypes.data, len(bad), C.byref(h)) in (ia._lib.E_VERSION, ia._lib.E_FORMAT)


@pytest.mark.gpu
def test_host_locate_keeps_the_callers_slots_beyond_the_hits():
    """`locations` of FmIndex.locate is the caller's array (FM:504): slots beyond a pattern's hits keep what they held — also
    for capped and failing patterns"""
    rnd = random.Random(4242)
    text = HD[:120_000]
    fm = ia.FmIndex(text, 16, True, device=0)
    o = orc.OracleFmIndex(text, 16, True)
    t16 = ia.as_chars(text)
    L = len(t16)
    pats = [t16[s:s + rnd.randrange(1, 14)] for s in (rnd.randrange(L - 16) for _ in range(400))]
    pats += [ia.as_chars("zzzzqq"), ia.as_chars(" "), ia.as_chars("INFO"), t16[:1]]
    ch, off = ia.pack_patterns(pats)
    off = np.concatenate([off, [off[-1]]]).astype(np.int32)  # plus one EMPTY pattern -> AIOOBE
    for mm, cap in ((8, 8), (-1, 5), (3, 8), (8, 3)):
        mine = np.arange(len(pats) + 1, dtype=np.int32)[:, None] * 1000 + np.arange(cap, dtype=np.int32)[None, :] - 5_000_000
        mine = np.ascontiguousarray(mine)
        locs, found, st = fm.locate_batch(ch, off, mm, cap, locs=mine)
        assert locs is mine
        for i, p in enumerate(pats):
            keep = i * 1000 + np.arange(cap) - 5_000_000
            try:
                n, l = o.locate(p, max_matches=mm, cap=cap)
                assert st[i] == 0 and found[i] == n and (locs[i, :n] == l).all() and (locs[i, n:] == keep[n:]).all(), (i, mm, cap)
            except IndexError:
                assert st[i] == 9 and (locs[i, found[i]:] == keep[found[i]:]).all()
        assert st[-1] == 9 and found[-1] == 0


def _window_bytes(f):
    n = C.c_int64(-1)
    assert ia.lib.fmx_window_cells_info(f.handle, C.byref(n)) == 0
    return n.value


@pytest.mark.parametrize("mode,entry_bytes", [(0, 0), (1, 0), (1, 4), (1, 6), (3, 0)])
def test_window_directory_changes_nothing_but_the_time(mode, entry_bytes):
    """option window_cells: every query kind vs the oracle with the directory grown (1) and without one (0) — on the fixture,
    on texts with sentinels, run blocks of wide symbols (Q1: never a directory entry) and a 900-symbol alphabet, on a compact
    image — and fmx_window_cells_info says which of the two an index got.  entry_bytes (option window_entry_bytes): the directory's
    entries in the form picked by the alphabet (0: four bytes — the row, its symbol found by a search over cumulativeCounts — up
    to 2,048 symbols, six beyond: the fixture has 2,061), all in four, all in six.  mode 3: the FLAT form of the directory (a word
    per position, 4 bytes per text byte: every step of a walk one sector)"""
    rng = np.random.default_rng(9)
    parts = []
    for i in range(6):
        parts.append("".join(chr(0x4E00 + int(x) * 7) for x in rng.integers(0, 900, 1500)))
        parts.append(chr(0x30A1 + i) * 70_000)
        parts.append("log line %d\n" % i * 50)
    wide = "".join(parts)
    rnd = random.Random(40 + mode)
    mod = list(HD[:40_000])
    for _ in range(300):
        mod[rnd.randrange(len(mod) - 2)] = "\0"
    try:
        assert ia.lib.fmx_set_option(b"window_cells", mode) == 0
        assert ia.lib.fmx_set_option(b"window_entry_bytes", entry_bytes) == 0
        f = ia.FmIndex(HD[:20_000], 8, True, device=0)
        assert (_window_bytes(f) > 0) == (mode != 0)
        # 64 bytes per 112 positions + 6 (4) per position none of its window's three classes holds; the flat form: 4 per position
        assert mode != 1 or 64 * (20_001 // 112) <= _window_bytes(f) <= 64 * (20_001 // 112 + 2) + 6 * 20_001
        assert mode != 3 or 4 * 20_001 <= _window_bytes(f) <= 4 * 20_001 + 8 * 200
        if mode == 1 and entry_bytes:
            other = {4: 6, 6: 4}[entry_bytes]
            assert ia.lib.fmx_set_option(b"window_entry_bytes", other) == 0
            g = ia.FmIndex(HD[:20_000], 8, True, device=0)
            assert ia.lib.fmx_set_option(b"window_entry_bytes", entry_bytes) == 0
            cells = 64 * (20_001 // 112 + 1)
            small, big = sorted((_window_bytes(f), _window_bytes(g)))
            assert 1.45 < (big - cells) / (small - cells) < 1.51  # the entries: six bytes against four (+ a few eight-byte slots)
            g.close()
        for sr in (1, 4, 32, 64):
            check_all(make_gpu, HD, sr, rnd, n_q=100)
        check_all(make_gpu, "".join(mod), 8, rnd, n_q=60)
        check_all(make_gpu, wide, 16, rnd, n_q=60)
        check_all(make_gpu, "ab" * 56, 4, rnd, n_q=20)  # wt_size = 113: the last position has a cell of its own
        check_all(make_gpu, "a", 1, rnd, n_q=5)
        assert ia.lib.fmx_set_option(b"image_compact", 1) == 0
        check_all(make_gpu, HD[:60_000], 32, rnd, n_q=60)
    finally:
        ia.lib.fmx_set_option(b"image_compact", 0)
        ia.lib.fmx_set_option(b"window_cells", 2)
        ia.lib.fmx_set_option(b"window_entry_bytes", 0)


@pytest.mark.parametrize("small_max", [0, 50, 2048])
def test_small_host_calls_through_one_mapped_block_change_nothing(small_max):
    """option host_small_max: host-array calls of at most that many patterns / queries go through ONE pinned block mapped into the
    device's address space (no copy calls: a scalar count() 104 -> 26 us) — every query kind against the oracle with the path off
    (0: the copying paths, as before round 6), on for part of check_all's batches (50) and on for all of them (the default);
    in / out arrays keep what a query does not write in every form"""
    rnd = random.Random(900 + small_max)
    try:
        assert ia.lib.fmx_set_option(b"host_small_max", small_max) == 0
        check_all(make_gpu, HD[:90_000], 16, rnd, n_q=100)
        fm = ia.FmIndex(HD[:90_000], 16, True, device=0)
        o = orc.OracleFmIndex(HD[:90_000], 16, True)
        t16 = ia.as_chars(HD[:90_000])
        pats = [t16[s:s + 9] for s in range(0, 40_000, 997)] + [ia.as_chars("zzzz")]
        ch, off = ia.pack_patterns(pats)
        n = len(pats)
        pre = np.full((n, 7), -5, np.int32)
        locs, found, st = fm.locate_batch(ch, off, 4, 7, locs=pre.copy())
        ol, of, os_ = o.locate_batch(ch, off, 4, 7, fill=-5)
        assert (locs == ol).all() and (found == of).all() and (st == os_).all()  # whole rows: the slots beyond the hits too
        a = np.arange(n, dtype=np.int32) * 501
        dst, ol2, st2 = fm.extract_batch(a, a + 20, 30, 3, dst=np.full((n, 30), 7, np.uint16))
        od, on, oe = o.extract_batch(a, a + 20, 30, 3, fill=7)
        assert (dst == od).all() and (ol2 == on).all() and (st2 == oe).all()
        for mode in (0, 1, 2):
            got = fm.extract_boundary_batch(a, "\n", mode, 48, 2, dst=np.full((n, 48), 7, np.uint16))
            exp = o.extract_until_boundary_batch(mode, a, "\n", 48, 2, fill=7)
            assert (got[0] == exp[0]).all() and (got[2] == exp[2]).all() and (got[1][exp[2] == 0] == exp[1][exp[2] == 0]).all()
            assert (got[3][exp[2] == 8] == exp[3][exp[2] == 8]).all()
        # the fused pipelines (locate -> extract, locate -> extractUntilBoundary): every array against the copying path's
        # (which test_locate_extract_pipeline_vs_oracle holds to the oracle)
        here = [fm.locate_extract_batch(ch, off, 3, 24, fill=9), fm.locate_lines_batch(ch, off, 3, "\n", 60, 0, fill=9)]
        assert ia.lib.fmx_set_option(b"host_small_max", 0) == 0
        there = [fm.locate_extract_batch(ch, off, 3, 24, fill=9), fm.locate_lines_batch(ch, off, 3, "\n", 60, 0, fill=9)]
        for x, y in zip(here, there):
            for key in x:
                assert (x[key] == y[key]).all(), key
        assert int(here[0]["found"].sum()) > 20
        fm.close()
    finally:
        ia.lib.fmx_set_option(b"host_small_max", 2048)
