"""tests/golden/vectors.json — brute-force expectations minted by tools/make_golden_vectors.py (plain Python, the
definitional oracles of the reference's tests; neither the product nor the C oracle involved) — against
(1) the C oracle, (2) the device source on the host simulation, (3) the HIP kernels through the C ABI."""
import hashlib
import json
import os

import numpy as np
import pytest

import hostsim
import index4j_amd as ia
import orc
from common import GOLDEN

V = json.load(open(os.path.join(GOLDEN, "vectors.json")))


def text_of(name):
    t = V["texts"][name]
    if isinstance(t, dict):
        raw = open(os.path.join(GOLDEN, t["file"]), "rb").read()
        assert hashlib.sha256(raw).hexdigest() == t["sha256"]
        return raw.decode("ascii")
    return t


def check_engine(case, count_batch, locate_batch, extract_batch, boundary_batch):
    """the four batch calls in index4j_amd.FmIndex's shapes"""
    text = text_of(case["text"])
    pats = [p for p, _ in case["count"]]
    ch, off = ia.pack_patterns(pats)
    cnt, st = count_batch(ch, off)[:2]
    assert (st == 0).all() and cnt.tolist() == [c for _, c in case["count"]]
    pats = [p for p, _ in case["locate_sorted"]]
    if pats:
        ch, off = ia.pack_patterns(pats)
        locs, found, st = locate_batch(ch, off, -1, 400)[:3]
        for i, (_, exp) in enumerate(case["locate_sorted"]):
            assert st[i] == 0 and found[i] == len(exp) and sorted(locs[i, :found[i]].tolist()) == exp
    pats = [p for p, _ in case["locate_sa_order_16"]]
    if pats:
        ch, off = ia.pack_patterns(pats)
        locs, found, st = locate_batch(ch, off, 16, 16)[:3]
        for i, (_, exp) in enumerate(case["locate_sa_order_16"]):
            assert st[i] == 0 and locs[i, :found[i]].tolist() == exp  # SA order, FM:527-547
    a = np.array([e[0] for e in case["extract"]], np.int32)
    b = np.array([e[1] for e in case["extract"]], np.int32)
    dst, ol, st = extract_batch(a, b, 80)[:3]
    for i, (_, _, exp) in enumerate(case["extract"]):
        assert st[i] == 0 and ol[i] == len(exp) and ia.chars_to_str(dst[i, :ol[i]]) == exp
    for mode in (0, 1, 2):
        rows = [r for r in case["boundary"] if r[0] == mode]
        fr = np.array([r[1] for r in rows], np.int32)
        dst, ol, st, aux = boundary_batch(fr, rows[0][2], mode, 2048)[:4]
        for i, (_, _, _, exp) in enumerate(rows):
            assert st[i] == 0 and ia.chars_to_str(dst[i, :ol[i]]) == exp, (case["text"], mode, rows[i][1])
    for e in case["errors"]:
        if e[0] == "extract":
            _, ol, st = extract_batch(np.array([e[1]], np.int32), np.array([e[2]], np.int32), e[3])[:3]
        else:
            _, ol, st, _ = boundary_batch(np.array([e[2]], np.int32), e[3], e[1], e[4])[:4]
        assert st[0] == e[-1], e


@pytest.mark.parametrize("case", V["cases"], ids=[c["text"] for c in V["cases"]])
@pytest.mark.parametrize("sr", [1, 4, 32])
def test_oracle_against_golden_vectors(case, sr):
    text = text_of(case["text"])
    o = orc.OracleFmIndex(text, sr, True)
    for p, c in case["count"]:
        assert o.count(p) == c
    for p, exp in case["locate_sorted"]:
        n, l = o.locate(p, max_matches=-1, cap=400)
        assert sorted(l.tolist()) == exp
    for p, exp in case["locate_sa_order_16"]:
        n, l = o.locate(p, max_matches=16, cap=16)
        assert l.tolist() == exp
    for a, b, exp in case["extract"]:
        n, d = o.extract(a, b, dest_len=80)
        assert ia.chars_to_str(d[:n]) == exp
    for mode, frm, bch, exp in case["boundary"]:
        n, d = o.extract_until_boundary(mode, frm, 2048, 0, bch)
        assert ia.chars_to_str(d[:n]) == exp, (mode, frm)


@pytest.mark.parametrize("case", V["cases"], ids=[c["text"] for c in V["cases"]])
def test_device_source_on_host_against_golden_vectors(case):
    h = hostsim.HostSim(ia.FmIndex(text_of(case["text"]), 8, True, device=None))
    check_engine(case, lambda ch, off: h.count_batch(ch, off), lambda ch, off, mm, cap: h.locate_batch(ch, off, mm, cap),
                 lambda a, b, n: h.extract_batch(a, b, n), lambda fr, bch, mode, n: h.extract_boundary_batch(fr, bch, mode, n))


@pytest.mark.gpu
@pytest.mark.parametrize("case", V["cases"], ids=[c["text"] for c in V["cases"]])
@pytest.mark.parametrize("sr", [2, 32])
def test_kernels_against_golden_vectors(case, sr):
    fm = ia.FmIndex(text_of(case["text"]), sr, True, device=0)
    check_engine(case, lambda ch, off: fm.count_batch(ch, off), lambda ch, off, mm, cap: fm.locate_batch(ch, off, mm, cap),
                 lambda a, b, n: fm.extract_batch(a, b, n), lambda fr, bch, mode, n: fm.extract_boundary_batch(fr, bch, mode, n))
