#!/usr/bin/env python3
"""Mints tests/golden/vectors.json: inputs and expected outputs of every query kind, computed by BRUTE FORCE in
plain Python — the definitional oracles of the reference's own tests (Util.java:111-258: overlapping
occurrences, substring, boundary scanners) plus a naive suffix sort under the first-appearance alphabet for the
order of a truncated locate (FM:396-435, 527-547).  Neither the product nor the C oracle is used here; both are
checked AGAINST this file (tests/test_golden_vectors.py).  The 64 KiB log text was produced once with the
repository's synthetic generator (fmx_synth_log, seed 42) and is committed as data (tests/golden/synth_64k.txt).
usage: python tools/make_golden_vectors.py"""
import hashlib
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from common import occurrences, until_boundary, until_boundary_left, until_boundary_right  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def sa_order_hits(text, pat, limit):
    """first `limit` occurrences in suffix-array order; codes by first appearance, the terminator smallest"""
    code = {}
    for ch in text:
        code.setdefault(ch, len(code) + 1)
    occ = occurrences(text, pat)
    occ.sort(key=lambda i: [code[c] for c in text[i:i + 4000]] + [0])
    return occ[:limit]


def cases_for(name, text, rnd, n_patterns):
    L = len(text)
    bch = "\n" if "\n" in text else " "
    # seeds at or behind the last boundary lie in an unterminated last field, where the reference itself returns a
    # length one short (FM:745-752, docs/DESIGN_HISTORY.md Q12): kept out of the definitional vectors
    last_safe = max(1, L if text.endswith(bch) else text.rfind(bch) + 1)
    pats = set()
    while len(pats) < n_patterns:
        s = rnd.randrange(L - 1)
        pats.add(text[s:s + rnd.randrange(1, 13)])
    pats = sorted(pats) + ["zzzzqq", "睷x"]
    count, loc_sorted, loc_sa = [], [], []
    for p in pats:
        occ = occurrences(text, p)
        count.append([p, len(occ)])
        if len(occ) <= 400:
            loc_sorted.append([p, occ])
        if 0 < len(occ) <= 60:
            loc_sa.append([p, sa_order_hits(text, p, 16)])
    extract = []
    for _ in range(25):
        a = rnd.randrange(L - 1)
        b = min(L - 1, a + rnd.randrange(0, 70))
        extract.append([a, b, text[a:b]])
    boundary = []
    for _ in range(30):
        seed = rnd.randrange(min(L, last_safe))
        boundary.append([0, seed, bch, until_boundary(text, seed, bch)])
        boundary.append([1, seed, bch, until_boundary_left(text, seed, bch)])
        boundary.append([2, seed, bch, until_boundary_right(text, seed, bch)])
    # statuses of include/fmx.h (FM:566-576, 591-593, 619-625, 659-661): [kind, args..., status]
    errors = [["extract", -1, 5, 50, 2], ["extract", 3, L + 5, 50, 3], ["extract", 0, min(40, L - 1), 10, 4 if L > 11 else 0],
              ["boundary", 0, L + 3, bch, 100, 5], ["boundary", 0, 1, bch, 0, 6], ["boundary", 0, 1, "睷", 100, 7]]
    return {"text": name, "count": count, "locate_sorted": loc_sorted, "locate_sa_order_16": loc_sa, "extract": extract,
            "boundary": boundary, "errors": errors}


def main():
    rnd = random.Random(20261003)
    synth_path = os.path.join(GOLDEN, "synth_64k.txt")
    if not os.path.exists(synth_path):
        sys.path.insert(0, ROOT)
        import index4j_amd as ia  # only to mint the text once; the vectors below never touch the library

        open(synth_path, "wb").write(bytes(ia.synth_log(1 << 16, seed=42).astype("uint8")))
    synth = open(synth_path, "rb").read().decode("ascii")
    texts = {
        "kat_fm": "This is a long string",
        "kat_lines": "What a string!\nNow this is long, indeed\nBut others could be longer.",
        "synth_64k": synth,
    }
    out = {"about": "brute-force expectations, see tools/make_golden_vectors.py",
           "texts": {"kat_fm": texts["kat_fm"], "kat_lines": texts["kat_lines"],
                     "synth_64k": {"file": "synth_64k.txt", "sha256": hashlib.sha256(synth.encode()).hexdigest()}},
           "cases": []}
    for name, n_pat in (("kat_fm", 25), ("kat_lines", 40), ("synth_64k", 90)):
        out["cases"].append(cases_for(name, texts[name], rnd, n_pat))
    with open(os.path.join(GOLDEN, "vectors.json"), "w") as f:
        json.dump(out, f, ensure_ascii=True, separators=(",", ":"))
    print("wrote", os.path.join(GOLDEN, "vectors.json"), os.path.getsize(os.path.join(GOLDEN, "vectors.json")), "bytes")


if __name__ == "__main__":
    main()
