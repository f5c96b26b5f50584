#!/usr/bin/env python3
"""Check (in the build container only) that the combinatorial generator used by oracle/ and the
product reproduces the reference's three literal RRR tables, and emit their SHA-256 digests as a
golden fixture.  Reads /root/reference (RrrVector.java:488-16900); never run on the GPU box.

Usage: python tools/check_rrr_tables.py [--write tests/golden/rrr_tables.json]
"""
import hashlib, itertools, json, re, sys
from math import comb

SRC = "/root/reference/indices/src/main/java/com/dynatrace/bitsequence/RrrVector.java"


def parse_tables():
    lines = open(SRC, encoding="utf-8").read().split("\n")
    def longs(lo, hi):
        out = []
        for ln in lines[lo:hi]:
            out += [int(x) for x in re.findall(r"(-?\d+)L", ln)]
        return out
    offs = longs(489, 8682)
    inv = longs(8706, 16899)
    card = [int(x) for x in re.findall(r"-?\d+", " ".join(lines[8693:8696]))]
    return offs, card, inv


def unpack16(longs):
    out = []
    for v in longs:
        v &= (1 << 64) - 1
        out += [(v >> (16 * j)) & 0xFFFF for j in range(4)]
    return out


def generate():
    """offset_of[value], base[k], value_of[base[k]+offset] from first principles."""
    base = [0] * 16
    for k in range(1, 16):
        base[k] = base[k - 1] + comb(15, k - 1)
    offset_of = [0] * 32768
    value_of = [0] * 32768
    for k in range(16):
        for idx, pos in enumerate(itertools.combinations(range(15), k)):
            v = 0
            for p in pos:
                v |= 1 << p
            offset_of[v] = idx
            value_of[base[k] + idx] = v
    return offset_of, base, value_of


def digest(vals):
    h = hashlib.sha256()
    for v in vals:
        h.update(int(v & 0xFFFF).to_bytes(2, "little"))
    return h.hexdigest()


def main():
    offs, card, inv = parse_tables()
    ref_off, ref_inv = unpack16(offs), unpack16(inv)
    assert len(ref_off) == 32768 and len(ref_inv) == 32768 and len(card) == 16, (len(ref_off), len(ref_inv), len(card))
    gen_off, gen_base, gen_inv = generate()
    assert [c & 0xFFFF for c in card] == [b & 0xFFFF for b in gen_base], (card, gen_base)
    assert ref_off == gen_off, "PRECOMPUTED_OFFSETS mismatch"
    assert ref_inv == gen_inv, "INVERSE_VALUES mismatch"
    out = {
        "source": "RrrVector.java:488-16900 (parsed in the build container by tools/check_rrr_tables.py)",
        "offset_of_value_sha256": digest(ref_off),
        "cardinality_offsets": [c & 0xFFFF for c in card],
        "value_of_offset_sha256": digest(ref_inv),
        "samples_offset_of_value": {str(v): ref_off[v] for v in (0, 1, 2, 3, 5, 1234, 16384, 21845, 32767)},
        "samples_value_of_offset": {str(i): ref_inv[i] for i in (0, 1, 15, 16, 17, 120, 121, 5000, 32767)},
    }
    print(json.dumps(out, indent=1))
    if len(sys.argv) > 2 and sys.argv[1] == "--write":
        json.dump(out, open(sys.argv[2], "w"), indent=1)
        print("wrote", sys.argv[2])


if __name__ == "__main__":
    main()
