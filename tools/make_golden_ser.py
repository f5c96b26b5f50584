#!/usr/bin/env python3
"""Mints tests/golden/ser/*: the serialized form (FmIndex.write, FM:948-975; raw DataOutput stream and ObjectOutputStream-framed,
SER:67-79) of the reference's known-answer texts, as THIS repository writes it today.

MINTED HERE, NOT BY A JVM: the reference holds no golden serialized file and this image has no JDK ("parity
unpinned" for the bytes).  What the files pin is DRIFT: the product's writer (fmx_save) and the oracle's (orc_fm_write) must
keep producing exactly these bytes — any change to the HashMap order replay, the Huffman tie-breaks, the block-size estimate
or the framing shows up as a diff against a committed file (tests/test_serial_golden.py).  A maintainer with a JDK runs
bindings/build.sh, whose parity test writes the same texts with index4j itself and compares.

    python tools/make_golden_ser.py          # rewrites tests/golden/ser/ (small texts: the bytes; the fixture: digests)"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

TEXTS = {
    "kat_fm": "This is a long string\0",                       # FmIndexTest.java:43-102
    "kat_wt": "aloha what a string this is string is eh",     # WaveletFixedBlockBoostingTest.java:50-68
}
SAMPLE_RATES = (1, 32)


def main():
    import index4j_amd as ia
    import orc
    from common import hdfs_text

    out = os.path.join(ROOT, "tests", "golden", "ser")
    os.makedirs(out, exist_ok=True)
    digests = {}
    texts = dict(TEXTS, hdfs_fixture=hdfs_text())  # FmIndexTest.java:195-200, 564-578 (HDFS_2k_multichar.log)
    for name, text in texts.items():
        for s in SAMPLE_RATES:
            fm = ia.FmIndexBuilder().setSampleRate(s).setEnableExtraction(True).build(text, device=None)
            o = orc.OracleFmIndex(text, s, True)
            for framed in (False, True):
                b = fm.write(framed)
                if b != o.write(framed):
                    raise SystemExit("product and oracle disagree on %s s=%d framed=%s: nothing minted" % (name, s, framed))
                key = "%s_s%d_%s" % (name, s, "framed" if framed else "raw")
                digests[key] = {"bytes": len(b), "sha256": hashlib.sha256(b).hexdigest(),
                                "key_order_modelled": fm.serialized_key_order_is_modelled()}
                if name != "hdfs_fixture":  # (the fixture's streams are 0.2-1 MB each: digests only)
                    with open(os.path.join(out, key + ".ser"), "wb") as f:
                        f.write(b)
    with open(os.path.join(out, "digests.json"), "w") as f:
        json.dump({"what": "FmIndex.write streams as this repository writes them (minted here, NOT by a JVM): drift detection, "
                           "tools/make_golden_ser.py", "streams": digests}, f, indent=1, sort_keys=True)
    print("minted %d streams into %s" % (len(digests), out))


if __name__ == "__main__":
    main()
