#!/usr/bin/env python3
"""Where the bytes of a flat index image (fmx_blob.hpp) go: per-section totals, per text character.

    python tools/image_sections.py [--text-log2 24] [--symbols 1100] [--sample-rate 32]
"""
import argparse
import os
import struct
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def sections(blob):
    """{section: bytes} of an FM-index image (BlobHeader / SbDesc / RrrDesc of fmx_blob.hpp)"""
    b = memoryview(blob)
    (magic, version, total, sample_rate, enable_extract, length, n_keys, bw_suf, bw_pos, n_c, n_look, sigma, n_sb, n_suf,
     n_posn, wt_size, off_c, off_look, off_c2c, off_suf, off_posw, off_sbc, off_sbd, off_inv) = struct.unpack_from(
        "<IIQ12iq8I", b, 0)
    sampled = struct.unpack_from("<II6i", b, 16 + 48 + 8 + 32)
    out = {"header + C + lookUp": 256 + (n_c + n_look) * 4, "char2code LUT": 131072,
           "suffixes (packed)": (n_suf * bw_suf + 63) // 64 * 8, "positions (packed)": (n_posn * bw_pos + 63) // 64 * 8,
           "sampled-row bitmap cells": sampled[4] * 16, "SbcEntry table": (n_sb + 1) * sigma * 8, "SbDesc": n_sb * 64,
           "mapping entries": 0, "path records": 0, "block headers": 0, "var header bytes": 0, "wavelet cells": 0,
           "inverseSelect section": 0}
    for s in range(n_sb):
        o = (off_sbd << 3) + 64 * s
        sg, bsl, off_map, off_bh, off_var, n_blocks, var_len, mapping_len, path_len = struct.unpack_from("<hhIII4i", b, o)
        rrr = struct.unpack_from("<II6i", b, o + 32)
        out["mapping entries"] += mapping_len * 16
        out["path records"] += path_len * 8
        out["block headers"] += n_blocks * 16
        out["var header bytes"] += var_len
        out["wavelet cells"] += rrr[4] * 16
        out["inverseSelect section"] += rrr[7] * 16
    out["(alignment, guards)"] = total - sum(out.values())
    return out, dict(sigma=sigma, n_sb=n_sb, length=length, total=total)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--compact", action="store_true", help="the compact image (option image_compact): the two 'cells' rows are "
                                                           "then 16-byte RRR records; their offsets streams show up under alignment")
    ap.add_argument("--text-log2", type=int, default=24)
    ap.add_argument("--symbols", type=int, default=1100)
    ap.add_argument("--sample-rate", type=int, default=32)
    ap.add_argument("--build-device", type=int, default=-1)
    args = ap.parse_args()
    import index4j_amd as ia

    if args.compact:
        ia.lib.fmx_set_option(b"image_compact", 1)
    n = 1 << args.text_log2
    text = ia.synth_log_multichar(n, args.symbols) if args.symbols > 70 else ia.synth_log(n)
    fm = ia.FmIndex(text, args.sample_rate, True, device=None, build_device=None if args.build_device < 0 else args.build_device)
    blob = fm.blob()
    sec, info = sections(blob)
    print("text 2^%d chars, %d distinct, sigma %d, %d superblocks, sampleRate %d" % (
        args.text_log2, len(np.unique(text)), info["sigma"], info["n_sb"], args.sample_rate))
    print("serialized (index4j layout): %.3f B/char" % (len(fm.write(False)) / n))
    for k, v in sorted(sec.items(), key=lambda kv: -kv[1]):
        print("  %-28s %12d  %.4f B/char" % (k, v, v / n))
    print("  %-28s %12d  %.4f B/char" % ("image", info["total"], info["total"] / n))


if __name__ == "__main__":
    main()
