#!/usr/bin/env python3
"""Which 128-byte lines of the index image one headline k_count launch touches, per XCD (diagnostic build of the
library: make -C index4j_amd/csrc EXTRA_DEFS="-DFMX_DIAG_LINES", saved as index4j_amd/libfmx_diag_lines.so; run with
FMX_LIBRARY pointing at it).  Fills the fabric must serve >= the sum over XCDs of the lines each one touches (every XCD
has its own L2); the union says what ONE shared cache would have had to fetch.  GPU box only."""
import ctypes as C
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def section_of_lines(blob):
    """line -> section id (0 other, 1 mapping entries, 2 path records, 3 wavelet cells, 4 SbcEntry table, 5 SbDesc)"""
    b = memoryview(blob)
    hdr = struct.unpack_from("<IIQ12iq8I", b, 0)
    total, sigma, n_sb = hdr[2], hdr[11], hdr[12]
    off_sbc, off_sbd = hdr[-3], hdr[-2]
    sec = np.zeros((total + 127) // 128, dtype=np.uint8)

    def mark(lo, n, v):
        if n > 0:
            sec[lo // 128:(lo + n + 127) // 128] = v

    mark(off_sbc << 3, (n_sb + 1) * sigma * 8, 4)
    mark(off_sbd << 3, n_sb * 64, 5)
    for s in range(n_sb):
        o = (off_sbd << 3) + 64 * s
        sg, bsl, off_map, off_bh, off_var, n_blocks, var_len, mapping_len, path_len = struct.unpack_from("<hhIII4i", b, o)
        off_rec, off_bits, length, ones, n_rec = struct.unpack_from("<II3i", b, o + 32)
        mark(off_map << 3, mapping_len * 16, 1)
        mark((off_map << 3) + mapping_len * 16, path_len * 8, 2)
        mark(off_rec << 3, n_rec * 16, 3)
    return sec


def main():
    import torch

    import bench
    import index4j_amd as ia

    text, fm, path = bench.build_or_load_index(ia, 28, 32, "/tmp/fmx_cache")
    fm.to_device(0)
    sec = section_of_lines(fm.blob())
    dev = torch.device("cuda", 0)
    n = 1 << 20
    pat, off, _ = ia.synth_patterns(text, 8, n, seed=43)
    d_pat, d_off = torch.from_numpy(pat.view(np.int16)).to(dev), torch.from_numpy(off).to(dev)
    d_cnt = torch.zeros(n, dtype=torch.int32, device=dev)
    sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    fn = ia.lib.fmx_diag_lines
    fn.argtypes = [C.c_void_p]
    plan = C.c_void_p()
    planned = "--planned" in sys.argv  # default: what fmx_count_batch_dev runs for this batch (round 4: the caller's order)

    def launch():
        if planned:
            assert ia.lib.fmx_count_ordered_dev(fm.handle, d_pat.data_ptr(), d_off.data_ptr(), plan, n, d_cnt.data_ptr(), None, None, sp) == 0
        else:
            assert ia.lib.fmx_count_batch_dev(fm.handle, d_pat.data_ptr(), d_off.data_ptr(), n, d_cnt.data_ptr(), None, None, sp) == 0

    if planned:
        assert ia.lib.fmx_count_plan_dev(fm.handle, d_pat.data_ptr(), d_off.data_ptr(), n, C.byref(plan), sp) == 0
    print("planned launch" if planned else "fmx_count_batch_dev (planned by the library: %d)" % ia.lib.fmx_count_batch_is_planned(fm.handle, n))
    launch()
    assert fn(None) == 0
    launch()
    words = np.zeros(8 << 19, dtype=np.uint32)
    assert fn(words.ctypes.data) == 0
    bits = np.unpackbits(words.view(np.uint8).reshape(8, -1), axis=1, bitorder="little")[:, :len(sec)].astype(bool)
    per = bits.sum(axis=1)
    union = bits.any(axis=0)
    k = bits.sum(axis=0)
    names = ["other", "mapping entries", "path records", "wavelet cells", "SbcEntry table (8-byte loads: not recorded)", "SbDesc"]
    print("image %d lines of 128 bytes (%.1f MB)" % (len(sec), len(sec) * 128 / 1e6))
    print("lines touched per XCD:", per.tolist(), " sum %d" % per.sum())
    print("union over the XCDs: %d lines; mean number of XCDs touching a touched line: %.2f" % (union.sum(), per.sum() / max(1, union.sum())))
    print("lines by number of XCDs touching them (1..8):", [int((k == i).sum()) for i in range(1, 9)])
    for i, nm in enumerate(names):
        m = sec == i
        if m.any():
            print("  %-44s lines %8d  touched (union) %8d  sum over XCDs %8d" % (nm, int(m.sum()), int(union[m].sum()), int(bits[:, m].sum())))


if __name__ == "__main__":
    main()
