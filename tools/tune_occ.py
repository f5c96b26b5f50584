#!/usr/bin/env python3
"""Occupancy experiment for k_count: pad dynamic LDS so that fewer workgroups fit per CU and time the
ordered kernel alone (GPU box only)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch

    import bench
    import index4j_amd as ia

    text, fm, path = bench.build_or_load_index(ia, 28, 32, "/tmp/fmx_cache")
    fm.to_device(0)
    dev = torch.device("cuda", 0)
    n = 1 << 20
    pat, off, _ = ia.synth_patterns(text, 8, n)
    d_pat = torch.from_numpy(pat.view(np.int16)).to(dev)
    d_off = torch.from_numpy(off).to(dev)
    d_cnt = torch.zeros(n, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream()
    perm = C.c_void_p()
    assert ia.lib.fmx_count_plan_dev(fm.handle, d_pat.data_ptr(), d_off.data_ptr(), n, C.byref(perm), C.c_void_p(stream.cuda_stream)) == 0
    torch.cuda.synchronize()
    for pad in (0, 43, 70, 96, 0):
        # 10 KiB static (superblock header cache) + pad: 160 KiB / (10 + pad) workgroups of 8 waves per CU
        assert ia.lib.fmx_set_option(b"lds_pad_kb", pad) == 0
        ts = []
        for _ in range(3):
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            ev0.record(stream)
            for _ in range(5):
                assert ia.lib.fmx_count_ordered_dev(fm.handle, d_pat.data_ptr(), d_off.data_ptr(), perm, n, d_cnt.data_ptr(), None,
                                                    None, C.c_void_p(stream.cuda_stream)) == 0
            ev1.record(stream)
            torch.cuda.synchronize()
            ts.append(ev0.elapsed_time(ev1) / 5)
        wgs = min(4, 160 // (10 + pad))
        print("lds pad %2d KiB -> %d workgroups (%2d waves) per CU: k_count %.4f ms" % (pad, wgs, wgs * 8, min(ts)), flush=True)


if __name__ == "__main__":
    main()
