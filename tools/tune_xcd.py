#!/usr/bin/env python3
"""A/B of the XCD-aware block order of k_count (interleaved, one process; GPU box only)."""
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
    res = {}
    ref = None
    for rnd in range(6):
        for remap in (0, 1):
            for gpc in (8, 16, 32):
                ia.lib.fmx_set_option(b"xcd_remap", remap)
                ia.lib.fmx_set_option(b"groups_per_cu", gpc)
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                ev0.record(stream)
                for _ in range(5):
                    assert ia.lib.fmx_count_ordered_dev(fm.handle, d_pat.data_ptr(), d_off.data_ptr(), perm, n, d_cnt.data_ptr(), None,
                                                        None, C.c_void_p(stream.cuda_stream)) == 0
                ev1.record(stream)
                torch.cuda.synchronize()
                res.setdefault((remap, gpc), []).append(ev0.elapsed_time(ev1) / 5)
                c = d_cnt.cpu().numpy()
                if ref is None:
                    ref = c
                assert (c == ref).all()
    for k, v in sorted(res.items(), key=lambda kv: np.median(kv[1])):
        print("xcd_remap=%d groups_per_cu=%2d  median %.4f ms  min %.4f ms" % (k[0], k[1], np.median(v), min(v)))


if __name__ == "__main__":
    main()
