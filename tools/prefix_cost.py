#!/usr/bin/env python3
"""Experiment: what do the FIRST LF-steps of a count() batch cost?  Times the planned k_count of configs[1]'s batch
truncated to its last L characters, L = 1 .. 8 (GPU box only)."""
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
    full = pat.reshape(n, 8)
    stream = torch.cuda.current_stream()
    sp = C.c_void_p(stream.cuda_stream)
    for L in (1, 2, 3, 4, 5, 6, 8):
        p = np.ascontiguousarray(full[:, 8 - L:]).reshape(-1)
        o = (np.arange(n + 1, dtype=np.int64) * L).astype(np.int32)
        d_pat = torch.from_numpy(p.view(np.int16)).to(dev)
        d_off = torch.from_numpy(o).to(dev)
        d_cnt = torch.zeros(n, dtype=torch.int32, device=dev)
        plan = C.c_void_p()
        assert ia.lib.fmx_count_plan_dev(fm.handle, d_pat.data_ptr(), d_off.data_ptr(), n, C.byref(plan), sp) == 0
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(5):
                assert ia.lib.fmx_count_ordered_dev(fm.handle, d_pat.data_ptr(), d_off.data_ptr(), plan, n, d_cnt.data_ptr(),
                                                    None, None, sp) == 0
            e1.record(stream)
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 5)
        print("last %d chars (%d LF-steps per pattern): k_count %.4f ms" % (L, 2 * (L - 1), min(ts)), flush=True)


if __name__ == "__main__":
    main()
