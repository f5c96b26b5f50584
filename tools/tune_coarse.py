#!/usr/bin/env python3
"""bins of the plan stage's bucket pass (2^coarse_bits): whole step (plan + count) on configs[1]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    import torch

    import bench
    import index4j_amd as ia
    from bench_configs import timed

    text, fm, path = bench.build_or_load_index(ia, 28, 32, "/tmp/fmx_cache")
    fm.to_device(0)
    dev = torch.device("cuda", 0)
    n = 1 << 20
    stream = torch.cuda.current_stream()
    sp = C.c_void_p(stream.cuda_stream)
    pat, off, _ = ia.synth_patterns(text, 8, n)
    d_pat = torch.from_numpy(pat.view(np.int16)).to(dev)
    d_off = torch.from_numpy(off).to(dev)
    d_cnt = torch.zeros(n, dtype=torch.int32, device=dev)
    ref = None
    for bits in (14, 13, 12, 11, 10, 8, 14):
        assert ia.lib.fmx_set_option(b"coarse_bits", bits) == 0

        def step():
            assert ia.lib.fmx_count_batch_dev(fm.handle, d_pat.data_ptr(), d_off.data_ptr(), n, d_cnt.data_ptr(), None, None, sp) == 0

        perm = C.c_void_p()

        def plan():
            assert ia.lib.fmx_count_plan_dev(fm.handle, d_pat.data_ptr(), d_off.data_ptr(), n, C.byref(perm), sp) == 0

        ms = timed(step, stream, 20)
        ms_plan = timed(plan, stream, 20)
        got = d_cnt.cpu().numpy().copy()
        if ref is None:
            ref = got
        assert (got == ref).all()
        print("coarse_bits %2d: step %.4f ms  (plan %.4f ms, count %.4f ms)" % (bits, ms, ms_plan, ms - ms_plan), flush=True)
    ia.lib.fmx_set_option(b"coarse_bits", 14)


if __name__ == "__main__":
    main()
