#!/usr/bin/env python3
"""Experiment: configs[1] batches on ONE stream (plan + count back to back) vs alternating batches on TWO streams, so that
one batch's plan stage overlaps the other's k_count (GPU box only).  Prints ms per batch."""
import ctypes as C
import os
import sys
import time

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
    batches = []
    for b in range(4):
        pat, off, _ = ia.synth_patterns(text, 8, n, seed=43 + b)
        batches.append((torch.from_numpy(pat.view(np.int16)).to(dev), torch.from_numpy(off).to(dev),
                        torch.zeros(n, dtype=torch.int32, device=dev)))
    for n_streams in (1, 2, 3, 1, 2):
        streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams)]

        def step(i):
            st = streams[i % n_streams]
            d_pat, d_off, d_cnt = batches[i % 4]
            assert ia.lib.fmx_count_batch_dev(fm.handle, d_pat.data_ptr(), d_off.data_ptr(), n, d_cnt.data_ptr(), None, None,
                                              C.c_void_p(st.cuda_stream)) == 0

        for i in range(8):
            step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        steps = 200
        for i in range(steps):
            step(i)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        sums = [int(b[2].sum().item()) for b in batches]
        print("%d stream(s): %.4f ms per batch of %d patterns (%.3e patterns/s)  checksums %s" % (n_streams, dt * 1e3, n, n / dt, sums), flush=True)


if __name__ == "__main__":
    main()
