#!/usr/bin/env python3
"""configs[1] step by suffix-table depth (option suffix_table_chars; arguments: depths, default 0 2 3 4 5 6 8).  GPU box only."""
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
    dev = torch.device("cuda", 0)
    n = 1 << 20
    batches = []
    for b in range(4):
        pat, off, _ = ia.synth_patterns(text, 8, n, seed=43 + b)
        batches.append((torch.from_numpy(pat.view(np.int16)).to(dev), torch.from_numpy(off).to(dev),
                        torch.zeros(n, dtype=torch.int32, device=dev)))
    stream = torch.cuda.current_stream()
    sp = C.c_void_p(stream.cuda_stream)
    depths = [int(x) for x in sys.argv[1:]] or [0, 2, 3, 4, 5, 6, 8]
    for depth in depths:
        ia.lib.fmx_set_option(b"suffix_table_mb", 0 if depth == 0 else 4096)
        ia.lib.fmx_set_option(b"suffix_table_image_fraction", 0)  # the depth asked for, whatever it takes
        ia.lib.fmx_set_option(b"suffix_table_chars", max(depth, 2))
        t0 = time.perf_counter()
        fm.to_device(0)
        t_dev = time.perf_counter() - t0
        k, nbytes = fm.suffix_table_info()

        def step(i):
            d_pat, d_off, d_cnt = batches[i % 4]
            assert ia.lib.fmx_count_batch_dev(fm.handle, d_pat.data_ptr(), d_off.data_ptr(), n, d_cnt.data_ptr(), None, None, sp) == 0

        for i in range(8):
            step(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for i in range(40):
            step(i)
        e1.record(stream)
        torch.cuda.synchronize()
        sums = [int(b[2].sum().item()) for b in batches]
        print("depth asked %d: table of %d chars, %10.2f MB, to_device %.3f s; step %.4f ms; checksums %s"
              % (depth, k, nbytes / 1e6, t_dev, e0.elapsed_time(e1) / 40, sums[:2]), flush=True)
    ia.lib.fmx_set_option(b"suffix_table_mb", 256)
    ia.lib.fmx_set_option(b"suffix_table_chars", 8)
    ia.lib.fmx_set_option(b"suffix_table_image_fraction", 8)


if __name__ == "__main__":
    main()
