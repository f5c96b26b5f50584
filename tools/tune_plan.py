#!/usr/bin/env python3
"""configs[1] step time by plan options (coarse_bits, plan_fine) with the suffix table resident.  GPU box only."""
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
    batches = []
    for b in range(4):
        pat, off, _ = ia.synth_patterns(text, 8, n, seed=43 + b)
        batches.append((torch.from_numpy(pat.view(np.int16)).to(dev), torch.from_numpy(off).to(dev),
                        torch.zeros(n, dtype=torch.int32, device=dev)))
    stream = torch.cuda.current_stream()
    sp = C.c_void_p(stream.cuda_stream)

    def run(label):
        def step(i):
            d_pat, d_off, d_cnt = batches[i % 4]
            assert ia.lib.fmx_count_batch_dev(fm.handle, d_pat.data_ptr(), d_off.data_ptr(), n, d_cnt.data_ptr(), None, None, sp) == 0
        for i in range(8):
            step(i)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for i in range(40):
                step(i)
            e1.record(stream)
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 40)
        print("%-28s step %.4f ms  checksum %d" % (label, best, int(batches[0][2].sum().item())), flush=True)

    for cb in (12, 10, 11, 13):
        for fine in (1, 0):
            ia.lib.fmx_set_option(b"coarse_bits", cb)
            ia.lib.fmx_set_option(b"plan_fine", fine)
            run("coarse_bits %d plan_fine %d" % (cb, fine))
    ia.lib.fmx_set_option(b"coarse_bits", 12)
    ia.lib.fmx_set_option(b"plan_fine", 1)


if __name__ == "__main__":
    main()
