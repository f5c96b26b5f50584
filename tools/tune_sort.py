#!/usr/bin/env python3
"""How much ordering does k_count need?  Times the kernel on batches sorted by their last k chars
(k = 0..8) and prices a device radix sort of the keys (torch.sort as a stand-in for hipCUB)."""
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
    P = pat.reshape(n, 8)
    d_off = torch.from_numpy(off).to(dev)
    d_cnt = torch.zeros(n, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream()

    def run(arr, reps=10):
        d_pat = torch.from_numpy(arr.view(np.int16)).to(dev)
        best = []
        for _ in range(3):
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            ev0.record(stream)
            for _ in range(reps):
                ia.lib.fmx_count_batch_dev(fm.handle, d_pat.data_ptr(), d_off.data_ptr(), n, d_cnt.data_ptr(), None, None,
                                           C.c_void_p(stream.cuda_stream))
            ev1.record(stream)
            torch.cuda.synchronize()
            best.append(ev0.elapsed_time(ev1) / reps)
        return min(best)

    for k in range(0, 9):
        key = np.zeros(n, dtype=np.uint64)
        for j in range(k):
            key = (key << np.uint64(7)) | (P[:, 7 - j].astype(np.uint64) & np.uint64(127))
        order = np.argsort(key, kind="stable") if k else np.arange(n)
        print("sorted by last %d chars: kernel %.4f ms" % (k, run(np.ascontiguousarray(P[order]).reshape(-1))), flush=True)
    # price of sorting on the device
    for bits in (14, 21, 28, 35, 56):
        keys = torch.randint(0, 1 << bits, (n,), dtype=torch.int64, device=dev)
        for dt in (torch.int64, torch.int32):
            if dt == torch.int32 and bits > 31:
                continue
            kk = keys.to(dt)
            torch.cuda.synchronize()
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            for _ in range(10):
                v, i = torch.sort(kk)
            ev1.record()
            torch.cuda.synchronize()
            print("torch.sort 1M %s keys (%d significant bits): %.4f ms" % (dt, bits, ev0.elapsed_time(ev1) / 10), flush=True)


if __name__ == "__main__":
    main()
