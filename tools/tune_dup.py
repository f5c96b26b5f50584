#!/usr/bin/env python3
"""Issue-vs-memory experiment for k_count: same instruction stream, collapsing address divergence.
  random      : the benchmark batch (suffix-ordered by the library)
  wave-uniform: every 32 consecutive patterns (= one wave) are copies of one pattern
  grid-uniform: the whole batch is one pattern
(GPU box only)"""
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
    batches = {
        "random (library order)": (P, True),
        "random (no order)": (P, False),
        "wave-uniform": (np.repeat(P[: n // 32], 32, axis=0), False),
        "grid-uniform": (np.repeat(P[:1], n, axis=0), False),
    }
    d_off = torch.from_numpy(off).to(dev)
    d_cnt = torch.zeros(n, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream()
    for name, (arr, order) in batches.items():
        d_pat = torch.from_numpy(np.ascontiguousarray(arr).reshape(-1).view(np.int16)).to(dev)
        perm = C.c_void_p()
        if order:
            assert ia.lib.fmx_count_plan_dev(fm.handle, d_pat.data_ptr(), d_off.data_ptr(), n, C.byref(perm), C.c_void_p(stream.cuda_stream)) == 0
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
        print("%-24s k_count %.4f ms" % (name, min(ts)), flush=True)


if __name__ == "__main__":
    main()
