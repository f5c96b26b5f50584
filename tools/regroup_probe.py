#!/usr/bin/env python3
"""A/B of option regroup_by_length: the headline batch (one length) and batches of mixed lengths (8..31 chars) on the same index.
GPU box only."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import index4j_amd as ia  # noqa: E402

dev = torch.device("cuda", 0)
sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
text, fm, _ = bench.build_or_load_index(ia, 28, 32, "/tmp/fmx_cache")
fm.to_device(0)
n_text = len(text)


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


rng = np.random.default_rng(7)
for label, n, lo, hi in (("8 chars", 1 << 20, 8, 8), ("8 chars", 1 << 18, 8, 8), ("8..31 chars", 1 << 20, 8, 31), ("8..31 chars", 1 << 18, 8, 31),
                         ("1..16 chars", 1 << 20, 1, 16)):
    lens = rng.integers(lo, hi + 1, n)
    starts = rng.integers(0, n_text - 32, n)
    off = np.zeros(n + 1, np.int32)
    off[1:] = np.cumsum(lens)
    idx = np.repeat(starts, lens) + (np.arange(off[-1]) - np.repeat(off[:-1], lens))
    pat = np.ascontiguousarray(ia.as_chars(text)[idx])
    d_pat = torch.from_numpy(pat.view(np.int16)).to(dev)
    d_off = torch.from_numpy(off).to(dev)
    d_cnt = torch.zeros(n, dtype=torch.int32, device=dev)
    out = []
    for opt in (0, 1):
        assert ia.lib.fmx_set_option(b"regroup_by_length", opt) == 0
        t = timed(lambda: ia.lib.fmx_count_batch_dev(fm.handle, d_pat.data_ptr(), d_off.data_ptr(), n, d_cnt.data_ptr(), None, None, sp))
        out.append((t, d_cnt.clone()))
    assert torch.equal(out[0][1], out[1][1])
    print("%-12s %8d patterns: as handed out %.4f ms, regrouped by length %.4f ms (%+.1f %%)" % (label, n, out[0][0], out[1][0],
                                                                                                (out[1][0] / out[0][0] - 1) * 100), flush=True)
