#!/usr/bin/env python3
"""What a count() step costs by pattern length on the headline index (suffix table: 5 characters): 1,048,576 patterns of m = 2..12
characters — m <= 5 is answered by the table alone (no rank at all), every further character is two ranks.  GPU box only."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
import index4j_amd as ia

text, fm, path = bench.build_or_load_index(ia, 28, 32, "/tmp/fmx_cache")
fm.to_device(0)
dev = torch.device("cuda", 0)
n = 1 << 20
sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for m in (2, 4, 5, 6, 7, 8, 10, 12):
    pat, off, _ = ia.synth_patterns(text, m, n, seed=43)
    d_pat, d_off = torch.from_numpy(pat.view(np.int16)).to(dev), torch.from_numpy(off).to(dev)
    d_cnt = torch.zeros(n, dtype=torch.int32, device=dev)

    def launch():
        assert ia.lib.fmx_count_batch_dev(fm.handle, d_pat.data_ptr(), d_off.data_ptr(), n, d_cnt.data_ptr(), None, None, sp) == 0

    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        launch()
    e1.record()
    torch.cuda.synchronize()
    print("m = %2d: %.4f ms per 1,048,576 patterns (planned: %d)" % (m, e0.elapsed_time(e1) / 20, ia.lib.fmx_count_batch_is_planned(fm.handle, n)))
