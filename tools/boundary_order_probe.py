#!/usr/bin/env python3
"""Experiment: configs[3] (extractUntilBoundary of 100,000 hit locations, sampleRate 64) with the queries sorted by text position
(equal and neighbouring `from` fetch the same sample intervals) against the caller's order.  Host-side rearrangement.  GPU box only."""
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
text, fm, _ = bench.build_or_load_index(ia, 28, 64, "/tmp/fmx_cache")
fm.to_device(0)
K, m, cap = 100_000, 8, 1024
pat, off, _ = ia.synth_patterns(text, m, K, seed=43)
locs, found, st = fm.locate_batch(pat, off, 16)
frm = np.ascontiguousarray(locs[:, 0])
print("distinct positions: %d of %d" % (len(np.unique(frm)), K))
d_dst = torch.zeros(K * cap, dtype=torch.int16, device=dev)
d_len = torch.zeros(K, dtype=torch.int32, device=dev)
d_st = torch.zeros(K, dtype=torch.int32, device=dev)
d_aux = torch.zeros(K, dtype=torch.int32, device=dev)
for label, f, opt in (("caller's order", frm, 0), ("sorted on the host", np.sort(frm), 0), ("ordered on the device", frm, 1), ("caller's order", frm, 0)):
    assert ia.lib.fmx_set_option(b"boundary_order_min", opt) == 0
    d_f = torch.from_numpy(f).to(dev)

    def run():
        assert ia.lib.fmx_extract_boundary_batch_dev(fm.handle, d_f.data_ptr(), K, 10, 0, d_dst.data_ptr(), cap, 0, d_len.data_ptr(), None,
                                                     d_st.data_ptr(), d_aux.data_ptr(), sp) == 0
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(8):
        run()
    e1.record()
    torch.cuda.synchronize()
    print("%-20s %.4f ms  (chars %d)" % (label, e0.elapsed_time(e1) / 8, int(d_len.sum().item())), flush=True)
