#!/usr/bin/env python3
"""Experiment: configs[2] locate with the batch rearranged by the SA row of each pattern's interval (what a plan-ordered
k_locate_walk would see) against the caller's (random) order.  Host-side rearrangement only.  GPU box only."""
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
for K in (8192, 16384, 32768, 100_000, 1 << 20):
    M, m = 16, 8
    pat, off, _ = ia.synth_patterns(text, m, K, seed=43)
    rows = pat.reshape(K, m)
    p_rows = rows
    d_off = torch.from_numpy(off).to(dev)
    d_locs = torch.zeros(K * M, dtype=torch.int32, device=dev)
    d_found = torch.zeros(K, dtype=torch.int32, device=dev)
    d_st = torch.zeros(K, dtype=torch.int32, device=dev)
    d_rng = torch.zeros(2 * K, dtype=torch.int32, device=dev)

    def run(p):
        d_pat = torch.from_numpy(np.ascontiguousarray(p).view(np.int16)).to(dev)

        def f():
            assert ia.lib.fmx_locate_batch_dev(fm.handle, d_pat.data_ptr(), d_off.data_ptr(), K, M, d_locs.data_ptr(), M, d_found.data_ptr(), None,
                                               d_st.data_ptr(), d_rng.data_ptr(), sp) == 0
        for _ in range(3):
            f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            f()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 10, int(d_found.sum().item())

    assert ia.lib.fmx_set_option(b"walk_order_min", 0) == 0
    t0, h0 = run(rows.reshape(-1))
    ref = d_locs.clone()
    assert ia.lib.fmx_set_option(b"walk_order_min", 1) == 0
    for fine, cb in ((0, 12), (1, 12), (0, 13), (1, 13), (1, 10)):
        assert ia.lib.fmx_set_option(b"walk_fine", fine) == 0 and ia.lib.fmx_set_option(b"coarse_bits", cb) == 0
        tw, hw = run(rows.reshape(-1))
        assert hw == h0 and torch.equal(ref, d_locs)
        print("   walk_fine %d coarse_bits %d: %.4f ms" % (fine, cb, tw), flush=True)
    assert ia.lib.fmx_set_option(b"walk_fine", 1) == 0 and ia.lib.fmx_set_option(b"coarse_bits", 12) == 0
    print("locate %8d patterns: walk in the caller's order %.4f ms, by the first row of the ranges %.4f ms (%+.1f %%)" % (K, t0, tw, (tw / t0 - 1) * 100), flush=True)
    assert ia.lib.fmx_set_option(b"walk_order_min", 0) == 0
    rng_ = d_rng.cpu().numpy().reshape(K, 2).astype(np.int64)
    start = rng_[:, 0]
    hits = np.minimum(np.maximum(rng_[:, 1] - rng_[:, 0], 0), M)
    hit_rows = np.concatenate([np.arange(a, a + h) for a, h in zip(start[:200000], hits[:200000])])
    print("   distinct patterns %d of %d, distinct ranges %d; of the first 200,000 patterns' %d hit rows %d are distinct"
          % (len(np.unique(rows_ := np.ascontiguousarray(p_rows).view([("", p_rows.dtype)] * m))), K, len(np.unique(start)), len(hit_rows), len(np.unique(hit_rows))), flush=True)
    order = np.argsort(start, kind="stable")
    t1, h1 = run(rows[order].reshape(-1))
    assert h0 == h1
    print("locate %8d patterns x <= %d hits: caller's order %.4f ms, sorted by SA row %.4f ms (%+.1f %%)" % (K, M, t0, t1, (t1 / t0 - 1) * 100), flush=True)
