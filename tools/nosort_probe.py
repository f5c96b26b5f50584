#!/usr/bin/env python3
"""Does the plan stage (suffix order of the batch) still pay with the deeper suffix table?  count() planned vs in the caller's order
(option sort_min above the batch size) on: the headline batch (8 chars, 70 symbols), 16- and 31-char patterns of the same text, the
reference-shaped queries (8..31 chars, 1,099 symbols), and locate of configs[2].  GPU box only."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import index4j_amd as ia  # noqa: E402
from index4j_amd import workload  # noqa: E402

dev = torch.device("cuda", 0)
sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def timed(fn, reps=30):
    for i in range(5):
        fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def count_case(name, fm, batches, n):
    out = []
    for sort_min in (16384, 1 << 30):
        ia.lib.fmx_set_option(b"sort_min", sort_min)

        def step(i):
            p, o, c = batches[i % len(batches)]
            assert ia.lib.fmx_count_batch_dev(fm.handle, p.data_ptr(), o.data_ptr(), n, c.data_ptr(), None, None, sp) == 0

        out.append((timed(step), int(batches[0][2].sum().item())))
    ia.lib.fmx_set_option(b"sort_min", 16384)
    assert out[0][1] == out[1][1]
    print("%-58s planned %.4f ms   caller's order %.4f ms   (%+.1f %%)" % (name, out[0][0], out[1][0], (out[1][0] / out[0][0] - 1) * 100), flush=True)


def dev_batch(pat, off, n):
    return (torch.from_numpy(np.ascontiguousarray(pat).view(np.int16)).to(dev), torch.from_numpy(np.ascontiguousarray(off)).to(dev),
            torch.zeros(n, dtype=torch.int32, device=dev))


text, fm, _ = bench.build_or_load_index(ia, 28, 32, "/tmp/fmx_cache")
fm.to_device(0)
n = 1 << 20
for m in (8, 12, 16, 31):
    bs = [dev_batch(*ia.synth_patterns(text, m, n, seed=43 + b)[:2], n) for b in range(4)]
    count_case("log text, %d-char patterns, table of %d chars" % (m, fm.suffix_table_info()[0]), fm, bs, n)
for k in (65536, 262144):
    bs = [dev_batch(*ia.synth_patterns(text, 8, k, seed=53 + b)[:2], k) for b in range(4)]
    count_case("log text, 8-char patterns, batch of %d" % k, fm, bs, k)
# locate of configs[2]
K, M = 100_000, 16
pat, off, _ = ia.synth_patterns(text, 8, K, seed=43)
p, o, _c = dev_batch(pat, off, K)
d_locs = torch.zeros(K * M, dtype=torch.int32, device=dev)
d_found = torch.zeros(K, dtype=torch.int32, device=dev)
d_st = torch.zeros(K, dtype=torch.int32, device=dev)
d_rng = torch.zeros(2 * K, dtype=torch.int32, device=dev)
res = []
for sort_min in (16384, 1 << 30):
    ia.lib.fmx_set_option(b"sort_min", sort_min)
    res.append(timed(lambda i: ia.lib.fmx_locate_batch_dev(fm.handle, p.data_ptr(), o.data_ptr(), K, M, d_locs.data_ptr(), M, d_found.data_ptr(),
                                                           None, d_st.data_ptr(), d_rng.data_ptr(), sp), 10))
ia.lib.fmx_set_option(b"sort_min", 16384)
print("%-58s planned %.4f ms   caller's order %.4f ms   (%+.1f %%)" % ("configs[2] locate, 100,000 patterns", res[0], res[1], (res[1] / res[0] - 1) * 100), flush=True)
fm.close()
t = workload.reference_text(28)
f2 = ia.FmIndex(t, 32, True, device=0, build_device=0)
pat, off, _ = workload.reference_queries(t, n)
count_case("1,099-symbol text, queries of 8..31 chars, table of %d" % f2.suffix_table_info()[0], f2, [dev_batch(pat, off, n)], n)
pat8, off8, _ = ia.synth_patterns(t, 8, n, seed=7)
count_case("1,099-symbol text, 8-char patterns", f2, [dev_batch(pat8, off8, n)], n)
