#!/usr/bin/env python3
"""Experiment: does it pay to give each XCD its own eighth of the SA?  LF-mapping keeps the order of rows that are preceded by the same
character, so patterns whose current SA interval lies in eighth g of the rows stay, after every further step, inside a set of rows
that is disjoint from the other eighths' — an XCD that only sees eighth g touches 1/8 of the lines (its 4 MB L2 against 32 MB now).
Here the batch is only REARRANGED on the host (no library change): workgroup b of k_count takes patterns [256 b, 256 b + 256) and runs
on XCD b mod 8, so the patterns of eighth g are put at the workgroups with b mod 8 == g.  GPU box only."""
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
n, m = 1 << 20, 8
depth = fm.suffix_table_info()[0]


def timed(fn, reps=40):
    for _ in range(6):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def run(pat):
    d_pat = torch.from_numpy(np.ascontiguousarray(pat).view(np.int16)).to(dev)
    d_off = torch.from_numpy((np.arange(n + 1, dtype=np.int64) * m).astype(np.int32)).to(dev)
    d_cnt = torch.zeros(n, dtype=torch.int32, device=dev)
    t = timed(lambda: ia.lib.fmx_count_batch_dev(fm.handle, d_pat.data_ptr(), d_off.data_ptr(), n, d_cnt.data_ptr(), None, None, sp))
    return t, int(d_cnt.sum(dtype=torch.int64).item())


pat, off, _ = ia.synth_patterns(text, m, n, seed=43)
rows = pat.reshape(n, m)
# SA interval start of every pattern's last `depth` characters: locate's range output of the cut patterns
tail = np.ascontiguousarray(rows[:, m - depth:]).reshape(-1)
d_tail = torch.from_numpy(tail.view(np.int16)).to(dev)
d_toff = torch.from_numpy((np.arange(n + 1, dtype=np.int64) * depth).astype(np.int32)).to(dev)
d_rng = torch.zeros(2 * n, dtype=torch.int32, device=dev)
d_locs = torch.zeros(n, dtype=torch.int32, device=dev)
d_found = torch.zeros(n, dtype=torch.int32, device=dev)
d_st = torch.zeros(n, dtype=torch.int32, device=dev)
assert ia.lib.fmx_locate_batch_dev(fm.handle, d_tail.data_ptr(), d_toff.data_ptr(), n, 1, d_locs.data_ptr(), 1, d_found.data_ptr(), None,
                                   d_st.data_ptr(), d_rng.data_ptr(), sp) == 0
torch.cuda.synchronize()
start = d_rng.cpu().numpy().reshape(n, 2)[:, 0].astype(np.int64)
t0, chk = run(pat)
print("caller's order (random)                              %.4f ms" % t0)
order = np.argsort(start, kind="stable")
t1, c1 = run(rows[order].reshape(-1))
print("sorted by SA position of the tabulated suffix        %.4f ms" % t1)
# eighth g of the sorted batch -> the workgroups with b mod 8 == g
blocks = n // 256
for within in ("sorted", "shuffled"):
    perm = np.empty(n, np.int64)
    rng = np.random.default_rng(1)
    for g in range(8):
        members = order[g * (n // 8):(g + 1) * (n // 8)]
        if within == "shuffled":
            members = rng.permutation(members)
        bs = np.arange(g, blocks, 8)  # workgroups of XCD g
        slots = (bs[:, None] * 256 + np.arange(256)[None, :]).reshape(-1)
        perm[slots] = members
    t2, c2 = run(rows[perm].reshape(-1))
    assert c2 == chk
    print("one eighth of the SA per XCD, %-8s inside           %.4f ms" % (within, t2))
assert c1 == chk
# how fine must the order be?  buckets of 2^s rows of the SA, random inside
rng = np.random.default_rng(2)
for s in (28, 24, 22, 20, 18, 16, 14, 12, 10, 8, 0):
    key = (start >> s) * (1 << 21) + rng.integers(0, 1 << 21, n)
    o2 = np.argsort(key, kind="stable")
    t3, c3 = run(rows[o2].reshape(-1))
    assert c3 == chk
    print("buckets of 2^%-2d SA rows (%8d buckets), random inside   %.4f ms" % (s, (1 << 28 >> s) if s <= 28 else 1, t3), flush=True)
