#!/usr/bin/env python3
"""The plan stage ordered by SA row (option plan_sa_key) against the code-key order and the caller's order: headline batch and other
shapes; coarse bits 12 / 13, with and without the fine pass.  GPU box only."""
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


def opt(**kw):
    for k, v in kw.items():
        assert ia.lib.fmx_set_option(k.encode(), v) == 0, (k, v)


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


def dev_batch(pat, off, n):
    return (torch.from_numpy(np.ascontiguousarray(pat).view(np.int16)).to(dev), torch.from_numpy(np.ascontiguousarray(off)).to(dev),
            torch.zeros(n, dtype=torch.int32, device=dev))


def case(name, fm, batches, n, settings):
    ref = None
    out = []
    for label, kw in settings:
        opt(**kw)

        def step(i):
            p, o, c = batches[i % len(batches)]
            assert ia.lib.fmx_count_batch_dev(fm.handle, p.data_ptr(), o.data_ptr(), n, c.data_ptr(), None, None, sp) == 0

        t = timed(step)
        chk = int(batches[0][2].sum().item())
        ref = chk if ref is None else ref
        assert chk == ref
        out.append("%s %.4f" % (label, t))
    print("%-46s %s" % (name, "  ".join(out)), flush=True)
    opt(plan_min_per_string=16, plan_sa_key=1, plan_fine=1, coarse_bits=12)


S = [("caller", dict(plan_min_per_string=1 << 30)), ("code-key", dict(plan_min_per_string=0, plan_sa_key=0)),
     ("sa12", dict(plan_min_per_string=0, plan_sa_key=1, coarse_bits=12)), ("sa13", dict(plan_min_per_string=0, plan_sa_key=1, coarse_bits=13)),
     ("sa12+fine", dict(plan_min_per_string=0, plan_sa_key=1, coarse_bits=12, plan_fine=2)), ("sa10", dict(plan_min_per_string=0, plan_sa_key=1, coarse_bits=10))]
text, fm, _ = bench.build_or_load_index(ia, 28, 32, "/tmp/fmx_cache")
fm.to_device(0)
n = 1 << 20
for m in (8, 16, 31):
    bs = [dev_batch(*ia.synth_patterns(text, m, n, seed=43 + b)[:2], n) for b in range(4)]
    case("log, %d chars, 1M" % m, fm, bs, n, S)
for k in (32768, 65536, 131072, 262144, 524288):
    bs = [dev_batch(*ia.synth_patterns(text, 8, k, seed=53 + b)[:2], k) for b in range(4)]
    case("log, 8 chars, batch %d" % k, fm, bs, k, S[:3])
fm.close()
t = workload.reference_text(28)
f2 = ia.FmIndex(t, 32, True, device=0, build_device=0)
pat, off, _ = workload.reference_queries(t, n)
case("1,099 symbols, 8..31 chars, 1M", f2, [dev_batch(pat, off, n)], n, S[:4])
pat8, off8, _ = ia.synth_patterns(t, 8, n, seed=7)
case("1,099 symbols, 8 chars, 1M", f2, [dev_batch(pat8, off8, n)], n, S[:4])
