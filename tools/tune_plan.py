#!/usr/bin/env python3
"""configs[1] step and its two stages by the plan stage's options (coarse_bits, sort_bits, plan_fine), final kernels.  GPU box only."""
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
    sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    batches = []
    for b in range(4):
        pat, off, _ = ia.synth_patterns(text, 8, n, seed=43 + b)
        batches.append((torch.from_numpy(pat.view(np.int16)).to(dev), torch.from_numpy(off).to(dev), torch.zeros(n, dtype=torch.int32, device=dev)))

    def timed(fn, reps=40):
        for i in range(6):
            fn(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(reps):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    def whole(i):
        p, o, c = batches[i % 4]
        assert ia.lib.fmx_count_batch_dev(fm.handle, p.data_ptr(), o.data_ptr(), n, c.data_ptr(), None, None, sp) == 0

    plan = C.c_void_p()

    def plan_only(i):
        p, o, c = batches[i % 4]
        assert ia.lib.fmx_count_plan_dev(fm.handle, p.data_ptr(), o.data_ptr(), n, C.byref(plan), sp) == 0

    def count_only(i):
        p, o, c = batches[0]
        assert ia.lib.fmx_count_ordered_dev(fm.handle, p.data_ptr(), o.data_ptr(), plan, n, c.data_ptr(), None, None, sp) == 0

    settings = [dict(), dict(plan_fine=0), dict(coarse_bits=10), dict(coarse_bits=11), dict(coarse_bits=13),
                dict(sort_bits=21), dict(sort_bits=24), dict(sort_bits=32), dict()]
    defaults = dict(plan_fine=1, coarse_bits=12, sort_bits=28)
    for st in settings:
        cur = dict(defaults, **st)
        for k, v in cur.items():
            rc = ia.lib.fmx_set_option(k.encode(), v)
            assert rc == 0, (k, v)
        t_whole = timed(whole)
        t_plan = timed(plan_only)
        plan_only(0)
        t_count = timed(count_only)
        print("%-28s step %.4f ms  plan %.4f  k_count %.4f  checksum %d" % (st or "defaults", t_whole, t_plan, t_count,
                                                                             int(batches[0][2].sum().item())), flush=True)
    for k, v in defaults.items():
        ia.lib.fmx_set_option(k.encode(), v)


if __name__ == "__main__":
    main()
