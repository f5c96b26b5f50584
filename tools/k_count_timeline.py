#!/usr/bin/env python3
"""When the workgroups of one headline k_count launch start and end, and on which XCD (diagnostic build of the library:
make -C index4j_amd/csrc EXTRA_DEFS=-DFMX_DIAG_TIMELINE, saved as index4j_amd/libfmx_diag.so; run with
FMX_LIBRARY=$PWD/index4j_amd/libfmx_diag.so).  GPU box only."""
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
    pat, off, _ = ia.synth_patterns(text, 8, n, seed=43)
    d_pat, d_off = torch.from_numpy(pat.view(np.int16)).to(dev), torch.from_numpy(off).to(dev)
    d_cnt = torch.zeros(n, dtype=torch.int32, device=dev)
    sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    groups = 4096
    buf = np.zeros(groups * 3, dtype=np.uint64)
    fn = ia.lib.fmx_diag_timeline
    fn.argtypes = [C.c_void_p, C.c_int]
    # k_count alone, back to back on one plan, against plan + k_count (what does the plan stage's traffic cost k_count?)
    plan = C.c_void_p()
    assert ia.lib.fmx_count_plan_dev(fm.handle, d_pat.data_ptr(), d_off.data_ptr(), n, C.byref(plan), sp) == 0

    def ordered():
        assert ia.lib.fmx_count_ordered_dev(fm.handle, d_pat.data_ptr(), d_off.data_ptr(), plan, n, d_cnt.data_ptr(), None, None, sp) == 0

    def whole():
        assert ia.lib.fmx_count_batch_dev(fm.handle, d_pat.data_ptr(), d_off.data_ptr(), n, d_cnt.data_ptr(), None, None, sp) == 0

    for name, fn_step in (("k_count alone, back to back", ordered), ("plan + k_count", whole), ("k_count alone, back to back", ordered)):
        if fn_step is ordered:
            assert ia.lib.fmx_count_plan_dev(fm.handle, d_pat.data_ptr(), d_off.data_ptr(), n, C.byref(plan), sp) == 0
        for _ in range(5):
            fn_step()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40):
            fn_step()
        e1.record()
        torch.cuda.synchronize()
        print("%s: %.4f ms per call" % (name, e0.elapsed_time(e1) / 40), flush=True)
    alone = "--alone" in sys.argv
    for rep in range(6):
        if alone:
            ordered()
        else:
            whole()
        torch.cuda.synchronize()
        assert fn(buf.ctypes.data, groups) == 0
        if rep < 3:
            continue
        t = buf.reshape(groups, 3)
        t0 = t[:, 0].min()
        start = (t[:, 0] - t0).astype(np.float64) / 100.0  # us
        end = (t[:, 1] - t0).astype(np.float64) / 100.0
        xcc = (t[:, 2] >> np.uint64(32)).astype(np.int64) & 0xf
        total = end.max()
        dur = end - start
        print("rep %d: kernel span %.1f us; workgroup duration min %.1f median %.1f p90 %.1f max %.1f us; sum of durations / (span x 1024 "
              "resident) = %.3f" % (rep, total, dur.min(), np.median(dur), np.percentile(dur, 90), dur.max(), dur.sum() / (total * 1024)))
        # active workgroups over time
        edges = np.linspace(0, total, 25)
        act = [int(((start < b) & (end > a)).sum()) for a, b in zip(edges[:-1], edges[1:])]
        print("   workgroups alive per 1/24 of the span:", act)
        print("   last start at %.1f us; ends: p50 %.1f p90 %.1f p99 %.1f" % (start.max(), np.median(end), np.percentile(end, 90), np.percentile(end, 99)))
        per = [(int((xcc == x).sum()), float(end[xcc == x].max()) if (xcc == x).any() else 0.0) for x in range(8)]
        print("   per XCD (workgroups, last end us):", [(c, round(e, 1)) for c, e in per])
        # duration by position in the sorted order
        q = groups // 8
        print("   mean duration by eighth of the sorted order:", [round(float(dur[i * q:(i + 1) * q].mean()), 1) for i in range(8)])


if __name__ == "__main__":
    main()
