#!/usr/bin/env python3
"""configs[3] (extractUntilBoundary of 100,000 hit locations, sampleRate 64): how the work is spread over the queries — line
lengths, executed LF-steps per query — and how the kernel's time grows with the batch (is the chip full?).  GPU box only."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


MODE = int(os.environ.get("FMX_MODE", "0"))


def main():
    import torch

    import bench
    import index4j_amd as ia

    text, fm, path = bench.build_or_load_index(ia, 28, 64, "/tmp/fmx_cache")
    fm.to_device(0)
    K0, m = 100_000, 8
    pat, off, pos = ia.synth_patterns(text, m, 8 * K0, seed=44)
    locs, found, st = fm.locate_batch(pat, off, 1)
    froms_all = np.ascontiguousarray(locs[:, 0])
    dst, ol, st2, aux, steps = fm.extract_boundary_batch(froms_all[:K0], "\n", 0, 1024, want_steps=True)
    q = lambda a, p: float(np.percentile(a, p))
    print("line lengths:   mean %.0f  p50 %.0f  p90 %.0f  p99 %.0f  max %d" % (ol.mean(), q(ol, 50), q(ol, 90), q(ol, 99), ol.max()))
    print("executed steps: mean %.0f  p50 %.0f  p90 %.0f  p99 %.0f  max %d   (sum %d)" % (steps.mean(), q(steps, 50), q(steps, 90),
                                                                                          q(steps, 99), steps.max(), steps.sum()))
    # steps of the slowest query of every group of 16 consecutive queries (one wave = 16 queries x 4 lanes)
    w = steps[: K0 // 16 * 16].reshape(-1, 16)
    print("per wave of 16 queries: mean of the maxima %.0f against the mean %.0f (x%.2f)" % (w.max(axis=1).mean(), steps.mean(),
                                                                                          w.max(axis=1).mean() / steps.mean()))
    dev = torch.device("cuda", 0)
    sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    fills = [int(x) for x in os.environ.get("FMX_FIRST_FILLS", "").split(",") if x] or [None]
    for fill in fills:
      if fill is not None:
        assert ia.lib.fmx_set_option(b"boundary_first_fill", fill) == 0
        print("boundary_first_fill = %d" % fill)
      for K in (25_000, 100_000, 800_000) if fill is not None else (25_000, 50_000, 100_000, 200_000, 400_000, 800_000):
          d_from = torch.from_numpy(froms_all[:K]).to(dev)
          d_dst = torch.zeros(K * 1024, dtype=torch.int16, device=dev)
          d_len = torch.zeros(K, dtype=torch.int32, device=dev)
          d_st = torch.zeros(K, dtype=torch.int32, device=dev)
          d_aux = torch.zeros(K, dtype=torch.int32, device=dev)

          def call():
              assert ia.lib.fmx_extract_boundary_batch_dev(fm.handle, d_from.data_ptr(), K, 10, MODE, d_dst.data_ptr(), 1024, 0, d_len.data_ptr(),
                                                           None, d_st.data_ptr(), d_aux.data_ptr(), sp) == 0

          for _ in range(2):
              call()
          e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
          e0.record()
          for _ in range(4):
              call()
          e1.record()
          torch.cuda.synchronize()
          ms = e0.elapsed_time(e1) / 4
          print("K = %7d queries: %.3f ms  (%.1f ns per query)" % (K, ms, ms * 1e6 / K), flush=True)
          del d_dst


if __name__ == "__main__":
    main()
