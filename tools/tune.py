#!/usr/bin/env python3
"""A/B timing of launch tunables for the full count path (order + k_count), interleaved in ONE
process (GPU box only).  usage: python tools/tune.py [--text-log2 28] [--rounds 5]"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--text-log2", type=int, default=28)
    ap.add_argument("--patterns", type=int, default=1 << 20)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--cache-dir", default="/tmp/fmx_cache")
    args = ap.parse_args()
    import torch

    import bench
    import index4j_amd as ia

    text, fm, path = bench.build_or_load_index(ia, args.text_log2, 32, args.cache_dir)
    fm.to_device(0)
    dev = torch.device("cuda", 0)
    n = args.patterns
    pat, off, _ = ia.synth_patterns(text, 8, n)
    d_pat = torch.from_numpy(pat.view(np.int16)).to(dev)
    d_off = torch.from_numpy(off).to(dev)
    d_cnt = torch.zeros(n, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream()
    results = {}
    ref = None
    variants = [(sb, b, g) for sb in (0, 21, 24, 28, 32) for b in (512, 1024) for g in (4, 8, 16, 32)]
    for rnd in range(args.rounds):
        for (sb, b, g) in variants:
            ia.lib.fmx_set_option(b"sort_min", 0 if sb == 0 else 16384)
            if sb:
                assert ia.lib.fmx_set_option(b"sort_bits", sb) == 0
            assert ia.lib.fmx_set_option(b"block", b) == 0 and ia.lib.fmx_set_option(b"groups_per_cu", g) == 0
            for timed in (False, True):  # first call of a variant may (re)allocate scratch
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                ev0.record(stream)
                for _ in range(5):
                    rc = ia.lib.fmx_count_batch_dev(fm.handle, d_pat.data_ptr(), d_off.data_ptr(), n, d_cnt.data_ptr(),
                                                    None, None, C.c_void_p(stream.cuda_stream))
                    assert rc == 0
                ev1.record(stream)
                torch.cuda.synchronize()
            results.setdefault((sb, b, g), []).append(ev0.elapsed_time(ev1) / 5)
            c = d_cnt.cpu().numpy()
            if ref is None:
                ref = c
            assert (c == ref).all()
    print("%9s %6s %4s %10s %10s" % ("sort_bits", "block", "gpc", "median_ms", "min_ms"))
    for (sb, b, g), v in sorted(results.items(), key=lambda kv: np.median(kv[1])):
        print("%9d %6d %4d %10.4f %10.4f" % (sb, b, g, np.median(v), min(v)))


if __name__ == "__main__":
    main()
