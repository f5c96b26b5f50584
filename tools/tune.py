#!/usr/bin/env python3
"""A/B timing of launch tunables and input orderings for k_count, interleaved in ONE process
(GPU box only).  usage: python tools/tune.py [--text-log2 28] [--rounds 5]"""
import argparse
import ctypes as C
import os
import sys
import time

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
    P = pat.reshape(n, 8)
    # lexicographic order of the REVERSED patterns: neighbours share their last characters, i.e. the
    # first backward-search steps
    key = np.zeros(n, dtype=np.uint64)
    for j in range(8):
        key = (key << np.uint64(8)) | P[:, 7 - j].astype(np.uint64)
    order = np.argsort(key, kind="stable")
    inputs = {"random": pat, "sorted_by_suffix": np.ascontiguousarray(P[order]).reshape(-1)}
    d_off = torch.from_numpy(off).to(dev)
    d_cnt = torch.zeros(n, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream()
    results = {}
    base_counts = None
    variants = [(b, g) for b in (512, 1024) for g in (2, 4, 8, 16, 32)]
    for name, arr in inputs.items():
        d_pat = torch.from_numpy(arr.view(np.int16)).to(dev)
        for rnd in range(args.rounds):
            for (b, g) in variants:
                assert ia.lib.fmx_set_option(b"block", b) == 0 and ia.lib.fmx_set_option(b"groups_per_cu", g) == 0
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                ev0.record(stream)
                for _ in range(5):
                    rc = ia.lib.fmx_count_batch_dev(fm.handle, d_pat.data_ptr(), d_off.data_ptr(), n, d_cnt.data_ptr(),
                                                    None, None, C.c_void_p(stream.cuda_stream))
                    assert rc == 0
                ev1.record(stream)
                torch.cuda.synchronize()
                results.setdefault((name, b, g), []).append(ev0.elapsed_time(ev1) / 5)
        c = d_cnt.cpu().numpy()
        if name == "random":
            base_counts = c
        else:
            assert (c == base_counts[order]).all(), "sorted batch gives different counts"
    print("%-18s %6s %4s %10s %10s" % ("input", "block", "gpc", "median_ms", "min_ms"))
    for (name, b, g), v in sorted(results.items(), key=lambda kv: np.median(kv[1])):
        print("%-18s %6d %4d %10.4f %10.4f" % (name, b, g, np.median(v), min(v)))


if __name__ == "__main__":
    main()
