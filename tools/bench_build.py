#!/usr/bin/env python3
"""Index construction: host builder (SA-IS, one thread, + parallel wavelet encode) vs the builder whose
suffix-array stage runs on the GPU (fmx_build_on_device).  Checks that both serialize to the same bytes.
usage: python tools/bench_build.py [--text-log2 28] [--sample-rate 32] [--out gpurun_out/build.jsonl]"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--text-log2", type=int, nargs="+", default=[24, 28])
    ap.add_argument("--sample-rate", type=int, default=32)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "build.jsonl"))
    args = ap.parse_args()
    import index4j_amd as ia

    ia.FmIndex("warm up the device", 4, True, device=None, build_device=0)
    rows = []
    for lg in args.text_log2:
        t = ia.synth_log(1 << lg)
        t0 = time.perf_counter()
        dev = ia.FmIndex(t, args.sample_rate, True, device=None, build_device=0)
        t_dev = time.perf_counter() - t0
        stats = dev.build_stats
        d_bytes = dev.write(False)
        d_hash = hashlib.sha256(d_bytes).hexdigest()
        n_bytes = len(d_bytes)
        del dev, d_bytes
        t0 = time.perf_counter()
        host = ia.FmIndex(t, args.sample_rate, True, device=None)
        t_host = time.perf_counter() - t0
        h_hash = hashlib.sha256(host.write(False)).hexdigest()
        del host
        row = {"text_chars": 1 << lg, "sample_rate": args.sample_rate, "host_build_s": t_host, "device_build_s": t_dev,
               "device_stage_s": stats["device_stage_seconds"], "doubling_rounds": stats["doubling_rounds"],
               "rows_sorted": stats["rows_sorted"], "serialized_bytes": n_bytes, "identical": d_hash == h_hash,
               "host_cores": os.cpu_count(), "chars_per_s_device_build": (1 << lg) / t_dev,
               "chars_per_s_host_build": (1 << lg) / t_host}
        print(json.dumps(row), flush=True)
        assert row["identical"]
        rows.append(row)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        for r in rows:
            f.write(json.dumps(r) + "\n")


if __name__ == "__main__":
    main()
