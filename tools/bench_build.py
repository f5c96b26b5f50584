#!/usr/bin/env python3
"""Index construction: fmx_build_on_device with the wavelet tree encoded in HBM, the same with the host's wavelet
encoder behind the device suffix-array stage, and (small sizes / --host) the all-host builder (SA-IS on one thread +
parallel wavelet encode).  Checks that all serialize to the same bytes.  FMX_BUILD_TIMING=1 prints the phases.
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
    ap.add_argument("--repeat", type=int, default=2, help="device builds per variant (the fastest counts)")
    ap.add_argument("--host", action="store_true", help="also time the all-host builder at every size (27 s at 2^28)")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "build.jsonl"))
    args = ap.parse_args()
    import index4j_amd as ia

    ia.FmIndex("warm up the device", 4, True, device=None, build_device=0)
    rows = []
    for lg in args.text_log2:
        t = ia.synth_log(1 << lg)
        runs = {}
        for name, on_device in (("device_build_s", 1), ("device_sa_host_wavelet_s", 0)):
            ia.lib.fmx_set_option(b"wavelet_on_device", on_device)
            best = None
            for _ in range(args.repeat):
                t0 = time.perf_counter()
                dev = ia.FmIndex(t, args.sample_rate, True, device=None, build_device=0)
                dt = time.perf_counter() - t0
                if best is None or dt < best[0]:
                    best = (dt, dict(dev.build_stats))
                d_bytes = dev.write(False)
                del dev
            runs[name] = (best, hashlib.sha256(d_bytes).hexdigest(), len(d_bytes))
            del d_bytes
        ia.lib.fmx_set_option(b"wavelet_on_device", 1)
        (t_dev, stats), d_hash, n_bytes = runs["device_build_s"]
        (t_mixed, _), m_hash, _ = runs["device_sa_host_wavelet_s"]
        row = {"text_chars": 1 << lg, "sample_rate": args.sample_rate, "device_build_s": t_dev,
               "device_sa_host_wavelet_s": t_mixed, "device_stage_s": stats["device_stage_seconds"],
               "wavelet_device_s": stats["wavelet_device_seconds"], "doubling_rounds": stats["doubling_rounds"],
               "rows_sorted": stats["rows_sorted"], "serialized_bytes": n_bytes, "identical": d_hash == m_hash,
               "host_cores": os.cpu_count(), "chars_per_s_device_build": (1 << lg) / t_dev}
        if args.host or lg <= 24:
            t0 = time.perf_counter()
            host = ia.FmIndex(t, args.sample_rate, True, device=None)
            t_host = time.perf_counter() - t0
            row["host_build_s"] = t_host
            row["chars_per_s_host_build"] = (1 << lg) / t_host
            row["identical"] = row["identical"] and hashlib.sha256(host.write(False)).hexdigest() == d_hash
            del host
        print(json.dumps(row), flush=True)
        assert row["identical"]
        rows.append(row)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        for r in rows:
            f.write(json.dumps(r) + "\n")


if __name__ == "__main__":
    main()
