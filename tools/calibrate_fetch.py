#!/usr/bin/env python3
"""Calibrates rocprofv3's FETCH_SIZE for SCATTERED 16-byte loads (k_count's access pattern) on this GPU and stores the
factor in profiles/pmc_latest.json ("calibration"), so that bench.py's `traffic` / `traffic_frac` are absolute bytes
rather than raw counter values (the guide calibrates the counter for wide streaming reads only: 1/2 of the bytes).

Runs tools/microbench/fetch_calibration.hip (loads from DISTINCT 128-byte lines of a 2 GiB table, each touched once) under
`rocprofv3 --pmc FETCH_SIZE` and `--pmc TCC_MISS_sum TCC_EA0_RDREQ_sum`, plus the streaming case as the control.
Run on the GPU box:  python tools/calibrate_fetch.py [--update profiles/pmc_latest.json]"""
import argparse
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_counter(exe, args, counters, outdir):
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = ["rocprofv3", "--pmc"] + counters + ["--output-format", "csv", "-d", outdir, "--", exe] + args
    r = subprocess.run(cmd, capture_output=True, text=True, cwd="/tmp", env=env, timeout=600)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    vals = {}
    for f in glob.glob(os.path.join(outdir, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "k_calib" in row.get("Kernel_Name", ""):
                vals[row["Counter_Name"]] = vals.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    return (json.loads(line[-1]) if line else None), vals, r.stderr[-500:]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--table-mib", type=int, default=2048)
    ap.add_argument("--loads", type=int, default=1 << 24)
    ap.add_argument("--update", default=None, help="pmc json to store the calibration in")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "fetch_calibration.json"))
    args = ap.parse_args()
    work = os.path.join(ROOT, "gpurun_out", "calib")
    os.makedirs(work, exist_ok=True)
    exe = os.path.join(work, "fetch_calibration")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950",
                           os.path.join(ROOT, "tools", "microbench", "fetch_calibration.hip"), "-o", exe])
    doc = {"what": "rocprofv3 FETCH_SIZE (KiB) against known bytes, gfx950: scattered = 16-byte loads from DISTINCT 128-byte "
                   "lines of a %d MiB table (each line once), stream = the same number of 16-byte loads, consecutive" % args.table_mib,
           "cases": {}}
    for name, extra in (("scatter", []), ("stream", ["1"]), ("halves", ["2"])):
        a = [str(args.table_mib), str(args.loads)] + extra
        info, v1, err1 = run_counter(exe, a, ["FETCH_SIZE"], os.path.join(work, name + "_fetch"))
        _, v2, err2 = run_counter(exe, a, ["TCC_MISS_sum", "TCC_EA0_RDREQ_sum"], os.path.join(work, name + "_tcc"))
        if not info or "FETCH_SIZE" not in v1:
            print("calibration pass failed:", err1, err2, file=sys.stderr)
            return 1
        fetch_bytes = v1["FETCH_SIZE"] * 1024.0
        case = dict(info, FETCH_SIZE_bytes=fetch_bytes, TCC_MISS=v2.get("TCC_MISS_sum"), TCC_EA0_RDREQ=v2.get("TCC_EA0_RDREQ_sum"),
                    FETCH_bytes_per_load=fetch_bytes / info["loads"])
        doc["cases"][name] = case
    sc, stc = doc["cases"]["scatter"], doc["cases"]["stream"]
    hv = doc["cases"]["halves"]
    # does a scattered miss fill the whole 128-byte line, or a 64-byte sector?  `halves` touches both halves of each line it
    # visits: ~1 request per line = whole lines (the x 2 below holds for scattered loads too), ~2 = sectors (x 1)
    doc["requests_per_line_when_both_halves_are_read"] = hv["TCC_EA0_RDREQ"] / hv["loads"] if hv.get("TCC_EA0_RDREQ") else None
    doc["scattered_16B_load"] = {"FETCH_SIZE_bytes_per_load": sc["FETCH_bytes_per_load"]}
    doc["streaming_16B_load"] = {"FETCH_SIZE_bytes_per_load": stc["FETCH_bytes_per_load"],
                                 "fraction_of_bytes_loaded": stc["FETCH_SIZE_bytes"] / stc["bytes_loaded"]}
    # What one fabric read request carries.  The STREAMING case loads every byte of the lines it touches:
    # bytes_loaded / TCC_EA0_RDREQ is its request size (128 on gfx950, tallied by FETCH_SIZE at 64: the guide's "double it").
    # A SCATTERED 16-byte load that misses costs one request too — of what size?  The `halves` case reads both 64-byte halves
    # of every line it visits, the second load depending on the first: ~1 request per line would mean a miss fills the whole
    # 128-byte line, ~2 mean it fills a 64-byte SECTOR — then FETCH_SIZE's 64 bytes per request are exactly what a scattered
    # miss moves, and the factor for scattered loads is 1, not 2.  (Rounds 3 and 4 assumed whole lines without this case and
    # doubled the LF kernels' traffic.)
    req_bytes = stc["bytes_loaded"] / stc["TCC_EA0_RDREQ"] if stc.get("TCC_EA0_RDREQ") else None
    doc["fabric_read_request_bytes_streaming"] = req_bytes
    doc["fabric_read_request_bytes"] = req_bytes
    doc["requests_per_scattered_load"] = sc["TCC_EA0_RDREQ"] / sc["loads"] if sc.get("TCC_EA0_RDREQ") else None
    per_line = doc.get("requests_per_line_when_both_halves_are_read")
    sectors = per_line is not None and per_line > 1.5
    doc["scattered_miss_fills"] = "a 64-byte sector" if sectors else "the whole 128-byte line"
    doc["fabric_bytes_per_FETCH_SIZE_byte_streaming"] = req_bytes / 64.0 if req_bytes else None
    doc["fabric_bytes_per_FETCH_SIZE_byte"] = 1.0 if sectors else (req_bytes / 64.0 if req_bytes else None)
    doc["reading"] = ("FETCH_SIZE = TCC_EA0_RDREQ x 64 B.  A streamed read issues %.0f-byte requests (FETCH_SIZE x %.2f = bytes moved); a "
                      "scattered 16-byte load that misses issues ONE request and — both halves of a line read one after the other cost "
                      "%.2f requests per line — fills %s: for the LF kernels' scattered loads bytes moved = FETCH_SIZE x %.2f" % (
                          req_bytes or 0, (req_bytes or 0) / 64.0, per_line or 0, doc["scattered_miss_fills"],
                          doc["fabric_bytes_per_FETCH_SIZE_byte"] or 0))
    # the ceiling of the LF kernels' access pattern: random 16-byte loads per second (each a sector fill)
    rl = os.path.join(work, "random_lines")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950",
                           os.path.join(ROOT, "tools", "microbench", "random_lines.hip"), "-o", rl])
    r = subprocess.run([rl], capture_output=True, text=True, timeout=300)
    os.remove(rl)
    rates = {}
    for ln in r.stdout.splitlines():
        parts = ln.split()
        if ln.startswith("table") and "chained" in ln:
            rates["%s_MiB" % parts[1]] = float(ln.split(":")[1].split()[0])
    doc["random_line_rate_Glines_per_s"] = dict(rates, what="random 16-byte loads per second, chained (one in flight per lane), 8 waves "
                                                            "per SIMD, by table size (tools/microbench/random_lines.hip): what the "
                                                            "memory system delivers to the LF kernels' access pattern")
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(doc, open(args.out, "w"), indent=1)
    print(json.dumps(doc, indent=1))
    if args.update and os.path.exists(args.update):
        p = json.load(open(args.update))
        p["calibration"] = doc
        json.dump(p, open(args.update, "w"), indent=1)
    os.remove(exe)
    return 0


if __name__ == "__main__":
    sys.exit(main())
