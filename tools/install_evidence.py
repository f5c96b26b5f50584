#!/usr/bin/env python3
"""After `gpurun -- bash tools/round_evidence.sh <tag>`: turn gpurun_out/ into the committed evidence of a round.

    python3 tools/install_evidence.py <tag>

* profiles/pmc_latest.json: the headline kernels' counters (tools/summarize_prof.py --pmc-json), the FETCH_SIZE calibration of the
  same session, the per-row counters (tools/summarize_rows.py --update) — stamped with the digest of the kernel sources as they
  are NOW (bench.py refuses counters taken on other sources: do not edit the kernels between the GPU run and this);
* profiles/<tag>_*: the rocprofv3 summary and kernel statistics of the profiled default run, the rows' report, the unprofiled
  bench stdout and its contract line, the calibration, the configs[4] run."""
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag = sys.argv[1]
    go, prof = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
    pmc = os.path.join(prof, "pmc_latest.json")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "summarize_prof.py"), os.path.join(go, "prof_" + tag), "--pmc-json", pmc])
    doc = json.load(open(pmc))
    doc["calibration"] = json.load(open(os.path.join(go, tag + "_fetch_calibration.json")))
    json.dump(doc, open(pmc, "w"), indent=1)
    # the rows were summarised ON THE BOX (rows.json, summary.txt) before its largest counter CSVs were dropped from the hand-back
    rows = json.load(open(os.path.join(go, "prof_rows_" + tag, "rows.json")))
    doc = json.load(open(pmc))
    doc.update(rows)
    if doc["rows_kernel_source_sha"] != doc["kernel_source_sha"]:
        raise SystemExit("the rows' counters and the headline's were taken on different kernel sources")
    json.dump(doc, open(pmc, "w"), indent=1)
    rows_report = open(os.path.join(go, "prof_rows_" + tag, "summary.txt")).read()
    open(os.path.join(prof, tag + "_rows_counters.txt"), "w").write(rows_report)
    copies = {os.path.join(go, "prof_" + tag, "summary.txt"): tag + "_final_summary.txt",
              os.path.join(go, tag + "_bench_stdout.txt"): tag + "_bench_unprofiled_stdout.txt",
              os.path.join(go, tag + "_bench_stderr.txt"): tag + "_bench_unprofiled_stderr.txt",
              os.path.join(go, tag + "_fetch_calibration.json"): tag + "_fetch_calibration.json",
              os.path.join(go, tag + "_segments_stdout.txt"): tag + "_segments_one_gpu_stdout.txt",
              os.path.join(go, "prof_" + tag, "bench_trace.json"): tag + "_final_bench_profiled_stdout.txt"}
    for f in glob.glob(os.path.join(go, "prof_" + tag, "trace", "**", "*kernel_stats.csv"), recursive=True):
        copies[f] = tag + "_final_kernel_stats.csv"
    for src, dst in copies.items():
        if os.path.exists(src):
            shutil.copyfile(src, os.path.join(prof, dst))
        else:
            print("missing:", src)
    lines = open(os.path.join(go, tag + "_bench_stdout.txt")).read().strip().splitlines()
    open(os.path.join(prof, tag + "_bench_contract_line.json"), "w").write(lines[-1] + "\n")
    print(rows_report)
    print("installed: profiles/pmc_latest.json (kernel sources %s) and profiles/%s_*" % (doc.get("kernel_source_sha"), tag))


if __name__ == "__main__":
    main()
