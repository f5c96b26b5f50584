#!/usr/bin/env python3
"""Counter passes of tools/profile_rows.sh -> per-row totals.

    python3 tools/summarize_rows.py gpurun_out/prof_rows_<tag>            # text report + <dir>/rows.json
    python3 tools/summarize_rows.py gpurun_out/prof_rows_<tag> --update   # ... and merged into profiles/pmc_latest.json `rows`

A row's figure = the SUM of a counter over every dispatch of the query kernels in the row's process (one row per process:
tools/pmc_rows.py) / the number of calls the process made.  Index construction, the suffix table's growth and torch's own
kernels are other kernels and are left out by name."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
QUERY_KERNELS = ("k_count<", "k_plan_codes", "k_plan_scatter", "k_plan_fine", "k_plan_fused", "k_walk_hist", "k_locate_walk", "k_extract<",
                 "k_extract_boundary", "k_segment_add_counts", "k_segment_append_hits", "k_segment_commit", "k_fill_offsets")
NAMES = {"FETCH_SIZE": "FETCH_SIZE_KiB", "WRITE_SIZE": "WRITE_SIZE_KiB", "TCC_HIT_sum": "TCC_HIT", "TCC_MISS_sum": "TCC_MISS"}


def short(kernel):
    m = re.search(r"(k_[a-z_0-9]+(<[^>(]*>)?)", kernel)
    return m.group(1) if m else kernel[:40]


def one_row(d):
    key = open(os.path.join(d, "row.txt")).read().strip()
    calls = queries = None
    for f in glob.glob(os.path.join(d, "*.out")):
        m = re.search(r"PMC_ROW_CALLS (\d+) QUERIES (\d+)", open(f).read())
        if m:
            calls, queries = int(m.group(1)), int(m.group(2))
    if not calls:
        return key, None
    total = defaultdict(float)
    per_kernel = defaultdict(lambda: defaultdict(float))
    launches = defaultdict(int)
    for f in sorted(glob.glob(os.path.join(d, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "")
            if not any(q in k for q in QUERY_KERNELS):
                continue
            c = NAMES.get(r["Counter_Name"], r["Counter_Name"])
            total[c] += float(r["Counter_Value"])
            per_kernel[short(k)][c] += float(r["Counter_Value"])
            if (r.get("Dispatch_Id"), k) not in seen and "pmc_fetch" in f:
                seen.add((r.get("Dispatch_Id"), k))
                launches[short(k)] += 1
    if "FETCH_SIZE_KiB" not in total or "WRITE_SIZE_KiB" not in total:
        return key, None
    row = {"queries": queries, "calls": calls}
    row.update({c: v / calls for c, v in sorted(total.items())})
    row["kernels"] = {k: dict({c: v / calls for c, v in sorted(cs.items())}, launches_per_call=launches[k] / calls)
                      for k, cs in sorted(per_kernel.items())}
    return key, row


def main():
    root = sys.argv[1]
    from summarize_prof import kernel_source_sha

    rows = {}
    for d in sorted(glob.glob(os.path.join(root, "row*"))):
        if not os.path.isdir(d):
            continue
        key, row = one_row(d)
        if row is None:
            print("row %-34s  NO DATA (see %s/*.err)" % (key, d))
            continue
        rows[key] = row
        # (scattered 16-byte loads: a miss fills a 64-byte sector, which is what FETCH_SIZE tallies — tools/calibrate_fetch.py)
        traffic = (row["FETCH_SIZE_KiB"] + row["WRITE_SIZE_KiB"]) * 1024
        print("row %-34s  calls %d  FETCH %.1f MiB + WRITE %.1f MiB = %.1f MB per call" % (
            key, row["calls"], row["FETCH_SIZE_KiB"] / 1024, row["WRITE_SIZE_KiB"] / 1024, traffic / 1e6))
        for k, cs in row["kernels"].items():
            print("      %-44s x%-5.2f %s" % (k, cs["launches_per_call"], "  ".join(
                "%s %.4g" % (c, v) for c, v in cs.items() if c != "launches_per_call")))
    doc = {"rows_kernel_source_sha": kernel_source_sha(), "rows": rows,
           "rows_source": "rocprofv3 --pmc passes of tools/pmc_rows.py, one row per process (tools/profile_rows.sh): sums over the "
                          "row's query kernels per call; FETCH_SIZE / WRITE_SIZE in KiB"}
    json.dump(doc, open(os.path.join(root, "rows.json"), "w"), indent=1)
    if "--update" in sys.argv:
        path = os.path.join(ROOT, "profiles", "pmc_latest.json")
        cur = json.load(open(path))
        cur.update(doc)
        json.dump(cur, open(path, "w"), indent=1)
        print("merged %d rows into %s (kernel sources %s)" % (len(rows), path, doc["rows_kernel_source_sha"]))


if __name__ == "__main__":
    main()
