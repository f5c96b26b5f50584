#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel trace stats + PMC passes) into a short text report.
usage: tools/summarize_prof.py gpurun_out/prof_<tag>"""
import csv
import glob
import os
import sys
from collections import defaultdict


def find(root, pattern):
    return sorted(glob.glob(os.path.join(root, "**", pattern), recursive=True))


def main():
    root = sys.argv[1]
    print("# rocprofv3 summary of", root)
    for f in find(os.path.join(root, "trace"), "*kernel_stats.csv"):
        print("\n## kernel stats (%s)" % os.path.relpath(f, root))
        rows = list(csv.DictReader(open(f)))
        for r in rows[:12]:
            print("  %-60s calls=%s total_ns=%s avg_ns=%s min_ns=%s max_ns=%s pct=%s" % (
                r.get("Name", "")[:60], r.get("Calls"), r.get("TotalDurationNs"), r.get("AverageNs"),
                r.get("MinNs"), r.get("MaxNs"), r.get("Percentage")))
    for f in find(os.path.join(root, "trace"), "*kernel_trace.csv"):
        rows = list(csv.DictReader(open(f)))
        agg = defaultdict(list)
        meta = {}
        for r in rows:
            name = r.get("Kernel_Name", "")
            agg[name].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            meta[name] = {k: r.get(k) for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size",
                                                 "Workgroup_Size", "Grid_Size")}
        print("\n## kernel trace (%s)" % os.path.relpath(f, root))
        for name, d in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:8]:
            d2 = sorted(d)
            print("  %-50s n=%d avg_us=%.1f med_us=%.1f min_us=%.1f max_us=%.1f %s" % (
                name[:50], len(d), sum(d) / len(d) / 1e3, d2[len(d2) // 2] / 1e3, d2[0] / 1e3, d2[-1] / 1e3, meta[name]))
    for sub in ("pmc_fetch", "pmc_tcc", "pmc_sq", "pmc_tcp"):
        for f in find(os.path.join(root, sub), "*counter_collection.csv"):
            rows = list(csv.DictReader(open(f)))
            agg = defaultdict(lambda: defaultdict(list))
            for r in rows:
                agg[r.get("Kernel_Name", "")][r.get("Counter_Name", "")].append(float(r.get("Counter_Value", 0)))
            print("\n## counters (%s)" % os.path.relpath(f, root))
            for name, ctrs in agg.items():
                if "k_" not in name:
                    continue
                for c, v in sorted(ctrs.items()):
                    print("  %-40s %-32s dispatches=%d avg=%.4g min=%.4g max=%.4g" % (name[:40], c, len(v), sum(v) / len(v), min(v), max(v)))


if __name__ == "__main__":
    main()
