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


KERNELS = {"k_count": "k_count", "k_locate_walk": "k_locate_walk", "k_extract_boundary_group": "k_extract_boundary_group",
           "k_plan_codes": "k_plan_codes", "k_plan_scatter": "k_plan_scatter", "k_plan_fine": "k_plan_fine",
           "k_extract": "k_extract<"}  # ("k_extract<": not k_extract_boundary*)


def kernel_source_sha():
    """the digest bench.py checks before it trusts this file (same function there)"""
    import hashlib

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for f in ("fmx_kernels.hip", "fmx_device.hpp", "fmx_blob.hpp", "fmx_blob.cpp", "fmx_plan.hpp"):
        h.update(open(os.path.join(root, "index4j_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def pmc_json(root, path):
    """per-launch averages of the kernels' counters -> the JSON bench.py reads (profiles/pmc_latest.json)"""
    import json

    per = {k: {} for k in KERNELS}
    for sub in ("pmc_fetch", "pmc_tcc", "pmc_sq", "pmc_tcp"):
        for f in find(os.path.join(root, sub), "*counter_collection.csv"):
            rows = list(csv.DictReader(open(f)))
            # the run also launches k_count / k_plan_* on the secondary configs' smaller batches: only the dispatches of
            # the headline batch (the largest grid of each kernel) count
            biggest = defaultdict(int)
            for r in rows:
                for key, needle in KERNELS.items():
                    if needle in r.get("Kernel_Name", ""):
                        biggest[key] = max(biggest[key], int(r.get("Grid_Size") or 0))
            agg = defaultdict(lambda: defaultdict(list))
            for r in rows:
                for key, needle in KERNELS.items():
                    if needle in r.get("Kernel_Name", "") and int(r.get("Grid_Size") or 0) == biggest[key]:
                        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            for key, ctrs in agg.items():
                for c, v in ctrs.items():
                    per[key][c] = sum(v) / len(v)

    def section(out):
        return {"FETCH_SIZE_KiB": out.get("FETCH_SIZE"), "WRITE_SIZE_KiB": out.get("WRITE_SIZE"),
                "TCC_HIT": out.get("TCC_HIT_sum"), "TCC_MISS": out.get("TCC_MISS_sum"),
                "TCP_TCC_READ_REQ": out.get("TCP_TCC_READ_REQ_sum"),
                "TCP_TOTAL_CACHE_ACCESSES": out.get("TCP_TOTAL_CACHE_ACCESSES_sum"),
                "SQ_INSTS_VALU": out.get("SQ_INSTS_VALU"), "SQ_INSTS_VMEM_RD": out.get("SQ_INSTS_VMEM_RD"),
                "SQ_WAVE_CYCLES": out.get("SQ_WAVE_CYCLES"), "SQ_WAIT_ANY": out.get("SQ_WAIT_ANY"),
                "SQ_ACTIVE_INST_ANY": out.get("SQ_ACTIVE_INST_ANY"), "GRBM_GUI_ACTIVE": out.get("GRBM_GUI_ACTIVE")}

    doc = {"source": "rocprofv3 --pmc passes of `bench.py --steps 3 --warmup 1` (tools/profile.sh), per-launch "
                     "averages; FETCH_SIZE / WRITE_SIZE are in KiB",
           "kernel_source_sha": kernel_source_sha(),
           "workload": {"text_log2": 28, "patterns": 1 << 20, "sample_rate": 32}}
    for key in KERNELS:
        if per[key]:
            doc[key] = section(per[key])
    json.dump(doc, open(path, "w"), indent=1)


def main():
    root = sys.argv[1]
    if len(sys.argv) > 3 and sys.argv[2] == "--pmc-json":
        pmc_json(root, sys.argv[3])
        return
    print("# rocprofv3 summary of", root)
    for f in find(os.path.join(root, "trace"), "*kernel_stats.csv"):
        print("\n## kernel stats (%s)" % os.path.relpath(f, root))
        rows = list(csv.DictReader(open(f)))
        for r in [r for i, r in enumerate(rows) if i < 8 or 'fmx::' in r.get('Name', '')]:
            print("  %-60s calls=%s total_ns=%s avg_ns=%s min_ns=%s max_ns=%s pct=%s" % (
                r.get("Name", "")[:60], r.get("Calls"), r.get("TotalDurationNs"), r.get("AverageNs"),
                r.get("MinNs"), r.get("MaxNs"), r.get("Percentage")))
    for f in find(os.path.join(root, "trace"), "*kernel_trace.csv"):
        rows = list(csv.DictReader(open(f)))
        agg = defaultdict(list)
        by_grid = defaultdict(lambda: defaultdict(list))  # kernel -> grid size -> durations
        meta = {}
        for r in rows:
            name = r.get("Kernel_Name", "")
            agg[name].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            by_grid[name][int(r.get("Grid_Size") or r.get("Grid_Size_X") or 0)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            meta[name] = {k: r.get(k) for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size",
                                                 "Workgroup_Size", "Grid_Size")}
        print("\n## kernel trace (%s)" % os.path.relpath(f, root))
        for name, d in [kv for i, kv in enumerate(sorted(agg.items(), key=lambda kv: -sum(kv[1]))) if i < 6 or 'fmx::' in kv[0]]:
            d2 = sorted(d)
            print("  %-50s n=%d avg_us=%.1f med_us=%.1f min_us=%.1f max_us=%.1f %s" % (
                name[:50], len(d), sum(d) / len(d) / 1e3, d2[len(d2) // 2] / 1e3, d2[0] / 1e3, d2[-1] / 1e3, meta[name]))
            if "fmx" in name and len(by_grid[name]) > 1:
                # the same kernel serves batches of several sizes in one bench run (headline, configs[2] / [3], table growth):
                # the HEADLINE launches are those of its largest grid — the average bench.py's HIP events measure
                g = max(by_grid[name])
                dg = sorted(by_grid[name][g])
                print("  %-50s   largest grid (%d threads) only: n=%d avg_us=%.1f med_us=%.1f min_us=%.1f max_us=%.1f" % (
                    "", g, len(dg), sum(dg) / len(dg) / 1e3, dg[len(dg) // 2] / 1e3, dg[0] / 1e3, dg[-1] / 1e3))
    for sub in ("pmc_fetch", "pmc_tcc", "pmc_sq", "pmc_tcp"):
        for f in find(os.path.join(root, sub), "*counter_collection.csv"):
            rows = list(csv.DictReader(open(f)))
            biggest = defaultdict(int)
            for r in rows:
                biggest[r.get("Kernel_Name", "")] = max(biggest[r.get("Kernel_Name", "")], int(r.get("Grid_Size") or 0))
            agg = defaultdict(lambda: defaultdict(list))
            for r in rows:  # per kernel: the dispatches with its largest grid (the headline batch, not the secondary legs')
                if int(r.get("Grid_Size") or 0) == biggest[r.get("Kernel_Name", "")]:
                    agg[r.get("Kernel_Name", "")][r.get("Counter_Name", "")].append(float(r.get("Counter_Value", 0)))
            print("\n## counters (%s)" % os.path.relpath(f, root))
            for name, ctrs in agg.items():
                if "k_" not in name:
                    continue
                for c, v in sorted(ctrs.items()):
                    print("  %-40s %-32s dispatches=%d avg=%.4g min=%.4g max=%.4g" % (name[:40], c, len(v), sum(v) / len(v), min(v), max(v)))


if __name__ == "__main__":
    main()
