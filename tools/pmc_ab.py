#!/usr/bin/env python3
"""Table of tools/pmc_ab.sh's passes: per option setting and query kernel, mean counter values per dispatch."""
import csv
import glob
import os
import sys
from collections import defaultdict

SKIP = ("k_suffix", "k_win_build", "rocprim", "k_sa_", "k_wt_", "Cijk", "at::", "elementwise", "fill")


def main():
    root = sys.argv[1]
    for d in sorted(glob.glob(os.path.join(root, "opt*"))):
        opt = open(os.path.join(d, "opt.txt")).read().strip()
        print("== %s" % opt)
        dur = defaultdict(list)
        for f in glob.glob(os.path.join(d, "trace", "**", "*kernel_trace.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        vals = defaultdict(lambda: defaultdict(list))
        for f in glob.glob(os.path.join(d, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                vals[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k in sorted(set(dur) | set(vals)):
            if any(s in k for s in SKIP):
                continue
            name = k.split("(")[0][-60:]
            us = dur.get(k, [])
            line = "  %-60s n=%d us=%.1f" % (name, len(us), sum(us) / max(1, len(us)))
            for c, v in sorted(vals.get(k, {}).items()):
                line += " %s=%.4g" % (c, sum(v) / len(v))
            print(line)


if __name__ == "__main__":
    main()
