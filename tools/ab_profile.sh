#!/bin/bash
# Kernel trace + SQ counter pass of the headline step under two (or more) option settings, in one session:
#   bash tools/ab_profile.sh <tag> "count_lean=1" "count_lean=0"   -> gpurun_out/abprof_<tag>_<i>/summary.txt
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
i=0
for o in "$@"; do
  OUT=$ROOT/gpurun_out/abprof_${TAG}_$i
  mkdir -p "$OUT"
  cd /tmp && export TMPDIR=/tmp
  FMX_OPTIONS="$o" rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-secondary --profiling --steps 12 > "$OUT/bench_trace.json" 2> "$OUT/bench_trace.err"
  FMX_OPTIONS="$o" rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD --output-format csv -d "$OUT/pmc_sq" -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-secondary --profiling --steps 3 --warmup 1 > "$OUT/bench_pmc_sq.json" 2> "$OUT/bench_pmc_sq.err"
  FMX_OPTIONS="$o" rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --output-format csv -d "$OUT/pmc_sq2" -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-secondary --profiling --steps 3 --warmup 1 > "$OUT/bench_pmc_sq2.json" 2> "$OUT/bench_pmc_sq2.err"
  cd "$ROOT"
  echo "== [$o]" > "$OUT/summary.txt"
  python3 tools/summarize_prof.py "$OUT" 2>&1 | head -30 >> "$OUT/summary.txt"
  python3 - "$OUT" >> "$OUT/summary.txt" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for d in ("pmc_sq", "pmc_sq2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob(out + "/" + d + "/**/*counter_collection.csv", recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:60]
            if not ("k_count" in k or "k_plan" in k): continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (k, r["Dispatch_Id"])
            if key not in seen: seen.add(key); cnt[k] += 1
    for k, v in acc.items():
        print(d, k, "dispatches", cnt[k], " ".join("%s=%.4g" % (c, x / cnt[k]) for c, x in sorted(v.items())))
PY
  find "$OUT" -name '*.csv' -size +8M -delete
  cat "$OUT/summary.txt"
  i=$((i+1))
done
