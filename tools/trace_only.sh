#!/bin/bash
# kernel trace only (no PMC passes): per-kernel durations of bench.py -> gpurun_out/trace_<tag>/summary.txt
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
OUT=$ROOT/gpurun_out/trace_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" --no-cpu-baseline "$@" > "$OUT/bench_trace.json" 2> "$OUT/bench_trace.err"
cd "$ROOT"
python3 tools/summarize_prof.py "$OUT" 2>&1 | head -14 > "$OUT/summary.txt"
find "$OUT" -name '*.csv' -size +8M -delete
cat "$OUT/summary.txt"
