#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel trace + separate PMC passes of bench.py.
# usage: tools/profile.sh <tag> [bench.py args...]   -> gpurun_out/prof_<tag>/
# (the default bench run: headline count() + the secondary configs[2] locate / configs[3] extractUntilBoundary legs, so
# the trace and the counters cover k_count, k_plan_*, k_locate_walk, k_extract_boundary_group and the k_sa_* builders)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# 1) per-kernel time (the program itself after --, no launcher hops)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" --cpu-budget 0.2 --profiling "$@" > "$OUT/bench_trace.json" 2> "$OUT/bench_trace.err"
# 2) counters, each group in its own pass (gfx950: TCC has 4 slots, FETCH_SIZE takes 3)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$ROOT/bench.py" --cpu-budget 0.2 --profiling --steps 3 --warmup 1 "$@" > "$OUT/bench_pmc_fetch.json" 2> "$OUT/bench_pmc_fetch.err"
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pmc_tcc" -- python3 "$ROOT/bench.py" --cpu-budget 0.2 --profiling --steps 3 --warmup 1 "$@" > "$OUT/bench_pmc_tcc.json" 2> "$OUT/bench_pmc_tcc.err"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD --output-format csv -d "$OUT/pmc_sq" -- python3 "$ROOT/bench.py" --cpu-budget 0.2 --profiling --steps 3 --warmup 1 "$@" > "$OUT/bench_pmc_sq.json" 2> "$OUT/bench_pmc_sq.err"
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_tcp" -- python3 "$ROOT/bench.py" --cpu-budget 0.2 --profiling --steps 3 --warmup 1 "$@" > "$OUT/bench_pmc_tcp.json" 2> "$OUT/bench_pmc_tcp.err"
cd "$ROOT"
python3 tools/summarize_prof.py "$OUT" > "$OUT/summary.txt" 2>&1
python3 tools/summarize_prof.py "$OUT" --pmc-json "$OUT/pmc.json" > /dev/null 2>&1
find "$OUT" -name '*.csv' -size +8M -delete
ls -R "$OUT" | head -60
cat "$OUT/summary.txt"
