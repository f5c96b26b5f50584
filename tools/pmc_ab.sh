#!/bin/bash
# Counter A/B of ONE bench row under two option settings (GPU box, through gpurun):
#   tools/pmc_ab.sh "series extract(32) s=32" "window_cells=0" "window_cells=1"
# per setting: L2 requests (TCC hit + miss = what the L1s send down), fabric fetches, SQ issue / wait counters, per dispatch of every
# query kernel (tools/pmc_ab.py prints the table).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
ROW=$1; shift
OUT=$ROOT/gpurun_out/pmc_ab
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for o in "$@"; do
  i=$((i+1))
  d="$OUT/opt$i"; mkdir -p "$d"; echo "$o" > "$d/opt.txt"
  export FMX_OPTIONS="$o"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$d/trace" -- python3 "$ROOT/tools/pmc_rows.py" --row "$ROW" > "$d/trace.out" 2> "$d/trace.err" || echo "trace pass failed ($o)"
  timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum FETCH_SIZE --output-format csv -d "$d/pmc_tcc" -- python3 "$ROOT/tools/pmc_rows.py" --row "$ROW" > "$d/tcc.out" 2> "$d/tcc.err" || echo "tcc pass failed ($o)"
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU --output-format csv -d "$d/pmc_sq" -- python3 "$ROOT/tools/pmc_rows.py" --row "$ROW" > "$d/sq.out" 2> "$d/sq.err" || echo "sq pass failed ($o)"
done
cd "$ROOT"
python3 tools/pmc_ab.py "$OUT" | tee "$OUT/summary.txt"
find "$OUT" -name '*.csv' -size +8M -delete
