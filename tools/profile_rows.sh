#!/bin/bash
# Run on the GPU box (through gpurun): counter passes of every secondary row of bench.py's line, ONE ROW PER PROCESS
# (tools/pmc_rows.py), each counter group in a pass of its own, then tools/summarize_rows.py -> gpurun_out/prof_rows_<tag>/rows.json
# (merge into profiles/pmc_latest.json with `python3 tools/summarize_rows.py <dir> --update`).
# usage: tools/profile_rows.sh <tag> [--sq]    (--sq: also the SQ instruction counters, for every row)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
OUT=$ROOT/gpurun_out/prof_rows_$TAG
mkdir -p "$OUT"
python3 "$ROOT/tools/pmc_rows.py" --row "configs[3]" --prepare > "$OUT/prepare.log" 2>&1 || { echo "prepare failed"; tail -5 "$OUT/prepare.log"; exit 1; }
cd /tmp && export TMPDIR=/tmp
python3 "$ROOT/tools/pmc_rows.py" --list > "$OUT/rows.txt"
i=0
while IFS= read -r row; do
  i=$((i+1))
  d="$OUT/row$i"
  mkdir -p "$d"
  echo "$row" > "$d/row.txt"
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$d/pmc_fetch" -- python3 "$ROOT/tools/pmc_rows.py" --row "$row" > "$d/fetch.out" 2> "$d/fetch.err" || echo "row $row: FETCH pass failed"
  timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$d/pmc_tcc" -- python3 "$ROOT/tools/pmc_rows.py" --row "$row" > "$d/tcc.out" 2> "$d/tcc.err" || echo "row $row: WRITE pass failed"
  if [ "$1" = "--sq" ] || [ "$row" = "configs[3]" ]; then
    timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD --output-format csv -d "$d/pmc_sq" -- python3 "$ROOT/tools/pmc_rows.py" --row "$row" > "$d/sq.out" 2> "$d/sq.err" || echo "row $row: SQ pass failed"
    timeout 300 rocprofv3 --pmc SQ_IFETCH SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD --output-format csv -d "$d/pmc_sq2" -- python3 "$ROOT/tools/pmc_rows.py" --row "$row" > "$d/sq2.out" 2> "$d/sq2.err" || echo "row $row: SQ2 pass failed"
  fi
  echo "row $i done: $row"
done < "$OUT/rows.txt"
cd "$ROOT"
python3 tools/summarize_rows.py "$OUT" > "$OUT/summary.txt" 2>&1
find "$OUT" -name '*.csv' -size +16M -delete
cat "$OUT/summary.txt"
