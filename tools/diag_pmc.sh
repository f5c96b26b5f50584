#!/bin/bash
# diagnostic counter passes for k_count (each group its own pass): where do the cycles go?
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/diag_$1
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; timeout 240 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$ROOT/bench.py" --no-cpu-baseline --steps 3 --warmup 1 > "$OUT/$name.json" 2> "$OUT/$name.err"; }
run sq1 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_BUSY_CU_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU
run sq2 SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_IFETCH
# NOTE: a pass with TA_* counters (TA_BUSY_avr, TA_*_sum) aborted inside rocprofv3 (signal 6) and then sat in its
# finalizer until the outer limit killed it (25 GPU-minutes lost in round 1) — keep TA_/TCP_ stall counters out.
run grbm GRBM_GUI_ACTIVE GRBM_TA_BUSY GRBM_TC_BUSY GRBM_SPI_BUSY
cd "$ROOT"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in sorted(glob.glob(out + "/*/*/*counter_collection.csv")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_count" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print("%-40s dispatches=%d avg=%.4g" % (k, len(v), sum(v) / len(v)))
PY
find "$OUT" -name '*.csv' -size +4M -delete
