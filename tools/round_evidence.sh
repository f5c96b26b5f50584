#!/bin/bash
# Run on the GPU box (through gpurun): the evidence of a round in one go -> gpurun_out/<tag>_*
#   1. the default bench line (N = 1) unprofiled              <tag>_bench_stdout.txt / _stderr.txt
#   2. rocprofv3 kernel trace + the headline's counter passes  prof_<tag>/  (tools/profile.sh)
#   3. the secondary rows' counter passes, one row per process prof_rows_<tag>/  (tools/profile_rows.sh)
#   4. FETCH_SIZE calibration (streamed / scattered / both halves of a line)  <tag>_fetch_calibration.json
#   5. configs[4] held by one GPU (8,388,608 patterns)         <tag>_segments_stdout.txt
# then, on the build machine: copy the summaries into profiles/, `python3 tools/summarize_prof.py gpurun_out/prof_<tag> --pmc-json
# profiles/pmc_latest.json`, `python3 tools/summarize_rows.py gpurun_out/prof_rows_<tag> --update`, merge the calibration.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1
cd "$ROOT"
mkdir -p gpurun_out
python bench.py > gpurun_out/${TAG}_bench_stdout.txt 2> gpurun_out/${TAG}_bench_stderr.txt; echo "bench rc=$?"
tail -c 3900 gpurun_out/${TAG}_bench_stdout.txt | tail -1 | cut -c1-400
bash tools/profile.sh $TAG > gpurun_out/${TAG}_profile.log 2>&1; echo "profile rc=$?"
bash tools/profile_rows.sh $TAG > gpurun_out/${TAG}_profile_rows.log 2>&1; echo "profile_rows rc=$?"
grep "^row " gpurun_out/${TAG}_profile_rows.log
python tools/calibrate_fetch.py --out gpurun_out/${TAG}_fetch_calibration.json > gpurun_out/${TAG}_calib.log 2>&1; echo "calibration rc=$?"
python bench.py --workload segments --steps 5 --warmup 2 > gpurun_out/${TAG}_segments_stdout.txt 2> gpurun_out/${TAG}_segments_stderr.txt; echo "segments rc=$?"
tail -1 gpurun_out/${TAG}_segments_stdout.txt | cut -c1-300
