# A/B of library OPTIONS on BASELINE configs[4] held by ONE GPU (8,388,608 patterns over 8 resident segment indexes):
# usage: bash tools/ab_segments.sh "segments_overlap=1" "segments_overlap=0" [reps]
show='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], "step ms", round(d["ms_per_step"],3), "patterns/s", "%.4g" % d["value"])'
A=$1; B=$2; N=${3:-1}
for i in $(seq $N); do
FMX_OPTIONS="$A" python bench.py --workload segments --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python tools/bench_detail.py | python -c "$show" "[$A]"
FMX_OPTIONS="$B" python bench.py --workload segments --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python tools/bench_detail.py | python -c "$show" "[$B]"
done
