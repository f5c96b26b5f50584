# A/B of library OPTIONS in one GPU session on the default bench line's headline (no secondary rows):
# usage: bash tools/ab_options.sh "plan_fused=1" "plan_fused=0" [reps]
show='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], "step", round(d["ms_per_step"],4), "k_count", round(r["kernel_ms"],4), "overlapped", round((d.get("overlapped") or {}).get("ms_per_step", 0),4))'
A=$1; B=$2; N=${3:-2}
for i in $(seq $N); do
FMX_OPTIONS="$A" python bench.py --cpu-budget 0.2 --no-secondary 2>/dev/null | python tools/bench_detail.py | python -c "$show" "[$A]"
FMX_OPTIONS="$B" python bench.py --cpu-budget 0.2 --no-secondary 2>/dev/null | python tools/bench_detail.py | python -c "$show" "[$B]"
done
