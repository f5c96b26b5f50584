# A/B of two builds of the library in one GPU session, headline only, no oracle (results of an EXPERIMENT build may be wrong):
# usage: bash tools/ab_lib.sh index4j_amd/libfmx_exp.so [reps]
show='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], "step", round(d["ms_per_step"],4), "overlapped", round((d.get("overlapped") or {}).get("ms_per_step", 0),4))'
LIB=$1; N=${2:-3}
for i in $(seq $N); do
python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | python tools/bench_detail.py | python -c "$show" "[libfmx.so]"
FMX_LIBRARY=$PWD/$LIB python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | python tools/bench_detail.py | python -c "$show" "[$LIB]"
done
