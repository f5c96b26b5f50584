# A/B of two builds of the library in one GPU session: index4j_amd/libfmx.so vs index4j_amd/libfmx_prev.so
# (build the other revision, copy its .so to libfmx_prev.so, then: gpurun -- bash tools/ab_prev.sh)
for i in 1 2; do
python bench.py --no-cpu-baseline 2>/dev/null | python tools/bench_detail.py | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new  step', round(d['ms_per_step'],4))"
FMX_LIBRARY=$PWD/index4j_amd/libfmx_prev.so python bench.py --no-cpu-baseline 2>/dev/null | python tools/bench_detail.py | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('prev step', round(d['ms_per_step'],4))"
done
python tools/bench_configs.py --out gpurun_out/cfg_new.jsonl > /dev/null 2>&1
FMX_LIBRARY=$PWD/index4j_amd/libfmx_prev.so python tools/bench_configs.py --out gpurun_out/cfg_prev.jsonl > /dev/null 2>&1
python - <<'PY'
import json
for w in ("new","prev"):
    for l in open('gpurun_out/cfg_%s.jsonl'%w):
        r=json.loads(l); print(w, r['config'][:40], round(r['ms'],4))
PY
