# A/B of two builds in one GPU session on the whole default bench line (headline + secondary configs[2]/[3]):
# index4j_amd/libfmx.so vs index4j_amd/libfmx_prev.so
show='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], "step", round(d["ms_per_step"],4), "k_count", round(d["roofline"]["kernel_ms"],4), " ".join("%s %.4f" % (s["config"][15:27], s["ms"]) for s in d.get("secondary", []) if "ms" in s))'
for i in 1 2; do
python bench.py --cpu-budget 0.2 2>/dev/null | python tools/bench_detail.py | python -c "$show" new
FMX_LIBRARY=$PWD/index4j_amd/libfmx_prev.so python bench.py --cpu-budget 0.2 2>/dev/null | python tools/bench_detail.py | python -c "$show" prev
done
