# A/B of library BUILDS on every row of the default bench line: bash tools/ab_lib_rows.sh index4j_amd/libfmx.so index4j_amd/libfmx_w6.so ...
show='import sys,json
d=json.loads(sys.stdin.read())
rows=[("headline", d["ms_per_step"])]
for s in d.get("secondary") or []:
    if "series" in s:
        rows += [(r["key"], r["ms_per_batch"]) for r in s["series"]["rows"]]
    elif s.get("ms") is not None:
        rows.append((s["config"][15:40], s["ms"]))
print(sys.argv[1], " | ".join("%s %.4f" % (k, v) for k, v in rows))'
for l in "$@"; do
FMX_LIBRARY=$PWD/$l python bench.py --cpu-budget 0.2 2>/dev/null | python tools/bench_detail.py | python -c "$show" "[$l]"
done
