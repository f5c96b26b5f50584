# A/B of library OPTIONS on EVERY row of the default bench line (headline + secondary rows):
# usage: bash tools/ab_options_rows.sh "groups_per_cu=8" "groups_per_cu=16"
show='import sys,json
d=json.loads(sys.stdin.read())
rows=[("headline", d["ms_per_step"])]
for s in d.get("secondary") or []:
    if "series" in s:
        rows += [(r["key"], r["ms_per_batch"]) for r in s["series"]["rows"]]
    elif s.get("ms") is not None:
        rows.append((s["config"][15:40], s["ms"]))
print(sys.argv[1], " | ".join("%s %.4f" % (k, v) for k, v in rows))'
for o in "$@"; do
FMX_OPTIONS="$o" python bench.py --cpu-budget 0.2 2>/dev/null | python tools/bench_detail.py | python -c "$show" "[$o]"
done
