# A/B in one GPU session: one call = two halves on two streams (default) vs one plan + one k_count (count_split_min 0)
for i in 1 2 3; do
for v in 524288 0; do
python bench.py --no-secondary --cpu-budget 0.2 --overlap-streams 1 --count-split-min $v 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('split_min', $v, 'ms_per_step', round(d['ms_per_step'],4), 'value %.3e' % d['value'])"
done
done
