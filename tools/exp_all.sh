for wl in arith cmp cast cast_string; do
echo "== $wl"; python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['metric'], d['value'], 'Mrows/s', 'ms/step', d['ms_per_step'], 'roofline', d['roofline']['achieved'], d['roofline']['frac'], 'kernel ms', d['roofline']['avg_launch_ms'])
    elif 'rror' in l: print(l.strip())
"
done
exit 0
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('take_ms', d['take_gather_ms'], 'scatter', d['filter_scatter_ms'])
"
