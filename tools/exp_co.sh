for br in 1048576 16777216 268435456; do
echo "== batch rows $br"; python bench.py --workload coalesce --batch-rows $br --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], 'Mrows/s', d['ms_per_step'], 'ms/step', d['roofline']['achieved'], 'GB/s alg', d['kernel_avg_ms'])
    elif 'rror' in l or 'Trace' in l: print(l.strip())
"
done
