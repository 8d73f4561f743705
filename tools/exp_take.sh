for nt in 0 1; do
echo "== NT=$nt"; AH_TAKE_NT=$nt python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('take_ms', d['take_gather_ms'], 'sorted', d['take_sorted_indices_ms'], 'step', d['ms_per_step'])
"
done
