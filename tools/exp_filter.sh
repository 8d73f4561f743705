for skip in 0 1; do
echo "== SKIP=$skip"; AH_FILTER_SKIP=$skip python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('scatter_ms', d['filter_scatter_ms'], 'frac', d['roofline']['frac'], 'step', d['ms_per_step'])
"
done
for sel in 0.01 0.5 0.9 ; do echo "== sel=$sel"; python bench.py --steps 10 --warmup 3 --no-cpu-baseline --selectivity $sel 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('scatter_ms', d['filter_scatter_ms'], 'frac', d['roofline']['frac'], 'take', d['take_gather_ms'])
"
done
