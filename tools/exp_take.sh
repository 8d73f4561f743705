for ku in 4 8; do
echo "== KU=$ku"; AH_TAKE_KU=$ku python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('take_ms', d['take_gather_ms'], 'step', d['ms_per_step'])
"
done
