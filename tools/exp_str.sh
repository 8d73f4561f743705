timeout 600 python -m pytest tests -m gpu -q -k "utf8 or cast" 2>&1 | grep -E "passed|failed|Error" | tail -3
python bench.py --workload cast_string --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['metric'], d['value'], 'ms/step', d['ms_per_step'], d['kernel_avg_ms'])
"
