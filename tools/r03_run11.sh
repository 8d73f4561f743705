#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03_run11; mkdir -p $O
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r03_run11/bench.json') if l.startswith('{')][-1])
print(d['metric'], d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], 'gap', d.get('host_gap_ms'))
print('kernels', d['kernel_avg_ms'])
for sec in ('configs','next_rows'):
    for k,v in d.get(sec,{}).items():
        if 'error' in v: print(sec,k,'ERROR',v['error']); continue
        print(sec,k,v['ms'],v['roofline']['frac'],v.get('ms_without_kernel_events'),v.get('single_push'),v['kernel_avg_ms'])
print('cpu', json.dumps(d.get('cpu_baseline'))[:1500])
print('traffic', d.get('pmc_traffic_bytes_per_launch'))
print({k:d[k] for k in d if k.startswith('take_') or k.startswith('roofline_take')})
PY
