#!/bin/bash
# sparse string filter: parity (both kernels forced) + string suites with sparse forced + string_filter at 0.1 % A/B
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03_run31; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_filter_sparse.py -q -m gpu > $O/sparse.log 2>&1; tail -1 $O/sparse.log; grep -E "^(FAILED|ERROR)" $O/sparse.log | head -5
AH_FILTER_SPARSE=1 timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "string or utf8 or byte" > $O/forced.log 2>&1; tail -1 $O/forced.log; grep -E "^(FAILED|ERROR)" $O/forced.log | head -5
for sp in 0 ""; do
  AH_FILTER_SPARSE=$sp timeout 200 python bench.py --workload string_filter --selectivity 0.001 --steps 10 --warmup 3 --no-cpu-baseline --pmc-traffic off > $O/sf_$sp.json 2>/dev/null
  python -c "
import json
j=json.loads(open('$O/sf_$sp.json').read().strip().splitlines()[-1]); print('sparse=$sp', j['ms_per_step'], j.get('kernel_avg_ms'))"
done
