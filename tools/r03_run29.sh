#!/bin/bash
# the coalescer's scatters through the sparse kernel: parity (both kernels forced), then a 0.1 % / 1 % coalesce step A/B
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03_run29; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_filter_sparse.py tests/test_gpu_parity.py -q -m gpu -k "sparse or coalesc" > $O/pytest.log 2>&1; tail -1 $O/pytest.log; grep -E "^(FAILED|ERROR)" $O/pytest.log | head
AH_FILTER_SPARSE=1 timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "coalesc" > $O/forced.log 2>&1; tail -1 $O/forced.log; grep -E "^(FAILED|ERROR)" $O/forced.log | head
run() {  # name, sparse env, selectivity, group
  AH_FILTER_SPARSE=$2 AH_COALESCE_GROUP=$4 timeout 300 python bench.py --workload coalesce --selectivity $3 --steps 6 --warmup 2 --no-cpu-baseline --pmc-traffic off > $O/$1.json 2> $O/$1.err
  python - "$O/$1.json" "$1" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], j["ms_per_step"], j.get("kernel_avg_ms"), j.get("ms_per_step_without_kernel_events"))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
run tiled_0.001_g8 0 0.001 8
run sparse_0.001_g8 "" 0.001 8
run tiled_0.001_g1 0 0.001 1
run sparse_0.001_g1 "" 0.001 1
run tiled_0.01_g8 0 0.01 8
run sparse_0.01_g8 "" 0.01 8
