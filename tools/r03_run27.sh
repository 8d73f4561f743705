#!/bin/bash
# valid rows counted after the scatter (tiled + sparse kernels): whole GPU suite, forced-sparse filter suites, the sweep
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03_run27; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_full.log 2>&1; grep -E "passed|failed" $O/pytest_full.log | tail -1; grep -E "^(FAILED|ERROR)" $O/pytest_full.log | head
AH_FILTER_SPARSE=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_filter_small.py tests/test_gpu_filter_expr.py tests/test_gpu_deferred.py -q -m gpu -k "filter or coalesc or record_batch or deferred" > $O/forced.log 2>&1; tail -1 $O/forced.log; grep -E "^(FAILED|ERROR)" $O/forced.log | head
bash tools/selectivity_sweep.sh > $O/selectivity_sweep.md 2> $O/selectivity_sweep.err; cat $O/selectivity_sweep.md
SEL_LIST="0.004 0.03 0.05" bash tools/selectivity_sweep.sh | tail -n +3 | tee $O/selectivity_sweep_more.md
