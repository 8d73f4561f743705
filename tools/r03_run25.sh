#!/bin/bash
# the sparse scatter kernel: its own tests, the filter / coalescer / record-batch suites with it FORCED everywhere, the sweep
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03_run25; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_filter_sparse.py -q -m gpu > $O/sparse.log 2>&1; tail -3 $O/sparse.log; grep -E "^(FAILED|ERROR)" $O/sparse.log | head
AH_FILTER_SPARSE=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_filter_small.py tests/test_gpu_filter_expr.py tests/test_gpu_deferred.py -q -m gpu -k "filter or coalesc or record_batch or deferred" > $O/forced.log 2>&1; tail -2 $O/forced.log; grep -E "^(FAILED|ERROR)" $O/forced.log | head
bash tools/selectivity_sweep.sh > $O/selectivity_sweep.md 2> $O/selectivity_sweep.err; cat $O/selectivity_sweep.md
