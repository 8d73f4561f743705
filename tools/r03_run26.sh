#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03_run26; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_filter_sparse.py -q -m gpu > $O/sparse.log 2>&1; tail -2 $O/sparse.log; grep -E "^(FAILED|ERROR)" $O/sparse.log | head
SEL_LIST="0.0009765625 0.004 0.01 0.015 0.03" bash tools/selectivity_sweep.sh > $O/sweep_default.md 2> $O/sweep.err; cat $O/sweep_default.md
echo "--- tiled kernel forced (AH_FILTER_SPARSE=0)"
AH_FILTER_SPARSE=0 SEL_LIST="0.0009765625 0.004 0.01 0.015 0.03" bash tools/selectivity_sweep.sh | tail -n +3
echo "--- sparse kernel forced (AH_FILTER_SPARSE=1)"
AH_FILTER_SPARSE=1 SEL_LIST="0.015 0.03 0.1" bash tools/selectivity_sweep.sh | tail -n +3
