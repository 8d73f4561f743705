#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03_run28; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_filter_sparse.py tests/test_gpu_filter_small.py tests/test_gpu_parity.py -q -m gpu -k "filter or coalesc or record_batch" > $O/pytest.log 2>&1; tail -1 $O/pytest.log; grep -E "^(FAILED|ERROR)" $O/pytest.log | head
VALID_LIST="0.9" SEL_LIST="0.0009765625 0.01 0.03 0.05 0.1 0.12 0.2 0.5 0.9990234375" bash tools/selectivity_sweep.sh | tee $O/selectivity_sweep.md
