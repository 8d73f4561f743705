#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03_run1; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_comm.py -x -q -m gpu > $O/pytest.log 2>&1; tail -30 $O/pytest.log
