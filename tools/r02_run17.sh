#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02_run17; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu --durations=8 > $O/pytest.log 2>&1; tail -25 $O/pytest.log
