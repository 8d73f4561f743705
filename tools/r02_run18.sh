#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02_run18; mkdir -p $O
AH_DEBUG_REDZONE=1 timeout 1200 python -m pytest tests -x -q -m gpu > $O/pytest_redzone.log 2>&1; grep -E "passed|failed" $O/pytest_redzone.log | tail -3; grep -E "^(FAILED|ERROR)|redzone" $O/pytest_redzone.log | head
