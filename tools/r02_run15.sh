#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02_run15; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -3; grep -E "^(FAILED|ERROR)|Error|assert" $O/pytest.log | head -20
