#!/bin/bash
# Round-3 closing run: the whole GPU suite (no -x), then the evidence collection on the same tree.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03_run17; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_full.log 2>&1
tail -5 $O/pytest_full.log
bash tools/collect_profiles_r03.sh > $O/collect.log 2>&1
tail -3 $O/collect.log
