#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03_run15; mkdir -p $O
timeout 1700 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/pytest.log 2>&1; tail -6 $O/pytest.log | grep -v "^\.\.\.\."; grep -E "^FAILED|^ERROR" $O/pytest.log | head -20
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
