#!/bin/bash
# soak: the whole GPU suite (minus the fixed-input full-size file) with shifted seeds
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03_run24; mkdir -p $O
for k in 7 11; do
  AH_SEED_OFFSET=$k timeout 900 python -m pytest tests -q -m gpu --deselect tests/test_gpu_fullsize.py > $O/soak$k.log 2>&1
  echo "seed offset $k: $(grep -E 'passed|failed|error' $O/soak$k.log | tail -1)"; grep -E "^(FAILED|ERROR)" $O/soak$k.log | head -8
done
