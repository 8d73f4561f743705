#!/bin/bash
# Randomized parity tests of the GPU suite with shifted seeds (tests/conftest.py: AH_SEED_OFFSET).
# usage (on the GPU box): bash tools/soak_gpu.sh [first] [last]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/soak; mkdir -p $O
for k in $(seq ${1:-1} ${2:-6}); do
  AH_SEED_OFFSET=$k timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu \
    -k "fuzz or coalescer or record_batch or sparse_and_dense or cast_f64 or cast_f32 or one_launch" > $O/seed$k.log 2>&1
  echo "seed offset $k: $(grep -E 'passed|failed|error' $O/seed$k.log | tail -1)"
  grep -E "^(FAILED|ERROR)" $O/seed$k.log | head -5
done
