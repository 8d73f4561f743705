#!/bin/bash
# soak of the round-3 parity tests with shifted seeds; string filter ranges stage size A/B (2048 vs 1024 rows per round)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03_run22; mkdir -p $O
for k in 1 2 3; do
  AH_SEED_OFFSET=$k timeout 900 python -m pytest tests/test_gpu_filter_expr.py tests/test_gpu_filter_small.py tests/test_gpu_parity.py -q -m gpu \
    -k "not golden" > $O/soak$k.log 2>&1
  echo "seed offset $k: $(grep -E 'passed|failed|error' $O/soak$k.log | tail -1)"; grep -E "^(FAILED|ERROR)" $O/soak$k.log | head -5
done
run() {  # name, lib, workload
  AH_LIB_PATH=$2 timeout 300 python bench.py --workload $3 --steps 10 --warmup 3 --no-cpu-baseline --pmc-traffic off > $O/$1.json 2> $O/$1.err
  python - "$1" <<'PY'
import json, sys
n = sys.argv[1]
try:
    j = json.loads(open(f"gpurun_out/r03_run22/{n}.json").read().strip().splitlines()[-1])
    print(n, j["ms_per_step"], j.get("kernel_avg_ms"))
except Exception as e:
    print(n, "failed", e)
PY
}
L=$GRAFT_REPO_ROOT/arrow-rs_amd/lib
run sf_cap2048 $L/libarrow_hip.so string_filter
run sf_cap1024 $L/ablate/libarrow_hip_cap1k.so string_filter
run sf_cap2048_b $L/libarrow_hip.so string_filter
run sf_cap1024_b $L/ablate/libarrow_hip_cap1k.so string_filter
AH_LIB_PATH=$L/ablate/libarrow_hip_cap1k.so timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "string or utf8 or Utf8" > $O/cap1k_strings.log 2>&1; tail -1 $O/cap1k_strings.log
