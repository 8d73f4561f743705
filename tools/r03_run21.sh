#!/bin/bash
# the plain lazy-predicate count pass with its two halves unrolled (sv5), and with 6 waves per SIMD demanded on top (sv6)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03_run21; mkdir -p $O
run() {  # name, lib, stash_min
  AH_LIB_PATH=$2 AH_FILTER_EXPR_STASH_MIN=$3 timeout 300 python bench.py --workload predicate_filter_fused --steps 8 --warmup 2 --no-cpu-baseline --pmc-traffic off > $O/$1.json 2> $O/$1.err
  python - "$1" <<'PY'
import json, sys
n = sys.argv[1]
try:
    j = json.loads(open(f"gpurun_out/r03_run21/{n}.json").read().strip().splitlines()[-1])
    print(n, j["ms_per_step"], j.get("kernel_avg_ms"))
except Exception as e:
    print(n, "failed", e)
PY
}
L=$GRAFT_REPO_ROOT/arrow-rs_amd/lib
run nostash $L/libarrow_hip.so -1
run sv5_unrolled $L/ablate/libarrow_hip_sv5.so -1
run sv6_unrolled_6waves $L/ablate/libarrow_hip_sv6.so -1
run nostash_again $L/libarrow_hip.so -1
