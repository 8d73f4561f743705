#!/bin/bash
# where does the stash form of the count pass lose its time?  ablation builds (csrc/Makefile EXTRA=-DAH_STASH_VARIANT=n)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03_run20; mkdir -p $O
run() {  # name, lib, stash_min
  AH_LIB_PATH=$2 AH_FILTER_EXPR_STASH_MIN=$3 timeout 300 python bench.py --workload predicate_filter_fused --steps 8 --warmup 2 --no-cpu-baseline --pmc-traffic off > $O/$1.json 2> $O/$1.err
  python - "$1" <<'PY'
import json, sys
n = sys.argv[1]
try:
    j = json.loads(open(f"gpurun_out/r03_run20/{n}.json").read().strip().splitlines()[-1])
    print(n, j["ms_per_step"], j.get("kernel_avg_ms"))
except Exception as e:
    print(n, "failed", e)
PY
}
L=$GRAFT_REPO_ROOT/arrow-rs_amd/lib
run stash $L/libarrow_hip.so 0
run nostash $L/libarrow_hip.so -1
run sv1_no_global_stores $L/ablate/libarrow_hip_sv1.so 0
run sv2_registers_only $L/ablate/libarrow_hip_sv2.so 0
run sv4_plain_capped_4_waves $L/ablate/libarrow_hip_sv4.so -1
