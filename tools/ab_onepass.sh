#!/bin/bash
# A/B of the one-pass string cast's ablation builds on ONE box (tools/ablate_build.sh op_<tag> cast_string.hip "<flags>").
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do for v in ${AB_VARIANTS:-default}; do
  if [ $v = default ]; then unset AH_LIB_PATH; else export AH_LIB_PATH=$PWD/arrow-rs_amd/lib/ablate/libarrow_hip_$v.so; fi
  for w in ${AB_WORKLOADS:-cast_string}; do
    echo "== $v $w rep $rep pure=${AH_BENCH_CAST_PURE:-0}"
    timeout 120 python bench.py --workload $w --no-cpu-baseline --pmc-traffic off 2>/dev/null | python tools/bench_pick.py kernel_avg_ms
  done
done; done
