#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03_run3; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_comm.py tests/test_gpu_filter_expr.py -x -q -m gpu > $O/pytest.log 2>&1; tail -15 $O/pytest.log
for wl in predicate_filter_fused; do
timeout 300 python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline --pmc-traffic off > $O/$wl.json 2> $O/$wl.err
grep "^{" $O/$wl.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl', d['ms_per_step'], d['kernel_avg_ms'], d['roofline']['frac'], d.get('host_gap_ms'))" || tail -5 $O/$wl.err
done
# non-temporal A/B over the four streaming kernels
for v in default ntL ntS ntLS; do
  if [ $v = default ]; then unset AH_LIB_PATH; else export AH_LIB_PATH=$GRAFT_REPO_ROOT/arrow-rs_amd/lib/ablate/libarrow_hip_$v.so; fi
  for wl in arith cmp cast filter_take; do
    timeout 300 python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline --pmc-traffic off --no-configs > $O/nt_${v}_$wl.json 2> $O/nt_${v}_$wl.err
    grep "^{" $O/nt_${v}_$wl.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('NT $v $wl', d['ms_per_step'], d['kernel_avg_ms'])" || tail -3 $O/nt_${v}_$wl.err
  done
done
unset AH_LIB_PATH
