#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02_run9; mkdir -p $O
L=$GRAFT_REPO_ROOT/arrow-rs_amd/lib
B="python bench.py --workload cast_string --steps 5 --warmup 2 --no-cpu-baseline --pmc-traffic off"
for v in base XNOTEXT XNOSTORE XNOSTREAM; do
  if [ $v = base ]; then AH_BENCH_CAST_PURE=1 $B > $O/$v.json 2> $O/$v.err; else AH_BENCH_CAST_PURE=1 AH_LIB_PATH=$L/libarrow_hip_$v.so $B > $O/$v.json 2> $O/$v.err; fi
  grep "^{" $O/$v.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['kernel_avg_ms'])"
done
python tools/size_sweep.py > $O/size_sweep.txt 2>&1; tail -12 $O/size_sweep.txt
