#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02_run8
mkdir -p $O
L=$GRAFT_REPO_ROOT/arrow-rs_amd/lib
T="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-configs --pmc-traffic off"
for v in base FSW6 FSW8 base FSW6; do
  if [ $v = base ]; then $T > $O/ft_$v.json 2> $O/ft_$v.err; else AH_LIB_PATH=$L/libarrow_hip_$v.so $T > $O/ft_$v.json 2> $O/ft_$v.err; fi
  grep "^{" $O/ft_$v.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['kernel_avg_ms'])"
done
python bench.py --workload cast_string --steps 5 --warmup 2 --no-cpu-baseline --pmc-traffic off | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cs', d['ms_per_step'], d['kernel_avg_ms'])"
python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -x -q -k "cast or string or config3" 2>&1 | tail -3
