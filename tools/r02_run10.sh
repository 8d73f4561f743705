#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02_run10; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "cast or config3" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for wl in cast cast_string; do
  timeout 300 python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline --pmc-traffic off > $O/$wl.json 2> $O/$wl.err
  grep "^{" $O/$wl.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl', d['ms_per_step'], d['kernel_avg_ms'], d['roofline']['frac'])"
done
AH_BENCH_CAST_PURE=1 timeout 300 python bench.py --workload cast_string --steps 10 --warmup 3 --no-cpu-baseline --pmc-traffic off > $O/cs_pure.json 2> $O/cs_pure.err
grep "^{" $O/cs_pure.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cs_pure', d['ms_per_step'], d['kernel_avg_ms'], d['roofline']['frac'])"
