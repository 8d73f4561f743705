#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02_run25; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_deferred.py tests/test_gpu_fullsize.py -x -q -m gpu -k "arith or cmp or compare or config2 or deferred or scalar or nan" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
for wl in arith cmp; do
timeout 300 python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline --pmc-traffic off > $O/$wl.json 2> $O/$wl.err
grep "^{" $O/$wl.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl', d['ms_per_step'], d['kernel_avg_ms'], d['roofline']['frac'], d.get('host_gap_ms'))"
done
python tools/size_sweep.py 2>&1 | tail -8
