#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03_run2; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_comm.py tests/test_gpu_filter_expr.py -x -q -m gpu > $O/pytest.log 2>&1; tail -15 $O/pytest.log
for wl in predicate_filter predicate_filter_fused; do
timeout 300 python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline --pmc-traffic off > $O/$wl.json 2> $O/$wl.err
grep "^{" $O/$wl.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl', d['ms_per_step'], d['kernel_avg_ms'], d['roofline']['frac'], d.get('host_gap_ms'))" || tail -5 $O/$wl.err
done
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k lazy > $O/pytest_full.log 2>&1; tail -5 $O/pytest_full.log
