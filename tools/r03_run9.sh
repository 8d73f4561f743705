#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03_run9; mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_filter_expr.py -q -m gpu -p no:cacheprovider -x > $O/pytest.log 2>&1; tail -15 $O/pytest.log | grep -v "^\.\.\.\."
for wl in predicate_filter_fused; do
timeout 200 python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline --pmc-traffic off > $O/$wl.json 2> $O/$wl.err
grep "^{" $O/$wl.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl', d['ms_per_step'], d['kernel_avg_ms'], d['roofline']['frac'], d.get('host_gap_ms'))" || tail -5 $O/$wl.err
done
AH_COALESCE_GROUP=8 timeout 300 python bench.py --workload coalesce --steps 5 --warmup 2 --no-cpu-baseline --pmc-traffic off > $O/coalesce_g8.json 2> $O/coalesce_g8.err
grep "^{" $O/coalesce_g8.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('coalesce group 8', d['ms_per_step'], d.get('ms_per_step_without_kernel_events'), d['kernel_avg_ms'], d['roofline']['frac'], d.get('host_gap_ms'))" || tail -5 $O/coalesce_g8.err
timeout 300 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -k lazy -p no:cacheprovider > $O/pytest_full.log 2>&1; tail -5 $O/pytest_full.log
