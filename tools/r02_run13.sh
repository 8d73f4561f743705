#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02_run13; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "filter or coalesc or config1 or record_batch" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --pmc-traffic off --no-configs > $O/ft.json 2> $O/ft.err || timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --pmc-traffic off > $O/ft.json 2> $O/ft.err
grep "^{" $O/ft.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ft', d['ms_per_step'], d['kernel_avg_ms'], d.get('host_gap_ms'))"
for wl in coalesce record_batch; do
timeout 300 python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline --pmc-traffic off > $O/$wl.json 2> $O/$wl.err
grep "^{" $O/$wl.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl', d['ms_per_step'], d.get('ms_per_step_without_kernel_events'), d['kernel_avg_ms'], d['roofline']['frac'])"
done
