#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03_run5; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu --deselect tests/test_gpu_fullsize.py -p no:cacheprovider > $O/pytest.log 2>&1; tail -40 $O/pytest.log | grep -v "^\.\.\.\." 
for wl in coalesce; do
timeout 300 python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --pmc-traffic off > $O/$wl.json 2> $O/$wl.err
grep "^{" $O/$wl.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl', d['ms_per_step'], d.get('ms_per_step_without_kernel_events'), d['kernel_avg_ms'], d['roofline']['frac'], d.get('host_gap_ms'))" || tail -5 $O/$wl.err
done
python tools/size_sweep.py 1e4 1e5 1e6 > $O/size_sweep.txt 2>&1; cat $O/size_sweep.txt
python tools/record_batch_latency.py > $O/rbl.txt 2>&1; tail -12 $O/rbl.txt
