#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03_run16; mkdir -p $O
timeout 1700 python -m pytest tests -q -m gpu -p no:cacheprovider --deselect tests/test_gpu_fullsize.py --deselect tests/test_gpu_comm.py > $O/pytest.log 2>&1; tail -6 $O/pytest.log | grep -v "^\.\.\.\."; grep -E "^FAILED|^ERROR" $O/pytest.log | head -20
python tools/size_sweep.py 1e4 1e5 1e6 > $O/size_sweep.txt 2>&1; cat $O/size_sweep.txt
for wl in arith string_filter; do
timeout 300 python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --pmc-traffic off > $O/$wl.json 2> $O/$wl.err
grep "^{" $O/$wl.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl', d['ms_per_step'], d['kernel_avg_ms'], d['roofline']['frac'], d.get('host_gap_ms'))" || tail -5 $O/$wl.err
done
