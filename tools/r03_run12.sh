#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03_run12; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_like.py tests/test_gpu_sort.py tests/test_gpu_comm.py -q -m gpu -k "string or utf8 or coalesc or bytes or str or exchange" -p no:cacheprovider > $O/pytest.log 2>&1; tail -12 $O/pytest.log | grep -v "^\.\.\.\."
for wl in string_filter; do
timeout 300 python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --pmc-traffic off > $O/$wl.json 2> $O/$wl.err
grep "^{" $O/$wl.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl', d['ms_per_step'], d['kernel_avg_ms'], d['roofline']['frac'], d['roofline']['algorithmic_bytes_per_launch'], d.get('host_gap_ms'))" || tail -5 $O/$wl.err
done
