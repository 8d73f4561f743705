#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02_run6
mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=3 ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -8 $O/pytest.log
B="python bench.py --workload cast_string --steps 5 --warmup 2 --no-cpu-baseline --pmc-traffic off"
run() { name=$1; shift; env "$@" $B > $O/$name.json 2> $O/$name.err; grep "^{" $O/$name.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', d['ms_per_step'], d['kernel_avg_ms'], d['roofline']['frac'])"; tail -2 $O/$name.err; }
run cs X=1
run cs_pure AH_BENCH_CAST_PURE=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_cs -o b -- $B > /dev/null 2> $O/trace_cs.log
head -7 $O/trace_cs/b_kernel_stats.csv | cut -c1-150
