#!/bin/bash
# round-2 GPU call 3: suite after the X2 side queue / test fixes; cast_string ablation
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02_run3
mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -16 $O/pytest.log
B="python bench.py --workload cast_string --steps 5 --warmup 2 --no-cpu-baseline --pmc-traffic off"
$B > $O/cs_sideq.json 2> $O/cs_sideq.err
AH_CAST_SIDEQ=0 $B > $O/cs_inkernel.json 2> $O/cs_inkernel.err
AH_BENCH_CAST_PURE=1 $B > $O/cs_pure.json 2> $O/cs_pure.err
for f in cs_sideq cs_inkernel cs_pure; do grep "^{" $O/$f.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$f', d['ms_per_step'], d['kernel_avg_ms'], d['roofline']['frac'])"; done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_cs -o b -- $B > /dev/null 2> $O/trace_cs.log
head -8 $O/trace_cs/b_kernel_stats.csv | cut -c1-160
python bench.py --workload coalesce --steps 3 --warmup 1 --no-cpu-baseline --pmc-traffic off > $O/coalesce.json 2> $O/coalesce.err
grep "^{" $O/coalesce.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('coalesce', d['ms_per_step'], d['kernel_avg_ms'], d['roofline']['frac'], d['host_gap_ms'])"
