#!/bin/bash
# round-2 GPU call 4: suite; cast_string A/B (digit code, declen, side queue, input); coalescer with no-wait pushes
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02_run4
mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=5 ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -12 $O/pytest.log
B="python bench.py --workload cast_string --steps 5 --warmup 2 --no-cpu-baseline --pmc-traffic off"
L=$GRAFT_REPO_ROOT/arrow-rs_amd/lib
run() { name=$1; shift; env "$@" $B > $O/$name.json 2> $O/$name.err; grep "^{" $O/$name.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', d['ms_per_step'], d['kernel_avg_ms'], d['roofline']['frac'])"; }
run cs_new X=1
run cs_olddigits AH_LIB_PATH=$L/libarrow_hip_OLD_DIGITS.so
run cs_tabledeclen AH_LIB_PATH=$L/libarrow_hip_TABLE_DECLEN.so
run cs_nosideq AH_CAST_SIDEQ=0
run cs_pure_new AH_BENCH_CAST_PURE=1
run cs_pure_olddigits AH_BENCH_CAST_PURE=1 AH_LIB_PATH=$L/libarrow_hip_OLD_DIGITS.so
run cs_pure_tabledeclen AH_BENCH_CAST_PURE=1 AH_LIB_PATH=$L/libarrow_hip_TABLE_DECLEN.so
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_cs -o b -- $B > /dev/null 2> $O/trace_cs.log
head -7 $O/trace_cs/b_kernel_stats.csv | cut -c1-150
python bench.py --workload coalesce --steps 3 --warmup 1 --no-cpu-baseline --pmc-traffic off > $O/coalesce.json 2> $O/coalesce.err
grep "^{" $O/coalesce.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('coalesce', d['ms_per_step'], d['kernel_avg_ms'], d['roofline']['frac'], d['host_gap_ms'])"
tail -3 $O/coalesce.err
