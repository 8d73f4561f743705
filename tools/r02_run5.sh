#!/bin/bash
# round-2 GPU call 5: cast_string tile-size ablation; the default bench line; per-kernel rocprof evidence
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02_run5
mkdir -p $O
B="python bench.py --workload cast_string --steps 5 --warmup 2 --no-cpu-baseline --pmc-traffic off"
L=$GRAFT_REPO_ROOT/arrow-rs_amd/lib
run() { name=$1; shift; env "$@" $B > $O/$name.json 2> $O/$name.err; grep "^{" $O/$name.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', d['ms_per_step'], d['kernel_avg_ms'], d['roofline']['frac'])"; tail -2 $O/$name.err; }
run cs_base X=1
run cs_T1024_L1 AH_LIB_PATH=$L/libarrow_hip_T1024_L1.so
run cs_T512_L4 AH_LIB_PATH=$L/libarrow_hip_T512_L4.so
run cs_T256_L4 AH_LIB_PATH=$L/libarrow_hip_T256_L4.so
run cs_pure_base AH_BENCH_CAST_PURE=1
run cs_pure_T1024_L1 AH_BENCH_CAST_PURE=1 AH_LIB_PATH=$L/libarrow_hip_T1024_L1.so
run cs_pure_T256_L4 AH_BENCH_CAST_PURE=1 AH_LIB_PATH=$L/libarrow_hip_T256_L4.so
( time python bench.py --steps 20 --warmup 5 ) > $O/bench_default.json 2> $O/bench_default.err
grep "^{" $O/bench_default.json | python -c "
import json,sys; d=json.loads(sys.stdin.read())
print('default', d['value'], d['ms_per_step'], d['host_gap_ms'], d['kernel_avg_ms'], d['roofline']['frac'], d['roofline'].get('traffic_frac'))
for k,v in d.get('configs',{}).items(): print(' ', k, v.get('ms'), v.get('roofline',{}).get('frac'), v.get('kernel_avg_ms'))
print(' cpu', d['cpu_baseline']['value'], d['cpu_baseline']['all_cores'])"
tail -3 $O/bench_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_all -o b -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --pmc-traffic off > $O/bench_trace_all.json 2> $O/trace_all.log
head -30 $O/trace_all/b_kernel_stats.csv | cut -c1-170
rocprofv3 --hip-trace --kernel-trace --stats --output-format csv -d $O/hiptrace -o b -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --pmc-traffic off > $O/bench_hiptrace.json 2> $O/hiptrace.log
head -14 $O/hiptrace/b_hip_api_stats.csv | cut -c1-120
