#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02_run7
mkdir -p $O
L=$GRAFT_REPO_ROOT/arrow-rs_amd/lib
B="python bench.py --workload cast_string --steps 5 --warmup 2 --no-cpu-baseline --pmc-traffic off"
run() { name=$1; shift; env "$@" $B > $O/$name.json 2> $O/$name.err; grep "^{" $O/$name.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', d['ms_per_step'], d['kernel_avg_ms'], d['roofline']['frac'])"; tail -2 $O/$name.err; }
run cs X=1
run cs_sgpr80 AH_LIB_PATH=$L/libarrow_hip_SGPR80.so
run cs_pure AH_BENCH_CAST_PURE=1
run cs_pure_sgpr80 AH_BENCH_CAST_PURE=1 AH_LIB_PATH=$L/libarrow_hip_SGPR80.so
T="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-configs --pmc-traffic off"
$T > $O/ft.json 2> $O/ft.err; grep "^{" $O/ft.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ft', d['ms_per_step'], d['kernel_avg_ms'])"
AH_LIB_PATH=$L/libarrow_hip_SGPR80.so $T > $O/ft80.json 2> $O/ft80.err; grep "^{" $O/ft80.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ft_sgpr80', d['ms_per_step'], d['kernel_avg_ms'])"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d $O/sq_cs -o b -- env AH_BENCH_CAST_PURE=1 $B > /dev/null 2> $O/sq_cs.log
python - <<'PY'
import csv,collections,re
rows=list(csv.DictReader(open('gpurun_out/r02_run7/sq_cs/b_counter_collection.csv')))
agg=collections.OrderedDict()
for r in rows:
    m=re.search(r'(string_\w+)',r['Kernel_Name'])
    if m: agg.setdefault(m.group(1),collections.OrderedDict()).setdefault(r['Counter_Name'],[]).append(float(r['Counter_Value']))
for k,v in agg.items(): print(k,{c: f"{sum(x)/len(x):.3g}" for c,x in v.items()})
PY
