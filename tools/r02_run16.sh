#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02_run16; mkdir -p $O
for wl in string_filter_take coalesce; do
timeout 600 python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline > $O/$wl.json 2> $O/$wl.err
grep "^{" $O/$wl.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl', d['ms_per_step'], d.get('ms_per_step_without_kernel_events'), d['kernel_avg_ms'], d['roofline'], d.get('pmc_traffic_bytes_per_launch'))"
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -o b -- python bench.py --workload string_filter_take --steps 5 --warmup 2 --no-cpu-baseline --pmc-traffic off > $O/sft_tr.json 2> $O/sft_tr.err
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r02_run16/tr/**/b_kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:22]:
    print(r['Name'][:70].replace('(anonymous namespace)::',''), r['Calls'], round(float(r['AverageNs'])/1e3,1), r['Percentage'])
PY
