#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02_run14; mkdir -p $O
for fz in 1 0; do
AH_COUNT_FUSE=$fz rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr$fz -o b -- python bench.py --workload coalesce --steps 5 --warmup 2 --no-cpu-baseline --pmc-traffic off > $O/co$fz.json 2> $O/co$fz.err
f=$(find $O/tr$fz -name "b_kernel_stats.csv" | head -1); echo "== fuse $fz"; grep -E "filter_count_small|group_scan|filter_scatter" "$f" | cut -d, -f1-4 | cut -c1-60,150-
AH_COUNT_FUSE=$fz timeout 300 python bench.py --workload coalesce --steps 10 --warmup 3 --no-cpu-baseline --pmc-traffic off > $O/co_f$fz.json 2> $O/co_f$fz.err
grep "^{" $O/co_f$fz.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fuse $fz', d['ms_per_step'], d.get('ms_per_step_without_kernel_events'), d['kernel_avg_ms'], d['roofline']['frac'])"
done
