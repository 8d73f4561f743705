#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03_run10; mkdir -p $O
for abl in 0 1; do
AH_SP_ABLATE=$abl timeout 200 python bench.py --workload predicate_filter_fused --steps 10 --warmup 3 --no-cpu-baseline --pmc-traffic off > $O/fused_abl$abl.json 2> $O/fused_abl$abl.err
grep "^{" $O/fused_abl$abl.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ablate $abl', d['ms_per_step'], d['kernel_avg_ms'], d['roofline']['frac'], d.get('host_gap_ms'))" || tail -5 $O/fused_abl$abl.err
done
