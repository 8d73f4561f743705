#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03_run8; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_deferred.py -q -m gpu -k "coalesc" -p no:cacheprovider > $O/pytest.log 2>&1; tail -12 $O/pytest.log | grep -v "^\.\.\.\."
for g in 8 1; do
AH_COALESCE_GROUP=$g timeout 300 python bench.py --workload coalesce --steps 5 --warmup 2 --no-cpu-baseline --pmc-traffic off > $O/coalesce_g$g.json 2> $O/coalesce_g$g.err
grep "^{" $O/coalesce_g$g.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('coalesce group $g', d['ms_per_step'], d.get('ms_per_step_without_kernel_events'), d['kernel_avg_ms'], d['roofline']['frac'], d.get('host_gap_ms'))" || tail -5 $O/coalesce_g$g.err
done
