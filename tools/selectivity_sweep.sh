#!/bin/bash
# filter (and the take of as many indices) across selectivities and null densities at 1e9 Int64 rows — the axis the
# reference's own bench sweeps (arrow/benches/filter_kernels.rs: kept 1/1024, 1/2, 1023/1024; with / without nulls).
# usage (GPU box): bash tools/selectivity_sweep.sh > gpurun_out/selectivity_sweep.md
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/sel_sweep; mkdir -p $O
echo "| selectivity | valid | filter_count ms | filter_scatter ms | filter frac (algorithmic bytes / 8 TB/s) | take_gather ms (selectivity x 1e9 random u32 indices) | take frac |"
echo "|---|---|---|---|---|---|---|"
for v in ${VALID_LIST:-0.9 1.0}; do
for s in ${SEL_LIST:-0.0009765625 0.01 0.1 0.5 0.8 0.9990234375}; do
  timeout 300 python bench.py --workload filter_take --selectivity $s --valid $v --steps 5 --warmup 2 --no-cpu-baseline --no-configs --pmc-traffic off > $O/s_${s}_v$v.json 2> $O/s_${s}_v$v.err
  python - "$O/s_${s}_v$v.json" $s $v <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k = j["kernel_avg_ms"]
    f = j.get("roofline_filter_scatter", {})
    print(f"| {float(sys.argv[2]):.4f} | {sys.argv[3]} | {k.get('filter_count', 0):.3f} | {k.get('filter_scatter', 0):.3f} | {f.get('frac', 0):.3f} | {k.get('take_gather', 0):.3f} | {j['roofline']['frac']:.3f} |")
except Exception as e:
    print(f"| {sys.argv[2]} | {sys.argv[3]} | failed: {e} |")
PY
done
done
