#!/bin/bash
# stash form of ah_filter_expr: parity, full-size check, A/B against the two-pass form; the world-8 exchange test again
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03_run18; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_filter_expr.py -q -m gpu > $O/pytest_expr.log 2>&1; tail -4 $O/pytest_expr.log
timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -k lazy_predicate > $O/pytest_full.log 2>&1; tail -3 $O/pytest_full.log
for i in 1 2 3; do
  timeout 600 python -m pytest tests/test_gpu_comm.py -q -m gpu -k "fake_transport and 8" > $O/pytest_comm$i.log 2>&1; tail -1 $O/pytest_comm$i.log
done
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 300 python bench.py --workload predicate_filter_fused --steps 10 --warmup 3 --no-cpu-baseline > $O/fused_stash.json 2> $O/fused_stash.err
AH_FILTER_EXPR_STASH_MIN=-1 timeout 300 python bench.py --workload predicate_filter_fused --steps 10 --warmup 3 --no-cpu-baseline --pmc-traffic off > $O/fused_nostash.json 2> $O/fused_nostash.err
python - <<'PY'
import json
for n in ("fused_stash", "fused_nostash"):
    try:
        j = json.loads(open(f"gpurun_out/r03_run18/{n}.json").read().strip().splitlines()[-1])
        print(n, j["ms_per_step"], j.get("kernel_avg_ms"), j["roofline"]["frac"], j["roofline"].get("traffic"))
    except Exception as e:
        print(n, "failed", e)
PY
