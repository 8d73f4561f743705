#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03_run14; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_deferred.py tests/test_gpu_filter_expr.py -q -m gpu -k "cmp or compare or lt or eq or distinct or golden or deferred or expr" -p no:cacheprovider > $O/pytest.log 2>&1; tail -8 $O/pytest.log | grep -v "^\.\.\.\."
python tools/size_sweep.py 1e4 1e5 1e6 > $O/size_sweep.txt 2>&1; cat $O/size_sweep.txt
timeout 300 python bench.py --reassemble allgatherv --steps 3 --warmup 1 --no-cpu-baseline --config-steps 2 --pmc-traffic off > $O/bench_exchange_world1.json 2> $O/bench_exchange_world1.err; echo rc=$?
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r03_run14/bench_exchange_world1.json') if l.startswith('{')][-1])
print(d['n_gpus'], d['config'], d['ms_per_step'], d.get('local_ms_per_step'), d.get('reassemble_last_ms'), d.get('exchange'))
print(d.get('configs'))
PY
tail -3 $O/bench_exchange_world1.err
