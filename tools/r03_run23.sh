#!/bin/bash
# what the driver runs at round end, in its order: smoke, then the default bench line; plus the --gpus 2 refusal on a 1-GPU box
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03_run23; mkdir -p $O
timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.log 2>&1; tail -3 $O/smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; cut -c1-600 $O/bench.json
timeout 120 python bench.py --gpus 2 --steps 2 --warmup 1 > $O/bench2.json 2> $O/bench2.err; echo "gpus2 rc $?"; cut -c1-400 $O/bench2.json
