#!/bin/bash
# A/B of the coalescer's table upload: the context's copy stream (default) against the context's main stream
# (AH_COALESCE_COPY_STREAM=0).   usage (GPU box): bash tools/ab_copy_stream.sh <out dir>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=${1:-gpurun_out/ab_copy_stream}
mkdir -p $O
KEYS="8192x8192,65536x2^20"
for rep in 1 2; do
  for cs in 1 0; do
    AH_COALESCE_COPY_STREAM=$cs timeout 200 python bench.py --only-coalesce-sweep "$KEYS" --no-cpu-baseline > $O/sweep_cs${cs}_$rep.json 2> $O/sweep_cs${cs}_$rep.err
    echo "== copy_stream=$cs rep $rep"; python tools/show_sweep.py $O/sweep_cs${cs}_$rep.json
  done
done
