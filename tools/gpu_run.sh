#!/bin/bash
# The ONE script every GPU-box call of a round goes through (replaces the 50 one-off tools/r0x_runN.sh of rounds 2-3;
# those are in the git history).  usage, from the build container:
#   gpurun --timeout S -- 'bash tools/gpu_run.sh <tag> "<command 1>" "<command 2>" ...'
# Every command runs from the repo root under its own `timeout` (AH_STEP_TIMEOUT, default 900 s), stdout + stderr into
# gpurun_out/<tag>/stepN.log; the tail of each log is echoed so it shows up in gpurun's own tail.  Commands may refer
# to $O (= gpurun_out/<tag>) for files they want merged back.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 9
tag=$1; shift
export O=gpurun_out/$tag; mkdir -p "$O"
i=0; rc=0
for cmd in "$@"; do
  i=$((i + 1))
  echo "=== [$tag] step $i: $cmd"
  t0=$(date +%s)
  timeout "${AH_STEP_TIMEOUT:-900}" bash -c "$cmd" > "$O/step$i.log" 2>&1; r=$?
  echo "--- rc=$r ($(( $(date +%s) - t0 )) s)"; tail -n "${AH_TAIL:-12}" "$O/step$i.log"
  grep -E "^(FAILED|ERROR)" "$O/step$i.log" | head -8
  [ $r -ne 0 ] && rc=$r
done
exit $rc
