#!/bin/bash
# One GPU-box session: steps run in order; a step that is KILLED (timeout / signal) ends the session (no further GPU work
# after a hang), a step that merely fails (non-zero exit) is reported and the session goes on.
# usage: tools/gpu_session.sh NAME 'cmd1' 'cmd2' ...      (logs under gpurun_out/NAME_stepK.log)
NAME=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out
k=0
for cmd in "$@"; do
  k=$((k+1))
  echo "=== step $k: $cmd"
  ( cd $ROOT && timeout -k 10 ${STEP_TIMEOUT:-900} bash -o pipefail -c "$cmd" ) > $ROOT/gpurun_out/${NAME}_step$k.log 2>&1
  rc=$?
  tail -n ${TAIL:-6} $ROOT/gpurun_out/${NAME}_step$k.log | cut -c1-600
  echo "=== step $k rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -ge 128 ]; then echo "step $k was killed: stopping the session"; exit $rc; fi
done
exit 0
