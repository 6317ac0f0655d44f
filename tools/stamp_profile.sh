#!/bin/bash
# Diagnostic only: a build of libmcq_hip.so with s_memtime stamps (-DMCQ_STAMPS) in build/, loaded through MCQ_DIAG_LIB,
# prints the share of wavefront cycles per section of the Metropolis step.  Read the SHARES, not the run time (the stamps
# fence the scheduler).  Build it where hipcc is cheap (the build container); build/ travels to the GPU box with gpurun.
# usage: tools/stamp_profile.sh [bench.py args...]
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
SRC=$ROOT/monte-carlo-collective_amd/csrc/mcq_hip.hip
LIB=$ROOT/build/libmcq_hip_stamps.so
mkdir -p $ROOT/build
if [ ! -f $LIB ] || [ $SRC -nt $LIB ]; then
  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -fPIC -shared -DMCQ_STAMPS -o $LIB $SRC
fi
[ "$1" = "--build-only" ] && exit 0
MCQ_ALLOW_DIAG=1 MCQ_DIAG_LIB=$LIB python $ROOT/bench.py --steps 1 --warmup 0 --n-steps 20000 --no-cpu-baseline "$@" > /dev/null
