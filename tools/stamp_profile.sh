#!/bin/bash
# Diagnostic only: build libmcq_hip.so with s_memtime stamps (-DMCQ_STAMPS) into a scratch directory, run a short
# bench with it and print the share of wavefront cycles per section of the Metropolis step.  Read the SHARES, not the
# run time (the stamps fence the scheduler).  usage: tools/stamp_profile.sh [bench.py args...]
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
SRC=$ROOT/monte-carlo-collective_amd/csrc
cp $SRC/libmcq_hip.so /tmp/libmcq_hip.so.keep
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -fPIC -shared -DMCQ_STAMPS -o $SRC/libmcq_hip.so $SRC/mcq_hip.hip
python $ROOT/bench.py --steps 1 --warmup 0 --n-steps 20000 --no-cpu-baseline "$@" > /dev/null || true
cp /tmp/libmcq_hip.so.keep $SRC/libmcq_hip.so
