#!/bin/bash
# Timing experiments only: build libmcq_hip.so with extra -D flags into place, run a short bench, restore the real library.
# usage: tools/exp_build.sh "-DFOO -DBAR" [bench.py args...]
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
SRC=$ROOT/monte-carlo-collective_amd/csrc
FLAGS=$1; shift
cp $SRC/libmcq_hip.so /tmp/libmcq_hip.so.keep
hipcc --offload-arch=${MCQ_ARCH:-gfx950} -O3 -ffp-contract=off -std=c++17 -fPIC -shared $FLAGS -o $SRC/libmcq_hip.so $SRC/mcq_hip.hip
python $ROOT/bench.py --steps 2 --warmup 1 --n-steps 20000 --no-cpu-baseline "$@" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('EXP', '$FLAGS', '%.3e'%d['value'], d['kernel_ms'])" || true
cp /tmp/libmcq_hip.so.keep $SRC/libmcq_hip.so
