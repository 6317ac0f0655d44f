#!/bin/bash
# Timing experiments only: build a variant of libmcq_hip.so with extra -D flags into build/ (never over the shipped library),
# run a short bench with it through MCQ_DIAG_LIB, print one EXP line.  Run on the GPU box, or build here and bench there:
#   tools/exp_build.sh --build-only NAME "-DFOO -DBAR"        (build/libmcq_hip_NAME.so travels with gpurun)
#   tools/exp_build.sh NAME "-DFOO -DBAR" [bench.py args...]
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
SRC=$ROOT/monte-carlo-collective_amd/csrc/mcq_hip.hip
BUILD_ONLY=0
if [ "$1" = "--build-only" ]; then BUILD_ONLY=1; shift; fi
NAME=$1; FLAGS=$2; shift 2
LIB=$ROOT/build/libmcq_hip_$NAME.so
mkdir -p $ROOT/build
if [ ! -f $LIB ] || [ $SRC -nt $LIB ] || [ $ROOT/include/mcq.h -nt $LIB ]; then
  hipcc --offload-arch=${MCQ_ARCH:-gfx950} -O3 -ffp-contract=off -std=c++17 -fPIC -shared $FLAGS -o $LIB $SRC
fi
[ $BUILD_ONLY = 1 ] && exit 0
MCQ_ALLOW_DIAG=1 MCQ_DIAG_LIB=$LIB python $ROOT/bench.py --steps 2 --warmup 1 --n-steps 20000 --no-cpu-baseline "$@" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('EXP', '$NAME', '$FLAGS', '%.3e'%d['value'], d['kernel_ms'])"
