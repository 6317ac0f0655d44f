#!/bin/bash
# Diagnostic only: a build with -DMCQ_WAVE_TIMES records start / end (s_memrealtime, 100 MHz) and placement (XCC, SE, CU,
# SIMD, wave slot) of every wavefront of the sweep; tools/wave_times_summary.py prints the distribution of end times.
# usage: tools/wave_times.sh OUT.txt [bench.py args...]     (build/ travels to the GPU box; --build-only to prebuild)
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
SRC=$ROOT/monte-carlo-collective_amd/csrc/mcq_hip.hip
LIB=$ROOT/build/libmcq_hip_wavetimes.so
mkdir -p $ROOT/build
if [ ! -f $LIB ] || [ $SRC -nt $LIB ]; then
  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -fPIC -shared -DMCQ_WAVE_TIMES -o $LIB $SRC
fi
[ "$1" = "--build-only" ] && exit 0
OUT=$1; shift
MCQ_ALLOW_DIAG=1 MCQ_DIAG_LIB=$LIB MCQ_WAVE_TIMES_OUT=$OUT python $ROOT/bench.py --steps 1 --warmup 0 --n-steps 20000 --no-cpu-baseline "$@" > /dev/null
python $ROOT/tools/wave_times_summary.py $OUT
