#!/bin/bash
# A/B timing of library builds on ONE box (boxes differ by a few per cent, so only same-session numbers compare):
# runs bench.py alternately with each library ROUNDS times and prints the sweep-kernel ms of every run.
#   tools/ab.sh "bench args" libA.so libB.so [libC.so ...]        ("shipped" = the in-tree csrc/libmcq_hip.so)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
ARGS=$1; shift
for r in $(seq 1 ${ROUNDS:-3}); do
  for lib in "$@"; do
    if [ "$lib" = "shipped" ]; then env=""; else env="MCQ_ALLOW_DIAG=1 MCQ_DIAG_LIB=$ROOT/$lib"; fi
    out=$(env $env python $ROOT/bench.py --no-cpu-baseline $ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms  %.4e moves/s' % (d['kernel_ms'].get('sweep', d['kernel_ms'].get('all_launches')), d['value']))")
    echo "AB round $r $lib: $out"
  done
done
