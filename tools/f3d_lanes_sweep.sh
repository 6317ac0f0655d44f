#!/bin/bash
# full_3d, 65 536 chains x 20 000 steps, full trace: 8 lanes per chain against the slim 4-lane layout, by N.   usage: tools/f3d_lanes_sweep.sh OUTFILE [N ...]
OUT=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for N in ${@:-9 10 11 12 13 14 15 16}; do
  for lanes in 8 4; do
    python $ROOT/bench.py --config c3 --N $N --steps 2 --warmup 1 --n-steps 20000 --no-cpu-baseline --lanes $lanes 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('full_3d N=$N lanes $lanes moves/s %.4e sweep_ms %.2f' % (d['value'], d['kernel_ms']['sweep']))" | tee -a $ROOT/$OUT
  done
done
