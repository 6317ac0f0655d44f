#!/bin/bash
# c4 at 1 024 chains per cell: lane plans for the long cells (N = 17..20), with the job priorities.
OUT=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
run() { python $ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 1 --config c4 "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%.4e moves/s  step %.2f ms  sweeps %.2f ms' % (d['value'], d['ms_per_step'], d['kernel_ms']['sweeps']))"; }
{
for r in 1 2; do
echo -n "plan: 4 lanes everywhere            : "; run
echo -n "plan: N=17..20 at 8 lanes           : "; MCQ_LANES_PLAN=17:8,18:8,19:8,20:8 run
echo -n "plan: N=19,20 at 8 lanes            : "; MCQ_LANES_PLAN=19:8,20:8 run
echo -n "plan: N=17..20 at 16 lanes          : "; MCQ_LANES_PLAN=17:16,18:16,19:16,20:16 run
echo -n "plan: N=13..20 at 8 lanes           : "; MCQ_LANES_PLAN=13:8,14:8,15:8,16:8,17:8,18:8,19:8,20:8 run
echo -n "plan: N=17..20 at 8, no priorities  : "; MCQ_JOB_PRIORITY=0 MCQ_LANES_PLAN=17:8,18:8,19:8,20:8 run
done
} | tee $ROOT/$OUT
