#!/bin/bash
# c4 at 1 024 chains per cell: balancing the launches of a job list -- FEWER lanes (2: half the wavefronts, a longer step) for the short launches, MORE for the long ones.
OUT=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
run() { python $ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 1 --config c4 "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%.4e moves/s  step %.2f ms  sweeps %.2f ms' % (d['value'], d['ms_per_step'], d['kernel_ms']['sweeps']))"; }
S2="3:2,4:2,5:2,6:2,7:2,8:2"; M2="9:2,10:2,11:2,12:2"; H8="17:8,18:8,19:8,20:8"; H16="17:16,18:16,19:16,20:16"
{
true
for r in 1 2; do
echo -n "A 4 lanes everywhere                      : "; run
echo -n "B N=3..8 at 2, N=17..20 at 8              : "; MCQ_LANES_PLAN=$S2,$H8 run
echo -n "C N=3..12 at 2, N=17..20 at 8             : "; MCQ_LANES_PLAN=$S2,$M2,$H8 run
echo -n "F N=3..8 at 2 only                        : "; MCQ_LANES_PLAN=$S2 run
echo -n "G N=3..12 at 2, N=17..20 at 16            : "; MCQ_LANES_PLAN=$S2,$M2,$H16 run
echo -n "H N=3..12 at 2, N=13..16 at 8, 17..20 at 8: "; MCQ_LANES_PLAN=$S2,$M2,13:8,14:8,15:8,16:8,$H8 run
done
} 2>&1 | grep -v amdgpu.ids | tee $ROOT/$OUT
