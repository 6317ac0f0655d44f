#!/bin/bash
# c4 at the per-GPU shapes: groups of launches on compute-unit layers of their own (MCQ_CU_PARTITION=1), with and without the launches pacing each other.
OUT=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
run() { python $ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%.4e moves/s  step %.2f ms  sweeps %.2f ms  groups %s' % (d['value'], d['ms_per_step'], d['kernel_ms']['sweeps'], d['config'].get('cu_partition')))"; }
{
for c in 1024 512; do
for r in 1 2; do
echo -n "c4 $c per cell, shared pacing                 : "; run --config c4 --chains $c
echo -n "c4 $c per cell, shared pacing + CU layers     : "; MCQ_CU_PARTITION=1 run --config c4 --chains $c
echo -n "c4 $c per cell, static priorities + CU layers : "; MCQ_CU_PARTITION=1 MCQ_JOB_PACING=0 run --config c4 --chains $c
done; done
} 2>&1 | grep -v amdgpu.ids | tee $ROOT/$OUT
