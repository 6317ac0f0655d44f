#!/bin/bash
# c4 at the per-GPU shape: static job priorities (default) / no priorities / launches pacing each other through one shared progress table / CU partition
OUT=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
run() { python $ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%.4e moves/s  step %.2f ms  sweeps %.2f ms  min %s' % (d['value'], d['ms_per_step'], d['kernel_ms']['sweeps'], d['min_energy']))"; }
{
for r in 1 2; do
echo -n "c4 1024 per cell, static priorities : "; run --config c4
echo -n "c4 1024 per cell, shared pacing     : "; MCQ_JOB_PACING=1 run --config c4
echo -n "c4 1024 per cell, neither           : "; MCQ_JOB_PRIORITY=0 run --config c4
done
echo -n "c4 512 per cell, static priorities : "; run --config c4 --chains 512
echo -n "c4 512 per cell, shared pacing     : "; MCQ_JOB_PACING=1 run --config c4 --chains 512
echo -n "c4 512 per cell, CU partition      : "; MCQ_CU_PARTITION=1 run --config c4 --chains 512
echo -n "c4 2048 per cell, static priorities: "; run --config c4 --chains 2048
echo -n "c4 2048 per cell, shared pacing    : "; MCQ_JOB_PACING=1 run --config c4 --chains 2048
} 2>&1 | grep -v amdgpu.ids | tee $ROOT/$OUT
