#!/bin/bash
# c4 / c5 at the per-GPU shapes with and without a CU partition of the job list's launches (MCQ_CU_PARTITION=0 switches it off).
OUT=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
run() { python $ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%.4e moves/s  step %.2f ms  sweeps %.2f ms  min %s  cus %s' % (d['value'], d['ms_per_step'], d['kernel_ms']['sweeps'], d['min_energy'], d['config'].get('cu_partition')))"; }
{
for r in 1 2; do
echo -n "c4 1024 per cell, CU partition on : "; run --config c4
echo -n "c4 1024 per cell, CU partition off: "; MCQ_CU_PARTITION=0 run --config c4
done
echo -n "c4 512 per cell, on : "; run --config c4 --chains 512
echo -n "c4 512 per cell, off: "; MCQ_CU_PARTITION=0 run --config c4 --chains 512
echo -n "c4 8192 per cell (launches too large: no partition) : "; run --config c4 --chains 8192
} 2>&1 | grep -v amdgpu.ids | tee $ROOT/$OUT
