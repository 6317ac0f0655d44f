#!/bin/bash
OUT=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
run() { python $ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%.4e moves/s  step %.2f ms  sweeps %.2f ms  groups %s' % (d['value'], d['ms_per_step'], d['kernel_ms']['sweeps'], 'yes' if d['config'].get('cu_partition') else 'no'))"; }
{
for c in 128 256 512 704 1024; do
echo -n "c4 $c per cell, auto : "; run --config c4 --chains $c
echo -n "c4 $c per cell, off  : "; MCQ_CU_PARTITION=0 run --config c4 --chains $c
done
for c in 128 256 512; do
echo -n "c5 $c per pair, auto (one launch: never) : "; run --config c5 --chains $c
done
} 2>&1 | grep -v amdgpu.ids | tee $ROOT/$OUT
