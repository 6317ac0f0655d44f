#!/bin/bash
# Round 4, review item 3: the shapes one GPU of an 8-GPU node runs (BASELINE configs[3] / [4] at 1 024 chains per cell / pair).
OUT=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
run() { python $ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%.4e moves/s  step %.2f ms  kernels %s  lanes %s' % (d['value'], d['ms_per_step'], d['kernel_ms'], d['config'].get('lanes_per_chain')))"; }
{
echo "# c5 (N = 24, 16 pairs, reduced trace), 1 024 chains per pair: library plan / 8 lanes / 16 lanes (unrolled since round 4)"
for l in 0 8 16; do echo -n "c5 1024 lanes=$l: "; run --config c5 --lanes $l; done
echo "# c5, 8 192 chains per pair"
for l in 0 8 16; do echo -n "c5 8192 lanes=$l: "; run --config c5 --chains 8192 --lanes $l; done
echo "# c4 (Ns 3..20 x 3 inits, no trace), 1 024 chains per cell: job priorities on / off, compile-time-N five-pass kernels on / off"
for r in 1 2; do
echo -n "c4 1024 priorities on : "; run --config c4
echo -n "c4 1024 priorities off: "; MCQ_JOB_PRIORITY=0 run --config c4
echo -n "c4 1024 priorities on, generic N=17,18,20: "; MCQ_ALLOW_DIAG=1 MCQ_DIAG_LIB=$ROOT/build/libmcq_hip_nonc5.so run --config c4
done
echo "# c4, 8 192 chains per cell"
echo -n "c4 8192 priorities on : "; run --config c4 --chains 8192
echo -n "c4 8192 priorities off: "; MCQ_JOB_PRIORITY=0 run --config c4 --chains 8192
} | tee $ROOT/$OUT
