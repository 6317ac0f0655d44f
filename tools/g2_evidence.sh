#!/bin/bash
# Two lanes per chain (32 chains per wavefront) against four on BASELINE configs[1]'s shape: occupancy sweep and instruction mix.
# usage (GPU box): tools/g2_evidence.sh OUTDIR      -> OUTDIR/occupancy.txt, OUTDIR/pmc_g{2,4}.txt
OUT=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/$OUT
for lanes in 4 2; do
  for chains in 16384 32768 65536 131072; do
    python $ROOT/bench.py --steps 2 --warmup 1 --n-steps 20000 --no-cpu-baseline --lanes $lanes --chains $chains 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('board N=12 lanes $lanes chains $chains moves/s %.4e sweep_ms %.2f' % (d['value'], d['kernel_ms']['sweep']))" | tee -a $ROOT/$OUT/occupancy.txt
  done
done
for lanes in 4 2; do
  $ROOT/tools/pmc_quick.sh $OUT/pmc_g$lanes --lanes $lanes | tee $ROOT/$OUT/pmc_g$lanes.txt
done
