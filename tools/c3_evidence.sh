#!/bin/bash
# full_3d (BASELINE configs[2]) at 8 lanes per chain against the slim 4-lane layout: instruction mix per wavefront-step.
# usage (GPU box): tools/c3_evidence.sh OUTDIR
OUT=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/$OUT
for lanes in 8 4; do
  $ROOT/tools/pmc_quick.sh $OUT/pmc_g$lanes --config c3 --lanes $lanes | tee $ROOT/$OUT/pmc_g$lanes.txt
done
