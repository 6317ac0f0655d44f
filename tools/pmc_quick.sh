#!/bin/bash
# Instruction mix, wait breakdown and LDS conflicts of one bench.py workload (three rocprofv3 passes).  usage: tools/pmc_quick.sh OUTDIR [bench args]
set -e
OUT=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/$OUT
ARGS="--steps 1 --warmup 0 --n-steps 20000 --no-cpu-baseline $@"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $ROOT/$OUT/insts -- python $ROOT/bench.py $ARGS > $ROOT/$OUT/insts.json 2> $ROOT/$OUT/insts.err
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $ROOT/$OUT/waits -- python $ROOT/bench.py $ARGS > $ROOT/$OUT/waits.json 2> $ROOT/$OUT/waits.err
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $ROOT/$OUT/lds -- python $ROOT/bench.py $ARGS > $ROOT/$OUT/lds.json 2> $ROOT/$OUT/lds.err
python $ROOT/tools/pmc_summary.py $ROOT/$OUT | python -c "
import sys,json
d=json.load(sys.stdin)['sweep']
w=d['SQ_WAVES']
print('per wavefront-step (20000 steps):', {k: round(v/w/20000,2) for k,v in sorted(d.items()) if k!='SQ_WAVES'})"
