#!/bin/bash
# Collect rocprofv3 PMC counters for the sweep kernel in separate passes (run on the GPU box).
# usage: tools/pmc_collect.sh <outdir> [bench.py args...]
set -e
OUT=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
pass() {
  name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $ROOT/$OUT/$name -- python $ROOT/bench.py $BENCH_ARGS > $ROOT/$OUT/$name.json 2> $ROOT/$OUT/$name.err || { tail -5 $ROOT/$OUT/$name.err; return 1; }
}
mkdir -p $ROOT/$OUT
export BENCH_ARGS="${@:---steps 1 --warmup 0 --n-steps 20000 --no-cpu-baseline}"
pass insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES
pass waits SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_BUSY_CYCLES
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE
pass fetch FETCH_SIZE
pass write WRITE_SIZE
