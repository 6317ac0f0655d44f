#!/bin/bash
# Refresh the PMC evidence bench.py quotes (profiles/hbm_traffic.json) for the CURRENT kernel source.  Run on the GPU box:
#   tools/pmc_refresh.sh gpurun_out/rNN_pmc            (headline workload: N=12 board, 65 536 chains x 100 000 steps, i32 trace)
# then, back in the build container:  python tools/pmc_refresh.py gpurun_out/rNN_pmc board_N12_c65536_s100000
# Each counter group is its own rocprofv3 pass (never combined with other trace domains).
set -e
OUT=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
ARGS="${@:---steps 1 --warmup 0 --no-cpu-baseline}"
mkdir -p $ROOT/$OUT
sha256sum $ROOT/monte-carlo-collective_amd/csrc/mcq_hip.hip | cut -d' ' -f1 > $ROOT/$OUT/kernel_sha256.txt
echo "$ARGS" > $ROOT/$OUT/bench_args.txt
cd /tmp && export TMPDIR=/tmp
pass() {
  name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $ROOT/$OUT/$name -- python $ROOT/bench.py $ARGS > $ROOT/$OUT/$name.json 2> $ROOT/$OUT/$name.err || { tail -5 $ROOT/$OUT/$name.err; return 1; }
  echo "pass $name done"
}
pass insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES
pass waits SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_BUSY_CYCLES
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE
pass rdreq TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
pass wrreq TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum
pass hit TCC_HIT_sum TCC_MISS_sum TCC_READ_sum TCC_WRITE_sum
pass fetch FETCH_SIZE
pass write WRITE_SIZE
python $ROOT/tools/pmc_summary.py $ROOT/$OUT > $ROOT/$OUT/summary.json
