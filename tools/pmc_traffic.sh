#!/bin/bash
# Memory-side accounting of the sweep kernel (run on the GPU box): L2 (TCC) hits / misses, fabric read requests by size
# (32 / 64 / 128 B), write requests, DRAM-destined requests, L1 (TCP) -> L2 requests.  Each group in its own rocprofv3 pass.
# usage: tools/pmc_traffic.sh <outdir> [bench.py args...]       (summary: python tools/pmc_summary.py <outdir>)
set -e
OUT=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
mkdir -p $ROOT/$OUT
BENCH_ARGS="${@:---steps 1 --warmup 0 --n-steps 20000 --no-cpu-baseline}"
pass() {
  name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $ROOT/$OUT/$name -- python $ROOT/bench.py $BENCH_ARGS > $ROOT/$OUT/$name.json 2> $ROOT/$OUT/$name.err || { tail -5 $ROOT/$OUT/$name.err; return 1; }
  echo "pass $name done"
}
pass rdreq TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
pass wrreq TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum
pass hit TCC_HIT_sum TCC_MISS_sum TCC_READ_sum TCC_WRITE_sum
pass sectors TCC_REQ_sum TCC_READ_SECTORS_sum TCC_WRITE_SECTORS_sum TCC_WRITEBACK_sum
pass tcp TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum
pass fetch FETCH_SIZE
pass write WRITE_SIZE
