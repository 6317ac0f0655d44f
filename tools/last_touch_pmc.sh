#!/bin/bash
# Last-touch experiment: the MT19937 state's write-back as a streaming (nt) store for the block that is the
# SECOND half of its 128-byte line, against plain stores, both with 128-byte aligned records (build/libmcq_hip_{lasttouch,rec128}.so
# from tools/exp_build.sh --build-only NAME "-DMCQ_EXP_LAST_TOUCH" / "-DMCQ_EXP_REC128") and against the shipped library.
# Time: tools/ab.sh.  Traffic: the L2's memory-side requests by size, one rocprofv3 --pmc pass each (headline workload, one launch).
# usage (GPU box): tools/last_touch_pmc.sh OUTDIR
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
case $1 in /*) OUT=$1;; *) OUT=$ROOT/$1;; esac
mkdir -p $OUT
ROUNDS=3 $ROOT/tools/ab.sh '--steps 2 --warmup 1' shipped build/libmcq_hip_rec128.so build/libmcq_hip_lasttouch.so
cd /tmp && export TMPDIR=/tmp
for v in shipped rec128 lasttouch; do
  if [ $v != shipped ]; then export MCQ_ALLOW_DIAG=1 MCQ_DIAG_LIB=$ROOT/build/libmcq_hip_$v.so; fi
  rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace --output-format csv -d $OUT/$v/rdreq -- python $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/$v.rd.json 2> $OUT/$v.rd.err
  rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/$v/wrreq -- python $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/$v.wr.json 2> $OUT/$v.wr.err
  python $ROOT/tools/pmc_summary.py $OUT/$v | python -c "
import sys, json
d = json.load(sys.stdin)['sweep']
moves = 65536 * 100000
rd = 32 * d['TCC_EA0_RDREQ_32B_sum'] + 64 * d['TCC_EA0_RDREQ_64B_sum'] + 128 * d['TCC_EA0_RDREQ_128B_sum']
wr = 64 * d['TCC_EA0_WRREQ_64B_sum'] + 32 * (d['TCC_EA0_WRREQ_sum'] - d['TCC_EA0_WRREQ_64B_sum'])
print('PMC $v read %.1f B/move  write %.1f B/move  total %.1f B/move  L2 hit rate %.3f' % (rd / moves, wr / moves, (rd + wr) / moves, d['TCC_HIT_sum'] / (d['TCC_HIT_sum'] + d['TCC_MISS_sum'])))"
done
