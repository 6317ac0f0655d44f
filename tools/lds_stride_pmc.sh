#!/bin/bash
# LDS bank-conflict counters of the headline sweep (20 000 steps) for the shipped LDS stride (4 mod 8 words per chain) and for the
# conflicting one (0 mod 8: build/libmcq_hip_stride0.so from tools/exp_build.sh --build-only stride0 "-DMCQ_EXP_LDS_STRIDE_0MOD8").
# usage (GPU box): tools/lds_stride_pmc.sh OUTDIR
OUT=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in shipped stride0; do
  if [ $v = stride0 ]; then export MCQ_ALLOW_DIAG=1 MCQ_DIAG_LIB=$ROOT/build/libmcq_hip_stride0.so; fi
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/lds_$v/lds -- python $ROOT/bench.py --steps 1 --warmup 0 --n-steps 20000 --no-cpu-baseline > $OUT/lds_$v.json 2> $OUT/lds_$v.err
  python $ROOT/tools/pmc_summary.py $OUT/lds_$v | python -c "
import sys, json
d = json.load(sys.stdin)['sweep']
print('$v', {k: d[k] for k in sorted(d)}, 'conflict share %.3f' % (d['SQ_LDS_BANK_CONFLICT'] / d['SQ_LDS_IDX_ACTIVE']))"
done
