#!/bin/bash
# Evidence for the N = 24 board kernel of BASELINE configs[4] (8 lanes, 3 probe passes, reduced trace: mcq_sweep_kernel<0,8,false,3,true,false,24>),
# which had none under profiles/ in round 2: stamp shares, instruction mix and waits (PMC), occupancy sweep.  Run on the GPU box:
#   tools/n24_evidence.sh gpurun_out/r03_n24
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
case $1 in /*) OUT=$1;; *) OUT=$ROOT/$1;; esac
mkdir -p $OUT
A="--N 24 --trace reduced --schedule sinusoidal_annealing --lanes 8 --no-states"
cd $ROOT
tools/stamp_profile.sh $A 2>&1 | grep STAMP | sed 's/^/N24 G8 reduced /' | tee $OUT/stamps.txt
for c in 8192 16384 32768 65536 131072; do
  python bench.py $A --chains $c --n-steps 20000 --steps 2 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('N=24 board reduced chains', $c, 'wavefronts per SIMD', $c / 8 / 1024, 'moves/s %.4e' % d['value'], 'sweep_ms %.2f' % d['kernel_ms']['sweep'])"
done | tee $OUT/occupancy.txt
tools/pmc_refresh.sh ${OUT#$ROOT/}/pmc --steps 1 --warmup 0 --no-cpu-baseline $A > $OUT/pmc.log 2>&1; tail -3 $OUT/pmc.log
