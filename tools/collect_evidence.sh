#!/bin/bash
# Everything profiles/rNN_* is made of, in ONE session on ONE box (boxes differ by a few per cent).  Run on the GPU box:
#   tools/collect_evidence.sh r02          -> gpurun_out/r02_evidence/...   (then tools/evidence_to_profiles.py r02 in the build container)
# Every rocprofv3 counter group is its own pass; --pmc is never combined with other trace domains.
R=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${R}_evidence
mkdir -p $OUT
cd $ROOT
J="python $ROOT/tools/last_json_line.py"
step() { echo "=== $1"; shift; timeout -k 10 ${STEP_TIMEOUT:-600} bash -o pipefail -c "$*"; rc=$?; echo "=== rc=$rc"; if [ $rc -eq 124 ] || [ $rc -ge 128 ]; then echo "killed: stopping"; exit $rc; fi; }

step "headline bench (with cpu baseline)" "python bench.py > $OUT/bench.json && tail -c 600 $OUT/bench.json"
step "kernel trace stats of the same command" "cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kernel_stats -- python $ROOT/bench.py --no-cpu-baseline > $OUT/kernel_stats_bench.json 2> $OUT/kernel_stats.err; ls $OUT/kernel_stats/*/ | head"
step "PMC refresh, headline" "tools/pmc_refresh.sh gpurun_out/${R}_evidence/pmc_headline"
step "PMC, full_3d (config 3)" "tools/pmc_refresh.sh gpurun_out/${R}_evidence/pmc_c3 --steps 1 --warmup 0 --no-cpu-baseline --config c3"
step "PMC, philox" "tools/pmc_refresh.sh gpurun_out/${R}_evidence/pmc_philox --steps 1 --warmup 0 --no-cpu-baseline --rng philox"
step "PMC, N = 24 board (config 5's kernel: 8 lanes, reduced trace, no state outputs)" "tools/pmc_refresh.sh gpurun_out/${R}_evidence/pmc_n24 --steps 1 --warmup 0 --no-cpu-baseline --N 24 --trace reduced --schedule sinusoidal_annealing --lanes 8 --no-states"
step "configs" "( python bench.py --config c3 --no-cpu-baseline | $J; python bench.py --config c4 --no-cpu-baseline | $J; python bench.py --config c4 --chains 8192 --no-cpu-baseline | $J; python bench.py --config c5 --no-cpu-baseline | $J; python bench.py --config c5 --chains 8192 --no-cpu-baseline | $J; python bench.py --trace reduced --n-steps 1000000 --steps 1 --no-cpu-baseline | $J; python bench.py --trace none --no-cpu-baseline | $J; python bench.py --rng philox --no-cpu-baseline | $J; python bench.py --rng philox --config c3 --no-cpu-baseline | $J ) > $OUT/configs.jsonl; cut -c1-160 $OUT/configs.jsonl"
step "patience" "python tools/bench_patience.py > $OUT/patience.txt; cat $OUT/patience.txt"
step "occupancy sweep, board" "for c in 16384 32768 65536 131072; do python bench.py --chains \$c --n-steps 20000 --steps 2 --no-cpu-baseline | python -c \"import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('board chains', \$c, 'moves/s %.4e' % d['value'], 'sweep_ms %.2f' % d['kernel_ms']['sweep'])\"; done | tee $OUT/occupancy_board.txt"
step "occupancy sweep, full_3d" "for c in 8192 16384 32768 65536 131072; do python bench.py --config c3 --chains \$c --n-steps 20000 --steps 2 --no-cpu-baseline | python -c \"import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('full_3d chains', \$c, 'moves/s %.4e' % d['value'], 'sweep_ms %.2f' % d['kernel_ms']['sweep'])\"; done | tee $OUT/occupancy_full3d.txt"
step "stamps" "(tools/stamp_profile.sh 2>&1 | grep STAMP | sed 's/^/board   /'; tools/stamp_profile.sh --config c3 2>&1 | grep STAMP | sed 's/^/full_3d /'; tools/stamp_profile.sh --N 24 --trace reduced --schedule sinusoidal_annealing --lanes 8 --no-states 2>&1 | grep STAMP | sed 's/^/N24     /') | tee $OUT/stamps.txt"
step "occupancy sweep, N = 24 board (8 lanes, reduced trace)" "for c in 8192 16384 32768 65536 131072; do python bench.py --N 24 --trace reduced --schedule sinusoidal_annealing --lanes 8 --no-states --chains \$c --n-steps 20000 --steps 2 --no-cpu-baseline | python -c \"import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('N=24 board chains', \$c, 'moves/s %.4e' % d['value'], 'sweep_ms %.2f' % d['kernel_ms']['sweep'])\"; done | tee $OUT/occupancy_n24.txt"
step "config 4 at SURVEY 8d's 10^6 steps (8 192 chains per cell, no trace)" "python bench.py --config c4 --chains 8192 --n-steps 1000000 --steps 1 --warmup 0 --no-cpu-baseline | $J | tee $OUT/c4_1e6.json | cut -c1-300"
step "small launches vs hardware queues" "for q in 4 8 16 24 32; do GPU_MAX_HW_QUEUES=\$q python bench.py --config c4 --no-cpu-baseline --steps 2 | python -c \"import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c4 1024 chains/cell, GPU_MAX_HW_QUEUES', \$q, 'moves/s %.4e' % d['value'], d['kernel_ms'], 'launches', d['config']['launches_per_rank'])\"; done | tee $OUT/small_launches.txt"
step "LDS stride A/B (same-offset accesses of the 8 chains of an access group: stride 4 mod 8 words vs 0 mod 8)" "ROUNDS=2 tools/ab.sh '--steps 2 --warmup 1' shipped build/libmcq_hip_stride0.so | tee $OUT/lds_stride_ab.txt"
step "LDS conflict counters for both strides" "tools/lds_stride_pmc.sh $OUT | tee $OUT/lds_stride_pmc.txt"
step "c5 host memory" "python -c \"
import resource, subprocess, sys
p = subprocess.run([sys.executable, 'bench.py', '--config', 'c5', '--chains', '8192', '--steps', '1', '--no-cpu-baseline'], capture_output=True, text=True)
print('c5 16 x 8192 x 1e5: child max RSS MB', resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss / 1024, 'rc', p.returncode)
\" | tee $OUT/c5_rss.txt"
echo "=== evidence complete"
