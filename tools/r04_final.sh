#!/bin/bash
# The round's final evidence in ONE session on ONE box, after the last kernel edit (boxes differ by a few per cent).  On the GPU box:
#   tools/r04_final.sh r04        -> gpurun_out/r04_final2/...
R=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${R}_final2
mkdir -p $OUT
cd $ROOT
J="python $ROOT/tools/last_json_line.py"
step() { echo "=== $1"; shift; timeout -k 10 ${STEP_TIMEOUT:-700} bash -o pipefail -c "$*"; rc=$?; echo "=== rc=$rc"; if [ $rc -eq 124 ] || [ $rc -ge 128 ]; then echo "killed: stopping"; exit $rc; fi; }

step "stamped evidence: lane table + PMC of c2 / c3 / c4 / c5" "tools/finalize.sh gpurun_out/${R}_final2"
step "headline bench (with cpu baseline)" "python bench.py > $OUT/bench.json && tail -c 300 $OUT/bench.json"
step "kernel trace stats of the same command" "cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kernel_stats -- python $ROOT/bench.py --no-cpu-baseline > $OUT/kernel_stats_bench.json 2> $OUT/kernel_stats.err; ls $OUT/kernel_stats/*/ | head"
step "configs" "( python bench.py --config c3 --no-cpu-baseline | $J; python bench.py --config c3 --lanes 8 --no-cpu-baseline | $J; python bench.py --config c4 --no-cpu-baseline | $J; python bench.py --config c4 --chains 8192 --no-cpu-baseline | $J; python bench.py --config c5 --no-cpu-baseline | $J; python bench.py --config c5 --chains 8192 --no-cpu-baseline | $J; python bench.py --trace reduced --n-steps 1000000 --steps 1 --no-cpu-baseline | $J; python bench.py --trace none --no-cpu-baseline | $J; python bench.py --rng philox --no-cpu-baseline | $J; python bench.py --lanes 2 --no-cpu-baseline | $J ) > $OUT/configs.jsonl; cut -c1-160 $OUT/configs.jsonl"
step "occupancy sweep, full_3d (slim)" "for c in 8192 16384 32768 65536 131072; do python bench.py --config c3 --chains \$c --n-steps 20000 --steps 2 --no-cpu-baseline | python -c \"import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('full_3d chains', \$c, 'moves/s %.4e' % d['value'], 'sweep_ms %.2f' % d['kernel_ms']['sweep'], 'lanes', d['config']['lanes_per_chain'])\"; done | tee $OUT/occupancy_full3d.txt"
step "full_3d beyond N = 32" "python tools/wide_timing.py | tee $OUT/full3d_wide.txt"
step "one-shot driver call against the steady state (config 5's per-GPU shape)" "python tools/c5_oneshot.py | tee $OUT/c5_oneshot.txt"
step "c5 host memory" "python -c \"
import resource, subprocess, sys
p = subprocess.run([sys.executable, 'bench.py', '--config', 'c5', '--chains', '8192', '--steps', '1', '--no-cpu-baseline'], capture_output=True, text=True)
print('c5 16 x 8192 x 1e5: child max RSS MB', resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss / 1024, 'rc', p.returncode)
\" | tee $OUT/c5_rss.txt"
echo "=== evidence complete"
