#!/bin/bash
# Replica exchange (not a mode of the reference) against plain annealing at equal moves, BASELINE configs[1] shape: min / mean best
# energy and moves/s.  Run on the GPU box: tools/exchange_evidence.sh > gpurun_out/r03_exchange.jsonl
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
J="python $ROOT/tools/last_json_line.py"
run() { python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" | $J; }
run
for lad in 0.7,1.4 0.85,1.2 1.0,1.5 0.5,1.0; do
  for k in 64 1000; do run --exchange $k --ladder $lad; done
done
run --exchange 1 --ladder 0.7,1.4
run --exchange 64 --ladder 0.7,1.4 --replicas 4
run --config c3
run --config c3 --exchange 64 --ladder 0.7,1.4 --replicas 8
python $ROOT/bench.py --steps 1 --warmup 0 --exchange 64 --ladder 0.7,1.4 --cpu-chains 256 | $J
