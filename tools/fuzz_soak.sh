#!/bin/bash
# A longer run of the random parity generators on the shipped kernels (GPU box): every generator, a fresh seed, then the long-chain form.
#   tools/fuzz_soak.sh SEED [scale]      -> gpurun_out/fuzz_soak_SEED.log
SEED=${1:-4242}; K=${2:-10}
mkdir -p gpurun_out
L=gpurun_out/fuzz_soak_$SEED.log
( MCQ_FUZZ_SEED=$SEED MCQ_FUZZ_CASES=$((400*K)) MCQ_FUZZ_R3_CASES=$((140*K)) MCQ_FUZZ_R4_CASES=$((220*K)) MCQ_FUZZ_WIDE_CASES=$((40*K)) MCQ_FUZZ_STREAM_CASES=$((60*K)) timeout -k 10 ${SOAK_TIMEOUT:-500} python -m pytest tests/test_fuzz_parity.py -m gpu -q -p no:cacheprovider -o addopts="" --timeout 1000 2>&1 | tail -3
  MCQ_FUZZ_LONG=1 MCQ_FUZZ_SEED=$((SEED+1)) MCQ_FUZZ_CASES=$((100*K)) MCQ_FUZZ_R3_CASES=$((40*K)) MCQ_FUZZ_R4_CASES=$((60*K)) MCQ_FUZZ_WIDE_CASES=$((10*K)) MCQ_FUZZ_STREAM_CASES=$((15*K)) timeout -k 10 ${SOAK_TIMEOUT:-500} python -m pytest tests/test_fuzz_parity.py -m gpu -q -p no:cacheprovider -o addopts="" --timeout 1000 2>&1 | tail -3 ) > $L 2>&1
cat $L
