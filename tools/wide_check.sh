#!/bin/bash
# usage (GPU box): tools/wide_check.sh [pytest -k expression] -- a part of the parity suite while a feature is being built
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "${1:-beyond_32 or golden_chains or queen_counts}" > gpurun_out/wide_check.log 2>&1
rc=$?
tail -30 gpurun_out/wide_check.log
exit $rc
