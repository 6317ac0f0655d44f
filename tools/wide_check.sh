#!/bin/bash
# usage (GPU box): tools/wide_check.sh -- the parity test of full_3d beyond N = 32, then the neighbours it could have disturbed
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "beyond_32 or golden_chains or queen_counts" > gpurun_out/wide_check.log 2>&1
rc=$?
tail -30 gpurun_out/wide_check.log
exit $rc
