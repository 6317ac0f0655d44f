#!/bin/bash
# Evidence that is stamped with the sha256 of csrc/mcq_hip.hip: run on the GPU box AFTER the last kernel edit of a round.
#   tools/finalize.sh gpurun_out/rNN_final
# then, in the build container:
#   python tools/pmc_refresh.py gpurun_out/rNN_final/pmc_c2 board_N12_c65536_s100000 profiles/rNN_pmc_summary.json      (and the other keys, see below)
#   cp gpurun_out/rNN_final/lane_table.json monte-carlo-collective_amd/lane_table.json
OUT=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/$OUT
python $ROOT/tools/lane_table.py --json $ROOT/$OUT/lane_table.json > $ROOT/$OUT/lane_table.txt 2>&1 && echo "lane table done" || { tail -3 $ROOT/$OUT/lane_table.txt; exit 1; }
$ROOT/tools/pmc_refresh.sh $OUT/pmc_c2 || exit 1
$ROOT/tools/pmc_refresh.sh $OUT/pmc_c3 --config c3 --steps 1 --warmup 0 --no-cpu-baseline || exit 1
$ROOT/tools/pmc_refresh.sh $OUT/pmc_c4 --config c4 --steps 1 --warmup 0 --no-cpu-baseline || exit 1
$ROOT/tools/pmc_refresh.sh $OUT/pmc_c5 --config c5 --steps 1 --warmup 0 --no-cpu-baseline || exit 1
echo "finalize done"
