#!/bin/bash
# After tools/r04_final.sh <tag> on the GPU box: bring the stamped evidence of gpurun_out/<tag>_final2 into the tree (build container).
#   tools/refresh_local.sh r04g
F=gpurun_out/${1}_final2
set -e
cp $F/lane_table.json monte-carlo-collective_amd/lane_table.json
python tools/pmc_refresh.py $F/pmc_c2 board_N12_c65536_s100000 profiles/r04_pmc_summary_c2.json > /dev/null
python tools/pmc_refresh.py $F/pmc_c3 full_3d_N12_c65536_s100000 profiles/r04_pmc_summary_c3.json > /dev/null
python tools/pmc_refresh.py $F/pmc_c4 c4_c1024_s100000 profiles/r04_pmc_summary_c4.json > /dev/null
python tools/pmc_refresh.py $F/pmc_c5 c5_c1024_s100000 profiles/r04_pmc_summary_c5.json > /dev/null
cp $F/configs.jsonl profiles/r04_configs.jsonl
cp $F/occupancy_full3d.txt profiles/r04_occupancy_full3d.txt
cp $F/lane_table.txt profiles/r04_lane_table.txt
cp $F/c5_oneshot.txt profiles/r04_c5_oneshot.txt
cp $F/full3d_wide.txt profiles/r04_full3d_wide.txt
python tools/register_table.py > profiles/r04_registers.txt
echo refreshed from $F
