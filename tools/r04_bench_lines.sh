#!/bin/bash
# The bench lines of the final kernel with the PMC entries of the same kernel source (profiles/hbm_traffic.json refreshed first), the
# rocprofv3 kernel-trace summary of the same command, and the whole GPU suite.   tools/r04_bench_lines.sh -> gpurun_out/r04_lines/
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_lines
mkdir -p $OUT
cd $ROOT
python bench.py > $OUT/bench.json 2> $OUT/bench.err && tail -c 200 $OUT/bench.json
( cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kernel_stats -- python $ROOT/bench.py --no-cpu-baseline > $OUT/kernel_stats_bench.json 2> $OUT/kernel_stats.err )
for c in c3 c4 c5; do python bench.py --config $c --no-cpu-baseline | python tools/last_json_line.py; done > $OUT/configs_pmc.jsonl
timeout -k 10 900 python -m pytest tests -m gpu -q > $OUT/suite.log 2>&1; tail -3 $OUT/suite.log
