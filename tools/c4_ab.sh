#!/bin/bash
# A/B on one box: config 4's per-GPU shape with and without the stream-word count in its launches (twice each, alternating)
J="python tools/last_json_line.py"
for r in 1 2; do
  for v in 0 1; do
    MCQ_JOB_STREAM_WORDS=$v python bench.py --config c4 --no-cpu-baseline | $J | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('stream_words=$v', '%.4e' % d['value'], round(d['ms_per_step'],1), d['kernel_ms'])"
  done
done
python bench.py --no-cpu-baseline | $J | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('headline', '%.4e' % d['value'], round(d['ms_per_step'],1), d['kernel_ms'])"
