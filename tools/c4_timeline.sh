#!/bin/bash
# Start / end of every sweep kernel of one config-4 step (rocprofv3 --kernel-trace): do the 18 launches of the job list overlap?
OUT=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/$OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $ROOT/$OUT/trace -- python $ROOT/bench.py --config c4 --steps 1 --warmup 1 --no-cpu-baseline "$@" > $ROOT/$OUT/bench.json 2> $ROOT/$OUT/err.txt
python - <<PY
import csv, glob
rows = []
for f in glob.glob("$ROOT/$OUT/trace/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "mcq_sweep" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("mcq_sweep_kernel")[1][:60], r.get("Grid_Size"), r.get("LDS_Block_Size"), r.get("Queue_Id")))
rows.sort()
rows = rows[-18:]  # the timed step
t0 = rows[0][0]
for s, e, k, g, l, q in rows:
    print(f"start {1e-6 * (s - t0):8.2f} ms  end {1e-6 * (e - t0):8.2f} ms  dur {1e-6 * (e - s):8.2f} ms  grid {g} lds {l} queue {q}  {k}")
PY
