#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV passes written by tools/pmc_collect.sh: per-kernel counter sums."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main(root):
    out = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(lambda: defaultdict(int))
    for path in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                k = row["Kernel_Name"]
                k = "sweep" if "mcq_sweep" in k else "init" if "mcq_init" in k else "beta" if "mcq_beta" in k else None
                if k is None:
                    continue
                out[k][row["Counter_Name"]] += float(row["Counter_Value"])
                calls[k][row["Counter_Name"]] += 1
    res = {k: {c: v / max(1, calls[k][c]) * 1.0 for c, v in d.items()} for k, d in out.items()}  # per dispatch
    for k, d in calls.items():  # dispatches of the kernel in one pass (job lists launch several sweep kernels per bench step)
        res[k]["_dispatches"] = min(d.values()) if d else 0  # (a counter may be collected in two passes: the smallest count is one pass)
    print(json.dumps(res, indent=1, sort_keys=True))


if __name__ == "__main__":
    main(sys.argv[1])
