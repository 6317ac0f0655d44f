#!/usr/bin/env python3
"""Print value / kernel_ms / lanes of the last JSON line on stdin (a bench.py run)."""
import json
import sys

d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print("%.4e moves/s" % d["value"], d["kernel_ms"], d["config"].get("lanes_per_chain"), flush=True)
