#!/usr/bin/env python3
"""stdin -> the last line that parses as JSON, re-emitted on one line (bench.py prints exactly one; libraries may chatter before it)."""
import json
import sys

last = None
for line in sys.stdin:
    line = line.strip()
    if line.startswith("{"):
        try:
            last = json.loads(line)
        except ValueError:
            pass
if last is None:
    sys.exit(1)
print(json.dumps(last))
