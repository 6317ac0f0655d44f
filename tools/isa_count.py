#!/usr/bin/env python3
"""Count instructions per class in one kernel of a hipcc -save-temps .s file (whole function: prologue, rare paths and
all; use it to compare two builds, not as a per-step count).  usage: isa_count.py file.s [mangled-name-substring]"""
import collections
import re
import sys

path = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else "mcq_sweep_kernelILi0ELi4ELb0ELi3ELb0"
inside = False
c = collections.Counter()
for line in open(path):
    if re.match(r"^_Z\w+:", line):
        inside = want in line
        continue
    if inside and line.startswith(".Lfunc_end"):
        break
    if not inside:
        continue
    m = re.match(r"^\s+([a-z_0-9]+)\s", line)
    if not m:
        continue
    op = m.group(1)
    if op.startswith("v_"):
        c["valu"] += 1
    elif op.startswith("s_load") or op.startswith("s_buffer"):
        c["smem"] += 1
    elif op.startswith("s_waitcnt"):
        c["waitcnt"] += 1
    elif op.startswith("s_cbranch") or op.startswith("s_branch"):
        c["branch"] += 1
    elif op.startswith("s_"):
        c["salu"] += 1
    elif op.startswith("ds_"):
        c["lds"] += 1
    elif op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_"):
        c["vmem"] += 1
    else:
        c["other"] += 1
print(dict(c), "total", sum(c.values()))
