#!/usr/bin/env python3
"""Instruction histogram of one kernel in a hipcc --save-temps .s file (static counts; a tuning aid).
usage: isa_hist.py file.s kernel-name-substring"""
import collections
import re
import sys

path, pat = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = None
for i, l in enumerate(lines):
    m = re.match(r"^(\S+):\s*;?\s*@?", l)
    if m and pat in m.group(1) and not l.startswith("\t") and not m.group(1).startswith("."):
        start = i
        break
if start is None:
    sys.exit("kernel not found")
hist = collections.Counter()
n = 0
for l in lines[start + 1:]:
    if l.startswith("\t.end_amdhsa_kernel") or l.startswith(".Lfunc_end"):
        break
    s = l.strip()
    if not s or s.startswith(";") or s.startswith(".") or s.endswith(":"):
        continue
    op = s.split()[0]
    hist[op] += 1
    n += 1
groups = collections.Counter()
for op, c in hist.items():
    if op.startswith("v_") and ("f64" in op):
        groups["valu_f64"] += c
    elif op.startswith("v_"):
        groups["valu_other"] += c
    elif op.startswith("s_"):
        groups["salu"] += c
    elif op.startswith("ds_"):
        groups["lds"] += c
    elif op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"):
        groups["vmem"] += c
    else:
        groups["other"] += c
print(lines[start], "total", n)
print(dict(groups))
for op, c in hist.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 40):
    print(f"{c:6d} {op}")
