#!/usr/bin/env python3
"""Loops of a gfx950 assembly listing (hipcc -S): every backward branch with the number of instructions of each class
between its target and itself.  Static counts of the code a loop trip walks through (both sides of the branches inside).
usage: asm_loops.py file.s [min_instructions]"""
import re, sys
lines = open(sys.argv[1]).read().split("\n")
minimum = int(sys.argv[2]) if len(sys.argv) > 2 else 8
labels = {}
for i, l in enumerate(lines):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m: labels[m.group(1)] = i
def classify(op):
    if op.startswith("v_"): return "valu"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "scratch_", "flat_")): return "vmem"
    return None
for i, l in enumerate(lines):
    m = re.match(r"^\s+(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)", l)
    if not m or m.group(2) not in labels or labels[m.group(2)] > i: continue
    start = labels[m.group(2)]
    count = {"valu": 0, "salu": 0, "lds": 0, "vmem": 0, "wait": 0}
    for k in range(start, i + 1):
        mm = re.match(r"^\s+([a-z_0-9]+)", lines[k])
        if mm and classify(mm.group(1)): count[classify(mm.group(1))] += 1
    total = sum(count.values())
    if total >= minimum:
        print("loop %s lines %d-%d: %d instructions  %s" % (m.group(2), start + 1, i + 1, total, count))
