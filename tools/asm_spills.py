#!/usr/bin/env python3
"""For every VGPR spill store in a function of an assembly listing built with -g1: the instruction that produced the spilled
value and the source line it came from.  usage: asm_spills.py file.s function-substring"""
import re, sys
lines = open(sys.argv[1]).read().split("\n")
fn = sys.argv[2]
files = {}
start = end = None
for i, l in enumerate(lines):
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m: files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
    if re.match(r"^_Z\w*%s\w*:" % fn, l): start = i
    if start is not None and end is None and re.match(r"^\.Lfunc_end", l): end = i
body = lines[start:end]
def loc_at(k):
    for j in range(k, -1, -1):
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", body[j])
        if m:
            cm = re.search(r";\s*(.*)$", body[j])
            return cm.group(1) if cm else "%s:%s" % (files.get(int(m.group(1)), "?"), m.group(2))
    return "?"
for k, l in enumerate(body):
    m = re.search(r"scratch_store_dword off, (v\d+), off(?: offset:(\d+))?", l)
    if not m: continue
    reg, off = m.group(1), m.group(2) or "0"
    d = None
    for j in range(k - 1, max(0, k - 400), -1):
        if re.match(r"\s+[a-z_0-9]+\s+%s\b" % reg, body[j]) or re.search(r"\[%s:|:%s\]" % (reg[1:], reg[1:]), body[j].split(",")[0]):
            d = j; break
    print("slot %3s  %-60s | def: %-50s @ %s" % (off, loc_at(k)[:60], body[d].strip()[:50] if d is not None else "?", loc_at(d)[:70] if d is not None else "?"))
