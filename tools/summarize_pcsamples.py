#!/usr/bin/env python3
"""Aggregate rocprofv3 PC-sampling CSV output by source file:line (needs a -gline-tables-only build)."""
import csv, glob, os, re, sys, collections
d = sys.argv[1]
files = glob.glob(os.path.join(d, "**", "*pc_sampling*.csv"), recursive=True)
by_line = collections.Counter(); by_inst = collections.Counter(); total = 0
for f in files:
    with open(f, newline="") as fh:
        rd = csv.DictReader(fh)
        print("columns:", rd.fieldnames)
        for row in rd:
            total += 1
            inst = row.get("Instruction", "") or ""
            com = row.get("Instruction_Comment", "") or ""
            by_inst[inst.split(" ")[0]] += 1
            m = re.search(r"([\w_]+\.(?:h|hip|cpp)):(\d+)", com)
            by_line[(m.group(1), int(m.group(2))) if m else ("?", 0)] += 1
print("total samples", total)
print("--- top opcodes")
for k, v in by_inst.most_common(80): print("%8d %5.1f%% %s" % (v, 100.0 * v / max(total, 1), k))
print("--- top source lines")
for k, v in by_line.most_common(120): print("%8d %5.1f%% %s:%d" % (v, 100.0 * v / max(total, 1), k[0], k[1]))
by_file = collections.Counter()
for (f, l), v in by_line.items(): by_file[f] += v
print("--- by file")
for k, v in by_file.most_common(): print("%8d %5.1f%% %s" % (v, 100.0 * v / max(total, 1), k))
