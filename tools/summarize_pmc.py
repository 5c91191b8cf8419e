#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter CSVs per (kernel, counter) for the compression kernel."""
import csv, glob, os, sys
from collections import defaultdict
tot = defaultdict(float); n = defaultdict(int)
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "")
            if "astc_compress" not in k:
                continue
            tot[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
for c in sorted(tot):
    print("%-24s sum=%.6g over %d dispatch records (per dispatch %.6g)" % (c, tot[c], n[c], tot[c] / max(n[c], 1)))
