#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter CSVs per counter for the compression kernel, and derive the HBM
traffic per launch the way /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes:
FETCH_SIZE and WRITE_SIZE are in KiB and come from separate passes; on gfx950 FETCH_SIZE counts
128-byte read requests as 64 bytes, so the read side is doubled.

usage: summarize_pmc.py [--json traffic.json] DIR...
"""
import csv, glob, json, os, sys
from collections import defaultdict

args = sys.argv[1:]
json_path = None
if args and args[0] == "--json":
    json_path, args = args[1], args[2:]
tot = defaultdict(float); n = defaultdict(int)
for d in args:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "")
            if "astc_compress" not in k:
                continue
            tot[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
for c in sorted(tot):
    print("%-24s sum=%.6g over %d dispatch records (per dispatch %.6g)" % (c, tot[c], n[c], tot[c] / max(n[c], 1)))
if "FETCH_SIZE" in tot and "WRITE_SIZE" in tot:
    rd = tot["FETCH_SIZE"] / n["FETCH_SIZE"] * 1024.0 * 2.0
    wr = tot["WRITE_SIZE"] / n["WRITE_SIZE"] * 1024.0
    print("HBM traffic per launch: read %.4g B (2 x FETCH_SIZE KiB), write %.4g B, total %.4g B" % (rd, wr, rd + wr))
    if json_path:
        out = {"kernel": "astcd::astc_compress_blocks_ldr", "read_bytes_per_launch": rd, "write_bytes_per_launch": wr,
               "hbm_bytes_per_launch": rd + wr,
               "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over bench.py --steps 1; "
                         "KiB -> bytes, FETCH_SIZE doubled (gfx950 counts 128 B requests as 64 B)"}
        def per(c):
            return tot[c] / n[c] if c in tot else None
        waves = per("SQ_WAVES")
        if waves and per("SQ_INSTS_VALU"):
            out["valu_insts_per_block"] = round(per("SQ_INSTS_VALU") / waves, 1)          # one wave = one block
        if waves and per("SQ_INSTS_SALU"):
            out["salu_insts_per_block"] = round(per("SQ_INSTS_SALU") / waves, 1)
        if waves and per("SQ_INSTS_LDS"):
            out["lds_insts_per_block"] = round(per("SQ_INSTS_LDS") / waves, 1)
        if per("SQ_THREAD_CYCLES_VALU") and per("SQ_ACTIVE_INST_VALU"):
            out["active_lanes_avg"] = round(per("SQ_THREAD_CYCLES_VALU") / per("SQ_ACTIVE_INST_VALU"), 2)   # of 64
        if per("SQ_ACTIVE_INST_VALU") and per("SQ_WAVE_CYCLES"):
            # both counters tick in units of 4 clocks; 4 waves share a SIMD (launch bounds), so wave residency / 4 = SIMD time.
            # Every VALU instruction counts as one 4-clock slot here whatever its real issue cost (2.6 .. 4.5 clocks, see
            # profiles/r02a/valu_microbench*.txt), so this is an upper bound of the VALU pipe's busy fraction.
            out["valu_issue_frac"] = round(per("SQ_ACTIVE_INST_VALU") / (per("SQ_WAVE_CYCLES") / 4.0), 4)
        json.dump(out, open(json_path, "w"), indent=1)
