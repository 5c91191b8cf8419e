#!/usr/bin/env python3
"""Per-stage instruction counts from the runs of tools/gpu_stage_counts.sh (stage doubled minus plain run)."""
import csv, glob, os, sys
NAMES = {1: "ideal endpoints+weights", 2: "decimate (all grids)", 3: "angular bounds", 4: "mode scoring", 5: "mode scoring + formats",
         6: "candidate quantize", 7: "candidate restore/staging", 8: "recompute endpoints", 9: "pack endpoints", 10: "difference (decode+score)",
         11: "partition order (k-means)", 12: "partition score", 13: "partition select", 14: "block statistics", 15: "load block", 16: "physical", 18: "weight realignment",
         19: "batch: rows + weights", 20: "batch: sums", 21: "batch: solve", 22: "batch: pack", 23: "batch: score",
         24: "TRIALS: A0 (1 partition, always modes)", 25: "TRIALS: A1 (1 partition, all modes)", 26: "TRIALS: two planes", 27: "TRIALS: 2 partitions", 28: "TRIALS: 3 partitions", 29: "TRIALS: 4 partitions", 30: "  realignment of two-plane candidates", 31: "  realignment: first pass over all weights"}
d = sys.argv[1]
def load(i):
    tot = {}
    for f in glob.glob(os.path.join(d, "dup_%d" % i, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "astc_compress" in row.get("Kernel_Name", ""):
                tot[row["Counter_Name"]] = tot.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    return tot
base = load(0)
if not base:
    sys.exit("no base run")
waves = base["SQ_WAVES"]
cols = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_WAVE_CYCLES"]
print("per block (one wave per block, %d blocks); plain run: " % waves + "  ".join("%s %.0f" % (c[3:], base[c] / waves) for c in cols))
print("avg active lanes per VALU instruction, whole kernel: %.1f" % (base["SQ_THREAD_CYCLES_VALU"] / base["SQ_ACTIVE_INST_VALU"]))
print("%-40s %9s %6s %8s %8s %8s %9s %7s" % ("stage", "VALU", "share", "SALU", "LDS", "VMEM_RD", "wavecyc", "lanes"))
rows = {}
for i in sorted(NAMES):
    t = load(i)
    if not t:
        continue
    dv = {c: (t[c] - base[c]) / waves for c in cols}
    dth, dac = t["SQ_THREAD_CYCLES_VALU"] - base["SQ_THREAD_CYCLES_VALU"], t["SQ_ACTIVE_INST_VALU"] - base["SQ_ACTIVE_INST_VALU"]
    rows[i] = dv
    print("%-40s %9.0f %5.1f%% %8.0f %8.0f %8.0f %9.0f %7.1f" % (NAMES[i], dv[cols[0]], 100 * dv[cols[0]] * waves / base[cols[0]], dv[cols[1]], dv[cols[2]], dv[cols[3]],
                                                              dv[cols[4]], dth / dac if dac else 0))
if 5 in rows and 4 in rows:
    print("%-28s %9.0f %5.1f%%" % ("  formats (5 minus 4)", rows[5][cols[0]] - rows[4][cols[0]], 100 * (rows[5][cols[0]] - rows[4][cols[0]]) * waves / base[cols[0]]))
acc = sum(v[cols[0]] for k, v in rows.items() if k != 4 and k < 24)
print("%-28s %9.0f %5.1f%%   (control code between the stages, block-level code)" % ("not doubled", base[cols[0]] / waves - acc, 100 * (1 - acc * waves / base[cols[0]])))
