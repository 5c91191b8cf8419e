#!/usr/bin/env python3
"""Turn the output of tools/gpu_evidence.sh into <dir>/traffic.json (per-config counters that bench.py reads back) and a
text summary.  HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are in KiB and come
from their own passes; on gfx950 FETCH_SIZE counts 128-byte read requests as 64 bytes, so the read side is doubled.

usage: summarize_evidence.py DIR c2 [c3 c4 ...]"""
import csv, glob, json, os, sys
from collections import defaultdict

d, configs = sys.argv[1], sys.argv[2:]
out = {"configs": {}, "method": "rocprofv3 --pmc over `bench.py --config <c> --steps 1 --warmup 0` (one launch of the compression kernel at the "
                               "config's full size); FETCH_SIZE and WRITE_SIZE in separate passes, KiB -> bytes, FETCH_SIZE doubled "
                               "(gfx950 counts 128 B requests as 64 B); per block = per wavefront (one wave per block)"}
for c in configs:
    tot, n = defaultdict(float), defaultdict(int)
    kernel = None
    for p in range(1, 5):
        for f in glob.glob(os.path.join(d, "%s_pmc%d" % (c, p), "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                k = row.get("Kernel_Name", "")
                if "astc_compress" not in k:
                    continue
                kernel = k.split("(")[0]
                tot[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
    per = lambda name: tot[name] / n[name] if n.get(name) else None
    e = {"kernel": kernel}
    if per("FETCH_SIZE") is not None and per("WRITE_SIZE") is not None:
        e["read_bytes_per_launch"] = per("FETCH_SIZE") * 1024.0 * 2.0
        e["write_bytes_per_launch"] = per("WRITE_SIZE") * 1024.0
        e["hbm_bytes_per_launch"] = e["read_bytes_per_launch"] + e["write_bytes_per_launch"]
    waves = per("SQ_WAVES")
    if waves:
        e["blocks"] = waves
        for key, counter in (("valu_insts_per_block", "SQ_INSTS_VALU"), ("salu_insts_per_block", "SQ_INSTS_SALU"), ("lds_insts_per_block", "SQ_INSTS_LDS"),
                             ("vmem_rd_insts_per_block", "SQ_INSTS_VMEM_RD"), ("vmem_wr_insts_per_block", "SQ_INSTS_VMEM_WR"), ("wave_quad_cycles_per_block", "SQ_WAVE_CYCLES")):
            if per(counter) is not None:
                e[key] = round(per(counter) / waves, 2)
    if per("SQ_THREAD_CYCLES_VALU") and per("SQ_ACTIVE_INST_VALU"):
        e["active_lanes_avg"] = round(per("SQ_THREAD_CYCLES_VALU") / per("SQ_ACTIVE_INST_VALU"), 2)
    if per("SQ_ACTIVE_INST_VALU") and per("SQ_WAVE_CYCLES"):
        # both tick in units of 4 clocks; 4 waves share a SIMD (c2, c4; config 3 runs 10 blocks per CU = 2.5 per SIMD, see
        # valu_busy_frac_of_kernel_time below), so wave residency / 4 = SIMD time.  Every VALU instruction counts as
        # one 4-clock slot whatever its real issue cost: an upper bound of the VALU pipe's busy fraction
        e["valu_issue_frac"] = round(per("SQ_ACTIVE_INST_VALU") / (per("SQ_WAVE_CYCLES") / 4.0), 4)
    if per("SQ_WAIT_ANY") and per("SQ_WAVE_CYCLES"):
        e["wait_any_frac_of_wave_cycles"] = round(per("SQ_WAIT_ANY") / per("SQ_WAVE_CYCLES"), 4)
    if per("SQ_LDS_BANK_CONFLICT") and per("SQ_ACTIVE_INST_LDS"):
        e["lds_bank_conflict_frac"] = round(per("SQ_LDS_BANK_CONFLICT") / per("SQ_ACTIVE_INST_LDS"), 4)
    # kernel time of the same command under --kernel-trace --stats
    for f in glob.glob(os.path.join(d, "%s_trace" % c, "**", "*kernel_stats.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "astc_compress" in row.get("Name", ""):
                e["rocprofv3_kernel_avg_ms"] = round(float(row["AverageNs"]) / 1e6, 3)
                e["rocprofv3_kernel_calls"] = int(row["Calls"])
    if per("SQ_ACTIVE_INST_VALU") and e.get("rocprofv3_kernel_avg_ms"):
        # occupancy-independent form of the same bound: VALU slots (4 clocks each) against the launch's wall time on 1024 SIMDs at the
        # nominal 2.4 GHz (the PMC pass and the traced pass run the same command line)
        e["valu_busy_frac_of_kernel_time"] = round(per("SQ_ACTIVE_INST_VALU") * 4.0 / (e["rocprofv3_kernel_avg_ms"] * 1e6 * 2.4 * 1024.0), 4)
    out["configs"][c] = e
    print("== %s" % c)
    for k, v in e.items():
        print("  %-32s %s" % (k, v))
if "c2" in out["configs"]:
    out.update({k: v for k, v in out["configs"]["c2"].items()})      # (top level = the headline workload, as in earlier rounds' files)
json.dump(out, open(os.path.join(d, "traffic.json"), "w"), indent=1)
