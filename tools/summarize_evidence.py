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
    for p in range(1, 7):
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
    if per("SQ_WAIT_ANY") and per("SQ_WAVE_CYCLES"):
        e["wait_any_frac_of_wave_cycles"] = round(per("SQ_WAIT_ANY") / per("SQ_WAVE_CYCLES"), 4)
    # measured occupancy: wave-cycles per busy CU-cycle.  SQ_WAVE_CYCLES counts in units of four cycles, SQ_BUSY_CU_CYCLES in
    # cycles, a CU has four SIMDs: the two factors cancel (config 3, whose LDS allows 10 workgroups per CU = 2.5 per SIMD,
    # reads 2.47 this way)
    if per("SQ_BUSY_CU_CYCLES") and per("SQ_WAVE_CYCLES"):
        e["waves_per_simd_measured"] = round(per("SQ_WAVE_CYCLES") / per("SQ_BUSY_CU_CYCLES"), 3)
    # (SQ_INST_CYCLES_SALU turned out to be four cycles per scalar instruction, i.e. the instruction count again, and
    #  SQ_INST_CYCLES_VALU is not delivered next to it on this stack: no issue-cycle figure from the counters)
    if waves and per("SQ_INSTS_SMEM") is not None:
        e["smem_insts_per_block"] = round(per("SQ_INSTS_SMEM") / waves, 2)
    if per("SQ_LDS_BANK_CONFLICT") and per("SQ_ACTIVE_INST_LDS"):
        e["lds_bank_conflict_frac"] = round(per("SQ_LDS_BANK_CONFLICT") / per("SQ_ACTIVE_INST_LDS"), 4)
    # kernel time of the same command under --kernel-trace --stats
    for f in glob.glob(os.path.join(d, "%s_trace" % c, "**", "*kernel_stats.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "astc_compress" in row.get("Name", ""):
                e["rocprofv3_kernel_avg_ms"] = round(float(row["AverageNs"]) / 1e6, 3)
                e["rocprofv3_kernel_calls"] = int(row["Calls"])
    # The VALU opcode classes the hardware counts, per block, and what they cost to issue: SIMD cycles per wave64
    # instruction measured with hand-written asm loops at 4 to 8 waves per SIMD (tools/valu_microbench3.hip,
    # profiles/r04a/valu_microbench3.txt): fp32 add / sub / mul 2.2; VOP3 fma 2.4; conversions 4.3; transcendentals 8.3.
    # INT32 mixes both classes (add / sub / and / or / lshr 2.2; lshl, mul, mad, bfe, min / max, compares 4.3) and so does
    # what no class counter sees (v_mov 2.2; v_cmp, v_cndmask, fp32 min / max, DPP forms, lane reads 4.3): the sum is given
    # as a range, every unclassified instruction at 2.2 / at 4.3.
    if waves and per("SQ_INSTS_VALU_ADD_F32") is not None and per("SQ_INSTS_VALU"):
        cls = {k: per("SQ_INSTS_VALU_" + k) / waves for k in ("ADD_F32", "MUL_F32", "FMA_F32", "TRANS_F32", "CVT", "INT32", "INT64")}
        total = per("SQ_INSTS_VALU") / waves
        known = cls["ADD_F32"] * 2.2 + cls["MUL_F32"] * 2.2 + cls["FMA_F32"] * 2.4 + cls["TRANS_F32"] * 8.3 + cls["CVT"] * 4.3
        mixed = cls["INT32"] + cls["INT64"] + max(0.0, total - sum(cls.values()))
        e["valu_class_insts_per_block"] = {k.lower(): round(v, 1) for k, v in cls.items()}
        e["valu_class_insts_per_block"]["unclassified"] = round(max(0.0, total - sum(cls.values())), 1)
        e["valu_issue_cycles_per_block"] = {"low": round(known + mixed * 2.2, 0), "high": round(known + mixed * 4.3, 0),
                                            "costs": "profiles/r04a/valu_microbench3.txt"}
    out["configs"][c] = e
    print("== %s" % c)
    for k, v in e.items():
        print("  %-32s %s" % (k, v))
if "c2" in out["configs"]:
    out.update({k: v for k, v in out["configs"]["c2"].items()})      # (top level = the headline workload, as in earlier rounds' files)
json.dump(out, open(os.path.join(d, "traffic.json"), "w"), indent=1)
