#!/usr/bin/env python3
"""An ESTIMATED dynamic opcode histogram of the compression kernel, for naming what the hardware's VALU class counters
leave unclassified (rocprofv3's PC sampling -- host_trap and stochastic -- is refused by this stack:
profiles/r05z/pc_sampling_*.log).

Method: every VALU / scalar / LDS / memory instruction of the build's -g1 assembly listing is put into a stage bucket (the
out-of-line stage function it sits in; inside the kernel body, the inlined function its .loc chain names: partition
scoring, realignment, endpoint packing, control) and weighted by 8^(loop depth) (the listing's "Loop ... Depth=d"
annotations); each bucket's weights are then scaled so that its VALU / SALU / LDS totals equal the bucket's MEASURED dynamic
counts from the stage-doubling runs (tools/gpu_stage_counts.sh -> stage_counts_*.txt).  So: measured per stage and per
instruction kind, estimated (by loop nesting) inside a stage.  The sum over the opcodes of a hardware class can be checked
against that class's counter (printed at the end).

usage: opcode_histogram.py <kernel name, e.g. ldr_6x6m> <stage_counts.txt> [traffic.json config]"""
import collections, json, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "astc-encoder_amd")
which, counts_path = sys.argv[1], sys.argv[2]
FLAGS = ("--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-math-errno -fno-slp-vectorize "
         "-fvisibility=hidden -DASTCENC_DYNAMIC_LIBRARY=1 -Icsrc -Wno-unused-function --cuda-device-only -S -g1").split()
if which in ("ldr_6x6m", "ldr_8x8t"):
    FLAGS += ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]
with tempfile.TemporaryDirectory() as tmp:
    asm = os.path.join(tmp, "k.s")
    subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-o", asm, "csrc/kernel_%s.hip" % which], cwd=PKG, check=True, stderr=subprocess.DEVNULL)
    lines = open(asm).read().split("\n")

# stage table: name -> (VALU, SALU, LDS) per block
STAGE_OF_FUNCTION = {"stage_ideal": "ideal endpoints+weights", "stage_decimate": "decimate (all grids)", "stage_angular": "angular bounds",
                     "stage_modes": "mode scoring", "stage_formats": "formats", "refine_quantize_candidates": "candidate quantize",
                     "refine_candidate_restore": "candidate restore/staging", "refine_recompute": "recompute endpoints",
                     "refine_difference": "difference (decode+score)", "stage_partition_order": "partition order (k-means)",
                     "stage_partition_select": "partition select", "batch_refit": "batch: refit", "batch_pack": "batch: pack",
                     "batch_score": "batch: score", "stage_block_statistics": "block statistics", "refine_accept": "control"}
measured = {}
for l in open(counts_path):
    m = re.match(r"^(.{40})\s+(-?\d+)\s+[\d.]+%\s+(-?\d+)\s+(-?\d+)", l)
    if m:
        measured[m.group(1).strip()] = (float(m.group(2)), float(m.group(3)), float(m.group(4)))
plain = re.search(r"INSTS_VALU (\d+)\s+INSTS_SALU (\d+)\s+INSTS_LDS (\d+)", open(counts_path).read())
total = tuple(float(x) for x in plain.groups())
bucket_measured = {
    "ideal endpoints+weights": measured["ideal endpoints+weights"], "decimate (all grids)": measured["decimate (all grids)"],
    "angular bounds": measured["angular bounds"], "mode scoring": measured["mode scoring"],
    "formats": tuple(a - b for a, b in zip(measured["mode scoring + formats"], measured["mode scoring"])),
    "candidate quantize": measured["candidate quantize"], "candidate restore/staging": measured["candidate restore/staging"],
    "recompute endpoints": measured["recompute endpoints"], "difference (decode+score)": measured["difference (decode+score)"],
    "partition order (k-means)": measured["partition order (k-means)"], "partition select": measured["partition select"],
    "batch: refit": tuple(sum(measured[k][i] for k in ("batch: rows + weights", "batch: sums", "batch: solve")) for i in range(3)),
    "batch: pack": measured["batch: pack"], "batch: score": measured["batch: score"],
    "partition score": measured["partition score"], "weight realignment": measured["weight realignment"], "pack endpoints": measured["pack endpoints"]}
rest = tuple(total[i] - sum(v[i] for v in bucket_measured.values()) for i in range(3))
bucket_measured["control"] = tuple(max(x, 0.0) for x in rest)     # the search driver, block statistics, load, physical ...

def kind(op):
    if op.startswith("v_"): return 0
    if op.startswith("s_") and not op.startswith(("s_waitcnt", "s_nop", "s_load", "s_buffer_load", "s_endpgm", "s_barrier", "s_sleep", "s_setprio")): return 1
    if op.startswith("ds_"): return 2
    return None

def body_bucket(chain):
    # inside the kernel body: which inlined function does the .loc chain name?
    if re.search(r"wave_partition\.h:(3[6-9]\d|[45]\d\d|6[0-6]\d)\b", chain): return "partition score"
    if re.search(r"wave_refine\.h:(6[4-9]\d|[7-9]\d\d|1[0-3]\d\d)\b", chain): return "weight realignment"
    if re.search(r"wave_color\.h|wave_quad\.h|wave_block\.h:(4[7-9]\d|5[0-6]\d)\b", chain): return "pack endpoints"
    return "control"

weights = collections.defaultdict(lambda: collections.Counter())   # bucket -> opcode -> weight
fn_bucket, depth, chain = None, 0, ""
for l in lines:
    m = re.match(r"^(_Z[\w]+):", l)
    if m:
        name = m.group(1)
        fn_bucket = "KERNEL" if "astc_compress_blocks" in name else next((b for f, b in STAGE_OF_FUNCTION.items() if f in name), "control")
        depth = 0
        continue
    m = re.match(r"^\.LBB\d+_\d+:(.*)$", l)
    if m:
        d = re.search(r"Depth=(\d+)", m.group(1))
        depth = int(d.group(1)) if d else 0
        continue
    m = re.match(r"\s*\.loc\s+\d+\s+\d+.*;\s*(.*)$", l)
    if m:
        chain = m.group(1)
        continue
    m = re.match(r"^\s+([a-z][a-z0-9_]+)\s", l)
    if not m or fn_bucket is None:
        continue
    op = m.group(1)
    if kind(op) is None:
        continue
    b = body_bucket(chain) if fn_bucket == "KERNEL" else fn_bucket
    weights[b][re.sub(r"_e32$|_e64$|_sdwa$|_dpp$", lambda s: s.group(0) if s.group(0) in ("_sdwa", "_dpp") else "", op)] += 8.0 ** depth

est = collections.Counter()
for b, ops in weights.items():
    if b not in bucket_measured:
        continue
    for k in range(3):
        w = sum(v for o, v in ops.items() if kind(o) == k)
        if w <= 0:
            continue
        scale = bucket_measured[b][k] / w
        for o, v in ops.items():
            if kind(o) == k:
                est[o] += v * scale

def hw_class(op):
    o = op.replace("_dpp", "").replace("_sdwa", "")
    if o in ("v_add_f32", "v_sub_f32", "v_subrev_f32"): return "add_f32"
    if o == "v_mul_f32": return "mul_f32"
    if o in ("v_fma_f32", "v_fmac_f32", "v_fmamk_f32", "v_fmaak_f32", "v_mad_f32", "v_mac_f32", "v_div_fmas_f32", "v_div_fixup_f32"): return "fma_f32"
    if o in ("v_rcp_f32", "v_sqrt_f32", "v_rsq_f32", "v_log_f32", "v_exp_f32", "v_rcp_iflag_f32", "v_sin_f32", "v_cos_f32"): return "trans_f32"
    if o.startswith("v_cvt_"): return "cvt"
    return None

print("estimated dynamic instructions per block, %s (totals measured: VALU %.0f, SALU %.0f, LDS %.0f)" % (which, *total))
for k, title in ((0, "VALU"), (1, "scalar"), (2, "LDS")):
    rows = [(o, v) for o, v in est.items() if kind(o) == k]
    tot = sum(v for _, v in rows)
    print("--- %s (%.0f)" % (title, tot))
    for o, v in sorted(rows, key=lambda r: -r[1])[:45 if k == 0 else 18]:
        print("  %-28s %8.0f  %5.1f%%  %s" % (o, v, 100.0 * v / tot, (hw_class(o) or "") if k == 0 else ""))
cls = collections.Counter()
for o, v in est.items():
    if kind(o) == 0:
        cls[hw_class(o) or "other (int32 / int64 / not counted by any class)"] += v
print("--- VALU by hardware counter class (estimate; compare with valu_class_insts_per_block of the evidence summary)")
for c, v in cls.most_common():
    print("  %-55s %8.0f" % (c, v))
if len(sys.argv) > 3:
    t = json.load(open(sys.argv[3]))["configs"][sys.argv[4] if len(sys.argv) > 4 else "c2"].get("valu_class_insts_per_block")
    print("  measured classes:", t)
