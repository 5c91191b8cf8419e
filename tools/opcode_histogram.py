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
est_by_bucket = collections.defaultdict(collections.Counter)
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
                est_by_bucket[b][o] += v * scale

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


# ---- count x measured issue cost, per stage ----------------------------------------------------------
# Costs from the issue microbenchmarks on the device (profiles/r04z/valu_microbench3_with_scalar_rows.txt, profiles/r06a/
# valu_microbench4.txt; SIMD cycles per wave-instruction with 4-8 waves on the SIMD):
#   full-rate class (add / sub / mul / fma / fmac / mov / and / or / xor / add_u32 / sub_u32 / lshrrev ...)  2.1 - 2.2
#   slow class (cndmask, every v_cmp, v_cvt_*, v_lshlrev, min / max, DPP and SDWA forms, lshl_add / add3 / bfe / mul_lo /
#               mul_u32_u24 / mad_u32_u24, readlane / readfirstlane, packed fp32)                                  4.1 - 4.3
#   transcendental (rcp, sqrt, rsq ...)                                                                           8.2
# ... but the classes do NOT add up: alternating rows cost what the full-rate class costs ("v_add + v_max alternating" 2.13
# per instruction, "3 x v_add + v_max" 2.06, "v_add + 3 x v_max" 3.11 = 12.4 / 4): a SIMD issues a slow-class instruction
# every 4.15 cycles and, beside it, full-rate ones -- VALU cycles = max(2.1 x all VALU, 4.15 x slow class + 8.2 x trans).
# The slow class costs extra only where it is more than HALF of the instructions in flight on the SIMD.
# And a wave issues ONE instruction of any kind per ~5.5 - 6 cycles of its own time ("v_add_f32" alone on a SIMD: 5.99;
# "3 x v_add + s_and_b64": 5.5 per instruction): with four waves per SIMD the instructions of a wave -- vector, scalar, LDS,
# branch, waitcnt alike -- are a floor under its residency of about 1.4 quad-cycles each.
FULL = 2.1; SLOW = 4.15; TRANS = 8.2; PER_WAVE = 5.5
def is_trans(o): return hw_class(o) == "trans_f32"
def is_slow(o):
    b = o.replace("_e32", "").replace("_e64", "")
    if is_trans(b): return False
    if b.endswith(("_dpp", "_sdwa")): return True
    return b.startswith(("v_cndmask", "v_cmp", "v_cmpx", "v_cvt_", "v_lshlrev", "v_min", "v_max", "v_med3", "v_lshl_add", "v_add3", "v_bfe", "v_bfi",
                         "v_mul_lo", "v_mul_hi", "v_mul_u32_u24", "v_mul_i32_i24", "v_mad_u32_u24", "v_mad_i32_i24", "v_readlane", "v_readfirstlane",
                         "v_writelane", "v_rndne", "v_pk_", "v_floor", "v_fract", "v_trunc", "v_ceil", "v_and_or", "v_or3", "v_lshl_or", "v_xad",
                         "v_bcnt", "v_ffbh", "v_ffbl", "v_alignbit", "v_perm", "v_sad", "v_div_scale", "v_div_fmas", "v_div_fixup", "v_ldexp",
                         "v_frexp", "v_ashrrev", "v_mbcnt", "v_dot"))
print()
print("--- count x measured issue cost per stage (estimate inside a stage, measured stage totals; cycles per block)")
print("%-28s %8s %8s %6s %8s %8s | %9s %9s  %s" % ("stage", "VALU", "slow", "trans", "scalar", "LDS", "VALU port", "wave issue", "slow share"))
tot = [0.0] * 7
for b in sorted(est_by_bucket, key=lambda b: -sum(v for o, v in est_by_bucket[b].items() if kind(o) == 0)):
    ops = est_by_bucket[b]
    valu = sum(v for o, v in ops.items() if kind(o) == 0)
    slow = sum(v for o, v in ops.items() if kind(o) == 0 and is_slow(o))
    trans = sum(v for o, v in ops.items() if kind(o) == 0 and is_trans(o))
    salu = sum(v for o, v in ops.items() if kind(o) == 1)
    lds = sum(v for o, v in ops.items() if kind(o) == 2)
    port = max(FULL * valu, SLOW * slow + TRANS * trans)
    wave = PER_WAVE * (valu + salu + lds)
    for i, x in enumerate((valu, slow, trans, salu, lds, port, wave)): tot[i] += x
    print("%-28s %8.0f %8.0f %6.0f %8.0f %8.0f | %9.0f %9.0f  %4.0f%%%s" % (b, valu, slow, trans, salu, lds, port, wave, 100.0 * slow / max(valu, 1), "  <- slow class binds" if SLOW * slow + TRANS * trans > FULL * valu else ""))
print("%-28s %8.0f %8.0f %6.0f %8.0f %8.0f | %9.0f %9.0f  %4.0f%%" % ("all", *tot, 100.0 * tot[1] / max(tot[0], 1)))
print("VALU port, all stages: %.0f cycles per block = %.0f quad-cycles of a SIMD; one wave's own issue floor: %.0f cycles = %.0f quad-cycles of its residency" % (tot[5], tot[5] / 4, tot[6], tot[6] / 4))
