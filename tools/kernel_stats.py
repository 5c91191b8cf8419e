#!/usr/bin/env python3
"""Static facts about the compression kernel's code object: per-function instruction counts, VGPRs, scratch frames,
SGPR spills (v_writelane / v_readlane), and the kernel descriptor's private segment size and spill counts.

Compiles csrc/kernel_<which>.hip to gfx950 assembly with the product's flags and reads the assembler's own function /
kernel info comments and metadata.  usage: kernel_stats.py [ldr|hdr|ldr64|hdr64] [--json out.json]
(ldr64 / hdr64: the builds for footprints of at most 64 texels, i.e. what BASELINE configs 2-4 run)"""
import json, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "astc-encoder_amd")
which = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "ldr"
out_json = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
FLAGS = os.environ.get("KS_EXTRA", "") + " --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-math-errno -fno-slp-vectorize " \
        "-fvisibility=hidden -DASTCENC_DYNAMIC_LIBRARY=1 -Icsrc -Wno-unused-function --cuda-device-only -S"
with tempfile.TemporaryDirectory() as tmp:
    asm = os.path.join(tmp, "k.s")
    subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS.split() + ["-o", asm, "csrc/kernel_%s.hip" % which], cwd=PKG, check=True,
                   stderr=subprocess.DEVNULL)
    lines = open(asm).read().split("\n")
funcs, cur = [], None
for i, l in enumerate(lines):
    m = re.match(r"^(_Z[\w]+):", l)
    if m:
        cur = {"name": m.group(1), "start": i}
        continue
    if re.match(r"^\.Lfunc_end\d+:", l) and cur:
        cur["end"] = i
        funcs.append(cur)
        cur = None
rows = []
for f in funcs:
    body = lines[f["start"]:f["end"]]
    tail = "\n".join(lines[f["end"]:f["end"] + 40])
    def info(key):
        m = re.search(r"; " + key + r": (\d+)", tail)
        return int(m.group(1)) if m else None
    name = re.sub(r"^_ZN5astcd5v_[a-z]+L?\d+", "", f["name"])
    name = re.sub(r"^_ZN5astcd\d+", "", name)
    name = re.sub(r"E[a-zA-Z0-9_]*$", "", name)
    rows.append({"function": name,
                 "instructions": sum(1 for x in body if re.match(r"^\s+[a-z][a-z_0-9]+ ", x) and not x.strip().startswith(".")),
                 "valu": sum(1 for x in body if re.match(r"^\s+v_", x)),
                 "scratch_stores": sum("scratch_store" in x for x in body), "scratch_loads": sum("scratch_load" in x for x in body),
                 "sgpr_spill_writelane": sum("v_writelane" in x for x in body), "sgpr_spill_readlane": sum("v_readlane" in x for x in body),
                 "vgprs": info("NumVgprs"), "frame_bytes": info("ScratchSize")})
meta = {}
for l in lines:
    for key in ("private_segment_fixed_size", "sgpr_spill_count", "vgpr_spill_count", "sgpr_count", "vgpr_count", "group_segment_fixed_size"):
        m = re.match(r"\s*\." + key + r":\s+(\d+)", l)
        if m:
            meta[key] = int(m.group(1))
print("%-36s %6s %6s %8s %8s %9s %6s %6s" % ("function", "instr", "VALU", "scr st", "scr ld", "wl/rl", "VGPR", "frame"))
for r in rows:
    print("%-36s %6d %6d %8d %8d %4d/%-4d %6s %6s" % (r["function"][:36], r["instructions"], r["valu"], r["scratch_stores"], r["scratch_loads"],
                                                     r["sgpr_spill_writelane"], r["sgpr_spill_readlane"], r["vgprs"], r["frame_bytes"]))
print("kernel descriptor:", meta)
if out_json:
    json.dump({"kernel": "astcd::astc_compress_blocks_%s" % which, "descriptor": meta, "functions": rows}, open(out_json, "w"), indent=1)
