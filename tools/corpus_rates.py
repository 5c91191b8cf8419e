#!/usr/bin/env python3
"""Coding rate per image of the reference's corpus (VERDICT r05 "missing" 5): the reference's own command line front end
prints "Coding rate" for the astcenc_compress_image calls it makes (Source/astcenccli_toplevel.cpp: best of -repeats runs);
here it runs every corpus image at 6x6 -medium (the command lines of tests/test_corpus.py) once linked against libastcenc_amd.so
and once against the reference's AVX2 library on this host's CPUs, and prints both next to the rate the reference recorded on
its own test machine (tests/corpus/manifest.json: ref_coding_rate, from Test/Images/*/astc_reference-main-avx2_medium_results.csv).
Host pointers in and out: PCIe both ways is inside the GPU figure.   usage: corpus_rates.py [block] [preset] [repeats]"""
import json, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI_AMD = os.path.join(ROOT, "oracle", "_ref", "astcenc-cli-amd")
CLI_REF = os.path.join(ROOT, "oracle", "_ref", "astcenc-cli-ref-avx2")
IMAGES = os.path.join(ROOT, "tests", "corpus", "_images")
MANIFEST = json.load(open(os.path.join(ROOT, "tests", "corpus", "manifest.json")))["images"]
COMPRESS = {"ldr": "-cl", "ldrs": "-cs", "hdr": "-ch"}
block = sys.argv[1] if len(sys.argv) > 1 else "6x6"
preset = sys.argv[2] if len(sys.argv) > 2 else "medium"
repeats = sys.argv[3] if len(sys.argv) > 3 else "5"
rate_re = re.compile(r"\s*Coding rate:\s*([0-9.]+) MT/s")
size_re = re.compile(r"\s*Dimensions:\s*(\d+)D, (\d+)x(\d+)")


def rate(exe, img, d, threads=None):
    src = os.path.join(IMAGES, img["set"], img["dir"], img["file"])
    extra = ["-repeats", repeats]
    if threads:
        extra += ["-j", str(threads)]
    if img["format"] == "xy":
        extra.append("-normal")
    if "a" in img["flags"]:
        extra += ["-a", "1"]
    r = subprocess.run([exe, COMPRESS[img["profile"]], src, "out.astc", block, "-" + preset] + extra, cwd=d, capture_output=True, text=True, timeout=1200)
    if r.returncode != 0:
        return None, None
    m = [rate_re.match(l) for l in r.stdout.splitlines()]
    m = [x for x in m if x]
    s = [size_re.match(l) for l in r.stdout.splitlines()]
    s = [x for x in s if x]
    return (float(m[0].group(1)) if m else None), ("%sx%s" % (s[0].group(2), s[0].group(3)) if s else "?")


print("%-10s %-34s %-11s %12s %12s %14s %18s" % ("set", "image", "size", "this library", "this library", "reference here", "reference recorded"))
print("%-10s %-34s %-11s %12s %12s %14s %18s" % ("", "", "", "MT/s, -j 1", "MT/s, -j all", "MT/s (AVX2)", "MT/s (its machine)"))
tot = [0.0, 0.0, 0]
with tempfile.TemporaryDirectory() as d:
    for img in MANIFEST:
        if "3" in img["flags"]:
            continue
        # (-j 1 for this library: the front end starts its -j worker threads anew for every call -- by default one per host CPU,
        #  256 on the test box, some 8 ms -- and here only the first of them does anything: INTEGRATION.md section 1)
        amd, size = rate(CLI_AMD, img, d, 1)
        amd_all, _ = rate(CLI_AMD, img, d)
        ref, _ = rate(CLI_REF, img, d)
        rec = img.get("ref_coding_rate", {}).get("%s/%s" % (preset, block))
        print("%-10s %-34s %-11s %12s %12s %14s %18s" % (img["set"], img["file"], size, "%.1f" % amd if amd else "-", "%.1f" % amd_all if amd_all else "-",
                                                    "%.2f" % ref if ref else "-", "%.2f" % rec if rec else "-"), flush=True)
        if amd and ref:
            tot[0] += amd; tot[1] += ref; tot[2] += 1
if tot[2]:
    print("mean over %d images: this library %.1f MT/s, the reference on this host %.2f MT/s" % (tot[2], tot[0] / tot[2], tot[1] / tot[2]))
