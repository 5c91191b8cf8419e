#!/usr/bin/env python3
"""Run the stage-timer build (libastcenc_amd_prof.so, -DASTC_PROFILE) on a synthetic image; the
library prints per-stage shader-clock cycles per block to stderr."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["ASTCENC_AMD_LIB"] = os.environ.get("PROF_LIB", os.path.join(ROOT, "astc-encoder_amd", "libastcenc_amd_prof.so"))
sys.path.insert(0, os.path.join(ROOT, "astc-encoder_amd", "python"))
import astcenc_amd as A
size = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
bx = int(sys.argv[2]) if len(sys.argv) > 2 else 6
q = float(sys.argv[3]) if len(sys.argv) > 3 else 60.0
lib = A.Library(A.LIB_PRODUCT)
img = A.synthetic_image(size, size)
lib.compress(img[:256, :256].copy(), (bx, bx), q)   # warm up
t = time.time(); lib.compress(img, (bx, bx), q); dt = time.time() - t
print("%dx%d %dx%d q=%.0f: %.3f s end-to-end (%.2f Mtexels/s incl. PCIe)" % (size, size, bx, bx, q, dt, size * size / dt / 1e6))
