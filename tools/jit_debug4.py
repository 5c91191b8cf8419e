#!/usr/bin/env python3
"""A misbehaving run-time build, block by block: every block of the image as a one-block image (one workgroup on the device at a
time) against the whole image at once.  Mismatches that disappear when a workgroup runs alone point at something workgroups
share -- registers or LDS a kernel descriptor under-reports -- not at the arithmetic."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "astc-encoder_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import astcenc_amd as A, images
torch.zeros(1, device="cuda")
lib = A.Library(A.LIB_PRODUCT)
bx, by, q = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
noisy = images.noisy(12 * bx, 15 * by, 21)
def run(mode, img):
    os.environ["ASTCENC_AMD_JIT"] = mode
    return lib.compress(img, (bx, by), q, specialize="try" if mode == "sync" else False).reshape(-1, 16)
want = run("off", noisy)
os.environ["ASTCENC_AMD_JIT_SELF_CHECK"] = "0"
got = run("sync", noisy)
print("whole image: kernel", lib.last_kernel[-12:], "mismatching", int((want != got).any(axis=1).sum()), "of", want.shape[0], flush=True)
bad_alone = 0
for r in range(15):
    for c in range(12):
        tile = np.ascontiguousarray(noisy[r * by:(r + 1) * by, c * bx:(c + 1) * bx])
        one = run("sync", tile)
        if (one[0] != want[r * 12 + c]).any(): bad_alone += 1
print("one block per launch: mismatching", bad_alone, "of", 180, flush=True)
for rows in (1, 2, 4, 8):
    strip = np.ascontiguousarray(noisy[:rows * by])
    g = run("sync", strip)
    print("%d block rows (%d workgroups): mismatching %d" % (rows, rows * 12, int((g != want[:rows * 12]).any(axis=1).sum())), flush=True)
