#!/usr/bin/env python3
"""Run-time builds against the generic build of the library, footprint by footprint (bytes compared, no reference needed)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "astc-encoder_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import astcenc_amd as A, images
torch.zeros(1, device="cuda")
lib = A.Library(A.LIB_PRODUCT)
noisy = images.noisy(120, 113, 21)
vol = np.stack([images.noisy(40, 36, 40 + z) for z in range(12)])
for block in [(8, 8), (10, 5), (10, 8), (10, 10), (12, 10), (12, 12), (3, 3, 3), (4, 4, 4), (5, 5, 5), (6, 6, 6)]:
    for q in (0.0, 10.0, 60.0, 98.0):
        img = vol if len(block) == 3 else noisy
        os.environ["ASTCENC_AMD_JIT"] = "off"
        want = lib.compress(img, block, q).reshape(-1, 16)
        os.environ["ASTCENC_AMD_JIT"] = "sync"
        got = lib.compress(img, block, q, specialize="try").reshape(-1, 16)
        bad = np.nonzero((want != got).any(axis=1))[0]
        print(block, q, lib.last_kernel[-12:], "mismatching %d of %d" % (bad.size, want.shape[0]), list(bad[:12]), flush=True)
