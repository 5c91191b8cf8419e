#!/usr/bin/env python3
"""Bisect a misbehaving run-time build: which stage functions need the live LDS layout (ASTC_DEBUG_LIVE_LAYOUT_IN) for the
bytes to match the generic build.  usage: jit_debug3.py bx by quality 'opts1' 'opts2' ..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "astc-encoder_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import astcenc_amd as A, images
torch.zeros(1, device="cuda")
lib = A.Library(A.LIB_PRODUCT)
noisy = images.noisy(120, 113, 21)
block, q = (int(sys.argv[1]), int(sys.argv[2])), float(sys.argv[3])
os.environ["ASTCENC_AMD_JIT"] = "off"
want = lib.compress(noisy, block, q).reshape(-1, 16)
os.environ["ASTCENC_AMD_JIT"] = "sync"
for opts in sys.argv[4:]:
    os.environ["ASTCENC_AMD_JIT_OPTIONS"] = opts
    got = lib.compress(noisy, block, q, specialize="try").reshape(-1, 16)
    bad = np.nonzero((want != got).any(axis=1))[0]
    print(block, q, "[%s]" % opts, lib.last_kernel[-12:], "mismatching %d of %d" % (bad.size, want.shape[0]), flush=True)
