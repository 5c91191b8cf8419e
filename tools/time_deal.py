#!/usr/bin/env python3
"""Dealing of a host image to the device slots of a context (backend_compress): per-slot busy time with one contiguous range
per slot (ASTCENC_AMD_DEAL=static) against portions taken from one counter (the default), on an image whose rows differ in cost
-- the top half the synthetic bench content (every trial of the search runs), the bottom half smooth content (most blocks stop
after the first trial).  Eight slots on the one GPU of the test box share that GPU: the call's wall time cannot change here; what
shows is how long each slot's host thread is busy -- on eight GPUs the longest of them is the call's time.
usage: time_deal.py [size]   (run with ASTCENC_AMD_DEVICES=0,0,0,0,0,0,0,0 ASTCENC_AMD_LOG=stderr)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "astc-encoder_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import astcenc_amd as A, images
size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
torch.zeros(1, device="cuda")
lib = A.Library(A.LIB_PRODUCT)
err, cfg = lib.config_init(A.PRF_LDR, 6, 6, 1, A.PRE_MEDIUM, 0); assert err == 0
err, ctx = lib.context_alloc(cfg, 1); assert err == 0
img = A.synthetic_image(size, size)
img[size // 2:] = images.smooth(size, size - size // 2)
img = np.ascontiguousarray(img)
nb = ((size + 5) // 6) ** 2
out = np.zeros(nb * 16, dtype=np.uint8)
for i in range(3):
    t = time.perf_counter()
    assert lib.compress_raw(ctx, img, out) == 0
    dt = time.perf_counter() - t
    assert lib.lib.astcenc_compress_reset(ctx) == 0
    print("call %d: %.1f ms (%d devices, deal=%s)" % (i, dt * 1e3, lib.lib.astcenc_amd_context_device_count(ctx), os.environ.get("ASTCENC_AMD_DEAL", "dynamic")), file=sys.stderr, flush=True)
import hashlib
print("blocks sha256 %s" % hashlib.sha256(out.tobytes()).hexdigest()[:16])
