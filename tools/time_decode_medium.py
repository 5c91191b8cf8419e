#!/usr/bin/env python3
"""Decode rate on -medium encoder output (more multi-partition / dual-plane blocks than the -fastest stream of
time_decode.py).  usage: time_decode_medium.py [size] [block]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "astc-encoder_amd", "python"))
import numpy as np, torch
import astcenc_amd as A
size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
b = int(sys.argv[2]) if len(sys.argv) > 2 else 6
torch.zeros(1, device="cuda")
lib = A.Library(A.LIB_PRODUCT)
err, cfg = lib.config_init(A.PRF_LDR, b, b, 1, 60.0, 0); assert err == 0
err, ctx = lib.context_alloc(cfg, 1); assert err == 0
img = torch.from_numpy(A.synthetic_image(size, size)).cuda()
nb = ((size + b - 1) // b) ** 2
blocks = torch.zeros(nb * 16, dtype=torch.uint8, device="cuda")
swz = A.Swizzle(*A.SWZ_RGBA); ms = ctypes.c_float()
assert lib.lib.astcenc_amd_compress_image_device(ctx, img.data_ptr(), size, size, 0, ctypes.byref(swz), blocks.data_ptr(), blocks.numel(), None, ctypes.byref(ms)) == 0
dec = torch.zeros((size, size, 4), dtype=torch.uint8, device="cuda")
best = 1e9
for i in range(5):
    torch.cuda.synchronize(); t = time.perf_counter()
    assert lib.lib.astcenc_amd_decompress_image_device(ctx, blocks.data_ptr(), blocks.numel(), dec.data_ptr(), size, size, 1, A.TYPE_U8, ctypes.byref(swz), None) == 0
    dt = time.perf_counter() - t
    if i: best = min(best, dt)
nbytes = nb * 16 + dec.numel()
print("decode of -medium output %dx%d %dx%d -> U8: %.3f ms, %.1f GB/s, checksum %d" % (size, size, b, b, best * 1e3, nbytes / best / 1e9, int(dec.to(torch.int64).sum().item())))
