#!/usr/bin/env python3
"""Rate of the decompression and image-comparison kernels on device-resident data (wall time of the
synchronous device-API calls).  usage: time_decode.py [size] [block]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "astc-encoder_amd", "python"))
import numpy as np, torch
import astcenc_amd as A
size = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
b = int(sys.argv[2]) if len(sys.argv) > 2 else 6
torch.zeros(1, device="cuda")
lib = A.Library(A.LIB_PRODUCT)
err, cfg = lib.config_init(A.PRF_LDR, b, b, 1, 10.0, 0); assert err == 0
err, ctx = lib.context_alloc(cfg, 1); assert err == 0
img = torch.from_numpy(A.synthetic_image(size, size)).cuda()
nb = ((size + b - 1) // b) ** 2
blocks = torch.zeros(nb * 16, dtype=torch.uint8, device="cuda")
swz = A.Swizzle(*A.SWZ_RGBA); ms = ctypes.c_float()
assert lib.lib.astcenc_amd_compress_image_device(ctx, img.data_ptr(), size, size, 0, ctypes.byref(swz), blocks.data_ptr(), blocks.numel(), None, ctypes.byref(ms)) == 0
for dtype, tname, tid in ((torch.uint8, "U8", A.TYPE_U8), (torch.float16, "F16", A.TYPE_F16)):
    dec = torch.zeros((size, size, 4), dtype=dtype, device="cuda")
    best = 1e9
    for i in range(4):
        torch.cuda.synchronize(); t = time.perf_counter()
        assert lib.lib.astcenc_amd_decompress_image_device(ctx, blocks.data_ptr(), blocks.numel(), dec.data_ptr(), size, size, 1, tid, ctypes.byref(swz), None) == 0
        dt = time.perf_counter() - t
        if i: best = min(best, dt)
    nbytes = nb * 16 + dec.numel() * dec.element_size()
    print("decode %dx%d %dx%d -> %s: %.2f ms, %.0f Mtexels/s, %.1f GB/s of %d MB algorithmic traffic" % (size, size, b, b, tname, best * 1e3, size * size / best / 1e6, nbytes / best / 1e9, nbytes >> 20))
    if tid == A.TYPE_U8:
        sums = A.ErrorSums(); best = 1e9
        for i in range(4):
            torch.cuda.synchronize(); t = time.perf_counter()
            assert lib.lib.astcenc_amd_compare_images_device(ctx, img.data_ptr(), 0, dec.data_ptr(), 0, size, size, 1, None, ctypes.byref(sums)) == 0
            dt = time.perf_counter() - t
            if i: best = min(best, dt)
        print("compare %dx%d U8/U8: %.2f ms, %.1f GB/s, PSNR %.4f dB" % (size, size, best * 1e3, 2 * size * size * 4 / best / 1e9, sums.psnr()))
