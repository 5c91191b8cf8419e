#!/usr/bin/env python3
"""Wall-clock rate of the host-pointer API (astcenc_compress_image: pageable host memory in, host memory out,
PCIe included) next to the kernel-only rate of the device API.  usage: time_host_api.py [size] [block] [quality]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "astc-encoder_amd", "python"))
import numpy as np, torch
import astcenc_amd as A
size = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
b = int(sys.argv[2]) if len(sys.argv) > 2 else 6
q = float(sys.argv[3]) if len(sys.argv) > 3 else 60.0
torch.zeros(1, device="cuda")
lib = A.Library(A.LIB_PRODUCT)
err, cfg = lib.config_init(A.PRF_LDR, b, b, 1, q, 0); assert err == 0
err, ctx = lib.context_alloc(cfg, 1); assert err == 0
img = A.synthetic_image(size, size)
nb = ((size + b - 1) // b) ** 2
out = np.zeros(nb * 16, dtype=np.uint8)
best = 1e9
for i in range(3):
    t = time.perf_counter()
    e = lib.compress_raw(ctx, img, out)
    dt = time.perf_counter() - t
    assert e == 0
    if i: best = min(best, dt)
d_img = torch.from_numpy(img).cuda(); d_out = torch.zeros(nb * 16, dtype=torch.uint8, device="cuda")
swz = A.Swizzle(*A.SWZ_RGBA); ms = ctypes.c_float()
for i in range(2):
    e = lib.lib.astcenc_amd_compress_image_device(ctx, d_img.data_ptr(), size, size, 0, ctypes.byref(swz), d_out.data_ptr(), d_out.numel(), None, ctypes.byref(ms))
    assert e == 0
same = bool((d_out.cpu().numpy() == out).all())
print("%dx%d %dx%d q=%.0f: host API %.1f ms (%.2f Mtexels/s incl. PCIe), kernel only %.1f ms (%.2f Mtexels/s), outputs identical: %s" %
      (size, size, b, b, q, best * 1e3, size * size / best / 1e6, ms.value, size * size / ms.value / 1e3, same))
