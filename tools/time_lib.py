#!/usr/bin/env python3
"""Time the compression kernel of a given library build on a device-resident synthetic image.
usage: time_lib.py <lib.so> [size] [block] [quality] [steps]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["ASTCENC_AMD_LIB"] = os.path.abspath(sys.argv[1])
sys.path.insert(0, os.path.join(ROOT, "astc-encoder_amd", "python"))
import numpy as np, torch
import astcenc_amd as A
size = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
b = int(sys.argv[3]) if len(sys.argv) > 3 else 6
q = float(sys.argv[4]) if len(sys.argv) > 4 else 60.0
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
lib = A.Library(A.LIB_PRODUCT)
err, cfg = lib.config_init(A.PRF_LDR, b, b, 1, q, 0); assert err == 0
err, ctx = lib.context_alloc(cfg, 1); assert err == 0, err
img = torch.from_numpy(A.synthetic_image(size, size)).cuda()
nb = ((size + b - 1) // b) ** 2
out = torch.zeros(nb * 16, dtype=torch.uint8, device="cuda")
swz = A.Swizzle(*A.SWZ_RGBA); ms = ctypes.c_float()
best = 1e9
for i in range(steps + 1):
    e = lib.lib.astcenc_amd_compress_image_device(ctx, img.data_ptr(), size, size, 0, ctypes.byref(swz), out.data_ptr(), out.numel(), torch.cuda.current_stream().cuda_stream, ctypes.byref(ms))
    assert e == 0
    if i > 0: best = min(best, ms.value)
print("%s [%s]: %dx%d %dx%d q=%.0f kernel %.2f ms -> %.2f Mtexels/s" % (os.path.basename(sys.argv[1]), lib.lib.astcenc_amd_context_kernel_name(ctx).decode(), size, size, b, b, q, best, size * size / best / 1e3))
if os.environ.get("CHECK", "1") != "0":
    # byte parity of a block-aligned 384^2 corner against the reference (checker only)
    sys.path.insert(0, os.path.join(ROOT, "oracle")); import oracle_libs as O
    if os.path.exists(O.LIB_REF_AVX2):
        n = 384 // b
        crop = np.ascontiguousarray(A.synthetic_image(size, size)[:n * b, :n * b])
        want = A.Library(O.LIB_REF_AVX2).compress(crop, (b, b), q).reshape(-1, 16)
        bxn = (size + b - 1) // b
        got = out.cpu().numpy().reshape(-1, 16)
        rows = np.concatenate([got[r * bxn: r * bxn + n] for r in range(n)])
        print("   parity: %d of %d blocks differ from the reference" % (int((rows != want).any(axis=1).sum()), want.shape[0]))
