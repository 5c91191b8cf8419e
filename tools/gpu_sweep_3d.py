#!/usr/bin/env python3
"""Bit-exactness sweep of the 3D footprints on the GPU box: every 3D footprint x presets x volume classes,
HIP library vs the reference (AVX2 build, all host threads), then kernel timings of a device-resident
volume.  usage: gpu_sweep_3d.py [edge]   (volume edge in texels for the sweep, default 40)"""
import ctypes, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "astc-encoder_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import astcenc_amd as A, images
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle")); import oracle_libs as O  # noqa: E402  (checker libraries: test infrastructure)

torch.zeros(1, device="cuda")
FOOT = [(3, 3, 3), (4, 3, 3), (4, 4, 3), (4, 4, 4), (5, 4, 4), (5, 5, 4), (5, 5, 5), (6, 5, 5), (6, 6, 5), (6, 6, 6)]
PRESETS = [("fastest", 0.0), ("fast", 10.0), ("medium", 60.0), ("thorough", 98.0)]
EDGE = int(sys.argv[1]) if len(sys.argv) > 1 else 40
gpu = A.Library(A.LIB_PRODUCT); ref = A.Library(O.LIB_REF_AVX2)
threads = len(os.sched_getaffinity(0))


def ref_compress(vol, block, quality, profile=A.PRF_LDR):
    err, cfg = ref.config_init(profile, block[0], block[1], block[2], quality, 0); assert err == 0
    err, ctx = ref.context_alloc(cfg, threads); assert err == 0
    d, h, w = vol.shape[:3]
    out = np.zeros(((w + block[0] - 1) // block[0]) * ((h + block[1] - 1) // block[1]) * ((d + block[2] - 1) // block[2]) * 16, dtype=np.uint8)
    ts = [threading.Thread(target=lambda i=i: ref.compress_raw(ctx, vol, out, thread_index=i)) for i in range(threads)]
    [t.start() for t in ts]; [t.join() for t in ts]
    ref.context_free(ctx)
    return out


cases = bad = blocks = 0
t0 = time.time()
vols = {k: images.volume(k, EDGE - 3, EDGE, EDGE + 5, seed=40 + i) for i, k in enumerate(("noise", "grad", "edges", "alpha", "flat"))}
for block in FOOT:
    for pname, q in PRESETS:
        for name, vol in vols.items():
            want = ref_compress(vol, block, q)
            got = gpu.compress(vol, block, q)
            n = int((want.reshape(-1, 16) != got.reshape(-1, 16)).any(axis=1).sum())
            cases += 1; blocks += want.size // 16
            if n:
                bad += 1
                print("MISMATCH %dx%dx%d %s %s: %d of %d blocks" % (block + (pname, name, n, want.size // 16)), flush=True)
print("3D sweep: %d cases, %d blocks, %d mismatching cases, %.0f s" % (cases, blocks, bad, time.time() - t0), flush=True)

# kernel time on a device-resident 256^3 RGBA8 volume (64 MiB), -medium
N = 256
vol = np.ascontiguousarray(A.synthetic_image(N, N * N).reshape(N, N, N, 4))
d_vol = torch.from_numpy(vol).cuda()
swz = A.Swizzle(*A.SWZ_RGBA); ms = ctypes.c_float()
for block in ((3, 3, 3), (4, 4, 4), (6, 6, 6)):
    err, cfg = gpu.config_init(A.PRF_LDR, block[0], block[1], block[2], 60.0, 0); assert err == 0
    err, ctx = gpu.context_alloc(cfg, 1); assert err == 0
    nb = ((N + block[0] - 1) // block[0]) * ((N + block[1] - 1) // block[1]) * ((N + block[2] - 1) // block[2])
    out = torch.zeros(nb * 16, dtype=torch.uint8, device="cuda")
    best = 1e9
    for i in range(3):
        e = gpu.lib.astcenc_amd_compress_volume_device(ctx, d_vol.data_ptr(), N, N, N, 0, ctypes.byref(swz), out.data_ptr(), out.numel(),
                                                       torch.cuda.current_stream().cuda_stream, ctypes.byref(ms))
        assert e == 0
        if i: best = min(best, ms.value)
    print("%d^3 RGBA8 %dx%dx%d -medium: kernel %.1f ms -> %.1f Mtexels/s" % ((N,) + block + (best, N ** 3 / best / 1e3)), flush=True)
    gpu.context_free(ctx)
