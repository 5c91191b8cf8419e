#!/usr/bin/env python3
"""Wide bit-exactness sweep on the GPU box: every 2D footprint x presets x image classes x profiles,
HIP library vs the reference (AVX2 build, all host threads; byte-identical to astcenc-none by the
reference's invariance guarantee).  Prints one line per mismatching case and a summary."""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "astc-encoder_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import astcenc_amd as A, images
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle")); import oracle_libs as O  # noqa: E402  (checker libraries: test infrastructure)

FOOT = [(4, 4), (5, 4), (5, 5), (6, 5), (6, 6), (8, 5), (8, 6), (8, 8), (10, 5), (10, 6), (10, 8), (10, 10), (12, 10), (12, 12)]
PRESETS = [("fastest", 0.0), ("fast", 10.0), ("medium", 60.0), ("thorough", 98.0)]
SIZE = int(sys.argv[1]) if len(sys.argv) > 1 else 192
gpu = A.Library(A.LIB_PRODUCT); ref = A.Library(O.LIB_REF_AVX2)
threads = len(os.sched_getaffinity(0))


def ref_compress(img, block, quality, profile, flags=0):
    err, cfg = ref.config_init(profile, block[0], block[1], 1, quality, flags); assert err == 0
    err, ctx = ref.context_alloc(cfg, threads); assert err == 0
    h, w = img.shape[:2]
    out = np.zeros(((w + block[0] - 1) // block[0]) * ((h + block[1] - 1) // block[1]) * 16, dtype=np.uint8)
    img = np.ascontiguousarray(img)
    ts = [threading.Thread(target=lambda i=i: ref.compress_raw(ctx, img, out, thread_index=i)) for i in range(threads)]
    [t.start() for t in ts]; [t.join() for t in ts]
    ref.context_free(ctx)
    return out


cases = bad = blocks = 0
t0 = time.time()
rng = np.random.default_rng(2024)
ldr_images = {"noisy": images.noisy(SIZE, SIZE - 7, 21), "random": images.random_u8(SIZE - 5, SIZE, 22), "two_colour": images.two_colour(SIZE, SIZE, 23),
              "gray": images.grayscale(SIZE, SIZE), "flat": images.flat_regions(SIZE, SIZE), "smooth": images.smooth(SIZE, SIZE)}
for block in FOOT:
    for pname, q in PRESETS:
        for name, img in ldr_images.items():
            for profile in ((A.PRF_LDR, A.PRF_LDR_SRGB) if name == "noisy" else (A.PRF_LDR,)):
                want = ref_compress(img, block, q, profile)
                got = gpu.compress(img, block, q, profile=profile)
                n = int((want.reshape(-1, 16) != got.reshape(-1, 16)).any(axis=1).sum())
                cases += 1; blocks += want.size // 16
                if n:
                    bad += 1
                    print("MISMATCH %dx%d %s %s profile %d: %d of %d blocks" % (block[0], block[1], pname, name, profile, n, want.size // 16), flush=True)
    # HDR profiles, medium only
    for profile in (A.PRF_HDR, A.PRF_HDR_RGB_LDR_A):
        for name, img in images.hdr_variants(96, 90).items():
            img = img.astype(np.float16)
            want = ref_compress(img, block, 60.0, profile)
            got = gpu.compress(img, block, 60.0, profile=profile)
            n = int((want.reshape(-1, 16) != got.reshape(-1, 16)).any(axis=1).sum())
            cases += 1; blocks += want.size // 16
            if n:
                bad += 1
                print("MISMATCH %dx%d medium hdr:%s profile %d: %d of %d blocks" % (block[0], block[1], name, profile, n, want.size // 16), flush=True)
# exhaustive on a small image, three footprints
for block in ((4, 4), (6, 6), (8, 8)):
    img = images.noisy(48, 48, 31)
    want = ref_compress(img, block, 100.0, A.PRF_LDR)
    got = gpu.compress(img, block, 100.0)
    n = int((want.reshape(-1, 16) != got.reshape(-1, 16)).any(axis=1).sum())
    cases += 1; blocks += want.size // 16
    if n:
        bad += 1; print("MISMATCH %dx%d exhaustive: %d blocks" % (block[0], block[1], n))
print("sweep: %d cases, %d blocks, %d mismatching cases, %.0f s" % (cases, blocks, bad, time.time() - t0))
