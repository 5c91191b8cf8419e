#!/usr/bin/env python3
"""Same-process A/B of the decompression kernel: the streams the product library produces (-medium, several footprints,
RGBA8 and RGBA16F output, an HDR stream) decoded by every library named on the command line; per library the best wall
time of the synchronous device call, the time between two events on the stream around it, and whether the decoded image
equals the first library's byte for byte.  usage: [AB_CASES=n] time_decode_ab.py <size> <lib.so> [<lib.so> ...]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "astc-encoder_amd", "python"))
import numpy as np, torch
import astcenc_amd as A
size = int(sys.argv[1])
paths = sys.argv[2:] or [A.LIB_PRODUCT]
torch.zeros(1, device="cuda")
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream(dev).cuda_stream
libs = [(os.path.basename(p), A.Library(p)) for p in paths]


def synthetic_image_on_device(n, seed=0x9E3779B1):
    """A.synthetic_image(n, n) computed with torch integer arithmetic on the device (the host generator takes half a
    minute at 8192^2); checked against the host generator on a corner below."""
    y, x = torch.meshgrid(torch.arange(n, dtype=torch.int64, device="cuda"), torch.arange(n, dtype=torch.int64, device="cuda"), indexing="ij")
    def tri(v):
        m = v & 511
        return torch.where(m < 256, m, 511 - m)
    r = (3 * tri(x + 2 * y) + tri((3 * x - y) >> 1)) >> 2
    checker = ((x >> 5) + (y >> 5)) & 1
    r = torch.where(checker == 1, 255 - r, r)
    g = (3 * tri(2 * x - y + 128) + tri((x + 3 * y) >> 2)) >> 2
    b = 255 - ((tri(x + y) + tri((x - y) >> 1)) >> 1)
    a = 192 + (tri((x >> 1) + (y >> 2)) >> 2)
    def noise(c, amp):
        h = (x * 0x85EBCA6B + y * 0xC2B2AE35 + c * 0x27D4EB2F + seed) & 0xFFFFFFFF
        h = h ^ (h >> 15)
        h = (h * 0x2C1B3C6D) & 0xFFFFFFFF
        h = h ^ (h >> 12)
        h = (h * 0x297A2D39) & 0xFFFFFFFF
        h = h ^ (h >> 15)
        return (h % (2 * amp + 1)) - amp
    out = torch.stack([r + noise(0, 10), g + noise(1, 10), b + noise(2, 10), a + noise(3, 3)], dim=-1)
    return out.clamp(0, 255).to(torch.uint8).contiguous()


ldr = synthetic_image_on_device(size)
assert np.array_equal(ldr[:64, :96].cpu().numpy(), A.synthetic_image(size, size)[:64, :96]) if size <= 1024 else np.array_equal(ldr[:64, :96].cpu().numpy(), A.synthetic_image(96, 64))
hsize = min(size, 2048)
hdr = torch.from_numpy(A.synthetic_hdr_image(hsize, hsize).astype(np.float16)).cuda()
swz = A.Swizzle(*A.SWZ_RGBA)
CASES = [("ldr 6x6 -medium -> U8", A.PRF_LDR, 6, 60.0, ldr, size, A.TYPE_U8, torch.uint8),
         ("ldr 8x8 -thorough -> U8", A.PRF_LDR, 8, 98.0, ldr, size, A.TYPE_U8, torch.uint8),
         ("ldr 4x4 -fast -> U8", A.PRF_LDR, 4, 10.0, ldr, size, A.TYPE_U8, torch.uint8),
         ("ldr 12x12 -medium -> U8", A.PRF_LDR, 12, 60.0, ldr, size, A.TYPE_U8, torch.uint8),
         ("ldr 6x6 -medium -> F16", A.PRF_LDR, 6, 60.0, ldr, size, A.TYPE_F16, torch.float16),
         ("hdr 6x6 -medium -> F16", A.PRF_HDR, 6, 60.0, hdr, hsize, A.TYPE_F16, torch.float16)]
for name, profile, b, quality, img, n, ttype, tdtype in CASES[:int(os.environ.get("AB_CASES", len(CASES)))]:
    first = None
    blocks = None
    for lname, lib in libs:
        err, cfg = lib.config_init(profile, b, b, 1, quality, 0); assert err == 0
        err, ctx = lib.context_alloc(cfg, 1); assert err == 0
        nb = ((n + b - 1) // b) ** 2
        if blocks is None:
            # (the stream every library decodes comes from the last library named: the product)
            plib = libs[-1][1]
            err, pcfg = plib.config_init(profile, b, b, 1, quality, 0); assert err == 0
            err, pctx = plib.context_alloc(pcfg, 1); assert err == 0
            blocks = torch.zeros(nb * 16, dtype=torch.uint8, device="cuda")
            ms = ctypes.c_float()
            assert plib.lib.astcenc_amd_compress_image_device(pctx, img.data_ptr(), n, n, 0 if img.dtype == torch.uint8 else 1, ctypes.byref(swz),
                                                              blocks.data_ptr(), blocks.numel(), None, ctypes.byref(ms)) == 0
            plib.context_free(pctx)
        dec = torch.zeros((n, n, 4), dtype=tdtype, device="cuda")
        wall = ev = 1e9
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for i in range(6):
            torch.cuda.synchronize(); t = time.perf_counter()
            e0.record()
            assert lib.lib.astcenc_amd_decompress_image_device(ctx, blocks.data_ptr(), blocks.numel(), dec.data_ptr(), n, n, 1, ttype, ctypes.byref(swz), stream) == 0
            e1.record(); torch.cuda.synchronize()
            dt = time.perf_counter() - t
            if i: wall = min(wall, dt * 1e3); ev = min(ev, e0.elapsed_time(e1))
        raw = dec.view(torch.uint8)
        same = "reference" if first is None else ("identical" if torch.equal(raw, first) else "DIFFERENT (%d bytes)" % int((raw != first).sum().item()))
        if first is None: first = raw.clone()
        nbytes = nb * 16 + dec.numel() * dec.element_size()
        print("%-26s %-28s wall %.3f ms  events %.3f ms  %.0f GB/s (events), %.1f %% of 8 TB/s; output %s" %
              (name, lname, wall, ev, nbytes / ev / 1e6, nbytes / ev / 1e6 / 80.0, same), flush=True)
        lib.context_free(ctx)
