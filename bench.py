#!/usr/bin/env python3
# SPDX-License-Identifier: Apache-2.0
"""Headline benchmark: Mtexels/s of ASTC LDR 6x6 -medium compression of an 8192x8192 RGBA8 image
(BASELINE.json configs[1]) on MI355X, inputs resident in HBM.

  python bench.py --gpus N --steps K --warmup W [--config c2|c3|c4|c1]
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)
  python bench.py --gpus N --single-process     one process, the library's own N-device split behind the C ABI

A "step" is one pass of the block compressor over one image per rank.  The path shards by independent
images / block rows with no data-path collective, so N ranks compress N images: weak
scaling, value = N * texels * K / max-over-ranks time.  torch is used for device memory, the stream and
the rank barrier only; the work is libastcenc_amd.so called through its C ABI
(astcenc_amd_compress_image_device).

Rank 0 prints ONE JSON line with the driver's fields plus:
  roofline       : algorithmic HBM bytes per launch / mean kernel time (HIP events recorded by the library on
                   the launch stream) vs the 8 TB/s HBM peak; `traffic` (HBM bytes per launch from the PMC passes) and
                   `valu` (the bound that actually binds: wave instructions per block, active lanes, and the issue rate
                   of THIS run = committed instruction count x blocks / measured kernel time, against the chip's
                   1.23e12 wave-instructions/s) come from the latest committed rocprofv3 PMC passes of the same
                   workload (profiles/*/traffic.json, named in traffic_source: rocprofv3 cannot run inside this process).
  parity_full    : config 2 only: the WHOLE 8192^2 image through the reference's AVX2 build, every one of the
                   1 865 956 blocks compared with the GPU's (ref: the block loop of astcenc_entry.cpp:1009-1038).
  value_host_api : the same image through astcenc_compress_image (host pointers, PCIe both ways included).
  cpu_baseline   : the reference encoder's AVX2 build (oracle/_ref/libastcenc-avx2.so; the build with
                   ASTCENC_X86_GATHERS=1, the reference's x86 default, is timed too and the faster one reported) on this
                   box's host cores: thread-count sweep on a 2048^2 crop, best of 3 with astcenc_compress_reset in
                   between, plus the 1-thread rate, the rate per host core-second, and the host's CPU model / cgroup
                   quota / load; then a byte comparison of the crop against the GPU output.  N = 1 only.
  extra_configs  : BASELINE configs[2] (8192^2 8x8 -thorough) and configs[3] (4096^2 RGBA16F HDR 6x6 -medium):
                   kernel time, Mtexels/s, roofline, byte parity of the WHOLE stream against the reference (crop + tail if
                   a probe says that would take more than a minute of host time), and (HDR) mPSNR / log RMSE from the
                   on-device comparison; and `photo`: the reference's Khronos test images tiled to 4096^2, 6x6 -medium,
                   with the reference's AVX2 rate on the same image and a whole-stream byte comparison.
"""
import argparse
import ctypes
import json
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "astc-encoder_amd", "python"))
import astcenc_amd as A  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle_libs as O  # noqa: E402  (checker libraries: used by the cpu_baseline / parity legs only, never inside the timed region)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
# VALU issue peak of the chip in wave-instructions per second: 1024 SIMDs x 2.4 GHz / 2 clocks per wave64 instruction
# (MI355X_MICROARCH.md: 256 CUs x 4 SIMDs, 32 lanes per clock; packed / dual-issue forms aside)
VALU_PEAK_WAVE_INSTS_S = 1024 * 2.4e9 / 2.0

# BASELINE.json configs (C1..C4); `index` = position in BASELINE.json's list
CONFIGS = {
    "c1": dict(index=0, size=512, block=(4, 4), quality=A.PRE_FASTEST, profile="PRF_LDR", hdr=False, plimit=1,
               label="512x512 RGBA8 LDR, 4x4 block, -fastest, 1 partition"),
    "c2": dict(index=1, size=8192, block=(6, 6), quality=A.PRE_MEDIUM, profile="PRF_LDR", hdr=False, plimit=None,
               label="8192x8192 RGBA8 LDR, 6x6 block, -medium"),
    "c3": dict(index=2, size=8192, block=(8, 8), quality=A.PRE_THOROUGH, profile="PRF_LDR", hdr=False, plimit=None,
               label="8192x8192 RGBA8 LDR, 8x8 block, -thorough"),
    "c4": dict(index=3, size=4096, block=(6, 6), quality=A.PRE_MEDIUM, profile="PRF_HDR", hdr=True, plimit=None,
               label="4096x4096 RGBA16F HDR, 6x6 block, -medium"),
}


def make_image(cfg, seed):
    if cfg["hdr"]:
        return A.synthetic_hdr_image(cfg["size"], cfg["size"], seed)
    return A.synthetic_image(cfg["size"], cfg["size"], seed)


def make_context(lib, cfg, threads=1):
    err, c = lib.config_init(getattr(A, cfg["profile"]), cfg["block"][0], cfg["block"][1], 1, cfg["quality"], 0)
    assert err == 0, err
    if cfg["plimit"]:
        c.tune_partition_count_limit = cfg["plimit"]
    err, ctx = lib.context_alloc(c, threads)
    assert err == 0, "context_alloc failed: %s" % lib.error_string(err)
    return ctx


def block_grid(cfg):
    s, (bx, by) = cfg["size"], cfg["block"]
    return (s + bx - 1) // bx, (s + by - 1) // by


def algorithmic_bytes(cfg):
    """every texel read once (4 B RGBA8, 8 B RGBA16F), 16 B per block written once."""
    nbx, nby = block_grid(cfg)
    return cfg["size"] * cfg["size"] * (8 if cfg["hdr"] else 4) + nbx * nby * 16


def to_device(img, dev):
    if img.dtype == np.float16:
        return torch.from_numpy(img.view(np.int16)).to(dev)
    return torch.from_numpy(img).to(dev)


# ---------------------------------------------------------------------------------------------------------
# checker legs (never timed as `value`)
# ---------------------------------------------------------------------------------------------------------

def reference_threads(ref, cfg, crop, threads, repeats):
    """astcenc_compress_image of the reference on `threads` caller threads (the API's own threading model,
    ref: astcenc.h:785-803), best of `repeats` with astcenc_compress_reset in between.  Returns (best s, blocks)."""
    ctx = make_context(ref, cfg, threads)
    h, w = crop.shape[:2]
    bx, by = cfg["block"]
    out = np.zeros(((w + bx - 1) // bx) * ((h + by - 1) // by) * 16, dtype=np.uint8)
    best = 1e30
    for _ in range(repeats):
        errs = [0] * threads
        workers = [threading.Thread(target=lambda i=i: errs.__setitem__(i, ref.compress_raw(ctx, crop, out, thread_index=i)))
                   for i in range(threads)]
        t0 = time.perf_counter()
        for t in workers:
            t.start()
        for t in workers:
            t.join()
        best = min(best, time.perf_counter() - t0)
        assert not any(errs), errs
        assert ref.lib.astcenc_compress_reset(ctx) == 0
    ref.context_free(ctx)
    return best, out


def host_description():
    info = {"nproc": os.cpu_count(), "affinity": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None}
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                info["cpu_model"] = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            info["cgroup_cpu_max"] = open(path).read().strip()
            break
        except OSError:
            continue
    try:
        info["loadavg"] = open("/proc/loadavg").read().split()[:3]
    except OSError:
        pass
    return info


def crop_blocks(gpu_blocks, cfg, x0, y0, w, h):
    """Blocks of the whole-image stream that cover texels [x0, x0+w) x [y0, y0+h) (origin on a block boundary)."""
    bx, by = cfg["block"]
    nbx, _ = block_grid(cfg)
    g = gpu_blocks.reshape(-1, 16)
    cx0, cy0, cw, ch = x0 // bx, y0 // by, (w + bx - 1) // bx, (h + by - 1) // by
    return np.concatenate([g[(cy0 + r) * nbx + cx0: (cy0 + r) * nbx + cx0 + cw] for r in range(ch)])


def cpu_reference_baseline(cfg, img, gpu_blocks, budget_s=30.0):
    """Thread sweep of the reference AVX2 encoder + byte parity against the GPU output."""
    if not os.path.exists(O.LIB_REF_AVX2):
        return None
    ref = A.Library(O.LIB_REF_AVX2)
    host = host_description()
    cores = host["affinity"] or host["nproc"] or 1
    t_start = time.perf_counter()
    bx, by = cfg["block"]

    # 1 thread on a 512^2 crop (a 2048^2 crop would take ~15 s per repeat on one core)
    small = np.ascontiguousarray(img[:510 // by * by, :510 // bx * bx])
    t1, _ = reference_threads(ref, cfg, small, 1, 3)
    rate1 = small.shape[0] * small.shape[1] / t1 / 1e6

    side = 2048 // by * by
    crop = np.ascontiguousarray(img[:side, :side])
    sweep = {}
    best_rate, best_threads, out_best = 0.0, 1, None
    for threads in sorted({8, 32, 64, 128, cores} | ({cores // 2} if cores >= 16 else set())):
        if threads > max(cores, 8) or threads < 1:
            continue
        if time.perf_counter() - t_start > budget_s and sweep:
            break
        dt, out = reference_threads(ref, cfg, crop, threads, 3)
        rate = side * side / dt / 1e6
        sweep[str(threads)] = round(rate, 3)
        if rate > best_rate:
            best_rate, best_threads, out_best = rate, threads, out
    want = crop_blocks(gpu_blocks, cfg, 0, 0, side, side)
    mismatch = int((want != out_best.reshape(-1, 16)).any(axis=1).sum())
    # the same crop, same thread count, through the build with hardware gathers (the reference's x86 default,
    # CMakeLists.txt:55); the faster of the two is the baseline
    gathers = {"0": round(best_rate, 3)}
    used_gathers = 0
    if os.path.exists(O.LIB_REF_AVX2_GATHERS):
        dt, out_g = reference_threads(A.Library(O.LIB_REF_AVX2_GATHERS), cfg, crop, best_threads, 3)
        gathers["1"] = round(side * side / dt / 1e6, 3)
        mismatch += int((want != out_g.reshape(-1, 16)).any(axis=1).sum())
        if gathers["1"] > best_rate:
            best_rate, used_gathers = gathers["1"], 1
    # ... and through the link-time optimised build (the reference's release setting for its command line tool,
    # Source/cmake_core.cmake:258-262)
    lto = None
    if os.path.exists(O.LIB_REF_AVX2_LTO):
        dt, out_l = reference_threads(A.Library(O.LIB_REF_AVX2_LTO), cfg, crop, best_threads, 3)
        lto = round(side * side / dt / 1e6, 3)
        mismatch += int((want != out_l.reshape(-1, 16)).any(axis=1).sum())
        if lto > best_rate:
            best_rate, used_gathers = lto, 0
    quota = host.get("cgroup_cpu_max", "").split()
    cpus_allowed = min(cores, float(quota[0]) / float(quota[1])) if len(quota) == 2 and quota[0] != "max" else float(cores)
    return {"value": round(best_rate, 3), "unit": "Mtexels/s", "cores": best_threads, "kind": "reference",
            "sample": "astcenc-avx2 (oracle/_ref), %dx%d top-left crop of the bench image, best of 3 per thread count with "
                      "astcenc_compress_reset in between; 1-thread figure on a %dx%d crop" % (side, side, small.shape[1], small.shape[0]),
            "x86_gathers_mtexels_s": gathers, "x86_gathers_used": used_gathers, "lto_build_mtexels_s": lto,
            "build": "g++ -O3 -mavx2 shared library from /root/reference/Source (oracle/Makefile); fastest of {no gathers, hardware gathers, -flto}",
            "cpus_allowed": round(cpus_allowed, 2), "mtexels_per_core_second": round(best_rate / max(min(cpus_allowed, best_threads), 1e-9), 4),
            "threads_at_best": best_threads, "value_1thread": round(rate1, 4), "per_thread_at_best": round(best_rate / best_threads, 4),
            "thread_sweep_mtexels_s": sweep, "cpu_model": host.get("cpu_model"), "nproc": host["nproc"], "affinity": host["affinity"],
            "cgroup_cpu_max": host.get("cgroup_cpu_max"), "loadavg": host.get("loadavg"),
            "blocks_compared_with_gpu": int(want.shape[0]), "blocks_mismatching_gpu": mismatch,
            "seconds": round(time.perf_counter() - t_start, 1)}


def parity_full(cfg, img, gpu_blocks, threads, gathers):
    """Every block of the whole image against the reference (AVX2 build, byte-identical to the scalar build by the
    reference's invariance mode -- tests/ pin that) at the thread count the baseline sweep found best."""
    path = O.LIB_REF_AVX2_GATHERS if gathers and os.path.exists(O.LIB_REF_AVX2_GATHERS) else O.LIB_REF_AVX2
    if not os.path.exists(path):
        return None
    dt, out = reference_threads(A.Library(path), cfg, img, threads, 1)
    want, got = out.reshape(-1, 16), gpu_blocks.reshape(-1, 16)
    bad = np.nonzero((want != got).any(axis=1))[0]
    return {"blocks_compared": int(want.shape[0]), "mismatch": int(bad.size), "first_mismatching_blocks": [int(b) for b in bad[:8]],
            "reference": os.path.basename(path), "threads": threads, "seconds": round(dt, 2),
            "reference_mtexels_s_whole_image": round(img.shape[0] * img.shape[1] / dt / 1e6, 3)}


def parity_crops(cfg, img, gpu_blocks, budget_s):
    """Byte parity of (a) a block-aligned top-left crop and (b) the bottom-right tail (clamped edge blocks included)
    against the reference run on those crops alone.  The crop grows until ~budget_s of host time is spent."""
    if not os.path.exists(O.LIB_REF_AVX2):
        return None
    ref = A.Library(O.LIB_REF_AVX2)
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    bx, by = cfg["block"]
    size = cfg["size"]
    t0 = time.perf_counter()
    probe = 256 // by * by
    dt, _ = reference_threads(ref, cfg, np.ascontiguousarray(img[:probe, :probe]), cores, 1)
    rate = probe * probe / max(dt, 1e-3)
    side = int(min(size // 2, max(probe, (rate * budget_s / 2.0) ** 0.5)))
    side_y, side_x = side // by * by, side // bx * bx
    res = {"threads": cores}
    # (a) top-left
    dt, out = reference_threads(ref, cfg, np.ascontiguousarray(img[:side_y, :side_x]), cores, 1)
    want = crop_blocks(gpu_blocks, cfg, 0, 0, side_x, side_y)
    res["crop"] = "%dx%d at (0,0)" % (side_x, side_y)
    res["crop_blocks"] = int(want.shape[0])
    res["crop_mismatch"] = int((want != out.reshape(-1, 16)).any(axis=1).sum())
    # (b) tail: from a block-aligned origin to the image's last texel
    tx0, ty0 = (size - side_x) // bx * bx, (size - side_y) // by * by
    dt, out = reference_threads(ref, cfg, np.ascontiguousarray(img[ty0:, tx0:]), cores, 1)
    want = crop_blocks(gpu_blocks, cfg, tx0, ty0, size - tx0, size - ty0)
    res["tail"] = "%dx%d at (%d,%d)" % (size - tx0, size - ty0, tx0, ty0)
    res["tail_blocks"] = int(want.shape[0])
    res["tail_mismatch"] = int((want != out.reshape(-1, 16)).any(axis=1).sum())
    res["seconds"] = round(time.perf_counter() - t0, 1)
    return res


def parity_whole_or_crops(cfg, img, gpu_blocks, budget_s):
    """Every block of the image against the reference when a probe says that fits `budget_s` of host time (it does on the
    GPU boxes seen so far), else the crop + tail comparison."""
    if not os.path.exists(O.LIB_REF_AVX2):
        return None
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    bx, by = cfg["block"]
    probe = 256 // by * by, 256 // bx * bx
    dt, _ = reference_threads(A.Library(O.LIB_REF_AVX2), cfg, np.ascontiguousarray(img[:probe[0], :probe[1]]), cores, 1)
    estimate = dt * img.shape[0] * img.shape[1] / (probe[0] * probe[1])
    if estimate <= budget_s:
        res = parity_full(cfg, img, gpu_blocks, cores, 0)
        res["coverage"] = "whole stream"
        return res
    res = parity_crops(cfg, img, gpu_blocks, budget_s / 4.0)
    res["coverage"] = "crop + tail (a whole-stream run was estimated at %.0f s of host time)" % estimate
    return res


def khronos_mosaic(size=4096):
    """The reference's Khronos test images (tests/corpus/_images, fetched by tests/corpus/make_corpus.py) tiled into one
    size x size RGBA8 image: the 2048^2 diffuse map in two quadrants, the 1024^2 maps (emissive, metal-roughness, base
    colour, specular-glossiness, normal, occlusion) in the other two.  None when the corpus is not there."""
    from PIL import Image
    d = os.path.join(ROOT, "tests", "corpus", "_images", "Khronos")
    names = ["LDR-RGB/ldr-rgb-diffuse.png", "LDR-RGB/ldr-rgb-emissive.png", "LDR-RGB/ldr-rgb-metalrough.png", "LDR-RGBA/ldr-rgba-base.png",
             "LDR-RGBA/ldr-rgba-diffuse.png", "LDR-RGBA/ldr-rgba-specgloss.png", "LDR-XY/ldr-xy-normal1.png", "LDR-L/ldr-l-occlusion.png"]
    if not all(os.path.exists(os.path.join(d, n)) for n in names):
        return None
    tiles = [np.array(Image.open(os.path.join(d, n)).convert("RGBA")) for n in names]
    big, small = tiles[0], tiles[1:]
    canvas = np.zeros((size, size, 4), dtype=np.uint8)
    q = size // 2
    k = 0
    for qy in range(2):
        for qx in range(2):
            if qx == qy:
                reps = (q + big.shape[0] - 1) // big.shape[0]
                canvas[qy * q:(qy + 1) * q, qx * q:(qx + 1) * q] = np.tile(big, (reps, reps, 1))[:q, :q]
            else:
                for ty in range(0, q, 1024):
                    for tx in range(0, q, 1024):
                        canvas[qy * q + ty: qy * q + ty + 1024, qx * q + tx: qx * q + tx + 1024] = small[k % len(small)][:min(1024, q - ty), :min(1024, q - tx)]
                        k += 1
    return np.ascontiguousarray(canvas)


def run_photo_config(lib, dev, steps, warmup):
    """Photographic content (VERDICT r03 "missing" 2: real content takes the early exits that noise never takes): the
    Khronos set tiled to 4096^2, 6x6 -medium, device-resident like the headline; the reference's AVX2 build on the same
    image beside it (whole stream, byte-compared)."""
    img = khronos_mosaic()
    if img is None:
        return {"config": "photo", "skipped": "tests/corpus/_images not present (tests/corpus/make_corpus.py needs /root/reference)"}
    cfg = dict(CONFIGS["c2"], size=img.shape[0], label="Khronos test images tiled to %dx%d RGBA8 LDR, 6x6 block, -medium" % img.shape[:2])
    ctx = make_context(lib, cfg)
    d_img = to_device(img, dev)
    nbx, nby = block_grid(cfg)
    d_out = torch.zeros(nbx * nby * 16, dtype=torch.uint8, device=dev)
    elapsed, kms = time_device_resident(lib, ctx, cfg, d_img, d_out, dev, steps, warmup, lambda: torch.cuda.synchronize(dev))
    gpu_blocks = d_out.cpu().numpy()
    texels = img.shape[0] * img.shape[1]
    value = texels * steps / elapsed / 1e6
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    par = parity_full(cfg, img, gpu_blocks, cores, 0)
    res = {"config": "photo", "workload": cfg["label"], "data": "reference Test/Images/Khronos", "value": round(value, 3), "unit": "Mtexels/s",
           "steps": steps, "warmup": warmup, "ms_per_step": round(elapsed / steps * 1e3, 3), "blocks_per_image": nbx * nby,
           "kernel_ms": round(sum(kms) / len(kms), 3), "parity_vs_reference": par,
           "quality": device_quality(lib, ctx, cfg, d_img, d_out, dev)}
    if par:
        res["cpu_baseline"] = {"value": par["reference_mtexels_s_whole_image"], "unit": "Mtexels/s", "cores": par["threads"], "kind": "reference",
                               "sample": "astcenc-avx2 (oracle/_ref) on the whole mosaic, one run"}
        res["speedup_vs_cpu_baseline"] = round(value / max(par["reference_mtexels_s_whole_image"], 1e-9), 2)
    lib.context_free(ctx)
    return res


def decoded_psnr(cfg, img, gpu_blocks):
    """PSNR (dB, RGBA, reference definition astcenccli_error_metrics.cpp:240-346) of the GPU's blocks
    decoded by the independent plain-C decoder in oracle/ (checker only, never timed)."""
    path = os.path.join(ROOT, "oracle", "_build", "libastc_decode.so")
    if not os.path.exists(path) or cfg["hdr"]:
        return None
    dec = ctypes.CDLL(path)
    dec.astc_oracle_decode_image.restype = ctypes.c_int
    dec.astc_oracle_decode_image.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 5 + [ctypes.c_void_p]
    h, w = img.shape[:2]
    out = np.zeros((h, w, 4), dtype=np.uint8)
    errors = dec.astc_oracle_decode_image(gpu_blocks.ctypes.data, cfg["block"][0], cfg["block"][1], w, h, 0, out.ctypes.data)
    sq = 0.0
    for y in range(0, h, 1024):             # chunked: keeps the float64 temporaries small
        d = (img[y:y + 1024].astype(np.float64) - out[y:y + 1024].astype(np.float64)) / 255.0
        sq += float((d * d).sum())
    psnr = float("inf") if sq == 0 else 10.0 * float(np.log10(img.size / sq))
    return {"psnr_db": round(psnr, 4), "error_blocks": int(errors), "decoder": "oracle/astc_decode.c (plain-C restatement)"}


def device_quality(lib, ctx, cfg, d_img, d_blocks, dev):
    """Quality figures without leaving the GPU: the product's decode kernel writes the decoded image into HBM and
    its comparison kernel reduces the error sums there (include/astcenc_amd.h).  Outside the timed region of the
    bench; the two calls are timed on their own (row f.1 of the scope table: the next-row kernels)."""
    size = cfg["size"]
    d_dec = torch.empty_like(d_img)
    swz = A.Swizzle(*A.SWZ_RGBA)
    stream = torch.cuda.current_stream(dev).cuda_stream
    ttype = A.TYPE_F16 if cfg["hdr"] else A.TYPE_U8
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    def timed(call, repeats=3):
        """Best wall time in ms of a synchronous device-API call (launch, kernel(s), the host's wait; the comparison
        also copies 18 doubles back), and the best time between two events on the call's stream around it (the kernel
        plus the launch gap, without the host's wait)."""
        best = best_ev = 1e9
        for _ in range(repeats):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            ev0.record()
            assert call() == 0
            ev1.record()
            torch.cuda.synchronize(dev)
            best = min(best, (time.perf_counter() - t0) * 1e3)
            best_ev = min(best_ev, ev0.elapsed_time(ev1))
        return best, best_ev

    decode = lambda: lib.lib.astcenc_amd_decompress_image_device(ctx, d_blocks.data_ptr(), d_blocks.numel(), d_dec.data_ptr(), size, size, 1,
                                                                ttype, ctypes.byref(swz), stream)
    decode_ms, decode_ev_ms = timed(decode, repeats=5)
    texel_bytes = d_img.element_size() * 4
    decode_bytes = d_blocks.numel() + size * size * texel_bytes
    timing = {"decode_ms": round(decode_ms, 3), "decode_gbps": round(decode_bytes / decode_ms / 1e6, 1),
              "decode_hbm_frac": round(decode_bytes / decode_ms / 1e6 / 8000.0, 4),
              "decode_events_ms": round(decode_ev_ms, 3), "decode_events_hbm_frac": round(decode_bytes / decode_ev_ms / 1e6 / 8000.0, 4)}
    sums = A.ErrorSums()
    if not cfg["hdr"]:
        compare = lambda: lib.lib.astcenc_amd_compare_images_device(ctx, d_img.data_ptr(), ttype, d_dec.data_ptr(), ttype, size, size, 1, stream,
                                                                   ctypes.byref(sums))
        compare_ms, _ = timed(compare)
        timing.update({"compare_ms": round(compare_ms, 3), "compare_gbps": round(2 * size * size * texel_bytes / compare_ms / 1e6, 1),
                       "what": "decode_ms / compare_ms: wall time of the synchronous astcenc_amd_decompress_image_device / _compare_images_device calls (best of 5 / 3); decode_events_ms: between two events on the stream around the decode call"})
        return {"psnr_db_on_device": round(sums.psnr(), 4), "decode_compare_timing": timing}
    hdr = A.HdrErrorSums()
    e = lib.lib.astcenc_amd_compare_images_hdr_device(ctx, d_img.data_ptr(), ttype, d_dec.data_ptr(), ttype, size, size, 1, -10, 10, stream,
                                                      ctypes.byref(sums), ctypes.byref(hdr))
    assert e == 0, e
    return {"decode_compare_timing": timing, "mpsnr_db_on_device": round(hdr.mpsnr(sums.texels), 4), "log_rmse_on_device": round(hdr.log_rmse(sums.texels), 4),
            "psnr_rgb_db_on_device": round(sums.psnr(3), 4), "rgb_peak": sums.rgb_peak, "fstops": [-10, 10],
            "definition": "astcenccli_error_metrics.cpp:60-107, :262-268, :389-403"}


def measured_counters(name="c2"):
    """HBM bytes per launch and the VALU figures of config `name` from the latest committed PMC summary
    (profiles/*/traffic.json, written by tools/gpu_evidence.sh); {} when there is none."""
    import glob
    # newest = highest round tag (profiles/r01a < r01b < ... < r02a); mtimes mean nothing in a fresh checkout
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "traffic.json")), reverse=True):
        data = json.load(open(path))
        per_config = data.get("configs", {"c2": data} if "hbm_bytes_per_launch" in data else {})
        if name in per_config:
            return per_config[name], os.path.relpath(path, ROOT)
    return {}, None


# ---------------------------------------------------------------------------------------------------------
# the timed path
# ---------------------------------------------------------------------------------------------------------

def time_device_resident(lib, ctx, cfg, d_img, d_out, dev, steps, warmup, barrier):
    """W warm-up steps, then K timed steps between barriers.  Returns (elapsed s, [kernel ms per step])."""
    size = cfg["size"]
    swz = A.Swizzle(*A.SWZ_RGBA)
    kernel_ms = ctypes.c_float(0.0)
    stream = torch.cuda.current_stream(dev)
    ttype = A.TYPE_F16 if cfg["hdr"] else A.TYPE_U8

    def step():
        e = lib.lib.astcenc_amd_compress_image_device(ctx, d_img.data_ptr(), size, size, ttype, ctypes.byref(swz),
                                                      d_out.data_ptr(), d_out.numel(), stream.cuda_stream, ctypes.byref(kernel_ms))
        assert e == 0, e
        return kernel_ms.value

    for _ in range(warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    kms = [step() for _ in range(steps)]
    barrier()
    return time.perf_counter() - t0, kms


def kernel_name_of(lib, ctx):
    """The build of the compression kernel the context launches (include/astcenc_amd.h: a fixed-context build for the BASELINE
    contexts, a generic one otherwise)."""
    return "astcd::" + lib.lib.astcenc_amd_context_kernel_name(ctx).decode()


def roofline_of(cfg, kernel_s, kernel_name, name):
    algo = algorithmic_bytes(cfg)
    achieved = algo / kernel_s / 1e9
    roof = {"bound": "hbm", "achieved": round(achieved, 4), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 7), "traffic": None,
            "kernel": kernel_name, "kernel_ms": round(kernel_s * 1e3, 3),
            "algorithmic_bytes_per_launch": algo}
    counters, src = measured_counters(name)
    if counters:
        roof["traffic"] = counters.get("hbm_bytes_per_launch")
        roof["traffic_unit"] = "bytes per launch"
        roof["traffic_source"] = "%s (committed rocprofv3 --pmc summary of this workload, not measured in this run)" % src
        if "valu_insts_per_block" in counters:
            nbx, nby = block_grid(cfg)
            # HBM is four orders of magnitude away; what binds the kernel is VALU issue.  Instructions per block are a
            # property of (library, workload) and come from the committed counters; the rate is this run's.
            roof["valu"] = {"insts_per_block": counters["valu_insts_per_block"], "active_lanes": counters.get("active_lanes_avg"),
                            "issue_frac_spec": round(counters["valu_insts_per_block"] * nbx * nby / kernel_s / VALU_PEAK_WAVE_INSTS_S, 4),
                            "peak_wave_insts_per_s": VALU_PEAK_WAVE_INSTS_S,
                            "what": "wave-level VALU instructions per block x blocks / kernel time of this run, against 1024 SIMDs x 2.4 GHz / 2 "
                                    "clocks; active_lanes = SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU (of 64)"}
            if "valu_issue_cycles_per_block" in counters:
                # Sum over the opcode classes of (count x measured issue cost) / (kernel time x SIMDs x clock): how busy the
                # vector pipes are by the costs of tools/valu_microbench3.hip instead of the spec's 2 clocks; a range, since
                # the hardware's class counters leave two fifths of the instructions unclassified (tools/summarize_evidence.py)
                simd_cycles = kernel_s * 1024 * 2.4e9
                cyc = counters["valu_issue_cycles_per_block"]
                roof["valu"]["issue_frac_measured"] = {"low": round(cyc["low"] * nbx * nby / simd_cycles, 4), "high": round(cyc["high"] * nbx * nby / simd_cycles, 4),
                                                       "class_insts_per_block": counters.get("valu_class_insts_per_block"), "costs": cyc.get("costs")}
        for key in ("salu_insts_per_block", "lds_insts_per_block", "wait_any_frac_of_wave_cycles", "lds_bank_conflict_frac"):
            if key in counters:
                roof[key] = counters[key]
    return roof


def run_extra_config(lib, name, dev, steps, warmup, shared_img, budget_s):
    cfg = CONFIGS[name]
    img = shared_img if shared_img is not None else make_image(cfg, 0x9E3779B1)
    ctx = make_context(lib, cfg)
    d_img = to_device(img, dev)
    nbx, nby = block_grid(cfg)
    d_out = torch.zeros(nbx * nby * 16, dtype=torch.uint8, device=dev)
    elapsed, kms = time_device_resident(lib, ctx, cfg, d_img, d_out, dev, steps, warmup, lambda: torch.cuda.synchronize(dev))
    kernel_s = sum(kms) / len(kms) / 1e3
    texels = cfg["size"] * cfg["size"]
    gpu_blocks = d_out.cpu().numpy()
    res = {"config": name, "baseline_config_index": cfg["index"], "workload": cfg["label"],
           "value": round(texels * steps / elapsed / 1e6, 3), "unit": "Mtexels/s", "steps": steps, "warmup": warmup,
           "ms_per_step": round(elapsed / steps * 1e3, 3), "blocks_per_image": nbx * nby,
           "roofline": roofline_of(cfg, kernel_s, kernel_name_of(lib, ctx), name),
           "parity_vs_reference": parity_whole_or_crops(cfg, img, gpu_blocks, budget_s),
           "quality": device_quality(lib, ctx, cfg, d_img, d_out, dev)}
    lib.context_free(ctx)
    del d_img, d_out
    return res


def run_jit_config(lib, dev, img_host):
    """A context none of the library's fixed-context builds serves (6x6 -thorough): the generic build against the build the
    library compiles for the context at run time (csrc/kernel_jit.cpp), cold compile time included, on a 4096^2 crop of the
    bench image; bytes compared."""
    import tempfile
    cfg = dict(CONFIGS["c2"], size=4096, quality=A.PRE_THOROUGH, label="4096x4096 RGBA8 LDR, 6x6 block, -thorough (crop of the bench image)")
    img = np.ascontiguousarray(img_host[:4096, :4096])
    d_img = to_device(img, dev)
    nbx, nby = block_grid(cfg)
    res = {"config": "run_time_build", "workload": cfg["label"]}
    saved = {k: os.environ.get(k) for k in ("ASTCENC_AMD_JIT", "ASTCENC_AMD_CACHE_DIR")}
    blocks = {}
    try:
        with tempfile.TemporaryDirectory() as cache:
            os.environ["ASTCENC_AMD_CACHE_DIR"] = cache
            for mode in ("off", "sync", "sync"):
                os.environ["ASTCENC_AMD_JIT"] = mode
                t0 = time.perf_counter()
                ctx = make_context(lib, cfg)
                alloc_ms = (time.perf_counter() - t0) * 1e3
                d_out = torch.zeros(nbx * nby * 16, dtype=torch.uint8, device=dev)
                elapsed, kms = time_device_resident(lib, ctx, cfg, d_img, d_out, dev, 2, 1, lambda: torch.cuda.synchronize(dev))
                key = "generic" if mode == "off" else ("run_time_cold" if "run_time_cold" not in res else "run_time_cached")
                res[key] = {"kernel": kernel_name_of(lib, ctx), "context_alloc_ms": round(alloc_ms, 1),
                            "value": round(cfg["size"] ** 2 * 2 / elapsed / 1e6, 3), "unit": "Mtexels/s", "kernel_ms": round(sum(kms) / len(kms), 3)}
                blocks[key] = d_out.cpu().numpy()
                lib.context_free(ctx)
                del d_out
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    res["identical_bytes"] = bool(np.array_equal(blocks["generic"], blocks["run_time_cold"]) and np.array_equal(blocks["generic"], blocks["run_time_cached"]))
    res["speedup_run_time_vs_generic"] = round(res["run_time_cached"]["value"] / res["generic"]["value"], 3)
    return res


def time_host_api(lib, cfg, img, devices, repeats=2):
    """Mtexels/s through astcenc_compress_image: pageable host memory in and out, PCIe both ways inside the timing."""
    old = os.environ.get("ASTCENC_AMD_DEVICES")
    os.environ["ASTCENC_AMD_DEVICES"] = devices
    try:
        ctx = make_context(lib, cfg)
    finally:
        if old is None:
            os.environ.pop("ASTCENC_AMD_DEVICES", None)
        else:
            os.environ["ASTCENC_AMD_DEVICES"] = old
    ndev = lib.lib.astcenc_amd_context_device_count(ctx)
    h, w = img.shape[:2]
    bx, by = cfg["block"]
    out = np.zeros(((w + bx - 1) // bx) * ((h + by - 1) // by) * 16, dtype=np.uint8)
    assert lib.compress_raw(ctx, img, out) == 0          # warm-up: staging buffers get allocated here
    best = 1e30
    for _ in range(repeats):
        t0 = time.perf_counter()
        assert lib.compress_raw(ctx, img, out) == 0
        best = min(best, time.perf_counter() - t0)
    lib.context_free(ctx)
    return h * w / best / 1e6, ndev, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2", help="BASELINE config timed as the headline line (default c2 = configs[1])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-quality", action="store_true", help="skip decoding the output for PSNR")
    ap.add_argument("--no-parity-full", action="store_true", help="skip the whole-image byte comparison with the reference (config 2, ~10 s of host time)")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra_configs leg (BASELINE configs[2] and [3])")
    ap.add_argument("--no-host-api", action="store_true", help="skip the host-pointer (PCIe-inclusive) leg")
    ap.add_argument("--single-process", action="store_true",
                    help="N GPUs from ONE process through the C ABI's own device split (host pointers, PCIe included)")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node == --gpus"
    # (debug aid for boxes with fewer GPUs than ranks: ASTC_BENCH_SHARE_GPU=1 puts every rank on GPU 0 and
    #  uses gloo for the barrier / max-time reduction, since RCCL refuses two ranks on one device)
    share_gpu = os.environ.get("ASTC_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # (ASTC_BENCH_FORCE_DIST=1: take the process-group path with ONE rank as well -- tests/test_bench_multirank.py runs the RCCL
    #  calls of the N > 1 launch on the one GPU of the test box, where RCCL refuses two ranks)
    use_dist = world > 1 or (os.environ.get("ASTC_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ)
    if use_dist:
        import torch.distributed as dist
        if share_gpu:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)

    lib = A.Library(A.LIB_PRODUCT)
    assert lib.backend_name() == "hip:gfx950"

    if args.single_process:
        return single_process(lib, cfg, args)

    # one process per GPU: this rank's context lives on this rank's device only
    os.environ["ASTCENC_AMD_DEVICES"] = str(local_rank)
    t_ctx = time.perf_counter()
    ctx = make_context(lib, cfg)
    context_alloc_ms = (time.perf_counter() - t_ctx) * 1e3

    # synthetic input of BASELINE's shape, one image per rank (different seeds), resident in HBM
    img_host = make_image(cfg, 0x9E3779B1 + rank)
    d_img = to_device(img_host, dev)
    nbx, nby = block_grid(cfg)
    nblocks = nbx * nby
    d_out = torch.zeros(nblocks * 16, dtype=torch.uint8, device=dev)

    def barrier():
        if use_dist:
            torch.distributed.barrier()
        torch.cuda.synchronize(dev)

    elapsed, kms = time_device_resident(lib, ctx, cfg, d_img, d_out, dev, args.steps, args.warmup, barrier)

    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if share_gpu else dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
        # (the group is done with: what rank 0 adds to its line below -- quality figures, the CPU baseline -- happens after
        #  the timed region and after the other ranks have left, so it neither perturbs their timing nor keeps a collective open)
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()

    if rank == 0:
        texels = cfg["size"] * cfg["size"]
        value = world * texels * args.steps / elapsed / 1e6
        kernel_s = sum(kms) / len(kms) / 1e3
        gpu_blocks = d_out.cpu().numpy()
        roof = roofline_of(cfg, kernel_s, kernel_name_of(lib, ctx), args.config)
        out = {
            "metric": "Mtexels/s + PSNR-dB, 8192x8192 RGBA8 LDR 6x6 -medium" if args.config == "c2" else "Mtexels/s, " + cfg["label"],
            "value": round(value, 3), "unit": "Mtexels/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s, one image per GPU (BASELINE configs[%d])" % (cfg["label"], cfg["index"]),
                       "blocks_per_image": nblocks, "block": "%dx%d" % cfg["block"], "sharding": "one image per rank, no collectives",
                       "inputs": "resident in HBM when the timed region starts (device API, the timing rule of this benchmark); "
                                 "value_host_api is the rate of the reference's own entry point astcenc_compress_image, "
                                 "host pointers in and out, PCIe both ways inside the timing (SURVEY.md 8d)"},
            "roofline": roof,
            "context": {"kernel": kernel_name_of(lib, ctx), "alloc_ms": round(context_alloc_ms, 1),
                        "what": "astcenc_context_alloc: table blob built on the host and uploaded, streams and events; the BASELINE contexts "
                                "launch a fixed-context build of the library, every other context gets one compiled at run time "
                                "(extra_configs: run_time_build)"},
        }
        if world == 1 and not args.no_host_api:
            rate, ndev, host_blocks = time_host_api(lib, cfg, img_host, str(local_rank))
            out["value_host_api"] = {"value": round(rate, 3), "unit": "Mtexels/s", "devices": ndev,
                                     "what": "astcenc_compress_image, pageable host buffers, H2D + kernels + D2H, best of 2",
                                     "identical_to_device_path": bool(np.array_equal(host_blocks, gpu_blocks))}
        if not args.no_quality:
            q = decoded_psnr(cfg, img_host, gpu_blocks) or {}
            q.update(device_quality(lib, ctx, cfg, d_img, d_out, dev))
            out["quality"] = q
        if not args.no_cpu_baseline:
            # (N > 1: timed on rank 0 once the other ranks are done, i.e. on the same idle host as at N = 1; the ratio below
            #  stays "one GPU's share against the host": the whole job's value / N)
            base = cpu_reference_baseline(cfg, img_host, gpu_blocks)
            if base:
                out["cpu_baseline"] = base
                out["speedup_vs_cpu_baseline"] = round(value / base["value"], 2)
                if world > 1:
                    out["speedup_vs_cpu_baseline_per_gpu"] = round(value / world / base["value"], 2)
                if world == 1 and args.config == "c2" and not args.no_parity_full:
                    par = parity_full(cfg, img_host, gpu_blocks, base["threads_at_best"], base["x86_gathers_used"])
                    out["parity_full"] = par
                    if par:
                        # the baseline figure is the WHOLE image's (the parity leg pays for that run anyway);
                        # the crop sweep above chose the thread count and the build, and stays in the line
                        base["value_crop_best_of_3"] = base["value"]
                        base["value"] = par["reference_mtexels_s_whole_image"]
                        base["sample"] = ("astcenc-avx2 (oracle/_ref) on the WHOLE %dx%d bench image, one run, %d threads (%s); thread count and build "
                                          "chosen on a %s" % (cfg["size"], cfg["size"], par["threads"], par["reference"], base["sample"]))
                        base["mtexels_per_core_second"] = round(base["value"] / max(min(base["cpus_allowed"], base["threads_at_best"]), 1e-9), 4)
                        out["speedup_vs_cpu_baseline"] = round(value / base["value"], 2)
        if world == 1 and not args.no_extra and args.config == "c2":
            del d_img, d_out
            extra = []
            for name in ("c3", "c4"):
                shared = img_host if name == "c3" else None           # c3 is the same RGBA8 image, other footprint / preset
                extra.append(run_extra_config(lib, name, dev, 2, 1, shared, 60.0))
            extra.append(run_photo_config(lib, dev, 3, 1))
            extra.append(run_jit_config(lib, dev, img_host))
            out["extra_configs"] = extra
        print(json.dumps(out), flush=True)

    lib.context_free(ctx)


def single_process(lib, cfg, args):
    """--single-process: what a plain C caller gets.  One context spanning N devices; one host image N times as
    tall as BASELINE's (so every device compresses one BASELINE image's worth: weak scaling, like the
    process-per-GPU mode) goes through astcenc_compress_image, which splits it by block rows and joins."""
    n = args.gpus
    devices = ",".join(str(i) for i in range(n))
    if os.environ.get("ASTC_BENCH_SHARE_GPU") == "1":
        devices = ",".join("0" for _ in range(n))
    base = make_image(cfg, 0x9E3779B1)
    img = np.ascontiguousarray(np.concatenate([base] * n, axis=0)) if n > 1 else base
    os.environ["ASTCENC_AMD_DEVICES"] = devices
    ctx = make_context(lib, cfg)
    ndev = lib.lib.astcenc_amd_context_device_count(ctx)
    h, w = img.shape[:2]
    bx, by = cfg["block"]
    out = np.zeros(((w + bx - 1) // bx) * ((h + by - 1) // by) * 16, dtype=np.uint8)
    for _ in range(max(args.warmup, 1)):
        assert lib.compress_raw(ctx, img, out) == 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        assert lib.compress_raw(ctx, img, out) == 0
    elapsed = time.perf_counter() - t0
    lib.context_free(ctx)
    print(json.dumps({
        "metric": "Mtexels/s, %s, single process / C-ABI device split, PCIe included" % cfg["label"],
        "value": round(h * w * args.steps / elapsed / 1e6, 3), "unit": "Mtexels/s", "n_gpus": ndev, "steps": args.steps,
        "warmup": max(args.warmup, 1), "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s x %d stacked vertically, one host image through astcenc_compress_image" % (cfg["label"], n),
                   "devices": devices, "sharding": "block rows dealt to devices inside libastcenc_amd.so, no collectives"}}), flush=True)


if __name__ == "__main__":
    main()
