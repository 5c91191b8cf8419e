#!/usr/bin/env python3
# SPDX-License-Identifier: Apache-2.0
"""Headline benchmark: Mtexels/s of ASTC LDR 6x6 -medium compression of an 8192x8192 RGBA8 image
(BASELINE.json configs[1]) on MI355X, inputs resident in HBM.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one pass of the block compressor over one 8192x8192 image per rank.  The path shards by
independent images/block rows with no data-path collective, so N ranks compress N
images: weak scaling, value = N * texels * K / max-over-ranks time.  torch is used for device
memory, the stream, and the rank barrier only; the work is libastcenc_amd.so called through its
C ABI (astcenc_amd_compress_image_device).

Rank 0 prints ONE JSON line with the driver's fields plus:
  roofline     : algorithmic HBM bytes per launch / mean kernel time (HIP events on the launch stream,
                 taken inside the library) vs the 8 TB/s HBM peak.  The kernel is compute/latency bound
                 by four orders of magnitude, which this fraction shows honestly.
  cpu_baseline : the reference encoder's AVX2 build (oracle/_ref/libastcenc-avx2.so) on all host cores
                 of this box, timed on a bounded crop of the same image (N = 1 only), plus a byte
                 comparison of that crop against the GPU output.
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

WIDTH = HEIGHT = 8192
BLOCK = (6, 6)
QUALITY = A.PRE_MEDIUM
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec


def cpu_reference_baseline(img, gpu_blocks, blocks_x):
    """Time the reference AVX2 encoder on all host cores on a crop sized for roughly 10-20 s."""
    if not os.path.exists(O.LIB_REF_AVX2):
        return None
    ref = A.Library(O.LIB_REF_AVX2)
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)

    def run(crop):
        err, cfg = ref.config_init(A.PRF_LDR, BLOCK[0], BLOCK[1], 1, QUALITY, 0)
        assert err == 0
        err, ctx = ref.context_alloc(cfg, cores)
        assert err == 0
        h, w = crop.shape[:2]
        out = np.zeros(((w + 5) // 6) * ((h + 5) // 6) * 16, dtype=np.uint8)
        errs = [0] * cores
        t0 = time.perf_counter()
        threads = [threading.Thread(target=lambda i=i: errs.__setitem__(i, ref.compress_raw(ctx, crop, out, thread_index=i)))
                   for i in range(cores)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        dt = time.perf_counter() - t0
        ref.context_free(ctx)
        assert not any(errs), errs
        return dt, out

    # calibrate on 1536x1536 (enough work for every thread), then scale the crop to ~15 s of CPU time
    probe = np.ascontiguousarray(img[:1536, :1536])
    run(probe)
    dt, _ = run(probe)
    rate = probe.shape[0] * probe.shape[1] / dt
    side = int(min(8190, max(1536, (rate * 15.0) ** 0.5)))
    side = (side // 6) * 6
    crop = np.ascontiguousarray(img[:side, :side])
    dt, out = run(crop)

    # byte parity of the crop: its block grid coincides with the image's top-left blocks
    nb = side // 6
    g = gpu_blocks.reshape(-1, 16)
    rows = np.concatenate([g[r * blocks_x: r * blocks_x + nb] for r in range(nb)])
    mismatch = int((rows != out.reshape(-1, 16)).any(axis=1).sum())
    return {"value": round(side * side / dt / 1e6, 3), "unit": "Mtexels/s", "cores": cores, "kind": "reference",
            "sample": "%dx%d top-left crop of the bench image, astcenc-avx2 %d threads, %.1f s" % (side, side, cores, dt),
            "blocks_compared_with_gpu": nb * nb, "blocks_mismatching_gpu": mismatch}


def decoded_psnr(img, gpu_blocks):
    """PSNR (dB, RGBA, reference definition astcenccli_error_metrics.cpp:240-346) of the GPU's blocks
    decoded by the independent plain-C decoder in oracle/ (checker only, never timed)."""
    path = os.path.join(ROOT, "oracle", "_build", "libastc_decode.so")
    if not os.path.exists(path):
        return None
    dec = ctypes.CDLL(path)
    dec.astc_oracle_decode_image.restype = ctypes.c_int
    dec.astc_oracle_decode_image.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 5 + [ctypes.c_void_p]
    h, w = img.shape[:2]
    out = np.zeros((h, w, 4), dtype=np.uint8)
    errors = dec.astc_oracle_decode_image(gpu_blocks.ctypes.data, BLOCK[0], BLOCK[1], w, h, 0, out.ctypes.data)
    sq = 0.0
    for y in range(0, h, 1024):             # chunked: keeps the float64 temporaries small
        d = (img[y:y + 1024].astype(np.float64) - out[y:y + 1024].astype(np.float64)) / 255.0
        sq += float((d * d).sum())
    psnr = float("inf") if sq == 0 else 10.0 * float(np.log10(img.size / sq))
    return {"psnr_db": round(psnr, 4), "error_blocks": int(errors), "decoder": "oracle/astc_decode.c (plain-C restatement)"}


def device_psnr(lib, ctx, d_img, d_blocks, dev):
    """The same figure without leaving the GPU: the product's decode kernel writes the decoded image into
    HBM and its comparison kernel reduces the squared error there (include/astcenc_amd.h); not timed."""
    d_dec = torch.empty_like(d_img)
    swz = A.Swizzle(*A.SWZ_RGBA)
    stream = torch.cuda.current_stream(dev).cuda_stream
    e = lib.lib.astcenc_amd_decompress_image_device(ctx, d_blocks.data_ptr(), d_blocks.numel(), d_dec.data_ptr(), WIDTH, HEIGHT, 1,
                                                    A.TYPE_U8, ctypes.byref(swz), stream)
    assert e == 0, e
    sums = A.ErrorSums()
    e = lib.lib.astcenc_amd_compare_images_device(ctx, d_img.data_ptr(), A.TYPE_U8, d_dec.data_ptr(), A.TYPE_U8, WIDTH, HEIGHT, 1,
                                                  stream, ctypes.byref(sums))
    assert e == 0, e
    return round(sums.psnr(), 4)


def measured_traffic():
    """HBM bytes per launch from the latest committed PMC summary of this same workload
    (profiles/*/traffic.json, written by tools/gpu_profile.sh); None when there is none."""
    import glob
    # newest = highest round tag (profiles/r01a < r01b < ... < r02a); mtimes mean nothing in a fresh checkout
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "traffic.json")))
    if not files:
        return None, None
    t = json.load(open(files[-1]))
    return t["hbm_bytes_per_launch"], os.path.relpath(files[-1], ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-quality", action="store_true", help="skip decoding the output for PSNR")
    args = ap.parse_args()

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
    if world > 1:
        import torch.distributed as dist
        if share_gpu:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)

    lib = A.Library(A.LIB_PRODUCT)
    assert lib.backend_name() == "hip:gfx950"
    err, cfg = lib.config_init(A.PRF_LDR, BLOCK[0], BLOCK[1], 1, QUALITY, 0)
    assert err == 0, err
    err, ctx = lib.context_alloc(cfg, 1)
    assert err == 0, "context_alloc failed: %s" % lib.error_string(err)

    # synthetic input of BASELINE's shape, one image per rank (different seeds), resident in HBM
    img_host = A.synthetic_image(WIDTH, HEIGHT, 0x9E3779B1 + rank)
    d_img = torch.from_numpy(img_host).to(dev)
    blocks_x, blocks_y = (WIDTH + BLOCK[0] - 1) // BLOCK[0], (HEIGHT + BLOCK[1] - 1) // BLOCK[1]
    nblocks = blocks_x * blocks_y
    d_out = torch.zeros(nblocks * 16, dtype=torch.uint8, device=dev)
    swz = A.Swizzle(*A.SWZ_RGBA)
    kernel_ms = ctypes.c_float(0.0)
    stream = torch.cuda.current_stream(dev)

    def step():
        e = lib.lib.astcenc_amd_compress_image_device(ctx, d_img.data_ptr(), WIDTH, HEIGHT, A.TYPE_U8, ctypes.byref(swz),
                                                      d_out.data_ptr(), d_out.numel(), stream.cuda_stream, ctypes.byref(kernel_ms))
        assert e == 0, e
        return kernel_ms.value

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    kms = [step() for _ in range(args.steps)]
    barrier()
    elapsed = time.perf_counter() - t0

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if share_gpu else dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        texels = WIDTH * HEIGHT
        value = world * texels * args.steps / elapsed / 1e6
        algo_bytes = texels * 4 + nblocks * 16            # 4 B/texel in + 16 B/block out
        kernel_s = sum(kms) / len(kms) / 1e3
        achieved = algo_bytes / kernel_s / 1e9
        gpu_blocks = d_out.cpu().numpy()
        traffic, traffic_src = measured_traffic()
        out = {
            "metric": "Mtexels/s + PSNR-dB, 8192x8192 RGBA8 LDR 6x6 -medium",
            "value": round(value, 3), "unit": "Mtexels/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "8192x8192 RGBA8 LDR, 6x6 block, -medium, one image per GPU (BASELINE configs[1])",
                       "blocks_per_image": nblocks, "block": "6x6", "preset": "medium", "sharding": "one image per rank, no collectives"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 4), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 7), "traffic": traffic, "traffic_unit": "bytes per launch",
                         "traffic_source": traffic_src,
                         "kernel": "astcd::astc_compress_blocks_ldr", "kernel_ms": round(kernel_s * 1e3, 3),
                         "algorithmic_bytes_per_launch": algo_bytes},
        }
        if world == 1 and not args.no_quality:
            out["quality"] = decoded_psnr(img_host, gpu_blocks) or {}
            out["quality"]["psnr_db_on_device"] = device_psnr(lib, ctx, d_img, d_out, dev)
        if world == 1 and not args.no_cpu_baseline:
            base = cpu_reference_baseline(img_host, gpu_blocks, blocks_x)
            if base:
                out["cpu_baseline"] = base
        print(json.dumps(out), flush=True)

    lib.context_free(ctx)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
