# SPDX-License-Identifier: Apache-2.0
"""The N-GPU split behind the C ABI (ref: the block loop of compress_image,
Source/astcenc_entry.cpp:1009-1038, whose N worker threads become N devices here).

astcenc_context_alloc builds one device slot per GPU (ASTCENC_AMD_DEVICES lists the ordinals, an ordinal
may repeat), astcenc_compress_image deals contiguous block-row ranges of the host image to the slots and
joins them.  On the one-GPU test box "0,0" / "0,0,0" give two / three slots on the same GPU: the split, the
per-slot streams and staging buffers, the host threads and the gather are exactly the multi-GPU code path.
"""
import os

import numpy as np
import pytest

import images

pytestmark = pytest.mark.gpu


def _compress(A, lib, img, block, quality, devices, progress=None, threads=1):
    old = os.environ.get("ASTCENC_AMD_DEVICES")
    os.environ["ASTCENC_AMD_DEVICES"] = devices
    try:
        err, cfg = lib.config_init(A.PRF_LDR, block[0], block[1], 1, quality, 0)
        assert err == A.SUCCESS
        if progress is not None:
            cfg.progress_callback = progress
        err, ctx = lib.context_alloc(cfg, threads)
        assert err == A.SUCCESS
    finally:
        if old is None:
            del os.environ["ASTCENC_AMD_DEVICES"]
        else:
            os.environ["ASTCENC_AMD_DEVICES"] = old
    try:
        ndev = lib.lib.astcenc_amd_context_device_count(ctx)
        h, w = img.shape[:2]
        out = np.zeros(((w + block[0] - 1) // block[0]) * ((h + block[1] - 1) // block[1]) * 16, dtype=np.uint8)
        assert lib.compress_raw(ctx, img, out) == A.SUCCESS
        return out, ndev
    finally:
        lib.context_free(ctx)


@pytest.mark.parametrize("block,size", [((6, 6), (1030, 1500)), ((4, 4), (700, 516)), ((8, 8), (1600, 1416))])
def test_two_and_three_slots_reproduce_one_device(product, A, block, size):
    img = A.synthetic_image(size[0], size[1], 5)
    one, n1 = _compress(A, product, img, block, A.PRE_FAST, "0")
    assert n1 == 1
    for devices, want_n in (("0,0", 2), ("0,0,0", 3)):
        got, n = _compress(A, product, img, block, A.PRE_FAST, devices)
        assert n == want_n
        assert np.array_equal(one, got), devices


def test_shards_match_reference_on_the_seams(product, ref, A):
    """The block rows either side of every shard boundary, and the clamped last row, against the reference."""
    w, h, block = 1200, 1210, (6, 6)
    img = A.synthetic_image(w, h, 9)
    got, n = _compress(A, product, img, block, A.PRE_FAST, "0,0,0")
    assert n == 3
    bx, by = (w + 5) // 6, (h + 5) // 6
    per = (by + 2) // 3
    got = got.reshape(by, bx, 16)
    for r0 in (0, per - 1, 2 * per - 1, by - 3):
        rows = slice(r0, min(r0 + 3, by))
        y0, y1 = rows.start * 6, min(rows.stop * 6, h)
        # (a crop that ends on a block boundary equals the full image's blocks; the last rows include the clamped tail)
        want = ref.compress(np.ascontiguousarray(img[y0:y1]), block, A.PRE_FAST).reshape(-1, bx, 16)
        assert np.array_equal(got[rows], want), r0


def test_small_images_stay_on_one_device_and_progress_is_monotonic(product, A):
    seen = []
    cb = A.PROGRESS_CB(lambda p: seen.append(p))
    img = A.synthetic_image(1536, 1536, 3)          # 65 536 blocks: four 16 384-block shards at most
    got, n = _compress(A, product, img, (6, 6), A.PRE_FASTEST, "0,0", progress=cb)
    one, _ = _compress(A, product, img, (6, 6), A.PRE_FASTEST, "0")
    assert n == 2 and np.array_equal(got, one)
    assert seen and seen == sorted(seen) and abs(seen[-1] - 100.0) < 1e-3
    small = images.noisy(96, 96, 2)                 # 256 blocks: never split
    a, _ = _compress(A, product, small, (6, 6), A.PRE_MEDIUM, "0,0")
    b, _ = _compress(A, product, small, (6, 6), A.PRE_MEDIUM, "0")
    assert np.array_equal(a, b)


def test_device_buffers_follow_their_device_and_restore_the_callers(product, A):
    import ctypes
    import torch
    img = torch.from_numpy(A.synthetic_image(512, 512, 1)).cuda()
    before = torch.cuda.current_device()
    err, cfg = product.config_init(A.PRF_LDR, 6, 6, 1, A.PRE_FAST, 0)
    err, ctx = product.context_alloc(cfg, 1)
    assert err == A.SUCCESS
    try:
        nb = ((512 + 5) // 6) ** 2
        out = torch.zeros(nb * 16, dtype=torch.uint8, device="cuda")
        swz = A.Swizzle(*A.SWZ_RGBA)
        e = product.lib.astcenc_amd_compress_image_device(ctx, img.data_ptr(), 512, 512, A.TYPE_U8, ctypes.byref(swz), out.data_ptr(), out.numel(),
                                                          torch.cuda.current_stream().cuda_stream, None)
        assert e == A.SUCCESS
        assert torch.cuda.current_device() == before
        assert np.array_equal(out.cpu().numpy(), product.compress(img.cpu().numpy(), (6, 6), A.PRE_FAST))
    finally:
        product.context_free(ctx)
