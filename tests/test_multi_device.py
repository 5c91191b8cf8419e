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


class _Devices:
    """ASTCENC_AMD_DEVICES for the contexts created inside the with-block."""
    def __init__(self, devices):
        self.devices = devices

    def __enter__(self):
        self.old = os.environ.get("ASTCENC_AMD_DEVICES")
        os.environ["ASTCENC_AMD_DEVICES"] = self.devices

    def __exit__(self, *exc):
        if self.old is None:
            del os.environ["ASTCENC_AMD_DEVICES"]
        else:
            os.environ["ASTCENC_AMD_DEVICES"] = self.old


def test_default_is_the_callers_device_only(product, A):
    """Without ASTCENC_AMD_DEVICES a context touches one GPU: a rank-per-GPU process must not initialise its neighbours'."""
    old = os.environ.pop("ASTCENC_AMD_DEVICES", None)
    try:
        err, cfg = product.config_init(A.PRF_LDR, 6, 6, 1, A.PRE_FAST, 0)
        err, ctx = product.context_alloc(cfg, 1)
        assert err == A.SUCCESS
        assert product.lib.astcenc_amd_context_device_count(ctx) == 1
        product.context_free(ctx)
    finally:
        if old is not None:
            os.environ["ASTCENC_AMD_DEVICES"] = old


def test_eight_slots_on_the_baseline_image(product, ref, A):
    """BASELINE's 8192^2 image dealt to eight slots (the 8-GPU node's code path on one GPU): the stream equals the
    one-slot stream, the rows either side of all seven seams equal the reference, the progress callback is monotonic
    although eight host threads feed it, and a cancel stops every shard within a few chunks."""
    size, block = 8192, (6, 6)
    img = A.synthetic_image(size, size)
    seen = []
    cb = A.PROGRESS_CB(lambda p: seen.append(p))
    got, n = _compress(A, product, img, block, A.PRE_MEDIUM, "0,0,0,0,0,0,0,0", progress=cb)
    assert n == 8
    assert seen and seen == sorted(seen) and abs(seen[-1] - 100.0) < 1e-3
    one, _ = _compress(A, product, img, block, A.PRE_MEDIUM, "0")
    assert np.array_equal(got, one)
    nb = (size + 5) // 6
    per = (nb + 7) // 8
    got = got.reshape(nb, nb, 16)
    for g in range(1, 8):
        r0 = g * per - 1                              # last row of shard g-1 and first row of shard g
        y0, y1 = r0 * 6, (r0 + 2) * 6
        want = ref.compress(np.ascontiguousarray(img[y0:y1, :1536]), block, A.PRE_MEDIUM).reshape(2, -1, 16)
        assert np.array_equal(got[r0:r0 + 2, :256], want), g

    # cancel from the progress callback: every shard stops at its next chunk boundary, the call still returns SUCCESS
    # (ref: astcenc_compress_cancel, astcenc.h:820) and nothing past the cancelled chunks is written
    with _Devices("0,0,0,0,0,0,0,0"):
        err, cfg = product.config_init(A.PRF_LDR, 6, 6, 1, A.PRE_MEDIUM, 0)
        holder = {}
        calls = []

        def on_progress(p):
            calls.append(p)
            if len(calls) == 2:
                product.lib.astcenc_compress_cancel(holder["ctx"])
        cb2 = A.PROGRESS_CB(on_progress)
        cfg.progress_callback = cb2
        err, ctx = product.context_alloc(cfg, 1)
        assert err == A.SUCCESS
    holder["ctx"] = ctx
    try:
        out = np.zeros(nb * nb * 16, dtype=np.uint8)
        assert product.compress_raw(ctx, img, out) == A.SUCCESS
        done = int(out.reshape(-1, 16).any(axis=1).sum())
        # each of the eight shards runs four bands and stays at most two bands ahead of what has been reported
        assert 0 < done < nb * nb * 7 // 8, done
        assert calls == sorted(calls) and calls[-1] < 100.0
    finally:
        product.context_free(ctx)


def test_decompress_is_dealt_to_the_slots_too(product, A):
    w, h, block = 2000, 1500, (5, 4)                   # 150 000 blocks: up to eight 16 384-block shards
    img = A.synthetic_image(w, h, 4)
    blocks = product.compress(img, block, A.PRE_FASTEST)
    one = product.decompress(blocks, w, h, block)
    for devices in ("0,0", "0,0,0,0,0,0,0,0"):
        with _Devices(devices):
            assert np.array_equal(product.decompress(blocks, w, h, block), one), devices
    for t in (np.float16, np.float32):
        with _Devices("0,0,0"):
            assert np.array_equal(product.decompress(blocks, w, h, block, out_type=t), product.decompress(blocks, w, h, block, out_type=t))


def test_volumes_and_slice_stacks_are_dealt_by_layers(product, ref, A):
    rng = np.random.default_rng(5)
    # a stack of 2D slices with a 2D footprint: F16 input (every slice from its own data) and RGBA8 input (the reference's
    # fast loader reads slice 0 for every slice -- a shard further up the stack must still see slice 0)
    stack8 = np.stack([A.synthetic_image(384, 256, 10 + z) for z in range(6)])            # 6 x 96 x 64 = 36 864 blocks
    stackf = (stack8.astype(np.float32) / 255.0).astype(np.float16)
    vol = rng.integers(0, 256, size=(40, 120, 120, 4), dtype=np.uint8)                    # 4x4x4: 10 x 30 x 30 = 9 000 ... 3x3x3 below
    for pixels, block in ((stackf, (4, 4)), (stack8, (4, 4)), (vol, (3, 3, 3))):
        with _Devices("0"):
            one = product.compress(pixels, block, A.PRE_FASTEST)
        with _Devices("0,0,0"):
            got = product.compress(pixels, block, A.PRE_FASTEST)
        assert np.array_equal(one, got), block
        d = pixels.shape[0]
        with _Devices("0"):
            back1 = product.decompress(got, pixels.shape[2], pixels.shape[1], block, depth=d)
        with _Devices("0,0,0"):
            back3 = product.decompress(got, pixels.shape[2], pixels.shape[1], block, depth=d)
        assert np.array_equal(back1, back3), block
    want = ref.compress(stack8, (4, 4), A.PRE_FASTEST)
    with _Devices("0,0,0"):
        assert np.array_equal(product.compress(stack8, (4, 4), A.PRE_FASTEST), want)


@pytest.mark.parametrize("radius,slots", [(1, "0,0"), (3, "0,0,0"), (9, "0,0,0,0,0"), (40, "0,0,0")])
def test_alpha_scale_prepass_is_sharded_with_a_halo(product, ref, A, radius, slots):
    """a_scale_radius != 0 (the CLI's -a): every shard runs the alpha-average pre-pass over its own rows plus the rows its
    averages reach into (ref: the reference deals the same pre-pass to its worker threads, astcenc_entry.cpp:1190-1211,
    astcenc_compute_variance.cpp:507).  The tiles of the pre-pass keep their places, so the averages -- and the blocks the
    test on them skips -- are the same floats as on one device, and the stream equals the reference's."""
    w, h, block = 1500, 2300, (6, 6)                     # 96 000 blocks: enough for five shards
    img = images.noisy(w, h, 21)
    # transparent bands and islands whose edges fall near the shard seams and on / off tile boundaries
    img[:, :, 3] = 255
    for y0, y1 in ((250, 520), (690, 705), (930, 1190), (1500, 1560), (1830, 1930)):
        img[y0:y1, :, 3] = 0
    img[300:420, 100:400, 3] = 7
    img[1000:1003, 500:503, 3] = 1
    img[1535:1538, 40:1400, 3] = 2

    def run(devices):
        old = os.environ.get("ASTCENC_AMD_DEVICES")
        os.environ["ASTCENC_AMD_DEVICES"] = devices
        try:
            return product.compress(img, block, A.PRE_FAST, flags=A.FLG_USE_ALPHA_WEIGHT, tweak=lambda c: setattr(c, "a_scale_radius", radius))
        finally:
            if old is None:
                del os.environ["ASTCENC_AMD_DEVICES"]
            else:
                os.environ["ASTCENC_AMD_DEVICES"] = old

    one = run("0")
    many = run(slots)
    assert np.array_equal(one, many), "the %s split differs from one device at radius %d" % (slots, radius)
    if radius == 3:
        want = ref.compress(img, block, A.PRE_FAST, flags=A.FLG_USE_ALPHA_WEIGHT, tweak=lambda c: setattr(c, "a_scale_radius", radius))
        assert np.array_equal(want, many)
    plain = product.compress(img, block, A.PRE_FAST, flags=A.FLG_USE_ALPHA_WEIGHT)
    assert (plain != many).any(), "the test image must contain blocks that the alpha test skips"


@pytest.mark.gpu
@pytest.mark.parametrize("radius", [0, 3, 40])
def test_portions_are_dealt_from_one_counter(product, ref, A, monkeypatch, radius):
    """Round 6: the devices TAKE portions of block rows from one counter (ref: the reference's workers take blocks from a
    ticket counter, astcenc_internal_entry.h:225-236) instead of owning one range each.  With the minimum portion size
    lowered (a test-only switch) a 96 000-block image on three slots is cut into twelve portions -- whichever slot takes
    which, the stream is the one-device stream, with the alpha-scale pre-pass too (every portion carries its own halo),
    and for a stack of slices (portions of layers)."""
    monkeypatch.setenv("ASTCENC_AMD_DEAL_MIN_BLOCKS", "4096")
    w, h, block = 1500, 2300, (6, 6)
    img = images.noisy(w, h, 23)
    img[:, :, 3] = 255
    for y0, y1 in ((180, 200), (760, 1010), (1530, 1550), (2100, 2290)):
        img[y0:y1, :, 3] = 0
    flags = A.FLG_USE_ALPHA_WEIGHT if radius else 0
    tweak = (lambda c: setattr(c, "a_scale_radius", radius)) if radius else None
    with _Devices("0"):
        one = product.compress(img, block, A.PRE_FAST, flags=flags, tweak=tweak)
    for deal in ("dynamic", "static"):
        monkeypatch.setenv("ASTCENC_AMD_DEAL", deal)
        with _Devices("0,0,0"):
            many = product.compress(img, block, A.PRE_FAST, flags=flags, tweak=tweak)
        assert np.array_equal(one, many), (deal, radius)
    if radius == 3:
        assert np.array_equal(ref.compress(img, block, A.PRE_FAST, flags=flags, tweak=tweak), one)
    if radius == 0:
        monkeypatch.setenv("ASTCENC_AMD_DEAL", "dynamic")
        stack = np.stack([A.synthetic_image(384, 256, 30 + z) for z in range(12)])            # 12 x 96 x 64 blocks at 4x4
        with _Devices("0"):
            one = product.compress(stack, (4, 4), A.PRE_FASTEST)
        with _Devices("0,0,0"):
            many = product.compress(stack, (4, 4), A.PRE_FASTEST)
        assert np.array_equal(one, many)


@pytest.mark.gpu
def test_two_multi_device_contexts_at_once(product, A):
    """Two host threads, each with its own three-slot context on the same devices, compress two images at the same time:
    the contexts share nothing but the GPUs (own tables, streams, staging, host threads), portions are dealt inside each --
    both streams equal the one-device streams, three times in a row."""
    import threading
    img_a, img_b = A.synthetic_image(1500, 1400, 41), A.synthetic_image(1400, 1500, 42)
    with _Devices("0"):
        want_a, want_b = product.compress(img_a, (6, 6), A.PRE_FAST), product.compress(img_b, (6, 6), A.PRE_FAST)
    with _Devices("0,0,0"):
        err, cfg = product.config_init(A.PRF_LDR, 6, 6, 1, A.PRE_FAST, 0)
        assert err == 0
        ctxs = []
        for _ in range(2):
            err, ctx = product.context_alloc(cfg, 1)
            assert err == 0
            ctxs.append(ctx)
        outs = [np.zeros(want_a.size, dtype=np.uint8), np.zeros(want_b.size, dtype=np.uint8)]
        rcs = [None, None]

        def run(i, img):
            rcs[i] = product.compress_raw(ctxs[i], np.ascontiguousarray(img), outs[i])
        for _ in range(3):
            ts = [threading.Thread(target=run, args=(0, img_a)), threading.Thread(target=run, args=(1, img_b))]
            [t.start() for t in ts]
            [t.join() for t in ts]
            assert rcs == [0, 0]
            assert np.array_equal(outs[0], want_a) and np.array_equal(outs[1], want_b)
            for ctx in ctxs:
                assert product.lib.astcenc_compress_reset(ctx) == 0
        for ctx in ctxs:
            product.context_free(ctx)
