# SPDX-License-Identifier: Apache-2.0
"""The C ABI seen from a caller: error codes, ownership and threading rules of the reference API,
plus the reference's own known-answer files for this path.

Mirrors Source/UnitTest/test_encode.cpp:32-299 (overflow / short-buffer errors, inf and NaN inputs
must encode without failing) and Test/astc_test_functional.py:457-503 (the three 1x1 .astc files in
Test/Data, quoted below as bytes).  Every case runs twice: on the scalar CPU build of the host layer +
kernel source (oracle/emu, here) and through libastcenc_amd.so on the GPU (-m gpu).
"""
import ctypes as C
import os
import threading
import oracle_libs as O  # (path set up by conftest.py)

import numpy as np
import pytest

import images

LIBS = [pytest.param("emu", id="emu"), pytest.param("product", id="hip", marks=pytest.mark.gpu)]


@pytest.fixture(params=LIBS)
def lib(request):
    return request.getfixturevalue(request.param)


def _ctx(lib, A, profile=None, block=(4, 4), quality=None, flags=0, threads=1):
    err, cfg = lib.config_init(A.PRF_LDR if profile is None else profile, block[0], block[1], 1,
                               A.PRE_MEDIUM if quality is None else quality, flags)
    assert err == A.SUCCESS
    err, ctx = lib.context_alloc(cfg, threads)
    assert err == A.SUCCESS, lib.error_string(err)
    return ctx


def _raw_compress(lib, A, ctx, dims, dtype, data_len, thread_index=0, swizzle=None):
    """astcenc_compress_image with arbitrary (possibly absurd) image dimensions and a 1-byte buffer."""
    inp = (C.c_uint8 * 64)()
    out = (C.c_uint8 * 64)()
    slices = (C.c_void_p * 1)(C.addressof(inp))
    img = A.Image(dims[0], dims[1], dims[2], dtype, slices)
    swz = A.Swizzle(*(swizzle or A.SWZ_RGBA))
    return lib.lib.astcenc_compress_image(ctx, C.byref(img), C.byref(swz), C.addressof(out), C.c_size_t(data_len), thread_index)


# ---- Source/UnitTest/test_encode.cpp:32-132 ----------------------------------------------------

def test_overflow_in_z(lib, A):
    ctx = _ctx(lib, A)
    assert _raw_compress(lib, A, ctx, (0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF), A.TYPE_U8, 2 ** 64 - 1) == A.ERR_BAD_PARAM
    lib.context_free(ctx)


def test_overflow_in_16(lib, A):
    ctx = _ctx(lib, A)
    assert _raw_compress(lib, A, ctx, (0x80000000, 0x80000000, 0x10), A.TYPE_U8, 2 ** 64 - 1) == A.ERR_BAD_PARAM
    lib.context_free(ctx)


def test_data_buffer_exceeded(lib, A):
    ctx = _ctx(lib, A)
    assert _raw_compress(lib, A, ctx, (4, 4, 1), A.TYPE_U8, 15) == A.ERR_OUT_OF_MEM
    lib.context_free(ctx)


# ---- Source/UnitTest/test_encode.cpp:170-299: non-finite inputs encode, and (beyond the reference's
# own assertion) to the same bytes as the reference ----------------------------------------------

@pytest.mark.parametrize("bad", [-np.inf, np.inf, np.nan], ids=["neg_inf", "pos_inf", "nan"])
@pytest.mark.parametrize("profile_name", ["PRF_LDR", "PRF_HDR", "PRF_HDR_RGB_LDR_A"])
def test_non_finite_input(lib, A, request, bad, profile_name):
    profile = getattr(A, profile_name)
    ref = request.getfixturevalue("ref") if os.path.exists(O.LIB_REF_NONE) else None
    for offset in range(4):
        img = np.full((4, 4, 4), 0.5, dtype=np.float32)
        flat = img.reshape(-1)
        flat[0 + offset] = bad
        if profile == A.PRF_LDR and bad == -np.inf:
            flat[4 + offset], flat[8 + offset], flat[12 + offset] = 0.75, 0.35, 0.0
        got = lib.compress(img, (4, 4), A.PRE_MEDIUM, profile=profile)      # raises on any error code
        assert got.size == 16
        if ref is not None:
            want = ref.compress(img, (4, 4), A.PRE_MEDIUM, profile=profile)
            assert got.tobytes() == want.tobytes(), (offset, got.tobytes().hex(), want.tobytes().hex())


# ---- Test/Data/{LDR,LDRS,HDR}-A-1x1.astc (Test/astc_test_functional.py:457-503) --------------------
# 1x1 inputs, 6x6 -exhaustive; the files' 16-byte payloads are quoted here.

KAT = [
    ("PRF_LDR", np.array([[[0x2B, 0xAD, 0x00, 0xFF]]], dtype=np.uint8), "fcfdffffffffffff2b2badad0000ffff"),
    ("PRF_LDR_SRGB", np.array([[[0x2B, 0x73, 0x00, 0xFF]]], dtype=np.uint8), "fcfdffffffffffff2b2b73730000ffff"),
    ("PRF_HDR_RGB_LDR_A", np.array([[[0x3E80, 0x3F60, 0x4040, 0x3C00]]], dtype=np.uint16).view(np.float16), "fcffffffffffffff803e603f4040003c"),
    ("PRF_HDR", np.array([[[0x3E80, 0x3F60, 0x4040, 0x3C00]]], dtype=np.uint16).view(np.float16), "fcffffffffffffff803e603f4040003c"),
]


@pytest.mark.parametrize("profile_name,pixel,payload", KAT, ids=[k[0] for k in KAT])
def test_reference_known_answer_files(lib, A, profile_name, pixel, payload):
    got = lib.compress(pixel, (6, 6), A.PRE_EXHAUSTIVE, profile=getattr(A, profile_name))
    assert got.tobytes().hex() == payload


# ---- argument validation (astcenc_entry.cpp:262-283, :494-498, :753-759, :1134-1182) -----------------

def test_config_init_errors(lib, A):
    assert lib.config_init(A.PRF_LDR, 3, 3, 1, A.PRE_MEDIUM, 0)[0] == A.ERR_BAD_BLOCK_SIZE
    assert lib.config_init(A.PRF_LDR, 6, 7, 1, A.PRE_MEDIUM, 0)[0] == A.ERR_BAD_BLOCK_SIZE
    assert lib.config_init(A.PRF_LDR, 6, 6, 1, 101.0, 0)[0] == A.ERR_BAD_QUALITY
    assert lib.config_init(A.PRF_LDR, 6, 6, 1, -1.0, 0)[0] == A.ERR_BAD_QUALITY
    assert lib.config_init(17, 6, 6, 1, A.PRE_MEDIUM, 0)[0] == A.ERR_BAD_PROFILE
    assert lib.config_init(A.PRF_LDR, 6, 6, 1, A.PRE_MEDIUM, 1 << 20)[0] == A.ERR_BAD_FLAGS
    assert lib.config_init(A.PRF_LDR, 6, 6, 1, A.PRE_MEDIUM, A.FLG_MAP_NORMAL | A.FLG_MAP_RGBM)[0] == A.ERR_BAD_FLAGS


def test_config_init_matches_reference_field_by_field(lib, ref, A):
    for profile in (A.PRF_LDR, A.PRF_LDR_SRGB, A.PRF_HDR, A.PRF_HDR_RGB_LDR_A):
        for block in ((4, 4), (5, 4), (6, 6), (8, 8), (10, 6), (12, 12)):
            for quality in (0.0, 5.0, 10.0, 35.0, 60.0, 80.0, 98.0, 99.5, 100.0):
                for flags in (0, A.FLG_MAP_NORMAL, A.FLG_USE_PERCEPTUAL, A.FLG_MAP_RGBM, A.FLG_USE_ALPHA_WEIGHT):
                    e0, want = ref.config_init(profile, block[0], block[1], 1, quality, flags)
                    e1, got = lib.config_init(profile, block[0], block[1], 1, quality, flags)
                    assert e0 == e1
                    assert want.as_dict() == got.as_dict(), (profile, block, quality, flags)


def test_context_alloc_errors(lib, A):
    err, cfg = lib.config_init(A.PRF_LDR, 6, 6, 1, A.PRE_MEDIUM, 0)
    assert lib.context_alloc(cfg, 0)[0] == A.ERR_BAD_PARAM
    cfg.cw_r_weight = cfg.cw_g_weight = cfg.cw_b_weight = cfg.cw_a_weight = 0.0
    assert lib.context_alloc(cfg, 1)[0] == A.ERR_BAD_PARAM
    lib.context_free(None)            # documented no-op


def test_compress_argument_errors(lib, A):
    ctx = _ctx(lib, A, threads=2)
    assert _raw_compress(lib, A, ctx, (4, 4, 1), A.TYPE_U8, 16, thread_index=2) == A.ERR_BAD_PARAM
    assert _raw_compress(lib, A, ctx, (4, 4, 1), A.TYPE_U8, 16, swizzle=(A.SWZ_R, A.SWZ_G, A.SWZ_B, A.SWZ_Z)) == A.ERR_BAD_SWIZZLE
    assert _raw_compress(lib, A, ctx, (4, 4, 1), A.TYPE_U8, 16, swizzle=(A.SWZ_R, A.SWZ_G, 9, A.SWZ_A)) == A.ERR_BAD_SWIZZLE
    lib.context_free(ctx)
    ctx = _ctx(lib, A, flags=A.FLG_DECOMPRESS_ONLY)
    assert _raw_compress(lib, A, ctx, (4, 4, 1), A.TYPE_U8, 16) == A.ERR_BAD_CONTEXT
    lib.context_free(ctx)


def test_error_strings_match_reference(lib, ref):
    for code in range(0, 14):
        assert lib.error_string(code) == ref.error_string(code), code


# ---- threading contract (astcenc.h:796-803, astcenc_entry.cpp:1185-1216) ---------------------------

def test_multiple_caller_threads_and_reset(lib, A):
    """N caller threads with unique thread_index all return only when the image is done; a reset is
    required between images for N > 1; the bytes equal a single-threaded context's."""
    img1, img2 = images.noisy(50, 44, 1), images.noisy(50, 44, 2)
    want1 = lib.compress(img1, (6, 6), A.PRE_FAST)
    want2 = lib.compress(img2, (6, 6), A.PRE_FAST)
    n = 4
    ctx = _ctx(lib, A, block=(6, 6), quality=A.PRE_FAST, threads=n)
    for img, want in ((img1, want1), (img2, want2)):
        out = np.zeros(want.size, dtype=np.uint8)
        errs = [None] * n
        seen_complete = [False] * n

        def work(i):
            errs[i] = lib.compress_raw(ctx, img, out, thread_index=i)
            seen_complete[i] = bool((out.reshape(-1, 16) == want.reshape(-1, 16)).all())
        ts = [threading.Thread(target=work, args=(i,)) for i in range(n)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        assert errs == [A.SUCCESS] * n
        assert all(seen_complete), "a caller returned before the whole image was written"
        assert lib.lib.astcenc_compress_reset(ctx) == A.SUCCESS
    lib.context_free(ctx)


def test_single_thread_context_auto_resets(lib, A):
    ctx = _ctx(lib, A, block=(6, 6), quality=A.PRE_FASTEST)
    for seed in (1, 2, 3):
        img = images.noisy(24, 24, seed)
        out = np.zeros(16 * 16, dtype=np.uint8)
        assert lib.compress_raw(ctx, img, out) == A.SUCCESS
        assert out.tobytes() == lib.compress(img, (6, 6), A.PRE_FASTEST).tobytes()
    lib.context_free(ctx)


def test_cancel_then_reset(lib, A):
    ctx = _ctx(lib, A, block=(6, 6), quality=A.PRE_FASTEST)
    assert lib.lib.astcenc_compress_cancel(ctx) == A.SUCCESS
    assert lib.lib.astcenc_compress_reset(ctx) == A.SUCCESS
    img = images.noisy(24, 24, 4)
    out = np.zeros(16 * 16, dtype=np.uint8)
    assert lib.compress_raw(ctx, img, out) == A.SUCCESS
    assert out.tobytes() == lib.compress(img, (6, 6), A.PRE_FASTEST).tobytes()
    lib.context_free(ctx)


def test_cancel_without_reset_on_single_thread_context(lib, A):
    """thread_count == 1: astcenc_compress_image resets implicitly, which also forgets a pending cancel
    (ref: astcenc_entry.cpp:1185-1188 -> astcenc_compress_reset -> ParallelManager::reset); the image after a
    cancel must therefore be compressed completely."""
    ctx = _ctx(lib, A, block=(6, 6), quality=A.PRE_FASTEST)
    img = images.noisy(48, 48, 8)
    want = lib.compress(img, (6, 6), A.PRE_FASTEST).tobytes()
    for _ in range(2):
        assert lib.lib.astcenc_compress_cancel(ctx) == A.SUCCESS
        out = np.full(64 * 16, 0xEE, dtype=np.uint8)
        assert lib.compress_raw(ctx, img, out) == A.SUCCESS
        assert out.tobytes() == want
    lib.context_free(ctx)


def test_cancel_is_sticky_on_multi_thread_context_until_reset(lib, A):
    """thread_count > 1: a cancelled context leaves the output untouched until astcenc_compress_reset."""
    ctx = _ctx(lib, A, block=(6, 6), quality=A.PRE_FASTEST, threads=2)
    img = images.noisy(48, 48, 9)
    assert lib.lib.astcenc_compress_cancel(ctx) == A.SUCCESS
    out = np.full(64 * 16, 0xEE, dtype=np.uint8)
    results = []
    ts = [threading.Thread(target=lambda i=i: results.append(lib.compress_raw(ctx, img, out, thread_index=i))) for i in range(2)]
    [t.start() for t in ts]; [t.join() for t in ts]
    assert results == [A.SUCCESS, A.SUCCESS]
    assert (out == 0xEE).all()                      # nothing was compressed, nothing was written
    assert lib.lib.astcenc_compress_reset(ctx) == A.SUCCESS
    ts = [threading.Thread(target=lambda i=i: results.append(lib.compress_raw(ctx, img, out, thread_index=i))) for i in range(2)]
    [t.start() for t in ts]; [t.join() for t in ts]
    assert out.tobytes() == lib.compress(img, (6, 6), A.PRE_FASTEST).tobytes()
    lib.context_free(ctx)


def test_progress_callback_monotonic_and_finishes(lib, A):
    seen = []
    cb = A.PROGRESS_CB(lambda p: seen.append(p))
    err, cfg = lib.config_init(A.PRF_LDR, 6, 6, 1, A.PRE_FASTEST, 0)
    cfg.progress_callback = cb
    err, ctx = lib.context_alloc(cfg, 1)
    assert err == A.SUCCESS
    img = images.noisy(96, 96, 6)
    out = np.zeros(16 * 16 * 16, dtype=np.uint8)
    assert lib.compress_raw(ctx, img, out) == A.SUCCESS
    lib.context_free(ctx)
    assert seen and seen == sorted(seen) and abs(seen[-1] - 100.0) < 1e-3


# ---- scope markers: what the drop-in declines, loudly -----------------------------------------------

def test_formerly_out_of_scope_paths_are_accepted(lib, A):
    """What earlier rounds answered with ASTCENC_ERR_NOT_IMPLEMENTED and the reference accepts: 3D footprints
    (tests/test_volume.py), alpha-scale radii above 80 and alpha-scale on a stack of slices (tests/test_alpha_scale.py)."""
    assert lib.config_init(A.PRF_LDR, 4, 4, 4, A.PRE_MEDIUM, 0)[0] == A.SUCCESS
    err, cfg = lib.config_init(A.PRF_LDR, 6, 6, 1, A.PRE_MEDIUM, 0)
    cfg.a_scale_radius = 81
    err, ctx = lib.context_alloc(cfg, 1)
    assert err == A.SUCCESS
    lib.context_free(ctx)
    cfg.a_scale_radius = 2
    err, ctx = lib.context_alloc(cfg, 1)
    assert err == A.SUCCESS
    try:
        import numpy as np
        vol = np.zeros((2, 12, 12, 4), dtype=np.uint8)
        assert lib.compress_raw(ctx, vol, np.zeros(2 * 4 * 16, dtype=np.uint8)) == A.SUCCESS
    finally:
        lib.context_free(ctx)
