# SPDX-License-Identifier: Apache-2.0
"""On-device quality metric (the CLI's -tl loop): blocks in device memory are
decoded into a device image and compared with the source there; only ten doubles come back.

  * astcenc_amd_decompress_image_device must produce the bytes astcenc_decompress_image produces;
  * astcenc_amd_compare_images_device must reproduce the sums of the reference's compute_error_metrics
    (Source/astcenccli_error_metrics.cpp:110-300), restated here in numpy: fp32 per-texel terms, fp64
    totals.  The totals are added in a different order, so the tolerance is 1e-12 relative.
Runs on the scalar CPU build here ("device" pointers are host pointers) and on the GPU with -m gpu."""
import ctypes as C

import numpy as np
import pytest

import images

LIBS = [pytest.param("emu", id="emu"), pytest.param("product", id="hip", marks=pytest.mark.gpu)]
REL = 1e-12


@pytest.fixture(params=LIBS)
def lib(request):
    return request.getfixturevalue(request.param)


class Dev:
    """A buffer in the library's "device" memory: HBM through torch for the product, numpy for the emulator."""

    def __init__(self, lib, array):
        self.gpu = lib.backend_name().startswith("hip")
        if self.gpu:
            import torch
            flat = np.ascontiguousarray(array).view(np.uint8).reshape(-1)
            self.t = torch.from_numpy(flat.copy()).cuda()
            self.ptr = self.t.data_ptr()
        else:
            self.a = np.ascontiguousarray(array).copy()
            self.ptr = self.a.ctypes.data
        self.dtype, self.shape = array.dtype, array.shape

    def host(self):
        if self.gpu:
            return self.t.cpu().numpy().view(self.dtype).reshape(self.shape)
        return self.a


def reference_sums(a, b):
    """compute_error_metrics' LDR accumulators, numpy restatement (fp32 terms, fp64 sums)."""
    def load(x):
        if x.dtype == np.uint8:
            return x.astype(np.float32) / np.float32(255.0)
        v = x.astype(np.float32)
        v = np.where(v > 0, v, np.float32(0))          # NaN -> 0 like the reference's max/min pair
        return np.minimum(v, np.float32(65504.0))
    c1, c2 = load(a).reshape(-1, 4), load(b).reshape(-1, 4)
    d = c1 - c2
    sq = (d * d).astype(np.float64).sum(axis=0)
    ds = d.copy()
    ds[:, :3] *= c1[:, 3:4]
    asq = (ds * ds).astype(np.float64).sum(axis=0)
    return sq, asq, float(c1[:, :3].max())


def compare(lib, ctx, A, a, b):
    da, db = Dev(lib, a), Dev(lib, b)
    types = {np.dtype(np.uint8): A.TYPE_U8, np.dtype(np.float16): A.TYPE_F16, np.dtype(np.float32): A.TYPE_F32}
    d = a.shape[0] if a.ndim == 4 else 1
    sums = A.ErrorSums()
    err = lib.lib.astcenc_amd_compare_images_device(ctx, da.ptr, types[a.dtype], db.ptr, types[b.dtype],
                                                    a.shape[-2], a.shape[-3], d, None, C.byref(sums))
    assert err == 0, lib.error_string(err)
    return sums


@pytest.fixture
def ctx66(lib, A):
    err, cfg = lib.config_init(A.PRF_LDR, 6, 6, 1, A.PRE_MEDIUM, 0)
    assert err == 0
    err, ctx = lib.context_alloc(cfg, 1)
    assert err == 0, lib.error_string(err)
    yield ctx
    lib.context_free(ctx)


def test_compare_matches_reference_formula(lib, A, ctx66):
    rng = np.random.default_rng(4)
    a = images.noisy(157, 93)
    b = np.clip(a.astype(np.int32) + rng.integers(-9, 10, a.shape), 0, 255).astype(np.uint8)
    half = (a.astype(np.float32) / 255.0 * 3.0).astype(np.float16)
    half[3, 5, 1] = np.float16(np.nan); half[4, 6, 2] = np.float16(np.inf); half[9, 9, 0] = np.float16(-2.0)
    f32 = (b.astype(np.float32) / 255.0 * 2.5).astype(np.float32)
    for x, y in ((a, b), (a, a), (half, f32), (a, f32), (f32, half)):
        sums = compare(lib, ctx66, A, x, y)
        sq, asq, peak = reference_sums(x, y)
        assert np.allclose(np.array(sums.squared_error), sq, rtol=REL, atol=0), (x.dtype, y.dtype)
        assert np.allclose(np.array(sums.alpha_scaled_squared_error), asq, rtol=REL, atol=0)
        assert sums.rgb_peak == peak and sums.texels == 157 * 93
    assert compare(lib, ctx66, A, a, a).psnr() == 999.0
    # the formula of the CLI report, against the module's host-side restatement
    sums = compare(lib, ctx66, A, a, b)
    assert abs(sums.psnr() - A.psnr_rgba8(a, b)) < 1e-5
    assert sums.psnr(3) > 0 and sums.psnr(4, alpha_scaled=True) >= sums.psnr() - 1e-9


def test_device_round_trip_psnr(lib, ref, A, ctx66):
    """compress -> (blocks stay on the device) -> decompress on the device -> compare on the device."""
    w, h = 200, 150
    img = images.noisy(w, h)
    blocks = lib.compress(img, (6, 6), A.PRE_MEDIUM)
    d_blocks = Dev(lib, blocks)
    d_out = Dev(lib, np.zeros_like(img))
    swz = A.Swizzle(*A.SWZ_RGBA)
    err = lib.lib.astcenc_amd_decompress_image_device(ctx66, d_blocks.ptr, blocks.nbytes, d_out.ptr, w, h, 1, A.TYPE_U8, C.byref(swz), None)
    assert err == 0, lib.error_string(err)
    decoded = d_out.host()
    assert np.array_equal(decoded, ref.decompress(blocks, w, h, (6, 6)))
    sums = compare(lib, ctx66, A, img, decoded)
    assert abs(sums.psnr() - A.psnr_rgba8(img, decoded)) < 1e-5
    assert sums.psnr() > 30.0
    # argument checks follow astcenc_decompress_image (ref: astcenc_entry.cpp:1296-1322)
    assert lib.lib.astcenc_amd_decompress_image_device(ctx66, d_blocks.ptr, blocks.nbytes - 1, d_out.ptr, w, h, 1, A.TYPE_U8, C.byref(swz), None) == A.ERR_OUT_OF_MEM
    assert lib.lib.astcenc_amd_decompress_image_device(ctx66, d_blocks.ptr, blocks.nbytes, d_out.ptr, 0, h, 1, A.TYPE_U8, C.byref(swz), None) == A.ERR_BAD_PARAM
    sums = A.ErrorSums()
    assert lib.lib.astcenc_amd_compare_images_device(ctx66, d_out.ptr, 0, d_out.ptr, 0, 0, h, 1, None, C.byref(sums)) == A.ERR_BAD_PARAM


def test_device_round_trip_volume(lib, ref, A):
    vol = images.volume("grad", 9, 14, 18)
    blocks = lib.compress(vol, (4, 4, 3), A.PRE_MEDIUM)
    err, cfg = lib.config_init(A.PRF_LDR, 4, 4, 3, A.PRE_MEDIUM, 0)
    err, ctx = lib.context_alloc(cfg, 1)
    assert err == 0
    try:
        d_blocks, d_out = Dev(lib, blocks), Dev(lib, np.zeros_like(vol))
        swz = A.Swizzle(*A.SWZ_RGBA)
        err = lib.lib.astcenc_amd_decompress_image_device(ctx, d_blocks.ptr, blocks.nbytes, d_out.ptr, 18, 14, 9, A.TYPE_U8, C.byref(swz), None)
        assert err == 0
        assert np.array_equal(d_out.host(), ref.decompress(blocks, 18, 14, (4, 4, 3), depth=9))
        sums = compare(lib, ctx, A, vol, d_out.host())
        assert sums.texels == 9 * 14 * 18 and sums.psnr() > 25.0
    finally:
        lib.context_free(ctx)
