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


def reference_hdr_sums(a, b, fstop_lo, fstop_hi):
    """The HDR accumulators of compute_error_metrics (astcenccli_error_metrics.cpp:60-107, :262-268), numpy
    restatement: fp32 terms, fp64 sums; powf = correctly rounded float power (double pow, rounded)."""
    def load(x):
        v = x.astype(np.float32)
        v = np.where(v > 0, v, np.float32(0))
        return np.minimum(v, np.float32(65504.0))

    def log2_poly(x):
        i = x.view(np.int32)
        e = (((i.astype(np.int64) & 0x7F800000) >> 23) - 127).astype(np.float32)
        m = ((i & 0x007FFFFF) | 0x3F800000).view(np.float32)
        p = np.float32(0.0596515482674574969533)
        for c in (-0.465725644288844778798, 1.48116647521213171641, -2.52074962577807006663, 2.8882704548164776201):
            p = (p * m).astype(np.float32) + np.float32(c)
        p = p * (m - np.float32(1.0))
        return (p + e).astype(np.float32)

    def operator(v, stop):
        scale = np.float32(2.0) ** np.float32(stop)
        t = np.power((v * scale).astype(np.float32).astype(np.float64), np.float64(np.float32(1.0) / np.float32(2.2))).astype(np.float32)
        return np.clip(t * np.float32(255.0), np.float32(0), np.float32(255.0)).astype(np.float32)

    c1, c2 = np.ascontiguousarray(load(a).reshape(-1, 4)), np.ascontiguousarray(load(b).reshape(-1, 4))
    ld = log2_poly(c1) - log2_poly(c2)
    log_sq = (ld * ld).astype(np.float64).sum(axis=0)
    summa = np.zeros_like(c1)
    for stop in range(fstop_lo, fstop_hi + 1):
        d = operator(c1, stop) - operator(c2, stop)
        summa = (summa + d * d).astype(np.float32)
    return log_sq, summa.astype(np.float64).sum(axis=0)


def test_hdr_sums_match_reference_formula(lib, A, ctx66):
    """mPSNR and log RMSE (what the reference CLI reports for BASELINE config 4).  Tolerance 1e-9 relative on the
    fp64 sums: the per-texel fp32 terms are the reference's arithmetic, the totals are added in another order, and
    the one libm call of the reference (powf) is a correctly rounded power on both sides."""
    rng = np.random.default_rng(12)
    src = images.hdr_f16(120, 88).astype(np.float16)
    noisy = (src.astype(np.float32) * (1.0 + rng.normal(0, 0.03, src.shape))).astype(np.float16)
    noisy[5, 7, 0] = np.float16(0.0); noisy[8, 3, 1] = np.float16(np.inf)
    types = {np.dtype(np.float16): A.TYPE_F16, np.dtype(np.float32): A.TYPE_F32}
    for x, y, lo, hi in ((src, noisy, -10, 10), (src.astype(np.float32), noisy, -4, 3), (src, src, 0, 0)):
        dx, dy = Dev(lib, x), Dev(lib, y)
        sums, hdr = A.ErrorSums(), A.HdrErrorSums()
        err = lib.lib.astcenc_amd_compare_images_hdr_device(ctx66, dx.ptr, types[x.dtype], dy.ptr, types[y.dtype], 120, 88, 1, lo, hi,
                                                            None, C.byref(sums), C.byref(hdr))
        assert err == 0, lib.error_string(err)
        log_sq, mp = reference_hdr_sums(x, y, lo, hi)
        assert np.allclose(np.array(hdr.log2_squared_error), log_sq, rtol=1e-9, atol=0)
        assert np.allclose(np.array(hdr.mpsnr_squared_error), mp, rtol=1e-9, atol=0)
        sq, asq, peak = reference_sums(x, y)
        assert np.allclose(np.array(sums.squared_error), sq, rtol=REL, atol=0) and sums.rgb_peak == peak
        assert (hdr.fstop_lo, hdr.fstop_hi) == (lo, hi)
        if x is y:
            assert hdr.mpsnr(sums.texels) == 999.0 and hdr.log_rmse(sums.texels) == 0.0
        else:
            assert 10.0 < hdr.mpsnr(sums.texels) < 80.0 and hdr.log_rmse(sums.texels) > 0.0
    sums, hdr = A.ErrorSums(), A.HdrErrorSums()
    d = Dev(lib, src)
    bad = lib.lib.astcenc_amd_compare_images_hdr_device(ctx66, d.ptr, A.TYPE_F16, d.ptr, A.TYPE_F16, 120, 88, 1, 3, 2, None, C.byref(sums), C.byref(hdr))
    assert bad == A.ERR_BAD_PARAM


def test_figures_match_the_reference_report(lib, A, ctx66, tmp_path):
    """The reference's own compute_error_metrics (astcenccli_error_metrics.cpp:110, compiled from where it lies into
    oracle/_ref/metrics_harness) prints its report for two raw images; the figures derived from the device sums must
    agree to the 4 decimals it prints -- LDR and HDR (mPSNR, LogRMSE, PSNR normalised to peak) alike."""
    import os
    import re
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "metrics_harness")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/metrics_harness not built (no /root/reference on this machine)")
    rng = np.random.default_rng(31)
    a8 = images.noisy(150, 110)
    b8 = np.clip(a8.astype(np.int32) + rng.integers(-7, 8, a8.shape), 0, 255).astype(np.uint8)
    ah = images.hdr_f16(150, 110)
    bh = (ah.astype(np.float32) * (1.0 + rng.normal(0, 0.02, ah.shape))).astype(np.float16)
    types = {np.dtype(np.uint8): ("u8", A.TYPE_U8), np.dtype(np.float16): ("f16", A.TYPE_F16)}
    for x, y, hdr in ((a8, b8, False), (ah, bh, True)):
        px, py = str(tmp_path / "x.raw"), str(tmp_path / "y.raw")
        x.tofile(px); y.tofile(py)
        tname, ttype = types[x.dtype]
        report = subprocess.run([exe, tname, "150", "110", px, py, "1" if hdr else "0", "-10", "10"], capture_output=True, text=True, check=True).stdout

        def figure(label):
            m = re.search(re.escape(label) + r"\s*:?\s*(-?[0-9.]+)", report)
            assert m, (label, report)
            return float(m.group(1))
        dx, dy = Dev(lib, x), Dev(lib, y)
        sums, hs = A.ErrorSums(), A.HdrErrorSums()
        err = lib.lib.astcenc_amd_compare_images_hdr_device(ctx66, dx.ptr, ttype, dy.ptr, ttype, 150, 110, 1, -10, 10, None, C.byref(sums), C.byref(hs))
        assert err == 0
        assert abs(sums.psnr() - figure("PSNR (LDR-RGBA):")) < 6e-5
        assert abs(sums.psnr(4, alpha_scaled=True) - figure("Alpha-weighted PSNR:")) < 6e-5
        assert abs(sums.psnr(3) - figure("PSNR (LDR-RGB):")) < 6e-5
        if hdr:
            assert abs(hs.mpsnr(sums.texels) - figure("mPSNR (RGB):")) < 6e-5
            assert abs(hs.log_rmse(sums.texels) - figure("LogRMSE (RGB):")) < 6e-5
            assert abs(sums.psnr(3) + 20.0 * np.log10(sums.rgb_peak) - figure("PSNR (RGB norm to peak):")) < 6e-5


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
