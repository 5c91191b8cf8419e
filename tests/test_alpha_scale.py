# SPDX-License-Identifier: Apache-2.0
"""config.a_scale_radius (the CLI's -a option): the alpha-average pre-pass and the per-block
transparency test, byte for byte against the reference (astcenc_compute_variance.cpp,
astcenc_entry.cpp:974-1034)."""
import numpy as np
import pytest

import images

LIBS = [pytest.param("emu", id="emu"), pytest.param("product", id="hip", marks=pytest.mark.gpu)]


@pytest.fixture(params=LIBS)
def lib(request):
    return request.getfixturevalue(request.param)


def holes(w, h, seed, dtype=np.uint8):
    """Noisy image with transparent rectangles of various sizes, isolated opaque specks inside them and
    very faint alpha near some edges."""
    rng = np.random.default_rng(seed)
    img = images.noisy(w, h, seed).copy()
    for _ in range(10):
        x0, y0 = rng.integers(0, w - 8), rng.integers(0, h - 8)
        ww, hh = rng.integers(6, max(7, w // 2)), rng.integers(6, max(7, h // 2))
        img[y0:y0 + hh, x0:x0 + ww, 3] = 0
    for _ in range(12):
        img[rng.integers(0, h), rng.integers(0, w), 3] = rng.integers(1, 4)
    img[: h // 3, : w // 3] = (9, 9, 9, 0)          # a large fully transparent corner (skippable blocks)
    if dtype == np.uint8:
        return img
    f = img.astype(np.float32) / 255.0
    f[h // 2:, : w // 4, 3] *= 1e-4                   # alpha well below the threshold in float inputs
    return f.astype(dtype)


@pytest.mark.parametrize("radius", [1, 2, 4, 8, 13, 40, 80])
@pytest.mark.parametrize("block", [(4, 4), (6, 6), (8, 5), (12, 12)])
def test_alpha_scale_matches_reference(lib, ref, A, radius, block):
    w, h = 97, 75                                        # several 32x32 regions with ragged edges
    img = holes(w, h, 3 + radius)

    def tweak(cfg):
        cfg.a_scale_radius = radius
    for flags in (0, A.FLG_USE_ALPHA_WEIGHT):
        want = ref.compress(img, block, 60.0, flags=flags, tweak=tweak)
        got = lib.compress(img, block, 60.0, flags=flags, tweak=tweak)
        bad = images.mismatches(want, got)
        assert len(bad) == 0, (flags, bad[:8])
    plain = ref.compress(img, block, 60.0)
    assert (plain != want).any(), "the test image must contain blocks that the alpha test skips"


@pytest.mark.parametrize("dtype", [np.float16, np.float32])
def test_alpha_scale_float_inputs_and_swizzle(lib, ref, A, dtype):
    img = holes(70, 66, 11, dtype)

    def tweak(cfg):
        cfg.a_scale_radius = 3
    for swz in (A.SWZ_RGBA, (A.SWZ_R, A.SWZ_G, A.SWZ_B, A.SWZ_1), (A.SWZ_A, A.SWZ_B, A.SWZ_G, A.SWZ_R), (A.SWZ_R, A.SWZ_G, A.SWZ_B, A.SWZ_0)):
        want = ref.compress(img, (6, 6), 10.0, swizzle=swz, tweak=tweak)
        got = lib.compress(img, (6, 6), 10.0, swizzle=swz, tweak=tweak)
        assert len(images.mismatches(want, got)) == 0, swz


@pytest.mark.parametrize("radius", [81, 130])
def test_alpha_scale_large_radius(lib, ref, radius):
    """Radii whose padded 32x32 tile no longer fits the CU's LDS (the reference has no limit): the pre-pass keeps the tile
    in a per-workgroup slice of device memory instead."""
    img = holes(150, 70, 40 + radius)
    img[:, 100:] = (9, 9, 9, 0)                          # transparent far beyond the reach of the filter on one side
    img[20:40, 120:140, 3] = 0

    def tweak(cfg):
        cfg.a_scale_radius = radius
    want = ref.compress(img, (6, 6), 10.0, tweak=tweak)
    got = lib.compress(img, (6, 6), 10.0, tweak=tweak)
    assert len(images.mismatches(want, got)) == 0


@pytest.mark.parametrize("radius,slices", [(1, 2), (2, 5), (3, 20), (9, 3)])
def test_alpha_scale_on_a_stack_of_slices(lib, ref, radius, slices):
    """Multi-slice image, 2D footprint: the reference averages alpha over a (2r+1)^3 box in 16x16x16 regions and then
    reads the averages around slice 0 for every slice (astcenc_entry.cpp:1001: `ay * dim_x + ax`); the same bytes
    come out here.  F16 input: every slice is loaded from its own data, so the slices differ."""
    w, h = 50, 37
    vol = np.stack([holes(w, h, 70 + z, np.float16) for z in range(slices)])
    vol[:, : h // 2, : w // 2, 3] = 0                     # transparent (but coloured) in every slice: skippable blocks
    vol[slices - 1, 3, 3, 3] = 1.0                       # ... except one texel of the last slice (in reach for small stacks)

    def tweak(cfg):
        cfg.a_scale_radius = radius
    want = ref.compress(vol, (5, 4), 10.0, tweak=tweak)
    got = lib.compress(vol, (5, 4), 10.0, tweak=tweak)
    assert len(images.mismatches(want, got)) == 0
    if radius <= 3:
        plain = ref.compress(vol, (5, 4), 10.0)
        assert (plain != want).any(), "the test volume must contain blocks that the alpha test skips"
