# SPDX-License-Identifier: Apache-2.0
"""CPU-side parity: the wave-emulator build of the kernel source vs the real reference encoder.

This is how the block compressor is debugged without a GPU: oracle/emu runs the same wave_*.h code
sequentially.  (The GPU parity tests in test_gpu_parity.py are the ones that count for the product.)
"""
import pytest

import images

CASES = [
    # (block, quality, image, size, partition limit override)
    ((4, 4), 0.0, "noisy", (128, 128), 1),      # BASELINE config 1 shape (1-partition only)
    ((4, 4), 0.0, "noisy", (64, 64), None),
    ((6, 6), 60.0, "noisy", (96, 96), None),     # BASELINE config 2 shape
    ((6, 6), 60.0, "flat", (72, 72), None),
    ((6, 6), 60.0, "gray", (72, 72), None),
    ((6, 6), 60.0, "two_colour", (72, 72), None),
    ((8, 8), 98.0, "noisy", (64, 64), None),     # BASELINE config 3 shape
    ((5, 5), 10.0, "random", (40, 45), None),    # ragged edges, texel count not a multiple of 4
    ((8, 5), 60.0, "smooth", (64, 40), None),
    ((6, 6), 60.0, "noisy", (7, 5), None),       # image smaller than two blocks
]


@pytest.mark.parametrize("block,quality,name,size,plimit", CASES)
def test_emu_matches_reference(ref, emu, block, quality, name, size, plimit):
    img = images.ALL[name](*size)
    tweak = (lambda c: setattr(c, "tune_partition_count_limit", plimit)) if plimit else None
    want = ref.compress(img, block, quality, tweak=tweak)
    got = emu.compress(img, block, quality, tweak=tweak)
    bad = images.mismatches(want, got)
    assert len(bad) == 0, "blocks differ: %s" % bad[:8]


def test_emu_flags_swizzle_profiles(ref, emu, A):
    img = images.noisy(60, 60, 5)
    for flags, swz in [(A.FLG_USE_ALPHA_WEIGHT, A.SWZ_RGBA), (A.FLG_USE_PERCEPTUAL, A.SWZ_RGBA),
                       (0, (A.SWZ_B, A.SWZ_G, A.SWZ_R, A.SWZ_1)), (A.FLG_MAP_NORMAL, (A.SWZ_R, A.SWZ_R, A.SWZ_R, A.SWZ_G)),
                       (A.FLG_USE_DECODE_UNORM8, A.SWZ_RGBA), (A.FLG_MAP_RGBM, A.SWZ_RGBA)]:
        want = ref.compress(img, (6, 6), 60.0, flags=flags, swizzle=swz)
        got = emu.compress(img, (6, 6), 60.0, flags=flags, swizzle=swz)
        assert len(images.mismatches(want, got)) == 0, (flags, swz)
    want = ref.compress(img, (6, 6), 60.0, profile=A.PRF_LDR_SRGB)
    got = emu.compress(img, (6, 6), 60.0, profile=A.PRF_LDR_SRGB)
    assert len(images.mismatches(want, got)) == 0


def test_emu_float_inputs(ref, emu):
    import numpy as np
    img8 = images.noisy(36, 36, 9)
    for dt in (np.float16, np.float32):
        img = (img8.astype(np.float32) / 255.0).astype(dt)
        want = ref.compress(img, (6, 6), 60.0)
        got = emu.compress(img, (6, 6), 60.0)
        assert len(images.mismatches(want, got)) == 0, dt


@pytest.mark.parametrize("block,quality", [((10, 10), 60.0), ((12, 12), 10.0), ((10, 5), 60.0), ((12, 10), 60.0), ((6, 6), 100.0), ((4, 4), 99.0)])
def test_emu_large_footprints_and_limits(ref, emu, block, quality):
    img = images.noisy(48, 48, 11)
    want = ref.compress(img, block, quality)
    got = emu.compress(img, block, quality)
    assert len(images.mismatches(want, got)) == 0


@pytest.mark.parametrize("profile_name", ["PRF_HDR", "PRF_HDR_RGB_LDR_A"])
def test_emu_hdr_profiles(ref, emu, A, profile_name):
    import numpy as np
    profile = getattr(A, profile_name)
    for name, img in images.hdr_variants(48, 48).items():
        for dt in (np.float16, np.float32):
            im = img.astype(dt)
            want = ref.compress(im, (6, 6), 60.0, profile=profile)
            got = emu.compress(im, (6, 6), 60.0, profile=profile)
            assert len(images.mismatches(want, got)) == 0, (name, dt)
    im = images.hdr_f16(40, 40)
    for block, q in [((4, 4), 0.0), ((8, 8), 98.0), ((5, 4), 60.0)]:
        want = ref.compress(im, block, q, profile=profile)
        got = emu.compress(im, block, q, profile=profile)
        assert len(images.mismatches(want, got)) == 0, (block, q)
