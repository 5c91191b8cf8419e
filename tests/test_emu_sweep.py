# SPDX-License-Identifier: Apache-2.0
"""Every footprint through the sequential build of the kernel source, on the CPU: the 14 2D and the 10 3D footprints x four
presets x two image classes against the reference encoder's bytes.  A footprint-specific slip -- an LDS layout that does
not fit (make_lds_layout checks itself in this build), a table whose unused entries are not what a kernel loop relies on,
a loop bound tied to the texel count -- shows up here without a GPU; tools/gpu_sweep.py / gpu_sweep_3d.py are the same
matrix, wider, through the HIP library on the GPU box.  (Test infrastructure: oracle/emu and oracle/_ref only.)"""
import numpy as np
import pytest

import images

FOOT_2D = [(4, 4), (5, 4), (5, 5), (6, 5), (6, 6), (8, 5), (8, 6), (8, 8), (10, 5), (10, 6), (10, 8), (10, 10), (12, 10), (12, 12)]
FOOT_3D = [(3, 3, 3), (4, 3, 3), (4, 4, 3), (4, 4, 4), (5, 4, 4), (5, 5, 4), (5, 5, 5), (6, 5, 5), (6, 6, 5), (6, 6, 6)]
PRESETS = [0.0, 10.0, 60.0, 98.0]      # -fastest, -fast, -medium, -thorough


@pytest.mark.parametrize("block", FOOT_2D, ids=lambda b: "%dx%d" % b)
def test_every_2d_footprint(ref, emu, block):
    w, h = block[0] * 3 + 1, block[1] * 2 + 2                      # ragged on both edges
    for name, img in (("noisy", images.noisy(w, h, 77)), ("two_colour", images.two_colour(w, h, 78))):
        for quality in PRESETS:
            want = ref.compress(img, block, quality)
            got = emu.compress(img, block, quality)
            bad = images.mismatches(want, got)
            assert len(bad) == 0, "%s q=%g: blocks differ: %s" % (name, quality, bad[:8])


@pytest.mark.parametrize("block", FOOT_3D, ids=lambda b: "%dx%dx%d" % b)
def test_every_3d_footprint(ref, emu, block):
    d, h, w = block[2] + 1, block[1] + 2, block[0] * 2 + 1           # ragged on all three edges
    for kind in ("grad", "alpha"):
        vol = images.volume(kind, d, h, w, seed=79)
        for quality in (0.0, 60.0, 98.0):
            want = ref.compress(vol, block, quality)
            got = emu.compress(vol, block, quality)
            assert np.array_equal(want, got), "%s q=%g" % (kind, quality)


@pytest.mark.parametrize("block", [(4, 4), (6, 6), (8, 8), (10, 6), (12, 12)], ids=lambda b: "%dx%d" % b)
def test_hdr_profiles_across_footprints(ref, emu, A, block):
    for name, img in images.hdr_variants(block[0] * 2 + 3, block[1] * 2 + 1).items():
        img = img.astype(np.float16)
        for profile in (A.PRF_HDR, A.PRF_HDR_RGB_LDR_A):
            want = ref.compress(img, block, 60.0, profile=profile)
            got = emu.compress(img, block, 60.0, profile=profile)
            bad = images.mismatches(want, got)
            assert len(bad) == 0, "%s profile %d: blocks differ: %s" % (name, profile, bad[:8])
