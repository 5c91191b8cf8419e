# SPDX-License-Identifier: Apache-2.0
"""Photographic sanity point: a window of the reference's own Khronos test image
(tests/golden/make_photo_fixture.py) instead of synthetic content.  The committed blocks were produced by
the real reference encoder; the HIP path must reproduce them byte for byte, decode them like the reference
decoder, and land on the PSNR recorded when the fixture was made."""
import json
import os

import numpy as np
import pytest

import images

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAME = "photo_khronos_rgba_base_500x460"
LIBS = [pytest.param("emu", id="emu"), pytest.param("product", id="hip", marks=pytest.mark.gpu)]


@pytest.fixture(params=LIBS)
def lib(request):
    return request.getfixturevalue(request.param)


def _photo():
    from PIL import Image
    img = np.array(Image.open(os.path.join(GOLDEN, NAME + ".png")))
    meta = json.load(open(os.path.join(GOLDEN, NAME + ".json")))
    assert img.shape == (460, 500, 4)
    import hashlib
    assert hashlib.sha256(img.tobytes()).hexdigest() == meta["input_sha256"]
    return np.ascontiguousarray(img), meta


@pytest.mark.parametrize("tag", ["6x6_medium", "8x8_thorough", "4x4_fast"])
def test_photo_matches_reference_blocks(lib, A, tag):
    if tag != "4x4_fast" and not lib.backend_name().startswith("hip"):
        pytest.skip("the sequential CPU build needs minutes for this setting; the GPU run covers it")
    img, meta = _photo()
    s = meta["settings"][tag]
    want = np.load(os.path.join(GOLDEN, "%s_%s.npy" % (NAME, tag)))
    got = lib.compress(img, tuple(s["block"]), s["quality"])
    bad = images.mismatches(want, got)
    assert len(bad) == 0, "%d of %d blocks differ: %s" % (len(bad), want.size // 16, bad[:8])
    dec = lib.decompress(got, 500, 460, tuple(s["block"]))
    assert abs(A.psnr_rgba8(img, dec) - s["psnr_rgba_db"]) < 1e-4
