# SPDX-License-Identifier: Apache-2.0
"""Lane-order race check of the kernel source.  The sequential build (oracle/emu) runs the lanes of a WV_FOR / WV_QUADS
loop one after the other, so a body that reads what ANOTHER lane of the same loop wrote -- a missing WV_SYNC(), i.e. a
race on the device, where the lanes run side by side -- can go unnoticed there.  A second build visits the lanes in
reverse order (-DASTC_EMU_REVERSE_LANES): any dependence between the lanes of one loop then changes the output bytes.
Test infrastructure only; no GPU needed."""
import os
import subprocess

import numpy as np
import pytest

import oracle_libs as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = [((6, 6), 60.0, "ldr"), ((4, 4), 60.0, "ldr"), ((8, 8), 98.0, "ldr"), ((5, 4), 10.0, "ldr"), ((10, 8), 60.0, "ldr"),
         ((12, 12), 60.0, "ldr"), ((6, 5), 0.0, "ldr"), ((8, 6), 100.0, "ldr"), ((6, 6), 60.0, "hdr"), ((8, 8), 98.0, "hdr"),
         ((4, 4, 4), 60.0, "ldr"), ((3, 3, 3), 98.0, "ldr"), ((6, 6, 6), 60.0, "ldr")]


@pytest.fixture(scope="module")
def libs(built, A):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle", "emu"), "reverse"])
    return A.Library(O.LIB_EMU), A.Library(O.LIB_EMU_REVERSE)


@pytest.mark.parametrize("block,quality,kind", CASES)
def test_output_does_not_depend_on_the_lane_order(libs, A, block, quality, kind):
    forward, reverse = libs
    if len(block) == 3:
        vol = np.stack([A.synthetic_image(16, 16, seed=7 + z) for z in range(8)])
        a = forward.compress(vol, block, quality)
        b = reverse.compress(vol, block, quality)
    elif kind == "hdr":
        img = A.synthetic_hdr_image(36, 36, 3)
        a = forward.compress(img, block, quality, profile=A.PRF_HDR)
        b = reverse.compress(img, block, quality, profile=A.PRF_HDR)
    else:
        img = A.synthetic_image(48, 40, seed=11)
        a = forward.compress(img, block, quality)
        b = reverse.compress(img, block, quality)
    a, b = np.asarray(a).reshape(-1, 16), np.asarray(b).reshape(-1, 16)
    differing = np.flatnonzero((a != b).any(axis=1))
    assert differing.size == 0, "%d of %d blocks depend on the lane order (first: %s)" % (differing.size, a.shape[0], differing[:8])


@pytest.mark.parametrize("block", [(4, 4), (6, 6), (8, 5), (12, 12), (4, 4, 4)])
def test_decoder_output_does_not_depend_on_the_lane_order(libs, A, block):
    """The batched decoder (runs of 32 blocks: headers, BISE groups, endpoints and texels on different lane maps) gives
    the same image with the lanes of every loop visited forwards and backwards -- real encoder output and random bit
    patterns, wide enough for full runs and a tail."""
    forward, reverse = libs
    rng = np.random.default_rng(5 + block[0])
    if len(block) == 3:
        w, h, d = 41 * block[0] - 1, 3 * block[1], 2 * block[2]
        n = 41 * 3 * 2
    else:
        w, h, d = 75 * block[0] - 2, 3 * block[1] + 1, None
        n = 75 * 4
    data = rng.integers(0, 256, size=n * 16, dtype=np.uint8)
    b = data.reshape(-1, 16)
    b[::7, 0] = 0xFC; b[::7, 1] |= 0x01; b[::14, 1] = 0xFD
    b[1::5, 1] &= 0xE7
    b[2::9, 0] &= 0xFC; b[2::9, 0] |= 0x01
    for profile in (A.PRF_LDR, A.PRF_HDR):
        for out_type in (np.uint8, np.float16):
            kw = dict(profile=profile, out_type=out_type)
            if d is not None:
                kw["depth"] = d
            a = forward.decompress(data, w, h, block, **kw)
            c = reverse.decompress(data, w, h, block, **kw)
            assert a.tobytes() == c.tobytes(), (block, profile, out_type)
