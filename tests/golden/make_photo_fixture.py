#!/usr/bin/env python3
# SPDX-License-Identifier: Apache-2.0
"""Photographic sanity fixture ("add at least one photographic image, e.g. Test/Images/Khronos/*").

Cuts a 500x460 window (not a multiple of any footprint: the clamped edge blocks are in) out of
/root/reference/Test/Images/Khronos/LDR-RGBA/ldr-rgba-base.png -- an extract of the KhronosGroup
glTF-Asset-Generator test suite, MIT licensed (Test/Images/Khronos/LICENSE.txt) -- stores it losslessly as
tests/golden/photo_khronos_rgba_base_500x460.png and stores what the REAL reference encoder
(oracle/_ref/libastcenc-none.so) makes of it for three BASELINE-shaped settings.

Run in the dev container:  python tests/golden/make_photo_fixture.py
"""
import hashlib
import json
import os
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "astc-encoder_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import astcenc_amd as A  # noqa: E402
import oracle_libs as O  # noqa: E402

SRC = "/root/reference/Test/Images/Khronos/LDR-RGBA/ldr-rgba-base.png"
NAME = "photo_khronos_rgba_base_500x460"
SETTINGS = {"6x6_medium": ((6, 6), 60.0), "8x8_thorough": ((8, 8), 98.0), "4x4_fast": ((4, 4), 10.0)}


def main():
    full = np.array(Image.open(SRC))
    assert full.shape == (1024, 1024, 4) and full.dtype == np.uint8
    crop = np.ascontiguousarray(full[300:760, 260:760])
    Image.fromarray(crop, "RGBA").save(os.path.join(HERE, NAME + ".png"), optimize=True)
    back = np.array(Image.open(os.path.join(HERE, NAME + ".png")))
    assert np.array_equal(back, crop)
    ref = A.Library(O.LIB_REF_NONE)
    manifest = {"source": "Test/Images/Khronos/LDR-RGBA/ldr-rgba-base.png [300:760, 260:760], MIT (Khronos Group, glTF-Asset-Generator)",
                "size": [500, 460], "input_sha256": hashlib.sha256(crop.tobytes()).hexdigest(), "settings": {}}
    for tag, (block, quality) in SETTINGS.items():
        blocks = ref.compress(crop, block, quality)
        np.save(os.path.join(HERE, "%s_%s.npy" % (NAME, tag)), blocks)
        dec = ref.decompress(blocks, 500, 460, block)
        manifest["settings"][tag] = {"block": block, "quality": quality, "blocks_sha256": hashlib.sha256(blocks.tobytes()).hexdigest(),
                                     "psnr_rgba_db": round(A.psnr_rgba8(crop, dec), 4)}
        print(tag, blocks.size // 16, "blocks", manifest["settings"][tag]["psnr_rgba_db"], "dB")
    with open(os.path.join(HERE, NAME + ".json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
