# SPDX-License-Identifier: Apache-2.0
"""Pins oracle/astc_decode.c (independent plain-C restatement of the block decoder) against the
reference's own astcenc_decompress_image, then uses it as a format oracle on the committed golden
encoder outputs: the bytes must decode, with no error blocks, to an image close to the source.
"""
import ctypes
import json
import os

import numpy as np
import pytest

import images

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_DECODE = os.path.join(ROOT, "oracle", "_build", "libastc_decode.so")


@pytest.fixture(scope="module")
def dec(built):
    if not os.path.exists(LIB_DECODE):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "decode"])
    lib = ctypes.CDLL(LIB_DECODE)
    lib.astc_oracle_decode_image.restype = ctypes.c_int
    lib.astc_oracle_decode_image.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                            ctypes.c_int, ctypes.c_void_p]

    def decode(blocks, block, w, h, srgb=False):
        blocks = np.ascontiguousarray(blocks, dtype=np.uint8)
        assert blocks.size == 16 * ((w + block[0] - 1) // block[0]) * ((h + block[1] - 1) // block[1])
        out = np.zeros((h, w, 4), dtype=np.uint8)
        errors = lib.astc_oracle_decode_image(blocks.ctypes.data, block[0], block[1], w, h, int(srgb), out.ctypes.data)
        return out, errors
    return decode


FOOTPRINTS = [(4, 4), (5, 4), (5, 5), (6, 5), (6, 6), (8, 5), (8, 6), (8, 8), (10, 5), (10, 6), (10, 8), (10, 10), (12, 10), (12, 12)]


@pytest.mark.parametrize("block", FOOTPRINTS)
def test_decoder_matches_reference_on_encoder_output(ref, dec, block, A):
    w, h = block[0] * 7 + 3, block[1] * 6 + 1            # ragged edges
    for name, quality in (("noisy", 60.0), ("random", 10.0), ("two_colour", 98.0)):
        img = images.ALL[name](w, h)
        data = ref.compress(img, block, quality)
        want = ref.decompress(data, w, h, block)
        got, errors = dec(data, block, w, h)
        assert errors == 0
        assert np.array_equal(want, got), (name, np.argwhere(want != got)[:4])
    data = ref.compress(images.noisy(w, h, 3), block, 60.0, profile=A.PRF_LDR_SRGB)
    want = ref.decompress(data, w, h, block, profile=A.PRF_LDR_SRGB)
    got, _ = dec(data, block, w, h, srgb=True)
    assert np.array_equal(want, got)


@pytest.mark.parametrize("block", [(4, 4), (6, 6), (8, 5), (12, 12)])
def test_decoder_matches_reference_on_random_bit_patterns(ref, dec, block):
    """Random 128-bit patterns reach reserved block modes, illegal void extents, HDR endpoint
    formats and over-long colour streams; both decoders must agree on what is an error."""
    rng = np.random.default_rng(1234 + block[0])
    nbx, nby = 64, 48
    data = rng.integers(0, 256, size=nbx * nby * 16, dtype=np.uint8)
    # make a share of them void-extent and single-partition headers so those paths are dense too
    blocks = data.reshape(-1, 16)
    blocks[::7, 0] = 0xFC
    blocks[::7, 1] |= 0x01
    blocks[::14, 1] = 0xFD
    blocks[::14, 2:8] = 0xFF
    blocks[1::5, 1] &= 0xE7
    w, h = nbx * block[0], nby * block[1]
    want = ref.decompress(data, w, h, block)
    got, errors = dec(data, block, w, h)
    assert errors > 0
    bad = np.argwhere((want != got).any(axis=2))
    assert len(bad) == 0, "first differing texels (y, x): %s" % bad[:4]


def test_golden_encoder_output_decodes_cleanly(dec):
    """Format check with no reference in the loop: the committed golden blocks decode without error
    blocks to something close to their source image (PSNR floor per case)."""
    gold = os.path.join(ROOT, "tests", "golden")
    manifest = json.load(open(os.path.join(gold, "manifest.json")))
    checked = 0
    for name, case in sorted(manifest.items()):
        if case["image"] not in images.ALL or case["image"] == "hdr":
            continue
        img = images.ALL[case["image"]](*case["size"])
        data = np.load(os.path.join(gold, name + ".npy"))
        out, errors = dec(data, tuple(case["block"]), case["size"][0], case["size"][1])
        assert errors == 0, name
        mse = np.mean((out.astype(np.float64) - img.astype(np.float64)) ** 2)
        psnr = 99.0 if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)
        floor = 11.0 if case["image"] == "random" else 20.0     # white noise is incompressible
        assert psnr > floor, (name, psnr)
        checked += 1
    assert checked >= 6


@pytest.fixture(scope="module")
def dec_volume(dec):
    lib = ctypes.CDLL(LIB_DECODE)
    lib.astc_oracle_decode_volume.restype = ctypes.c_int
    lib.astc_oracle_decode_volume.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 7 + [ctypes.c_void_p]

    def decode(blocks, block, w, h, d, srgb=False):
        blocks = np.ascontiguousarray(blocks, dtype=np.uint8)
        out = np.zeros((d, h, w, 4), dtype=np.uint8)
        errors = lib.astc_oracle_decode_volume(blocks.ctypes.data, block[0], block[1], block[2], w, h, d, int(srgb), out.ctypes.data)
        return out, errors
    return decode


FOOTPRINTS_3D = [(3, 3, 3), (4, 3, 3), (4, 4, 3), (4, 4, 4), (5, 4, 4), (5, 5, 4), (5, 5, 5), (6, 5, 5), (6, 6, 5), (6, 6, 6)]


@pytest.mark.parametrize("block", FOOTPRINTS_3D)
def test_3d_decoder_matches_reference(ref, dec_volume, block):
    """The 3D block modes, simplex weight infill, z-aware partition hash and 3D void extents of the
    oracle decoder against the reference decoder: encoder output and random bit patterns."""
    d, h, w = 2 * block[2] + 1, 2 * block[1] + 2, 3 * block[0] + 1
    for kind, quality in (("noise", 60.0), ("edges", 60.0), ("alpha", 10.0)):
        data = ref.compress(images.volume(kind, d, h, w), block, quality)
        want = ref.decompress(data, w, h, block, depth=d)
        got, errors = dec_volume(data, block, w, h, d)
        assert errors == 0
        assert np.array_equal(want, got), (kind, np.argwhere(want != got)[:4])
    rng = np.random.default_rng(99 + block[0] + block[2])
    junk = rng.integers(0, 256, size=3 * 3 * 4 * 16, dtype=np.uint8)
    junk.reshape(-1, 16)[::3, 0] = 0xFC
    junk.reshape(-1, 16)[::3, 1] |= 0x01
    junk.reshape(-1, 16)[::6, 1:8] = 0xFF
    want = ref.decompress(junk, w, h, block, depth=d)
    got, errors = dec_volume(junk, block, w, h, d)
    assert errors > 0 and np.array_equal(want, got), np.argwhere(want != got)[:4]


def test_golden_volume_output_decodes_cleanly(dec_volume):
    gold = os.path.join(ROOT, "tests", "golden")
    manifest = json.load(open(os.path.join(gold, "manifest.json")))
    checked = 0
    for name, case in sorted(manifest.items()):
        if not case["image"].startswith("volume:"):
            continue
        vol = images.volume(case["image"][7:], *case["size"])
        d, h, w = case["size"]
        out, errors = dec_volume(np.load(os.path.join(gold, name + ".npy")), tuple(case["block"]), w, h, d)
        assert errors == 0, name
        mse = np.mean((out.astype(np.float64) - vol.astype(np.float64)) ** 2)
        psnr = 99.0 if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)
        assert psnr > (9.0 if case["image"] == "volume:noise" else 18.0), (name, psnr)
        checked += 1
    assert checked >= 4
