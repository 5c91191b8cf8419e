# SPDX-License-Identifier: Apache-2.0
"""Seeded test images covering the content classes the reference's search treats differently."""
import numpy as np

import astcenc_amd as A


def noisy(w, h, seed=0x9E3779B1):
    return A.synthetic_image(w, h, seed)


def flat_regions(w, h):
    img = A.synthetic_image(w, h, 1)
    img[: h // 2, : w // 2] = (17, 99, 201, 255)      # constant -> void-extent blocks
    img[h // 2:, : w // 2, 3] = 255                    # opaque -> 3-component paths
    return img


def grayscale(w, h):
    img = A.synthetic_image(w, h, 2)
    img[..., 1] = img[..., 0]
    img[..., 2] = img[..., 0]
    img[: h // 2, :, 3] = 255                          # luminance vs luminance+alpha blocks
    return img


def smooth(w, h):
    y, x = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    img = np.stack([(x * 255) // max(w - 1, 1), (y * 255) // max(h - 1, 1), ((x + y) * 255) // max(w + h - 2, 1),
                    np.full_like(x, 255)], axis=-1)
    return img.astype(np.uint8)


def random_u8(w, h, seed=7):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)


def two_colour(w, h, seed=3):
    """Hard two-region blocks: exercises 2+ partition encodings."""
    rng = np.random.default_rng(seed)
    y, x = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    mask = ((x * 3 + y * 5) // 7) % 2
    a = np.array([220, 40, 30, 255]); b = np.array([20, 60, 230, 128])
    img = np.where(mask[..., None] == 1, a, b) + rng.integers(-6, 7, size=(h, w, 4))
    return np.clip(img, 0, 255).astype(np.uint8)


def volume(kind, d, h, w, seed=11):
    """[D, H, W, 4] RGBA8 volumes for the 3D footprints: random noise, noisy ramps along all three axes,
    hard two-material cells, and a transparent/opaque alpha lattice (2-plane territory)."""
    rng = np.random.default_rng(seed)
    z, y, x = np.meshgrid(np.arange(d), np.arange(h), np.arange(w), indexing="ij")
    if kind == "noise":
        return rng.integers(0, 256, (d, h, w, 4), dtype=np.uint8)
    if kind == "grad":
        v = np.stack([x * 9 + z * 13, y * 11 + z * 5, (x + y + z) * 7, 255 - (x * 3 + y * 2 + z * 17)], -1) % 256
        return (v + rng.integers(-6, 7, v.shape)).clip(0, 255).astype(np.uint8)
    if kind == "edges":
        m = (x // 3 + y // 2 + z // 2) % 2
        v = np.stack([m * 200 + 20, (1 - m) * 180 + x * 3, z * 30 + m * 40, 255 * np.ones_like(x)], -1)
        return (v + rng.integers(-3, 4, v.shape)).clip(0, 255).astype(np.uint8)
    if kind == "alpha":
        m = (x + z) // 4 % 2
        v = np.stack([x * 8, y * 8, z * 20, m * 255], -1)
        return (v + rng.integers(-2, 3, v.shape)).clip(0, 255).astype(np.uint8)
    if kind == "flat":
        v = np.empty((d, h, w, 4), dtype=np.uint8)
        v[...] = (90, 14, 200, 255)
        v[:, :, w // 2:] = rng.integers(0, 256, (d, h, w - w // 2, 4), dtype=np.uint8)
        return v
    raise KeyError(kind)


ALL = {"noisy": noisy, "flat": flat_regions, "gray": grayscale, "smooth": smooth, "random": random_u8, "two_colour": two_colour}


def mismatches(a, b):
    a = a.reshape(-1, 16); b = b.reshape(-1, 16)
    return np.where((a != b).any(axis=1))[0]


def hdr_f16(w, h, seed=0x9E3779B1):
    """HDR test image: the LDR generator scaled by 2^(-2..5) in 8x8 patches,
    alpha kept in 0..1, stored as RGBA16F."""
    return A.synthetic_hdr_image(w, h, seed)


def hdr_variants(w, h):
    """HDR images that steer the encoder into the different HDR endpoint formats."""
    base = hdr_f16(w, h).astype(np.float32)
    opaque = base.copy(); opaque[..., 3] = 1.0                       # HDR RGB / RGB+offset
    gray = base.copy(); gray[..., 1] = gray[..., 0]; gray[..., 2] = gray[..., 0]; gray[..., 3] = 1.0   # HDR luminance
    dim = base.copy(); dim[..., :3] *= 0.02                            # small magnitudes / small-range modes
    bright = base.copy(); bright[..., :3] *= 400.0                     # near the fp16 limit
    return {"rgba": base, "opaque": opaque, "gray": gray, "dim": dim, "bright": bright}
