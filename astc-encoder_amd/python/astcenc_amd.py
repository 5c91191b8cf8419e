# SPDX-License-Identifier: Apache-2.0
"""ctypes binding of the astcenc C ABI (include/astcenc.h + include/astcenc_amd.h).

LIB_PRODUCT is astc-encoder_amd/libastcenc_amd.so (HIP kernels, gfx950).  Library(path) binds any shared
object with the astcenc ABI, which is how the tests drive the checker libraries (their paths live in
oracle/oracle_libs.py, not here) through the very same calls: compress with A, compress with B, compare bytes.

Names mirror the C API (ref: Source/astcenc.h); see include/astcenc.h for field meaning.
"""
import ctypes as C
import os

import numpy as np

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
LIB_PRODUCT = os.environ.get("ASTCENC_AMD_LIB", os.path.join(REPO, "astc-encoder_amd", "libastcenc_amd.so"))

# enum astcenc_error
(SUCCESS, ERR_OUT_OF_MEM, ERR_BAD_CPU_FLOAT, ERR_BAD_PARAM, ERR_BAD_BLOCK_SIZE, ERR_BAD_PROFILE,
 ERR_BAD_QUALITY, ERR_BAD_SWIZZLE, ERR_BAD_FLAGS, ERR_BAD_CONTEXT, ERR_NOT_IMPLEMENTED,
 ERR_BAD_DECODE_MODE) = range(12)
# enum astcenc_profile
PRF_LDR_SRGB, PRF_LDR, PRF_HDR_RGB_LDR_A, PRF_HDR = range(4)
# presets
PRE_FASTEST, PRE_FAST, PRE_MEDIUM, PRE_THOROUGH, PRE_VERYTHOROUGH, PRE_EXHAUSTIVE = 0.0, 10.0, 60.0, 98.0, 99.0, 100.0
# enum astcenc_swz
SWZ_R, SWZ_G, SWZ_B, SWZ_A, SWZ_0, SWZ_1, SWZ_Z = range(7)
# enum astcenc_type
TYPE_U8, TYPE_F16, TYPE_F32 = range(3)
# flags
FLG_MAP_NORMAL = 1 << 0
FLG_USE_DECODE_UNORM8 = 1 << 1
FLG_USE_ALPHA_WEIGHT = 1 << 2
FLG_USE_PERCEPTUAL = 1 << 3
FLG_DECOMPRESS_ONLY = 1 << 4
FLG_SELF_DECOMPRESS_ONLY = 1 << 5
FLG_MAP_RGBM = 1 << 6

PROGRESS_CB = C.CFUNCTYPE(None, C.c_float)


class Config(C.Structure):
    """struct astcenc_config (ref: astcenc.h:427)."""
    _fields_ = [
        ("profile", C.c_int), ("flags", C.c_uint),
        ("block_x", C.c_uint), ("block_y", C.c_uint), ("block_z", C.c_uint),
        ("cw_r_weight", C.c_float), ("cw_g_weight", C.c_float), ("cw_b_weight", C.c_float), ("cw_a_weight", C.c_float),
        ("a_scale_radius", C.c_uint), ("rgbm_m_scale", C.c_float),
        ("tune_partition_count_limit", C.c_uint),
        ("tune_2partition_index_limit", C.c_uint), ("tune_3partition_index_limit", C.c_uint), ("tune_4partition_index_limit", C.c_uint),
        ("tune_block_mode_limit", C.c_uint), ("tune_refinement_limit", C.c_uint), ("tune_candidate_limit", C.c_uint),
        ("tune_2partitioning_candidate_limit", C.c_uint), ("tune_3partitioning_candidate_limit", C.c_uint),
        ("tune_4partitioning_candidate_limit", C.c_uint),
        ("tune_db_limit", C.c_float), ("tune_mse_overshoot", C.c_float),
        ("tune_2partition_early_out_limit_factor", C.c_float), ("tune_3partition_early_out_limit_factor", C.c_float),
        ("tune_2plane_early_out_limit_correlation", C.c_float), ("tune_search_mode0_enable", C.c_float),
        ("progress_callback", PROGRESS_CB),
    ]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_ if n != "progress_callback"}


class Image(C.Structure):
    """struct astcenc_image (ref: astcenc.h:613)."""
    _fields_ = [("dim_x", C.c_uint), ("dim_y", C.c_uint), ("dim_z", C.c_uint), ("data_type", C.c_int),
                ("data", C.POINTER(C.c_void_p))]


class Swizzle(C.Structure):
    """struct astcenc_swizzle (ref: astcenc.h:294)."""
    _fields_ = [("r", C.c_int), ("g", C.c_int), ("b", C.c_int), ("a", C.c_int)]


class BlockInfo(C.Structure):
    """struct astcenc_block_info (ref: astcenc.h:637-704)."""
    _fields_ = [("profile", C.c_int), ("block_x", C.c_uint), ("block_y", C.c_uint), ("block_z", C.c_uint), ("texel_count", C.c_uint),
                ("is_error_block", C.c_bool), ("is_constant_block", C.c_bool), ("is_hdr_block", C.c_bool), ("is_dual_plane_block", C.c_bool),
                ("partition_count", C.c_uint), ("partition_index", C.c_uint), ("dual_plane_component", C.c_uint),
                ("color_endpoint_modes", C.c_uint * 4), ("color_level_count", C.c_uint), ("weight_level_count", C.c_uint),
                ("weight_x", C.c_uint), ("weight_y", C.c_uint), ("weight_z", C.c_uint),
                ("color_endpoints", ((C.c_float * 4) * 2) * 4), ("weight_values_plane1", C.c_float * 216),
                ("weight_values_plane2", C.c_float * 216), ("partition_assignment", C.c_uint8 * 216)]


SWZ_RGBA = (SWZ_R, SWZ_G, SWZ_B, SWZ_A)

EXPORTS = ["astcenc_config_init", "astcenc_context_alloc", "astcenc_compress_image", "astcenc_compress_reset",
           "astcenc_compress_cancel", "astcenc_decompress_image", "astcenc_decompress_reset",
           "astcenc_context_free", "astcenc_get_block_info", "astcenc_get_error_string"]
EXPORTS_AMD = ["astcenc_amd_compress_image_device", "astcenc_amd_compress_volume_device", "astcenc_amd_decompress_image_device",
               "astcenc_amd_compare_images_device", "astcenc_amd_backend_name", "astcenc_amd_context_device_count",
               "astcenc_amd_context_set_option", "astcenc_amd_compare_images_hdr_device", "astcenc_amd_context_kernel_name",
               "astcenc_amd_set_log_callback", "astcenc_amd_context_specialize"]
OPT_PER_SLICE_FAST_LOAD = 1


class ErrorSums(C.Structure):
    """struct astcenc_amd_error_sums (include/astcenc_amd.h)."""
    _fields_ = [("squared_error", C.c_double * 4), ("alpha_scaled_squared_error", C.c_double * 4),
                ("rgb_peak", C.c_double), ("texels", C.c_double)]

    def psnr(self, channels=4, alpha_scaled=False):
        """PSNR over the first `channels` channels as the reference CLI reports it (999 dB when identical)."""
        src = self.alpha_scaled_squared_error if alpha_scaled else self.squared_error
        num = sum(src[k] for k in range(channels))
        return 999.0 if num == 0 else 10.0 * float(np.log10(self.texels * channels / num))


class HdrErrorSums(C.Structure):
    """struct astcenc_amd_hdr_error_sums (include/astcenc_amd.h)."""
    _fields_ = [("log2_squared_error", C.c_double * 4), ("mpsnr_squared_error", C.c_double * 4),
                ("fstop_lo", C.c_int), ("fstop_hi", C.c_int)]

    def mpsnr(self, texels):
        """mPSNR (RGB) as the reference CLI prints it (astcenccli_error_metrics.cpp:389-397)."""
        num = sum(self.mpsnr_squared_error[k] for k in range(3))
        stops = self.fstop_hi - self.fstop_lo + 1
        return 999.0 if num == 0 else 10.0 * float(np.log10(texels * 3.0 * stops * 255.0 * 255.0 / num))

    def log_rmse(self, texels):
        """LogRMSE (RGB) (astcenccli_error_metrics.cpp:402-403)."""
        return float(np.sqrt(sum(self.log2_squared_error[k] for k in range(3)) / texels))


class AstcError(RuntimeError):
    def __init__(self, code, where):
        super().__init__("%s failed with astcenc_error %d" % (where, code))
        self.code = code


class Library:
    """One loaded astcenc-ABI shared object."""

    def __init__(self, path):
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.path = path
        self.lib = C.CDLL(path)
        L = self.lib
        L.astcenc_config_init.argtypes = [C.c_int, C.c_uint, C.c_uint, C.c_uint, C.c_float, C.c_uint, C.POINTER(Config)]
        L.astcenc_config_init.restype = C.c_int
        L.astcenc_context_alloc.argtypes = [C.POINTER(Config), C.c_uint, C.POINTER(C.c_void_p), C.c_void_p]
        L.astcenc_context_alloc.restype = C.c_int
        L.astcenc_compress_image.argtypes = [C.c_void_p, C.POINTER(Image), C.POINTER(Swizzle), C.c_void_p, C.c_size_t, C.c_uint]
        L.astcenc_compress_image.restype = C.c_int
        L.astcenc_compress_reset.argtypes = [C.c_void_p]
        L.astcenc_compress_reset.restype = C.c_int
        L.astcenc_compress_cancel.argtypes = [C.c_void_p]
        L.astcenc_compress_cancel.restype = C.c_int
        L.astcenc_decompress_image.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(Image), C.POINTER(Swizzle), C.c_uint]
        L.astcenc_decompress_image.restype = C.c_int
        L.astcenc_decompress_reset.argtypes = [C.c_void_p]
        L.astcenc_decompress_reset.restype = C.c_int
        L.astcenc_context_free.argtypes = [C.c_void_p]
        L.astcenc_context_free.restype = None
        L.astcenc_get_block_info.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(BlockInfo)]
        L.astcenc_get_block_info.restype = C.c_int
        L.astcenc_get_error_string.argtypes = [C.c_int]
        L.astcenc_get_error_string.restype = C.c_char_p
        self.has_amd = hasattr(L, "astcenc_amd_compress_image_device")
        if self.has_amd:
            L.astcenc_amd_compress_image_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_int,
                                                            C.POINTER(Swizzle), C.c_void_p, C.c_size_t, C.c_void_p,
                                                            C.POINTER(C.c_float)]
            L.astcenc_amd_compress_image_device.restype = C.c_int
            L.astcenc_amd_backend_name.restype = C.c_char_p
        if hasattr(L, "astcenc_amd_compare_images_device"):
            L.astcenc_amd_decompress_image_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_uint, C.c_uint, C.c_uint,
                                                              C.c_int, C.POINTER(Swizzle), C.c_void_p]
            L.astcenc_amd_decompress_image_device.restype = C.c_int
            L.astcenc_amd_compare_images_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_uint, C.c_uint, C.c_uint,
                                                            C.c_void_p, C.POINTER(ErrorSums)]
            L.astcenc_amd_compare_images_device.restype = C.c_int
        if hasattr(L, "astcenc_amd_compare_images_hdr_device"):
            L.astcenc_amd_compare_images_hdr_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_uint, C.c_uint, C.c_uint,
                                                                C.c_int, C.c_int, C.c_void_p, C.POINTER(ErrorSums), C.POINTER(HdrErrorSums)]
            L.astcenc_amd_compare_images_hdr_device.restype = C.c_int
        if hasattr(L, "astcenc_amd_context_device_count"):
            L.astcenc_amd_context_device_count.argtypes = [C.c_void_p]
            L.astcenc_amd_context_device_count.restype = C.c_int
            L.astcenc_amd_context_set_option.argtypes = [C.c_void_p, C.c_int, C.c_int]
            L.astcenc_amd_context_set_option.restype = C.c_int
        if hasattr(L, "astcenc_amd_context_kernel_name"):
            L.astcenc_amd_context_kernel_name.argtypes = [C.c_void_p]
            L.astcenc_amd_context_kernel_name.restype = C.c_char_p
        if hasattr(L, "astcenc_amd_context_specialize"):
            L.astcenc_amd_context_specialize.argtypes = [C.c_void_p]
            L.astcenc_amd_context_specialize.restype = C.c_int
        if hasattr(L, "astcenc_amd_compress_volume_device"):
            L.astcenc_amd_compress_volume_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_uint, C.c_int,
                                                             C.POINTER(Swizzle), C.c_void_p, C.c_size_t, C.c_void_p,
                                                             C.POINTER(C.c_float)]
            L.astcenc_amd_compress_volume_device.restype = C.c_int

    # -- thin wrappers returning error codes, as the C API does --
    def config_init(self, profile, bx, by, bz, quality, flags):
        cfg = Config()
        err = self.lib.astcenc_config_init(profile, bx, by, bz, quality, flags, C.byref(cfg))
        return err, cfg

    def context_alloc(self, cfg, thread_count=1, parent=None):
        ctx = C.c_void_p()
        err = self.lib.astcenc_context_alloc(C.byref(cfg) if cfg is not None else None, thread_count, C.byref(ctx), parent)
        return err, ctx

    def context_free(self, ctx):
        self.lib.astcenc_context_free(ctx)

    def error_string(self, code):
        s = self.lib.astcenc_get_error_string(code)
        return s.decode() if s else None

    def backend_name(self):
        return self.lib.astcenc_amd_backend_name().decode() if self.has_amd else "reference"

    def compress_raw(self, ctx, pixels, out, swizzle=SWZ_RGBA, thread_index=0, data_len=None):
        """pixels: contiguous array [H, W, 4] (or [D, H, W, 4] for a volume / array image) of
        uint8 / float16 / float32; out: uint8 array."""
        d = pixels.shape[0] if pixels.ndim == 4 else 1
        h, w = pixels.shape[-3], pixels.shape[-2]
        dtype = {np.dtype(np.uint8): TYPE_U8, np.dtype(np.float16): TYPE_F16, np.dtype(np.float32): TYPE_F32}[pixels.dtype]
        slice_bytes = h * w * 4 * pixels.dtype.itemsize
        slices = (C.c_void_p * d)(*[pixels.ctypes.data + z * slice_bytes for z in range(d)])
        img = Image(w, h, d, dtype, slices)
        swz = Swizzle(*swizzle)
        return self.lib.astcenc_compress_image(ctx, C.byref(img), C.byref(swz), out.ctypes.data,
                                               out.nbytes if data_len is None else data_len, thread_index)

    def compress(self, pixels, block=(6, 6), quality=PRE_MEDIUM, profile=PRF_LDR, flags=0, swizzle=SWZ_RGBA, tweak=None, options=None,
                 specialize=False):
        """Convenience: config_init -> context_alloc -> compress_image -> free. Returns uint8 [blocks*16].
        block is (x, y) or (x, y, z); pixels is [H, W, 4] or [D, H, W, 4]; options: {OPT_*: value} for
        astcenc_amd_context_set_option (product library only); specialize: wait for the context's specialised kernel
        build first (astcenc_amd_context_specialize; "try": go on with the generic build if there is none) -- self.last_kernel then
        names the build that ran."""
        bz = block[2] if len(block) > 2 else 1
        err, cfg = self.config_init(profile, block[0], block[1], bz, quality, flags)
        if err:
            raise AstcError(err, "astcenc_config_init")
        if tweak:
            tweak(cfg)
        err, ctx = self.context_alloc(cfg, 1)
        if err:
            raise AstcError(err, "astcenc_context_alloc")
        try:
            for opt, value in (options or {}).items():
                err = self.lib.astcenc_amd_context_set_option(ctx, opt, value)
                if err:
                    raise AstcError(err, "astcenc_amd_context_set_option")
            if specialize:
                err = self.lib.astcenc_amd_context_specialize(ctx)
                if err and specialize != "try":          # ("try": a refused build leaves the context on the generic kernel)
                    raise AstcError(err, "astcenc_amd_context_specialize")
            if hasattr(self.lib, "astcenc_amd_context_kernel_name"):
                self.last_kernel = self.lib.astcenc_amd_context_kernel_name(ctx).decode()
            d = pixels.shape[0] if pixels.ndim == 4 else 1
            h, w = pixels.shape[-3], pixels.shape[-2]
            bx, by = (w + block[0] - 1) // block[0], (h + block[1] - 1) // block[1]
            out = np.zeros(bx * by * ((d + bz - 1) // bz) * 16, dtype=np.uint8)
            err = self.compress_raw(ctx, np.ascontiguousarray(pixels), out, swizzle)
            if err:
                raise AstcError(err, "astcenc_compress_image")
            return out
        finally:
            self.context_free(ctx)

    def decompress(self, data, width, height, block=(6, 6), profile=PRF_LDR, out_type=np.uint8, depth=None):
        """Decode blocks back to [H, W, 4] ([D, H, W, 4] when depth is given) through
        astcenc_decompress_image of whichever library this is."""
        bz = block[2] if len(block) > 2 else 1
        err, cfg = self.config_init(profile, block[0], block[1], bz, PRE_MEDIUM, FLG_DECOMPRESS_ONLY)
        if err:
            raise AstcError(err, "astcenc_config_init")
        err, ctx = self.context_alloc(cfg, 1)
        if err:
            raise AstcError(err, "astcenc_context_alloc")
        try:
            d = 1 if depth is None else depth
            out = np.zeros((height, width, 4) if depth is None else (depth, height, width, 4), dtype=out_type)
            dtype = {np.dtype(np.uint8): TYPE_U8, np.dtype(np.float16): TYPE_F16, np.dtype(np.float32): TYPE_F32}[out.dtype]
            slice_bytes = height * width * 4 * out.dtype.itemsize
            slices = (C.c_void_p * d)(*[out.ctypes.data + z * slice_bytes for z in range(d)])
            img = Image(width, height, d, dtype, slices)
            swz = Swizzle(*SWZ_RGBA)
            data = np.ascontiguousarray(data, dtype=np.uint8)
            err = self.lib.astcenc_decompress_image(ctx, data.ctypes.data, data.nbytes, C.byref(img), C.byref(swz), 0)
            if err:
                raise AstcError(err, "astcenc_decompress_image")
            return out
        finally:
            self.context_free(ctx)


def synthetic_image(width, height, seed=0x9E3779B1):
    """Deterministic integer-only RGBA8 test image: smooth ramps, a 32x32 checker of
    hard edges on R, and per-channel hash noise.  No libm, so every host produces identical bytes."""
    y, x = np.meshgrid(np.arange(height, dtype=np.int64), np.arange(width, dtype=np.int64), indexing="ij")

    def tri(v):
        m = v & 511
        return np.where(m < 256, m, 511 - m)

    r = (3 * tri(x + 2 * y) + tri((3 * x - y) >> 1)) >> 2
    checker = ((x >> 5) + (y >> 5)) & 1
    r = np.where(checker == 1, 255 - r, r)
    g = (3 * tri(2 * x - y + 128) + tri((x + 3 * y) >> 2)) >> 2
    b = 255 - ((tri(x + y) + tri((x - y) >> 1)) >> 1)
    a = 192 + (tri((x >> 1) + (y >> 2)) >> 2)

    def noise(c, amp):
        h = (x * 0x85EBCA6B + y * 0xC2B2AE35 + c * 0x27D4EB2F + seed) & 0xFFFFFFFF
        h ^= h >> 15
        h = (h * 0x2C1B3C6D) & 0xFFFFFFFF
        h ^= h >> 12
        h = (h * 0x297A2D39) & 0xFFFFFFFF
        h ^= h >> 15
        return (h % (2 * amp + 1)) - amp

    out = np.stack([r + noise(0, 10), g + noise(1, 10), b + noise(2, 10), a + noise(3, 3)], axis=-1)
    return np.clip(out, 0, 255).astype(np.uint8)


def synthetic_hdr_image(width, height, seed=0x9E3779B1):
    """HDR companion of synthetic_image: the LDR generator / 255 with RGB scaled by
    2^(-2..5) in 8x8 patches (exponent ((tri(x >> 3) + tri(y >> 3)) >> 6) - 2), alpha kept in 0..1, stored as
    RGBA16F (round to nearest even).  Power-of-two scaling of exact quotients: identical bytes on every host."""
    base = synthetic_image(width, height, seed).astype(np.float32) / np.float32(255.0)
    y, x = np.meshgrid(np.arange(height, dtype=np.int64), np.arange(width, dtype=np.int64), indexing="ij")

    def tri(v):
        m = v & 511
        return np.where(m < 256, m, 511 - m)

    expo = ((tri(x >> 3) + tri(y >> 3)) >> 6) - 2
    scale = np.ldexp(np.float32(1.0), expo.astype(np.int32)).astype(np.float32)
    base[..., :3] *= scale[..., None]
    return base.astype(np.float16)


def psnr_rgba8(a, b):
    """10*log10(samples / sum((a-b)^2)) on values/255, double accumulation (ref: astcenccli_error_metrics.cpp:240-346)."""
    d = (a.astype(np.float64) - b.astype(np.float64)) / 255.0
    s = float((d * d).sum())
    return float("inf") if s == 0 else 10.0 * np.log10(a.size / s)


# ---- multi-GPU sharding: contiguous block rows per rank, no data-path collective ----

def block_row_shard(dim_y, block_y, rank, world):
    """Rows of blocks [row0, row1) owned by `rank`, and the texel rows [y0, y1) that feed them.

    Blocks are independent (ref: astcenc_entry.cpp:1009-1038: compress_block reads only its own
    texels and writes its own 16 bytes), so a shard is just a sub-image whose top edge sits on a block
    boundary; only the last shard can contain a partial (edge-clamped) block row.  Ranks beyond the
    number of block rows get an empty shard.
    """
    blocks_y = (dim_y + block_y - 1) // block_y
    per = (blocks_y + world - 1) // world
    row0 = min(rank * per, blocks_y)
    row1 = min(row0 + per, blocks_y)
    return row0, row1, min(row0 * block_y, dim_y), min(row1 * block_y, dim_y)


def compress_shard(lib, ctx, pixels, block, rank, world, out=None):
    """Compress this rank's block rows of `pixels` ([H, W, 4]) with context `ctx`.

    Returns (byte offset into the whole image's block stream, uint8 blocks of the shard).  When
    `out` (the whole image's output array) is given, the shard is also written in place.
    """
    h, w = pixels.shape[0], pixels.shape[1]
    row0, row1, y0, y1 = block_row_shard(h, block[1], rank, world)
    blocks_x = (w + block[0] - 1) // block[0]
    offset = row0 * blocks_x * 16
    part = np.zeros((row1 - row0) * blocks_x * 16, dtype=np.uint8)
    if row1 > row0:
        err = lib.compress_raw(ctx, np.ascontiguousarray(pixels[y0:y1]), part)
        if err:
            raise AstcError(err, "astcenc_compress_image (shard %d/%d)" % (rank, world))
    if out is not None:
        out[offset: offset + part.size] = part
    return offset, part


# ---- .astc container (ref: Docs/FileFormat.md, astcenccli_image_load_store.cpp: 16-byte header) ----

ASTC_MAGIC = bytes([0x13, 0xAB, 0xA1, 0x5C])


def write_astc(path, blocks, width, height, block, depth=1):
    """Write a block stream as an .astc file: magic, block dims (3 x u8), image dims (3 x 24-bit LE), data."""
    def u24(v):
        return bytes([v & 0xFF, (v >> 8) & 0xFF, (v >> 16) & 0xFF])
    bz = block[2] if len(block) > 2 else 1
    header = ASTC_MAGIC + bytes([block[0], block[1], bz]) + u24(width) + u24(height) + u24(depth)
    data = np.ascontiguousarray(blocks, dtype=np.uint8).tobytes()
    with open(path, "wb") as f:
        f.write(header + data)


def read_astc(path):
    """-> (blocks uint8[n*16], width, height, depth, (bx, by, bz)); raises ValueError on a malformed file
    (bad magic, zero dimensions, truncated payload), like the reference loader."""
    raw = open(path, "rb").read()
    if len(raw) < 16 or raw[:4] != ASTC_MAGIC:
        raise ValueError("not an .astc file")
    bx, by, bz = raw[4], raw[5], raw[6]
    dims = [raw[7 + 3 * i] | (raw[8 + 3 * i] << 8) | (raw[9 + 3 * i] << 16) for i in range(3)]
    if 0 in (bx, by, bz) or 0 in dims:
        raise ValueError("zero dimension in .astc header")
    n = ((dims[0] + bx - 1) // bx) * ((dims[1] + by - 1) // by) * ((dims[2] + bz - 1) // bz)
    if len(raw) - 16 < n * 16:
        raise ValueError("truncated .astc payload")
    return np.frombuffer(raw, dtype=np.uint8, count=n * 16, offset=16).copy(), dims[0], dims[1], dims[2], (bx, by, bz)


# ---- KTX 1.1 container for compressed data (ref: store_ktx_compressed_image / load_ktx_compressed_image,
# astcenccli_image_load_store.cpp:1294-1437; GL enums :725-775) ----

KTX_MAGIC = bytes([0xAB, 0x4B, 0x54, 0x58, 0x20, 0x31, 0x31, 0xBB, 0x0D, 0x0A, 0x1A, 0x0A])
GL_RGBA = 0x1908
_KTX_2D = [(4, 4), (5, 4), (5, 5), (6, 5), (6, 6), (8, 5), (8, 6), (8, 8), (10, 5), (10, 6), (10, 8), (10, 10), (12, 10), (12, 12)]
_KTX_3D = [(3, 3, 3), (4, 3, 3), (4, 4, 3), (4, 4, 4), (5, 4, 4), (5, 5, 4), (5, 5, 5), (6, 5, 5), (6, 6, 5), (6, 6, 6)]


def ktx_gl_format(block, srgb=False):
    """glInternalFormat of an ASTC footprint: COMPRESSED_RGBA_ASTC_* / COMPRESSED_SRGB8_ALPHA8_ASTC_* (+ _OES for 3D)."""
    bz = block[2] if len(block) > 2 else 1
    if bz <= 1:
        return (0x93D0 if srgb else 0x93B0) + _KTX_2D.index((block[0], block[1]))
    return (0x93E0 if srgb else 0x93C0) + _KTX_3D.index((block[0], block[1], bz))


def write_ktx(path, blocks, width, height, block, depth=1, srgb=False):
    """Write a block stream as a single-level KTX 1.1 file, little endian, no key/value data."""
    import struct
    data = np.ascontiguousarray(blocks, dtype=np.uint8).tobytes()
    header = KTX_MAGIC + struct.pack("<13I", 0x04030201, 0, 1, 0, ktx_gl_format(block, srgb), GL_RGBA,
                                     width, height, 0 if depth == 1 else depth, 0, 1, 1, 0)
    with open(path, "wb") as f:
        f.write(header + struct.pack("<I", len(data)) + data)


def read_ktx(path):
    """-> (blocks uint8[], width, height, depth, (bx, by, bz), is_srgb); either byte order; raises ValueError
    on anything that is not a compressed ASTC KTX file, like the reference loader."""
    import struct
    raw = open(path, "rb").read()
    if len(raw) < 68 or raw[:12] != KTX_MAGIC:
        raise ValueError("not a KTX file")
    endian = struct.unpack_from("<I", raw, 12)[0]
    if endian not in (0x04030201, 0x01020304):
        raise ValueError("corrupt KTX header")
    e = "<" if endian == 0x04030201 else ">"
    (gl_type, type_size, gl_format, internal, base, w, h, d, _arrays, _faces, _mips, kv) = struct.unpack_from(e + "12I", raw, 16)
    if gl_type != 0 or gl_format != 0 or type_size != 1 or base != GL_RGBA:
        raise ValueError("unsupported KTX format")
    for first, table, srgb in ((0x93B0, _KTX_2D, False), (0x93D0, _KTX_2D, True), (0x93C0, _KTX_3D, False), (0x93E0, _KTX_3D, True)):
        if first <= internal < first + len(table):
            blk = table[internal - first]
            break
    else:
        raise ValueError("unsupported KTX format")
    at = 64 + kv
    if len(raw) < at + 4:
        raise ValueError("truncated KTX file")
    n = struct.unpack_from(e + "I", raw, at)[0]
    if len(raw) < at + 4 + n:
        raise ValueError("truncated KTX file")
    block = (blk[0], blk[1], blk[2] if len(blk) > 2 else 1)
    return np.frombuffer(raw, dtype=np.uint8, count=n, offset=at + 4).copy(), w, h, d if d else 1, block, srgb
