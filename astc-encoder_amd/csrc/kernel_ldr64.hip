// SPDX-License-Identifier: Apache-2.0
// LDR / sRGB build of the compression kernel for footprints of at most 64 texels (every 2D footprint up to 8x8, 3D up to
// 4x4x4): the loops over a block's texels make exactly one trip, and the compiler is told so (WV_FOR_T, wave.h).
#define ASTC_VARIANT v_ldr64
#define ASTC_ENABLE_HDR 0
#define ASTC_TEXELS_LE_64 1
#define ASTC_KERNEL_NAME astc_compress_blocks_ldr64
#define ASTC_PREPARE_NAME astc_kernel_prepare_ldr64
#define ASTC_LAUNCH_NAME astc_kernel_launch_ldr64
#include "kernel_impl.h"
