// SPDX-License-Identifier: Apache-2.0
// Fixed-context build of the compression kernel for 8x8 -thorough, LDR profile (BASELINE.json configs[2]); see
// kernel_ldr_6x6m.hip.
#define ASTC_VARIANT v_ldr_8x8t
#define ASTC_ENABLE_HDR 0
#define ASTC_TEXELS_LE_64 1
#define ASTC_FIXED_CONTEXT 1
#define ASTC_FIXED_ldr_8x8_thorough 1
#define ASTC_KERNEL_NAME astc_compress_blocks_ldr_8x8t
#define ASTC_PREPARE_NAME astc_kernel_prepare_ldr_8x8t
#define ASTC_LAUNCH_NAME astc_kernel_launch_ldr_8x8t
#include "kernel_impl.h"
