// SPDX-License-Identifier: Apache-2.0
// LDR / sRGB build of the compression kernel (HDR endpoint coders compiled out).
#define ASTC_VARIANT v_ldr
#define ASTC_ENABLE_HDR 0
#define ASTC_KERNEL_NAME astc_compress_blocks_ldr
#define ASTC_PREPARE_NAME astc_kernel_prepare_ldr
#define ASTC_LAUNCH_NAME astc_kernel_launch_ldr
#include "kernel_impl.h"
