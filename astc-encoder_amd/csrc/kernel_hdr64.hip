// SPDX-License-Identifier: Apache-2.0
// HDR build of the compression kernel for footprints of at most 64 texels (see kernel_ldr64.hip).
#define ASTC_VARIANT v_hdr64
#define ASTC_ENABLE_HDR 1
#define ASTC_TEXELS_LE_64 1
#define ASTC_KERNEL_NAME astc_compress_blocks_hdr64
#define ASTC_PREPARE_NAME astc_kernel_prepare_hdr64
#define ASTC_LAUNCH_NAME astc_kernel_launch_hdr64
#include "kernel_impl.h"
