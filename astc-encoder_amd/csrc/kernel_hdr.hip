// SPDX-License-Identifier: Apache-2.0
// HDR build of the compression kernel (HDR_RGB_LDR_A and HDR profiles).
#define ASTC_VARIANT v_hdr
#define ASTC_ENABLE_HDR 1
#define ASTC_KERNEL_NAME astc_compress_blocks_hdr
#define ASTC_PREPARE_NAME astc_kernel_prepare_hdr
#define ASTC_LAUNCH_NAME astc_kernel_launch_hdr
#include "kernel_impl.h"
