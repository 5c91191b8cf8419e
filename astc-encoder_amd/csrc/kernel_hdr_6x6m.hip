// SPDX-License-Identifier: Apache-2.0
// Fixed-context build of the compression kernel for 6x6 -medium, HDR profile (BASELINE.json configs[3]); see
// kernel_ldr_6x6m.hip.
#define ASTC_VARIANT v_hdr_6x6m
#define ASTC_ENABLE_HDR 1
#define ASTC_TEXELS_LE_64 1
#define ASTC_FIXED_CONTEXT 1
#define ASTC_FIXED_hdr_6x6_medium 1
#define ASTC_KERNEL_NAME astc_compress_blocks_hdr_6x6m
#define ASTC_PREPARE_NAME astc_kernel_prepare_hdr_6x6m
#define ASTC_LAUNCH_NAME astc_kernel_launch_hdr_6x6m
#include "kernel_impl.h"
