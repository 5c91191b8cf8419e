// SPDX-License-Identifier: Apache-2.0
// Fixed-context build of the compression kernel for 6x6 -medium, LDR profile (BASELINE.json configs[1], the headline
// workload): LdsLayout, DeviceConfig and TableRoot are compile-time constants (wave_ctx.h, fixed_contexts.inc).  The
// backend uses it when the live context matches those records byte for byte, kernel_ldr64.hip otherwise.
#define ASTC_VARIANT v_ldr_6x6m
#define ASTC_ENABLE_HDR 0
#define ASTC_TEXELS_LE_64 1
#define ASTC_FIXED_CONTEXT 1
#define ASTC_FIXED_ldr_6x6_medium 1
#define ASTC_KERNEL_NAME astc_compress_blocks_ldr_6x6m
#define ASTC_PREPARE_NAME astc_kernel_prepare_ldr_6x6m
#define ASTC_LAUNCH_NAME astc_kernel_launch_ldr_6x6m
#include "kernel_impl.h"
