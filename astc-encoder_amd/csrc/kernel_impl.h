// SPDX-License-Identifier: Apache-2.0
// The compression kernel.  Included by kernel_ldr.hip and kernel_hdr.hip, which set
//   ASTC_VARIANT     inline-namespace tag of this build of the wave_*.h code
//   ASTC_ENABLE_HDR  0: LDR/sRGB profiles only (HDR endpoint coders compiled out), 1: everything
//   ASTC_KERNEL_NAME / ASTC_PREPARE_NAME / ASTC_LAUNCH_NAME
// One 64-lane wavefront (= one workgroup) compresses one ASTC block; its working set is a
// dynamic-LDS region laid out by make_lds_layout().
#include "backend.h"
#include "kernel_device.h"
#include <hip/hip_runtime.h>
#include <cstring>
#include <cstdlib>

namespace astcd {

int ASTC_PREPARE_NAME(const TableRoot& root, const DeviceConfig& cfg, uint32_t* lds_bytes, void* layout_out, uint32_t* layout_bytes)
{
	LdsLayout L;
	make_lds_layout(root, cfg, L);
#if ASTC_FIXED
	// This build is compiled for ONE context (wave_ctx.h): the live context must be that one, record for record.  (The
	// instrumentation builds' stage selector is read from the live record, DUP_STAGE_ID.)
	{
		DeviceConfig live = cfg;
		live.debug_dup_stage = kFixedConfig.debug_dup_stage;
		if (memcmp(&L, &kFixedLayout, sizeof(L)) != 0 || memcmp(&live, &kFixedConfig, sizeof(live)) != 0 || memcmp(&root, &kFixedRoot, sizeof(root)) != 0)
			return ASTC_PREPARE_NOT_THIS_CONTEXT;
	}
#endif
	*lds_bytes = L.total;
	static_assert(sizeof(LdsLayout) <= CTX_LAYOUT_BACK - CTX_CONFIG_BACK, "layout record grew past the space the backend reserves");
	memcpy(layout_out, &L, sizeof(L));
	*layout_bytes = (uint32_t)sizeof(L);
	return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(ASTC_KERNEL_NAME),
	                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)L.total);
}

int ASTC_LAUNCH_NAME(const KernelLaunch& k)
{
#if defined(ASTC_LDS_PAD_ENV)
	// (occupancy experiment builds only: extra dynamic LDS per workgroup from the environment)
	static const unsigned pad = getenv("ASTC_LDS_PAD") ? (unsigned)atoi(getenv("ASTC_LDS_PAD")) : 0u;
	if (pad) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ASTC_KERNEL_NAME), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(k.lds_bytes + pad));
	hipLaunchKernelGGL(ASTC_KERNEL_NAME, dim3(k.count), dim3(64), k.lds_bytes + pad, static_cast<hipStream_t>(k.stream),
	                   k.d_tab, k.img, k.d_out, k.first, k.count, k.d_prof);
	return (int)hipGetLastError();
#endif
	hipLaunchKernelGGL(ASTC_KERNEL_NAME, dim3(k.count), dim3(64), k.lds_bytes, static_cast<hipStream_t>(k.stream),
	                   k.d_tab, k.img, k.d_out, k.first, k.count, k.d_prof);
	return (int)hipGetLastError();
}

} // namespace astcd
