// SPDX-License-Identifier: Apache-2.0
// Alpha-average pre-pass kernel (config.a_scale_radius): one wavefront per 32x32 texel tile.
#define ASTC_VARIANT v_alpha
#include "backend.h"
#include "wave_alpha.h"
#include <hip/hip_runtime.h>

namespace astcd {

__global__ void __launch_bounds__(64)
astc_alpha_averages(AlphaJob job, uint32_t tiles_x)
{
	extern __shared__ __attribute__((aligned(16))) float alpha_buf[];
	const uint32_t ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
	alpha_average_tile(job, tx, ty, alpha_buf);
}

int astc_alpha_launch(const AlphaLaunch& a)
{
	AlphaJob job;
	job.image = a.d_image;
	job.averages = a.d_averages;
	job.dim_x = a.dim_x; job.dim_y = a.dim_y; job.data_type = a.data_type;
	job.swz_a = a.swz_a; job.radius = a.radius;
	const uint32_t tiles_x = (a.dim_x + ALPHA_TILE - 1) / ALPHA_TILE, tiles_y = (a.dim_y + ALPHA_TILE - 1) / ALPHA_TILE;
	const uint32_t pad = ALPHA_TILE + 2 * a.radius + 1;
	const uint32_t lds_bytes = pad * pad * (uint32_t)sizeof(float);
	if (lds_bytes > 48u * 1024u)
	{
		// large radii: opt in to more than the default dynamic-LDS allowance
		hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(astc_alpha_averages), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
		if (e != hipSuccess) return (int)e;
	}
	hipLaunchKernelGGL(astc_alpha_averages, dim3(tiles_x * tiles_y), dim3(64), pad * pad * sizeof(float), static_cast<hipStream_t>(a.stream), job, tiles_x);
	return (int)hipGetLastError();
}

} // namespace astcd
