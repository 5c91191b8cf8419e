// SPDX-License-Identifier: Apache-2.0
// Alpha-average pre-pass kernel (config.a_scale_radius): one wavefront per 32x32 texel tile (16x16 for stacks of slices).
#define ASTC_VARIANT v_alpha
#include "backend.h"
#include "wave_alpha.h"
#include <hip/hip_runtime.h>

namespace astcd {
static_assert(ALPHA_TILE == (int)ALPHA_TILE_ROWS_2D, "the backend's shard halo is counted in tiles of the pre-pass");

/* The padded tile in LDS: one workgroup per tile. */
__global__ void __launch_bounds__(64)
astc_alpha_averages(AlphaJob job, uint32_t tiles_x)
{
	extern __shared__ __attribute__((aligned(16))) float alpha_buf[];
	const uint32_t ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
	alpha_average_tile(job, tx, ty, alpha_buf);
}

/* The padded tile in HBM (it outgrew the 160 KiB of LDS): a fixed number of workgroups, each with its own slice of
 * `scratch`, walk the tiles. */
__global__ void __launch_bounds__(64)
astc_alpha_averages_big(AlphaJob job, uint32_t tiles_x, uint32_t tiles, float* scratch, size_t floats_per_workgroup)
{
	float* buf = scratch + (size_t)blockIdx.x * floats_per_workgroup;
	for (uint32_t t = blockIdx.x; t < tiles; t += gridDim.x)
	{
		const uint32_t ty = t / tiles_x, tx = t - ty * tiles_x;
		alpha_average_tile(job, tx, ty, buf);
	}
}

size_t astc_alpha_scratch_bytes(uint32_t dim_x, uint32_t dim_y, uint32_t dim_z, uint32_t radius, uint32_t* workgroups)
{
	AlphaJob job;
	job.image = nullptr; job.averages = nullptr;
	job.dim_x = dim_x; job.dim_y = dim_y; job.dim_z = dim_z; job.data_type = 0; job.swz_a = 3; job.radius = radius;
	const size_t bytes = alpha_scratch_floats(job) * sizeof(float);
	*workgroups = 0;
	if (bytes <= ALPHA_LDS_LIMIT) return 0;
	const uint32_t tile = (uint32_t)alpha_tile_size(job);
	const uint32_t tiles = ((dim_x + tile - 1) / tile) * ((dim_y + tile - 1) / tile);
	// One workgroup's summed-area table must stay below 1 GiB: the tile arithmetic of alpha_average_tile (plane sizes,
	// row offsets) is 32-bit, good for 2^28 floats; a radius beyond that (about 8000 texels on one slice, a few hundred
	// on a stack of slices) is refused -- the reference would be asking its allocator for the same gigabytes per thread.
	if (bytes > ((size_t)1 << 30)) return (size_t)-1;
	// two workgroups per CU keep a bandwidth-bound gather busy; no more scratch than ~1 GiB in total
	uint32_t wg = tiles < 512u ? tiles : 512u;
	while (wg > 1 && (size_t)wg * bytes > ((size_t)1 << 30)) wg /= 2;
	*workgroups = wg;
	return (size_t)wg * bytes;
}

int astc_alpha_launch(const AlphaLaunch& a)
{
	AlphaJob job;
	job.image = a.d_image;
	job.averages = a.d_averages;
	job.dim_x = a.dim_x; job.dim_y = a.dim_y; job.dim_z = a.dim_z ? a.dim_z : 1u; job.data_type = a.data_type;
	job.swz_a = a.swz_a; job.radius = a.radius;
	const uint32_t tile = (uint32_t)alpha_tile_size(job);
	const uint32_t tiles_x = (a.dim_x + tile - 1) / tile, tiles_y = (a.dim_y + tile - 1) / tile;
	const size_t bytes = alpha_scratch_floats(job) * sizeof(float);
	if (bytes <= ALPHA_LDS_LIMIT)
	{
		if (bytes > 48u * 1024u)
		{
			// large radii: opt in to more than the default dynamic-LDS allowance
			hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(astc_alpha_averages), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
			if (e != hipSuccess) return (int)e;
		}
		hipLaunchKernelGGL(astc_alpha_averages, dim3(tiles_x * tiles_y), dim3(64), bytes, static_cast<hipStream_t>(a.stream), job, tiles_x);
		return (int)hipGetLastError();
	}
	if (!a.d_scratch || a.scratch_workgroups == 0) return (int)hipErrorInvalidValue;
	hipLaunchKernelGGL(astc_alpha_averages_big, dim3(a.scratch_workgroups), dim3(64), 0, static_cast<hipStream_t>(a.stream), job, tiles_x, tiles_x * tiles_y,
	                   a.d_scratch, bytes / sizeof(float));
	return (int)hipGetLastError();
}

} // namespace astcd
