// SPDX-License-Identifier: Apache-2.0
// Image comparison kernel: squared-error sums of two images resident in HBM, for PSNR without a host
// round trip (ref: compute_error_metrics, Source/astcenccli_error_metrics.cpp:110).  Streaming and
// HBM-bound: every lane walks texels with a grid stride (coalesced 4 / 8 / 16-byte texel loads), keeps
// fp64 partial sums, the wave folds them with DPP shuffles and lane 0 adds them to the totals.
#define ASTC_VARIANT v_metrics
#include "backend.h"
#include "wave_metrics.h"
#include <hip/hip_runtime.h>

namespace astcd {

__global__ void __launch_bounds__(256)
astc_compare_images(const void* __restrict__ a, uint32_t type_a, const void* __restrict__ b, uint32_t type_b,
                    size_t texels, double* __restrict__ sums)
{
	double acc[8];
	for (int k = 0; k < 8; k++) acc[k] = 0.0;
	float peak = 0.0f;
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < texels; t += stride)
	{
		float e[8];
		float m = metric_texel_terms(a, type_a, b, type_b, t, e);
		peak = m > peak ? m : peak;
		for (int k = 0; k < 8; k++) acc[k] += (double)e[k];
	}
	for (int off = 32; off > 0; off >>= 1)
	{
		for (int k = 0; k < 8; k++) acc[k] += __shfl_down(acc[k], off);
		float o = __shfl_down(peak, off);
		peak = o > peak ? o : peak;
	}
	if ((threadIdx.x & 63) == 0)
	{
		for (int k = 0; k < 8; k++) atomicAdd(&sums[k], acc[k]);
		// the peak is never negative, so its bit pattern orders like the value
		atomicMax(reinterpret_cast<unsigned long long*>(&sums[8]), (unsigned long long)__double_as_longlong((double)peak));
	}
}

int astc_compare_launch(const CompareLaunch& c)
{
	size_t groups = (c.texels + 255) / 256;
	if (groups > 256 * 16) groups = 256 * 16;          // 16 workgroups per CU is plenty for a streaming pass
	if (groups == 0) groups = 1;
	hipLaunchKernelGGL(astc_compare_images, dim3((uint32_t)groups), dim3(256), 0, static_cast<hipStream_t>(c.stream),
	                   c.d_a, c.type_a, c.d_b, c.type_b, c.texels, c.d_sums);
	return (int)hipGetLastError();
}

} // namespace astcd
