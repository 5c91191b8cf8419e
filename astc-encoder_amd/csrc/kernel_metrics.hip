// SPDX-License-Identifier: Apache-2.0
// Image comparison kernel: squared-error sums of two images resident in HBM, for PSNR without a host
// round trip (ref: compute_error_metrics, Source/astcenccli_error_metrics.cpp:110).  Streaming and
// HBM-bound: every lane walks texels with a grid stride (coalesced 4 / 8 / 16-byte texel loads), keeps
// fp64 partial sums, the workgroup folds them in a fixed order and a second pass (one wavefront per quantity)
// adds the workgroups' partials in a fixed order (no atomics: the totals are reproducible).
#define ASTC_VARIANT v_metrics
#include "backend.h"
#include "wave_metrics.h"
#include <hip/hip_runtime.h>

namespace astcd {

constexpr uint32_t COMPARE_MAX_GROUPS = 2048;

/* Pass 1: every workgroup reduces its texels to fp64 partials (fixed order inside the group: lane shuffles,
 * then one thread per quantity adds the four waves' results) and writes them to its own slot -- no atomics,
 * so the totals do not depend on scheduling.  HDR: also the eight HDR sums (log2 and mPSNR terms). */
template <bool HDR>
__global__ void __launch_bounds__(256)
astc_compare_images(const void* __restrict__ a, uint32_t type_a, const void* __restrict__ b, uint32_t type_b,
                    size_t texels, int fstop_lo, int fstop_hi, double* __restrict__ partials)
{
	constexpr int NACC = HDR ? 16 : 8;                       // sums (the peak rides along separately)
	__shared__ float unorm8[256];
	__shared__ double wave_sums[4][NACC + 1];
	unorm8[threadIdx.x] = (float)threadIdx.x / 255.0f;      // blockDim.x == 256
	__syncthreads();
	double acc[NACC];
	for (int k = 0; k < NACC; k++) acc[k] = 0.0;
	float peak = 0.0f;
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	size_t first = 0;
	// Two RGBA8 images (the LDR quality figure of a compressed texture): FOUR texels per lane and trip through one 16-byte
	// load per image -- a lane with one 4-byte load per image in flight keeps 4 MB of the chip's HBM requests busy, a quarter
	// of what the bandwidth-latency product asks for (2.8 TB/s measured, profiles/r06z); the terms are added texel by texel
	// in index order as before.  What is left past the last whole group of four goes through the loop below.
	if (!HDR && type_a == 0 && type_b == 0 && ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15u) == 0)
	{
		const size_t quads = texels >> 2;
		const uint4* qa = static_cast<const uint4*>(a);
		const uint4* qb = static_cast<const uint4*>(b);
		for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += stride)
		{
			const uint4 pa = qa[q], pb = qb[q];
			const uint32_t xa[4] = { pa.x, pa.y, pa.z, pa.w }, xb[4] = { pb.x, pb.y, pb.z, pb.w };
			#pragma unroll
			for (int i = 0; i < 4; i++)
			{
				float e[8], c1[4], c2[4];
				metric_unpack_rgba8(xa[i], unorm8, c1);
				metric_unpack_rgba8(xb[i], unorm8, c2);
				const float m = metric_terms_of(c1, c2, e);
				peak = m > peak ? m : peak;
				for (int k = 0; k < 8; k++) acc[k] += (double)e[k];
			}
		}
		first = quads << 2;
	}
	for (size_t t = first + (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < texels; t += stride)
	{
		float e[8], c1[4], c2[4];
		float m = metric_texel_terms(a, type_a, b, type_b, t, unorm8, e, c1, c2);
		peak = m > peak ? m : peak;
		for (int k = 0; k < 8; k++) acc[k] += (double)e[k];
		if (HDR)
		{
			float h[8];
			metric_hdr_terms(c1, c2, fstop_lo, fstop_hi, h);
			for (int k = 0; k < 8; k++) acc[8 + k] += (double)h[k];
		}
	}
	for (int off = 32; off > 0; off >>= 1)
	{
		for (int k = 0; k < NACC; k++) acc[k] += __shfl_down(acc[k], off);
		float o = __shfl_down(peak, off);
		peak = o > peak ? o : peak;
	}
	const int wave = threadIdx.x >> 6;
	if ((threadIdx.x & 63) == 0)
	{
		for (int k = 0; k < NACC; k++) wave_sums[wave][k] = acc[k];
		wave_sums[wave][NACC] = (double)peak;
	}
	__syncthreads();
	if (threadIdx.x <= NACC)
	{
		const int k = threadIdx.x;
		double v = wave_sums[0][k];
		for (int w = 1; w < 4; w++) v = k < NACC ? v + wave_sums[w][k] : (wave_sums[w][k] > v ? wave_sums[w][k] : v);
		// slot layout = layout of the totals: [0..7] LDR sums, [8] peak, [10..17] HDR sums
		const int dst = k == NACC ? 8 : k < 8 ? k : METRIC_HDR_FIRST + (k - 8);
		partials[(size_t)blockIdx.x * METRIC_STRIDE + dst] = v;
	}
}

/* Pass 2: one wavefront per quantity.  Lane l adds the partials of workgroups l, l + 64, ... in index order, then
 * the 64 lane sums are folded with a fixed shuffle tree: the same totals on every run, whatever the scheduling. */
__global__ void __launch_bounds__(64)
astc_compare_finish(const double* __restrict__ partials, uint32_t groups, int hdr, double* __restrict__ sums)
{
	const int k = (int)blockIdx.x;
	if (k >= (hdr ? METRIC_SUMS_HDR : 9) || k == 9) return;
	const bool is_peak = k == 8;
	double v = 0.0;                                         // (every quantity is a sum of squares or a peak of non-negative values)
	for (uint32_t g = threadIdx.x; g < groups; g += 64)
	{
		const double p = partials[(size_t)g * METRIC_STRIDE + k];
		v = !is_peak ? v + p : (p > v ? p : v);
	}
	for (int off = 32; off > 0; off >>= 1)
	{
		const double o = __shfl_down(v, off);
		v = !is_peak ? v + o : (o > v ? o : v);
	}
	if (threadIdx.x == 0) sums[k] = v;
}

int astc_compare_launch(const CompareLaunch& c)
{
	size_t groups = (c.texels + 255) / 256;
	if (groups > COMPARE_MAX_GROUPS) groups = COMPARE_MAX_GROUPS;     // 8 workgroups of 256 per CU: plenty for a streaming pass
	if (groups == 0) groups = 1;
	double* partials = c.d_sums + METRIC_STRIDE;
	if (c.hdr)
		hipLaunchKernelGGL(astc_compare_images<true>, dim3((uint32_t)groups), dim3(256), 0, static_cast<hipStream_t>(c.stream),
		                   c.d_a, c.type_a, c.d_b, c.type_b, c.texels, c.fstop_lo, c.fstop_hi, partials);
	else
		hipLaunchKernelGGL(astc_compare_images<false>, dim3((uint32_t)groups), dim3(256), 0, static_cast<hipStream_t>(c.stream),
		                   c.d_a, c.type_a, c.d_b, c.type_b, c.texels, 0, 0, partials);
	hipLaunchKernelGGL(astc_compare_finish, dim3(METRIC_SUMS_HDR), dim3(64), 0, static_cast<hipStream_t>(c.stream), partials, (uint32_t)groups, c.hdr, c.d_sums);
	return (int)hipGetLastError();
}

/* Doubles the caller must provide at d_sums: the totals (one slot) followed by the per-workgroup partials. */
size_t astc_compare_scratch_doubles() { return METRIC_STRIDE + (size_t)COMPARE_MAX_GROUPS * METRIC_STRIDE; }

} // namespace astcd
