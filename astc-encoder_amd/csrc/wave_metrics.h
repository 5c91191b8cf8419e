// SPDX-License-Identifier: Apache-2.0
// Image comparison for the on-device quality metric (the CLI's -tl test mode).
//   ref: compute_error_metrics   Source/astcenccli_error_metrics.cpp:110-300 (LDR sums: per-channel
//        squared error, alpha-scaled squared error, RGB peak of the first image; HDR sums: squared log2
//        difference and the mPSNR tone-mapped squared difference over a range of f-stops, :60-107, :262-268)
// Per-texel arithmetic is the reference's (fp32 differences and squares of values / 255, or of
// half/float values clamped to 0..65504); the sums are fp64 as there, but added in a tree instead of
// texel by texel, so the totals agree to fp64 rounding rather than bit for bit.
#pragma once
#include "wave.h"

namespace astcd { inline namespace ASTC_VARIANT {

constexpr int METRIC_SUMS = 10;   // [0..3] squared error rgba, [4..7] alpha-scaled squared error rgba, [8] rgb peak, [9] unused
constexpr int METRIC_HDR_FIRST = 10;   // HDR comparisons add [10..13] squared log2 difference rgba, [14..17] mPSNR squared difference rgba
constexpr int METRIC_SUMS_HDR = 18;
constexpr int METRIC_STRIDE = 32;      // doubles per workgroup slot in the partials array

/* `unorm8` is an optional table of (float)i / 255.0f, i = 0..255 (the same correctly rounded quotients,
 * computed once per workgroup instead of eight times per texel). */
WV_FN void metric_load_texel(const void* img, size_t texel, uint32_t data_type, const float* unorm8, float c[4])
{
	if (data_type == 0)
	{
		uint32_t px;                                   // one 32-bit load per texel
		__builtin_memcpy(&px, static_cast<const uint8_t*>(img) + texel * 4, 4);
		for (int k = 0; k < 4; k++)
		{
			const uint32_t v = (px >> (8 * k)) & 0xFFu;
			c[k] = unorm8 ? unorm8[v] : (float)v / 255.0f;
		}
	}
	else
	{
		uint16_t h4[4] = { 0, 0, 0, 0 };
		float f4[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
		if (data_type == 1) __builtin_memcpy(h4, static_cast<const uint16_t*>(img) + texel * 4, 8);
		else __builtin_memcpy(f4, static_cast<const float*>(img) + texel * 4, 16);
		for (int k = 0; k < 4; k++)
		{
			float v = data_type == 1 ? half_to_float(h4[k]) : f4[k];
			v = v > 0.0f ? v : 0.0f;               // clamp(0, 65504, v), NaN -> 0 as the reference's max/min pair
			v = v < 65504.0f ? v : 65504.0f;
			c[k] = v;
		}
	}
}

/* The same for an RGBA8 texel that is already in a register (the four-texels-per-lane loop of kernel_metrics.hip). */
WV_FN void metric_unpack_rgba8(uint32_t px, const float* unorm8, float c[4])
{
	for (int k = 0; k < 4; k++)
	{
		const uint32_t v = (px >> (8 * k)) & 0xFFu;
		c[k] = unorm8 ? unorm8[v] : (float)v / 255.0f;
	}
}

/* Error terms of one texel pair whose components are loaded (see metric_texel_terms). */
WV_FN float metric_terms_of(const float c1[4], const float c2[4], float e[8])
{
	for (int k = 0; k < 4; k++)
	{
		float d = c1[k] - c2[k];
		e[k] = d * d;
		float ds = k < 3 ? d * c1[3] : d;
		e[4 + k] = ds * ds;
	}
	float m = c1[0] > c1[1] ? c1[0] : c1[1];
	return m > c1[2] ? m : c1[2];
}

/* log2 as the reference's metric code evaluates it (ref: log2(vfloat4), astcenc_vecmathlib.h:416-440:
 * exponent + 5th degree polynomial in the mantissa, Horner form). */
WV_FN float metric_log2(float x)
{
	const int i = float_as_int(x);
	const float e = (float)((int)(((unsigned)i & 0x7F800000u) >> 23) - 127);
	const float m = int_as_float((i & 0x007FFFFF) | 0x3F800000);
	float p = 0.0596515482674574969533f;
	p = p * m + -0.465725644288844778798f;
	p = p * m + 1.48116647521213171641f;
	p = p * m + -2.52074962577807006663f;
	p = p * m + 2.8882704548164776201f;
	p = p * (m - 1.0f);
	return p + e;
}

/* mpsnr_operator (ref: astcenccli_error_metrics.cpp:70-82): val * 2^fstop, gamma 1/2.2, scaled to 0..255.
 * powf there is libm's; here it is a double-precision pow rounded to float (the correctly rounded result in
 * all but a vanishing fraction of cases, on the device and on the host alike). */
WV_FN float metric_mpsnr_operator(float val, int fstop)
{
	const unsigned int uscale = 0x3f800000u + ((unsigned int)fstop << 23);
	const float scale = int_as_float((int)uscale);
#if WV_DEVICE
	const float v = (float)::pow((double)(val * scale), (double)(1.0f / 2.2f));     // ocml's double-precision pow
#else
	const float v = (float)__builtin_pow((double)(val * scale), (double)(1.0f / 2.2f));
#endif
	return f_clamp(v * 255.0f, 0.0f, 255.0f);
}

/* HDR terms of one texel pair: e[0..3] squared log2 difference, e[4..7] mpsnr_sumdiff over fstop_lo..fstop_hi
 * (ref: :262-268, :84-107). */
WV_FN void metric_hdr_terms(const float c1[4], const float c2[4], int fstop_lo, int fstop_hi, float e[8])
{
	for (int k = 0; k < 4; k++)
	{
		const float ld = metric_log2(c1[k]) - metric_log2(c2[k]);
		e[k] = ld * ld;
		float summa = 0.0f;
		for (int i = fstop_lo; i <= fstop_hi; i++)
		{
			const float mdiff = metric_mpsnr_operator(c1[k], i) - metric_mpsnr_operator(c2[k], i);
			summa += mdiff * mdiff;
		}
		e[4 + k] = summa;
	}
}

/* Error terms of one texel: e[0..3] squared difference, e[4..7] the same with RGB differences scaled by
 * the first image's alpha; returns max(r, g, b) of the first image. */
WV_FN float metric_texel_terms(const void* a, uint32_t type_a, const void* b, uint32_t type_b, size_t texel, const float* unorm8, float e[8],
                               float c1[4], float c2[4])
{
	metric_load_texel(a, texel, type_a, unorm8, c1);
	metric_load_texel(b, texel, type_b, unorm8, c2);
	return metric_terms_of(c1, c2, e);
}

} } // namespace astcd::ASTC_VARIANT
