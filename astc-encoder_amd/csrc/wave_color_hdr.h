// SPDX-License-Identifier: Apache-2.0
// HDR endpoint formats: quantizers (pack) and decoders (unpack).
//   ref: quantize_and_unquantize_retain_top_{two,four}_bits, quantize_hdr_rgbo, quantize_hdr_rgb,
//        quantize_hdr_rgb_ldr_alpha, quantize_hdr_luminance_large_range,
//        try_quantize_hdr_luminance_small_range, quantize_hdr_alpha, quantize_hdr_rgb_alpha
//                                   Source/astcenc_color_quantize.cpp:848-1906
//        hdr_rgbo_unpack, hdr_rgb_unpack, hdr_rgb_ldr_alpha_unpack, hdr_luminance_*_unpack,
//        hdr_alpha_unpack, hdr_rgb_hdr_alpha_unpack
//                                   Source/astcenc_color_unquantize.cpp:310-841
// The bit layouts are those of ASTC spec C.2.14 (HDR endpoint modes 2, 3, 7, 11, 14, 15).
#pragma once
#include "wave_ctx.h"

namespace astcd { inline namespace ASTC_VARIANT {

struct ColorTabs;
WV_FN int quant_color(const ColorTabs& t, int value);
WV_FN int quant_color(const ColorTabs& t, int value, float valuef);

/* Quantize `value` so that its top bits survive the quantize/unquantize round trip.
 * (ref: color_quantize.cpp:848-916; keep_mask = 0xC0 or 0xF0) */
WV_FN uint8_t quant_retain_top_bits(const ColorTabs& t, uint8_t value, int keep_mask)
{
	bool perform_loop;
	uint8_t quantval;
	do
	{
		quantval = (uint8_t)quant_color(t, value);
		perform_loop = (value & keep_mask) != (quantval & keep_mask);
		if ((quantval & keep_mask) > (value & keep_mask)) value--;
		else if ((quantval & keep_mask) < (value & keep_mask)) value--;
	} while (perform_loop);
	return quantval;
}

/* (ref: quantize_hdr_rgbo :925) */
WV_FN void quantize_hdr_rgbo(const ColorTabs& t, f4 color, uint8_t* output)
{
	color.x = color.x + color.w;
	color.y = color.y + color.w;
	color.z = color.z + color.w;
	color = v4_clamp(0.0f, 65535.0f, color);
	f4 color_bak = color;

	int majcomp;
	if (color.x > color.y && color.x > color.z) majcomp = 0;
	else if (color.y > color.z) majcomp = 1;
	else majcomp = 2;

	if (majcomp == 1) color = mk4(color.y, color.x, color.z, color.w);
	else if (majcomp == 2) color = mk4(color.z, color.y, color.x, color.w);

	const int mode_bits[5][3] = { {11, 5, 7}, {11, 6, 5}, {10, 5, 8}, {9, 6, 7}, {8, 7, 6} };
	const float mode_cutoffs[5][2] = { {1024, 4096}, {2048, 1024}, {2048, 16384}, {8192, 16384}, {32768, 16384} };
	const float mode_rscales[5] = { 32.0f, 32.0f, 64.0f, 128.0f, 256.0f };
	const float mode_scales[5] = { 1.0f / 32.0f, 1.0f / 32.0f, 1.0f / 64.0f, 1.0f / 128.0f, 1.0f / 256.0f };

	float r_base = color.x;
	float g_base = color.x - color.y;
	float b_base = color.x - color.z;
	float s_base = color.w;

	for (int mode = 0; mode < 5; mode++)
	{
		if (g_base > mode_cutoffs[mode][0] || b_base > mode_cutoffs[mode][0] || s_base > mode_cutoffs[mode][1]) continue;

		int mode_enc = mode < 4 ? (mode | (majcomp << 2)) : (majcomp | 0xC);
		float mode_scale = mode_scales[mode];
		float mode_rscale = mode_rscales[mode];
		int gb_intcutoff = 1 << mode_bits[mode][1];
		int s_intcutoff = 1 << mode_bits[mode][2];

		int r_intval = flt2int_rtn(r_base * mode_scale);
		int r_lowbits = r_intval & 0x3f;
		r_lowbits |= (mode_enc & 3) << 6;
		uint8_t r_quantval = quant_retain_top_bits(t, (uint8_t)r_lowbits, 0xC0);
		r_intval = (r_intval & ~0x3f) | (r_quantval & 0x3f);
		float r_fval = (float)r_intval * mode_rscale;

		float g_fval = r_fval - color.y;
		float b_fval = r_fval - color.z;
		g_fval = f_clamp(g_fval, 0.0f, 65535.0f);
		b_fval = f_clamp(b_fval, 0.0f, 65535.0f);
		int g_intval = flt2int_rtn(g_fval * mode_scale);
		int b_intval = flt2int_rtn(b_fval * mode_scale);
		if (g_intval >= gb_intcutoff || b_intval >= gb_intcutoff) continue;

		int g_lowbits = g_intval & 0x1f;
		int b_lowbits = b_intval & 0x1f;

		int bit0 = 0, bit1 = 0, bit2 = 0, bit3 = 0;
		switch (mode)
		{
		case 0: case 2: bit0 = (r_intval >> 9) & 1; break;
		case 1: case 3: bit0 = (r_intval >> 8) & 1; break;
		default: bit0 = (g_intval >> 6) & 1; break;
		}
		switch (mode)
		{
		case 0: case 1: case 2: case 3: bit2 = (r_intval >> 7) & 1; break;
		default: bit2 = (b_intval >> 6) & 1; break;
		}
		switch (mode)
		{
		case 0: case 2: bit1 = (r_intval >> 8) & 1; break;
		default: bit1 = (g_intval >> 5) & 1; break;
		}
		switch (mode)
		{
		case 0: bit3 = (r_intval >> 10) & 1; break;
		case 2: bit3 = (r_intval >> 6) & 1; break;
		default: bit3 = (b_intval >> 5) & 1; break;
		}

		g_lowbits |= (mode_enc & 0x4) << 5;
		b_lowbits |= (mode_enc & 0x8) << 4;
		g_lowbits |= bit0 << 6;
		g_lowbits |= bit1 << 5;
		b_lowbits |= bit2 << 6;
		b_lowbits |= bit3 << 5;

		uint8_t g_quantval = quant_retain_top_bits(t, (uint8_t)g_lowbits, 0xF0);
		uint8_t b_quantval = quant_retain_top_bits(t, (uint8_t)b_lowbits, 0xF0);

		g_intval = (g_intval & ~0x1f) | (g_quantval & 0x1f);
		b_intval = (b_intval & ~0x1f) | (b_quantval & 0x1f);
		g_fval = (float)g_intval * mode_rscale;
		b_fval = (float)b_intval * mode_rscale;

		float rgb_errorsum = (r_fval - color.x) + (r_fval - g_fval - color.y) + (r_fval - b_fval - color.z);
		float s_fval = s_base + rgb_errorsum * (1.0f / 3.0f);
		s_fval = f_clamp(s_fval, 0.0f, 1e9f);
		int s_intval = flt2int_rtn(s_fval * mode_scale);
		if (s_intval >= s_intcutoff) continue;

		int s_lowbits = s_intval & 0x1f;
		int bit4, bit5, bit6;
		bit6 = mode == 1 ? (r_intval >> 9) & 1 : (s_intval >> 5) & 1;
		bit5 = mode == 4 ? (r_intval >> 7) & 1 : mode == 1 ? (r_intval >> 10) & 1 : (s_intval >> 6) & 1;
		bit4 = mode == 2 ? (s_intval >> 7) & 1 : (r_intval >> 6) & 1;

		s_lowbits |= bit6 << 5;
		s_lowbits |= bit5 << 6;
		s_lowbits |= bit4 << 7;
		uint8_t s_quantval = quant_retain_top_bits(t, (uint8_t)s_lowbits, 0xF0);

		output[0] = r_quantval; output[1] = g_quantval; output[2] = b_quantval; output[3] = s_quantval;
		return;
	}

	// no sub-mode fits: flat 7/7/7/7-bit layout ("mode 5")
	float vals[4] = { color_bak.x, color_bak.y, color_bak.z, color_bak.w };
	int ivals[4];
	float cvals[3];
	for (int i = 0; i < 3; i++)
	{
		vals[i] = f_clamp(vals[i], 0.0f, 65020.0f);
		ivals[i] = flt2int_rtn(vals[i] * (1.0f / 512.0f));
		cvals[i] = (float)ivals[i] * 512.0f;
	}
	float rgb_errorsum = (cvals[0] - vals[0]) + (cvals[1] - vals[1]) + (cvals[2] - vals[2]);
	vals[3] += rgb_errorsum * (1.0f / 3.0f);
	vals[3] = f_clamp(vals[3], 0.0f, 65020.0f);
	ivals[3] = flt2int_rtn(vals[3] * (1.0f / 512.0f));

	int encvals[4];
	encvals[0] = (ivals[0] & 0x3f) | 0xC0;
	encvals[1] = (ivals[1] & 0x7f) | 0x80;
	encvals[2] = (ivals[2] & 0x7f) | 0x80;
	encvals[3] = (ivals[3] & 0x7f) | ((ivals[0] & 0x40) << 1);
	for (int i = 0; i < 4; i++) output[i] = quant_retain_top_bits(t, (uint8_t)encvals[i], 0xF0);
}

/* (ref: quantize_hdr_rgb :1253) */
WV_FN void quantize_hdr_rgb(const ColorTabs& t, f4 color0, f4 color1, uint8_t* output)
{
	color0 = v4_clamp(0.0f, 65535.0f, color0);
	color1 = v4_clamp(0.0f, 65535.0f, color1);
	f4 color0_bak = color0, color1_bak = color1;

	int majcomp;
	if (color1.x > color1.y && color1.x > color1.z) majcomp = 0;
	else if (color1.y > color1.z) majcomp = 1;
	else majcomp = 2;

	if (majcomp == 1)
	{
		color0 = mk4(color0.y, color0.x, color0.z, color0.w);
		color1 = mk4(color1.y, color1.x, color1.z, color1.w);
	}
	else if (majcomp == 2)
	{
		color0 = mk4(color0.z, color0.y, color0.x, color0.w);
		color1 = mk4(color1.z, color1.y, color1.x, color1.w);
	}

	float a_base = color1.x;
	a_base = f_clamp(a_base, 0.0f, 65535.0f);
	float b0_base = a_base - color1.y;
	float b1_base = a_base - color1.z;
	float c_base = a_base - color0.x;
	float d0_base = a_base - b0_base - c_base - color0.y;
	float d1_base = a_base - b1_base - c_base - color0.z;

	const int mode_bits[8][4] = { {9, 7, 6, 7}, {9, 8, 6, 6}, {10, 6, 7, 7}, {10, 7, 7, 6}, {11, 8, 6, 5}, {11, 6, 8, 6}, {12, 7, 7, 5}, {12, 6, 7, 6} };
	const float mode_cutoffs[8][3] = { {16384, 8192, 8192}, {32768, 8192, 4096}, {4096, 8192, 4096}, {8192, 8192, 2048},
	                                   {8192, 2048, 512}, {2048, 8192, 1024}, {2048, 2048, 256}, {1024, 2048, 512} };
	const float mode_scales[8] = { 1.0f / 128.0f, 1.0f / 128.0f, 1.0f / 64.0f, 1.0f / 64.0f, 1.0f / 32.0f, 1.0f / 32.0f, 1.0f / 16.0f, 1.0f / 16.0f };
	const float mode_rscales[8] = { 128.0f, 128.0f, 64.0f, 64.0f, 32.0f, 32.0f, 16.0f, 16.0f };

	for (int mode = 7; mode >= 0; mode--)
	{
		float b_cutoff = mode_cutoffs[mode][0], c_cutoff = mode_cutoffs[mode][1], d_cutoff = mode_cutoffs[mode][2];
		if (b0_base > b_cutoff || b1_base > b_cutoff || c_base > c_cutoff || f_abs(d0_base) > d_cutoff || f_abs(d1_base) > d_cutoff) continue;

		float mode_scale = mode_scales[mode];
		float mode_rscale = mode_rscales[mode];
		int b_intcutoff = 1 << mode_bits[mode][1];
		int c_intcutoff = 1 << mode_bits[mode][2];
		int d_intcutoff = 1 << (mode_bits[mode][3] - 1);

		int a_intval = flt2int_rtn(a_base * mode_scale);
		int a_lowbits = a_intval & 0xFF;
		int a_quantval = quant_color(t, a_lowbits);
		int a_uquantval = a_quantval;
		a_intval = (a_intval & ~0xFF) | a_uquantval;
		float a_fval = (float)a_intval * mode_rscale;

		float c_fval = a_fval - color0.x;
		c_fval = f_clamp(c_fval, 0.0f, 65535.0f);
		int c_intval = flt2int_rtn(c_fval * mode_scale);
		if (c_intval >= c_intcutoff) continue;

		int c_lowbits = c_intval & 0x3f;
		c_lowbits |= (mode & 1) << 7;
		c_lowbits |= (a_intval & 0x100) >> 2;
		uint8_t c_quantval = quant_retain_top_bits(t, (uint8_t)c_lowbits, 0xC0);
		c_intval = (c_intval & ~0x3F) | (c_quantval & 0x3F);
		c_fval = (float)c_intval * mode_rscale;

		float b0_fval = a_fval - color1.y;
		float b1_fval = a_fval - color1.z;
		b0_fval = f_clamp(b0_fval, 0.0f, 65535.0f);
		b1_fval = f_clamp(b1_fval, 0.0f, 65535.0f);
		int b0_intval = flt2int_rtn(b0_fval * mode_scale);
		int b1_intval = flt2int_rtn(b1_fval * mode_scale);
		if (b0_intval >= b_intcutoff || b1_intval >= b_intcutoff) continue;

		int b0_lowbits = b0_intval & 0x3f;
		int b1_lowbits = b1_intval & 0x3f;

		int bit0 = 0, bit1 = 0;
		switch (mode)
		{
		case 0: case 1: case 3: case 4: case 6: bit0 = (b0_intval >> 6) & 1; break;
		default: bit0 = (a_intval >> 9) & 1; break;
		}
		switch (mode)
		{
		case 0: case 1: case 3: case 4: case 6: bit1 = (b1_intval >> 6) & 1; break;
		case 2: bit1 = (c_intval >> 6) & 1; break;
		default: bit1 = (a_intval >> 10) & 1; break;
		}

		b0_lowbits |= bit0 << 6;
		b1_lowbits |= bit1 << 6;
		b0_lowbits |= ((mode >> 1) & 1) << 7;
		b1_lowbits |= ((mode >> 2) & 1) << 7;

		uint8_t b0_quantval = quant_retain_top_bits(t, (uint8_t)b0_lowbits, 0xC0);
		uint8_t b1_quantval = quant_retain_top_bits(t, (uint8_t)b1_lowbits, 0xC0);

		b0_intval = (b0_intval & ~0x3f) | (b0_quantval & 0x3f);
		b1_intval = (b1_intval & ~0x3f) | (b1_quantval & 0x3f);
		b0_fval = (float)b0_intval * mode_rscale;
		b1_fval = (float)b1_intval * mode_rscale;

		float d0_fval = a_fval - b0_fval - c_fval - color0.y;
		float d1_fval = a_fval - b1_fval - c_fval - color0.z;
		d0_fval = f_clamp(d0_fval, -65535.0f, 65535.0f);
		d1_fval = f_clamp(d1_fval, -65535.0f, 65535.0f);
		int d0_intval = flt2int_rtn(d0_fval * mode_scale);
		int d1_intval = flt2int_rtn(d1_fval * mode_scale);
		int ad0 = d0_intval < 0 ? -d0_intval : d0_intval;
		int ad1 = d1_intval < 0 ? -d1_intval : d1_intval;
		if (ad0 >= d_intcutoff || ad1 >= d_intcutoff) continue;

		int d0_lowbits = d0_intval & 0x1f;
		int d1_lowbits = d1_intval & 0x1f;

		int bit2 = 0, bit3 = 0, bit4, bit5;
		switch (mode)
		{
		case 0: case 2: bit2 = (d0_intval >> 6) & 1; break;
		case 1: case 4: bit2 = (b0_intval >> 7) & 1; break;
		case 3: bit2 = (a_intval >> 9) & 1; break;
		case 5: bit2 = (c_intval >> 7) & 1; break;
		default: bit2 = (a_intval >> 11) & 1; break;
		}
		switch (mode)
		{
		case 0: case 2: bit3 = (d1_intval >> 6) & 1; break;
		case 1: case 4: bit3 = (b1_intval >> 7) & 1; break;
		default: bit3 = (c_intval >> 6) & 1; break;
		}
		if (mode == 4 || mode == 6)
		{
			bit4 = (a_intval >> 9) & 1;
			bit5 = (a_intval >> 10) & 1;
		}
		else
		{
			bit4 = (d0_intval >> 5) & 1;
			bit5 = (d1_intval >> 5) & 1;
		}

		d0_lowbits |= bit2 << 6;
		d1_lowbits |= bit3 << 6;
		d0_lowbits |= bit4 << 5;
		d1_lowbits |= bit5 << 5;
		d0_lowbits |= (majcomp & 1) << 7;
		d1_lowbits |= ((majcomp >> 1) & 1) << 7;

		uint8_t d0_quantval = quant_retain_top_bits(t, (uint8_t)d0_lowbits, 0xF0);
		uint8_t d1_quantval = quant_retain_top_bits(t, (uint8_t)d1_lowbits, 0xF0);

		output[0] = (uint8_t)a_quantval;
		output[1] = c_quantval;
		output[2] = b0_quantval;
		output[3] = b1_quantval;
		output[4] = d0_quantval;
		output[5] = d1_quantval;
		return;
	}

	// no sub-mode fits: direct 8/8/7-bit endpoints (majcomp == 3 encoding)
	float vals[6] = { color0_bak.x, color1_bak.x, color0_bak.y, color1_bak.y, color0_bak.z, color1_bak.z };
	for (int i = 0; i < 6; i++) vals[i] = f_clamp(vals[i], 0.0f, 65020.0f);
	for (int i = 0; i < 4; i++)
	{
		int idx = flt2int_rtn(vals[i] * 1.0f / 256.0f);
		output[i] = (uint8_t)quant_color(t, idx);
	}
	for (int i = 4; i < 6; i++)
	{
		int idx = flt2int_rtn(vals[i] * 1.0f / 512.0f) + 128;
		output[i] = quant_retain_top_bits(t, (uint8_t)idx, 0xC0);
	}
}

/* (ref: quantize_hdr_luminance_large_range :1644) */
WV_FN void quantize_hdr_luminance_large_range(const ColorTabs& t, f4 color0, f4 color1, uint8_t* output)
{
	float lum0 = hadd_rgb_s(color0) * (1.0f / 3.0f);
	float lum1 = hadd_rgb_s(color1) * (1.0f / 3.0f);
	if (lum1 < lum0)
	{
		float avg = (lum0 + lum1) * 0.5f;
		lum0 = avg;
		lum1 = avg;
	}
	int ilum1 = flt2int_rtn(lum1);
	int ilum0 = flt2int_rtn(lum0);

	int upper_v0 = i_clamp((ilum0 + 128) >> 8, 0, 255);
	int upper_v1 = i_clamp((ilum1 + 128) >> 8, 0, 255);
	int lower_v0 = i_clamp((ilum1 + 256) >> 8, 0, 255);
	int lower_v1 = i_clamp(ilum0 >> 8, 0, 255);

	int upper0_dec = upper_v0 << 8;
	int upper1_dec = upper_v1 << 8;
	int lower0_dec = (lower_v1 << 8) + 128;
	int lower1_dec = (lower_v0 << 8) - 128;

	int upper0_diff = upper0_dec - ilum0, upper1_diff = upper1_dec - ilum1;
	int lower0_diff = lower0_dec - ilum0, lower1_diff = lower1_dec - ilum1;
	int upper_error = (upper0_diff * upper0_diff) + (upper1_diff * upper1_diff);
	int lower_error = (lower0_diff * lower0_diff) + (lower1_diff * lower1_diff);

	int v0, v1;
	if (upper_error < lower_error) { v0 = upper_v0; v1 = upper_v1; }
	else { v0 = lower_v0; v1 = lower_v1; }
	output[0] = (uint8_t)quant_color(t, v0);
	output[1] = (uint8_t)quant_color(t, v1);
}

/* (ref: try_quantize_hdr_luminance_small_range :1718) */
WV_FN bool try_quantize_hdr_luminance_small_range(const ColorTabs& t, f4 color0, f4 color1, uint8_t* output)
{
	float lum0 = hadd_rgb_s(color0) * (1.0f / 3.0f);
	float lum1 = hadd_rgb_s(color1) * (1.0f / 3.0f);
	if (lum1 < lum0)
	{
		float avg = (lum0 + lum1) * 0.5f;
		lum0 = avg;
		lum1 = avg;
	}
	int ilum1 = flt2int_rtn(lum1);
	int ilum0 = flt2int_rtn(lum0);
	if (ilum1 - ilum0 > 2048) return false;

	int lowval, highval, diffval, v0, v1, v0e, v1e, v0d, v1d;

	// sub-mode with 11-bit base, 4-bit offset
	lowval = i_clamp((ilum0 + 16) >> 5, 0, 2047);
	highval = i_clamp((ilum1 + 16) >> 5, 0, 2047);
	v0 = lowval & 0x7F;
	v0e = quant_color(t, v0);
	v0d = v0e;
	if (v0d < 0x80)
	{
		lowval = (lowval & ~0x7F) | v0d;
		diffval = highval - lowval;
		if (diffval >= 0 && diffval <= 15)
		{
			v1 = ((lowval >> 3) & 0xF0) | diffval;
			v1e = quant_color(t, v1);
			v1d = v1e;
			if ((v1d & 0xF0) == (v1 & 0xF0))
			{
				output[0] = (uint8_t)v0e;
				output[1] = (uint8_t)v1e;
				return true;
			}
		}
	}

	// sub-mode with 10-bit base, 5-bit offset
	lowval = i_clamp((ilum0 + 32) >> 6, 0, 1023);
	highval = i_clamp((ilum1 + 32) >> 6, 0, 1023);
	v0 = (lowval & 0x7F) | 0x80;
	v0e = quant_color(t, v0);
	v0d = v0e;
	if ((v0d & 0x80) == 0) return false;

	lowval = (lowval & ~0x7F) | (v0d & 0x7F);
	diffval = highval - lowval;
	if (diffval < 0 || diffval > 31) return false;

	v1 = ((lowval >> 2) & 0xE0) | diffval;
	v1e = quant_color(t, v1);
	v1d = v1e;
	if ((v1d & 0xE0) != (v1 & 0xE0)) return false;

	output[0] = (uint8_t)v0e;
	output[1] = (uint8_t)v1e;
	return true;
}

/* (ref: quantize_hdr_alpha :1820) */
WV_FN void quantize_hdr_alpha(const ColorTabs& t, float alpha0, float alpha1, uint8_t* output)
{
	alpha0 = f_clamp(alpha0, 0.0f, 65280.0f);
	alpha1 = f_clamp(alpha1, 0.0f, 65280.0f);
	int ialpha0 = flt2int_rtn(alpha0);
	int ialpha1 = flt2int_rtn(alpha1);

	int val0, val1, diffval, v6, v7, v6e, v7e, v6d, v7d;
	const int testbits[3] = { 0xE0, 0xF0, 0xF8 };

	for (int i = 2; i >= 0; i--)
	{
		val0 = (ialpha0 + (128 >> i)) >> (8 - i);
		val1 = (ialpha1 + (128 >> i)) >> (8 - i);

		v6 = (val0 & 0x7F) | ((i & 1) << 7);
		v6e = quant_color(t, v6);
		v6d = v6e;
		if ((v6 ^ v6d) & 0x80) continue;

		val0 = (val0 & ~0x7f) | (v6d & 0x7f);
		diffval = val1 - val0;
		int cutoff = 32 >> i;
		int mask = 2 * cutoff - 1;
		if (diffval < -cutoff || diffval >= cutoff) continue;

		v7 = ((i & 2) << 6) | ((val0 >> 7) << (6 - i)) | (diffval & mask);
		v7e = quant_color(t, v7);
		v7d = v7e;
		if ((v7 ^ v7d) & testbits[i]) continue;

		output[0] = (uint8_t)v6e;
		output[1] = (uint8_t)v7e;
		return;
	}

	val0 = (ialpha0 + 256) >> 9;
	val1 = (ialpha1 + 256) >> 9;
	v6 = val0 | 0x80;
	v7 = val1 | 0x80;
	output[0] = (uint8_t)quant_color(t, v6);
	output[1] = (uint8_t)quant_color(t, v7);
}

// ---------------------------------------------------------------------------------------------
// Decoders
// ---------------------------------------------------------------------------------------------

WV_FN int safe_signed_lsh(int val, int shift)
{
	return (int)((unsigned int)val << shift);
}

/* (ref: hdr_rgbo_unpack :310) */
WV_FN void hdr_rgbo_unpack(const uint8_t* input, i4& output0, i4& output1)
{
	int v0 = input[0], v1 = input[1], v2 = input[2], v3 = input[3];
	int modeval = ((v0 & 0xC0) >> 6) | (((v1 & 0x80) >> 7) << 2) | (((v2 & 0x80) >> 7) << 3);

	int majcomp, mode;
	if ((modeval & 0xC) != 0xC) { majcomp = modeval >> 2; mode = modeval & 3; }
	else if (modeval != 0xF) { majcomp = modeval & 3; mode = 4; }
	else { majcomp = 0; mode = 5; }

	int red = v0 & 0x3F, green = v1 & 0x1F, blue = v2 & 0x1F, scale = v3 & 0x1F;
	int bit0 = (v1 >> 6) & 1, bit1 = (v1 >> 5) & 1, bit2 = (v2 >> 6) & 1, bit3 = (v2 >> 5) & 1;
	int bit4 = (v3 >> 7) & 1, bit5 = (v3 >> 6) & 1, bit6 = (v3 >> 5) & 1;

	int ohcomp = 1 << mode;
	if (ohcomp & 0x30) green |= bit0 << 6;
	if (ohcomp & 0x3A) green |= bit1 << 5;
	if (ohcomp & 0x30) blue |= bit2 << 6;
	if (ohcomp & 0x3A) blue |= bit3 << 5;
	if (ohcomp & 0x3D) scale |= bit6 << 5;
	if (ohcomp & 0x2D) scale |= bit5 << 6;
	if (ohcomp & 0x04) scale |= bit4 << 7;
	if (ohcomp & 0x3B) red |= bit4 << 6;
	if (ohcomp & 0x04) red |= bit3 << 6;
	if (ohcomp & 0x10) red |= bit5 << 7;
	if (ohcomp & 0x0F) red |= bit2 << 7;
	if (ohcomp & 0x05) red |= bit1 << 8;
	if (ohcomp & 0x0A) red |= bit0 << 8;
	if (ohcomp & 0x05) red |= bit0 << 9;
	if (ohcomp & 0x02) red |= bit6 << 9;
	if (ohcomp & 0x01) red |= bit3 << 10;
	if (ohcomp & 0x02) red |= bit5 << 10;

	const int shamts[6] = { 1, 1, 2, 3, 4, 5 };
	int shamt = shamts[mode];
	red <<= shamt; green <<= shamt; blue <<= shamt; scale <<= shamt;

	if (mode != 5)
	{
		green = red - green;
		blue = red - blue;
	}

	int temp;
	if (majcomp == 1) { temp = red; red = green; green = temp; }
	else if (majcomp == 2) { temp = red; red = blue; blue = temp; }

	int red0 = red - scale, green0 = green - scale, blue0 = blue - scale;
	red = i_max(red, 0); green = i_max(green, 0); blue = i_max(blue, 0);
	red0 = i_max(red0, 0); green0 = i_max(green0, 0); blue0 = i_max(blue0, 0);

	output0 = mki4(red0 << 4, green0 << 4, blue0 << 4, 0x7800);
	output1 = mki4(red << 4, green << 4, blue << 4, 0x7800);
}

/* (ref: hdr_rgb_unpack :498) */
WV_FN void hdr_rgb_unpack(const uint8_t* input, i4& output0, i4& output1)
{
	int v0 = input[0], v1 = input[1], v2 = input[2], v3 = input[3], v4 = input[4], v5 = input[5];
	int modeval = ((v1 & 0x80) >> 7) | (((v2 & 0x80) >> 7) << 1) | (((v3 & 0x80) >> 7) << 2);
	int majcomp = ((v4 & 0x80) >> 7) | (((v5 & 0x80) >> 7) << 1);

	if (majcomp == 3)
	{
		output0 = mki4(v0 << 8, v2 << 8, (v4 & 0x7F) << 9, 0x7800);
		output1 = mki4(v1 << 8, v3 << 8, (v5 & 0x7F) << 9, 0x7800);
		return;
	}

	int a = v0 | ((v1 & 0x40) << 2);
	int b0 = v2 & 0x3f, b1 = v3 & 0x3f, c = v1 & 0x3f, d0 = v4 & 0x7f, d1 = v5 & 0x7f;

	const int dbits_tab[8] = { 7, 6, 7, 6, 5, 6, 5, 6 };
	int dbits = dbits_tab[modeval];

	int bit0 = (v2 >> 6) & 1, bit1 = (v3 >> 6) & 1, bit2 = (v4 >> 6) & 1, bit3 = (v5 >> 6) & 1;
	int bit4 = (v4 >> 5) & 1, bit5 = (v5 >> 5) & 1;

	int ohmod = 1 << modeval;
	if (ohmod & 0xA4) a |= bit0 << 9;
	if (ohmod & 0x8) a |= bit2 << 9;
	if (ohmod & 0x50) a |= bit4 << 9;
	if (ohmod & 0x50) a |= bit5 << 10;
	if (ohmod & 0xA0) a |= bit1 << 10;
	if (ohmod & 0xC0) a |= bit2 << 11;
	if (ohmod & 0x4) c |= bit1 << 6;
	if (ohmod & 0xE8) c |= bit3 << 6;
	if (ohmod & 0x20) c |= bit2 << 7;
	if (ohmod & 0x5B) { b0 |= bit0 << 6; b1 |= bit1 << 6; }
	if (ohmod & 0x12) { b0 |= bit2 << 7; b1 |= bit3 << 7; }
	if (ohmod & 0xAF) { d0 |= bit4 << 5; d1 |= bit5 << 5; }
	if (ohmod & 0x5) { d0 |= bit2 << 6; d1 |= bit3 << 6; }

	// sign-extend d0 / d1 from dbits
	int sx_shamt = 32 - dbits;
	int d0x = safe_signed_lsh(d0, sx_shamt); d0x >>= sx_shamt;
	int d1x = safe_signed_lsh(d1, sx_shamt); d1x >>= sx_shamt;
	d0 = d0x; d1 = d1x;

	int val_shamt = (modeval >> 1) ^ 3;
	a = safe_signed_lsh(a, val_shamt);
	b0 = safe_signed_lsh(b0, val_shamt);
	b1 = safe_signed_lsh(b1, val_shamt);
	c = safe_signed_lsh(c, val_shamt);
	d0 = safe_signed_lsh(d0, val_shamt);
	d1 = safe_signed_lsh(d1, val_shamt);

	int red1 = a, green1 = a - b0, blue1 = a - b1;
	int red0 = a - c, green0 = a - b0 - c - d0, blue0 = a - b1 - c - d1;

	red0 = i_clamp(red0, 0, 4095); green0 = i_clamp(green0, 0, 4095); blue0 = i_clamp(blue0, 0, 4095);
	red1 = i_clamp(red1, 0, 4095); green1 = i_clamp(green1, 0, 4095); blue1 = i_clamp(blue1, 0, 4095);

	int temp0, temp1;
	if (majcomp == 1)
	{
		temp0 = red0; temp1 = red1; red0 = green0; red1 = green1; green0 = temp0; green1 = temp1;
	}
	else if (majcomp == 2)
	{
		temp0 = red0; temp1 = red1; red0 = blue0; red1 = blue1; blue0 = temp0; blue1 = temp1;
	}

	output0 = mki4(red0 << 4, green0 << 4, blue0 << 4, 0x7800);
	output1 = mki4(red1 << 4, green1 << 4, blue1 << 4, 0x7800);
}

/* (ref: hdr_luminance_small_range_unpack :708) */
WV_FN void hdr_luminance_small_range_unpack(const uint8_t* input, i4& output0, i4& output1)
{
	int v0 = input[0], v1 = input[1];
	int y0, y1;
	if (v0 & 0x80)
	{
		y0 = ((v1 & 0xE0) << 4) | ((v0 & 0x7F) << 2);
		y1 = (v1 & 0x1F) << 2;
	}
	else
	{
		y0 = ((v1 & 0xF0) << 4) | ((v0 & 0x7F) << 1);
		y1 = (v1 & 0xF) << 1;
	}
	y1 += y0;
	if (y1 > 0xFFF) y1 = 0xFFF;
	output0 = mki4(y0 << 4, y0 << 4, y0 << 4, 0x7800);
	output1 = mki4(y1 << 4, y1 << 4, y1 << 4, 0x7800);
}

/* (ref: hdr_luminance_large_range_unpack :745) */
WV_FN void hdr_luminance_large_range_unpack(const uint8_t* input, i4& output0, i4& output1)
{
	int v0 = input[0], v1 = input[1];
	int y0, y1;
	if (v1 >= v0) { y0 = v0 << 4; y1 = v1 << 4; }
	else { y0 = (v1 << 4) + 8; y1 = (v0 << 4) - 8; }
	output0 = mki4(y0 << 4, y0 << 4, y0 << 4, 0x7800);
	output1 = mki4(y1 << 4, y1 << 4, y1 << 4, 0x7800);
}

/* (ref: hdr_alpha_unpack :776) */
WV_FN void hdr_alpha_unpack(const uint8_t* input, int& output0, int& output1)
{
	int v6 = input[0], v7 = input[1];
	int modeval = ((v6 >> 7) & 1) | ((v7 >> 6) & 2);
	v6 &= 0x7F;
	v7 &= 0x7F;
	if (modeval == 3)
	{
		output0 = v6 << 5;
		output1 = v7 << 5;
	}
	else
	{
		v6 |= (v7 << (modeval + 1)) & 0x780;
		v7 &= (0x3f >> modeval);
		v7 ^= 32 >> modeval;
		v7 -= 32 >> modeval;
		v6 = v6 << (4 - modeval);
		v7 = safe_signed_lsh(v7, 4 - modeval);
		v7 = i_clamp(v6 + v7, 0, 0xFFF);
		output0 = v6;
		output1 = v7;
	}
	output0 <<= 4;
	output1 <<= 4;
}

} } // namespace astcd::ASTC_VARIANT
