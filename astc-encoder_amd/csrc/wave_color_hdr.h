// SPDX-License-Identifier: Apache-2.0
// HDR endpoint formats: coders (sub-modes side by side on lanes, driven by width / spare-bit tables) and decoders.
//   ref (behaviour): quantize_hdr_rgbo, quantize_hdr_rgb, quantize_hdr_rgb_ldr_alpha, quantize_hdr_luminance_large_range,
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

// ---- coders ------------------------------------------------------------------------------------------------
//
// An HDR endpoint format is a family of sub-modes -- different splits of the available bits between a base value
// and its offsets -- and the coder's job is to find the first sub-mode of a fixed preference order in which every
// field fits (ref: the mode loops of quantize_hdr_rgbo :1002 and quantize_hdr_rgb :1353, the three tries of
// quantize_hdr_alpha :1838).  The tries are independent of each other, so they run SIDE BY SIDE: sixteen lanes per
// partition, one sub-mode per lane (lanes 0..7 colour, 8..10 alpha), each leaving "fits" and its bytes in a small
// LDS record; one lane per partition then takes the first record that fits, or codes the format's escape layout.
// The sub-modes are data: field widths per sub-mode (every cut-off and scale follows from them) and, per spare bit
// of the byte layout, which field's which bit it carries (spec tables 26, 28: "Endpoint Unquantization" of modes
// 7 and 11).

/* Largest value <= `value` whose bits under `keep` survive the quantize / unquantize round trip, quantized
 * (ref: quantize_and_unquantize_retain_top_{two,four}_bits :848-916). */
WV_FN int quant_keeping(const ColorTabs& t, int value, int keep)
{
	for (;;)
	{
		const int q = quant_color(t, value & 0xFF);
		if ((((value & 0xFF) ^ q) & keep) == 0) return q;
		value--;
	}
}

/* One spare bit of a byte layout: bit `shift` of field `field`. */
struct SpareBit { uint8_t field, shift; };
WV_FN int spare_bit(SpareBit s, int f0, int f1, int f2, int f3, int f4, int f5)
{
	const int v = s.field == 0 ? f0 : s.field == 1 ? f1 : s.field == 2 ? f2 : s.field == 3 ? f3 : s.field == 4 ? f4 : f5;
	return (v >> s.shift) & 1;
}

/* The record a sub-mode lane leaves: [0] fits, [1..6] bytes. */
constexpr int HDR_TRY_BYTES = 8;
constexpr int HDR_TRY_LANES = 16;       // per partition: 0..7 colour sub-modes, 8..10 alpha sub-modes

/* Colour with its largest component moved to x (ties: the reference's comparison order). */
WV_FN int major_component(f4 v)
{
	if (v.x > v.y && v.x > v.z) return 0;
	return v.y > v.z ? 1 : 2;
}
WV_FN f4 major_first(f4 v, int major) { return major == 1 ? mk4(v.y, v.x, v.z, v.w) : major == 2 ? mk4(v.z, v.y, v.x, v.w) : v; }

/* RGB + offset format, sub-mode `mode` 0..4: base R (the major component), G and B as differences from it, the
 * offset S as the last field.  (ref: quantize_hdr_rgbo :925, one iteration of its mode loop) */
WV_FN void hdr_try_rgbo(const ColorTabs& t, f4 rgbo, int mode, uint8_t* rec)
{
	// widths of R, G/B, S per sub-mode
	const uint8_t widths[5][3] = { { 11, 5, 7 }, { 11, 6, 5 }, { 10, 5, 8 }, { 9, 6, 7 }, { 8, 7, 6 } };
	// spare bits 0..6 of bytes 1..3 (fields: 0 R, 1 G, 2 B, 3 S)
	const SpareBit spare[5][7] = {
		{ { 0, 9 }, { 0, 8 }, { 0, 7 }, { 0, 10 }, { 0, 6 }, { 3, 6 }, { 3, 5 } },
		{ { 0, 8 }, { 1, 5 }, { 0, 7 }, { 2, 5 },  { 0, 6 }, { 0, 10 }, { 0, 9 } },
		{ { 0, 9 }, { 0, 8 }, { 0, 7 }, { 0, 6 },  { 3, 7 }, { 3, 6 }, { 3, 5 } },
		{ { 0, 8 }, { 1, 5 }, { 0, 7 }, { 2, 5 },  { 0, 6 }, { 3, 6 }, { 3, 5 } },
		{ { 1, 6 }, { 1, 5 }, { 2, 6 }, { 2, 5 },  { 0, 6 }, { 0, 7 }, { 3, 5 } } };
	rec[0] = 0;

	f4 v = v4_clamp(0.0f, 65535.0f, mk4(rgbo.x + rgbo.w, rgbo.y + rgbo.w, rgbo.z + rgbo.w, rgbo.w));
	const int major = major_component(v);
	v = major_first(v, major);

	const int wr = widths[mode][0], wgb = widths[mode][1], ws = widths[mode][2];
	const float step = (float)(1 << (16 - wr)), rstep = 1.0f / step;
	if (v.x - v.y > (float)(1 << wgb) * step || v.x - v.z > (float)(1 << wgb) * step || v.w > (float)(1 << ws) * step) return;
	const int tag = mode < 4 ? (mode | (major << 2)) : (major | 0xC);

	int r = flt2int_rtn(v.x * rstep);
	const int byte0 = quant_keeping(t, (r & 0x3F) | ((tag & 3) << 6), 0xC0);
	r = (r & ~0x3F) | (byte0 & 0x3F);
	const float rf = (float)r * step;

	int g = flt2int_rtn(f_clamp(rf - v.y, 0.0f, 65535.0f) * rstep);
	int b = flt2int_rtn(f_clamp(rf - v.z, 0.0f, 65535.0f) * rstep);
	if (g >= (1 << wgb) || b >= (1 << wgb)) return;
	const SpareBit* sp = spare[mode];
	const int byte1 = quant_keeping(t, (g & 0x1F) | ((tag & 4) << 5) | (spare_bit(sp[0], r, g, b, 0, 0, 0) << 6) | (spare_bit(sp[1], r, g, b, 0, 0, 0) << 5), 0xF0);
	const int byte2 = quant_keeping(t, (b & 0x1F) | ((tag & 8) << 4) | (spare_bit(sp[2], r, g, b, 0, 0, 0) << 6) | (spare_bit(sp[3], r, g, b, 0, 0, 0) << 5), 0xF0);
	g = (g & ~0x1F) | (byte1 & 0x1F);
	b = (b & ~0x1F) | (byte2 & 0x1F);
	const float gf = (float)g * step, bf = (float)b * step;

	// the offset absorbs a third of the colour's coding error
	const float drift = (rf - v.x) + (rf - gf - v.y) + (rf - bf - v.z);
	const int sv = flt2int_rtn(f_clamp(v.w + drift * (1.0f / 3.0f), 0.0f, 1e9f) * rstep);
	if (sv >= (1 << ws)) return;
	const int byte3 = quant_keeping(t, (sv & 0x1F) | (spare_bit(sp[6], r, g, b, sv, 0, 0) << 5) | (spare_bit(sp[5], r, g, b, sv, 0, 0) << 6) |
	                                   (spare_bit(sp[4], r, g, b, sv, 0, 0) << 7), 0xF0);
	rec[1] = (uint8_t)byte0; rec[2] = (uint8_t)byte1; rec[3] = (uint8_t)byte2; rec[4] = (uint8_t)byte3;
	rec[0] = 1;
}

/* ... no sub-mode fits: four plain 7-bit fields (ref: :1179-1240). */
WV_FN void hdr_escape_rgbo(const ColorTabs& t, f4 rgbo, uint8_t* output)
{
	const f4 v = v4_clamp(0.0f, 65535.0f, mk4(rgbo.x + rgbo.w, rgbo.y + rgbo.w, rgbo.z + rgbo.w, rgbo.w));
	const float cr = f_clamp(v.x, 0.0f, 65020.0f), cg = f_clamp(v.y, 0.0f, 65020.0f), cb = f_clamp(v.z, 0.0f, 65020.0f);
	const int ir = flt2int_rtn(cr * (1.0f / 512.0f)), ig = flt2int_rtn(cg * (1.0f / 512.0f)), ib = flt2int_rtn(cb * (1.0f / 512.0f));
	const float drift = ((float)ir * 512.0f - cr) + ((float)ig * 512.0f - cg) + ((float)ib * 512.0f - cb);
	const int is = flt2int_rtn(f_clamp(v.w + drift * (1.0f / 3.0f), 0.0f, 65020.0f) * (1.0f / 512.0f));
	output[0] = (uint8_t)quant_keeping(t, (ir & 0x3F) | 0xC0, 0xF0);
	output[1] = (uint8_t)quant_keeping(t, (ig & 0x7F) | 0x80, 0xF0);
	output[2] = (uint8_t)quant_keeping(t, (ib & 0x7F) | 0x80, 0xF0);
	output[3] = (uint8_t)quant_keeping(t, (is & 0x7F) | ((ir & 0x40) << 1), 0xF0);
}

/* Direct RGB format, sub-mode `mode` 0..7: A = the major component of the high endpoint, B0 / B1 the other two as
 * differences from it, C the step down to the low endpoint, D0 / D1 what the low endpoint's minor components still
 * differ by.  (ref: quantize_hdr_rgb :1253, one iteration of its mode loop) */
WV_FN void hdr_try_rgb(const ColorTabs& t, f4 low, f4 high, int mode, uint8_t* rec)
{
	// widths of A, B, C, D per sub-mode
	const uint8_t widths[8][4] = { { 9, 7, 6, 7 }, { 9, 8, 6, 6 }, { 10, 6, 7, 7 }, { 10, 7, 7, 6 }, { 11, 8, 6, 5 }, { 11, 6, 8, 6 }, { 12, 7, 7, 5 }, { 12, 6, 7, 6 } };
	// spare bits 0..5 of bytes 2..5 (fields: 0 A, 1 B0, 2 B1, 3 C, 4 D0, 5 D1)
	const SpareBit spare[8][6] = {
		{ { 1, 6 }, { 2, 6 },  { 4, 6 },  { 5, 6 }, { 4, 5 }, { 5, 5 } },
		{ { 1, 6 }, { 2, 6 },  { 1, 7 },  { 2, 7 }, { 4, 5 }, { 5, 5 } },
		{ { 0, 9 }, { 3, 6 },  { 4, 6 },  { 5, 6 }, { 4, 5 }, { 5, 5 } },
		{ { 1, 6 }, { 2, 6 },  { 0, 9 },  { 3, 6 }, { 4, 5 }, { 5, 5 } },
		{ { 1, 6 }, { 2, 6 },  { 1, 7 },  { 2, 7 }, { 0, 9 }, { 0, 10 } },
		{ { 0, 9 }, { 0, 10 }, { 3, 7 },  { 3, 6 }, { 4, 5 }, { 5, 5 } },
		{ { 1, 6 }, { 2, 6 },  { 0, 11 }, { 3, 6 }, { 0, 9 }, { 0, 10 } },
		{ { 0, 9 }, { 0, 10 }, { 0, 11 }, { 3, 6 }, { 4, 5 }, { 5, 5 } } };
	rec[0] = 0;

	low = v4_clamp(0.0f, 65535.0f, low);
	high = v4_clamp(0.0f, 65535.0f, high);
	const int major = major_component(high);
	low = major_first(low, major);
	high = major_first(high, major);

	const int wa = widths[mode][0], wb = widths[mode][1], wc = widths[mode][2], wd = widths[mode][3];
	const float step = (float)(1 << (16 - wa)), rstep = 1.0f / step;
	{
		// coarse test on the real-valued fields
		const float fa = f_clamp(high.x, 0.0f, 65535.0f);
		const float fb0 = fa - high.y, fb1 = fa - high.z, fc = fa - low.x;
		const float fd0 = fa - fb0 - fc - low.y, fd1 = fa - fb1 - fc - low.z;
		const float bmax = (float)(1 << wb) * step, cmax = (float)(1 << wc) * step, dmax = (float)(1 << (wd - 1)) * step;
		if (fb0 > bmax || fb1 > bmax || fc > cmax || f_abs(fd0) > dmax || f_abs(fd1) > dmax) return;
	}

	int a = flt2int_rtn(f_clamp(high.x, 0.0f, 65535.0f) * rstep);
	const int byte0 = quant_color(t, a & 0xFF);
	a = (a & ~0xFF) | byte0;
	const float af = (float)a * step;

	int cv = flt2int_rtn(f_clamp(af - low.x, 0.0f, 65535.0f) * rstep);
	if (cv >= (1 << wc)) return;
	const int byte1 = quant_keeping(t, (cv & 0x3F) | ((mode & 1) << 7) | ((a & 0x100) >> 2), 0xC0);
	cv = (cv & ~0x3F) | (byte1 & 0x3F);
	const float cf = (float)cv * step;

	int b0 = flt2int_rtn(f_clamp(af - high.y, 0.0f, 65535.0f) * rstep);
	int b1 = flt2int_rtn(f_clamp(af - high.z, 0.0f, 65535.0f) * rstep);
	if (b0 >= (1 << wb) || b1 >= (1 << wb)) return;
	const SpareBit* sp = spare[mode];
	const int byte2 = quant_keeping(t, (b0 & 0x3F) | (spare_bit(sp[0], a, b0, b1, cv, 0, 0) << 6) | (((mode >> 1) & 1) << 7), 0xC0);
	const int byte3 = quant_keeping(t, (b1 & 0x3F) | (spare_bit(sp[1], a, b0, b1, cv, 0, 0) << 6) | (((mode >> 2) & 1) << 7), 0xC0);
	b0 = (b0 & ~0x3F) | (byte2 & 0x3F);
	b1 = (b1 & ~0x3F) | (byte3 & 0x3F);
	const float b0f = (float)b0 * step, b1f = (float)b1 * step;

	const int d0 = flt2int_rtn(f_clamp(af - b0f - cf - low.y, -65535.0f, 65535.0f) * rstep);
	const int d1 = flt2int_rtn(f_clamp(af - b1f - cf - low.z, -65535.0f, 65535.0f) * rstep);
	if ((d0 < 0 ? -d0 : d0) >= (1 << (wd - 1)) || (d1 < 0 ? -d1 : d1) >= (1 << (wd - 1))) return;
	const int byte4 = quant_keeping(t, (d0 & 0x1F) | (spare_bit(sp[2], a, b0, b1, cv, d0, d1) << 6) | (spare_bit(sp[4], a, b0, b1, cv, d0, d1) << 5) | ((major & 1) << 7), 0xF0);
	const int byte5 = quant_keeping(t, (d1 & 0x1F) | (spare_bit(sp[3], a, b0, b1, cv, d0, d1) << 6) | (spare_bit(sp[5], a, b0, b1, cv, d0, d1) << 5) | (((major >> 1) & 1) << 7), 0xF0);

	rec[1] = (uint8_t)byte0; rec[2] = (uint8_t)byte1; rec[3] = (uint8_t)byte2; rec[4] = (uint8_t)byte3; rec[5] = (uint8_t)byte4; rec[6] = (uint8_t)byte5;
	rec[0] = 1;
}

/* ... no sub-mode fits: both endpoints as plain 8 / 8 / 7-bit values, flagged by major component "3" (ref: :1571-1610). */
WV_FN void hdr_escape_rgb(const ColorTabs& t, f4 low, f4 high, uint8_t* output)
{
	low = v4_clamp(0.0f, 65020.0f, v4_clamp(0.0f, 65535.0f, low));
	high = v4_clamp(0.0f, 65020.0f, v4_clamp(0.0f, 65535.0f, high));
	output[0] = (uint8_t)quant_color(t, flt2int_rtn(low.x * 1.0f / 256.0f));
	output[1] = (uint8_t)quant_color(t, flt2int_rtn(high.x * 1.0f / 256.0f));
	output[2] = (uint8_t)quant_color(t, flt2int_rtn(low.y * 1.0f / 256.0f));
	output[3] = (uint8_t)quant_color(t, flt2int_rtn(high.y * 1.0f / 256.0f));
	output[4] = (uint8_t)quant_keeping(t, flt2int_rtn(low.z * 1.0f / 512.0f) + 128, 0xC0);
	output[5] = (uint8_t)quant_keeping(t, flt2int_rtn(high.z * 1.0f / 512.0f) + 128, 0xC0);
}

/* HDR alpha pair, sub-mode `fine` 2..0: a 7-bit base at (9 - fine)-bit precision and a signed offset of 6 - fine
 * bits (ref: quantize_hdr_alpha :1820, one iteration of its loop). */
WV_FN void hdr_try_alpha(const ColorTabs& t, float alpha0, float alpha1, int fine, uint8_t* rec)
{
	rec[0] = 0;
	const int a0 = flt2int_rtn(f_clamp(alpha0, 0.0f, 65280.0f)), a1 = flt2int_rtn(f_clamp(alpha1, 0.0f, 65280.0f));
	int base = (a0 + (128 >> fine)) >> (8 - fine);
	const int other = (a1 + (128 >> fine)) >> (8 - fine);
	const int byte0_in = (base & 0x7F) | ((fine & 1) << 7);
	const int byte0 = quant_color(t, byte0_in);
	if ((byte0_in ^ byte0) & 0x80) return;
	base = (base & ~0x7F) | (byte0 & 0x7F);
	const int reach = 32 >> fine;
	const int offset = other - base;
	if (offset < -reach || offset >= reach) return;
	const int byte1_in = ((fine & 2) << 6) | ((base >> 7) << (6 - fine)) | (offset & (2 * reach - 1));
	const int byte1 = quant_color(t, byte1_in);
	// everything but the offset's own bits must survive
	if ((byte1_in ^ byte1) & (0xFF & ~(reach - 1))) return;
	rec[1] = (uint8_t)byte0; rec[2] = (uint8_t)byte1;
	rec[0] = 1;
}

/* HDR luminance pair (ref: try_quantize_hdr_luminance_small_range :1718, quantize_hdr_luminance_large_range :1644):
 * returns the format used. */
WV_FN int hdr_code_luminance(const ColorTabs& t, f4 color0, f4 color1, uint8_t* output)
{
	float lum0 = hadd_rgb_s(color0) * (1.0f / 3.0f);
	float lum1 = hadd_rgb_s(color1) * (1.0f / 3.0f);
	if (lum1 < lum0)
	{
		const float avg = (lum0 + lum1) * 0.5f;
		lum0 = avg;
		lum1 = avg;
	}
	const int y0 = flt2int_rtn(lum0), y1 = flt2int_rtn(lum1);

	// small range: a base with `wb` bits and a `wo`-bit unsigned offset -- (11, 4) first, then (10, 5)
	if (y1 - y0 <= 2048)
	{
		for (int wo = 4; wo <= 5; wo++)
		{
			const int drop = wo + 1;                       // 16 -> 15 - wo bits
			int base = i_clamp((y0 + (1 << (drop - 1))) >> drop, 0, (1 << (15 - wo)) - 1);
			const int top = i_clamp((y1 + (1 << (drop - 1))) >> drop, 0, (1 << (15 - wo)) - 1);
			const int flag = wo == 5 ? 0x80 : 0;
			const int b0 = quant_color(t, (base & 0x7F) | flag);
			if ((b0 & 0x80) != flag) { if (wo == 5) break; continue; }
			base = (base & ~0x7F) | (b0 & 0x7F);
			const int offset = top - base;
			if (offset < 0 || offset >= (1 << wo)) { if (wo == 5) break; continue; }
			const int high_mask = 0xFF & ~((1 << wo) - 1);
			const int b1_in = ((base >> (7 - wo)) & high_mask) | offset;
			const int b1 = quant_color(t, b1_in);
			if ((b1 & high_mask) != (b1_in & high_mask)) { if (wo == 5) break; continue; }
			output[0] = (uint8_t)b0;
			output[1] = (uint8_t)b1;
			return FMT_HDR_LUMINANCE_SMALL_RANGE;
		}
	}

	// large range: two 8-bit values, either rounded (stored ascending) or offset by half a step (stored descending);
	// take the layout with the smaller squared error
	const int up0 = i_clamp((y0 + 128) >> 8, 0, 255), up1 = i_clamp((y1 + 128) >> 8, 0, 255);
	const int dn0 = i_clamp((y1 + 256) >> 8, 0, 255), dn1 = i_clamp(y0 >> 8, 0, 255);
	const int eu0 = (up0 << 8) - y0, eu1 = (up1 << 8) - y1;
	const int ed0 = ((dn1 << 8) + 128) - y0, ed1 = ((dn0 << 8) - 128) - y1;
	const bool rounded = eu0 * eu0 + eu1 * eu1 < ed0 * ed0 + ed1 * ed1;
	output[0] = (uint8_t)quant_color(t, rounded ? up0 : dn0);
	output[1] = (uint8_t)quant_color(t, rounded ? up1 : dn1);
	return FMT_HDR_LUMINANCE_LARGE_RANGE;
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
