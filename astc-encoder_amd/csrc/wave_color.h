// SPDX-License-Identifier: Apache-2.0
// Colour endpoint quantization (pack) and its inverse (unpack) for the LDR endpoint formats.
//   ref: quantize_* / try_quantize_* / pack_color_endpoints   Source/astcenc_color_quantize.cpp:53-839, :1909-2147
//        *_unpack / unpack_color_endpoints                    Source/astcenc_color_unquantize.cpp:35-301, :844-1023
// These are short, branchy, strictly scalar routines: one lane handles one partition's endpoint pair.
#pragma once
#include "wave_ctx.h"

namespace astcd { inline namespace ASTC_VARIANT {

struct ColorTabs {
	const uint8_t* unq_to_uq;   // [512] for the current quant level
};

/* The rows of the colour quant level staged in LDS by stage_color_rows(): pack_color_endpoints does ~50 dependent
 * lookups per call, and a pointer that could be LDS or HBM would make each of them a flat load with 64-bit address
 * arithmetic.  The caller stages the level it packs at (refine_candidate_setup; refine_pack for its retry level). */
WV_FN ColorTabs color_tabs(const Ctx& c, int quant_level)
{
	ColorTabs t;
#if !WV_DEVICE
	if (quant_level != c.tr().staged_color_quant[0]) __builtin_trap();
#endif
	(void)quant_level;
	t.unq_to_uq = c.lds + c.L->ctab;
	return t;
}

/* Stage the rows of a colour quant level. */
WV_FN void stage_color_rows(const Ctx& c, int q0)
{
	TrialInfo& tr = c.tr();
	const uint32_t* s0 = reinterpret_cast<const uint32_t*>(c.table(c.root->off_color_unquant_to_uquant) + (q0 - QUANT_6) * 512);
	stage_words_nosync(c.lds + c.L->ctab, reinterpret_cast<const uint8_t*>(s0), 128);
	WV_ONE { tr.staged_color_quant[0] = q0; tr.staged_color_quant[1] = -1; }
	WV_SYNC();
}

/* (ref: quant_color :72) */
WV_FN int quant_color(const ColorTabs& t, int value) { return t.unq_to_uq[value * 2 + 1]; }
/* (ref: quant_color :109) rounding direction follows the residual */
WV_FN int quant_color(const ColorTabs& t, int value, float valuef)
{
	int index = value * 2;
	float residual = valuef - (float)value;
	if (residual >= -0.1f) index++;
	return t.unq_to_uq[index];
}

} } // namespace astcd::ASTC_VARIANT
#include "wave_color_hdr.h"
namespace astcd { inline namespace ASTC_VARIANT {

WV_FN i4 quant_color3(const ColorTabs& t, i4 v) { return mki4(quant_color(t, v.x), quant_color(t, v.y), quant_color(t, v.z), 0); }
WV_FN i4 quant_color3(const ColorTabs& t, i4 v, f4 f) { return mki4(quant_color(t, v.x, f.x), quant_color(t, v.y, f.y), quant_color(t, v.z, f.z), 0); }

WV_FN i4 float_to_int_rtn4(f4 a) { return mki4((int)(a.x + 0.5f), (int)(a.y + 0.5f), (int)(a.z + 0.5f), (int)(a.w + 0.5f)); }
WV_FN int hadd_rgb_i(i4 a) { return a.x + a.y + a.z; }

/* (ref: bit_transfer_signed, vecmathlib_common_4.h:364) */
WV_FN void bit_transfer_signed1(int& input0, int& input1)
{
	input1 = (int)((unsigned)input1 >> 1) | (input0 & 0x80);
	input0 = (int)((unsigned)input0 >> 1) & 0x3F;
	if (input0 & 0x20) input0 -= 0x40;
}
WV_FN void bit_transfer_signed4(i4& a, i4& b)
{
	bit_transfer_signed1(a.x, b.x); bit_transfer_signed1(a.y, b.y);
	bit_transfer_signed1(a.z, b.z); bit_transfer_signed1(a.w, b.w);
}

/* (ref: uncontract_color, color_unquantize.cpp:35) */
WV_FN i4 uncontract_color(i4 in)
{
	return mki4((in.x + in.z) >> 1, (in.y + in.z) >> 1, in.z, in.w);
}

WV_FN i4 clamp_i4(int lo, int hi, i4 a)
{
	return mki4(i_min(i_max(a.x, lo), hi), i_min(i_max(a.y, lo), hi), i_min(i_max(a.z, lo), hi), i_min(i_max(a.w, lo), hi));
}

/* (ref: rgba_delta_unpack :61) */
WV_FN void rgba_delta_unpack(i4 input0, i4 input1, i4& output0, i4& output1)
{
	bit_transfer_signed4(input1, input0);
	int rgb_sum = hadd_rgb_i(input1);
	input1 = input1 + input0;
	if (rgb_sum < 0)
	{
		input0 = uncontract_color(input0);
		input1 = uncontract_color(input1);
		i4 t = input0; input0 = input1; input1 = t;
	}
	output0 = clamp_i4(0, 255, input0);
	output1 = clamp_i4(0, 255, input1);
}

/* (ref: rgba_unpack :105) */
WV_FN void rgba_unpack(i4 input0, i4 input1, i4& output0, i4& output1)
{
	if (hadd_rgb_i(input0) > hadd_rgb_i(input1))
	{
		input0 = uncontract_color(input0);
		input1 = uncontract_color(input1);
		i4 t = input0; input0 = input1; input1 = t;
	}
	output0 = input0;
	output1 = input1;
}

/* Decode one endpoint pair to 16-bit integer colours (LDR formats). (ref: unpack_color_endpoints :844)
 * HDR formats decode to the LDR error colour in LDR profiles, as in the reference. */
WV_FN void unpack_color_endpoints(int profile, int format, const uint8_t* in, i4& out0, i4& out1)
{
	bool rgb_hdr = false, alpha_hdr = false, alpha_hdr_default = false;
	switch (format)
	{
	case FMT_LUMINANCE:
		out0 = mki4(in[0], in[0], in[0], 255);
		out1 = mki4(in[1], in[1], in[1], 255);
		break;
	case FMT_LUMINANCE_DELTA:
		{
			int l0 = (in[0] >> 2) | (in[1] & 0xC0);
			int l1 = l0 + (in[1] & 0x3F);
			l1 = i_min(l1, 255);
			out0 = mki4(l0, l0, l0, 255);
			out1 = mki4(l1, l1, l1, 255);
		}
		break;
	case FMT_LUMINANCE_ALPHA:
		out0 = mki4(in[0], in[0], in[0], in[2]);
		out1 = mki4(in[1], in[1], in[1], in[3]);
		break;
	case FMT_LUMINANCE_ALPHA_DELTA:
		{
			int lum0 = in[0], lum1 = in[1], alpha0 = in[2], alpha1 = in[3];
			lum0 |= (lum1 & 0x80) << 1;
			alpha0 |= (alpha1 & 0x80) << 1;
			lum1 &= 0x7F;
			alpha1 &= 0x7F;
			if (lum1 & 0x40) lum1 -= 0x80;
			if (alpha1 & 0x40) alpha1 -= 0x80;
			lum0 >>= 1; lum1 >>= 1; alpha0 >>= 1; alpha1 >>= 1;
			lum1 += lum0;
			alpha1 += alpha0;
			lum1 = i_clamp(lum1, 0, 255);
			alpha1 = i_clamp(alpha1, 0, 255);
			out0 = mki4(lum0, lum0, lum0, alpha0);
			out1 = mki4(lum1, lum1, lum1, alpha1);
		}
		break;
	case FMT_RGB_SCALE:
		{
			int scale = in[3];
			out1 = mki4(in[0], in[1], in[2], 255);
			out0 = mki4((in[0] * scale) >> 8, (in[1] * scale) >> 8, (in[2] * scale) >> 8, 255);
		}
		break;
	case FMT_RGB_SCALE_ALPHA:
		{
			int scale = in[3];
			out1 = mki4(in[0], in[1], in[2], in[5]);
			out0 = mki4((in[0] * scale) >> 8, (in[1] * scale) >> 8, (in[2] * scale) >> 8, in[4]);
		}
		break;
	case FMT_RGB:
		rgba_unpack(mki4(in[0], in[2], in[4], 0), mki4(in[1], in[3], in[5], 0), out0, out1);
		out0.w = 255; out1.w = 255;
		break;
	case FMT_RGB_DELTA:
		rgba_delta_unpack(mki4(in[0], in[2], in[4], 0), mki4(in[1], in[3], in[5], 0), out0, out1);
		out0.w = 255; out1.w = 255;
		break;
	case FMT_RGBA:
		rgba_unpack(mki4(in[0], in[2], in[4], in[6]), mki4(in[1], in[3], in[5], in[7]), out0, out1);
		break;
	case FMT_RGBA_DELTA:
		rgba_delta_unpack(mki4(in[0], in[2], in[4], in[6]), mki4(in[1], in[3], in[5], in[7]), out0, out1);
		break;
#if ASTC_ENABLE_HDR
	case FMT_HDR_LUMINANCE_SMALL_RANGE:
		rgb_hdr = true; alpha_hdr_default = true;
		hdr_luminance_small_range_unpack(in, out0, out1);
		break;
	case FMT_HDR_LUMINANCE_LARGE_RANGE:
		rgb_hdr = true; alpha_hdr_default = true;
		hdr_luminance_large_range_unpack(in, out0, out1);
		break;
	case FMT_HDR_RGB_SCALE:
		rgb_hdr = true; alpha_hdr_default = true;
		hdr_rgbo_unpack(in, out0, out1);
		break;
	case FMT_HDR_RGB:
		rgb_hdr = true; alpha_hdr_default = true;
		hdr_rgb_unpack(in, out0, out1);
		break;
	case FMT_HDR_RGB_LDR_ALPHA:
		rgb_hdr = true;
		hdr_rgb_unpack(in, out0, out1);
		out0.w = in[6]; out1.w = in[7];
		break;
	default: // FMT_HDR_RGBA
		{
			rgb_hdr = true; alpha_hdr = true;
			hdr_rgb_unpack(in, out0, out1);
			int a0, a1;
			hdr_alpha_unpack(in + 6, a0, a1);
			out0.w = a0; out1.w = a1;
		}
		break;
#else
	default:
		// LDR-only kernel: HDR formats never reach the encoder's decode loops
		rgb_hdr = true; alpha_hdr_default = true;
		out0 = mki4(0, 0, 0, 0); out1 = mki4(0, 0, 0, 0);
		break;
#endif
	}

	// formats without their own alpha take the profile's default (ref: color_unquantize.cpp:963-977)
	if (alpha_hdr_default)
	{
		if (profile == 3 /* HDR */)
		{
			out0.w = 0x7800; out1.w = 0x7800;
			alpha_hdr = true;
		}
		else
		{
			out0.w = 0x00FF; out1.w = 0x00FF;
			alpha_hdr = false;
		}
	}

	if (profile == 1 /* LDR */)
	{
		// an HDR endpoint format in an LDR profile decodes to the error colour
		if (rgb_hdr || alpha_hdr)
		{
			out0 = mki4(0xFF, 0x00, 0xFF, 0xFF);
			out1 = mki4(0xFF, 0x00, 0xFF, 0xFF);
		}
		out0 = out0 * 257;
		out1 = out1 * 257;
	}
	else if (profile == 0 /* LDR_SRGB */)
	{
		if (rgb_hdr || alpha_hdr)
		{
			out0 = mki4(0xFF, 0x00, 0xFF, 0xFF);
			out1 = mki4(0xFF, 0x00, 0xFF, 0xFF);
		}
		out0 = mki4((out0.x << 8) | 0x80, (out0.y << 8) | 0x80, (out0.z << 8) | 0x80, (out0.w << 8) | 0x80);
		out1 = mki4((out1.x << 8) | 0x80, (out1.y << 8) | 0x80, (out1.z << 8) | 0x80, (out1.w << 8) | 0x80);
	}
	else
	{
		// HDR profiles: LDR lanes expand 8 -> 16 bit, HDR lanes are already 16-bit LNS codes
		int sr = rgb_hdr ? 1 : 257, sa = alpha_hdr ? 1 : 257;
		out0 = mki4(out0.x * sr, out0.y * sr, out0.z * sr, out0.w * sa);
		out1 = mki4(out1.x * sr, out1.y * sr, out1.z * sr, out1.w * sa);
	}
}

// ---------------------------------------------------------------------------------------------
// Quantizers
// ---------------------------------------------------------------------------------------------

/* (ref: get_rgba_encoding_error :53) */
WV_FN float rgba_encoding_error(f4 uquant0, f4 uquant1, i4 quant0, i4 quant1)
{
	f4 error0 = uquant0 - int_to_float4(quant0);
	f4 error1 = uquant1 - int_to_float4(quant1);
	return hadd_s(error0 * error0 + error1 * error1);
}

/* (ref: quantize_rgb :169) */
WV_FN void quantize_rgb(const ColorTabs& t, f4 color0, f4 color1, i4& color0_out, i4& color1_out)
{
	i4 color0i, color1i;
	f4 nudge = splat4(0.2f);
	do
	{
		i4 q0 = float_to_int_rtn4(color0);
		q0 = mki4(i_max(q0.x, 0), i_max(q0.y, 0), i_max(q0.z, 0), i_max(q0.w, 0));
		color0i = quant_color3(t, q0, color0);
		color0 = color0 - nudge;

		i4 q1 = float_to_int_rtn4(color1);
		q1 = mki4(i_min(q1.x, 255), i_min(q1.y, 255), i_min(q1.z, 255), i_min(q1.w, 255));
		color1i = quant_color3(t, q1, color1);
		color1 = color1 + nudge;
	} while (hadd_rgb_i(color0i) > hadd_rgb_i(color1i));
	color0_out = color0i;
	color1_out = color1i;
}

/* (ref: quantize_rgba :208) */
WV_FN void quantize_rgba(const ColorTabs& t, f4 color0, f4 color1, i4& color0_out, i4& color1_out)
{
	quantize_rgb(t, color0, color1, color0_out, color1_out);
	color0_out.w = quant_color(t, flt2int_rtn(color0.w), color0.w);
	color1_out.w = quant_color(t, flt2int_rtn(color1.w), color1.w);
}

WV_FN bool out_of_byte_range(f4 a)
{
	return (a.x < 0.0f) || (a.x > 255.0f) || (a.y < 0.0f) || (a.y > 255.0f) ||
	       (a.z < 0.0f) || (a.z > 255.0f) || (a.w < 0.0f) || (a.w > 255.0f);
}

/* (ref: try_quantize_rgb_blue_contract :237) */
WV_FN bool try_quantize_rgb_blue_contract(const ColorTabs& t, f4 color0, f4 color1, i4& color0_out, i4& color1_out)
{
	color0 = color0 + (color0 - mk4(color0.z, color0.z, color0.z, color0.w));
	color1 = color1 + (color1 - mk4(color1.z, color1.z, color1.z, color1.w));
	if (out_of_byte_range(color0) || out_of_byte_range(color1)) return false;

	i4 color0i = quant_color3(t, float_to_int_rtn4(color0), color0);
	i4 color1i = quant_color3(t, float_to_int_rtn4(color1), color1);
	if (hadd_rgb_i(color1i) <= hadd_rgb_i(color0i)) return false;

	color0_out = color1i;
	color1_out = color0i;
	return true;
}

/* (ref: try_quantize_rgba_blue_contract :285) */
WV_FN bool try_quantize_rgba_blue_contract(const ColorTabs& t, f4 color0, f4 color1, i4& color0_out, i4& color1_out)
{
	if (try_quantize_rgb_blue_contract(t, color0, color1, color0_out, color1_out))
	{
		float a0 = color0.w, a1 = color1.w;
		color0_out.w = quant_color(t, flt2int_rtn(a1), a1);
		color1_out.w = quant_color(t, flt2int_rtn(a0), a0);
		return true;
	}
	return false;
}

/* Shared tail of the two RGB delta encoders (ref: :329-387, :426-484). want_negative_sum selects the
 * blue-contracted flavour. */
WV_FN bool quantize_rgb_delta_core(const ColorTabs& t, f4 color0, f4 color1, bool want_negative_sum, i4& color0_out, i4& color1_out)
{
	i4 color0a = float_to_int_rtn4(color0);
	color0a = mki4(color0a.x << 1, color0a.y << 1, color0a.z << 1, color0a.w << 1);
	i4 color0b = mki4(color0a.x & 0xFF, color0a.y & 0xFF, color0a.z & 0xFF, color0a.w & 0xFF);
	i4 color0be = quant_color3(t, color0b);
	color0b = mki4(color0be.x | (color0a.x & 0x100), color0be.y | (color0a.y & 0x100), color0be.z | (color0a.z & 0x100), color0be.w | (color0a.w & 0x100));

	i4 color1d = float_to_int_rtn4(color1);
	color1d = mki4(color1d.x << 1, color1d.y << 1, color1d.z << 1, 0);
	color1d = mki4(color1d.x - color0b.x, color1d.y - color0b.y, color1d.z - color0b.z, 0);

	if (color1d.x > 63 || color1d.x < -64 || color1d.y > 63 || color1d.y < -64 || color1d.z > 63 || color1d.z < -64)
	{
		return false;
	}

	color1d = mki4((color1d.x & 0x7F) | ((color0b.x & 0x100) >> 1),
	               (color1d.y & 0x7F) | ((color0b.y & 0x100) >> 1),
	               (color1d.z & 0x7F) | ((color0b.z & 0x100) >> 1), 0);

	i4 color1de = quant_color3(t, color1d);
	if (((color1d.x ^ color1de.x) & 0xC0) || ((color1d.y ^ color1de.y) & 0xC0) || ((color1d.z ^ color1de.z) & 0xC0))
	{
		return false;
	}

	i4 ep0 = color0be;
	i4 ep1 = color1de;
	ep0.w = 0; ep1.w = 0;
	bit_transfer_signed4(ep1, ep0);
	int s = hadd_rgb_i(ep1);
	if (want_negative_sum ? (s >= 0) : (s < 0)) return false;

	ep0 = ep0 + ep1;
	if (ep0.x < 0 || ep0.x > 0xFF || ep0.y < 0 || ep0.y > 0xFF || ep0.z < 0 || ep0.z > 0xFF) return false;

	color0_out = color0be;
	color1_out = color1de;
	return true;
}

/* (ref: try_quantize_rgb_delta :321) */
WV_FN bool try_quantize_rgb_delta(const ColorTabs& t, f4 color0, f4 color1, i4& color0_out, i4& color1_out)
{
	return quantize_rgb_delta_core(t, color0, color1, false, color0_out, color1_out);
}

/* (ref: try_quantize_rgb_delta_blue_contract :403) */
WV_FN bool try_quantize_rgb_delta_blue_contract(const ColorTabs& t, f4 color0, f4 color1, i4& color0_out, i4& color1_out)
{
	f4 tmp = color0; color0 = color1; color1 = tmp;
	color0 = color0 + (color0 - mk4(color0.z, color0.z, color0.z, color0.w));
	color1 = color1 + (color1 - mk4(color1.z, color1.z, color1.z, color1.w));
	if (out_of_byte_range(color0) || out_of_byte_range(color1)) return false;
	return quantize_rgb_delta_core(t, color0, color1, true, color0_out, color1_out);
}

/* (ref: try_quantize_alpha_delta :504) */
WV_FN bool try_quantize_alpha_delta(const ColorTabs& t, f4 color0, f4 color1, i4& color0_out, i4& color1_out)
{
	float a0 = color0.w, a1 = color1.w;
	int a0a = flt2int_rtn(a0);
	a0a <<= 1;
	int a0b = a0a & 0xFF;
	int a0be = quant_color(t, a0b);
	a0b = a0be;
	a0b |= a0a & 0x100;
	int a1d = flt2int_rtn(a1);
	a1d <<= 1;
	a1d -= a0b;
	if (a1d > 63 || a1d < -64) return false;
	a1d &= 0x7F;
	a1d |= (a0b & 0x100) >> 1;
	int a1de = quant_color(t, a1d);
	int a1du = a1de;
	if ((a1d ^ a1du) & 0xC0) return false;
	a1du &= 0x7F;
	if (a1du & 0x40) a1du -= 0x80;
	a1du += a0b;
	if (a1du < 0 || a1du > 0x1FF) return false;
	color0_out.w = a0be;
	color1_out.w = a1de;
	return true;
}

/* (ref: try_quantize_luminance_alpha_delta :573) */
WV_FN bool try_quantize_luminance_alpha_delta(const ColorTabs& t, f4 color0, f4 color1, uint8_t* output)
{
	float l0 = hadd_rgb_s(color0) * (1.0f / 3.0f);
	float l1 = hadd_rgb_s(color1) * (1.0f / 3.0f);
	float a0 = color0.w, a1 = color1.w;

	int l0a = flt2int_rtn(l0), a0a = flt2int_rtn(a0);
	l0a <<= 1; a0a <<= 1;
	int l0b = l0a & 0xFF, a0b = a0a & 0xFF;
	int l0be = quant_color(t, l0b), a0be = quant_color(t, a0b);
	l0b = l0be; a0b = a0be;
	l0b |= l0a & 0x100; a0b |= a0a & 0x100;

	int l1d = flt2int_rtn(l1), a1d = flt2int_rtn(a1);
	l1d <<= 1; a1d <<= 1;
	l1d -= l0b; a1d -= a0b;
	if (l1d > 63 || l1d < -64) return false;
	if (a1d > 63 || a1d < -64) return false;

	l1d &= 0x7F; a1d &= 0x7F;
	l1d |= (l0b & 0x100) >> 1;
	a1d |= (a0b & 0x100) >> 1;

	int l1de = quant_color(t, l1d), a1de = quant_color(t, a1d);
	int l1du = l1de, a1du = a1de;
	if ((l1d ^ l1du) & 0xC0) return false;
	if ((a1d ^ a1du) & 0xC0) return false;

	l1du &= 0x7F; a1du &= 0x7F;
	if (l1du & 0x40) l1du -= 0x80;
	if (a1du & 0x40) a1du -= 0x80;
	l1du += l0b; a1du += a0b;
	if (l1du < 0 || l1du > 0x1FF) return false;
	if (a1du < 0 || a1du > 0x1FF) return false;

	output[0] = (uint8_t)l0be; output[1] = (uint8_t)l1de;
	output[2] = (uint8_t)a0be; output[3] = (uint8_t)a1de;
	return true;
}

/* (ref: quantize_rgbs :734) */
WV_FN void quantize_rgbs(const ColorTabs& t, f4 color, uint8_t* output)
{
	float scale = 1.0f / 257.0f;
	float r = f_clamp255(color.x * scale);
	float g = f_clamp255(color.y * scale);
	float b = f_clamp255(color.z * scale);

	int ri = quant_color(t, flt2int_rtn(r), r);
	int gi = quant_color(t, flt2int_rtn(g), g);
	int bi = quant_color(t, flt2int_rtn(b), b);

	float oldcolorsum = hadd_rgb_s(color) * scale;
	float newcolorsum = (float)(ri + gi + bi);

	float scalea = f_clamp1(color.w * (oldcolorsum + 1e-10f) / (newcolorsum + 1e-10f));
	int scale_idx = flt2int_rtn(scalea * 256.0f);
	scale_idx = i_clamp(scale_idx, 0, 255);

	output[0] = (uint8_t)ri; output[1] = (uint8_t)gi; output[2] = (uint8_t)bi;
	output[3] = (uint8_t)quant_color(t, scale_idx);
}

/* (ref: pack_color_endpoints :1909).  Returns the format actually used. */
WV_FN int pack_color_endpoints(const Ctx& c, f4 color0, f4 color1, f4 rgbs_color, f4 rgbo_color, int format, uint8_t* output, int quant_level)
{
	(void)rgbo_color;
	ColorTabs t = color_tabs(c, quant_level);

	color0 = v4_clamp(0.0f, 65535.0f, color0);
	color1 = v4_clamp(0.0f, 65535.0f, color1);
	f4 color0_ldr = color0 * (1.0f / 257.0f);
	f4 color1_ldr = color1 * (1.0f / 257.0f);

	int retval = 0;
	float best_error = ERROR_CALC_DEFAULT;
	i4 c0 = mki4(0, 0, 0, 0), c1 = mki4(0, 0, 0, 0), c0b = c0, c1b = c1, u0, u1;

	switch (format)
	{
	case FMT_RGB:
	case FMT_RGBA:
		{
			const bool al = format == FMT_RGBA;
			const int fmt_delta = al ? FMT_RGBA_DELTA : FMT_RGB_DELTA;
			if (quant_level <= QUANT_160)
			{
				bool ok = try_quantize_rgb_delta_blue_contract(t, color0_ldr, color1_ldr, c0, c1);
				if (ok && al) ok = try_quantize_alpha_delta(t, color1_ldr, color0_ldr, c0, c1);
				if (ok)
				{
					rgba_delta_unpack(c0, c1, u0, u1);
					retval = fmt_delta;
					best_error = rgba_encoding_error(color0_ldr, color1_ldr, u0, u1);
				}

				ok = try_quantize_rgb_delta(t, color0_ldr, color1_ldr, c0b, c1b);
				if (ok && al) ok = try_quantize_alpha_delta(t, color0_ldr, color1_ldr, c0b, c1b);
				if (ok)
				{
					rgba_delta_unpack(c0b, c1b, u0, u1);
					float error = rgba_encoding_error(color0_ldr, color1_ldr, u0, u1);
					if (error < best_error)
					{
						retval = fmt_delta;
						best_error = error;
						c0 = c0b; c1 = c1b;
					}
				}
			}

			if (quant_level < QUANT_256)
			{
				bool ok = al ? try_quantize_rgba_blue_contract(t, color0_ldr, color1_ldr, c0b, c1b)
				             : try_quantize_rgb_blue_contract(t, color0_ldr, color1_ldr, c0b, c1b);
				if (ok)
				{
					rgba_unpack(c0b, c1b, u0, u1);
					float error = rgba_encoding_error(color0_ldr, color1_ldr, u0, u1);
					if (error < best_error)
					{
						retval = format;
						best_error = error;
						c0 = c0b; c1 = c1b;
					}
				}
			}

			{
				if (al) quantize_rgba(t, color0_ldr, color1_ldr, c0b, c1b);
				else quantize_rgb(t, color0_ldr, color1_ldr, c0b, c1b);
				rgba_unpack(c0b, c1b, u0, u1);
				float error = rgba_encoding_error(color0_ldr, color1_ldr, u0, u1);
				if (error < best_error)
				{
					retval = format;
					c0 = c0b; c1 = c1b;
				}
			}

			output[0] = (uint8_t)c0.x; output[1] = (uint8_t)c1.x;
			output[2] = (uint8_t)c0.y; output[3] = (uint8_t)c1.y;
			output[4] = (uint8_t)c0.z; output[5] = (uint8_t)c1.z;
			if (al) { output[6] = (uint8_t)c0.w; output[7] = (uint8_t)c1.w; }
		}
		break;

	case FMT_RGB_SCALE:
		quantize_rgbs(t, rgbs_color, output);
		retval = FMT_RGB_SCALE;
		break;

	case FMT_RGB_SCALE_ALPHA:
		{
			float a0 = color0_ldr.w, a1 = color1_ldr.w;
			output[4] = (uint8_t)quant_color(t, flt2int_rtn(a0), a0);
			output[5] = (uint8_t)quant_color(t, flt2int_rtn(a1), a1);
			quantize_rgbs(t, rgbs_color, output);
			retval = FMT_RGB_SCALE_ALPHA;
		}
		break;

	case FMT_LUMINANCE:
		{
			float lum0 = hadd_rgb_s(color0_ldr) * (1.0f / 3.0f);
			float lum1 = hadd_rgb_s(color1_ldr) * (1.0f / 3.0f);
			if (lum0 > lum1)
			{
				float avg = (lum0 + lum1) * 0.5f;
				lum0 = avg;
				lum1 = avg;
			}
			output[0] = (uint8_t)quant_color(t, flt2int_rtn(lum0), lum0);
			output[1] = (uint8_t)quant_color(t, flt2int_rtn(lum1), lum1);
			retval = FMT_LUMINANCE;
		}
		break;

	case FMT_LUMINANCE_ALPHA:
		{
			if (quant_level <= 18)
			{
				if (try_quantize_luminance_alpha_delta(t, color0_ldr, color1_ldr, output))
				{
					retval = FMT_LUMINANCE_ALPHA_DELTA;
					break;
				}
			}
			float lum0 = hadd_rgb_s(color0_ldr) * (1.0f / 3.0f);
			float lum1 = hadd_rgb_s(color1_ldr) * (1.0f / 3.0f);
			float a0 = color0_ldr.w, a1 = color1_ldr.w;
			output[0] = (uint8_t)quant_color(t, flt2int_rtn(lum0), lum0);
			output[1] = (uint8_t)quant_color(t, flt2int_rtn(lum1), lum1);
			output[2] = (uint8_t)quant_color(t, flt2int_rtn(a0), a0);
			output[3] = (uint8_t)quant_color(t, flt2int_rtn(a1), a1);
			retval = FMT_LUMINANCE_ALPHA;
		}
		break;

#if ASTC_ENABLE_HDR
	case FMT_HDR_RGB_SCALE:
		quantize_hdr_rgbo(t, rgbo_color, output);
		retval = FMT_HDR_RGB_SCALE;
		break;

	case FMT_HDR_RGB:
		quantize_hdr_rgb(t, color0, color1, output);
		retval = FMT_HDR_RGB;
		break;

	case FMT_HDR_LUMINANCE_SMALL_RANGE:
	case FMT_HDR_LUMINANCE_LARGE_RANGE:
		if (try_quantize_hdr_luminance_small_range(t, color0, color1, output))
		{
			retval = FMT_HDR_LUMINANCE_SMALL_RANGE;
			break;
		}
		quantize_hdr_luminance_large_range(t, color0, color1, output);
		retval = FMT_HDR_LUMINANCE_LARGE_RANGE;
		break;

	case FMT_HDR_RGB_LDR_ALPHA:
		{
			float scale = 1.0f / 257.0f;
			float a0 = f_clamp255(color0.w * scale);
			float a1 = f_clamp255(color1.w * scale);
			output[6] = (uint8_t)quant_color(t, flt2int_rtn(a0), a0);
			output[7] = (uint8_t)quant_color(t, flt2int_rtn(a1), a1);
			quantize_hdr_rgb(t, color0, color1, output);
			retval = FMT_HDR_RGB_LDR_ALPHA;
		}
		break;

	case FMT_HDR_RGBA:
		quantize_hdr_rgb(t, color0, color1, output);
		quantize_hdr_alpha(t, color0.w, color1.w, output + 6);
		retval = FMT_HDR_RGBA;
		break;
#endif

	default:
		retval = format;
		break;
	}

	return retval;
}

} } // namespace astcd::ASTC_VARIANT
