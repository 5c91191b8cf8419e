// SPDX-License-Identifier: Apache-2.0
// Colour endpoint quantization (pack) and its inverse (unpack) for the LDR endpoint formats.
//   ref: quantize_* / try_quantize_* / pack_color_endpoints   Source/astcenc_color_quantize.cpp:53-839, :1909-2147
//        *_unpack / unpack_color_endpoints                    Source/astcenc_color_unquantize.cpp:35-301, :844-1023
// The quantizers are quad code (wave_quad.h): four lanes per partition, one colour channel each.  The decoder of an
// endpoint pair (unpack_color_endpoints, shared with the decompression kernel) is scalar code on one lane.
#pragma once
#include "wave_ctx.h"
#include "wave_quad.h"

namespace astcd { inline namespace ASTC_VARIANT {

struct ColorTabs {
	const uint8_t* unq_to_uq;   // [512] for the current quant level
};

/* The rows of the colour quant level staged in LDS by stage_color_rows(): pack_color_endpoints does ~50 dependent
 * lookups per call, and a pointer that could be LDS or HBM would make each of them a flat load with 64-bit address
 * arithmetic.  The caller stages the level it packs at (refine_candidate_restore; refine_pack for its retry level). */
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
	// (the rows of this level are there already: the previous candidate of the trial used the same one)
	if (wv_uniform(tr.staged_color_quant[0]) == q0) { WV_SYNC(); return; }
	const uint32_t* s0 = reinterpret_cast<const uint32_t*>(c.table(c.root->off_color_unquant_to_uquant) + (q0 - QUANT_6) * 512);
	stage_quads_nosync(c.lds + c.L->ctab, reinterpret_cast<const uint8_t*>(s0), 512);
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
// Quantizers: quad code (wave_quad.h) -- partition p of the block on lanes 4p .. 4p+3, one colour channel per lane
// ---------------------------------------------------------------------------------------------

WV_FN qi q_quant(const ColorTabs& t, qi value) { return q_mapi(value, [&t](int v) { return quant_color(t, v); }); }
WV_FN qi q_quant(const ColorTabs& t, qi value, qf valuef) { return q_zip_if(value, valuef, [&t](int v, float f) { return quant_color(t, v, f); }); }

/* Two decoded or encoded endpoints. */
struct QPair { qi e0, e1; };

/* The blue-contracted colour: R and G move away from B, B and A stay (ref: try_quantize_rgb_blue_contract :248-252). */
WV_FN qf q_blue_contract(qf colour)
{
	const float blue = q_get<2>(colour);
	return colour + q_map_ch(colour, [blue](int ch, float x) { return x - (ch == 3 ? x : blue); });
}

/* (ref: uncontract_color, color_unquantize.cpp:35) */
WV_FN qi q_uncontract(qi in)
{
	const int blue = q_get<2>(in);
	return q_mapi_ch(in, [blue](int ch, int x) { return ch < 2 ? (x + blue) >> 1 : x; });
}

/* Direct endpoints -> decoded pair: a pair stored in descending RGB-sum order means blue contraction
 * (ref: rgba_unpack, color_unquantize.cpp:105). */
WV_FN QPair q_decode_direct(QPair in)
{
	QPair out = in;
	if (q_sum_rgb(in.e0) > q_sum_rgb(in.e1))
	{
		out.e0 = q_uncontract(in.e1);
		out.e1 = q_uncontract(in.e0);
	}
	return out;
}

/* Base + offset endpoints -> decoded pair (ref: rgba_delta_unpack, color_unquantize.cpp:61 with bit_transfer_signed). */
WV_FN QPair q_decode_delta(QPair in)
{
	// the offset's top bit is the base's ninth bit; what is left of the offset is a 6-bit signed number
	qi base = (in.e0 >> 1) | (in.e1 & 0x80);
	qi offset = q_mapi((in.e1 >> 1) & 0x3F, [](int x) { return (x & 0x20) ? x - 0x40 : x; });
	const int offset_sum = q_sum_rgb(offset);
	qi other = offset + base;
	if (offset_sum < 0)
	{
		const qi t = q_uncontract(base);
		base = q_uncontract(other);
		other = t;
	}
	QPair out;
	out.e0 = q_clamp(0, 255, base);
	out.e1 = q_clamp(0, 255, other);
	return out;
}

/* Squared distance of a decoded pair to the real-valued endpoints (ref: get_rgba_encoding_error :53). */
WV_FN float q_encoding_error(qf colour0, qf colour1, QPair decoded)
{
	const qf d0 = colour0 - q_to_float(decoded.e0);
	const qf d1 = colour1 - q_to_float(decoded.e1);
	return q_hadd(d0 * d0 + d1 * d1);
}

/* Base + offset encoding of (colour0, colour1): 8.1-bit base, 7-bit signed offset.  `alpha`: the alpha lane takes part
 * (its own validity rule; ref: try_quantize_alpha_delta :504), otherwise it computes along and is ignored.
 * `contracted`: the decoder must see a negative offset sum, i.e. the pair is the swapped, blue-contracted one; else a
 * non-negative sum.  (ref: try_quantize_rgb_delta :321, try_quantize_rgb_delta_blue_contract :403) */
WV_FN bool q_encode_delta(const ColorTabs& t, qf colour0, qf colour1, bool contracted, bool alpha, QPair& out)
{
	auto any = [alpha](qb test) { return alpha ? q_any(test) : q_any_rgb(test); };

	const qi base9 = q_round_to_int(colour0) << 1;
	const qi base_q = q_quant(t, base9 & 0xFF);
	const qi base = base_q | (base9 & 0x100);

	qi offset = (q_round_to_int(colour1) << 1) - base;
	if (any(q_test(offset, [](int x) { return x > 63 || x < -64; }))) return false;
	offset = (offset & 0x7F) | ((base & 0x100) >> 1);
	const qi offset_q = q_quant(t, offset);
	// quantization must not touch the sign bit nor the base's ninth bit
	if (any(q_test((offset ^ offset_q) & 0xC0, [](int x) { return x != 0; }))) return false;

	// what a decoder makes of it: colour lanes must stay within a byte and show the right contraction flag; the alpha
	// lane, decoded at 9 bits, must stay within 0 .. 0x1FF
	QPair in; in.e0 = base_q; in.e1 = offset_q;
	const qi dec_base = (base_q >> 1) | (offset_q & 0x80);
	const qi dec_offset = q_mapi((offset_q >> 1) & 0x3F, [](int x) { return (x & 0x20) ? x - 0x40 : x; });
	const int offset_sum = q_sum_rgb(dec_offset);
	if (contracted ? offset_sum >= 0 : offset_sum < 0) return false;
	if (q_any_rgb_outside(dec_base + dec_offset, 0, 0xFF)) return false;
	if (alpha)
	{
		const qi alpha9 = q_mapi(offset_q & 0x7F, [](int x) { return (x & 0x40) ? x - 0x80 : x; }) + base;
		if (q_get<3>(q_test(alpha9, [](int x) { return x < 0 || x > 0x1FF; }))) return false;
	}
	out = in;
	return true;
}

/* Blue-contracted direct encoding of the (already contracted, range-checked) pair: stored in descending order
 * (ref: try_quantize_rgb[a]_blue_contract :237, :285). */
WV_FN bool q_encode_contracted(const ColorTabs& t, qf contracted0, qf contracted1, QPair& out)
{
	const qi q0 = q_quant(t, q_round_to_int(contracted0), contracted0);
	const qi q1 = q_quant(t, q_round_to_int(contracted1), contracted1);
	if (q_sum_rgb(q1) <= q_sum_rgb(q0)) return false;
	out.e0 = q1;
	out.e1 = q0;
	return true;
}

/* Plain direct encoding; the colour lanes are nudged apart until they are stored in ascending order, alpha is
 * quantized as it is (ref: quantize_rgb :169, quantize_rgba :208). */
WV_FN QPair q_encode_direct(const ColorTabs& t, qf colour0, qf colour1)
{
	// The first round quantizes all four lanes of the values as they are (they lie within 0 .. 255, so the clamps of
	// the rounded index do nothing yet): that is the alpha lane's result, whatever the colour lanes go on to do.
	QPair first;
	first.e0 = q_quant(t, q_round_to_int(colour0), colour0);
	first.e1 = q_quant(t, q_round_to_int(colour1), colour1);
	QPair out = first;
	while (q_sum_rgb(out.e0) > q_sum_rgb(out.e1))
	{
		colour0 = colour0 - 0.2f;
		colour1 = colour1 + 0.2f;
		out.e0 = q_quant(t, q_mapi(q_round_to_int(colour0), [](int x) { return i_max(x, 0); }), colour0);
		out.e1 = q_quant(t, q_mapi(q_round_to_int(colour1), [](int x) { return i_min(x, 255); }), colour1);
	}
	out.e0 = q_with_w(out.e0, first.e0);
	out.e1 = q_with_w(out.e1, first.e1);
	return out;
}

/* RGB + scale (ref: quantize_rgbs :734): components 0..2 = colour, 3 = scale -> four bytes */
WV_FN qi q_encode_rgb_scale(const ColorTabs& t, qf rgbs)
{
	const qf c = q_map(rgbs * (1.0f / 257.0f), [](float x) { return f_clamp255(x); });
	const qi cq = q_quant(t, q_round_to_int(c), c);
	const float old_sum = q_hadd_rgb(rgbs) * (1.0f / 257.0f);
	const float new_sum = (float)q_sum_rgb(cq);
	const float scale = f_clamp1(q_get<3>(rgbs) * (old_sum + 1e-10f) / (new_sum + 1e-10f));
	const int scale_index = i_clamp(flt2int_rtn(scale * 256.0f), 0, 255);
	return q_with_w(cq, quant_color(t, scale_index));
}

/* What pack_endpoints_quad found. */
struct QPacked {
	int format;          // the format actually used (same value on the quad's lanes)
	bool decoded_valid;  // `decoded` holds the 8-bit endpoints a decoder reconstructs (the direct and base + offset formats)
	QPair decoded;
};

/* One partition's endpoints -> bytes, LDR formats (ref: pack_color_endpoints :1909; the HDR formats are coded by
 * pack_endpoints_hdr in wave_color_hdr.h).  colour0 / colour1 are the real-valued endpoints (0 .. 65535), rgbs the
 * RGB + scale vector.  output: 8 bytes. */
WV_FN QPacked pack_endpoints_quad(const ColorTabs& t, qf colour0, qf colour1, qf rgbs, int format, uint8_t* output, int quant_level)
{
	QPacked r;
	r.format = format;
	r.decoded_valid = false;
	r.decoded.e0 = q_splat(0); r.decoded.e1 = q_splat(0);

	colour0 = q_clamp(0.0f, 65535.0f, colour0) * (1.0f / 257.0f);
	colour1 = q_clamp(0.0f, 65535.0f, colour1) * (1.0f / 257.0f);

	if (format == FMT_RGB || format == FMT_RGBA)
	{
		// Up to four encodings in the reference's order of preference; a later one replaces an earlier one only if its
		// decoded endpoints are strictly closer (ref: :1936-1999, :2010-2073).
		const bool alpha = format == FMT_RGBA;
		const int format_delta = alpha ? FMT_RGBA_DELTA : FMT_RGB_DELTA;
		float best_error = ERROR_CALC_DEFAULT;
		QPair best, trial;
		best.e0 = q_splat(0); best.e1 = q_splat(0);
		auto consider = [&](QPair enc, bool delta) {
			if (!alpha) { enc.e0 = q_with_w(enc.e0, 0); enc.e1 = q_with_w(enc.e1, 0); }
			const QPair dec = delta ? q_decode_delta(enc) : q_decode_direct(enc);
			const float error = q_encoding_error(colour0, colour1, dec);
			if (error < best_error)
			{
				best_error = error;
				best = enc;
				r.format = delta ? format_delta : format;
				r.decoded = dec;
			}
		};
		// both blue-contracting encodings start from the same contracted pair, and fall together if it leaves the byte range
		const qf contracted0 = q_blue_contract(colour0), contracted1 = q_blue_contract(colour1);
		const bool contractible = quant_level < QUANT_256 &&
		    !q_any(q_zipb(q_testf(contracted0, [](float x) { return x < 0.0f || x > 255.0f; }), q_testf(contracted1, [](float x) { return x < 0.0f || x > 255.0f; })));
		if (quant_level <= QUANT_160)
		{
			if (contractible && q_encode_delta(t, contracted1, contracted0, true, alpha, trial)) consider(trial, true);
			if (q_encode_delta(t, colour0, colour1, false, alpha, trial)) consider(trial, true);
		}
		if (contractible && q_encode_contracted(t, contracted0, contracted1, trial)) consider(trial, false);
		consider(q_encode_direct(t, colour0, colour1), false);

		q_store_u8(output, 2, best.e0, alpha ? 4 : 3);
		q_store_u8(output + 1, 2, best.e1, alpha ? 4 : 3);
		if (!alpha) { r.decoded.e0 = q_with_w(r.decoded.e0, 255); r.decoded.e1 = q_with_w(r.decoded.e1, 255); }
		r.decoded_valid = true;
		return r;
	}

	if (format == FMT_RGB_SCALE || format == FMT_RGB_SCALE_ALPHA)
	{
		q_store_u8(output, 1, q_encode_rgb_scale(t, rgbs), 4);
		if (format == FMT_RGB_SCALE_ALPHA)
		{
			// alpha of both endpoints behind the four bytes: lane 3 computes both
			const float a0 = q_get<3>(colour0), a1 = q_get<3>(colour1);
			Q_ONCE
			{
				output[4] = (uint8_t)quant_color(t, flt2int_rtn(a0), a0);
				output[5] = (uint8_t)quant_color(t, flt2int_rtn(a1), a1);
			}
		}
		return r;
	}

	if (format == FMT_LUMINANCE || format == FMT_LUMINANCE_ALPHA)
	{
		// (luminance, alpha) of both endpoints: two values per endpoint, the same arithmetic on every lane of the quad
		float lum0 = q_hadd_rgb(colour0) * (1.0f / 3.0f);
		float lum1 = q_hadd_rgb(colour1) * (1.0f / 3.0f);
		const float a0 = q_get<3>(colour0), a1 = q_get<3>(colour1);
		if (format == FMT_LUMINANCE)
		{
			if (lum0 > lum1)
			{
				const float avg = (lum0 + lum1) * 0.5f;
				lum0 = avg;
				lum1 = avg;
			}
			Q_ONCE
			{
				output[0] = (uint8_t)quant_color(t, flt2int_rtn(lum0), lum0);
				output[1] = (uint8_t)quant_color(t, flt2int_rtn(lum1), lum1);
			}
			return r;
		}
		if (quant_level <= 18)
		{
			// base + offset form of the (luminance, alpha) pair (ref: try_quantize_luminance_alpha_delta :573): components
			// 0 and 1 of a quad vector, coded like the alpha lane of q_encode_delta
			const qf v0 = q_make(lum0, a0, 0.0f, 0.0f), v1 = q_make(lum1, a1, 0.0f, 0.0f);
			const qi base9 = q_round_to_int(v0) << 1;
			const qi base_q = q_quant(t, base9 & 0xFF);
			const qi base = base_q | (base9 & 0x100);
			qi offset = (q_round_to_int(v1) << 1) - base;
			bool ok = !q_any(q_test(offset, [](int x) { return x > 63 || x < -64; }));
			offset = (offset & 0x7F) | ((base & 0x100) >> 1);
			const qi offset_q = q_quant(t, offset);
			ok = ok && !q_any(q_test((offset ^ offset_q) & 0xC0, [](int x) { return x != 0; }));
			const qi value9 = q_mapi(offset_q & 0x7F, [](int x) { return (x & 0x40) ? x - 0x80 : x; }) + base;
			ok = ok && !q_any(q_test(value9, [](int x) { return x < 0 || x > 0x1FF; }));
			if (ok)
			{
				// bytes: l0 l1 a0 a1
				q_store_u8(output, 2, base_q, 2);
				q_store_u8(output + 1, 2, offset_q, 2);
				r.format = FMT_LUMINANCE_ALPHA_DELTA;
				return r;
			}
		}
		Q_ONCE
		{
			output[0] = (uint8_t)quant_color(t, flt2int_rtn(lum0), lum0);
			output[1] = (uint8_t)quant_color(t, flt2int_rtn(lum1), lum1);
			output[2] = (uint8_t)quant_color(t, flt2int_rtn(a0), a0);
			output[3] = (uint8_t)quant_color(t, flt2int_rtn(a1), a1);
		}
		return r;
	}
	return r;
}

/* ... with the rows of the candidate being refined (staged by stage_color_rows). */
WV_FN QPacked pack_endpoints_quad(const Ctx& c, qf colour0, qf colour1, qf rgbs, int format, uint8_t* output, int quant_level)
{
	return pack_endpoints_quad(color_tabs(c, quant_level), colour0, colour1, rgbs, format, output, quant_level);
}

/* The endpoint formats coded by pack_endpoints_hdr. */
WV_FN bool endpoint_format_is_hdr(int format)
{
	return ((1u << format) & ((1u << FMT_HDR_LUMINANCE_LARGE_RANGE) | (1u << FMT_HDR_LUMINANCE_SMALL_RANGE) | (1u << FMT_HDR_RGB_SCALE) |
	                          (1u << FMT_HDR_RGB) | (1u << FMT_HDR_RGB_LDR_ALPHA) | (1u << FMT_HDR_RGBA))) != 0;
}

#if ASTC_ENABLE_HDR
/* HDR endpoint formats of all partitions of the candidate being refined (ref: the HDR cases of pack_color_endpoints
 * :2076-2140).  requested[p] = format asked for; values + 8 p receives the bytes, formats_out[p] the format used.
 * Partitions with an LDR format are left alone (pack_endpoints_quad codes those).  `tries`: LDS scratch,
 * 4 * HDR_TRY_LANES * HDR_TRY_BYTES bytes. */
WV_FN void pack_endpoints_hdr(const ColorTabs& t, const float* wep0, const float* wep1, const float* rgbo_in,
                              int partition_count, const uint8_t* requested, uint8_t* values, uint8_t* formats_out, uint8_t* tries)
{
	// the sub-modes, side by side
	WV_FOR64(k, partition_count * HDR_TRY_LANES)
	{
		const int p = k / HDR_TRY_LANES, lane = k % HDR_TRY_LANES;
		const int format = requested[p];
		uint8_t* rec = tries + k * HDR_TRY_BYTES;
		const f4 low = v4_clamp(0.0f, 65535.0f, load4(wep0 + 4 * p)), high = v4_clamp(0.0f, 65535.0f, load4(wep1 + 4 * p));
		if (format == FMT_HDR_RGB_SCALE)
		{
			if (lane < 5) hdr_try_rgbo(t, load4(rgbo_in + 4 * p), lane, rec);                       // preference: sub-mode 0 first
		}
		else if (format == FMT_HDR_RGB || format == FMT_HDR_RGB_LDR_ALPHA || format == FMT_HDR_RGBA)
		{
			if (lane < 8) hdr_try_rgb(t, low, high, 7 - lane, rec);                             // preference: sub-mode 7 first
			else if (lane < 11 && format == FMT_HDR_RGBA) hdr_try_alpha(t, low.w, high.w, 10 - lane, rec);   // finest first
		}
	}
	WV_SYNC();
	// first sub-mode that fits, else the escape layout
	WV_FOR64(p, partition_count)
	{
		const int format = requested[p];
		if (!endpoint_format_is_hdr(format)) continue;
		uint8_t* out = values + p * 8;
		const uint8_t* recs = tries + p * HDR_TRY_LANES * HDR_TRY_BYTES;
		const f4 low = v4_clamp(0.0f, 65535.0f, load4(wep0 + 4 * p)), high = v4_clamp(0.0f, 65535.0f, load4(wep1 + 4 * p));
		auto first_fit = [recs](int begin, int end) {
			for (int m = begin; m < end; m++) if (recs[m * HDR_TRY_BYTES]) return m;
			return -1;
		};
		int used = format;
		if (format == FMT_HDR_RGB_SCALE)
		{
			const int m = first_fit(0, 5);
			if (m >= 0) { for (int i = 0; i < 4; i++) out[i] = recs[m * HDR_TRY_BYTES + 1 + i]; }
			else hdr_escape_rgbo(t, load4(rgbo_in + 4 * p), out);
		}
		else if (format == FMT_HDR_LUMINANCE_SMALL_RANGE || format == FMT_HDR_LUMINANCE_LARGE_RANGE)
		{
			used = hdr_code_luminance(t, low, high, out);
		}
		else
		{
			const int m = first_fit(0, 8);
			if (m >= 0) { for (int i = 0; i < 6; i++) out[i] = recs[m * HDR_TRY_BYTES + 1 + i]; }
			else hdr_escape_rgb(t, low, high, out);
			if (format == FMT_HDR_RGB_LDR_ALPHA)
			{
				const float a0 = f_clamp255(low.w * (1.0f / 257.0f)), a1 = f_clamp255(high.w * (1.0f / 257.0f));
				out[6] = (uint8_t)quant_color(t, flt2int_rtn(a0), a0);
				out[7] = (uint8_t)quant_color(t, flt2int_rtn(a1), a1);
			}
			else if (format == FMT_HDR_RGBA)
			{
				const int ma = first_fit(8, 11);
				if (ma >= 0) { out[6] = recs[ma * HDR_TRY_BYTES + 1]; out[7] = recs[ma * HDR_TRY_BYTES + 2]; }
				else
				{
					// plain 7-bit alphas (ref: :1890-1906)
					const int a0 = flt2int_rtn(f_clamp(low.w, 0.0f, 65280.0f)), a1 = flt2int_rtn(f_clamp(high.w, 0.0f, 65280.0f));
					out[6] = (uint8_t)quant_color(t, ((a0 + 256) >> 9) | 0x80);
					out[7] = (uint8_t)quant_color(t, ((a1 + 256) >> 9) | 0x80);
				}
			}
		}
		formats_out[p] = (uint8_t)used;
	}
}
#endif

} } // namespace astcd::ASTC_VARIANT
