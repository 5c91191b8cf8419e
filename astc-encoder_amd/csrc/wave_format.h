// SPDX-License-Identifier: Apache-2.0
// Endpoint format / colour quant level selection for every block mode of a trial, and the top-N
// candidate pick.
//   ref: compute_error_squared_rgb_single_partition  Source/astcenc_pick_best_endpoint_format.cpp:72-208
//        compute_encoding_choice_errors              :222-300
//        compute_color_error_for_every_integer_count_and_quant_level :315-665
//        {one..four}_partition(s)_find_best_combination_* :678-1093
//        compute_ideal_endpoint_formats              :1096-1357
#pragma once
#include "wave_ctx.h"
#include "wave_ideal.h"
#include "wave_weights.h"

namespace astcd { inline namespace ASTC_VARIANT {

/* Wave-wide argmin with lowest-index tie break over v(i), i in [start, end); entries >= 1e30 are
 * never selected (returns -1 if none).  Uniform result. */
#if WV_DEVICE
/* Fold per-lane (value, index) pairs across the wave: smaller value wins, equal values go to the smaller index; the
 * winning pair is returned on every lane.  DPP steps (row_shr 1 / 2 / 4 / 8, then row_bcast 15 / 31; lanes without a
 * source keep their own pair), no LDS traffic: the ds_bpermute butterfly this replaces was a dozen dependent LDS round
 * trips per pick. */
WV_FN void wave_argmin_fold(float& best, int& idx)
{
	#define WV_ARGMIN_STEP(CTRL, ROW_MASK) do { \
		const float ov = int_as_float(__builtin_amdgcn_update_dpp(float_as_int(best), float_as_int(best), CTRL, ROW_MASK, 0xF, false)); \
		const int oi = __builtin_amdgcn_update_dpp(idx, idx, CTRL, ROW_MASK, 0xF, false); \
		const bool take = ov < best || (ov == best && oi < idx); \
		best = take ? ov : best; idx = take ? oi : idx; } while (0)
	WV_ARGMIN_STEP(0x111, 0xF); WV_ARGMIN_STEP(0x112, 0xF); WV_ARGMIN_STEP(0x114, 0xF); WV_ARGMIN_STEP(0x118, 0xF);
	WV_ARGMIN_STEP(0x142, 0xA); WV_ARGMIN_STEP(0x143, 0xC);
	#undef WV_ARGMIN_STEP
	idx = __builtin_amdgcn_readlane(idx, 63);
	best = int_as_float(__builtin_amdgcn_readlane(float_as_int(best), 63));
}
#endif

template <typename ValFn>
WV_FN int wave_argmin(const Ctx& c, int start, int end, ValFn v)
{
	(void)c;
#if WV_DEVICE
	float best = ERROR_CALC_DEFAULT;
	int idx = 0x7FFFFFFF;                    // (no candidate: compares above every index)
	for (int i = start + WV_LANE; i < end; i += 64)
	{
		float e = v(i);
		if (e < best) { best = e; idx = i; }
	}
	wave_argmin_fold(best, idx);
	return best < ERROR_CALC_DEFAULT ? idx : -1;
#else
	float best = ERROR_CALC_DEFAULT;
	int idx = -1;
	for (int i = start; i < end; i++)
	{
		float e = v(i);
		if (e < best) { best = e; idx = i; }
	}
	return idx;
#endif
}

/* Endpoint-format tables of one trial, laid out in the `uni` LDS region (fmt_scratch_bytes()):
 *   best_error[P][17][4], format_of_choice[P][17][4]   per partition x quant level x integer count
 *   comb_error[17][cols], comb_format[17][cols] (4 x 4 bits) best combination over partitions
 * Quant rows are indexed by (quant - QUANT_6): lower levels are never legal for colour endpoints
 * (the reference fills them with ERROR_CALC_DEFAULT and never reads them back, ref :328-346). */
struct FmtView {
	float*   best_error_;
	uint8_t* format_of_choice_;
	float*   comb_error_;
	uint16_t* comb_format_;      // four 4-bit endpoint formats per cell, partition p in bits 4p .. 4p+3
	int      cols;

	WV_FN float* best_error(int p, int quant) const { return best_error_ + (p * (int)FMT_QUANT_ROWS + (quant - QUANT_6)) * 4; }
	WV_FN uint8_t* format_of_choice(int p, int quant) const { return format_of_choice_ + (p * (int)FMT_QUANT_ROWS + (quant - QUANT_6)) * 4; }
	WV_FN float* comb_error(int quant) const { return comb_error_ + (quant - QUANT_6) * cols; }
	WV_FN uint16_t& comb_format(int quant, int col) const { return comb_format_[(quant - QUANT_6) * cols + col]; }
};

WV_FN FmtView fmt_view(const Ctx& c)
{
	uint32_t P = c.cfg->tune_partition_count_limit;
	P = P < 1 ? 1u : P > 4 ? 4u : P;
	FmtView v;
	uint8_t* base = c.fmt();
	v.best_error_ = reinterpret_cast<float*>(base);
	v.format_of_choice_ = base + P * FMT_QUANT_ROWS * 16;
	v.cols = (int)fmt_comb_cols(P);
	v.comb_error_ = reinterpret_cast<float*>(base + P * FMT_QUANT_ROWS * 20);
	v.comb_format_ = reinterpret_cast<uint16_t*>(base + P * FMT_QUANT_ROWS * 20 + FMT_QUANT_ROWS * (uint32_t)v.cols * 4);
	return v;
}

WV_FN float blk_default_alpha(const BlkInfo& blk) { return blk.alpha_lns ? (float)0x7800 : (float)0xFFFF; }
WV_FN bool blk_is_luminance(const BlkInfo& blk)
{
	float da = blk_default_alpha(blk);
	bool alpha1 = (blk.data_min[3] == da) && (blk.data_max[3] == da);
	return blk.grayscale && alpha1;
}
WV_FN bool blk_is_luminancealpha(const BlkInfo& blk)
{
	float da = blk_default_alpha(blk);
	bool alpha1 = (blk.data_min[3] == da) && (blk.data_max[3] == da);
	return blk.grayscale && !alpha1;
}

/* (ref: compute_encoding_choice_errors :222) ep = endpoints [partition][channel] */
WV_FN void compute_encoding_choice_errors(const Ctx& c, const PartView& pv, const float (*ep0)[4], const float (*ep1)[4])
{
	TrialInfo& tr = c.tr();
	const BlkInfo& blk = c.blk();
	const int T = c.T, pc = pv.pcount;

	// With one partition the four errors below depend on the block alone: the first single-partition trial of the block
	// leaves them in tr.eci1 and the later ones (second run of trial A, the two-plane trials) read them back.
	const bool cached = pc == 1 && wv_uniform(tr.eci1_valid) != 0;
	if (!cached)
	{
		CompSel rgb; rgb.ncomp = 3; rgb.set(0, 1, 2, 0);
		compute_avgs_and_dirs(c, pv, rgb);

		// processed lines per partition -> fbox[p*16 + ..]: uncor amod(3) bs(3), samec bs(3), rgbl amod(3)
		WV_FOR64(p, pc)
		{
			f4 avg = load4(tr.pm_avg[p]);
			f4 dir = load4(tr.pm_dir[p]);
			f4 uncor_b = normalize_safe4(dir, unit3());
			f4 samec_b = normalize_safe4(avg, unit3());
			f4 luma_b = unit3();
			float d_uncor = dot3_s(avg, uncor_b);
			f4 uncor_amod = avg - uncor_b * mk4(d_uncor, d_uncor, d_uncor, 0.0f);
			float d_luma = dot3_s(avg, luma_b);
			f4 luma_amod = avg - luma_b * mk4(d_luma, d_luma, d_luma, 0.0f);
			float* o = &tr.fbox[p * 16];
			o[0] = uncor_amod.x; o[1] = uncor_amod.y; o[2] = uncor_amod.z;
			o[3] = uncor_b.x;    o[4] = uncor_b.y;    o[5] = uncor_b.z;
			o[6] = samec_b.x;    o[7] = samec_b.y;    o[8] = samec_b.z;
			o[9] = luma_amod.x;  o[10] = luma_amod.y; o[11] = luma_amod.z;
		}
		WV_SYNC();

		// per-texel error terms in partition order (ref: :124-201)
		const float default_a = blk_default_alpha(blk);
		const float ew0 = cw_of(blk, 0), ew1 = cw_of(blk, 1), ew2 = cw_of(blk, 2);
		WV_FOR_T(i, T)
		{
			int t = pv.sorted[i];
			int p = pv.of_texel[t];
			const float* o = &tr.fbox[p * 16];
			float r = c.data(0)[t], g = c.data(1)[t], b = c.data(2)[t], a = c.data(3)[t];

			float alpha_diff = a - default_a;
			c.tsc_f(0)[i] = alpha_diff * alpha_diff;

			float param = r * o[3] + g * o[4] + b * o[5];
			float dist0 = (o[0] + param * o[3]) - r;
			float dist1 = (o[1] + param * o[4]) - g;
			float dist2 = (o[2] + param * o[5]) - b;
			c.tsc_f(1)[i] = dist0 * dist0 * ew0 + dist1 * dist1 * ew1 + dist2 * dist2 * ew2;

			param = r * o[6] + g * o[7] + b * o[8];
			dist0 = (param * o[6]) - r;
			dist1 = (param * o[7]) - g;
			dist2 = (param * o[8]) - b;
			c.tsc_f(2)[i] = dist0 * dist0 * ew0 + dist1 * dist1 * ew1 + dist2 * dist2 * ew2;

			const float u = 0.577350258827209473f;
			param = r * u + g * u + b * u;
			dist0 = (o[9] + param * u) - r;
			dist1 = (o[10] + param * u) - g;
			dist2 = (o[11] + param * u) - b;
			c.tsc_f(3)[i] = dist0 * dist0 * ew0 + dist1 * dist1 * ew1 + dist2 * dist2 * ew2;

			dist0 = (param * u) - r;
			dist1 = (param * u) - g;
			dist2 = (param * u) - b;
			c.tsc_f(4)[i] = dist0 * dist0 * ew0 + dist1 * dist1 * ew1 + dist2 * dist2 * ew2;
		}
		WV_SYNC();

		WV_FOR64(k, pc * 5)
		{
			int p = k / 5, which = k % 5;
			tr.fbox[64 - 20 + k] = sum4(c.tsc_f(which) + pv.off(p), pv.cnt(p));
		}
		WV_SYNC();

	}

	WV_FOR64(p, pc)
	{
		const float* s = &tr.fbox[64 - 20 + p * 5];
		float e_scale, e_luma, e_lum, e_drop;
		if (cached)
		{
			e_scale = tr.eci1[0]; e_luma = tr.eci1[1]; e_lum = tr.eci1[2]; e_drop = tr.eci1[3];
		}
		else
		{
			float a_drop = s[0] * cw_of(blk, 3);
			float uncor = s[1], samec = s[2], rgbl = s[3], lum = s[4];
			e_scale = (samec - uncor) * 0.7f;
			e_luma = (rgbl - uncor) * 1.5f;
			e_lum = (lum - uncor) * 3.0f;
			e_drop = a_drop * 3.0f;
			if (pc == 1) { tr.eci1[0] = e_scale; tr.eci1[1] = e_luma; tr.eci1[2] = e_lum; tr.eci1[3] = e_drop; tr.eci1_valid = 1; }
		}
		bool can_offset = true;
		for (int k = 0; k < 3; k++)
		{
			float diff = f_abs(ep1[p][k] - ep0[p][k]);
			can_offset = can_offset && (diff < (0.12f * 65535.0f));
		}
		tr.eci_rgb_scale[p] = e_scale;
		tr.eci_rgb_luma[p] = e_luma;
		tr.eci_luminance[p] = e_lum;
		tr.eci_alpha_drop[p] = e_drop;
		tr.eci_can_offset[p] = can_offset ? 1 : 0;
		tr.eci_can_blue_contract[p] = blk_is_luminance(blk) ? 0 : 1;
	}
	WV_SYNC();
}

WV_FN float baseline_quant_error(int i /* quant - QUANT_6 */)
{
	const float t[17] = {
		(65536.0f * 65536.0f / 18.0f) / (5 * 5),
		(65536.0f * 65536.0f / 18.0f) / (7 * 7),
		(65536.0f * 65536.0f / 18.0f) / (9 * 9),
		(65536.0f * 65536.0f / 18.0f) / (11 * 11),
		(65536.0f * 65536.0f / 18.0f) / (15 * 15),
		(65536.0f * 65536.0f / 18.0f) / (19 * 19),
		(65536.0f * 65536.0f / 18.0f) / (23 * 23),
		(65536.0f * 65536.0f / 18.0f) / (31 * 31),
		(65536.0f * 65536.0f / 18.0f) / (39 * 39),
		(65536.0f * 65536.0f / 18.0f) / (47 * 47),
		(65536.0f * 65536.0f / 18.0f) / (63 * 63),
		(65536.0f * 65536.0f / 18.0f) / (79 * 79),
		(65536.0f * 65536.0f / 18.0f) / (95 * 95),
		(65536.0f * 65536.0f / 18.0f) / (127 * 127),
		(65536.0f * 65536.0f / 18.0f) / (159 * 159),
		(65536.0f * 65536.0f / 18.0f) / (191 * 191),
		(65536.0f * 65536.0f / 18.0f) / (255 * 255)
	};
	return t[i];
}

/* One (partition, quant level) cell of the table. (ref: :315-665) */
WV_FN void color_error_for_quant_level(const Ctx& c, const PartView& pv, int p, int i,
                                       const float* ep0p, const float* ep1p, const FmtView& fs)
{
	const TrialInfo& tr = c.tr();
	const BlkInfo& blk = c.blk();
	bool encode_hdr_rgb = kHdr && blk.rgb_lns != 0;
	bool encode_hdr_alpha = kHdr && blk.alpha_lns != 0;
	int partition_size = pv.cnt(p);
	float* best_error = fs.best_error(p, i);
	uint8_t* fmt = fs.format_of_choice(p, i);

	f4 ep0 = load4(ep0p), ep1 = load4(ep1p);
	f4 ew = cw4_of(blk);

	float ep1_min = hmin4(ep1.x, ep1.y, ep1.z, ep1.x);
	ep1_min = f_max(ep1_min, 0.0f);

	float error_weight_rgbsum = hadd_rgb_s(ew);
	float range_upper_limit_rgb = encode_hdr_rgb ? 61440.0f : 65535.0f;
	float range_upper_limit_alpha = encode_hdr_alpha ? 61440.0f : 65535.0f;

	f4 offset = mk4(range_upper_limit_rgb, range_upper_limit_rgb, range_upper_limit_rgb, range_upper_limit_alpha);
	f4 ep0_high = v4_max(ep0 - offset, splat4(0.0f));
	f4 ep1_high = v4_max(ep1 - offset, splat4(0.0f));
	f4 ep0_low = v4_min(ep0, splat4(0.0f));
	f4 ep1_low = v4_min(ep1, splat4(0.0f));

	f4 sum_range_error = (ep0_low * ep0_low) + (ep1_low * ep1_low) + (ep0_high * ep0_high) + (ep1_high * ep1_high);

	float rgb_range_error = dot3_s(sum_range_error, ew) * 0.5f * (float)partition_size;
	float alpha_range_error = sum_range_error.w * ew.w * 0.5f * (float)partition_size;

	float eci_alpha_drop = tr.eci_alpha_drop[p], eci_rgb_luma = tr.eci_rgb_luma[p];
	float eci_luminance = tr.eci_luminance[p], eci_rgb_scale = tr.eci_rgb_scale[p];

	if (encode_hdr_rgb)
	{
		if (i < QUANT_16)
		{
			best_error[3] = best_error[2] = best_error[1] = best_error[0] = ERROR_CALC_DEFAULT;
			fmt[3] = (uint8_t)(encode_hdr_alpha ? FMT_HDR_RGBA : FMT_HDR_RGB_LDR_ALPHA);
			fmt[2] = FMT_HDR_RGB;
			fmt[1] = FMT_HDR_RGB_SCALE;
			fmt[0] = FMT_HDR_LUMINANCE_LARGE_RANGE;
			return;
		}

		// heuristic sub-mode pick from the endpoint spread (ref: :383-512)
		float af, cf;
		if (ep1.x > ep1.y && ep1.x > ep1.z) { af = ep1.x; cf = ep1.x - ep0.x; }
		else if (ep1.y > ep1.z) { af = ep1.y; cf = ep1.y - ep0.y; }
		else { af = ep1.z; cf = ep1.z - ep0.z; }

		float bf = af - ep1_min;
		f4 prd = xyz0(ep1 - splat4(cf));
		f4 pdif = prd - xyz0(ep0);
		f4 ap = v4_abs(pdif);
		float df = hmax4(ap.x, ap.y, ap.z, ap.w);

		int b = (int)f_clamp(bf, 0.0f, 65536.0f);
		int cc = (int)f_clamp(cf, 0.0f, 65536.0f);
		int d = (int)f_clamp(df, 0.0f, 65536.0f);

		// (each comparison is turned into a 0 / 1 integer at once: as thirteen && chains the compiler keeps some twenty
		//  lane masks alive at the same time, more scalar registers than an out-of-line stage has without parking
		//  some in a vector register -- which costs a scratch frame)
		auto below = [](int v, int limit) { return wv_opaque(v < limit ? 1 : 0); };
		int rgbo_mode = 5;
		if (below(b, 32768) & below(cc, 16384)) rgbo_mode = 4;
		if (below(b, 8192) & below(cc, 16384)) rgbo_mode = 3;
		if (below(b, 2048) & below(cc, 16384)) rgbo_mode = 2;
		if (below(b, 2048) & below(cc, 1024)) rgbo_mode = 1;
		if (below(b, 1024) & below(cc, 4096)) rgbo_mode = 0;

		int rgb_mode = 8;
		if (below(b, 16384) & below(cc, 8192) & below(d, 8192)) rgb_mode = 0;
		if (below(b, 32768) & below(cc, 8192) & below(d, 4096)) rgb_mode = 1;
		if (below(b, 4096) & below(cc, 8192) & below(d, 4096)) rgb_mode = 2;
		if (below(b, 8192) & below(cc, 8192) & below(d, 2048)) rgb_mode = 3;
		if (below(b, 8192) & below(cc, 2048) & below(d, 512)) rgb_mode = 4;
		if (below(b, 2048) & below(cc, 8192) & below(d, 1024)) rgb_mode = 5;
		if (below(b, 2048) & below(cc, 2048) & below(d, 256)) rgb_mode = 6;
		if (below(b, 1024) & below(cc, 2048) & below(d, 512)) rgb_mode = 7;

		const float rgbo_error_scales[6] = { 4.0f, 4.0f, 16.0f, 64.0f, 256.0f, 1024.0f };
		const float rgb_error_scales[9] = { 64.0f, 64.0f, 16.0f, 16.0f, 4.0f, 4.0f, 1.0f, 1.0f, 384.0f };

		float mode7mult = rgbo_error_scales[rgbo_mode] * 0.0015f;
		float mode11mult = rgb_error_scales[rgb_mode] * 0.010f;

		float lum_high = hadd_rgb_s(ep1) * (1.0f / 3.0f);
		float lum_low = hadd_rgb_s(ep0) * (1.0f / 3.0f);
		float lumdif = lum_high - lum_low;
		float mode23mult = lumdif < 960 ? 4.0f : lumdif < 3968 ? 16.0f : 128.0f;
		mode23mult *= 0.0005f;

		float base_quant_error = baseline_quant_error(i - QUANT_6) * (float)partition_size;
		float rgb_quantization_error = error_weight_rgbsum * base_quant_error * 2.0f;
		float alpha_quantization_error = ew.w * base_quant_error * 2.0f;
		float rgba_quantization_error = rgb_quantization_error + alpha_quantization_error;

		best_error[3] = rgba_quantization_error + rgb_range_error + alpha_range_error;
		fmt[3] = (uint8_t)(encode_hdr_alpha ? FMT_HDR_RGBA : FMT_HDR_RGB_LDR_ALPHA);

		best_error[2] = (rgb_quantization_error * mode11mult) + rgb_range_error + eci_alpha_drop;
		fmt[2] = FMT_HDR_RGB;

		best_error[1] = (rgb_quantization_error * mode7mult) + rgb_range_error + eci_alpha_drop + eci_rgb_luma;
		fmt[1] = FMT_HDR_RGB_SCALE;

		best_error[0] = (rgb_quantization_error * mode23mult) + rgb_range_error + eci_alpha_drop + eci_luminance;
		fmt[0] = FMT_HDR_LUMINANCE_LARGE_RANGE;
		return;
	}

	if (i < QUANT_6)
	{
		best_error[3] = best_error[2] = best_error[1] = best_error[0] = ERROR_CALC_DEFAULT;
		fmt[3] = FMT_RGBA; fmt[2] = FMT_RGB; fmt[1] = FMT_RGB_SCALE; fmt[0] = FMT_LUMINANCE;
		return;
	}

	float base_quant_error_rgb = error_weight_rgbsum * (float)partition_size;
	float base_quant_error_a = ew.w * (float)partition_size;
	float base_quant_error_rgba = base_quant_error_rgb + base_quant_error_a;

	float error_scale_bc_rgba = tr.eci_can_blue_contract[p] ? 0.625f : 1.0f;
	float error_scale_oe_rgba = tr.eci_can_offset[p] ? 0.5f : 1.0f;
	float error_scale_bc_rgb = tr.eci_can_blue_contract[p] ? 0.5f : 1.0f;
	float error_scale_oe_rgb = tr.eci_can_offset[p] ? 0.25f : 1.0f;
	if (i >= QUANT_192)
	{
		error_scale_oe_rgba = 1.0f;
		error_scale_oe_rgb = 1.0f;
	}

	float base_quant_error = baseline_quant_error(i - QUANT_6);
	float quant_error_rgb = base_quant_error_rgb * base_quant_error;
	float quant_error_rgba = base_quant_error_rgba * base_quant_error;

	float full_ldr_rgba_error = quant_error_rgba * error_scale_bc_rgba * error_scale_oe_rgba + rgb_range_error + alpha_range_error;
	best_error[3] = full_ldr_rgba_error;
	fmt[3] = FMT_RGBA;

	float full_ldr_rgb_error = quant_error_rgb * error_scale_bc_rgb * error_scale_oe_rgb + rgb_range_error + eci_alpha_drop;
	float rgbs_alpha_error = quant_error_rgba + eci_rgb_scale + rgb_range_error + alpha_range_error;
	if (rgbs_alpha_error < full_ldr_rgb_error)
	{
		best_error[2] = rgbs_alpha_error;
		fmt[2] = FMT_RGB_SCALE_ALPHA;
	}
	else
	{
		best_error[2] = full_ldr_rgb_error;
		fmt[2] = FMT_RGB;
	}

	float ldr_rgbs_error = quant_error_rgb + rgb_range_error + eci_alpha_drop + eci_rgb_scale;
	float lum_alpha_error = quant_error_rgba + rgb_range_error + alpha_range_error + eci_luminance;
	if (ldr_rgbs_error < lum_alpha_error)
	{
		best_error[1] = ldr_rgbs_error;
		fmt[1] = FMT_RGB_SCALE;
	}
	else
	{
		best_error[1] = lum_alpha_error;
		fmt[1] = FMT_LUMINANCE_ALPHA;
	}

	best_error[0] = quant_error_rgb + rgb_range_error + eci_alpha_drop + eci_luminance;
	fmt[0] = FMT_LUMINANCE;
}

/* Combine the per-partition tables for one quant level (ref: :728-766, :842-891, :967-1027). */
WV_FN void combine_partitions_for_quant(int pc, int quant, const FmtView& fs)
{
	const int ncols = pc == 2 ? 7 : pc == 3 ? 10 : 13;
	for (int j = 0; j < ncols; j++) fs.comb_error(quant)[j] = ERROR_CALC_DEFAULT;
	if (quant < QUANT_6) return;

	for (int i = 0; i < 4; i++)
	{
		for (int j = 0; j < 4; j++)
		{
			int low2 = i_min(i, j), high2 = i_max(i, j);
			if ((high2 - low2) > 1) continue;
			if (pc == 2)
			{
				int intcnt = i + j;
				float errorterm = f_min(fs.best_error(0, quant)[i] + fs.best_error(1, quant)[j], 1e10f);
				if (errorterm <= fs.comb_error(quant)[intcnt])
				{
					fs.comb_error(quant)[intcnt] = errorterm;
					fs.comb_format(quant, intcnt) = (uint16_t)(fs.format_of_choice(0, quant)[i] | (fs.format_of_choice(1, quant)[j] << 4));
				}
				continue;
			}
			for (int k = 0; k < 4; k++)
			{
				int low3 = i_min(k, low2), high3 = i_max(k, high2);
				if ((high3 - low3) > 1) continue;
				if (pc == 3)
				{
					int intcnt = i + j + k;
					float errorterm = f_min(fs.best_error(0, quant)[i] + fs.best_error(1, quant)[j] + fs.best_error(2, quant)[k], 1e10f);
					if (errorterm <= fs.comb_error(quant)[intcnt])
					{
						fs.comb_error(quant)[intcnt] = errorterm;
						fs.comb_format(quant, intcnt) = (uint16_t)(fs.format_of_choice(0, quant)[i] | (fs.format_of_choice(1, quant)[j] << 4) |
						                                           (fs.format_of_choice(2, quant)[k] << 8));
					}
					continue;
				}
				for (int l = 0; l < 4; l++)
				{
					int low4 = i_min(l, low3), high4 = i_max(l, high3);
					if ((high4 - low4) > 1) continue;
					int intcnt = i + j + k + l;
					float errorterm = f_min(fs.best_error(0, quant)[i] + fs.best_error(1, quant)[j] + fs.best_error(2, quant)[k] + fs.best_error(3, quant)[l], 1e10f);
					if (errorterm <= fs.comb_error(quant)[intcnt])
					{
						fs.comb_error(quant)[intcnt] = errorterm;
						fs.comb_format(quant, intcnt) = (uint16_t)(fs.format_of_choice(0, quant)[i] | (fs.format_of_choice(1, quant)[j] << 4) |
						                                           (fs.format_of_choice(2, quant)[k] << 8) | (fs.format_of_choice(3, quant)[l] << 12));
					}
				}
			}
		}
	}
}

/* The colour quant level of every integer-pair count for one bit budget: one 16-byte row of the transposed quant mode
 * table (TableRoot::off_quant_mode_by_bits), fetched with a single load. */
struct QuantLevels
{
	uint32_t w[4];
	WV_FN int of(int pairs) const
	{
		const uint32_t word = pairs < 4 ? w[0] : pairs < 8 ? w[1] : w[2];
		return (int)(int8_t)(uint8_t)(word >> (8 * (pairs & 3)));
	}
	/* The row moved down by `first` entries (0 .. 7): result.of(k) == of(first + k), -1 past the end.  Loops over
	 * "first + k" with a wave-uniform `first` then index with constants (a run-time entry index costs a scalar lane mask
	 * per word choice). */
	WV_FN QuantLevels from(int first) const
	{
		QuantLevels r;
		const bool word_up = first >= 4;
		const uint32_t a0 = word_up ? w[1] : w[0], a1 = word_up ? w[2] : w[1], a2 = word_up ? w[3] : w[2], a3 = word_up ? 0xFFFFFFFFu : w[3];
		const uint32_t s = 8u * ((uint32_t)first & 3u);
#if WV_DEVICE
		r.w[0] = __builtin_amdgcn_alignbit(a1, a0, s);
		r.w[1] = __builtin_amdgcn_alignbit(a2, a1, s);
		r.w[2] = __builtin_amdgcn_alignbit(a3, a2, s);
		r.w[3] = __builtin_amdgcn_alignbit(0xFFFFFFFFu, a3, s);
#else
		r.w[0] = (uint32_t)((((uint64_t)a1 << 32) | a0) >> s);
		r.w[1] = (uint32_t)((((uint64_t)a2 << 32) | a1) >> s);
		r.w[2] = (uint32_t)((((uint64_t)a3 << 32) | a2) >> s);
		r.w[3] = (uint32_t)(((0xFFFFFFFFull << 32) | a3) >> s);
#endif
		return r;
	}
};
static_assert(sizeof(QuantLevels) == 16, "one row of the transposed quant mode table");

/* `e` if `level` is a legal colour quant level (>= QUANT_6), else ERROR_CALC_DEFAULT -- as bit arithmetic on the vector
 * unit: a handful of these in flight as compare + select would each hold a lane mask in a scalar register pair. */
WV_FN float error_unless_legal(float e, int level)
{
	const int illegal = (level - (int)QUANT_6) >> 31;            // all ones below QUANT_6
	return int_as_float((float_as_int(e) & ~illegal) | (float_as_int(ERROR_CALC_DEFAULT) & illegal));
}

WV_FN QuantLevels quant_levels_for_bits(const Ctx& c, int bits_available)
{
	return table_at(reinterpret_cast<const QuantLevels*>(c.table(c.root->off_quant_mode_by_bits)), (uint32_t)bits_available);
}

/* Best (quant level, formats) for one block mode's colour bit budget. (ref: :678-718, :780-832,
 * :905-957, :1041-1093) */
/* `quant` receives the colour quant level, `quant_mod` the level that matched formats would allow, `formats` [4] the
 * endpoint formats; all three null in the scoring pass, which only wants the error. */
WV_FN float best_combination_for_levels(const Ctx& c, int pc, const FmtView& fs, const QuantLevels levels, int bits_available,
                                        uint8_t* quant, uint8_t* quant_mod, uint8_t* formats)
{
	float best_integer_count_error = ERROR_CALC_DEFAULT;

	if (pc == 1)
	{
		int best_integer_count = 0;
		// (all four table reads are issued together -- a level below QUANT_6 reads the QUANT_6 row and is ignored --, then
		// the reference's loop runs over the values: one LDS round trip instead of four dependent ones)
		float e[4];
		#pragma unroll
		for (int k = 0; k < 4; k++) e[k] = fs.best_error(0, i_max(levels.of(k + 1), (int)QUANT_6))[k];
		#pragma unroll
		for (int k = 0; k < 4; k++)
		{
			const float ek = error_unless_legal(e[k], levels.of(k + 1));                   // (1e30: never below the running best)
			if (ek < best_integer_count_error)
			{
				best_integer_count_error = ek;
				best_integer_count = k;
			}
		}
		if (quant)
		{
			int ql = levels.of(best_integer_count + 1);
			*quant = (uint8_t)ql;
			*quant_mod = (uint8_t)ql;
			formats[0] = ql >= QUANT_6 ? fs.format_of_choice(0, ql)[best_integer_count] : (uint8_t)FMT_LUMINANCE;
		}
		return best_integer_count_error;
	}

	const int lo = pc;                       // minimum integer-pair count
	// (integer-pair counts lo .. 8 for two partitions, lo .. 9 for three and four: at most seven)
	const int mod_bits = pc == 2 ? 2 : pc == 3 ? 5 : 8;
	int best_integer_count = 0;
	// (the same with up to seven reads in flight; the reference stops at the first integer count whose level is too low)
	const QuantLevels from_lo = levels.from(lo);
	float e[7];
	#pragma unroll
	for (int k = 0; k < 7; k++) e[k] = fs.comb_error(i_max(from_lo.of(k), (int)QUANT_6))[k];
	// (the levels never rise with the integer count -- more values in the same bits --, so "stop at the first level that
	// is too low" and "skip every level that is too low" are the same thing; tests/test_tables.py checks the table)
	#pragma unroll
	for (int k = 0; k < 7; k++)
	{
		const float ek = error_unless_legal(e[k], from_lo.of(k));          // (integer counts past `hi`: the rows hold -1 there)
		if (ek < best_integer_count_error)
		{
			best_integer_count_error = ek;
			best_integer_count = lo + k;
		}
	}
	if (quant)
	{
		int ql = levels.of(best_integer_count);
		int ql_mod = quant_levels_for_bits(c, bits_available + mod_bits).of(best_integer_count);
		*quant = (uint8_t)ql;
		*quant_mod = (uint8_t)ql_mod;
		for (int i = 0; i < pc; i++)
		{
			formats[i] = ql >= QUANT_6 ? (uint8_t)((fs.comb_format(ql, best_integer_count - lo) >> (4 * i)) & 0xF) : (uint8_t)FMT_LUMINANCE;
		}
	}
	return best_integer_count_error;
}

WV_FN float best_combination_for_bitcount(const Ctx& c, int pc, const FmtView& fs, int bits_available,
                                          uint8_t* quant, uint8_t* quant_mod, uint8_t* formats)
{
	return best_combination_for_levels(c, pc, fs, quant_levels_for_bits(c, bits_available), bits_available, quant, quant_mod, formats);
}

/* Colour bits left after the weights. (ref: compress_symbolic.cpp:434-453, :817) */
WV_FN int mode_bitcount(int partition_count, const BlockMode& bm)
{
	if (bm.is_dual_plane) return 109 - bm.weight_bits;
	const int free_bits[4] = { 111, 97, 94, 91 };
	return free_bits[partition_count - 1] - bm.weight_bits;
}

/* (ref: compute_ideal_endpoint_formats :1096), first half: the error tables of every (partition, quant level, format
 * class) and their best combinations over the partitions, in the `uni` LDS region (FmtView). */
WV_FN void compute_ideal_endpoint_formats(const Ctx& c, const PartView& pv, const float (*ep0)[4], const float (*ep1)[4],
                                          int start_block_mode, int end_block_mode)
{
	TrialInfo& tr = c.tr();
	const int pc = pv.pcount;
	const FmtView fs = fmt_view(c);
	ModeRec* modes = c.modes(start_block_mode);

	{ PROF_SCOPE(c, PS_FMT1); compute_encoding_choice_errors(c, pv, ep0, ep1); }

	{ PROF_SCOPE(c, PS_FMT2);
	WV_FOR(k, pc * (int)FMT_QUANT_ROWS)
	{
		int p = k / (int)FMT_QUANT_ROWS, i = k % (int)FMT_QUANT_ROWS + QUANT_6;
		color_error_for_quant_level(c, pv, p, i, ep0[p], ep1[p], fs);
	}
	WV_SYNC(); }

	if (pc >= 2)
	{
		PROF_SCOPE(c, PS_FMT3);
		WV_FOR(q, (int)FMT_QUANT_ROWS) { combine_partitions_for_quant(pc, q + QUANT_6, fs); }
		WV_SYNC();
	}

	(void)tr; (void)modes; (void)start_block_mode; (void)end_block_mode;
}

/* Second half (ref: :1133-1333): the best colour encoding of every scored block mode, then the tune_candidate_limit best
 * modes with their quant levels and formats -> tr.cand_*.  Its own out-of-line stage on the device: together with the
 * table building above it runs out of scalar registers. */
WV_FN void select_candidate_modes(const Ctx& c, int pc, int start_block_mode, int end_block_mode)
{
	TrialInfo& tr = c.tr();
	const FmtView fs = fmt_view(c);
	ModeRec* modes = c.modes(start_block_mode);
	PROF_SCOPE(c, PS_FMT4);
	const QuantLevels* mode_levels = reinterpret_cast<const QuantLevels*>(c.table(c.root->off_mode_levels))
	                                 + (uint32_t)(pc - 1) * (uint32_t)i_max(1, (int)c.root->block_mode_count_1plane_2plane_selected);
	WV_FOR(i, end_block_mode - start_block_mode)
	{
		ModeRec& m = modes[start_block_mode + i];
		if (m.error >= ERROR_CALC_DEFAULT)
		{
			m.error = ERROR_CALC_DEFAULT;
		}
		else
		{
			// (the levels the mode's colour bit budget allows: TableRoot::off_mode_levels)
			const QuantLevels levels = table_at(mode_levels, (uint32_t)(start_block_mode + i));
			float error_of_best = best_combination_for_levels(c, pc, fs, levels, 0, nullptr, nullptr, nullptr);
			m.error = error_of_best + m.error;
		}
	}
	WV_SYNC();

	// top-N by repeated argmin, lowest index on ties (ref: :1156-1333)
	int limit = (int)c.cfg->tune_candidate_limit;
	int count = 0;
#if WV_DEVICE
	if (end_block_mode - start_block_mode <= 128)
	{
		// up to two modes per lane: their errors stay in registers while the winners are picked (no LDS reads, no
		// hand-offs between the picks); a winner is struck out in the register of the lane that holds it
		const int i0 = start_block_mode + WV_LANE, i1 = i0 + 64;
		float e0 = i0 < end_block_mode ? modes[i0].error : ERROR_CALC_DEFAULT;
		float e1 = i1 < end_block_mode ? modes[i1].error : ERROR_CALC_DEFAULT;
		for (int n = 0; n < limit; n++)
		{
			float best = ERROR_CALC_DEFAULT;
			int idx = 0x7FFFFFFF;
			if (e0 < best) { best = e0; idx = i0; }
			if (e1 < best) { best = e1; idx = i1; }
			wave_argmin_fold(best, idx);
			if (!(best < ERROR_CALC_DEFAULT)) break;
			WV_ONE { tr.cand_block_mode[n] = idx; }
			e0 = idx == i0 ? ERROR_CALC_DEFAULT : e0;
			e1 = idx == i1 ? ERROR_CALC_DEFAULT : e1;
			count++;
		}
		WV_SYNC();
	}
	else
#endif
	for (int n = 0; n < limit; n++)
	{
		int best = wave_argmin(c, start_block_mode, end_block_mode, [&](int i) { return modes[i].error; });
		if (best < 0) break;
		WV_SYNC();
		WV_ONE
		{
			tr.cand_block_mode[n] = best;
			modes[best].error = ERROR_CALC_DEFAULT;
		}
		WV_SYNC();
		count++;
	}
	// the quant levels and formats of the winners: the same pick as in the scoring pass, now with its outputs, one lane
	// per winner (the table lookups of all of them in flight together)
	WV_FOR64(n, count)
	{
		const int mode = tr.cand_block_mode[n];
		for (int j = 0; j < 4; j++) tr.cand_formats[n][j] = 0;
		// (the mode's levels row and its record are requested side by side; the bit count only selects the "matched formats" row)
		const QuantLevels levels = table_at(mode_levels, (uint32_t)mode);
		const int bitcount = mode_bitcount(pc, c.block_mode(mode));
		(void)best_combination_for_levels(c, pc, fs, levels, bitcount, &tr.cand_quant[n], &tr.cand_quant_mod[n], tr.cand_formats[n]);
	}
	WV_ONE { tr.cand_count = count; }
	WV_SYNC();
}

} } // namespace astcd::ASTC_VARIANT
