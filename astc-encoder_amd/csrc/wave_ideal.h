// SPDX-License-Identifier: Apache-2.0
// Ideal endpoints and per-texel weights for one plane.
//   ref: compute_partition_averages_{rgb,rgba}        Source/astcenc_averages_and_directions.cpp:47-385
//        compute_avgs_and_dirs_{4,3,3_rgb,2}_comp      Source/astcenc_averages_and_directions.cpp:388-720
//        compute_ideal_colors_and_weights_{1..4}_comp  Source/astcenc_ideal_endpoints_and_weights.cpp:107-609
//        compute_ideal_colors_and_weights_{1,2}plane(s) Source/astcenc_ideal_endpoints_and_weights.cpp:612-685
//
// Vectors are handled in "component-lane space": lane j holds image channel comp(j); lanes >= ncomp
// are zero, exactly like the reference's vfloat2/vfloat3 helpers.
#pragma once
#include "wave_ctx.h"

namespace astcd { inline namespace ASTC_VARIANT {

/* Which image channels a computation runs over.  The channel list is a packed 2-bit-per-entry word
 * rather than an array: it is indexed with run-time values, and an indexed array would live in scratch
 * memory. */
struct CompSel {
	int ncomp;
	uint32_t packed;
	WV_FN int comp(int j) const { return (int)((packed >> (2 * j)) & 3u); }
	WV_FN void set(int c0, int c1, int c2, int c3) { packed = (uint32_t)(c0 | (c1 << 2) | (c2 << 4) | (c3 << 6)); }
};

/* Partition means and dominant directions -> tr.pm_avg / tr.pm_dir (component-lane space). */
WV_FN void compute_avgs_and_dirs(const Ctx& c, const PartView& pv, const CompSel& cs)
{
	TrialInfo& tr = c.tr();
	const BlkInfo& blk = c.blk();
	const int T = c.T, n = cs.ncomp, pc = pv.pcount;

	if (pc == 1)
	{
		WV_FOR(j, 4)
		{
			tr.pm_avg[0][j] = j < n ? blk.data_mean[cs.comp(j)] : 0.0f;
		}
	}
	else
	{
		// 4-accumulator masked sums for all but the last partition (ref: :67-84, :238-258)
		WV_FOR64(k, (pc - 1) * n * 4)
		{
			const uint32_t n_inv = n == 4 ? 64u : n == 3 ? 86u : n == 2 ? 128u : 256u;      // (k >> 2) / n by multiply-shift
			int l = k & 3, p = (int)((((uint32_t)k >> 2) * n_inv) >> 8), j = (k >> 2) - p * n;
			const float* d = c.data(cs.comp(j));
			float acc = 0.0f;
			for_texels_of_quarter(l, T, [&](int i)
			{
				// (`in partition p ? d : 0` as d * (1 or 0): the texel data are non-negative numbers, so d * 0 is +0 like the
				//  reference's masked lane -- and the read is not hidden behind a lane mask)
				acc = f_add_masked(acc, d[i], pv.of_texel[i] == p ? 1.0f : 0.0f);
			});
			tr.fbox[k] = acc;
		}
		WV_SYNC();
		WV_FOR(j, 4)
		{
			if (j < n)
			{
				float block_total = blk.data_mean[cs.comp(j)] * (float)T;
				float rest = block_total;
				for (int p = 0; p < pc - 1; p++)
				{
					const float* a = &tr.fbox[(p * n + j) * 4];
					float total = hadd4(a[0], a[1], a[2], a[3]);
					rest = rest - total;
					tr.pm_avg[p][j] = total / (float)pv.cnt(p);
				}
				tr.pm_avg[pc - 1][j] = rest / (float)pv.cnt(pc - 1);
			}
			else
			{
				for (int p = 0; p < pc; p++) tr.pm_avg[p][j] = 0.0f;
			}
		}
	}
	WV_SYNC();

	// sum of offsets over the texels whose component `which` is above the mean (ref: :409-433)
	// (one partition: the entries are the block's, whatever the trial -- TrialInfo::dirsum1)
	// (bit a * 4 + b for every pair of the trial's components: the columns once, then one shifted copy per row)
	uint32_t needed = 0;
	{
		uint32_t cols = 0;
		for (int b = 0; b < n; b++) cols |= 1u << cs.comp(b);
		for (int a = 0; a < n; a++) needed |= cols << (cs.comp(a) * 4);
	}
	const bool cached = pc == 1 && (wv_uniform(tr.dirsum1_mask) & needed) == needed;
	if (cached)
	{
		WV_FOR64(k, n * n)
		{
			const uint32_t n_inv = n == 4 ? 64u : n == 3 ? 86u : n == 2 ? 128u : 256u;
			const int which = (int)(((uint32_t)k * n_inv) >> 8), j = k - which * n;
			tr.fbox[which * 4 + j] = tr.dirsum1[cs.comp(which)][cs.comp(j)];
		}
	}
	else
	WV_FOR64(k, pc * n * n)
	{
		// (n is 2, 3 or 4: divisions by multiply-shift, exact for k < 128 -- the device has no integer divide)
		const uint32_t n_inv = n == 4 ? 64u : n == 3 ? 86u : n == 2 ? 128u : 256u;
		const int q = (int)(((uint32_t)k * n_inv) >> 8);
		const int p = (int)(((uint32_t)q * n_inv) >> 8);
		const int j = k - q * n, which = q - p * n;
		const float* dj = c.data(cs.comp(j));
		const float* dw = c.data(cs.comp(which));
		float avg_j = tr.pm_avg[p][j], avg_w = tr.pm_avg[p][which];
		const uint8_t* tix = pv.sorted + pv.off(p);
		float sum = 0.0f;
		for (int i = 0; i < pv.cnt(p); i++)
		{
			int t = tix[i];
			float dat_w = dw[t] - avg_w;
			float dat_j = dj[t] - avg_j;
			sum = sum + (dat_w > 0.0f ? dat_j : 0.0f);
		}
		tr.fbox[(p * 4 + which) * 4 + j] = sum;
		if (pc == 1) tr.dirsum1[cs.comp(which)][cs.comp(j)] = sum;
	}
	if (pc == 1 && !cached) { WV_ONE { tr.dirsum1_mask |= needed; } }
	WV_SYNC();

	WV_FOR64(p, pc)
	{
		float best[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
		float best_sum = 0.0f;
		for (int which = 0; which < n; which++)
		{
			float s[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
			for (int j = 0; j < n; j++) s[j] = tr.fbox[(p * 4 + which) * 4 + j];
			float prod = hadd4(s[0] * s[0], s[1] * s[1], s[2] * s[2], s[3] * s[3]);
			if (which == 0 || prod > best_sum)
			{
				best_sum = prod;
				for (int j = 0; j < 4; j++) best[j] = s[j];
			}
		}
		for (int j = 0; j < 4; j++) tr.pm_dir[p][j] = best[j];
	}
	WV_SYNC();
}

/* One-component plane: endpoints are the channel's min/max. (ref: :107-206) */
WV_FN void ideal_colors_and_weights_1comp(const Ctx& c, const PartView& pv, int plane, int component)
{
	TrialInfo& tr = c.tr();
	const BlkInfo& blk = c.blk();
	const int T = c.T, pc = pv.pcount;
	const float* d = c.data(component);
	float* w = c.ei_w(plane);
	float* wes = c.ei_wes(plane);
	float error_weight = cw_of(blk, component);

	// (one partition -- the only case on the search path, the second plane of a two-plane trial -- : the channel's range
	//  over the block by a wave reduction; minimum / maximum of finite values are exact whatever the order)
	float lo_part = 1e10f, hi_part = -1e10f;
	if (pc == 1)
	{
		WV_FOR_T(t, T)
		{
			const float value = d[t];
			lo_part = f_min(value, lo_part);
			hi_part = f_max(value, hi_part);
		}
		wv_all_minmax(lo_part, hi_part);
	}
	WV_FOR64(p, pc)
	{
		float lowvalue = 1e10f, highvalue = -1e10f;
		if (pc == 1)
		{
			lowvalue = lo_part;
			highvalue = hi_part;
		}
		else
		{
			const uint8_t* tix = pv.sorted + pv.off(p);
			for (int j = 0; j < pv.cnt(p); j++)
			{
				float value = d[tix[j]];
				lowvalue = f_min(value, lowvalue);
				highvalue = f_max(value, highvalue);
			}
		}
		if (highvalue <= lowvalue)
		{
			lowvalue = 0.0f;
			highvalue = 1e-7f;
		}
		float length = highvalue - lowvalue;
		tr.fbox[p * 4 + 0] = lowvalue;
		tr.fbox[p * 4 + 1] = 1.0f / length;
		tr.fbox[p * 4 + 2] = length * length;
		for (int k = 0; k < 4; k++)
		{
			tr.ep0[plane][p][k] = k == component ? lowvalue : blk.data_min[k];
			tr.ep1[plane][p][k] = k == component ? highvalue : blk.data_max[k];
		}
	}
	WV_SYNC();
	WV_FOR_T(t, c.Tp)
	{
		if (t < T)
		{
			int p = pv.of_texel[t];
			float value = (d[t] - tr.fbox[p * 4 + 0]) * tr.fbox[p * 4 + 1];
			w[t] = f_clamp1(value);
			wes[t] = tr.fbox[p * 4 + 2] * error_weight;
		}
		else
		{
			w[t] = 0.0f;
			wes[t] = 0.0f;
		}
	}
	WV_ONE
	{
		bool cw = true;
		for (int p = 1; p < pc; p++) cw = cw && tr.fbox[p * 4 + 2] == tr.fbox[2];
		tr.is_constant_wes[plane] = cw ? 1 : 0;
	}
	WV_SYNC();
}

/* 2-, 3- or 4-component plane: least-squares-ish line through the partition mean. (ref: :217-609) */
WV_FN void ideal_colors_and_weights_ncomp(const Ctx& c, const PartView& pv, int plane, const CompSel& cs, float error_weight)
{
	TrialInfo& tr = c.tr();
	const BlkInfo& blk = c.blk();
	const int T = c.T, pc = pv.pcount, n = cs.ncomp;
	float* w = c.ei_w(plane);
	float* wes = c.ei_wes(plane);

	compute_avgs_and_dirs(c, pv, cs);

	// normalised line direction per partition -> fbox[32 + p*4 + j]
	WV_FOR64(p, pc)
	{
		f4 dir = load4(tr.pm_dir[p]);
		float s = n == 2 ? hadd_s(dir) : hadd_rgb_s(dir);
		if (s < 0.0f)
		{
			dir = splat4(0.0f) - dir;
		}
		f4 safe = n == 4 ? unit4() : n == 3 ? unit3() : unit2();
		f4 b = normalize_safe4(dir, safe);
		store4(&tr.fbox[32 + p * 4], b);
	}
	WV_SYNC();

	// raw line parameter of every texel (ref: :282-291, :431-440, :553-562); with one partition its range is folded
	// across the wave on the way (minimum / maximum of finite values: exact whatever the order)
	float lo_part = 1e10f, hi_part = -1e10f;
	WV_FOR_T(t, T)
	{
		int p = pv.of_texel[t];
		f4 pt = mk4(0.0f, 0.0f, 0.0f, 0.0f);
		for (int j = 0; j < n; j++) set_lane(pt, j, c.data(cs.comp(j))[t]);
		f4 a = load4(tr.pm_avg[p]);
		f4 b = load4(&tr.fbox[32 + p * 4]);
		float param = n == 3 ? dot3_s(pt - a, b) : dot_s(pt - a, b);
		w[t] = param;
		lo_part = f_min(param, lo_part);
		hi_part = f_max(param, hi_part);
	}
	if (pc == 1) wv_all_minmax(lo_part, hi_part);
	WV_SYNC();

	WV_FOR64(p, pc)
	{
		float lowparam = 1e10f, highparam = -1e10f;
		if (pc == 1)
		{
			lowparam = lo_part;
			highparam = hi_part;
		}
		else
		{
			const uint8_t* tix = pv.sorted + pv.off(p);
			for (int j = 0; j < pv.cnt(p); j++)
			{
				float param = w[tix[j]];
				lowparam = f_min(param, lowparam);
				highparam = f_max(param, highparam);
			}
		}
		if (highparam <= lowparam)
		{
			lowparam = 0.0f;
			highparam = 1e-7f;
		}
		float length = highparam - lowparam;
		tr.fbox[p * 4 + 0] = lowparam;
		tr.fbox[p * 4 + 1] = 1.0f / length;
		tr.fbox[p * 4 + 2] = length * length;

		f4 a = load4(tr.pm_avg[p]);
		f4 b = load4(&tr.fbox[32 + p * 4]);
		f4 lo = a + b * lowparam;
		f4 hi = a + b * highparam;
		for (int k = 0; k < 4; k++)
		{
			tr.ep0[plane][p][k] = blk.data_min[k];
			tr.ep1[plane][p][k] = blk.data_max[k];
		}
		for (int j = 0; j < n; j++)
		{
			tr.ep0[plane][p][cs.comp(j)] = lane(lo, j);
			tr.ep1[plane][p][cs.comp(j)] = lane(hi, j);
		}
	}
	WV_SYNC();

	WV_FOR_T(t, c.Tp)
	{
		if (t < T)
		{
			int p = pv.of_texel[t];
			float idx = (w[t] - tr.fbox[p * 4 + 0]) * tr.fbox[p * 4 + 1];
			w[t] = f_clamp1(idx);
			wes[t] = tr.fbox[p * 4 + 2] * error_weight;
		}
		else
		{
			w[t] = 0.0f;
			wes[t] = 0.0f;
		}
	}
	WV_ONE
	{
		bool cw = true;
		for (int p = 1; p < pc; p++) cw = cw && tr.fbox[p * 4 + 2] == tr.fbox[2];
		tr.is_constant_wes[plane] = cw ? 1 : 0;
	}
	WV_SYNC();
}

WV_FN bool is_constant_channel(const BlkInfo& blk, int ch) { return blk.data_min[ch] == blk.data_max[ch]; }

/* (ref: compute_ideal_colors_and_weights_1plane :612) */
WV_FN void ideal_colors_and_weights_1plane(const Ctx& c, const PartView& pv)
{
	const BlkInfo& blk = c.blk();
	bool uses_alpha = !is_constant_channel(blk, 3);
	CompSel cs;
	float ew;
	if (uses_alpha)
	{
		cs.ncomp = 4; cs.set(0, 1, 2, 3);
		ew = hadd4(cw_of(blk, 0), cw_of(blk, 1), cw_of(blk, 2), cw_of(blk, 3)) / 4.0f;
	}
	else
	{
		cs.ncomp = 3; cs.set(0, 1, 2, 0);
		ew = hadd4(cw_of(blk, 0), cw_of(blk, 1), cw_of(blk, 2), 0.0f) * (1.0f / 3.0f);
	}
	ideal_colors_and_weights_ncomp(c, pv, 0, cs, ew);
}

/* (ref: compute_ideal_colors_and_weights_2planes :630) plane 0 = everything but the separated
 * component, plane 1 = the separated component. */
WV_FN void ideal_colors_and_weights_2planes(const Ctx& c, const PartView& pv, int plane2_component)
{
	const BlkInfo& blk = c.blk();
	bool uses_alpha = !is_constant_channel(blk, 3);
	CompSel cs;
	float ew;
	if (uses_alpha || plane2_component == 3)
	{
		// three remaining components; NB the reference's weight for omitted component 0 uses
		// channels <0,1,2> (ref: ideal_endpoints_and_weights.cpp:373-404)
		cs.ncomp = 3;
		// the three channels other than plane2_component, ascending
		cs.set(plane2_component == 0 ? 1 : 0, plane2_component <= 1 ? 2 : 1, plane2_component <= 2 ? 3 : 2, 0);
		float a, b, d;
		switch (plane2_component)
		{
		case 0: a = cw_of(blk, 0); b = cw_of(blk, 1); d = cw_of(blk, 2); break;
		case 1: a = cw_of(blk, 0); b = cw_of(blk, 2); d = cw_of(blk, 3); break;
		case 2: a = cw_of(blk, 0); b = cw_of(blk, 1); d = cw_of(blk, 3); break;
		default: a = cw_of(blk, 0); b = cw_of(blk, 1); d = cw_of(blk, 2); break;
		}
		ew = hadd4(a, b, d, 0.0f) * (1.0f / 3.0f);
	}
	else
	{
		cs.ncomp = 2;
		// the two colour channels other than plane2_component (0..2), ascending
		const int c0 = plane2_component == 0 ? 1 : 0, c1 = plane2_component <= 1 ? 2 : 1;
		cs.set(c0, c1, 0, 0);
		ew = hadd4(cw_of(blk, c0), cw_of(blk, c1), 0.0f, 0.0f) / 2.0f;
	}
	ideal_colors_and_weights_ncomp(c, pv, 0, cs, ew);
	ideal_colors_and_weights_1comp(c, pv, 1, plane2_component);
}

} } // namespace astcd::ASTC_VARIANT
