// SPDX-License-Identifier: Apache-2.0
// Candidate refinement: re-fit endpoints to quantized weights, score by decoding, nudge weights.
//   ref: recompute_ideal_colors_{1plane,2planes}, compute_rgbo_vector
//                                   Source/astcenc_ideal_endpoints_and_weights.cpp:1099-1650
//        unpack_weights, lerp_color_int, compute_symbolic_block_difference_{1plane_1partition,1plane,2plane}
//                                   Source/astcenc_decompress_symbolic.cpp:37-155, :313-618
//        realign_weights_{undecimated,decimated}
//                                   Source/astcenc_compress_symbolic.cpp:69-338
#pragma once
#include "wave_ctx.h"
#include "wave_weights.h"
#include "wave_color.h"

namespace astcd { inline namespace ASTC_VARIANT {

/* (ref: compute_rgbo_vector :1099) */
WV_FN f4 compute_rgbo_vector(f4 rgba_weight_sum, f4 weight_weight_sum, f4 rgbq_sum, float psum)
{
	float X = rgba_weight_sum.x, Y = rgba_weight_sum.y, Z = rgba_weight_sum.z;
	float P = weight_weight_sum.x, Q = weight_weight_sum.y, R = weight_weight_sum.z;
	float S = psum;

	float PP = P * P, QQ = Q * Q, RR = R * R;
	float SZmRR = S * Z - RR;
	float DT = SZmRR * Y - Z * QQ;
	float YP = Y * P, QX = Q * X, YX = Y * X;
	float mZYP = -Z * YP, mZQX = -Z * QX, mRYX = -R * YX;
	float ZQP = Z * Q * P, RYP = R * YP, RQX = R * QX;

	float rdet = 1.0f / (DT * X + mZYP * P);

	f4 mat0 = mk4(DT, ZQP, RYP, mZYP);
	f4 mat1 = mk4(ZQP, SZmRR * X - Z * PP, RQX, mZQX);
	f4 mat2 = mk4(RYP, RQX, (S * Y - QQ) * X - Y * PP, mRYX);
	f4 mat3 = mk4(mZYP, mZQX, mRYX, Z * YX);
	f4 vect = rgbq_sum * rdet;

	return mk4(dot_s(mat0, vect), dot_s(mat1, vect), dot_s(mat2, vect), dot_s(mat3, vect));
}

/* Expand quantized grid weights (0..64 in `uq`) to per-texel float weights in dst[0..T). */
WV_FN void expand_weights(const Ctx& c, const DecView& di, const uint8_t* uq, float* grid, float* dst)
{
	(void)c;
	const int T = di.T, W = di.W;
	WV_FOR64(i, W) { grid[i] = (float)uq[i] * (1.0f / 64.0f); }
	WV_SYNC();
	const uint8_t* tw = di.tw;
	const float* tcf = di.tcf;
	if (di.max_texel_weight_count == 1)
	{
		WV_FOR_T(t, T) { dst[t] = grid[t]; }
	}
	else if (di.max_texel_weight_count <= 2)
	{
		WV_FOR_T(t, T) { dst[t] = infill2(grid, tw, tcf, T, t); }
	}
	else
	{
		WV_FOR_T(t, T) { dst[t] = infill4(grid, tw, tcf, T, t); }
	}
	WV_SYNC();
}

/* Endpoint solve shared by both recompute variants: given the accumulated sums of one weight
 * plane, produce (ep0, ep1, mask) for the lanes it owns. */
struct PlaneSolve {
	f4 ep0, ep1;
	uint32_t mask;            // bit k: lane k solved (a packed word: a run-time indexed bool[4] would live in scratch)
	WV_FN bool m(int k) const { return ((mask >> k) & 1u) != 0; }
};

WV_FN PlaneSolve solve_plane(f4 left_sum, f4 middle_sum, f4 right_sum, f4 color_vec_x, f4 color_vec_y)
{
	PlaneSolve r;
	f4 color_det1 = (left_sum * right_sum) - (middle_sum * middle_sum);
	f4 color_rdet1 = splat4(1.0f) / color_det1;
	f4 color_mss1 = (left_sum * left_sum) + (splat4(2.0f) * middle_sum * middle_sum) + (right_sum * right_sum);
	r.ep0 = (right_sum * color_vec_x - middle_sum * color_vec_y) * color_rdet1;
	r.ep1 = (left_sum * color_vec_y - middle_sum * color_vec_x) * color_rdet1;
	f4 ad = v4_abs(color_det1);
	f4 th = color_mss1 * 1e-4f;
	r.mask = 0;
	for (int k = 0; k < 4; k++)
	{
		float e0 = lane(r.ep0, k), e1 = lane(r.ep1, k);
		if ((lane(ad, k) > lane(th, k)) && (e0 == e0) && (e1 == e1)) r.mask |= 1u << k;
	}
	return r;
}

/* The re-fit's solve for one (partition, channel) of a one-plane candidate (ref: recompute_ideal_colors_1plane :1271-1340;
 * every quantity is channel-wise in the reference's vector code, the per-partition scalars are simply recomputed by the
 * four lanes of a partition).  `s`: the partition's sums -- 0 wmin 1 wmax 2 scale_min 3 scale_max 4-6 left / middle / right
 * 7 weight_weight_sum 8-11 color_vec_x 12-15 color_vec_y 16-17 scale_vec; ep0 / ep1: in = the endpoints so far (kept when
 * the solve is degenerate), out = the new ones; rgbs: lane ch of the RGB + scale vector. */
WV_FN void refit_solve_1plane(const float* s, const BlkInfo& blk, float scale_dir, int texels, float ls_weight, int ch,
                              float& ep0, float& ep1, float& rgbs)
{
	const float wmin1 = s[0], wmax1 = s[1], scale_min = s[2], scale_max = s[3];
	const float left_sum_s = s[4], middle_sum_s = s[5], right_sum_s = s[6];
	const float color_weight = cw_of(blk, ch);
	const float cwn = color_weight * (float)texels;
	const float rgba_weight_sum = cwn > 1e-17f ? cwn : 1e-17f;

	const float left_sum = left_sum_s * color_weight;
	const float middle_sum = middle_sum_s * color_weight;
	const float right_sum = right_sum_s * color_weight;
	const float lm_x = left_sum_s * ls_weight, lm_y = middle_sum_s * ls_weight, lm_z = right_sum_s * ls_weight;

	const float color_vec_x = s[8 + ch] * color_weight;
	const float color_vec_y = s[12 + ch] * color_weight;

	float scalediv = scale_min / f_max(scale_max, 1e-10f);
	scalediv = f_clamp1(scalediv);
	// lane ch of rgbs: scale_dir * scale for RGB, the scale ratio for A
	rgbs = ch < 3 ? scale_dir * scale_max : scalediv;

	if (wmin1 >= wmax1 * 0.999f)
	{
		// all weights (nearly) equal: both endpoints become the mean
		const float avg = (color_vec_x + color_vec_y) / rgba_weight_sum;
		if (avg == avg) { ep0 = avg; ep1 = avg; }
		if (ch == 3) rgbs = 1.0f;
	}
	else
	{
		// (ref: solve as in solve_plane(), one channel)
		const float color_det1 = (left_sum * right_sum) - (middle_sum * middle_sum);
		const float color_rdet1 = 1.0f / color_det1;
		const float color_mss1 = (left_sum * left_sum) + (2.0f * middle_sum * middle_sum) + (right_sum * right_sum);
		const float e0 = (right_sum * color_vec_x - middle_sum * color_vec_y) * color_rdet1;
		const float e1 = (left_sum * color_vec_y - middle_sum * color_vec_x) * color_rdet1;
		if ((f_abs(color_det1) > (color_mss1 * 1e-4f)) && (e0 == e0) && (e1 == e1)) { ep0 = e0; ep1 = e1; }

		const float scale_vec0 = s[16], scale_vec1 = s[17];
		const float ls_det1 = (lm_x * lm_z) - (lm_y * lm_y);
		const float ls_rdet1 = 1.0f / ls_det1;
		const float ls_mss1 = (lm_x * lm_x) + (2.0f * lm_y * lm_y) + (lm_z * lm_z);
		const float scale_ep0 = (lm_z * scale_vec0 - lm_y * scale_vec1) * ls_rdet1;
		const float scale_ep1 = (lm_x * scale_vec1 - lm_y * scale_vec0) * ls_rdet1;

		if (f_abs(ls_det1) > (ls_mss1 * 1e-4f) && scale_ep0 == scale_ep0 && scale_ep1 == scale_ep1 && scale_ep0 < scale_ep1)
		{
			rgbs = ch < 3 ? scale_dir * scale_ep1 : scale_ep0 / scale_ep1;
		}
	}
}

/* The HDR RGB + offset vector of one partition of a one-plane candidate (ref: :1342-1364); v0 / v1: the new endpoints. */
WV_FN f4 refit_rgbo_1plane(const float* s, const BlkInfo& blk, int texels, f4 v0, f4 v1)
{
	const float right_sum_s = s[6], weight_weight_sum_s = s[7];
	f4 color_weight = cw4_of(blk);
	f4 rgba_weight_sum = v4_max(color_weight * (float)texels, splat4(1e-17f));
	f4 color_vec_x = load4(&s[8]) * color_weight, color_vec_y = load4(&s[12]) * color_weight;
	f4 weight_weight_sum = splat4(weight_weight_sum_s) * color_weight;
	float psum = right_sum_s * hadd_rgb_s(color_weight);
	f4 rgbq_sum = color_vec_x + color_vec_y;
	rgbq_sum.w = hadd_rgb_s(color_vec_y);
	f4 rgbovec = compute_rgbo_vector(rgba_weight_sum, weight_weight_sum, rgbq_sum, psum);
	if (f_isnan(dot_s(rgbovec, rgbovec)))
	{
		float avgdif = hadd_rgb_s(v1 - v0) * (1.0f / 3.0f);
		avgdif = f_max(avgdif, 0.0f);
		f4 avg = (v0 + v1) * 0.5f;
		f4 e0 = avg - splat4(avgdif) * 0.5f;
		rgbovec = mk4(e0.x, e0.y, e0.z, avgdif);
	}
	return rgbovec;
}

/* The same for one channel of a two-plane candidate (ref: recompute_ideal_colors_2planes :1514-1612).  `s`: 0 wmin1 1 wmax1
 * 2 wmin2 3 wmax2 4 scale_min 5 scale_max 6-8 left / middle / right of plane 1, 9-11 of plane 2, 12-15 color_vec_x
 * 16-19 color_vec_y 20-21 scale_vec 22-25 weight_weight_sum. */
WV_FN void refit_solve_2planes(const float* s, const BlkInfo& blk, float scale_dir_ch, int T, float ls_weight, int ch, int plane2_component,
                               float& ep0, float& ep1, float& rgbs)
{
	const bool second = ch == plane2_component;
	const float wmin1 = s[0], wmax1 = s[1], scale_min = s[4], scale_max = s[5];
	const float wmin = second ? s[2] : wmin1, wmax = second ? s[3] : wmax1;
	const float color_weight = cw_of(blk, ch);
	const float cwn = color_weight * (float)T;
	const float rgba_weight_sum = cwn > 1e-17f ? cwn : 1e-17f;
	const float left_sum = (second ? s[9] : s[6]) * color_weight;
	const float middle_sum = (second ? s[10] : s[7]) * color_weight;
	const float right_sum = (second ? s[11] : s[8]) * color_weight;
	const float color_vec_x = s[12 + ch] * color_weight;
	const float color_vec_y = s[16 + ch] * color_weight;

	if (wmin >= wmax * 0.999f)
	{
		// all weights of the channel's plane (nearly) equal: both endpoints become the mean
		const float avg = (color_vec_x + color_vec_y) / rgba_weight_sum;
		if (avg == avg) { ep0 = avg; ep1 = avg; }
	}
	else
	{
		const float color_det1 = (left_sum * right_sum) - (middle_sum * middle_sum);
		const float color_rdet1 = 1.0f / color_det1;
		const float color_mss1 = (left_sum * left_sum) + (2.0f * middle_sum * middle_sum) + (right_sum * right_sum);
		const float e0 = (right_sum * color_vec_x - middle_sum * color_vec_y) * color_rdet1;
		const float e1 = (left_sum * color_vec_y - middle_sum * color_vec_x) * color_rdet1;
		if ((f_abs(color_det1) > (color_mss1 * 1e-4f)) && (e0 == e0) && (e1 == e1)) { ep0 = e0; ep1 = e1; }
	}

	// the RGB + scale vector follows plane 1 (lane ch: scale_dir * scale for RGB, the scale ratio for A)
	float scalediv = scale_min / f_max(scale_max, 1e-10f);
	scalediv = f_clamp1(scalediv);
	rgbs = ch < 3 ? scale_dir_ch * scale_max : scalediv;
	if (wmin1 >= wmax1 * 0.999f)
	{
		if (ch == 3) rgbs = 1.0f;
	}
	else
	{
		const float lm_x = s[6] * ls_weight, lm_y = s[7] * ls_weight, lm_z = s[8] * ls_weight;
		const float scale_vec0 = s[20], scale_vec1 = s[21];
		const float ls_det1 = (lm_x * lm_z) - (lm_y * lm_y);
		const float ls_rdet1 = 1.0f / ls_det1;
		const float ls_mss1 = (lm_x * lm_x) + (2.0f * lm_y * lm_y) + (lm_z * lm_z);
		const float scale_ep0 = (lm_z * scale_vec0 - lm_y * scale_vec1) * ls_rdet1;
		const float scale_ep1 = (lm_x * scale_vec1 - lm_y * scale_vec0) * ls_rdet1;
		if (f_abs(ls_det1) > (ls_mss1 * 1e-4f) && scale_ep0 == scale_ep0 && scale_ep1 == scale_ep1 && scale_ep0 < scale_ep1)
		{
			rgbs = ch < 3 ? scale_dir_ch * scale_ep1 : scale_ep0 / scale_ep1;
		}
	}
}

/* ... and its HDR RGB + offset vector (ref: :1614-1640). */
WV_FN f4 refit_rgbo_2planes(const float* s, const BlkInfo& blk, int T, int plane2_component, f4 v0, f4 v1)
{
	const f4 color_weight = cw4_of(blk);
	const f4 rgba_weight_sum = v4_max(color_weight * (float)T, splat4(1e-17f));
	const f4 right1_sum = splat4(s[8]) * color_weight, right2_sum = splat4(s[11]) * color_weight;
	const f4 color_vec_x = load4(&s[12]) * color_weight;
	const f4 color_vec_y = load4(&s[16]) * color_weight;
	const f4 weight_weight_sum = load4(&s[22]) * color_weight;
	f4 sel = mk4(plane2_component == 0 ? right2_sum.x : right1_sum.x, plane2_component == 1 ? right2_sum.y : right1_sum.y,
	             plane2_component == 2 ? right2_sum.z : right1_sum.z, plane2_component == 3 ? right2_sum.w : right1_sum.w);
	float psum = dot3_s(sel, color_weight);
	f4 rgbq_sum = color_vec_x + color_vec_y;
	rgbq_sum.w = hadd_rgb_s(color_vec_y);
	f4 rgbovec = compute_rgbo_vector(rgba_weight_sum, weight_weight_sum, rgbq_sum, psum);
	if (f_isnan(dot_s(rgbovec, rgbovec)))
	{
		float avgdif = hadd_rgb_s(v1 - v0) * (1.0f / 3.0f);
		avgdif = f_max(avgdif, 0.0f);
		f4 avg = (v0 + v1) * 0.5f;
		f4 e0 = avg - splat4(avgdif) * 0.5f;
		rgbovec = mk4(e0.x, e0.y, e0.z, avgdif);
	}
	return rgbovec;
}

/* Scale direction of every partition of the trial (ref: recompute_ideal_colors_1plane :1198-1219, _2planes :1433-1437):
 * the normalised colour sum of the partition's texels.  It depends on the block and the partitioning only, not on
 * the weights, so it is computed once per trial (the reference recomputes the same values in every refinement step)
 * into tr.pm_dir, whose search-phase contents are dead by then. */
WV_FN void trial_scale_directions(const Ctx& c, const PartView& pv, bool dual)
{
	TrialInfo& tr = c.tr();
	const BlkInfo& blk = c.blk();
	const int pc = pv.pcount;
	if (dual)
	{
		WV_ONE { store4(tr.pm_dir[0], normalize4(xyz0(load4(blk.data_mean)))); }
		WV_SYNC();
		return;
	}
	if (pc > 1)
	{
		WV_FOR64(k, pc * 4)
		{
			int p = k >> 2, ch = k & 3;
			const float* d = c.data(ch);
			const uint8_t* tix = pv.sorted + pv.off(p);
			float s = 0.0f;
			for (int j = 0; j < pv.cnt(p); j++) s += d[tix[j]];
			tr.fbox[96 + k] = s;
		}
	}
	else
	{
		WV_FOR(k, 4) { tr.fbox[96 + k] = blk.data_mean[k] * (float)c.T; }
	}
	WV_SYNC();
	WV_FOR64(p, pc)
	{
		f4 rgba_sum = load4(&tr.fbox[96 + p * 4]) * cw4_of(blk);
		f4 rgba_weight_sum = v4_max(cw4_of(blk) * (float)pv.cnt(p), splat4(1e-17f));
		f4 scale_dir = normalize4(xyz0(rgba_sum / rgba_weight_sum));
		store4(tr.pm_dir[p], scale_dir);
	}
	WV_SYNC();
}

/* (ref: recompute_ideal_colors_1plane :1146).  Reads wscb().weights, updates tr.wep0/wep1/rgbs/rgbo. */
WV_FN void recompute_ideal_colors_1plane(const Ctx& c, const PartView& pv, const DecView& di)
{
	TrialInfo& tr = c.tr();
	const BlkInfo& blk = c.blk();
	const int T = c.T, pc = pv.pcount;
	float* undec = c.tsc_r(0);
	expand_weights(c, di, c.wscb().weights, c.wsc(0), undec);

	// (pass 1, the scale direction of each partition, is per trial: trial_scale_directions() -> tr.pm_dir)

	// pass 2 (ref: :1241-1269): the reference accumulates 15 running sums per partition in partition-texel order.
	// Per-texel terms are produced lane-parallel (row r, position i in the partition-sorted order), then each chain
	// is summed sequentially -- additions only -- by its own lane.
	const float ls_weight = hadd_rgb_s(cw4_of(blk));
	// Running minima / maxima of the weights and of the scale projection (ref: :1230-1235, :1247-1253).  With a single
	// partition they are wave-wide reductions of per-lane partials (exact: the values are finite); with several
	// partitions the chain lanes of rows 0 and 1 pick them up below.
	float wmin_part = 1.0f, wmax_part = 0.0f, smin_part = 1e10f, smax_part = 0.0f;
	WV_FOR_T(i, T)
	{
		int t = pv.sorted[i];
		int p = pv.of_texel[t];
		f4 scale_dir = load4(tr.pm_dir[p]);
		f4 rgba = mk4(c.data(0)[t], c.data(1)[t], c.data(2)[t], c.data(3)[t]);
		float idx0 = undec[t];
		float om_idx0 = 1.0f - idx0;
		float scale = dot3_s(scale_dir, rgba);
		wmin_part = idx0 < wmin_part ? idx0 : wmin_part; wmax_part = idx0 > wmax_part ? idx0 : wmax_part;
		smin_part = scale < smin_part ? scale : smin_part; smax_part = scale > smax_part ? scale : smax_part;
		c.rsc(0)[i] = idx0;
		c.rsc(1)[i] = scale;
		c.rsc(2)[i] = om_idx0 * om_idx0;
		c.rsc(3)[i] = om_idx0 * idx0;
		c.rsc(4)[i] = idx0 * idx0;
		f4 cwiprod = rgba * splat4(idx0);
		f4 xterm = rgba - cwiprod;
		c.rsc(5)[i] = xterm.x; c.rsc(6)[i] = xterm.y; c.rsc(7)[i] = xterm.z; c.rsc(8)[i] = xterm.w;
		c.rsc(9)[i] = cwiprod.x; c.rsc(10)[i] = cwiprod.y; c.rsc(11)[i] = cwiprod.z; c.rsc(12)[i] = cwiprod.w;
		c.rsc(13)[i] = om_idx0 * (scale * ls_weight);
		c.rsc(14)[i] = idx0 * (scale * ls_weight);
	}
	WV_SYNC();
	if (pc == 1)
	{
		wv_all_minmax(wmin_part, wmax_part, smin_part, smax_part);
		WV_ONE { tr.fbox[0] = wmin_part; tr.fbox[1] = wmax_part; tr.fbox[2] = smin_part; tr.fbox[3] = smax_part; }
		// the 14 sums: additions only, four loads in flight
		WV_FOR(k, 14)
		{
			const int r = k >= 1 ? k + 1 : k;                           // rows 0, 2..14 (row 1 only feeds the scale min / max)
			const float* v = c.rsc(r);
			float acc = r == 0 ? 1e-17f : 0.0f;   // row 0 is weight_weight_sum (starts at 1e-17)
			int j = 0;
			for (; j + 4 <= T; j += 4)
			{
				const float x0 = v[j], x1 = v[j + 1], x2 = v[j + 2], x3 = v[j + 3];
				acc += x0; acc += x1; acc += x2; acc += x3;
			}
			for (; j < T; j++) acc += v[j];
			float* s = tr.fbox;
			if (r == 0) s[7] = acc;                                     // weight_weight_sum
			else if (r <= 4) s[4 + (r - 2)] = acc;                      // left, middle, right
			else if (r <= 12) s[8 + (r - 5)] = acc;                     // color_vec_x[4], color_vec_y[4]
			else s[16 + (r - 13)] = acc;                                // scale_vec
		}
	}
	else
	{
		WV_FOR64(k, pc * 15)
		{
			int p = k / 15, r = k % 15;
			const float* v = c.rsc(r) + pv.off(p);
			const int n = pv.cnt(p);
			// one branch-free loop for all chains: running sum, min and max of the row
			float acc = r == 0 ? 1e-17f : 0.0f;   // row 0 doubles as weight_weight_sum (starts at 1e-17)
			float mn = r == 0 ? 1.0f : 1e10f;
			float mx = 0.0f;
			for (int j = 0; j < n; j++)
			{
				float x = v[j];
				acc += x;
				mn = x < mn ? x : mn;
				mx = x > mx ? x : mx;
			}
			float* s = &tr.fbox[p * 24];
			if (r == 0) { s[0] = mn; s[1] = mx; s[7] = acc; }       // wmin1, wmax1; weight_weight_sum
			else if (r == 1) { s[2] = mn; s[3] = mx; }              // scale_min, scale_max
			else if (r <= 4) s[4 + (r - 2)] = acc;                  // left, middle, right
			else if (r <= 12) s[8 + (r - 5)] = acc;                 // color_vec_x[4], color_vec_y[4]
			else s[16 + (r - 13)] = acc;                            // scale_vec
		}
	}
	WV_SYNC();

	// the solve, one lane per (partition, channel): every quantity below is channel-wise in the reference's vector
	// code (ref: :1271-1340), the per-partition scalars are simply recomputed by the four lanes of a partition
	WV_FOR64(k, pc * 4)
	{
		const int p = k >> 2, ch = k & 3;
		float ep0 = tr.wep0[p][ch], ep1 = tr.wep1[p][ch], rgbs;
		refit_solve_1plane(&tr.fbox[p * 24], blk, tr.pm_dir[p][ch], pv.cnt(p), ls_weight, ch, ep0, ep1, rgbs);
		tr.wep0[p][ch] = ep0;
		tr.wep1[p][ch] = ep1;
		tr.rgbs[p][ch] = rgbs;
	}
	WV_SYNC();
	if (kHdr && (blk.rgb_lns || blk.alpha_lns))
	{
		WV_FOR64(p, pc)
		{
			store4(tr.rgbo[p], refit_rgbo_1plane(&tr.fbox[p * 24], blk, pv.cnt(p), load4(tr.wep0[p]), load4(tr.wep1[p])));
		}
		WV_SYNC();
	}
}

/* (ref: recompute_ideal_colors_2planes :1369) */
WV_FN void recompute_ideal_colors_2planes(const Ctx& c, const DecView& di, int plane2_component)
{
	TrialInfo& tr = c.tr();
	const BlkInfo& blk = c.blk();
	const int T = c.T;
	float* undec1 = c.tsc_r(0);
	float* undec2 = c.tsc_r(1);
	expand_weights(c, di, c.wscb().weights, c.wsc(0), undec1);
	expand_weights(c, di, c.wscb().weights + PLANE2_OFFSET, c.wsc(1), undec2);

	const float ls_weight = hadd_rgb_s(cw4_of(blk));
	const f4 scale_dir = load4(tr.pm_dir[0]);              // trial_scale_directions()

	// running minima / maxima (ref: :1455-1462): per-lane partials, reduced over the wave after the texel pass
	float wmin1_part = 1.0f, wmax1_part = 0.0f, wmin2_part = 1.0f, wmax2_part = 0.0f, smin_part = 1e10f, smax_part = 0.0f;

	// Per-texel terms lane-parallel, then one sequential lane per running sum (ref: :1474-1512).
	// rows: 0 idx0, 1 idx1, 3-5 l/m/r plane1, 6-8 l/m/r plane2, 9-12 x, 13-16 y, 17-18 scale_vec
	WV_FOR_T(j, T)
	{
		f4 rgba = mk4(c.data(0)[j], c.data(1)[j], c.data(2)[j], c.data(3)[j]);
		float idx0 = undec1[j], idx1 = undec2[j];
		float om_idx0 = 1.0f - idx0, om_idx1 = 1.0f - idx1;
		float scale = dot3_s(scale_dir, rgba);
		wmin1_part = idx0 < wmin1_part ? idx0 : wmin1_part; wmax1_part = idx0 > wmax1_part ? idx0 : wmax1_part;
		wmin2_part = idx1 < wmin2_part ? idx1 : wmin2_part; wmax2_part = idx1 > wmax2_part ? idx1 : wmax2_part;
		smin_part = scale < smin_part ? scale : smin_part; smax_part = scale > smax_part ? scale : smax_part;
		c.rsc(0)[j] = idx0; c.rsc(1)[j] = idx1;
		c.rsc(3)[j] = om_idx0 * om_idx0; c.rsc(4)[j] = om_idx0 * idx0; c.rsc(5)[j] = idx0 * idx0;
		c.rsc(6)[j] = om_idx1 * om_idx1; c.rsc(7)[j] = om_idx1 * idx1; c.rsc(8)[j] = idx1 * idx1;
		f4 color_idx = mk4(plane2_component == 0 ? idx1 : idx0, plane2_component == 1 ? idx1 : idx0,
		                   plane2_component == 2 ? idx1 : idx0, plane2_component == 3 ? idx1 : idx0);
		f4 cwiprod = rgba * color_idx;
		f4 xterm = rgba - cwiprod;
		c.rsc(9)[j] = xterm.x; c.rsc(10)[j] = xterm.y; c.rsc(11)[j] = xterm.z; c.rsc(12)[j] = xterm.w;
		c.rsc(13)[j] = cwiprod.x; c.rsc(14)[j] = cwiprod.y; c.rsc(15)[j] = cwiprod.z; c.rsc(16)[j] = cwiprod.w;
		c.rsc(17)[j] = om_idx0 * (ls_weight * scale);
		c.rsc(18)[j] = idx0 * (ls_weight * scale);
	}
	WV_SYNC();
	// outputs: fbox 0 wmin1 1 wmax1 2 wmin2 3 wmax2 4 scale_min 5 scale_max 6-8 lmr1 9-11 lmr2
	//          12-15 color_vec_x 16-19 color_vec_y 20-21 scale_vec 22-25 weight_weight_sum
	wv_all_minmax(wmin1_part, wmax1_part, wmin2_part, wmax2_part);
	wv_all_minmax(smin_part, smax_part);
	WV_ONE
	{
		tr.fbox[0] = wmin1_part; tr.fbox[1] = wmax1_part; tr.fbox[2] = wmin2_part; tr.fbox[3] = wmax2_part;
		tr.fbox[4] = smin_part; tr.fbox[5] = smax_part;
	}
	WV_FOR(k, 18)
	{
		const int r = k >= 2 ? k + 1 : k;        // rows 0, 1, 3..18 (the scale projection only feeds the min / max)
		const float* v = c.rsc(r);
		float acc = r <= 1 ? 1e-17f : 0.0f;      // rows 0/1 are the per-plane weight sums
		int j = 0;
		for (; j + 4 <= T; j += 4)
		{
			const float x0 = v[j], x1 = v[j + 1], x2 = v[j + 2], x3 = v[j + 3];
			acc += x0; acc += x1; acc += x2; acc += x3;
		}
		for (; j < T; j++) acc += v[j];
		if (r == 0) tr.fbox[26] = acc;
		else if (r == 1) tr.fbox[27] = acc;
		else if (r <= 8) tr.fbox[6 + (r - 3)] = acc;
		else if (r <= 16) tr.fbox[12 + (r - 9)] = acc;
		else tr.fbox[20 + (r - 17)] = acc;
	}
	WV_SYNC();
	WV_FOR(k, 4) { tr.fbox[22 + k] = k == plane2_component ? tr.fbox[27] : tr.fbox[26]; }
	WV_SYNC();

	// the solve, one lane per channel (every quantity below is channel-wise in the reference's vector code, ref: :1514-1612;
	// the per-plane scalars are simply recomputed by the four lanes): a channel belongs to plane 2 if it is the separated
	// component, else to plane 1
	WV_FOR(ch, 4)
	{
		float ep0 = tr.wep0[0][ch], ep1 = tr.wep1[0][ch], rgbs;
		refit_solve_2planes(tr.fbox, blk, tr.pm_dir[0][ch], T, ls_weight, ch, plane2_component, ep0, ep1, rgbs);
		tr.wep0[0][ch] = ep0;
		tr.wep1[0][ch] = ep1;
		tr.rgbs[0][ch] = rgbs;
	}
	WV_SYNC();

	// HDR endpoints: the rgbo vector (ref: :1614-1640).  Its inputs are read back from the sums in LDS instead of being
	// kept in registers across the two solves above (the HDR build of this function needed callee-saved registers,
	// i.e. a scratch frame, for them).
	if (kHdr && (blk.rgb_lns || blk.alpha_lns))
	{
		WV_ONE { store4(tr.rgbo[0], refit_rgbo_2planes(tr.fbox, blk, T, plane2_component, load4(tr.wep0[0]), load4(tr.wep1[0]))); }
		WV_SYNC();
	}
}

// ---------------------------------------------------------------------------------------------
// Decode-and-score
// ---------------------------------------------------------------------------------------------

/* Integer texel weights of one plane (ref: unpack_weights :89).  `taps`: how many grid weights a texel of this grid
 * interpolates at most (1, 2 or 4); the taps beyond that have contribution 0, so leaving them out changes nothing. */
WV_FN int unpack_texel_weight(const uint8_t* uq, const uint8_t* tw, const uint8_t* tci, int T, int t, int taps)
{
	if (taps == 1) return uq[t];                  // an undecimated grid: one tap of weight 16, (8 + 16 w) >> 4 = w
	(void)T;
	// (the texel's four indices and four contributions: one 32-bit read each)
	const uint32_t idx = reinterpret_cast<const uint32_t*>(tw)[t], contrib = reinterpret_cast<const uint32_t*>(tci)[t];
	int sum = 8;
	sum += (int)uq[idx & 0xFFu] * (int)(contrib & 0xFFu) + (int)uq[(idx >> 8) & 0xFFu] * (int)((contrib >> 8) & 0xFFu);
	if (taps > 2) sum += (int)uq[(idx >> 16) & 0xFFu] * (int)((contrib >> 16) & 0xFFu) + (int)uq[idx >> 24] * (int)(contrib >> 24);
	return sum >> 4;
}

WV_FN int lerp_channel(bool u8, int c0, int c1, int w)
{
	int color = (c0 * (64 - w)) + (c1 * w) + 32;
	color = color >> 6;
	if (u8) color = (color >> 8) * 257;
	return color;
}

/* Squared error of wscb() against the block, with the reference's summation order for each of
 * its three variants (ref: :313, :407, :505).  Uniform return value. */
WV_FN float compute_symbolic_block_difference(const Ctx& c, const PartView& pv, const DecView& di)
{
	const Scb& scb = c.wscb();
	TrialInfo& tr = c.tr();
	const BlkInfo& blk = c.blk();
	const int T = c.T;
	if (wv_uniform((int)scb.block_type) == SYM_BTYPE_ERROR) return ERROR_CALC_DEFAULT;

	const uint8_t* tw = di.tw;
	const uint8_t* tci = di.tci;
	const bool dual = wv_uniform((int)scb.plane2_component) >= 0;      // (refine_pack: -1 for a one-plane block)
	const int pc = wv_uniform((int)scb.partition_count);
	const int profile = c.cfg->profile;
	const bool u8 = (c.cfg->flags & (1u << 1)) || profile == 0;
	const bool rgbm = (c.cfg->flags & (1u << 6)) != 0;
	const bool fast_1p = !dual && pc == 1 && !rgbm;
	const int p2c = wv_uniform((int)scb.plane2_component);
	const int taps = di.W == T ? 1 : di.max_texel_weight_count <= 2 ? 2 : 4;

	// (the decoded endpoints of every partition are in tr.ibox[p * 8 ..]: refine_pack() unpacks them once per packing)

	float* term = c.rsc(0);        // (the endpoint re-fit rows are free between re-fits)
	bool bad_here = false;         // RGBM: a texel of this lane decodes to M = 0
	WV_FOR_T(i, T)
	{
		// 1-plane multi-partition sums in partition order, the other two in texel order
		int t = (!dual && !fast_1p) ? pv.sorted[i] : i;
		int p = pv.of_texel[t];
		const int* e = &tr.ibox[p * 8];
		int w1 = unpack_texel_weight(scb.weights, tw, tci, T, t, taps);
		int w2 = dual ? unpack_texel_weight(scb.weights + PLANE2_OFFSET, tw, tci, T, t, taps) : w1;

		float col[4], old[4];
		for (int k = 0; k < 4; k++)
		{
			int w = (k == p2c) ? w2 : w1;
			col[k] = (float)lerp_channel(u8, e[k], e[4 + k], w);
			old[k] = c.data(k)[t];
		}

		if (rgbm)
		{
			if (col[3] == 0.0f) bad_here = true;
			float ms = c.cfg->rgbm_m_scale;
			for (int k = 0; k < 3; k++)
			{
				col[k] = col[k] * col[3] * ms;
				old[k] = old[k] * old[3] * ms;
			}
			col[3] = 1.0f; old[3] = 1.0f;
		}

		float err[4];
		for (int k = 0; k < 4; k++)
		{
			float e1 = f_abs(old[k] - col[k]);
			e1 = e1 < 1e15f ? e1 : 1e15f;
			err[k] = e1 * e1;
		}

		if (fast_1p)
		{
			term[i] = err[0] * cw_of(blk, 0) + err[1] * cw_of(blk, 1) + err[2] * cw_of(blk, 2) + err[3] * cw_of(blk, 3);
		}
		else
		{
			float d = hadd4(err[0] * cw_of(blk, 0), err[1] * cw_of(blk, 1), err[2] * cw_of(blk, 2), err[3] * cw_of(blk, 3));
			term[i] = d < ERROR_CALC_DEFAULT ? d : ERROR_CALC_DEFAULT;
		}
	}
	WV_SYNC();

	if (fast_1p)
	{
		return wv_sum4_texels(term, T);
	}
	// (the reference leaves its texel loop with the sentinel at the first texel whose M is zero, ref: :366-394, :470-480:
	//  whichever texel that is, the result is the same)
	if (rgbm && wv_any(bad_here)) return -ERROR_CALC_DEFAULT;

	// strictly sequential sum (additions only; four terms fetched per step)
	float summa = 0.0f;
	int i = 0;
	for (; i + 4 <= T; i += 4)
	{
		const float x0 = term[i], x1 = term[i + 1], x2 = term[i + 2], x3 = term[i + 3];
		summa += x0; summa += x1; summa += x2; summa += x3;
	}
	for (; i < T; i++) summa += term[i];
	return summa;
}

// ---------------------------------------------------------------------------------------------
// Weight realignment
// ---------------------------------------------------------------------------------------------

/* realign_weights() for a TWO-PLANE candidate on a grid that is walked speculatively (every 2D grid; two planes: at most
 * 32 weights each, one partition).  The reference does one plane after the other (ref: the `pl` loop of
 * realign_weights_decimated, astcenc_weight_align... astcenc_compress_symbolic.cpp:188-338); the planes share nothing --
 * a plane's endpoint step is zero on the other plane's channels, so its verdicts only read its own weights -- and so
 * both run side by side here: one first pass over the weights of both planes, then per round the first mover of EACH plane
 * moves and the later neighbours of both are looked at again on the sixteen quads of the wave.  Same verdicts, same moves,
 * half the passes.  State per plane pp: endpoints fbox[pp*8 ..], prev/next and float weights at [pp*32 + weight],
 * infilled weights wb[pp*Tp + texel], colour differences tdiff[(pp*Tp + texel)*4 ..], verdicts verdict[pp*32 + weight]. */
WV_FN bool realign_weights_2planes(const Ctx& c, const DecView& di, const QuantXfer& qat)
{
	WV_LANE_SCOPE;
	Scb& scb = c.wscb();
	TrialInfo& tr = c.tr();
	const BlkInfo& blk = c.blk();
	const int T = c.T, Tp = c.Tp;
	const int W = di.W;
	const int p2c = wv_uniform((int)scb.plane2_component);
	const uint8_t* wtc = di.wtc;
	const uint8_t* wt = di.wt;
	const float* tcw = di.tcw;
	const uint8_t* tw = di.tw;
	const float* tcf = di.tcf;
	const bool two_taps = di.max_texel_weight_count <= 2;
	const int last_row = di.rows - 1;

	uint32_t* pn = reinterpret_cast<uint32_t*>(c.rsc(0));   // [2][32]
	float* uqf = reinterpret_cast<float*>(pn + 64);         // [2][32]
	float* wb = uqf + 64;                                   // [2][Tp]
	float* tdiff = wb + 2 * Tp;                             // [2][Tp][4]
	uint8_t* verdict = reinterpret_cast<uint8_t*>(&tr.ibox[40]);    // [2][32] new quantized value, 255 = stays
	uint8_t* items = reinterpret_cast<uint8_t*>(&tr.ibox[32]);      // [32] weights to evaluate: plane * 32 + weight

	// (the decoded endpoints are in tr.ibox[0 .. 7]: refine_pack() unpacks them once per packing)
	WV_FOR64(k, 8)
	{
		const int pp = k >> 2, ch = k & 3;
		const int* e = &tr.ibox[0];
		const bool masked = (pp == 0) ? (ch == p2c) : (ch != p2c);
		const int epd = masked ? 0 : e[4 + ch] - e[ch];
		tr.fbox[pp * 8 + ch] = (float)e[ch];
		tr.fbox[pp * 8 + 4 + ch] = (float)epd * (1.0f / 64.0f);
	}
	WV_FOR64(k, 64)
	{
		const int we = k & 31;
		if (we < W)
		{
			const int u = scb.weights[k];                        // (PLANE2_OFFSET == 32: the index is plane * 32 + weight)
			uqf[k] = (float)u;
			pn[k] = qat.prev_next_values[u];
		}
	}
	WV_SYNC();
	auto refresh_texel = [&](int pp, int t)
	{
		const float* wts = uqf + pp * 32;
		const float w = two_taps ? infill2(wts, tw, tcf, T, t) : infill4(wts, tw, tcf, T, t);
		wb[pp * Tp + t] = w;
		const f4 color_offset = load4(&tr.fbox[pp * 8 + 4]);
		const f4 color_base = load4(&tr.fbox[pp * 8]);
		const f4 color = color_base + color_offset * w;
		const f4 orig_color = mk4(c.data(0)[t], c.data(1)[t], c.data(2)[t], c.data(3)[t]);
		store4_aligned(&tdiff[(pp * Tp + t) * 4], color - orig_color);
	};
	WV_FOR(k, 2 * T)
	{
		const int pp = k >= T ? 1 : 0;
		refresh_texel(pp, k - pp * T);
	}
	WV_SYNC();

	// ---- first pass: every weight of both planes against the current state ----
	int count = 2 * W;
#if defined(ASTC_DUPSTAGE)
	bool dup_done = false;
	for (int rep = 0; rep < (DUP_STAGE_ID(c) == (uint32_t)DUP_REALIGN_FIRST_PASS ? 2 : 1); rep++)
#endif
	if (count > 16)
	{
		// one lane per weight (see realign_weights: the twelve running sums in registers, the table reads one row ahead)
		const f4 error_weight = cw4_of(blk);
		WV_FOR64(k, 64)
		{
			const int pp = k >> 5, we = k & 31;
			if (we >= W) continue;
			const int uqw = scb.weights[k];
			const uint32_t prev_and_next = pn[k];
			const float* wbp = wb + pp * Tp;
			const float* tdp = tdiff + pp * Tp * 4;
			const f4 color_offset = load4(&tr.fbox[pp * 8 + 4]);
			const int n = wtc[we];
			int texel = wt[we];
			float tw_base = tcw[we];
			const float uqw_base = (float)uqw;
			const float uqw_diff_down = (float)(prev_and_next & 0xFF) - uqw_base;
			const float uqw_diff_up = (float)((prev_and_next >> 8) & 0xFF) - uqw_base;
			f4 sb = splat4(0.0f), sd = splat4(0.0f), su = splat4(0.0f);
			for (int te = 0; te < n; te++)
			{
				const int row1 = i_min(te + 1, last_row);
				const int texel_next = wt[row1 * W + we];
				const float tw_next = tcw[row1 * W + we];
				const float weight_base = wbp[texel];
				const float weight_down = weight_base + uqw_diff_down * tw_base - weight_base;
				const float weight_up = weight_base + uqw_diff_up * tw_base - weight_base;
				const f4 color_diff = load4_aligned(&tdp[texel * 4]);        // (base + step * weight_base) - source colour
				const f4 color_down_diff = color_diff + color_offset * weight_down;
				const f4 color_up_diff = color_diff + color_offset * weight_up;
				sb = sb + color_diff * color_diff;
				sd = sd + color_down_diff * color_down_diff;
				su = su + color_up_diff * color_up_diff;
				texel = texel_next; tw_base = tw_next;
			}
			const float error_base = hadd_s(sb * error_weight);
			const float error_down = hadd_s(sd * error_weight);
			const float error_up = hadd_s(su * error_weight);
			int new_value = 255;
			if ((error_up < error_base) && (error_up < error_down) && (uqw < 64)) new_value = (int)((prev_and_next >> 8) & 0xFF);
			else if ((error_down < error_base) && (uqw > 0)) new_value = (int)(prev_and_next & 0xFF);
			verdict[k] = (uint8_t)new_value;
		}
		count = 0;
	}
	else
	{
		WV_FOR64(k, count) { items[k] = (uint8_t)(k >= W ? 32 + (k - W) : k); }
	}
	WV_SYNC();

	bool adjustments = false;
	int start0 = 0, start1 = 0;                     // verdicts below these are final
	for (;;)
	{
		// ---- the listed weights, one QUAD per weight, lane = colour channel (see realign_weights) ----
		if (count != 0)
		{
			const qf error_weight_q = q_cw_of(blk);
			WV_QUADS(k, count)
			{
				const int item = items[k];
				const int pp = item >> 5, we = item & 31;
				const int uqw = scb.weights[item];
				const uint32_t prev_and_next = pn[item];
				const float* wbp = wb + pp * Tp;
				const float* tdp = tdiff + pp * Tp * 4;
				const qf color_offset_q = q_load(&tr.fbox[pp * 8 + 4]);
				const int n = wtc[we];
				int texel_next = wt[we];
				float tw_cur = tcw[we];
				const int row1 = i_min(1, last_row);
				int texel_ahead = wt[row1 * W + we];
				float tw_next = tcw[row1 * W + we];
				const float uqw_base = (float)uqw;
				const float uqw_diff_down = (float)(prev_and_next & 0xFF) - uqw_base;
				const float uqw_diff_up = (float)((prev_and_next >> 8) & 0xFF) - uqw_base;
				qf sb = q_splat(0.0f), sd = q_splat(0.0f), su = q_splat(0.0f);
				float weight_cur = wbp[texel_next];
				qf diff_cur = q_load(&tdp[texel_next * 4]);
				texel_next = texel_ahead;
				for (int te = 0; te < n; te++)
				{
					const int row2 = i_min(te + 2, last_row);
					texel_ahead = wt[row2 * W + we];
					const float tw_ahead = tcw[row2 * W + we];
					const float weight_next = wbp[texel_next];
					const qf diff_next = q_load(&tdp[texel_next * 4]);

					const float weight_base = weight_cur;
					const float weight_down = weight_base + uqw_diff_down * tw_cur - weight_base;
					const float weight_up = weight_base + uqw_diff_up * tw_cur - weight_base;
					const qf color_diff = diff_cur;
					const qf color_down_diff = color_diff + color_offset_q * weight_down;
					const qf color_up_diff = color_diff + color_offset_q * weight_up;
					sb = sb + color_diff * color_diff;
					sd = sd + color_down_diff * color_down_diff;
					su = su + color_up_diff * color_up_diff;

					texel_next = texel_ahead; tw_cur = tw_next; tw_next = tw_ahead;
					weight_cur = weight_next; diff_cur = diff_next;
				}
				const float error_base = q_hadd(sb * error_weight_q);
				const float error_down = q_hadd(sd * error_weight_q);
				const float error_up = q_hadd(su * error_weight_q);
				int new_value = 255;
				if ((error_up < error_base) && (error_up < error_down) && (uqw < 64)) new_value = (int)((prev_and_next >> 8) & 0xFF);
				else if ((error_down < error_base) && (uqw > 0)) new_value = (int)(prev_and_next & 0xFF);
				Q_ONCE { verdict[item] = (uint8_t)new_value; }
			}
			WV_SYNC();
#if defined(ASTC_DUPSTAGE)
			if (!dup_done && !adjustments && DUP_STAGE_ID(c) == (uint32_t)DUP_REALIGN_FIRST_PASS) { dup_done = true; continue; }
#endif
		}

		// ---- per plane: the first weight (in index order) whose verdict is "move" moves; what that invalidates -- the
		//      later weights sharing a texel with it -- is listed for the next round ----
		count = 0;
		for (;;)
		{
			int m0, m1, nv0 = 0, nv1 = 0;
#if WV_DEVICE
			int entry = 255;
			{
				const int we = WV_LANE & 31;
				const int mine = we < W ? (int)verdict[WV_LANE] : 255;
				const unsigned long long movers = __ballot(mine != 255 && we >= (WV_LANE < 32 ? start0 : start1));
				const uint32_t lo = (uint32_t)movers, hi = (uint32_t)(movers >> 32);
				m0 = lo ? (int)__builtin_ctz(lo) : -1;
				m1 = hi ? (int)__builtin_ctz(hi) : -1;
				if (m0 < 0 && m1 < 0) break;
				if (m0 >= 0) nv0 = __builtin_amdgcn_readlane(mine, m0);
				if (m1 >= 0) nv1 = __builtin_amdgcn_readlane(mine, 32 + m1);
				// the movers' later neighbours (global memory): requested now, needed after the moves
				const int m = WV_LANE < 16 ? m0 : m1;
				if (WV_LANE < 32 && m >= 0) entry = di.later[m * REALIGN_LATER_MAX + (WV_LANE & 15)];
			}
#else
			m0 = wv_find_first(W, [&](int w) { return w >= start0 && verdict[w] != 255; });
			m1 = wv_find_first(W, [&](int w) { return w >= start1 && verdict[32 + w] != 255; });
			if (m0 < 0 && m1 < 0) break;
			if (m0 >= 0) nv0 = verdict[m0];
			if (m1 >= 0) nv1 = verdict[32 + m1];
#endif
			adjustments = true;
			WV_FOR64(k, 2)
			{
				const int m = k ? m1 : m0, nv = k ? nv1 : nv0;
				if (m >= 0)
				{
					scb.weights[k * 32 + m] = (uint8_t)nv;
					uqf[k * 32 + m] = (float)nv;
				}
			}
			WV_SYNC();
			WV_FOR(k, 2 * di.rows)
			{
				const int pp = k >= di.rows ? 1 : 0;
				const int te = k - pp * di.rows;
				const int m = pp ? m1 : m0;
				if (m >= 0 && te < (int)wtc[m]) refresh_texel(pp, (int)wt[te * W + m]);
			}
			if (m0 >= 0) start0 = m0 + 1;
			if (m1 >= 0) start1 = m1 + 1;
			int count0 = 0, count1 = 0;
#if WV_DEVICE
			{
				const unsigned long long listed = __ballot(entry != 255);
				count0 = (int)__builtin_popcount((uint32_t)listed & 0xFFFFu);
				count1 = (int)__builtin_popcount((uint32_t)listed >> 16);
				if (entry != 255) items[(WV_LANE < 16 ? 0 : count0) + (WV_LANE & 15)] = (uint8_t)((WV_LANE < 16 ? 0 : 32) + entry);
			}
#else
			if (m0 >= 0) { const uint8_t* row = di.later + m0 * REALIGN_LATER_MAX; while (count0 < REALIGN_LATER_MAX && row[count0] != 255) { items[count0] = row[count0]; count0++; } }
			if (m1 >= 0) { const uint8_t* row = di.later + m1 * REALIGN_LATER_MAX; while (count1 < REALIGN_LATER_MAX && row[count1] != 255) { items[count0 + count1] = (uint8_t)(32 + row[count1]); count1++; } }
#endif
			WV_SYNC();
			count = count0 + count1;
			if (count != 0) break;
		}
		if (count == 0) break;
	}
	return adjustments;
}

/* (ref: realign_weights_undecimated :69, realign_weights_decimated :188).  Operates on wscb();
 * uniform return: true if any weight moved. */
WV_FN bool realign_weights(const Ctx& c, const PartView& pv, const DecView& di, const QuantXfer& qat)
{
	WV_LANE_SCOPE;
	Scb& scb = c.wscb();
	TrialInfo& tr = c.tr();
	const BlkInfo& blk = c.blk();
	const int T = c.T;
	const int W = di.W;
	const int pc = wv_uniform((int)scb.partition_count);
	const int p2c = wv_uniform((int)scb.plane2_component);
	const int max_plane = p2c >= 0 ? 1 : 0;                             // (refine_pack: -1 for a one-plane block)
	const bool decimated = W != T;

	// (the decoded endpoints of every partition are in tr.ibox[p * 8 ..]: refine_pack() unpacks them once per packing)
	// (not in the device build for footprints of more than 64 texels: inlined next to that build's general texel loops the
	//  kernel body would need a scratch frame; the sequential build runs it for every footprint)
#if !WV_DEVICE || defined(ASTC_TEXELS_LE_64)
	if (max_plane == 1 && decimated && pc == 1 && di.later != nullptr && W <= 32) return realign_weights_2planes(c, di, qat);
#endif

	bool adjustments = false;
	// wave-uniform values read from LDS: keep them in scalar registers
	auto uniform4 = [](f4 v) {
#if WV_DEVICE
		return mk4(wv_uniform(v.x), wv_uniform(v.y), wv_uniform(v.z), wv_uniform(v.w));
#else
		return v;
#endif
	};
	const f4 error_weight = uniform4(cw4_of(blk));

	for (int pl = 0; pl <= max_plane; pl++)
	{
		uint8_t* uq = scb.weights + pl * PLANE2_OFFSET;

		// endpoint base and per-weight-step offset, with the other plane's channels frozen
		// -> fbox[p*8 + 0..3] = endpnt0f, fbox[p*8 + 4..7] = offset
		WV_FOR64(k, pc * 4)
		{
			int p = k >> 2, ch = k & 3;
			const int* e = &tr.ibox[p * 8];
			bool masked = (pl == 0) ? (ch == p2c) : (ch != p2c);
			int epd = masked ? 0 : e[4 + ch] - e[ch];
			tr.fbox[p * 8 + ch] = (float)e[ch];
			tr.fbox[p * 8 + 4 + ch] = (float)epd * (1.0f / 64.0f);
		}
		WV_SYNC();

		if (!decimated)
		{
			PROF_SCOPE(c, PS_Y5);
			bool moved_here = false;         // (per lane on the device; wv_any folds the lanes)
			WV_FOR_T(texel, T)
			{
				int uqw = uq[texel];
				uint32_t prev_and_next = qat.prev_next_values[uqw];
				int uqw_down = prev_and_next & 0xFF;
				int uqw_up = (prev_and_next >> 8) & 0xFF;

				float weight_base = (float)uqw;
				float weight_down = (float)(uqw_down - uqw);
				float weight_up = (float)(uqw_up - uqw);

				int p = pv.of_texel[texel];
				f4 color_offset = load4(&tr.fbox[p * 8 + 4]);
				f4 color_base = load4(&tr.fbox[p * 8]);

				f4 color = color_base + color_offset * weight_base;
				f4 orig_color = mk4(c.data(0)[texel], c.data(1)[texel], c.data(2)[texel], c.data(3)[texel]);

				f4 color_diff = color - orig_color;
				f4 color_diff_down = color_diff + color_offset * weight_down;
				f4 color_diff_up = color_diff + color_offset * weight_up;

				float error_base = dot_s(color_diff * color_diff, error_weight);
				float error_down = dot_s(color_diff_down * color_diff_down, error_weight);
				float error_up = dot_s(color_diff_up * color_diff_up, error_weight);

				float mv = 0.0f;
				if ((error_up < error_base) && (error_up < error_down) && (uqw < 64))
				{
					uq[texel] = (uint8_t)uqw_up;
					mv = 1.0f;
				}
				else if ((error_down < error_base) && (uqw > 0))
				{
					uq[texel] = (uint8_t)uqw_down;
					mv = 1.0f;
				}
				moved_here = moved_here || mv != 0.0f;
			}
			WV_SYNC();
			adjustments = adjustments || wv_any(moved_here);
		}
		else
		{
			const uint8_t* wtc = di.wtc;
			const uint8_t* wt = di.wt;
			const float* tcw = di.tcw;
			const uint8_t* tw = di.tw;
			const float* tcf = di.tcf;
			const int rs = (di.rows + 3) & ~3;                       // row stride of the term rows for this grid
			const uint32_t rs_inv = (65536u + (uint32_t)rs - 1u) / (uint32_t)rs;   // k / rs == (k * rs_inv) >> 16 for k < 1024 (one divide per call)
			uint32_t* pn = reinterpret_cast<uint32_t*>(c.rsc(0));   // prev/next quant values of each weight's current value
			float* uqf = c.rsc(1);
			float* rt = c.rsc(2);                                    // [slots of this grid][12] rows of `rs` floats
			float* wb = rt + (int)c.root->realign_rt_floats;         // [T] current weights infilled to texel resolution

			WV_FOR64(i, W)
			{
				int u = uq[i];
				uqf[i] = (float)u;
				pn[i] = qat.prev_next_values[u];
			}
			WV_SYNC();
			// The reference re-infills a texel's weight every time it looks at it; the value only changes
			// when one of its grid weights moves, so it is kept here and refreshed on moves.
			// (grids decimated in one dimension only: two taps; the other two would add 0.0 * weight, i.e. nothing)
			const bool two_taps = di.max_texel_weight_count <= 2;
			// with one partition the endpoint base / step are the same for every texel
			const bool one_partition = pc == 1;
			const f4 color_offset_1 = uniform4(load4(&tr.fbox[4]));
			const f4 color_base_1 = uniform4(load4(&tr.fbox[0]));
			const bool speculative = di.later != nullptr;
			// The one-lane-per-weight evaluator (below) keeps, per texel, what every weight reaching the texel would compute
			// again: the decoded colour at the current weights minus the source colour, and the texel's endpoint step.  The
			// term rows of the group evaluator are not used on that path; the three arrays take their place.
			float* tdiff = rt + c.Tp;                          // [T][4]
			float* toff = tdiff + 4 * c.Tp;                    // [T][4], more than one partition only
			if (speculative) wb = rt;
			auto refresh_texel = [&](int t)
			{
				const float w = two_taps ? infill2(uqf, tw, tcf, T, t) : infill4(uqf, tw, tcf, T, t);
				wb[t] = w;
				if (speculative)
				{
					f4 color_offset = color_offset_1, color_base = color_base_1;
					if (!one_partition)
					{
						const int p = pv.of_texel[t];
						color_offset = load4(&tr.fbox[p * 8 + 4]);
						color_base = load4(&tr.fbox[p * 8]);
						store4_aligned(&toff[t * 4], color_offset);
					}
					const f4 color = color_base + color_offset * w;
					const f4 orig_color = mk4(c.data(0)[t], c.data(1)[t], c.data(2)[t], c.data(3)[t]);
					store4_aligned(&tdiff[t * 4], color - orig_color);
				}
			};
			{ PROF_SCOPE(c, PS_Y4);
			WV_FOR_T(t, T) { refresh_texel(t); }
			WV_SYNC(); }

			// Two ways through a grid's weights (ref: the one-by-one sweep of realign_weights_decimated :188-338):
			//
			// * speculation (every grid whose later-neighbour lists fit, i.e. every 2D grid): all weights are evaluated
			//   against the current state, one lane or one quad each; then, in index order, the first weight whose verdict is
			//   "move" is moved for real and only the later weights that share a texel with it
			//   (DecimationInfo::off_realign_later) are evaluated again.  A verdict only depends on the weight itself and on
			//   the weights it shares texels with, moves are applied in index order, and every verdict a move could
			//   invalidate is recomputed before it is looked at: the outcome is the reference's sweep.  (On the bench content
			//   1.5 to 3.5 weights of ~26 move per call.)
			//
			// * the level schedule (DecimationInfo::off_realign_order; some 3D grids): a weight's verdict only depends on
			//   earlier weights that share a texel with it, so the host-built schedule groups weights that touch disjoint
			//   texels and a whole group is evaluated -- one lane per (weight, texel) -- and moved at once.
			uint8_t* verdict = reinterpret_cast<uint8_t*>(&tr.ibox[40]);    // [W <= 64] new quantized value, 255 = stays
			const uint8_t* order = di.ro;
			const uint8_t* group_count = di.rc;
			int lv = 0, pos = 0;                          // level schedule: next group
			int start = 0;                                // speculation: verdicts below `start` are final
			// Speculative grids (decimated in two dimensions: every weight reaches a dozen texels or more, and no two
			// neighbours can be decided together): ONE LANE PER WEIGHT walks the weight's texels, keeps the twelve running
			// sums in registers in the reference's order and decides on the spot -- all weights in a single pass, no term
			// rows in LDS, no second and third phase per handful of weights.  (The (weight, texel)-lane evaluator below
			// stays for the level schedule, whose groups are many weights with few texels each.)
			if (speculative)
			{
				int items = W;                              // pass 1: every weight; then: the later neighbours of each mover
				bool all = true;
#if defined(ASTC_DUPSTAGE)
				bool dup_done = false;
#endif
				const int last_row = di.rows - 1;
				// the mover's later neighbours: on the device entry k sits in a register of the lanes of quad k
				const uint8_t* later_row = di.later;
#if WV_DEVICE
				int later_mine = 255;
#define REALIGN_LATER_ENTRY(k) later_mine
#else
#define REALIGN_LATER_ENTRY(k) ((int)later_row[k])
#endif
				for (;;)
				{
					// Both evaluators walk a weight's texel list with the table reads two rows ahead and the per-texel reads one
					// row ahead of the arithmetic (rows past the list are clamped to the grid's last row and never used): the
					// loop is a chain of dependent LDS reads otherwise, two round trips per texel.
					{ PROF_SCOPE(c, PS_Y6);
					if (items <= 16)
					{
						// Up to sixteen weights (every pass after a move, and the first pass of the small grids): one QUAD per
						// weight, lane = colour channel.  The three running sums are one register each, the channel arithmetic one
						// instruction instead of four, and the error is the quad's hadd in the reference's order.
						const qf error_weight_q = q_cw_of(blk);
						const qf color_offset_q = q_load(&tr.fbox[4]);
						WV_QUADS16(k, items)
						{
							const int we = all ? k : REALIGN_LATER_ENTRY(k);
							const int uqw = uq[we];
							const uint32_t prev_and_next = pn[we];
							const int n = wtc[we];
							int texel_next = wt[we];
							float tw_cur = tcw[we];
							const int row1 = i_min(1, last_row);
							int texel_ahead = wt[row1 * W + we];
							float tw_next = tcw[row1 * W + we];
							const float uqw_base = (float)uqw;
							const float uqw_diff_down = (float)(prev_and_next & 0xFF) - uqw_base;
							const float uqw_diff_up = (float)((prev_and_next >> 8) & 0xFF) - uqw_base;
							qf sb = q_splat(0.0f), sd = q_splat(0.0f), su = q_splat(0.0f);
							float weight_cur = wb[texel_next];
							qf diff_cur = q_load(&tdiff[texel_next * 4]);
							qf offset_cur = one_partition ? color_offset_q : q_load(&toff[texel_next * 4]);
							texel_next = texel_ahead;
							for (int te = 0; te < n; te++)
							{
								const int row2 = i_min(te + 2, last_row);
								texel_ahead = wt[row2 * W + we];
								const float tw_ahead = tcw[row2 * W + we];
								const float weight_next = wb[texel_next];
								const qf diff_next = q_load(&tdiff[texel_next * 4]);
								const qf offset_next = one_partition ? color_offset_q : q_load(&toff[texel_next * 4]);

								const float weight_base = weight_cur;
								const float weight_down = weight_base + uqw_diff_down * tw_cur - weight_base;
								const float weight_up = weight_base + uqw_diff_up * tw_cur - weight_base;
								const qf color_diff = diff_cur;                         // (base + step * weight_base) - source colour
								const qf color_down_diff = color_diff + offset_cur * weight_down;
								const qf color_up_diff = color_diff + offset_cur * weight_up;
								sb = sb + color_diff * color_diff;
								sd = sd + color_down_diff * color_down_diff;
								su = su + color_up_diff * color_up_diff;

								texel_next = texel_ahead; tw_cur = tw_next; tw_next = tw_ahead;
								weight_cur = weight_next; diff_cur = diff_next; offset_cur = offset_next;
							}
							const float error_base = q_hadd(sb * error_weight_q);
							const float error_down = q_hadd(sd * error_weight_q);
							const float error_up = q_hadd(su * error_weight_q);
							int new_value = 255;
							if ((error_up < error_base) && (error_up < error_down) && (uqw < 64)) new_value = (int)((prev_and_next >> 8) & 0xFF);
							else if ((error_down < error_base) && (uqw > 0)) new_value = (int)(prev_and_next & 0xFF);
							Q_ONCE { verdict[we] = (uint8_t)new_value; }
						}
					}
					else
					{
						// (more than sixteen weights: only ever the first pass, one lane per weight; the table reads run one row ahead)
						WV_FOR(we, items)
						{
							const int uqw = uq[we];
							const uint32_t prev_and_next = pn[we];
							const int n = wtc[we];
							int texel = wt[we];
							float tw_base = tcw[we];
							const float uqw_base = (float)uqw;
							const float uqw_diff_down = (float)(prev_and_next & 0xFF) - uqw_base;
							const float uqw_diff_up = (float)((prev_and_next >> 8) & 0xFF) - uqw_base;
							f4 sb = splat4(0.0f), sd = splat4(0.0f), su = splat4(0.0f);
							for (int te = 0; te < n; te++)
							{
								const int row1 = i_min(te + 1, last_row);
								const int texel_next = wt[row1 * W + we];
								const float tw_next = tcw[row1 * W + we];
								const float weight_base = wb[texel];
								const float weight_down = weight_base + uqw_diff_down * tw_base - weight_base;
								const float weight_up = weight_base + uqw_diff_up * tw_base - weight_base;
								const f4 color_offset = one_partition ? color_offset_1 : load4_aligned(&toff[texel * 4]);
								const f4 color_diff = load4_aligned(&tdiff[texel * 4]);        // (base + step * weight_base) - source colour
								const f4 color_down_diff = color_diff + color_offset * weight_down;
								const f4 color_up_diff = color_diff + color_offset * weight_up;
								sb = sb + color_diff * color_diff;
								sd = sd + color_down_diff * color_down_diff;
								su = su + color_up_diff * color_up_diff;
								texel = texel_next; tw_base = tw_next;
							}
							const float error_base = hadd_s(sb * error_weight);
							const float error_down = hadd_s(sd * error_weight);
							const float error_up = hadd_s(su * error_weight);
							int new_value = 255;
							if ((error_up < error_base) && (error_up < error_down) && (uqw < 64)) new_value = (int)((prev_and_next >> 8) & 0xFF);
							else if ((error_down < error_base) && (uqw > 0)) new_value = (int)(prev_and_next & 0xFF);
							verdict[we] = (uint8_t)new_value;
						}
					}
					WV_SYNC(); }
#if defined(ASTC_DUPSTAGE)
					if (all && !dup_done && DUP_STAGE_ID(c) == (uint32_t)DUP_REALIGN_FIRST_PASS) { dup_done = true; continue; }
#endif
					PROF_SCOPE(c, PS_Y7);
					// the first weight (in index order) whose verdict is "move" moves; what it invalidates is evaluated again
					int mover, later_count = 0;
					for (;;)
					{
						int new_value;
#if WV_DEVICE
						{
							// one LDS read per lane answers both "who moves" and "to which value"
							const int mine = WV_LANE < W ? (int)verdict[WV_LANE] : 255;
							const unsigned long long movers = __ballot(WV_LANE >= start && mine != 255);
							mover = movers ? (int)__builtin_ctzll(movers) : -1;
							if (mover < 0) break;
							new_value = __builtin_amdgcn_readlane(mine, mover);
							// the list of the weights to look at again (global memory): requested now, needed after the move
							later_row = di.later + mover * REALIGN_LATER_MAX;
							later_mine = later_row[WV_LANE >> 2];
						}
#else
						mover = wv_find_first(W, [&](int w) { return w >= start && verdict[w] != 255; });
						if (mover < 0) break;
						new_value = verdict[mover];
						later_row = di.later + mover * REALIGN_LATER_MAX;
#endif
						adjustments = true;
						WV_ONE
						{
							uq[mover] = (uint8_t)new_value;
							uqf[mover] = (float)new_value;
						}
						WV_SYNC();
						// (the list length and the list entries are read side by side, not one after the other)
						WV_FOR_T(te, di.rows) { if (te < (int)wtc[mover]) refresh_texel((int)wt[te * W + mover]); }
						WV_SYNC();
#if WV_DEVICE
						later_count = popcount64(__ballot((WV_LANE & 3) == 0 && later_mine != 255));
#else
						later_count = 0;
						while (later_count < REALIGN_LATER_MAX && later_row[later_count] != 255) later_count++;
#endif
						start = mover + 1;
						if (later_count != 0) break;
					}
					if (mover < 0) break;
					items = later_count;
					all = false;
				}
#undef REALIGN_LATER_ENTRY
				continue;                                   // (next plane)
			}
			for (;;)
			{
				// ---- the next group of the schedule: `gn` weights, weight of slot s = src[s] ----
				if (lv >= di.levels) break;
				const int gn = group_count[lv];
				const uint8_t* src = order + pos;

				// ---- evaluate the group: one lane per (weight of the group, texel row of that weight) writes the squared
				//      differences for the current / next lower / next higher quantized value, 4 channels each ----
				WV_FOR_T(k, gn * rs)
				{
					const int slot = (int)(((uint32_t)k * rs_inv) >> 16), te = k - slot * rs;
					const int we = (int)src[slot];
					if (te >= (int)wtc[we]) continue;
					const int uqw = uq[we];
					const uint32_t prev_and_next = pn[we];
					const float uqw_base = (float)uqw;
					const float uqw_diff_down = (float)(prev_and_next & 0xFF) - uqw_base;
					const float uqw_diff_up = (float)((prev_and_next >> 8) & 0xFF) - uqw_base;

					int texel = wt[te * W + we];
					float tw_base = tcw[te * W + we];
					float weight_base = wb[texel];
					float weight_down = weight_base + uqw_diff_down * tw_base - weight_base;
					float weight_up = weight_base + uqw_diff_up * tw_base - weight_base;

					f4 color_offset = color_offset_1, color_base = color_base_1;
					if (!one_partition)
					{
						int p = pv.of_texel[texel];
						color_offset = load4(&tr.fbox[p * 8 + 4]);
						color_base = load4(&tr.fbox[p * 8]);
					}
					f4 color = color_base + color_offset * weight_base;
					f4 orig_color = mk4(c.data(0)[texel], c.data(1)[texel], c.data(2)[texel], c.data(3)[texel]);
					f4 color_diff = color - orig_color;
					f4 color_down_diff = color_diff + color_offset * weight_down;
					f4 color_up_diff = color_diff + color_offset * weight_up;
					f4 b = color_diff * color_diff, d = color_down_diff * color_down_diff, u = color_up_diff * color_up_diff;
					float* o = rt + slot * 12 * rs + te;
					o[0] = b.x; o[rs] = b.y; o[2 * rs] = b.z; o[3 * rs] = b.w;
					o[4 * rs] = d.x; o[5 * rs] = d.y; o[6 * rs] = d.z; o[7 * rs] = d.w;
					o[8 * rs] = u.x; o[9 * rs] = u.y; o[10 * rs] = u.z; o[11 * rs] = u.w;
				}
				WV_SYNC();
				// the 12 sums of each weight, in place: lane (slot, j) adds up its own row and leaves the total in the
				// row's first element
				WV_FOR(k, gn * 12)
				{
					const int slot = k / 12;
					const int n = wtc[(int)src[slot]];
					float* v = rt + k * rs;                   // == rt + slot * 12 * rs + (k % 12) * rs
					float acc = 0.0f;
					for (int te = 0; te < n; te++) acc += v[te];
					v[0] = acc;
				}
				WV_SYNC();
				// one lane per weight decides (ref: :250-316) and moves right away
				bool moved_here = false;                  // per lane on the device; wv_any() folds the lanes
				WV_FOR64(slot, gn)
				{
					const int we = (int)src[slot];
					const int uqw = uq[we];
					const uint32_t prev_and_next = pn[we];
					const float* sm = rt + wv_opaque(slot * 12) * rs;
					float error_base = hadd_s(mk4(sm[0], sm[rs], sm[2 * rs], sm[3 * rs]) * error_weight);
					float error_down = hadd_s(mk4(sm[4 * rs], sm[5 * rs], sm[6 * rs], sm[7 * rs]) * error_weight);
					float error_up = hadd_s(mk4(sm[8 * rs], sm[9 * rs], sm[10 * rs], sm[11 * rs]) * error_weight);
					int new_value = -1;
					if ((error_up < error_base) && (error_up < error_down) && (uqw < 64)) new_value = (int)((prev_and_next >> 8) & 0xFF);
					else if ((error_down < error_base) && (uqw > 0)) new_value = (int)(prev_and_next & 0xFF);
					verdict[we] = (uint8_t)(new_value < 0 ? 255 : new_value);
					if (new_value >= 0)
					{
						uqf[we] = (float)new_value;
						uq[we] = (uint8_t)new_value;
						moved_here = true;
					}
				}
				WV_SYNC();

				// ---- the texels of the weights that moved see different infilled weights now ----
				if (wv_any(moved_here))
				{
					adjustments = true;
					WV_FOR_T(k, gn * rs)
					{
						const int slot = (int)(((uint32_t)k * rs_inv) >> 16), te = k - slot * rs;
						const int we = order[pos + slot];
						if (verdict[we] == 255 || te >= (int)wtc[we]) continue;
						int texel = wt[te * W + we];
						wb[texel] = two_taps ? infill2(uqf, tw, tcf, T, texel) : infill4(uqf, tw, tcf, T, texel);
					}
					WV_SYNC();
				}
				pos += gn;
				lv++;
			}
		}
	}
	return adjustments;
}

} } // namespace astcd::ASTC_VARIANT
