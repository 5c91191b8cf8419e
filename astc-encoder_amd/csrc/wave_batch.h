// SPDX-License-Identifier: Apache-2.0
// The first refinement step of a trial's candidates, for several candidates at once.
//
// The reference refines the candidate block modes of a trial one after the other
// (compress_symbolic_block_for_partition_1plane, Source/astcenc_compress_symbolic.cpp:497-700; _2planes :880-1040).
// Every candidate starts the same way: re-fit the endpoints to its quantized weights (recompute_ideal_colors_*), pack
// them (pack_color_endpoints), decode and score (compute_symbolic_block_difference_*).  Nothing in that first step
// depends on what an earlier candidate did -- only the DECISIONS taken on its error do (the early-outs against the best
// error so far, :621-697).  So the step runs here for a whole batch of candidates side by side, and the decisions are
// replayed afterwards in candidate order on the errors it leaves behind; a candidate that survives its first test is
// then continued by the one-candidate code of wave_refine.h exactly where the reference would be.
//
// Why: on one candidate the step is a chain of a dozen LDS hand-offs on 4 to 36 lanes (14 running sums, partitions x 4
// solve lanes, one quad per partition in the endpoint coders).  The candidate axis multiplies the lanes of every one of
// those phases and leaves their number unchanged.
//
// How the re-fit fits three or four candidates into the LDS of one: the one-candidate code writes fifteen to nineteen
// term rows per candidate (om*om, om*w, w*w, c - c*w, c*w, ...) for its summing lanes; here a summing lane forms its term
// on the fly from the candidate's weight row and a texel row that all candidates share (the four colour channels, the
// scale projection, a row of ones) as
//         term = (A + B * x) * (E * d + F * (d * x)),     x = the candidate's weight, d = the shared value,
// with the lane's constants A, B, E, F in {0, 1, -1}: every product with 0 or 1 and every sum with 0 is exact, so each
// term has the bits of the reference's expression (1 - x, x * x, d - d * x, (1 - x) * (scale * ls_weight) ...), and the
// sums run in the reference's order.
#pragma once
#include "wave_ctx.h"
#include "wave_quad.h"
#include "wave_weights.h"
#include "wave_color.h"
#include "wave_refine.h"

namespace astcd { inline namespace ASTC_VARIANT {

/* What the step leaves per candidate for the candidate's own turn (LdsLayout::cstate; the first candidate of a batch
 * is consumed before anything overwrites the scratch, its record lives there). */
struct CandState {
	float* wep0;        // [P][4]  the re-fitted endpoints: the next re-fit falls back on them (ref: :1290-1300)
	float* wep1;        // [P][4]
	uint8_t* colors;    // [P][8]  the packed endpoints
	uint8_t* formats;   // [4]
	uint8_t* meta;      // [4]: colour quant level of the packed values, formats matched, RGBM error flag
	float* errorval;
	// (the RGB + scale and RGB + offset vectors are not kept: every re-fit computes them anew before a packing reads them)
};
WV_FN CandState cand_state_at(uint8_t* p, int P)
{
	CandState s;
	float* f = reinterpret_cast<float*>(p);
	s.wep0 = f; s.wep1 = f + 4 * P;
	uint8_t* b = p + 32 * P;
	s.colors = b; s.formats = b + 8 * P; s.meta = b + 8 * P + 4;
	s.errorval = reinterpret_cast<float*>(b + 8 * P + 8);
	return s;
}

/* Small per-candidate record of a batch. */
struct BatchCand {
	uint32_t tw_off, tcf_off;     // the candidate grid's per-texel records in the blob (ModeStatic)
	int32_t taps;                 // 1, 2 or 4 grid weights per texel
	int32_t color_quant;          // colour quant level of the first packing
	int32_t color_quant_mod;      // ... of the matched-format retry
	int32_t retry;                // the retry is due
	int32_t pad[2];
};
static_assert(sizeof(BatchCand) == 32, "batch_scratch_bytes() counts 32 bytes per candidate");

/* The scratch of the step (LdsLayout::bat; sizes: batch_scratch_bytes in wave_ctx.h). */
struct BatchView {
	uint8_t* base;
	int nb, planes, pc, Tp, Ts, P;      // Ts: floats between rows (lds_row_stride: 8x8 rows would all start in one LDS bank)
	uint32_t o_x, o_iw, o_sum, o_dec, o_term, o_ctab, o_cand, o_vec, o_state0;
	// shared rows: 0-3 the colour channels, 4 scale * ls_weight, 5 scale, 6 ones -- in the order the sums visit the texels
	WV_FN float* shared(int row) const { return reinterpret_cast<float*>(base) + row * Ts; }
	WV_FN float* x(int ci, int plane) const { return reinterpret_cast<float*>(base + o_x) + (ci * planes + plane) * Ts; }
	WV_FN uint8_t* iw(int ci, int plane) const { return base + o_iw + (ci * planes + plane) * Tp; }
	WV_FN float* sums(int ci, int p) const { return reinterpret_cast<float*>(base + o_sum) + (ci * pc + p) * 28; }
	WV_FN int* dec(int ci, int p) const { return reinterpret_cast<int*>(base + o_dec) + (ci * pc + p) * 8; }
	WV_FN float* term(int ci) const { return reinterpret_cast<float*>(base + o_term) + ci * Ts; }
	WV_FN uint8_t* ctab(int slot) const { return base + o_ctab + slot * 512; }
	WV_FN BatchCand& cand(int ci) const { return reinterpret_cast<BatchCand*>(base + o_cand)[ci]; }
	// the re-fit's RGB + scale / RGB + offset vectors of candidate ci: [P][4] each
	WV_FN float* rgbs(int ci) const { return reinterpret_cast<float*>(base + o_vec) + ci * 8 * P; }
	WV_FN float* rgbo(int ci) const { return rgbs(ci) + 4 * P; }
};

WV_FN BatchView batch_view(const Ctx& c, bool dual, int pc)
{
	BatchView v;
	v.base = c.lds + c.L->bat;
	v.nb = (int)c.L->bat_max[dual ? 1 : 0];
	v.planes = dual ? 2 : 1;
	v.pc = pc;
	v.Tp = c.Tp;
	v.Ts = c.Ts;
	v.P = i_max(1, i_min(4, (int)c.cfg->tune_partition_count_limit));
	const BatchOffsets& o = c.L->batview[dual ? 1 : 0][pc - 1];      // (host-computed: batch_offsets, wave_ctx.h)
	v.o_x = o.o_x; v.o_iw = o.o_iw; v.o_sum = o.o_sum; v.o_dec = o.o_dec; v.o_term = o.o_term;
	v.o_ctab = o.o_ctab; v.o_cand = o.o_cand; v.o_vec = o.o_vec; v.o_state0 = o.o_state0;
	return v;
}

/* The record of candidate `ci` of the current batch. */
WV_FN CandState batch_state(const Ctx& c, const BatchView& bv, int ci)
{
	uint8_t* p = ci == 0 ? bv.base + bv.o_state0 : c.lds + c.L->cstate + (uint32_t)(ci - 1) * c.L->cstate_stride;
	return cand_state_at(p, bv.P);
}

// ---------------------------------------------------------------------------------------------
// Stage 1: shared texel rows, the candidates' expanded weights, their colour quantization rows
// ---------------------------------------------------------------------------------------------
WV_FN void batch_prepare_body(const Ctx& c, const BatchView& bv, const PartView& pv, bool dual, int partition_count, int first, int count)
{
	TrialInfo& tr = c.tr();
	const BlkInfo& blk = c.blk();
	const int T = c.T;
	const ModeStatic* mstat = reinterpret_cast<const ModeStatic*>(c.table(c.root->off_mode_static));

	WV_FOR64(ci, count)
	{
		const ModeStatic ms = table_at(mstat, (uint32_t)tr.cand_block_mode[first + ci]);
		BatchCand m;
		m.tw_off = ms.tw_off; m.tcf_off = ms.tcf_off; m.taps = ms.taps;
		m.color_quant = tr.cand_quant[first + ci];
		m.color_quant_mod = tr.cand_quant_mod[first + ci];
		m.retry = 0; m.pad[0] = 0; m.pad[1] = 0;
		bv.cand(ci) = m;
	}
	// the texel rows every candidate's sums read, in visiting order: partition by partition for a one-plane trial
	// (ref: recompute_ideal_colors_1plane :1241), texel order for a two-plane one (:1474)
	const float ls_weight = hadd_rgb_s(cw4_of(blk));
	WV_FOR_T(i, T)
	{
		const int t = dual ? i : (int)pv.sorted[i];
		const int p = dual ? 0 : (int)pv.of_texel[t];
		const f4 scale_dir = load4(tr.pm_dir[p]);
		const f4 rgba = mk4(c.data(0)[t], c.data(1)[t], c.data(2)[t], c.data(3)[t]);
		const float scale = dot3_s(scale_dir, rgba);
		bv.shared(0)[i] = rgba.x; bv.shared(1)[i] = rgba.y; bv.shared(2)[i] = rgba.z; bv.shared(3)[i] = rgba.w;
		bv.shared(4)[i] = scale * ls_weight;
		bv.shared(5)[i] = scale;
		bv.shared(6)[i] = 1.0f;
	}
	// the rows of every candidate's colour quant level (512 bytes each)
	for (int ci = 0; ci < count; ci++)
	{
		const int q = wv_uniform((int)tr.cand_quant[first + ci]);
		stage_quads_nosync(bv.ctab(ci), c.table(c.root->off_color_unquant_to_uquant) + (uint32_t)(q - QUANT_6) * 512u, 512);
	}
	WV_SYNC();

	// The weights of every (candidate, texel) at texel resolution: as floats for the re-fit (ref: the bilinear infill of
	// unquantized weights / 64, :1184-1196) and as integers for the scoring (ref: unpack_weights, decompress_symbolic.cpp:89).
	// The texel's taps come straight from the grid's records in the blob (one 32-bit + one 128-bit load): no candidate's
	// tables are staged for this step.  A float contribution is the integer one / 16, exactly.
	const uint32_t t_inv = c.L->t_inv24;
	WV_FOR(k, count * T)
	{
		const int ci = (int)(((uint32_t)k * t_inv) >> 24), i = k - ci * T;
		const int t = dual ? i : (int)pv.sorted[i];
		const BatchCand m = bv.cand(ci);
		const uint8_t* uq = c.candw(first + ci);
		for (int plane = 0; plane < bv.planes; plane++)
		{
			const uint8_t* u = uq + plane * PLANE2_OFFSET;
			// (every grid through the four-tap form: a tap a grid does not have has index 0 and factor 0 in its records, which
			//  leaves both values as the shorter forms give them -- x + (0 + 0) is x, (8 + 16 w) >> 4 is w -- and the lanes of
			//  a wave, which belong to two candidates with different grids, do not branch apart)
			const TexelTaps tp = texel_taps_at(c.tab, m.tw_off, m.tcf_off, (uint32_t)t);
			const int u0 = u[tp.idx & 0xFFu], u1 = u[(tp.idx >> 8) & 0xFFu], u2 = u[(tp.idx >> 16) & 0xFFu], u3 = u[tp.idx >> 24];
			const float g0 = (float)u0 * (1.0f / 64.0f), g1 = (float)u1 * (1.0f / 64.0f);
			const float g2 = (float)u2 * (1.0f / 64.0f), g3 = (float)u3 * (1.0f / 64.0f);
			const float xf = (g0 * tp.c0 + g1 * tp.c1) + (g2 * tp.c2 + g3 * tp.c3);
			const int sum = 8 + u0 * (int)(tp.c0 * 16.0f) + u1 * (int)(tp.c1 * 16.0f) + u2 * (int)(tp.c2 * 16.0f) + u3 * (int)(tp.c3 * 16.0f);
			const int wi = sum >> 4;
			bv.x(ci, plane)[i] = xf;
			bv.iw(ci, plane)[i] = (uint8_t)wi;
		}
	}
	WV_SYNC();
}

// ---------------------------------------------------------------------------------------------
// Stage 2: the running sums of the re-fit, one lane per (candidate, partition, sum)
// ---------------------------------------------------------------------------------------------

/* One summing lane's constants. */
struct SumLane {
	const float* xrow;      // the candidate's weights (or the scale row for the scale minimum / maximum)
	const float* drow;      // the shared texel row of the term
	float A, B, E, F;       // term = (A + B x) (E d + F (d x))
	float init;             // the sum starts here (1e-17 for the sums of weights)
	float mn0;              // the minimum starts here
	int out, out_mn;        // slots of the sum / of (minimum, maximum) in the partition's sums; -1: not kept
};

/* Sum r of a one-plane candidate (ref: recompute_ideal_colors_1plane :1241-1269).  Slots as refit_solve_1plane reads them. */
WV_FN SumLane sum_lane_1plane(const BatchView& bv, int ci, int r)
{
	SumLane s;
	s.xrow = r == 1 ? bv.shared(5) : bv.x(ci, 0);
	const int ch = r >= 9 ? r - 9 : r - 5;                         // colour channel of sums 5..12
	s.drow = r >= 13 ? bv.shared(4) : r >= 5 ? bv.shared(ch & 3) : bv.shared(6);
	// P = A + B x:   1 (0, 1: the row itself)   1 - x (2, 3, 13)   x (4, 14)   1 (5..12)
	const bool p_om = r == 2 || r == 3 || r == 13, p_x = r == 4 || r == 14;
	s.A = p_x ? 0.0f : 1.0f;
	s.B = p_om ? -1.0f : p_x ? 1.0f : 0.0f;
	// Q = E d + F (d x):   x (0, 1, 3, 4; d = 1)   1 - x (2; d = 1)   d - d x (5..8)   d x (9..12)   d (13, 14)
	const bool q_x = r == 0 || r == 1 || r == 3 || r == 4 || (r >= 9 && r <= 12), q_d = r >= 13;
	s.E = q_x ? 0.0f : 1.0f;
	s.F = q_x ? 1.0f : q_d ? 0.0f : -1.0f;
	s.init = r == 0 ? 1e-17f : 0.0f;
	s.mn0 = r == 0 ? 1.0f : 1e10f;
	s.out = r == 0 ? 7 : r == 1 ? -1 : r <= 4 ? 4 + (r - 2) : r <= 12 ? 8 + (r - 5) : 16 + (r - 13);
	s.out_mn = r == 0 ? 0 : r == 1 ? 2 : -1;
	return s;
}

/* Sum r of a two-plane candidate (ref: recompute_ideal_colors_2planes :1474-1512): 0 / 1 the weights of plane 1 / 2,
 * 2 the scale projection (minimum and maximum only), 3-5 / 6-8 left, middle, right of plane 1 / 2, 9-12 / 13-16 the colour
 * sums (a channel follows plane 2 if it is the separated component), 17-18 the scale sums.  Slots as
 * refit_solve_2planes reads them, plus 26 / 27 for the two weight sums. */
WV_FN SumLane sum_lane_2planes(const BatchView& bv, int ci, int r, int plane2_component)
{
	SumLane s;
	const int ch = r >= 13 ? r - 13 : r - 9;
	const bool second = r == 1 || (r >= 6 && r <= 8) || (r >= 9 && r <= 16 && ch == plane2_component);
	s.xrow = r == 2 ? bv.shared(5) : bv.x(ci, second ? 1 : 0);
	s.drow = r >= 17 ? bv.shared(4) : r >= 9 ? bv.shared(ch & 3) : bv.shared(6);
	const bool p_om = r == 3 || r == 4 || r == 6 || r == 7 || r == 17, p_x = r == 5 || r == 8 || r == 18;
	s.A = p_x ? 0.0f : 1.0f;
	s.B = p_om ? -1.0f : p_x ? 1.0f : 0.0f;
	const bool q_x = r <= 2 || r == 4 || r == 5 || r == 7 || r == 8 || (r >= 13 && r <= 16), q_d = r >= 17;
	s.E = q_x ? 0.0f : 1.0f;
	s.F = q_x ? 1.0f : q_d ? 0.0f : -1.0f;
	s.init = r <= 1 ? 1e-17f : 0.0f;
	s.mn0 = r <= 1 ? 1.0f : 1e10f;
	s.out = r == 0 ? 26 : r == 1 ? 27 : r == 2 ? -1 : r <= 8 ? 6 + (r - 3) : r <= 16 ? 12 + (r - 9) : 20 + (r - 17);
	s.out_mn = r == 0 ? 0 : r == 1 ? 2 : r == 2 ? 4 : -1;
	return s;
}

WV_FN void batch_sums_body(const Ctx& c, const BatchView& bv, const PartView& pv, bool dual, int partition_count, int plane2_component, int count)
{
	const int rows = dual ? 19 : 15;
	const int per_cand = partition_count * rows;
	const bool uniform_walk = dual || partition_count == 1;
	WV_FOR(k, count * per_cand)
	{
		const int ci = k / per_cand, rem = k - ci * per_cand;
		const int p = rem / rows, r = rem - p * rows;
		const SumLane s = dual ? sum_lane_2planes(bv, ci, r, plane2_component) : sum_lane_1plane(bv, ci, r);
		const int off = uniform_walk ? 0 : pv.off(p);
		const int n = uniform_walk ? c.T : pv.cnt(p);
		const float* xr = s.xrow + off;
		const float* dr = s.drow + off;
		float acc = s.init, mn = s.mn0, mx = 0.0f;
		auto step = [&](float x, float d)
		{
			const float P = s.A + s.B * x;
			const float dx = d * x;
			const float Q = s.E * d + s.F * dx;
			acc += P * Q;
			mn = f_run_min(x, mn);            // (weights and scale projections are >= +0, or NaN for an all-black partition's scale)
			mx = f_run_max(x, mx);
		};
		if (uniform_walk)
		{
			// every lane walks all T positions: a uniform loop, four positions per 128-bit read of each row
			int j = 0;
			for (; j + 4 <= n; j += 4)
			{
				const f4 xv = load4_aligned(xr + j), dv = load4_aligned(dr + j);
				step(xv.x, dv.x); step(xv.y, dv.y); step(xv.z, dv.z); step(xv.w, dv.w);
			}
			for (; j < n; j++) step(xr[j], dr[j]);
		}
		else
		{
			for (int j = 0; j < n; j++) step(xr[j], dr[j]);
		}
		float* o = bv.sums(ci, p);
		if (s.out >= 0) o[s.out] = acc;
		if (s.out_mn >= 0) { o[s.out_mn] = mn; o[s.out_mn + 1] = mx; }
	}
	WV_SYNC();
	if (dual)
	{
		// the weight sum every channel's plane goes with (ref: :1498-1500)
		WV_FOR64(k, count * 4)
		{
			const int ci = k >> 2, ch = k & 3;
			float* o = bv.sums(ci, 0);
			o[22 + ch] = ch == plane2_component ? o[27] : o[26];
		}
		WV_SYNC();
	}
}

// ---------------------------------------------------------------------------------------------
// Stage 3: the solve, one lane per (candidate, partition, channel)
// ---------------------------------------------------------------------------------------------
WV_FN void batch_solve_body(const Ctx& c, const BatchView& bv, const PartView& pv, bool dual, int partition_count, int plane2_component, int count)
{
	const TrialInfo& tr = c.tr();
	const BlkInfo& blk = c.blk();
	const float ls_weight = hadd_rgb_s(cw4_of(blk));
	const int pc4 = partition_count * 4;
	// (the endpoints a candidate starts from are the trial's ideal ones, merged across the planes: ref :497-540)
	WV_FOR(k, count * pc4)
	{
		const int ci = k / pc4, rem = k - ci * pc4;
		const int p = rem >> 2, ch = rem & 3;
		const int plane = (dual && ch == plane2_component) ? 1 : 0;
		float ep0 = tr.ep0[plane][p][ch], ep1 = tr.ep1[plane][p][ch], rgbs;
		if (dual) refit_solve_2planes(bv.sums(ci, 0), blk, tr.pm_dir[0][ch], c.T, ls_weight, ch, plane2_component, ep0, ep1, rgbs);
		else refit_solve_1plane(bv.sums(ci, p), blk, tr.pm_dir[p][ch], pv.cnt(p), ls_weight, ch, ep0, ep1, rgbs);
		const CandState st = batch_state(c, bv, ci);
		st.wep0[p * 4 + ch] = ep0;
		st.wep1[p * 4 + ch] = ep1;
		bv.rgbs(ci)[p * 4 + ch] = rgbs;
	}
	WV_SYNC();
	if (kHdr && (blk.rgb_lns || blk.alpha_lns))
	{
		WV_FOR64(k, count * partition_count)
		{
			const int ci = k / partition_count, p = k - ci * partition_count;
			const CandState st = batch_state(c, bv, ci);
			const f4 v0 = load4(&st.wep0[p * 4]), v1 = load4(&st.wep1[p * 4]);
			const f4 rgbo = dual ? refit_rgbo_2planes(bv.sums(ci, 0), blk, c.T, plane2_component, v0, v1)
			                     : refit_rgbo_1plane(bv.sums(ci, p), blk, pv.cnt(p), v0, v1);
			store4(&bv.rgbo(ci)[p * 4], rgbo);
		}
		WV_SYNC();
	}
}

/* Stages 1 to 3 as one out-of-line stage (a stage call costs some fifty instructions of call sequence and context
 * rebuild; the three share the views). */
WV_OUT void batch_refit(bool dual, int partition_count, int partition_packed, int plane2_component, int first, int count)
{
	const Ctx c = ctx_make();
	dual = wv_uniform(dual); partition_count = wv_uniform(partition_count); partition_packed = wv_uniform(partition_packed);
	plane2_component = wv_uniform(plane2_component); first = wv_uniform(first); count = wv_uniform(count);
	const BatchView bv = batch_view(c, dual, partition_count);
	const PartView pv = partition_count == 1 ? part_view_lds(c, 1, 0) : part_view_lds(c, partition_count, partition_packed);
	DUP_STAGE(c, DUP_BATCH_PREPARE, batch_prepare_body(c, bv, pv, dual, partition_count, first, count));
	DUP_STAGE(c, DUP_BATCH_SUMS, batch_sums_body(c, bv, pv, dual, partition_count, plane2_component, count));
	DUP_STAGE(c, DUP_BATCH_SOLVE, batch_solve_body(c, bv, pv, dual, partition_count, plane2_component, count));
}

// ---------------------------------------------------------------------------------------------
// Stage 4: pack the endpoints, one quad per (candidate, partition); the matched-format retry; decoded endpoints
// (ref: compress_symbolic.cpp:561-598 around pack_color_endpoints, astcenc_color_quantize.cpp:1909)
// ---------------------------------------------------------------------------------------------
/* the per-candidate "decoded endpoints are in place" flags: four bytes in the candidate's batch record */
WV_FN uint8_t* batch_have_decoded(const BatchView& bv, int ci) { return reinterpret_cast<uint8_t*>(bv.cand(ci).pad); }

#if ASTC_ENABLE_HDR
/* The HDR endpoint formats of candidate `ci`'s partitions (sub-modes side by side, wave_color_hdr.h), out of line like
 * refine_pack_hdr; `retry`: at the retry level into the retry buffers. */
WV_OUT void batch_pack_hdr(bool dual, int partition_count, int first, int ci, int retry)
{
	const Ctx c = ctx_make();
	dual = wv_uniform(dual); partition_count = wv_uniform(partition_count); first = wv_uniform(first); ci = wv_uniform(ci); retry = wv_uniform(retry);
	TrialInfo& tr = c.tr();
	const BatchView bv = batch_view(c, dual, partition_count);
	const CandState st = batch_state(c, bv, ci);
	uint8_t* colorvals = reinterpret_cast<uint8_t*>(&tr.ibox[32]);
	uint8_t* fmts = colorvals + 32;
	uint8_t* tries = reinterpret_cast<uint8_t*>(tr.fbox);              // sub-mode records (the mailbox is idle in this step)
	static_assert(sizeof(tr.fbox) >= 4 * HDR_TRY_LANES * HDR_TRY_BYTES, "sub-mode records do not fit the mailbox");
	WV_FOR64(j, partition_count) { if (!retry && endpoint_format_is_hdr(tr.cand_formats[first + ci][j])) batch_have_decoded(bv, ci)[j] = 0; }
	ColorTabs t; t.unq_to_uq = bv.ctab(retry ? bv.nb : ci);
	pack_endpoints_hdr(t, st.wep0, st.wep1, bv.rgbo(ci), partition_count, tr.cand_formats[first + ci], retry ? colorvals : st.colors,
	                   retry ? fmts : st.formats, tries);
	WV_SYNC();
}
#endif

/* Which candidates get the retry at the quant level that matched formats allow (ref: :571-598). */
WV_FN void batch_pack_decide_body(const Ctx& c, const BatchView& bv, bool dual, int partition_count, int count)
{
	WV_FOR64(ci, count)
	{
		const CandState st = batch_state(c, bv, ci);
		BatchCand& m = bv.cand(ci);
		bool retry = !dual && partition_count >= 2 && m.color_quant != m.color_quant_mod;
		for (int j = 1; j < partition_count; j++) retry = retry && st.formats[j] == st.formats[0];
		m.retry = retry ? 1 : 0;
		st.meta[0] = (uint8_t)m.color_quant;
		st.meta[1] = 0;
		st.meta[2] = 0;
		st.meta[3] = 0;
	}
	WV_SYNC();
}

#if ASTC_ENABLE_HDR
/* (builds with HDR formats: the HDR packing stages run between the first packing and this) */
WV_OUT void batch_pack_decide(bool dual, int partition_count, int count)
{
	const Ctx c = ctx_make();
	dual = wv_uniform(dual); partition_count = wv_uniform(partition_count); count = wv_uniform(count);
	batch_pack_decide_body(c, batch_view(c, dual, partition_count), dual, partition_count, count);
}
#endif

/* First packing of every (candidate, partition). */
WV_OUT void batch_pack_first(bool dual, int partition_count, int first, int count)
{
	const Ctx c = ctx_make();
	dual = wv_uniform(dual); partition_count = wv_uniform(partition_count); first = wv_uniform(first); count = wv_uniform(count);
	TrialInfo& tr = c.tr();
	const BatchView bv = batch_view(c, dual, partition_count);
	const int profile = c.cfg->profile;
	// (at most sixteen quads: make_lds_layout keeps candidates per batch x partition limit within that)
	const uint32_t pc_inv = (65536u + (uint32_t)partition_count - 1u) / (uint32_t)partition_count;     // k / pc == (k * inv) >> 16 for small k
	WV_QUADS16(k, count * partition_count)
	{
		const int ci = (int)(((uint32_t)k * pc_inv) >> 16), p = k - ci * partition_count;
		const CandState st = batch_state(c, bv, ci);
		const int requested = tr.cand_formats[first + ci][p];
		if (kHdr && endpoint_format_is_hdr(requested)) continue;
		ColorTabs t; t.unq_to_uq = bv.ctab(ci);
		const QPacked r = pack_endpoints_quad(t, q_load(st.wep0 + 4 * p), q_load(st.wep1 + 4 * p), q_load(bv.rgbs(ci) + 4 * p), requested,
		                                      st.colors + 8 * p, bv.cand(ci).color_quant);
		Q_ONCE { st.formats[p] = (uint8_t)r.format; batch_have_decoded(bv, ci)[p] = r.decoded_valid ? 1 : 0; }
		if (r.decoded_valid)
		{
			// the 8-bit endpoints a decoder sees, expanded to the 16 bits the scoring works in
			// (ref: unpack_color_endpoints, color_unquantize.cpp:980-1022: LDR formats in every profile)
			auto widen = [profile](int v) { return profile == 0 ? (v << 8) | 0x80 : v * 257; };
			q_store_i32(bv.dec(ci, p), q_mapi(r.decoded.e0, widen));
			q_store_i32(bv.dec(ci, p) + 4, q_mapi(r.decoded.e1, widen));
		}
	}
	WV_SYNC();
#if !ASTC_ENABLE_HDR
	batch_pack_decide_body(c, bv, dual, partition_count, count);      // (same stage: nothing runs in between in the LDR builds)
#endif
}

/* The retry of candidate `ci` at the higher quant level, into the retry buffers ... */
WV_OUT void batch_pack_retry(int partition_count, int first, int ci)
{
	const Ctx c = ctx_make();
	partition_count = wv_uniform(partition_count); first = wv_uniform(first); ci = wv_uniform(ci);
	TrialInfo& tr = c.tr();
	const BatchView bv = batch_view(c, false, partition_count);
	uint8_t* colorvals = reinterpret_cast<uint8_t*>(&tr.ibox[32]);   // [4][8] retry copy
	uint8_t* fmts = colorvals + 32;                                   // [4]
	const CandState st = batch_state(c, bv, ci);
	const int qmod = wv_uniform(bv.cand(ci).color_quant_mod);
	stage_quads_nosync(bv.ctab(bv.nb), c.table(c.root->off_color_unquant_to_uquant) + (uint32_t)(qmod - QUANT_6) * 512u, 512);
	WV_SYNC();
	ColorTabs t; t.unq_to_uq = bv.ctab(bv.nb);
	WV_QUADS16(p, partition_count)
	{
		const int requested = tr.cand_formats[first + ci][p];
		if (kHdr && endpoint_format_is_hdr(requested)) continue;
		const QPacked r = pack_endpoints_quad(t, q_load(st.wep0 + 4 * p), q_load(st.wep1 + 4 * p), q_load(bv.rgbs(ci) + 4 * p), requested,
		                                      colorvals + 8 * p, qmod);
		Q_ONCE { fmts[p] = (uint8_t)r.format; }
	}
	WV_SYNC();
}

/* ... its values replace the first packing's if every partition got the same format again. */
WV_OUT void batch_pack_retry_finish(int partition_count, int ci)
{
	const Ctx c = ctx_make();
	partition_count = wv_uniform(partition_count); ci = wv_uniform(ci);
	TrialInfo& tr = c.tr();
	const BatchView bv = batch_view(c, false, partition_count);
	const uint8_t* colorvals = reinterpret_cast<const uint8_t*>(&tr.ibox[32]);
	const uint8_t* fmts = colorvals + 32;
	const CandState st = batch_state(c, bv, ci);
	const int qmod = wv_uniform(bv.cand(ci).color_quant_mod);
	bool all_same_mod = true;
	for (int j = 1; j < partition_count; j++) all_same_mod = all_same_mod && fmts[j] == fmts[0];
	if (wv_uniform(all_same_mod))
	{
		WV_FOR64(k, partition_count * 8) { st.colors[k] = colorvals[k]; }
		WV_FOR64(j, partition_count) { st.formats[j] = fmts[j]; batch_have_decoded(bv, ci)[j] = 0; }
		WV_ONE { st.meta[0] = (uint8_t)qmod; st.meta[1] = 1; }
	}
	WV_SYNC();
}

/* The decoded endpoints the packing did not leave behind. */
WV_FN void batch_decode_body(const Ctx& c, const BatchView& bv, int partition_count, int count)
{
	const int profile = c.cfg->profile;
	const uint32_t pc_inv = (65536u + (uint32_t)partition_count - 1u) / (uint32_t)partition_count;
	WV_FOR64(k, count * partition_count)
	{
		const int ci = (int)(((uint32_t)k * pc_inv) >> 16), p = k - ci * partition_count;
		if (batch_have_decoded(bv, ci)[p]) continue;
		const CandState st = batch_state(c, bv, ci);
		i4 e0, e1;
		unpack_color_endpoints(profile, st.formats[p], st.colors + 8 * p, e0, e1);
		int* o = bv.dec(ci, p);
		o[0] = e0.x; o[1] = e0.y; o[2] = e0.z; o[3] = e0.w;
		o[4] = e1.x; o[5] = e1.y; o[6] = e1.z; o[7] = e1.w;
	}
	WV_SYNC();
}

/* (the pieces are called from the kernel body: a stage function that called another one would have to save its return
 *  address, i.e. need a stack frame -- DESIGN.md section 3.1, "No scratch memory") */
__attribute__((always_inline)) WV_FN void batch_pack(bool dual, int partition_count, int first, int count)
{
	batch_pack_first(dual, partition_count, first, count);
#if ASTC_ENABLE_HDR
	for (int ci = 0; ci < count; ci++) batch_pack_hdr(dual, partition_count, first, ci, 0);
#endif
#if ASTC_ENABLE_HDR
	batch_pack_decide(dual, partition_count, count);
#endif
	if (!dual && partition_count >= 2)
	{
		const Ctx c = ctx_make();
		const BatchView bv = batch_view(c, dual, partition_count);
		for (int ci = 0; ci < count; ci++)
		{
			if (wv_uniform(bv.cand(ci).retry) == 0) continue;
			batch_pack_retry(partition_count, first, ci);
#if ASTC_ENABLE_HDR
			batch_pack_hdr(false, partition_count, first, ci, 1);
#endif
			batch_pack_retry_finish(partition_count, ci);
		}
	}
}

// ---------------------------------------------------------------------------------------------
// Stage 5: decode and score, one lane per (candidate, texel) (ref: compute_symbolic_block_difference_*,
// decompress_symbolic.cpp:313-505) -> CandState::errorval
// ---------------------------------------------------------------------------------------------
WV_FN void batch_score_terms_body(const Ctx& c, const BatchView& bv, bool dual, int partition_count, int partition_packed, int plane2_component, int count)
{
	const BlkInfo& blk = c.blk();
	const PartView pv = partition_count == 1 ? part_view_lds(c, 1, 0) : part_view_lds(c, partition_count, partition_packed);
	const int T = c.T;
	const int profile = c.cfg->profile;
	const bool u8 = (c.cfg->flags & (1u << 1)) || profile == 0;
	const bool rgbm = (c.cfg->flags & (1u << 6)) != 0;
	const bool fast_1p = !dual && partition_count == 1 && !rgbm;
	const int p2c = dual ? plane2_component : -1;
	const uint32_t t_inv = c.L->t_inv24;

	WV_FOR(k, count * T)
	{
		const int ci = (int)(((uint32_t)k * t_inv) >> 24), i = k - ci * T;
		// the position's texel: partition order for one-plane candidates, texel order for two-plane ones (batch_prepare)
		const int t = dual ? i : (int)pv.sorted[i];
		const int p = dual ? 0 : (int)pv.of_texel[t];
		const int* e = bv.dec(ci, p);
		const int w1 = bv.iw(ci, 0)[i];
		const int w2 = dual ? (int)bv.iw(ci, 1)[i] : w1;

		float col[4], old[4];
		for (int ch = 0; ch < 4; ch++)
		{
			const int w = (ch == p2c) ? w2 : w1;
			col[ch] = (float)lerp_channel(u8, e[ch], e[4 + ch], w);
			old[ch] = c.data(ch)[t];
		}
		if (rgbm)
		{
			if (col[3] == 0.0f) batch_state(c, bv, ci).meta[2] = 1;
			const float ms = c.cfg->rgbm_m_scale;
			for (int ch = 0; ch < 3; ch++)
			{
				col[ch] = col[ch] * col[3] * ms;
				old[ch] = old[ch] * old[3] * ms;
			}
			col[3] = 1.0f; old[3] = 1.0f;
		}
		float err[4];
		for (int ch = 0; ch < 4; ch++)
		{
			float e1 = f_abs(old[ch] - col[ch]);
			e1 = e1 < 1e15f ? e1 : 1e15f;
			err[ch] = e1 * e1;
		}
		float term;
		if (fast_1p)
		{
			term = err[0] * cw_of(blk, 0) + err[1] * cw_of(blk, 1) + err[2] * cw_of(blk, 2) + err[3] * cw_of(blk, 3);
		}
		else
		{
			const float d = hadd4(err[0] * cw_of(blk, 0), err[1] * cw_of(blk, 1), err[2] * cw_of(blk, 2), err[3] * cw_of(blk, 3));
			term = d < ERROR_CALC_DEFAULT ? d : ERROR_CALC_DEFAULT;
		}
		bv.term(ci)[i] = term;
	}
	WV_SYNC();
}

/* ... and the sums of the terms, in the reference's order. */
WV_FN void batch_score_sums_body(const Ctx& c, const BatchView& bv, bool dual, int partition_count, int count)
{
	const int T = c.T;
	const bool rgbm = (c.cfg->flags & (1u << 6)) != 0;
	const bool fast_1p = !dual && partition_count == 1 && !rgbm;
	if (fast_1p)
	{
		// four accumulators per candidate, one per lane of a quad, folded (a0 + a2) + (a1 + a3) (ref: :340-356 with hadd_s)
		WV_QUADS(ci, count)
		{
			const float* v = bv.term(ci);
			const qf acc = q_map_ch(q_splat(0.0f), [v, T](int l, float a) { for_texels_of_quarter(l, T, [&](int i) { a += v[i]; }); return a; });
			const float total = q_hadd(acc);
			Q_ONCE { *batch_state(c, bv, ci).errorval = total; }
		}
	}
	else
	{
		// strictly sequential sums (ref: :407-505); an RGBM block whose M decodes to zero somewhere is an error block (:366-394)
		WV_FOR64(ci, count)
		{
			const CandState st = batch_state(c, bv, ci);
			const float* v = bv.term(ci);
			float summa = 0.0f;
			for (int i = 0; i < T; i++) summa += v[i];
			if (rgbm && st.meta[2]) summa = ERROR_CALC_DEFAULT;
			*st.errorval = summa;
		}
	}
	WV_SYNC();
}

/* One stage for the three steps (a stage call costs some fifty instructions of call sequence and context rebuild). */
WV_OUT void batch_score(bool dual, int partition_count, int partition_packed, int plane2_component, int count)
{
	const Ctx c = ctx_make();
	dual = wv_uniform(dual); partition_count = wv_uniform(partition_count); partition_packed = wv_uniform(partition_packed);
	plane2_component = wv_uniform(plane2_component); count = wv_uniform(count);
	const BatchView bv = batch_view(c, dual, partition_count);
	batch_decode_body(c, bv, partition_count, count);
	batch_score_terms_body(c, bv, dual, partition_count, partition_packed, plane2_component, count);
	batch_score_sums_body(c, bv, dual, partition_count, count);
}

} } // namespace astcd::ASTC_VARIANT
