// SPDX-License-Identifier: Apache-2.0
// Weight-grid stages of one trial:
//   * ideal weights on every decimated grid      ref: compute_ideal_weights_for_decimation
//                                                     Source/astcenc_ideal_endpoints_and_weights.cpp:845-971
//   * angular search for low/high weight bounds  ref: compute_angular_offsets / compute_lowest_and_highest_weight /
//                                                     compute_angular_endpoints_for_quant_levels
//                                                     Source/astcenc_weight_align.cpp:94-355
//   * quantize a grid and score it               ref: compute_quantized_weights_for_decimation :974-1080
//                                                     compute_error_of_weight_set_{1plane,2planes} :688-842
#pragma once
#include "wave_ctx.h"
#include "wave_quad.h"

namespace astcd { inline namespace ASTC_VARIANT {

/* Bilinear infill of one texel from a weight array (ref: bilinear_infill_vla[_2] :38-97).  `tw` / `tcf` are the
 * per-texel records of a grid (DecimationInfo: [T][4] indices / contributions, usually the copy staged in LDS): one
 * 32-bit and one 128-bit read.  Unused taps have zero contribution, so the 4-tap form equals the reference's
 * count-specialised forms whenever max_texel_weight_count > 2; the 2-tap form must be used otherwise. */
WV_FN float infill4(const float* wts, const uint8_t* tw, const float* tcf, int T, int t)
{
	(void)T;
	const uint32_t idx = reinterpret_cast<const uint32_t*>(tw)[t];
	const f4 cf = load4_aligned(tcf + 4 * t);
	return (wts[idx & 0xFFu] * cf.x + wts[(idx >> 8) & 0xFFu] * cf.y) + (wts[(idx >> 16) & 0xFFu] * cf.z + wts[idx >> 24] * cf.w);
}
WV_FN float infill2(const float* wts, const uint8_t* tw, const float* tcf, int T, int t)
{
	(void)T;
	const uint32_t idx = reinterpret_cast<const uint32_t*>(tw)[t];
	const float c0 = tcf[4 * t], c1 = tcf[4 * t + 1];
	return (wts[idx & 0xFFu] * c0 + wts[(idx >> 8) & 0xFFu] * c1);
}

/* The same from the table blob itself (DecimationInfo::off_texel_weights / off_texel_contribs_f as byte offsets): the
 * texel's four weight indices arrive in one 32-bit load and its four contributions in one 128-bit load, addressed as
 * uniform base + 32-bit offset.  `four_taps`: the grid has texels with more than two weights (else the 2-tap form). */
struct TexelTaps { uint32_t idx; float c0, c1, c2, c3; };
struct TapContribs { float c0, c1, c2, c3; };
WV_FN TexelTaps texel_taps_at(const uint8_t* tab, uint32_t idx_off, uint32_t f4_off, uint32_t t)
{
	TexelTaps r;
	r.idx = table_at_byte<uint32_t>(tab, idx_off + 4u * t);
	const TapContribs cf = table_at_byte<TapContribs>(tab, f4_off + 16u * t);
	r.c0 = cf.c0; r.c1 = cf.c1; r.c2 = cf.c2; r.c3 = cf.c3;
	return r;
}
WV_FN float infill_taps_at(const float* wts, const uint8_t* tab, uint32_t idx_off, uint32_t f4_off, uint32_t t, bool four_taps)
{
	const TexelTaps k = texel_taps_at(tab, idx_off, f4_off, t);
	const float lo = wts[k.idx & 0xFFu] * k.c0 + wts[(k.idx >> 8) & 0xFFu] * k.c1;
	const float hi = wts[(k.idx >> 16) & 0xFFu] * k.c2 + wts[k.idx >> 24] * k.c3;
	return four_taps ? lo + hi : lo;
}

// weights per round trip in the angular search's two per-(grid, step) loops.  Measured on config 2: 8 -> 4 +1.2 %
// (the tail of a group is padded work for the whole wavefront, and four weights are eight table loads in flight)
#ifndef ASTC_ANG_GROUP
#define ASTC_ANG_GROUP 4
#endif
// ... and whether the next group's table loads are requested before the current group is added up: on in the decimation
// sweeps (ASTC_DWI_PREFETCH: a weight of a coarse grid has up to six groups of taps, each an L2 round trip), off in the angular
// search (ASTC_ANG_PREFETCH: measured +-0 in every build, profiles/r05h)
#ifndef ASTC_DWI_PREFETCH
#define ASTC_DWI_PREFETCH 1
#endif
#ifndef ASTC_ANG_PREFETCH
#define ASTC_ANG_PREFETCH 0
#endif

/* Row of the sin/cos tables an ideal weight selects in the angular search (ref: compute_angular_offsets,
 * weight_align.cpp:110-118).  It depends on the weight only, not on the angular step, so it is computed once when
 * the weight is final instead of once per (weight, step) in the search. */
WV_FN uint8_t angular_sample_row(float weight)
{
	float sample = v_clampzo(weight) * (SINCOS_STEPS - 1.0f);
	return (uint8_t)(int)(sample + 0.5f);
}

/* Tap k (0 .. 7) of a packed group: texel index and contribution. */
WV_FN uint32_t dwi_tap_texel(const DwiTap8& g, int k) { return (g.w[k >> 1] >> (16 * (k & 1))) & 0xFFu; }
WV_FN float dwi_tap_contrib(const DwiTap8& g, int k) { return (float)(int)((g.w[k >> 1] >> (16 * (k & 1) + 8)) & 0xFFu); }

/* One slot of sweep 1: initial guess for weight `sl.index` of its grid (ref: :877-905; direct grids copy, :858-866).
 * Eight taps per round trip to the table: the group's (texel, contribution) pairs arrive with one 128-bit load, then all
 * gathers are issued, then the (strictly ordered) accumulation runs.  The padding taps of the last group have
 * contribution 0: they add +0.0 to both sums (which are never -0), i.e. nothing. */
WV_FN float dwi_initial_weight(const Ctx& c, const DwiSlot& sl)
{
	const int plane = (sl.flags >> 1) & 1;
	const float* eiw = c.ei_w(plane);
	const float* eiwes = c.ei_wes(plane);
	if (sl.flags & 1) return eiw[sl.index];
	// (ref: :872-905 reads one scale for every texel when the trial's scales are all equal, is_constant_weight_error_scale; the
	//  per-texel row holds that same value in every entry then (ideal_colors_and_weights_*), so reading the row gives the same
	//  bits -- and a select per tap on a lane-variant flag, the plane's, cost more than the read it saved)
	float weight_weight = 1e-10f;
	float initial_weight = 0.0f;
	const int cnt = sl.taps;
	// (the next group of eight taps is requested before this group's gathers and sums: a weight of a coarse grid on a large
	//  footprint has up to six groups, each an L2 round trip)
	DwiTap8 g_next = {};
	if (ASTC_DWI_PREFETCH) g_next = table_at_byte<DwiTap8>(c.tab, sl.wt_off);
	for (int j0 = 0; j0 < cnt; j0 += 8)
	{
		DwiTap8 g;
		if (ASTC_DWI_PREFETCH)
		{
			g = g_next;
			if (j0 + 8 < cnt) g_next = table_at_byte<DwiTap8>(c.tab, sl.wt_off + (uint32_t)(j0 + 8) * 2u);
		}
		else g = table_at_byte<DwiTap8>(c.tab, sl.wt_off + (uint32_t)j0 * 2u);
		// (in two halves: most weights of the larger grids have four taps or fewer, and the second half of their only group
		//  is all padding)
		#pragma unroll
		for (int h = 0; h < 8; h += 4)
		{
			if (h != 0 && j0 + h >= cnt) break;
			float iw[4], es[4];
			#pragma unroll
			for (int k = 0; k < 4; k++)
			{
				const uint32_t t = dwi_tap_texel(g, h + k);
				iw[k] = eiw[t];
				es[k] = eiwes[t];
			}
			#pragma unroll
			for (int k = 0; k < 4; k++)
			{
				const float contrib_weight = dwi_tap_contrib(g, h + k) * es[k];
				weight_weight += contrib_weight;
				initial_weight += iw[k] * contrib_weight;
			}
		}
	}
	return initial_weight / weight_weight;
}

/* One slot of sweep 3: the clamped gradient step (ref: :930-970); `inf` = the grid's infill at texel resolution. */
WV_FN float dwi_refined_weight(const Ctx& c, const DwiSlot& sl, const float* inf, float weight_val)
{
	const int plane = (sl.flags >> 1) & 1;
	const float* eiw = c.ei_w(plane);
	const float* eiwes = c.ei_wes(plane);
	float error_change0 = 1e-10f;
	float error_change1 = 0.0f;
	const int cnt = sl.taps;
	DwiTap8 g_next = {};
	if (ASTC_DWI_PREFETCH) g_next = table_at_byte<DwiTap8>(c.tab, sl.wt_off);
	for (int j0 = 0; j0 < cnt; j0 += 8)
	{
		DwiTap8 g;
		if (ASTC_DWI_PREFETCH)
		{
			g = g_next;
			if (j0 + 8 < cnt) g_next = table_at_byte<DwiTap8>(c.tab, sl.wt_off + (uint32_t)(j0 + 8) * 2u);
		}
		else g = table_at_byte<DwiTap8>(c.tab, sl.wt_off + (uint32_t)j0 * 2u);
		#pragma unroll
		for (int h = 0; h < 8; h += 4)
		{
			if (h != 0 && j0 + h >= cnt) break;
			float iw[4], ow[4], es[4];
			#pragma unroll
			for (int k = 0; k < 4; k++)
			{
				const uint32_t t = dwi_tap_texel(g, h + k);
				iw[k] = eiw[t];
				ow[k] = inf[t];
				es[k] = eiwes[t];
			}
			#pragma unroll
			for (int k = 0; k < 4; k++)
			{
				const float contrib = dwi_tap_contrib(g, h + k);
				const float scale = es[k] * contrib;
				error_change0 += contrib * scale;
				error_change1 += (ow[k] - iw[k]) * scale;
			}
		}
	}
	float step = (error_change1 * -16.0f) / error_change0;
	step = v_clamp(-0.25f, 0.25f, step);
	return weight_val + step;
}

/* Ideal weights on ALL referenced grids of a trial in three lane-parallel sweeps instead of three
 * per grid.
 *   nplanes        : 1 or 2 weight planes in this trial
 *   ref_mask       : quant levels allowed in this trial (grid is used if refprec & ref_mask)
 *   max_dm         : grids [0, max_dm) are considered
 * Uses the `uni` LDS region for the texel-resolution infill of up to uni/Tp (grid, plane) sets at a time. */
WV_FN void ideal_weights_all_grids(const Ctx& c, int nplanes, uint16_t ref_mask, int max_dm)
{
	const TableRoot& r = *c.root;
	const int T = c.T, Tp = c.Tp;
	const int cls = nplanes == 2 ? 1 : 0;          // trial class: selects the packing of the dwi region
	const DwiSlot* slots = reinterpret_cast<const DwiSlot*>(c.table(r.off_dwi_slots[cls]));
	const InfillSet* isets = reinterpret_cast<const InfillSet*>(c.table(r.off_infill_sets[cls]));
	float* dwi_base = reinterpret_cast<float*>(c.lds + c.L->dwi);
	uint8_t* isamp = c.isample();
	float* infilled = c.uni_f();
	const int cap_sets = (int)r.dwi_sets_per_chunk;
	// Sets are packed by ascending lowest quant level, so the sets this trial can use are a prefix
	// (the single-grid "always" pass of trial A is the exception and simply filters by grid index).
	const int nsets_all = (int)r.dwi_sets[cls];
	int quant_limit = 0;
	while (quant_limit < 11 && (ref_mask >> (quant_limit + 1))) quant_limit++;
	const bool all_grids = max_dm >= (int)r.decimation_mode_count_selected;
	const int nsets_used = all_grids ? (int)r.dwi_used_sets[cls][quant_limit] : nsets_all;
	const uint32_t t_inv = c.L->t_inv24;                                       // k / T == (k * t_inv) >> 24
	const DwiOrderDir& dir = reinterpret_cast<const DwiOrderDir*>(c.table(r.off_dwi_order[cls]))[quant_limit];
	const bool sorted = all_grids && dir.chunks != 0;
	const DwiSlot* order = reinterpret_cast<const DwiSlot*>(c.table(dir.list_off));     // records in processing order, refprec = packed index

	// chunks of (grid, plane) sets whose texel-resolution infill fits the scratch region
	int p0 = 0, chunk = 0;
	while (p0 < nsets_used)
	{
		int p1 = p0 + cap_sets;
		if (p1 > nsets_used) p1 = nsets_used;
		const int nsets = p1 - p0;
		// the chunk's slots: in the table's balanced order (longest tap lists first), or -- for the single-grid
		// "always" pass -- every slot of the chunk's sets, filtered by grid
		const int k_begin = (int)isets[p0].dwi_offset;
		const int k_end = p1 < nsets_all ? (int)isets[p1].dwi_offset : (int)r.dwi_total_floats[cls];
		const int o_begin = sorted ? (int)dir.chunk_start[chunk] : 0;
		const int n_items = sorted ? (int)dir.chunk_start[chunk + 1] - o_begin : k_end - k_begin;

		// sweep 1: initial guess for every (grid, plane, weight).  The next iteration's record is requested before this
		// one's taps are (the sweeps are bound by the latency of their table loads, not by arithmetic).
		{ PROF_SCOPE(c, PS_DEC1);
		if (sorted)
		{
#if WV_DEVICE
			DwiSlot ahead = {};
			if (WV_LANE < n_items) ahead = table_at(order, (uint32_t)(o_begin + WV_LANE));
#endif
			WV_FOR(jj, n_items)
			{
#if WV_DEVICE
				const DwiSlot sl = ahead;
				if (jj + 64 < n_items) ahead = table_at(order, (uint32_t)(o_begin + jj + 64));
#else
				const DwiSlot sl = table_at(order, (uint32_t)(o_begin + jj));
#endif
				const int k = sl.refprec;
				const float w0 = dwi_initial_weight(c, sl);
				dwi_base[k] = w0;
				if (sl.flags & 1) isamp[k] = angular_sample_row(w0);     // copied weights are final here
			}
		}
		else
		{
			WV_FOR(j, n_items)
			{
				const int k = k_begin + j;
				const DwiSlot sl = table_at(slots, (uint32_t)k);
				if (sl.taps == 0 || (int)sl.dm >= max_dm || !(sl.refprec & ref_mask)) continue;
				const float w0 = dwi_initial_weight(c, sl);
				dwi_base[k] = w0;
				if (sl.flags & 1) isamp[k] = angular_sample_row(w0);
			}
		}
		WV_SYNC(); }

		// sweep 2: infill to texel resolution (ref: :910-926); the set record of the next iteration is requested before this
		// iteration's table loads are
		{ PROF_SCOPE(c, PS_DEC2);
#if WV_DEVICE
		InfillSet ahead = {};
		if (WV_LANE < nsets * T) ahead = table_at(isets, (uint32_t)(p0 + (int)(((uint32_t)WV_LANE * t_inv) >> 24)));
#endif
		WV_FOR(k, nsets * T)
		{
			int set = (int)(((uint32_t)k * t_inv) >> 24), t = k - set * T;
#if WV_DEVICE
			const InfillSet is = ahead;
			if (k + 64 < nsets * T) ahead = table_at(isets, (uint32_t)(p0 + (int)(((uint32_t)(k + 64) * t_inv) >> 24)));
#else
			const InfillSet is = table_at(isets, (uint32_t)(p0 + set));
#endif
			if (is.direct || (int)is.dm >= max_dm || !(is.refprec & ref_mask)) continue;
			const float* wts = dwi_base + is.dwi_offset;
			infilled[set * Tp + t] = infill_taps_at(wts, c.tab, is.tw_off, is.tcf_off, (uint32_t)t, is.taps > 2);
		}
		WV_SYNC(); }

		// sweep 3: one clamped gradient step
		{ PROF_SCOPE(c, PS_DEC3);
		if (sorted)
		{
#if WV_DEVICE
			DwiSlot ahead = {};
			if (WV_LANE < n_items) ahead = table_at(order, (uint32_t)(o_begin + WV_LANE));
#endif
			WV_FOR(jj, n_items)
			{
#if WV_DEVICE
				const DwiSlot sl = ahead;
				if (jj + 64 < n_items) ahead = table_at(order, (uint32_t)(o_begin + jj + 64));
#else
				const DwiSlot sl = table_at(order, (uint32_t)(o_begin + jj));
#endif
				if (sl.flags & 1) continue;
				const int k = sl.refprec;
				const float w1 = dwi_refined_weight(c, sl, infilled + ((int)sl.set - p0) * Tp, dwi_base[k]);
				dwi_base[k] = w1;
				isamp[k] = angular_sample_row(w1);
			}
		}
		else
		{
			WV_FOR(j, n_items)
			{
				const int k = k_begin + j;
				const DwiSlot sl = table_at(slots, (uint32_t)k);
				if ((sl.flags & 1) || sl.taps == 0 || (int)sl.dm >= max_dm || !(sl.refprec & ref_mask)) continue;
				const float w1 = dwi_refined_weight(c, sl, infilled + ((int)sl.set - p0) * Tp, dwi_base[k]);
				dwi_base[k] = w1;
				isamp[k] = angular_sample_row(w1);
			}
		}
		WV_SYNC(); }
		p0 = p1;
		chunk++;
	}
}

// ---------------------------------------------------------------------------------------------
// Angular endpoint search
// ---------------------------------------------------------------------------------------------

WV_FN int steps_for_quant_level(int q)
{
	const uint8_t s[12] = { 2, 3, 4, 5, 6, 8, 10, 12, 16, 20, 24, 32 };
	return s[q];
}

/* Run the angular search for `nsets` weight sets at once.  Set s reads its ideal weights from
 * set_weights(s) (W = set_wcount(s) values) with max quant level set_maxq(s) and writes
 * low/high[quant 0..maxq] to set_out(s)[q*2 + {0,1}].
 *
 * Phase 1: one lane per (set, angular step)   -> offset, lowest weight, span, 3 error terms
 * Phase 2: one lane per (set, quant level)    -> best (error, step, cut) for that span, low/high
 * Sets are processed in batches of up to 64 (set, step) pairs. */
struct AngSet {
	const float* weights;
	const uint8_t* rows;    // sin/cos table row per weight (angular_sample_row)
	float* out;             // (low, high) pairs of the quant levels in `used`, ascending
	int wcount;
	int maxq;
	uint16_t used;          // quant levels <= QUANT_12 that a block mode of this grid uses (bit mask)
};

/* What the search keeps per set while it runs (in the trial's float mailbox, 64 sets at a time): the phases read this
 * one 8-byte record instead of walking decimation mode -> decimation info -> layout again (get_set) on every lane of
 * every phase. */
struct AngRec {
	uint16_t w_off;     // the set's ideal weights: float index from the start of LDS
	uint16_t out_off;   // ... its (low, high) pairs
	uint8_t wcount, maxq, used, steps;
};
static_assert(sizeof(AngRec) == 8, "64 records fill the 512-byte mailbox");

template <typename SetFn>
WV_FN void angular_endpoints(const Ctx& c, int nsets_all, SetFn get_set)
{
	float* ang = c.ang();               // [64][ANG_PAIR_STRIDE]: offset, lowest, span, err, cut_low, cut_high
	TrialInfo& tr = c.tr();
	struct CosSin { float cs, sn; };
	const CosSin* cos_sin_table = reinterpret_cast<const CosSin*>(c.table(c.root->off_cos_sin_table));
	float* ldsf = reinterpret_cast<float*>(c.lds);
	AngRec* recs = reinterpret_cast<AngRec*>(tr.fbox);
	static_assert(sizeof(tr.fbox) >= 64 * sizeof(AngRec), "set records do not fit the mailbox");
	const int row_bias = (int)(c.L->dwi >> 2);       // isample()[k] belongs to the weight at float index row_bias + k
	uint8_t* pair_set = reinterpret_cast<uint8_t*>(&tr.ibox[32]);      // [64] batch-local set of each (set, step) pair
	uint8_t* set_steps = reinterpret_cast<uint8_t*>(&tr.ibox[48]);     // [32] steps of each set of the batch

	for (int g0 = 0; g0 < nsets_all; g0 += 64)
	{
	const int nsets = i_min(64, nsets_all - g0);
	// Per set: its record, and the lowest and highest weight (ref: compute_angular_offsets keeps them per step, :104-127:
	// they are the same for every step of a set) -- one QUAD per set, lane l looks at weights l, l + 4, ... (exact: the
	// weights are finite) -- parked in the first two floats of the set's output row until the set's own phase 2 overwrites
	// them with the bounds of quant level 0.
	WV_QUADS(s, nsets)
	{
		const AngSet a = get_set(g0 + s);
		qf mn = q_splat(3.402823466e+38f), mx = q_splat(-3.402823466e+38f);
		Q_LANES(l)
		{
			for (int j = l; j < a.wcount; j += 4)
			{
				const float w = a.weights[j];
				QV(mn, l) = w < QV(mn, l) ? w : QV(mn, l);
				QV(mx, l) = w > QV(mx, l) ? w : QV(mx, l);
			}
		}
		const float min_weight = q_hmin(mn), max_weight = q_hmax(mx);
		Q_ONCE
		{
			a.out[0] = min_weight;
			a.out[1] = max_weight;
			// The slots between the set's last weight and the next multiple of four (they belong to the set: the packing is
			// in fours) select the table's all-zero row: phase 1 walks the weights four at a time and needs no tail handling
			// in its sum of cosines / sines.
			static_assert(ASTC_ANG_GROUP <= 4, "the padding of a set's slots is to a multiple of four");
			for (int j = a.wcount; j < ((a.wcount + 3) & ~3); j++) const_cast<uint8_t*>(a.rows)[j] = (uint8_t)SINCOS_STEPS;
			AngRec r;
			r.w_off = (uint16_t)(a.weights - ldsf);
			r.out_off = (uint16_t)(a.out - ldsf);
			r.wcount = (uint8_t)a.wcount; r.maxq = (uint8_t)a.maxq; r.used = (uint8_t)a.used;
			r.steps = (uint8_t)steps_for_quant_level(a.maxq);
			recs[s] = r;
		}
	}
	WV_SYNC();
	// Batches of sets: as many consecutive sets as give at most 64 (set, step) pairs (and at most 32 sets).
	// ibox[s - s0] = first pair slot of set s.  On the device lane s holds the steps of set s and their running total, so a
	// batch boundary is one ballot and every set of the batch writes its own two entries -- no loop over the sets.
#if WV_DEVICE
	const int my_steps = WV_LANE < nsets ? (int)recs[WV_LANE].steps : 0;
	int steps_through = my_steps;                  // steps of sets 0 .. lane (integers: any order of adding is exact)
	for (int d = 1; d < 64; d <<= 1)
	{
		const int up = __shfl_up(steps_through, d);
		if (WV_LANE >= d) steps_through += up;
	}
	int steps_before_batch = 0;
#else
	LaneArray128 steps_of;
	steps_of.clear();
	WV_FOR64(s, nsets) { steps_of.set(s, recs[s].steps); }
#endif

	int s0 = 0;
	while (s0 < nsets)
	{
		// batch [s0, s1): total steps <= 64; ibox[s - s0] = first pair slot of set s
		int s1, pairs;
#if WV_DEVICE
		{
			const bool fits = WV_LANE >= s0 && WV_LANE < nsets && WV_LANE - s0 < 32 && steps_through - steps_before_batch <= 64;
			const unsigned long long run = __ballot(fits) >> s0;                 // (the running total only grows: a run of ones from bit 0)
			const int n = ~run ? (int)__builtin_ctzll(~run) : 64;
			s1 = s0 + n;
			pairs = __builtin_amdgcn_readlane(steps_through, s1 - 1) - steps_before_batch;
			if (WV_LANE >= s0 && WV_LANE < s1)
			{
				tr.ibox[WV_LANE - s0] = steps_through - my_steps - steps_before_batch;
				set_steps[WV_LANE - s0] = (uint8_t)my_steps;
			}
			steps_before_batch += pairs;
		}
#else
		s1 = s0; pairs = 0;
		while (s1 < nsets && s1 - s0 < 32)
		{
			int steps = steps_of.get(s1);
			if (pairs + steps > 64) break;
			WV_ONE { tr.ibox[s1 - s0] = pairs; set_steps[s1 - s0] = (uint8_t)steps; }
			pairs += steps;
			s1++;
		}
#endif
		WV_SYNC();
#if WV_DEVICE
		{
			// the set of every pair slot: each set marks its first slot, the running maximum fills in the rest (every set
			// has at least two steps, so no two sets start in one slot)
			pair_set[WV_LANE] = 0;
			WV_SYNC();
			if (WV_LANE < s1 - s0) pair_set[tr.ibox[WV_LANE]] = (uint8_t)WV_LANE;
			WV_SYNC();
			int v = pair_set[WV_LANE];
			for (int d = 1; d < 64; d <<= 1)
			{
				const int up = __shfl_up(v, d);
				if (WV_LANE >= d) v = i_max(v, up);
			}
			WV_SYNC();
			pair_set[WV_LANE] = (uint8_t)v;
		}
#else
		WV_FOR64(sl, s1 - s0)
		{
			const int base = tr.ibox[sl], steps = set_steps[sl];
			for (int j = 0; j < steps; j++) pair_set[base + j] = (uint8_t)sl;
		}
#endif
		WV_SYNC();

		{ PROF_SCOPE(c, PS_ANG1);
		WV_FOR64(k, pairs)
		{
			// (set, step) of pair k
			const int sl = pair_set[k];
			const int base = tr.ibox[sl];
			const AngRec a = recs[s0 + sl];
			const int steps = a.steps;
			int sp = k - base;
			const int W = a.wcount;
			const float* wv = ldsf + a.w_off;
			const uint8_t* rows = c.isample() + ((int)a.w_off - row_bias);
			const float* aout = ldsf + a.out_off;
			const float min_weight = aout[0], max_weight = aout[1];

			// compute_angular_offsets (ref: weight_align.cpp:94-140)
			float anglesum_x = 0.0f, anglesum_y = 0.0f;
			// groups of ASTC_ANG_GROUP weights: LDS reads, then all table loads, then the ordered accumulation -- the next group's
			// (cos, sin) pairs are requested before this group's are added up (they come from L2: one such round trip per group,
			// one after the other, was most of this loop's time)
			{
				float cs[ASTC_ANG_GROUP], sn[ASTC_ANG_GROUP];
				auto fetch = [&](int j0)
				{
					uint32_t row[ASTC_ANG_GROUP];
					#pragma unroll
					for (int u = 0; u < ASTC_ANG_GROUP; u++) { row[u] = rows[j0 + u]; }      // (past W: the all-zero row, see the pre-pass)
					#pragma unroll
					for (int u = 0; u < ASTC_ANG_GROUP; u++)
					{
						const uint32_t at = row[u] * (uint32_t)ANGULAR_STEPS + (uint32_t)sp;
						const CosSin both = table_at(cos_sin_table, at);      // one 64-bit load
						cs[u] = both.cs;
						sn[u] = both.sn;
					}
				};
				if (ASTC_ANG_PREFETCH && W > 0) fetch(0);
				for (int j0 = 0; j0 < W; j0 += ASTC_ANG_GROUP)
				{
					float c_now[ASTC_ANG_GROUP], s_now[ASTC_ANG_GROUP];
					if (!ASTC_ANG_PREFETCH) fetch(j0);
					#pragma unroll
					for (int u = 0; u < ASTC_ANG_GROUP; u++) { c_now[u] = cs[u]; s_now[u] = sn[u]; }
					if (ASTC_ANG_PREFETCH && j0 + ASTC_ANG_GROUP < W) fetch(j0 + ASTC_ANG_GROUP);
					#pragma unroll
					for (int u = 0; u < ASTC_ANG_GROUP; u++)
					{
						// (past the set's last weight: +0.0 from the table's zero row; the sums start at +0.0 and are never -0.0)
						anglesum_x += c_now[u];
						anglesum_y += s_now[u];
					}
				}
			}
			float angle = ref_atan2(anglesum_y, anglesum_x);
			angle = angle == angle ? angle : 0.0f;
			float offset = angle * (1.0f / (2.0f * 3.14159265358979323846f));

			// compute_lowest_and_highest_weight (ref: weight_align.cpp:160-245)
			float rcp_stepsize = (float)sp + 1.0f;
			float errval = 0.0f, cut_low = 0.0f, cut_high = 0.0f;
			float minidx = f_round(min_weight * rcp_stepsize - offset);
			float maxidx = f_round(max_weight * rcp_stepsize - offset);
			// (four weights per round trip while four are left, then the tail one by one: every lane walks exactly its own
			//  set's weights -- the sets of a batch are sorted by weight count, so the lanes mostly stop together -- and a
			//  weight costs its sixteen operations, nothing for masking slots past the end)
			auto one_weight = [&](float w)
			{
				float sval = w * rcp_stepsize - offset;
				float svalrte = f_round(sval);
				float diff = sval - svalrte;
				errval += diff * diff;
				// (ref: (cut + 1) -/+ (2 dif), weight_align.cpp:212-224; |dif| <= 0.5, so 2 dif is exact and the fused form is the same float)
				if (svalrte == minidx) cut_low = f_add_doubled(cut_low + 1.0f, diff, -2.0f);
				if (svalrte == maxidx) cut_high = f_add_doubled(cut_high + 1.0f, diff, 2.0f);
			};
			{
				int j = 0;
				for (; j + 4 <= W; j += 4)
				{
					const float w0 = wv[j], w1 = wv[j + 1], w2 = wv[j + 2], w3 = wv[j + 3];
					one_weight(w0); one_weight(w1); one_weight(w2); one_weight(w3);
				}
				for (; j < W; j++) one_weight(wv[j]);
			}
			int max_quant_steps = steps;
			int span = (int)(maxidx - minidx + 1.0f);
			span = i_min(span, max_quant_steps + 3);
			span = i_max(span, 2);
			float ssize = 1.0f / rcp_stepsize;
			float errscale = ssize * ssize;

			float* o = ang + k * (int)ANG_PAIR_STRIDE;
			o[0] = offset;
			o[1] = minidx;
			o[2] = int_as_float(span);
			o[3] = errval * errscale;
			o[4] = cut_low * errscale;
			o[5] = cut_high * errscale;
		} }
		// phase 2: (set, quant) lanes (ref: weight_align.cpp:285-354)
		PROF_SCOPE(c, PS_ANG2);
		WV_FOR(k, (s1 - s0) * 8)
		{
			int s = s0 + (k >> 3), qi = k & 7;
			const AngRec a = recs[s];
			float* aout = ldsf + a.out_off;
			if (qi <= a.maxq && ((a.used >> qi) & 1u))
			{
				int steps = a.steps;
				const float* base = ang + tr.ibox[s - s0] * (int)ANG_PAIR_STRIDE;
				int q = steps_for_quant_level(qi);

				// sequential scan with the reference's update order and strict '>' tests,
				// restricted to the updates that land on span index q
				float best_err = ERROR_CALC_DEFAULT;
				float best_idx = -1.0f;
				float best_cut = 0.0f;
				for (int i = 0; i < steps; i++)
				{
					const float* r = base + i * (int)ANG_PAIR_STRIDE;
					const int idx_span = float_as_int(r[2]);
					const float err = r[3], cl = r[4], ch = r[5];
					// Which of the reference's three cases this step is for span index q (ref: :300-340), without branching
					// (every lane of the wave looks at a different (set, quant level)): the step offers up to two
					// candidates in the reference's order -- a case that does not apply offers an error no best can exceed.
					const int d = idx_span - q;                      // 0: exact span, 1: one end cut, 2: both ends cut
					const float e_cut_low = err + cl;
					const float first = d == 0 ? err : d == 1 ? e_cut_low : d == 2 ? e_cut_low + ch : 3.0e38f;
					const float first_cut = d == 0 ? 0.0f : 1.0f;
					const float second = d == 1 ? err + ch : 3.0e38f;
					if (best_err > first) { best_err = first; best_idx = (float)i; best_cut = first_cut; }
					if (best_err > second) { best_err = second; best_idx = (float)i; best_cut = 0.0f; }
				}

				int bsi = (int)best_idx;
				bsi = i_max(0, bsi);
				const float* r = base + bsi * (int)ANG_PAIR_STRIDE;
				float lwi = r[1] + best_cut;
				float hwi = lwi + (float)q - 1.0f;
				float stepsize = 1.0f / (1.0f + (float)bsi);
				const int slot = popcount32((uint32_t)a.used & ((1u << qi) - 1u));
				aout[slot * 2 + 0] = (r[0] + lwi) * stepsize;
				aout[slot * 2 + 1] = (r[0] + hwi) * stepsize;
			}
		}
		WV_SYNC();
		s0 = s1;
	}
	}
}

// ---------------------------------------------------------------------------------------------
// Quantize + score
// ---------------------------------------------------------------------------------------------

/* Quantization of one ideal weight (ref: compute_quantized_weights_for_decimation :1023-1046).
 * Returns the unquantized integer level 0..64; *outf receives the float reconstruction. */
struct QuantParams {
	float scale, scaled_low_bound, quant_level_m1, rscale, low_bound;
	int steps_m1;
};

WV_FN QuantParams quant_params(float low_bound, float high_bound, int quant_level)
{
	const float quant_levels_m1[12] = { 1.0f, 2.0f, 3.0f, 4.0f, 5.0f, 7.0f, 9.0f, 11.0f, 15.0f, 19.0f, 23.0f, 31.0f };
	const uint8_t levels[12] = { 2, 3, 4, 5, 6, 8, 10, 12, 16, 20, 24, 32 };
	QuantParams p;
	if (high_bound <= low_bound)
	{
		low_bound = 0.0f;
		high_bound = 1.0f;
	}
	float rscale = high_bound - low_bound;
	p.scale = 1.0f / rscale;
	p.scaled_low_bound = low_bound * p.scale;
	p.rscale = rscale * (1.0f / 64.0f);
	p.quant_level_m1 = quant_levels_m1[quant_level];
	p.low_bound = low_bound;
	p.steps_m1 = levels[quant_level] - 1;
	return p;
}

WV_FN int quantize_weight(const QuantParams& p, const uint8_t* quant_to_unquant, float ideal, float* outf)
{
	float ix = ideal * p.scale - p.scaled_low_bound;
	ix = v_clampzo(ix);
	float ix1 = ix * p.quant_level_m1;
	int weightl = (int)ix1;
	int weighth = i_min(weightl + 1, p.steps_m1);
	int ixli = quant_to_unquant[weightl];
	int ixhi = quant_to_unquant[weighth];
	float ixl = (float)ixli;
	float ixh = (float)ixhi;
	bool mask = (ixl + ixh) < (128.0f * ix);
	int weight = mask ? ixhi : ixli;
	ixl = mask ? ixh : ixl;
	*outf = ixl * p.rscale + p.low_bound;
	return weight;
}

/* 4-accumulator sum of v[0..n) in the reference's order: acc[l] += v[i] for i = l mod 4, then
 * (acc0 + acc2) + (acc1 + acc3).  Uniform (every lane computes the same value). */
WV_FN float sum4(const float* v, int n)
{
	float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
	int i = 0;
	for (; i + 3 < n; i += 4)
	{
		a0 += v[i]; a1 += v[i + 1]; a2 += v[i + 2]; a3 += v[i + 3];
	}
	if (i < n) a0 += v[i];
	if (i + 1 < n) a1 += v[i + 1];
	if (i + 2 < n) a2 += v[i + 2];
	return (a0 + a2) + (a1 + a3);
}

/* wv_sum4() over the T per-texel terms of a block (wave_ctx.h: for_texels_of_quarter). */
WV_FN float wv_sum4_texels(const float* v, int T)
{
#if WV_DEVICE
	float acc = 0.0f;
	if (WV_LANE < 4) for_texels_of_quarter(WV_LANE, T, [&](int i) { acc += v[i]; });
	const int bits = float_as_int(acc);
	const float a0 = int_as_float(__builtin_amdgcn_readlane(bits, 0)), a1 = int_as_float(__builtin_amdgcn_readlane(bits, 1));
	const float a2 = int_as_float(__builtin_amdgcn_readlane(bits, 2)), a3 = int_as_float(__builtin_amdgcn_readlane(bits, 3));
	return (a0 + a2) + (a1 + a3);
#else
	float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
	int i = 0;
	for (; i + 3 < T; i += 4) { a0 += v[i]; a1 += v[i + 1]; a2 += v[i + 2]; a3 += v[i + 3]; }
	if (i < T) a0 += v[i];
	if (i + 1 < T) a1 += v[i + 1];
	if (i + 2 < T) a2 += v[i + 2];
	return (a0 + a2) + (a1 + a3);
#endif
}

/* The same sum as a wave-level operation: the four accumulators run side by side on lanes 0..3 (a quarter of the
 * dependent additions of the one-lane form), then (acc0 + acc2) + (acc1 + acc3).  Uniform result; v must be visible to
 * all lanes (a WV_SYNC() after it was written). */
WV_FN float wv_sum4(const float* v, int n)
{
#if WV_DEVICE
	float acc = 0.0f;
	if (WV_LANE < 4) for (int i = WV_LANE; i < n; i += 4) acc += v[i];      // (n is not always the texel count: see wv_sum4_texels)
	const int bits = float_as_int(acc);
	const float a0 = int_as_float(__builtin_amdgcn_readlane(bits, 0)), a1 = int_as_float(__builtin_amdgcn_readlane(bits, 1));
	const float a2 = int_as_float(__builtin_amdgcn_readlane(bits, 2)), a3 = int_as_float(__builtin_amdgcn_readlane(bits, 3));
	return (a0 + a2) + (a1 + a3);
#else
	return sum4(v, n);
#endif
}

} } // namespace astcd::ASTC_VARIANT
