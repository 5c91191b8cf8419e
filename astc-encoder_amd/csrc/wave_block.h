// SPDX-License-Identifier: Apache-2.0
// The per-block search controller and the two trial drivers.
//   ref: compress_block                                  Source/astcenc_compress_symbolic.cpp:1162-1456
//        compress_symbolic_block_for_partition_1plane    :353-702
//        compress_symbolic_block_for_partition_2planes   :715-1037
//        prepare_block_statistics                        :1047-1159
// All control flow here is wave-uniform: every decision is taken on values read back from LDS.
#pragma once
#if !defined(__HIP_DEVICE_COMPILE__)
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#endif
#include "wave_ctx.h"
#include "wave_quad.h"
#include "wave_load.h"
#include "wave_ideal.h"
#include "wave_weights.h"
#include "wave_format.h"
#include "wave_color.h"
#include "wave_refine.h"
#include "wave_batch.h"
#include "wave_partition.h"
#include "wave_pack.h"

namespace astcd { inline namespace ASTC_VARIANT {

WV_FN void copy_scb(Scb& dst, const Scb& src)
{
	const uint32_t* s = reinterpret_cast<const uint32_t*>(&src);
	uint32_t* d = reinterpret_cast<uint32_t*>(&dst);
	WV_FOR(i, (int)(sizeof(Scb) / 4)) { d[i] = s[i]; }
}

/* Low/high weight bounds of block mode `bm` for `plane`, with the high-bound override
 * (ref: compress_symbolic.cpp:459-462, :819-827; weight_align.cpp:404-422). */
WV_FN void mode_weight_bounds(const Ctx& c, const BlockMode& bm, int plane, float& low, float& high)
{
	if (bm.quant_mode <= MAX_ANGULAR_QUANT)
	{
		// the pair of quant level q is stored at the rank of q among the levels this grid's block modes use
		const DecimationMode& m = c.dec_mode(bm.decimation_mode);
		const uint32_t used = bm.is_dual_plane ? m.refprec_2planes : m.refprec_1plane;
		const int rank = popcount32(used & ((1u << bm.quant_mode) - 1u));
		const float* lh = c.lowhigh(plane, bm.decimation_mode, bm.is_dual_plane != 0);
		low = lh[rank * 2];
		high = lh[rank * 2 + 1];
	}
	else
	{
		low = 0.0f;
		high = 1.0f;
	}
	if (high > 1.02f * c.tr().min_wt_cutoff[plane])
	{
		high = 1.0f;
	}
}

/* Quantize the ideal weights of `plane` for block mode i into dst_u8 (and float copy dst_f). */
WV_FN void quantize_mode_weights(const Ctx& c, const BlockMode& bm, int plane, float* dst_f, uint8_t* dst_u8)
{
	const DecimationInfo& di = c.dec_info(bm.decimation_mode);
	float low, high;
	mode_weight_bounds(c, bm, plane, low, high);
	QuantParams qp = quant_params(low, high, bm.quant_mode);
	const uint8_t* q2u = c.qxfer(bm.quant_mode).quant_to_unquant;
	const float* ideal = c.dwi(bm.decimation_mode, plane, bm.is_dual_plane != 0);
	WV_FOR64(i, di.weight_count)
	{
		float f;
		int w = quantize_weight(qp, q2u, ideal[i], &f);
		if (dst_f) dst_f[i] = f;
		if (dst_u8) dst_u8[i] = (uint8_t)w;
	}
}

/* Per-mode records of one scoring chunk, built once by one lane per (mode, plane) so that the
 * (mode, texel) sweep reads them with a single LDS access instead of chasing
 * block mode -> decimation info -> table pointers through global memory in every lane. */
struct ModeHdr { uint32_t tw_off, tcf_off; int16_t taps, weights; int32_t mode; };   // mode = packed block mode index; weights per plane
struct ModeQ { float scale, scaled_low_bound, quant_level_m1, rscale, low_bound; int32_t steps_m1; uint32_t q2u_off; uint32_t dwi_off; };
static_assert(sizeof(ModeHdr) + 2 * sizeof(ModeQ) == MODE_DESC_BYTES, "mode descriptor size");

WV_FN float quantize_weight_q(const ModeQ& q, const uint8_t* tab, float ideal)
{
	// table reads as uniform base + 32-bit lane offset (one address register, no 64-bit pointer math)
	const uint32_t q2u = q.q2u_off;
	// same arithmetic as quantize_weight() (ref: compute_quantized_weights_for_decimation :974)
	float ix = ideal * q.scale - q.scaled_low_bound;
	ix = v_clampzo(ix);
	float ix1 = ix * q.quant_level_m1;
	int weightl = (int)ix1;
	int weighth = i_min(weightl + 1, q.steps_m1);
	float ixl = (float)(int)tab[q2u + (uint32_t)weightl];
	float ixh = (float)(int)tab[q2u + (uint32_t)weighth];
	bool mask = (ixl + ixh) < (128.0f * ix);
	ixl = mask ? ixh : ixl;
	return ixl * q.rscale + q.low_bound;
}

/* Quantize-and-score every block mode in [start, end) (ref: compress_symbolic.cpp:438-485 / :806-860 with
 * compute_quantized_weights_for_decimation + compute_error_of_weight_set_{1plane,2planes}).
 *
 * The reference walks the modes one by one.  Here a chunk of up to sixteen modes is scored at once: one lane per
 * mode prepares the mode's quantization parameters; one lane per (mode, weight) quantizes the grid weights; one
 * quad per mode runs the reference's four interleaved accumulators, each lane forming its texels' error terms itself
 * (the infill of the <= 4 quantized weights the texel interpolates) in texel order. */
template <bool DUAL>
WV_FN void score_block_modes(const Ctx& c, int partition_count, int start, int end, int max_weight_quant)
{
	// (one or two weight planes: a template parameter -- as a run-time flag the second plane was a wave-uniform branch in
	//  every trip of the three passes' inner loops)
	constexpr bool dual = DUAL;
	ModeRec* modes = c.modes(start);
	const int T = c.T;
	(void)T;        // (the fixed-context builds use the literal)
	constexpr int planes = DUAL ? 2 : 1;
	const int chunk_modes = (int)c.L->mode_chunk;
	ModeHdr* hdr = reinterpret_cast<ModeHdr*>(c.lds + c.L->uni);
	ModeQ* mq = reinterpret_cast<ModeQ*>(c.lds + c.L->uni + (uint32_t)chunk_modes * sizeof(ModeHdr));
	uint8_t* uqw = c.lds + c.L->uni + (uint32_t)chunk_modes * MODE_DESC_BYTES;   // [mode][MODE_WEIGHT_BYTES]
	const int wcap = (int)c.L->mode_wcap[dual ? 1 : 0];
	const uint32_t wcap_inv = c.L->mode_wcap_inv[dual ? 1 : 0];
	const float* ldsf = reinterpret_cast<const float*>(c.lds);
	const float* eiw0 = c.ei_w(0); const float* eiwes0 = c.ei_wes(0);
	const float* eiw1 = c.ei_w(1); const float* eiwes1 = c.ei_wes(1);

	// The unquantized value of every weight quant level's steps (QuantXfer::quant_to_unquant, 12 x 32 bytes) goes to the
	// tail of the scratch region once per call: pass A looks two of them up per weight, and from the blob every chunk
	// would wait for one more round trip to L2.
	const uint32_t q2u_lds = c.L->uni + c.L->uni_bytes - MODE_Q2U_BYTES;
	WV_FOR(k, (int)(MODE_Q2U_BYTES / 4))
	{
		const uint32_t level = (uint32_t)k >> 3, w = (uint32_t)k & 7u;
		reinterpret_cast<uint32_t*>(c.lds + q2u_lds)[k] = table_at_byte<uint32_t>(c.tab, c.root->off_quant_xfer + level * (uint32_t)sizeof(QuantXfer) + 4u * w);
	}
	// (visible to pass A after the descriptor pass's hand-off)

	// Windows of 64 block modes.  Only the modes that are legal under this trial's weight quant limit
	// (less than half of them, typically, once trial A has set the limit) are scored: their window
	// positions are compacted through a validity bit mask into chunks of descriptor slots.
	for (int base = start; base < end; base += 64)
	{
	const int nwin = i_min(64, end - base);
	const ModeStatic* mstat = reinterpret_cast<const ModeStatic*>(c.table(c.root->off_mode_static));
	const int free_bits = dual ? 109 : partition_count == 1 ? 111 : partition_count == 2 ? 97 : partition_count == 3 ? 94 : 91;   // (ref: mode_bitcount)
	unsigned long long vmask = 0;
#if WV_DEVICE
	// the record of window mode `lane`: read once, kept in registers (as scalars: a struct indexed by the plane would live
	// in scratch memory) for every chunk's descriptor pass
	uint32_t my_tw_off = 0, my_tcf_off = 0, my_dwi = 0, my_lh = 0, my_misc = 0;
	{
		bool ok = false;
		if (WV_LANE < nwin)
		{
			const ModeStatic ms = table_at(mstat, (uint32_t)(base + WV_LANE));
			ok = ms.quant_mode <= max_weight_quant && (dual || free_bits - (int)ms.weight_bits > 0);
			my_tw_off = ms.tw_off; my_tcf_off = ms.tcf_off;
			my_dwi = (uint32_t)ms.dwi_off[0] | ((uint32_t)ms.dwi_off[1] << 16);
			my_lh = (uint32_t)ms.lh_off[0] | ((uint32_t)ms.lh_off[1] << 16);
			my_misc = (uint32_t)ms.taps | ((uint32_t)ms.weights << 8) | ((uint32_t)ms.quant_mode << 16);
		}
		vmask = __ballot(ok);
	}
#else
	for (int i = 0; i < nwin; i++)
	{
		const ModeStatic ms = table_at(mstat, (uint32_t)(base + i));
		if (ms.quant_mode <= max_weight_quant && (dual || free_bits - (int)ms.weight_bits > 0)) vmask |= 1ull << i;
	}
#endif
	const int nvalid = popcount64(vmask);
	WV_FOR64(i, nwin) { if (!((vmask >> i) & 1ull)) modes[base + i].error = 1e38f; }

	for (int first = 0; first < nvalid; first += chunk_modes)
	{
		const int nm = i_min(chunk_modes, nvalid - first);

		{ PROF_SCOPE(c, PS_MODE3);
		WV_FOR64(i, nwin)
		{
			if (!((vmask >> i) & 1ull)) continue;
			const int m = popcount64(vmask & ((1ull << i) - 1ull)) - first;       // descriptor slot of window mode i
			if (m < 0 || m >= nm) continue;
#if !WV_DEVICE
			const ModeStatic ms = table_at(mstat, (uint32_t)(base + i));
			const uint32_t my_tw_off = ms.tw_off, my_tcf_off = ms.tcf_off, my_dwi = (uint32_t)ms.dwi_off[0] | ((uint32_t)ms.dwi_off[1] << 16);
			const uint32_t my_lh = (uint32_t)ms.lh_off[0] | ((uint32_t)ms.lh_off[1] << 16);
			const uint32_t my_misc = (uint32_t)ms.taps | ((uint32_t)ms.weights << 8) | ((uint32_t)ms.quant_mode << 16);
#endif
			const int quant_mode = (int)(my_misc >> 16);
			ModeHdr h;
			h.tw_off = my_tw_off;
			h.tcf_off = my_tcf_off;
			h.taps = (int16_t)(my_misc & 0xFFu);
			h.weights = (int16_t)((my_misc >> 8) & 0xFFu);
			h.mode = base + i;
			hdr[m] = h;
			for (int plane = 0; plane < planes; plane++)
			{
				// the mode's weight range (ref: compress_symbolic.cpp:459-462, :819-827; the same as mode_weight_bounds())
				const uint32_t lh_off = plane ? my_lh >> 16 : my_lh & 0xFFFFu;
				const uint32_t dwi_off = plane ? my_dwi >> 16 : my_dwi & 0xFFFFu;
				float low = 0.0f, high = 1.0f;
				if (lh_off != 0xFFFF)
				{
					const float* lh = reinterpret_cast<const float*>(c.lds + c.L->lowhigh) + lh_off;
					low = lh[0];
					high = lh[1];
				}
				if (high > 1.02f * c.tr().min_wt_cutoff[plane]) high = 1.0f;
				QuantParams qp = quant_params(low, high, quant_mode);
				ModeQ q;
				q.scale = qp.scale; q.scaled_low_bound = qp.scaled_low_bound; q.quant_level_m1 = qp.quant_level_m1;
				q.rscale = qp.rscale; q.low_bound = qp.low_bound; q.steps_m1 = qp.steps_m1;
				q.q2u_off = q2u_lds + (uint32_t)quant_mode * 32u;                 // (LDS byte offset of the level's 32 entries)
				q.dwi_off = (c.L->dwi >> 2) + dwi_off;
				mq[m * 2 + plane] = q;
			}
		}
		WV_SYNC(); }

		// Pass A: quantize every grid weight of every mode of the chunk once (a texel interpolates up to four grid
		// weights and a weight feeds many texels, so quantizing per (mode, texel, tap) would repeat each weight's
		// quantization several times over); the unquantized levels 0..64 go to LDS as bytes.
		{ PROF_SCOPE(c, PS_MODE1);
		WV_FOR(k, nm * wcap)
		{
			const int m = (int)(((uint32_t)k * wcap_inv) >> 16), i = k - m * wcap;
			const ModeHdr h = hdr[m];
			if (i >= h.weights) continue;
			for (int plane = 0; plane < planes; plane++)
			{
				const ModeQ q = mq[m * 2 + plane];
				const float ideal = ldsf[q.dwi_off + (uint32_t)i];
				// (ref: compute_quantized_weights_for_decimation :974; same arithmetic as quantize_weight())
				float ix = ideal * q.scale - q.scaled_low_bound;
				ix = v_clampzo(ix);
				const float ix1 = ix * q.quant_level_m1;
				const int weightl = (int)ix1;
				const int weighth = i_min(weightl + 1, q.steps_m1);
				const int ixli = c.lds[q.q2u_off + (uint32_t)weightl];
				const int ixhi = c.lds[q.q2u_off + (uint32_t)weighth];
				const bool up = ((float)ixli + (float)ixhi) < (128.0f * ix);
				uqw[m * (int)MODE_WEIGHT_BYTES + plane * PLANE2_OFFSET + i] = (uint8_t)(up ? ixhi : ixli);
			}
		}
		WV_SYNC(); }

		// Pass B: one QUAD per mode, lane l of it = the reference's accumulator l: the lane infills the quantized weights at
		// its texels l, l + 4, ... , takes the squared difference to the texel's ideal weight and adds the terms up in that
		// order (ref: compute_error_of_weight_set_{1plane,2planes} :688-842, four interleaved sums folded (a0 + a2) + (a1 + a3)).
		// (Until round 4 one lane per (mode, texel) wrote the terms to a row per mode and a second pass summed them: the rows
		//  were most of a mode's LDS, i.e. set the modes per chunk -- 10 for 6x6, 7 for 8x8, now 16 -- and every chunk pays
		//  the descriptor pass and four hand-offs.)
		{ PROF_SCOPE(c, PS_MODE1);
		WV_QUADS16(m, nm)
		{
			const ModeHdr h = hdr[m];
			const uint8_t* uq = uqw + m * (int)MODE_WEIGHT_BYTES;
			const ModeQ q0 = mq[m * 2], q1 = mq[m * 2 + 1];
			qf acc = q_splat(0.0f);
			Q_LANES(l)
			{
				float sum = 0.0f;
				// (the texel's record of the mode's grid: one 32-bit and one 128-bit load, whatever the tap count.  Every
				//  grid goes through the four-tap form: a tap a grid does not have has index 0 and factor 0.0f in the tables
				//  (ref: init_decimation_info_2d fills the unused entries with zeros), and (v0 + v1) + (0 + 0) is v0 + v1)
#if ASTC_FIXED
				auto texel_term = [&](const TexelTaps& taps, int t) -> float
				{
					const int i0 = (int)(taps.idx & 0xFFu), i1 = (int)((taps.idx >> 8) & 0xFFu), i2 = (int)((taps.idx >> 16) & 0xFFu), i3 = (int)(taps.idx >> 24);
					const float c0 = taps.c0, c1 = taps.c1, c2 = taps.c2, c3 = taps.c3;
					float term = 0.0f;
					for (int plane = 0; plane < planes; plane++)
					{
						const ModeQ& q = plane ? q1 : q0;
						const uint8_t* u = uq + plane * PLANE2_OFFSET;
						const float v0 = ((float)(int)u[i0] * q.rscale + q.low_bound) * c0;
						const float v1 = ((float)(int)u[i1] * q.rscale + q.low_bound) * c1;
						const float v2 = ((float)(int)u[i2] * q.rscale + q.low_bound) * c2;
						const float v3 = ((float)(int)u[i3] * q.rscale + q.low_bound) * c3;
						float current = (v0 + v1) + (v2 + v3);
						float diff = current - (plane ? eiw1[t] : eiw0[t]);
						float e = diff * diff * (plane ? eiwes1[t] : eiwes0[t]);
						term = plane ? term + e : e;
					}
					return term;
				};
				// A fixed-context build knows the texel count: every accumulator lane makes the same ceil(T / 4) trips (the last
				// one masked when T is not a multiple of four), a few of them unrolled together, with the next trip's record
				// requested before this trip's arithmetic -- the records come from L2, and the `t < T` loop of the generic build
				// waits out one such round trip per texel row, one after the other (8x8 -thorough +2.6 %, profiles/r05f).
				constexpr int kTrips = ((int)kFixedRoot.texel_count + 3) >> 2;
				constexpr int kLast = (int)kFixedRoot.texel_count - 1;
				TexelTaps next = texel_taps_at(c.tab, h.tw_off, h.tcf_off, (uint32_t)i_min(l, kLast));
				constexpr int kUnroll = DUAL ? (kTrips <= 9 ? 2 : 1) : 3;      // (two planes: twice the values in flight per trip)
				#pragma unroll kUnroll
				for (int trip = 0; trip < kTrips; trip++)
				{
					const int t = l + 4 * trip;
					const TexelTaps taps = next;
					if (trip + 1 < kTrips) next = texel_taps_at(c.tab, h.tw_off, h.tcf_off, (uint32_t)i_min(t + 4, kLast));
					if ((kLast & 3) != 3 && t > kLast) continue;
					sum += texel_term(taps, t);
				}
#else
				for (int t = l; t < T; t += 4)
				{
					const TexelTaps taps = texel_taps_at(c.tab, h.tw_off, h.tcf_off, (uint32_t)t);
					const int i0 = (int)(taps.idx & 0xFFu), i1 = (int)((taps.idx >> 8) & 0xFFu), i2 = (int)((taps.idx >> 16) & 0xFFu), i3 = (int)(taps.idx >> 24);
					const float c0 = taps.c0, c1 = taps.c1, c2 = taps.c2, c3 = taps.c3;
					float term = 0.0f;
					for (int plane = 0; plane < planes; plane++)
					{
						const ModeQ& q = plane ? q1 : q0;
						const uint8_t* u = uq + plane * PLANE2_OFFSET;
						const float v0 = ((float)(int)u[i0] * q.rscale + q.low_bound) * c0;
						const float v1 = ((float)(int)u[i1] * q.rscale + q.low_bound) * c1;
						const float v2 = ((float)(int)u[i2] * q.rscale + q.low_bound) * c2;
						const float v3 = ((float)(int)u[i3] * q.rscale + q.low_bound) * c3;
						float current = (v0 + v1) + (v2 + v3);
						float diff = current - (plane ? eiw1[t] : eiw0[t]);
						float e = diff * diff * (plane ? eiwes1[t] : eiwes0[t]);
						term = plane ? term + e : e;
					}
					sum += term;
				}
#endif
				QV(acc, l) = sum;
			}
			const float error = q_hadd(acc);
			Q_ONCE { modes[h.mode].error = error; }
		}
		WV_SYNC(); }
	}
	}
}

// Out-of-line pieces of the refinement loop (see the note on stage functions below): each one
// rebuilds the views it needs from LDS.  refine_candidates() itself is scalar control flow around
// calls: it keeps no vector value alive across a call, so it has no callee-saved registers to spill.

/* The quantized weights of the chosen candidates (the reference keeps them for every block mode,
 * compress_symbolic.cpp:469-478); after this the search-phase LDS (ideal weights, angular bounds, mode
 * records) is dead and the refine-phase tables reuse it. */
WV_OUT void refine_quantize_candidates(bool dual, int partition_count, int partition_packed)
{
	const Ctx c = ctx_make();
	dual = wv_uniform(dual); partition_count = wv_uniform(partition_count); partition_packed = wv_uniform(partition_packed);
	// per-trial constants of the refinement steps
	if (partition_count == 1) trial_scale_directions(c, part_view_lds(c, 1, 0), dual);
	else trial_scale_directions(c, part_view_lds(c, partition_count, partition_packed), false);
	TrialInfo& tr = c.tr();
	const int candidate_count = wv_uniform(tr.cand_count);
	// (the search phase of this trial has used the bytes the staged colour rows live in)
	WV_ONE { tr.staged_color_quant[0] = -1; }
	PROF_SCOPE(c, PS_Y0);
	// quantization parameters of each (candidate, plane) once, then one lane per (candidate, plane, weight)
	const int plane_shift = dual ? 1 : 0;
	ModeQ* cq = reinterpret_cast<ModeQ*>(c.lds + c.L->uni);          // [candidate][plane]; the scoring scratch is idle now
	const float* ldsf = reinterpret_cast<const float*>(c.lds);
	WV_FOR64(k, candidate_count << plane_shift)
	{
		const int ci = k >> plane_shift, plane = k & plane_shift;
		const BlockMode& bm = c.block_mode(tr.cand_block_mode[ci]);
		float low, high;
		mode_weight_bounds(c, bm, plane, low, high);
		QuantParams qp = quant_params(low, high, bm.quant_mode);
		ModeQ q;
		q.scale = qp.scale; q.scaled_low_bound = qp.scaled_low_bound; q.quant_level_m1 = qp.quant_level_m1;
		q.rscale = (float)c.dec_info(bm.decimation_mode).weight_count;   // (field reused: weight count of the grid)
		q.low_bound = 0.0f; q.steps_m1 = qp.steps_m1;
		q.q2u_off = c.root->off_quant_xfer + (uint32_t)bm.quant_mode * (uint32_t)sizeof(QuantXfer);
		q.dwi_off = (uint32_t)(c.dwi(bm.decimation_mode, plane, dual) - ldsf);
		cq[k] = q;
	}
	WV_SYNC();
	WV_FOR(k, candidate_count << (plane_shift + 6))
	{
		const int cp = k >> 6, i = k & 63;
		const ModeQ q = cq[cp];
		if (i >= (int)q.rscale) continue;
		// (ref: compute_quantized_weights_for_decimation :974, integer output)
		const float ideal = (ldsf + q.dwi_off)[i];
		float ix = ideal * q.scale - q.scaled_low_bound;
		ix = v_clampzo(ix);
		const float ix1 = ix * q.quant_level_m1;
		const int weightl = (int)ix1;
		const int weighth = i_min(weightl + 1, q.steps_m1);
		const int ixli = c.tab[q.q2u_off + (uint32_t)weightl];
		const int ixhi = c.tab[q.q2u_off + (uint32_t)weighth];
		const bool mask = ((float)ixli + (float)ixhi) < (128.0f * ix);
		c.candw(cp >> plane_shift)[(cp & plane_shift) * PLANE2_OFFSET + i] = (uint8_t)(mask ? ixhi : ixli);
	}
	WV_SYNC();
}

/* A candidate that passed the test on the error of its (batched) first step takes its turn: what the step left for
 * it (wave_batch.h: CandState) becomes the working block, endpoints and decoded endpoints, and the candidate's tables
 * are staged for the steps that follow -- the grid's tables and the quant transfer table only when they differ from
 * what is staged (ref: :497-540). */
WV_OUT void refine_candidate_restore(bool dual, int partition_count, int partition_packed, int plane2_component, int candidate, int slot,
                                     int stage_dm, int stage_wq, int color_quant_level, int block_mode_packed)
{
	const Ctx c = ctx_make();
	dual = wv_uniform(dual); partition_count = wv_uniform(partition_count); partition_packed = wv_uniform(partition_packed);
	plane2_component = wv_uniform(plane2_component); candidate = wv_uniform(candidate); slot = wv_uniform(slot);
	stage_dm = wv_uniform(stage_dm); stage_wq = wv_uniform(stage_wq); color_quant_level = wv_uniform(color_quant_level);
	block_mode_packed = wv_uniform(block_mode_packed);
	TrialInfo& tr = c.tr();
	Scb& workscb = c.wscb();
	{
		// (first: the record of a batch's first candidate lies in the scratch that the staged tables are about to overwrite)
		const BatchView bv = batch_view(c, dual, partition_count);
		const CandState st = batch_state(c, bv, slot);
		WV_FOR64(k, partition_count * 4)
		{
			(&tr.wep0[0][0])[k] = st.wep0[k];
			(&tr.wep1[0][0])[k] = st.wep1[k];
		}
		WV_FOR64(k, partition_count * 8) { workscb.color_values[k >> 3][k & 7] = st.colors[k]; }
		WV_FOR64(j, partition_count) { workscb.color_formats[j] = st.formats[j]; }
		{
			const uint32_t* src = reinterpret_cast<const uint32_t*>(c.candw(candidate));
			uint32_t* dst = reinterpret_cast<uint32_t*>(workscb.weights);
			WV_FOR(k, 16) { dst[k] = src[k]; }
		}
		const int formats_matched = wv_uniform((int)st.meta[1]);
		const int quant_mode = wv_uniform((int)st.meta[0]);
		const bool rgbm_error = wv_uniform((int)st.meta[2]) != 0;
		WV_ONE
		{
			// the header of the working block (as refine_pack writes it); an RGBM block whose M decodes to zero is an
			// error block from here on (ref: :612-616)
			uint32_t* head = reinterpret_cast<uint32_t*>(&workscb);
			head[0] = (uint32_t)(rgbm_error ? SYM_BTYPE_ERROR : SYM_BTYPE_NONCONST) | ((uint32_t)partition_count << 8) |
			          ((uint32_t)formats_matched << 16) | ((uint32_t)((dual ? plane2_component : -1) & 0xFF) << 24);
			head[1] = (uint32_t)block_mode_packed | ((uint32_t)partition_packed << 16);
			workscb.quant_mode = (uint8_t)quant_mode;
		}
		WV_SYNC();
	}
	if (stage_dm >= 0)
	{
		const DecimationInfo& dinfo = c.dec_info(stage_dm);
		stage_quads_nosync(c.lds + c.L->dtab, reinterpret_cast<const uint8_t*>(&dinfo), (int)sizeof(DecimationInfo));
		stage_quads_nosync(c.lds + c.L->dtab + DTAB_RECORD_BYTES, c.table(dinfo.off_texel_weights), (int)dinfo.table_bytes);
	}
	if (stage_wq >= 0)
	{
		stage_words_nosync(c.lds + c.L->qtab, reinterpret_cast<const uint8_t*>(&c.qxfer(stage_wq)), (int)(sizeof(QuantXfer) / 4));
	}
	stage_color_rows(c, color_quant_level);      // ends with a sync
	// the decoded endpoints of the packed values, where the scoring and realignment steps expect them
	WV_FOR64(p, partition_count)
	{
		i4 e0, e1;
		unpack_color_endpoints(c.cfg->profile, workscb.color_formats[p], workscb.color_values[p], e0, e1);
		int* o = &tr.ibox[p * 8];
		o[0] = e0.x; o[1] = e0.y; o[2] = e0.z; o[3] = e0.w;
		o[4] = e1.x; o[5] = e1.y; o[6] = e1.z; o[7] = e1.w;
	}
	WV_SYNC();
}

/* One refinement step's front half, part 1: least-squares endpoints for the current weights
 * (ref: :542-555, :925-931).  Three out-of-line variants -- the single-partition case is by far the most
 * frequent and should not carry the register needs of the other two. */
WV_OUT void refine_recompute_1partition(int decimation_mode)
{
	const Ctx c = ctx_make();
	decimation_mode = wv_uniform(decimation_mode);
	PROF_SCOPE(c, PS_RECOMPUTE);
	recompute_ideal_colors_1plane(c, part_view_lds(c, 1, 0), dec_view_lds(c, decimation_mode));
}
WV_OUT void refine_recompute_partitions(int partition_count, int partition_packed, int decimation_mode)
{
	const Ctx c = ctx_make();
	partition_count = wv_uniform(partition_count); partition_packed = wv_uniform(partition_packed); decimation_mode = wv_uniform(decimation_mode);
	PROF_SCOPE(c, PS_RECOMPUTE);
	recompute_ideal_colors_1plane(c, part_view_lds(c, partition_count, partition_packed), dec_view_lds(c, decimation_mode));
}
WV_OUT void refine_recompute_2planes(int decimation_mode, int plane2_component)
{
	const Ctx c = ctx_make();
	decimation_mode = wv_uniform(decimation_mode); plane2_component = wv_uniform(plane2_component);
	PROF_SCOPE(c, PS_RECOMPUTE);
	recompute_ideal_colors_2planes(c, dec_view_lds(c, decimation_mode), plane2_component);
}

#if ASTC_ENABLE_HDR
/* The HDR endpoint formats of the candidate's partitions (sub-modes side by side, wave_color_hdr.h), out of line: the
 * LDR-profile kernel has none of it, and the HDR kernel's refinement loop should not carry its registers. */
WV_OUT void refine_pack_hdr(int partition_count, int candidate, int to_scratch, int quant_level)
{
	const Ctx c = ctx_make();
	partition_count = wv_uniform(partition_count); candidate = wv_uniform(candidate);
	to_scratch = wv_uniform(to_scratch); quant_level = wv_uniform(quant_level);
	TrialInfo& tr = c.tr();
	Scb& workscb = c.wscb();
	uint8_t* colorvals = reinterpret_cast<uint8_t*>(&tr.ibox[32]);    // (the retry buffers of refine_pack)
	uint8_t* fmts = colorvals + 32;
	uint8_t* have_decoded = fmts + 4;
	uint8_t* tries = reinterpret_cast<uint8_t*>(tr.fbox);             // (the re-fit's sums are consumed by now)
	static_assert(sizeof(tr.fbox) >= 4 * HDR_TRY_LANES * HDR_TRY_BYTES, "sub-mode records do not fit the mailbox");
	WV_FOR64(j, partition_count) { if (!to_scratch && endpoint_format_is_hdr(tr.cand_formats[candidate][j])) have_decoded[j] = 0; }
	pack_endpoints_hdr(color_tabs(c, quant_level), &tr.wep0[0][0], &tr.wep1[0][0], &tr.rgbo[0][0], partition_count, tr.cand_formats[candidate],
	                   to_scratch ? colorvals : &workscb.color_values[0][0], to_scratch ? fmts : workscb.color_formats, tries);
}
#endif

/* Part 2: pack the endpoints (one lane per partition), retry at the higher quant level that matched
 * formats allow (ref: :561-598), and fill in the header of the working block. */
__attribute__((always_inline)) WV_FN void refine_pack(bool dual, int partition_count, int partition_packed, int plane2_component,
                        int candidate, int quant_level, int quant_level_mod, int block_mode_packed)
{
	WV_LANE_SCOPE;
	const Ctx c = ctx_make();
	dual = wv_uniform(dual); partition_count = wv_uniform(partition_count); partition_packed = wv_uniform(partition_packed);
	plane2_component = wv_uniform(plane2_component);
	candidate = wv_uniform(candidate); quant_level = wv_uniform(quant_level); quant_level_mod = wv_uniform(quant_level_mod);
	block_mode_packed = wv_uniform(block_mode_packed);
	TrialInfo& tr = c.tr();
	Scb& workscb = c.wscb();

	uint8_t* colorvals = reinterpret_cast<uint8_t*>(&tr.ibox[32]);   // [4][8] scratch copy for the matched-format retry
	uint8_t* fmts = colorvals + 32;                                   // [4]
	uint8_t* have_decoded = fmts + 4;                                 // [4] the pack left partition p's decoded endpoints in tr.ibox
	int formats_matched = 0;
	const int profile = c.cfg->profile;
	{ PROF_SCOPE(c, PS_PACK);
	// pass 0 packs into the working block; pass 1 (only when every partition got the same format and a
	// higher quant level is then possible) packs again at that level into the retry buffer
	for (int pass = 0; pass < 2; pass++)
	{
		const bool to_scratch = pass == 1;
		const int q = to_scratch ? quant_level_mod : quant_level;
		if (to_scratch) stage_color_rows(c, q);        // (rare: the retry level's rows replace the candidate's, put back below)
		// four lanes per partition, one colour channel each (wave_quad.h)
		WV_QUADS16(j, partition_count)
		{
			const int requested = tr.cand_formats[candidate][j];
			if (kHdr && endpoint_format_is_hdr(requested)) continue;
			uint8_t* vals = to_scratch ? colorvals + j * 8 : workscb.color_values[j];
			const QPacked r = pack_endpoints_quad(c, q_load(tr.wep0[j]), q_load(tr.wep1[j]), q_load(tr.rgbs[j]), requested, vals, q);
			Q_ONCE
			{
				if (to_scratch) fmts[j] = (uint8_t)r.format;
				else { workscb.color_formats[j] = (uint8_t)r.format; have_decoded[j] = r.decoded_valid ? 1 : 0; }
			}
			if (!to_scratch && r.decoded_valid)
			{
				// the 8-bit endpoints a decoder sees, expanded to the 16 bits the scoring works in
				// (ref: unpack_color_endpoints, color_unquantize.cpp:980-1022: LDR formats in every profile)
				auto widen = [profile](int v) { return profile == 0 ? (v << 8) | 0x80 : v * 257; };
				q_store_i32(&tr.ibox[j * 8], q_mapi(r.decoded.e0, widen));
				q_store_i32(&tr.ibox[j * 8 + 4], q_mapi(r.decoded.e1, widen));
			}
		}
#if ASTC_ENABLE_HDR
		refine_pack_hdr(partition_count, candidate, to_scratch ? 1 : 0, q);
#endif
		WV_SYNC();
		if (pass == 1) { stage_color_rows(c, quant_level); break; }
		if (dual || partition_count < 2 || quant_level == quant_level_mod) break;
		bool all_same = true;
		for (int j = 1; j < partition_count; j++) all_same = all_same && workscb.color_formats[j] == workscb.color_formats[0];
		if (!wv_uniform(all_same)) break;
	}
	}

	PROF_SCOPE(c, PS_Y2);
	if (!dual && partition_count >= 2 && quant_level != quant_level_mod)
	{
		bool all_same = true;
		for (int j = 1; j < partition_count; j++) all_same = all_same && workscb.color_formats[j] == workscb.color_formats[0];
		if (wv_uniform(all_same))
		{
			bool all_same_mod = true;
			for (int j = 1; j < partition_count; j++) all_same_mod = all_same_mod && fmts[j] == fmts[0];
			if (wv_uniform(all_same_mod))
			{
				formats_matched = 1;
				WV_FOR64(k, partition_count * 8) { workscb.color_values[k >> 3][k & 7] = colorvals[k]; }
				WV_FOR64(j, partition_count) { workscb.color_formats[j] = fmts[j]; have_decoded[j] = 0; }
			}
			WV_SYNC();
		}
	}
	// The decoded endpoints of what was just packed: the scoring and weight realignment steps that follow (up to three
	// of them before the next packing) all start from these (tr.ibox[p * 8 ..]).  The direct and base + offset formats
	// left them there while packing; the others (and a retry that replaced the values) are decoded here.
	WV_FOR64(p, partition_count)
	{
		if (have_decoded[p]) continue;
		i4 e0, e1;
		unpack_color_endpoints(c.cfg->profile, workscb.color_formats[p], workscb.color_values[p], e0, e1);
		int* o = &tr.ibox[p * 8];
		o[0] = e0.x; o[1] = e0.y; o[2] = e0.z; o[3] = e0.w;
		o[4] = e1.x; o[5] = e1.y; o[6] = e1.z; o[7] = e1.w;
	}
	WV_ONE
	{
		// the header of the working block: its first eight bytes as two words, then the colour quant level
		static_assert(offsetof(Scb, block_type) == 0 && offsetof(Scb, partition_count) == 1 && offsetof(Scb, color_formats_matched) == 2 &&
		              offsetof(Scb, plane2_component) == 3 && offsetof(Scb, block_mode) == 4 && offsetof(Scb, partition_index) == 6, "Scb header layout");
		uint32_t* head = reinterpret_cast<uint32_t*>(&workscb);
		head[0] = (uint32_t)SYM_BTYPE_NONCONST | ((uint32_t)partition_count << 8) | ((uint32_t)formats_matched << 16) | ((uint32_t)(plane2_component & 0xFF) << 24);
		head[1] = (uint32_t)block_mode_packed | ((uint32_t)partition_packed << 16);
		workscb.quant_mode = (uint8_t)(formats_matched ? quant_level_mod : quant_level);
	}
	WV_SYNC();
}

WV_OUT float refine_difference(int partition_count, int partition_packed, int decimation_mode)
{
	const Ctx c = ctx_make();
	partition_count = wv_uniform(partition_count); partition_packed = wv_uniform(partition_packed); decimation_mode = wv_uniform(decimation_mode);
	float errorval;
	{
		PROF_SCOPE(c, PS_DIFF);
		if (partition_count == 1) errorval = compute_symbolic_block_difference(c, part_view_lds(c, 1, 0), dec_view_lds(c, decimation_mode));
		else errorval = compute_symbolic_block_difference(c, part_view_lds(c, partition_count, partition_packed), dec_view_lds(c, decimation_mode));
	}
	errorval = wv_uniform(errorval);
	if (errorval == -ERROR_CALC_DEFAULT)
	{
		// (RGBM blocks whose M channel decodes to zero: ref :612-616)
		errorval = -errorval;
		WV_ONE { c.wscb().block_type = SYM_BTYPE_ERROR; }
		WV_SYNC();
	}
	return errorval;
}

__attribute__((always_inline)) WV_FN bool refine_realign(int partition_count, int partition_packed, int decimation_mode)
{
	const Ctx c = ctx_make();
	partition_count = wv_uniform(partition_count); partition_packed = wv_uniform(partition_packed); decimation_mode = wv_uniform(decimation_mode);
	const QuantXfer& qat = *reinterpret_cast<const QuantXfer*>(c.lds + c.L->qtab);
	PROF_SCOPE(c, PS_REALIGN);
	// (one copy for every partition count: the per-texel partition lookup is a run-time branch in there anyway)
	return realign_weights(c, part_view_lds(c, partition_count, partition_packed), dec_view_lds(c, decimation_mode), qat);
}

/* The working block becomes the best block so far (ref: :633-636, :676-679). */
WV_OUT void refine_accept(float errorval)
{
	const Ctx c = ctx_make();
	errorval = wv_uniform(errorval);
	WV_SYNC();
	WV_ONE { c.wscb().errorval = errorval; }
	WV_SYNC();
	copy_scb(c.scb(), c.wscb());
	WV_SYNC();
}

/* Shared tail of both trials: refine the chosen candidates. Returns best error seen in this trial.
 * (ref: the candidate loops of compress_symbolic_block_for_partition_1plane :497-700 / _2planes :880-1040) */
WV_FN float refine_candidates(const Ctx& c, int partition_count, int partition_packed,
                              int plane2_component, float tune_errorval_threshold)
{
	WV_LANE_SCOPE;
	TrialInfo& tr = c.tr();
	const bool dual = plane2_component >= 0;

	float best_errorval_in_mode = ERROR_CALC_DEFAULT;
	// (values read from LDS look lane-variant to the compiler; wv_uniform() puts them in SGPRs so that the
	//  control flow runs on the scalar unit)
	float best_errorval_in_scb = wv_uniform(c.scb().errorval);
	const int candidate_count = wv_uniform(tr.cand_count);
	const int refinement_limit = (int)c.cfg->tune_refinement_limit;

	DUP_STAGE(c, DUP_CAND_QUANTIZE, refine_quantize_candidates(dual, partition_count, partition_packed));

	// The first step of every candidate -- re-fit, pack, decode and score -- runs for a batch of candidates at once
	// (wave_batch.h); then each candidate of the batch takes its turn in the reference's order: the test on that first
	// error, and for a candidate that passes it the rest of its refinement, one candidate at a time.
	const int batch_max = wv_uniform((int)c.L->bat_max[dual ? 1 : 0]);
	bool stop_all = false;
	for (int first = 0; first < candidate_count && !stop_all; first += batch_max)
	{
		const int batch = i_min(batch_max, candidate_count - first);
		batch_refit(dual, partition_count, partition_packed, plane2_component, first, batch);
		DUP_STAGE(c, DUP_BATCH_PACK, batch_pack(dual, partition_count, first, batch));
		DUP_STAGE(c, DUP_BATCH_SCORE, batch_score(dual, partition_count, partition_packed, plane2_component, batch));
		// (the step's scratch is the region the staged candidate tables live in)
		int staged_dm = -1, staged_wq = -1;
		WV_ONE { tr.staged_color_quant[0] = -1; }
		WV_SYNC();

		for (int slot = 0; slot < batch; slot++)
		{
			const int i = first + slot;
			const int bm_packed_index = wv_uniform(tr.cand_block_mode[i]);
			const BlockMode& qw_bm = c.block_mode(bm_packed_index);
			const int color_quant_level = wv_uniform((int)tr.cand_quant[i]);
			const int color_quant_level_mod = wv_uniform((int)tr.cand_quant_mod[i]);
			const int cand_dm = wv_uniform((int)qw_bm.decimation_mode);
			const int cand_wq = wv_uniform((int)qw_bm.quant_mode);

			TRACE_PUT(c, TR_CANDIDATE, (float)cand_wq);
			{
				// the first step's error (ref: :600-640)
				const BatchView bv = batch_view(c, dual, partition_count);
				const float errorval = wv_uniform(*batch_state(c, bv, slot).errorval);
				TRACE_PUT(c, TR_ERR_PRE, errorval);
				best_errorval_in_mode = wv_uniform(f_min(errorval, best_errorval_in_mode));

				const float threshold = (0.045f * (float)refinement_limit) + 1.08f;
				if (errorval > (threshold * best_errorval_in_scb)) continue;

				DUP_STAGE(c, DUP_CAND_SETUP, refine_candidate_restore(dual, partition_count, partition_packed, plane2_component, i, slot,
				                       cand_dm != staged_dm ? cand_dm : -1, cand_wq != staged_wq ? cand_wq : -1, color_quant_level, bm_packed_index));
				staged_dm = cand_dm;
				staged_wq = cand_wq;

				if (errorval < best_errorval_in_scb)
				{
					best_errorval_in_scb = errorval;
					refine_accept(errorval);
					if (errorval < tune_errorval_threshold)
					{
						stop_all = true;
						break;
					}
				}
			}

			for (int l = 0; l < refinement_limit; l++)
			{
				if (l > 0)
				{
					DUP_STAGE(c, DUP_RECOMPUTE, {
					if (dual) refine_recompute_2planes(cand_dm, plane2_component);
					else if (partition_count == 1) refine_recompute_1partition(cand_dm);
					else refine_recompute_partitions(partition_count, partition_packed, cand_dm); });
					DUP_STAGE(c, DUP_PACK, refine_pack(dual, partition_count, partition_packed, plane2_component, i,
					            color_quant_level, color_quant_level_mod, bm_packed_index));
				}

#if defined(ASTC_DUPSTAGE)
				// (realignment changes the weights: for the doubled run they are put back first)
				uint32_t saved_weights = 0;
				if (DUP_STAGE_ID(c) == (uint32_t)DUP_REALIGN || (dual && DUP_STAGE_ID(c) == (uint32_t)DUP_REALIGN_2PLANES))
				{
					WV_FOR(k, 16) { saved_weights = reinterpret_cast<const uint32_t*>(c.wscb().weights)[k]; }
					(void)refine_realign(partition_count, partition_packed, cand_dm);
					WV_SYNC();
					WV_FOR(k, 16) { reinterpret_cast<uint32_t*>(c.wscb().weights)[k] = saved_weights; }
					WV_SYNC();
				}
#endif
				const bool adjustments = wv_uniform(refine_realign(partition_count, partition_packed, cand_dm));

				float errorval;
				DUP_STAGE(c, DUP_DIFF, errorval = wv_uniform(refine_difference(partition_count, partition_packed, cand_dm)));
				TRACE_PUT(c, TR_ERR_POST, errorval);
				best_errorval_in_mode = wv_uniform(f_min(errorval, best_errorval_in_mode));

				int iters_remaining = refinement_limit - 1 - l;
				float threshold = (0.045f * (float)iters_remaining) + 1.0f;
				if (errorval > (threshold * best_errorval_in_scb))
				{
					break;
				}

				if (errorval < best_errorval_in_scb)
				{
					best_errorval_in_scb = errorval;
					refine_accept(errorval);
					if (errorval < tune_errorval_threshold)
					{
						stop_all = true;
						break;
					}
				}

				if (!adjustments)
				{
					break;
				}
			}
			if (stop_all) break;
		}
	}
	return best_errorval_in_mode;
}

// ---------------------------------------------------------------------------------------------
// Out-of-line stages.  Every stage of a trial is its own (non-inlined) function that rebuilds the
// wave context from LDS (ctx_make) and takes only uniform scalars.  The register allocator then
// works on one stage at a time: the hot inner loops no longer carry (and spill/reload around) the
// live state of the whole search, and each stage exists once in the kernel instead of once per
// call site.
// ---------------------------------------------------------------------------------------------

WV_OUT void stage_ideal(bool dual, int partition_count, int partition_packed, int plane2_component)
{
	const Ctx c = ctx_make();
	dual = wv_uniform(dual); partition_count = wv_uniform(partition_count);
	partition_packed = wv_uniform(partition_packed); plane2_component = wv_uniform(plane2_component);
	PROF_SCOPE(c, PS_IDEAL);
	TrialInfo& tr = c.tr();
	if (partition_count == 1)
	{
		PartView pv = part_view_staged(c, 1, 0);
		if (dual) ideal_colors_and_weights_2planes(c, pv, plane2_component);
		else
		{
			// the second run of trial A (ref: compress_symbolic.cpp:1292-1318: both runs call
			// compute_ideal_colors_and_weights_1plane on the same block and partitioning) finds the first run's result
			// still in place: nothing in between writes plane 0 of ei_w / ei_wes / ep0 / ep1
			if (wv_uniform(tr.ideal_1p1p_valid) != 0) return;
			ideal_colors_and_weights_1plane(c, pv);
		}
		WV_ONE { tr.ideal_1p1p_valid = dual ? 0 : 1; }
		WV_SYNC();
	}
	else
	{
		PartView pv = part_view_staged(c, partition_count, partition_packed);
		ideal_colors_and_weights_1plane(c, pv);
		WV_ONE { tr.ideal_1p1p_valid = 0; }
		WV_SYNC();
	}
}

WV_OUT void stage_decimate(int nplanes, int ref_mask, int max_decimation_modes)
{
	const Ctx c = ctx_make_vector_tables();
	nplanes = wv_uniform(nplanes); ref_mask = wv_uniform(ref_mask); max_decimation_modes = wv_uniform(max_decimation_modes);
	PROF_SCOPE(c, PS_DECIMATE);
	ideal_weights_all_grids(c, nplanes, (uint16_t)ref_mask, max_decimation_modes);
}

/* Weight cut-offs, the list of grids of this trial and their angular bounds
 * (ref: compress_symbolic.cpp:409-418 / :765-785; weight_align.cpp:358-426). */
WV_OUT void stage_angular(bool dual, int partition_count, int plane2_component, int max_decimation_modes, int ref_mask_i, int max_weight_quant)
{
	const Ctx c = ctx_make();
	dual = wv_uniform(dual); partition_count = wv_uniform(partition_count); plane2_component = wv_uniform(plane2_component);
	max_decimation_modes = wv_uniform(max_decimation_modes); max_weight_quant = wv_uniform(max_weight_quant);
	const uint16_t ref_mask = (uint16_t)wv_uniform(ref_mask_i);
	TrialInfo& tr = c.tr();

	// Weight cut-off of each plane (ref: compress_symbolic.cpp:409-418, :765-785): the smallest (1 - e0) / (e1 - e0) above
	// 0.5 over the plane's channels, one channel per lane of a quad (wave_quad.h).  The values compared are finite
	// (a NaN quotient fails both tests), so the minimum is the same in any order.
	WV_QUADS(q, 1)
	{
		(void)q;
		if (!dual)
		{
			qf min_ep = q_splat(10.0f);
			for (int i = 0; i < partition_count; i++)
			{
				const qf ep = (q_splat(1.0f) - q_load(tr.ep0[0][i])) / (q_load(tr.ep1[0][i]) - q_load(tr.ep0[0][i]));
				min_ep = q_zip(ep, min_ep, [](float v, float m) { return (v > 0.5f && v < m) ? v : m; });
			}
			const float cut = q_hmin(min_ep);
			Q_ONCE { tr.min_wt_cutoff[0] = cut; }
		}
		else
		{
			for (int plane = 0; plane < 2; plane++)
			{
				const qf ep = (q_splat(1.0f) - q_load(tr.ep0[plane][0])) / (q_load(tr.ep1[plane][0]) - q_load(tr.ep0[plane][0]));
				qf min_ep = q_map(ep, [](float v) { return (v > 0.5f && v < 10.0f) ? v : 10.0f; });
				// plane 0 ignores the separated component, plane 1 only looks at it
				min_ep = q_map_ch(min_ep, [plane, plane2_component](int k, float v) {
					const bool is_p2 = k == plane2_component;
					return (plane == 0 ? is_p2 : !is_p2) ? ERROR_CALC_DEFAULT : v; });
				const float cut = q_hmin(min_ep);
				Q_ONCE { tr.min_wt_cutoff[plane] = cut; }
			}
		}
	}
	// list of the grids this trial uses, largest weight count first (the order the search batches them in).  Only grids with
	// a block mode at a quant level that HAS angular bounds (<= QUANT_12, ref: weight_align.cpp:404-422) and that this trial
	// may use: the reference also runs the search for grids whose every mode is above QUANT_12 and never reads the result.
	const uint16_t ang_mask = (uint16_t)(ref_mask & ((1u << (MAX_ANGULAR_QUANT + 1)) - 1u));
	const uint8_t* by_weights = c.table(c.root->off_dm_by_weights);
	const int all_dms = (int)c.root->decimation_mode_count_selected;
#if WV_DEVICE
	{
		int n = 0;
		for (int base = 0; base < all_dms; base += 64)
		{
			const int k = base + WV_LANE;
			bool used = false;
			int i = 0;
			if (k < all_dms)
			{
				i = by_weights[k];
				const DecimationMode& m = c.dec_mode(i);
				used = i < max_decimation_modes && ((dual ? m.refprec_2planes : m.refprec_1plane) & ang_mask) != 0;
			}
			const unsigned long long mask = __ballot(used);
			if (used) tr.dm_list[n + __popcll(mask & ((1ull << WV_LANE) - 1ull))] = (uint8_t)i;
			n += __popcll(mask);
		}
		WV_ONE { tr.dm_count = n; }
	}
#else
	{
		int n = 0;
		for (int k = 0; k < all_dms; k++)
		{
			const int i = by_weights[k];
			const DecimationMode& m = c.dec_mode(i);
			if (i < max_decimation_modes && ((dual ? m.refprec_2planes : m.refprec_1plane) & ang_mask)) tr.dm_list[n++] = (uint8_t)i;
		}
		tr.dm_count = n;
	}
#endif
	WV_SYNC();

	auto get_set = [&](int s) {
		int plane = dual ? (s & 1) : 0;
		int dm = tr.dm_list[dual ? (s >> 1) : s];
		const DecimationMode& m = c.dec_mode(dm);
		int max_precision = dual ? m.maxprec_2planes : m.maxprec_1plane;
		max_precision = i_min(max_precision, MAX_ANGULAR_QUANT);
		max_precision = i_min(max_precision, max_weight_quant);
		AngSet a;
		a.weights = c.dwi(dm, plane, dual);
		a.rows = c.isample() + (a.weights - reinterpret_cast<const float*>(c.lds + c.L->dwi));
		a.out = c.lowhigh(plane, dm, dual);
		a.wcount = c.dec_info(dm).weight_count;
		a.maxq = max_precision;
		a.used = (uint16_t)((dual ? m.refprec_2planes : m.refprec_1plane) & 0xFFu);
		return a;
	};
	PROF_SCOPE(c, PS_ANGULAR);
#if !WV_DEVICE
	if (getenv("ASTC_EMU_DUMP_ANG")) { fprintf(stderr, "ang: dual %d pc %d maxwq %d sets %d:", (int)dual, partition_count, max_weight_quant, tr.dm_count * (dual ? 2 : 1));
		for (int s = 0; s < tr.dm_count * (dual ? 2 : 1); s++) { AngSet a = get_set(s); fprintf(stderr, " (w%d q%d u%x)", a.wcount, a.maxq, a.used); } fprintf(stderr, "\n"); }
#endif
	angular_endpoints(c, tr.dm_count * (dual ? 2 : 1), get_set);
}

/* (one stage function per plane count: each gets its own register allocation -- the two-plane copy's loops keep twice the
 *  values in flight) */
WV_OUT void stage_modes_1plane(int partition_count, int start, int end, int max_weight_quant)
{
	const Ctx c = ctx_make_vector_tables();
	partition_count = wv_uniform(partition_count); start = wv_uniform(start); end = wv_uniform(end);
	max_weight_quant = wv_uniform(max_weight_quant);
	PROF_SCOPE(c, PS_MODES);
	score_block_modes<false>(c, partition_count, start, end, max_weight_quant);
}
WV_OUT void stage_modes_2planes(int partition_count, int start, int end, int max_weight_quant)
{
	const Ctx c = ctx_make_vector_tables();
	partition_count = wv_uniform(partition_count); start = wv_uniform(start); end = wv_uniform(end);
	max_weight_quant = wv_uniform(max_weight_quant);
	PROF_SCOPE(c, PS_MODES);
	score_block_modes<true>(c, partition_count, start, end, max_weight_quant);
}
WV_FN void stage_modes(int partition_count, int start, int end, int max_weight_quant, bool dual)
{
	if (dual) stage_modes_2planes(partition_count, start, end, max_weight_quant);
	else stage_modes_1plane(partition_count, start, end, max_weight_quant);
}

WV_OUT void stage_formats(bool dual, int partition_count, int partition_packed, int plane2_component, int start, int end)
{
	const Ctx c = ctx_make();
	dual = wv_uniform(dual); partition_count = wv_uniform(partition_count); partition_packed = wv_uniform(partition_packed);
	plane2_component = wv_uniform(plane2_component); start = wv_uniform(start); end = wv_uniform(end);
	TrialInfo& tr = c.tr();
	const PartView pv = part_view_lds(c, partition_count, partition_packed);
	if (dual)
	{
		// merged endpoints (ref: merge_endpoints :37) are what the format search sees
		WV_FOR(ch, 4)
		{
			int plane = ch == plane2_component ? 1 : 0;
			tr.rgbs[1][ch] = tr.ep0[plane][0][ch];   // staging rows (rgbs[1..2] are unused with 1 partition)
			tr.rgbs[2][ch] = tr.ep1[plane][0][ch];
		}
		WV_SYNC();
	}
	PROF_SCOPE(c, PS_FORMATS);
	if (partition_count == 1)
		compute_ideal_endpoint_formats(c, part_view_lds(c, 1, 0), dual ? &tr.rgbs[1] : tr.ep0[0], dual ? &tr.rgbs[2] : tr.ep1[0], start, end);
	else
		compute_ideal_endpoint_formats(c, pv, tr.ep0[0], tr.ep1[0], start, end);
	// (the candidate selection in the same stage: a stage call costs some fifty instructions of call sequence and context
	//  rebuild, and the selection shares this stage's uniform values)
	select_candidate_modes(c, partition_count, start, end);
}

WV_FN float stage_refine(int partition_count, int partition_packed, int plane2_component, float tune_errorval_threshold)
{
	const Ctx c = ctx_make();
	partition_count = wv_uniform(partition_count); partition_packed = wv_uniform(partition_packed);
	plane2_component = wv_uniform(plane2_component); tune_errorval_threshold = wv_uniform(tune_errorval_threshold);
	PROF_SCOPE(c, PS_X2);
	return refine_candidates(c, partition_count, partition_packed, plane2_component, tune_errorval_threshold);
}

/* One trial of the search (ref: compress_symbolic_block_for_partition_1plane :353, _2planes :715). */
__attribute__((always_inline)) WV_FN float compress_trial(const Ctx& c, bool dual, bool only_always, float tune_errorval_threshold,
                           int partition_count, int partition_packed, int plane2_component, int quant_limit)
{
	PROF_SCOPE(c, PS_X3);
	tune_errorval_threshold = wv_uniform(tune_errorval_threshold);
	TRACE_PUT(c, TR_PASS, (float)(partition_count * 64 + (dual ? 2 : 1) * 8 + (plane2_component + 1)));
#if defined(ASTC_TRACE)
	if (partition_count > 1) TRACE_PUT(c, TR_PARTITION_INDEX, (float)part_view(c, partition_count, partition_packed).h->partition_index);
#endif
	const int max_weight_quant = i_min((int)QUANT_32, quant_limit);
	const int max_decimation_modes = wv_uniform(only_always ? (int)c.root->decimation_mode_count_always : (int)c.root->decimation_mode_count_selected);
	const int ref_mask = (int)((1u << (max_weight_quant + 1)) - 1);
	int mode_start, mode_end;
	if (dual)
	{
		mode_start = wv_uniform((int)c.root->block_mode_count_1plane_selected);
		mode_end = wv_uniform((int)c.root->block_mode_count_1plane_2plane_selected);
	}
	else
	{
		mode_start = 0;
		mode_end = wv_uniform(only_always ? (int)c.root->block_mode_count_1plane_always : (int)c.root->block_mode_count_1plane_selected);
	}

	DUP_STAGE(c, DUP_IDEAL, stage_ideal(dual, partition_count, partition_packed, plane2_component));
	DUP_STAGE(c, DUP_DECIMATE, stage_decimate(dual ? 2 : 1, ref_mask, max_decimation_modes));
	DUP_STAGE(c, DUP_ANGULAR, stage_angular(dual, partition_count, plane2_component, max_decimation_modes, ref_mask, max_weight_quant));
	// (the format search adds to the mode records in place, so it is doubled together with the scoring that resets them)
	DUP_STAGE(c, DUP_MODES_FORMATS, {
	DUP_STAGE(c, DUP_MODES, stage_modes(partition_count, mode_start, mode_end, max_weight_quant, dual));
	stage_formats(dual, partition_count, partition_packed, plane2_component, mode_start, mode_end); });
	return wv_uniform(stage_refine(partition_count, partition_packed, dual ? plane2_component : -1, tune_errorval_threshold));
}

__attribute__((always_inline)) WV_FN float compress_block_1plane(const Ctx& c, bool only_always, float tune_errorval_threshold,
                                  int partition_count, int partition_packed, int quant_limit)
{
	return compress_trial(c, false, only_always, tune_errorval_threshold, partition_count, partition_packed, -1, quant_limit);
}

__attribute__((always_inline)) WV_FN float compress_block_2planes(const Ctx& c, float tune_errorval_threshold, int plane2_component, int quant_limit)
{
	return compress_trial(c, true, false, tune_errorval_threshold, 1, 0, plane2_component, quant_limit);
}

/* (ref: prepare_block_statistics :1047) */
WV_FN float prepare_block_statistics(const Ctx& c)
{
	TrialInfo& tr = c.tr();
	const BlkInfo& blk = c.blk();
	const int T = c.T;

	// chains: 0 weight_sum, 1-4 rs gs bs as, 5 rr 6 gg 7 bb 8 aa, 9 rg 10 rb 11 ra 12 gb 13 ga 14 ba
	WV_FOR(k, 15)
	{
		const int ca[15] = { 0, 0, 1, 2, 3, 0, 1, 2, 3, 0, 0, 0, 1, 1, 2 };
		const int cb[15] = { -1, -1, -1, -1, -1, 0, 1, 2, 3, 1, 2, 3, 2, 3, 3 };
		const float* da = c.data(ca[k]);
		const float* db = cb[k] >= 0 ? c.data(cb[k]) : da;
		float acc = 0.0f;
		for (int i = 0; i < T; i++)
		{
			float weight = hadd4(cw_of(blk, 0), cw_of(blk, 1), cw_of(blk, 2), cw_of(blk, 3)) / 4.0f;
			if (k == 0) acc += weight;
			else
			{
				float aw = da[i] * weight;
				if (cb[k] < 0) acc += aw;
				else acc += db[i] * aw;
			}
		}
		tr.fbox[k] = acc;
	}
	WV_SYNC();

	const float* s = tr.fbox;
	float weight_sum = s[0], rs = s[1], gs = s[2], bs = s[3], as = s[4];
	float rr_var = s[5], gg_var = s[6], bb_var = s[7], aa_var = s[8];
	float rg_cov = s[9], rb_cov = s[10], ra_cov = s[11], gb_cov = s[12], ga_cov = s[13], ba_cov = s[14];

	float rpt = 1.0f / f_max(weight_sum, 1e-7f);

	rr_var -= rs * (rs * rpt);
	rg_cov -= gs * (rs * rpt);
	rb_cov -= bs * (rs * rpt);
	ra_cov -= as * (rs * rpt);

	gg_var -= gs * (gs * rpt);
	gb_cov -= bs * (gs * rpt);
	ga_cov -= as * (gs * rpt);

	bb_var -= bs * (bs * rpt);
	ba_cov -= as * (bs * rpt);

	aa_var -= as * (as * rpt);

	rg_cov *= 1.0f / f_sqrt(rr_var * gg_var);
	rb_cov *= 1.0f / f_sqrt(rr_var * bb_var);
	ra_cov *= 1.0f / f_sqrt(rr_var * aa_var);
	gb_cov *= 1.0f / f_sqrt(gg_var * bb_var);
	ga_cov *= 1.0f / f_sqrt(gg_var * aa_var);
	ba_cov *= 1.0f / f_sqrt(bb_var * aa_var);

	if (f_isnan(rg_cov)) rg_cov = 1.0f;
	if (f_isnan(rb_cov)) rb_cov = 1.0f;
	if (f_isnan(ra_cov)) ra_cov = 1.0f;
	if (f_isnan(gb_cov)) gb_cov = 1.0f;
	if (f_isnan(ga_cov)) ga_cov = 1.0f;
	if (f_isnan(ba_cov)) ba_cov = 1.0f;

	float lowest_correlation = f_min(f_abs(rg_cov), f_abs(rb_cov));
	lowest_correlation = f_min(lowest_correlation, f_abs(ra_cov));
	lowest_correlation = f_min(lowest_correlation, f_abs(gb_cov));
	lowest_correlation = f_min(lowest_correlation, f_abs(ga_cov));
	lowest_correlation = f_min(lowest_correlation, f_abs(ba_cov));
	WV_SYNC();
	return lowest_correlation;
}

WV_OUT int stage_partition_order(int partition_count)
{
	const Ctx c = ctx_make_vector_tables();
	partition_count = wv_uniform(partition_count);
	PROF_SCOPE(c, PS_KMEANS);
	return partition_search_order(c, partition_count);
}
WV_FN void stage_partition_score(int partition_count, int search_limit)
{
	const Ctx c = ctx_make();
	partition_count = wv_uniform(partition_count); search_limit = wv_uniform(search_limit);
	PROF_SCOPE(c, PS_KMEANS);
	partition_search_score(c, partition_count, search_limit);
}
WV_OUT int stage_partition_select(int search_limit, int requested_trials)
{
	const Ctx c = ctx_make();
	search_limit = wv_uniform(search_limit); requested_trials = wv_uniform(requested_trials);
	PROF_SCOPE(c, PS_KMEANS);
	return partition_search_select(c, search_limit, requested_trials);
}

/* (ref: find_best_partition_candidates :551) candidates -> PartScratch::best[], returns how many */
WV_FN int stage_partition_search(const Ctx& c, int partition_count, int requested_indices, int requested_trials)
{
	(void)c;      // (read by DUP_STAGE in instruction-count builds)
	int sequence_len;
	DUP_STAGE(c, DUP_PART_ORDER, sequence_len = wv_uniform(stage_partition_order(partition_count)));
	const int search_limit = i_min(requested_indices, sequence_len);
	DUP_STAGE(c, DUP_PART_SCORE, stage_partition_score(partition_count, search_limit));
	int found;
	DUP_STAGE(c, DUP_PART_SELECT, found = stage_partition_select(search_limit, i_min(search_limit, requested_trials)));
	return found;
}

WV_OUT float stage_block_statistics()
{
	const Ctx c = ctx_make();
	PROF_SCOPE(c, PS_STATS);
	return prepare_block_statistics(c);
}

/* The search of compress_block for a block that is not one constant colour: leaves the best encoding found in
 * c.scb(), or block_type SYM_BTYPE_ERROR.  (ref: compress_block :1247-1430) */
__attribute__((always_inline)) WV_FN void search_block(const Ctx& c)
{
	const BlkInfo& blk = c.blk();
	const DeviceConfig& cfg = *c.cfg;
	Scb& scb = c.scb();
	const int T = c.T;

	bool block_is_l = blk_is_luminance(blk);
	float block_is_l_scale = block_is_l ? 1.0f / 1.5f : 1.0f;
	bool block_is_la = blk_is_luminancealpha(blk);
	float block_is_la_scale = block_is_la ? 1.0f / 1.05f : 1.0f;

	float error_weight_sum = hadd4(cw_of(blk, 0), cw_of(blk, 1), cw_of(blk, 2), cw_of(blk, 3)) * (float)T;
	// (driver state of the whole block: wave-uniform, kept in scalar registers -- values derived from LDS or table
	//  loads look lane-variant to the compiler and would be carried, and spilled, as vector registers)
	const float error_threshold = wv_uniform(cfg.tune_db_limit * error_weight_sum * block_is_l_scale * block_is_la_scale);

	TRACE_PUT(c, TR_THRESHOLD, error_threshold);
	WV_ONE
	{
		scb.errorval = ERROR_CALC_DEFAULT;
		scb.block_type = SYM_BTYPE_ERROR;
		c.tr().staged_color_quant[0] = -1;
		c.tr().staged_color_quant[1] = -1;
		c.tr().eci1_valid = 0;
		c.tr().ideal_1p1p_valid = 0;
		c.tr().dirsum1_mask = 0;
	}
	WV_SYNC();

	// (the reference's best_errorvals_for_pcount[] / exit_thresholds_for_pcount[] / errorval_mult[] as scalars:
	//  run-time indexed local arrays would live in scratch memory)
	float best_errorval_prev_pcount = ERROR_CALC_DEFAULT;      // best error with one partition fewer
	const float errorval_overshoot = wv_uniform(1.0f / cfg.tune_mse_overshoot);
	const float trial_threshold = wv_uniform(error_threshold * errorval_overshoot);     // what the 2-plane and partitioned trials aim for

	int start_trial = 1;
	if (cfg.tune_search_mode0_enable >= 0.85f && c.root->dim_z == 1) start_trial = 0;   // ref: compress_symbolic.cpp:1287

	int quant_limit = QUANT_32;
	bool done = false;

	// trial A: 1 partition, 1 plane (ref: :1292-1318)
	for (int i = start_trial; i < 2 && !done; i++)
	{
		const float errorval_mult = i == 0 ? 1.0f / cfg.tune_mse_overshoot : 1.0f;
		float errorval;
		DUP_STAGE(c, i == 0 ? DUP_TRIAL_A0 : DUP_TRIAL_A1, errorval = compress_block_1plane(c, i == 0, error_threshold * errorval_mult * errorval_overshoot, 1, 0, QUANT_32));
		WV_SYNC();
		if (scb.block_type != SYM_BTYPE_ERROR)
		{
			quant_limit = wv_uniform((int)c.block_mode(scb.block_mode).quant_mode);
		}
		best_errorval_prev_pcount = wv_uniform(f_min(best_errorval_prev_pcount, errorval));
		if (errorval < (error_threshold * errorval_mult)) done = true;
	}

	// trial B: 1 partition, 2 planes (ref: :1320-1369)
	if (!done)
	{
		float lowest_correl;
		DUP_STAGE(c, DUP_STATS, lowest_correl = wv_uniform(stage_block_statistics()));
		TRACE_PUT(c, TR_LOWEST_CORREL, lowest_correl);
		bool block_skip_two_plane = lowest_correl > cfg.tune_2plane_early_out_limit_correlation;
		for (int i = 3; i >= 0 && !done; i--)
		{
			if (block_skip_two_plane) continue;
			if (blk.grayscale && i != 3) continue;
			if (is_constant_channel(blk, i)) continue;

			float errorval;
			DUP_STAGE(c, DUP_TRIAL_2PLANES, errorval = compress_block_2planes(c, trial_threshold, i, quant_limit));
			WV_SYNC();
			if (errorval > (best_errorval_prev_pcount * 1.85f)) break;
			if (errorval < error_threshold) done = true;
		}
	}

	// trial C: 2..4 partitions (ref: :1372-1429)
	if (!done)
	{
		const int max_partitions = wv_uniform((int)cfg.tune_partition_count_limit);
		for (int partition_count = 2; partition_count <= max_partitions && !done; partition_count++)
		{
			int requested_indices = wv_uniform((int)cfg.tune_partition_index_limit[partition_count - 2]);
			int requested_trials = wv_uniform((int)cfg.tune_partitioning_candidate_limit[partition_count - 2]);
			requested_trials = i_min(requested_trials, requested_indices);

			int actual_trials;
			actual_trials = wv_uniform(stage_partition_search(c, partition_count, requested_indices, requested_trials));
			// copy out of the scratch region (the trials below reuse it): candidate i sits in lane i
			LaneArray128 partition_indices;
			partition_indices.clear();
			{
				const PartScratch& ps = *reinterpret_cast<const PartScratch*>(c.part());
				WV_FOR64(i, actual_trials) { partition_indices.set(i, ps.best[i]); }
			}
			WV_SYNC();

			const float best_error_in_prev = best_errorval_prev_pcount;
			const float exit_threshold = wv_uniform(partition_count == 2 ? cfg.tune_partition_early_out_limit_factor[0]
			                           : partition_count == 3 ? cfg.tune_partition_early_out_limit_factor[1] : 0.0f);
			float best_error = ERROR_CALC_DEFAULT;               // best error with this partition count

			for (int i = 0; i < actual_trials; i++)
			{
				float errorval;
				DUP_STAGE(c, DUP_TRIAL_2PARTITIONS + (partition_count - 2), errorval = compress_block_1plane(c, false, trial_threshold,
				                                       partition_count, partition_indices.get(i), quant_limit));
				WV_SYNC();
				best_error = wv_uniform(f_min(best_error, errorval));

				if (best_error > (best_error_in_prev * (exit_threshold * 1.85f))) { done = true; break; }
				if (errorval < error_threshold) { done = true; break; }
			}
			if (done) break;

			if (best_error > (best_error_in_prev * exit_threshold)) { done = true; break; }
			best_errorval_prev_pcount = best_error;
		}
	}

}

/* Compress the block currently loaded in LDS and write its 16 bytes. (ref: compress_block :1162)
 * `out`: the launch's output array; the block's index in it is blk.block_index (kept in LDS rather than in a
 * register pair across the whole search). */
WV_FN void compress_block(const Ctx& c, uint8_t* out)
{
	bool constant_color;
	{
		const BlkInfo& blk = c.blk();
		constant_color = blk.data_min[0] == blk.data_max[0] && blk.data_min[1] == blk.data_max[1] &&
		                 blk.data_min[2] == blk.data_max[2] && blk.data_min[3] == blk.data_max[3];
	}
	if (!wv_uniform(constant_color)) search_block(c);

	// (the context is rebuilt from the LDS header here, like in an out-of-line stage: anything derived from `c`
	//  above would have to stay in registers across the whole search)
	WV_SYNC();
	WV_ONE
	{
		const Ctx ce = ctx_make();
		Scb& scb = ce.scb();
		const BlkInfo& blk = ce.blk();
		if (constant_color)
		{
			// constant colour -> void extent (ref: :1216-1245)
			scb.partition_count = 0;
			if (ce.cfg->profile == 3 || ce.cfg->profile == 2)
			{
				scb.block_type = SYM_BTYPE_CONST_F16;
				for (int k = 0; k < 4; k++) scb.constant_color[k] = float_to_half(blk.origin[k]);
			}
			else
			{
				scb.block_type = SYM_BTYPE_CONST_U16;
				for (int k = 0; k < 4; k++)
				{
					float v = v_clamp(0.0f, 1.0f, blk.origin[k]) * 65535.0f;
					scb.constant_color[k] = (int)(v + 0.5f);
				}
			}
		}
		else if (scb.block_type == SYM_BTYPE_ERROR)
		{
			// no valid encoding found: constant colour of texel 0 (ref: :1436-1452)
			scb.block_type = SYM_BTYPE_CONST_U16;
			for (int k = 0; k < 4; k++)
			{
				float v = v_clamp(0.0f, 1.0f, blk.origin[k]) * 65535.0f;
				scb.constant_color[k] = (int)(v + 0.5f);
			}
		}
	}
	WV_SYNC();
	{
		const Ctx ce = ctx_make();
		PROF_SCOPE(ce, PS_X1);
		uint8_t* pcb = out + (size_t)wv_uniform(ce.blk().block_index) * 16;
		DUP_STAGE(ce, DUP_PHYSICAL, symbolic_to_physical(ce, ce.scb(), pcb));
	}
}

} } // namespace astcd::ASTC_VARIANT
