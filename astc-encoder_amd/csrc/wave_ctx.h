// SPDX-License-Identifier: Apache-2.0
// Per-block working set of the wavefront compressor: LDS layout and the uniform context object.
//
// The reference keeps this state in image_block (astcenc_internal.h:749), symbolic_compressed_block
// (:1077) and compression_working_buffers (:946; 231 KB per thread).  Here it is an LDS region whose
// layout is computed once per context from the real texel / mode counts (15 KB for 6x6 -medium).
#pragma once
#include "astc_tables.h"
#include "wave.h"

namespace astcd { inline namespace ASTC_VARIANT {

/* Block-lifetime scalars. (ref: image_block :749) */
struct BlkInfo {
	float data_min[4];
	float data_mean[4];
	float data_max[4];
	float origin[4];
	float cw[4];          // channel_weight
	int   grayscale;
	int   rgb_lns;        // blk.rgb_lns[0]
	int   alpha_lns;      // blk.alpha_lns[0]
	uint32_t block_index; // which 16 bytes of the output this block writes (read back when the block is written)
};

/* Symbolic block. (ref: symbolic_compressed_block :1077).  block_mode / partition_index hold PACKED
 * table indices while searching; they are translated when the physical block is written. */
struct Scb {
	uint8_t  block_type;
	uint8_t  partition_count;
	uint8_t  color_formats_matched;
	int8_t   plane2_component;
	uint16_t block_mode;
	uint16_t partition_index;
	uint8_t  color_formats[4];
	uint8_t  quant_mode;
	uint8_t  pad[3];
	float    errorval;
	union {
		uint8_t color_values[4][8];
		int     constant_color[4];
	};
	uint8_t  weights[64];
};

/* Trial-lifetime scalars: ideal endpoints of both planes etc. (ref: endpoints_and_weights :904) */
struct TrialInfo {
	float ep0[2][4][4];       // [plane][partition][channel]  endpt0
	float ep1[2][4][4];
	int   is_constant_wes[2];
	float min_wt_cutoff[2];
	// candidate search results (ref: compute_ideal_endpoint_formats outputs)
	int   cand_count;
	int   cand_block_mode[MAX_TRIAL_CANDIDATES];
	uint8_t cand_quant[MAX_TRIAL_CANDIDATES];
	uint8_t cand_quant_mod[MAX_TRIAL_CANDIDATES];
	uint8_t cand_formats[MAX_TRIAL_CANDIDATES][4];
	// working endpoints of the candidate being refined
	float wep0[4][4];
	float wep1[4][4];
	float rgbs[4][4];
	float rgbo[4][4];
	// partition metrics
	float pm_avg[4][4];
	float pm_dir[4][4];
	// encoding choice errors per partition (ref: encoding_choice_errors :922)
	float eci_rgb_scale[4], eci_rgb_luma[4], eci_luminance[4], eci_alpha_drop[4];
	int   eci_can_offset[4], eci_can_blue_contract[4];
	// Block-lifetime cache of what the single-partition trials share (search_block clears the flags): the four
	// encoding choice errors depend on the block and the partitioning only, so the two runs of trial A and the four
	// two-plane trials (all with the one-partition "partitioning") compute them once; the ideal endpoints and weights
	// of the one-plane trial are identical in both runs of trial A.
	int   eci1_valid;
	float eci1[4];            // rgb_scale, rgb_luma, luminance, alpha_drop of partition 0 of 1
	int   ideal_1p1p_valid;   // ei_w / ei_wes / ep0 / ep1 / is_constant_wes hold the 1-partition 1-plane result
	// ... and the direction sums of the one-partition "partitioning" (compute_avgs_and_dirs): entry [which][j] -- the sum of
	// channel j's offsets from the block mean over the texels whose channel `which` is above its mean -- depends on the two
	// channels only, so the one-plane trial's 3 x 3 or 4 x 4 entries serve the channel subsets of the two-plane trials
	uint32_t dirsum1_mask;    // bit which * 4 + j: dirsum1[which][j] is valid
	float dirsum1[4][4];
	// the staged partitioning's per-partition start / size bytes (PartView::offsets / counts): every refinement step
	// builds a view of it, and two dependent L2 round trips for the header each time would be most of such a step's latency
	uint32_t part_offsets, part_counts;
	// generic small uniform mailboxes
	float fbox[128];
	int   ibox[64];
	// decimation modes referenced by the current trial (compacted list)
	int   dm_count;
	int   staged_color_quant[2];
	uint8_t dm_list[96];
};

/* Per block mode record of one trial: the weight quantization error, then (in place) + the error of the best colour
 * combination.  Only the error is kept per mode: the quant levels and endpoint formats that go with it are recomputed
 * for the handful of modes that become candidates (compute_ideal_endpoint_formats) -- at 217 one-plane modes (8x8
 * -thorough) eight more bytes per mode were 1.7 KB of every block's LDS. */
struct ModeRec {
	float error;
};

/* First bytes of every workgroup's LDS: what an out-of-line stage function needs to rebuild its Ctx. */
struct LdsHeader {
	const uint8_t* base;        // start of [LdsLayout][DeviceConfig][table blob] in device memory (blob = base + CTX_LAYOUT_BACK):
	                            // every field of the three is then a non-negative immediate offset of one scalar base
	unsigned long long* prof;
};

/* The offsets inside the scratch of the batched first refinement step (BatchView, wave_batch.h), per trial class and
 * partition count: computed once on the host instead of in the prologue of each of the step's stages. */
struct BatchOffsets { uint32_t o_x, o_iw, o_sum, o_dec, o_term, o_ctab, o_cand, o_vec, o_state0; };

struct LdsLayout {
	uint32_t hdr;        // LdsHeader (offset 0)
	// ---- block / trial lifetime ----
	uint32_t data;       // f32 [4][Tp]
	uint32_t blk;        // BlkInfo
	uint32_t scb;        // Scb (best so far)
	uint32_t wscb;       // Scb (candidate being refined)
	uint32_t trial;      // TrialInfo
	uint32_t ei_w;       // f32 [2][Tp]    ideal weights per plane
	uint32_t ei_wes;     // f32 [2][Tp]    weight error scale per plane
	uint32_t ptab;       // u8  [2][Tp]    staged partition record of the current trial
	uint32_t candw;      // u8  [candidates][64]  quantized weights of the chosen candidates
	// ---- phase-multiplexed region: {search | refine | partition search} never overlap in time ----
	uint32_t dwi;        // f32 packed     search: ideal weights of every grid (DecimationMode::dwi_offset)
	uint32_t lowhigh;    // f32 [slots][16] search: angular low/high per quant level
	uint32_t modes;      // ModeRec [NBM]  search
	uint32_t uni;        // search: infill rows, angular batch [64][8] f32, mode-score terms, then
	                     //         5 texel rows of the encoding-choice errors, then the format tables
	uint32_t dtab;       // refine: staged decimation tables of the candidate
	uint32_t ctab;       // refine: u8 [512] staged colour quant rows of the candidate's quant level
	uint32_t qtab;       // refine: QuantXfer of the candidate's weight quant level
	uint32_t rsc;        // refine: f32 [19][Tp] per-texel term rows of the endpoint re-fit
	uint32_t tsc_r;      // refine: f32 [2][Tp] weights expanded to texel resolution
	uint32_t wsc;        // refine: f32 [2][64] per-weight scratch rows
	uint32_t part;       // partition search scratch
	uint32_t tsc_p;      // partition search: f32 [2][Tp] k-means rows
	uint32_t part_tabs;  // partition search: staged records (header + texel lists) of the candidates being scored
	uint32_t part_chunk; // candidates per staging pass
	uint32_t part_rec_words;   // 32-bit words of one staged record
	uint32_t part_rec_inv24;   // ceil(2^24 / part_rec_words): k / words == (k * inv) >> 24 for k < chunk * words
	uint32_t uni_bytes;  // size of the `uni` region
	uint32_t tsc_stride; // floats between tsc rows
	uint32_t t_inv24;    // ceil(2^24 / texel_count): k / T == (k * t_inv24) >> 24 for k < 2^24 / T (no integer divide on the device)
	uint32_t mode_chunk; // block modes scored per pass of score_block_modes (descriptors + quantized weights + texel terms fit `uni`)
	uint32_t mode_wcap[2];     // weights per plane the scoring passes provide lanes for, per trial class (1-plane, 2-plane)
	uint32_t mode_wcap_inv[2]; // ceil(2^16 / mode_wcap): k / wcap == (k * inv) >> 16
	// refine: the first refinement step of a trial's candidates runs for several of them at once (wave_batch.h)
	uint32_t bat;              // scratch of that step: the whole refine region up to `cstate`
	uint32_t cstate;           // what the step leaves for the candidates' own turns: one record per candidate but the first
	uint32_t cstate_stride;    // bytes per record (cand_state_bytes)
	uint32_t bat_max[2];       // candidates per batch in a 1-plane / 2-plane trial (1: nothing to gain, still the same code)
	uint32_t total;
	uint32_t texel_count;      // (TableRoot::texel_count as a 32-bit word: a scalar load in the stages' prologues)
	BatchOffsets batview[2][4];   // [2-plane trial][partition count - 1]
};

/* Staged copy of a grid's tables in LDS (LdsLayout::dtab): its DecimationInfo record first (the refinement steps then
 * find sizes and offsets with an LDS read instead of two dependent global loads), the tables DTAB_RECORD_BYTES in. */
constexpr uint32_t DTAB_RECORD_BYTES = 64;

/* Stride (floats) of the texel-length scratch rows that several lanes walk side by side, one row each (the running sums
 * of the endpoint re-fit, the term rows of the mode scoring): a texel count that is a multiple of 16 would put the same
 * column of every row into the same LDS bank -- 8x8 blocks: 64 floats = twice the 32 banks, a 14-way conflict on every
 * read of the re-fit's chains -- so those footprints get four floats of padding per row (rows stay 16-byte aligned). */
WV_FN int lds_row_stride(int Tp) { return Tp + ((Tp & 15) == 0 ? 4 : 0); }
/* ... and the re-fit rows (Ctx::rsc): up to 18 lanes each walk their own row; Tp is a multiple of 4, so Tp + 2 shares
 * only a factor 2 with the 32 banks: sixteen consecutive rows start in sixteen different banks (even rows stay 16-byte
 * aligned, which is what the arrays carved out of row 2 onwards rely on). */
WV_FN int lds_refit_stride(int Tp) { return Tp + 2; }
static_assert(sizeof(DecimationInfo) <= DTAB_RECORD_BYTES, "DecimationInfo outgrew its staged slot");

constexpr uint32_t LDS_ALLOC_GRANULE = 1280;   // gfx950: 160 KiB of LDS per CU in 128 allocation units

/* Sizes in bytes of the variable scratch regions. */

/* (MODE_DESC_BYTES, the endpoint-format table sizes and uni_region_bytes() live in astc_tables.h: the table
 * builder needs them too.) */

WV_FN uint32_t part_scratch_bytes(uint32_t max_partitionings, uint32_t max_index_limit)
{
	// see struct PartScratch (wave_partition.h): fixed header, then ordering u16[n], errors f32[2][limit], mismatch u8[n]
	uint32_t n = (max_partitionings + 3u) & ~3u;
	uint32_t lim = max_index_limit < n ? max_index_limit : n;
	return 256 + n * 2 + lim * 8 + n;
}

/* Per-candidate record of the batched first refinement step (wave_batch.h: CandState): the endpoints of every partition
 * as floats, the packed colour bytes, formats, flags, the step's error. */
WV_FN uint32_t cand_state_bytes(uint32_t partition_limit)
{
	const uint32_t P = partition_limit < 1 ? 1u : partition_limit > 4 ? 4u : partition_limit;
	return (2u * P * 16u + P * 8u + 4u + 4u + 4u + 15u) & ~15u;
}
/* Bytes of the step's scratch for `nb` candidates (BatchView, wave_batch.h): seven shared texel rows, then per candidate
 * the expanded weights (float and integer, per plane), the sums of every partition, the decoded endpoints, the error
 * terms, the colour quantization rows (one set more for the matched-format retry), a small record, the first state. */
WV_FN uint32_t batch_scratch_bytes(uint32_t nb, uint32_t planes, uint32_t pc, uint32_t Tp, uint32_t state_bytes)
{
	const uint32_t iw = (nb * planes * Tp + 15u) & ~15u;
	const uint32_t Ts = (uint32_t)lds_row_stride((int)Tp);
	return 7u * Ts * 4u + nb * planes * Ts * 4u + iw + nb * pc * 112u + nb * pc * 32u + nb * Ts * 4u + (nb + 1u) * 512u + nb * 32u + nb * 4u * 32u + state_bytes;
}

/* Where the parts of the step's scratch start (bytes from LdsLayout::bat), in the order batch_scratch_bytes() adds them up. */
WV_FN BatchOffsets batch_offsets(uint32_t nb, uint32_t planes, uint32_t pc, uint32_t Tp, uint32_t P)
{
	const uint32_t Ts = (uint32_t)lds_row_stride((int)Tp);
	BatchOffsets v;
	uint32_t o = 7u * Ts * 4u;
	v.o_x = o;     o += nb * planes * Ts * 4u;
	v.o_iw = o;    o += (nb * planes * Tp + 15u) & ~15u;
	v.o_sum = o;   o += nb * pc * 112u;
	v.o_dec = o;   o += nb * pc * 32u;
	v.o_term = o;  o += nb * Ts * 4u;
	v.o_ctab = o;  o += (nb + 1u) * 512u;
	v.o_cand = o;  o += nb * 32u;
	v.o_vec = o;   o += nb * P * 32u;
	v.o_state0 = o;
	return v;
}

WV_FN void make_lds_layout(const TableRoot& r, const DeviceConfig& cfg, LdsLayout& L)
{
	uint32_t Tp = (r.texel_count + 3u) & ~3u;
	uint32_t nbm = r.block_mode_count_1plane_2plane_selected;
	uint32_t o = 0;
	auto take = [&](uint32_t bytes) { uint32_t at = o; o += (bytes + 15u) & ~15u; return at; };
	L.hdr = take(sizeof(LdsHeader));
	L.data = take(4 * Tp * 4);
	L.blk = take(sizeof(BlkInfo));
	L.scb = take(2 * sizeof(Scb));                               // best + working symbolic block, back to back
	L.wscb = L.scb + (uint32_t)sizeof(Scb);
	L.trial = take(sizeof(TrialInfo));
	L.ei_w = take(2 * Tp * 4);
	L.ei_wes = take(2 * Tp * 4);
	L.ptab = take(2 * Tp);
	L.candw = take(cfg.tune_candidate_limit * 64);
	L.tsc_stride = Tp;

	const uint32_t begin = o;
	// search phase
	// one trial is either 1-plane or 2-plane: size for the larger class
	uint32_t nbm1 = r.block_mode_count_1plane_selected;
	uint32_t nbm_max = nbm1 > nbm - nbm1 ? nbm1 : nbm - nbm1;
	L.dwi = take((r.dwi_total_floats[0] > r.dwi_total_floats[1] ? r.dwi_total_floats[0] : r.dwi_total_floats[1]) * 4);
	L.lowhigh = take((r.lowhigh_floats[0] > r.lowhigh_floats[1] ? r.lowhigh_floats[0] : r.lowhigh_floats[1]) * 4);
	{
		// the mode records; before the modes are scored the same bytes hold one sin/cos table row index per ideal
		// weight (decimation sweeps -> angular search), so the region is at least one byte per packed weight slot
		uint32_t bytes = nbm_max * (uint32_t)sizeof(ModeRec);
		const uint32_t slots = r.dwi_total_floats[0] > r.dwi_total_floats[1] ? r.dwi_total_floats[0] : r.dwi_total_floats[1];
		if (slots > bytes) bytes = slots;
		L.modes = take(bytes);
	}
	L.uni_bytes = uni_region_bytes(r.texel_count, cfg.tune_partition_count_limit);
	L.t_inv24 = ((1u << 24) + (uint32_t)r.texel_count - 1u) / (uint32_t)r.texel_count;
	L.mode_chunk = (L.uni_bytes - MODE_Q2U_BYTES) / (MODE_DESC_BYTES + MODE_WEIGHT_BYTES);
	if (L.mode_chunk > 16u) L.mode_chunk = 16u;
	for (int cls = 0; cls < 2; cls++)
	{
		const uint32_t cap = (r.max_weights[cls] + 3u) & ~3u;
		L.mode_wcap[cls] = cap;
		L.mode_wcap_inv[cls] = (65536u + cap - 1u) / cap;
	}
	L.uni = take(L.uni_bytes);
	uint32_t end = o;
	// refine phase
	o = begin;
	L.dtab = take(DTAB_RECORD_BYTES + r.max_decimation_table_bytes);    // the candidate grid's DecimationInfo record, then its tables
	L.ctab = take(512);
	L.qtab = take(sizeof(QuantXfer));
	const uint32_t Tr = (uint32_t)lds_refit_stride((int)Tp);
	L.rsc = take(19 * Tr * 4);
	{
		// the re-fit rows double as scratch of the difference / realign steps; realign needs 3 texel rows + 12 rows of
		// one weight's texel list
		uint32_t need = 3 * Tr + r.realign_rt_floats;
		if (need > 19 * Tr) { o = L.rsc; take(need * 4); }
	}
	L.tsc_r = take(2 * Tp * 4);                                  // expanded weights of plane 0 / 1
	L.wsc = take(2 * 64 * 4);
	const uint32_t classic_end = o;
	if (o > end) end = o;
	// The hardware hands out LDS in units of LDS_ALLOC_GRANULE bytes (measured on MI355X, DESIGN.md section 6: the
	// workgroups per CU step at multiples of 1280 B): what the largest phase leaves of its last unit costs no occupancy.
	// The mode scoring takes it -- more block modes per pass of its descriptor / quantize / score / sum sequence, whose
	// table loads are latency-bound -- and so do the staged partition records below.
	{
		const uint32_t rounded = (end + LDS_ALLOC_GRANULE - 1u) / LDS_ALLOC_GRANULE * LDS_ALLOC_GRANULE;
		if (rounded <= 64u * 1024u)
		{
			L.uni_bytes = (rounded - L.uni) & ~15u;
			L.mode_chunk = (L.uni_bytes - MODE_Q2U_BYTES) / (MODE_DESC_BYTES + MODE_WEIGHT_BYTES);
			if (L.mode_chunk > 16u) L.mode_chunk = 16u;       // (four accumulator lanes per mode of a chunk: one wave-trip, score_block_modes)
			end = rounded;
		}
	}
	// partition search phase
	o = begin;
	uint32_t lim = cfg.tune_partition_index_limit[0];
	if (cfg.tune_partition_index_limit[1] > lim) lim = cfg.tune_partition_index_limit[1];
	if (cfg.tune_partition_index_limit[2] > lim) lim = cfg.tune_partition_index_limit[2];
	L.part = take(part_scratch_bytes(r.max_partitionings, lim));
	L.tsc_p = take(2 * Tp * 4);
	{
		// as many candidate records as fit without growing the block's LDS (at least 8, at most one per lane)
		uint32_t rec = (uint32_t)(sizeof(PartitionHeader) + 2 * r.texel_count + 3u) & ~3u;
		// (an odd number of words: the candidates' lanes read the same position of consecutive records, which then fall
		// into different banks; the extra word is the first one of the next record in the blob)
		if (((rec >> 2) & 1u) == 0) rec += 4;
		uint32_t room = end > o ? (end - o) / rec : 0;
		uint32_t chunk = room < 8 ? 8 : room;
		if (chunk > 64) chunk = 64;
		if (chunk > lim) chunk = lim < 1 ? 1 : lim;
		L.part_chunk = chunk;
		L.part_rec_words = rec >> 2;
		L.part_rec_inv24 = ((1u << 24) + (rec >> 2) - 1u) / (rec >> 2);
		L.part_tabs = take(chunk * rec);
	}
	if (o > end) end = o;
	L.total = end;
	// The batched first refinement step: its scratch is the refine region itself (nothing of a candidate's own turn is live
	// while it runs), the records it leaves behind sit between the end of that region and the end of the allocation --
	// as many candidates per batch as both allow.
	{
		L.bat = begin;
		L.cstate = classic_end;
		L.cstate_stride = cand_state_bytes(cfg.tune_partition_count_limit);
		const uint32_t room = L.total - classic_end;
		const uint32_t pcl = cfg.tune_partition_count_limit < 1 ? 1u : cfg.tune_partition_count_limit;
		for (int cls = 0; cls < 2; cls++)
		{
			uint32_t nb = 1u + room / L.cstate_stride;
			if (nb > cfg.tune_candidate_limit) nb = cfg.tune_candidate_limit;
			if (nb > (uint32_t)MAX_TRIAL_CANDIDATES) nb = (uint32_t)MAX_TRIAL_CANDIDATES;
			if (nb * (cls ? 1u : pcl) > 16u) nb = 16u / (cls ? 1u : pcl);        // (one quad per (candidate, partition) in one trip)
			while (nb > 1u && batch_scratch_bytes(nb, cls ? 2u : 1u, cls ? 1u : pcl, Tp, L.cstate_stride) > classic_end - begin) nb--;
			if (nb < 1u) nb = 1u;
			L.bat_max[cls] = nb;
			// (a footprint whose one-candidate scratch outgrows the refine region: the region grows)
			const uint32_t need = batch_scratch_bytes(nb, cls ? 2u : 1u, cls ? 1u : pcl, Tp, L.cstate_stride);
			if (begin + need > L.cstate) { L.cstate = (begin + need + 15u) & ~15u; if (L.cstate > L.total) L.total = L.cstate; }
		}
		// (a class processed later may have moved cstate up: the records of a class with more than one candidate per batch
		//  -- (bat_max - 1) of them from cstate on -- must still end inside the allocation)
		{
			const uint32_t nbmax = L.bat_max[0] > L.bat_max[1] ? L.bat_max[0] : L.bat_max[1];
			const uint32_t state_end = L.cstate + (nbmax - 1u) * L.cstate_stride;
			if (state_end > L.total) L.total = (state_end + 15u) & ~15u;
		}
		const uint32_t P = pcl > 4u ? 4u : pcl;
		for (uint32_t cls = 0; cls < 2; cls++)
			for (uint32_t pc = 1; pc <= 4; pc++)
			{
				L.batview[cls][pc - 1] = batch_offsets(L.bat_max[cls], cls ? 2u : 1u, pc, Tp, P);
#if !WV_DEVICE
				// (the two descriptions of the scratch -- where its parts start, how much is set aside -- must agree: the second
				//  counts the vector rows of four partitions whatever the partition limit)
				if (pc <= (cls ? 1u : pcl) && L.batview[cls][pc - 1].o_state0 + L.cstate_stride > batch_scratch_bytes(L.bat_max[cls], cls ? 2u : 1u, pc, Tp, L.cstate_stride)) __builtin_trap();
#endif
			}
	}
	L.texel_count = r.texel_count;
}

/* FIXED-CONTEXT BUILDS (kernel_*_6x6m.hip, kernel_ldr_8x8t.hip).  A kernel build may be compiled for
 * ONE named context -- footprint x preset x profile with the default channel weights and flags, i.e. exactly what a BASELINE
 * config runs: ASTC_FIXED_CONTEXT names a record of fixed_contexts.inc, which holds that context's LdsLayout, DeviceConfig
 * and TableRoot word for word (generated from the sequential build of this source by tools/gen_fixed_contexts.py,
 * tests/test_fixed_contexts.py keeps it current).  In such a build the three records are compile-time constants: every LDS
 * region offset is an immediate of its ds_read / ds_write, every table offset an immediate of its load, the texel count,
 * the trip counts of the texel loops, the candidate / partition limits and the profile are literals, and a stage
 * function's prologue shrinks to fetching the blob pointer.  The host side of the same translation unit compares the live
 * context's three records with the constants byte for byte (ASTC_PREPARE_NAME) and the backend falls back to the generic
 * build of the footprint class when anything differs. */
#if defined(ASTC_FIXED_CONTEXT)
#if defined(ASTC_FIXED_RECORDS_FILE)
// (the sequential build compiled for one context: the `fixed` target of the debugging build's Makefile, tests/test_emu_fixed.py)
#include ASTC_FIXED_RECORDS_FILE
#else
#include "fixed_contexts.inc"      // (the record selected by the translation unit's ASTC_FIXED_<name>: kFixedLayout, kFixedConfig, kFixedRoot;
                                   //  a run-time build gets its own file under this name, kernel_jit.cpp)
#endif
#define ASTC_FIXED 1
#else
#define ASTC_FIXED 0
#endif

/* wv_uniform(v): `v` is the same on every lane (a value read from LDS or from a table looks lane-variant to the
 * compiler); returns it in a scalar register, so that control flow on it runs on the scalar unit and addresses built
 * on it use the scalar-base addressing mode. */
#if WV_DEVICE
WV_FN uint32_t wv_uniform(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
WV_FN int wv_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
WV_FN bool wv_uniform(bool v) { return __builtin_amdgcn_readfirstlane(v ? 1 : 0) != 0; }
WV_FN float wv_uniform(float v) { return int_as_float(__builtin_amdgcn_readfirstlane(float_as_int(v))); }
/* The same value, but the optimiser cannot see through it: an expression built on wv_opaque(lane) is not loop
 * invariant, so it is computed where it is used instead of being hoisted to the top of the kernel and carried (or
 * spilled to scratch memory) across every stage in between.  No instruction is emitted. */
WV_FN int wv_opaque(int v) { asm volatile("" : "+v"(v)); return v; }
WV_FN float wv_opaque_f(float v) { asm volatile("" : "+v"(v)); return v; }
WV_FN uint64_t wv_uniform(uint64_t v)
{
	uint32_t lo = wv_uniform((uint32_t)v), hi = wv_uniform((uint32_t)(v >> 32));
	return ((uint64_t)hi << 32) | lo;
}
#else
WV_FN uint32_t wv_uniform(uint32_t v) { return v; }
WV_FN int wv_uniform(int v) { return v; }
WV_FN int wv_opaque(int v) { return v; }
WV_FN float wv_opaque_f(float v) { return v; }
WV_FN bool wv_uniform(bool v) { return v; }
WV_FN float wv_uniform(float v) { return v; }
WV_FN uint64_t wv_uniform(uint64_t v) { return v; }
#endif

/* Uniform per-wave context. */
struct Ctx {
	const uint8_t* tab;          // table blob (HBM, read-only)
	bool tab_constant;           // read through the constant address space (scalar cache) rather than the global one
	const TableRoot* root;
	const DeviceConfig* cfg;
	uint8_t* lds;
	const LdsLayout* L;          // in the table blob (constant memory): fields are scalar loads
	int T;                       // texels per block
	int Tp;                      // T rounded up to 4
	int Ts;                      // lds_row_stride(Tp): stride of the scratch rows that lanes walk side by side
	unsigned long long* prof;    // stage cycle counters (profiling builds only), else null

	// a table of the blob at byte offset `off` (an offset read from the blob): scalar base pointer, so that lane-variant
	// indexing costs one 32-bit offset per access instead of 64-bit pointer arithmetic
	WV_FN const uint8_t* table(uint32_t off) const
	{
#if ASTC_FIXED && WV_DEVICE
		// `off` is a literal here (a field of the constant TableRoot).  Left to itself the compiler adds it to every lane's
		// address -- scalar base + 32-bit lane offset + a constant too large for the load's immediate field becomes a 64-bit
		// vector addition per access (v_lshl_add_u64, v_add_co_u32, v_addc_co_u32: four instructions where the generic
		// build has one).  Formed once on the scalar unit and handed on as an opaque scalar pair, the table's base is a
		// scalar register pair again and the load keeps the scalar-base + vector-offset form.
		typedef const __attribute__((address_space(1))) uint8_t* global_bytes;
		typedef const __attribute__((address_space(4))) uint8_t* constant_bytes;
		if (__builtin_constant_p(off))
		{
			uintptr_t p = (uintptr_t)wv_uniform((uint64_t)(reinterpret_cast<uintptr_t>(tab) + off));
			asm("" : "+s"(p));
			return tab_constant ? (const uint8_t*)(constant_bytes)p : (const uint8_t*)(global_bytes)p;
		}
		return tab + wv_uniform(off);
#else
		return tab + wv_uniform(off);
#endif
	}
	// typed views
	WV_FN float* data(int c) const { return reinterpret_cast<float*>(lds + L->data) + c * Tp; }
	WV_FN BlkInfo& blk() const { return *reinterpret_cast<BlkInfo*>(lds + L->blk); }
	WV_FN Scb& scb() const { return *reinterpret_cast<Scb*>(lds + L->scb); }
	WV_FN Scb& wscb() const { return *reinterpret_cast<Scb*>(lds + L->wscb); }
	WV_FN TrialInfo& tr() const { return *reinterpret_cast<TrialInfo*>(lds + L->trial); }
	WV_FN float* ei_w(int plane) const { return reinterpret_cast<float*>(lds + L->ei_w) + plane * Tp; }
	WV_FN float* ei_wes(int plane) const { return reinterpret_cast<float*>(lds + L->ei_wes) + plane * Tp; }
	// per-grid results of the current trial; `dual` = the trial has two weight planes
	WV_FN float* dwi(int dm, int plane, bool dual) const { return reinterpret_cast<float*>(lds + L->dwi) + dec_mode(dm).dwi_offset[dual ? 1 + plane : 0]; }
	WV_FN float* lowhigh(int plane, int dm, bool dual) const { return reinterpret_cast<float*>(lds + L->lowhigh) + dec_mode(dm).lowhigh_offset[dual ? 1 + plane : 0]; }
	WV_FN float* ang() const { return reinterpret_cast<float*>(lds + L->uni); }
	// sin/cos table row of every packed ideal weight (same indexing as the dwi region), see ideal_weights_all_grids
	WV_FN uint8_t* isample() const { return lds + L->modes; }
	WV_FN float* uni_f() const { return reinterpret_cast<float*>(lds + L->uni); }
	// records of the block modes [first, ...) scored by the current trial; index with the packed mode index
	WV_FN ModeRec* modes(int first) const { return reinterpret_cast<ModeRec*>(lds + L->modes) - first; }
	// texel-length scratch rows; one set per phase because the phases' regions alias each other
	WV_FN float* tsc_f(int row) const { return reinterpret_cast<float*>(lds + L->uni) + row * L->tsc_stride; }      // format search
	WV_FN float* tsc_r(int row) const { return reinterpret_cast<float*>(lds + L->tsc_r) + row * L->tsc_stride; }    // refinement
	WV_FN float* tsc_p(int row) const { return reinterpret_cast<float*>(lds + L->tsc_p) + row * L->tsc_stride; }    // partition search
	WV_FN float* wsc(int row) const { return reinterpret_cast<float*>(lds + L->wsc) + row * 64; }
	WV_FN uint8_t* fmt() const { return lds + L->uni; }
	WV_FN uint8_t* part() const { return lds + L->part; }
	WV_FN float* rsc(int row) const { return reinterpret_cast<float*>(lds + L->rsc) + row * lds_refit_stride(Tp); }
	WV_FN uint8_t* candw(int n) const { return lds + L->candw + n * 64; }

	// table accessors
	WV_FN const BlockMode& block_mode(int i) const { return table_at(reinterpret_cast<const BlockMode*>(table(root->off_block_modes)), (uint32_t)i); }
	WV_FN const DecimationMode& dec_mode(int i) const { return table_at(reinterpret_cast<const DecimationMode*>(table(root->off_decimation_modes)), (uint32_t)i); }
	WV_FN const DecimationInfo& dec_info(int i) const { return table_at(reinterpret_cast<const DecimationInfo*>(table(root->off_decimation_infos)), (uint32_t)i); }
	WV_FN const uint8_t* part_rec(int pcount, int packed) const { return table(root->off_partitions[pcount - 1]) + (uint32_t)packed * root->partition_stride; }
	WV_FN const QuantXfer& qxfer(int q) const { return table_at(reinterpret_cast<const QuantXfer*>(table(root->off_quant_xfer)), (uint32_t)q); }
};

/* Rebuild the wave's context inside an out-of-line stage function.  On the device everything comes
 * from the LDS header and the table blob, so stage functions need no context argument (arguments of
 * non-kernel functions travel in VGPRs and would make every table address look lane-variant). */
#if defined(__HIPCC__)
extern __shared__ __attribute__((aligned(16))) uint8_t astc_lds[];
#endif
#if WV_DEVICE
/* SCALAR_TABLES: the table blob, the config and the LDS layout never change while a kernel runs, so they may be read
 * through the constant address space: every read at a wave-uniform address then goes through the scalar cache instead
 * of the vector memory pipeline (shorter latency, no vector register).  The values live in scalar registers, though:
 * the two stages that run out of those (mode scoring, decimation) keep the global address space. */
/* LDS address 0 as a pointer the optimiser can see through.  The kernels use dynamic LDS only, so it starts at LDS
 * address 0 (tests/test_code_object.py checks that the static size stays 0).  Naming `astc_lds` in a function that is not a
 * kernel costs a look-up of the caller's static LDS size in a table in memory (nine scalar instructions and a load per stage
 * call), and a literal 0 cast to a pointer is the null pointer, which is -1 in this address space; 16 - 16 is neither: every
 * `lds + offset` folds into the offset field of its ds_read / ds_write.  (Until round 5 this was an opaque zero in a scalar
 * register, added to every LDS address formed.) */
WV_FN uint8_t* lds_base()
{
	typedef __attribute__((address_space(3))) uint8_t* lds_bytes;
	return (uint8_t*)((lds_bytes)(uintptr_t)16 - 16);
}

template <bool SCALAR_TABLES>
WV_FN Ctx ctx_make_as()
{
	uint8_t* const lds = lds_base();
	const LdsHeader* h = reinterpret_cast<const LdsHeader*>(lds);
	Ctx c;
	// A pointer rebuilt from integers would be a generic (flat) pointer to the compiler: go through
	// explicit address-space pointers so that table reads stay s_load / global_load.
	typedef const __attribute__((address_space(1))) uint8_t* global_bytes;
	typedef const __attribute__((address_space(4))) uint8_t* constant_bytes;
	const uintptr_t base = (uintptr_t)wv_uniform((uint64_t)reinterpret_cast<uintptr_t>(h->base));
	const uint8_t* const b = SCALAR_TABLES ? (const uint8_t*)(constant_bytes)base : (const uint8_t*)(global_bytes)base;
	c.tab = b + CTX_LAYOUT_BACK;
	c.tab_constant = SCALAR_TABLES;
#if defined(ASTC_PROFILE) || defined(ASTC_TRACE)
	typedef __attribute__((address_space(1))) unsigned long long* global_u64;
	c.prof = (unsigned long long*)(global_u64)(uintptr_t)wv_uniform((uint64_t)reinterpret_cast<uintptr_t>(h->prof));
#else
	c.prof = nullptr;
#endif
	c.lds = lds;
#if ASTC_FIXED
	c.root = &kFixedRoot;
	c.cfg = &kFixedConfig;
	c.L = &kFixedLayout;
#else
	c.root = reinterpret_cast<const TableRoot*>(c.tab);
	c.cfg = reinterpret_cast<const DeviceConfig*>(b + (CTX_LAYOUT_BACK - CTX_CONFIG_BACK));
	c.L = reinterpret_cast<const LdsLayout*>(b);
#endif
	c.T = (int)c.L->texel_count;
#if defined(ASTC_FIXED_OPAQUE_TEXEL_COUNT)
	// (run-time builds for footprints of more than 64 texels, kernel_jit.cpp: the texel count of the lane loops is NOT a
	//  compile-time constant there.  With it, the builds of the 10x8 and 12x12 footprints -- 80 and 144 texels: the last trip of
	//  a texel loop has exactly sixteen lanes -- produce other bytes than the generic build on a quarter of noisy blocks; the
	//  same source with the same constants through g++ (the sequential build, tests/test_emu_fixed.py) does not, and neither does
	//  this build with the count read back through a register.  Not root-caused: DESIGN.md section 3.1.)
	c.T = wv_uniform(wv_opaque(c.T));
#endif
	c.Tp = (c.T + 3) & ~3;
	c.Ts = lds_row_stride(c.Tp);
	return c;
}
WV_FN Ctx ctx_make() { return ctx_make_as<true>(); }
WV_FN Ctx ctx_make_vector_tables() { return ctx_make_as<false>(); }
#else
WV_FN uint8_t* lds_base() { return nullptr; }        // (host pass of a kernel translation unit: never executed)
extern thread_local const Ctx* g_wave_ctx;      // set by the CPU emulation backend around each block
WV_FN Ctx ctx_make() { return *g_wave_ctx; }
WV_FN Ctx ctx_make_vector_tables() { return *g_wave_ctx; }
#endif

/* The block's channel weights (BlkInfo::cw; ref: image_block::channel_weight).  They are the context's weights unless
 * ASTCENC_FLG_USE_ALPHA_WEIGHT scales them by the block's alpha (load_block) -- so in a fixed-context build whose context has
 * the flag clear they are literals (1, 1, 1, 1 for the BASELINE contexts: every `x * cw` then IS x, the multiplication by one
 * is exact and the compiler drops it), and everything else reads the block's record. */
WV_FN float cw_of(const BlkInfo& blk, int k)
{
#if ASTC_FIXED
	if ((kFixedConfig.flags & (1u << 2)) == 0) return k == 0 ? kFixedConfig.cw[0] : k == 1 ? kFixedConfig.cw[1] : k == 2 ? kFixedConfig.cw[2] : kFixedConfig.cw[3];
#endif
	return blk.cw[k];
}
WV_FN f4 cw4_of(const BlkInfo& blk)
{
#if ASTC_FIXED
	if ((kFixedConfig.flags & (1u << 2)) == 0) return mk4(kFixedConfig.cw[0], kFixedConfig.cw[1], kFixedConfig.cw[2], kFixedConfig.cw[3]);
#endif
	return load4(blk.cw);
}

/* for (i = l; i < T; i += 4) body(i), T = the block's texel count, l = 0 .. 3 (an accumulator lane of a quad walking its
 * texels).  With a lane-dependent start the compiler cannot tell the trip count even when T is a literal and builds a
 * divergent loop (an exec-mask exit test per trip, the LDS reads one after the other); in a fixed-context build the trips are
 * ceil(T / 4) for every l, unrolled, the tail trip masked when T is not a multiple of four. */
template <typename Body>
WV_FN void for_texels_of_quarter(int l, int T, Body body)
{
#if ASTC_FIXED
	(void)T;
	constexpr int kT = (int)kFixedRoot.texel_count, kTrips = (kT + 3) >> 2;
	#pragma unroll
	for (int trip = 0; trip < kTrips; trip++)
	{
		const int i = l + 4 * trip;
		if ((kT & 3) != 0 && i >= kT) continue;
		body(i);
	}
#else
	for (int i = l; i < T; i += 4) body(i);
#endif
}

/* Lowest index i in [0, n), n <= 64, for which pred(i) holds, or -1; the same value on every lane. */
template <typename Pred>
WV_FN int wv_find_first(int n, Pred pred)
{
#if WV_DEVICE
	const bool mine = WV_LANE < n && pred(WV_LANE);
	const unsigned long long mask = __ballot(mine);
	return mask ? (int)__builtin_ctzll(mask) : -1;
#else
	for (int i = 0; i < n; i++) if (pred(i)) return i;
	return -1;
#endif
}

/* Stage timers for profiling builds (-DASTC_PROFILE): lane 0 accumulates shader-clock cycles per
 * stage into c.prof[].  Compiled out otherwise. */
enum { PS_LOAD, PS_IDEAL, PS_DECIMATE, PS_ANGULAR, PS_MODES, PS_FORMATS, PS_RECOMPUTE, PS_PACK, PS_DIFF, PS_REALIGN,
       PS_KMEANS, PS_PSCORE, PS_PHYSICAL, PS_STATS, PS_TOTAL, PS_BLOCKS,
       // fine-grained sub-stage slots (names in backend_hip.hip)
       PS_DEC1, PS_DEC2, PS_DEC3, PS_ANG1, PS_ANG2, PS_MODE1, PS_MODE2, PS_MODE3, PS_FMT1, PS_FMT2, PS_FMT3, PS_FMT4,
       PS_X0, PS_X1, PS_X2, PS_X3,
       PS_Y0, PS_Y1, PS_Y2, PS_Y3, PS_Y4, PS_Y5, PS_Y6, PS_Y7,
       PS_COUNT };   // slots [PS_COUNT, 2 * PS_COUNT) count how often each scope was entered
#if defined(ASTC_PROFILE) && WV_DEVICE
struct ProfScope {
	unsigned long long* p; unsigned long long t0;
	__device__ ProfScope(unsigned long long* prof, int id) : p(prof ? prof + id : nullptr), t0(__builtin_amdgcn_s_memtime()) {}
	__device__ ~ProfScope() { if (p && threadIdx.x == 0) { atomicAdd(p, __builtin_amdgcn_s_memtime() - t0); atomicAdd(p + PS_COUNT, 1ull); } }
};
#define PROF_SCOPE(c, id) ProfScope prof_scope_##id((c).prof, id)
#else
#define PROF_SCOPE(c, id) ((void)0)
#endif

/* Search trace for debug builds (-DASTC_TRACE): lane 0 appends (tag, float) records to the block's slice of a
 * global buffer that the backend dumps to $ASTCENC_AMD_TRACE_FILE; tests/test_trace.py diffs it against the
 * -dtrace JSON of the reference's ASTCENC_DIAGNOSTICS build (ref: trace_add_data calls in
 * astcenc_compress_symbolic.cpp:618/:667/:952/:1002 and the pass nodes :1295-1392).  In trace builds c.prof is this
 * block's slice: word 0 counts the records, records follow as (tag, value bits) pairs.  Compiled out otherwise. */
constexpr uint32_t TRACE_WORDS_PER_BLOCK = 1024;      // 4 KiB per block: 511 records
enum { TR_PASS = 1,            // value = partition_count * 64 + plane_count * 8 + (plane 2 component + 1) as float
       TR_PARTITION_INDEX = 2, // value = partition index (the format's, not the packed table index) as float
       TR_CANDIDATE = 3,       // value = weight quant mode of the candidate as float
       TR_ERR_PRE = 4,         // error_prerealign
       TR_ERR_POST = 5,        // error_postrealign
       TR_THRESHOLD = 6,       // tune_error_threshold
       TR_LOWEST_CORREL = 7 };
#if defined(ASTC_TRACE)
WV_FN void trace_put(const Ctx& c, uint32_t tag, float value)
{
	WV_ONE
	{
		uint32_t* t = reinterpret_cast<uint32_t*>(c.prof);
		if (t)
		{
			uint32_t n = t[0];
			if (2 * n + 2 < TRACE_WORDS_PER_BLOCK)
			{
				t[1 + 2 * n] = tag;
				t[2 + 2 * n] = (uint32_t)float_as_int(value);
				t[0] = n + 1;
			}
		}
	}
}
#define TRACE_PUT(c, tag, value) trace_put(c, tag, value)
#else
#define TRACE_PUT(c, tag, value) ((void)0)
#endif

/* Instruction-count builds (-DASTC_DUPSTAGE): DeviceConfig::debug_dup_stage names one stage that is executed twice
 * (every stage listed here recomputes its outputs from unchanged inputs, so the result bytes do not change).  The
 * difference of the SQ_INSTS_* counters between a run with the stage doubled and a plain run is that stage's dynamic
 * instruction count -- a per-stage VALU / SALU / LDS / VMEM profile without PC sampling (tools/gpu_stage_counts.sh). */
enum { DUP_IDEAL = 1, DUP_DECIMATE, DUP_ANGULAR, DUP_MODES, DUP_MODES_FORMATS, DUP_CAND_QUANTIZE, DUP_CAND_SETUP, DUP_RECOMPUTE,
       DUP_PACK, DUP_DIFF, DUP_PART_ORDER, DUP_PART_SCORE, DUP_PART_SELECT, DUP_STATS, DUP_LOAD, DUP_PHYSICAL, DUP_ACCEPT,
       DUP_REALIGN, DUP_BATCH_PREPARE, DUP_BATCH_SUMS, DUP_BATCH_SOLVE, DUP_BATCH_PACK, DUP_BATCH_SCORE,
       // whole trials by class (a second run of a trial finds nothing better than its first and changes nothing)
       DUP_TRIAL_A0, DUP_TRIAL_A1, DUP_TRIAL_2PLANES, DUP_TRIAL_2PARTITIONS, DUP_TRIAL_3PARTITIONS, DUP_TRIAL_4PARTITIONS,
       DUP_REALIGN_2PLANES,          // the realignment of two-plane candidates only
       DUP_REALIGN_FIRST_PASS };     // the first evaluation of all weights inside every realignment (idempotent)
#if defined(ASTC_DUPSTAGE)
/* (read from the live context's record: in a fixed-context build c.cfg is the constant record of the named context) */
#define DUP_STAGE_ID(c) (reinterpret_cast<const DeviceConfig*>((c).tab - CTX_CONFIG_BACK)->debug_dup_stage)
#define DUP_STAGE(c, id, call) do { call; if (DUP_STAGE_ID(c) == (uint32_t)(id)) { call; } } while (0)
#else
#define DUP_STAGE(c, id, call) do { call; } while (0)
#endif

/* View of one partition record. */
struct PartView {
	const PartitionHeader* h;
	const uint8_t* of_texel;    // [T]
	const uint8_t* sorted;      // [T]  texels grouped by partition
	uint32_t offsets;           // start of each partition's run in sorted[], one byte per partition
	uint32_t counts;            // texels per partition, one byte per partition
	int pcount;
	// (packed bytes instead of int[4] arrays: a run-time partition index would put arrays in scratch)
	WV_FN int off(int p) const { return (int)((offsets >> (8 * p)) & 0xFFu); }
	WV_FN int cnt(int p) const { return (int)((counts >> (8 * p)) & 0xFFu); }
};

WV_FN PartView part_view(const Ctx& c, int pcount, int packed)
{
	PartView v;
	const uint8_t* rec = c.part_rec(pcount, packed);
	v.h = reinterpret_cast<const PartitionHeader*>(rec);
	v.of_texel = rec + sizeof(PartitionHeader);
	v.sorted = v.of_texel + c.T;
	v.pcount = pcount;
	uint32_t o = 0;
	v.offsets = 0; v.counts = 0;
	for (int i = 0; i < 4; i++)
	{
		uint32_t n = v.h->texel_count[i];
		v.offsets |= o << (8 * i);
		v.counts |= n << (8 * i);
		o += n;
	}
	return v;
}


/* Copy `words` 32-bit words global -> LDS with all lanes (coalesced), then sync. */
WV_FN void stage_words_nosync(uint8_t* lds_dst, const uint8_t* src, int words)
{
	const uint32_t* s = reinterpret_cast<const uint32_t*>(src);
	uint32_t* d = reinterpret_cast<uint32_t*>(lds_dst);
#if WV_DEVICE
	// up to 8 loads in flight per lane before the first store: one memory round trip per 512 words (and no more load
	// instructions than the copy has 64-word pieces: most copies are one or two)
	for (int base = 0; base < words; base += 512)
	{
		const int pieces = i_min(8, (words - base + 63) >> 6);        // wave-uniform
		uint32_t v[8];
		#pragma unroll
		for (int u = 0; u < 8; u++)
		{
			if (u < pieces)
			{
				int i = base + u * 64 + WV_LANE;
				v[u] = s[i < words ? i : 0];
			}
		}
		#pragma unroll
		for (int u = 0; u < 8; u++)
		{
			int i = base + u * 64 + WV_LANE;
			if (u < pieces && i < words) d[i] = v[u];
		}
	}
#else
	for (int i = 0; i < words; i++) d[i] = s[i];
#endif
}
/* The same in 16-byte pieces (both addresses 16-byte aligned; the source may be read up to 12 bytes past `bytes`, the
 * destination must hold the rounded size): a quarter of the load / store instructions of the word copy. */
WV_FN void stage_quads_nosync(uint8_t* lds_dst, const uint8_t* src, int bytes)
{
	const int quads = (bytes + 15) >> 4;
#if WV_DEVICE
	typedef uint32_t Quad __attribute__((vector_size(16)));       // (a native 128-bit value: stays in registers)
	const Quad* s = static_cast<const Quad*>(__builtin_assume_aligned(src, 16));
	Quad* d = static_cast<Quad*>(__builtin_assume_aligned(lds_dst, 16));
	// up to four loads in flight per lane, and no more load instructions than the copy has 64-quad pieces (named
	// registers, not an indexed array: a conditionally filled array would live in scratch memory)
	for (int base = 0; base < quads; base += 256)
	{
		const int n = quads - base;                                   // wave-uniform
		const int i0 = base + WV_LANE, i1 = i0 + 64, i2 = i0 + 128, i3 = i0 + 192;
		Quad q0 = table_at(s, (uint32_t)(i0 < quads ? i0 : 0)), q1 = q0, q2 = q0, q3 = q0;
		if (n > 64) q1 = table_at(s, (uint32_t)(i1 < quads ? i1 : 0));
		if (n > 128) q2 = table_at(s, (uint32_t)(i2 < quads ? i2 : 0));
		if (n > 192) q3 = table_at(s, (uint32_t)(i3 < quads ? i3 : 0));
		if (i0 < quads) d[i0] = q0;
		if (n > 64 && i1 < quads) d[i1] = q1;
		if (n > 128 && i2 < quads) d[i2] = q2;
		if (n > 192 && i3 < quads) d[i3] = q3;
	}
#else
	__builtin_memcpy(lds_dst, src, (size_t)quads * 16);
#endif
}

WV_FN void stage_words(uint8_t* lds_dst, const uint8_t* src, int words)
{
	stage_words_nosync(lds_dst, src, words);
	WV_SYNC();
}

/* Partition view whose texel arrays are staged in LDS: every trial walks them in serial loops, and
 * an L2 round trip per element is what those loops would otherwise pay. */
WV_FN PartView part_view_staged(const Ctx& c, int pcount, int packed)
{
	PartView v = part_view(c, pcount, packed);
	uint8_t* dst = c.lds + c.L->ptab;
	// of_texel[T] and sorted[T] are adjacent in the record; record stride is a multiple of 4
	const uint8_t* src = reinterpret_cast<const uint8_t*>(v.h) + sizeof(PartitionHeader);
	WV_ONE { c.tr().part_offsets = v.offsets; c.tr().part_counts = v.counts; }
	stage_words(dst, src, (2 * c.T + 3) / 4);
	v.of_texel = dst;
	v.sorted = dst + c.T;
	return v;
}

/* The same view without copying: for stage functions that run after part_view_staged() did (no global memory access). */
WV_FN PartView part_view_lds(const Ctx& c, int pcount, int packed)
{
	PartView v;
	const uint8_t* dst = c.lds + c.L->ptab;
	v.h = reinterpret_cast<const PartitionHeader*>(c.part_rec(pcount, packed));     // (address only; not dereferenced on the hot paths)
	v.of_texel = dst;
	v.sorted = dst + c.T;
	v.offsets = wv_uniform(c.tr().part_offsets);
	v.counts = wv_uniform(c.tr().part_counts);
	v.pcount = pcount;
	return v;
}

/* Pointer view of one decimation mode's tables (global or staged in LDS). */
struct DecView {
	int T, W, max_texel_weight_count, rows;
	const uint8_t* tw;     // [4][T]
	const uint8_t* tci;    // [4][T]
	const float*   tcf;    // [4][T]
	const uint8_t* wtc;    // [W]
	const uint8_t* wt;     // [rows][W]
	const float*   wc;     // [rows][W]
	const float*   tcw;    // [rows][W]
	const uint8_t* ro;     // [W]  realign schedule: weights in processing order
	const uint8_t* rc;     // [levels] weights per group
	int levels;
	int slots;             // most weights evaluated at once (LDS rows, lanes)
	const uint8_t* later;  // [W][REALIGN_LATER_MAX] later neighbours (global memory), or null: use the level schedule
};

WV_FN DecView dec_view_at(const DecimationInfo& di, const uint8_t* base /* address of the texel_weights array */)
{
	DecView v;
	v.T = di.texel_count; v.W = di.weight_count; v.max_texel_weight_count = di.max_texel_weight_count;
	v.rows = di.max_weight_texel_count;
	v.tw = base;
	v.tci = base + (di.off_texel_contribs_int - di.off_texel_weights);
	v.tcf = reinterpret_cast<const float*>(base + (di.off_texel_contribs_f - di.off_texel_weights));
	v.wtc = base + (di.off_weight_texel_count - di.off_texel_weights);
	v.wt = base + (di.off_weight_texels - di.off_texel_weights);
	v.wc = reinterpret_cast<const float*>(base + (di.off_weight_contribs - di.off_texel_weights));
	v.tcw = reinterpret_cast<const float*>(base + (di.off_texel_contrib_for_weight - di.off_texel_weights));
	v.ro = base + (di.off_realign_order - di.off_texel_weights);
	v.rc = base + (di.off_realign_counts - di.off_texel_weights);
	v.levels = di.realign_levels;
	v.slots = di.realign_slots;
	v.later = nullptr;
	return v;
}

WV_FN DecView dec_view_global(const Ctx& c, int dm)
{
	const DecimationInfo& di = c.dec_info(dm);
	return dec_view_at(di, c.table(di.off_texel_weights));
}

/* The staged view (after refine_candidate_restore staged the grid): sizes, offsets and tables all come from LDS. */
WV_FN DecView dec_view_lds(const Ctx& c, int dm)
{
	(void)dm;
	const DecimationInfo& di = *reinterpret_cast<const DecimationInfo*>(c.lds + c.L->dtab);
	DecView v = dec_view_at(di, c.lds + c.L->dtab + DTAB_RECORD_BYTES);
	if (di.realign_speculative) v.later = c.table(di.off_realign_later);
	return v;
}

} } // namespace astcd::ASTC_VARIANT
