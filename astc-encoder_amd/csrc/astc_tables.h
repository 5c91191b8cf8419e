// SPDX-License-Identifier: Apache-2.0
// Flat, pointer-free table layout shared by the host table builder (host_tables.cpp) and the
// wavefront block compressor (wave_*.h).  Everything the kernels read lives in ONE byte blob
// that is uploaded once per context; records refer to their arrays by byte offset into the blob.
//
// The reference keeps the same information in block_size_descriptor (Source/astcenc_internal.h:533,
// 14.7 MB, fixed 216x64 strides).  Here arrays are sized to the real texel/weight counts and only
// the entries the compressor can select are kept (a few hundred KB, L2 resident).
#pragma once
#if !defined(__HIPCC_RTC__)
#include <stdint.h>
#include <stddef.h>
#else
// The run-time compiler (hipRTC, kernel_jit.cpp) has no C library headers: the fixed-width types by hand.
typedef unsigned char uint8_t; typedef signed char int8_t; typedef unsigned short uint16_t; typedef short int16_t;
typedef unsigned int uint32_t; typedef int int32_t; typedef unsigned long uint64_t; typedef long int64_t;
typedef unsigned long size_t; typedef unsigned long uintptr_t; typedef long ptrdiff_t;
#define offsetof(type, member) __builtin_offsetof(type, member)
#endif

#if defined(__HIPCC__)
	#define ASTC_HD __host__ __device__
#else
	#define ASTC_HD
#endif

namespace astcd {

constexpr int MAX_TEXELS       = 216;  // largest footprint (6x6x6)        ref: BLOCK_MAX_TEXELS astcenc_internal.h:68
constexpr int MAX_WEIGHTS      = 64;   // ref: BLOCK_MAX_WEIGHTS           astcenc_internal.h:88
constexpr int PLANE2_OFFSET    = 32;   // ref: WEIGHTS_PLANE2_OFFSET       astcenc_internal.h:109
constexpr int MAX_PARTITIONS   = 4;    // ref: BLOCK_MAX_PARTITIONS        astcenc_internal.h:79
constexpr int MAX_PARTITIONINGS = 1024;
constexpr int MAX_KMEANS_TEXELS = 64;  // ref: BLOCK_MAX_KMEANS_TEXELS     astcenc_internal.h:85
constexpr int MAX_BLOCK_MODES  = 2048; // ref: WEIGHTS_MAX_BLOCK_MODES     astcenc_internal.h:115
constexpr int MAX_DECIMATION_MODES = 87;
constexpr int MAX_TRIAL_CANDIDATES = 8;       // ref: TUNE_MAX_TRIAL_CANDIDATES
constexpr int MAX_PARTITIONING_CANDIDATES = 8; // ref: TUNE_MAX_PARTITIONING_CANDIDATES
constexpr int MAX_ANGULAR_QUANT = 7;          // ref: TUNE_MAX_ANGULAR_QUANT (QUANT_12)
constexpr int ANGULAR_STEPS = 32;             // ref: astcenc_weight_align.cpp:48
constexpr int SINCOS_STEPS  = 64;             // ref: astcenc_weight_align.cpp:58
constexpr float ERROR_CALC_DEFAULT = 1e30f;   // ref: astcenc_internal.h:121

// Endpoint formats (ref: enum endpoint_formats, astcenc_internal.h:179)
enum {
	FMT_LUMINANCE = 0, FMT_LUMINANCE_DELTA = 1, FMT_HDR_LUMINANCE_LARGE_RANGE = 2,
	FMT_HDR_LUMINANCE_SMALL_RANGE = 3, FMT_LUMINANCE_ALPHA = 4, FMT_LUMINANCE_ALPHA_DELTA = 5,
	FMT_RGB_SCALE = 6, FMT_HDR_RGB_SCALE = 7, FMT_RGB = 8, FMT_RGB_DELTA = 9,
	FMT_RGB_SCALE_ALPHA = 10, FMT_HDR_RGB = 11, FMT_RGBA = 12, FMT_RGBA_DELTA = 13,
	FMT_HDR_RGB_LDR_ALPHA = 14, FMT_HDR_RGBA = 15
};

// Quantization methods (ref: enum quant_method, astcenc_internal.h:204)
enum {
	QUANT_2 = 0, QUANT_3, QUANT_4, QUANT_5, QUANT_6, QUANT_8, QUANT_10, QUANT_12, QUANT_16,
	QUANT_20, QUANT_24, QUANT_32, QUANT_40, QUANT_48, QUANT_64, QUANT_80, QUANT_96, QUANT_128,
	QUANT_160, QUANT_192, QUANT_256
};

// Size of the phase-shared `uni` LDS region (see make_lds_layout, wave_ctx.h).  It depends only on the footprint
// and the partition-count limit, so the table builder can cut the decimation sweeps into the same chunks the
// kernel will use.
constexpr uint32_t MODE_DESC_BYTES = 16 + 2 * 32;   // ModeHdr + ModeQ[2], see score_block_modes (wave_block.h)
constexpr uint32_t MODE_WEIGHT_BYTES = 68;          // quantized weights of one block mode (second plane at + 32); 17 words: the quads of a chunk's modes read their records side by side, in different LDS banks
constexpr uint32_t MODE_Q2U_BYTES = 12 * 32;        // mode scoring's LDS copy of the quant_to_unquant rows of the weight quant levels (tail of `uni`)
constexpr uint32_t FMT_QUANT_ROWS = 17;
ASTC_HD inline uint32_t fmt_comb_cols(uint32_t partition_limit) { return partition_limit <= 1 ? 0u : partition_limit == 2 ? 7u : partition_limit == 3 ? 10u : 13u; }
ASTC_HD inline uint32_t fmt_scratch_bytes(uint32_t partition_limit)
{
	uint32_t P = partition_limit < 1 ? 1u : partition_limit > 4 ? 4u : partition_limit;
	// best_error f32 [P][17][4], format_of_choice u8 [P][17][4]; comb_error f32 [17][cols], comb_format u16 [17][cols] (four 4-bit formats)
	return P * FMT_QUANT_ROWS * 4 * 4 + P * FMT_QUANT_ROWS * 4 + FMT_QUANT_ROWS * fmt_comb_cols(P) * (4 + 2);
}
// floats per (grid, step) record of the angular search's batch (six used; nine: an odd stride, so that the lanes of a
// batch -- one record each -- do not all land in the same four LDS banks)
constexpr uint32_t ANG_PAIR_STRIDE = 9;
ASTC_HD inline uint32_t uni_region_bytes(uint32_t texel_count, uint32_t partition_limit)
{
	const uint32_t Tp = (texel_count + 3u) & ~3u;
	uint32_t bytes = 64 * ANG_PAIR_STRIDE * 4;                     // angular batch
	if (8u * (MODE_DESC_BYTES + 64u + Tp * 4) > bytes) bytes = 8u * (MODE_DESC_BYTES + 64u + Tp * 4);   // (what mode scoring needed while it kept a row of texel terms per mode; it takes less now -- make_lds_layout fits the modes per chunk to the region -- but the decimation sweeps' sets per chunk are sized by this too)
	if (fmt_scratch_bytes(partition_limit) > bytes) bytes = fmt_scratch_bytes(partition_limit);
	if (5 * Tp * 4 > bytes) bytes = 5 * Tp * 4;                    // encoding-choice rows
	return (bytes + 15u) & ~15u;
}

// Symbolic block types (ref: astcenc_internal.h:1059-1068)
enum { SYM_BTYPE_ERROR = 0, SYM_BTYPE_CONST_F16 = 1, SYM_BTYPE_CONST_U16 = 2, SYM_BTYPE_NONCONST = 3 };

// One legal (weight grid, weight quant, plane count) combination. (ref: struct block_mode :418)
struct BlockMode {
	uint16_t mode_index;      // the 11-bit value stored in the physical block
	uint8_t  decimation_mode; // index into DecimationMode / DecimationInfo arrays
	uint8_t  quant_mode;      // weight quant_method
	uint8_t  weight_bits;
	uint8_t  is_dual_plane;
	uint8_t  pad[2];
};

// What scoring a block mode needs to know about it, gathered from its BlockMode, DecimationMode and DecimationInfo records
// into one 24-byte record (TableRoot::off_mode_static): the mode lanes of score_block_modes then start with one load
// instead of a chain of three dependent ones.
struct ModeStatic {
	uint32_t tw_off;          // DecimationInfo::off_texel_weights
	uint32_t tcf_off;         // DecimationInfo::off_texel_contribs_f
	uint16_t dwi_off[2];      // packed ideal-weight slot of plane 0 / 1 in the mode's trial class
	uint16_t lh_off[2];       // float offset of the mode's (low, high) pair per plane in the angular bounds; 0xFFFF: quant level above QUANT_12
	uint8_t  taps;            // 1, 2 or 4 grid weights per texel
	uint8_t  weights;         // per plane
	uint8_t  quant_mode;
	uint8_t  weight_bits;
	uint8_t  is_dual_plane;
	uint8_t  pad[3];
};

// (ref: struct decimation_mode :449)
struct DecimationMode {
	int8_t   maxprec_1plane;
	int8_t   maxprec_2planes;
	uint16_t refprec_1plane;
	uint16_t refprec_2planes;
	// LDS slots of this grid's per-trial results.  1-plane and 2-plane trials never run at the same
	// time, so each trial class has its own dense packing: slot 0 = the plane of a 1-plane trial,
	// slots 1 / 2 = plane 0 / 1 of a 2-plane trial.  Within a class the grids are packed by ascending
	// lowest quant level of their block modes (TableRoot::dwi_used_sets); grids the class cannot use have no slot.
	uint16_t dwi_offset[3];      // float offset of the ideal weights in the packed dwi region
	uint16_t lowhigh_offset[3];  // float offset of the angular (low, high) pairs: one per quant level <= QUANT_12 in refprec, in ascending order
};

// Everything the decimation sweeps need about one packed ideal-weight slot (TableRoot::off_dwi_slots):
// one 16-byte load per lane instead of walking owner -> decimation mode -> decimation info.
// The taps of a weight -- weight_texels[j][i], weight_contribs[j][i] -- two bytes each (texel index, the contribution: an
// integer 0 .. 16), eight to a 16-byte group: tap k of a group = bytes 2k, 2k + 1.  The last group of a weight is padded
// with (texel 0, contribution 0), which adds +0.0 to every sum of the sweeps.
struct DwiTap8 { uint32_t w[4]; };
struct DwiSlot {
	uint32_t wt_off;        // blob offset of the weight's DwiTap8 groups, taps in order, 16-byte aligned: a sweep fetches eight taps
	                        // with one 128-bit load
	uint32_t wc_off;        // (unused)
	uint16_t refprec;       // quant levels (bit mask) of the block modes using this grid in this trial class
	uint8_t  weight_count;
	uint8_t  taps;          // texels this weight touches; 0 for the padding slots past weight_count
	uint8_t  flags;         // bit 0: grid == texels, the ideal weight is copied; bit 1: weight plane
	uint8_t  dm;            // decimation mode
	uint8_t  set;           // index of the (grid, plane) set in packing order (InfillSet index)
	uint8_t  index;         // weight index in the grid
};

// Processing order of the decimation sweeps (TableRoot::off_dwi_order): for every weight quant limit q of a trial,
// the live slots of the sets that trial uses (a prefix of the packing, TableRoot::dwi_used_sets), cut into the
// chunks of sets whose texel-resolution infill fits the scratch region together (dwi_sets_per_chunk) and, inside a
// chunk, sorted by descending tap count: the 64 lanes of a sweep iteration then walk tap lists of similar
// length instead of waiting for the longest one in packing order.  Copied ("direct") slots come last.
constexpr int DWI_MAX_CHUNKS = 15;
struct DwiOrderDir {
	uint32_t list_off;                       // blob offset of the DwiSlot records of this (class, q) in processing order; DwiSlot::refprec = the slot's packed index there
	uint16_t chunk_start[DWI_MAX_CHUNKS + 1];  // list positions [chunk_start[c], chunk_start[c + 1]) belong to chunk c
	uint16_t chunks;
	uint16_t pad;
};

// The same for the texel-resolution infill of one (grid, plane) of a trial class (TableRoot::off_infill_sets).
struct InfillSet {
	uint32_t tw_off;        // blob offset of the grid's per-texel index words (DecimationInfo::off_texel_weights)
	uint32_t tcf_off;       // blob offset of the grid's per-texel contributions (DecimationInfo::off_texel_contribs_f)
	uint16_t dwi_offset;    // float offset of the grid's ideal weights in the packed region
	uint16_t refprec;
	uint8_t  taps;          // 1, 2 or 4 weights per texel
	uint8_t  direct;
	uint8_t  dm;
	uint8_t  plane;
};

// Bilinear-infill tables of one weight grid. (ref: struct decimation_info :347)
// Arrays are [row][T] or [row][W] with row stride = texel_count / weight_count.
struct DecimationInfo {
	uint8_t  texel_count;
	uint8_t  weight_count;
	uint8_t  max_texel_weight_count;
	uint8_t  weight_x;
	uint8_t  weight_y;
	uint8_t  max_weight_texel_count;     // rows in the per-weight arrays
	uint8_t  realign_levels;             // number of groups in the realign schedule below
	uint8_t  realign_slots;              // most weights per group for this grid (lanes: slots * rows rounded to 4 <= 64)
	// one record per texel: its (up to) four grid weights and their contributions side by side, so that a lane fetches them
	// with one 32-bit / 128-bit access (the reference keeps them transposed, [4][T], for its SIMD gathers)
	uint32_t off_texel_weights;          // u8  [T][4]   ref: texel_weights_tr[j][t] at [t][j]; 16-byte aligned
	uint32_t off_texel_contribs_int;     // u8  [T][4]   ref: texel_weight_contribs_int_tr
	uint32_t off_texel_contribs_f;       // f32 [T][4]   ref: texel_weight_contribs_float_tr; 16-byte aligned
	uint32_t off_weight_texel_count;     // u8  [W]      ref: weight_texel_count
	uint32_t off_weight_texels;          // u8  [rows][W] ref: weight_texels_tr
	uint32_t off_weight_contribs;        // f32 [rows][W] ref: weights_texel_contribs_tr
	uint32_t off_texel_contrib_for_weight; // f32 [rows][W] ref: texel_contrib_for_weight
	// Realign schedule: the weights in processing order, cut into groups of weights that share no
	// texel (so a group can be evaluated at once) while every pair of weights that does share a texel
	// keeps its index order across groups -- equivalent to the reference's one-by-one sweep.
	uint32_t off_realign_order;          // u8 [W]      weight indices
	uint32_t off_realign_counts;         // u8 [W]      weights per group, realign_levels entries used
	uint32_t table_bytes;                // all nine arrays are contiguous from off_texel_weights
	// Grids whose schedule is long (weights decimated in two dimensions: every weight shares texels with its eight
	// neighbours, so the groups hold one to three weights) are realigned speculatively instead: every weight is
	// evaluated as if no weight before it moved -- all of them lane-parallel -- and after each weight that does move
	// (few do) only the later weights that share a texel with it, listed here, are evaluated again.
	uint32_t off_realign_later;          // u8 [W][REALIGN_LATER_MAX] later neighbours of each weight, 255-terminated (not staged in LDS)
	uint32_t realign_speculative;        // 1: use the scheme above
	uint32_t pad[2];
};
static_assert(sizeof(DecimationInfo) == 64, "DecimationInfo is staged as one 64-byte record");
constexpr int REALIGN_LATER_MAX = 16;
// (measured on MI355X, same-call A/B: with the lane-per-weight evaluator of realign_weights the speculative scheme wins
//  for every decimated grid -- 6x6 -medium 124.9 -> 127.8 Mtexels/s going from 7 to 2 -- so the level schedule is left
//  for the grids whose later-neighbour lists do not fit, i.e. some 3D grids)
#ifndef ASTC_REALIGN_SPEC_MIN
#define ASTC_REALIGN_SPEC_MIN 2
#endif
constexpr int REALIGN_SPECULATIVE_MIN_LEVELS = ASTC_REALIGN_SPEC_MIN;   // schedules at least this long are replaced by the speculative scheme

// One partitioning; fixed-stride record followed by two u8[T] arrays:
//   partition_of_texel[T], texels_sorted[T] (texel indices grouped by partition, ascending inside
//   each group == ref texels_of_partition[p][0..count) laid end to end).  (ref: partition_info :313)
struct PartitionHeader {
	uint16_t partition_index;    // the 10-bit seed stored in the physical block
	uint8_t  partition_count;
	uint8_t  pad;
	uint8_t  texel_count[4];     // texels per partition
};

// Weight quantization transfer table. (ref: quant_and_transfer_table :1036)
struct QuantXfer {
	uint8_t  quant_to_unquant[32];
	uint8_t  scramble_map[32];
	uint16_t prev_next_values[65];
	uint16_t pad;
};

// The device copy of the blob is preceded by two 256-byte records of the owning context, so that a
// kernel or stage function that knows the blob pointer reaches them without a dependent load:
//   blob - CTX_LAYOUT_BACK : LdsLayout      blob - CTX_CONFIG_BACK : DeviceConfig
constexpr uint32_t CTX_CONFIG_BACK = 256;
constexpr uint32_t CTX_LAYOUT_BACK = 1024;

// Root record at blob offset 0.
struct TableRoot {
	uint8_t  dim_x, dim_y, texel_count, dim_z;
	uint32_t block_mode_count_1plane_always;
	uint32_t block_mode_count_1plane_selected;
	uint32_t block_mode_count_1plane_2plane_selected;
	uint32_t decimation_mode_count_always;
	uint32_t decimation_mode_count_selected;
	uint32_t partitioning_count_selected[4];  // [partition_count - 1]
	uint32_t partition_stride;                // bytes per partition record
	uint32_t off_block_modes;                 // BlockMode[]
	uint32_t off_decimation_modes;            // DecimationMode[]
	uint32_t off_decimation_infos;            // DecimationInfo[]
	uint32_t off_partitions[4];               // [partition_count - 1] -> records
	uint32_t off_coverage[4];                 // [partition_count - 1] -> u64[count][partition_count]
	uint32_t off_kmeans_texels;               // u8[min(T,64)]
	uint32_t off_color_unquant_to_uquant;     // u8[17][512]
	uint32_t off_color_uquant_to_pquant;      // u8[17][256]
	uint32_t off_quant_xfer;                  // QuantXfer[12]
	uint32_t off_quant_mode_table;            // i8[10][128]
	uint32_t off_quant_mode_by_bits;          // i8[128][16]: the same table transposed, one 16-byte row per bit budget
	uint32_t off_mode_levels;                 // i8[4][block_mode_count_1plane_2plane_selected][16]: that row for every (partition count, block mode)
	uint32_t off_integer_of_trits;            // u8[243]  index ((((t4*3+t3)*3+t2)*3+t1)*3+t0)
	uint32_t off_integer_of_quints;           // u8[125]  index ((q2*5+q1)*5+q0)
	uint32_t off_sin_table;                   // f32[64][32]
	uint32_t off_cos_table;                   // f32[64][32]
	uint32_t off_cos_sin_table;               // f32[65][32][2]: (cos, sin) side by side, one 64-bit load per (row, step) in the angular search; row 64 is all +0.0
	uint32_t off_dm_by_weights;               // u8[decimation_mode_count_selected]: the grids by descending weight count (angular batching order)
	uint32_t off_mode_static;                 // ModeStatic[block_mode_count_1plane_2plane_selected]
	uint32_t max_decimation_table_bytes;      // largest DecimationInfo::table_bytes (LDS staging size)
	uint32_t max_weight_texel_rows;           // largest DecimationInfo::max_weight_texel_count
	uint32_t realign_rt_floats;               // LDS floats the realign term rows need: max over grids of slots * 12 * rows4
	uint32_t max_weights[2];                  // largest weight count per plane among the grids of [1-plane, 2-plane] trials
	uint32_t dwi_total_floats[2];             // size of the packed ideal-weight region, [1-plane trials, 2-plane trials]
	uint32_t off_dwi_owner[2];                // u16[dwi_total_floats[class]]: (decimation mode << 1) | plane owning each packed slot
	uint32_t off_dwi_slots[2];                // DwiSlot[dwi_total_floats[class]]
	uint32_t off_infill_sets[2];              // InfillSet[dwi_sets[class]] in packing order
	uint32_t dwi_sets[2];                     // (grid, plane) sets per class; packed by ascending lowest usable quant level,
	uint32_t dwi_used_sets[2][12];            // so the sets a trial with weight quant limit q uses are the first [class][q]
	uint32_t lowhigh_floats[2];               // size of the packed low/high region per trial class
	uint32_t off_dwi_order[2];                // DwiOrderDir[12] per trial class, indexed by the trial's weight quant limit
	uint32_t dwi_sets_per_chunk;              // (grid, plane) sets whose infill fits the `uni` LDS region at once
	uint32_t max_partitionings;               // largest partitioning_count_selected[1..3]
	uint32_t total_bytes;
};

// Per-context scalars consumed by the kernels. (ref: astcenc_config + derived values)
struct DeviceConfig {
	int32_t  profile;
	uint32_t flags;
	float    cw[4];
	float    rgbm_m_scale;
	uint32_t tune_partition_count_limit;
	uint32_t tune_partition_index_limit[3];       // 2,3,4 partitions
	uint32_t tune_refinement_limit;
	uint32_t tune_candidate_limit;
	uint32_t tune_partitioning_candidate_limit[3];
	float    tune_db_limit;                        // already converted (ref: astcenc_entry.cpp:816)
	float    tune_mse_overshoot;
	float    tune_partition_early_out_limit_factor[2]; // 2,3 partitions
	float    tune_2plane_early_out_limit_correlation;
	float    tune_search_mode0_enable;
	uint32_t debug_dup_stage;                      // instruction-count builds only (-DASTC_DUPSTAGE, tools/gpu_stage_counts.sh): stage to run twice
};

// One image (or image slice) handed to the kernel.
struct ImageDesc {
	const void* data;      // device pointer, tightly packed RGBA rows, dim_z slices back to back
	uint32_t dim_x, dim_y; // texels
	uint32_t data_type;    // astcenc_type
	uint32_t swz[4];       // astcenc_swz per output channel
	uint32_t blocks_x, blocks_y;
	uint32_t use_fast_load; // ref: astcenc_entry.cpp:946
	const float* alpha_avg; // per-texel alpha averages of the a_scale_radius pre-pass, or null
	uint32_t a_scale_radius;
	uint32_t dim_z, blocks_z; // slices of a volume / 2D array image (1 for a plain 2D image)
	uint32_t fast_load_slice0; // reference behaviour of the RGBA8 fast loader on a multi-slice image: every slice's
	                           // blocks are read from slice 0 (ref: astcenc_image.cpp:304); 0 = read the block's own slice
};

} // namespace astcd
