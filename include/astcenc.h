/* SPDX-License-Identifier: Apache-2.0
 *
 * astcenc.h -- C ABI of the MI355X-native ASTC encoder (libastcenc_amd.so).
 *
 * This header is the drop-in boundary.  It declares exactly the ten entry points the
 * reference library exports, with struct layouts, enum values and argument meaning kept
 * ABI-identical so that a caller built against the reference header links and runs
 * against this library unchanged.  Each declaration cites the reference declaration it
 * replaces (paths relative to the astcenc 5.x source tree, Source/astcenc.h).
 *
 * What differs from the reference: astcenc_compress_image() runs the per-block
 * compressor as HIP kernels on the current device; the caller-side threading contract
 * (N callers with distinct thread_index, all return when the image is complete) is kept,
 * the first caller drives the GPU and the others wait on the same completion.
 */
#ifndef ASTCENC_INCLUDED
#define ASTCENC_INCLUDED

#if defined(__cplusplus)
	#include <cstddef>
	#include <cstdint>
	#define ASTCENC_EXTERN_C extern "C"
#else
	#include <stddef.h>
	#include <stdint.h>
	#include <stdbool.h>
	#define ASTCENC_EXTERN_C
#endif

#if defined(ASTCENC_DYNAMIC_LIBRARY)
	#define ASTCENC_PUBLIC ASTCENC_EXTERN_C __attribute__ ((visibility ("default")))
#else
	#define ASTCENC_PUBLIC ASTCENC_EXTERN_C
#endif

/* Opaque codec context. (ref: astcenc.h:205) */
struct astcenc_context;

/* Return codes. (ref: astcenc.h:207-236; same numeric values) */
enum astcenc_error {
	ASTCENC_SUCCESS = 0,
	ASTCENC_ERR_OUT_OF_MEM,
	ASTCENC_ERR_BAD_CPU_FLOAT,
	ASTCENC_ERR_BAD_PARAM,
	ASTCENC_ERR_BAD_BLOCK_SIZE,
	ASTCENC_ERR_BAD_PROFILE,
	ASTCENC_ERR_BAD_QUALITY,
	ASTCENC_ERR_BAD_SWIZZLE,
	ASTCENC_ERR_BAD_FLAGS,
	ASTCENC_ERR_BAD_CONTEXT,
	ASTCENC_ERR_NOT_IMPLEMENTED,
	ASTCENC_ERR_BAD_DECODE_MODE
};

/* Color profiles. (ref: astcenc.h:241-251) */
enum astcenc_profile {
	ASTCENC_PRF_LDR_SRGB = 0,
	ASTCENC_PRF_LDR,
	ASTCENC_PRF_HDR_RGB_LDR_A,
	ASTCENC_PRF_HDR
};

/* Quality presets; any float in [0,100] is legal and interpolates. (ref: astcenc.h:254-269) */
static const float ASTCENC_PRE_FASTEST = 0.0f;
static const float ASTCENC_PRE_FAST = 10.0f;
static const float ASTCENC_PRE_MEDIUM = 60.0f;
static const float ASTCENC_PRE_THOROUGH = 98.0f;
static const float ASTCENC_PRE_VERYTHOROUGH = 99.0f;
static const float ASTCENC_PRE_EXHAUSTIVE = 100.0f;

/* Component selectors for swizzles. (ref: astcenc.h:274-289) */
enum astcenc_swz {
	ASTCENC_SWZ_R = 0,
	ASTCENC_SWZ_G = 1,
	ASTCENC_SWZ_B = 2,
	ASTCENC_SWZ_A = 3,
	ASTCENC_SWZ_0 = 4,
	ASTCENC_SWZ_1 = 5,
	ASTCENC_SWZ_Z = 6
};

/* (ref: astcenc.h:294-304) */
struct astcenc_swizzle {
	enum astcenc_swz r;
	enum astcenc_swz g;
	enum astcenc_swz b;
	enum astcenc_swz a;
};

/* Texel component storage type. (ref: astcenc.h:309-317) */
enum astcenc_type {
	ASTCENC_TYPE_U8 = 0,
	ASTCENC_TYPE_F16 = 1,
	ASTCENC_TYPE_F32 = 2
};

/* Progress callback, percentage 0..100. (ref: astcenc.h:322) */
ASTCENC_EXTERN_C typedef void (*astcenc_progress_callback)(float);

/* Flag bits. (ref: astcenc.h:332-415) */
static const unsigned int ASTCENC_FLG_MAP_NORMAL           = 1 << 0;
static const unsigned int ASTCENC_FLG_USE_DECODE_UNORM8    = 1 << 1;
static const unsigned int ASTCENC_FLG_USE_ALPHA_WEIGHT     = 1 << 2;
static const unsigned int ASTCENC_FLG_USE_PERCEPTUAL       = 1 << 3;
static const unsigned int ASTCENC_FLG_DECOMPRESS_ONLY      = 1 << 4;
static const unsigned int ASTCENC_FLG_SELF_DECOMPRESS_ONLY = 1 << 5;
static const unsigned int ASTCENC_FLG_MAP_RGBM             = 1 << 6;

static const unsigned int ASTCENC_ALL_FLAGS =
	ASTCENC_FLG_MAP_NORMAL | ASTCENC_FLG_MAP_RGBM | ASTCENC_FLG_USE_ALPHA_WEIGHT |
	ASTCENC_FLG_USE_PERCEPTUAL | ASTCENC_FLG_USE_DECODE_UNORM8 |
	ASTCENC_FLG_DECOMPRESS_ONLY | ASTCENC_FLG_SELF_DECOMPRESS_ONLY;

/* Codec configuration; field order/size is ABI. (ref: astcenc.h:427-605) */
struct astcenc_config {
	enum astcenc_profile profile;
	unsigned int flags;
	unsigned int block_x;
	unsigned int block_y;
	unsigned int block_z;
	float cw_r_weight;
	float cw_g_weight;
	float cw_b_weight;
	float cw_a_weight;
	unsigned int a_scale_radius;
	float rgbm_m_scale;
	unsigned int tune_partition_count_limit;
	unsigned int tune_2partition_index_limit;
	unsigned int tune_3partition_index_limit;
	unsigned int tune_4partition_index_limit;
	unsigned int tune_block_mode_limit;
	unsigned int tune_refinement_limit;
	unsigned int tune_candidate_limit;
	unsigned int tune_2partitioning_candidate_limit;
	unsigned int tune_3partitioning_candidate_limit;
	unsigned int tune_4partitioning_candidate_limit;
	float tune_db_limit;
	float tune_mse_overshoot;
	float tune_2partition_early_out_limit_factor;
	float tune_3partition_early_out_limit_factor;
	float tune_2plane_early_out_limit_correlation;
	float tune_search_mode0_enable;
	astcenc_progress_callback progress_callback;
};

/* Uncompressed image: dim_z slice pointers of tightly packed RGBA. (ref: astcenc.h:613-629) */
struct astcenc_image {
	unsigned int dim_x;
	unsigned int dim_y;
	unsigned int dim_z;
	enum astcenc_type data_type;
	void** data;
};

/* Block metadata query result. (ref: astcenc.h:637-704) */
struct astcenc_block_info {
	enum astcenc_profile profile;
	unsigned int block_x;
	unsigned int block_y;
	unsigned int block_z;
	unsigned int texel_count;
	bool is_error_block;
	bool is_constant_block;
	bool is_hdr_block;
	bool is_dual_plane_block;
	unsigned int partition_count;
	unsigned int partition_index;
	unsigned int dual_plane_component;
	unsigned int color_endpoint_modes[4];
	unsigned int color_level_count;
	unsigned int weight_level_count;
	unsigned int weight_x;
	unsigned int weight_y;
	unsigned int weight_z;
	float color_endpoints[4][2][4];
	float weight_values_plane1[216];
	float weight_values_plane2[216];
	uint8_t partition_assignment[216];
};

/* Fill a config from (profile, block size, quality preset, flags). (ref: astcenc.h:725) */
ASTCENC_PUBLIC enum astcenc_error astcenc_config_init(
	enum astcenc_profile profile,
	unsigned int block_x, unsigned int block_y, unsigned int block_z,
	float quality, unsigned int flags,
	struct astcenc_config* config);

/* Create a context: validates the config, builds the block-size tables on the host and
 * uploads them to the current HIP device. Exactly one of config / parent_context must be
 * given. (ref: astcenc.h:761) */
ASTCENC_PUBLIC enum astcenc_error astcenc_context_alloc(
	const struct astcenc_config* config,
	unsigned int thread_count,
	struct astcenc_context** context,
	const struct astcenc_context* parent_context);

/* Compress a host image into data_out (16 bytes per block, raster block order).
 * Must be called by every one of the context's thread_count threads. (ref: astcenc.h:785) */
ASTCENC_PUBLIC enum astcenc_error astcenc_compress_image(
	struct astcenc_context* context,
	struct astcenc_image* image,
	const struct astcenc_swizzle* swizzle,
	uint8_t* data_out, size_t data_len,
	unsigned int thread_index);

/* (ref: astcenc.h:806) */
ASTCENC_PUBLIC enum astcenc_error astcenc_compress_reset(struct astcenc_context* context);

/* (ref: astcenc.h:820) */
ASTCENC_PUBLIC enum astcenc_error astcenc_compress_cancel(struct astcenc_context* context);

/* (ref: astcenc.h:835) */
ASTCENC_PUBLIC enum astcenc_error astcenc_decompress_image(
	struct astcenc_context* context,
	const uint8_t* data, size_t data_len,
	struct astcenc_image* image_out,
	const struct astcenc_swizzle* swizzle,
	unsigned int thread_index);

/* (ref: astcenc.h:856) */
ASTCENC_PUBLIC enum astcenc_error astcenc_decompress_reset(struct astcenc_context* context);

/* (ref: astcenc.h:864) */
ASTCENC_PUBLIC void astcenc_context_free(struct astcenc_context* context);

/* (ref: astcenc.h:881) */
ASTCENC_PUBLIC enum astcenc_error astcenc_get_block_info(
	struct astcenc_context* context,
	const uint8_t data[16],
	struct astcenc_block_info* info);

/* (ref: astcenc.h:893) */
ASTCENC_PUBLIC const char* astcenc_get_error_string(enum astcenc_error status);

#endif
