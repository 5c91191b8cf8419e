// SPDX-License-Identifier: Apache-2.0
// Host side of the drop-in boundary: the ten astcenc_* entry points (include/astcenc.h).
//
// Mirrors the argument checking, preset interpolation, config clamping and caller-threading
// contract of the reference API layer so that error codes and the search parameters derived from
// (profile, block size, quality, flags) are identical:
//   ref: presets                    Source/astcenc_entry.cpp:65-135
//        validate_*                 :215-501
//        astcenc_config_init        :504-723
//        astcenc_context_alloc      :726-859
//        astcenc_compress_image     :1113-1228   (+ compress_image :891 for the block order)
//        astcenc_compress_reset/cancel :1231-1271
//        ParallelManager semantics  Source/astcenc_internal_entry.h:97-329
// The per-block work itself is handed to the backend (HIP kernels).
#include "../../include/astcenc.h"
#include "../../include/astcenc_amd.h"
#include "backend.h"
#include "host_tables.h"

// sequential (host) build of the block decoder source, for astcenc_get_block_info
#define ASTC_VARIANT v_host
#define ASTC_ENABLE_HDR 1
#include "wave_decode.h"
namespace block_info_host = astcd::v_host;

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <condition_variable>
#include <new>
#include <vector>

using namespace astcd;

namespace {

// ---------------------------------------------------------------------------------------------
// The reference library's own pow() approximation (ref: astcenc_vecmathlib.h:388-454); used once
// per context to turn the dB limit into a squared-error threshold, so it has to match bit for bit.
// ---------------------------------------------------------------------------------------------
float as_float(int v) { float f; memcpy(&f, &v, 4); return f; }
int as_int(float v) { int i; memcpy(&i, &v, 4); return i; }

float ref_log2(float x)
{
	int i = as_int(x);
	float e = (float)((int)(((unsigned)(i & 0x7F800000)) >> 23) - 127);
	float m = as_float((i & 0x007FFFFF) | 0x3F800000);
	float p = 0.0596515482674574969533f;
	p = (p * m) + -0.465725644288844778798f;
	p = (p * m) + 1.48116647521213171641f;
	p = (p * m) + -2.52074962577807006663f;
	p = (p * m) + 2.8882704548164776201f;
	p = p * (m - 1.0f);
	return p + e;
}

float ref_exp2(float x)
{
	float t = x > -126.99999f ? x : -126.99999f;
	x = t < 129.0f ? t : 129.0f;
	int ipart = (int)(x - 0.5f);
	float fpart = x - (float)ipart;
	float iexp = as_float((int)((unsigned)(ipart + 127) << 23));
	float f = 1.8775767e-3f;
	f = (f * fpart) + 8.9893397e-3f;
	f = (f * fpart) + 5.5826318e-2f;
	f = (f * fpart) + 2.4015361e-1f;
	f = (f * fpart) + 6.9315308e-1f;
	f = (f * fpart) + 9.9999994e-1f;
	return iexp * f;
}

float ref_pow(float x, float y)
{
	if (y == 0.0f) return 1.0f;
	return ref_exp2(ref_log2(x) * y);
}

// ---------------------------------------------------------------------------------------------
// Quality presets (search-effort knobs per footprint class).  These numbers are part of the
// behavioural contract: preset -> tuning values -> output bytes.  (ref: astcenc_entry.cpp:65-135)
// Columns: quality | partition count | 2/3/4-partition index limits | block mode percentile |
//          refinement | candidates | 2/3/4-partitioning candidates | dB a-base, b-base |
//          mse overshoot | 2/3-partition early-out | 2-plane correlation | mode0 enable
// ---------------------------------------------------------------------------------------------
struct Preset {
	float quality;
	unsigned int pcount, i2, i3, i4, modes, refine, cand, c2, c3, c4;
	float db_a, db_b, overshoot, e2, e3, corr, mode0;
};

const Preset presets_small[6] = {   // fewer than 25 texels
	{   0.0f, 2,  10,   6,   4,  43, 2, 2, 2, 2, 2,  85.2f,  63.2f,  3.5f, 1.00f, 1.00f, 0.85f, 0.0f },
	{  10.0f, 3,  18,  10,   8,  55, 3, 3, 2, 2, 2,  85.2f,  63.2f,  3.5f, 1.00f, 1.00f, 0.90f, 0.0f },
	{  60.0f, 4,  34,  28,  16,  77, 3, 3, 2, 2, 2,  95.0f,  70.0f,  2.5f, 1.10f, 1.05f, 0.95f, 0.0f },
	{  98.0f, 4,  82,  60,  30,  94, 4, 4, 3, 2, 2, 105.0f,  77.0f, 10.0f, 1.35f, 1.15f, 0.97f, 0.0f },
	{  99.0f, 4, 256, 128,  64,  98, 4, 6, 8, 6, 4, 200.0f, 200.0f, 10.0f, 1.60f, 1.40f, 0.98f, 0.0f },
	{ 100.0f, 4, 512, 512, 512, 100, 4, 8, 8, 8, 8, 200.0f, 200.0f, 10.0f, 2.00f, 2.00f, 0.99f, 0.0f }
};
const Preset presets_mid[6] = {     // 25..63 texels
	{   0.0f, 2,  10,   6,   4,  43, 2, 2, 2, 2, 2,  85.2f,  63.2f,  3.5f, 1.00f, 1.00f, 0.80f, 1.0f },
	{  10.0f, 3,  18,  12,  10,  55, 3, 3, 2, 2, 2,  85.2f,  63.2f,  3.5f, 1.00f, 1.00f, 0.85f, 1.0f },
	{  60.0f, 3,  34,  28,  16,  77, 3, 3, 2, 2, 2,  95.0f,  70.0f,  3.0f, 1.10f, 1.05f, 0.90f, 1.0f },
	{  98.0f, 4,  82,  60,  30,  94, 4, 4, 3, 2, 2, 105.0f,  77.0f, 10.0f, 1.40f, 1.20f, 0.95f, 0.0f },
	{  99.0f, 4, 256, 128,  64,  98, 4, 6, 8, 6, 3, 200.0f, 200.0f, 10.0f, 1.60f, 1.40f, 0.98f, 0.0f },
	{ 100.0f, 4, 256, 256, 256, 100, 4, 8, 8, 8, 8, 200.0f, 200.0f, 10.0f, 2.00f, 2.00f, 0.99f, 0.0f }
};
const Preset presets_large[6] = {   // 64 texels and up
	{   0.0f, 2,  10,   6,   4,  40, 2, 2, 2, 2, 2,  85.0f,  63.0f,  3.5f, 1.00f, 1.00f, 0.80f, 1.0f },
	{  10.0f, 2,  18,  12,  10,  55, 3, 3, 2, 2, 2,  85.0f,  63.0f,  3.5f, 1.00f, 1.00f, 0.85f, 1.0f },
	{  60.0f, 3,  34,  28,  16,  77, 3, 3, 2, 2, 2,  95.0f,  70.0f,  3.5f, 1.10f, 1.05f, 0.90f, 1.0f },
	{  98.0f, 4,  82,  60,  30,  93, 4, 4, 3, 2, 2, 105.0f,  77.0f, 10.0f, 1.30f, 1.20f, 0.97f, 1.0f },
	{  99.0f, 4, 256, 128,  64,  98, 4, 6, 8, 5, 2, 200.0f, 200.0f, 10.0f, 1.60f, 1.40f, 0.98f, 1.0f },
	{ 100.0f, 4, 256, 256, 256, 100, 4, 8, 8, 8, 8, 200.0f, 200.0f, 10.0f, 2.00f, 2.00f, 0.99f, 1.0f }
};

astcenc_error validate_cpu_float()
{
	volatile float xprec_testval = 2.51f;
	float store = xprec_testval + 12582912.0f;
	float q = store - 12582912.0f;
	return q == 3.0f ? ASTCENC_SUCCESS : ASTCENC_ERR_BAD_CPU_FLOAT;
}

astcenc_error validate_profile(astcenc_profile profile)
{
	switch ((int)profile)
	{
	case ASTCENC_PRF_LDR_SRGB: case ASTCENC_PRF_LDR: case ASTCENC_PRF_HDR_RGB_LDR_A: case ASTCENC_PRF_HDR:
		return ASTCENC_SUCCESS;
	default:
		return ASTCENC_ERR_BAD_PROFILE;
	}
}

astcenc_error validate_block_size(unsigned int bx, unsigned int by, unsigned int bz)
{
	bool is_legal = ((bz <= 1) && is_legal_2d_block_size(bx, by)) || ((bz >= 2) && is_legal_3d_block_size(bx, by, bz));
	if (!is_legal) return ASTCENC_ERR_BAD_BLOCK_SIZE;
	if (bx * by * bz > 216) return ASTCENC_ERR_NOT_IMPLEMENTED;
	return ASTCENC_SUCCESS;
}

int popcount32(unsigned int v) { return __builtin_popcount(v); }

astcenc_error validate_flags(astcenc_profile profile, unsigned int flags)
{
	if (popcount32(flags & ~ASTCENC_ALL_FLAGS) != 0) return ASTCENC_ERR_BAD_FLAGS;
	if (popcount32(flags & (ASTCENC_FLG_MAP_NORMAL | ASTCENC_FLG_MAP_RGBM)) > 1) return ASTCENC_ERR_BAD_FLAGS;
	bool is_unorm8 = flags & ASTCENC_FLG_USE_DECODE_UNORM8;
	bool is_hdr = (profile == ASTCENC_PRF_HDR) || (profile == ASTCENC_PRF_HDR_RGB_LDR_A);
	if (is_unorm8 && is_hdr) return ASTCENC_ERR_BAD_DECODE_MODE;
	return ASTCENC_SUCCESS;
}

bool swz_ok(astcenc_swz s, bool allow_z)
{
	return (int)s >= ASTCENC_SWZ_R && ((int)s <= ASTCENC_SWZ_1 || (allow_z && s == ASTCENC_SWZ_Z));
}

template <typename T> T clampv(T v, T lo, T hi) { return v > hi ? hi : (v > lo ? v : lo); }
float maxf(float a, float b) { return a > b ? a : b; }
unsigned int maxu(unsigned int a, unsigned int b) { return a > b ? a : b; }

/* (ref: validate_config :434) */
astcenc_error validate_config(astcenc_config& config)
{
	astcenc_error status;
	if ((status = validate_profile(config.profile)) != ASTCENC_SUCCESS) return status;
	if ((status = validate_flags(config.profile, config.flags)) != ASTCENC_SUCCESS) return status;
	if ((status = validate_block_size(config.block_x, config.block_y, config.block_z)) != ASTCENC_SUCCESS) return status;

	config.rgbm_m_scale = maxf(config.rgbm_m_scale, 1.0f);
	config.tune_partition_count_limit = clampv(config.tune_partition_count_limit, 1u, 4u);
	config.tune_2partition_index_limit = clampv(config.tune_2partition_index_limit, 1u, 1024u);
	config.tune_3partition_index_limit = clampv(config.tune_3partition_index_limit, 1u, 1024u);
	config.tune_4partition_index_limit = clampv(config.tune_4partition_index_limit, 1u, 1024u);
	config.tune_block_mode_limit = clampv(config.tune_block_mode_limit, 1u, 100u);
	config.tune_refinement_limit = maxu(config.tune_refinement_limit, 1u);
	config.tune_candidate_limit = clampv(config.tune_candidate_limit, 1u, 8u);
	config.tune_2partitioning_candidate_limit = clampv(config.tune_2partitioning_candidate_limit, 1u, 8u);
	config.tune_3partitioning_candidate_limit = clampv(config.tune_3partitioning_candidate_limit, 1u, 8u);
	config.tune_4partitioning_candidate_limit = clampv(config.tune_4partitioning_candidate_limit, 1u, 8u);
	config.tune_db_limit = maxf(config.tune_db_limit, 0.0f);
	config.tune_mse_overshoot = maxf(config.tune_mse_overshoot, 1.0f);
	config.tune_2partition_early_out_limit_factor = maxf(config.tune_2partition_early_out_limit_factor, 0.0f);
	config.tune_3partition_early_out_limit_factor = maxf(config.tune_3partition_early_out_limit_factor, 0.0f);
	config.tune_2plane_early_out_limit_correlation = maxf(config.tune_2plane_early_out_limit_correlation, 0.0f);

	// channel weights below 1/1000 of the largest are raised; all-zero is rejected
	float max_weight = maxf(maxf(config.cw_r_weight, config.cw_g_weight), maxf(config.cw_b_weight, config.cw_a_weight));
	if (max_weight > 0.0f)
	{
		max_weight /= 1000.0f;
		config.cw_r_weight = maxf(config.cw_r_weight, max_weight);
		config.cw_g_weight = maxf(config.cw_g_weight, max_weight);
		config.cw_b_weight = maxf(config.cw_b_weight, max_weight);
		config.cw_a_weight = maxf(config.cw_a_weight, max_weight);
	}
	else
	{
		return ASTCENC_ERR_BAD_PARAM;
	}
	return ASTCENC_SUCCESS;
}

size_t mul_safe(size_t a, size_t b, bool& overflow)
{
	size_t r = a * b;
	overflow = overflow || ((b != 0) && ((r / b) != a));
	return r;
}

size_t block_count_axis(size_t dim, size_t block)
{
	size_t n = dim / block;
	if (dim != block * n) n++;
	return n;
}

} // namespace

// ---------------------------------------------------------------------------------------------
// Context
// ---------------------------------------------------------------------------------------------
struct astcenc_context {
	astcenc_config config;            // validated copy; tune_db_limit already converted
	unsigned int thread_count;
	bool owns_tables;
	const astcenc_context* parent;
	std::vector<uint8_t>* blob;       // table blob (shared with child contexts)
	HostTables* host_tables;
	Backend* backend;

	// caller-thread rendezvous for compress (ref: ParallelManager)
	std::mutex lock;
	std::condition_variable cv;
	enum { IDLE, RUNNING, DONE } state;
	astcenc_error result;
	int dstate;                       // same protocol for decompress (ref: manage_decompress)
	astcenc_error dresult;
	std::atomic<int> cancel_flag;     // (ref: ParallelManager::m_is_cancelled, astcenc_internal_entry.h:104)
	int per_slice_fast_load;          // ASTCENC_AMD_OPT_PER_SLICE_FAST_LOAD: -1 not set (each entry point's default), 0, 1
};

extern "C" {

astcenc_error astcenc_config_init(astcenc_profile profile, unsigned int block_x, unsigned int block_y, unsigned int block_z,
                                  float quality, unsigned int flags, astcenc_config* configp)
{
	astcenc_error status = validate_cpu_float();
	if (status != ASTCENC_SUCCESS) return status;

	astcenc_config& config = *configp;
	memset(&config, 0, sizeof(config));

	block_z = block_z > 1u ? block_z : 1u;
	status = validate_block_size(block_x, block_y, block_z);
	if (status != ASTCENC_SUCCESS) return status;

	config.block_x = block_x;
	config.block_y = block_y;
	config.block_z = block_z;

	float texels = (float)(block_x * block_y * block_z);
	float ltexels = logf(texels) / logf(10.0f);

	if (quality < ASTCENC_PRE_FASTEST || quality > ASTCENC_PRE_EXHAUSTIVE) return ASTCENC_ERR_BAD_QUALITY;

	size_t texels_int = block_x * block_y * block_z;
	const Preset* presets = texels_int < 25 ? presets_small : texels_int < 64 ? presets_mid : presets_large;

	size_t end;
	for (end = 0; end < 6; end++)
	{
		if (presets[end].quality >= quality) break;
	}
	size_t start = end == 0 ? 0 : end - 1;

	if (start == end)
	{
		const Preset& p = presets[start];
		config.tune_partition_count_limit = p.pcount;
		config.tune_2partition_index_limit = p.i2;
		config.tune_3partition_index_limit = p.i3;
		config.tune_4partition_index_limit = p.i4;
		config.tune_block_mode_limit = p.modes;
		config.tune_refinement_limit = p.refine;
		config.tune_candidate_limit = p.cand;
		config.tune_2partitioning_candidate_limit = p.c2;
		config.tune_3partitioning_candidate_limit = p.c3;
		config.tune_4partitioning_candidate_limit = p.c4;
		config.tune_db_limit = maxf(p.db_a - 35 * ltexels, p.db_b - 19 * ltexels);
		config.tune_mse_overshoot = p.overshoot;
		config.tune_2partition_early_out_limit_factor = p.e2;
		config.tune_3partition_early_out_limit_factor = p.e3;
		config.tune_2plane_early_out_limit_correlation = p.corr;
		config.tune_search_mode0_enable = p.mode0;
	}
	else
	{
		// linear blend of the two bracketing presets
		const Preset& a = presets[start];
		const Preset& b = presets[end];
		float wt_range = b.quality - a.quality;
		float wa = (b.quality - quality) / wt_range;
		float wb = (quality - a.quality) / wt_range;
		auto lerp = [&](float x, float y) { return (x * wa) + (y * wb); };
		auto lerpi = [&](unsigned int x, unsigned int y) { return (unsigned int)(int)(((float)x * wa) + ((float)y * wb) + 0.5f); };

		config.tune_partition_count_limit = lerpi(a.pcount, b.pcount);
		config.tune_2partition_index_limit = lerpi(a.i2, b.i2);
		config.tune_3partition_index_limit = lerpi(a.i3, b.i3);
		config.tune_4partition_index_limit = lerpi(a.i4, b.i4);
		config.tune_block_mode_limit = lerpi(a.modes, b.modes);
		config.tune_refinement_limit = lerpi(a.refine, b.refine);
		config.tune_candidate_limit = lerpi(a.cand, b.cand);
		config.tune_2partitioning_candidate_limit = lerpi(a.c2, b.c2);
		config.tune_3partitioning_candidate_limit = lerpi(a.c3, b.c3);
		config.tune_4partitioning_candidate_limit = lerpi(a.c4, b.c4);
		config.tune_db_limit = maxf(lerp(a.db_a, b.db_a) - 35 * ltexels, lerp(a.db_b, b.db_b) - 19 * ltexels);
		config.tune_mse_overshoot = lerp(a.overshoot, b.overshoot);
		config.tune_2partition_early_out_limit_factor = lerp(a.e2, b.e2);
		config.tune_3partition_early_out_limit_factor = lerp(a.e3, b.e3);
		config.tune_2plane_early_out_limit_correlation = lerp(a.corr, b.corr);
		config.tune_search_mode0_enable = lerp(a.mode0, b.mode0);
	}

	config.cw_r_weight = 1.0f;
	config.cw_g_weight = 1.0f;
	config.cw_b_weight = 1.0f;
	config.cw_a_weight = 1.0f;
	config.a_scale_radius = 0;
	config.rgbm_m_scale = 0.0f;
	config.profile = profile;

	switch ((int)profile)
	{
	case ASTCENC_PRF_LDR: case ASTCENC_PRF_LDR_SRGB:
		break;
	case ASTCENC_PRF_HDR_RGB_LDR_A: case ASTCENC_PRF_HDR:
		config.tune_db_limit = 999.0f;
		config.tune_search_mode0_enable = 0.0f;
		break;
	default:
		return ASTCENC_ERR_BAD_PROFILE;
	}

	status = validate_flags(profile, flags);
	if (status != ASTCENC_SUCCESS) return status;

	if (flags & ASTCENC_FLG_MAP_NORMAL)
	{
		config.tune_partition_count_limit = config.tune_partition_count_limit + 1u < 4u ? config.tune_partition_count_limit + 1u : 4u;
		config.cw_g_weight = 0.0f;
		config.cw_b_weight = 0.0f;
		config.tune_2partition_early_out_limit_factor *= 1.5f;
		config.tune_3partition_early_out_limit_factor *= 1.5f;
		config.tune_2plane_early_out_limit_correlation = 0.99f;
		config.tune_db_limit *= 1.03f;
	}
	else if (flags & ASTCENC_FLG_MAP_RGBM)
	{
		config.rgbm_m_scale = 5.0f;
		config.cw_a_weight = 2.0f * config.rgbm_m_scale;
	}
	else if (flags & ASTCENC_FLG_USE_PERCEPTUAL)
	{
		config.cw_r_weight = 0.30f * 2.25f;
		config.cw_g_weight = 0.59f * 2.25f;
		config.cw_b_weight = 0.11f * 2.25f;
	}
	config.flags = flags;
	return ASTCENC_SUCCESS;
}

astcenc_error astcenc_context_alloc(const astcenc_config* configp, unsigned int thread_count,
                                    astcenc_context** context, const astcenc_context* parent_context)
{
	astcenc_error status = validate_cpu_float();
	if (status != ASTCENC_SUCCESS) return status;
	if (thread_count == 0) return ASTCENC_ERR_BAD_PARAM;

	bool has_config = configp != nullptr;
	bool has_parent = parent_context != nullptr;
	if (!(has_config ^ has_parent)) return ASTCENC_ERR_BAD_PARAM;
	if (has_parent) configp = &parent_context->config;

	astcenc_context* ctx = new (std::nothrow) astcenc_context;
	if (!ctx) return ASTCENC_ERR_OUT_OF_MEM;
	ctx->thread_count = thread_count;
	ctx->config = *configp;
	ctx->owns_tables = false;
	ctx->parent = parent_context;
	ctx->blob = nullptr;
	ctx->host_tables = nullptr;
	ctx->backend = nullptr;
	ctx->state = astcenc_context::IDLE;
	ctx->dstate = astcenc_context::IDLE;
	ctx->result = ASTCENC_SUCCESS;
	ctx->cancel_flag.store(0);
	ctx->per_slice_fast_load = -1;

	// NB: like the reference, a child context re-validates (and below re-converts) the parent's
	// already processed config (ref: astcenc_entry.cpp:761-777, :811-821).
	status = validate_config(ctx->config);
	if (status != ASTCENC_SUCCESS)
	{
		delete ctx;
		return status;
	}

	const astcenc_config& config = ctx->config;

	bool is_hdr = config.profile == ASTCENC_PRF_HDR || config.profile == ASTCENC_PRF_HDR_RGB_LDR_A;
	bool compress = !(config.flags & ASTCENC_FLG_DECOMPRESS_ONLY);

	if (!has_parent)
	{
		ctx->blob = new std::vector<uint8_t>();
		ctx->host_tables = new HostTables();
		ctx->owns_tables = true;
		if (!build_tables(config.block_x, config.block_y, config.block_z, config.tune_partition_count_limit,
		                  (float)config.tune_block_mode_limit / 100.0f, *ctx->blob, *ctx->host_tables))
		{
			delete ctx->blob; delete ctx->host_tables; delete ctx;
			return ASTCENC_ERR_OUT_OF_MEM;
		}

	}
	else
	{
		ctx->blob = parent_context->blob;
		ctx->host_tables = parent_context->host_tables;
	}

	// dB limit -> per-texel squared error threshold (ref: astcenc_entry.cpp:814-821)
	if (compress)
	{
		if (!is_hdr) ctx->config.tune_db_limit = ref_pow(0.1f, ctx->config.tune_db_limit * 0.1f) * 65535.0f * 65535.0f;
		else ctx->config.tune_db_limit = 0.0f;
	}

	// decompress-only contexts need the device too (the decode kernel), so the backend always exists
	{
		DeviceConfig dc;
		memset(&dc, 0, sizeof(dc));
		dc.profile = (int32_t)config.profile;
		dc.flags = config.flags;
		dc.cw[0] = config.cw_r_weight; dc.cw[1] = config.cw_g_weight; dc.cw[2] = config.cw_b_weight; dc.cw[3] = config.cw_a_weight;
		dc.rgbm_m_scale = config.rgbm_m_scale;
		dc.tune_partition_count_limit = config.tune_partition_count_limit;
		dc.tune_partition_index_limit[0] = config.tune_2partition_index_limit;
		dc.tune_partition_index_limit[1] = config.tune_3partition_index_limit;
		dc.tune_partition_index_limit[2] = config.tune_4partition_index_limit;
		dc.tune_refinement_limit = config.tune_refinement_limit;
		dc.tune_candidate_limit = config.tune_candidate_limit;
		dc.tune_partitioning_candidate_limit[0] = config.tune_2partitioning_candidate_limit;
		dc.tune_partitioning_candidate_limit[1] = config.tune_3partitioning_candidate_limit;
		dc.tune_partitioning_candidate_limit[2] = config.tune_4partitioning_candidate_limit;
		dc.tune_db_limit = ctx->config.tune_db_limit;
		dc.tune_mse_overshoot = config.tune_mse_overshoot;
		dc.tune_partition_early_out_limit_factor[0] = config.tune_2partition_early_out_limit_factor;
		dc.tune_partition_early_out_limit_factor[1] = config.tune_3partition_early_out_limit_factor;
		dc.tune_2plane_early_out_limit_correlation = config.tune_2plane_early_out_limit_correlation;
		dc.tune_search_mode0_enable = config.tune_search_mode0_enable;
#if defined(ASTC_DUPSTAGE)
		if (const char* dup = getenv("ASTC_DUP_STAGE")) dc.debug_dup_stage = (uint32_t)atoi(dup);   // instruction-count builds only
#endif

		int bstatus = 0;
		ctx->backend = backend_create(ctx->blob->data(), ctx->blob->size(), dc, &bstatus);
		if (!ctx->backend)
		{
			if (ctx->owns_tables) { delete ctx->blob; delete ctx->host_tables; }
			delete ctx;
			// no silent CPU fallback: without a usable HIP device the context cannot exist
			return bstatus == 1 ? ASTCENC_ERR_OUT_OF_MEM : ASTCENC_ERR_NOT_IMPLEMENTED;
		}
	}

	*context = ctx;
	return ASTCENC_SUCCESS;
}

void astcenc_context_free(astcenc_context* ctx)
{
	if (!ctx) return;
	if (ctx->backend) backend_destroy(ctx->backend);
	if (ctx->owns_tables)
	{
		delete ctx->blob;
		delete ctx->host_tables;
	}
	delete ctx;
}

static astcenc_error check_compress_args(astcenc_context* ctx, unsigned int dim_x, unsigned int dim_y, unsigned int dim_z,
                                         const astcenc_swizzle* swizzle, size_t data_len, unsigned int thread_index, size_t& block_count)
{
	if (ctx->config.flags & ASTCENC_FLG_DECOMPRESS_ONLY) return ASTCENC_ERR_BAD_CONTEXT;
	if (!swz_ok(swizzle->r, false) || !swz_ok(swizzle->g, false) || !swz_ok(swizzle->b, false) || !swz_ok(swizzle->a, false))
	{
		return ASTCENC_ERR_BAD_SWIZZLE;
	}
	if (thread_index >= ctx->thread_count) return ASTCENC_ERR_BAD_PARAM;

	bool overflow = false;
	size_t texel_count = mul_safe(mul_safe(dim_x, dim_y, overflow), dim_z, overflow);
	if (overflow || texel_count == 0) return ASTCENC_ERR_BAD_PARAM;

	size_t bx = block_count_axis(dim_x, ctx->config.block_x);
	size_t by = block_count_axis(dim_y, ctx->config.block_y);
	size_t bz = block_count_axis(dim_z, ctx->config.block_z);
	overflow = false;
	block_count = mul_safe(mul_safe(bx, by, overflow), bz, overflow);
	mul_safe(block_count, 16, overflow);
	if (overflow || block_count == 0) return ASTCENC_ERR_BAD_PARAM;
	if (data_len < block_count * 16) return ASTCENC_ERR_OUT_OF_MEM;
	return ASTCENC_SUCCESS;
}

static astcenc_error run_job(astcenc_context* ctx, CompressJob& job)
{
	// Every caller thread of the context enters here; the first one drives the device, the rest
	// wait for the same completion (ref: ParallelManager init/wait, astcenc_internal_entry.h:97-329).
	std::unique_lock<std::mutex> lk(ctx->lock);
	if (ctx->thread_count == 1)
	{
		// a single caller resets implicitly, which also clears a pending cancel (ref: astcenc_entry.cpp:1185-1188,
		// astcenc_compress_reset -> ParallelManager::reset)
		ctx->state = astcenc_context::IDLE;
		ctx->cancel_flag.store(0);
	}

	if (ctx->state == astcenc_context::IDLE)
	{
		ctx->state = astcenc_context::RUNNING;
		lk.unlock();
		job.cancel_flag = &ctx->cancel_flag;
		job.progress = ctx->config.progress_callback;
		int rc = backend_compress(ctx->backend, job);
		lk.lock();
		ctx->result = rc == 0 ? ASTCENC_SUCCESS : rc == 1 ? ASTCENC_ERR_OUT_OF_MEM : ASTCENC_ERR_BAD_CONTEXT;
		ctx->state = astcenc_context::DONE;
		ctx->cv.notify_all();
		return ctx->result;
	}
	while (ctx->state == astcenc_context::RUNNING) ctx->cv.wait(lk);
	return ctx->result;
}

astcenc_error astcenc_compress_image(astcenc_context* ctx, astcenc_image* imagep, const astcenc_swizzle* swizzle,
                                     uint8_t* data_out, size_t data_len, unsigned int thread_index)
{
	astcenc_image& image = *imagep;
	size_t block_count;
	astcenc_error status = check_compress_args(ctx, image.dim_x, image.dim_y, image.dim_z, swizzle, data_len, thread_index, block_count);
	if (status != ASTCENC_SUCCESS) return status;
	// The alpha-scale test only exists for 2D footprints (ref: astcenc_entry.cpp:975).  On a multi-slice image the
	// reference averages over a 3D box and then reads the averages around slice 0 for every slice; the pre-pass
	// reproduces exactly that (wave_alpha.h).
	const bool alpha_scale = ctx->config.a_scale_radius != 0 && ctx->config.block_z <= 1;

	CompressJob job;
	memset(&job, 0, sizeof(job));
	job.host_slices = image.data;
	job.dim_x = image.dim_x;
	job.dim_y = image.dim_y;
	job.dim_z = image.dim_z;
	job.data_type = (uint32_t)image.data_type;
	job.swz[0] = swizzle->r; job.swz[1] = swizzle->g; job.swz[2] = swizzle->b; job.swz[3] = swizzle->a;
	job.host_out = data_out;
	job.a_scale_radius = alpha_scale ? ctx->config.a_scale_radius : 0u;
	// default: the reference's bytes (its fast loader reads slice 0 for every slice, astcenc_image.cpp:304)
	job.fast_load_slice0 = ctx->per_slice_fast_load == 1 ? 0u : 1u;
	return run_job(ctx, job);
}

astcenc_error astcenc_amd_compress_image_device(astcenc_context* ctx, const void* device_image,
                                                unsigned int dim_x, unsigned int dim_y, astcenc_type data_type,
                                                const astcenc_swizzle* swizzle, void* device_out, size_t data_len,
                                                void* hip_stream, float* kernel_ms)
{
	return astcenc_amd_compress_volume_device(ctx, device_image, dim_x, dim_y, 1, data_type, swizzle, device_out, data_len, hip_stream, kernel_ms);
}

astcenc_error astcenc_amd_compress_volume_device(astcenc_context* ctx, const void* device_image,
                                                 unsigned int dim_x, unsigned int dim_y, unsigned int dim_z, astcenc_type data_type,
                                                 const astcenc_swizzle* swizzle, void* device_out, size_t data_len,
                                                 void* hip_stream, float* kernel_ms)
{
	size_t block_count;
	astcenc_error status = check_compress_args(ctx, dim_x, dim_y, dim_z, swizzle, data_len, 0, block_count);
	if (status != ASTCENC_SUCCESS) return status;
	const bool alpha_scale = ctx->config.a_scale_radius != 0 && ctx->config.block_z <= 1;

	CompressJob job;
	memset(&job, 0, sizeof(job));
	job.device_data = device_image;
	job.dim_x = dim_x;
	job.dim_y = dim_y;
	job.dim_z = dim_z;
	job.data_type = (uint32_t)data_type;
	job.swz[0] = swizzle->r; job.swz[1] = swizzle->g; job.swz[2] = swizzle->b; job.swz[3] = swizzle->a;
	job.device_out = static_cast<uint8_t*>(device_out);
	job.stream = hip_stream;
	job.kernel_ms = kernel_ms;
	job.a_scale_radius = alpha_scale ? ctx->config.a_scale_radius : 0u;
	// default of this entry point (which has no reference counterpart to match): every slice from its own data
	job.fast_load_slice0 = ctx->per_slice_fast_load == 0 ? 1u : 0u;

	// A device-resident call is a single-caller operation.  On a thread_count == 1 context it starts from a clean
	// state like astcenc_compress_image does there (a cancel issued before the call is forgotten; one issued while
	// it runs stops it at the next chunk).  On a multi-thread context a cancel is sticky until
	// astcenc_compress_reset -- a device call must not swallow the cancel of a concurrent or later
	// astcenc_compress_image -- so a pending one stops this call as well.  Calls on one context are serialised per
	// device inside the backend.
	if (ctx->thread_count == 1) ctx->cancel_flag.store(0);
	job.cancel_flag = &ctx->cancel_flag;
	job.progress = ctx->config.progress_callback;
	int rc = backend_compress(ctx->backend, job);
	return rc == 0 ? ASTCENC_SUCCESS : rc == 1 ? ASTCENC_ERR_OUT_OF_MEM : rc == 3 ? ASTCENC_ERR_BAD_PARAM : ASTCENC_ERR_BAD_CONTEXT;
}

astcenc_error astcenc_amd_decompress_image_device(astcenc_context* ctx, const void* device_blocks, size_t data_len,
                                                  void* device_image, unsigned int dim_x, unsigned int dim_y, unsigned int dim_z,
                                                  astcenc_type data_type, const astcenc_swizzle* swizzle, void* hip_stream)
{
	if (!swz_ok(swizzle->r, true) || !swz_ok(swizzle->g, true) || !swz_ok(swizzle->b, true) || !swz_ok(swizzle->a, true))
	{
		return ASTCENC_ERR_BAD_SWIZZLE;
	}
	bool overflow = false;
	size_t texel_count = mul_safe(mul_safe(dim_x, dim_y, overflow), dim_z, overflow);
	if (overflow || texel_count == 0 || !device_blocks || !device_image) return ASTCENC_ERR_BAD_PARAM;
	size_t block_count = mul_safe(mul_safe(block_count_axis(dim_x, ctx->config.block_x), block_count_axis(dim_y, ctx->config.block_y), overflow),
	                              block_count_axis(dim_z, ctx->config.block_z), overflow);
	mul_safe(block_count, 16, overflow);
	if (overflow || block_count == 0) return ASTCENC_ERR_BAD_PARAM;
	if (data_len < block_count * 16) return ASTCENC_ERR_OUT_OF_MEM;

	DecompressDeviceJob job;
	memset(&job, 0, sizeof(job));
	job.device_blocks = static_cast<const uint8_t*>(device_blocks);
	job.device_image = device_image;
	job.dim_x = dim_x; job.dim_y = dim_y; job.dim_z = dim_z;
	job.data_type = (uint32_t)data_type;
	job.swz[0] = swizzle->r; job.swz[1] = swizzle->g; job.swz[2] = swizzle->b; job.swz[3] = swizzle->a;
	job.stream = hip_stream;
	int rc = backend_decompress_device(ctx->backend, job);
	return rc == 0 ? ASTCENC_SUCCESS : rc == 1 ? ASTCENC_ERR_OUT_OF_MEM : rc == 3 ? ASTCENC_ERR_BAD_PARAM : ASTCENC_ERR_BAD_CONTEXT;
}

static astcenc_error compare_images(astcenc_context* ctx, const void* device_image1, astcenc_type type1,
                                    const void* device_image2, astcenc_type type2,
                                    unsigned int dim_x, unsigned int dim_y, unsigned int dim_z, void* hip_stream,
                                    astcenc_amd_error_sums* sums, astcenc_amd_hdr_error_sums* hdr_sums, int fstop_lo, int fstop_hi)
{
	bool overflow = false;
	size_t texel_count = mul_safe(mul_safe(dim_x, dim_y, overflow), dim_z, overflow);
	if (overflow || texel_count == 0 || !device_image1 || !device_image2 || !sums) return ASTCENC_ERR_BAD_PARAM;
	if ((unsigned)type1 > 2u || (unsigned)type2 > 2u) return ASTCENC_ERR_BAD_PARAM;
	// the f-stop becomes a float exponent (ref: mpsnr_operator: "should be in range [-125, 125]")
	if (hdr_sums && (fstop_lo < -125 || fstop_hi > 125 || fstop_hi < fstop_lo)) return ASTCENC_ERR_BAD_PARAM;

	double raw[METRIC_SUMS_HOST];
	CompareJob job;
	memset(&job, 0, sizeof(job));
	job.device_a = device_image1; job.type_a = (uint32_t)type1;
	job.device_b = device_image2; job.type_b = (uint32_t)type2;
	job.texels = texel_count;
	job.stream = hip_stream;
	job.sums = raw;
	job.hdr = hdr_sums ? 1 : 0; job.fstop_lo = fstop_lo; job.fstop_hi = fstop_hi;
	int rc = backend_compare(ctx->backend, job);
	if (rc != 0) return rc == 1 ? ASTCENC_ERR_OUT_OF_MEM : rc == 3 ? ASTCENC_ERR_BAD_PARAM : ASTCENC_ERR_BAD_CONTEXT;
	for (int k = 0; k < 4; k++) { sums->squared_error[k] = raw[k]; sums->alpha_scaled_squared_error[k] = raw[4 + k]; }
	sums->rgb_peak = raw[8];
	sums->texels = (double)texel_count;
	if (hdr_sums)
	{
		for (int k = 0; k < 4; k++) { hdr_sums->log2_squared_error[k] = raw[10 + k]; hdr_sums->mpsnr_squared_error[k] = raw[14 + k]; }
		hdr_sums->fstop_lo = fstop_lo; hdr_sums->fstop_hi = fstop_hi;
	}
	return ASTCENC_SUCCESS;
}

astcenc_error astcenc_amd_compare_images_device(astcenc_context* ctx, const void* device_image1, astcenc_type type1,
                                                const void* device_image2, astcenc_type type2,
                                                unsigned int dim_x, unsigned int dim_y, unsigned int dim_z,
                                                void* hip_stream, astcenc_amd_error_sums* sums)
{
	return compare_images(ctx, device_image1, type1, device_image2, type2, dim_x, dim_y, dim_z, hip_stream, sums, nullptr, 0, 0);
}

astcenc_error astcenc_amd_compare_images_hdr_device(astcenc_context* ctx, const void* device_image1, astcenc_type type1,
                                                    const void* device_image2, astcenc_type type2,
                                                    unsigned int dim_x, unsigned int dim_y, unsigned int dim_z,
                                                    int fstop_lo, int fstop_hi, void* hip_stream,
                                                    astcenc_amd_error_sums* sums, astcenc_amd_hdr_error_sums* hdr_sums)
{
	if (!hdr_sums) return ASTCENC_ERR_BAD_PARAM;
	return compare_images(ctx, device_image1, type1, device_image2, type2, dim_x, dim_y, dim_z, hip_stream, sums, hdr_sums, fstop_lo, fstop_hi);
}

const char* astcenc_amd_backend_name(void)
{
	return backend_name();
}

void astcenc_amd_set_log_callback(void (*callback)(const char* message))
{
	backend_set_log_callback(callback);
}

astcenc_error astcenc_amd_context_specialize(astcenc_context* ctx)
{
	if (!ctx || !ctx->backend) return ASTCENC_ERR_BAD_CONTEXT;
	return backend_specialize(ctx->backend) == 0 ? ASTCENC_SUCCESS : ASTCENC_ERR_NOT_IMPLEMENTED;
}

int astcenc_amd_context_device_count(const astcenc_context* ctx)
{
	return ctx && ctx->backend ? backend_device_count(ctx->backend) : 0;
}

const char* astcenc_amd_context_kernel_name(const astcenc_context* ctx)
{
	return ctx && ctx->backend ? backend_kernel_name(ctx->backend) : "";
}

astcenc_error astcenc_amd_context_set_option(astcenc_context* ctx, astcenc_amd_option option, int value)
{
	if (!ctx) return ASTCENC_ERR_BAD_CONTEXT;
	switch ((int)option)
	{
	case ASTCENC_AMD_OPT_PER_SLICE_FAST_LOAD:
		ctx->per_slice_fast_load = value != 0 ? 1 : 0;
		return ASTCENC_SUCCESS;
	default:
		return ASTCENC_ERR_BAD_PARAM;
	}
}

astcenc_error astcenc_compress_reset(astcenc_context* ctx)
{
	if (ctx->config.flags & ASTCENC_FLG_DECOMPRESS_ONLY) return ASTCENC_ERR_BAD_CONTEXT;
	std::lock_guard<std::mutex> lk(ctx->lock);
	ctx->state = astcenc_context::IDLE;
	ctx->cancel_flag.store(0);
	return ASTCENC_SUCCESS;
}

astcenc_error astcenc_compress_cancel(astcenc_context* ctx)
{
	if (ctx->config.flags & ASTCENC_FLG_DECOMPRESS_ONLY) return ASTCENC_ERR_BAD_CONTEXT;
	ctx->cancel_flag.store(1);
	return ASTCENC_SUCCESS;
}

/* (ref: astcenc_decompress_image, astcenc_entry.cpp:1274-1390).  Blocks go to the device, one wavefront
 * decodes each block straight into the output image, the image comes back. */
astcenc_error astcenc_decompress_image(astcenc_context* ctx, const uint8_t* data, size_t data_len,
                                       astcenc_image* image_outp, const astcenc_swizzle* swizzle, unsigned int thread_index)
{
	if (thread_index >= ctx->thread_count) return ASTCENC_ERR_BAD_PARAM;
	if (!swz_ok(swizzle->r, true) || !swz_ok(swizzle->g, true) || !swz_ok(swizzle->b, true) || !swz_ok(swizzle->a, true))
	{
		return ASTCENC_ERR_BAD_SWIZZLE;
	}
	bool overflow = false;
	size_t texel_count = mul_safe(mul_safe(image_outp->dim_x, image_outp->dim_y, overflow), image_outp->dim_z, overflow);
	if (overflow || texel_count == 0) return ASTCENC_ERR_BAD_PARAM;
	size_t bx = block_count_axis(image_outp->dim_x, ctx->config.block_x);
	size_t by = block_count_axis(image_outp->dim_y, ctx->config.block_y);
	size_t bz = block_count_axis(image_outp->dim_z, ctx->config.block_z);
	overflow = false;
	size_t block_count = mul_safe(mul_safe(bx, by, overflow), bz, overflow);
	mul_safe(block_count, 16, overflow);
	if (overflow || block_count == 0) return ASTCENC_ERR_BAD_PARAM;
	if (data_len < block_count * 16) return ASTCENC_ERR_OUT_OF_MEM;

	DecompressJob job;
	memset(&job, 0, sizeof(job));
	job.host_blocks = data;
	job.block_bytes = block_count * 16;
	job.host_slices = image_outp->data;
	job.dim_x = image_outp->dim_x;
	job.dim_y = image_outp->dim_y;
	job.dim_z = image_outp->dim_z;
	job.data_type = (uint32_t)image_outp->data_type;
	job.swz[0] = swizzle->r; job.swz[1] = swizzle->g; job.swz[2] = swizzle->b; job.swz[3] = swizzle->a;

	// every caller thread arrives here; the first drives the device, the others wait for it
	std::unique_lock<std::mutex> lk(ctx->lock);
	if (ctx->thread_count == 1) ctx->dstate = astcenc_context::IDLE;
	if (ctx->dstate == astcenc_context::IDLE)
	{
		ctx->dstate = astcenc_context::RUNNING;
		lk.unlock();
		int rc = backend_decompress(ctx->backend, job);
		lk.lock();
		ctx->dresult = rc == 0 ? ASTCENC_SUCCESS : rc == 1 ? ASTCENC_ERR_OUT_OF_MEM : ASTCENC_ERR_BAD_CONTEXT;
		ctx->dstate = astcenc_context::DONE;
		ctx->cv.notify_all();
		return ctx->dresult;
	}
	while (ctx->dstate == astcenc_context::RUNNING) ctx->cv.wait(lk);
	return ctx->dresult;
}

astcenc_error astcenc_decompress_reset(astcenc_context* ctx)
{
	std::unique_lock<std::mutex> lk(ctx->lock);
	ctx->dstate = astcenc_context::IDLE;
	return ASTCENC_SUCCESS;
}

/* (ref: astcenc_get_block_info, astcenc_entry.cpp:1401-1517).  A single block is described on the host by
 * the same decoder source the decompression kernel is built from (wave_decode.h, plain sequential build). */
astcenc_error astcenc_get_block_info(astcenc_context* ctx, const uint8_t data[16], astcenc_block_info* info)
{
	memset(info, 0, sizeof(*info));
	info->profile = ctx->config.profile;
	block_info_host::DecodeScratch scratch;
	memset(&scratch, 0, sizeof(scratch));
	block_info_host::describe_block(data, (int)ctx->config.block_x, (int)ctx->config.block_y, (int)(ctx->config.block_z > 1 ? ctx->config.block_z : 1),
	                                (int)ctx->config.profile, info, scratch);
	return ASTCENC_SUCCESS;
}

const char* astcenc_get_error_string(astcenc_error status)
{
	switch ((int)status)
	{
	case ASTCENC_SUCCESS: return "ASTCENC_SUCCESS";
	case ASTCENC_ERR_OUT_OF_MEM: return "ASTCENC_ERR_OUT_OF_MEM";
	case ASTCENC_ERR_BAD_CPU_FLOAT: return "ASTCENC_ERR_BAD_CPU_FLOAT";
	case ASTCENC_ERR_BAD_PARAM: return "ASTCENC_ERR_BAD_PARAM";
	case ASTCENC_ERR_BAD_BLOCK_SIZE: return "ASTCENC_ERR_BAD_BLOCK_SIZE";
	case ASTCENC_ERR_BAD_PROFILE: return "ASTCENC_ERR_BAD_PROFILE";
	case ASTCENC_ERR_BAD_QUALITY: return "ASTCENC_ERR_BAD_QUALITY";
	case ASTCENC_ERR_BAD_FLAGS: return "ASTCENC_ERR_BAD_FLAGS";
	case ASTCENC_ERR_BAD_SWIZZLE: return "ASTCENC_ERR_BAD_SWIZZLE";
	case ASTCENC_ERR_BAD_CONTEXT: return "ASTCENC_ERR_BAD_CONTEXT";
	case ASTCENC_ERR_NOT_IMPLEMENTED: return "ASTCENC_ERR_NOT_IMPLEMENTED";
	case ASTCENC_ERR_BAD_DECODE_MODE: return "ASTCENC_ERR_BAD_DECODE_MODE";
	default: return nullptr;
	}
}

} // extern "C"
