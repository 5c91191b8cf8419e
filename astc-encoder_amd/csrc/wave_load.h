// SPDX-License-Identifier: Apache-2.0
// Stage 0: fetch one block of texels from the image in HBM into LDS as SoA floats in 0..65535.
//   ref: load_image_block_fast_ldr   Source/astcenc_image.cpp:278-342
//        load_image_block            Source/astcenc_image.cpp:162-275
//        float_to_lns / lns_to_sf16  Source/astcenc_vecmathlib.h:537-620
// One texel per lane; rows of the block are contiguous 4*dim_x (U8) byte runs in the image, so a
// wave's loads coalesce into dim_y segments.
#pragma once
#include "wave_ctx.h"

namespace astcd { inline namespace ASTC_VARIANT {

/* (ref: vecmathlib.h:582 float_to_lns, scalar form) */
WV_FN float float_to_lns(float a)
{
	int ai = float_as_int(a);
	int exp = ((int)((unsigned)ai >> 23) & 0xFF) - 126;
	float mant = int_as_float((ai & (int)0x807FFFFF) | 0x3F000000);

	bool mask_underflow_nan = !(a > (1.0f / 67108864.0f));
	bool mask_infinity = a >= 65536.0f;
	bool exp_lt_m13 = exp < -13;

	float a1a = a * 33554432.0f;
	float a1b = (mant - 0.5f) * 4096;
	int expb = exp + 14;

	a = exp_lt_m13 ? a1a : a1b;
	exp = exp_lt_m13 ? 0 : expb;

	bool a_lt_384 = a < 384.0f;
	bool a_lt_1408 = a <= 1408.0f;

	float a2a = a * (4.0f / 3.0f);
	float a2b = a + 128.0f;
	float a2c = (a + 512.0f) * (4.0f / 5.0f);

	a = a2c;
	a = a_lt_1408 ? a2b : a;
	a = a_lt_384 ? a2a : a;

	a = a + ((float)exp * 2048.0f) + 1.0f;

	a = mask_infinity ? 65535.0f : a;
	a = mask_underflow_nan ? 0.0f : a;
	return a;
}

/* (ref: vecmathlib.h:537 lns_to_sf16, scalar form) */
WV_FN int lns_to_sf16(int p)
{
	int mc = p & 0x7FF;
	int ec = (int)((unsigned)p >> 11);
	int mt;
	if (mc < 512) mt = mc * 3;
	else if (mc < 1536) mt = mc * 4 - 512;
	else mt = mc * 5 - 2048;
	int res = (ec << 10) | (int)((unsigned)mt >> 3);
	return i_min(res, 0x7BFF);
}

/* Load block (bx, by, bz) of the image into LDS and compute the block statistics. */
WV_FN void load_block(const Ctx& c, const ImageDesc& img, unsigned int bx, unsigned int by, unsigned int bz)
{
	const int T = c.T;
	const int dim_x = c.root->dim_x;
	const unsigned int plane_texels = (unsigned)dim_x * (unsigned)c.root->dim_y;
	const bool volume = c.root->dim_z > 1;
	BlkInfo& blk = c.blk();
	const int profile = c.cfg->profile;
	float* dr = c.data(0); float* dg = c.data(1); float* db = c.data(2); float* da = c.data(3);

	const bool fast = img.use_fast_load != 0;
	const int rgb_lns = (kHdr && (profile == 3 /*HDR*/ || profile == 2 /*HDR_RGB_LDR_A*/)) ? 1 : 0;
	const int a_lns = (kHdr && profile == 3) ? 1 : 0;

	WV_FOR_T(t, T)
	{
		// texel order inside a block is x fastest, then y, then z (ref: image.cpp:221-233)
		unsigned int tz = volume ? (unsigned)t / plane_texels : 0u;
		unsigned int trem = (unsigned)t - tz * plane_texels;
		unsigned int ty = trem / (unsigned)dim_x;
		unsigned int tx = trem - ty * (unsigned)dim_x;
		unsigned int xi = bx * (unsigned)dim_x + tx;
		unsigned int yi = by * (unsigned)c.root->dim_y + ty;
		unsigned int zi = bz * (unsigned)c.root->dim_z + tz;
		xi = xi < img.dim_x - 1 ? xi : img.dim_x - 1;
		yi = yi < img.dim_y - 1 ? yi : img.dim_y - 1;
		zi = zi < img.dim_z - 1 ? zi : img.dim_z - 1;
		// the reference's RGBA8 fast loader reads slice 0 whatever the block's z (ref: astcenc_image.cpp:304)
		if (fast && img.fast_load_slice0) zi = 0;
		size_t base = (size_t)4 * ((size_t)img.dim_x * ((size_t)img.dim_y * zi + yi) + xi);

		float v[6];
		if (img.data_type == 0)
		{
			uint32_t px = *reinterpret_cast<const uint32_t*>(static_cast<const uint8_t*>(img.data) + base);
			float r = (float)(px & 0xFF), g = (float)((px >> 8) & 0xFF), b = (float)((px >> 16) & 0xFF), a = (float)(px >> 24);
			if (fast)
			{
				dr[t] = r * (65535.0f / 255.0f);
				dg[t] = g * (65535.0f / 255.0f);
				db[t] = b * (65535.0f / 255.0f);
				da[t] = a * (65535.0f / 255.0f);
				continue;
			}
			v[0] = r / 255.0f; v[1] = g / 255.0f; v[2] = b / 255.0f; v[3] = a / 255.0f;
		}
		else if (img.data_type == 1)
		{
			const uint16_t* p = static_cast<const uint16_t*>(img.data) + base;
			v[0] = half_to_float(p[0]); v[1] = half_to_float(p[1]); v[2] = half_to_float(p[2]); v[3] = half_to_float(p[3]);
		}
		else
		{
			const float* p = static_cast<const float*>(img.data) + base;
			v[0] = p[0]; v[1] = p[1]; v[2] = p[2]; v[3] = p[3];
		}

		v[4] = 0.0f; v[5] = 1.0f;
		float s[4];
		for (int k = 0; k < 4; k++) s[k] = v[img.swz[k] < 6 ? img.swz[k] : 4];

		for (int k = 0; k < 4; k++)
		{
			float unorm = v_clamp(0.0f, 65535.0f, s[k] * 65535.0f);
			bool use_lns = k < 3 ? rgb_lns : a_lns;
			float out = use_lns ? float_to_lns(s[k]) : unorm;
			c.data(k)[t] = out;
		}
	}
	// zero the SIMD padding tail so later 4-wide loops see defined values
	WV_FOR(t, c.Tp - T)
	{
		for (int k = 0; k < 4; k++) c.data(k)[T + t] = 0.0f;
	}
	WV_SYNC();

	// per-channel min / mean / max in texel order (ref: image.cpp:239-241, :317-319, :339)
	WV_FOR(k, 4)
	{
		const float* d = c.data(k);
		float mn = 1e38f, mx = -1e38f, mean = 0.0f;
		float scale = 1.0f / (float)T;
		for (int t = 0; t < T; t++)
		{
			float v = d[t];
			mn = mn < v ? mn : v;
			mx = mx > v ? mx : v;
			if (fast) mean += v; else mean += v * scale;
		}
		if (fast) mean = mean / (float)T;
		blk.data_min[k] = mn;
		blk.data_max[k] = mx;
		blk.data_mean[k] = mean;
		blk.cw[k] = c.cfg->cw[k];

		float e = d[0];
		float origin = e / 65535.0f;
		bool use_lns = !fast && (k < 3 ? rgb_lns : a_lns);
		if (use_lns)
		{
			origin = half_to_float((uint16_t)lns_to_sf16((int)e));
		}
		blk.origin[k] = origin;

		if (k == 0)
		{
			bool gray = true;
			for (int t = 0; t < T; t++)
			{
				gray = gray && (dr[t] == dg[t]) && (dr[t] == db[t]);
			}
			blk.grayscale = gray ? 1 : 0;
			blk.rgb_lns = fast ? 0 : rgb_lns;
			blk.alpha_lns = fast ? 0 : a_lns;
		}
	}
	WV_SYNC();

	// (ref: astcenc_entry.cpp:1017-1024) alpha-weighted colour error
	if (c.cfg->flags & (1u << 2))
	{
		WV_ONE
		{
			float alpha_scale = blk.data_max[3] * (1.0f / 65535.0f);
			blk.cw[0] = c.cfg->cw[0] * alpha_scale;
			blk.cw[1] = c.cfg->cw[1] * alpha_scale;
			blk.cw[2] = c.cfg->cw[2] * alpha_scale;
			blk.cw[3] = c.cfg->cw[3];
		}
		WV_SYNC();
	}
}

/* Alpha-scale test of one block (ref: astcenc_entry.cpp:974-1007): is any texel of the block within
 * reach of non-transparent content?  If not the block is encoded as constant zero without being read. */
WV_FN bool block_has_visible_alpha(const Ctx& c, const ImageDesc& img, unsigned int bx, unsigned int by)
{
	const int dim_x = c.root->dim_x, dim_y = c.root->dim_y;
	const unsigned int r = img.a_scale_radius;
	const float footprint = (float)((size_t)(dim_x + 2 * (r - 1)) * (size_t)(dim_y + 2 * (r - 1)));
	const float threshold = 0.9f / (255.0f * footprint);
	bool seen = false;
	WV_FOR_T(t, c.T)
	{
		unsigned int ty = (unsigned)t / (unsigned)dim_x;
		unsigned int tx = (unsigned)t - ty * (unsigned)dim_x;
		unsigned int xi = bx * (unsigned)dim_x + tx, yi = by * (unsigned)dim_y + ty;
		if (xi < img.dim_x && yi < img.dim_y && img.alpha_avg[(size_t)yi * img.dim_x + xi] > threshold) seen = true;
	}
	return wv_any(seen);
}

/* The image_block of a skipped block: everything zero, so compress_block() emits the constant block
 * (ref: astcenc_entry.cpp:1027-1034). */
WV_FN void load_transparent_block(const Ctx& c)
{
	BlkInfo& blk = c.blk();
	WV_ONE
	{
		for (int k = 0; k < 4; k++)
		{
			blk.origin[k] = 0.0f; blk.data_min[k] = 0.0f; blk.data_mean[k] = 0.0f; blk.data_max[k] = 0.0f;
			blk.cw[k] = c.cfg->cw[k];
		}
		blk.grayscale = 1;
		blk.rgb_lns = 0;
		blk.alpha_lns = 0;
	}
	WV_SYNC();
}

} } // namespace astcd::ASTC_VARIANT
