// SPDX-License-Identifier: Apache-2.0
// Alpha-scale pre-pass (config.a_scale_radius, the CLI's -a option): per-texel average of alpha over a
// (2r+1)^2 window, used to skip blocks that are fully transparent everywhere within reach of a filter.
//   ref: compute_pixel_region_variance / init_compute_averages   Source/astcenc_compute_variance.cpp:103-556
//        brent_kung_prefix_sum                                    :48-100
//        per-block test                                           Source/astcenc_entry.cpp:974-1034
//
// The reference works on 32x32 texel regions: it gathers the (edge-clamped) alpha of the padded
// region, builds a summed-area table with Brent-Kung prefix sums along x and then y, and reads each
// window sum from four corners.  The same tiling, the same pairing of additions in the prefix sums
// and the same corner expression are kept, so the float averages are identical.  One wavefront per
// tile: lanes own rows, then columns, then texels.
#pragma once
#include "wave.h"
#include "astc_tables.h"

namespace astcd { inline namespace ASTC_VARIANT {

constexpr int ALPHA_TILE = 32;                 // ref: max_blk_size_xy for 2D images
constexpr int ALPHA_MAX_RADIUS = 80;           // padded tile (32 + 2r + 1)^2 floats must fit the 160 KiB of LDS

struct AlphaJob {
	const void* image;        // tightly packed RGBA rows
	float*      averages;     // [dim_y][dim_x]
	uint32_t dim_x, dim_y, data_type, swz_a, radius;
};

/* Brent-Kung inclusive prefix sum of `items` floats `stride` apart, additions paired exactly as the reference. */
WV_FN void brent_kung(float* d, int items, int stride)
{
	if (items < 2) return;
	int lc_stride = 2, log2_stride = 1;
	do
	{
		const int step = lc_stride >> 1;
		int iters = items >> log2_stride;
		float* da = d + (lc_stride - 1) * stride;
		const int ofs = -step * stride, ofs_stride = stride << log2_stride;
		while (iters) { *da = *da + da[ofs]; da += ofs_stride; iters--; }
		log2_stride += 1;
		lc_stride <<= 1;
	} while (lc_stride <= items);
	do
	{
		log2_stride -= 1;
		lc_stride >>= 1;
		const int step = lc_stride >> 1;
		int iters = (items - step) >> log2_stride;
		float* da = d + (step + lc_stride - 1) * stride;
		const int ofs = -step * stride, ofs_stride = stride << log2_stride;
		while (iters) { *da = *da + da[ofs]; da += ofs_stride; iters--; }
	} while (lc_stride > 2);
}

/* Averages of tile (tx, ty).  `buf` holds (ALPHA_TILE + 2 * radius + 1)^2 floats of scratch. */
WV_FN void alpha_average_tile(const AlphaJob& j, uint32_t tx, uint32_t ty, float* buf)
{
	const int r = (int)j.radius, kd = 2 * r + 1;
	const int off_x = (int)tx * ALPHA_TILE, off_y = (int)ty * ALPHA_TILE;
	const int size_x = i_min(ALPHA_TILE, (int)j.dim_x - off_x), size_y = i_min(ALPHA_TILE, (int)j.dim_y - off_y);
	const int pad_x = size_x + kd, pad_y = size_y + kd;

	// gather: row 0 / column 0 are zero, the rest is alpha of the edge-clamped source texel (ref: :154-364)
	WV_FOR(k, pad_x * pad_y)
	{
		const int y = k / pad_x, x = k - y * pad_x;
		float v = 0.0f;
		if (x > 0 && y > 0)
		{
			int xs = (x - 1) + off_x, ys = (y - 1) + off_y;
			xs = xs <= r ? 0 : xs - r;
			ys = ys <= r ? 0 : ys - r;
			xs = i_min(xs, (int)j.dim_x - 1);
			ys = i_min(ys, (int)j.dim_y - 1);
			const size_t at = ((size_t)ys * j.dim_x + (size_t)xs) * 4;
			if (j.swz_a == 4) v = 0.0f;
			else if (j.swz_a == 5) v = j.data_type == 0 ? 255.0f * (1.0f / 255.0f) : 1.0f;
			else if (j.data_type == 0) v = (float)static_cast<const uint8_t*>(j.image)[at + j.swz_a] * (1.0f / 255.0f);
			else if (j.data_type == 1) v = half_to_float(static_cast<const uint16_t*>(j.image)[at + j.swz_a]);
			else v = static_cast<const float*>(j.image)[at + j.swz_a];
		}
		buf[k] = v;
	}
	WV_SYNC();
	// summed-area table (ref: :394-410): prefix sums along x for every row, then along y for every column
	WV_FOR(y, pad_y - 1) { brent_kung(buf + (y + 1) * pad_x + 1, pad_x - 1, 1); }
	WV_SYNC();
	WV_FOR(x, pad_x - 1) { brent_kung(buf + pad_x + (x + 1), pad_y - 1, pad_x); }
	WV_SYNC();
	// window sums from four corners (ref: :479-502)
	const float kdim = (float)kd;
	const float rsamples = 1.0f / (kdim * kdim);
	WV_FOR(k, size_x * size_y)
	{
		const int y = k / size_x, x = k - y * size_x;
		const int y_low = y, y_high = y + kd, x_low = x, x_high = x + kd;
		float vasum = buf[y_low * pad_x + x_low] - buf[y_low * pad_x + x_high] - buf[y_high * pad_x + x_low] + buf[y_high * pad_x + x_high];
		j.averages[(size_t)(y + off_y) * j.dim_x + (size_t)(x + off_x)] = vasum * rsamples;
	}
	WV_SYNC();
}

} } // namespace astcd::ASTC_VARIANT
