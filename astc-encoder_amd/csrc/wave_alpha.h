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
// tile: lanes own rows, then columns, then texels.  Any radius: the padded tile lives in LDS while it fits
// (radius <= 80 for one slice) and in a per-workgroup slice of HBM beyond that.
#pragma once
#include "wave.h"
#include "astc_tables.h"

namespace astcd { inline namespace ASTC_VARIANT {

constexpr int ALPHA_TILE = 32;                 // ref: max_blk_size_xy for 2D images
constexpr int ALPHA_TILE_3D = 16;              // ... and for images with more than one slice (astcenc_compute_variance.cpp:523)
constexpr int ALPHA_MAX_SLICES = 16;           // z extent of a region of a multi-slice image (:525-526)

struct AlphaJob {
	const void* image;        // tightly packed RGBA rows, dim_z slices back to back
	float*      averages;     // [dim_y][dim_x]
	uint32_t dim_x, dim_y, dim_z, data_type, swz_a, radius;
};

/* The padded region's floats live in LDS when they fit and in a per-workgroup slice of HBM when they do not (large
 * radii): the hand-off between lanes then has to wait for the stores, which the wave-level fence of WV_SYNC() does not. */
#if WV_DEVICE
#define ALPHA_SYNC() __syncthreads()
#else
#define ALPHA_SYNC() ((void)0)
#endif

WV_FN int alpha_tile_size(const AlphaJob& j) { return j.dim_z > 1 ? ALPHA_TILE_3D : ALPHA_TILE; }
/* Floats of scratch one region needs. */
WV_FN size_t alpha_scratch_floats(const AlphaJob& j)
{
	const size_t kd = 2 * (size_t)j.radius + 1, pad = (size_t)alpha_tile_size(j) + kd;
	const size_t planes = j.dim_z > 1 ? (size_t)i_min((int)j.dim_z, ALPHA_MAX_SLICES) + kd : 1;
	return pad * pad * planes;
}

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

/* Averages of the region at tile (tx, ty) -> j.averages.  `buf`: alpha_scratch_floats(j) floats.
 *
 * One slice: the (2r+1)^2 box average of a 32 x 32 tile.  More than one slice (and a 2D footprint, the only case in
 * which the block loop looks at the averages, astcenc_entry.cpp:975): the reference averages over a (2r+1)^3 box in
 * 16 x 16 x min(dim_z, 16) regions and then reads the result with `y * dim_x + x` for every slice, i.e. it uses the
 * averages around slice 0 for the whole stack.  Those are what is computed here: the summed-volume table of the first
 * region in z, prefix sums along x, y, then z with the reference's pairing, and the eight-corner expression for z = 0. */
WV_FN void alpha_average_tile(const AlphaJob& j, uint32_t tx, uint32_t ty, float* buf)
{
	const bool have_z = j.dim_z > 1;
	const int tile = alpha_tile_size(j);
	const int r = (int)j.radius, kd = 2 * r + 1;
	const int off_x = (int)tx * tile, off_y = (int)ty * tile;
	const int size_x = i_min(tile, (int)j.dim_x - off_x), size_y = i_min(tile, (int)j.dim_y - off_y);
	const int size_z = have_z ? i_min((int)j.dim_z, ALPHA_MAX_SLICES) : 1;
	const int pad_x = size_x + kd, pad_y = size_y + kd, pad_z = have_z ? size_z + kd : 1;
	const int zd_start = have_z ? 1 : 0;
	const int plane = pad_x * pad_y;
	const size_t texel_bytes = j.data_type == 0 ? 4 : j.data_type == 1 ? 8 : 16;
	const size_t slice_bytes = (size_t)j.dim_x * j.dim_y * texel_bytes;

	// gather: row 0 / column 0 (and, with slices, plane 0) are zero, the rest is alpha of the edge-clamped source texel
	// (ref: :154-392)
	for (int z = 0; z < pad_z; z++)
	{
		int zs = 0;
		if (have_z)
		{
			zs = z - zd_start;
			zs = zs <= r ? 0 : zs - r;
			zs = i_min(zs, (int)j.dim_z - 1);
		}
		const uint8_t* slice = static_cast<const uint8_t*>(j.image) + (size_t)zs * slice_bytes;
		WV_FOR(k, plane)
		{
			const int y = k / pad_x, x = k - y * pad_x;
			float v = 0.0f;
			if (x > 0 && y > 0 && z >= zd_start)
			{
				int xs = (x - 1) + off_x, ys = (y - 1) + off_y;
				xs = xs <= r ? 0 : xs - r;
				ys = ys <= r ? 0 : ys - r;
				xs = i_min(xs, (int)j.dim_x - 1);
				ys = i_min(ys, (int)j.dim_y - 1);
				const size_t at = ((size_t)ys * j.dim_x + (size_t)xs) * 4;
				if (j.swz_a == 4) v = 0.0f;
				else if (j.swz_a == 5) v = j.data_type == 0 ? 255.0f * (1.0f / 255.0f) : 1.0f;
				else if (j.data_type == 0) v = (float)slice[at + j.swz_a] * (1.0f / 255.0f);
				else if (j.data_type == 1) v = half_to_float(reinterpret_cast<const uint16_t*>(slice)[at + j.swz_a]);
				else v = reinterpret_cast<const float*>(slice)[at + j.swz_a];
			}
			buf[(size_t)z * plane + k] = v;
		}
	}
	ALPHA_SYNC();
	// summed-area tables (ref: :394-420): prefix sums along x for every row, along y for every column, then along z
	WV_FOR(k, (pad_z - zd_start) * (pad_y - 1))
	{
		const int z = zd_start + k / (pad_y - 1), y = k % (pad_y - 1);
		brent_kung(buf + (size_t)z * plane + (y + 1) * pad_x + 1, pad_x - 1, 1);
	}
	ALPHA_SYNC();
	WV_FOR(k, (pad_z - zd_start) * (pad_x - 1))
	{
		const int z = zd_start + k / (pad_x - 1), x = k % (pad_x - 1);
		brent_kung(buf + (size_t)z * plane + pad_x + (x + 1), pad_y - 1, pad_x);
	}
	ALPHA_SYNC();
	if (have_z)
	{
		WV_FOR(k, (pad_y - 1) * (pad_x - 1))
		{
			const int y = 1 + k / (pad_x - 1), x = 1 + k % (pad_x - 1);
			brent_kung(buf + (size_t)plane + y * pad_x + x, pad_z - 1, plane);
		}
		ALPHA_SYNC();
	}
	// window sums from the corners (ref: :441-502); with slices: around slice 0
	const float kdim = (float)kd;
	const float rsamples = have_z ? 1.0f / (kdim * kdim * kdim) : 1.0f / (kdim * kdim);
	const float* lo_plane = buf;                                     // z_low = 0 (all zeros when there are slices)
	const float* hi_plane = buf + (size_t)(have_z ? kd : 0) * plane;  // z_high = 2r + 1
	WV_FOR(k, size_x * size_y)
	{
		const int y = k / size_x, x = k - y * size_x;
		const int y_low = y, y_high = y + kd, x_low = x, x_high = x + kd;
		float vasum = hi_plane[y_low * pad_x + x_low] - hi_plane[y_low * pad_x + x_high] - hi_plane[y_high * pad_x + x_low] + hi_plane[y_high * pad_x + x_high];
		if (have_z)
		{
			vasum = vasum - (lo_plane[y_low * pad_x + x_low] - lo_plane[y_low * pad_x + x_high] - lo_plane[y_high * pad_x + x_low] + lo_plane[y_high * pad_x + x_high]);
		}
		j.averages[(size_t)(y + off_y) * j.dim_x + (size_t)(x + off_x)] = vasum * rsamples;
	}
	ALPHA_SYNC();
}

} } // namespace astcd::ASTC_VARIANT
