// SPDX-License-Identifier: Apache-2.0
// Decompression kernel: one wavefront per run of DECODE_BATCH consecutive blocks of a block row, texels written
// straight into the output image in HBM (astcenc_decompress_image; ref: Source/astcenc_entry.cpp:1274-1390).
#define ASTC_VARIANT v_dec
#define ASTC_ENABLE_HDR 1
#include "backend.h"
#include "wave_decode.h"
#include <hip/hip_runtime.h>
#include <cstdlib>

namespace astcd {

/* One block is a few hundred instructions that keep under half of a wavefront busy, so every wavefront takes a run of
 * DECODE_BATCH consecutive blocks of one block row and decodes it together (decode_row_batch).  The grid is (runs per block
 * row, block rows, layers of blocks): a run's place in the image needs no division.  `row0` / `layer0`: the launch covers
 * block rows [row0, row0 + gridDim.y) and layers [layer0, layer0 + gridDim.z) of the stream (astc_decode_launch). */
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(6, 8)))      // (LDS allows 5.75 waves per SIMD: keep the registers under that)
astc_decompress_blocks(const uint8_t* __restrict__ blocks, DecodeImage img, uint32_t row0, uint32_t layer0)
{
	__shared__ DecodeBatch batch;
	const uint32_t bx0 = blockIdx.x * (uint32_t)DECODE_BATCH;
	const uint32_t left = img.blocks_x - bx0;
	decode_row_batch(img, blocks, bx0, row0 + blockIdx.y, layer0 + blockIdx.z, (int)(left < (uint32_t)DECODE_BATCH ? left : (uint32_t)DECODE_BATCH), batch);
}

size_t astc_decode_tables_bytes() { return sizeof(DecodeTables); }

void astc_decode_tables_build(void* out, uint32_t block_x, uint32_t block_y, uint32_t block_z)
{
	decode_tables_build(*static_cast<DecodeTables*>(out), (int)block_x, (int)block_y, (int)block_z);
}

int astc_decode_launch(const DecodeLaunch& d)
{
	DecodeImage img;
	img.data = d.d_image;
	img.tabs = static_cast<const DecodeTables*>(d.d_tables);
	img.dim_x = d.dim_x; img.dim_y = d.dim_y; img.dim_z = d.dim_z;
	img.data_type = d.data_type;
	for (int i = 0; i < 4; i++) img.swz[i] = d.swz[i];
	img.block_x = d.block_x; img.block_y = d.block_y; img.block_z = d.block_z;
	img.blocks_x = (d.dim_x + d.block_x - 1) / d.block_x;
	img.blocks_y = (d.dim_y + d.block_y - 1) / d.block_y;
	img.blocks_z = (d.dim_z + d.block_z - 1) / d.block_z;
	img.profile = d.profile;
	decode_image_prepare(img);
	// Grid y / z hold block rows / layers of blocks, at most 65535 each: a taller stream (more than 262 140 texel rows at the
	// smallest footprint, or as many slices) is covered by several launches, each told where its rows and layers start.  The
	// image record stays the whole image's, so every address is formed from the real dimensions.
	// (ASTCENC_AMD_DECODE_GRID_LIMIT: a smaller limit for tests/test_decode.py, which cannot allocate a 262 144-row image)
	static const uint32_t limit = []() {
		const char* e = getenv("ASTCENC_AMD_DECODE_GRID_LIMIT");
		const long v = e ? strtol(e, nullptr, 10) : 0;
		return (uint32_t)(v >= 1 && v < 65535 ? v : 65535);
	}();
	const uint32_t runs = (img.blocks_x + (uint32_t)DECODE_BATCH - 1u) / (uint32_t)DECODE_BATCH;
	for (uint32_t layer0 = 0; layer0 < img.blocks_z; layer0 += limit)
	{
		const uint32_t layers = img.blocks_z - layer0 < limit ? img.blocks_z - layer0 : limit;
		for (uint32_t row0 = 0; row0 < img.blocks_y; row0 += limit)
		{
			const uint32_t rows = img.blocks_y - row0 < limit ? img.blocks_y - row0 : limit;
			hipLaunchKernelGGL(astc_decompress_blocks, dim3(runs, rows, layers), dim3(64), 0, static_cast<hipStream_t>(d.stream), d.d_blocks, img, row0, layer0);
		}
	}
	return (int)hipGetLastError();
}

} // namespace astcd
