// SPDX-License-Identifier: Apache-2.0
// Decompression kernel: one wavefront per run of DECODE_BATCH consecutive blocks of a block row, texels written
// straight into the output image in HBM (astcenc_decompress_image; ref: Source/astcenc_entry.cpp:1274-1390).
#define ASTC_VARIANT v_dec
#define ASTC_ENABLE_HDR 1
#include "backend.h"
#include "wave_decode.h"
#include <hip/hip_runtime.h>

namespace astcd {

/* One block is a few hundred instructions that keep under half of a wavefront busy, so every wavefront takes runs of
 * DECODE_BATCH consecutive blocks of one block row and decodes each run together (decode_row_batch) -- DECODE_RUNS_PER_WAVE
 * of them one after the other (one, as measured: wave_decode.h).  The grid is (waves per block row, block rows, layers of blocks): a run's place in the image needs no division. */
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(6, 8)))      // (LDS allows 5.75 waves per SIMD: keep the registers under that)
astc_decompress_blocks(const uint8_t* __restrict__ blocks, DecodeImage img)
{
	__shared__ DecodeBatch batch;
	for (int run = 0; run < DECODE_RUNS_PER_WAVE; run++)
	{
		const uint32_t bx0 = (blockIdx.x * (uint32_t)DECODE_RUNS_PER_WAVE + (uint32_t)run) * (uint32_t)DECODE_BATCH;
		if (bx0 >= img.blocks_x) break;
		const uint32_t left = img.blocks_x - bx0;
		decode_row_batch(img, blocks, bx0, blockIdx.y, blockIdx.z, (int)(left < (uint32_t)DECODE_BATCH ? left : (uint32_t)DECODE_BATCH), batch);
	}
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
	const uint32_t per_wave = (uint32_t)(DECODE_BATCH * DECODE_RUNS_PER_WAVE);
	// Grid y / z hold block rows / layers, at most 65535 each.  A taller 2D image (more than 262 140 texel rows at the
	// smallest footprint) is decoded in bands of 65535 block rows: a band is an image of its own -- its rows, its blocks.
	if (img.blocks_z > 65535u || (img.blocks_y > 65535u && img.blocks_z > 1u)) return (int)hipErrorInvalidConfiguration;
	const size_t texel_bytes = d.data_type == 0 ? 4 : d.data_type == 1 ? 8 : 16;
	for (uint32_t row0 = 0; row0 < img.blocks_y; row0 += 65535u)
	{
		DecodeImage band = img;
		const uint32_t rows = img.blocks_y - row0 < 65535u ? img.blocks_y - row0 : 65535u;
		const uint32_t y0 = row0 * d.block_y;
		band.blocks_y = rows;
		band.dim_y = d.dim_y - y0 < rows * d.block_y ? d.dim_y - y0 : rows * d.block_y;
		band.data = static_cast<uint8_t*>(d.d_image) + (size_t)y0 * d.dim_x * texel_bytes;
		const uint8_t* blocks = d.d_blocks + (size_t)row0 * img.blocks_x * 16;
		const dim3 grid((img.blocks_x + per_wave - 1) / per_wave, rows, img.blocks_z);
		hipLaunchKernelGGL(astc_decompress_blocks, grid, dim3(64), 0, static_cast<hipStream_t>(d.stream), blocks, band);
	}
	return (int)hipGetLastError();
}

} // namespace astcd
