// SPDX-License-Identifier: Apache-2.0
// Decompression kernel: one wavefront per ASTC block, texels written straight into the output image
// in HBM (astcenc_decompress_image; ref: Source/astcenc_entry.cpp:1274-1390).
#define ASTC_VARIANT v_dec
#define ASTC_ENABLE_HDR 1
#include "backend.h"
#include "wave_decode.h"
#include <hip/hip_runtime.h>

namespace astcd {

/* Blocks per workgroup.  One block is a few microseconds of work for one wavefront, so a launch with one
 * workgroup per block is bound by the rate at which workgroups can be dispatched, not by the decoding:
 * each wavefront takes a run of consecutive blocks instead. */
constexpr uint32_t DECODE_BLOCKS_PER_WAVE = 8;

__global__ void __launch_bounds__(64)
astc_decompress_blocks(const uint8_t* __restrict__ blocks, DecodeImage img, uint32_t num_blocks)
{
	__shared__ DecodeScratch scratch;
	const uint32_t first = blockIdx.x * DECODE_BLOCKS_PER_WAVE;
	for (uint32_t b = first; b < first + DECODE_BLOCKS_PER_WAVE && b < num_blocks; b++)
	{
		const uint32_t row = b / img.blocks_x;
		const uint32_t bx = b - row * img.blocks_x;
		const uint32_t bz = row / img.blocks_y;
		const uint32_t by = row - bz * img.blocks_y;
		decode_block(img, blocks + (size_t)b * 16, bx, by, bz, scratch);
		WV_SYNC();          // the scratch is reused by the next block
	}
}

int astc_decode_launch(const DecodeLaunch& d)
{
	DecodeImage img;
	img.data = d.d_image;
	img.dim_x = d.dim_x; img.dim_y = d.dim_y; img.dim_z = d.dim_z;
	img.data_type = d.data_type;
	for (int i = 0; i < 4; i++) img.swz[i] = d.swz[i];
	img.block_x = d.block_x; img.block_y = d.block_y; img.block_z = d.block_z;
	img.blocks_x = (d.dim_x + d.block_x - 1) / d.block_x;
	img.blocks_y = (d.dim_y + d.block_y - 1) / d.block_y;
	img.blocks_z = (d.dim_z + d.block_z - 1) / d.block_z;
	img.profile = d.profile;
	const uint32_t n = img.blocks_x * img.blocks_y * img.blocks_z;
	hipLaunchKernelGGL(astc_decompress_blocks, dim3((n + DECODE_BLOCKS_PER_WAVE - 1) / DECODE_BLOCKS_PER_WAVE), dim3(64), 0, static_cast<hipStream_t>(d.stream), d.d_blocks, img, n);
	return (int)hipGetLastError();
}

} // namespace astcd
