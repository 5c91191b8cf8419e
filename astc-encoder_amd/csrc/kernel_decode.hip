// SPDX-License-Identifier: Apache-2.0
// Decompression kernel: one wavefront per ASTC block, texels written straight into the output image
// in HBM (astcenc_decompress_image; ref: Source/astcenc_entry.cpp:1274-1390).
#define ASTC_VARIANT v_dec
#define ASTC_ENABLE_HDR 1
#include "backend.h"
#include "wave_decode.h"
#include <hip/hip_runtime.h>

namespace astcd {

/* One block is a few microseconds of work that keeps under half of a wavefront busy, so every wavefront
 * takes a run of DECODE_BATCH consecutive blocks and decodes them together (decode_block_batch). */
__global__ void __launch_bounds__(64)
astc_decompress_blocks(const uint8_t* __restrict__ blocks, DecodeImage img, uint32_t num_blocks)
{
	__shared__ DecodeBatch batch;
	const uint32_t first = blockIdx.x * (uint32_t)DECODE_BATCH;
	if (first >= num_blocks) return;
	const uint32_t left = num_blocks - first;
	decode_block_batch(img, blocks, first, (int)(left < (uint32_t)DECODE_BATCH ? left : (uint32_t)DECODE_BATCH), batch);
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
	decode_image_prepare(img);
	const uint32_t n = img.blocks_x * img.blocks_y * img.blocks_z;
	hipLaunchKernelGGL(astc_decompress_blocks, dim3((n + (uint32_t)DECODE_BATCH - 1) / (uint32_t)DECODE_BATCH), dim3(64), 0, static_cast<hipStream_t>(d.stream), d.d_blocks, img, n);
	return (int)hipGetLastError();
}

} // namespace astcd
