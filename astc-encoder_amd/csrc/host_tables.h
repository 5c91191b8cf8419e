// SPDX-License-Identifier: Apache-2.0
// Host-side interface of the table builder (host_tables.cpp).
#pragma once
#include <vector>
#include <stdint.h>
#include "astc_tables.h"

namespace astcd {

/* Host-only side tables (not needed by the kernels). */
struct HostTables {
	// seed -> packed index of the deduplicated partition table, per partition count 2..4
	// (ref: block_size_descriptor::partitioning_packed_index, astcenc_internal.h:602)
	std::vector<uint16_t> partition_packed_index;
};

/* Build the table blob for a 2D (block_z == 1) or 3D footprint.
 *   partition_count_cutoff : config.tune_partition_count_limit
 *   mode_cutoff            : config.tune_block_mode_limit / 100
 * (ref: init_block_size_descriptor, astcenc_block_sizes.cpp:1199) */
bool build_tables(unsigned int block_x, unsigned int block_y, unsigned int block_z, unsigned int partition_count_cutoff,
                  float mode_cutoff, std::vector<uint8_t>& blob, HostTables& host);

bool is_legal_2d_block_size(unsigned int x, unsigned int y);
bool is_legal_3d_block_size(unsigned int x, unsigned int y, unsigned int z);
unsigned int ise_sequence_bitcount(unsigned int count, unsigned int quant);
unsigned int get_quant_level(unsigned int quant);

} // namespace astcd
