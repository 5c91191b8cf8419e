// SPDX-License-Identifier: Apache-2.0
// Symbolic block -> 128-bit physical block, including BISE packing.
//   ref: symbolic_to_physical   Source/astcenc_symbolic_physical.cpp:102-286
//        encode_ise             Source/astcenc_integer_sequence.cpp:493-648
// The bits are assembled from pieces that cannot overlap, so every BISE symbol and every header field is ORed into the
// block independently: one lane per symbol, its bit position in closed form.
#pragma once
#include "wave_ctx.h"

namespace astcd { inline namespace ASTC_VARIANT {

struct Btq { uint8_t bits, trits, quints; };

WV_FN Btq btq_of(int q)
{
	// bits / trit / quint composition of each quant level (ASTC spec table C.2.7)
	const uint8_t bits[21]   = { 1,0,2,0,1,3,1,2,4,2,3,5,3,4,6,4,5,7,5,6,8 };
	const uint8_t kind[21]   = { 0,1,0,2,1,0,2,1,0,2,1,0,2,1,0,2,1,0,2,1,0 };
	Btq b; b.bits = bits[q]; b.trits = kind[q] == 1; b.quints = kind[q] == 2;
	return b;
}

WV_FN unsigned int ise_bitcount(unsigned int count, int q)
{
	Btq b = btq_of(q);
	unsigned int total = b.bits * count;
	if (b.trits)  total += (8 * count + 4) / 5;
	if (b.quints) total += (7 * count + 2) / 3;
	return total;
}

WV_FN unsigned int quant_level_count(int q)
{
	const uint16_t lv[21] = { 2,3,4,5,6,8,10,12,16,20,24,32,40,48,64,80,96,128,160,192,256 };
	return lv[q];
}

/* OR `count` bits of `value` into a little-endian bit string of 32-bit words at bit `offset`.  The string starts out
 * zero and no two fields overlap, so the pieces can arrive in any order -- from any lane. */
WV_FN void bits_or(uint32_t* words, uint32_t value, uint32_t count, uint32_t offset)
{
	value &= count >= 32u ? 0xFFFFFFFFu : (1u << count) - 1u;
	const uint32_t w = offset >> 5, sh = offset & 31u;
	const uint32_t lo = value << sh, hi = sh ? value >> (32u - sh) : 0u;
#if WV_DEVICE
	if (lo) __hip_atomic_fetch_or(words + w, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
	if (hi) __hip_atomic_fetch_or(words + w + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
	words[w] |= lo;
	if (hi) words[w + 1] |= hi;
#endif
}

/* BISE: every symbol of the sequence on its own lane.  A symbol's place in the bit stream is a closed form of its index
 * -- a block of five trit symbols takes 5 b + 8 bits, one of three quint symbols 3 b + 7, and inside a block symbol k
 * sits after k b bits plus the (2,2,1,2,1) / (3,2,2) trit / quint bits of its predecessors -- so nothing is serial here
 * except reading the block's other symbols for the shared trit / quint word.  (ref: encode_ise,
 * astcenc_integer_sequence.cpp:493-648, which walks the symbols one after the other)
 * sym: the `count` symbols in LDS; words: the destination bit string; base: its first bit. */
WV_FN void encode_ise_lanes(const Ctx& c, int quant, int count, const uint8_t* sym, uint32_t* words, uint32_t base)
{
	const Btq b = btq_of(quant);
	const uint32_t bits = b.bits, low_mask = (1u << bits) - 1u;
	const uint8_t* trit_tab = c.table(c.root->off_integer_of_trits);
	const uint8_t* quint_tab = c.table(c.root->off_integer_of_quints);
	WV_FOR64(s, count)
	{
		const uint32_t v = sym[s];
		if (b.trits)
		{
			const uint32_t g = ((uint32_t)s * 13u) >> 6, k = (uint32_t)s - g * 5u;          // s / 5, s % 5 for s < 64
			uint32_t t[5];
			for (uint32_t j = 0; j < 5; j++) t[j] = (int)(g * 5u + j) < count ? (uint32_t)(sym[g * 5u + j] >> bits) : 0u;
			const uint32_t word = trit_tab[(((t[4] * 3 + t[3]) * 3 + t[2]) * 3 + t[1]) * 3 + t[0]];
			const uint32_t place = (0x75420u >> (4u * k)) & 0xFu;                            // 0, 2, 4, 5, 7
			const uint32_t width = (0x12122u >> (4u * k)) & 0xFu;                            // 2, 2, 1, 2, 1
			bits_or(words, (v & low_mask) | (((word >> place) & ((1u << width) - 1u)) << bits), bits + width, base + g * (5u * bits + 8u) + k * bits + place);
		}
		else if (b.quints)
		{
			const uint32_t g = ((uint32_t)s * 43u) >> 7, k = (uint32_t)s - g * 3u;          // s / 3, s % 3 for s < 64
			uint32_t q[3];
			for (uint32_t j = 0; j < 3; j++) q[j] = (int)(g * 3u + j) < count ? (uint32_t)(sym[g * 3u + j] >> bits) : 0u;
			const uint32_t word = quint_tab[(q[2] * 5 + q[1]) * 5 + q[0]];
			const uint32_t place = (0x530u >> (4u * k)) & 0xFu;                              // 0, 3, 5
			const uint32_t width = (0x223u >> (4u * k)) & 0xFu;                              // 3, 2, 2
			bits_or(words, (v & low_mask) | (((word >> place) & ((1u << width) - 1u)) << bits), bits + width, base + g * (3u * bits + 7u) + k * bits + place);
		}
		else
		{
			bits_or(words, v, bits, base + (uint32_t)s * bits);
		}
	}
}

WV_FN uint32_t bit_reverse32(uint32_t v)
{
#if WV_DEVICE
	return __builtin_bitreverse32(v);
#else
	v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
	v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
	v = ((v >> 4) & 0x0F0F0F0Fu) | ((v & 0x0F0F0F0Fu) << 4);
	v = ((v >> 8) & 0x00FF00FFu) | ((v & 0x00FF00FFu) << 8);
	return (v >> 16) | (v << 16);
#endif
}

/* The physical block of `scb` -> pcb_out[16] (ref: symbolic_to_physical, astcenc_symbolic_physical.cpp:102-286).
 * All lanes call this.  The 128 bits are assembled in LDS from pieces that do not overlap: the weight symbols (one
 * lane each) into a stream that is then bit-reversed into the top of the block, the colour symbols (one lane each)
 * behind the header, the header fields from lane 0. */
WV_FN void symbolic_to_physical(const Ctx& c, const Scb& scb, uint8_t* pcb_out)
{
	// scratch in LDS (the search regions are idle by now)
	uint32_t* block = reinterpret_cast<uint32_t*>(c.lds + c.L->uni);            // [4] + 1 word of slack
	uint32_t* stream = block + 8;                                                // [4] + 1: the weight bit stream
	uint8_t* weight_sym = reinterpret_cast<uint8_t*>(block + 16);                // [64]
	uint8_t* colour_sym = weight_sym + 64;                                       // [32]
	WV_FOR(k, 16) { block[k] = 0u; }
	WV_SYNC();

	const int block_type = wv_uniform((int)scb.block_type);
	if (block_type == SYM_BTYPE_CONST_U16 || block_type == SYM_BTYPE_CONST_F16)
	{
		// void-extent block: LDR header FC FD FF.., HDR header FC FF FF.. (spec C.2.23)
		WV_ONE
		{
			uint32_t* out = reinterpret_cast<uint32_t*>(pcb_out);
			out[0] = block_type == SYM_BTYPE_CONST_U16 ? 0xFFFFFDFCu : 0xFFFFFFFCu;
			out[1] = 0xFFFFFFFFu;
			out[2] = (uint32_t)(scb.constant_color[0] & 0xFFFF) | ((uint32_t)(scb.constant_color[1] & 0xFFFF) << 16);
			out[3] = (uint32_t)(scb.constant_color[2] & 0xFFFF) | ((uint32_t)(scb.constant_color[3] & 0xFFFF) << 16);
		}
		return;
	}

	const int partition_count = wv_uniform((int)scb.partition_count);
	const BlockMode& bm = c.block_mode(wv_uniform((int)scb.block_mode));
	const int weight_count = wv_uniform((int)c.dec_info(bm.decimation_mode).weight_count);
	const int wq = wv_uniform((int)bm.quant_mode);
	const bool dual = wv_uniform((int)bm.is_dual_plane) != 0;
	const int real_weight_count = dual ? 2 * weight_count : weight_count;
	const int bits_for_weights = (int)ise_bitcount((unsigned)real_weight_count, wq);
	const int colour_quant = wv_uniform((int)scb.quant_mode);

	// ---- symbols: weights (planes interleaved) and colour values ----
	{
		const QuantXfer& qat = c.qxfer(wq);
		const float top = (float)quant_level_count(wq) - 1.0f;
		WV_FOR64(s, real_weight_count)
		{
			const int i = dual ? s >> 1 : s, plane = dual ? s & 1 : 0;
			const float uqw = (float)scb.weights[i + plane * PLANE2_OFFSET];
			const int level = (int)((uqw / 64.0f) * top + 0.5f);
			weight_sym[s] = qat.scramble_map[level];
		}
	}
	int value_count = 0;
	for (int p = 0; p < partition_count; p++) value_count += 2 * (scb.color_formats[p] >> 2) + 2;
	value_count = wv_uniform(value_count);
	{
		const uint8_t* pack_table = c.table(c.root->off_color_uquant_to_pquant) + (colour_quant - QUANT_6) * 256;
		WV_FOR64(v, value_count)
		{
			// value v of the block = value j of partition p
			int p = 0, j = v;
			for (int q = 0; q < 3; q++)
			{
				const int n = 2 * (scb.color_formats[q] >> 2) + 2;
				if (p == q && q + 1 < partition_count && j >= n) { j -= n; p++; }
			}
			colour_sym[v] = pack_table[scb.color_values[p][j]];
		}
	}
	WV_SYNC();

	// ---- the two BISE sequences ----
	encode_ise_lanes(c, wq, real_weight_count, weight_sym, stream, 0u);
	encode_ise_lanes(c, colour_quant, value_count, colour_sym, block, partition_count == 1 ? 17u : 29u);

	// ---- header fields (ref: :180-238) ----
	WV_ONE
	{
		bits_or(block, bm.mode_index, 11, 0);
		bits_or(block, (uint32_t)(partition_count - 1), 2, 11);
		int below_weights_pos = 128 - bits_for_weights;
		if (partition_count > 1)
		{
			const uint32_t seed = reinterpret_cast<const PartitionHeader*>(c.part_rec(partition_count, scb.partition_index))->partition_index;
			bits_or(block, seed, 10, 13);
			if (scb.color_formats_matched)
			{
				bits_or(block, (uint32_t)scb.color_formats[0] << 2, 6, 23);
			}
			else
			{
				// base class = lowest format class, capped so every class bit is 0 or 1; then a class bit and the two low
				// format bits per partition
				int low_class = 4;
				for (int i = 0; i < partition_count; i++) low_class = i_min(scb.color_formats[i] >> 2, low_class);
				if (low_class == 3) low_class = 2;
				uint32_t encoded = (uint32_t)(low_class + 1);
				for (int i = 0; i < partition_count; i++)
				{
					encoded |= (uint32_t)((scb.color_formats[i] >> 2) - low_class) << (2 + i);
					encoded |= (uint32_t)(scb.color_formats[i] & 3) << (2 + partition_count + 2 * i);
				}
				const int high_size = 3 * partition_count - 4;
				bits_or(block, encoded & 0x3Fu, 6, 23);
				bits_or(block, encoded >> 6, (uint32_t)high_size, (uint32_t)(128 - bits_for_weights - high_size));
				below_weights_pos -= high_size;
			}
		}
		else
		{
			bits_or(block, scb.color_formats[0], 4, 13);
		}
		if (dual) bits_or(block, (uint32_t)scb.plane2_component, 2, (uint32_t)(below_weights_pos - 2));
	}
	WV_SYNC();

	// ---- the weight stream goes in from the top, bit-reversed; then the 16 bytes leave ----
	WV_FOR(d, 4)
	{
		reinterpret_cast<uint32_t*>(pcb_out)[d] = block[d] | bit_reverse32(stream[3 - d]);
	}
}

} } // namespace astcd::ASTC_VARIANT
