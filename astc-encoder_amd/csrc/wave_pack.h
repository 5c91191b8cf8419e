// SPDX-License-Identifier: Apache-2.0
// Symbolic block -> 128-bit physical block, including BISE packing.
//   ref: symbolic_to_physical   Source/astcenc_symbolic_physical.cpp:102-286
//        encode_ise             Source/astcenc_integer_sequence.cpp:493-648
// Strictly sequential bit packing: executed by one lane, once per block.
#pragma once
#include "wave_ctx.h"

namespace astcd { inline namespace ASTC_VARIANT {

WV_FN void pk_write_bits(unsigned int value, unsigned int bitcount, unsigned int bitoffset, uint8_t* ptr)
{
	unsigned int mask = (1u << bitcount) - 1;
	value &= mask;
	ptr += bitoffset >> 3;
	bitoffset &= 7;
	value <<= bitoffset;
	mask <<= bitoffset;
	mask = ~mask;
	ptr[0] &= (uint8_t)mask;
	ptr[0] |= (uint8_t)value;
	ptr[1] &= (uint8_t)(mask >> 8);
	ptr[1] |= (uint8_t)(value >> 8);
}

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

/* BISE encode `count` symbols into `out` (read-modify-write) from bit_offset. out must have one
 * byte of slack beyond the last written bit. */
WV_FN void encode_ise(const Ctx& c, int quant, unsigned int count, const uint8_t* in, uint8_t* out, unsigned int bit_offset)
{
	Btq b = btq_of(quant);
	unsigned int bits = b.bits;
	unsigned int mask = (1u << bits) - 1;
	const uint8_t* trit_tab = c.table(c.root->off_integer_of_trits);
	const uint8_t* quint_tab = c.table(c.root->off_integer_of_quints);

	if (b.trits)
	{
		const uint8_t tbits[5] = { 2, 2, 1, 2, 1 };
		const uint8_t tshift[5] = { 0, 2, 4, 5, 7 };
		for (unsigned int i = 0; i < count; i += 5)
		{
			unsigned int t[5];
			for (unsigned int k = 0; k < 5; k++) t[k] = (i + k < count) ? (unsigned)(in[i + k] >> bits) : 0u;
			unsigned int T = trit_tab[(((t[4] * 3 + t[3]) * 3 + t[2]) * 3 + t[1]) * 3 + t[0]];
			for (unsigned int k = 0; k < 5 && i + k < count; k++)
			{
				unsigned int pack = (in[i + k] & mask) | (((T >> tshift[k]) & ((1u << tbits[k]) - 1)) << bits);
				pk_write_bits(pack, bits + tbits[k], bit_offset, out);
				bit_offset += bits + tbits[k];
			}
		}
	}
	else if (b.quints)
	{
		const uint8_t qbits[3] = { 3, 2, 2 };
		const uint8_t qshift[3] = { 0, 3, 5 };
		for (unsigned int i = 0; i < count; i += 3)
		{
			unsigned int q[3];
			for (unsigned int k = 0; k < 3; k++) q[k] = (i + k < count) ? (unsigned)(in[i + k] >> bits) : 0u;
			unsigned int Q = quint_tab[(q[2] * 5 + q[1]) * 5 + q[0]];
			for (unsigned int k = 0; k < 3 && i + k < count; k++)
			{
				unsigned int pack = (in[i + k] & mask) | (((Q >> qshift[k]) & ((1u << qbits[k]) - 1)) << bits);
				pk_write_bits(pack, bits + qbits[k], bit_offset, out);
				bit_offset += bits + qbits[k];
			}
		}
	}
	else
	{
		for (unsigned int i = 0; i < count; i++)
		{
			pk_write_bits(in[i], bits, bit_offset, out);
			bit_offset += bits;
		}
	}
}

WV_FN int bitrev8(int p)
{
	p = ((p & 0x0F) << 4) | ((p >> 4) & 0x0F);
	p = ((p & 0x33) << 2) | ((p >> 2) & 0x33);
	p = ((p & 0x55) << 1) | ((p >> 1) & 0x55);
	return p;
}

/* Write the physical block for `scb` into pcb[16].  Call from ONE lane. */
WV_FN void symbolic_to_physical(const Ctx& c, const Scb& scb, uint8_t* pcb_out)
{
	// byte buffers in LDS (the search regions are idle by now): run-time indexed private arrays would
	// live in scratch memory
	uint8_t* pcb = c.lds + c.L->uni;            // [18] (+2 pad)
	uint8_t* weightbuf = pcb + 20;              // [18] (+2 pad)
	uint8_t* weights = pcb + 40;                // [64]
	uint8_t* values_to_encode = pcb + 104;      // [32]
	for (int i = 0; i < 18; i++) pcb[i] = 0;

	if (scb.block_type == SYM_BTYPE_CONST_U16 || scb.block_type == SYM_BTYPE_CONST_F16)
	{
		// void-extent block: LDR header FC FD FF.., HDR header FC FF FF.. (spec C.2.23)
		pcb[0] = 0xFC;
		pcb[1] = scb.block_type == SYM_BTYPE_CONST_U16 ? 0xFD : 0xFF;
		for (int i = 2; i < 8; i++) pcb[i] = 0xFF;
		for (int i = 0; i < 4; i++)
		{
			pcb[2 * i + 8] = (uint8_t)(scb.constant_color[i] & 0xFF);
			pcb[2 * i + 9] = (uint8_t)((scb.constant_color[i] >> 8) & 0xFF);
		}
		for (int i = 0; i < 16; i++) pcb_out[i] = pcb[i];
		return;
	}

	unsigned int partition_count = scb.partition_count;
	const BlockMode& bm = c.block_mode(scb.block_mode);
	const DecimationInfo& di = c.dec_info(bm.decimation_mode);
	int weight_count = di.weight_count;
	int wq = bm.quant_mode;
	float weight_quant_levels = (float)quant_level_count(wq);
	int is_dual_plane = bm.is_dual_plane;
	const QuantXfer& qat = c.qxfer(wq);

	int real_weight_count = is_dual_plane ? 2 * weight_count : weight_count;
	int bits_for_weights = (int)ise_bitcount((unsigned)real_weight_count, wq);

	for (int i = 0; i < 18; i++) weightbuf[i] = 0;

	for (int i = 0; i < weight_count; i++)
	{
		float uqw = (float)scb.weights[i];
		float qw = (uqw / 64.0f) * (weight_quant_levels - 1.0f);
		int qwi = (int)(qw + 0.5f);
		if (is_dual_plane)
		{
			weights[2 * i] = qat.scramble_map[qwi];
			uqw = (float)scb.weights[i + PLANE2_OFFSET];
			qw = (uqw / 64.0f) * (weight_quant_levels - 1.0f);
			qwi = (int)(qw + 0.5f);
			weights[2 * i + 1] = qat.scramble_map[qwi];
		}
		else
		{
			weights[i] = qat.scramble_map[qwi];
		}
	}

	encode_ise(c, wq, (unsigned)real_weight_count, weights, weightbuf, 0);

	for (int i = 0; i < 16; i++)
	{
		pcb[i] = (uint8_t)bitrev8(weightbuf[15 - i]);
	}

	pk_write_bits(bm.mode_index, 11, 0, pcb);
	pk_write_bits(partition_count - 1, 2, 11, pcb);

	int below_weights_pos = 128 - bits_for_weights;

	if (partition_count > 1)
	{
		unsigned int seed = reinterpret_cast<const PartitionHeader*>(c.part_rec((int)partition_count, scb.partition_index))->partition_index;
		pk_write_bits(seed, 6, 13, pcb);
		pk_write_bits(seed >> 6, 4, 19, pcb);

		if (scb.color_formats_matched)
		{
			pk_write_bits((unsigned)scb.color_formats[0] << 2, 6, 23, pcb);
		}
		else
		{
			// base class = lowest format class, capped so every class bit is 0 or 1
			int low_class = 4;
			for (unsigned int i = 0; i < partition_count; i++)
			{
				int cls = scb.color_formats[i] >> 2;
				low_class = i_min(cls, low_class);
			}
			if (low_class == 3) low_class = 2;

			int encoded_type = low_class + 1;
			int bitpos = 2;
			for (unsigned int i = 0; i < partition_count; i++)
			{
				int classbit = (scb.color_formats[i] >> 2) - low_class;
				encoded_type |= classbit << bitpos;
				bitpos++;
			}
			for (unsigned int i = 0; i < partition_count; i++)
			{
				int lowbits = scb.color_formats[i] & 3;
				encoded_type |= lowbits << bitpos;
				bitpos += 2;
			}

			int low_part = encoded_type & 0x3F;
			int high_part = encoded_type >> 6;
			int high_size = (3 * (int)partition_count) - 4;
			int high_pos = 128 - bits_for_weights - high_size;
			pk_write_bits((unsigned)low_part, 6, 23, pcb);
			pk_write_bits((unsigned)high_part, (unsigned)high_size, (unsigned)high_pos, pcb);
			below_weights_pos -= high_size;
		}
	}
	else
	{
		pk_write_bits(scb.color_formats[0], 4, 13, pcb);
	}

	if (is_dual_plane)
	{
		pk_write_bits((unsigned)scb.plane2_component, 2, (unsigned)(below_weights_pos - 2), pcb);
	}

	int valuecount = 0;
	const uint8_t* pack_table = c.table(c.root->off_color_uquant_to_pquant) + (scb.quant_mode - QUANT_6) * 256;
	for (unsigned int i = 0; i < partition_count; i++)
	{
		int vals = 2 * (scb.color_formats[i] >> 2) + 2;
		for (int j = 0; j < vals; j++)
		{
			values_to_encode[j + valuecount] = pack_table[scb.color_values[i][j]];
		}
		valuecount += vals;
	}

	encode_ise(c, scb.quant_mode, (unsigned)valuecount, values_to_encode, pcb, partition_count == 1 ? 17 : 29);

	for (int i = 0; i < 16; i++) pcb_out[i] = pcb[i];
}

} } // namespace astcd::ASTC_VARIANT
