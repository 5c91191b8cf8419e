// SPDX-License-Identifier: Apache-2.0
// Test infrastructure: the batched decoder's table-driven routines -- BISE group decode (group_layout / group_symbols /
// group_symbols_split, *_unquant_lut), the packed quant-level constants, the header parse from DecodeTables -- against
// the arithmetic per-element routines it replaces (ise_symbol, unquant_*_symbol), which are the ones the single-block
// decoder and astcenc_get_block_info use and which are pinned to the reference decoder by tests/test_decode.py.
//   g++ -std=c++17 -O1 -DASTC_WAVE_EMU=1 -I astc-encoder_amd/csrc tests/harness/ise_lut_check.cpp -o ise_lut_check
#define ASTC_VARIANT v_check
#define ASTC_ENABLE_HDR 1
#include "backend.h"
#include "wave_decode.h"
#include <cstdio>
#include <cstdint>

using namespace astcd;

int main()
{
	uint64_t x = 0x9E3779B97F4A7C15ull;
	auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (uint32_t)(x >> 16); };
	long checked = 0, bad = 0;
	for (int quant = 0; quant <= 20; quant++)
	{
		const Btq q = btq_of(quant);
		const int kind = q.trits ? 1 : q.quints ? 2 : 0;
		// the packed level description and the division-free group count of the batched decoder
		{
			const uint32_t pk = btq_packed(quant);
			checked++;
			if ((int)(pk & 15u) != q.bits || (int)(pk >> 4) != kind) { bad++; fprintf(stderr, "btq_packed %d\n", quant); }
			for (int count = 1; count < 128; count++)
			{
				const int per = ise_group_size(kind);
				checked++;
				if (ise_group_count(count, kind) != (count + per - 1) / per) { bad++; fprintf(stderr, "group count %d %d\n", kind, count); }
			}
		}
		// weight levels: the straight-line group decode (group_layout / group_symbols) on a stream cut off at its length (what decode_row_batch stores)
		if (quant <= 11)
		{
			for (int rep = 0; rep < 4000; rep++)
			{
				Bits128 b;
				const int count = 1 + (int)(rnd() % 64);
				const int len = (int)ise_bitcount((unsigned)count, quant);
				if (len > 96) continue;
				uint32_t ws[4];
				for (int k = 0; k < 4; k++) b.w[k] = rnd();
				for (int k = 0; k < 3; k++)
				{
					const int left = len - 32 * k;
					ws[k] = left >= 32 ? b.w[k] : left <= 0 ? 0u : b.w[k] & ((1u << left) - 1u);
				}
				ws[3] = 0u;
				const int per = ise_group_size(kind);
				const int glen = kind == 1 ? 5 * q.bits + 8 : kind == 2 ? 3 * q.bits + 7 : 4 * q.bits;
				for (int group = 0; group * per < count; group++)
				{
					uint32_t sym[5];
					group_symbols(group_layout(q.bits, kind), bits_window32(ws, group * glen), sym);
					for (int e = 0; e < 5; e++) if (sym[e] > 31u) { bad++; fprintf(stderr, "weight group: symbol out of range\n"); }
					for (int e = 0; e < per && group * per + e < count; e++)
					{
						checked++;
						if ((int)sym[e] != ise_symbol(b, 0, quant, count, group * per + e))
						{
							if (bad++ < 10) fprintf(stderr, "weight group: quant %d count %d group %d element %d\n", quant, count, group, e);
						}
					}
				}
			}
		}
		// colour levels: the straight-line group decode over two 32-bit windows (group_layout_split / group_symbols_split) on a stream cut off at the end of the colour values
		if (quant >= 4)
		{
			for (int rep = 0; rep < 4000; rep++)
			{
				Bits128 b;
				const int count = 2 * (1 + (int)(rnd() % 9));
				const int start = (rep & 1) ? 17 : 29;
				const int end = start + (int)ise_bitcount((unsigned)count, quant);
				if (end > 128 - 24) continue;
				uint32_t cs[5];
				for (int k = 0; k < 4; k++) b.w[k] = rnd();
				for (int k = 0; k < 4; k++)
				{
					const int left = end - 32 * k;
					cs[k] = left >= 32 ? b.w[k] : left <= 0 ? 0u : b.w[k] & ((1u << left) - 1u);
				}
				cs[4] = 0u;
				const int per = ise_group_size(kind);
				const int glen = kind == 1 ? 5 * q.bits + 8 : kind == 2 ? 3 * q.bits + 7 : 4 * q.bits;
				for (int group = 0; group * per < count; group++)
				{
					uint32_t sym[5];
					int second_at;
					const GroupLayout L = group_layout_split(q.bits, kind, second_at);
					group_symbols_split(L, bits_window32(cs, start + group * glen), bits_window32(cs, start + group * glen + second_at), sym);
					for (int e = 0; e < per && group * per + e < count; e++)
					{
						checked++;
						if ((int)sym[e] != ise_symbol(b, start, quant, count, group * per + e))
						{
							if (bad++ < 10) fprintf(stderr, "colour group: quant %d count %d group %d element %d\n", quant, count, group, e);
						}
					}
					for (int e = 0; e < 5; e++) if (sym[e] > 255u) { bad++; fprintf(stderr, "colour group: symbol out of range\n"); }
				}
			}
		}
		// unquantization tables: every symbol the level can produce
		for (int v = 0; v < 256; v++)
		{
			const int top = v >> q.bits;
			const bool valid = q.trits ? top < 3 : q.quints ? top < 5 : v < (1 << q.bits);
			if (!valid) continue;
			if (quant <= 11 && v < 32) { checked++; if (weight_unquant_lut(quant, v) != unquant_weight_symbol(v, quant)) { bad++; fprintf(stderr, "weight unquant %d %d\n", quant, v); } }
			if (quant >= 4) { checked++; if (color_unquant_lut(quant, v) != unquant_color_symbol(v, quant)) { bad++; fprintf(stderr, "colour unquant %d %d\n", quant, v); } }
		}
	}
	// the table-driven header parse of the batched decoder (DecodeTables, built from the arithmetic routines) against the
	// arithmetic one, field by field, on random blocks -- reserved modes, void extents, every partition count
	{
		const int foot[8][3] = { { 4, 4, 1 }, { 6, 6, 1 }, { 8, 5, 1 }, { 10, 6, 1 }, { 12, 12, 1 }, { 3, 3, 3 }, { 4, 4, 3 }, { 6, 6, 6 } };
		static DecodeTables tabs;
		for (int f = 0; f < 8; f++)
		{
			decode_tables_build(tabs, foot[f][0], foot[f][1], foot[f][2]);
			for (int rep = 0; rep < 40000; rep++)
			{
				Bits128 b;
				for (int k = 0; k < 4; k++) b.w[k] = rnd();
				if (rep % 7 == 0) b.w[0] = (b.w[0] & ~0x1FFu) | 0x1FCu;          // void extent
				if (rep % 5 == 1) b.w[0] &= ~0x1800u;                            // one partition
				if (rep % 9 == 2) b.w[0] = (b.w[0] & ~3u) | 1u;                  // the common rows of the mode table
				const BlockHeader x = parse_block_header(b, foot[f][0], foot[f][1], foot[f][2], &tabs);
				const BlockHeader y = parse_block_header(b, foot[f][0], foot[f][1], foot[f][2], nullptr);
				checked++;
				bool same = x.error == y.error && x.constant == y.constant;
				if (same && !x.error && x.constant)
				{
					same = x.constant_f16 == y.constant_f16;
					for (int k = 0; k < 4; k++) same = same && x.const_color[k] == y.const_color[k];
				}
				if (same && !x.error && !x.constant)
				{
					same = x.wx == y.wx && x.wy == y.wy && x.wz == y.wz && x.wquant == y.wquant && x.wbits == y.wbits && x.dual == y.dual && x.parts == y.parts &&
					       x.seed == y.seed && x.plane2 == y.plane2 && x.nvals == y.nvals && x.cquant == y.cquant && x.color_start == y.color_start;
					for (int k = 0; k < x.parts; k++) same = same && x.fmt[k] == y.fmt[k];
				}
				if (!same && bad++ < 10) fprintf(stderr, "header: footprint %dx%dx%d block %08x %08x %08x %08x\n", foot[f][0], foot[f][1], foot[f][2], b.w[0], b.w[1], b.w[2], b.w[3]);
			}
		}
	}
	printf("%ld checks, %ld mismatches\n", checked, bad);
	return bad ? 1 : 0;
}
