// SPDX-License-Identifier: Apache-2.0
// Block decoder: 128-bit physical block -> texels of the output image.
//   ref: physical_to_symbolic        Source/astcenc_symbolic_physical.cpp:291-556
//        decode_ise                  Source/astcenc_integer_sequence.cpp:651-739
//        decompress_symbolic_block   Source/astcenc_decompress_symbolic.cpp:170-308
//        unpack_weights / lerp_color_int / decode_texel       :37-155
//        store_image_block           Source/astcenc_image.cpp:345-573
//
// One wavefront decodes a run of 32 consecutive blocks of a block row (decode_row_batch): headers on a lane per block, BISE
// groups of weights and colour values and endpoint pairs on two lanes per block, then the texels of the image rows the run
// covers, a lane per column.  The decoder must accept every legal block mode and partitioning, not just the ones a
// compression preset selects: grid weights are infilled with the format's arithmetic rule and texels are assigned to
// partitions with the hash function.  What it reads from tables: per footprint, what a block mode field says and which colour
// quant level a bit budget affords (DecodeTables, built on the host from the arithmetic routines below); as format
// constants, the trit / quint groups and the unquantized weight and colour values (decode_luts.inc, generated from the
// arithmetic routines below).  The single-block routines (parse_block_header without tables, unpack_block_payload) are what
// astcenc_get_block_info runs on the host.
#pragma once
#include "wave_color.h"
#include "wave_load.h"
#include "wave_pack.h"

namespace astcd { inline namespace ASTC_VARIANT {

struct DecodeTables;

struct DecodeImage {
	void*    data;            // tightly packed RGBA rows of data_type, dim_z slices back to back
	const DecodeTables* tabs; // per-footprint tables of the batched decoder (decode_tables_build), device memory
	uint32_t dim_x, dim_y, dim_z;
	uint32_t data_type;       // astcenc_type
	uint32_t swz[4];          // astcenc_swz per output channel
	uint32_t blocks_x, blocks_y, blocks_z;
	uint32_t block_x, block_y, block_z;
	uint32_t profile;         // astcenc_profile
	// multiply-shift inverses (the device has no integer divide); filled by decode_image_prepare()
	uint32_t t_inv24;         // ceil(2^24 / texels per block):   j / T == (j * t_inv24) >> 24
	uint32_t bx_inv16;        // ceil(2^16 / block_x):            t / block_x == (t * bx_inv16) >> 16 for t < 256
	uint32_t bxy_inv16;       // ceil(2^16 / (block_x * block_y))
	uint32_t ds, dt, dr;      // texel -> grid scale factors (1024 + block / 2) / (block - 1) (ref: astcenc_block_sizes.cpp:261-262)
};

/* Derived fields of a DecodeImage (host side, once per launch). */
inline void decode_image_prepare(DecodeImage& img)
{
	const uint32_t T = img.block_x * img.block_y * img.block_z;
	img.t_inv24 = ((1u << 24) + T - 1u) / T;
	img.bx_inv16 = (65536u + img.block_x - 1u) / img.block_x;
	img.bxy_inv16 = (65536u + img.block_x * img.block_y - 1u) / (img.block_x * img.block_y);
	img.ds = img.block_x > 1 ? (1024u + img.block_x / 2u) / (img.block_x - 1u) : 0u;
	img.dt = img.block_y > 1 ? (1024u + img.block_y / 2u) / (img.block_y - 1u) : 0u;
	img.dr = img.block_z > 1 ? (1024u + img.block_z / 2u) / (img.block_z - 1u) : 0u;
}

/* Per-wave scratch (LDS). */
struct DecodeScratch {
	uint8_t weights[2][64];   // unquantized grid weights per plane, 0..64
	uint8_t colors[32];       // unquantized colour values, 0..255
	uint16_t ep[4][8];        // endpoint0.rgba, endpoint1.rgba per partition (16-bit domain)
	uint8_t  lns[4][2];       // rgb / alpha are LNS encoded, per partition
};

/* 128-bit block held as four dwords, bit 0 = LSB of byte 0. */
struct Bits128 { uint32_t w[4]; };

WV_FN uint32_t bits_get(const Bits128& b, int off, int n)
{
	// n <= 16, may run past bit 127 (reads zeros there, like the reference's padded buffer)
	if (n <= 0 || off >= 128) return 0u;
	// The two words the field lies in, picked with bit masks in two halving steps: an indexed read of a register array would
	// put the block in scratch memory, and chains of ?: become chains of branches.
	const uint32_t m64 = 0u - (((uint32_t)off >> 6) & 1u), m32 = 0u - (((uint32_t)off >> 5) & 1u);
	const uint32_t a = (b.w[2] & m64) | (b.w[0] & ~m64), c = (b.w[3] & m64) | (b.w[1] & ~m64), d = b.w[2] & ~m64;    // words 0..2 of the string from bit (off & 64) on
	const uint32_t lo = (c & m32) | (a & ~m32), hi = (d & m32) | (c & ~m32);
	const int sh = off & 31;
	const uint32_t v = sh ? (lo >> sh) | (hi << (32 - sh)) : lo;
	return v & ((1u << n) - 1u);
}

WV_FN uint32_t rev32(uint32_t v)
{
#if WV_DEVICE
	return __builtin_bitreverse32(v);       // v_bfrev_b32
#else
	v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
	v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
	v = ((v >> 4) & 0x0F0F0F0Fu) | ((v & 0x0F0F0F0Fu) << 4);
	v = ((v >> 8) & 0x00FF00FFu) | ((v & 0x00FF00FFu) << 8);
	return (v >> 16) | (v << 16);
#endif
}

/* a * b (+ c) for factors that fit 24 bits (signed / unsigned): one full-rate instruction each on the device, where a
 * 32-bit multiply is a quarter-rate one (and where the compiler, left to itself, forms 64-bit multiply-adds).  The device
 * forms are spelled out because the 24-bit intrinsics are folded back into 32-bit multiplies when an operand's range is unknown. */
WV_FN int mul24(int a, int b)
{
#if WV_DEVICE
	int r; asm("v_mul_i32_i24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r;
#else
	return a * b;
#endif
}
WV_FN uint32_t umul24(uint32_t a, uint32_t b)
{
#if WV_DEVICE
	uint32_t r; asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r;
#else
	return a * b;
#endif
}
WV_FN int mad24(int a, int b, int c)
{
#if WV_DEVICE
	int r; asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r;
#else
	return a * b + c;
#endif
}
WV_FN uint32_t umad24(uint32_t a, uint32_t b, uint32_t c)
{
#if WV_DEVICE
	uint32_t r; asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r;
#else
	return a * b + c;
#endif
}

/* 8 + t0 w0 + t1 w1 + t2 w2 + t3 w3 for factors that fit 24 bits: the weight infill's sum (one block of four multiply-adds
 * on the device: separate ones are scheduled with idle states between them). */
WV_FN int tap_sum4(int t0, int w0, int t1, int w1, int t2, int w2, int t3, int w3)
{
#if WV_DEVICE
	int r;
	asm("v_mad_i32_i24 %0, %1, %2, 8\n\tv_mad_i32_i24 %0, %3, %4, %0\n\tv_mad_i32_i24 %0, %5, %6, %0\n\tv_mad_i32_i24 %0, %7, %8, %0"
	    : "=&v"(r) : "v"(t0), "v"(w0), "v"(t1), "v"(w1), "v"(t2), "v"(w2), "v"(t3), "v"(w3));
	return r;
#else
	return 8 + t0 * w0 + t1 * w1 + t2 * w2 + t3 * w3;
#endif
}

/* a.lo16 * b.lo16 + a.hi16 * b.hi16 + c (v_dot2_u32_u16). */
WV_FN uint32_t udot2(uint32_t a, uint32_t b, uint32_t c)
{
#if WV_DEVICE
	typedef unsigned short wv_u16x2 __attribute__((ext_vector_type(2)));
	return __builtin_amdgcn_udot2(__builtin_bit_cast(wv_u16x2, a), __builtin_bit_cast(wv_u16x2, b), c, false);
#else
	return (a & 0xFFFFu) * (b & 0xFFFFu) + (a >> 16) * (b >> 16) + c;
#endif
}

/* Four consecutive words at a 16-byte aligned address: one 128-bit LDS read on the device. */
struct U32x4 { uint32_t x, y, z, w; };
WV_FN U32x4 load_u32x4_aligned(const uint32_t* p)
{
	U32x4 r;
#if WV_DEVICE
	typedef uint32_t wv_u32x4 __attribute__((ext_vector_type(4)));
	const wv_u32x4 v = *reinterpret_cast<const wv_u32x4*>(p);
	r.x = v.x; r.y = v.y; r.z = v.z; r.w = v.w;
#else
	r.x = p[0]; r.y = p[1]; r.z = p[2]; r.w = p[3];
#endif
	return r;
}

/* Byte permutes (v_perm_b32): byte k of the result is picked by byte k of `sel` out of the eight bytes { hi, lo } -- 0..3 a
 * byte of lo, 4..7 a byte of hi, 0x0C the constant 0x00, 0x0D the constant 0xFF. */
WV_FN uint32_t byte_perm(uint32_t hi, uint32_t lo, uint32_t sel)
{
#if WV_DEVICE
	return __builtin_amdgcn_perm(hi, lo, sel);
#else
	uint32_t r = 0;
	for (int k = 0; k < 4; k++)
	{
		const uint32_t s = (sel >> (8 * k)) & 0xFFu;
		const uint32_t v = s == 0x0Cu ? 0u : s >= 0x0Du ? 255u : s < 4u ? (lo >> (8 * s)) & 0xFFu : (hi >> (8 * (s - 4u))) & 0xFFu;
		r |= v << (8 * k);
	}
	return r;
#endif
}
/* The selector that takes the channels r, g out of the low two bytes of `lo` and b, a out of the low two bytes of `hi`
 * through a swizzle that picks channels or constants (astcenc_swz 0..5). */
WV_FN uint32_t swizzle_selector(const uint32_t swz[4])
{
	uint32_t sel = 0;
	for (int k = 0; k < 4; k++)
	{
		const uint32_t s = swz[k] == 4 ? 0x0Cu : swz[k] == 5 ? 0x0Du : swz[k] < 2 ? swz[k] : swz[k] + 2u;
		sel |= s << (8 * k);
	}
	return sel;
}

/* The weight stream is stored from the top of the block downwards: bit i of it is bit 127 - i. */
WV_FN Bits128 bits_reversed(const Bits128& b)
{
	Bits128 r;
	r.w[0] = rev32(b.w[3]); r.w[1] = rev32(b.w[2]); r.w[2] = rev32(b.w[1]); r.w[3] = rev32(b.w[0]);
	return r;
}

/* Five trits of a packed 8-bit group -> the one at `pos`. */
WV_FN int trit_of(uint32_t t8, int pos)
{
	int t[5];
	uint32_t cbits;
	if (((t8 >> 2) & 7u) == 7u)
	{
		cbits = (((t8 >> 5) & 7u) << 2) | (t8 & 3u);
		t[4] = 2; t[3] = 2;
	}
	else
	{
		cbits = t8 & 0x1Fu;
		if (((t8 >> 5) & 3u) == 3u) { t[4] = 2; t[3] = (int)((t8 >> 7) & 1u); }
		else { t[4] = (int)((t8 >> 7) & 1u); t[3] = (int)((t8 >> 5) & 3u); }
	}
	if ((cbits & 3u) == 3u)
	{
		t[2] = 2; t[1] = (int)((cbits >> 4) & 1u);
		uint32_t c3 = (cbits >> 3) & 1u, c2 = (cbits >> 2) & 1u;
		t[0] = (int)((c3 << 1) | (c2 & ~c3 & 1u));
	}
	else if (((cbits >> 2) & 3u) == 3u)
	{
		t[2] = 2; t[1] = 2; t[0] = (int)(cbits & 3u);
	}
	else
	{
		t[2] = (int)((cbits >> 4) & 1u); t[1] = (int)((cbits >> 2) & 3u);
		uint32_t c1 = (cbits >> 1) & 1u, c0 = cbits & 1u;
		t[0] = (int)((c1 << 1) | (c0 & ~c1 & 1u));
	}
	int r = t[0];
	r = pos == 1 ? t[1] : r; r = pos == 2 ? t[2] : r; r = pos == 3 ? t[3] : r; r = pos == 4 ? t[4] : r;
	return r;
}

/* Three quints of a packed 7-bit group -> the one at `pos`. */
WV_FN int quint_of(uint32_t q7, int pos)
{
	int q[3];
	if (((q7 >> 1) & 3u) == 3u && ((q7 >> 5) & 3u) == 0u)
	{
		uint32_t n0 = ~q7 & 1u;
		q[2] = (int)(((q7 & 1u) << 2) | ((((q7 >> 4) & 1u) & n0) << 1) | (((q7 >> 3) & 1u) & n0));
		q[1] = 4; q[0] = 4;
	}
	else
	{
		uint32_t cbits;
		if (((q7 >> 1) & 3u) == 3u)
		{
			q[2] = 4;
			cbits = (((q7 >> 3) & 3u) << 3) | ((~(q7 >> 5) & 3u) << 1) | (q7 & 1u);
		}
		else
		{
			q[2] = (int)((q7 >> 5) & 3u);
			cbits = q7 & 0x1Fu;
		}
		if ((cbits & 7u) == 5u) { q[1] = 4; q[0] = (int)((cbits >> 3) & 3u); }
		else { q[1] = (int)((cbits >> 3) & 3u); q[0] = (int)(cbits & 7u); }
	}
	int r = q[0];
	r = pos == 1 ? q[1] : r; r = pos == 2 ? q[2] : r;
	return r;
}

/* Symbol `index` of a BISE sequence of `count` symbols at `offset`:  (trit or quint) << bits | low bits.
 * (ref: decode_ise, per element instead of streaming) */
WV_FN int ise_symbol(const Bits128& b, int offset, int quant, int count, int index)
{
	const Btq q = btq_of(quant);
	const int bits = q.bits;
	if (q.trits)
	{
		const int tb[5] = { 2, 2, 1, 2, 1 };
		const int group = index / 5, pos = index - group * 5;
		int at = offset + group * (5 * bits + 8);
		const int in_group = i_min(5, count - group * 5);
		uint32_t t8 = 0, low = 0;
		int shift = 0;
		for (int k = 0; k < 5; k++)
		{
			if (k < in_group)
			{
				uint32_t m = bits_get(b, at, bits);
				at += bits;
				t8 |= bits_get(b, at, tb[k]) << shift;
				at += tb[k];
				low = k == pos ? m : low;
			}
			shift += tb[k];
		}
		return (trit_of(t8, pos) << bits) | (int)low;
	}
	if (q.quints)
	{
		const int qb[3] = { 3, 2, 2 };
		const int group = index / 3, pos = index - group * 3;
		int at = offset + group * (3 * bits + 7);
		const int in_group = i_min(3, count - group * 3);
		uint32_t q7 = 0, low = 0;
		int shift = 0;
		for (int k = 0; k < 3; k++)
		{
			if (k < in_group)
			{
				uint32_t m = bits_get(b, at, bits);
				at += bits;
				q7 |= bits_get(b, at, qb[k]) << shift;
				at += qb[k];
				low = k == pos ? m : low;
			}
			shift += qb[k];
		}
		return (quint_of(q7, pos) << bits) | (int)low;
	}
	return (int)bits_get(b, offset + index * bits, bits);
}

/* BISE weight symbol -> 0..64 (format rule "weight unquantization"). */
WV_FN int unquant_weight_symbol(int v, int quant)
{
	const Btq q = btq_of(quant);
	const int bits = q.bits;
	int r;
	if (!q.trits && !q.quints)
	{
		// replicate the pattern into 6 bits
		r = 0;
		int have = 0;
		for (int k = 0; k < 6 && have < 6; k++) { r = (r << bits) | v; have += bits; }
		r >>= (have - 6);
	}
	else if (bits == 0)
	{
		r = q.trits ? (v == 0 ? 0 : v == 1 ? 32 : 63) : (v == 0 ? 0 : v == 1 ? 16 : v == 2 ? 32 : v == 3 ? 47 : 63);
	}
	else
	{
		const int d = v >> bits;
		const int m = v & ((1 << bits) - 1);
		const int a = m & 1, b = (m >> 1) & 1, cc = (m >> 2) & 1;
		const int A = a ? 0x7F : 0;
		int B, C;
		if (q.trits)
		{
			if (bits == 1) { C = 50; B = 0; }
			else if (bits == 2) { C = 23; B = (b << 6) | (b << 2) | b; }
			else { C = 11; B = (cc << 6) | (b << 5) | (cc << 1) | b; }
		}
		else
		{
			if (bits == 1) { C = 28; B = 0; }
			else { C = 13; B = (b << 6) | (b << 1); }
		}
		int t = d * C + B;
		t ^= A;
		r = (A & 0x20) | (t >> 2);
	}
	return r > 32 ? r + 1 : r;
}

/* BISE colour symbol -> 0..255 (format rule "endpoint unquantization"). */
WV_FN int unquant_color_symbol(int v, int quant)
{
	const Btq q = btq_of(quant);
	const int bits = q.bits;
	if (!q.trits && !q.quints)
	{
		int r = 0, have = 0;
		for (int k = 0; k < 8 && have < 8; k++) { r = (r << bits) | v; have += bits; }
		return r >> (have - 8);
	}
	const int dd = v >> bits;
	const int m = v & ((1 << bits) - 1);
	const int a = m & 1, b = (m >> 1) & 1, c = (m >> 2) & 1, d = (m >> 3) & 1, e = (m >> 4) & 1, f = (m >> 5) & 1;
	const int A = a ? 0x1FF : 0;
	int B = 0, C = 0;
	if (q.trits)
	{
		if (bits == 1) { C = 204; }
		else if (bits == 2) { C = 93; B = (b << 8) | (b << 4) | (b << 2) | (b << 1); }
		else if (bits == 3) { C = 44; B = (c << 8) | (b << 7) | (c << 3) | (b << 2) | (c << 1) | b; }
		else if (bits == 4) { C = 22; B = (d << 8) | (c << 7) | (b << 6) | (d << 2) | (c << 1) | b; }
		else if (bits == 5) { C = 11; B = (e << 8) | (d << 7) | (c << 6) | (b << 5) | (e << 1) | d; }
		else { C = 5; B = (f << 8) | (e << 7) | (d << 6) | (c << 5) | (b << 4) | f; }
	}
	else
	{
		if (bits == 1) { C = 113; }
		else if (bits == 2) { C = 54; B = (b << 8) | (b << 3) | (b << 2); }
		else if (bits == 3) { C = 26; B = (c << 8) | (b << 7) | (c << 2) | (b << 1) | c; }
		else if (bits == 4) { C = 13; B = (d << 8) | (c << 7) | (b << 6) | (d << 1) | c; }
		else { C = 6; B = (e << 8) | (d << 7) | (c << 6) | (b << 5) | e; }
	}
	int t = dd * C + B;
	t ^= A;
	return (A & 0x80) | (t >> 2);
}

#if !defined(ASTC_DECODE_NO_LUTS)
// ---------------------------------------------------------------------------------------------
// Table-driven symbol decode of the batched decoder.  The four tables are generated from the arithmetic
// routines above (tools/gen_decode_luts.cpp -> decode_luts.inc); they hold format constants, nothing that
// depends on a block size or preset.
// ---------------------------------------------------------------------------------------------
#include "decode_luts.inc"

WV_FN uint32_t trit_group_lut(uint32_t t8) { const uint16_t t[256] = { ASTC_TRIT_LUT_VALUES }; return table_at(t, t8); }
WV_FN uint32_t quint_group_lut(uint32_t q7) { const uint16_t t[128] = { ASTC_QUINT_LUT_VALUES }; return table_at(t, q7); }
WV_FN int weight_unquant_lut(int quant, int sym) { const uint8_t t[12 * 32] = { ASTC_WEIGHT_UNQUANT_LUT_VALUES }; return t[quant * 32 + sym]; }
/* both group tables in one: trit groups at 0..255, quint groups at 256..383 (entry 0 -- all trits zero -- also serves the
 * levels without trits or quints) */
WV_FN uint32_t group_lut(uint32_t index) { const uint16_t t[256 + 128] = { ASTC_TRIT_LUT_VALUES ASTC_QUINT_LUT_VALUES }; return table_at(t, index); }
/* four consecutive entries of the weight table as one word (entry 4 i in the low byte) */
WV_FN uint32_t weight_unquant_lut_word(int i)
{
	alignas(4) const uint8_t t[12 * 32] = { ASTC_WEIGHT_UNQUANT_LUT_VALUES };
	uint32_t v;
	__builtin_memcpy(&v, t + 4 * i, 4);
	return v;
}
WV_FN int color_unquant_lut(int quant, int sym) { const uint8_t t[21 * 256] = { ASTC_COLOR_UNQUANT_LUT_VALUES }; return table_at(t, (uint32_t)(quant * 256 + sym)); }

/* (hi:lo) >> sh, low 32 bits; sh in 0..31. */
WV_FN uint32_t funnel_shift_right(uint32_t hi, uint32_t lo, int sh)
{
#if WV_DEVICE
	return __builtin_amdgcn_alignbit(hi, lo, (uint32_t)sh);
#else
	return sh ? (lo >> sh) | (hi << (32 - sh)) : lo;
#endif
}

/* Symbols per lane-trip in the batched decoder: one BISE group (five trits or three quints share packed bits, so the
 * group is the natural unit: its window and its table entry are fetched once), four symbols for plain bit fields. */
WV_FN int ise_group_size(int kind) { return kind == 1 ? 5 : kind == 2 ? 3 : 4; }
/* btq_of without table reads: symbol bits | kind << 4 (kind 0 plain bits, 1 trits, 2 quints) of quant level q, out of
 * packed constants (spec table C.2.7: four bits per level for the bit counts, two for the kinds). */
WV_FN uint32_t btq_packed(int q)
{
	const uint64_t kinds = 0x6186186184ull, bits_lo = 0x4643532421310201ull;
	const uint32_t bits_hi = 0x86575u;
	const uint32_t bits = q < 16 ? (uint32_t)(bits_lo >> (4 * q)) & 15u : (bits_hi >> (4 * (q - 16))) & 15u;
	return bits | (((uint32_t)(kinds >> (2 * q)) & 3u) << 4);
}
/* groups of a sequence of `count` symbols (count < 128), without a division */
WV_FN int ise_group_count(int count, int kind) { return kind == 1 ? ((count + 4) * 205) >> 10 : kind == 2 ? ((count + 2) * 171) >> 9 : (count + 3) >> 2; }

#endif // !ASTC_DECODE_NO_LUTS

/* Block mode field -> grid size, planes, weight quant.  False for reserved / oversized modes.
 * (ref: decode_block_mode_2d / _3d, astcenc_block_sizes.cpp:37-243 + the checks in construct_block_size_descriptor_2d / _3d) */
WV_FN bool decode_block_mode(uint32_t mode, int block_x, int block_y, int block_z, int& wx, int& wy, int& wz, bool& dual, int& wquant)
{
	int r, h, d, w = 0, ht = 0, dp = 1;
	if (block_z > 1)
	{
		// 3D footprints (spec table C.2.10)
		const int a = (int)((mode >> 5) & 3u);
		d = (int)((mode >> 10) & 1u); h = (int)((mode >> 9) & 1u);
		if (mode & 3u)
		{
			r = (int)(((mode >> 4) & 1u) | ((mode & 3u) << 1));
			w = a + 2; ht = (int)((mode >> 7) & 3u) + 2; dp = (int)((mode >> 2) & 3u) + 2;
		}
		else
		{
			if ((mode & 0xFu) == 0u) return false;
			r = (int)(((mode >> 4) & 1u) | (((mode >> 2) & 3u) << 1));
			const int b = (int)((mode >> 9) & 3u);
			const int sel = (int)((mode >> 7) & 3u);
			if (sel != 3) { d = 0; h = 0; }
			if (sel == 0) { w = 6; ht = b + 2; dp = a + 2; }
			else if (sel == 1) { w = a + 2; ht = 6; dp = b + 2; }
			else if (sel == 2) { w = a + 2; ht = b + 2; dp = 6; }
			else
			{
				w = 2; ht = 2; dp = 2;
				if (a == 0) w = 6; else if (a == 1) ht = 6; else if (a == 2) dp = 6; else return false;
			}
		}
	}
	else if (mode & 3u)
	{
		r = (int)(((mode >> 4) & 1u) | ((mode & 3u) << 1));
		const int a = (int)((mode >> 5) & 3u);
		int b = (int)((mode >> 7) & 3u);
		const int sel = (int)((mode >> 2) & 3u);
		if (sel == 0) { w = b + 4; ht = a + 2; }
		else if (sel == 1) { w = b + 8; ht = a + 2; }
		else if (sel == 2) { w = a + 2; ht = b + 8; }
		else
		{
			b &= 1;
			if (mode & 0x100u) { w = b + 2; ht = a + 2; }
			else { w = a + 2; ht = b + 6; }
		}
		d = (int)((mode >> 10) & 1u); h = (int)((mode >> 9) & 1u);
	}
	else
	{
		if ((mode & 0xFu) == 0u) return false;
		r = (int)(((mode >> 4) & 1u) | (((mode >> 2) & 3u) << 1));
		const int a = (int)((mode >> 5) & 3u);
		const int b = (int)((mode >> 9) & 3u);
		d = (int)((mode >> 10) & 1u); h = (int)((mode >> 9) & 1u);
		const int sel = (int)((mode >> 7) & 3u);
		if (sel == 0) { w = 12; ht = a + 2; }
		else if (sel == 1) { w = a + 2; ht = 12; }
		else if (sel == 2) { w = a + 6; ht = b + 6; d = 0; h = 0; }
		else
		{
			if (a == 0) { w = 6; ht = 10; }
			else if (a == 1) { w = 10; ht = 6; }
			else return false;
		}
	}
	if (r < 2) return false;
	wquant = (r - 2) + 6 * h;
	wx = w; wy = ht; wz = dp; dual = d != 0;
	if (w > block_x || ht > block_y || dp > block_z) return false;
	const int count = w * ht * dp * (d ? 2 : 1);
	if (count > 64) return false;
	const int wbits = (int)ise_bitcount((unsigned)count, wquant);
	return wbits >= 24 && wbits <= 96;
}

/* Per-footprint tables of the batched decoder: what an 11-bit block mode field says for this footprint (the arithmetic of
 * decode_block_mode and its BISE size, one word per mode) and the colour quant level a bit budget affords (the search of
 * physical_to_symbolic, astcenc_symbolic_physical.cpp:467-486, one byte per (value pairs, bits)).  Built on the host
 * by decode_tables_build() from those routines when a context is created and kept in HBM beside its other tables. */
struct DecodeTables {
	uint32_t mode[2048];       // bit 31 legal | weight count over both planes << 24 | weight stream bits << 17 | weight quant << 13 | dual << 12 | grid z << 8 | y << 4 | x
	uint8_t  cquant[9][128];   // [colour value pairs - 1][bits] -> highest quant level whose BISE size fits + 1, 0 = none
};

inline void decode_tables_build(DecodeTables& t, int block_x, int block_y, int block_z)
{
	for (uint32_t mode = 0; mode < 2048u; mode++)
	{
		int wx = 0, wy = 0, wz = 1, wquant = 0;
		bool dual = false;
		uint32_t e = 0u;
		if ((mode & 0x1FFu) != 0x1FCu && decode_block_mode(mode, block_x, block_y, block_z, wx, wy, wz, dual, wquant))
		{
			const int count = wx * wy * wz * (dual ? 2 : 1);
			e = 0x80000000u | ((uint32_t)count << 24) | ((uint32_t)ise_bitcount((unsigned)count, wquant) << 17) | ((uint32_t)wquant << 13) |
			    ((dual ? 1u : 0u) << 12) | ((uint32_t)wz << 8) | ((uint32_t)wy << 4) | (uint32_t)wx;
		}
		t.mode[mode] = e;
	}
	for (int pairs = 1; pairs <= 9; pairs++)
	{
		for (int bits = 0; bits < 128; bits++)
		{
			int best = -1;
			for (int q = 20; q >= 0; q--)
			{
				if (best < 0 && (int)ise_bitcount((unsigned)(2 * pairs), q) <= bits) best = q;
			}
			t.cquant[pairs - 1][bits] = (uint8_t)(best + 1);
		}
	}
}

/* The partition hash (ref: select_partition / hash52, astcenc_partition_tables.cpp:66-245) in two steps: everything
 * that depends on the block only -- the scrambled seed, its twelve 4-bit multipliers after squaring and shifting, the
 * four offsets -- is packed into four words (x | y << 8 | z << 16 multipliers and the offset << 24 of the a, b, c, d
 * terms; the terms a partition count does not use are zero), then a texel needs three multiply-adds per term. */
struct PartitionHash { uint32_t term[4]; };

WV_FN PartitionHash partition_hash_setup(int seed, int partition_count)
{
	seed += (partition_count - 1) * 1024;
	uint32_t rnum = (uint32_t)seed;
	rnum ^= rnum >> 15; rnum -= rnum << 17; rnum += rnum << 7; rnum += rnum << 4;
	rnum ^= rnum >> 5; rnum += rnum << 16; rnum ^= rnum >> 7; rnum ^= rnum >> 3;
	rnum ^= rnum << 6; rnum ^= rnum >> 17;

	uint32_t s1 = rnum & 0xF, s2 = (rnum >> 4) & 0xF, s3 = (rnum >> 8) & 0xF, s4 = (rnum >> 12) & 0xF;
	uint32_t s5 = (rnum >> 16) & 0xF, s6 = (rnum >> 20) & 0xF, s7 = (rnum >> 24) & 0xF, s8 = (rnum >> 28) & 0xF;
	uint32_t s9 = (rnum >> 18) & 0xF, s10 = (rnum >> 22) & 0xF, s11 = (rnum >> 26) & 0xF, s12 = ((rnum >> 30) | (rnum << 2)) & 0xF;
	s1 *= s1; s2 *= s2; s3 *= s3; s4 *= s4; s5 *= s5; s6 *= s6; s7 *= s7; s8 *= s8;
	s9 *= s9; s10 *= s10; s11 *= s11; s12 *= s12;

	int sh1, sh2;
	if (seed & 1) { sh1 = (seed & 2) ? 4 : 5; sh2 = partition_count == 3 ? 6 : 5; }
	else { sh1 = partition_count == 3 ? 6 : 5; sh2 = (seed & 2) ? 4 : 5; }
	const int sh3 = (seed & 0x10) ? sh1 : sh2;
	s1 >>= sh1; s2 >>= sh2; s3 >>= sh1; s4 >>= sh2; s5 >>= sh1; s6 >>= sh2; s7 >>= sh1; s8 >>= sh2;
	s9 >>= sh3; s10 >>= sh3; s11 >>= sh3; s12 >>= sh3;

	// only the low six bits of a term matter, so the offsets are kept modulo 64
	PartitionHash h;
	h.term[0] = s1 | (s2 << 8) | (s11 << 16) | (((rnum >> 14) & 0x3Fu) << 24);
	h.term[1] = s3 | (s4 << 8) | (s12 << 16) | (((rnum >> 10) & 0x3Fu) << 24);
	h.term[2] = partition_count <= 2 ? 0u : s5 | (s6 << 8) | (s9 << 16) | (((rnum >> 6) & 0x3Fu) << 24);
	h.term[3] = partition_count <= 3 ? 0u : s7 | (s8 << 8) | (s10 << 16) | (((rnum >> 2) & 0x3Fu) << 24);
	return h;
}

WV_FN int partition_from_hash(const PartitionHash& h, int x, int y, int z, bool small_block)
{
	if (small_block) { x <<= 1; y <<= 1; z <<= 1; }
	int v[4];
	for (int k = 0; k < 4; k++)
	{
		const uint32_t t = h.term[k];
		v[k] = (int)(((t & 0xFFu) * (uint32_t)x + ((t >> 8) & 0xFFu) * (uint32_t)y + ((t >> 16) & 0xFFu) * (uint32_t)z + (t >> 24)) & 0x3Fu);
	}
	const int a = v[0], b = v[1], c = v[2], d = v[3];
	if (a >= b && a >= c && a >= d) return 0;
	if (b >= c && b >= d) return 1;
	if (c >= d) return 2;
	return 3;
}

WV_FN int partition_of_texel(int seed, int x, int y, int z, int partition_count, bool small_block)
{
	return partition_from_hash(partition_hash_setup(seed, partition_count), x, y, z, small_block);
}

/* (ref: unorm16_to_sf16, astcenc_vecmathlib.h:503) */
WV_FN int unorm16_to_sf16(int p)
{
	if (p == 0xFFFF) return 0x3C00;
	if (p < 4) return p << 8;
	const int lz = __builtin_clz((unsigned)p) - 16;   // leading zeros within 16 bits (4 <= p < 0xFFFF here)
	int v = (p << (lz + 1)) & 0xFFFF;
	v >>= 6;
	return v | ((14 - lz) << 10);
}

/* The swizzle's sources of one decoded texel: r, g, b, a, 0, 1 and the reconstructed normal z. (ref: store_image_block :345-573) */
WV_FN void swizzle_sources(float r, float g, float b, float a, float src[7])
{
	src[0] = r; src[1] = g; src[2] = b; src[3] = a; src[4] = 0.0f; src[5] = 1.0f;
	float xn = (r * 2.0f) - 1.0f;
	float yn = (a * 2.0f) - 1.0f;
	float zn = 1.0f - xn * xn - yn * yn;
	if (zn < 0.0f) zn = 0.0f;
	src[6] = (f_sqrt(zn) * 0.5f) + 0.5f;
}

/* One decoded texel (floats) as an RGBA8 pixel through the swizzle; an error texel (NaN) is opaque magenta. */
WV_FN uint32_t pack_texel_u8(const DecodeImage& img, float r, float g, float b, float a)
{
	if (r != r) return 0xFFFF00FFu;
	float src[7];
	swizzle_sources(r, g, b, a, src);
	uint32_t px = 0;
	for (int k = 0; k < 4; k++)
	{
		uint32_t sw = img.swz[k];
		int v;
		if (sw == 4) v = 0;
		else if (sw == 5) v = 255;
		else
		{
			float f = src[sw];
			if (sw == 6) f = f < 1.0f ? f : 1.0f;                    // min(z, 1), z is never negative
			else f = v_clampzo(f);
			v = (int)(f * 255.0f + 0.5f);
		}
		px |= ((uint32_t)v & 0xFFu) << (8 * k);
	}
	return px;
}

/* Write one decoded texel (floats) through the swizzle; `at` = index of its first component.  (ref: store_image_block :345-573) */
WV_FN void store_texel_at(const DecodeImage& img, size_t at, float r, float g, float b, float a)
{
	if (img.data_type == 0)
	{
		// one 32-bit store per texel (the image base is at least 4-byte aligned, rows are tightly packed)
		const uint32_t px = pack_texel_u8(img, r, g, b, a);
		__builtin_memcpy(static_cast<uint8_t*>(img.data) + at, &px, 4);
		return;
	}
	float src[7];
	swizzle_sources(r, g, b, a, src);
	if (img.data_type == 1)
	{
		uint16_t h4[4];
		for (int k = 0; k < 4; k++) h4[k] = float_to_half(src[img.swz[k]]);
		__builtin_memcpy(static_cast<uint16_t*>(img.data) + at, h4, 8);
	}
	else
	{
		float* o = static_cast<float*>(img.data) + at;
		for (int k = 0; k < 4; k++) o[k] = src[img.swz[k]];
	}
}

WV_FN void store_texel(const DecodeImage& img, uint32_t x, uint32_t y, uint32_t z, float r, float g, float b, float a)
{
	store_texel_at(img, (((size_t)z * img.dim_y + y) * img.dim_x + x) * 4, r, g, b, a);
}

/* Everything the header of a block says (ref: physical_to_symbolic :291-556 up to the ISE decode). */
struct BlockHeader {
	bool error, constant, constant_f16;
	int  const_color[4];      // void-extent colour, raw 16-bit fields
	int  wx, wy, wz, wquant;  // weight grid and its quant level
	int  wbits;               // length of the weight stream
	bool dual;
	int  parts, seed, plane2;
	int  fmt[4];              // colour endpoint mode per partition
	int  nvals, cquant, color_start;
};

/* tabs: the footprint's DecodeTables (the batched decoder), or null: the mode and the colour quant level by arithmetic. */
WV_FN BlockHeader parse_block_header(const Bits128& blk, int block_x, int block_y, int block_z, const DecodeTables* tabs = nullptr)
{
	BlockHeader h;
	h.error = false; h.constant = false; h.constant_f16 = false;
	h.wx = 0; h.wy = 0; h.wz = 1; h.wquant = 0; h.wbits = 0; h.dual = false;
	h.parts = 1; h.seed = 0; h.plane2 = -1;
	h.nvals = 0; h.cquant = 0; h.color_start = 17;
	for (int k = 0; k < 4; k++) { h.const_color[k] = 0; h.fmt[k] = 0; }

	const uint32_t mode = bits_get(blk, 0, 11);
	if ((mode & 0x1FFu) == 0x1FCu)
	{
		// void extent (ref: :302-370): reserved bits set, coordinates all ones or properly ordered
		h.constant = true;
		h.constant_f16 = (mode & 0x200u) != 0;
		if (block_z > 1)
		{
			// 3D layout: six 9-bit coordinates from bit 10, no reserved bits
			const uint32_t ls = bits_get(blk, 10, 9), hs = bits_get(blk, 19, 9), lt = bits_get(blk, 28, 9), ht = bits_get(blk, 37, 9);
			const uint32_t lr = bits_get(blk, 46, 9), hr = bits_get(blk, 55, 9);
			const bool all_ones = ls == 0x1FFu && hs == 0x1FFu && lt == 0x1FFu && ht == 0x1FFu && lr == 0x1FFu && hr == 0x1FFu;
			if ((ls >= hs || lt >= ht || lr >= hr) && !all_ones) h.error = true;
		}
		else
		{
			const uint32_t ls = bits_get(blk, 12, 13), hs = bits_get(blk, 25, 13), lt = bits_get(blk, 38, 13), ht = bits_get(blk, 51, 13);
			const bool all_ones = ls == 0x1FFFu && hs == 0x1FFFu && lt == 0x1FFFu && ht == 0x1FFFu;
			if (bits_get(blk, 10, 2) != 3u || ((ls >= hs || lt >= ht) && !all_ones)) h.error = true;
		}
		for (int k = 0; k < 4; k++) h.const_color[k] = (int)bits_get(blk, 64 + 16 * k, 16);
		return h;
	}
	int wbits;
	if (tabs)
	{
		const uint32_t e = table_at(tabs->mode, mode);
		if (!(e >> 31))
		{
			h.error = true;
			return h;
		}
		h.wx = (int)(e & 15u); h.wy = (int)((e >> 4) & 15u); h.wz = (int)((e >> 8) & 15u);
		h.dual = ((e >> 12) & 1u) != 0u;
		h.wquant = (int)((e >> 13) & 15u);
		wbits = (int)((e >> 17) & 127u);
	}
	else
	{
		if (!decode_block_mode(mode, block_x, block_y, block_z, h.wx, h.wy, h.wz, h.dual, h.wquant))
		{
			h.error = true;
			return h;
		}
		const int wcount = h.wx * h.wy * h.wz;
		wbits = (int)ise_bitcount((unsigned)(h.dual ? 2 * wcount : wcount), h.wquant);
	}
	h.wbits = wbits;
	h.parts = (int)bits_get(blk, 11, 2) + 1;
	if (h.dual && h.parts == 4) h.error = true;

	int below = 128 - wbits;
	if (h.parts == 1)
	{
		h.fmt[0] = (int)bits_get(blk, 13, 4);
	}
	else
	{
		h.seed = (int)bits_get(blk, 13, 10);
		h.color_start = 29;
		const uint32_t cem = bits_get(blk, 23, 6);
		if ((cem & 3u) == 0u)
		{
			for (int i = 0; i < 4; i++) h.fmt[i] = (int)((cem >> 2) & 0xFu);
		}
		else
		{
			const int extra = 3 * h.parts - 4;
			below -= extra;
			const uint32_t enc = cem | (bits_get(blk, below, extra) << 6);
			const int base = (int)(enc & 3u) - 1;
			for (int i = 0; i < 4; i++)
			{
				const int cls = base + (int)((enc >> (2 + i)) & 1u);
				const int low = (int)((enc >> (2 + h.parts + 2 * i)) & 3u);
				h.fmt[i] = i < h.parts ? cls * 4 + low : 0;
			}
		}
	}
	if (h.dual)
	{
		below -= 2;
		h.plane2 = (int)bits_get(blk, below, 2);
	}

	for (int i = 0; i < 4; i++) h.nvals += i < h.parts ? 2 * (h.fmt[i] >> 2) + 2 : 0;
	if (h.nvals > 18) h.error = true;

	// the colour stream uses the highest quant level whose BISE size fits the bits left
	int cbits = below - h.color_start;
	if (cbits < 0) cbits = 0;
	if (tabs)
	{
		// (an over-long value count is an error already; the table has rows for the legal counts)
		h.cquant = h.nvals <= 18 ? (int)table_at(&tabs->cquant[0][0], (uint32_t)((h.nvals >> 1) - 1) * 128u + (uint32_t)cbits) - 1 : -1;
	}
	else
	{
		h.cquant = -1;
		for (int q = 20; q >= 0; q--)
		{
			if (h.cquant < 0 && (int)ise_bitcount((unsigned)h.nvals, q) <= cbits) h.cquant = q;
		}
	}
	if (h.cquant < QUANT_6) h.error = true;
	return h;
}

/* Which endpoint lanes hold LNS codes (ref: color_unquantize.cpp:854-1022). */
WV_FN void endpoint_lns_flags(int profile, int f, bool& rgb_lns, bool& alpha_lns)
{
	const bool hdr_fmt = f == 2 || f == 3 || f == 7 || f == 11 || f == 14 || f == 15;
	const bool hdr_profile = profile == 2 || profile == 3;
	const bool alpha_default = f == 2 || f == 3 || f == 7 || f == 11;
	rgb_lns = hdr_fmt && hdr_profile;
	alpha_lns = hdr_profile && (f == 15 || (alpha_default && profile == 3));
}

/* Weights of one texel from the grid (format rule "weight infill"; ref: unpack_weights :89). */
/* g0 / g1: grid weight 0 of plane 0 / 1, `stride` bytes from one grid position to the next. */
WV_FN void infill_texel_weights(int wx, int wy, int wz, bool dual, const uint8_t* g0, const uint8_t* g1, int stride, int ds, int dt, int dr, int block_z, int tx, int ty, int tz, int wp[2])
{
	// ds, dt, dr = (1024 + block / 2) / (block - 1) per axis
	if (block_z > 1)
	{
		// 3D: simplex interpolation inside the grid cell -- from the low corner step along the axes in
		// descending order of fraction (ref: init_decimation_info_3d, astcenc_block_sizes.cpp:483-600)
		const int gs = (ds * tx * (wx - 1) + 32) >> 6;
		const int gt = (dt * ty * (wy - 1) + 32) >> 6;
		const int gr = (dr * tz * (wz - 1) + 32) >> 6;
		const int fs = gs & 0xF, ft = gt & 0xF, fp = gr & 0xF;
		const int N = wx, NM = wx * wy;
		const int cas = ((fs > ft) << 2) + ((ft > fp) << 1) + (fs > fp);
		int s1, s2, w0, w1, w2, w3;
		switch (cas)
		{
		case 7: s1 = 1;  s2 = N;  w0 = 16 - fs; w1 = fs - ft; w2 = ft - fp; w3 = fp; break;
		case 3: s1 = N;  s2 = 1;  w0 = 16 - ft; w1 = ft - fs; w2 = fs - fp; w3 = fp; break;
		case 5: s1 = 1;  s2 = NM; w0 = 16 - fs; w1 = fs - fp; w2 = fp - ft; w3 = ft; break;
		case 4: s1 = NM; s2 = 1;  w0 = 16 - fp; w1 = fp - fs; w2 = fs - ft; w3 = ft; break;
		case 2: s1 = N;  s2 = NM; w0 = 16 - ft; w1 = ft - fp; w2 = fp - fs; w3 = fs; break;
		default: s1 = NM; s2 = N; w0 = 16 - fp; w1 = fp - ft; w2 = ft - fs; w3 = fs; break;
		}
		const int v0 = ((gr >> 4) * wy + (gt >> 4)) * wx + (gs >> 4);
		const int v1 = v0 + s1, v2 = v1 + s2, v3 = v0 + NM + N + 1;
		for (int pl = 0; pl < 2; pl++)
		{
			const uint8_t* g = pl ? g1 : g0;
			int sum = 8;
			sum += w0 ? g[v0 * stride] * w0 : 0;
			sum += w1 ? g[v1 * stride] * w1 : 0;
			sum += w2 ? g[v2 * stride] * w2 : 0;
			sum += w3 ? g[v3 * stride] * w3 : 0;
			wp[pl] = sum >> 4;
		}
		return;
	}
	wp[1] = 0;
	const int cs = ds * tx, ct = dt * ty;
	const int gs = (cs * (wx - 1) + 32) >> 6;
	const int gt = (ct * (wy - 1) + 32) >> 6;
	const int js = gs >> 4, fs = gs & 0xF, jt = gt >> 4, ft = gt & 0xF;
	const int w11 = (fs * ft + 8) >> 4;
	const int w10 = ft - w11, w01 = fs - w11, w00 = 16 - fs - ft + w11;
	const int v0 = js + jt * wx;
	const int wcount = wx * wy;
	// A tap whose factor is zero may lie past the grid (the last row / column interpolates with factor 0): its product is
	// zero whatever is read there, and v0 + wx + 1 <= 76 stays inside the scratch record the weights are the first 128
	// bytes of -- so the four taps are read unconditionally instead of behind eight tests.
	(void)wcount;
	for (int pl = 0; pl < (dual ? 2 : 1); pl++)
	{
		const uint8_t* g = pl ? g1 : g0;
		const int sum = 8 + g[v0 * stride] * w00 + g[(v0 + 1) * stride] * w01 + g[(v0 + wx) * stride] * w10 + g[(v0 + wx + 1) * stride] * w11;
		wp[pl] = sum >> 4;
	}
}

WV_FN void infill_texel_weights(const BlockHeader& h, const uint8_t gw[2][64], int ds, int dt, int dr, int block_z, int tx, int ty, int tz, int wp[2])
{
	infill_texel_weights(h.wx, h.wy, h.wz, h.dual, gw[0], gw[1], 1, ds, dt, dr, block_z, tx, ty, tz, wp);
}

/* Unpack weights, colour values and endpoints of a non-constant, legal block into the scratch. */
WV_FN void unpack_block_payload(const Bits128& blk, const BlockHeader& h, int profile, DecodeScratch& s)
{
	const int wcount = h.wx * h.wy * h.wz;
	const int real_wcount = h.dual ? 2 * wcount : wcount;
	const Bits128 rev = bits_reversed(blk);
	WV_FOR(i, real_wcount)
	{
		int sym = ise_symbol(rev, 0, h.wquant, real_wcount, i);
		int w = unquant_weight_symbol(sym, h.wquant);
		if (h.dual) s.weights[i & 1][i >> 1] = (uint8_t)w;
		else s.weights[0][i] = (uint8_t)w;
	}
	WV_FOR(i, h.nvals)
	{
		int sym = ise_symbol(blk, h.color_start, h.cquant, h.nvals, i);
		s.colors[i] = (uint8_t)unquant_color_symbol(sym, h.cquant);
	}
	WV_SYNC();
	WV_FOR(p, h.parts)
	{
		int first = 0;
		for (int i = 0; i < 4; i++) first += i < p ? 2 * (h.fmt[i] >> 2) + 2 : 0;
		const int f = p == 0 ? h.fmt[0] : p == 1 ? h.fmt[1] : p == 2 ? h.fmt[2] : h.fmt[3];
		uint8_t in[8];
		const int n = 2 * (f >> 2) + 2;
		for (int j = 0; j < 8; j++) in[j] = j < n ? s.colors[first + j] : 0;
		i4 e0, e1;
		unpack_color_endpoints(profile, f, in, e0, e1);
		uint16_t* o = s.ep[p];
		o[0] = (uint16_t)e0.x; o[1] = (uint16_t)e0.y; o[2] = (uint16_t)e0.z; o[3] = (uint16_t)e0.w;
		o[4] = (uint16_t)e1.x; o[5] = (uint16_t)e1.y; o[6] = (uint16_t)e1.z; o[7] = (uint16_t)e1.w;
		bool rgb_lns, alpha_lns;
		endpoint_lns_flags(profile, f, rgb_lns, alpha_lns);
		s.lns[p][0] = rgb_lns ? 1 : 0;
		s.lns[p][1] = alpha_lns ? 1 : 0;
	}
	WV_SYNC();
}

#if !defined(ASTC_DECODE_NO_LUTS)
/* A run of consecutive blocks of one block row decoded together by one wavefront.  Decoding one block keeps few lanes
 * busy (a header, <= 64 weights, <= 18 colour values, <= 4 endpoint pairs, T texels, one after the other); over a run
 * every phase has a lane per (block, element), and the per-block phases cost a pass per run instead of one per block.
 * The blocks of a run share their rows of the image, so the texel phase walks the image row by row: a lane keeps its
 * column -- block, texel x, everything the weight infill and the partition hash derive from them -- while the rows go by,
 * and consecutive lanes store consecutive pixels. */
#ifndef ASTC_DECODE_BATCH
#define ASTC_DECODE_BATCH 32
#endif
// (one run per wavefront.  Measured in round 5, profiles/r05zz/decode_runs_per_wave.log, 8192^2 6x6: 1 run 0.258 ms, 2 runs
//  0.276, 4 runs 0.285, 8 runs 0.343 -- launching a wave and filling its table is 0.016 ms of the 0.258, and longer-lived
//  waves cost registers and a longer tail.  The phase-by-phase timings of profiles/r05zz came from measurement switches
//  that are no longer in this source.)
constexpr int DECODE_BATCH = ASTC_DECODE_BATCH;
static_assert(DECODE_BATCH == 32 || DECODE_BATCH == 16, "the lane maps of decode_row_batch pair lane l with block l & (DECODE_BATCH - 1)");
constexpr int DECODE_SLOTS = 64 / DECODE_BATCH;             // lanes per block in the element phases
constexpr int DECODE_BATCH_LOG2 = DECODE_BATCH == 32 ? 5 : 4;

/* Per-wave scratch (LDS).  Strides are chosen so that lanes on neighbouring blocks fall on different banks. */
struct alignas(16) DecodeBatch {
	uint32_t rec[DECODE_BATCH][4];       // what the element and texel phases need to know about a block (rec_* below), one 16-byte read
	uint32_t hash[DECODE_BATCH][4];      // multi-partition blocks: the per-block part of the partition hash (PartitionHash)
	uint8_t  lns[DECODE_BATCH][4];       // per partition: 1 = rgb, 2 = alpha endpoints are LNS codes
	uint8_t  fmt[DECODE_BATCH][4];       // colour endpoint mode per partition
	uint8_t  weights[DECODE_BATCH][76];  // unquantized grid weights in stream order (the two planes of a dual-plane block interleaved), 0..64
	uint8_t  colors[DECODE_BATCH][28];   // unquantized colour values, 0..255
	// Endpoints per partition, one word per channel: endpoint0 | endpoint1 << 16 (16-bit values).  With the weight pair
	// (256 - 4 w) | 4 w << 16 one dot product per channel, plus 128, is the interpolated 16-bit value in bits 8..23 (its top
	// byte in byte 2).  Until the endpoint phase writes them the words hold the block's bit streams, both cut off at their
	// lengths: five words -- the block up to the end of its colour values and a zero word -- and four -- the weight stream,
	// i.e. the block bit-reversed -- at decode_batch_bits() / decode_batch_wstream().
	// A constant-colour block keeps its four floats in [0..3].
	uint32_t ep[DECODE_BATCH][4 * 4];
	uint8_t  wunq[12 * 32];              // weight_unquant_lut, copied once per wave
};

// rec[0]: bit 0 no payload (error or constant colour), bit 1 error, bit 2 constant colour, bit 3 fast (RGBA8 pixel from integers),
//         4..6 weight symbol bits, 8..9 weight kind, 12..15 weight quant, 16..21 weight groups, 24..28 bits per weight group
// rec[1]: 0..3 colour symbol bits, 4..5 colour kind, 8..12 colour quant, 16..20 colour values, 24..28 first colour bit
// rec[2]: 0..3 grid x, 4..7 y, 8..11 z, 12 dual plane, 13..15 partitions, 16..17 second plane's component, 20..22 colour groups
// rec[3]: error / constant block with RGBA8 output: the pixel

// (the 128-bit reads of load_u32x4_aligned)
static_assert(__builtin_offsetof(DecodeBatch, rec) % 16 == 0 && __builtin_offsetof(DecodeBatch, hash) % 16 == 0 && __builtin_offsetof(DecodeBatch, ep) % 16 == 0,
              "DecodeBatch: records read as four words are 16-byte aligned");

// The streams' place inside the block's sixteen words turns with the block index: the element phases have their lanes on
// 32 different blocks, whose records are 16 words -- half the banks -- apart; unturned, every stream read is a 16-way bank
// conflict (measured: the weight and colour phases did not get faster when their instructions were halved).
WV_FN uint32_t* decode_batch_bits(DecodeBatch& s, int k) { return s.ep[k] + ((k >> 1) & 7); }
WV_FN uint32_t* decode_batch_wstream(DecodeBatch& s, int k) { return s.ep[k] + ((k >> 1) & 7) + 5; }

/* 32 bits of a bit string from bit `at` on (the word after the last one read must exist). */
WV_FN uint32_t bits_window32(const uint32_t* w, int at)
{
	const int word = at >> 5, sh = at & 31;
	return funnel_shift_right(w[word + 1], w[word], sh);
}

/* What the decode of a BISE group needs to know about a quant level, as bit positions: the five symbols' low bits start at
 * sym_at[e], the bits of the packed trit / quint word are `width[e]` bits at word_at[e] that go to bit word_to[e] of the
 * table index, and a symbol's trit / quint is `digit_bits` bits of the table entry.  One straight-line routine then serves
 * the three kinds of level (levels of plain bits have zero-width fields and read entry 0: all digits zero); a wave whose
 * lanes work on levels of different kinds runs it once, not once per kind.
 * (ref: decode_ise, astcenc_integer_sequence.cpp:651-739; the layout of a group: ASTC specification C.2.12) */
struct GroupLayout {
	int sym_at[5], word_at[5], width[5], word_to[5];
	int bits, digit_bits, per;
	uint32_t table_base;
};
WV_FN GroupLayout group_layout(int bits, int kind)
{
	GroupLayout L;
	const uint32_t to = kind == 1 ? 0x75420u : kind == 2 ? 0x00530u : 0u;                                   // four bits per symbol
	const uint32_t wd = kind == 1 ? (2u | 2u << 2 | 1u << 4 | 2u << 6 | 1u << 8) : kind == 2 ? (3u | 2u << 2 | 2u << 4) : 0u;   // two bits per symbol
	for (int e = 0; e < 5; e++)
	{
		L.word_to[e] = (int)((to >> (4 * e)) & 15u);
		L.sym_at[e] = e * bits + L.word_to[e];
		L.word_at[e] = L.sym_at[e] + bits;
		L.width[e] = (int)((wd >> (2 * e)) & 3u);
	}
	L.bits = bits;
	L.digit_bits = kind == 1 ? 2 : kind == 2 ? 3 : 0;
	L.per = ise_group_size(kind);
	L.table_base = kind == 2 ? 256u : 0u;
	return L;
}
/* The five symbols of the group in the 32-bit window g (bits past the end of the stream zero; a symbol past the group's
 * last one comes out as some value below 32: in range of the unquantization table, never stored). */
WV_FN void group_symbols(const GroupLayout& L, uint32_t g, uint32_t out[5])
{
	uint32_t index = 0;
	for (int e = 0; e < 5; e++) index |= ((g >> L.word_at[e]) & ((1u << L.width[e]) - 1u)) << L.word_to[e];
	const uint32_t digits = group_lut(L.table_base + index);
	const uint32_t low_mask = (1u << L.bits) - 1u, digit_mask = (1u << L.digit_bits) - 1u;
	for (int e = 0; e < 5; e++) out[e] = (((digits >> (e * L.digit_bits)) & digit_mask) << L.bits) | ((g >> L.sym_at[e]) & low_mask);
}

/* The same for the colour levels, whose groups can be longer than a window (five trit symbols of six bits: 38 bits): the
 * first three symbols out of the window a at the group's first bit, the last two out of the window b at the fourth symbol's
 * first bit (sym_at[3] of the unsplit layout, returned in `second_window_at`). */
WV_FN GroupLayout group_layout_split(int bits, int kind, int& second_window_at)
{
	GroupLayout L = group_layout(bits, kind);
	second_window_at = L.sym_at[3];
	for (int e = 3; e < 5; e++) { L.sym_at[e] -= second_window_at; L.word_at[e] -= second_window_at; }
	return L;
}
WV_FN void group_symbols_split(const GroupLayout& L, uint32_t a, uint32_t b, uint32_t out[5])
{
	uint32_t index = 0;
	for (int e = 0; e < 5; e++) index |= (((e < 3 ? a : b) >> L.word_at[e]) & ((1u << L.width[e]) - 1u)) << L.word_to[e];
	const uint32_t digits = group_lut(L.table_base + index);
	const uint32_t low_mask = (1u << L.bits) - 1u, digit_mask = (1u << L.digit_bits) - 1u;
	for (int e = 0; e < 5; e++) out[e] = (((digits >> (e * L.digit_bits)) & digit_mask) << L.bits) | (((e < 3 ? a : b) >> L.sym_at[e]) & low_mask);
}

/* A texel that does not leave as an RGBA8 pixel built from integers: error and constant-colour blocks, LNS endpoints,
 * FP16 / FP32 output, the Z swizzle.  cv = the four interpolated 16-bit values (blocks with a payload).
 * `at` = index of the texel's first component.  (ref: decode_texel :66, store_image_block :345) */
WV_FN void store_texel_general(const DecodeImage& img, const DecodeBatch& s, int k, uint32_t flags, int p, const int cv[4], size_t at)
{
	const int profile = (int)img.profile;
	const bool u8_out = img.data_type == 0 || profile == 0;        // (ref: get_u8_component_mask)
	float r, g, bl, a;
	if (flags & 2u)
	{
		r = g = bl = a = int_as_float((int)0xFFFFE000u);
	}
	else if (flags & 4u)
	{
		r = int_as_float((int)s.ep[k][0]); g = int_as_float((int)s.ep[k][1]); bl = int_as_float((int)s.ep[k][2]); a = int_as_float((int)s.ep[k][3]);
	}
	else
	{
		const uint32_t lns = s.lns[k][p];
		int hf[4];
		for (int q = 0; q < 4; q++)
		{
			int cval = cv[q];
			if (u8_out) cval = (cval >> 8) * 257;
			const bool is_lns = (lns & (q == 3 ? 2u : 1u)) != 0u;
			hf[q] = is_lns ? lns_to_sf16(cval) : unorm16_to_sf16(cval);  // (ref: decode_texel :66)
		}
		// FP16 output through the identity swizzle: binary16 -> float -> binary16 gives every finite value back
		// unchanged, so the four halves are stored as they are (infinities / NaNs take the general route)
		const bool halves_out = img.data_type == 1 && img.swz[0] == 0 && img.swz[1] == 1 && img.swz[2] == 2 && img.swz[3] == 3;
		if (halves_out && ((hf[0] & 0x7C00) != 0x7C00) && ((hf[1] & 0x7C00) != 0x7C00) && ((hf[2] & 0x7C00) != 0x7C00) && ((hf[3] & 0x7C00) != 0x7C00))
		{
			const uint32_t lo = (uint32_t)(hf[0] & 0xFFFF) | ((uint32_t)hf[1] << 16), hi = (uint32_t)(hf[2] & 0xFFFF) | ((uint32_t)hf[3] << 16);
			const uint64_t px = ((uint64_t)hi << 32) | lo;
			__builtin_memcpy(static_cast<uint8_t*>(img.data) + at * 2, &px, 8);
			return;
		}
		r = half_to_float((uint16_t)hf[0]); g = half_to_float((uint16_t)hf[1]); bl = half_to_float((uint16_t)hf[2]); a = half_to_float((uint16_t)hf[3]);
	}
	store_texel_at(img, at, r, g, bl, a);
}

/* The RGBA8 pixel of four interpolation results x[q] = lerp_terms(endpoint word q, weight pair) (byte 2 = the value's top
 * byte, byte 3 = 0) through a swizzle that picks channels or constants (sel = swizzle_selector). */
/* (ref: lerp_color_int :37)  The weight w as the pair (256 - 4 w) | 4 w << 16, and (e0 (64 - w) + e1 w + 32) * 4 from an
 * endpoint word e0 | e1 << 16: the reference's value (... + 32) >> 6 is bits 8..23 of it. */
WV_FN uint32_t lerp_weight_pair(int w) { return umad24((uint32_t)w, 4u * 0xFFFFu, 256u); }
WV_FN uint32_t lerp_terms(uint32_t endpoints, uint32_t weight_pair) { return udot2(endpoints, weight_pair, 128u); }

WV_FN uint32_t pixel_from_lerps(uint32_t sel, const uint32_t x[4])
{
	const uint32_t rg = byte_perm(x[1], x[0], 0x0C0C0602u), ba = byte_perm(x[3], x[2], 0x0C0C0602u);   // byte 2 of each, side by side
	return byte_perm(ba, rg, sel);
}

/* The texel phase of a run of 2D blocks: the image rows the run covers, one after the other.  kMulti / kDual: the run
 * has blocks with more than one partition (1: with two at most, 2: with three or four) / with two weight planes
 * (wave-uniform; a run without them skips the partition hash and reads its endpoints once per column / reads one plane).  kGeneral: some texel of the run does not leave as an
 * integer-built RGBA8 pixel (store_texel_general); the builds without it are the RGBA8 decoder's inner loops. */
template <int kMulti, bool kDual, bool kGeneral>
WV_FN void decode_row_texels(const DecodeImage& img, uint32_t bx0, uint32_t by, uint32_t bz, int count, DecodeBatch& s)
{
	const int block_x = (int)img.block_x, block_y = (int)img.block_y;
	const bool small_block = block_x * block_y < 31;
	const bool bytes_out = img.data_type == 0;
	const uint32_t swz_sel = swizzle_selector(img.swz);
	const uint32_t y0 = by * (uint32_t)block_y;
	const int rows = (int)(img.dim_y - y0 < (uint32_t)block_y ? img.dim_y - y0 : (uint32_t)block_y);
	const int row_len = count * block_x;
	const uint32_t x0 = bx0 * (uint32_t)block_x;
	for (int c0 = 0; c0 < row_len; c0 += 64)
	{
		WV_FOR64(l, i_min(64, row_len - c0))
		{
			const int col = c0 + l;
			const int k = (int)(umul24((uint32_t)col, img.bx_inv16) >> 16);
			const int tx = col - mul24(k, block_x);
			const uint32_t xi = x0 + (uint32_t)col;
			if (xi >= img.dim_x) continue;
			const U32x4 rec = load_u32x4_aligned(s.rec[k]);
			const uint32_t ra = rec.x, rc = rec.z, cpx = rec.w;
			const bool fast = !kGeneral || (bytes_out && (ra & 8u) != 0u);         // the pixel comes from integers (or is the block's constant one)
			const bool payload = (ra & 1u) == 0u;
			const int wx = (int)(rc & 15u), wy = (int)((rc >> 4) & 15u);
			const bool dual = kDual && (rc & 0x1000u) != 0u;
			const int parts = (int)((rc >> 13) & 7u);
			const int plane2 = dual ? (int)((rc >> 16) & 3u) : -1;
			// infill: the column's share (ref: unpack_weights :89 via init_decimation_info_2d, astcenc_block_sizes.cpp:261-330)
			const int gs = mad24(mul24((int)img.ds, tx), wx - 1, 32) >> 6;
			const int js = gs >> 4, fs = gs & 0xF;
			const int stride = dual ? 2 : 1;
			const uint8_t* wcol = s.weights[k] + mul24(js, stride);
			const int wrow = mul24(wx, stride);
			const int dty = mul24((int)img.dt, wy - 1);
			// partition hash: the column's share of the four terms, times four -- (hx + hy * y) & 0xFC is a term's six bits above
			// two bits that rank the terms, so that the first of the largest terms is the largest key
			uint32_t hx[4] = { 0u, 0u, 0u, 0u }, hy[4] = { 0u, 0u, 0u, 0u };
			if (kMulti)
			{
				const uint32_t xs = (uint32_t)(small_block ? tx << 1 : tx);
				const U32x4 ht = load_u32x4_aligned(s.hash[k]);
				const uint32_t term[4] = { ht.x, ht.y, ht.z, ht.w };
				for (int q = 0; q < 4; q++)
				{
					hx[q] = umad24(term[q] & 0xFFu, xs, term[q] >> 24) << 2;
					hy[q] = ((term[q] >> 8) & 0xFFu) << 2;
				}
			}
			const uint32_t* epk = s.ep[k];
			U32x4 e = { 0u, 0u, 0u, 0u };
			if (!kMulti) e = load_u32x4_aligned(epk);
			size_t at = (((size_t)bz * img.dim_y + y0) * img.dim_x + xi) * 4;      // (2D blocks: layer bz of the stream is slice bz of the image)
			int gt64 = 32;                                                          // dt * ty * (wy - 1) + 32, row by row
			for (int ty = 0; ty < rows; ty++, at += (size_t)img.dim_x * 4, gt64 += dty)
			{
				int cv[4] = { 0, 0, 0, 0 };
				uint32_t px = cpx;
				int p = 0;
				if (payload)
				{
					const int gt = gt64 >> 6;
					const int jt = gt >> 4, ft = gt & 0xF;
					const int w11 = mad24(fs, ft, 8) >> 4;
					const int w10 = ft - w11, w01 = fs - w11, w00 = 16 - fs - ft + w11;
					// A tap whose factor is zero may lie past the grid (the last row / column interpolates with factor 0): its product is
					// zero whatever is read there, and the read stays inside the scratch -- so the four taps are read unconditionally.
					const uint8_t* g = wcol + mul24(jt, wrow);
					const int w0 = tap_sum4(g[0], w00, g[stride], w01, g[wrow], w10, g[wrow + stride], w11) >> 4;
					int w1 = w0;
					if (kDual && dual) w1 = tap_sum4(g[1], w00, g[3], w01, g[wrow + 1], w10, g[wrow + 3], w11) >> 4;
					if (kMulti)
					{
						if (parts > 1)
						{
							const uint32_t ys = (uint32_t)(small_block ? ty << 1 : ty);
							// (ref: select_partition, astcenc_partition_tables.cpp:216-245: a >= b && a >= c && a >= d -> 0, b >= c && b >= d -> 1, c >= d -> 2, else 3)
							if (kMulti == 1)
							{
								// (two partitions at most in this run: the terms c and d are zero)
								p = (umad24(hy[1], ys, hx[1]) & 0xFCu) > (umad24(hy[0], ys, hx[0]) & 0xFCu) ? 1 : 0;
							}
							else
							{
								const uint32_t a = (umad24(hy[0], ys, hx[0]) & 0xFCu) | 3u, b = (umad24(hy[1], ys, hx[1]) & 0xFCu) | 2u;
								const uint32_t c = (umad24(hy[2], ys, hx[2]) & 0xFCu) | 1u, d = umad24(hy[3], ys, hx[3]) & 0xFCu;
								const uint32_t ab = a > b ? a : b, cd = c > d ? c : d;
								p = 3 - (int)((ab > cd ? ab : cd) & 3u);
							}
						}
						e = load_u32x4_aligned(epk + p * 4);
					}
					const uint32_t wp0 = lerp_weight_pair(w0), wp1 = kDual ? lerp_weight_pair(w1) : wp0;
					uint32_t x[4];
					x[0] = lerp_terms(e.x, kDual && plane2 == 0 ? wp1 : wp0);
					x[1] = lerp_terms(e.y, kDual && plane2 == 1 ? wp1 : wp0);
					x[2] = lerp_terms(e.z, kDual && plane2 == 2 ? wp1 : wp0);
					x[3] = lerp_terms(e.w, kDual && plane2 == 3 ? wp1 : wp0);
					if (fast) px = pixel_from_lerps(swz_sel, x);
					else for (int q = 0; q < 4; q++) cv[q] = (int)(x[q] >> 8);
				}
				if (fast) __builtin_memcpy(static_cast<uint8_t*>(img.data) + at, &px, 4);
				else store_texel_general(img, s, k, ra, p, cv, at);
			}
		}
	}
}

/* Decode blocks bx0 .. bx0 + count - 1 (count <= DECODE_BATCH) of block row `by`, layer `bz` of the stream into the image.
 * All 64 lanes call this.  The arithmetic, block by block, is that of the single-block routines above (parse_block_header,
 * unpack_block_payload, infill_texel_weights: what astcenc_get_block_info runs on the host). */
WV_FN void decode_row_batch(const DecodeImage& img, const uint8_t* blocks, uint32_t bx0, uint32_t by, uint32_t bz, int count, DecodeBatch& s)
{
	const int block_x = (int)img.block_x, block_y = (int)img.block_y, block_z = (int)img.block_z;
	const int T = block_x * block_y * block_z;
	const int profile = (int)img.profile;
	const bool u8_out = img.data_type == 0 || profile == 0;        // (ref: get_u8_component_mask)
	const float error_nan = int_as_float((int)0xFFFFE000u);
	// RGBA8 output whose swizzle only picks channels or constants: a decoded UNORM16 value v leaves as the byte
	// v >> 8 (exact: for every byte b, b * 257 -> FP16 -> float -> * 255 + 0.5 -> int gives b back, which is the
	// reference's route, astcenc_image.cpp:345-420 after decompress_symbolic.cpp:66-120), so the texel is built
	// from integers alone; LNS (HDR) endpoints and the Z swizzle take the general route.
	const bool bytes_swz = img.data_type == 0 && img.swz[0] < 6 && img.swz[1] < 6 && img.swz[2] < 6 && img.swz[3] < 6;
	const size_t first = ((size_t)bz * img.blocks_y + by) * img.blocks_x + bx0;

	// ---- headers and constant colours: one lane per block ----
	bool multi_part = false, many_part = false, dual_part = false;      // per-lane partials, folded below
	WV_FOR64(k, 64)
	{
		// (every lane loads a block -- the lanes past the run's last block that one again -- and its share of the weight
		//  unquantization table the wave keeps in LDS, so that all these loads are in flight together)
		Bits128 blk;
		const uint32_t* p = reinterpret_cast<const uint32_t*>(blocks + (first + (size_t)i_min(k, count - 1)) * 16);
		blk.w[0] = p[0]; blk.w[1] = p[1]; blk.w[2] = p[2]; blk.w[3] = p[3];
		{
			const uint32_t t0 = weight_unquant_lut_word(k), t1 = weight_unquant_lut_word(64 + (k & 31));
			__builtin_memcpy(s.wunq + 4 * k, &t0, 4);
			if (k < 32) __builtin_memcpy(s.wunq + 4 * (64 + k), &t1, 4);
		}
		if (k >= count) continue;
		const DecodeTables* tabs = img.tabs;
#if WV_DEVICE
		__builtin_assume(tabs != nullptr);       // (the launch always passes the tables: no arithmetic fallback in the kernel)
#endif
		const BlockHeader h = parse_block_header(blk, block_x, block_y, block_z, tabs);
		bool error = h.error;
		float cc[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
		if (h.constant && !error)
		{
			// constant colour (ref: decompress_symbolic.cpp:204-255)
			if (h.constant_f16)
			{
				// FP16 constant colour: legal in the HDR profiles only
				if (profile == 2 || profile == 3)
				{
					for (int q = 0; q < 4; q++) cc[q] = half_to_float((uint16_t)h.const_color[q]);
				}
				else error = true;
			}
			else
			{
				for (int q = 0; q < 4; q++)
				{
					int v = u8_out ? (h.const_color[q] >> 8) * 257 : h.const_color[q];
					cc[q] = half_to_float((uint16_t)unorm16_to_sf16(v));
				}
			}
		}
		const bool skip = error || h.constant;
		const uint32_t wq = btq_packed(h.wquant), cq = btq_packed(skip ? 0 : h.cquant);
		const int wkind = (int)(wq >> 4), ckind = (int)(cq >> 4), wqbits = (int)(wq & 15u), cqbits = (int)(cq & 15u);
		const int wcount = h.wx * h.wy * h.wz;
		const int real_wcount = h.dual ? 2 * wcount : wcount;
		const int wgroups = skip ? 0 : ise_group_count(real_wcount, wkind), cgroups = skip ? 0 : ise_group_count(h.nvals, ckind);
		const int wglen = wkind == 1 ? 5 * wqbits + 8 : wkind == 2 ? 3 * wqbits + 7 : 4 * wqbits;
		uint32_t cpx = 0u;
		if (skip && img.data_type == 0) cpx = error ? pack_texel_u8(img, error_nan, error_nan, error_nan, error_nan) : pack_texel_u8(img, cc[0], cc[1], cc[2], cc[3]);
		// (fast: the texel phase forms -- or, for these two kinds of block, already has -- the RGBA8 pixel; LNS endpoints clear the flag)
		const bool fast = skip ? img.data_type == 0 : bytes_swz;
		s.rec[k][0] = (skip ? 1u : 0u) | (error ? 2u : 0u) | ((h.constant && !error) ? 4u : 0u) | (fast ? 8u : 0u) | ((uint32_t)wqbits << 4) | ((uint32_t)wkind << 8) |
		              ((uint32_t)h.wquant << 12) | ((uint32_t)wgroups << 16) | ((uint32_t)wglen << 24);
		s.rec[k][1] = (uint32_t)cqbits | ((uint32_t)ckind << 4) | ((uint32_t)(skip ? 0 : h.cquant) << 8) | ((uint32_t)h.nvals << 16) | ((uint32_t)h.color_start << 24);
		s.rec[k][2] = (uint32_t)h.wx | ((uint32_t)h.wy << 4) | ((uint32_t)h.wz << 8) | ((h.dual ? 1u : 0u) << 12) | ((uint32_t)h.parts << 13) |
		              (((uint32_t)h.plane2 & 3u) << 16) | ((uint32_t)cgroups << 20);
		s.rec[k][3] = cpx;
		if (!skip && h.parts > 1)
		{
			const PartitionHash ph = partition_hash_setup(h.seed, h.parts);
			for (int q = 0; q < 4; q++) s.hash[k][q] = ph.term[q];
		}
		else for (int q = 0; q < 4; q++) s.hash[k][q] = 0u;
		for (int q = 0; q < 4; q++) { s.lns[k][q] = 0; s.fmt[k][q] = (uint8_t)h.fmt[q]; }
		if (h.constant && !error)
		{
			for (int q = 0; q < 4; q++) s.ep[k][q] = (uint32_t)float_as_int(cc[q]);
		}
		if (!skip)
		{
			uint32_t* bits = decode_batch_bits(s, k);
			uint32_t* ws = decode_batch_wstream(s, k);
			const Bits128 rv = bits_reversed(blk);
			// the colour values end at bit color_start + their BISE size; the weight stream is 24..96 bits long.  Bits past the
			// end of either read as zeros, so that a short last group needs no mask
			const int cend = h.color_start + h.nvals * cqbits + (ckind == 1 ? (8 * h.nvals + 4) / 5 : ckind == 2 ? (7 * h.nvals + 2) / 3 : 0);    // (ise_bitcount)
			for (int q = 0; q < 4; q++)
			{
				const int left = cend - 32 * q;
				bits[q] = left >= 32 ? blk.w[q] : left <= 0 ? 0u : blk.w[q] & ((1u << left) - 1u);
			}
			bits[4] = 0u;
			for (int q = 0; q < 3; q++)
			{
				const int left = h.wbits - 32 * q;
				ws[q] = left >= 32 ? rv.w[q] : left <= 0 ? 0u : rv.w[q] & ((1u << left) - 1u);
			}
			ws[3] = 0u;
			multi_part = multi_part || h.parts > 1;
			many_part = many_part || h.parts > 2;
			dual_part = dual_part || h.dual;
		}
	}
	const bool any_multi = wv_any(multi_part), any_many = wv_any(many_part), any_dual = wv_any(dual_part);
	WV_SYNC();

	// ---- weights and colour values: lane l works on block l & (DECODE_BATCH - 1), on every DECODE_SLOTS-th group of it ----
	WV_FOR64(l, 64)
	{
		const int k = l & (DECODE_BATCH - 1);
		if (k >= count) continue;
		const uint32_t ra = s.rec[k][0];
		const int groups = (int)((ra >> 16) & 63u);
		const int glen = (int)((ra >> 24) & 31u);
		const GroupLayout L = group_layout((int)((ra >> 4) & 7u), (int)((ra >> 8) & 3u));
		const uint32_t* ws = decode_batch_wstream(s, k);
		const uint8_t* unq = s.wunq + ((ra >> 12) & 15u) * 32u;
		// Three groups per trip, side by side: the phase is a chain of dependent reads (stream window -> group table -> weight
		// table) that a wave cannot overlap with anything but another group's chain.  A slot past the block's last group decodes
		// the last group again and stores nothing.
		for (int g0 = l >> DECODE_BATCH_LOG2; g0 < groups; g0 += 3 * DECODE_SLOTS)
		{
			uint32_t sym[3][5];
			for (int u = 0; u < 3; u++) group_symbols(L, bits_window32(ws, mul24(i_min(g0 + u * DECODE_SLOTS, groups - 1), glen)), sym[u]);
			uint8_t w[3][5];
			for (int u = 0; u < 3; u++) for (int e = 0; e < 5; e++) w[u][e] = unq[sym[u][e]];
			for (int u = 0; u < 3; u++)
			{
				const int g = g0 + u * DECODE_SLOTS;
				if (g >= groups) break;
				uint8_t* out = s.weights[k] + mul24(g, L.per);
				out[0] = w[u][0]; out[1] = w[u][1]; out[2] = w[u][2];
				if (L.per > 3) out[3] = w[u][3];
				if (L.per > 4) out[4] = w[u][4];
			}
		}
	}
	WV_FOR64(l, 64)
	{
		const int k = l & (DECODE_BATCH - 1);
		if (k >= count) continue;
		const uint32_t rb = s.rec[k][1];
		const int groups = (int)((s.rec[k][2] >> 20) & 7u);
		const int nvals = (int)((rb >> 16) & 31u);
		const int bits = (int)(rb & 15u), kind = (int)((rb >> 4) & 3u);
		const int glen = kind == 1 ? 5 * bits + 8 : kind == 2 ? 3 * bits + 7 : 4 * bits;
		const int cquant = (int)((rb >> 8) & 31u), cstart = (int)((rb >> 24) & 31u);
		int second_at;
		const GroupLayout L = group_layout_split(bits, kind, second_at);
		const uint32_t* cs = decode_batch_bits(s, k);
		// (three groups per trip, as for the weights; a block has at most six groups of colour values)
		for (int g0 = l >> DECODE_BATCH_LOG2; g0 < groups; g0 += 3 * DECODE_SLOTS)
		{
			uint32_t sym[3][5];
			for (int u = 0; u < 3; u++)
			{
				const int at = cstart + mul24(i_min(g0 + u * DECODE_SLOTS, groups - 1), glen);
				group_symbols_split(L, bits_window32(cs, at), bits_window32(cs, at + second_at), sym[u]);
			}
			// (the table reads side by side, then the stores: one wait; the symbols past a short group are in range -- below 256)
			uint8_t c[3][5];
			for (int u = 0; u < 3; u++) for (int e = 0; e < 5; e++) c[u][e] = (uint8_t)color_unquant_lut(cquant, (int)sym[u][e]);
			for (int u = 0; u < 3; u++)
			{
				const int g = g0 + u * DECODE_SLOTS;
				if (g >= groups) break;
				const int n = nvals - mul24(g, L.per);
				uint8_t* out = s.colors[k] + mul24(g, L.per);
				out[0] = c[u][0];
				if (n > 1) out[1] = c[u][1];
				if (n > 2) out[2] = c[u][2];
				if (n > 3 && L.per > 3) out[3] = c[u][3];
				if (n > 4 && L.per > 4) out[4] = c[u][4];
			}
		}
	}
	WV_SYNC();
	// ---- endpoints: lane l works on block l & (DECODE_BATCH - 1), on every DECODE_SLOTS-th partition of it ----
	WV_FOR64(l, 64)
	{
		const int k = l & (DECODE_BATCH - 1);
		if (k >= count) continue;
		const uint32_t ra = s.rec[k][0];
		if (ra & 1u) continue;
		const int parts = (int)((s.rec[k][2] >> 13) & 7u);
		for (int p = l >> DECODE_BATCH_LOG2; p < parts; p += DECODE_SLOTS)
		{
			int start = 0;
			for (int i = 0; i < 4; i++) start += i < p ? 2 * (s.fmt[k][i] >> 2) + 2 : 0;
			const int f = s.fmt[k][p];
			uint8_t in[8];
			const int n = 2 * (f >> 2) + 2;
			for (int q = 0; q < 8; q++) in[q] = q < n ? s.colors[k][start + q] : 0;
			i4 e0, e1;
			unpack_color_endpoints(profile, f, in, e0, e1);
			bool rgb_lns, alpha_lns;
			endpoint_lns_flags(profile, f, rgb_lns, alpha_lns);
			s.lns[k][p] = (uint8_t)((rgb_lns ? 1 : 0) | (alpha_lns ? 2 : 0));
			uint32_t* o = s.ep[k] + p * 4;
			o[0] = (uint32_t)e0.x | ((uint32_t)e1.x << 16); o[1] = (uint32_t)e0.y | ((uint32_t)e1.y << 16);
			o[2] = (uint32_t)e0.z | ((uint32_t)e1.z << 16); o[3] = (uint32_t)e0.w | ((uint32_t)e1.w << 16);
		}
	}
	WV_SYNC();
	// (a block with LNS endpoints in any partition leaves the integer pixel path)
	bool general_part = false;
	WV_FOR64(k, count)
	{
		uint32_t any_lns;
		__builtin_memcpy(&any_lns, s.lns[k], 4);
		uint32_t ra = s.rec[k][0];
		if (any_lns) { ra &= ~8u; s.rec[k][0] = ra; }
		general_part = general_part || !(ra & 8u);
	}
	const bool any_general = wv_any(general_part);
	WV_SYNC();

	// ---- texels ----
	if (block_z == 1)
	{
		if (any_general) decode_row_texels<2, true, true>(img, bx0, by, bz, count, s);
		else if (any_many)
		{
			if (any_dual) decode_row_texels<2, true, false>(img, bx0, by, bz, count, s);
			else decode_row_texels<2, false, false>(img, bx0, by, bz, count, s);
		}
		else if (any_multi)
		{
			if (any_dual) decode_row_texels<1, true, false>(img, bx0, by, bz, count, s);
			else decode_row_texels<1, false, false>(img, bx0, by, bz, count, s);
		}
		else
		{
			if (any_dual) decode_row_texels<0, true, false>(img, bx0, by, bz, count, s);
			else decode_row_texels<0, false, false>(img, bx0, by, bz, count, s);
		}
	}
	else
	{
		// 3D blocks: one lane per (block, texel)
		const bool small_block = T < 31;
		const uint32_t swz_sel = swizzle_selector(img.swz);
		WV_FOR(j, count * T)
		{
			const int k = (int)(((uint32_t)j * img.t_inv24) >> 24);
			const int t = j - k * T;
			const int tz = (int)(((uint32_t)t * img.bxy_inv16) >> 16);
			const int trem = t - tz * (block_x * block_y);
			const int ty = (int)(((uint32_t)trem * img.bx_inv16) >> 16);
			const int tx = trem - ty * block_x;
			const uint32_t xi = (bx0 + (uint32_t)k) * (uint32_t)block_x + (uint32_t)tx;
			const uint32_t yi = by * (uint32_t)block_y + (uint32_t)ty;
			const uint32_t zi = bz * (uint32_t)block_z + (uint32_t)tz;
			if (xi >= img.dim_x || yi >= img.dim_y || zi >= img.dim_z) continue;
			const size_t at = (((size_t)zi * img.dim_y + yi) * img.dim_x + xi) * 4;
			const uint32_t ra = s.rec[k][0], rc = s.rec[k][2];
			const bool fast = img.data_type == 0 && (ra & 8u) != 0u;
			uint32_t px = s.rec[k][3];
			int cv[4] = { 0, 0, 0, 0 };
			int p = 0;
			if (!(ra & 1u))
			{
				const bool dual = (rc & 0x1000u) != 0u;
				const int parts = (int)((rc >> 13) & 7u);
				const int plane2 = dual ? (int)((rc >> 16) & 3u) : -1;
				int wp[2];
				infill_texel_weights((int)(rc & 15u), (int)((rc >> 4) & 15u), (int)((rc >> 8) & 15u), dual, s.weights[k], s.weights[k] + (dual ? 1 : 0), dual ? 2 : 1,
				                     (int)img.ds, (int)img.dt, (int)img.dr, block_z, tx, ty, tz, wp);
				if (parts > 1)
				{
					PartitionHash ph;
					for (int q = 0; q < 4; q++) ph.term[q] = s.hash[k][q];
					p = partition_from_hash(ph, tx, ty, tz, small_block);
				}
				uint32_t x[4];
				for (int q = 0; q < 4; q++) x[q] = lerp_terms(s.ep[k][p * 4 + q], lerp_weight_pair(q == plane2 ? wp[1] : wp[0]));
				if (fast) px = pixel_from_lerps(swz_sel, x);
				else for (int q = 0; q < 4; q++) cv[q] = (int)(x[q] >> 8);
			}
			if (fast) __builtin_memcpy(static_cast<uint8_t*>(img.data) + at, &px, 4);
			else store_texel_general(img, s, k, ra, p, cv, at);
		}
	}
	WV_SYNC();          // the batch scratch is reused by the next call
}

#endif // !ASTC_DECODE_NO_LUTS

/* astcenc_get_block_info for one block, as plain sequential code (runs on the host).
 * (ref: astcenc_get_block_info, astcenc_entry.cpp:1401-1517)  `info` is a struct astcenc_block_info. */
template <typename BlockInfo>
WV_FN void describe_block(const uint8_t* pcb, int block_x, int block_y, int block_z, int profile, BlockInfo* info, DecodeScratch& s)
{
	const int T = block_x * block_y * block_z;
	Bits128 blk;
	for (int k = 0; k < 4; k++) blk.w[k] = (uint32_t)pcb[4 * k] | ((uint32_t)pcb[4 * k + 1] << 8) | ((uint32_t)pcb[4 * k + 2] << 16) | ((uint32_t)pcb[4 * k + 3] << 24);
	const BlockHeader h = parse_block_header(blk, block_x, block_y, block_z);

	info->block_x = (unsigned)block_x; info->block_y = (unsigned)block_y; info->block_z = (unsigned)block_z;
	info->texel_count = (unsigned)T;
	info->is_error_block = h.error;
	if (h.error) return;
	info->is_constant_block = h.constant;
	if (h.constant) return;

	unpack_block_payload(blk, h, profile, s);
	info->weight_x = (unsigned)h.wx; info->weight_y = (unsigned)h.wy; info->weight_z = (unsigned)h.wz;
	info->is_dual_plane_block = h.dual;
	info->partition_count = (unsigned)h.parts;
	info->partition_index = (unsigned)h.seed;
	info->dual_plane_component = (unsigned)h.plane2;          // -1 (all ones) for single-plane blocks, as the reference
	info->color_level_count = quant_level_count(h.cquant);
	info->weight_level_count = quant_level_count(h.wquant);
	for (int p = 0; p < h.parts; p++)
	{
		info->color_endpoint_modes[p] = (unsigned)h.fmt[p];
		const bool rgb_lns = s.lns[p][0] != 0, alpha_lns = s.lns[p][1] != 0;
		info->is_hdr_block = info->is_hdr_block || rgb_lns || alpha_lns;
		for (int j = 0; j < 2; j++)
		{
			for (int k = 0; k < 4; k++)
			{
				const int v = s.ep[p][j * 4 + k];
				const bool lns = k == 3 ? alpha_lns : rgb_lns;
				info->color_endpoints[p][j][k] = half_to_float((uint16_t)(lns ? lns_to_sf16(v) : unorm16_to_sf16(v)));
			}
		}
	}
	const bool small_block = T < 31;
	for (int t = 0; t < T; t++)
	{
		const int tz = t / (block_x * block_y), trem = t - tz * (block_x * block_y);
		const int ty = trem / block_x, tx = trem - ty * block_x;
		int wp[2];
		infill_texel_weights(h, s.weights, (1024 + block_x / 2) / (block_x - 1), (1024 + block_y / 2) / (block_y - 1),
		                     block_z > 1 ? (1024 + block_z / 2) / (block_z - 1) : 0, block_z, tx, ty, tz, wp);
		info->weight_values_plane1[t] = (float)wp[0] * (1.0f / 16.0f);
		if (h.dual) info->weight_values_plane2[t] = (float)wp[1] * (1.0f / 16.0f);
		info->partition_assignment[t] = (uint8_t)(h.parts == 1 ? 0 : partition_of_texel(h.seed, tx, ty, tz, h.parts, small_block));
	}
}

} } // namespace astcd::ASTC_VARIANT
