// SPDX-License-Identifier: Apache-2.0
// Host-side builder of the read-only table blob (astc_tables.h) that the wavefront block
// compressor reads.  Runs once per context; the blob is then uploaded to HBM.
//
// Behavioural reference (what the tables must contain so that the search visits the same
// candidates in the same order as the reference encoder):
//   block modes / decimation tables : Source/astcenc_block_sizes.cpp:36-137, :252-436, :822-1012
//   partition tables + coverage     : Source/astcenc_partition_tables.cpp:114-497
//   percentile cut                  : Source/astcenc_percentile_tables.cpp:1165-1200
//   colour / weight quant tables    : ASTC spec C.2.13 / C.2.17 (the reference ships them as data,
//                                     Source/astcenc_quantization.cpp, astcenc_weight_quant_xfer_tables.cpp)
//   BISE trit/quint packing         : ASTC spec C.2.12 (reference data: astcenc_integer_sequence.cpp:28-330)
//   angular sin/cos tables          : Source/astcenc_weight_align.cpp:72-84 (host libm sinf/cosf)
#include "host_tables.h"

#include <cmath>
#include <cstring>
#include <algorithm>

#include "percentile_data.inc"

namespace astcd {

// ---------------------------------------------------------------------------------------------
// BISE sizes
// ---------------------------------------------------------------------------------------------
struct BtqCount { uint8_t bits, trits, quints; };
static const BtqCount btq_counts[21] = {
	{1,0,0},{0,1,0},{2,0,0},{0,0,1},{1,1,0},{3,0,0},{1,0,1},{2,1,0},{4,0,0},{2,0,1},{3,1,0},
	{5,0,0},{3,0,1},{4,1,0},{6,0,0},{4,0,1},{5,1,0},{7,0,0},{5,0,1},{6,1,0},{8,0,0}
};

static const unsigned int quant_levels[21] = {
	2,3,4,5,6,8,10,12,16,20,24,32,40,48,64,80,96,128,160,192,256
};

unsigned int get_quant_level(unsigned int q) { return quant_levels[q]; }

/* Bits needed to BISE-encode `count` symbols at quant level q. (ref: integer_sequence.cpp:419) */
unsigned int ise_sequence_bitcount(unsigned int count, unsigned int q)
{
	if (q >= 21) return 1024;
	const BtqCount& b = btq_counts[q];
	// n bits each, plus 8 bits per 5 trits or 7 bits per 3 quints, rounded up
	unsigned int total = b.bits * count;
	if (b.trits)  total += (8 * count + 4) / 5;
	if (b.quints) total += (7 * count + 2) / 3;
	return total;
}

// ---------------------------------------------------------------------------------------------
// Blob writer
// ---------------------------------------------------------------------------------------------
struct Blob {
	std::vector<uint8_t> d;
	uint32_t alloc(size_t bytes, size_t align = 16)
	{
		size_t off = (d.size() + align - 1) / align * align;
		d.resize(off + bytes, 0);
		return static_cast<uint32_t>(off);
	}
	template <typename T> T* at(uint32_t off) { return reinterpret_cast<T*>(d.data() + off); }
};

// ---------------------------------------------------------------------------------------------
// Block mode decode (2D). (ref: block_sizes.cpp:36-137; ASTC spec table C.2.8)
// ---------------------------------------------------------------------------------------------
static bool decode_block_mode_2d(unsigned int mode, unsigned int& wx, unsigned int& wy,
                                 bool& dual, unsigned int& quant, unsigned int& wbits)
{
	unsigned int r0 = (mode >> 4) & 1;
	unsigned int H = (mode >> 9) & 1;
	unsigned int D = (mode >> 10) & 1;
	unsigned int A = (mode >> 5) & 3;
	unsigned int R;
	wx = wy = 0;

	if (mode & 3)
	{
		R = r0 | ((mode & 3) << 1);
		unsigned int B = (mode >> 7) & 3;
		switch ((mode >> 2) & 3)
		{
		case 0: wx = B + 4; wy = A + 2; break;
		case 1: wx = B + 8; wy = A + 2; break;
		case 2: wx = A + 2; wy = B + 8; break;
		default:
			B &= 1;
			if (mode & 0x100) { wx = B + 2; wy = A + 2; }
			else              { wx = A + 2; wy = B + 6; }
			break;
		}
	}
	else
	{
		unsigned int r21 = (mode >> 2) & 3;
		if (r21 == 0) return false;
		R = r0 | (r21 << 1);
		unsigned int B = (mode >> 9) & 3;
		switch ((mode >> 7) & 3)
		{
		case 0: wx = 12; wy = A + 2; break;
		case 1: wx = A + 2; wy = 12; break;
		case 2: wx = A + 6; wy = B + 6; D = 0; H = 0; break;
		default:
			switch ((mode >> 5) & 3)
			{
			case 0: wx = 6; wy = 10; break;
			case 1: wx = 10; wy = 6; break;
			default: return false;
			}
			break;
		}
	}

	unsigned int count = wx * wy * (D + 1);
	quant = (R - 2) + 6 * H;
	dual = D != 0;
	wbits = ise_sequence_bitcount(count, quant);
	return count <= (unsigned)MAX_WEIGHTS && wbits >= 24 && wbits <= 96;
}

// ---------------------------------------------------------------------------------------------
// Block mode decode (3D). (ref: block_sizes.cpp:152-243; ASTC spec table C.2.10)
// ---------------------------------------------------------------------------------------------
static bool decode_block_mode_3d(unsigned int mode, unsigned int& wx, unsigned int& wy, unsigned int& wz,
                                 bool& dual, unsigned int& quant, unsigned int& wbits)
{
	unsigned int r0 = (mode >> 4) & 1;
	unsigned int H = (mode >> 9) & 1;
	unsigned int D = (mode >> 10) & 1;
	const unsigned int A = (mode >> 5) & 3;
	unsigned int R;
	wx = wy = wz = 0;

	if (mode & 3)
	{
		R = r0 | ((mode & 3) << 1);
		wx = A + 2;
		wy = ((mode >> 7) & 3) + 2;
		wz = ((mode >> 2) & 3) + 2;
	}
	else
	{
		const unsigned int r21 = (mode >> 2) & 3;
		if (r21 == 0) return false;
		R = r0 | (r21 << 1);
		const unsigned int B = (mode >> 9) & 3;
		const unsigned int sel = (mode >> 7) & 3;
		if (sel != 3) { D = 0; H = 0; }
		switch (sel)
		{
		case 0: wx = 6; wy = B + 2; wz = A + 2; break;
		case 1: wx = A + 2; wy = 6; wz = B + 2; break;
		case 2: wx = A + 2; wy = B + 2; wz = 6; break;
		default:
			wx = wy = wz = 2;
			switch (A)
			{
			case 0: wx = 6; break;
			case 1: wy = 6; break;
			case 2: wz = 6; break;
			default: return false;
			}
			break;
		}
	}

	unsigned int count = wx * wy * wz * (D + 1);
	quant = (R - 2) + 6 * H;
	dual = D != 0;
	wbits = ise_sequence_bitcount(count, quant);
	return count <= (unsigned)MAX_WEIGHTS && wbits >= 24 && wbits <= 96;
}

// ---------------------------------------------------------------------------------------------
// Decimation (infill) tables for one weight grid: bilinear for 2D footprints (ref:
// block_sizes.cpp:252-436), simplex interpolation over the 3D grid cell for 3D ones (ref: :450-700).
// ---------------------------------------------------------------------------------------------
static void build_decimation_info(Blob& blob, uint32_t di_off, unsigned int tx, unsigned int ty, unsigned int tz,
                                  unsigned int wx, unsigned int wy, unsigned int wz)
{
	const unsigned int T = tx * ty * tz, W = wx * wy * wz;
	std::vector<uint8_t> cnt_t(T, 0), cnt_w(W, 0);
	std::vector<uint8_t> gw(T * 4, 0), gc(T * 4, 0);            // per texel: weight ids, contribs
	std::vector<std::vector<uint8_t>> tw(W), tc(W);            // per weight: texel ids, contribs

	for (unsigned int z = 0; z < tz; z++)
	for (unsigned int y = 0; y < ty; y++)
	{
		for (unsigned int x = 0; x < tx; x++)
		{
			unsigned int texel = (z * ty + y) * tx + x;
			unsigned int xs = (((1024 + tx / 2) / (tx - 1)) * x * (wx - 1) + 32) >> 6;
			unsigned int ys = (((1024 + ty / 2) / (ty - 1)) * y * (wy - 1) + 32) >> 6;
			unsigned int xf = xs & 0xF, yf = ys & 0xF, xi = xs >> 4, yi = ys >> 4;

			unsigned int q[4], w[4];
			if (tz == 1)
			{
				q[0] = xi + yi * wx; q[1] = q[0] + 1; q[2] = q[0] + wx; q[3] = q[2] + 1;
				unsigned int prod = xf * yf;
				w[3] = (prod + 8) >> 4;
				w[1] = xf - w[3];
				w[2] = yf - w[3];
				w[0] = 16 - xf - yf + w[3];
			}
			else
			{
				unsigned int zs = (((1024 + tz / 2) / (tz - 1)) * z * (wz - 1) + 32) >> 6;
				unsigned int zf = zs & 0xF, zi = zs >> 4;
				// walk from the cell's low corner to its high corner along the axes in descending
				// order of fraction; ties resolved as the spec's comparison chain does
				const unsigned int N = wx, NM = wx * wy;
				const unsigned int fs = xf, ft = yf, fp = zf;
				const unsigned int cas = ((fs > ft) << 2) + ((ft > fp) << 1) + (fs > fp);
				unsigned int s1, s2;
				switch (cas)
				{
				case 7: s1 = 1;  s2 = N;  w[0] = 16 - fs; w[1] = fs - ft; w[2] = ft - fp; w[3] = fp; break;
				case 3: s1 = N;  s2 = 1;  w[0] = 16 - ft; w[1] = ft - fs; w[2] = fs - fp; w[3] = fp; break;
				case 5: s1 = 1;  s2 = NM; w[0] = 16 - fs; w[1] = fs - fp; w[2] = fp - ft; w[3] = ft; break;
				case 4: s1 = NM; s2 = 1;  w[0] = 16 - fp; w[1] = fp - fs; w[2] = fs - ft; w[3] = ft; break;
				case 2: s1 = N;  s2 = NM; w[0] = 16 - ft; w[1] = ft - fp; w[2] = fp - fs; w[3] = fs; break;
				default: s1 = NM; s2 = N; w[0] = 16 - fp; w[1] = fp - ft; w[2] = ft - fs; w[3] = fs; break;
				}
				q[0] = (zi * wy + yi) * wx + xi;
				q[3] = ((zi + 1) * wy + (yi + 1)) * wx + (xi + 1);
				q[1] = q[0] + s1;
				q[2] = q[1] + s2;
			}

			for (int i = 0; i < 4; i++)
			{
				if (!w[i]) continue;
				gw[texel * 4 + cnt_t[texel]] = (uint8_t)q[i];
				gc[texel * 4 + cnt_t[texel]] = (uint8_t)w[i];
				cnt_t[texel]++;
				tw[q[i]].push_back((uint8_t)texel);
				tc[q[i]].push_back((uint8_t)w[i]);
				cnt_w[q[i]]++;
			}
		}
	}

	unsigned int rows = 0, max_tw = 0;
	for (unsigned int i = 0; i < W; i++) rows = std::max<unsigned>(rows, cnt_w[i]);
	for (unsigned int i = 0; i < T; i++) max_tw = std::max<unsigned>(max_tw, cnt_t[i]);

	// the three texel tables as one record per texel ([T][4]; the staged copy in LDS and the sweeps that read the blob both
	// fetch a texel's four indices / contributions with one 32-bit / 128-bit access)
	uint32_t o_tw  = blob.alloc(4 * T, 16);
	uint32_t o_tci = blob.alloc(4 * T, 4);
	uint32_t o_tcf = blob.alloc(4 * T * sizeof(float), 16);
	uint32_t o_wtc = blob.alloc(W, 4);
	uint32_t o_wt  = blob.alloc(rows * W, 4);
	uint32_t o_tcw = blob.alloc(rows * W * sizeof(float), 4);
	uint32_t o_ro  = blob.alloc(W, 4);          // realign schedule, filled by build_realign_schedules()
	uint32_t o_rc  = blob.alloc(W, 4);
	// (the weight contributions feed the decimation sweeps only, which read them from HBM / L2: behind the range that a
	//  refined candidate stages into LDS, DecimationInfo::table_bytes)
	uint32_t o_wc  = blob.alloc(rows * W * sizeof(float), 4);
	uint32_t o_later = blob.alloc((size_t)W * REALIGN_LATER_MAX, 4);   // later neighbours (not part of the staged range), see build_realign_schedule()

	uint8_t* p_tw = blob.at<uint8_t>(o_tw);
	uint8_t* p_tci = blob.at<uint8_t>(o_tci);
	float*   p_tcf = blob.at<float>(o_tcf);
	for (unsigned int t = 0; t < T; t++)
	{
		for (unsigned int j = 0; j < 4; j++)
		{
			bool used = j < cnt_t[t];
			p_tw[t * 4 + j]  = used ? gw[t * 4 + j] : 0;
			p_tci[t * 4 + j] = used ? gc[t * 4 + j] : 0;
			p_tcf[t * 4 + j] = used ? (float)gc[t * 4 + j] * (1.0f / 16.0f) : 0.0f;
		}
	}

	uint8_t* p_wtc = blob.at<uint8_t>(o_wtc);
	uint8_t* p_wt = blob.at<uint8_t>(o_wt);
	float* p_wc = blob.at<float>(o_wc);
	float* p_tcw = blob.at<float>(o_tcw);
	for (unsigned int w = 0; w < W; w++)
	{
		p_wtc[w] = cnt_w[w];
		for (unsigned int j = 0; j < rows; j++)
		{
			if (j < cnt_w[w])
			{
				unsigned int texel = tw[w][j];
				p_wt[j * W + w] = (uint8_t)texel;
				p_wc[j * W + w] = (float)tc[w][j];
				float c = 0.0f;
				for (unsigned int k = 0; k < 4; k++)
				{
					if (p_tw[texel * 4 + k] == w && p_tcf[texel * 4 + k] != 0.0f)
					{
						c = p_tcf[texel * 4 + k];
						break;
					}
				}
				p_tcw[j * W + w] = c;
			}
			else
			{
				// padding rows contribute exactly zero (ref: block_sizes.cpp:387-392)
				p_wt[j * W + w] = tw[w][cnt_w[w] - 1];
				p_wc[j * W + w] = 0.0f;
				p_tcw[j * W + w] = 0.0f;
			}
		}
	}

	DecimationInfo* di = blob.at<DecimationInfo>(di_off);
	di->texel_count = (uint8_t)T;
	di->weight_count = (uint8_t)W;
	di->max_texel_weight_count = (uint8_t)max_tw;
	di->weight_x = (uint8_t)wx;
	di->weight_y = (uint8_t)wy;
	di->max_weight_texel_count = (uint8_t)rows;
	di->off_texel_weights = o_tw;
	di->off_texel_contribs_int = o_tci;
	di->off_texel_contribs_f = o_tcf;
	di->off_weight_texel_count = o_wtc;
	di->off_weight_texels = o_wt;
	di->off_weight_contribs = o_wc;
	di->off_texel_contrib_for_weight = o_tcw;
	di->off_realign_order = o_ro;
	di->off_realign_counts = o_rc;
	di->off_realign_later = o_later;
	di->realign_speculative = 0;
	di->table_bytes = (uint32_t)(o_rc + ((W + 3u) & ~3u) - o_tw);
}

/* Realign schedule of one decimation grid (see DecimationInfo): greedy levelling in index order.
 * level[w] = first level above every earlier weight that shares a texel with w, and with room left. */
static void build_realign_schedule(Blob& blob, uint32_t di_off, unsigned int max_slots)
{
	DecimationInfo* di = blob.at<DecimationInfo>(di_off);
	const unsigned int W = di->weight_count, T = di->texel_count, rows = di->max_weight_texel_count;
	const uint8_t* wtc = blob.at<uint8_t>(di->off_weight_texel_count);
	const uint8_t* wt = blob.at<uint8_t>(di->off_weight_texels);
	std::vector<int> level(W, 0), fill;
	std::vector<int> texel_level(T, -1);          // highest level of any earlier weight touching the texel
	(void)rows;
	for (unsigned int w = 0; w < W; w++)
	{
		int need = 0;
		for (unsigned int j = 0; j < wtc[w]; j++) need = std::max(need, texel_level[wt[j * W + w]] + 1);
		int L = need;
		while ((int)fill.size() <= L) fill.push_back(0);
		while (fill[L] >= (int)max_slots) { L++; if ((int)fill.size() <= L) fill.push_back(0); }
		level[w] = L;
		fill[L]++;
		for (unsigned int j = 0; j < wtc[w]; j++) texel_level[wt[j * W + w]] = std::max(texel_level[wt[j * W + w]], L);
	}
	uint8_t* order = blob.at<uint8_t>(di->off_realign_order);
	uint8_t* counts = blob.at<uint8_t>(di->off_realign_counts);
	unsigned int pos = 0, nlev = 0;
	for (size_t L = 0; L < fill.size(); L++)
	{
		if (fill[L] == 0) continue;
		for (unsigned int w = 0; w < W; w++) if (level[w] == (int)L) order[pos++] = (uint8_t)w;
		counts[nlev++] = (uint8_t)fill[L];
	}
	blob.at<DecimationInfo>(di_off)->realign_levels = (uint8_t)nlev;

	// later neighbours of every weight (see DecimationInfo::off_realign_later)
	std::vector<uint8_t> later((size_t)W * REALIGN_LATER_MAX, 255);
	bool fits = true;
	{
		const DecimationInfo* d = blob.at<DecimationInfo>(di_off);
		const uint8_t* wtc2 = blob.at<uint8_t>(d->off_weight_texel_count);
		const uint8_t* wt2 = blob.at<uint8_t>(d->off_weight_texels);
		std::vector<uint64_t> touches(W, 0), touches_hi(W, 0), touches_3(W, 0), touches_4(W, 0);   // texel sets as 4 x 64 bits (T <= 216)
		for (unsigned int w = 0; w < W; w++)
			for (unsigned int j = 0; j < wtc2[w]; j++)
			{
				unsigned int t = wt2[j * W + w];
				(t < 64 ? touches[w] : t < 128 ? touches_hi[w] : t < 192 ? touches_3[w] : touches_4[w]) |= 1ull << (t & 63);
			}
		for (unsigned int w = 0; w < W; w++)
		{
			unsigned int n = 0;
			for (unsigned int k = w + 1; k < W; k++)
			{
				if (!((touches[w] & touches[k]) | (touches_hi[w] & touches_hi[k]) | (touches_3[w] & touches_3[k]) | (touches_4[w] & touches_4[k]))) continue;
				if (n >= (unsigned)REALIGN_LATER_MAX - 1) { fits = false; break; }      // (the last entry stays the terminator)
				later[(size_t)w * REALIGN_LATER_MAX + n++] = (uint8_t)k;
			}
		}
	}
	// (no allocation here: callers hold pointers into the blob)
	DecimationInfo* d = blob.at<DecimationInfo>(di_off);
	memcpy(blob.at<uint8_t>(d->off_realign_later), later.data(), later.size());
	d->realign_speculative = (fits && nlev >= (unsigned)REALIGN_SPECULATIVE_MIN_LEVELS && W <= 64) ? 1u : 0u;
}

// ---------------------------------------------------------------------------------------------
// Partition function (ASTC spec C.2.21; ref: partition_tables.cpp:114-263)
// ---------------------------------------------------------------------------------------------
static uint32_t hash52(uint32_t p)
{
	p ^= p >> 15; p *= 0xEEDE0891u; p ^= p >> 5; p += p << 16;
	p ^= p >> 7; p ^= p >> 3; p ^= p << 6; p ^= p >> 17;
	return p;
}

static uint8_t select_partition(int seed, int x, int y, int z, int pcount, bool small_block)
{
	if (small_block) { x <<= 1; y <<= 1; z <<= 1; }
	seed += (pcount - 1) * 1024;
	uint32_t rnum = hash52((uint32_t)seed);

	uint8_t s[12];
	s[0] = rnum & 0xF;          s[1] = (rnum >> 4) & 0xF;   s[2] = (rnum >> 8) & 0xF;
	s[3] = (rnum >> 12) & 0xF;  s[4] = (rnum >> 16) & 0xF;  s[5] = (rnum >> 20) & 0xF;
	s[6] = (rnum >> 24) & 0xF;  s[7] = (rnum >> 28) & 0xF;  s[8] = (rnum >> 18) & 0xF;
	s[9] = (rnum >> 22) & 0xF;  s[10] = (rnum >> 26) & 0xF; s[11] = ((rnum >> 30) | (rnum << 2)) & 0xF;
	for (int i = 0; i < 12; i++) s[i] = (uint8_t)(s[i] * s[i]);

	int sh1, sh2;
	if (seed & 1) { sh1 = (seed & 2) ? 4 : 5; sh2 = (pcount == 3) ? 6 : 5; }
	else          { sh1 = (pcount == 3) ? 6 : 5; sh2 = (seed & 2) ? 4 : 5; }
	int sh3 = (seed & 0x10) ? sh1 : sh2;

	s[0] >>= sh1; s[1] >>= sh2; s[2] >>= sh1; s[3] >>= sh2;
	s[4] >>= sh1; s[5] >>= sh2; s[6] >>= sh1; s[7] >>= sh2;
	s[8] >>= sh3; s[9] >>= sh3; s[10] >>= sh3; s[11] >>= sh3;

	int a = s[0] * x + s[1] * y + s[10] * z + (rnum >> 14);
	int b = s[2] * x + s[3] * y + s[11] * z + (rnum >> 10);
	int c = s[4] * x + s[5] * y + s[8] * z + (rnum >> 6);
	int d = s[6] * x + s[7] * y + s[9] * z + (rnum >> 2);
	a &= 0x3F; b &= 0x3F; c &= 0x3F; d &= 0x3F;
	if (pcount <= 3) d = 0;
	if (pcount <= 2) c = 0;
	if (pcount <= 1) b = 0;

	if (a >= b && a >= c && a >= d) return 0;
	if (b >= c && b >= d) return 1;
	if (c >= d) return 2;
	return 3;
}

struct PartTmp {
	uint8_t of_texel[MAX_TEXELS];
	uint8_t counts[4];
	unsigned int pcount;
};

static void gen_partition(unsigned int tx, unsigned int ty, unsigned int tz, unsigned int pcount, unsigned int seed, PartTmp& p)
{
	bool small_block = (tx * ty * tz) < 32;
	memset(p.counts, 0, sizeof(p.counts));
	unsigned int t = 0;
	for (unsigned int z = 0; z < tz; z++)
	for (unsigned int y = 0; y < ty; y++)
		for (unsigned int x = 0; x < tx; x++)
		{
			uint8_t part = select_partition((int)seed, (int)x, (int)y, (int)z, (int)pcount, small_block);
			p.of_texel[t++] = part;
			p.counts[part]++;
		}
	if (p.counts[0] == 0) p.pcount = 0;
	else if (p.counts[1] == 0) p.pcount = 1;
	else if (p.counts[2] == 0) p.pcount = 2;
	else if (p.counts[3] == 0) p.pcount = 3;
	else p.pcount = 4;
}

/* Canonical form: partitions renumbered in order of first appearance. (ref: partition_tables.cpp:38) */
static void canonical_pattern(unsigned int T, const uint8_t* of_texel, uint64_t pat[7])
{
	for (int i = 0; i < 7; i++) pat[i] = 0;
	int map[4] = { -1, -1, -1, -1 };
	int next = 0;
	for (unsigned int i = 0; i < T; i++)
	{
		int idx = of_texel[i];
		if (map[idx] < 0) map[idx] = next++;
		pat[i >> 5] |= (uint64_t)map[idx] << (2 * (i & 0x1F));
	}
}

// ---------------------------------------------------------------------------------------------
// xoroshiro128+ with the reference's fixed seed (ref: mathlib.cpp:26-48) -- only used to pick the
// 64-texel k-means subset of footprints above 64 texels (ref: block_sizes.cpp:717-754).
// ---------------------------------------------------------------------------------------------
static inline uint64_t rotl64(uint64_t v, int c) { return (v << c) | (v >> (64 - c)); }
static uint64_t ref_rand(uint64_t st[2])
{
	uint64_t s0 = st[0], s1 = st[1], res = s0 + s1;
	s1 ^= s0;
	st[0] = rotl64(s0, 24) ^ s1 ^ (s1 << 16);
	st[1] = rotl64(s1, 37);
	return res;
}

// ---------------------------------------------------------------------------------------------
// Quantization tables from the spec's unquantization rules
// ---------------------------------------------------------------------------------------------

/* Colour unquantization of one (trit/quint D, bits) symbol. (ASTC spec C.2.13, table C.2.16) */
static unsigned int color_unquant_symbol(unsigned int q, unsigned int hi, unsigned int lo)
{
	unsigned int bits = btq_counts[q].bits;
	if (!btq_counts[q].trits && !btq_counts[q].quints)
	{
		// bit replication to 8 bits
		unsigned int v = lo << (8 - bits);
		int rem = 8 - (int)bits;
		while (rem > 0)
		{
			int shift = rem - (int)bits;
			v |= shift > 0 ? lo << shift : lo >> -shift;
			rem -= (int)bits;
		}
		return v & 0xFF;
	}

	unsigned int a = lo & 1, b = (lo >> 1) & 1, c = (lo >> 2) & 1, d = (lo >> 3) & 1, e = (lo >> 4) & 1, f = (lo >> 5) & 1;
	unsigned int A = a ? 0x1FF : 0, B = 0, C = 0;
	if (btq_counts[q].trits)
	{
		switch (bits)
		{
		case 1: C = 204; B = 0; break;
		case 2: C = 93;  B = (b << 8) | (b << 4) | (b << 2) | (b << 1); break;
		case 3: C = 44;  B = (c << 8) | (b << 7) | (c << 3) | (b << 2) | (c << 1) | b; break;
		case 4: C = 22;  B = (d << 8) | (c << 7) | (b << 6) | (d << 2) | (c << 1) | b; break;
		case 5: C = 11;  B = (e << 8) | (d << 7) | (c << 6) | (b << 5) | (e << 1) | d; break;
		default: C = 5;  B = (f << 8) | (e << 7) | (d << 6) | (c << 5) | (b << 4) | f; break;
		}
	}
	else
	{
		switch (bits)
		{
		case 1: C = 113; B = 0; break;
		case 2: C = 54;  B = (b << 8) | (b << 3) | (b << 2); break;
		case 3: C = 26;  B = (c << 8) | (b << 7) | (c << 2) | (b << 1) | c; break;
		case 4: C = 13;  B = (d << 8) | (c << 7) | (b << 6) | (d << 1) | c; break;
		default: C = 6;  B = (e << 8) | (d << 7) | (c << 6) | (b << 5) | e; break;
		}
	}
	unsigned int T = hi * C + B;
	T ^= A;
	return (A & 0x80) | (T >> 2);
}

/* Weight unquantization of one symbol to 0..64. (ASTC spec C.2.17, table C.2.19) */
static unsigned int weight_unquant_symbol(unsigned int q, unsigned int hi, unsigned int lo)
{
	unsigned int bits = btq_counts[q].bits;
	unsigned int r;
	if (!btq_counts[q].trits && !btq_counts[q].quints)
	{
		switch (bits)
		{
		case 1: r = lo ? 63 : 0; break;
		case 2: r = lo | (lo << 2) | (lo << 4); break;
		case 3: r = lo | (lo << 3); break;
		case 4: r = (lo >> 2) | (lo << 2); break;
		default: r = (lo >> 4) | (lo << 1); break;
		}
	}
	else if (bits == 0)
	{
		static const unsigned int t3[3] = { 0, 32, 63 };
		static const unsigned int t5[5] = { 0, 16, 32, 47, 63 };
		r = btq_counts[q].trits ? t3[hi] : t5[hi];
	}
	else
	{
		unsigned int a = lo & 1, b = (lo >> 1) & 1, c = (lo >> 2) & 1;
		unsigned int A = a ? 0x7F : 0, B = 0, C = 0;
		if (btq_counts[q].trits)
		{
			switch (bits)
			{
			case 1: C = 50; B = 0; break;
			case 2: C = 23; B = (b << 6) | (b << 2) | b; break;
			default: C = 11; B = (c << 6) | (b << 5) | (c << 1) | b; break;
			}
		}
		else
		{
			switch (bits)
			{
			case 1: C = 28; B = 0; break;
			default: C = 13; B = (b << 6) | (b << 1); break;
			}
		}
		unsigned int T = hi * C + B;
		T ^= A;
		r = (A & 0x20) | (T >> 2);
	}
	if (r > 32) r++;
	return r;
}

static void build_color_quant_tables(uint8_t* unq_to_uq /*[17][512]*/, uint8_t* uq_to_pq /*[17][256]*/)
{
	for (unsigned int qi = 0; qi < 17; qi++)
	{
		unsigned int q = qi + QUANT_6;
		unsigned int bits = btq_counts[q].bits;
		unsigned int nhi = btq_counts[q].trits ? 3 : btq_counts[q].quints ? 5 : 1;
		bool valid[256] = { false };
		uint8_t* pq = uq_to_pq + qi * 256;
		memset(pq, 0, 256);
		for (unsigned int hi = 0; hi < nhi; hi++)
			for (unsigned int lo = 0; lo < (1u << bits); lo++)
			{
				unsigned int v = color_unquant_symbol(q, hi, lo);
				valid[v] = true;
				pq[v] = (uint8_t)((hi << bits) | lo);   // the BISE symbol value
			}

		// nearest representable value; exact ties resolved down (even slot) or up (odd slot)
		for (unsigned int i = 0; i < 256; i++)
		{
			unsigned int best = 256, lo_v = 256, hi_v = 0;
			for (unsigned int v = 0; v < 256; v++)
			{
				if (!valid[v]) continue;
				unsigned int dist = i > v ? i - v : v - i;
				if (dist < best) { best = dist; lo_v = hi_v = v; }
				else if (dist == best) { lo_v = std::min(lo_v, v); hi_v = std::max(hi_v, v); }
			}
			unq_to_uq[qi * 512 + 2 * i] = (uint8_t)lo_v;
			unq_to_uq[qi * 512 + 2 * i + 1] = (uint8_t)hi_v;
		}
	}
}

static void build_weight_quant_tables(QuantXfer* qx /*[12]*/)
{
	for (unsigned int q = 0; q < 12; q++)
	{
		unsigned int bits = btq_counts[q].bits;
		unsigned int nhi = btq_counts[q].trits ? 3 : btq_counts[q].quints ? 5 : 1;
		unsigned int n = quant_levels[q];
		std::vector<std::pair<unsigned, unsigned>> vals; // (unquant value, symbol)
		for (unsigned int hi = 0; hi < nhi; hi++)
			for (unsigned int lo = 0; lo < (1u << bits); lo++)
				vals.push_back({ weight_unquant_symbol(q, hi, lo), (hi << bits) | lo });
		std::sort(vals.begin(), vals.end());

		memset(&qx[q], 0, sizeof(QuantXfer));
		for (unsigned int i = 0; i < n; i++)
		{
			qx[q].quant_to_unquant[i] = (uint8_t)vals[i].first;
			qx[q].scramble_map[i] = (uint8_t)vals[i].second;
			unsigned int prev = vals[i == 0 ? 0 : i - 1].first;
			unsigned int next = vals[i == n - 1 ? n - 1 : i + 1].first;
			qx[q].prev_next_values[vals[i].first] = (uint16_t)((next << 8) | prev);
		}
	}
}

/* quant_mode_table[pairs][bits] = best colour quant level whose BISE size for 2*pairs integers
 * fits in `bits`, or -1. (ref data: quantization.cpp:802; rule: astc_quant_generator / spec C.2.22) */
static void build_quant_mode_table(int8_t* tab /*[10][128]*/)
{
	for (unsigned int i = 0; i < 10; i++)
		for (unsigned int j = 0; j < 128; j++)
		{
			int best = -1;
			if (i > 0)
			{
				for (int q = 0; q < 21; q++)
					if (ise_sequence_bitcount(2 * i, (unsigned)q) <= j) best = q;
			}
			tab[i * 128 + j] = (int8_t)best;
		}
}

/* Inverse of the spec's trit/quint block decode; on duplicates the highest packed value wins
 * (that is the choice baked into the reference's data tables, integer_sequence.cpp:64-330). */
static void build_trit_quint_tables(uint8_t* trits /*[243]*/, uint8_t* quints /*[125]*/)
{
	for (unsigned int T = 0; T < 256; T++)
	{
		unsigned int t0, t1, t2, t3, t4, C;
		if (((T >> 2) & 7) == 7)
		{
			C = ((T >> 5) << 2) | (T & 3);
			t4 = t3 = 2;
		}
		else
		{
			C = T & 0x1F;
			if (((T >> 5) & 3) == 3) { t4 = 2; t3 = (T >> 7) & 1; }
			else { t4 = (T >> 7) & 1; t3 = (T >> 5) & 3; }
		}
		if ((C & 3) == 3)
		{
			t2 = 2; t1 = (C >> 4) & 1;
			t0 = (((C >> 3) & 1) << 1) | (((C >> 2) & 1) & ~((C >> 3) & 1) & 1);
		}
		else if (((C >> 2) & 3) == 3)
		{
			t2 = 2; t1 = 2; t0 = C & 3;
		}
		else
		{
			t2 = (C >> 4) & 1; t1 = (C >> 2) & 3;
			t0 = (((C >> 1) & 1) << 1) | ((C & 1) & ~((C >> 1) & 1) & 1);
		}
		unsigned int idx = (((t4 * 3 + t3) * 3 + t2) * 3 + t1) * 3 + t0;
		trits[idx] = (uint8_t)T;
	}
	for (unsigned int Q = 0; Q < 128; Q++)
	{
		unsigned int q0, q1, q2;
		if (((Q >> 1) & 3) == 3 && ((Q >> 5) & 3) == 0)
		{
			unsigned int b0 = Q & 1;
			q2 = (b0 << 2) | ((((Q >> 4) & 1) & ~b0 & 1) << 1) | (((Q >> 3) & 1) & ~b0 & 1);
			q1 = q0 = 4;
		}
		else
		{
			unsigned int C;
			if (((Q >> 1) & 3) == 3)
			{
				q2 = 4;
				C = (((Q >> 3) & 3) << 3) | ((~(Q >> 5) & 3) << 1) | (Q & 1);
			}
			else
			{
				q2 = (Q >> 5) & 3;
				C = Q & 0x1F;
			}
			if ((C & 7) == 5) { q1 = 4; q0 = (C >> 3) & 3; }
			else { q1 = (C >> 3) & 3; q0 = C & 7; }
		}
		unsigned int idx = (q2 * 5 + q1) * 5 + q0;
		quints[idx] = (uint8_t)Q;
	}
}

// ---------------------------------------------------------------------------------------------
// Legal footprints (ASTC spec table C.2.7; ref: percentile_tables.cpp:1203-1229)
// ---------------------------------------------------------------------------------------------
bool is_legal_2d_block_size(unsigned int x, unsigned int y)
{
	static const uint8_t legal[14][2] = {
		{4,4},{5,4},{5,5},{6,5},{6,6},{8,5},{8,6},{8,8},{10,5},{10,6},{10,8},{10,10},{12,10},{12,12}
	};
	for (auto& l : legal) if (l[0] == x && l[1] == y) return true;
	return false;
}

bool is_legal_3d_block_size(unsigned int x, unsigned int y, unsigned int z)
{
	static const uint8_t legal[10][3] = {
		{3,3,3},{4,3,3},{4,4,3},{4,4,4},{5,4,4},{5,5,4},{5,5,5},{6,5,5},{6,6,5},{6,6,6}
	};
	for (auto& l : legal) if (l[0] == x && l[1] == y && l[2] == z) return true;
	return false;
}

// ---------------------------------------------------------------------------------------------
// The builder
// ---------------------------------------------------------------------------------------------
bool build_tables(unsigned int tx, unsigned int ty, unsigned int tz, unsigned int partition_count_cutoff,
                  float mode_cutoff, std::vector<uint8_t>& out, HostTables& host)
{
	if (tz < 1) tz = 1;
	const bool is_3d = tz > 1;
	const unsigned int T = tx * ty * tz;
	Blob blob;
	uint32_t root_off = blob.alloc(sizeof(TableRoot));
	(void)root_off;

	// ---- percentile table for this footprint (ref: percentile_tables.cpp:1165) ----
	std::vector<float> percentiles(MAX_BLOCK_MODES, 1.0f);
	if (!is_3d)
	{
		const uint16_t* pt = nullptr;
		for (auto* t : pct_tables) if (t[0] == tx && t[1] == ty) pt = t;
		if (!pt) return false;
		const uint16_t* item = pt + 6;
		for (int p = 0; p < 2; p++)
		{
			float div = (float)pt[2 + p];
			for (unsigned int j = 0; j < pt[4 + p]; j++, item += 2)
				percentiles[item[0]] = (float)item[1] / div;
		}
	}

	// ---- block modes in search order: always / 1-plane selected / 2-plane selected ----
	// (ref: block_sizes.cpp:865-988; the 4th "unselected" pass is never searched and is omitted)
	std::vector<BlockMode> bms;
	std::vector<DecimationMode> dms;
	struct Grid { unsigned x, y, z; unsigned count() const { return x * y * z; } };
	std::vector<Grid> dm_grid;
	int dm_index[8 * 16 * 16];
	for (int& v : dm_index) v = -1;
	std::vector<bool> taken(MAX_BLOCK_MODES, false);
	unsigned int bm_counts[3] = { 0, 0, 0 }, dm_counts[3] = { 0, 0, 0 };

	auto new_decimation_mode = [&](unsigned int wx, unsigned int wy, unsigned int wz) {
		int dm = (int)dms.size();
		dm_index[(wz * 16 + wy) * 16 + wx] = dm;
		unsigned int wc = wx * wy * wz;
		int mp1 = -1, mp2 = -1;
		for (int q = 0; q < 12; q++)
		{
			unsigned int b1 = ise_sequence_bitcount(wc, (unsigned)q);
			if (b1 >= 24 && b1 <= 96) mp1 = q;
			if (2 * wc <= (unsigned)MAX_WEIGHTS)
			{
				unsigned int b2 = ise_sequence_bitcount(2 * wc, (unsigned)q);
				if (b2 >= 24 && b2 <= 96) mp2 = q;
			}
		}
		DecimationMode d = { (int8_t)mp1, (int8_t)mp2, 0, 0, {0, 0}, {0, 0} };
		dms.push_back(d);
		dm_grid.push_back({ wx, wy, wz });
		return dm;
	};

	if (is_3d)
	{
		// the reference numbers the grids of a 3D footprint up front, in this order, used or not
		// (block_sizes.cpp:1052-1100); grids no block mode refers to stay out of every search list here
		for (unsigned int wx = 2; wx <= tx; wx++)
			for (unsigned int wy = 2; wy <= ty; wy++)
				for (unsigned int wz = 2; wz <= tz; wz++)
					if (wx * wy * wz <= (unsigned)MAX_WEIGHTS) { new_decimation_mode(wx, wy, wz); dm_counts[1]++; }
	}

	for (unsigned int pass = 0; pass < 3; pass++)
	{
		for (unsigned int i = 0; i < (unsigned)MAX_BLOCK_MODES; i++)
		{
			if (taken[i]) continue;
			unsigned int wx, wy, wz = 1, quant, wbits; bool dual;
			bool valid = is_3d ? decode_block_mode_3d(i, wx, wy, wz, dual, quant, wbits)
			                   : decode_block_mode_2d(i, wx, wy, dual, quant, wbits);
			if (!valid || wx > tx || wy > ty || wz > tz) continue;
			if ((pass <= 1 && dual) || (pass == 2 && !dual)) continue;
			if ((dual ? 109 : 111) <= (int)wbits) continue;
			// 3D footprints have no percentile data: every legal mode is "selected", none "always"
			// (ref: construct_block_size_descriptor_3d, block_sizes.cpp:1025-1190)
			bool hit = is_3d ? pass != 0 : percentiles[i] <= (pass == 0 ? 0.0f : mode_cutoff);
			if (!hit) continue;

			int dm = dm_index[(wz * 16 + wy) * 16 + wx];
			if (dm < 0)
			{
				dm = new_decimation_mode(wx, wy, wz);
				dm_counts[pass]++;
			}

			BlockMode bm = { (uint16_t)i, (uint8_t)dm, (uint8_t)quant, (uint8_t)wbits, (uint8_t)dual, {0, 0} };
			bms.push_back(bm);
			if (dual) dms[dm].refprec_2planes |= (uint16_t)(1u << quant);
			else      dms[dm].refprec_1plane  |= (uint16_t)(1u << quant);
			taken[i] = true;
			bm_counts[pass]++;
		}
	}
	if ((bm_counts[0] == 0 && !is_3d) || dms.empty()) return false;

	// packed LDS slots for the per-trial ideal weights and angular bounds of each grid, one dense
	// packing per trial class (see DecimationMode).  Grids are packed by ascending lowest quant level
	// of the block modes that use them, so that a trial limited to quant level q touches a prefix.
	uint32_t dwi_total[2] = { 0, 0 }, lh_total[2] = { 0, 0 };
	std::vector<uint32_t> pack_order[2];          // decimation modes of each class in packing order
	auto lowest_bit = [](uint16_t v) { int b = 0; while (!((v >> b) & 1)) b++; return b; };
	for (int cls = 0; cls < 2; cls++)
	{
		for (size_t i = 0; i < dms.size(); i++)
		{
			if ((cls == 0 ? dms[i].refprec_1plane : dms[i].refprec_2planes) != 0) pack_order[cls].push_back((uint32_t)i);
			for (int plane = 0; plane <= cls; plane++) { dms[i].dwi_offset[cls + plane] = 0; dms[i].lowhigh_offset[cls + plane] = 0; }
		}
		std::stable_sort(pack_order[cls].begin(), pack_order[cls].end(), [&](uint32_t a, uint32_t b) {
			return lowest_bit(cls == 0 ? dms[a].refprec_1plane : dms[a].refprec_2planes) < lowest_bit(cls == 0 ? dms[b].refprec_1plane : dms[b].refprec_2planes);
		});
		for (uint32_t i : pack_order[cls])
		{
			uint32_t wc4 = (dm_grid[i].count() + 3u) & ~3u;
			int maxprec = cls == 0 ? dms[i].maxprec_1plane : dms[i].maxprec_2planes;
			for (int plane = 0; plane <= cls; plane++)
			{
				dms[i].dwi_offset[cls + plane] = (uint16_t)dwi_total[cls];
				dms[i].lowhigh_offset[cls + plane] = (uint16_t)lh_total[cls];
				dwi_total[cls] += wc4;
				// one (low, high) pair per angular quant level (<= QUANT_12) that a block mode of this grid uses in this class
				// (at least one: the angular search parks the set's weight range there); the pair of level q sits at the rank
				// of q among the used levels (mode_weight_bounds, angular_endpoints)
				(void)maxprec;
				const uint32_t used = (cls == 0 ? dms[i].refprec_1plane : dms[i].refprec_2planes) & 0xFFu;
				lh_total[cls] += 2u * std::max<uint32_t>(1u, (uint32_t)__builtin_popcount(used));
			}
		}
	}

	uint32_t off_bm = blob.alloc(bms.size() * sizeof(BlockMode));
	memcpy(blob.at<uint8_t>(off_bm), bms.data(), bms.size() * sizeof(BlockMode));
	uint32_t off_dm = blob.alloc(dms.size() * sizeof(DecimationMode));
	memcpy(blob.at<uint8_t>(off_dm), dms.data(), dms.size() * sizeof(DecimationMode));
	uint32_t off_di = blob.alloc(dms.size() * sizeof(DecimationInfo));
	for (size_t i = 0; i < dms.size(); i++)
		build_decimation_info(blob, (uint32_t)(off_di + i * sizeof(DecimationInfo)), tx, ty, tz, dm_grid[i].x, dm_grid[i].y, dm_grid[i].z);

	// ---- k-means texel subset (ref: block_sizes.cpp:717-754) ----
	unsigned int kcount = std::min<unsigned>(T, MAX_KMEANS_TEXELS);
	uint32_t off_km = blob.alloc(MAX_KMEANS_TEXELS);
	{
		uint8_t* km = blob.at<uint8_t>(off_km);
		if (T <= (unsigned)MAX_KMEANS_TEXELS)
		{
			for (unsigned int i = 0; i < T; i++) km[i] = (uint8_t)i;
		}
		else
		{
			uint64_t st[2] = { 0xfaf9e171cea1ec6bULL, 0xf1b318cc06af5d71ULL };
			std::vector<bool> seen(T, false);
			unsigned int n = 0;
			while (n < (unsigned)MAX_KMEANS_TEXELS)
			{
				uint8_t texel = (uint8_t)ref_rand(st);
				texel = (uint8_t)(texel % T);
				if (!seen[texel]) { km[n++] = texel; seen[texel] = true; }
			}
		}
	}
	std::vector<uint8_t> kmeans_texels(blob.at<uint8_t>(off_km), blob.at<uint8_t>(off_km) + kcount);

	// ---- partition tables (ref: partition_tables.cpp:389-497), selected (deduplicated) entries only ----
	const uint32_t pstride = (uint32_t)((sizeof(PartitionHeader) + 2 * T + 3) & ~3u);
	uint32_t off_part[4] = { 0, 0, 0, 0 }, off_cov[4] = { 0, 0, 0, 0 }, pcounts[4] = { 1, 0, 0, 0 };
	host.partition_packed_index.assign(3 * MAX_PARTITIONINGS, 0xFFFF);

	auto write_partition = [&](uint32_t rec, const PartTmp& p, unsigned int seed) {
		PartitionHeader* h = blob.at<PartitionHeader>(rec);
		h->partition_index = (uint16_t)seed;
		h->partition_count = (uint8_t)p.pcount;
		for (int i = 0; i < 4; i++) h->texel_count[i] = p.counts[i];
		uint8_t* pot = blob.at<uint8_t>(rec + sizeof(PartitionHeader));
		uint8_t* sorted = pot + T;
		memcpy(pot, p.of_texel, T);
		unsigned int n = 0;
		for (unsigned int part = 0; part < 4; part++)
			for (unsigned int t = 0; t < T; t++)
				if (p.of_texel[t] == part) sorted[n++] = (uint8_t)t;
	};

	{
		PartTmp p;
		gen_partition(tx, ty, tz, 1, 0, p);
		off_part[0] = blob.alloc(pstride);
		write_partition(off_part[0], p, 0);
	}

	for (unsigned int pc = 2; pc <= 4; pc++)
	{
		std::vector<PartTmp> kept;
		std::vector<unsigned int> kept_seed;
		std::vector<uint64_t> pats;
		// The reference only drops table entries above the partition count limit when the
		// context is SELF_DECOMPRESS_ONLY; the encoder never reads them either way.
		if (pc <= partition_count_cutoff)
		{
			for (unsigned int seed = 0; seed < (unsigned)MAX_PARTITIONINGS; seed++)
			{
				PartTmp p;
				gen_partition(tx, ty, tz, pc, seed, p);
				if (p.pcount != pc) continue;
				uint64_t pat[7];
				canonical_pattern(T, p.of_texel, pat);
				bool dup = false;
				for (size_t j = 0; j < kept.size() && !dup; j++)
					dup = memcmp(&pats[j * 7], pat, sizeof(pat)) == 0;
				if (dup) continue;
				host.partition_packed_index[(pc - 2) * MAX_PARTITIONINGS + seed] = (uint16_t)kept.size();
				kept.push_back(p);
				kept_seed.push_back(seed);
				pats.insert(pats.end(), pat, pat + 7);
			}
		}
		pcounts[pc - 1] = (uint32_t)kept.size();
		off_part[pc - 1] = blob.alloc(std::max<size_t>(kept.size(), 1) * pstride);
		off_cov[pc - 1] = blob.alloc(std::max<size_t>(kept.size(), 1) * pc * sizeof(uint64_t));
		for (size_t i = 0; i < kept.size(); i++)
		{
			write_partition((uint32_t)(off_part[pc - 1] + i * pstride), kept[i], kept_seed[i]);
			uint64_t* cov = blob.at<uint64_t>((uint32_t)(off_cov[pc - 1] + i * pc * sizeof(uint64_t)));
			for (unsigned int k = 0; k < kcount; k++)
				cov[kept[i].of_texel[kmeans_texels[k]]] |= 1ULL << k;
		}
	}

	// owner of every packed ideal-weight slot
	uint32_t off_owner[2];
	for (int cls = 0; cls < 2; cls++)
	{
		off_owner[cls] = blob.alloc(std::max<uint32_t>(dwi_total[cls], 1) * sizeof(uint16_t));
		uint16_t* own = blob.at<uint16_t>(off_owner[cls]);
		for (uint32_t i : pack_order[cls])
		{
			uint32_t wc4 = (dm_grid[i].count() + 3u) & ~3u;
			for (int plane = 0; plane <= cls; plane++)
				for (uint32_t k = 0; k < wc4; k++) own[dms[i].dwi_offset[cls + plane] + k] = (uint16_t)((i << 1) | (unsigned)plane);
		}
	}

	// the taps of every weight as packed DwiTap8 groups (see DwiSlot::wt_off): two bytes per tap -- texel index, integer
	// contribution 0 .. 16 -- eight taps per 16-byte group, the last group padded with (texel 0, contribution 0)
	std::vector<uint32_t> weight_taps_off(dms.size(), 0), weight_taps_stride(dms.size(), 0);
	for (size_t i = 0; i < dms.size(); i++)
	{
		const DecimationInfo di = *blob.at<DecimationInfo>((uint32_t)(off_di + i * sizeof(DecimationInfo)));
		const uint32_t W = di.weight_count, rows = di.max_weight_texel_count, rows8 = (rows + 7u) & ~7u;
		const uint32_t off = blob.alloc((size_t)std::max<uint32_t>(W * rows8, 8u) * 2u, 16);
		for (uint32_t w = 0; w < W; w++)
			for (uint32_t j = 0; j < rows8; j++)
			{
				uint8_t* tap = blob.at<uint8_t>((uint32_t)(off + (w * rows8 + j) * 2u));
				const float contrib = j < rows ? *blob.at<float>((uint32_t)(di.off_weight_contribs + (j * W + w) * sizeof(float))) : 0.0f;
				tap[0] = j < rows && contrib != 0.0f ? *blob.at<uint8_t>((uint32_t)(di.off_weight_texels + j * W + w)) : (uint8_t)0;
				tap[1] = (uint8_t)(int)contrib;      // (integer valued: ref block_sizes.cpp weights_texel_contribs)
				if ((float)tap[1] != contrib) abort();
			}
		weight_taps_off[i] = off;
		weight_taps_stride[i] = rows8 * 2u;
	}

	// per-slot / per-set records of the decimation sweeps (see DwiSlot, InfillSet), in packing order
	uint32_t off_slots[2], off_isets[2], n_sets[2], used_sets[2][12];
	for (int cls = 0; cls < 2; cls++)
	{
		const int planes = cls + 1;
		n_sets[cls] = (uint32_t)pack_order[cls].size() * planes;
		off_slots[cls] = blob.alloc(std::max<uint32_t>(dwi_total[cls], 1) * sizeof(DwiSlot));
		off_isets[cls] = blob.alloc(std::max<uint32_t>(n_sets[cls], 1) * sizeof(InfillSet));
		for (int q = 0; q < 12; q++) used_sets[cls][q] = 0;
		uint32_t set = 0;
		for (uint32_t i : pack_order[cls])
		{
			const DecimationInfo di = *blob.at<DecimationInfo>((uint32_t)(off_di + i * sizeof(DecimationInfo)));
			const uint16_t refprec = cls == 0 ? dms[i].refprec_1plane : dms[i].refprec_2planes;
			const uint32_t wc4 = ((uint32_t)di.weight_count + 3u) & ~3u;
			const uint8_t* wtc = blob.at<uint8_t>(di.off_weight_texel_count);
			for (int q = lowest_bit(refprec); q < 12; q++) used_sets[cls][q] += planes;
			for (int plane = 0; plane < planes; plane++, set++)
			{
				InfillSet* is = blob.at<InfillSet>((uint32_t)(off_isets[cls] + set * sizeof(InfillSet)));
				is->tw_off = di.off_texel_weights;
				is->tcf_off = di.off_texel_contribs_f;
				is->dwi_offset = dms[i].dwi_offset[cls + plane];
				is->refprec = refprec;
				is->taps = (uint8_t)(di.max_texel_weight_count > 2 ? 4 : di.max_texel_weight_count > 1 ? 2 : 1);
				is->direct = di.texel_count == di.weight_count;
				is->dm = (uint8_t)i;
				is->plane = (uint8_t)plane;
				for (uint32_t k = 0; k < wc4; k++)
				{
					DwiSlot* sl = blob.at<DwiSlot>((uint32_t)(off_slots[cls] + (dms[i].dwi_offset[cls + plane] + k) * sizeof(DwiSlot)));
					sl->wt_off = weight_taps_off[i] + (k < di.weight_count ? k : 0u) * weight_taps_stride[i];
					sl->wc_off = 0;
					sl->refprec = refprec;
					sl->weight_count = di.weight_count;
					sl->taps = k < di.weight_count ? wtc[k] : 0;
					sl->flags = (uint8_t)((di.texel_count == di.weight_count ? 1 : 0) | (plane << 1));
					sl->dm = (uint8_t)i;
					sl->set = (uint8_t)set;
					sl->index = (uint8_t)k;
				}
			}
		}
	}

	// processing order of the decimation sweeps (see DwiOrderDir): per trial class and weight quant limit, the live
	// slots of the used sets, chunked like the kernel chunks the infill, longest tap lists first inside a chunk
	uint32_t off_order[2] = { 0, 0 };
	const uint32_t texel_count_all = tx * ty * tz;
	uint32_t sets_per_chunk = (uni_region_bytes(texel_count_all, partition_count_cutoff) / 4u) / ((texel_count_all + 3u) & ~3u);
	if (sets_per_chunk < 1) sets_per_chunk = 1;
	for (int cls = 0; cls < 2; cls++)
	{
		off_order[cls] = blob.alloc(12 * sizeof(DwiOrderDir));
		for (int q = 0; q < 12; q++)
		{
			const uint32_t nsets = used_sets[cls][q];
			std::vector<uint16_t> list;
			DwiOrderDir dir;
			memset(&dir, 0, sizeof(dir));
			uint32_t chunks = 0;
			for (uint32_t p0 = 0; p0 < nsets && chunks < (uint32_t)DWI_MAX_CHUNKS; p0 += sets_per_chunk, chunks++)
			{
				const uint32_t p1 = std::min(p0 + sets_per_chunk, nsets);
				std::vector<uint16_t> part;
				for (uint32_t k = 0; k < dwi_total[cls]; k++)
				{
					const DwiSlot* sl = blob.at<DwiSlot>((uint32_t)(off_slots[cls] + k * sizeof(DwiSlot)));
					if (sl->taps != 0 && sl->set >= p0 && sl->set < p1) part.push_back((uint16_t)k);
				}
				std::stable_sort(part.begin(), part.end(), [&](uint16_t a, uint16_t b) {
					const DwiSlot* sa = blob.at<DwiSlot>((uint32_t)(off_slots[cls] + a * sizeof(DwiSlot)));
					const DwiSlot* sb = blob.at<DwiSlot>((uint32_t)(off_slots[cls] + b * sizeof(DwiSlot)));
					const int ta = (sa->flags & 1) ? 0 : sa->taps, tb = (sb->flags & 1) ? 0 : sb->taps;
					return ta > tb;
				});
				dir.chunk_start[chunks] = (uint16_t)list.size();
				list.insert(list.end(), part.begin(), part.end());
			}
			// (more sets than DWI_MAX_CHUNKS chunks can hold: chunks = 0 tells the kernel to use the unsorted sweeps)
			if (chunks * sets_per_chunk < nsets) { chunks = 0; list.clear(); }
			dir.chunk_start[chunks] = (uint16_t)list.size();
			dir.chunks = (uint16_t)chunks;
			// the list holds the slots' records themselves, in processing order, with the slot's packed index in the place
			// of the quant-level mask (which only the unsorted sweeps look at): a sweep iteration then starts with ONE
			// coalesced 16-byte load per lane instead of an index load and a dependent gather of the record
			dir.list_off = blob.alloc(std::max<size_t>(list.size(), 1) * sizeof(DwiSlot), 16);
			for (size_t n = 0; n < list.size(); n++)
			{
				DwiSlot rec = *blob.at<DwiSlot>((uint32_t)(off_slots[cls] + list[n] * sizeof(DwiSlot)));
				rec.refprec = list[n];
				*blob.at<DwiSlot>((uint32_t)(dir.list_off + n * sizeof(DwiSlot))) = rec;
			}
			*blob.at<DwiOrderDir>((uint32_t)(off_order[cls] + q * sizeof(DwiOrderDir))) = dir;
		}
	}

	// ---- static tables ----
	uint32_t off_cq = blob.alloc(17 * 512);
	uint32_t off_cp = blob.alloc(17 * 256);
	build_color_quant_tables(blob.at<uint8_t>(off_cq), blob.at<uint8_t>(off_cp));
	uint32_t off_qx = blob.alloc(12 * sizeof(QuantXfer));
	build_weight_quant_tables(blob.at<QuantXfer>(off_qx));
	uint32_t off_qm = blob.alloc(10 * 128);
	build_quant_mode_table(blob.at<int8_t>(off_qm));
	// ... and transposed: the levels of all integer counts for one bit budget in one 16-byte row (best_combination_for_bitcount)
	uint32_t off_qm_bits = blob.alloc(128 * 16, 16);
	for (int bits = 0; bits < 128; bits++)
		for (int pairs = 0; pairs < 16; pairs++)
			*blob.at<int8_t>((uint32_t)(off_qm_bits + bits * 16 + pairs)) = pairs < 10 ? *blob.at<int8_t>((uint32_t)(off_qm + pairs * 128 + bits)) : (int8_t)-1;
	uint32_t off_tr = blob.alloc(243);
	uint32_t off_qu = blob.alloc(125);
	build_trit_quint_tables(blob.at<uint8_t>(off_tr), blob.at<uint8_t>(off_qu));

	uint32_t off_sin = blob.alloc(SINCOS_STEPS * ANGULAR_STEPS * sizeof(float));
	uint32_t off_cos = blob.alloc(SINCOS_STEPS * ANGULAR_STEPS * sizeof(float));
	{
		// ref: weight_align.cpp:72-84 -- host libm, float argument arithmetic
		float* s = blob.at<float>(off_sin);
		float* c = blob.at<float>(off_cos);
		const float pi = 3.14159265358979323846f;
		for (unsigned int i = 0; i < (unsigned)ANGULAR_STEPS; i++)
		{
			float angle_step = (float)(i + 1);
			for (unsigned int j = 0; j < (unsigned)SINCOS_STEPS; j++)
			{
				float arg = (2.0f * pi / (SINCOS_STEPS - 1.0f)) * angle_step * (float)j;
				s[j * ANGULAR_STEPS + i] = sinf(arg);
				c[j * ANGULAR_STEPS + i] = cosf(arg);
			}
		}
	}

	TableRoot* r = blob.at<TableRoot>(0);
	r->dim_x = (uint8_t)tx; r->dim_y = (uint8_t)ty; r->dim_z = (uint8_t)tz; r->texel_count = (uint8_t)T;
	r->block_mode_count_1plane_always = bm_counts[0];
	r->block_mode_count_1plane_selected = bm_counts[0] + bm_counts[1];
	r->block_mode_count_1plane_2plane_selected = bm_counts[0] + bm_counts[1] + bm_counts[2];
	r->decimation_mode_count_always = dm_counts[0];
	r->decimation_mode_count_selected = dm_counts[0] + dm_counts[1] + dm_counts[2];
	for (int i = 0; i < 4; i++)
	{
		r->partitioning_count_selected[i] = pcounts[i];
		r->off_partitions[i] = off_part[i];
		r->off_coverage[i] = off_cov[i];
	}
	r->partition_stride = pstride;
	r->off_block_modes = off_bm;
	r->off_decimation_modes = off_dm;
	r->off_decimation_infos = off_di;
	r->off_kmeans_texels = off_km;
	r->off_color_unquant_to_uquant = off_cq;
	r->off_color_uquant_to_pquant = off_cp;
	r->off_quant_xfer = off_qx;
	r->off_quant_mode_table = off_qm;
	r->off_quant_mode_by_bits = off_qm_bits;
	r->off_integer_of_trits = off_tr;
	r->off_integer_of_quints = off_qu;
	{
		// grids by descending weight count: the angular search batches (grid, step) pairs onto lanes that each walk
		// the grid's weights, so a batch of similar grids wastes the fewest iterations
		std::vector<uint8_t> by_w(dms.size());
		for (size_t i = 0; i < dms.size(); i++) by_w[i] = (uint8_t)i;
		std::stable_sort(by_w.begin(), by_w.end(), [&](uint8_t a, uint8_t b) { return dm_grid[a].count() > dm_grid[b].count(); });
		uint32_t off = blob.alloc(std::max<size_t>(by_w.size(), 1));
		memcpy(blob.at<uint8_t>(off), by_w.data(), by_w.size());
		r = blob.at<TableRoot>(0);
		r->off_dm_by_weights = off;
	}
	r->off_sin_table = off_sin;
	r->off_cos_table = off_cos;
	{
		const uint32_t n = (uint32_t)SINCOS_STEPS * (uint32_t)ANGULAR_STEPS;
		// (one more row, all +0.0: the angular search points the slots past a grid's last weight at it -- adding it leaves
		// the sums as they are)
		uint32_t off_cs = blob.alloc((size_t)(n + (uint32_t)ANGULAR_STEPS) * 2 * sizeof(float), 8);
		for (uint32_t i = 0; i < n; i++)
		{
			*blob.at<float>((uint32_t)(off_cs + (2 * i) * sizeof(float))) = *blob.at<float>((uint32_t)(off_cos + i * sizeof(float)));
			*blob.at<float>((uint32_t)(off_cs + (2 * i + 1) * sizeof(float))) = *blob.at<float>((uint32_t)(off_sin + i * sizeof(float)));
		}
		r = blob.at<TableRoot>(0);
		r->off_cos_sin_table = off_cs;
	}
	{
		// per block mode: everything its scoring reads from the mode / grid records (ModeStatic)
		uint32_t off_ms = blob.alloc(std::max<size_t>(bms.size(), 1) * sizeof(ModeStatic), 8);
		for (size_t i = 0; i < bms.size(); i++)
		{
			const BlockMode& bm = bms[i];
			const DecimationMode& dm = dms[bm.decimation_mode];
			const DecimationInfo di = *blob.at<DecimationInfo>((uint32_t)(off_di + bm.decimation_mode * sizeof(DecimationInfo)));
			ModeStatic ms;
			memset(&ms, 0, sizeof(ms));
			ms.tw_off = di.off_texel_weights;
			ms.tcf_off = di.off_texel_contribs_f;
			const int cls = bm.is_dual_plane ? 1 : 0;
			const uint32_t used = (cls ? dm.refprec_2planes : dm.refprec_1plane) & 0xFFu;
			for (int plane = 0; plane <= cls; plane++)
			{
				ms.dwi_off[plane] = dm.dwi_offset[cls + plane];
				ms.lh_off[plane] = bm.quant_mode <= MAX_ANGULAR_QUANT
				    ? (uint16_t)(dm.lowhigh_offset[cls + plane] + 2u * (uint32_t)__builtin_popcount(used & ((1u << bm.quant_mode) - 1u))) : (uint16_t)0xFFFF;
			}
			ms.taps = (uint8_t)(di.max_texel_weight_count > 2 ? 4 : di.max_texel_weight_count > 1 ? 2 : 1);
			ms.weights = di.weight_count;
			ms.quant_mode = bm.quant_mode;
			ms.weight_bits = bm.weight_bits;
			ms.is_dual_plane = bm.is_dual_plane;
			*blob.at<ModeStatic>((uint32_t)(off_ms + i * sizeof(ModeStatic))) = ms;
		}
		// ... and the colour quant levels its bit budget allows, per partition count (format selection: one load per mode
		// instead of block mode -> bit count -> table row)
		const size_t nm = std::max<size_t>(bms.size(), 1);
		uint32_t off_ml = blob.alloc(4 * nm * 16, 16);
		for (int pc = 1; pc <= 4; pc++)
			for (size_t i = 0; i < bms.size(); i++)
			{
				const int free_bits[4] = { 111, 97, 94, 91 };
				const int bits = (bms[i].is_dual_plane ? 109 : free_bits[pc - 1]) - (int)bms[i].weight_bits;      // (ref: compress_symbolic.cpp:434-453, :817)
				int8_t* row = blob.at<int8_t>((uint32_t)(off_ml + ((size_t)(pc - 1) * nm + i) * 16));
				for (int pairs = 0; pairs < 16; pairs++)
					row[pairs] = (bits > 0 && bits < 128) ? *blob.at<int8_t>((uint32_t)(off_qm_bits + bits * 16 + pairs)) : (int8_t)-1;
			}
		r = blob.at<TableRoot>(0);
		r->off_mode_static = off_ms;
		r->off_mode_levels = off_ml;
	}
	{
		auto used = [&](size_t i) { return dms[i].refprec_1plane != 0 || dms[i].refprec_2planes != 0; };
		uint32_t mx = 0;
		for (size_t i = 0; i < dms.size(); i++) if (used(i)) mx = std::max(mx, blob.at<DecimationInfo>((uint32_t)(off_di + i * sizeof(DecimationInfo)))->table_bytes);
		r->max_decimation_table_bytes = mx;
		uint32_t mr = 0;
		for (size_t i = 0; i < dms.size(); i++) if (used(i)) mr = std::max<uint32_t>(mr, blob.at<DecimationInfo>((uint32_t)(off_di + i * sizeof(DecimationInfo)))->max_weight_texel_count);
		r->max_weight_texel_rows = mr;
		{
			// weights per realign group, per grid: one lane per (weight, texel row) and 12 LDS rows of
			// that many floats per weight, inside the 16 texel-length rows the refit scratch has to spare
			uint32_t Tp = ((uint32_t)T + 3u) & ~3u;
			uint32_t rt_floats = 0;
			for (size_t i = 0; i < dms.size(); i++)
			{
				uint32_t di_off = (uint32_t)(off_di + i * sizeof(DecimationInfo));
				uint32_t rs = ((uint32_t)blob.at<DecimationInfo>(di_off)->max_weight_texel_count + 3u) & ~3u;
				// (a group's (weight, texel row) pairs fill at most one wave; its 12 term rows per weight live in LDS;
				//  16 = the per-weight decision records the kernel keeps, see realign_weights)
				uint32_t slots = std::min<uint32_t>(std::min(64u / rs, (16u * Tp) / (12u * rs)), 16u);
				if (slots < 1) slots = 1;
				blob.at<DecimationInfo>(di_off)->realign_slots = (uint8_t)slots;
				if (used(i)) rt_floats = std::max(rt_floats, slots * 12u * rs);
				build_realign_schedule(blob, di_off, slots);
			}
			r->realign_rt_floats = rt_floats;
		}
		r->max_weights[0] = r->max_weights[1] = 1;
		for (size_t i = 0; i < dms.size(); i++)
		{
			uint32_t wc = dm_grid[i].count();
			if (dms[i].refprec_1plane != 0) r->max_weights[0] = std::max(r->max_weights[0], wc);
			if (dms[i].refprec_2planes != 0) r->max_weights[1] = std::max(r->max_weights[1], wc);
		}
		for (int cls = 0; cls < 2; cls++)
		{
			r->dwi_total_floats[cls] = dwi_total[cls];
			r->off_dwi_owner[cls] = off_owner[cls];
			r->off_dwi_slots[cls] = off_slots[cls];
			r->off_infill_sets[cls] = off_isets[cls];
			r->dwi_sets[cls] = n_sets[cls];
			for (int q = 0; q < 12; q++) r->dwi_used_sets[cls][q] = used_sets[cls][q];
			r->lowhigh_floats[cls] = lh_total[cls];
			r->off_dwi_order[cls] = off_order[cls];
		}
		r->dwi_sets_per_chunk = sets_per_chunk;
		r->max_partitionings = std::max(pcounts[1], std::max(pcounts[2], pcounts[3]));
	}
	blob.alloc(0, 256);
	r = blob.at<TableRoot>(0);
	r->total_bytes = (uint32_t)blob.d.size();

	out.swap(blob.d);
	return true;
}

} // namespace astcd
