// SPDX-License-Identifier: Apache-2.0
// Quad vectors: a 4-component value (R, G, B, A of one endpoint) whose components live on the four lanes of a quad.
//
// The endpoint coders are channel-wise arithmetic with a handful of cross-channel steps (the RGB sum that orders
// two endpoints, "does any channel leave the byte range", the blue channel that the blue-contraction subtracts).
// Run on one lane per partition they are scalar code on a vector unit: every vector operation of the reference costs
// four instructions with one to four lanes doing anything.  Here partition p of the block owns lanes 4p .. 4p+3, lane
// 4p + ch holds component ch, a vector operation is ONE instruction for all partitions, and the cross-channel steps
// are quad-permute DPP moves (no LDS, no ballots).
//
//   device : qf / qi / qb hold one float / int / bool per lane; Q_CH is the lane's component
//   CPU    : the same names are 4-element structs and every operation loops over the components, so that the
//            sequential debug build (oracle/emu) runs the identical source
//
// A quad's lanes always execute together: every branch condition in quad code is built from q_any / q_all / values
// broadcast with q_get, i.e. it is the same on the four lanes.
#pragma once
#include "wave.h"
#include "wave_ctx.h"

namespace astcd { inline namespace ASTC_VARIANT {

#if WV_DEVICE

#define Q_CH (WV_LANE & 3)
/* Quad p runs on lanes 4p .. 4p+3; at most 16 quads per pass. */
#define WV_QUADS(p, n) for (int p = WV_LANE >> 2; p < (int)(n); p += 16)
/* ... at most 16 of them: one trip by construction (see WV_FOR64) */
#define WV_QUADS16(p, n) for (int p = WV_LANE >> 2, wv_once_##p = 1; wv_once_##p && p < (int)(n); wv_once_##p = 0)

struct qf { float v; };
struct qi { int v; };
struct qb { int v; };      // 0 / 1 (a bool member makes the compiler keep these temporaries in scratch memory)

template <int SEL> WV_FN int q_perm_i(int v) { return __builtin_amdgcn_update_dpp(0, v, SEL, 0xF, 0xF, true); }
template <int SEL> WV_FN float q_perm_f(float v) { return int_as_float(q_perm_i<SEL>(float_as_int(v))); }
constexpr int Q_SWAP1 = 1 | (0 << 2) | (3 << 4) | (2 << 6);      // lanes (0 1)(2 3) exchanged
constexpr int Q_SWAP2 = 2 | (3 << 2) | (0 << 4) | (1 << 6);      // lanes (0 2)(1 3) exchanged

/* component K of the quad on every lane */
template <int K> WV_FN float q_get(qf a) { return q_perm_f<K * 0x55>(a.v); }
template <int K> WV_FN int q_get(qi a) { return q_perm_i<K * 0x55>(a.v); }
template <int K> WV_FN bool q_get(qb a) { return q_perm_i<K * 0x55>(a.v) != 0; }

WV_FN qf q_splat(float s) { qf r; r.v = s; return r; }
WV_FN qi q_splat(int s) { qi r; r.v = s; return r; }
/* (x, y, z, w) */
WV_FN qf q_make(float x, float y, float z, float w) { const int c = Q_CH; qf r; r.v = c == 0 ? x : c == 1 ? y : c == 2 ? z : w; return r; }
WV_FN qi q_make(int x, int y, int z, int w) { const int c = Q_CH; qi r; r.v = c == 0 ? x : c == 1 ? y : c == 2 ? z : w; return r; }
WV_FN qf q_load(const float* p4) { qf r; r.v = p4[Q_CH]; return r; }
/* per-component map */
template <typename F> WV_FN qf q_map(qf a, F f) { qf r; r.v = f(a.v); return r; }
template <typename F> WV_FN qi q_mapi(qi a, F f) { qi r; r.v = f(a.v); return r; }
template <typename F> WV_FN qf q_zip(qf a, qf b, F f) { qf r; r.v = f(a.v, b.v); return r; }
template <typename F> WV_FN qi q_zipi(qi a, qi b, F f) { qi r; r.v = f(a.v, b.v); return r; }
template <typename F> WV_FN qi q_zip_if(qi a, qf b, F f) { qi r; r.v = f(a.v, b.v); return r; }
template <typename F> WV_FN qi q_zipi_ch(qi a, qi b, F f) { qi r; r.v = f(Q_CH, a.v, b.v); return r; }
template <typename F> WV_FN qf q_zip_ch(qf a, qf b, F f) { qf r; r.v = f(Q_CH, a.v, b.v); return r; }
template <typename F> WV_FN qb q_test(qi a, F f) { qb r; r.v = f(a.v) ? 1 : 0; return r; }
template <typename F> WV_FN qb q_testf(qf a, F f) { qb r; r.v = f(a.v) ? 1 : 0; return r; }
/* f(component index, value): component-dependent maps */
template <typename F> WV_FN qi q_mapi_ch(qi a, F f) { qi r; r.v = f(Q_CH, a.v); return r; }
template <typename F> WV_FN qf q_map_ch(qf a, F f) { qf r; r.v = f(Q_CH, a.v); return r; }
WV_FN qb q_zipb(qb a, qb b) { qb r; r.v = a.v | b.v; return r; }       // either test
WV_FN qi q_to_int(qf a) { qi r; r.v = (int)a.v; return r; }
WV_FN qf q_to_float(qi a) { qf r; r.v = (float)a.v; return r; }
WV_FN qf q_select(qb c, qf a, qf b) { qf r; r.v = c.v ? a.v : b.v; return r; }
WV_FN qi q_select(qb c, qi a, qi b) { qi r; r.v = c.v ? a.v : b.v; return r; }
/* table[index] per component */
WV_FN qi q_lookup(const uint8_t* table, qi index) { qi r; r.v = table[index.v]; return r; }

/* true on every lane of the quad if the test holds on any of its components (RGB only: the first three) */
WV_FN bool q_any(qb a)
{
	int v = a.v;
	v |= q_perm_i<Q_SWAP1>(v);
	v |= q_perm_i<Q_SWAP2>(v);
	return v != 0;
}
WV_FN bool q_any_rgb(qb a) { qb m; m.v = Q_CH != 3 ? a.v : 0; return q_any(m); }

/* x + y + z of the integer components (any association is exact) */
WV_FN int q_sum_rgb(qi a)
{
	int v = Q_CH == 3 ? 0 : a.v;
	v += q_perm_i<Q_SWAP1>(v);
	v += q_perm_i<Q_SWAP2>(v);
	return v;
}
/* (x + y) + z in that order (ref: hadd_rgb_s, vecmathlib_common_4.h:287) */
WV_FN float q_hadd_rgb(qf a) { return (q_get<0>(a) + q_get<1>(a)) + q_get<2>(a); }
/* (x + z) + (y + w) (ref: hadd_s, vecmathlib_none_4.h:907); float addition commutes, so both halves may arrive swapped */
WV_FN float q_hadd(qf a)
{
	float v = a.v + q_perm_f<Q_SWAP2>(a.v);
	return v + q_perm_f<Q_SWAP1>(v);
}
/* smallest component (finite values: exact whatever the order; ref: hmin, vecmathlib_none_4.h:888) */
WV_FN float q_hmin(qf a)
{
	float v = a.v;
	float o = q_perm_f<Q_SWAP1>(v); v = o < v ? o : v;
	o = q_perm_f<Q_SWAP2>(v); v = o < v ? o : v;
	return v;
}
/* largest component (finite values) */
WV_FN float q_hmax(qf a)
{
	float v = a.v;
	float o = q_perm_f<Q_SWAP1>(v); v = o > v ? o : v;
	o = q_perm_f<Q_SWAP2>(v); v = o > v ? o : v;
	return v;
}
/* stores: component ch of `a` to p[ch * stride] for ch < count */
WV_FN void q_store_u8(uint8_t* p, int stride, qi a, int count) { if (Q_CH < count) p[Q_CH * stride] = (uint8_t)a.v; }
WV_FN void q_store_i32(int* p, qi a) { p[Q_CH] = a.v; }
/* executed by one lane of the quad (scalar results: formats, flags) */
#define Q_ONCE if (Q_CH == 0)
/* Per-component code with loops and state of its own (a running sum per component, each over its own index sequence):
 * Q_LANES(l) { ... QV(a, l) ... } runs the body for component l = this lane's on the device, for l = 0 .. 3 in turn on the CPU. */
#define Q_LANES(l) for (int l = Q_CH, q_once_##l = 1; q_once_##l; q_once_##l = 0)
#define QV(a, l) ((a).v)

#else // ------------------------------------------------------------------------------------------------------

#if defined(ASTC_EMU_REVERSE_LANES)
#define WV_QUADS(p, n) for (int p = (int)(n) - 1; p >= 0; p--)
#else
#define WV_QUADS(p, n) for (int p = 0; p < (int)(n); p++)
#endif
#define WV_QUADS16(p, n) WV_QUADS(p, wv_checked_count((int)(n), 16))

struct qf { float v[4]; };
struct qi { int v[4]; };
struct qb { bool v[4]; };

template <int K> WV_FN float q_get(qf a) { return a.v[K]; }
template <int K> WV_FN int q_get(qi a) { return a.v[K]; }
template <int K> WV_FN bool q_get(qb a) { return a.v[K]; }
WV_FN qf q_splat(float s) { qf r; for (int k = 0; k < 4; k++) r.v[k] = s; return r; }
WV_FN qi q_splat(int s) { qi r; for (int k = 0; k < 4; k++) r.v[k] = s; return r; }
WV_FN qf q_make(float x, float y, float z, float w) { qf r; r.v[0] = x; r.v[1] = y; r.v[2] = z; r.v[3] = w; return r; }
WV_FN qi q_make(int x, int y, int z, int w) { qi r; r.v[0] = x; r.v[1] = y; r.v[2] = z; r.v[3] = w; return r; }
WV_FN qf q_load(const float* p4) { return q_make(p4[0], p4[1], p4[2], p4[3]); }
template <typename F> WV_FN qf q_map(qf a, F f) { qf r; for (int k = 0; k < 4; k++) r.v[k] = f(a.v[k]); return r; }
template <typename F> WV_FN qi q_mapi(qi a, F f) { qi r; for (int k = 0; k < 4; k++) r.v[k] = f(a.v[k]); return r; }
template <typename F> WV_FN qf q_zip(qf a, qf b, F f) { qf r; for (int k = 0; k < 4; k++) r.v[k] = f(a.v[k], b.v[k]); return r; }
template <typename F> WV_FN qi q_zipi(qi a, qi b, F f) { qi r; for (int k = 0; k < 4; k++) r.v[k] = f(a.v[k], b.v[k]); return r; }
template <typename F> WV_FN qi q_zip_if(qi a, qf b, F f) { qi r; for (int k = 0; k < 4; k++) r.v[k] = f(a.v[k], b.v[k]); return r; }
template <typename F> WV_FN qi q_zipi_ch(qi a, qi b, F f) { qi r; for (int k = 0; k < 4; k++) r.v[k] = f(k, a.v[k], b.v[k]); return r; }
template <typename F> WV_FN qf q_zip_ch(qf a, qf b, F f) { qf r; for (int k = 0; k < 4; k++) r.v[k] = f(k, a.v[k], b.v[k]); return r; }
template <typename F> WV_FN qb q_test(qi a, F f) { qb r; for (int k = 0; k < 4; k++) r.v[k] = f(a.v[k]); return r; }
template <typename F> WV_FN qb q_testf(qf a, F f) { qb r; for (int k = 0; k < 4; k++) r.v[k] = f(a.v[k]); return r; }
template <typename F> WV_FN qi q_mapi_ch(qi a, F f) { qi r; for (int k = 0; k < 4; k++) r.v[k] = f(k, a.v[k]); return r; }
template <typename F> WV_FN qf q_map_ch(qf a, F f) { qf r; for (int k = 0; k < 4; k++) r.v[k] = f(k, a.v[k]); return r; }
WV_FN qb q_zipb(qb a, qb b) { qb r; for (int k = 0; k < 4; k++) r.v[k] = a.v[k] || b.v[k]; return r; }
WV_FN qi q_to_int(qf a) { qi r; for (int k = 0; k < 4; k++) r.v[k] = (int)a.v[k]; return r; }
WV_FN qf q_to_float(qi a) { qf r; for (int k = 0; k < 4; k++) r.v[k] = (float)a.v[k]; return r; }
WV_FN qf q_select(qb c, qf a, qf b) { qf r; for (int k = 0; k < 4; k++) r.v[k] = c.v[k] ? a.v[k] : b.v[k]; return r; }
WV_FN qi q_select(qb c, qi a, qi b) { qi r; for (int k = 0; k < 4; k++) r.v[k] = c.v[k] ? a.v[k] : b.v[k]; return r; }
WV_FN qi q_lookup(const uint8_t* table, qi index) { qi r; for (int k = 0; k < 4; k++) r.v[k] = table[index.v[k]]; return r; }
WV_FN bool q_any(qb a) { return a.v[0] || a.v[1] || a.v[2] || a.v[3]; }
WV_FN bool q_any_rgb(qb a) { return a.v[0] || a.v[1] || a.v[2]; }
WV_FN int q_sum_rgb(qi a) { return a.v[0] + a.v[1] + a.v[2]; }
WV_FN float q_hadd_rgb(qf a) { return (a.v[0] + a.v[1]) + a.v[2]; }
WV_FN float q_hadd(qf a) { return (a.v[0] + a.v[2]) + (a.v[1] + a.v[3]); }
WV_FN float q_hmin(qf a) { float m = a.v[0]; for (int k = 1; k < 4; k++) m = a.v[k] < m ? a.v[k] : m; return m; }
WV_FN float q_hmax(qf a) { float m = a.v[0]; for (int k = 1; k < 4; k++) m = a.v[k] > m ? a.v[k] : m; return m; }
WV_FN void q_store_u8(uint8_t* p, int stride, qi a, int count) { for (int k = 0; k < count; k++) p[k * stride] = (uint8_t)a.v[k]; }
WV_FN void q_store_i32(int* p, qi a) { for (int k = 0; k < 4; k++) p[k] = a.v[k]; }
#define Q_ONCE if (true)
#define Q_LANES(l) for (int l = 0; l < 4; l++)
#define QV(a, l) ((a).v[l])

#endif

// ---- arithmetic shared by both builds ----
WV_FN qf operator+(qf a, qf b) { return q_zip(a, b, [](float x, float y) { return x + y; }); }
WV_FN qf operator-(qf a, qf b) { return q_zip(a, b, [](float x, float y) { return x - y; }); }
WV_FN qf operator*(qf a, qf b) { return q_zip(a, b, [](float x, float y) { return x * y; }); }
WV_FN qf operator/(qf a, qf b) { return q_zip(a, b, [](float x, float y) { return x / y; }); }
WV_FN qf operator*(qf a, float s) { return q_map(a, [s](float x) { return x * s; }); }
WV_FN qf operator+(qf a, float s) { return q_map(a, [s](float x) { return x + s; }); }
WV_FN qf operator-(qf a, float s) { return q_map(a, [s](float x) { return x - s; }); }
WV_FN qi operator+(qi a, qi b) { return q_zipi(a, b, [](int x, int y) { return x + y; }); }
WV_FN qi operator-(qi a, qi b) { return q_zipi(a, b, [](int x, int y) { return x - y; }); }
WV_FN qi operator|(qi a, qi b) { return q_zipi(a, b, [](int x, int y) { return x | y; }); }
WV_FN qi operator^(qi a, qi b) { return q_zipi(a, b, [](int x, int y) { return x ^ y; }); }
WV_FN qi operator&(qi a, int m) { return q_mapi(a, [m](int x) { return x & m; }); }
WV_FN qi operator<<(qi a, int s) { return q_mapi(a, [s](int x) { return (int)((unsigned)x << s); }); }
WV_FN qi operator>>(qi a, int s) { return q_mapi(a, [s](int x) { return x >> s; }); }
WV_FN qi operator*(qi a, int s) { return q_mapi(a, [s](int x) { return x * s; }); }
/* min(max(a, lo), hi) as compare-selects in the reference's operand order (ref: vecmathlib_common_4.h:225) */
WV_FN qf q_clamp(float lo, float hi, qf a) { return q_map(a, [lo, hi](float x) { return v_clamp(lo, hi, x); }); }
WV_FN qi q_clamp(int lo, int hi, qi a) { return q_mapi(a, [lo, hi](int x) { return i_min(i_max(x, lo), hi); }); }
/* (int)(a + 0.5f) (ref: float_to_int_rtn) */
WV_FN qi q_round_to_int(qf a) { return q_to_int(a + 0.5f); }
/* a value outside lo .. hi in any component? */
WV_FN bool q_any_rgb_outside(qi a, int lo, int hi) { return q_any_rgb(q_test(a, [lo, hi](int x) { return x < lo || x > hi; })); }
/* component 3 replaced by that of `w` / by a constant */
WV_FN qi q_with_w(qi a, qi w) { return q_zipi_ch(a, w, [](int ch, int x, int y) { return ch == 3 ? y : x; }); }
WV_FN qi q_with_w(qi a, int w) { return q_mapi_ch(a, [w](int ch, int x) { return ch == 3 ? w : x; }); }

/* the block's channel weights as a quad vector (cw4_of, wave_ctx.h: literals in a fixed-context build) */
WV_FN qf q_cw_of(const BlkInfo& blk) { const f4 w = cw4_of(blk); return q_make(w.x, w.y, w.z, w.w); }

} } // namespace astcd::ASTC_VARIANT
