// SPDX-License-Identifier: Apache-2.0
// Execution model + exact-arithmetic helpers for the wavefront block compressor.
//
// EXECUTION MODEL.  One ASTC block is compressed by ONE 64-lane wavefront (workgroup = 64
// threads).  All block state lives in LDS.  Code is written in two kinds of region:
//
//   * uniform code  - every lane executes it with identical operands (values come from kernel
//                     arguments, read-only tables, or LDS).  Control flow of the search
//                     (trial order, early outs, candidate loops) is uniform code.
//   * WV_FOR(i, n)  - a lane-parallel loop: iteration i runs on lane i % 64.  Iterations must be
//                     independent; they communicate only through LDS and a following WV_SYNC().
//
// The same source also compiles as plain C++ (ASTC_WAVE_EMU) where WV_FOR is a sequential loop and
// WV_SYNC() is a no-op.  That build is a debugging aid for machines without a GPU (oracle/emu); the
// product library only contains the HIP build.
//
// NUMERICS CONTRACT.  Output bytes must equal the reference's scalar ("none") build, whose results
// depend on evaluation order and on IEEE single precision with no contraction.  Hence:
//   * this file's helpers restate the reference's scalar definitions operation for operation
//     (Source/astcenc_mathlib.h, astcenc_vecmathlib_none_4.h, astcenc_vecmathlib_common_4.h,
//     astcenc_vecmathlib.h); the translation units are built with -ffp-contract=off;
//   * min/max are compare-selects with the reference's operand order (NaN -> second operand);
//   * 4-lane horizontal sums use the reference's (l0 + l2) + (l1 + l3) order.
#pragma once
#include "astc_tables.h"      // (fixed-width integer types, also for the run-time compiler)

// Build variant: the LDR and HDR kernels are separate translation units of the same source so that
// the HDR endpoint coders do not weigh on the register allocation of the LDR hot path.
#ifndef ASTC_VARIANT
	#define ASTC_VARIANT v_all
#endif
#ifndef ASTC_ENABLE_HDR
	#define ASTC_ENABLE_HDR 1
#endif
#if defined(__HIPCC__) && !defined(__HIPCC_RTC__)      // (the run-time compiler brings the HIP device API with it)
#include <hip/hip_runtime.h>
#endif

#if defined(__HIP_DEVICE_COMPILE__)
	#define WV_DEVICE 1
	#define WV_FN __host__ __device__ inline
	#define WV_OUT static __host__ __device__ __attribute__((noinline))
	// The lane's index.  WV_LANE names `wv_lane_v`, which at namespace scope is the hardware's value -- ONE value for the whole
	// kernel, so everything derived from it (a lane's row address, its slot in a table) is loop-invariant everywhere and the
	// optimiser hoists such terms out of the search loops of compress_block and carries them, in vector registers, across
	// every trial.  A function that is inlined into the kernel body shadows the name with a local, opaque copy
	// (WV_LANE_SCOPE as its first statement): what is derived from the lane index is then computed, and dies, inside it.
	// (a workgroup is one wavefront: the index is below 64.  A kernel knows that from its launch bounds; a stage function reads
	//  the index out of the packed work-item register and would otherwise test `lane < 64` -- a compare, three exec-mask
	//  instructions -- around every one-trip loop over 64 items: the texel loops of an 8x8 block, the weight loops of two planes)
	struct WvLaneId { __device__ operator int() const { const unsigned l = threadIdx.x; __builtin_assume(l < 64u); return (int)l; } };
	static constexpr WvLaneId wv_lane_v{};
	#define WV_LANE ((int)wv_lane_v)
	#define WV_LANE_SCOPE int wv_lane_scope_src = (int)threadIdx.x; asm volatile("" : "+v"(wv_lane_scope_src)); __builtin_assume((unsigned)wv_lane_scope_src < 64u); const int wv_lane_v = wv_lane_scope_src
	// A workgroup is exactly one wavefront, and a wavefront's LDS instructions execute in issue
	// order, so a cross-lane hand-off through LDS needs no s_barrier and no s_waitcnt: it only needs
	// the compiler not to move or cache LDS accesses across this point.  (__syncthreads() would add
	// an `s_waitcnt lgkmcnt(0)` LDS round trip at every one of the thousands of hand-offs per block.)
	#define WV_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
	                       __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
	#define WV_FOR(i, n) for (int i = WV_LANE; i < (int)(n); i += 64)
	#define WV_ONE if (WV_LANE == 0)
	// WV_FOR for a count that is known to be at most 64 (the weights of a grid, partitions x channels, candidates ...): one
	// trip by construction -- the compiler cannot see the bound of a run-time count and would otherwise wrap the body in a
	// divergent loop (an exec-mask exit test of half a dozen instructions) that never takes its back edge
	#define WV_FOR64(i, n) for (int i = WV_LANE, wv_once_##i = 1; wv_once_##i && i < (int)(n); wv_once_##i = 0)
	// loops over the texels of a block: one trip in the kernel build for footprints of at most 64 texels (every 2D
	// footprint up to 8x8, 3D up to 4x4x4), the general loop in the build for the larger ones
	#if defined(ASTC_TEXELS_LE_64)
	#define WV_FOR_T(i, n) WV_FOR64(i, n)
	#else
	#define WV_FOR_T(i, n) WV_FOR(i, n)
	#endif
#else
	#define WV_DEVICE 0
	#if defined(__HIPCC__)
		#define WV_FN __host__ __device__ inline
		#define WV_OUT static __host__ __device__ __attribute__((noinline))
	#else
		#define WV_FN inline
		#define WV_OUT __attribute__((noinline)) inline
	#endif
	#define WV_SYNC() ((void)0)
	#if defined(ASTC_EMU_REVERSE_LANES)
	// debugging aid: run the lanes of every WV_FOR in reverse order; any output change means a loop
	// body depends on another lane's writes without a WV_SYNC() in between
	#define WV_FOR(i, n) for (int i = (int)(n) - 1; i >= 0; i--)
	#else
	#define WV_FOR(i, n) for (int i = 0; i < (int)(n); i++)
	#endif
	// (the sequential build checks what the device build relies on: a count above the promised bound stops the run)
	WV_FN int wv_checked_count(int n, int bound) { if (n > bound) __builtin_trap(); return n; }
	#define WV_FOR64(i, n) WV_FOR(i, wv_checked_count((int)(n), 64))
	// ... and the texel loops: one trip in the device builds for footprints of at most 64 texels, where loops such as
	// WV_FOR_T(k, groups * rows) lean on host-table invariants (slots x rows <= 64); the backend of the sequential build
	// sets this flag for such footprints so that a table change that breaks an invariant stops the run here
	extern thread_local bool g_wave_one_trip_texel_loops;
	#define WV_FOR_T(i, n) WV_FOR(i, (g_wave_one_trip_texel_loops ? wv_checked_count((int)(n), 64) : (int)(n)))
	#define WV_ONE if (true)
	#define WV_LANE_SCOPE ((void)0)
#endif

/* True on every lane if `flag` is true on any lane (flags are set inside WV_FOR bodies). */
WV_FN bool wv_any(bool flag)
{
#if WV_DEVICE
	return __ballot(flag) != 0ull;
#else
	return flag;
#endif
}

/* Minimum / maximum over the whole wave of per-lane partial results.  Usage: a variable declared outside a WV_FOR
 * is folded inside the loop body (`m = x < m ? x : m`), which on the device leaves one partial per lane (lanes
 * that ran no iteration keep the initial value) and in the sequential CPU build already the total; afterwards
 * wv_all_minmax(...) makes every argument the wave-wide result on every lane.  Operands must be finite and non-NaN:
 * then the result is the exact minimum / maximum whatever the association order.  Device: six DPP steps per value
 * (row_shr 1/2/4/8, row_bcast 15/31: v_min_f32_dpp / v_max_f32_dpp, one instruction each, the values interleaved so
 * that no wait states are needed between dependent steps) and a v_readlane of lane 63 -- no LDS traffic. */
#if WV_DEVICE
#define WV_DPP_MINMAX2(ctrl) \
	"v_min_f32_dpp %0, %0, %0 " ctrl "\n v_max_f32_dpp %1, %1, %1 " ctrl "\n"
#define WV_DPP_MINMAX4(ctrl) \
	"v_min_f32_dpp %0, %0, %0 " ctrl "\n v_max_f32_dpp %1, %1, %1 " ctrl "\n" \
	"v_min_f32_dpp %2, %2, %2 " ctrl "\n v_max_f32_dpp %3, %3, %3 " ctrl "\n"
#endif
/* (mn0, mx0), (mn1, mx1): two (minimum, maximum) pairs at once */
WV_FN void wv_all_minmax(float& mn0, float& mx0, float& mn1, float& mx1)
{
#if WV_DEVICE
	asm volatile("s_nop 1\n"
	    WV_DPP_MINMAX4("row_shr:1 row_mask:0xf bank_mask:0xf") WV_DPP_MINMAX4("row_shr:2 row_mask:0xf bank_mask:0xf")
	    WV_DPP_MINMAX4("row_shr:4 row_mask:0xf bank_mask:0xf") WV_DPP_MINMAX4("row_shr:8 row_mask:0xf bank_mask:0xf")
	    WV_DPP_MINMAX4("row_bcast:15 row_mask:0xa bank_mask:0xf") WV_DPP_MINMAX4("row_bcast:31 row_mask:0xc bank_mask:0xf")
	    "s_nop 1\n"
	    : "+v"(mn0), "+v"(mx0), "+v"(mn1), "+v"(mx1));
	mn0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mn0), 63));
	mx0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mx0), 63));
	mn1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mn1), 63));
	mx1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mx1), 63));
#else
	(void)mn0; (void)mx0; (void)mn1; (void)mx1;
#endif
}
WV_FN void wv_all_minmax(float& mn0, float& mx0)
{
#if WV_DEVICE
	asm volatile("s_nop 1\n"
	    WV_DPP_MINMAX2("row_shr:1 row_mask:0xf bank_mask:0xf") "s_nop 0\n" WV_DPP_MINMAX2("row_shr:2 row_mask:0xf bank_mask:0xf") "s_nop 0\n"
	    WV_DPP_MINMAX2("row_shr:4 row_mask:0xf bank_mask:0xf") "s_nop 0\n" WV_DPP_MINMAX2("row_shr:8 row_mask:0xf bank_mask:0xf") "s_nop 0\n"
	    WV_DPP_MINMAX2("row_bcast:15 row_mask:0xa bank_mask:0xf") "s_nop 0\n" WV_DPP_MINMAX2("row_bcast:31 row_mask:0xc bank_mask:0xf")
	    "s_nop 1\n"
	    : "+v"(mn0), "+v"(mx0));
	mn0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mn0), 63));
	mx0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mx0), 63));
#else
	(void)mn0; (void)mx0;
#endif
}

/* Maximum over the wave of a per-lane partial (same usage as wv_all_minmax): ballot-free butterfly through DPP. */
WV_FN int wv_all_imax(int v)
{
#if WV_DEVICE
	asm volatile("s_nop 1\n"
	    "v_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
	    "v_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
	    "v_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
	    "v_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
	    "v_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n s_nop 1\n"
	    "v_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n s_nop 1\n"
	    : "+v"(v));
	return __builtin_amdgcn_readlane(v, 63);
#else
	return v;
#endif
}

/* A small array with one element per lane index 0..127, written from WV_FOR bodies (element i by
 * the lane that runs iteration i) and read back with a wave-uniform index.  On the device it is two
 * VGPRs and a v_readlane, i.e. no memory at all; on the CPU it is an array. */
struct LaneArray128 {
#if WV_DEVICE
	int v0, v1;
	WV_FN void clear() { v0 = 0; v1 = 0; }
	WV_FN void set(int i, int value) { if (i < 64) v0 = value; else v1 = value; }
	WV_FN int get(int i) const { return i < 64 ? __builtin_amdgcn_readlane(v0, i) : __builtin_amdgcn_readlane(v1, i - 64); }
#else
	int v[128];
	WV_FN void clear() { for (int i = 0; i < 128; i++) v[i] = 0; }
	WV_FN void set(int i, int value) { v[i] = value; }
	WV_FN int get(int i) const { return v[i]; }
#endif
};

// Cold, bulky routines (HDR endpoint coders) are kept out of line so that they do not inflate the
// register pressure and code size of the LDR hot path.
#if defined(__HIPCC__)
	#define WV_NOINLINE __host__ __device__ __attribute__((noinline))
#else
	#define WV_NOINLINE __attribute__((noinline)) inline
#endif

namespace astcd { inline namespace ASTC_VARIANT {

constexpr bool kHdr = ASTC_ENABLE_HDR != 0;

// ---- scalar helpers (ref: astcenc_mathlib.h:168-331) ----
WV_FN float f_min(float p, float q) { return p < q ? p : q; }
WV_FN float f_max(float p, float q) { return p > q ? p : q; }
WV_FN int   i_min(int p, int q) { return p < q ? p : q; }
WV_FN int   i_max(int p, int q) { return p > q ? p : q; }

/* `x < m ? x : m` / `x > m ? x : m` for a running minimum / maximum m that is never NaN and values that are never -0
 * (the compare-select keeps m when x is NaN; so does the hardware's IEEE minimum / maximum, which returns the operand
 * that is a number): one instruction on the device instead of a compare and a select. */
WV_FN float f_run_min(float x, float m)
{
#if WV_DEVICE
	float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(m)); return r;
#else
	return x < m ? x : m;
#endif
}
WV_FN float f_run_max(float x, float m)
{
#if WV_DEVICE
	float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(m)); return r;
#else
	return x > m ? x : m;
#endif
}

/* The median of (v, mn, mx) in ONE instruction on the device.  For mn <= mx that are numbers it equals both clamp forms
 * of the reference below bit for bit -- NaN gives mn, -0 against a +0 bound gives +0 -- as tools/minmax_semantics.hip
 * checks on the hardware (v_med3_f32 returns the minimum of the numbers when an operand is NaN and orders -0 below +0);
 * the compare-select forms are two compares and two selects, four instructions of the slow issue class. */
#if WV_DEVICE
WV_FN float f_med3(float v, float mn, float mx)
{
	float r; asm("v_med3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(v), "v"(mn), "v"(mx)); return r;
}
#endif

/* acc + d * m for a mask m that is exactly 1.0f or 0.0f: the product d * m is d or a zero -- exact either way -- so the fused
 * form rounds once where the two-instruction form rounds once too, to the same bits (a zero product added to acc gives acc,
 * whose sums here are never -0); one instruction instead of two on the device.  (The translation units are built with
 * -ffp-contract=off: this is the one place where a fused multiply-add is asked for by name.) */
WV_FN float f_add_masked(float acc, float d, float m)
{
#if WV_DEVICE
	return __builtin_fmaf(d, m, acc);
#else
	return acc + d * m;
#endif
}

/* acc + k * d for k = 2 or -2: doubling is exact (no rounding, no overflow for the magnitudes here), so the fused form rounds
 * once -- like the reference's two-instruction acc + (2 d) -- to the same bits; one instruction less on the device. */
WV_FN float f_add_doubled(float acc, float d, float k)
{
#if WV_DEVICE
	return __builtin_fmaf(k, d, acc);
#else
	return acc + k * d;
#endif
}

/* astc::clamp (ref: astcenc_mathlib.h:271) */
WV_FN float f_clamp(float v, float mn, float mx)
{
#if WV_DEVICE
	return f_med3(v, mn, mx);
#else
	if (v > mx) return mx;
	if (v > mn) return v;
	return mn;
#endif
}
WV_FN int i_clamp(int v, int mn, int mx)
{
	if (v > mx) return mx;
	if (v > mn) return v;
	return mn;
}
WV_FN float f_clamp1(float v) { return f_clamp(v, 0.0f, 1.0f); }
WV_FN float f_clamp255(float v) { return f_clamp(v, 0.0f, 255.0f); }

/* vector-lane clamp: min(max(a, lo), hi) with compare-select (ref: vecmathlib_common_4.h:225) */
WV_FN float v_clamp(float lo, float hi, float a)
{
#if WV_DEVICE
	return f_med3(a, lo, hi);
#else
	float t = a > lo ? a : lo;
	return t < hi ? t : hi;
#endif
}
WV_FN float v_clampzo(float a) { return v_clamp(0.0f, 1.0f, a); }

WV_FN int flt2int_rtn(float v) { return (int)(v + 0.5f); }
WV_FN bool f_isnan(float v) { return v != v; }

WV_FN float f_abs(float v)
{
#if WV_DEVICE
	return __builtin_fabsf(v);
#else
	return __builtin_fabsf(v);
#endif
}

WV_FN float f_sqrt(float v)
{
	// correctly rounded on both sides (device build: -fhip-fp32-correctly-rounded-divide-sqrt)
	return __builtin_sqrtf(v);
}

/* round to nearest even (ref: vecmathlib_none_4.h:876 uses nearbyint under FE_TONEAREST) */
WV_FN float f_round(float v)
{
#if WV_DEVICE
	return __builtin_rintf(v);
#else
	return __builtin_nearbyintf(v);
#endif
}

WV_FN int float_as_int(float v) { int i; __builtin_memcpy(&i, &v, 4); return i; }
WV_FN float int_as_float(int v) { float f; __builtin_memcpy(&f, &v, 4); return f; }

/* horizontal add of 4 lanes (ref: vecmathlib_none_4.h:907) */
WV_FN float hadd4(float a, float b, float c, float d) { return (a + c) + (b + d); }
/* hadd_rgb_s (ref: vecmathlib_common_4.h:287) */
WV_FN float hadd3(float a, float b, float c) { return a + b + c; }
/* hmin / hmax (ref: vecmathlib_none_4.h:888-902; std::min(a,b) = b < a ? b : a) */
WV_FN float std_min(float a, float b) { return b < a ? b : a; }
WV_FN float std_max(float a, float b) { return a < b ? b : a; }
WV_FN float hmin4(float a, float b, float c, float d) { return std_min(std_min(a, b), std_min(c, d)); }
WV_FN float hmax4(float a, float b, float c, float d) { return std_max(std_max(a, b), std_max(c, d)); }

/* base[index] for a table in HBM with a lane-variant index: the byte offset is formed in 32 bits (every table is far
 * smaller than 4 GB), which lets the load use the scalar-base + 32-bit vector offset addressing mode; plain
 * `base[index]` with an unsigned index makes the compiler build a 64-bit address per access (shift + add pairs). */
template <typename T>
WV_FN const T& table_at(const T* base, uint32_t index)
{
	return *reinterpret_cast<const T*>(reinterpret_cast<const uint8_t*>(base) + (uint32_t)(index * (uint32_t)sizeof(T)));
}

/* The T at byte offset `offset` of a table (same addressing as table_at). */
template <typename T>
WV_FN const T& table_at_byte(const uint8_t* base, uint32_t offset)
{
	return *reinterpret_cast<const T*>(base + offset);
}

// ---- a 4-lane value type for the strictly scalar sections ----
struct f4 {
	float x, y, z, w;
};
WV_FN f4 mk4(float x, float y, float z, float w) { f4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
WV_FN f4 splat4(float v) { return mk4(v, v, v, v); }
WV_FN f4 operator+(f4 a, f4 b) { return mk4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
WV_FN f4 operator-(f4 a, f4 b) { return mk4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
WV_FN f4 operator*(f4 a, f4 b) { return mk4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
WV_FN f4 operator/(f4 a, f4 b) { return mk4(a.x / b.x, a.y / b.y, a.z / b.z, a.w / b.w); }
WV_FN f4 operator*(f4 a, float b) { return mk4(a.x * b, a.y * b, a.z * b, a.w * b); }
WV_FN f4 operator/(f4 a, float b) { return mk4(a.x / b, a.y / b, a.z / b, a.w / b); }
WV_FN float lane(f4 a, int i) { return i == 0 ? a.x : i == 1 ? a.y : i == 2 ? a.z : a.w; }
WV_FN void set_lane(f4& a, int i, float v) { if (i == 0) a.x = v; else if (i == 1) a.y = v; else if (i == 2) a.z = v; else a.w = v; }
WV_FN float hadd_s(f4 a) { return hadd4(a.x, a.y, a.z, a.w); }
WV_FN float hadd_rgb_s(f4 a) { return a.x + a.y + a.z; }
WV_FN float dot_s(f4 a, f4 b) { return hadd_s(a * b); }
WV_FN float dot3_s(f4 a, f4 b) { f4 m = a * b; return m.x + m.y + m.z; }
WV_FN f4 v4_min(f4 a, f4 b) { return mk4(a.x < b.x ? a.x : b.x, a.y < b.y ? a.y : b.y, a.z < b.z ? a.z : b.z, a.w < b.w ? a.w : b.w); }
WV_FN f4 v4_max(f4 a, f4 b) { return mk4(a.x > b.x ? a.x : b.x, a.y > b.y ? a.y : b.y, a.z > b.z ? a.z : b.z, a.w > b.w ? a.w : b.w); }
WV_FN f4 v4_clamp(float lo, float hi, f4 a) { return mk4(v_clamp(lo, hi, a.x), v_clamp(lo, hi, a.y), v_clamp(lo, hi, a.z), v_clamp(lo, hi, a.w)); }
WV_FN f4 v4_abs(f4 a) { return mk4(f_abs(a.x), f_abs(a.y), f_abs(a.z), f_abs(a.w)); }
WV_FN f4 v4_sqrt(f4 a) { return mk4(f_sqrt(a.x), f_sqrt(a.y), f_sqrt(a.z), f_sqrt(a.w)); }
WV_FN f4 load4(const float* p) { return mk4(p[0], p[1], p[2], p[3]); }
WV_FN void store4(float* p, f4 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w; }
/* the same for 16-byte aligned addresses: one 128-bit LDS access on the device */
WV_FN f4 load4_aligned(const float* p) { return load4(static_cast<const float*>(__builtin_assume_aligned(p, 16))); }
WV_FN void store4_aligned(float* p, f4 v) { store4(static_cast<float*>(__builtin_assume_aligned(p, 16)), v); }
/* swz<0,1,2>: 4th lane zero */
WV_FN f4 xyz0(f4 a) { return mk4(a.x, a.y, a.z, 0.0f); }

/* normalize (ref: vecmathlib.h:353) -- note dot() is the 4-lane hadd */
WV_FN f4 normalize4(f4 a)
{
	float len = dot_s(a, a);
	return a / splat4(f_sqrt(len));
}
/* normalize_safe (ref: vecmathlib.h:362) */
WV_FN f4 normalize_safe4(f4 a, f4 safe)
{
	float len = dot_s(a, a);
	if (len != 0.0f)
	{
		return a / splat4(f_sqrt(len));
	}
	return safe;
}
WV_FN f4 unit4() { return splat4(0.5f); }
WV_FN f4 unit3() { float v = 0.577350258827209473f; return mk4(v, v, v, 0.0f); }
WV_FN f4 unit2() { float v = 0.707106769084930420f; return mk4(v, v, 0.0f, 0.0f); }

struct i4 {
	int x, y, z, w;
};
WV_FN i4 mki4(int x, int y, int z, int w) { i4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
WV_FN int ilane(i4 a, int i) { return i == 0 ? a.x : i == 1 ? a.y : i == 2 ? a.z : a.w; }
WV_FN void set_ilane(i4& a, int i, int v) { if (i == 0) a.x = v; else if (i == 1) a.y = v; else if (i == 2) a.z = v; else a.w = v; }
WV_FN i4 operator+(i4 a, i4 b) { return mki4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
WV_FN i4 operator-(i4 a, i4 b) { return mki4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
WV_FN i4 operator*(i4 a, int b) { return mki4(a.x * b, a.y * b, a.z * b, a.w * b); }
WV_FN f4 int_to_float4(i4 a) { return mk4((float)a.x, (float)a.y, (float)a.z, (float)a.w); }

/* c ? a : b on all four components of an f4 with one condition.  (A VOP3 v_cndmask_b32_e64 version on a ballot mask,
 * tried because runs of three or more VOP2 v_cndmask_b32 cost ~22 cycles each in tools/valu_microbench2.hip, made no
 * difference in the kernel in two same-call A/B runs and cost ~1.4 k instructions per block: dropped.) */
WV_FN f4 select4(bool c, f4 a, f4 b) { return c ? a : b; }

/* atan2 approximation (ref: vecmathlib.h:275-306) */
WV_FN float ref_change_sign(float a, float b)
{
	int ia = float_as_int(a), ib = float_as_int(b);
	return int_as_float(ia ^ (ib & (int)0x80000000));
}
WV_FN float ref_atan(float x)
{
	const float PI_OVER_TWO = 1.57079632679489661923f;
	bool c = f_abs(x) > 1.0f;
	float z = ref_change_sign(PI_OVER_TWO, x);
	float y = c ? 1.0f / x : x;
	y = y / (y * y * 0.28f + 1.0f);
	return c ? z - y : y;
}
WV_FN float ref_atan2(float y, float x)
{
	const float PI = 3.14159265358979323846f;
	float z = ref_atan(f_abs(y / x));
	bool xmask = x < 0.0f;
	return ref_change_sign(xmask ? PI - z : z, y);
}

/* popcount of a 64-bit word */
WV_FN int popcount64(uint64_t v)
{
#if WV_DEVICE
	return __popcll(v);
#else
	return __builtin_popcountll(v);
#endif
}

WV_FN int popcount32(uint32_t v)
{
#if WV_DEVICE
	return __popc(v);
#else
	return __builtin_popcount(v);
#endif
}

/* IEEE binary16 -> binary32 (exact). (ref semantics: sf16_to_float, mathlib_softfloat.cpp) */
WV_FN float half_to_float(uint16_t h)
{
	uint32_t sign = (uint32_t)(h & 0x8000) << 16;
	uint32_t exp = (h >> 10) & 0x1F;
	uint32_t man = h & 0x3FF;
	uint32_t out;
	if (exp == 0)
	{
		if (man == 0) out = sign;
		else
		{
			// denormal: normalise
			int e = -1;
			do { man <<= 1; e++; } while (!(man & 0x400));
			out = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3FF) << 13);
		}
	}
	else if (exp == 31)
	{
		// inf / nan; the reference quiets NaNs
		out = sign | 0x7F800000u | (man << 13) | (man ? 0x400000u : 0);
	}
	else
	{
		out = sign | ((exp + 112) << 23) | (man << 13);
	}
	float f; __builtin_memcpy(&f, &out, 4); return f;
}

/* binary32 -> binary16, round to nearest even. (ref semantics: float_to_sf16 with SF_NEARESTEVEN) */
WV_FN uint16_t float_to_half(float f)
{
	uint32_t x; __builtin_memcpy(&x, &f, 4);
	uint32_t sign = (x >> 16) & 0x8000;
	uint32_t ax = x & 0x7FFFFFFF;
	if (ax >= 0x7F800000u)
	{
		if (ax > 0x7F800000u) return (uint16_t)(sign | 0x7C00 | ((ax >> 13) & 0x3FF) | 0x200); // quiet NaN
		return (uint16_t)(sign | 0x7C00);
	}
	if (ax >= 0x477FF000u) // rounds to >= 65520 -> inf
	{
		return (uint16_t)(sign | 0x7C00);
	}
	if (ax < 0x33000001u) // below half the smallest denormal -> zero
	{
		return (uint16_t)sign;
	}
	int e = (int)(ax >> 23) - 127;
	uint32_t m = (ax & 0x7FFFFF) | 0x800000;
	if (e < -14)
	{
		// denormal half: shift so that the result has 10 fraction bits at exponent -14
		int shift = 13 + (-14 - e);
		uint32_t r = m >> shift;
		uint32_t rem = m & ((1u << shift) - 1);
		uint32_t half = 1u << (shift - 1);
		if (rem > half || (rem == half && (r & 1))) r++;
		return (uint16_t)(sign | r);
	}
	uint32_t r = ((uint32_t)(e + 15) << 10) | ((m >> 13) & 0x3FF);
	uint32_t rem = m & 0x1FFF;
	if (rem > 0x1000 || (rem == 0x1000 && (r & 1))) r++;
	return (uint16_t)(sign | r);
}

} } // namespace astcd::ASTC_VARIANT
