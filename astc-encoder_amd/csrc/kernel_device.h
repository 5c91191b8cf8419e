// SPDX-License-Identifier: Apache-2.0
// The compression kernel's device side: one 64-lane wavefront (= one workgroup) compresses one ASTC block; its working set
// is a dynamic-LDS region laid out by make_lds_layout().  Included by kernel_impl.h (the builds that are part of the library)
// and, as it is, by the translation unit the library writes and compiles at run time for one context (kernel_jit.cpp): no
// host code and no C library header in here or in anything it includes.  The including file sets
//   ASTC_VARIANT     inline-namespace tag of this build of the wave_*.h code
//   ASTC_ENABLE_HDR  0: LDR/sRGB profiles only (HDR endpoint coders compiled out), 1: everything
//   ASTC_KERNEL_NAME
#pragma once
#include "wave_block.h"

#ifndef ASTC_KERNEL_LINKAGE
#define ASTC_KERNEL_LINKAGE
#endif

#ifndef ASTC_WAVES_PER_EU
#define ASTC_WAVES_PER_EU 4
#endif

namespace astcd {

/* blockIdx -> ASTC block.  Workgroups are dealt round-robin to the 8 XCDs (block b -> XCD b % 8),
 * each with its own L2.  Raster-adjacent ASTC blocks share input cache lines, so every XCD gets a
 * contiguous run of the chunk rather than every 8th block. */
__device__ inline uint32_t xcd_block_remap(uint32_t b, uint32_t n)
{
	const uint32_t per = n / 8u;
	const uint32_t even = per * 8u;
	if (b >= even) return b;               // ragged tail keeps identity order
	return (b % 8u) * per + (b / 8u);
}

// (the occupancy bound twice: __launch_bounds__ is a macro of the HIP headers, and the run-time compiler of ROCm 7.0 drops its
//  second argument -- 160 VGPRs, three waves per SIMD -- where hipcc and the ROCm 7.2 run-time compiler honour it)
ASTC_KERNEL_LINKAGE __global__ void __launch_bounds__(64, ASTC_WAVES_PER_EU) __attribute__((amdgpu_waves_per_eu(ASTC_WAVES_PER_EU)))
ASTC_KERNEL_NAME(const uint8_t* __restrict__ tab, ImageDesc img,
                 uint8_t* __restrict__ out, uint32_t first_block, uint32_t num_blocks, unsigned long long* prof)
{
	uint8_t* lds = lds_base();

	uint32_t b = xcd_block_remap(blockIdx.x, num_blocks) + first_block;
	// raster block order: x fastest, then y, then z (ref: astcenc_entry.cpp:961-966)
	uint32_t row = b / img.blocks_x;
	uint32_t bx = b - row * img.blocks_x;
	uint32_t bz = img.blocks_z > 1 ? row / img.blocks_y : 0u;
	uint32_t by = row - bz * img.blocks_y;

	// one scalar base for the layout, the config and the tables: every field is then a non-negative immediate offset of
	// it (fields of `tab - CTX_LAYOUT_BACK` written as such cost a 64-bit subtraction per field)
	// (through an integer the optimiser cannot see through, and back as a pointer to constant memory -- a generic pointer
	//  would make every table read a flat load)
	typedef const __attribute__((address_space(4))) uint8_t* constant_bytes;
	uintptr_t base_bits = reinterpret_cast<uintptr_t>(tab) - CTX_LAYOUT_BACK;
	asm volatile("" : "+s"(base_bits));
	const uint8_t* const base = (const uint8_t*)(constant_bytes)base_bits;
	tab = base + CTX_LAYOUT_BACK;
	Ctx c;
	c.tab = tab;
	c.tab_constant = true;
	c.lds = lds;
#if ASTC_FIXED
	// (a fixed-context build: the three records are constants of this translation unit, wave_ctx.h)
	c.root = &kFixedRoot;
	c.cfg = &kFixedConfig;
	c.L = &kFixedLayout;
#else
	c.root = reinterpret_cast<const TableRoot*>(tab);
	c.cfg = reinterpret_cast<const DeviceConfig*>(base + (CTX_LAYOUT_BACK - CTX_CONFIG_BACK));
	c.L = reinterpret_cast<const LdsLayout*>(base);
#endif
	c.T = (int)c.L->texel_count;
#if defined(ASTC_FIXED_OPAQUE_TEXEL_COUNT)
	// (run-time builds for footprints of more than 64 texels, kernel_jit.cpp: the texel count of the lane loops is NOT a
	//  compile-time constant there.  With it, the builds of the 10x8 and 12x12 footprints -- 80 and 144 texels: the last trip of
	//  a texel loop has exactly sixteen lanes -- produce other bytes than the generic build on a quarter of noisy blocks; the
	//  same source with the same constants through g++ (the sequential build, tests/test_emu_fixed.py) does not, and neither does
	//  this build with the count read back through a register.  Not root-caused: DESIGN.md section 3.1.)
	c.T = wv_uniform(wv_opaque(c.T));
#endif
	c.Tp = (c.T + 3) & ~3;
	c.Ts = lds_row_stride(c.Tp);
#if defined(ASTC_TRACE)
	// trace builds: `prof` is the search trace buffer, one slice per block of the image (wave_ctx.h: TRACE_PUT)
	if (prof) prof = reinterpret_cast<unsigned long long*>(reinterpret_cast<uint32_t*>(prof) + (size_t)b * TRACE_WORDS_PER_BLOCK);
#endif
	c.prof = prof;

	// header for the out-of-line stage functions (ctx_make)
	WV_ONE
	{
		LdsHeader* h = reinterpret_cast<LdsHeader*>(lds);
		h->base = base;
		h->prof = prof;
		c.blk().block_index = b;
	}
	WV_SYNC();

	PROF_SCOPE(c, PS_TOTAL);
	{
		PROF_SCOPE(c, PS_LOAD);
		if (img.alpha_avg && !block_has_visible_alpha(c, img, bx, by)) load_transparent_block(c);
		else DUP_STAGE(c, DUP_LOAD, load_block(c, img, bx, by, bz));
	}
	compress_block(c, out);
}

} // namespace astcd
