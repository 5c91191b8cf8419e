#!/usr/bin/env python3
"""Generate tools/valu_microbench3.hip: VALU issue cost per wave64 instruction on gfx950 from hand-written asm loops.

What round 3's review asked for (VERDICT.md "settle the VALU ceiling"): the earlier tables
(tools/valu_microbench*.hip) let the compiler pick registers, so a 4.3-cycle class could have been VGPR bank
conflicts or SGPR operands rather than the pipe.  Here every loop is one asm block with explicit registers:

  * 16 independent chains, accumulator of chain i = v[32+i] (VGPR bank = register number mod 4 = i mod 4),
    sources v[8 + (i+1)%4] and v[8 + (i+2)%4]: the three operands of an instruction sit in three different banks
    ("banks: distinct"); a second variant puts all operands of an instruction into ONE bank ("banks: same");
  * no SGPR operands unless the row says so;
  * 64 instructions per loop iteration (4 x the 16 chains), `iters` iterations, s_memtime before and after
    (shader clock ticks, MI355X_MICROARCH.md) -> ticks per instruction per wave; with w waves resident on a
    SIMD and the pipe saturated, SIMD cycles per wave-instruction = ticks / (instructions x w);
  * w = 1, 2, 4, 8 waves per SIMD enforced by the LDS allocation of 256-thread workgroups (one wave per SIMD
    each): 160 KiB / w per workgroup, grid = 256 CUs x w;
  * the effective clock of the run = ticks / hipEvent time is printed too (DVFS: MI355X_MICROARCH.md).
"""
import os

ROWS = []   # (label, instruction template(s) per chain, instructions per chain entry, flags)

def row(label, tmpl, n=1, same_bank=False, setup=""):
    ROWS.append((label, tmpl, n, same_bank, setup))

for sb in (False, True):
    tag = " [same bank]" if sb else ""
    row("v_add_f32" + tag, "v_add_f32 {d}, {a}, {d}", same_bank=sb)
    row("v_mul_f32" + tag, "v_mul_f32 {d}, {a}, {d}", same_bank=sb)
    row("v_fma_f32" + tag, "v_fma_f32 {d}, {a}, {b}, {d}", same_bank=sb)
    row("v_max_f32" + tag, "v_max_f32 {d}, {a}, {d}", same_bank=sb)
    row("v_cndmask_b32_e64 (sgpr pair)" + tag, "v_cndmask_b32_e64 {d}, {d}, {a}, s[20:21]", same_bank=sb)
row("v_sub_f32", "v_sub_f32 {d}, {a}, {d}")
row("v_fmac_f32", "v_fmac_f32 {d}, {a}, {b}")
row("v_min_f32", "v_min_f32 {d}, {a}, {d}")
row("v_cmp_lt_f32 vcc + v_cndmask_b32 vcc (2)", "v_cmp_lt_f32 vcc, {a}, {d}\\n v_cndmask_b32 {d}, {d}, {b}, vcc", n=2)
row("v_cmp_lt_f32_e64 + v_cndmask_b32_e64 (2)", "v_cmp_lt_f32_e64 s[22:23], {a}, {d}\\n v_cndmask_b32_e64 {d}, {d}, {b}, s[22:23]", n=2)
row("v_cmp_lt_f32 vcc alone", "v_cmp_lt_f32 vcc, {a}, {d}")
row("v_cvt_f32_i32", "v_cvt_f32_i32 {d}, {d}")
row("v_cvt_i32_f32", "v_cvt_i32_f32 {d}, {d}")
row("v_cvt_f32_ubyte0", "v_cvt_f32_ubyte0 {d}, {d}")
row("v_rndne_f32", "v_rndne_f32 {d}, {d}")
row("v_add_f32_dpp row_shr:1", "v_add_f32_dpp {d}, {a}, {d} row_shr:1 row_mask:0xf bank_mask:0xf")
row("v_mov_b32_dpp row_shr:1", "v_mov_b32_dpp {d}, {a} row_shr:1 row_mask:0xf bank_mask:0xf")
row("v_mov_b32_dpp quad_perm", "v_mov_b32_dpp {d}, {a} quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
row("v_mov_b32", "v_mov_b32 {d}, {a}")
row("v_and_b32", "v_and_b32 {d}, {a}, {d}")
row("v_or_b32", "v_or_b32 {d}, {a}, {d}")
row("v_lshlrev_b32 by 1", "v_lshlrev_b32 {d}, 1, {d}")
row("v_lshrrev_b32 by 1", "v_lshrrev_b32 {d}, 1, {d}")
row("v_add_u32", "v_add_u32 {d}, {a}, {d}")
row("v_sub_u32", "v_sub_u32 {d}, {a}, {d}")
row("v_mul_u32_u24", "v_mul_u32_u24 {d}, {a}, {d}")
row("v_mad_u32_u24", "v_mad_u32_u24 {d}, {a}, {b}, {d}")
row("v_mul_lo_u32", "v_mul_lo_u32 {d}, {a}, {d}")
row("v_bfe_u32", "v_bfe_u32 {d}, {d}, 3, 8")
row("v_lshl_add_u32", "v_lshl_add_u32 {d}, {d}, 2, {a}")
row("v_add3_u32", "v_add3_u32 {d}, {d}, {a}, {b}")
row("v_min_u32", "v_min_u32 {d}, {a}, {d}")
row("v_mul_f32 sgpr src0", "v_mul_f32 {d}, s24, {d}")
row("v_add_f32 sgpr src0", "v_add_f32 {d}, s24, {d}")
row("v_mul_f32 literal", "v_mul_f32 {d}, 0x3f8ccccd, {d}")
row("v_add_f32 inline const", "v_add_f32 {d}, 1.0, {d}")
row("v_readlane_b32", "v_readlane_b32 s25, {d}, 3")
row("v_readfirstlane_b32", "v_readfirstlane_b32 s25, {d}")
row("v_rcp_f32", "v_rcp_f32 {d}, {d}")
row("v_sqrt_f32", "v_sqrt_f32 {d}, {d}")
row("v_pk_add_f32", "v_pk_add_f32 {D}, {A}, {D}")
row("v_pk_mul_f32", "v_pk_mul_f32 {D}, {A}, {D}")
row("v_pk_fma_f32", "v_pk_fma_f32 {D}, {A}, {B}, {D}")
row("v_add_f32 + v_max_f32 alternating (2)", "v_add_f32 {d}, {a}, {d}\\n v_max_f32 {d}, {b}, {d}", n=2)
row("v_add_f32 + v_cvt_f32_i32 alternating (2)", "v_add_f32 {d}, {a}, {d}\\n v_cvt_f32_i32 {d}, {d}", n=2)
row("v_add_f32 + s_and_b64 alternating (1 valu)", "v_add_f32 {d}, {a}, {d}\\n s_and_b64 s[26:27], s[26:27], s[20:21]", n=1)
row("v_max_f32 + s_and_b64 alternating (1 valu)", "v_max_f32 {d}, {a}, {d}\\n s_and_b64 s[26:27], s[26:27], s[20:21]", n=1)
row("v_add_f32 + s_waitcnt lgkmcnt(0) (1 valu)", "v_add_f32 {d}, {a}, {d}\\n s_waitcnt lgkmcnt(0)", n=1)
row("v_add_f32 + s_nop 0 (1 valu)", "v_add_f32 {d}, {a}, {d}\\n s_nop 0", n=1)
row("v_add_f32 + s_nop 1 (1 valu)", "v_add_f32 {d}, {a}, {d}\\n s_nop 1", n=1)
row("v_max_f32 + s_nop 1 (1 valu)", "v_max_f32 {d}, {a}, {d}\\n s_nop 1", n=1)
row("v_add_f32 + s_mov_b32 (1 valu)", "v_add_f32 {d}, {a}, {d}\\n s_mov_b32 s25, s24", n=1)
row("3 x v_add_f32 + v_max_f32 (4)", "v_add_f32 {d}, {a}, {d}\\n v_mul_f32 {d}, {b}, {d}\\n v_add_f32 {d}, {b}, {d}\\n v_max_f32 {d}, {b}, {d}", n=4)
row("v_add_f32 + 3 x v_max_f32 (4)", "v_add_f32 {d}, {a}, {d}\\n v_max_f32 {d}, {b}, {d}\\n v_min_f32 {d}, {a}, {d}\\n v_max_f32 {d}, {b}, {d}", n=4)
row("v_add_f32 dependent chain (1 chain)", "v_add_f32 v32, {a}, v32")
row("v_max_f32 dependent chain (1 chain)", "v_max_f32 v32, {a}, v32")
row("v_fma_f32 dependent chain (1 chain)", "v_fma_f32 v32, {a}, {b}, v32")

# round 6: is the scalar unit a pipe of its own (one instruction per ~4.3 cycles per SIMD beside the vector issue) or does a
# scalar / LDS instruction take a vector issue slot?  two and three fast vector instructions per scalar one decide.
row("2 x v_add_f32 + s_and_b64 (2 valu)", "v_add_f32 {d}, {a}, {d}\\n v_mul_f32 {d}, {b}, {d}\\n s_and_b64 s[26:27], s[26:27], s[20:21]", n=2)
row("3 x v_add_f32 + s_and_b64 (3 valu)", "v_add_f32 {d}, {a}, {d}\\n v_mul_f32 {d}, {b}, {d}\\n v_add_f32 {d}, {b}, {d}\\n s_and_b64 s[26:27], s[26:27], s[20:21]", n=3)
row("3 x v_add_f32 + s_add_u32 independent (3 valu)", "v_add_f32 {d}, {a}, {d}\\n v_mul_f32 {d}, {b}, {d}\\n v_add_f32 {d}, {b}, {d}\\n s_add_u32 s25, s24, 1", n=3)
row("3 x v_add_f32 + v_max_f32 + s_and_b64 (4 valu)", "v_add_f32 {d}, {a}, {d}\\n v_mul_f32 {d}, {b}, {d}\\n v_add_f32 {d}, {b}, {d}\\n v_max_f32 {d}, {b}, {d}\\n s_and_b64 s[26:27], s[26:27], s[20:21]", n=4)
row("v_add_f32 + ds_read_b32 (1 valu)", "v_add_f32 {d}, {a}, {d}\\n ds_read_b32 v20, v21", n=1)
row("3 x v_add_f32 + ds_read_b32 (3 valu)", "v_add_f32 {d}, {a}, {d}\\n v_mul_f32 {d}, {b}, {d}\\n v_add_f32 {d}, {b}, {d}\\n ds_read_b32 v20, v21", n=3)
row("3 x v_add_f32 + ds_read_b128 (3 valu)", "v_add_f32 {d}, {a}, {d}\\n v_mul_f32 {d}, {b}, {d}\\n v_add_f32 {d}, {b}, {d}\\n ds_read_b128 v[24:27], v21", n=3)
row("v_pk_add_f32 + v_add_f32 alternating (2 insts)", "v_pk_add_f32 {D}, {A}, {D}\\n v_add_f32 {d}, {a}, {d}", n=2)
row("v_pk_mul_f32 + 2 x v_add_f32 (3 insts)", "v_pk_mul_f32 {D}, {A}, {D}\\n v_add_f32 {d}, {a}, {d}\\n v_mul_f32 {d}, {b}, {d}", n=3)
row("v_pk_add_f32 + s_and_b64 (1 valu)", "v_pk_add_f32 {D}, {A}, {D}\\n s_and_b64 s[26:27], s[26:27], s[20:21]", n=1)
row("v_pk_add_f32 + v_max_f32 alternating (2 insts)", "v_pk_add_f32 {D}, {A}, {D}\\n v_max_f32 {d}, {b}, {d}", n=2)
row("v_fmac_f32 + v_max_f32 alternating (2)", "v_fmac_f32 {d}, {a}, {b}\\n v_max_f32 {d}, {b}, {d}", n=2)
row("v_cndmask_b32 vcc alone", "v_cndmask_b32 {d}, {d}, {a}, vcc")
row("v_cmp_lt_f32 vcc + 2 x v_add_f32 (3)", "v_cmp_lt_f32 vcc, {a}, {d}\\n v_add_f32 {d}, {a}, {d}\\n v_mul_f32 {d}, {b}, {d}", n=3)

def body(tmpl, same_bank):
    out = []
    for rep in range(4):
        for i in range(16):
            d = 32 + i
            if same_bank:
                a = 8 + (d % 4); b = 12 + (d % 4)
            else:
                a = 8 + ((i + 1) % 4); b = 8 + ((i + 2) % 4)
                if a == b: b = 8 + ((i + 3) % 4)
            # 64-bit operands for the packed forms: even-aligned pairs, chain i -> v[64+2i : 65+2i]; sources v[16:17] / v[18:19]
            D = "v[%d:%d]" % (64 + 2 * i, 65 + 2 * i)
            A = "v[16:17]" if i % 2 == 0 else "v[18:19]"
            B = "v[18:19]" if i % 2 == 0 else "v[16:17]"
            out.append(tmpl.format(d="v%d" % d, a="v%d" % a, b="v%d" % b, D=D, A=A, B=B))
    return "\\n ".join(out)

HDR = r'''// SPDX-License-Identifier: Apache-2.0
// GENERATED by tools/gen_valu_microbench3.py -- do not edit.  VALU issue cost per wave64 instruction on gfx950,
// hand-written asm loops with explicit registers (operand banks controlled, no SGPR operands unless named).
// build: hipcc --offload-arch=gfx950 -O2 -o tools/_build/valu_microbench3 tools/valu_microbench3.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
extern __shared__ unsigned char dyn_lds[];
#define CLOBBERS "v20","v21","v24","v25","v26","v27","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19", \
	"v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47", \
	"v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79", \
	"v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95", \
	"s20","s21","s22","s23","s24","s25","s26","s27","s28","s30","s31","s36","s37","vcc","scc","memory"
#define PROLOGUE \
	"v_mov_b32 v8, 1.0\n v_mov_b32 v9, 0.5\n v_mov_b32 v10, 2.0\n v_mov_b32 v11, 4.0\n" \
	"v_mov_b32 v12, 1.0\n v_mov_b32 v13, 0.5\n v_mov_b32 v14, 2.0\n v_mov_b32 v15, 4.0\n" \
	"v_mov_b32 v16, 1.0\n v_mov_b32 v17, 0.5\n v_mov_b32 v18, 2.0\n v_mov_b32 v19, 1.0\n" \
	"s_mov_b32 s20, 0x55555555\n s_mov_b32 s21, 0x33333333\n s_mov_b32 s24, 0x3f800000\n s_mov_b64 s[26:27], -1\n" \
	"s_mov_b32 s25, 0\n v_mov_b32 v21, 0\n"
'''

KERNEL = r'''
__global__ __launch_bounds__(256) void k%(idx)d(unsigned long long* out, int iters)
{
	unsigned long long t0, t1; unsigned hwid = 0, xcc = 0;
	// (initial accumulators: small distinct values per lane)
	asm volatile(PROLOGUE
	    "v_cvt_f32_u32 v32, %%4\n"
	    "v_mov_b32 v33, v32\n v_mov_b32 v34, v32\n v_mov_b32 v35, v32\n v_mov_b32 v36, v32\n v_mov_b32 v37, v32\n v_mov_b32 v38, v32\n v_mov_b32 v39, v32\n"
	    "v_mov_b32 v40, v32\n v_mov_b32 v41, v32\n v_mov_b32 v42, v32\n v_mov_b32 v43, v32\n v_mov_b32 v44, v32\n v_mov_b32 v45, v32\n v_mov_b32 v46, v32\n v_mov_b32 v47, v32\n"
	    "v_mov_b32 v64, v32\n v_mov_b32 v65, v32\n v_mov_b32 v66, v32\n v_mov_b32 v67, v32\n v_mov_b32 v68, v32\n v_mov_b32 v69, v32\n v_mov_b32 v70, v32\n v_mov_b32 v71, v32\n"
	    "v_mov_b32 v72, v32\n v_mov_b32 v73, v32\n v_mov_b32 v74, v32\n v_mov_b32 v75, v32\n v_mov_b32 v76, v32\n v_mov_b32 v77, v32\n v_mov_b32 v78, v32\n v_mov_b32 v79, v32\n"
	    "v_mov_b32 v80, v32\n v_mov_b32 v81, v32\n v_mov_b32 v82, v32\n v_mov_b32 v83, v32\n v_mov_b32 v84, v32\n v_mov_b32 v85, v32\n v_mov_b32 v86, v32\n v_mov_b32 v87, v32\n"
	    "v_mov_b32 v88, v32\n v_mov_b32 v89, v32\n v_mov_b32 v90, v32\n v_mov_b32 v91, v32\n v_mov_b32 v92, v32\n v_mov_b32 v93, v32\n v_mov_b32 v94, v32\n v_mov_b32 v95, v32\n"
	    "s_mov_b32 s28, %%5\n"
	    "s_barrier\n"
	    "s_memtime s[30:31]\n s_waitcnt lgkmcnt(0)\n"
	    "1:\n %(body)s\n"
	    "s_sub_u32 s28, s28, 1\n s_cmp_lg_u32 s28, 0\n s_cbranch_scc1 1b\n"
	    "s_memtime s[36:37]\n s_waitcnt lgkmcnt(0)\n"
	    "s_mov_b64 %%0, s[30:31]\n s_mov_b64 %%1, s[36:37]\n"
	    "s_getreg_b32 %%2, hwreg(HW_REG_HW_ID)\n s_getreg_b32 %%3, hwreg(HW_REG_XCC_ID)\n"
	    : "=s"(t0), "=s"(t1), "=s"(hwid), "=s"(xcc) : "v"(threadIdx.x), "s"(iters) : CLOBBERS);
	if ((threadIdx.x & 63) == 0)
	{
		unsigned long long* o = out + (size_t)(blockIdx.x * 4 + (threadIdx.x >> 6)) * 3;
		o[0] = t0; o[1] = t1; o[2] = ((unsigned long long)xcc << 32) | hwid;
	}
	if (iters < 0) dyn_lds[threadIdx.x] = 1;    // (keeps the dynamic LDS allocation referenced)
}
'''

MAIN = r'''
struct Row { const char* name; void (*fn)(unsigned long long*, int); int insts_per_iter; };
struct Wave { unsigned long long t0, t1, where; };
int main(int argc, char** argv)
{
	const int iters = argc > 1 ? atoi(argv[1]) : 4000;
	hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
	const int cus = prop.multiProcessorCount;
	const int max_waves = 4 * cus * 8;
	unsigned long long* d_out; CHECK(hipMalloc(&d_out, sizeof(unsigned long long) * 3 * max_waves));
	std::vector<unsigned long long> h(3 * max_waves);
	Row rows[] = {
%(rows)s	};
	printf("device: %%s, %%d CUs, nominal clock %%d MHz; %%d iterations of the listed instructions x 64 per wave\n", prop.gcnArchName, cus, prop.clockRate / 1000, iters);
	printf("per column (workgroups of 4 waves, 1 / 2 / 4 / 8 workgroups per CU launched): SIMD cycles per wave-instruction = (last end - first start of the waves that shared\n"
	       "a SIMD, in s_memtime ticks) / (their instructions), median over the SIMDs; (n) = median number of waves that shared a SIMD; [MHz] = ticks per microsecond of\n"
	       "the kernel's hipEvent time in the 4-per-CU run\n");
	hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	for (const Row& r : rows)
	{
		printf("%%-50s", r.name);
		double mhz4 = 0;
		for (int w = 1; w <= 8; w *= 2)
		{
			const int grid = cus * w, waves = 4 * grid;
			r.fn<<<grid, 256, 0>>>(d_out, 64);          // warm-up
			CHECK(hipDeviceSynchronize());
			CHECK(hipEventRecord(e0));
			r.fn<<<grid, 256, 0>>>(d_out, iters);
			CHECK(hipEventRecord(e1));
			CHECK(hipDeviceSynchronize());
			float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
			CHECK(hipMemcpy(h.data(), d_out, sizeof(unsigned long long) * 3 * waves, hipMemcpyDeviceToHost));
			// group the waves by the SIMD they ran on: XCC id, and HW_ID without its wave slot (bits 3:0) and without the
			// queue / pipe / VM id fields: SIMD 5:4, CU 11:8, SH 12, SE 15:13
			std::vector<Wave> v(waves);
			for (int i = 0; i < waves; i++) { v[i].t0 = h[3 * i]; v[i].t1 = h[3 * i + 1]; v[i].where = ((h[3 * i + 2] >> 32) << 16) | (h[3 * i + 2] & 0xFF30u); }
			std::sort(v.begin(), v.end(), [](const Wave& a, const Wave& b) { return a.where < b.where; });
			std::vector<double> per; std::vector<int> cnt;
			unsigned long long tmin = ~0ull, tmax = 0;
			for (int i = 0; i < waves; )
			{
				int j = i; unsigned long long a = ~0ull, b = 0;
				while (j < waves && v[j].where == v[i].where) { a = std::min(a, v[j].t0); b = std::max(b, v[j].t1); j++; }
				per.push_back((double)(b - a) / ((double)(j - i) * iters * r.insts_per_iter));
				cnt.push_back(j - i);
				tmin = std::min(tmin, a); tmax = std::max(tmax, b);
				i = j;
			}
			std::sort(per.begin(), per.end()); std::sort(cnt.begin(), cnt.end());
			printf(" %%6.2f (%%d)", per[per.size() / 2], cnt[cnt.size() / 2]);
			if (w == 4) mhz4 = (double)(tmax - tmin) / (ms * 1e3);
		}
		printf("   [%%.0f]\n", mhz4);
	}
	return 0;
}
'''

def main():
    here = os.path.dirname(os.path.abspath(__file__))
    parts = [HDR]
    rows = []
    only = os.environ.get("MB_FROM")
    rows_sel = ROWS[[r[0] for r in ROWS].index(only):] if only else ROWS
    for idx, (label, tmpl, n, sb, setup) in enumerate(rows_sel):
        parts.append(KERNEL % {"idx": idx, "body": body(tmpl, sb)})
        rows.append('\t\t{ "%s", k%d, %d },\n' % (label, idx, 64 * n))
    parts.append(MAIN % {"rows": "".join(rows)})
    with open(os.path.join(here, os.environ.get("MB_OUT", "valu_microbench3.hip")), "w") as f:
        f.write("".join(parts))

if __name__ == "__main__":
    main()
