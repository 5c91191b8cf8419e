// SPDX-License-Identifier: Apache-2.0
// Second VALU / LDS issue-cost table for gfx950: one row per instruction form the compression kernel uses.
// Same method as valu_microbench.hip (16 independent chains, 64 instructions per loop body, W waves per SIMD,
// one wave per workgroup); prints SIMD-cycles per wave-instruction at the nominal 2.4 GHz clock, so the
// rows are comparable with each other whatever the real clock is.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

#define KERNEL(NAME, BODY, POST) \
__global__ __launch_bounds__(64) void NAME(float* out, const unsigned char* tab, int iters, float seed) \
{ \
	float r[16]; unsigned u[16]; __shared__ float lds[2048]; \
	for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = (float)i; \
	__syncthreads(); \
	_Pragma("unroll") for (int i = 0; i < 16; i++) { r[i] = seed + i + threadIdx.x; u[i] = threadIdx.x * 4u + i * 256u; } \
	float s = seed * 1.0001f; unsigned long long m = 0x5555555555555555ull; unsigned sacc = 0; (void)m; (void)sacc; \
	for (int it = 0; it < iters; it++) { \
		_Pragma("unroll") for (int rep = 0; rep < 4; rep++) { REP16(BODY) } \
		POST \
	} \
	float acc = 0.0f; \
	_Pragma("unroll") for (int i = 0; i < 16; i++) acc += r[i] + (float)u[i]; \
	out[blockIdx.x * 64 + threadIdx.x] = acc + (float)sacc + lds[threadIdx.x]; \
}

#define NOPOST
#define LGKM asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#define VMW  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

#define B_CND_VCC(i)   asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(s));
#define B_CND_SGPR(i)  asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(r[i]) : "v"(s), "s"(m));
#define B_CMP_CND(i)   asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(s) : "vcc");
#define B_CMP64_CND(i) asm volatile("v_cmp_lt_f32_e64 %2, %0, %1\n v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(r[i]) : "v"(s), "s"(m));
#define B_MAX(i)       asm volatile("v_max_f32 %0, %0, %1" : "+v"(r[i]) : "v"(s));
#define B_MED3(i)      asm volatile("v_med3_f32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(s));
#define B_MOV(i)       asm volatile("v_mov_b32 %0, %1" : "=v"(r[i]) : "v"(s));
#define B_AND(i)       asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
#define B_OR(i)        asm volatile("v_or_b32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
#define B_XOR(i)       asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
#define B_LSHL(i)      asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(u[i]));
#define B_LSHR(i)      asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(u[i]));
#define B_BFE(i)       asm volatile("v_bfe_u32 %0, %0, 3, 8" : "+v"(u[i]));
#define B_ADD3(i)      asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
#define B_SUB_F32(i)   asm volatile("v_sub_f32 %0, %0, %1" : "+v"(r[i]) : "v"(s));
#define B_SUBREV(i)    asm volatile("v_subrev_f32 %0, %1, %0" : "+v"(r[i]) : "v"(s));
#define B_FMAC(i)      asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(r[i]) : "v"(s));
#define B_MUL_SGPR(i)  asm volatile("v_mul_f32 %0, %1, %0" : "+v"(r[i]) : "s"(seed));
#define B_MUL_LIT(i)   asm volatile("v_mul_f32 %0, 0x3f8ccccd, %0" : "+v"(r[i]));
#define B_ADD_E64(i)   asm volatile("v_add_f32_e64 %0, %0, %1" : "+v"(r[i]) : "v"(s));
#define B_ADD_ABS(i)   asm volatile("v_add_f32_e64 %0, |%0|, %1" : "+v"(r[i]) : "v"(s));
#define B_CVT_U32(i)   asm volatile("v_cvt_u32_f32 %0, %1" : "=v"(u[i]) : "v"(r[i]));
#define B_CVT_I32(i)   asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(u[i]) : "v"(r[i]));
#define B_CVT_UB0(i)   asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(r[i]) : "v"(u[i]));
#define B_RNDNE(i)     asm volatile("v_rndne_f32 %0, %0" : "+v"(r[i]));
#define B_MUL24(i)     asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
#define B_ADDCO(i)     asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 15]) : "vcc");
#define B_SUB_U32(i)   asm volatile("v_sub_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
#define B_RFL(i)       asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(sacc) : "v"(u[i]));
#define B_WRL(i)       asm volatile("v_writelane_b32 %0, %1, 5" : "+v"(u[i]) : "s"(iters));
#define B_PERM(i)      asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
#define B_MIN_U32(i)   asm volatile("v_min_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
#define B_MIN_I32(i)   asm volatile("v_min_i32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
#define B_CMP_U32(i)   asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(u[i]), "v"(u[(i + 1) & 15]) : "vcc");
#define B_CMPX(i)      asm volatile("v_cmp_lt_u32_e64 %0, %1, %2" : "=s"(m) : "v"(u[i]), "v"(u[(i + 1) & 15]));
#define B_MAD_F32(i)   asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
#define B_ADD_MIX(i)   asm volatile("v_add_f32 %0, %0, %1\n v_mul_lo_u32 %2, %2, %3" : "+v"(r[i]), "+v"(u[i]) : "v"(s), "v"(u[(i + 1) & 15]));
#define B_ADD_DEP(i)   asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[0]) : "v"(s));
#define B_FMA_DEP(i)   asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r[0]) : "v"(s));
#define B_DS_U8(i)     asm volatile("ds_read_u8 %0, %1" : "=v"(u[i]) : "v"((threadIdx.x * 4u + i * 64u) & 4095u) : "memory");
#define B_DS_B32(i)    asm volatile("ds_read_b32 %0, %1" : "=v"(u[i]) : "v"((threadIdx.x * 4u + i * 64u) & 4095u) : "memory");
#define B_DS_B64(i)    { unsigned long long t_; asm volatile("ds_read_b64 %0, %1" : "=v"(t_) : "v"((threadIdx.x * 8u + i * 64u) & 4095u) : "memory"); u[i] = (unsigned)t_; }
#define B_DS_W32(i)    asm volatile("ds_write_b32 %0, %1" : : "v"((threadIdx.x * 4u + i * 64u) & 4095u), "v"(u[i]) : "memory");
#define B_DS_SAME(i)   asm volatile("ds_read_b32 %0, %1" : "=v"(u[i]) : "v"(i * 64u) : "memory");
#define B_DS_GATHER(i) asm volatile("ds_read_b32 %0, %1" : "=v"(u[i]) : "v"(((threadIdx.x * 37u + i * 11u) & 1023u) * 4u) : "memory");
#define B_BPERM(i)     asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(u[i]) : "v"(threadIdx.x * 4u ^ 16u), "v"(u[(i + 1) & 15]) : "memory");
#define B_GL_U8(i)     asm volatile("global_load_ubyte %0, %1, %2" : "=v"(u[i]) : "v"((threadIdx.x + i * 64u) & 4095u), "s"(tab) : "memory");
#define B_GL_B32(i)    asm volatile("global_load_dword %0, %1, %2" : "=v"(u[i]) : "v"((threadIdx.x * 4u + i * 256u) & 4095u), "s"(tab) : "memory");


#define B_CMP_CND2(i)  asm volatile("v_cmp_lt_f32 vcc, %0, %2\n v_cndmask_b32 %0, %0, %2, vcc\n v_cndmask_b32 %1, %1, %2, vcc" : "+v"(r[i]), "+v"(u[i]) : "v"(s) : "vcc");
#define B_CMP_CND4(i)  asm volatile("v_cmp_lt_f32 vcc, %0, %2\n v_cndmask_b32 %0, %0, %2, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %0, %2, %0, vcc\n v_cndmask_b32 %1, %2, %1, vcc" : "+v"(r[i]), "+v"(u[i]) : "v"(s) : "vcc");
#define B_CMP64_CND4(i) asm volatile("v_cmp_lt_f32_e64 %3, %0, %2\n v_cndmask_b32_e64 %0, %0, %2, %3\n v_cndmask_b32_e64 %1, %1, %2, %3\n v_cndmask_b32_e64 %0, %2, %0, %3\n v_cndmask_b32_e64 %1, %2, %1, %3" : "+v"(r[i]), "+v"(u[i]) : "v"(s), "s"(m));
#define B_CND_ADD(i)   asm volatile("v_cndmask_b32 %0, %0, %2, vcc\n v_add_f32 %1, %1, %2" : "+v"(r[i]), "+v"(u[i]) : "v"(s));
#define B_CND_NEWDST(i) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(r[i]) : "v"(u[i]), "v"(s));
#define B_CND_E64VCC(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(s));
#define B_CND_SCMP(i)  asm volatile("s_mov_b64 vcc, %2\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(s), "s"(m) : "vcc");
#define B_CND_CONST(i) asm volatile("v_cndmask_b32 %0, 0, %1, vcc" : "+v"(r[i]) : "v"(s));
#define B_CND_DENORM(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(u[(i+1)&15]));
#define B_DIV(i)       r[i] = r[i] / s;
#define B_SQRTF(i)     r[i] = __builtin_sqrtf(r[i]);

KERNEL(k_cnd_vcc, B_CND_VCC, NOPOST)
KERNEL(k_cnd_sgpr, B_CND_SGPR, NOPOST)
KERNEL(k_cmp_cnd, B_CMP_CND, NOPOST)
KERNEL(k_cmp64_cnd, B_CMP64_CND, NOPOST)
KERNEL(k_max, B_MAX, NOPOST)
KERNEL(k_med3, B_MED3, NOPOST)
KERNEL(k_mov, B_MOV, NOPOST)
KERNEL(k_and, B_AND, NOPOST)
KERNEL(k_or, B_OR, NOPOST)
KERNEL(k_xor, B_XOR, NOPOST)
KERNEL(k_lshl, B_LSHL, NOPOST)
KERNEL(k_lshr, B_LSHR, NOPOST)
KERNEL(k_bfe, B_BFE, NOPOST)
KERNEL(k_add3, B_ADD3, NOPOST)
KERNEL(k_sub_f32, B_SUB_F32, NOPOST)
KERNEL(k_subrev, B_SUBREV, NOPOST)
KERNEL(k_fmac, B_FMAC, NOPOST)
KERNEL(k_mul_sgpr, B_MUL_SGPR, NOPOST)
KERNEL(k_mul_lit, B_MUL_LIT, NOPOST)
KERNEL(k_add_e64, B_ADD_E64, NOPOST)
KERNEL(k_add_abs, B_ADD_ABS, NOPOST)
KERNEL(k_cvt_u32, B_CVT_U32, NOPOST)
KERNEL(k_cvt_i32, B_CVT_I32, NOPOST)
KERNEL(k_cvt_ub0, B_CVT_UB0, NOPOST)
KERNEL(k_rndne, B_RNDNE, NOPOST)
KERNEL(k_mul24, B_MUL24, NOPOST)
KERNEL(k_addco, B_ADDCO, NOPOST)
KERNEL(k_sub_u32, B_SUB_U32, NOPOST)
KERNEL(k_rfl, B_RFL, NOPOST)
KERNEL(k_wrl, B_WRL, NOPOST)
KERNEL(k_perm, B_PERM, NOPOST)
KERNEL(k_min_u32, B_MIN_U32, NOPOST)
KERNEL(k_min_i32, B_MIN_I32, NOPOST)
KERNEL(k_cmp_u32, B_CMP_U32, NOPOST)
KERNEL(k_cmpx, B_CMPX, NOPOST)
KERNEL(k_add_mix, B_ADD_MIX, NOPOST)
KERNEL(k_add_dep, B_ADD_DEP, NOPOST)
KERNEL(k_fma_dep, B_FMA_DEP, NOPOST)
KERNEL(k_ds_u8, B_DS_U8, LGKM)
KERNEL(k_ds_b32, B_DS_B32, LGKM)
KERNEL(k_ds_b64, B_DS_B64, LGKM)
KERNEL(k_ds_w32, B_DS_W32, LGKM)
KERNEL(k_ds_same, B_DS_SAME, LGKM)
KERNEL(k_ds_gather, B_DS_GATHER, LGKM)
KERNEL(k_bperm, B_BPERM, LGKM)
KERNEL(k_gl_u8, B_GL_U8, VMW)
KERNEL(k_gl_b32, B_GL_B32, VMW)

KERNEL(k_cmp_cnd2, B_CMP_CND2, NOPOST)
KERNEL(k_cmp_cnd4, B_CMP_CND4, NOPOST)
KERNEL(k_cmp64_cnd4, B_CMP64_CND4, NOPOST)
KERNEL(k_cnd_add, B_CND_ADD, NOPOST)
KERNEL(k_cnd_newdst, B_CND_NEWDST, NOPOST)
KERNEL(k_cnd_e64vcc, B_CND_E64VCC, NOPOST)
KERNEL(k_cnd_scmp, B_CND_SCMP, NOPOST)
KERNEL(k_cnd_const, B_CND_CONST, NOPOST)
KERNEL(k_cnd_denorm, B_CND_DENORM, NOPOST)
KERNEL(k_div, B_DIV, NOPOST)
KERNEL(k_sqrtf, B_SQRTF, NOPOST)

typedef void (*kernel_fn)(float*, const unsigned char*, int, float);
struct Row { const char* name; kernel_fn fn; int per_body; };
static Row rows[] = {
	{ "cmp + 2 cndmask vcc (3)", k_cmp_cnd2, 3 }, { "cmp + 4 cndmask vcc (5)", k_cmp_cnd4, 5 }, { "cmp64 + 4 cndmask sgpr (5)", k_cmp64_cnd4, 5 },
	{ "cndmask vcc + v_add_f32 (2)", k_cnd_add, 2 }, { "cndmask vcc new dst", k_cnd_newdst, 1 }, { "cndmask e64 vcc", k_cnd_e64vcc, 1 },
	{ "s_mov vcc + cndmask (1 valu)", k_cnd_scmp, 1 }, { "cndmask vcc src0=0", k_cnd_const, 1 }, { "cndmask vcc int regs", k_cnd_denorm, 1 },
	{ "IEEE fp32 divide (per div)", k_div, 1 }, { "IEEE fp32 sqrt (per sqrt)", k_sqrtf, 1 },
	{ "v_cndmask_b32 vcc", k_cnd_vcc, 1 }, { "v_cndmask_b32 sgpr-pair", k_cnd_sgpr, 1 }, { "v_cmp+v_cndmask (vcc) pair", k_cmp_cnd, 2 },
	{ "v_cmp+v_cndmask (sgpr) pair", k_cmp64_cnd, 2 }, { "v_max_f32", k_max, 1 }, { "v_med3_f32", k_med3, 1 }, { "v_mov_b32", k_mov, 1 },
	{ "v_and_b32", k_and, 1 }, { "v_or_b32", k_or, 1 }, { "v_xor_b32", k_xor, 1 }, { "v_lshlrev_b32", k_lshl, 1 }, { "v_lshrrev_b32", k_lshr, 1 },
	{ "v_bfe_u32", k_bfe, 1 }, { "v_add3_u32", k_add3, 1 }, { "v_sub_f32", k_sub_f32, 1 }, { "v_subrev_f32", k_subrev, 1 }, { "v_fmac_f32", k_fmac, 1 },
	{ "v_mul_f32 sgpr src", k_mul_sgpr, 1 }, { "v_mul_f32 literal", k_mul_lit, 1 }, { "v_add_f32_e64", k_add_e64, 1 }, { "v_add_f32 |abs|", k_add_abs, 1 },
	{ "v_cvt_u32_f32", k_cvt_u32, 1 }, { "v_cvt_i32_f32", k_cvt_i32, 1 }, { "v_cvt_f32_ubyte0", k_cvt_ub0, 1 }, { "v_rndne_f32", k_rndne, 1 },
	{ "v_mul_u32_u24", k_mul24, 1 }, { "v_add_co_u32", k_addco, 1 }, { "v_sub_u32", k_sub_u32, 1 }, { "v_readfirstlane_b32", k_rfl, 1 },
	{ "v_writelane_b32", k_wrl, 1 }, { "v_perm_b32", k_perm, 1 }, { "v_min_u32", k_min_u32, 1 }, { "v_min_i32", k_min_i32, 1 },
	{ "v_cmp_lt_u32 vcc", k_cmp_u32, 1 }, { "v_cmp_lt_u32 sgpr", k_cmpx, 1 }, { "v_add_f32+v_mul_lo_u32 pair", k_add_mix, 2 },
	{ "v_add_f32 dependent chain", k_add_dep, 1 }, { "v_fma_f32 dependent chain", k_fma_dep, 1 },
	{ "ds_read_u8", k_ds_u8, 1 }, { "ds_read_b32", k_ds_b32, 1 }, { "ds_read_b64", k_ds_b64, 1 }, { "ds_write_b32", k_ds_w32, 1 },
	{ "ds_read_b32 same addr", k_ds_same, 1 }, { "ds_read_b32 gather", k_ds_gather, 1 }, { "ds_bpermute_b32", k_bperm, 1 },
	{ "global_load_ubyte (L2 hit)", k_gl_u8, 1 }, { "global_load_dword (L2 hit)", k_gl_b32, 1 },
};

int main(int argc, char** argv)
{
	int iters = argc > 1 ? atoi(argv[1]) : 1000;
	hipDeviceProp_t prop;
	CHECK(hipGetDeviceProperties(&prop, 0));
	int cus = prop.multiProcessorCount;
	double clock_hz = 2.4e9;
	int max_blocks = cus * 4 * 8;
	float* d_out; unsigned char* d_tab;
	CHECK(hipMalloc(&d_out, (size_t)max_blocks * 64 * 4));
	CHECK(hipMalloc(&d_tab, 8192)); CHECK(hipMemset(d_tab, 1, 8192));
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	printf("%-30s %s\n", "instruction", "SIMD-cycles per wave-instruction @2.4GHz at 1 / 2 / 4 / 8 waves per SIMD");
	for (size_t k = 0; k < sizeof(rows) / sizeof(rows[0]); k++)
	{
		printf("%-30s", rows[k].name);
		for (int w = 1; w <= 8; w *= 2)
		{
			int blocks = cus * 4 * w;
			hipLaunchKernelGGL(rows[k].fn, dim3(blocks), dim3(64), 0, 0, d_out, d_tab, 10, 1.0f);
			CHECK(hipDeviceSynchronize());
			CHECK(hipEventRecord(e0));
			hipLaunchKernelGGL(rows[k].fn, dim3(blocks), dim3(64), 0, 0, d_out, d_tab, iters, 1.0f);
			CHECK(hipEventRecord(e1));
			CHECK(hipEventSynchronize(e1));
			float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
			double per_simd = (double)w * iters * 64.0 * rows[k].per_body;
			printf(" %8.2f", ms * 1e-3 * clock_hz / per_simd);
		}
		printf("\n");
	}
	return 0;
}
