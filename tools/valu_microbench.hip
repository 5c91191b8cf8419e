// SPDX-License-Identifier: Apache-2.0
// VALU issue-rate microbenchmark for gfx950 ("next round" item 1.i).
//
// Question: does a wave64 fp32 VALU instruction occupy its SIMD for 2 or for 4 cycles, and do the packed
// fp32 forms (v_pk_add_f32 / v_pk_mul_f32: two floats per lane per instruction) issue at the same
// per-instruction rate?  The answer decides whether SQ_ACTIVE_INST_VALU / SIMD-cycles = 85 % means "issue
// bound" or "42 % busy, latency bound", and whether packed maths halves the cost of RGBA channel arithmetic.
//
// Method: every wave runs ITER iterations of an unrolled body of 64 instructions of one kind over 16
// independent register chains (no dependency stall at >= 1 wave/SIMD for 4+ cycle latencies), with W waves
// per SIMD resident (grid = 256 CUs x 4 SIMDs x W waves, one wave per workgroup).  Reports SIMD-cycles per
// instruction = elapsed * clock * 1024 SIMDs / total wave-instructions, with the clock taken from
// s_memrealtime-independent wall time and the device's reported clock, and also from a lane-0 s_memtime delta
// (shader clock counter) so that the figure does not depend on the assumed frequency.
//
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_microbench tools/valu_microbench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

enum Kind { K_ADD_F32, K_MUL_F32, K_FMA_F32, K_PK_ADD_F32, K_PK_MUL_F32, K_PK_FMA_F32, K_ADD_U32, K_MAD_U32_U24, K_CNDMASK,
            K_CVT_F32_I32, K_RCP_F32, K_SQRT_F32, K_MUL_LO_U32, K_LSHL_ADD, K_READLANE, K_DPP_ADD, K_MIN_F32, K_CMP_F32,
            K_ADD_F32_HALFEXEC, K_ADD_F32_16LANES, K_DS_READ_B32, K_COUNT };
static const char* kind_name[K_COUNT] = { "v_add_f32", "v_mul_f32", "v_fma_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_pk_fma_f32",
            "v_add_u32", "v_mad_u32_u24", "v_cndmask_b32", "v_cvt_f32_i32", "v_rcp_f32", "v_sqrt_f32", "v_mul_lo_u32", "v_lshl_add_u32",
            "v_readlane_b32", "v_add_f32 dpp", "v_min_f32", "v_cmp_lt_f32", "v_add_f32 (32 lanes on)", "v_add_f32 (16 lanes on)",
            "ds_read_b32" };

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int KIND>
__global__ __launch_bounds__(64) void bench_kernel(float* out, unsigned long long* cycles, int iters, float seed)
{
	float r[16];
	float2 p[16];
	unsigned u[16];
	__shared__ float lds[1024];
	for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (float)i;
	__syncthreads();
	#pragma unroll
	for (int i = 0; i < 16; i++) { r[i] = seed + i + threadIdx.x; p[i] = make_float2(seed + i, seed - i); u[i] = threadIdx.x * 4u + i * 256u; }
	float s = seed * 1.0001f;
	unsigned sacc = 0;
	if (KIND == K_ADD_F32_HALFEXEC && threadIdx.x >= 32) { iters = 0; }
	if (KIND == K_ADD_F32_16LANES && threadIdx.x >= 16) { iters = 0; }
	unsigned long long t0 = __builtin_amdgcn_s_memtime();
	for (int it = 0; it < iters; it++)
	{
		#pragma unroll
		for (int rep = 0; rep < 4; rep++)
		{
#define A_ADD(i)  asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[i]) : "v"(s));
#define A_MUL(i)  asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[i]) : "v"(s));
#define A_FMA(i)  asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(s));
#define A_PKADD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 15]));
#define A_PKMUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 15]));
#define A_PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 15]));
#define A_ADDU(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
#define A_MAD24(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
#define A_CND(i)  asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(s));
#define A_CVT(i)  asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(r[i]));
#define A_RCP(i)  asm volatile("v_rcp_f32 %0, %0" : "+v"(r[i]));
#define A_SQRT(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(r[i]));
#define A_MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
#define A_LSHLADD(i) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
#define A_RDLANE(i) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(sacc) : "v"(u[i]));
#define A_DPP(i)  asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(r[i]));
#define A_MIN(i)  asm volatile("v_min_f32 %0, %0, %1" : "+v"(r[i]) : "v"(s));
#define A_CMP(i)  asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(r[i]), "v"(s) : "vcc");
#define A_DSRD(i) asm volatile("ds_read_b32 %0, %1" : "=v"(r[i]) : "v"(u[i] & 4095u) : "memory");
			if (KIND == K_ADD_F32 || KIND == K_ADD_F32_HALFEXEC || KIND == K_ADD_F32_16LANES) { REP16(A_ADD) }
			if (KIND == K_MUL_F32) { REP16(A_MUL) }
			if (KIND == K_FMA_F32) { REP16(A_FMA) }
			if (KIND == K_PK_ADD_F32) { REP16(A_PKADD) }
			if (KIND == K_PK_MUL_F32) { REP16(A_PKMUL) }
			if (KIND == K_PK_FMA_F32) { REP16(A_PKFMA) }
			if (KIND == K_ADD_U32) { REP16(A_ADDU) }
			if (KIND == K_MAD_U32_U24) { REP16(A_MAD24) }
			if (KIND == K_CNDMASK) { REP16(A_CND) }
			if (KIND == K_CVT_F32_I32) { REP16(A_CVT) }
			if (KIND == K_RCP_F32) { REP16(A_RCP) }
			if (KIND == K_SQRT_F32) { REP16(A_SQRT) }
			if (KIND == K_MUL_LO_U32) { REP16(A_MULLO) }
			if (KIND == K_LSHL_ADD) { REP16(A_LSHLADD) }
			if (KIND == K_READLANE) { REP16(A_RDLANE) }
			if (KIND == K_DPP_ADD) { REP16(A_DPP) }
			if (KIND == K_MIN_F32) { REP16(A_MIN) }
			if (KIND == K_CMP_F32) { REP16(A_CMP) }
			if (KIND == K_DS_READ_B32) { REP16(A_DSRD) }
		}
		if (KIND == K_DS_READ_B32) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
	}
	unsigned long long t1 = __builtin_amdgcn_s_memtime();
	float acc = 0.0f;
	#pragma unroll
	for (int i = 0; i < 16; i++) acc += r[i] + p[i].x + p[i].y + (float)u[i];
	out[blockIdx.x * 64 + threadIdx.x] = acc + (float)sacc;
	if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

typedef void (*kernel_fn)(float*, unsigned long long*, int, float);
template <int K> static kernel_fn get() { return bench_kernel<K>; }
static kernel_fn table[K_COUNT] = { get<0>(), get<1>(), get<2>(), get<3>(), get<4>(), get<5>(), get<6>(), get<7>(), get<8>(), get<9>(), get<10>(),
                                   get<11>(), get<12>(), get<13>(), get<14>(), get<15>(), get<16>(), get<17>(), get<18>(), get<19>(), get<20>() };

int main(int argc, char** argv)
{
	int iters = argc > 1 ? atoi(argv[1]) : 4000;
	hipDeviceProp_t prop;
	CHECK(hipGetDeviceProperties(&prop, 0));
	int cus = prop.multiProcessorCount;
	double clock_hz = prop.clockRate * 1e3;
	printf("device %s, %d CUs, clockRate %.0f MHz\n", prop.name, cus, clock_hz / 1e6);
	int max_blocks = cus * 4 * 8;
	float* d_out; unsigned long long* d_cyc;
	CHECK(hipMalloc(&d_out, (size_t)max_blocks * 64 * 4));
	CHECK(hipMalloc(&d_cyc, (size_t)max_blocks * 8));
	std::vector<unsigned long long> cyc(max_blocks);
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	printf("%-26s %5s %10s %14s %16s %18s\n", "instruction", "w/SIMD", "ms", "Ginstr/s", "SIMDcyc/instr@clk", "memtime ticks/instr");
	for (int k = 0; k < K_COUNT; k++)
	{
		for (int w = 1; w <= 8; w *= 2)
		{
			int blocks = cus * 4 * w;
			hipLaunchKernelGGL(table[k], dim3(blocks), dim3(64), 0, 0, d_out, d_cyc, 10, 1.0f);   // warm
			CHECK(hipDeviceSynchronize());
			CHECK(hipEventRecord(e0));
			hipLaunchKernelGGL(table[k], dim3(blocks), dim3(64), 0, 0, d_out, d_cyc, iters, 1.0f);
			CHECK(hipEventRecord(e1));
			CHECK(hipEventSynchronize(e1));
			float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
			CHECK(hipMemcpy(cyc.data(), d_cyc, (size_t)blocks * 8, hipMemcpyDeviceToHost));
			double tick = 0; for (int b = 0; b < blocks; b++) tick += (double)cyc[b];
			tick /= blocks;
			double wave_instr = (double)blocks * iters * 64.0;
			double per_simd = wave_instr / (cus * 4.0);
			double simd_cycles = ms * 1e-3 * clock_hz;
			// ticks per instruction as seen by one wave, divided by resident waves = SIMD ticks per instruction
			printf("%-26s %5d %10.3f %14.1f %16.2f %18.3f\n", kind_name[k], w, ms, wave_instr / (ms * 1e-3) / 1e9, simd_cycles / per_simd,
			       tick / (iters * 64.0) / w);
		}
	}
	return 0;
}
