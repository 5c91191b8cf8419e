// SPDX-License-Identifier: Apache-2.0
// What v_min_f32 / v_max_f32 / v_med3_f32 do with -0, +0 and NaN on gfx950 (IEEE mode as HIP kernels run): decides where
// the compare-select forms of the reference's min / max / clamp may become one hardware instruction (wave.h).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
__global__ void k(const float* a, const float* b, float* out, int n)
{
	int i = threadIdx.x;
	if (i >= n) return;
	float x = a[i], y = b[i], mn, mx, md;
	asm volatile("v_min_f32 %0, %1, %2" : "=v"(mn) : "v"(x), "v"(y));
	asm volatile("v_max_f32 %0, %1, %2" : "=v"(mx) : "v"(x), "v"(y));
	asm volatile("v_med3_f32 %0, %1, %2, %3" : "=v"(md) : "v"(x), "v"(0.0f), "v"(1.0f));
	out[3 * i] = mn; out[3 * i + 1] = mx; out[3 * i + 2] = md;
}
static unsigned bits(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
int main()
{
	const float nan = NAN, vals[] = { 0.0f, -0.0f, 1.0f, -1.0f, nan, 0.5f, 2.0f };
	float ha[49], hb[49]; int n = 0;
	for (float x : vals) for (float y : vals) { ha[n] = x; hb[n] = y; n++; }
	float *da, *db, *dout, ho[49 * 3];
	hipMalloc(&da, sizeof(ha)); hipMalloc(&db, sizeof(hb)); hipMalloc(&dout, sizeof(ho));
	hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
	k<<<1, 64>>>(da, db, dout, n);
	hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost);
	int bad_min = 0, bad_max = 0, bad_med = 0;
	for (int i = 0; i < n; i++)
	{
		const float x = ha[i], y = hb[i];
		const float sel_min = x < y ? x : y, sel_max = x > y ? x : y;                 // f_min / f_max of wave.h (NaN in x -> y)
		const float t = x > 0.0f ? x : 0.0f, clampzo = t < 1.0f ? t : 1.0f;           // v_clampzo
		const bool m1 = bits(ho[3 * i]) != bits(sel_min), m2 = bits(ho[3 * i + 1]) != bits(sel_max), m3 = bits(ho[3 * i + 2]) != bits(clampzo);
		if (m1 || m2 || (m3 && i % 7 == 0))
			printf("x=%g y=%g: v_min %g (select %g)%s  v_max %g (select %g)%s  med3(x,0,1) %g (clampzo %g)%s\n", x, y, ho[3 * i], sel_min, m1 ? " DIFF" : "",
			       ho[3 * i + 1], sel_max, m2 ? " DIFF" : "", ho[3 * i + 2], clampzo, m3 ? " DIFF" : "");
		bad_min += m1; bad_max += m2; if (i % 7 == 0) bad_med += m3;
	}
	printf("differences from the compare-select forms: v_min %d, v_max %d, v_med3 clamp %d (of %d / %d / 7 cases)\n", bad_min, bad_max, bad_med, n, n);
	// What wave.h relies on (tests/test_hw_semantics.py runs this on the GPU box and checks the exit status):
	//   f_clamp / v_clamp: v_med3_f32(x, 0, 1) has the bits of the compare-select clamp for EVERY x, NaN and -0 included;
	//   f_run_min / f_run_max: with a running value y that is a number, and no -0 on either side, v_min_f32 / v_max_f32
	//   have the bits of `x < y ? x : y` / `x > y ? x : y` -- for a NaN x that is y.
	int contract = bad_med;
	for (int i = 0; i < n; i++)
	{
		const float x = ha[i], y = hb[i];
		if (y != y || bits(x) == 0x80000000u || bits(y) == 0x80000000u) continue;
		const float sel_min = x < y ? x : y, sel_max = x > y ? x : y;
		contract += bits(ho[3 * i]) != bits(sel_min);
		contract += bits(ho[3 * i + 1]) != bits(sel_max);
	}
	printf("violations of the contract wave.h relies on: %d\n", contract);
	return contract == 0 ? 0 : 1;
}
