// SPDX-License-Identifier: Apache-2.0
// TEST INFRASTRUCTURE ONLY.  Runs the reference's own quality report -- compute_error_metrics(),
// Source/astcenccli_error_metrics.cpp:110, compiled from where it lies by oracle/Makefile -- on two raw
// images, so that the product's on-device metric (astcenc_amd_compare_images[_hdr]_device) can be checked
// against the figures the reference CLI would print, not only against a restatement of its formulas.
//
//   metrics_harness <u8|f16|f32> <dim_x> <dim_y> <image1.raw> <image2.raw> <hdr 0|1> <fstop_lo> <fstop_hi>
// The report goes to stdout exactly as the CLI prints it.
#include "astcenccli_internal.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

static std::vector<unsigned char> slurp(const char* path, size_t expect)
{
	std::vector<unsigned char> buf(expect);
	FILE* f = fopen(path, "rb");
	if (!f || fread(buf.data(), 1, expect, f) != expect) { fprintf(stderr, "cannot read %zu bytes from %s\n", expect, path); exit(2); }
	fclose(f);
	return buf;
}

int main(int argc, char** argv)
{
	if (argc != 9) { fprintf(stderr, "usage: metrics_harness <u8|f16|f32> x y img1 img2 hdr fstop_lo fstop_hi\n"); return 1; }
	astcenc_type type = !strcmp(argv[1], "u8") ? ASTCENC_TYPE_U8 : !strcmp(argv[1], "f16") ? ASTCENC_TYPE_F16 : ASTCENC_TYPE_F32;
	size_t texel = type == ASTCENC_TYPE_U8 ? 4 : type == ASTCENC_TYPE_F16 ? 8 : 16;
	unsigned int x = (unsigned)atoi(argv[2]), y = (unsigned)atoi(argv[3]);
	std::vector<unsigned char> a = slurp(argv[4], texel * x * y), b = slurp(argv[5], texel * x * y);
	void* sa = a.data(); void* sb = b.data();
	astcenc_image i1, i2;
	i1.dim_x = i2.dim_x = x; i1.dim_y = i2.dim_y = y; i1.dim_z = i2.dim_z = 1;
	i1.data_type = i2.data_type = type;
	i1.data = &sa; i2.data = &sb;
	compute_error_metrics(atoi(argv[6]) != 0, false, 4, &i1, &i2, atoi(argv[7]), atoi(argv[8]));
	return 0;
}
