// SPDX-License-Identifier: Apache-2.0
// HIP backend for gfx950 (MI355X): device-resident tables, staging buffers, and the compression
// kernel launch.  One 64-lane wavefront (one workgroup) compresses one ASTC block; its working
// set is a dynamic-LDS region laid out by make_lds_layout().
//
// Replaces the reference's CPU block loop (compress_image, Source/astcenc_entry.cpp:891-1043):
// instead of threads pulling 16-block tickets from an atomic counter, the block range of a chunk
// is one kernel launch; chunks give the host cancel/progress points.
#include "backend.h"

#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

namespace astcd {

struct Backend {
	int device;
	uint8_t* d_base;              // device allocation: context records, then the table blob
	uint8_t* d_tab;               // the blob inside it
	size_t tab_bytes;
	DeviceConfig cfg;
	uint32_t lds_bytes;
	bool hdr;
	TableRoot root;
	hipStream_t stream;
	hipStream_t copy_stream;      // PCIe traffic of the banded host-pointer path
	hipEvent_t ev0, ev1, ev_copy[2], ev_band;
	// staging for the host-pointer API
	void* d_image; size_t image_cap;
	uint8_t* d_out; size_t out_cap;
	float* d_alpha; size_t alpha_cap;   // alpha averages of the a_scale_radius pre-pass
	unsigned long long* d_prof;   // stage timers (ASTC_PROFILE builds)
	double* d_sums;               // totals of the image comparison kernel
};

// The library's own device buffers end in a little slack, so that a buffer never stops exactly at the end
// of a mapped page.
constexpr size_t ALLOC_SLACK = 4096;

#define HIP_TRY(expr, fail) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { \
	fprintf(stderr, "astcenc_amd: %s -> %s\n", #expr, hipGetErrorString(e_)); fail; } } while (0)

const char* backend_name() { return "hip:gfx950"; }

Backend* backend_create(const uint8_t* blob, size_t blob_bytes, const DeviceConfig& cfg, int* status)
{
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
	{
		fprintf(stderr, "astcenc_amd: no HIP device available; this library has no CPU fallback\n");
		*status = 2;
		return nullptr;
	}

	Backend* b = new Backend;
	memset(b, 0, sizeof(*b));
	HIP_TRY(hipGetDevice(&b->device), { delete b; *status = 2; return nullptr; });
	b->cfg = cfg;
	b->tab_bytes = blob_bytes;
	memcpy(&b->root, blob, sizeof(TableRoot));
	b->hdr = cfg.profile >= 2;
	uint8_t layout[256];
	uint32_t layout_bytes = 0;
	int prc = b->hdr ? astc_kernel_prepare_hdr(b->root, b->cfg, &b->lds_bytes, layout, &layout_bytes)
	                 : astc_kernel_prepare_ldr(b->root, b->cfg, &b->lds_bytes, layout, &layout_bytes);

	if (prc != 0 || b->lds_bytes > 160 * 1024)
	{
		fprintf(stderr, "astcenc_amd: kernel setup failed (hip error %d, block working set %u B; a CU has 160 KiB of LDS)\n", prc, b->lds_bytes);
		delete b; *status = 2; return nullptr;
	}

	// device allocation = [LdsLayout, 256 B][DeviceConfig, 256 B][table blob]; kernels get the blob pointer
	std::vector<uint8_t> full(CTX_LAYOUT_BACK + blob_bytes, 0);
	memcpy(full.data(), layout, layout_bytes);
	static_assert(sizeof(DeviceConfig) <= 256, "DeviceConfig outgrew its slot");
	memcpy(full.data() + (CTX_LAYOUT_BACK - CTX_CONFIG_BACK), &b->cfg, sizeof(DeviceConfig));
	memcpy(full.data() + CTX_LAYOUT_BACK, blob, blob_bytes);
	b->tab_bytes = full.size();
	HIP_TRY(hipMalloc(&b->d_base, full.size() + ALLOC_SLACK), { delete b; *status = 1; return nullptr; });
	HIP_TRY(hipMemcpy(b->d_base, full.data(), full.size(), hipMemcpyHostToDevice), { (void)hipFree(b->d_base); delete b; *status = 2; return nullptr; });
	b->d_tab = b->d_base + CTX_LAYOUT_BACK;
	HIP_TRY(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking), { (void)hipFree(b->d_base); delete b; *status = 2; return nullptr; });
	HIP_TRY(hipStreamCreateWithFlags(&b->copy_stream, hipStreamNonBlocking), { *status = 2; return nullptr; });
	HIP_TRY(hipEventCreateWithFlags(&b->ev_copy[0], hipEventDisableTiming), { *status = 2; return nullptr; });
	HIP_TRY(hipEventCreateWithFlags(&b->ev_copy[1], hipEventDisableTiming), { *status = 2; return nullptr; });
	HIP_TRY(hipEventCreateWithFlags(&b->ev_band, hipEventDisableTiming), { *status = 2; return nullptr; });
	HIP_TRY(hipEventCreate(&b->ev0), { *status = 2; return nullptr; });
	HIP_TRY(hipEventCreate(&b->ev1), { *status = 2; return nullptr; });
#if defined(ASTC_PROFILE)
	enum { PS_COUNT = 40, PS_TOTAL = 14 };
	HIP_TRY(hipMalloc(&b->d_prof, 2 * PS_COUNT * sizeof(unsigned long long)), { *status = 1; return nullptr; });
	HIP_TRY(hipMemset(b->d_prof, 0, 2 * PS_COUNT * sizeof(unsigned long long)), { *status = 2; return nullptr; });
#endif
	*status = 0;
	return b;
}

void backend_destroy(Backend* b)
{
	if (!b) return;
	(void)hipSetDevice(b->device);
	if (b->d_image) (void)hipFree(b->d_image);
	if (b->d_out) (void)hipFree(b->d_out);
	if (b->d_alpha) (void)hipFree(b->d_alpha);
	if (b->d_sums) (void)hipFree(b->d_sums);
	(void)hipEventDestroy(b->ev_copy[0]);
	(void)hipEventDestroy(b->ev_copy[1]);
	(void)hipEventDestroy(b->ev_band);
	(void)hipStreamDestroy(b->copy_stream);
	(void)hipEventDestroy(b->ev0);
	(void)hipEventDestroy(b->ev1);
	(void)hipStreamDestroy(b->stream);
	(void)hipFree(b->d_base);
	delete b;
}

int backend_compress(Backend* b, const CompressJob& job)
{
	HIP_TRY(hipSetDevice(b->device), return 2);

	const uint32_t bsx = b->root.dim_x, bsy = b->root.dim_y, bsz = b->root.dim_z;
	const uint32_t dim_z = job.dim_z ? job.dim_z : 1u;
	const uint32_t blocks_x = (job.dim_x + bsx - 1) / bsx;
	const uint32_t blocks_y = (job.dim_y + bsy - 1) / bsy;
	const uint32_t blocks_z = (dim_z + bsz - 1) / bsz;
	const size_t nblocks = (size_t)blocks_x * blocks_y * blocks_z;
	const size_t texel_bytes = job.data_type == 0 ? 4 : job.data_type == 1 ? 8 : 16;
	const size_t slice_bytes = (size_t)job.dim_x * job.dim_y * texel_bytes;
	const size_t image_bytes = slice_bytes * dim_z;
	const size_t out_bytes = nblocks * 16;

	hipStream_t stream = job.stream ? static_cast<hipStream_t>(job.stream) : b->stream;

	const void* d_image = job.device_data;
	uint8_t* d_out = job.device_out;

	if (job.host_slices)
	{
		if (b->image_cap < image_bytes)
		{
			if (b->d_image) (void)hipFree(b->d_image);
			b->d_image = nullptr; b->image_cap = 0;
			HIP_TRY(hipMalloc(&b->d_image, image_bytes + ALLOC_SLACK), return 1);
			b->image_cap = image_bytes;
		}
		d_image = b->d_image;
	}
	// Host-pointer calls on a plain 2D image are pipelined by bands of block rows: band k+1 travels over
	// PCIe on the copy stream while band k is being compressed, and band k's blocks travel back while band
	// k+1 runs (a block only reads texel rows of its own band).  The alpha-scale pre-pass and volumes need
	// the whole image first.
	const bool banded = job.host_slices && job.host_out && dim_z == 1 && job.a_scale_radius == 0;
	if (job.host_slices && !banded)
	{
		for (uint32_t z = 0; z < dim_z; z++)
			HIP_TRY(hipMemcpyAsync(static_cast<uint8_t*>(b->d_image) + z * slice_bytes, job.host_slices[z], slice_bytes, hipMemcpyHostToDevice, stream), return 2);
	}
	if (job.host_out)
	{
		if (b->out_cap < out_bytes)
		{
			if (b->d_out) (void)hipFree(b->d_out);
			b->d_out = nullptr; b->out_cap = 0;
			HIP_TRY(hipMalloc(&b->d_out, out_bytes + ALLOC_SLACK), return 1);
			b->out_cap = out_bytes;
		}
		d_out = b->d_out;
	}
	if (!d_image || !d_out) return 2;

	ImageDesc img;
	img.data = d_image;
	img.dim_x = job.dim_x; img.dim_y = job.dim_y;
	img.data_type = job.data_type;
	for (int i = 0; i < 4; i++) img.swz[i] = job.swz[i];
	img.blocks_x = blocks_x; img.blocks_y = blocks_y;
	img.dim_z = dim_z; img.blocks_z = blocks_z;
	bool needs_swz = job.swz[0] != 0 || job.swz[1] != 1 || job.swz[2] != 2 || job.swz[3] != 3;
	bool hdr = b->cfg.profile >= 2;
	img.use_fast_load = (!needs_swz && !hdr && job.data_type == 0 && bsz == 1) ? 1 : 0;   // ref: astcenc_entry.cpp:946
	img.alpha_avg = nullptr;
	img.a_scale_radius = job.a_scale_radius;
	if (job.a_scale_radius != 0)
	{
		const size_t need = (size_t)job.dim_x * job.dim_y * sizeof(float);
		if (b->alpha_cap < need)
		{
			if (b->d_alpha) (void)hipFree(b->d_alpha);
			b->d_alpha = nullptr; b->alpha_cap = 0;
			HIP_TRY(hipMalloc(&b->d_alpha, need + ALLOC_SLACK), return 1);
			b->alpha_cap = need;
		}
		AlphaLaunch a;
		a.d_image = d_image; a.d_averages = b->d_alpha;
		a.dim_x = job.dim_x; a.dim_y = job.dim_y; a.data_type = job.data_type;
		a.swz_a = job.swz[3]; a.radius = job.a_scale_radius; a.stream = stream;
		int arc = astc_alpha_launch(a);
		if (arc != 0) { fprintf(stderr, "astcenc_amd: alpha pre-pass launch failed (hip error %d)\n", arc); return 2; }
		img.alpha_avg = b->d_alpha;
	}

	// Chunks bound the time between cancel checks / progress callbacks on huge images; a chunk is
	// still tens of thousands of workgroups, far more than the 256 CUs need to stay full.
	size_t chunk = (job.progress || job.host_slices) ? (size_t)1 << 18 : nblocks;
	if (banded)
	{
		// whole block rows per band, at least four bands when the image has that many block rows
		size_t rows = chunk / blocks_x;
		if (rows * 4 > blocks_y) rows = (blocks_y + 3) / 4;
		if (rows < 1) rows = 1;
		chunk = rows * blocks_x;
	}
	const size_t row_bytes = (size_t)job.dim_x * texel_bytes;
	auto upload_band = [&](size_t band_first, size_t band_blocks) -> int {
		const size_t y0 = (band_first / blocks_x) * bsy;
		size_t y1 = ((band_first + band_blocks) / blocks_x) * bsy;
		if (y1 > job.dim_y) y1 = job.dim_y;
		HIP_TRY(hipMemcpyAsync(static_cast<uint8_t*>(b->d_image) + y0 * row_bytes, static_cast<const uint8_t*>(job.host_slices[0]) + y0 * row_bytes,
		                       (y1 - y0) * row_bytes, hipMemcpyHostToDevice, b->copy_stream), return 2);
		HIP_TRY(hipEventRecord(b->ev_copy[(band_first / chunk) & 1], b->copy_stream), return 2);
		return 0;
	};
	if (job.kernel_ms) HIP_TRY(hipEventRecord(b->ev0, stream), return 2);
	for (size_t first = 0; first < nblocks; first += chunk)
	{
		if (job.cancel_flag && *job.cancel_flag) break;
		size_t n = nblocks - first < chunk ? nblocks - first : chunk;
		if (banded)
		{
			// band 0 first; from then on the next band is queued before this band's kernel, so that it
			// crosses PCIe while the kernel runs (and ahead of this band's results on the copy stream)
			if (first == 0 && upload_band(0, n) != 0) return 2;
			const size_t next = first + n;
			if (next < nblocks && upload_band(next, nblocks - next < chunk ? nblocks - next : chunk) != 0) return 2;
			HIP_TRY(hipStreamWaitEvent(stream, b->ev_copy[(first / chunk) & 1], 0), return 2);
		}
		KernelLaunch k;
		k.d_tab = b->d_tab; k.lds_bytes = b->lds_bytes; k.img = img; k.d_out = d_out;
		k.first = (uint32_t)first; k.count = (uint32_t)n; k.stream = stream; k.d_prof = b->d_prof;
		int lrc = b->hdr ? astc_kernel_launch_hdr(k) : astc_kernel_launch_ldr(k);
		if (lrc != 0) { fprintf(stderr, "astcenc_amd: kernel launch failed (hip error %d)\n", lrc); return 2; }
		if (banded)
		{
			// this band's blocks go home on the copy stream once its kernel is done
			HIP_TRY(hipEventRecord(b->ev_band, stream), return 2);
			HIP_TRY(hipStreamWaitEvent(b->copy_stream, b->ev_band, 0), return 2);
			HIP_TRY(hipMemcpyAsync(job.host_out + first * 16, d_out + first * 16, n * 16, hipMemcpyDeviceToHost, b->copy_stream), return 2);
		}
		if (job.progress)
		{
			HIP_TRY(hipStreamSynchronize(stream), return 2);
			job.progress(100.0f * (float)(first + n) / (float)nblocks);
		}
	}
	if (job.kernel_ms) HIP_TRY(hipEventRecord(b->ev1, stream), return 2);

	if (job.host_out && !banded)
	{
		HIP_TRY(hipMemcpyAsync(job.host_out, d_out, out_bytes, hipMemcpyDeviceToHost, stream), return 2);
	}
	HIP_TRY(hipStreamSynchronize(stream), return 2);
	if (banded) HIP_TRY(hipStreamSynchronize(b->copy_stream), return 2);
	if (job.kernel_ms) HIP_TRY(hipEventElapsedTime(job.kernel_ms, b->ev0, b->ev1), return 2);
#if defined(ASTC_PROFILE)
	{
		enum { PS_COUNT = 40, PS_TOTAL = 14 };
		static const char* names[PS_COUNT] = { "load", "ideal", "decimate", "angular", "modes", "formats", "recompute", "pack", "diff",
		                                        "realign", "kmeans+partsearch", "  partscore", "physical", "stats", "TOTAL", "blocks",
		                                        "  dec sweep1", "  dec infill", "  dec sweep3", "  ang phase1", "  ang phase2",
		                                        "  mode terms", "  mode acc", "  mode quant", "  fmt eci", "  fmt table", "  fmt combine", "  fmt select",
		                                        "  cand staging", "  physical", "  refine (all)", "  trial (all)",
		                                        "  y0 cand quantize", "  y1 cand setup", "  y2 after pack", "  y3 accept/copy", "  y4", "  y5", "  y6", "  y7" };
		unsigned long long h[2 * PS_COUNT];
		HIP_TRY(hipMemcpy(h, b->d_prof, sizeof(h), hipMemcpyDeviceToHost), return 2);
		HIP_TRY(hipMemset(b->d_prof, 0, sizeof(h)), return 2);
		fprintf(stderr, "stage cycles per block (lane-0 shader clock), %zu blocks:   [calls per block]\n", nblocks);
		for (int i = 0; i < PS_COUNT; i++)
			if (i != 15 && h[i])
				fprintf(stderr, "  %-18s %12.0f  %5.1f%%   [%6.2f]\n", names[i], (double)h[i] / (double)nblocks, 100.0 * (double)h[i] / (double)h[PS_TOTAL],
				        (double)h[PS_COUNT + i] / (double)nblocks);
	}
#endif
	return 0;
}

int backend_decompress(Backend* b, const DecompressJob& job)
{
	HIP_TRY(hipSetDevice(b->device), return 2);
	const size_t texel_bytes = job.data_type == 0 ? 4 : job.data_type == 1 ? 8 : 16;
	const uint32_t dim_z = job.dim_z ? job.dim_z : 1u;
	const size_t slice_bytes = (size_t)job.dim_x * job.dim_y * texel_bytes;
	const size_t image_bytes = slice_bytes * dim_z;

	// the staging buffers of the compress path are reused the other way round
	if (b->image_cap < image_bytes)
	{
		if (b->d_image) (void)hipFree(b->d_image);
		b->d_image = nullptr; b->image_cap = 0;
		HIP_TRY(hipMalloc(&b->d_image, image_bytes + ALLOC_SLACK), return 1);
		b->image_cap = image_bytes;
	}
	if (b->out_cap < job.block_bytes)
	{
		if (b->d_out) (void)hipFree(b->d_out);
		b->d_out = nullptr; b->out_cap = 0;
		HIP_TRY(hipMalloc(&b->d_out, job.block_bytes + ALLOC_SLACK), return 1);
		b->out_cap = job.block_bytes;
	}
	HIP_TRY(hipMemcpyAsync(b->d_out, job.host_blocks, job.block_bytes, hipMemcpyHostToDevice, b->stream), return 2);

	DecodeLaunch d;
	d.d_blocks = b->d_out;
	d.d_image = b->d_image;
	d.dim_x = job.dim_x; d.dim_y = job.dim_y; d.dim_z = dim_z; d.data_type = job.data_type;
	for (int i = 0; i < 4; i++) d.swz[i] = job.swz[i];
	d.block_x = b->root.dim_x; d.block_y = b->root.dim_y; d.block_z = b->root.dim_z;
	d.profile = b->cfg.profile;
	d.stream = b->stream;
	int lrc = astc_decode_launch(d);
	if (lrc != 0) { fprintf(stderr, "astcenc_amd: decode kernel launch failed (hip error %d)\n", lrc); return 2; }
	for (uint32_t z = 0; z < dim_z; z++)
		HIP_TRY(hipMemcpyAsync(job.host_slices[z], static_cast<uint8_t*>(b->d_image) + z * slice_bytes, slice_bytes, hipMemcpyDeviceToHost, b->stream), return 2);
	HIP_TRY(hipStreamSynchronize(b->stream), return 2);
	return 0;
}

int backend_decompress_device(Backend* b, const DecompressDeviceJob& job)
{
	HIP_TRY(hipSetDevice(b->device), return 2);
	hipStream_t stream = job.stream ? static_cast<hipStream_t>(job.stream) : b->stream;
	DecodeLaunch d;
	d.d_blocks = job.device_blocks;
	d.d_image = job.device_image;
	d.dim_x = job.dim_x; d.dim_y = job.dim_y; d.dim_z = job.dim_z ? job.dim_z : 1u; d.data_type = job.data_type;
	for (int i = 0; i < 4; i++) d.swz[i] = job.swz[i];
	d.block_x = b->root.dim_x; d.block_y = b->root.dim_y; d.block_z = b->root.dim_z;
	d.profile = b->cfg.profile;
	d.stream = stream;
	int lrc = astc_decode_launch(d);
	if (lrc != 0) { fprintf(stderr, "astcenc_amd: decode kernel launch failed (hip error %d)\n", lrc); return 2; }
	HIP_TRY(hipStreamSynchronize(stream), return 2);
	return 0;
}

int backend_compare(Backend* b, const CompareJob& job)
{
	HIP_TRY(hipSetDevice(b->device), return 2);
	hipStream_t stream = job.stream ? static_cast<hipStream_t>(job.stream) : b->stream;
	if (!b->d_sums) HIP_TRY(hipMalloc(&b->d_sums, astc_compare_scratch_doubles() * sizeof(double)), return 1);
	CompareLaunch c;
	c.d_a = job.device_a; c.type_a = job.type_a; c.d_b = job.device_b; c.type_b = job.type_b;
	c.texels = job.texels; c.d_sums = b->d_sums; c.stream = stream;
	int lrc = astc_compare_launch(c);
	if (lrc != 0) { fprintf(stderr, "astcenc_amd: compare kernel launch failed (hip error %d)\n", lrc); return 2; }
	HIP_TRY(hipMemcpyAsync(job.sums, b->d_sums, 10 * sizeof(double), hipMemcpyDeviceToHost, stream), return 2);
	HIP_TRY(hipStreamSynchronize(stream), return 2);
	return 0;
}

} // namespace astcd
