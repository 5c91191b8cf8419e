// SPDX-License-Identifier: Apache-2.0
// Seam between the host API layer (astcenc_entry.cpp) and whatever executes the per-block
// compressor.  The product library links backend_hip.hip (HIP kernels on the current device).
// oracle/emu links backend_emu.cpp, which runs the same wave_*.h source sequentially on the CPU as
// a debugging aid -- it is never part of libastcenc_amd.so.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <atomic>
#include "astc_tables.h"

namespace astcd {

struct Backend;

struct CompressJob {
	const void* const* host_slices; // dim_z pointers to tightly packed RGBA rows (one 2D slice each), host memory; may be null
	const void* device_data;   // the slices back to back, already resident in HBM; used when host_slices is null
	uint32_t dim_x, dim_y, dim_z;
	uint32_t data_type;        // astcenc_type
	uint32_t swz[4];
	uint8_t* host_out;         // 16 bytes per block, host memory; may be null
	uint8_t* device_out;       // HBM destination when host_out is null
	void*    stream;           // hipStream_t for the device-resident path (null = backend's own)
	float*   kernel_ms;        // optional: elapsed kernel time measured with HIP events
	uint32_t a_scale_radius;   // != 0: alpha-average pre-pass, fully transparent neighbourhoods encode as constant zero
	const std::atomic<int>* cancel_flag; // polled between chunks
	void (*progress)(float);   // optional; called with a monotonically increasing percentage
	uint32_t fast_load_slice0; // multi-slice RGBA8 / LDR / identity-swizzle input with a 2D footprint: 1 = every slice reads
	                           // slice 0 like the reference's fast loader (astcenc_image.cpp:304), 0 = each slice reads itself
	// A block-row shard of a 2D host image with the alpha-scale pre-pass: host_slices[0] starts halo_above texel rows
	// above the shard's first row and halo_below rows follow its last one (dim_y counts the shard's own rows only).  The
	// pre-pass runs over all of them, the blocks of the shard's own rows are compressed (backend_compress sets these).
	uint32_t halo_above, halo_below;
};

struct DecompressJob {
	const uint8_t* host_blocks;   // 16 bytes per block, raster block order
	size_t   block_bytes;
	void* const* host_slices;     // dim_z pointers to tightly packed RGBA rows (one 2D slice each) of data_type, host memory
	uint32_t dim_x, dim_y, dim_z;
	uint32_t data_type;           // astcenc_type
	uint32_t swz[4];
};

/* Decompression with both ends in device memory (emu: host memory). */
struct DecompressDeviceJob {
	const uint8_t* device_blocks;
	void*    device_image;        // dim_z slices back to back
	uint32_t dim_x, dim_y, dim_z;
	uint32_t data_type;
	uint32_t swz[4];
	void*    stream;
};

/* Squared-error sums of two images of the same size (wave_metrics.h); sums[METRIC_SUMS] on the host. */
struct CompareJob {
	const void* device_a; uint32_t type_a;
	const void* device_b; uint32_t type_b;
	size_t texels;
	void* stream;
	double* sums;              // METRIC_SUMS_HOST doubles: [0..3] squared error, [4..7] alpha-scaled, [8] rgb peak, [10..13] log2, [14..17] mPSNR
	int hdr, fstop_lo, fstop_hi;   // hdr != 0: also the HDR sums over f-stops fstop_lo..fstop_hi
};
constexpr int METRIC_SUMS_HOST = 18;

/* status / return codes: 0 ok, 1 out of memory, 2 no usable device / launch failure, 3 bad argument
 * (a stream of another device than the buffers).
 * backend_create builds one device slot per GPU the context may use: every visible device by default, or the
 * ordinals listed in the environment variable ASTCENC_AMD_DEVICES.  backend_compress deals contiguous ranges of
 * block rows of a host image to those devices and joins them (the reference's N worker threads, N = devices);
 * device-resident buffers are compressed on the device that owns them. */
Backend* backend_create(const uint8_t* blob, size_t blob_bytes, const DeviceConfig& cfg, int* status);
void backend_destroy(Backend* b);
int backend_device_count(const Backend* b);
int backend_compress(Backend* b, const CompressJob& job);
int backend_decompress(Backend* b, const DecompressJob& job);
int backend_decompress_device(Backend* b, const DecompressDeviceJob& job);
int backend_compare(Backend* b, const CompareJob& job);
const char* backend_name();
/* Where the library's diagnostics go (null: nowhere, the default; see include/astcenc_amd.h). */
void backend_set_log_callback(void (*callback)(const char* message));


/* One kernel launch over blocks [first, first + count) of an image.  The kernel exists in two
 * builds of the same source (kernel_ldr.hip / kernel_hdr.hip). */
struct KernelLaunch {
	const uint8_t* d_tab;            // table blob in HBM (with the context's DeviceConfig and LdsLayout appended)
	uint32_t lds_bytes;              // dynamic LDS per workgroup
	ImageDesc img;
	uint8_t* d_out;
	uint32_t first, count;
	void* stream;                    // hipStream_t
	unsigned long long* d_prof;      // stage timers (profiling builds) or null
};

/* Return 0 on success, a hipError_t value otherwise. `prepare` sets the dynamic-LDS attribute and
 * reports the per-workgroup LDS bytes and the LdsLayout record (<= 256 bytes) that the backend
 * appends to the device copy of the table blob together with the DeviceConfig. */
int astc_kernel_prepare_ldr(const TableRoot& root, const DeviceConfig& cfg, uint32_t* lds_bytes, void* layout_out, uint32_t* layout_bytes);
int astc_kernel_prepare_hdr(const TableRoot& root, const DeviceConfig& cfg, uint32_t* lds_bytes, void* layout_out, uint32_t* layout_bytes);
int astc_kernel_launch_ldr(const KernelLaunch& k);
int astc_kernel_launch_hdr(const KernelLaunch& k);
// ... and the builds for footprints of at most 64 texels (kernel_ldr64.hip / kernel_hdr64.hip)
int astc_kernel_prepare_ldr64(const TableRoot& root, const DeviceConfig& cfg, uint32_t* lds_bytes, void* layout_out, uint32_t* layout_bytes);
int astc_kernel_prepare_hdr64(const TableRoot& root, const DeviceConfig& cfg, uint32_t* lds_bytes, void* layout_out, uint32_t* layout_bytes);
int astc_kernel_launch_ldr64(const KernelLaunch& k);
int astc_kernel_launch_hdr64(const KernelLaunch& k);
// ... and the fixed-context builds (kernel_ldr_6x6m.hip, kernel_ldr_8x8t.hip, kernel_hdr_6x6m.hip): `prepare` returns
// ASTC_PREPARE_NOT_THIS_CONTEXT when the context is not the one the build was compiled for
constexpr int ASTC_PREPARE_NOT_THIS_CONTEXT = -1;
#define ASTC_DECLARE_KERNEL_VARIANT(tag) \
	int astc_kernel_prepare_##tag(const TableRoot& root, const DeviceConfig& cfg, uint32_t* lds_bytes, void* layout_out, uint32_t* layout_bytes); \
	int astc_kernel_launch_##tag(const KernelLaunch& k);
ASTC_DECLARE_KERNEL_VARIANT(ldr_6x6m)
ASTC_DECLARE_KERNEL_VARIANT(ldr_8x8t)
ASTC_DECLARE_KERNEL_VARIANT(hdr_6x6m)
#undef ASTC_DECLARE_KERNEL_VARIANT
const char* backend_kernel_name(const Backend* b);   // the build of the compression kernel this context launches
/* Waits for the context's specialised build (compiling it now if that has not started) and switches every device of the
 * context to it.  0: the context launches a specialised build (one of the library's fixed-context builds or its own run-time
 * build); 1: it stays on the generic build (kernel_jit.h says when). */
int backend_specialize(Backend* b);

/* Alpha-average pre-pass launch (kernel_alpha.hip).  The padded tile of a region lives in LDS while it fits
 * (ALPHA_LDS_LIMIT) and otherwise in d_scratch: astc_alpha_scratch_bytes() says how much of it and for how many
 * workgroups the launch needs (0 / 0: the LDS kernel is used). */
constexpr size_t ALPHA_LDS_LIMIT = 160u * 1024u;
// tile edge of the pre-pass on a single slice (= ALPHA_TILE of wave_alpha.h, kernel_alpha.hip asserts it): the halo rows a
// block-row shard takes along are counted in these tiles (backend_compress)
constexpr uint32_t ALPHA_TILE_ROWS_2D = 32;
struct AlphaLaunch {
	const void* d_image;
	float* d_averages;
	uint32_t dim_x, dim_y, dim_z, data_type, swz_a, radius;
	float* d_scratch; uint32_t scratch_workgroups;
	void* stream;
};
size_t astc_alpha_scratch_bytes(uint32_t dim_x, uint32_t dim_y, uint32_t dim_z, uint32_t radius, uint32_t* workgroups);
int astc_alpha_launch(const AlphaLaunch& a);

/* Decompression kernel launch (kernel_decode.hip). */
struct DecodeLaunch {
	const uint8_t* d_blocks;
	void* d_image;
	const void* d_tables;            // the footprint's decoder tables in HBM (astc_decode_tables_build)
	uint32_t dim_x, dim_y, dim_z, data_type, swz[4];
	uint32_t block_x, block_y, block_z, profile;
	void* stream;
};
int astc_decode_launch(const DecodeLaunch& d);
/* The per-footprint tables of the decoder (block mode field -> weight grid, bit budget -> colour quant level): built on
 * the host once per context into astc_decode_tables_bytes() bytes, uploaded with the context's other tables. */
size_t astc_decode_tables_bytes();
void astc_decode_tables_build(void* out, uint32_t block_x, uint32_t block_y, uint32_t block_z);

/* Image comparison launch (kernel_metrics.hip); d_sums = astc_compare_scratch_doubles() doubles of device memory,
 * the totals arrive in the first ten. */
struct CompareLaunch {
	const void* d_a; uint32_t type_a;
	const void* d_b; uint32_t type_b;
	size_t texels;
	double* d_sums;
	void* stream;
	int hdr, fstop_lo, fstop_hi;
};
int astc_compare_launch(const CompareLaunch& c);
size_t astc_compare_scratch_doubles();

} // namespace astcd
