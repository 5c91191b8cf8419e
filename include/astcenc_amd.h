/* SPDX-License-Identifier: Apache-2.0
 *
 * astcenc_amd.h -- MI355X-specific additions to the astcenc C ABI (libastcenc_amd.so).
 *
 * The reference API (include/astcenc.h) takes host pointers, so every astcenc_compress_image()
 * pays a PCIe round trip.  Pipelines that already hold their textures in HBM (and the benchmark,
 * which must time the kernels with inputs resident) use the entry point below instead.  It runs
 * the same kernels with the same context; nothing else about the contract changes.
 */
#ifndef ASTCENC_AMD_INCLUDED
#define ASTCENC_AMD_INCLUDED

#include "astcenc.h"

/* Compress a 2D image that is already resident in device memory.
 *
 *   device_image : device pointer, tightly packed RGBA rows, dim_x * dim_y texels of data_type
 *   device_out   : device pointer, receives 16 bytes per block in raster block order
 *   data_len     : bytes available at device_out (>= 16 * blocks, else ASTCENC_ERR_OUT_OF_MEM)
 *   hip_stream   : hipStream_t to launch on (NULL = the context's own stream); the call returns
 *                  after the work on that stream has completed
 *   kernel_ms    : optional; receives the elapsed time of the compression kernel(s) measured with
 *                  HIP events recorded on that stream around the launches
 *
 * Same argument checks and error codes as astcenc_compress_image (ref: Source/astcenc_entry.cpp:1134-1182).
 * Single caller per context (no thread_index rendezvous). */
ASTCENC_PUBLIC enum astcenc_error astcenc_amd_compress_image_device(
	struct astcenc_context* context,
	const void* device_image,
	unsigned int dim_x, unsigned int dim_y,
	enum astcenc_type data_type,
	const struct astcenc_swizzle* swizzle,
	void* device_out, size_t data_len,
	void* hip_stream,
	float* kernel_ms);

/* The same for a volume or 2D array image: dim_z slices of dim_x * dim_y texels back to back at
 * device_volume, compressed with the context's 2D or 3D footprint into blocks in x, y, z raster order
 * (ref: the z loop of compress_image, Source/astcenc_entry.cpp:961-966, and astcenc_image::data[z]).
 * Every slice is compressed from its own data (see ASTCENC_AMD_OPT_PER_SLICE_FAST_LOAD below for the one case in
 * which astcenc_compress_image deliberately does not).
 * The context's progress_callback, if any, is called from the calling thread here; astcenc_compress_image on a
 * context that shards over several devices (ASTCENC_AMD_DEVICES) calls it from library-created threads as well,
 * serialised, with a monotonic percentage. */
ASTCENC_PUBLIC enum astcenc_error astcenc_amd_compress_volume_device(
	struct astcenc_context* context,
	const void* device_volume,
	unsigned int dim_x, unsigned int dim_y, unsigned int dim_z,
	enum astcenc_type data_type,
	const struct astcenc_swizzle* swizzle,
	void* device_out, size_t data_len,
	void* hip_stream,
	float* kernel_ms);

/* Decompress blocks that are resident in device memory into a device image (dim_z slices of tightly
 * packed RGBA rows of data_type, back to back).  Same checks, profiles, output types and swizzles as
 * astcenc_decompress_image (ref: Source/astcenc_entry.cpp:1274-1390); returns when the work on
 * hip_stream (NULL = the context's own stream) has completed. */
ASTCENC_PUBLIC enum astcenc_error astcenc_amd_decompress_image_device(
	struct astcenc_context* context,
	const void* device_blocks, size_t data_len,
	void* device_image,
	unsigned int dim_x, unsigned int dim_y, unsigned int dim_z,
	enum astcenc_type data_type,
	const struct astcenc_swizzle* swizzle,
	void* hip_stream);

/* Error sums of two device-resident images of the same size, the quantities the reference CLI's quality
 * report is made of (ref: compute_error_metrics, Source/astcenccli_error_metrics.cpp:110-300):
 *   PSNR (LDR-RGBA)     = 10 log10(4 texels / (squared_error[0] + .. + [3]))
 *   PSNR (LDR-RGB)      = 10 log10(3 texels / (squared_error[0] + .. + [2]))
 *   alpha-weighted PSNR = the RGBA form over alpha_scaled_squared_error
 * U8 texels are compared as value / 255, F16 / F32 texels clamped to 0..65504, exactly as there.  The
 * HDR figures (mPSNR, log RMSE) come from astcenc_amd_compare_images_hdr_device below. */
struct astcenc_amd_error_sums {
	double squared_error[4];               /* per channel, image1 - image2 */
	double alpha_scaled_squared_error[4];  /* RGB differences scaled by image1's alpha first */
	double rgb_peak;                       /* largest R, G or B value of image1 */
	double texels;
};

ASTCENC_PUBLIC enum astcenc_error astcenc_amd_compare_images_device(
	struct astcenc_context* context,
	const void* device_image1, enum astcenc_type type1,
	const void* device_image2, enum astcenc_type type2,
	unsigned int dim_x, unsigned int dim_y, unsigned int dim_z,
	void* hip_stream,
	struct astcenc_amd_error_sums* sums);

/* The HDR part of the same report (ref: Source/astcenccli_error_metrics.cpp:60-107 mpsnr_operator / mpsnr_sumdiff,
 * :262-268 the log2 terms, :389-403 the printed figures), over the f-stops fstop_lo..fstop_hi (the CLI's
 * -mpsnr option, default -10..10; both within -125..125):
 *   mPSNR (RGB)   = 10 log10(texels * 3 * (fstop_hi - fstop_lo + 1) * 255^2 / (mpsnr_squared_error[0] + [1] + [2]))
 *   LogRMSE (RGB) = sqrt((log2_squared_error[0] + [1] + [2]) / texels)
 *   PSNR (RGB normalised to peak) = PSNR (LDR-RGB) + 20 log10(rgb_peak)
 * log2 is the reference's own polynomial; the tone-mapping power is evaluated in double precision and rounded
 * to float where the reference calls libm's powf. */
struct astcenc_amd_hdr_error_sums {
	double log2_squared_error[4];    /* per channel, log2(image1) - log2(image2) */
	double mpsnr_squared_error[4];   /* per channel, summed over the f-stops */
	int fstop_lo, fstop_hi;
};

ASTCENC_PUBLIC enum astcenc_error astcenc_amd_compare_images_hdr_device(
	struct astcenc_context* context,
	const void* device_image1, enum astcenc_type type1,
	const void* device_image2, enum astcenc_type type2,
	unsigned int dim_x, unsigned int dim_y, unsigned int dim_z,
	int fstop_lo, int fstop_hi,
	void* hip_stream,
	struct astcenc_amd_error_sums* sums,
	struct astcenc_amd_hdr_error_sums* hdr_sums);

/* "hip:gfx950" for the product library. */
ASTCENC_PUBLIC const char* astcenc_amd_backend_name(void);

/* Diagnostics.  The library reports through the reference's error codes and prints nothing.  What it knows beyond the code
 * -- which HIP call failed, why a device of ASTCENC_AMD_DEVICES was skipped -- is handed, one line per event and without a
 * trailing newline, to the callback installed here (process-wide; null, the default, switches it off; the callback may be
 * called from any thread that is inside a library call, and from the library's own threads -- a device's host thread, the
 * worker that waits for a run-time kernel build).  ASTCENC_AMD_LOG=stderr in the environment prints the same
 * lines to stderr when no callback is installed. */
ASTCENC_PUBLIC void astcenc_amd_set_log_callback(void (*callback)(const char* message));

/* Number of GPUs the context shards host images over.  By default astcenc_context_alloc() prepares the calling
 * thread's current device only (one process per GPU is the usual deployment, and a context must not touch its
 * neighbours' GPUs).  With the environment variable ASTCENC_AMD_DEVICES = "all" or a list of ordinals ("0,1,2,3")
 * it prepares those devices, and astcenc_compress_image() / astcenc_decompress_image() then deal contiguous ranges
 * of block rows of the host image to them -- each with its own tables, streams and PCIe pipeline -- and join them,
 * the way the reference deals blocks to its N worker threads (ref: Source/astcenc_entry.cpp:1009-1038, :1340-1385).
 * Buffers that already live on a device (the *_device entry points) are processed on the device that owns them,
 * whatever the list says. */
ASTCENC_PUBLIC int astcenc_amd_context_device_count(const struct astcenc_context* context);

/* Name of the build of the compression kernel the context launches (what a rocprofv3 kernel trace shows, without the
 * namespace).  The library holds generic builds -- "astc_compress_blocks_{ldr,hdr}64" for footprints of at most 64 texels,
 * "astc_compress_blocks_{ldr,hdr}" for the larger ones -- and builds compiled for one context each, whose LDS layout,
 * configuration and table root are compile-time constants: "astc_compress_blocks_ldr_6x6m" (6x6 -medium, LDR),
 * "astc_compress_blocks_ldr_8x8t" (8x8 -thorough, LDR), "astc_compress_blocks_hdr_6x6m" (6x6 -medium, HDR).  A context gets
 * such a build when its records equal the build's byte for byte (default flags and channel weights), the generic one
 * otherwise; both produce the same bytes.  ASTCENC_AMD_KERNEL=generic in the environment keeps every context on the
 * generic builds. */
ASTCENC_PUBLIC const char* astcenc_amd_context_kernel_name(const struct astcenc_context* context);

/* Every other context gets a build of its own at run time: the library carries its device source, writes the context's
 * records as constants, compiles the kernel with hipRTC for the device's architecture on a background thread and keeps the
 * code object on disk (ASTCENC_AMD_CACHE_DIR, else $XDG_CACHE_HOME/astcenc_amd, else ~/.cache/astcenc_amd) under a hash of
 * source, records, options and compiler version; the name is then "astc_compress_blocks_jit_<hash>".  Until that build is
 * there the context runs the generic one -- same bytes.  ASTCENC_AMD_JIT in the environment: "lazy" (default: a cached
 * build is used at once, a compile is started when the context has compressed 2^18 blocks), "eager" (started in
 * astcenc_context_alloc), "sync" (finished inside astcenc_context_alloc), "off".
 *
 * astcenc_amd_context_specialize() waits for the context's specialised build -- starting the compile if need be -- and
 * switches the context to it: ASTCENC_SUCCESS when the context now launches a specialised build (one of the library's own or
 * its run-time build), ASTCENC_ERR_NOT_IMPLEMENTED when it stays generic (no hipRTC library on the box, "off", a compile
 * error: astcenc_amd_set_log_callback says which).  Must not run concurrently with a compression on the same context. */
ASTCENC_PUBLIC enum astcenc_error astcenc_amd_context_specialize(struct astcenc_context* context);

/* Behaviour switches that have no counterpart in the reference API. */
enum astcenc_amd_option {
	/* Multi-slice RGBA8 input (image.dim_z > 1) with a 2D footprint, LDR profile and identity swizzle: the
	 * reference's fast block loader reads slice 0 for every slice (Source/astcenc_image.cpp:304), so it emits
	 * slice 0's blocks dim_z times.  0: do exactly that, byte for byte (the default of astcenc_compress_image, which
	 * promises the reference's bytes).  1: every slice is loaded from its own data, the output equals compressing the
	 * slices one by one (the default of astcenc_amd_compress_volume_device, which has no reference counterpart). */
	ASTCENC_AMD_OPT_PER_SLICE_FAST_LOAD = 1
};

ASTCENC_PUBLIC enum astcenc_error astcenc_amd_context_set_option(
	struct astcenc_context* context,
	enum astcenc_amd_option option,
	int value);

#endif
