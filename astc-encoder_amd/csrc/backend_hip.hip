// SPDX-License-Identifier: Apache-2.0
// HIP backend for gfx950 (MI355X): device-resident tables, staging buffers, and the compression
// kernel launch.  One 64-lane wavefront (one workgroup) compresses one ASTC block; its working
// set is a dynamic-LDS region laid out by make_lds_layout().
//
// Replaces the reference's CPU block loop (compress_image, Source/astcenc_entry.cpp:891-1043):
// instead of threads pulling 16-block tickets from an atomic counter, the block range of a chunk
// is one kernel launch; chunks give the host cancel/progress points.
#include "backend.h"
#include "kernel_jit.h"

#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <sched.h>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

namespace astcd {

/* The host thread that drives a further device of a sharded call (the calling thread drives the first).  Created with
 * the context's slot, parked on a condition variable between calls, joined when the context goes: a call on N devices
 * wakes N - 1 threads instead of creating and joining them (round 5; a 4096^2 image is 26 ms of work per device, a
 * std::thread launch + join some 50-100 us each).  Concurrent calls on one context take turns per worker (`mu`). */
struct SlotWorker {
	std::mutex mu;                    // one task at a time
	std::mutex state_mu;
	std::condition_variable wake, done_cv;
	std::function<void()> task;
	bool has_task = false, done = true, quit = false, bound = false;
	std::thread thread;
	bool started = false;
};

// Everything that lives on one GPU: the table blob, two streams, events and the staging buffers of the
// host-pointer API.  A context owns one slot per device it may run on (see backend_create).
struct DeviceSlot {
	int device;
	uint8_t* d_base;              // device allocation: context records, then the table blob
	uint8_t* d_tab;               // the blob inside it
	uint8_t* d_dectab;            // the decoder's per-footprint tables inside it (after the blob)
	hipStream_t stream;
	hipStream_t copy_stream;      // PCIe traffic of the banded host-pointer path
	hipEvent_t ev0, ev1, ev_copy[2], ev_band, ev_done[3], ev_out[2];
	// pinned staging of the banded host-pointer path: two bands of input in flight, the blocks of two bands on their way back
	uint8_t* h_in[2]; size_t h_in_cap;
	uint8_t* h_out[2]; size_t h_out_cap;
	// staging for the host-pointer API
	void* d_image; size_t image_cap;
	uint8_t* d_out; size_t out_cap;
	float* d_alpha; size_t alpha_cap;   // alpha averages of the a_scale_radius pre-pass
	float* d_alpha_scratch; size_t alpha_scratch_cap;   // padded tiles of the pre-pass when they outgrow LDS (large radii)
	unsigned long long* d_prof;   // stage timers (ASTC_PROFILE builds) / search trace (ASTC_TRACE builds)
	size_t trace_cap;             // bytes at d_prof in ASTC_TRACE builds
	double* d_sums;               // totals of the image comparison kernel
	std::mutex busy;              // one call at a time per slot: the staging buffers and events are shared state
	std::vector<int> local_cpus;  // host CPUs on the device's NUMA node (Linux sysfs); empty: unknown, no binding
	SlotWorker* worker;           // the slot's parked host thread (slots 1.. of a multi-device context), or null
	// the context's run-time specialised build on this device (kernel_jit.h): loaded once it is ready, then launched
	// instead of the library's own generic build
	hipModule_t jit_module;
	hipFunction_t jit_fn;
	bool jit_tried;
};

struct Backend {
	std::vector<DeviceSlot*> slots;   // the devices host images are sharded over; fixed at backend_create (slot 0 = default device)
	std::vector<DeviceSlot*> extra;   // slots created on first use for device pointers that live on other GPUs; never sharded over
	std::mutex slots_mu;              // guards `extra`
	std::vector<uint8_t> full;        // host copy of [LdsLayout][DeviceConfig][table blob][decoder tables], uploaded to every slot
	size_t dectab_offset;             // of the decoder tables in it
	DeviceConfig cfg;
	uint32_t lds_bytes;
	bool hdr;
	TableRoot root;
	int variant;                      // index into kernel_variants (chosen by the first kernel_prepare)
	JitKernel* jit;                   // the context's run-time build (null: none asked for -- a fixed-context build of the library
	                                  // serves this context, ASTCENC_AMD_JIT=off, an instrumentation build, no hipRTC)
	JitMode jit_mode;
	std::atomic<unsigned long long> blocks_done;     // blocks this context has compressed (JIT_LAZY's trigger)
	unsigned long long jit_lazy_blocks;              // ... and the count at which the compile is queued (JIT_LAZY_BLOCKS)
	std::atomic<bool> jit_active;     // some slot launches the run-time build
};

// The library's own device buffers end in a little slack, so that a buffer never stops exactly at the end
// of a mapped page.
constexpr size_t ALLOC_SLACK = 4096;
/* Smallest host image whose PCIe transfers are pipelined against its kernels in bands (compress_on_slot_locked).  A band
 * costs about half a millisecond of its own (a block takes 0.8 ms from load to store, so every kernel ends in a tail that does
 * not fill the device); four bands pay for themselves once the image's transfers take longer than that: measured break-even
 * near 4096 x 4096 at 6x6 (profiles/r06z/host_api_small_images.txt). */
constexpr size_t BAND_MIN_BLOCKS = (size_t)1 << 18;
#if defined(ASTC_TRACE)
constexpr size_t TRACE_WORDS_PER_BLOCK_HOST = 1024;   // = TRACE_WORDS_PER_BLOCK (wave_ctx.h)
#endif

/* Diagnostics.  The library reports through return codes only; what it has to say beyond them -- which HIP call failed,
 * why a device was skipped -- goes to a callback the host application may install (astcenc_amd_set_log_callback,
 * include/astcenc_amd.h), or to stderr when ASTCENC_AMD_LOG=stderr is in the environment.  Nothing is printed otherwise. */
static std::atomic<void (*)(const char*)> g_log_callback{nullptr};
void backend_set_log_callback(void (*cb)(const char*)) { g_log_callback.store(cb); }
static void log_msg(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
static void log_msg(const char* fmt, ...)
{
	void (*cb)(const char*) = g_log_callback.load();
	static const bool to_stderr = []() { const char* e = getenv("ASTCENC_AMD_LOG"); return e && strcmp(e, "stderr") == 0; }();
	if (!cb && !to_stderr) return;
	char line[512];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(line, sizeof(line), fmt, ap);
	va_end(ap);
	if (cb) cb(line);
	else fprintf(stderr, "astcenc_amd: %s\n", line);
}

#define HIP_TRY(expr, fail) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { \
	log_msg("%s -> %s", #expr, hipGetErrorString(e_)); fail; } } while (0)

const char* backend_name() { return "hip:gfx950"; }

namespace {

/* The calling thread's current device is the caller's business (a torch process, an engine): every
 * entry point puts it back on exit. */
struct DeviceGuard {
	int saved; bool ok;
	DeviceGuard() : saved(0), ok(hipGetDevice(&saved) == hipSuccess) {}
	~DeviceGuard() { if (ok) (void)hipSetDevice(saved); }
};

/* The builds of the compression kernel.  Generic ones: LDR / HDR coders x footprints of at most 64 texels (whose texel
 * loops make one trip, kernel_ldr64.hip) / the larger ones.  Fixed-context ones (kernel_ldr_6x6m.hip, ...): compiled for
 * one named context each, its LDS layout, configuration and table root as constants; their `prepare` returns
 * ASTC_PREPARE_NOT_THIS_CONTEXT unless the live context is that one, record for record.  The first variant that accepts
 * the context is used (fixed ones first); ASTCENC_AMD_KERNEL=generic in the environment skips the fixed ones. */
struct KernelVariant {
	const char* name;
	bool fixed, hdr, small;
	int (*prepare)(const TableRoot&, const DeviceConfig&, uint32_t*, void*, uint32_t*);
	int (*launch)(const KernelLaunch&);
};
const KernelVariant kernel_variants[] = {
	{ "astc_compress_blocks_ldr_6x6m", true, false, true, astc_kernel_prepare_ldr_6x6m, astc_kernel_launch_ldr_6x6m },
	{ "astc_compress_blocks_ldr_8x8t", true, false, true, astc_kernel_prepare_ldr_8x8t, astc_kernel_launch_ldr_8x8t },
	{ "astc_compress_blocks_hdr_6x6m", true, true, true, astc_kernel_prepare_hdr_6x6m, astc_kernel_launch_hdr_6x6m },
	{ "astc_compress_blocks_ldr64", false, false, true, astc_kernel_prepare_ldr64, astc_kernel_launch_ldr64 },
	{ "astc_compress_blocks_hdr64", false, true, true, astc_kernel_prepare_hdr64, astc_kernel_launch_hdr64 },
	{ "astc_compress_blocks_ldr", false, false, false, astc_kernel_prepare_ldr, astc_kernel_launch_ldr },
	{ "astc_compress_blocks_hdr", false, true, false, astc_kernel_prepare_hdr, astc_kernel_launch_hdr },
};
/* Picks the context's variant (b->variant < 0: not chosen yet) and prepares it on the current device. */
int kernel_prepare(Backend* b, uint32_t* lds_bytes, void* layout_out, uint32_t* layout_bytes)
{
	if (b->variant >= 0) return kernel_variants[b->variant].prepare(b->root, b->cfg, lds_bytes, layout_out, layout_bytes);
	const bool small = b->root.texel_count <= 64;
	const char* want = getenv("ASTCENC_AMD_KERNEL");
	const bool generic_only = want && strcmp(want, "generic") == 0;
	for (int i = 0; i < (int)(sizeof(kernel_variants) / sizeof(kernel_variants[0])); i++)
	{
		const KernelVariant& v = kernel_variants[i];
		if (v.hdr != b->hdr || v.small != small || (v.fixed && generic_only)) continue;
		const int rc = v.prepare(b->root, b->cfg, lds_bytes, layout_out, layout_bytes);
		if (rc == ASTC_PREPARE_NOT_THIS_CONTEXT) continue;
		if (rc == 0) b->variant = i;
		return rc;
	}
	return (int)hipErrorInvalidDeviceFunction;
}
int kernel_launch(const Backend* b, const DeviceSlot* s, const KernelLaunch& k)
{
	if (s->jit_fn)
	{
		// the run-time build: same parameters as the library's builds (kernel_device.h), launched through the module API
		KernelLaunch a = k;
		void* args[] = { &a.d_tab, &a.img, &a.d_out, &a.first, &a.count, &a.d_prof };
		return (int)hipModuleLaunchKernel(s->jit_fn, k.count, 1, 1, 64, 1, 1, k.lds_bytes, static_cast<hipStream_t>(k.stream), args, nullptr);
	}
	return kernel_variants[b->variant].launch(k);
}

/* The self-check of slot_adopt_jit: 16 x 16 blocks (volumes: 8 x 8 x 4) of deterministic RGBA8 content through `fn` and through
 * the library's build of the context; true when the two block streams are byte-identical. */
bool jit_self_check(Backend* b, DeviceSlot* s, hipFunction_t fn)
{
	const uint32_t bsx = b->root.dim_x, bsy = b->root.dim_y, bsz = b->root.dim_z;
	const uint32_t nbx = bsz > 1 ? 8u : 16u, nby = bsz > 1 ? 8u : 16u, nbz = bsz > 1 ? 4u : 1u;
	const uint32_t dim_x = nbx * bsx - 1u, dim_y = nby * bsy - 1u, dim_z = bsz > 1 ? nbz * bsz - 1u : 1u;      // (the last blocks are partial)
	const size_t texels = (size_t)dim_x * dim_y * dim_z, nblocks = (size_t)nbx * nby * nbz;
	std::vector<uint8_t> img(texels * 4);
	uint32_t rng = 0x9E3779B1u;
	for (uint32_t z = 0; z < dim_z; z++)
		for (uint32_t y = 0; y < dim_y; y++)
			for (uint32_t x = 0; x < dim_x; x++)
			{
				// per block-sized tile: noise amplitude 0 (ramps), 6, 40, 255 (pure noise), two colours, a constant
				const uint32_t tile = (x / bsx + 3u * (y / bsy) + 5u * (z / bsz)) % 6u;
				uint8_t* px = &img[(((size_t)z * dim_y + y) * dim_x + x) * 4];
				for (int ch = 0; ch < 4; ch++)
				{
					rng = rng * 1664525u + 1013904223u;
					const int noise = (int)(rng >> 24);
					int v = (int)((x * (3u + ch) + y * (7u - ch) + z * 11u) & 255u);
					if (tile == 1) v += (noise & 15) - 6;
					else if (tile == 2) v += (noise & 63) - 24;
					else if (tile == 3) v = noise;
					else if (tile == 4) v = ((x + ch) ^ (y >> 1)) & 2 ? 220 - 20 * ch : 30 + 25 * ch;
					else if (tile == 5) v = 40 + 50 * ch;
					if (ch == 3 && tile != 3 && tile != 2) v = 255;
					px[ch] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
				}
			}
	void* d_img = nullptr;
	uint8_t* d_out = nullptr;
	bool ok = hipMalloc(&d_img, img.size() + ALLOC_SLACK) == hipSuccess && hipMalloc(reinterpret_cast<void**>(&d_out), 2 * nblocks * 16 + ALLOC_SLACK) == hipSuccess;
	std::vector<uint8_t> out(2 * nblocks * 16);
	if (ok)
	{
		KernelLaunch k;
		k.d_tab = s->d_tab; k.lds_bytes = b->lds_bytes; k.d_prof = nullptr; k.stream = s->stream;
		k.first = 0; k.count = (uint32_t)nblocks;
		ImageDesc& im = k.img;
		im.data = d_img; im.dim_x = dim_x; im.dim_y = dim_y; im.data_type = 0;
		for (int i = 0; i < 4; i++) im.swz[i] = (uint32_t)i;
		im.blocks_x = nbx; im.blocks_y = nby; im.dim_z = dim_z; im.blocks_z = nbz;
		im.use_fast_load = (b->cfg.profile < 2 && bsz == 1) ? 1 : 0;
		im.fast_load_slice0 = 0; im.alpha_avg = nullptr; im.a_scale_radius = 0;
		ok = hipMemcpyAsync(d_img, img.data(), img.size(), hipMemcpyHostToDevice, s->stream) == hipSuccess;
		k.d_out = d_out;
		ok = ok && kernel_variants[b->variant].launch(k) == 0;
		k.d_out = d_out + nblocks * 16;
		void* args[] = { &k.d_tab, &k.img, &k.d_out, &k.first, &k.count, &k.d_prof };
		ok = ok && hipModuleLaunchKernel(fn, k.count, 1, 1, 64, 1, 1, k.lds_bytes, s->stream, args, nullptr) == hipSuccess;
		ok = ok && hipMemcpyAsync(out.data(), d_out, out.size(), hipMemcpyDeviceToHost, s->stream) == hipSuccess;
		ok = ok && hipStreamSynchronize(s->stream) == hipSuccess;
		if (!ok) (void)hipGetLastError();
	}
	if (d_img) (void)hipFree(d_img);
	if (d_out) (void)hipFree(d_out);
	return ok && memcmp(out.data(), out.data() + nblocks * 16, nblocks * 16) == 0;
}

/* Loads the context's run-time build on the slot's device once the compiler has delivered it (current device = the slot's).
 * Anything that goes wrong leaves the slot on the library's generic build. */
void slot_adopt_jit(Backend* b, DeviceSlot* s)
{
	if (!b->jit || s->jit_tried || jit_state(b->jit) != JIT_READY) return;
	s->jit_tried = true;
	size_t bytes = 0;
	const void* code = jit_code(b->jit, &bytes);
	hipModule_t mod = nullptr;
	hipFunction_t fn = nullptr;
	if (hipModuleLoadData(&mod, code) != hipSuccess || hipModuleGetFunction(&fn, mod, JIT_ENTRY_POINT) != hipSuccess)
	{
		(void)hipGetLastError();
		if (mod) (void)hipModuleUnload(mod);
		log_msg("run-time build %s does not load on device %d: the generic build stays", jit_kernel_name(b->jit), s->device);
		return;
	}
	// Trust, but verify: before the build takes over, it and the library's own build compress the same 256 blocks of built-in
	// content -- noise of several amplitudes over ramps, flat and two-colour stretches, i.e. blocks that run every trial of the
	// search -- and the bytes must be equal.  (A build is compiled from the same source with the same numerics flags, so they
	// are -- unless the compiler did something with the constants that it does not do without them: the run-time builds of the
	// 10x8 and 12x12 footprints do differ, DESIGN.md section 3.1, and are turned away here.)
	// (ASTCENC_AMD_JIT_SELF_CHECK=0: debugging only -- tools/jit_debug4.py looks at a build the check turns away)
	const char* check = getenv("ASTCENC_AMD_JIT_SELF_CHECK");
	if (!(check && strcmp(check, "0") == 0) && !jit_self_check(b, s, fn))
	{
		(void)hipModuleUnload(mod);
		log_msg("run-time build %s does not reproduce the generic build's bytes on the self-check image: the generic build stays", jit_kernel_name(b->jit));
		return;
	}
	s->jit_module = mod;
	s->jit_fn = fn;
	b->jit_active.store(true);
}

void worker_stop(SlotWorker* w)
{
	if (!w) return;
	if (w->started)
	{
		{ std::lock_guard<std::mutex> g(w->state_mu); w->quit = true; }
		w->wake.notify_one();
		w->thread.join();
	}
	delete w;
}

void slot_destroy(DeviceSlot* s)
{
	if (!s) return;
	worker_stop(s->worker);
	s->worker = nullptr;
	(void)hipSetDevice(s->device);
	if (s->d_image) (void)hipFree(s->d_image);
	if (s->d_out) (void)hipFree(s->d_out);
	if (s->d_alpha) (void)hipFree(s->d_alpha);
	if (s->d_alpha_scratch) (void)hipFree(s->d_alpha_scratch);
	if (s->d_sums) (void)hipFree(s->d_sums);
	if (s->d_prof) (void)hipFree(s->d_prof);
	for (int i = 0; i < 2; i++) { if (s->h_in[i]) (void)hipHostFree(s->h_in[i]); if (s->h_out[i]) (void)hipHostFree(s->h_out[i]); }
	for (hipEvent_t e : { s->ev_copy[0], s->ev_copy[1], s->ev_band, s->ev_done[0], s->ev_done[1], s->ev_done[2], s->ev_out[0], s->ev_out[1], s->ev0, s->ev1 })
		if (e) (void)hipEventDestroy(e);
	if (s->copy_stream) (void)hipStreamDestroy(s->copy_stream);
	if (s->stream) (void)hipStreamDestroy(s->stream);
	if (s->jit_module) (void)hipModuleUnload(s->jit_module);
	if (s->d_base) (void)hipFree(s->d_base);
	delete s;
}

/* The host CPUs next to `device` (Linux: /sys/bus/pci/devices/<bus id>/local_cpulist, e.g. "0-31,128-159").  A shard's
 * worker thread runs on them (bind_worker_to_device), so the pinned staging buffers it allocates on first use -- pinned
 * pages are placed by first touch -- and the memcpy into them stay on the device's NUMA node: on an 8-GPU node the
 * other placement sends every band across the inter-socket link twice.  Empty when the platform does not say. */
static std::vector<int> device_local_cpus(int device)
{
	std::vector<int> cpus;
	char bus[64] = { 0 };
	if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), device) != hipSuccess) return cpus;
	for (char* p = bus; *p; p++) if (*p >= 'A' && *p <= 'F') *p = (char)(*p - 'A' + 'a');
	char path[160];
	snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/local_cpulist", bus);
	FILE* f = fopen(path, "r");
	if (!f) return cpus;
	char line[1024] = { 0 };
	if (fgets(line, (int)sizeof(line), f))
	{
		for (char* p = line; *p && *p != '\n'; )
		{
			char* end = nullptr;
			long a = strtol(p, &end, 10);
			if (end == p) break;
			long bnd = a;
			p = end;
			if (*p == '-') { bnd = strtol(p + 1, &end, 10); p = end; }
			for (long c = a; c <= bnd && c < 4096; c++) cpus.push_back((int)c);
			if (*p == ',') p++;
		}
	}
	fclose(f);
	return cpus;
}

/* Called on a shard's worker thread (never on the caller's own thread: its affinity is the caller's business).
 * ASTCENC_AMD_NUMA_BIND=0 switches it off. */
static void bind_worker_to_device(const DeviceSlot* s)
{
	static const bool enabled = []() { const char* e = getenv("ASTCENC_AMD_NUMA_BIND"); return !(e && e[0] == '0'); }();
	if (!enabled || s->local_cpus.empty()) return;
	cpu_set_t allowed, want;
	CPU_ZERO(&want);
	if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return;
	int n = 0;
	for (int c : s->local_cpus) if (c < CPU_SETSIZE && CPU_ISSET(c, &allowed)) { CPU_SET(c, &want); n++; }
	if (n > 0) (void)sched_setaffinity(0, sizeof(want), &want);        // (best effort: a container may forbid it)
}

/* Hands `task` to the slot's parked thread (started on first use, bound to the device's CPUs once); false when no
 * thread can be had -- std::system_error must not cross the C ABI --, or when the thread is busy with another call on
 * the same context (a second call used to wait here for the first call's whole shard before it even started its
 * own first one): a compression then deals that device's share to the others, a decompression runs the shard itself.  The
 * worker stays locked (w->mu) until worker_wait(). */
static bool worker_run(DeviceSlot* s, std::function<void()> task)
{
	SlotWorker* w = s->worker;
	if (!w) return false;
	if (!w->mu.try_lock()) return false;
	if (!w->started)
	{
		try
		{
			w->thread = std::thread([w, s]()
			{
				bind_worker_to_device(s);
				std::unique_lock<std::mutex> lk(w->state_mu);
				for (;;)
				{
					w->wake.wait(lk, [w]() { return w->has_task || w->quit; });
					if (w->quit) return;
					std::function<void()> t = std::move(w->task);
					w->has_task = false;
					lk.unlock();
					t();
					lk.lock();
					w->done = true;
					w->done_cv.notify_all();
				}
			});
			w->started = true;
		}
		catch (...) { w->mu.unlock(); return false; }
	}
	{
		std::lock_guard<std::mutex> g(w->state_mu);
		w->task = std::move(task);
		w->has_task = true;
		w->done = false;
	}
	w->wake.notify_one();
	return true;
}
static void worker_wait(DeviceSlot* s)
{
	SlotWorker* w = s->worker;
	{
		std::unique_lock<std::mutex> lk(w->state_mu);
		w->done_cv.wait(lk, [w]() { return w->done; });
	}
	w->mu.unlock();
}

/* One slot on `device`: uploads the tables, sets the kernels' dynamic-LDS attribute there, creates streams
 * and events.  Every failure leaves through slot_destroy (the record starts zeroed). status: 1 = out of
 * memory, 2 = anything else. */
DeviceSlot* slot_create(Backend* b, int device, int* status)
{
	DeviceSlot* s = new DeviceSlot();
	s->device = device;
	s->d_base = nullptr; s->d_tab = nullptr; s->stream = nullptr; s->copy_stream = nullptr;
	s->ev0 = s->ev1 = s->ev_copy[0] = s->ev_copy[1] = s->ev_band = nullptr;
	s->ev_done[0] = s->ev_done[1] = s->ev_done[2] = nullptr;
	s->ev_out[0] = s->ev_out[1] = nullptr;
	s->h_in[0] = s->h_in[1] = nullptr; s->h_in_cap = 0; s->h_out[0] = s->h_out[1] = nullptr; s->h_out_cap = 0;
	s->d_image = nullptr; s->image_cap = 0; s->d_out = nullptr; s->out_cap = 0; s->d_alpha = nullptr; s->alpha_cap = 0;
	s->d_alpha_scratch = nullptr; s->alpha_scratch_cap = 0;
	s->d_prof = nullptr; s->trace_cap = 0; s->d_sums = nullptr;
	s->worker = nullptr;
	s->jit_module = nullptr; s->jit_fn = nullptr; s->jit_tried = false;
#define SLOT_TRY(expr, code) HIP_TRY(expr, { slot_destroy(s); *status = code; return nullptr; })
	SLOT_TRY(hipSetDevice(device), 2);
	{
		uint8_t layout[CTX_LAYOUT_BACK - CTX_CONFIG_BACK]; uint32_t layout_bytes = 0, lds_bytes = 0;
		int prc = kernel_prepare(b, &lds_bytes, layout, &layout_bytes);
		if (prc != 0)
		{
			log_msg("kernel setup failed on device %d (hip error %d)", device, prc);
			slot_destroy(s); *status = 2; return nullptr;
		}
	}
	SLOT_TRY(hipMalloc(&s->d_base, b->full.size() + ALLOC_SLACK), 1);
	SLOT_TRY(hipMemcpy(s->d_base, b->full.data(), b->full.size(), hipMemcpyHostToDevice), 2);
	s->d_tab = s->d_base + CTX_LAYOUT_BACK;
	s->d_dectab = s->d_base + b->dectab_offset;
	SLOT_TRY(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking), 2);
	SLOT_TRY(hipStreamCreateWithFlags(&s->copy_stream, hipStreamNonBlocking), 2);
	SLOT_TRY(hipEventCreateWithFlags(&s->ev_copy[0], hipEventDisableTiming), 2);
	SLOT_TRY(hipEventCreateWithFlags(&s->ev_copy[1], hipEventDisableTiming), 2);
	SLOT_TRY(hipEventCreateWithFlags(&s->ev_band, hipEventDisableTiming), 2);
	for (int i = 0; i < 3; i++) SLOT_TRY(hipEventCreateWithFlags(&s->ev_done[i], hipEventDisableTiming), 2);
	for (int i = 0; i < 2; i++) SLOT_TRY(hipEventCreateWithFlags(&s->ev_out[i], hipEventDisableTiming), 2);
	SLOT_TRY(hipEventCreate(&s->ev0), 2);
	SLOT_TRY(hipEventCreate(&s->ev1), 2);
#if defined(ASTC_PROFILE)
	enum { PS_COUNT = 40 };
	SLOT_TRY(hipMalloc(&s->d_prof, 2 * PS_COUNT * sizeof(unsigned long long)), 1);
	SLOT_TRY(hipMemset(s->d_prof, 0, 2 * PS_COUNT * sizeof(unsigned long long)), 2);
#endif
#undef SLOT_TRY
	s->local_cpus = device_local_cpus(device);
	slot_adopt_jit(b, s);      // (a slot created after the build arrived: devices joined on first use)
	*status = 0;
	return s;
}

/* Device list of a context.  Default: the calling thread's current device only (a rank-per-GPU process must not
 * initialise, or contend for, its neighbours' GPUs).  The N-device split of host images is opt-in:
 * ASTCENC_AMD_DEVICES = "all" (every visible device, the current one first) or a comma separated list of device
 * ordinals; an ordinal may repeat ("0,0": two slots on one GPU, which is how the multi-device path is tested on a
 * one-GPU box). */
std::vector<int> device_list(int ndev)
{
	std::vector<int> out;
	const char* env = getenv("ASTCENC_AMD_DEVICES");
	const bool all = env && strcmp(env, "all") == 0;
	if (env && *env && !all)
	{
		const char* p = env;
		while (*p)
		{
			char* e = nullptr;
			long v = strtol(p, &e, 10);
			if (e == p) break;
			if (v >= 0 && v < ndev) out.push_back((int)v);
			else log_msg("ASTCENC_AMD_DEVICES names device %ld, %d visible; ignored", v, ndev);
			p = e;
			while (*p == ',' || *p == ' ') p++;
		}
	}
	if (out.empty())
	{
		// current device first, so that a one-device caller keeps what it had selected
		int cur = 0;
		if (hipGetDevice(&cur) != hipSuccess) cur = 0;
		out.push_back(cur);
		if (all) for (int d = 0; d < ndev; d++) if (d != cur) out.push_back(d);
	}
	return out;
}

/* Slot that owns `ptr` (device memory); slot 0 when the runtime does not know the pointer. */
DeviceSlot* slot_for_pointer(Backend* b, const void* ptr, int* status)
{
	*status = 0;
	hipPointerAttribute_t attr;
	memset(&attr, 0, sizeof(attr));
	if (!ptr || hipPointerGetAttributes(&attr, ptr) != hipSuccess)
	{
		(void)hipGetLastError();
		return b->slots[0];
	}
	for (DeviceSlot* s : b->slots) if (s->device == attr.device) return s;     // (immutable after backend_create: no lock)
	std::lock_guard<std::mutex> lk(b->slots_mu);
	for (DeviceSlot* s : b->extra) if (s->device == attr.device) return s;
	DeviceSlot* s = slot_create(b, attr.device, status);
	if (s) b->extra.push_back(s);
	return s;
}

/* The stream a device-resident call runs on: the caller's, which must belong to the device the buffers (and the
 * slot's tables) live on, or the slot's own.  Returns false (-> rc 3, a bad argument) on a foreign stream. */
bool pick_stream(const DeviceSlot* s, void* caller_stream, hipStream_t* out)
{
	*out = s->stream;
	if (!caller_stream) return true;
	hipStream_t stream = static_cast<hipStream_t>(caller_stream);
	hipDevice_t sdev = -1;
	if (hipStreamGetDevice(stream, &sdev) != hipSuccess) { (void)hipGetLastError(); sdev = s->device; }
	if ((int)sdev != s->device)
	{
		log_msg("the stream belongs to device %d, the buffers to device %d", (int)sdev, s->device);
		return false;
	}
	*out = stream;
	return true;
}

/* Completed-block counter shared by the slots of one call; the callback sees a monotonic percentage
 * (ref: ParallelManager::complete_task_assignment, astcenc_internal_entry.h:255-290). */
struct Progress {
	std::mutex mu;
	size_t done, total;
	void (*callback)(float);
	void add(size_t n)
	{
		if (!callback) return;
		std::lock_guard<std::mutex> lk(mu);
		done += n;
		callback(100.0f * (float)done / (float)total);
	}
};

} // namespace

Backend* backend_create(const uint8_t* blob, size_t blob_bytes, const DeviceConfig& cfg, int* status)
{
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
	{
		log_msg("no HIP device available; this library has no CPU fallback");
		*status = 2;
		return nullptr;
	}
	DeviceGuard guard;

	Backend* b = new Backend;
	b->cfg = cfg;
	memcpy(&b->root, blob, sizeof(TableRoot));
	b->hdr = cfg.profile >= 2;
	b->lds_bytes = 0;
	b->variant = -1;
	b->jit = nullptr; b->jit_mode = JIT_OFF; b->blocks_done.store(0); b->jit_active.store(false);
	b->jit_lazy_blocks = JIT_LAZY_BLOCKS;
	// (ASTCENC_AMD_JIT_LAZY_BLOCKS: another trigger count -- tests/test_jit.py watches the background compile take over)
	if (const char* e = getenv("ASTCENC_AMD_JIT_LAZY_BLOCKS")) { const long long v = strtoll(e, nullptr, 10); if (v >= 1) b->jit_lazy_blocks = (unsigned long long)v; }
	uint8_t layout[CTX_LAYOUT_BACK - CTX_CONFIG_BACK];
	uint32_t layout_bytes = 0;
	{
		// layout record of the block's LDS working set (the kernels' dynamic-LDS attribute is per device: slot_create)
		int prc = kernel_prepare(b, &b->lds_bytes, layout, &layout_bytes);
		if (prc != 0)
		{
			log_msg("kernel setup failed (hip error %d)", prc);
			delete b; *status = 2; return nullptr;
		}
	}
	if (b->lds_bytes > 160 * 1024)
	{
		log_msg("block working set %u B; a CU has 160 KiB of LDS", b->lds_bytes);
		delete b; *status = 2; return nullptr;
	}

	// device allocation = [LdsLayout, 256 B][DeviceConfig, 256 B][table blob][decoder tables]; kernels get the blob pointer
	b->dectab_offset = (CTX_LAYOUT_BACK + blob_bytes + 255) & ~(size_t)255;
	b->full.assign(b->dectab_offset + astc_decode_tables_bytes(), 0);
	astc_decode_tables_build(b->full.data() + b->dectab_offset, b->root.dim_x, b->root.dim_y, b->root.dim_z);
	memcpy(b->full.data(), layout, layout_bytes);
	static_assert(sizeof(DeviceConfig) <= 256, "DeviceConfig outgrew its slot");
	memcpy(b->full.data() + (CTX_LAYOUT_BACK - CTX_CONFIG_BACK), &b->cfg, sizeof(DeviceConfig));
	memcpy(b->full.data() + CTX_LAYOUT_BACK, blob, blob_bytes);

#if !defined(ASTC_TRACE) && !defined(ASTC_DUPSTAGE) && !defined(ASTC_PROFILE) && !defined(ASTC_LDS_PAD_ENV)
	// A context that none of the library's fixed-context builds serves gets its own (kernel_jit.h): found in the disk cache,
	// or compiled in the background -- by default once the context has shown that it is used for more than a thumbnail.
	{
		const char* want = getenv("ASTCENC_AMD_KERNEL");
		b->jit_mode = want && strcmp(want, "generic") == 0 ? JIT_OFF : jit_mode_from_environment();
		if (b->jit_mode != JIT_OFF && !kernel_variants[b->variant].fixed)
		{
			int dev = 0;
			hipDeviceProp_t prop;
			if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
				b->jit = jit_acquire(layout, layout_bytes, b->cfg, b->root, b->hdr, prop.gcnArchName, [](const char* line) { log_msg("%s", line); });
			if (b->jit && b->jit_mode == JIT_EAGER) jit_start(b->jit);
			if (b->jit && b->jit_mode == JIT_SYNC) (void)jit_wait(b->jit);
		}
	}
#endif
	for (int device : device_list(ndev))
	{
		int st = 0;
		DeviceSlot* s = slot_create(b, device, &st);
		if (!s)
		{
			if (b->slots.empty()) { jit_release(b->jit); delete b; *status = st; return nullptr; }
			log_msg("device %d not usable, continuing with %zu device(s)", device, b->slots.size());
			continue;
		}
		// (every device but the first gets its parked host thread: the record here, the thread itself on first use)
		if (!b->slots.empty()) s->worker = new (std::nothrow) SlotWorker();
		b->slots.push_back(s);
	}
	*status = 0;
	return b;
}

void backend_destroy(Backend* b)
{
	if (!b) return;
	DeviceGuard guard;
	for (DeviceSlot* s : b->slots) slot_destroy(s);
	for (DeviceSlot* s : b->extra) slot_destroy(s);
	jit_release(b->jit);
	delete b;
}

int backend_device_count(const Backend* b) { return (int)b->slots.size(); }
const char* backend_kernel_name(const Backend* b)
{
	if (b->jit && b->jit_active.load()) return jit_kernel_name(b->jit);
	return b->variant >= 0 ? kernel_variants[b->variant].name : "";
}

int backend_specialize(Backend* b)
{
	if (b->variant >= 0 && kernel_variants[b->variant].fixed) return 0;
	if (!b->jit || jit_wait(b->jit) != JIT_READY) return 1;
	DeviceGuard guard;
	for (DeviceSlot* s : b->slots)
	{
		std::lock_guard<std::mutex> busy(s->busy);
		if (hipSetDevice(s->device) == hipSuccess) slot_adopt_jit(b, s);
	}
	return b->jit_active.load() ? 0 : 1;
}

static int compress_on_slot_locked(Backend* b, DeviceSlot* s, const CompressJob& job, Progress* progress);

/* The blocks of `job` on one slot.  Returns 0 ok, 1 out of memory, 2 device failure, 3 bad argument. */
static int compress_on_slot(Backend* b, DeviceSlot* s, const CompressJob& job, Progress* progress)
{
	std::lock_guard<std::mutex> busy(s->busy);
	HIP_TRY(hipSetDevice(s->device), return 2);
	const int rc = compress_on_slot_locked(b, s, job, progress);
	if (rc != 0)
	{
		// A failed call may leave asynchronous copies of the banded pipeline in flight that read or write the pinned
		// staging buffers; the next call refills (or regrows and frees) those buffers without waiting for band 0 and 1.
		// Drain both streams before anybody can get there (best effort: the device may be the thing that failed).
		(void)hipStreamSynchronize(s->copy_stream);
		(void)hipStreamSynchronize(s->stream);
	}
	return rc;
}

static int compress_on_slot_locked(Backend* b, DeviceSlot* s, const CompressJob& job, Progress* progress)
{

	const uint32_t bsx = b->root.dim_x, bsy = b->root.dim_y, bsz = b->root.dim_z;
	const uint32_t dim_z = job.dim_z ? job.dim_z : 1u;
	const uint32_t blocks_x = (job.dim_x + bsx - 1) / bsx;
	const uint32_t blocks_y = (job.dim_y + bsy - 1) / bsy;
	const uint32_t blocks_z = (dim_z + bsz - 1) / bsz;
	const size_t nblocks = (size_t)blocks_x * blocks_y * blocks_z;
	// the context's run-time build: adopted as soon as the compiler has delivered it; asked for (JIT_LAZY) once the context
	// has compressed enough to be worth a compile
	slot_adopt_jit(b, s);
	if (b->jit && b->jit_mode == JIT_LAZY && b->blocks_done.fetch_add(nblocks) + nblocks >= b->jit_lazy_blocks) jit_start(b->jit);
	const size_t texel_bytes = job.data_type == 0 ? 4 : job.data_type == 1 ? 8 : 16;
	// (a shard of the alpha-scale split carries halo rows around its own: they are uploaded and averaged, not compressed)
	const uint32_t halo_above = dim_z == 1 && job.a_scale_radius != 0 ? job.halo_above : 0u;
	const uint32_t halo_below = dim_z == 1 && job.a_scale_radius != 0 ? job.halo_below : 0u;
	const uint32_t rows_with_halo = job.dim_y + halo_above + halo_below;
	const size_t slice_bytes = (size_t)job.dim_x * rows_with_halo * texel_bytes;
	const size_t image_bytes = slice_bytes * dim_z;
	const size_t out_bytes = nblocks * 16;

	hipStream_t stream;
	if (!pick_stream(s, job.stream, &stream)) return 3;

	const void* d_image = job.device_data;
	uint8_t* d_out = job.device_out;

	if (job.host_slices)
	{
		if (s->image_cap < image_bytes)
		{
			if (s->d_image) (void)hipFree(s->d_image);
			s->d_image = nullptr; s->image_cap = 0;
			HIP_TRY(hipMalloc(&s->d_image, image_bytes + ALLOC_SLACK), return 1);
			s->image_cap = image_bytes;
		}
		d_image = s->d_image;
	}
	// Host-pointer calls on a plain 2D image are pipelined by bands of block rows: band k+1 travels over
	// PCIe on the copy stream while band k is being compressed, and band k's blocks travel back while band
	// k+1 runs (a block only reads texel rows of its own band).  The alpha-scale pre-pass and volumes need
	// the whole image first.
	const bool banded = job.host_slices && job.host_out && dim_z == 1 && job.a_scale_radius == 0;
	if (job.host_slices && !banded)
	{
		for (uint32_t z = 0; z < dim_z; z++)
			HIP_TRY(hipMemcpyAsync(static_cast<uint8_t*>(s->d_image) + z * slice_bytes, job.host_slices[z], slice_bytes, hipMemcpyHostToDevice, stream), return 2);
	}
	if (job.host_out)
	{
		if (s->out_cap < out_bytes)
		{
			if (s->d_out) (void)hipFree(s->d_out);
			s->d_out = nullptr; s->out_cap = 0;
			HIP_TRY(hipMalloc(&s->d_out, out_bytes + ALLOC_SLACK), return 1);
			s->out_cap = out_bytes;
		}
		d_out = s->d_out;
	}
	if (!d_image || !d_out) return 2;

	ImageDesc img;
	img.data = static_cast<const uint8_t*>(d_image) + (size_t)halo_above * job.dim_x * texel_bytes;
	img.dim_x = job.dim_x; img.dim_y = job.dim_y;
	img.data_type = job.data_type;
	for (int i = 0; i < 4; i++) img.swz[i] = job.swz[i];
	img.blocks_x = blocks_x; img.blocks_y = blocks_y;
	img.dim_z = dim_z; img.blocks_z = blocks_z;
	bool needs_swz = job.swz[0] != 0 || job.swz[1] != 1 || job.swz[2] != 2 || job.swz[3] != 3;
	bool hdr = b->cfg.profile >= 2;
	img.use_fast_load = (!needs_swz && !hdr && job.data_type == 0 && bsz == 1) ? 1 : 0;   // ref: astcenc_entry.cpp:946
	img.fast_load_slice0 = job.fast_load_slice0;
	img.alpha_avg = nullptr;
	img.a_scale_radius = job.a_scale_radius;
	if (job.a_scale_radius != 0)
	{
		const size_t need = (size_t)job.dim_x * rows_with_halo * sizeof(float);
		if (s->alpha_cap < need)
		{
			if (s->d_alpha) (void)hipFree(s->d_alpha);
			s->d_alpha = nullptr; s->alpha_cap = 0;
			HIP_TRY(hipMalloc(&s->d_alpha, need + ALLOC_SLACK), return 1);
			s->alpha_cap = need;
		}
		AlphaLaunch a;
		a.d_image = d_image; a.d_averages = s->d_alpha;
		a.dim_x = job.dim_x; a.dim_y = rows_with_halo; a.dim_z = dim_z; a.data_type = job.data_type;
		a.swz_a = job.swz[3]; a.radius = job.a_scale_radius; a.stream = stream;
		a.d_scratch = nullptr; a.scratch_workgroups = 0;
		const size_t scratch = astc_alpha_scratch_bytes(job.dim_x, rows_with_halo, dim_z, job.a_scale_radius, &a.scratch_workgroups);
		if (scratch == (size_t)-1)
		{
			log_msg("a_scale_radius %u needs more than 1 GiB of pre-pass scratch per tile: refused", job.a_scale_radius);
			return 1;
		}
		if (scratch)
		{
			if (s->alpha_scratch_cap < scratch)
			{
				if (s->d_alpha_scratch) (void)hipFree(s->d_alpha_scratch);
				s->d_alpha_scratch = nullptr; s->alpha_scratch_cap = 0;
				HIP_TRY(hipMalloc(&s->d_alpha_scratch, scratch + ALLOC_SLACK), return 1);
				s->alpha_scratch_cap = scratch;
			}
			a.d_scratch = s->d_alpha_scratch;
		}
		int arc = astc_alpha_launch(a);
		if (arc != 0) { log_msg("alpha pre-pass launch failed (hip error %d)", arc); return 2; }
		img.alpha_avg = s->d_alpha + (size_t)halo_above * job.dim_x;
		// (a large scratch is not kept for the life of the context: the pre-pass runs once per call, its scratch goes back
		//  as soon as the stream is past the kernel -- hipFree waits for that)
		if (s->alpha_scratch_cap > ((size_t)64 << 20))
		{
			(void)hipStreamSynchronize(stream);
			(void)hipFree(s->d_alpha_scratch);
			s->d_alpha_scratch = nullptr; s->alpha_scratch_cap = 0;
		}
	}

	// Chunks bound the time between cancel checks / progress callbacks on huge images; a chunk is
	// still tens of thousands of workgroups, far more than the 256 CUs need to stay full.
	const bool chunked = (progress && progress->callback) || job.host_slices;
	size_t chunk = chunked ? (size_t)1 << 18 : nblocks;
	if (banded)
	{
		// whole block rows per band; at least four bands when the image is large enough for a band to fill the device.  (A
		// block takes 0.8 ms from load to store whatever runs beside it -- its own instruction stream on one wave -- so a
		// kernel over a band is never shorter than that: four bands of a 256 x 256 image were four times 0.8 ms one after
		// the other, 3.4 ms for a call whose one launch takes 0.9; below BAND_MIN_BLOCKS the image is one band.)
		size_t rows = chunk / blocks_x;
		if (rows * 4 > blocks_y) rows = nblocks >= BAND_MIN_BLOCKS ? (blocks_y + 3) / 4 : blocks_y;
		if (rows < 1) rows = 1;
		chunk = rows * blocks_x;
	}
	const size_t row_bytes = (size_t)job.dim_x * texel_bytes;
	if (banded)
	{
		// Pinned staging: a copy from or to the caller's pageable memory would hold the host thread until it is done (and the
		// D2H one until the band's kernel is done), so the device would sit idle while the next band is queued.  The host
		// copies a band into a pinned buffer (two of them: band k+1 is staged while band k's transfer is in flight) and the
		// blocks come back through pinned buffers as well; every transfer is then a true asynchronous DMA.
		const size_t band_rows = (chunk / blocks_x) * bsy;
		const size_t in_need = band_rows * row_bytes, out_need = chunk * 16;
		if (s->h_in_cap < in_need)
		{
			for (int i = 0; i < 2; i++) { if (s->h_in[i]) (void)hipHostFree(s->h_in[i]); s->h_in[i] = nullptr; }
			s->h_in_cap = 0;
			for (int i = 0; i < 2; i++) HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&s->h_in[i]), in_need, hipHostMallocDefault), return 1);
			s->h_in_cap = in_need;
		}
		if (s->h_out_cap < out_need)
		{
			for (int i = 0; i < 2; i++) { if (s->h_out[i]) (void)hipHostFree(s->h_out[i]); s->h_out[i] = nullptr; }
			s->h_out_cap = 0;
			for (int i = 0; i < 2; i++) HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&s->h_out[i]), out_need, hipHostMallocDefault), return 1);
			s->h_out_cap = out_need;
		}
	}
	auto upload_band = [&](size_t band_first, size_t band_blocks) -> int {
		const size_t band = band_first / chunk;
		const size_t y0 = (band_first / blocks_x) * bsy;
		size_t y1 = ((band_first + band_blocks) / blocks_x) * bsy;
		if (y1 > job.dim_y) y1 = job.dim_y;
		// the staging buffer's previous transfer (band - 2) must have left it
		if (band >= 2) HIP_TRY(hipEventSynchronize(s->ev_copy[band & 1]), return 2);
		memcpy(s->h_in[band & 1], static_cast<const uint8_t*>(job.host_slices[0]) + y0 * row_bytes, (y1 - y0) * row_bytes);
		HIP_TRY(hipMemcpyAsync(static_cast<uint8_t*>(s->d_image) + y0 * row_bytes, s->h_in[band & 1], (y1 - y0) * row_bytes, hipMemcpyHostToDevice, s->copy_stream), return 2);
		HIP_TRY(hipEventRecord(s->ev_copy[band & 1], s->copy_stream), return 2);
		return 0;
	};
	// a band's blocks, once its transfer has landed in the pinned buffer, go to the caller's memory (and to the progress count)
	auto collect_band = [&](size_t band_first, size_t band_blocks) -> int {
		const size_t band = band_first / chunk;
		HIP_TRY(hipEventSynchronize(s->ev_out[band & 1]), return 2);
		memcpy(job.host_out + band_first * 16, s->h_out[band & 1], band_blocks * 16);
		if (progress) progress->add(band_blocks);
		return 0;
	};
#if defined(ASTC_TRACE)
	{
		// debug build: one trace slice per block, dumped to $ASTCENC_AMD_TRACE_FILE after the call (wave_ctx.h: TRACE_PUT)
		const size_t need = nblocks * TRACE_WORDS_PER_BLOCK_HOST * sizeof(uint32_t);
		if (s->trace_cap < need)
		{
			if (s->d_prof) (void)hipFree(s->d_prof);
			s->d_prof = nullptr; s->trace_cap = 0;
			HIP_TRY(hipMalloc(&s->d_prof, need), return 1);
			s->trace_cap = need;
		}
		HIP_TRY(hipMemsetAsync(s->d_prof, 0, need, stream), return 2);
	}
#endif
	if (job.kernel_ms) HIP_TRY(hipEventRecord(s->ev0, stream), return 2);
	// The host stays at most two chunks ahead of the device: chunk k-1's completion event is waited for
	// (and reported to the progress callback) after chunk k has been queued, so the device always has its
	// next kernel waiting, a cancel takes effect within two chunks, and nothing synchronises a whole stream.
	size_t launched = 0, chunk_index = 0, prev_blocks = 0;
	for (size_t first = 0; first < nblocks; first += chunk, chunk_index++)
	{
		if (job.cancel_flag && job.cancel_flag->load(std::memory_order_relaxed)) break;
		size_t n = nblocks - first < chunk ? nblocks - first : chunk;
		if (banded)
		{
			// band 0 first; from then on the next band is queued before this band's kernel, so that it
			// crosses PCIe while the kernel runs (and ahead of this band's results on the copy stream)
			if (first == 0 && upload_band(0, n) != 0) return 2;
			const size_t next = first + n;
			if (next < nblocks && upload_band(next, nblocks - next < chunk ? nblocks - next : chunk) != 0) return 2;
			HIP_TRY(hipStreamWaitEvent(stream, s->ev_copy[(first / chunk) & 1], 0), return 2);
		}
		KernelLaunch k;
		k.d_tab = s->d_tab; k.lds_bytes = b->lds_bytes; k.img = img; k.d_out = d_out;
		k.first = (uint32_t)first; k.count = (uint32_t)n; k.stream = stream; k.d_prof = s->d_prof;
		int lrc = kernel_launch(b, s, k);
		if (lrc != 0) { log_msg("kernel launch failed (hip error %d)", lrc); return 2; }
		launched = first + n;
		if (banded)
		{
			// this band's blocks go home on the copy stream once its kernel is done; the band before it is collected
			// meanwhile (its transfer was queued one iteration ago, behind its own kernel)
			HIP_TRY(hipEventRecord(s->ev_band, stream), return 2);
			HIP_TRY(hipStreamWaitEvent(s->copy_stream, s->ev_band, 0), return 2);
			HIP_TRY(hipMemcpyAsync(s->h_out[chunk_index & 1], d_out + first * 16, n * 16, hipMemcpyDeviceToHost, s->copy_stream), return 2);
			HIP_TRY(hipEventRecord(s->ev_out[chunk_index & 1], s->copy_stream), return 2);
			if (chunk_index > 0 && collect_band(first - prev_blocks, prev_blocks) != 0) return 2;
			prev_blocks = n;
		}
		else if (chunked)
		{
			HIP_TRY(hipEventRecord(s->ev_done[chunk_index % 3], stream), return 2);
			if (chunk_index > 0)
			{
				HIP_TRY(hipEventSynchronize(s->ev_done[(chunk_index - 1) % 3]), return 2);
				if (progress) progress->add(prev_blocks);
			}
			prev_blocks = n;
		}
	}
	if (job.kernel_ms) HIP_TRY(hipEventRecord(s->ev1, stream), return 2);

	if (job.host_out && !banded && launched)
	{
		// after a cancel only the blocks that were compressed go home; the rest of the caller's buffer stays untouched
		HIP_TRY(hipMemcpyAsync(job.host_out, d_out, launched * 16, hipMemcpyDeviceToHost, stream), return 2);
	}
	HIP_TRY(hipStreamSynchronize(stream), return 2);
	if (banded)
	{
		// the last band that was launched (after a cancel: the last one before it) is still on its way
		if (launched && collect_band(launched - prev_blocks, prev_blocks) != 0) return 2;
		HIP_TRY(hipStreamSynchronize(s->copy_stream), return 2);
	}
	else if (chunked && progress && prev_blocks && launched) progress->add(prev_blocks);
	if (job.kernel_ms) HIP_TRY(hipEventElapsedTime(job.kernel_ms, s->ev0, s->ev1), return 2);
#if defined(ASTC_TRACE)
	if (const char* path = getenv("ASTCENC_AMD_TRACE_FILE"))
	{
		std::vector<uint32_t> host(nblocks * TRACE_WORDS_PER_BLOCK_HOST);
		HIP_TRY(hipMemcpy(host.data(), s->d_prof, host.size() * sizeof(uint32_t), hipMemcpyDeviceToHost), return 2);
		if (FILE* f = fopen(path, "wb")) { fwrite(host.data(), sizeof(uint32_t), host.size(), f); fclose(f); }
	}
#endif
#if defined(ASTC_PROFILE)
	{
		enum { PS_COUNT = 40, PS_TOTAL = 14 };
		static const char* names[PS_COUNT] = { "load", "ideal", "decimate", "angular", "modes", "formats", "recompute", "pack", "diff",
		                                        "realign", "kmeans+partsearch", "  partscore", "physical", "stats", "TOTAL", "blocks",
		                                        "  dec sweep1", "  dec infill", "  dec sweep3", "  ang phase1", "  ang phase2",
		                                        "  mode terms", "  mode acc", "  mode quant", "  fmt eci", "  fmt table", "  fmt combine", "  fmt select",
		                                        "  cand staging", "  physical", "  refine (all)", "  trial (all)",
		                                        "  y0 cand quantize", "  y1 cand setup", "  y2 after pack", "  y3 accept/copy", "  y4 realign infill", "  y5 realign undecimated", "  y6 realign lane-per-weight eval", "  y7 realign movers" };
		unsigned long long h[2 * PS_COUNT];
		HIP_TRY(hipMemcpy(h, s->d_prof, sizeof(h), hipMemcpyDeviceToHost), return 2);
		HIP_TRY(hipMemset(s->d_prof, 0, sizeof(h)), return 2);
		fprintf(stderr, "stage cycles per block (lane-0 shader clock), %zu blocks:   [calls per block]\n", nblocks);
		for (int i = 0; i < PS_COUNT; i++)
			if (i != 15 && h[i])
				fprintf(stderr, "  %-18s %12.0f  %5.1f%%   [%6.2f]\n", names[i], (double)h[i] / (double)nblocks, 100.0 * (double)h[i] / (double)h[PS_TOTAL],
				        (double)h[PS_COUNT + i] / (double)nblocks);
	}
#endif
	return 0;
}

/* Smallest shard worth a device of its own: below this a second GPU's fixed costs (its PCIe transfers start
 * later, its L2 has to fetch the tables again) outweigh the kernel time it takes over. */
constexpr size_t MIN_BLOCKS_PER_DEVICE = 16384;
/* Portions a sharded call is cut into per device (backend_compress): enough for a device that got cheap content to take
 * over work from one that did not, few enough for the pipeline restart at a portion's first band not to show. */
constexpr size_t DEAL_PORTIONS_PER_DEVICE = 4;
/* ... and a portion is at least this many blocks (a 2048^2 image at 6x6): every portion starts its PCIe pipeline anew -- its
 * first band crosses before anything is compressed -- which eight portions of 16 k blocks pay for visibly (profiles/r06z). */
constexpr size_t MIN_BLOCKS_PER_PORTION = 4 * MIN_BLOCKS_PER_DEVICE;

int backend_compress(Backend* b, const CompressJob& job)
{
	DeviceGuard guard;
	Progress progress;
	progress.done = 0; progress.callback = job.progress;
	{
		const uint32_t bsx = b->root.dim_x, bsy = b->root.dim_y, bsz = b->root.dim_z;
		const uint32_t dz = job.dim_z ? job.dim_z : 1u;
		progress.total = (size_t)((job.dim_x + bsx - 1) / bsx) * ((job.dim_y + bsy - 1) / bsy) * ((dz + bsz - 1) / bsz);
	}

	// Buffers that already live on a device are compressed there.
	if (!job.host_slices)
	{
		int st = 0;
		DeviceSlot* s = slot_for_pointer(b, job.device_data, &st);
		if (!s) return st ? st : 2;
		return compress_on_slot(b, s, job, &progress);
	}

	// Host images: contiguous ranges of block rows (2D) or of block layers (volumes, stacks of slices), one per device,
	// each running its own pipeline on its own streams from its own host thread; the caller's thread takes the first
	// shard and joins the rest (ref: the block loop of compress_image, astcenc_entry.cpp:1009-1038 -- blocks are
	// independent, so the split needs no exchange).  With the alpha-scale pre-pass (ref: the same split over the
	// worker threads, astcenc_entry.cpp:1190-1211, astcenc_compute_variance.cpp:507) a shard also takes the texel rows its
	// averages reach into: enough rows above and below for every 32 x 32 tile of the pre-pass that touches the shard to
	// see exactly the texels it sees in the whole image -- same tiles, same summed-area arithmetic, same floats.
	const uint32_t bsy = b->root.dim_y, bsz = b->root.dim_z;
	const uint32_t blocks_x = (job.dim_x + b->root.dim_x - 1) / b->root.dim_x;
	const uint32_t blocks_y = (job.dim_y + bsy - 1) / bsy;
	const uint32_t dim_z = job.dim_z ? job.dim_z : 1u;
	const uint32_t blocks_z = (dim_z + bsz - 1) / bsz;
	size_t ndev = b->slots.size();
	// (a stack of slices with the alpha-scale pre-pass averages around slice 0 for the whole stack: one device)
	if ((job.a_scale_radius != 0 && dim_z > 1) || !job.host_out) ndev = 1;
	const size_t by_size = progress.total / MIN_BLOCKS_PER_DEVICE;
	if (ndev > by_size) ndev = by_size < 1 ? 1 : by_size;
	// a 2D image is cut into block rows, a volume / stack of slices into layers of blocks
	const uint32_t units = dim_z == 1 ? blocks_y : blocks_z;
	if (ndev > units) ndev = units;
#if defined(ASTC_TRACE)
	ndev = 1;      // (debug build: every shard would write the same trace file)
#endif
	if (ndev <= 1) return compress_on_slot(b, b->slots[0], job, &progress);

	const size_t texel_bytes = job.data_type == 0 ? 4 : job.data_type == 1 ? 8 : 16;
	// The units are DEALT, not pre-assigned (ref: the reference's workers take blocks from one atomic ticket counter, 16 at a
	// time, astcenc_internal_entry.h:225-236 / astcenc_entry.cpp:959 -- a worker that lands on cheap content takes more):
	// block cost depends on content (a photograph runs three times faster per texel than noise), so with one contiguous range
	// per device the device that got the hard part of the image sets the call's time.  The image is cut into portions --
	// DEAL_PORTIONS_PER_DEVICE per device, none below MIN_BLOCKS_PER_DEVICE blocks -- and every device's host thread takes
	// the next portion from one counter when it has finished its last; a portion runs the banded PCIe pipeline of the
	// single-device path, so consecutive bands still overlap inside it.  ASTCENC_AMD_DEAL=static: one portion per device.
	size_t nportions = ndev;
	{
		const char* deal = getenv("ASTCENC_AMD_DEAL");
		if (!(deal && strcmp(deal, "static") == 0))
		{
			nportions = ndev * DEAL_PORTIONS_PER_DEVICE;
			// (ASTCENC_AMD_DEAL_MIN_BLOCKS: a smaller minimum, for tests/test_multi_device.py -- many portions on a small image)
			size_t min_portion = MIN_BLOCKS_PER_PORTION;
			if (const char* e = getenv("ASTCENC_AMD_DEAL_MIN_BLOCKS")) { const long v = strtol(e, nullptr, 10); if (v >= 1) min_portion = (size_t)v; }
			const size_t by_portion_size = progress.total / min_portion;
			if (nportions > by_portion_size) nportions = by_portion_size;
			if (nportions > units) nportions = units;
			if (nportions < ndev) nportions = ndev;
		}
	}
	const uint32_t units_per = (uint32_t)((units + nportions - 1) / nportions);
	struct Shard { CompressJob job; std::vector<const void*> slices; int rc; };
	std::vector<Shard> shards;
	// The reference's fast loader reads slice 0 whatever the block's z (CompressJob::fast_load_slice0): a shard that
	// starts further up the stack must then see the image's first slice as its own first slice.
	const bool needs_swz = job.swz[0] != 0 || job.swz[1] != 1 || job.swz[2] != 2 || job.swz[3] != 3;
	const bool slice0_quirk = dim_z > 1 && job.fast_load_slice0 && !needs_swz && b->cfg.profile < 2 && job.data_type == 0 && bsz == 1;
	for (size_t g = 0; g < nportions; g++)
	{
		const uint32_t u0 = (uint32_t)g * units_per;
		if (u0 >= units) break;
		const uint32_t u1 = u0 + units_per < units ? u0 + units_per : units;
		Shard sh;
		sh.job = job;
		sh.job.progress = nullptr;
		sh.rc = 0;
		if (dim_z == 1)
		{
			const uint32_t y0 = u0 * bsy;
			const uint32_t y1 = u1 * bsy < job.dim_y ? u1 * bsy : job.dim_y;
			uint32_t first_row = y0;
			if (job.a_scale_radius != 0)
			{
				// tiles of the pre-pass start at multiples of ALPHA_TILE from the image origin; a tile's summed-area table
				// takes in radius + 1 rows above its first row and radius rows below its last
				const uint32_t tile = ALPHA_TILE_ROWS_2D, r = job.a_scale_radius;
				const uint32_t tile_top = y0 / tile * tile, tile_bottom = (y1 + tile - 1) / tile * tile;
				const uint32_t want_top = tile_top > r + 1 ? (tile_top - (r + 1)) / tile * tile : 0u;
				const uint64_t want_bottom = (uint64_t)tile_bottom + r + 1;
				first_row = want_top;
				const uint32_t last_row = want_bottom < job.dim_y ? (uint32_t)want_bottom : job.dim_y;
				sh.job.halo_above = y0 - first_row;
				sh.job.halo_below = last_row - y1;
			}
			sh.slices.push_back(static_cast<const uint8_t*>(job.host_slices[0]) + (size_t)first_row * job.dim_x * texel_bytes);
			sh.job.dim_y = y1 - y0;
			sh.job.host_out = job.host_out + (size_t)u0 * blocks_x * 16;
		}
		else
		{
			const uint32_t z0 = u0 * bsz;
			const uint32_t z1 = u1 * bsz < dim_z ? u1 * bsz : dim_z;
			for (uint32_t z = z0; z < z1; z++) sh.slices.push_back(slice0_quirk ? job.host_slices[0] : job.host_slices[z]);
			sh.job.dim_z = z1 - z0;
			sh.job.host_out = job.host_out + (size_t)u0 * blocks_x * blocks_y * 16;
		}
		shards.push_back(std::move(sh));
	}
	for (Shard& sh : shards) sh.job.host_slices = sh.slices.data();      // (after the vector stopped growing)
	// one host thread per further device; a device whose thread cannot be created (std::system_error must not cross the
	// C ABI, and the earlier workers must still be joined) simply takes no portions: the others deal its share out
	std::atomic<size_t> next_portion{0};
	std::atomic<bool> failed{false};
	const size_t nslots = ndev < shards.size() ? ndev : shards.size();
	std::vector<double> busy_ms(nslots, 0.0);
	std::vector<unsigned> taken(nslots, 0u);
	auto deal = [&](size_t g)
	{
		const auto t0 = std::chrono::steady_clock::now();
		for (;;)
		{
			if (failed.load()) break;                          // (a failed portion fails the call: nobody starts another)
			const size_t p = next_portion.fetch_add(1);
			if (p >= shards.size()) break;
			shards[p].rc = compress_on_slot(b, b->slots[g], shards[p].job, &progress);
			if (shards[p].rc != 0) failed.store(true);
			taken[g]++;
		}
		busy_ms[g] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
	};
	std::vector<size_t> on_workers;
	for (size_t g = 1; g < nslots; g++)
	{
		// (building the task may allocate: nothing may be thrown across the C ABI -- ADVICE r05)
		try { if (worker_run(b->slots[g], [&deal, g]() { deal(g); })) on_workers.push_back(g); }
		catch (...) { }
	}
	deal(0);
	for (size_t g : on_workers) worker_wait(b->slots[g]);
	for (size_t g = 0; g < nslots; g++) log_msg("compress: device slot %zu took %u of %zu portions, busy %.2f ms", g, taken[g], shards.size(), busy_ms[g]);
	int rc = 0;
	for (const Shard& sh : shards) if (sh.rc != 0 && (rc == 0 || sh.rc == 1)) rc = sh.rc;
	return rc;
}

/* The blocks of `job` decoded on one slot. */
static int decompress_on_slot(Backend* bk, DeviceSlot* b, const DecompressJob& job)
{
	std::lock_guard<std::mutex> busy(b->busy);
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
	d.d_tables = b->d_dectab;
	d.dim_x = job.dim_x; d.dim_y = job.dim_y; d.dim_z = dim_z; d.data_type = job.data_type;
	for (int i = 0; i < 4; i++) d.swz[i] = job.swz[i];
	d.block_x = bk->root.dim_x; d.block_y = bk->root.dim_y; d.block_z = bk->root.dim_z;
	d.profile = bk->cfg.profile;
	d.stream = b->stream;
	int lrc = astc_decode_launch(d);
	if (lrc != 0) { log_msg("decode kernel launch failed (hip error %d)", lrc); return 2; }
	for (uint32_t z = 0; z < dim_z; z++)
		HIP_TRY(hipMemcpyAsync(job.host_slices[z], static_cast<uint8_t*>(b->d_image) + z * slice_bytes, slice_bytes, hipMemcpyDeviceToHost, b->stream), return 2);
	HIP_TRY(hipStreamSynchronize(b->stream), return 2);
	return 0;
}

/* Host blocks -> host image.  Like compression the work is dealt to the context's devices in contiguous ranges --
 * block rows of a 2D image, layers of blocks of a volume -- each decoded into its own rows / slices of the caller's
 * image (ref: the block loop of astcenc_decompress_image, astcenc_entry.cpp:1340-1385: blocks are independent). */
int backend_decompress(Backend* bk, const DecompressJob& job)
{
	DeviceGuard guard;
	const uint32_t bsx = bk->root.dim_x, bsy = bk->root.dim_y, bsz = bk->root.dim_z;
	const uint32_t dim_z = job.dim_z ? job.dim_z : 1u;
	const uint32_t blocks_x = (job.dim_x + bsx - 1) / bsx, blocks_y = (job.dim_y + bsy - 1) / bsy, blocks_z = (dim_z + bsz - 1) / bsz;
	const size_t total = (size_t)blocks_x * blocks_y * blocks_z;
	size_t ndev = bk->slots.size();
	const size_t by_size = total / MIN_BLOCKS_PER_DEVICE;
	if (ndev > by_size) ndev = by_size < 1 ? 1 : by_size;
	const uint32_t units = dim_z == 1 ? blocks_y : blocks_z;
	if (ndev > units) ndev = units;
	if (ndev <= 1) return decompress_on_slot(bk, bk->slots[0], job);

	const size_t texel_bytes = job.data_type == 0 ? 4 : job.data_type == 1 ? 8 : 16;
	const uint32_t units_per = (uint32_t)((units + ndev - 1) / ndev);
	struct Shard { DecompressJob job; std::vector<void*> slices; int rc; };
	std::vector<Shard> shards;
	for (size_t g = 0; g < ndev; g++)
	{
		const uint32_t u0 = (uint32_t)g * units_per;
		if (u0 >= units) break;
		const uint32_t u1 = u0 + units_per < units ? u0 + units_per : units;
		Shard sh;
		sh.job = job;
		sh.rc = 0;
		if (dim_z == 1)
		{
			const uint32_t y0 = u0 * bsy;
			const uint32_t y1 = u1 * bsy < job.dim_y ? u1 * bsy : job.dim_y;
			sh.slices.push_back(static_cast<uint8_t*>(job.host_slices[0]) + (size_t)y0 * job.dim_x * texel_bytes);
			sh.job.dim_y = y1 - y0;
			sh.job.host_blocks = job.host_blocks + (size_t)u0 * blocks_x * 16;
			sh.job.block_bytes = (size_t)(u1 - u0) * blocks_x * 16;
		}
		else
		{
			const uint32_t z0 = u0 * bsz;
			const uint32_t z1 = u1 * bsz < dim_z ? u1 * bsz : dim_z;
			for (uint32_t z = z0; z < z1; z++) sh.slices.push_back(job.host_slices[z]);
			sh.job.dim_z = z1 - z0;
			sh.job.host_blocks = job.host_blocks + (size_t)u0 * blocks_x * blocks_y * 16;
			sh.job.block_bytes = (size_t)(u1 - u0) * blocks_x * blocks_y * 16;
		}
		shards.push_back(std::move(sh));
	}
	for (Shard& sh : shards) sh.job.host_slices = sh.slices.data();
	std::vector<size_t> on_workers, inline_shards;
	for (size_t g = 1; g < shards.size(); g++)
	{
		bool handed = false;
		try { handed = worker_run(bk->slots[g], [&shards, bk, g]() { shards[g].rc = decompress_on_slot(bk, bk->slots[g], shards[g].job); }); }
		catch (...) { }                                     // (building the task may allocate: nothing is thrown across the C ABI)
		if (handed) on_workers.push_back(g);
		else inline_shards.push_back(g);
	}
	shards[0].rc = decompress_on_slot(bk, bk->slots[0], shards[0].job);
	for (size_t g : inline_shards) shards[g].rc = decompress_on_slot(bk, bk->slots[g], shards[g].job);
	for (size_t g : on_workers) worker_wait(bk->slots[g]);
	int rc = 0;
	for (const Shard& sh : shards) if (sh.rc != 0 && (rc == 0 || sh.rc == 1)) rc = sh.rc;
	return rc;
}

int backend_decompress_device(Backend* bk, const DecompressDeviceJob& job)
{
	DeviceGuard guard;
	int st = 0;
	DeviceSlot* b = slot_for_pointer(bk, job.device_image, &st);
	if (!b) return st ? st : 2;
	std::lock_guard<std::mutex> busy(b->busy);
	HIP_TRY(hipSetDevice(b->device), return 2);
	hipStream_t stream;
	if (!pick_stream(b, job.stream, &stream)) return 3;
	DecodeLaunch d;
	d.d_blocks = job.device_blocks;
	d.d_image = job.device_image;
	d.d_tables = b->d_dectab;
	d.dim_x = job.dim_x; d.dim_y = job.dim_y; d.dim_z = job.dim_z ? job.dim_z : 1u; d.data_type = job.data_type;
	for (int i = 0; i < 4; i++) d.swz[i] = job.swz[i];
	d.block_x = bk->root.dim_x; d.block_y = bk->root.dim_y; d.block_z = bk->root.dim_z;
	d.profile = bk->cfg.profile;
	d.stream = stream;
	int lrc = astc_decode_launch(d);
	if (lrc != 0) { log_msg("decode kernel launch failed (hip error %d)", lrc); return 2; }
	HIP_TRY(hipStreamSynchronize(stream), return 2);
	return 0;
}

int backend_compare(Backend* bk, const CompareJob& job)
{
	DeviceGuard guard;
	int st = 0;
	DeviceSlot* b = slot_for_pointer(bk, job.device_a, &st);
	if (!b) return st ? st : 2;
	std::lock_guard<std::mutex> busy(b->busy);
	HIP_TRY(hipSetDevice(b->device), return 2);
	hipStream_t stream;
	if (!pick_stream(b, job.stream, &stream)) return 3;
	if (!b->d_sums) HIP_TRY(hipMalloc(&b->d_sums, astc_compare_scratch_doubles() * sizeof(double)), return 1);
	CompareLaunch c;
	c.d_a = job.device_a; c.type_a = job.type_a; c.d_b = job.device_b; c.type_b = job.type_b;
	c.texels = job.texels; c.d_sums = b->d_sums; c.stream = stream;
	c.hdr = job.hdr; c.fstop_lo = job.fstop_lo; c.fstop_hi = job.fstop_hi;
	int lrc = astc_compare_launch(c);
	if (lrc != 0) { log_msg("compare kernel launch failed (hip error %d)", lrc); return 2; }
	HIP_TRY(hipMemcpyAsync(job.sums, b->d_sums, METRIC_SUMS_HOST * sizeof(double), hipMemcpyDeviceToHost, stream), return 2);
	HIP_TRY(hipStreamSynchronize(stream), return 2);
	return 0;
}

} // namespace astcd
