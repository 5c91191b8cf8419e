// SPDX-License-Identifier: Apache-2.0
// Run-time specialised builds of the compression kernel (kernel_jit.h).  Host code only.
#include "kernel_jit.h"
#include "wave_ctx.h"      // LdsLayout (host side of the shared source)

#include <hip/hiprtc.h>
#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

// The device source, byte for byte as it was compiled into this library's own kernels: the assembler copies the files into
// .rodata (found through the -I of the build).
#define ASTC_EMBED(sym, file) \
	__asm__(".pushsection .rodata\n" ".balign 16\n" #sym ":\n" ".incbin \"" file "\"\n" ".byte 0\n" #sym "_end:\n" ".popsection\n"); \
	extern "C" const char sym[]; extern "C" const char sym##_end[];
ASTC_EMBED(astc_src_kernel_device, "kernel_device.h")
ASTC_EMBED(astc_src_wave_block, "wave_block.h")
ASTC_EMBED(astc_src_wave, "wave.h")
ASTC_EMBED(astc_src_wave_ctx, "wave_ctx.h")
ASTC_EMBED(astc_src_astc_tables, "astc_tables.h")
ASTC_EMBED(astc_src_wave_quad, "wave_quad.h")
ASTC_EMBED(astc_src_wave_load, "wave_load.h")
ASTC_EMBED(astc_src_wave_ideal, "wave_ideal.h")
ASTC_EMBED(astc_src_wave_weights, "wave_weights.h")
ASTC_EMBED(astc_src_wave_format, "wave_format.h")
ASTC_EMBED(astc_src_wave_color, "wave_color.h")
ASTC_EMBED(astc_src_wave_color_hdr, "wave_color_hdr.h")
ASTC_EMBED(astc_src_wave_refine, "wave_refine.h")
ASTC_EMBED(astc_src_wave_batch, "wave_batch.h")
ASTC_EMBED(astc_src_wave_partition, "wave_partition.h")
ASTC_EMBED(astc_src_wave_pack, "wave_pack.h")

namespace astcd {
namespace {

struct EmbeddedHeader { const char* name; const char* text; const char* end; };
const EmbeddedHeader kHeaders[] = {
	{ "kernel_device.h", astc_src_kernel_device, astc_src_kernel_device_end }, { "wave_block.h", astc_src_wave_block, astc_src_wave_block_end },
	{ "wave.h", astc_src_wave, astc_src_wave_end }, { "wave_ctx.h", astc_src_wave_ctx, astc_src_wave_ctx_end },
	{ "astc_tables.h", astc_src_astc_tables, astc_src_astc_tables_end }, { "wave_quad.h", astc_src_wave_quad, astc_src_wave_quad_end },
	{ "wave_load.h", astc_src_wave_load, astc_src_wave_load_end }, { "wave_ideal.h", astc_src_wave_ideal, astc_src_wave_ideal_end },
	{ "wave_weights.h", astc_src_wave_weights, astc_src_wave_weights_end }, { "wave_format.h", astc_src_wave_format, astc_src_wave_format_end },
	{ "wave_color.h", astc_src_wave_color, astc_src_wave_color_end }, { "wave_color_hdr.h", astc_src_wave_color_hdr, astc_src_wave_color_hdr_end },
	{ "wave_refine.h", astc_src_wave_refine, astc_src_wave_refine_end }, { "wave_batch.h", astc_src_wave_batch, astc_src_wave_batch_end },
	{ "wave_partition.h", astc_src_wave_partition, astc_src_wave_partition_end }, { "wave_pack.h", astc_src_wave_pack, astc_src_wave_pack_end },
};
constexpr int kHeaderCount = (int)(sizeof(kHeaders) / sizeof(kHeaders[0]));

/* hipRTC, looked up when the first build is asked for.  (Not a link-time dependency: a box without the library -- or a
 * process that must not load a compiler -- keeps the generic kernels and loses nothing else.  This is the only dlopen of
 * the library and it names the ROCm run-time compiler, nothing of the test infrastructure.) */
struct Rtc {
	void* handle = nullptr;
	decltype(&hiprtcCreateProgram) create = nullptr;
	decltype(&hiprtcCompileProgram) compile = nullptr;
	decltype(&hiprtcGetCodeSize) code_size = nullptr;
	decltype(&hiprtcGetCode) code = nullptr;
	decltype(&hiprtcGetProgramLogSize) log_size = nullptr;
	decltype(&hiprtcGetProgramLog) log = nullptr;
	decltype(&hiprtcDestroyProgram) destroy = nullptr;
	decltype(&hiprtcVersion) version = nullptr;
	int major = 0, minor = 0;
	bool ok = false;
};
const Rtc& rtc()
{
	static const Rtc r = []() {
		Rtc t;
		const char* names[] = { getenv("ASTCENC_AMD_HIPRTC"), "libhiprtc.so.7", "libhiprtc.so", "/opt/rocm/lib/libhiprtc.so" };
		for (const char* n : names)
		{
			if (!n || !*n) continue;
			t.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
			if (t.handle) break;
		}
		if (!t.handle) return t;
		t.create = reinterpret_cast<decltype(t.create)>(dlsym(t.handle, "hiprtcCreateProgram"));
		t.compile = reinterpret_cast<decltype(t.compile)>(dlsym(t.handle, "hiprtcCompileProgram"));
		t.code_size = reinterpret_cast<decltype(t.code_size)>(dlsym(t.handle, "hiprtcGetCodeSize"));
		t.code = reinterpret_cast<decltype(t.code)>(dlsym(t.handle, "hiprtcGetCode"));
		t.log_size = reinterpret_cast<decltype(t.log_size)>(dlsym(t.handle, "hiprtcGetProgramLogSize"));
		t.log = reinterpret_cast<decltype(t.log)>(dlsym(t.handle, "hiprtcGetProgramLog"));
		t.destroy = reinterpret_cast<decltype(t.destroy)>(dlsym(t.handle, "hiprtcDestroyProgram"));
		t.version = reinterpret_cast<decltype(t.version)>(dlsym(t.handle, "hiprtcVersion"));
		t.ok = t.create && t.compile && t.code_size && t.code && t.log_size && t.log && t.destroy;
		if (t.ok && t.version) (void)t.version(&t.major, &t.minor);
		return t;
	}();
	return r;
}

uint64_t fnv1a(uint64_t h, const void* data, size_t n)
{
	const uint8_t* p = static_cast<const uint8_t*>(data);
	for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 1099511628211ull; }
	return h;
}

/* The context's three records as the `constexpr` initializers a fixed-context build includes as "fixed_contexts.inc"
 * (wave_ctx.h).  LdsLayout is 32-bit words throughout, TableRoot four bytes and then words; floats go by bit pattern. */
std::string records_text(const LdsLayout& L, const DeviceConfig& cfg, const TableRoot& r)
{
	static_assert(sizeof(LdsLayout) % 4 == 0 && sizeof(TableRoot) % 4 == 0, "records are word arrays");
	static_assert(sizeof(DeviceConfig) == 92, "DeviceConfig changed: update records_text (and oracle/emu/backend_emu.cpp)");
	std::string s;
	char t[96];
	auto word = [&](uint32_t v) { snprintf(t, sizeof(t), "%uu", v); s += t; };
	auto flt = [&](float f) { uint32_t v; memcpy(&v, &f, 4); snprintf(t, sizeof(t), "__builtin_bit_cast(float, %uu)", v); s += t; };
	s += "constexpr LdsLayout kFixedLayout = { ";
	for (size_t i = 0; i < sizeof(LdsLayout) / 4; i++) { if (i) s += ", "; word(reinterpret_cast<const uint32_t*>(&L)[i]); }
	s += " };\nconstexpr DeviceConfig kFixedConfig = { ";
	snprintf(t, sizeof(t), "%d, ", (int)cfg.profile); s += t;
	word(cfg.flags); s += ", { ";
	for (int i = 0; i < 4; i++) { if (i) s += ", "; flt(cfg.cw[i]); }
	s += " }, "; flt(cfg.rgbm_m_scale); s += ", "; word(cfg.tune_partition_count_limit); s += ", { ";
	for (int i = 0; i < 3; i++) { if (i) s += ", "; word(cfg.tune_partition_index_limit[i]); }
	s += " }, "; word(cfg.tune_refinement_limit); s += ", "; word(cfg.tune_candidate_limit); s += ", { ";
	for (int i = 0; i < 3; i++) { if (i) s += ", "; word(cfg.tune_partitioning_candidate_limit[i]); }
	s += " }, "; flt(cfg.tune_db_limit); s += ", "; flt(cfg.tune_mse_overshoot); s += ", { ";
	flt(cfg.tune_partition_early_out_limit_factor[0]); s += ", "; flt(cfg.tune_partition_early_out_limit_factor[1]);
	s += " }, "; flt(cfg.tune_2plane_early_out_limit_correlation); s += ", "; flt(cfg.tune_search_mode0_enable); s += ", 0u };\n";
	snprintf(t, sizeof(t), "constexpr TableRoot kFixedRoot = { %u, %u, %u, %u", (unsigned)r.dim_x, (unsigned)r.dim_y, (unsigned)r.texel_count, (unsigned)r.dim_z);
	s += t;
	for (size_t i = 1; i < sizeof(TableRoot) / 4; i++) { s += ", "; word(reinterpret_cast<const uint32_t*>(&r)[i]); }
	s += " };\n";
	return s;
}

std::string cache_dir()
{
	const char* e = getenv("ASTCENC_AMD_CACHE_DIR");
	if (e) return *e ? std::string(e) : std::string();      // (set but empty: no disk cache)
	if ((e = getenv("XDG_CACHE_HOME")) && *e) return std::string(e) + "/astcenc_amd";
	if ((e = getenv("HOME")) && *e) return std::string(e) + "/.cache/astcenc_amd";
	return std::string();
}
void make_dirs(const std::string& path)
{
	for (size_t i = 1; i <= path.size(); i++)
		if (i == path.size() || path[i] == '/') (void)mkdir(path.substr(0, i).c_str(), 0755);
}

} // namespace

struct JitKernel {
	std::string key;                // hash of everything that decides the code object, in hex
	std::string name;               // astc_compress_blocks_jit_<key>
	std::string records;            // "fixed_contexts.inc" of this build
	std::string unit;               // the translation unit
	std::vector<std::string> options;
	std::vector<char> code;
	double seconds = 0.0;
	int refs = 0;
	JitState state = JIT_IDLE;
	void (*log)(const char*) = nullptr;
};

namespace {

/* One worker thread compiles the queued builds one after the other (a compile is seven to fifteen seconds of one core: ten
 * contexts created at once must not become ten compilers).  The thread is started with the first queued build and joined
 * when the library is unloaded -- which waits for a compile in flight: the compiler must not be inside LLVM while the
 * process tears its statics down. */
struct Queue {
	std::mutex mu;
	std::condition_variable cv;
	std::deque<JitKernel*> pending;
	std::map<std::string, JitKernel*> by_key;
	std::thread worker;
	bool started = false, quit = false;
	~Queue()
	{
		{ std::lock_guard<std::mutex> g(mu); quit = true; pending.clear(); }
		cv.notify_all();
		if (worker.joinable()) worker.join();
	}
};
Queue& queue() { static Queue q; return q; }

void say(const JitKernel* k, const char* fmt, const char* a, double b = 0.0, size_t c = 0)
{
	if (!k->log) return;
	char line[768];
	snprintf(line, sizeof(line), fmt, a, b, c);
	k->log(line);
}

bool compile_now(JitKernel* k)
{
	const Rtc& r = rtc();
	if (!r.ok) return false;
	std::vector<const char*> texts, names;
	for (int i = 0; i < kHeaderCount; i++) { texts.push_back(kHeaders[i].text); names.push_back(kHeaders[i].name); }
	texts.push_back(k->records.c_str()); names.push_back("fixed_contexts.inc");
	hiprtcProgram prog = nullptr;
	if (r.create(&prog, k->unit.c_str(), "astc_compress_blocks_jit.hip", (int)texts.size(), texts.data(), names.data()) != HIPRTC_SUCCESS) return false;
	std::vector<const char*> opts;
	for (const std::string& o : k->options) opts.push_back(o.c_str());
	const auto t0 = std::chrono::steady_clock::now();
	const hiprtcResult rc = r.compile(prog, (int)opts.size(), opts.data());
	k->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	bool ok = rc == HIPRTC_SUCCESS;
	if (!ok)
	{
		size_t n = 0;
		std::string text;
		if (r.log_size(prog, &n) == HIPRTC_SUCCESS && n > 1) { text.resize(n); (void)r.log(prog, &text[0]); }
		if (text.size() > 600) text.resize(600);
		say(k, "run-time build %s failed to compile: %.600s", (k->name + " " + text).c_str());
	}
	size_t bytes = 0;
	if (ok) ok = r.code_size(prog, &bytes) == HIPRTC_SUCCESS && bytes > 0;
	if (ok) { k->code.resize(bytes); ok = r.code(prog, k->code.data()) == HIPRTC_SUCCESS; }
	(void)r.destroy(&prog);
	if (!ok) { k->code.clear(); return false; }
	// the disk cache: written under a temporary name, renamed into place (readers see a whole file or none)
	const std::string dir = cache_dir();
	if (!dir.empty())
	{
		make_dirs(dir);
		const std::string path = dir + "/" + k->key + ".hsaco";
		char tmp[32];
		snprintf(tmp, sizeof(tmp), ".tmp%ld", (long)getpid());
		const std::string tpath = path + tmp;
		FILE* f = fopen(tpath.c_str(), "wb");
		if (f)
		{
			const bool wrote = fwrite(k->code.data(), 1, k->code.size(), f) == k->code.size();
			if (fclose(f) == 0 && wrote) (void)rename(tpath.c_str(), path.c_str());
			else (void)unlink(tpath.c_str());
		}
	}
	say(k, "run-time build %s compiled in %.1f s (%zu bytes)", k->name.c_str(), k->seconds, k->code.size());
	return true;
}

bool load_cached(JitKernel* k)
{
	const std::string dir = cache_dir();
	if (dir.empty()) return false;
	FILE* f = fopen((dir + "/" + k->key + ".hsaco").c_str(), "rb");
	if (!f) return false;
	std::vector<char> data;
	char buf[65536];
	size_t n;
	while ((n = fread(buf, 1, sizeof(buf), f)) > 0) data.insert(data.end(), buf, buf + n);
	fclose(f);
	if (data.size() < 64 || memcmp(data.data(), "\177ELF", 4) != 0) return false;
	k->code.swap(data);
	return true;
}

void worker_main()
{
	Queue& q = queue();
	std::unique_lock<std::mutex> g(q.mu);
	for (;;)
	{
		q.cv.wait(g, [&] { return q.quit || !q.pending.empty(); });
		if (q.quit) return;
		JitKernel* k = q.pending.front();
		q.pending.pop_front();
		k->state = JIT_COMPILING;
		k->refs++;                           // (the build outlives its contexts while the compiler works on it)
		g.unlock();
		const bool ok = compile_now(k);
		g.lock();
		k->state = ok ? JIT_READY : JIT_FAILED;
		q.cv.notify_all();
		if (--k->refs == 0) { q.by_key.erase(k->key); delete k; }
	}
}

void start_locked(Queue& q, JitKernel* k)
{
	if (k->state != JIT_IDLE) return;
	k->state = JIT_QUEUED;
	q.pending.push_back(k);
	if (!q.started) { q.started = true; q.worker = std::thread(worker_main); }
	q.cv.notify_all();
}

} // namespace

JitMode jit_mode_from_environment()
{
	const char* e = getenv("ASTCENC_AMD_JIT");
	if (!e || !*e || strcmp(e, "lazy") == 0) return JIT_LAZY;
	if (strcmp(e, "off") == 0 || strcmp(e, "0") == 0) return JIT_OFF;
	if (strcmp(e, "eager") == 0 || strcmp(e, "async") == 0) return JIT_EAGER;
	if (strcmp(e, "sync") == 0) return JIT_SYNC;
	return JIT_LAZY;
}

JitKernel* jit_acquire(const void* layout, size_t layout_bytes, const DeviceConfig& cfg_in, const TableRoot& root, bool hdr, const char* arch,
                       void (*log)(const char*))
{
	if (layout_bytes != sizeof(LdsLayout) || !rtc().ok) return nullptr;
	LdsLayout L;
	memcpy(&L, layout, sizeof(L));
	DeviceConfig cfg = cfg_in;
	cfg.debug_dup_stage = 0;
	JitKernel* k = new JitKernel;
	k->log = log;
	k->records = records_text(L, cfg, root);
	// the translation unit: what kernel_ldr_6x6m.hip is for its context, with the records from the "header" above
	k->unit = std::string("#define ASTC_VARIANT v_jit\n") + (hdr ? "#define ASTC_ENABLE_HDR 1\n" : "#define ASTC_ENABLE_HDR 0\n") +
	          (root.texel_count <= 64 ? "#define ASTC_TEXELS_LE_64 1\n" : "") +
	          "#define ASTC_FIXED_CONTEXT 1\n#define ASTC_KERNEL_NAME astc_compress_blocks_jit\n#define ASTC_KERNEL_LINKAGE extern \"C\"\n#include \"kernel_device.h\"\n";
	// the numerics flags of the Makefile are part of the bit-exactness contract (wave.h)
	std::string a = arch && *arch ? arch : "gfx950";
	a = a.substr(0, a.find(':'));
	k->options = { "--offload-arch=" + a, "-O3", "-std=c++17", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-math-errno",
	               "-fno-slp-vectorize", "-Wno-unused-function" };
	uint64_t h = 14695981039346656037ull;
	for (int i = 0; i < kHeaderCount; i++) h = fnv1a(h, kHeaders[i].text, (size_t)(kHeaders[i].end - kHeaders[i].text));
	h = fnv1a(h, k->records.data(), k->records.size());
	h = fnv1a(h, k->unit.data(), k->unit.size());
	for (const std::string& o : k->options) h = fnv1a(h, o.data(), o.size() + 1);
	const int ver[2] = { rtc().major, rtc().minor };
	h = fnv1a(h, ver, sizeof(ver));
	char hex[24];
	snprintf(hex, sizeof(hex), "%016llx", (unsigned long long)h);
	k->key = hex;
	k->name = std::string(JIT_ENTRY_POINT) + "_" + hex;

	Queue& q = queue();
	std::lock_guard<std::mutex> g(q.mu);
	auto it = q.by_key.find(k->key);
	if (it != q.by_key.end()) { delete k; it->second->refs++; return it->second; }
	if (load_cached(k)) k->state = JIT_READY;
	k->refs = 1;
	q.by_key[k->key] = k;
	return k;
}

void jit_release(JitKernel* k)
{
	if (!k) return;
	Queue& q = queue();
	std::lock_guard<std::mutex> g(q.mu);
	if (--k->refs != 0) return;
	if (k->state == JIT_QUEUED)
		for (auto it = q.pending.begin(); it != q.pending.end(); ++it) if (*it == k) { q.pending.erase(it); break; }
	q.by_key.erase(k->key);
	delete k;
}

void jit_start(JitKernel* k)
{
	if (!k) return;
	Queue& q = queue();
	std::lock_guard<std::mutex> g(q.mu);
	start_locked(q, k);
}

JitState jit_wait(JitKernel* k)
{
	if (!k) return JIT_FAILED;
	Queue& q = queue();
	std::unique_lock<std::mutex> g(q.mu);
	start_locked(q, k);
	q.cv.wait(g, [&] { return k->state == JIT_READY || k->state == JIT_FAILED || q.quit; });
	return k->state;
}

JitState jit_state(const JitKernel* k)
{
	if (!k) return JIT_FAILED;
	Queue& q = queue();
	std::lock_guard<std::mutex> g(q.mu);
	return k->state;
}

const void* jit_code(const JitKernel* k, size_t* bytes) { *bytes = k->code.size(); return k->code.data(); }
const char* jit_kernel_name(const JitKernel* k) { return k->name.c_str(); }
double jit_compile_seconds(const JitKernel* k) { return k->seconds; }

} // namespace astcd
