// SPDX-License-Identifier: Apache-2.0
// Run-time specialised builds of the compression kernel (kernel_jit.h).  Host code only.
#include "kernel_jit.h"
#include "wave_ctx.h"      // LdsLayout (host side of the shared source)

#include <hip/hiprtc.h>
#include <dlfcn.h>
#include <fcntl.h>
#include <link.h>
#include <signal.h>
#include <spawn.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>
#include <atomic>
#include <cerrno>
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

extern char** environ;

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
constexpr long JIT_MAX_SCRATCH_BYTES = 32;

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


/* Where the library's own file lies: the compiler process (astcenc_amd_jitc, jitc_main.cpp) is installed next to it. */
std::string helper_path()
{
	static const std::string path = []() {
		const char* e = getenv("ASTCENC_AMD_JITC");
		if (e) return std::string(e);                      // (set but empty: compile inside this process)
		Dl_info info;
		if (!dladdr(reinterpret_cast<const void*>(&helper_path), &info) || !info.dli_fname) return std::string();
		std::string p(info.dli_fname);
		const size_t slash = p.rfind('/');
		p = (slash == std::string::npos ? std::string(".") : p.substr(0, slash)) + "/astcenc_amd_jitc";
		return access(p.c_str(), X_OK) == 0 ? p : std::string();
	}();
	return path;
}

/* Runs the compiler process and waits for it; `pid_out` names it while it runs (the queue kills it when the library is
 * unloaded).  stdout / stderr of the child end up in the given files (may be null: inherited). */
std::atomic<pid_t> g_child{0};
int run_helper(const std::vector<std::string>& args, const char* stdout_path, const char* stderr_path, const char* scratch_dir = nullptr)
{
	std::vector<char*> argv;
	for (const std::string& a : args) argv.push_back(const_cast<char*>(a.c_str()));
	argv.push_back(nullptr);
	posix_spawn_file_actions_t fa;
	posix_spawn_file_actions_init(&fa);
	// The child may outlive this process: it must not hold the host's standard streams or any other descriptor of the host
	// (a pipe that somebody waits on to close) -- its own go to the given files or to /dev/null.
	posix_spawn_file_actions_addopen(&fa, 0, "/dev/null", O_RDONLY, 0);
	posix_spawn_file_actions_addopen(&fa, 1, stdout_path ? stdout_path : "/dev/null", O_WRONLY | O_CREAT | O_TRUNC, 0644);
	posix_spawn_file_actions_addopen(&fa, 2, stderr_path ? stderr_path : "/dev/null", O_WRONLY | O_CREAT | O_TRUNC, 0644);
#if defined(__GLIBC__) && __GLIBC_PREREQ(2, 34)
	posix_spawn_file_actions_addclosefrom_np(&fa, 3);
#endif
	pid_t pid = 0;
	// (the child's environment: this process's, plus the scratch directory it is to remove when it is done)
	std::vector<char*> envp;
	std::string scratch_var;
	for (char** e = environ; e && *e; e++) if (strncmp(*e, "ASTCENC_AMD_JITC_SCRATCH=", 25) != 0) envp.push_back(*e);
	if (scratch_dir) { scratch_var = std::string("ASTCENC_AMD_JITC_SCRATCH=") + scratch_dir; envp.push_back(const_cast<char*>(scratch_var.c_str())); }
	envp.push_back(nullptr);
	const int rc = posix_spawn(&pid, argv[0], &fa, nullptr, argv.data(), envp.data());
	posix_spawn_file_actions_destroy(&fa);
	if (rc != 0) return -1;
	g_child.store(pid);
	int status = 0;
	while (waitpid(pid, &status, 0) < 0 && errno == EINTR) { }
	g_child.store(0);
	return WIFEXITED(status) ? WEXITSTATUS(status) : -1;
}

std::string slurp(const std::string& path)
{
	std::string out;
	FILE* f = fopen(path.c_str(), "rb");
	if (!f) return out;
	char buf[65536];
	size_t n;
	while ((n = fread(buf, 1, sizeof(buf), f)) > 0) out.append(buf, n);
	fclose(f);
	return out;
}
bool spill(const std::string& path, const void* data, size_t n)
{
	FILE* f = fopen(path.c_str(), "wb");
	if (!f) return false;
	const bool ok = fwrite(data, 1, n, f) == n;
	return fclose(f) == 0 && ok;
}
std::string temp_dir()
{
	std::string base = cache_dir();
	if (!base.empty()) make_dirs(base); else { const char* t = getenv("TMPDIR"); base = t && *t ? t : "/tmp"; }
	std::string tmpl = base + "/jitXXXXXX";
	return mkdtemp(&tmpl[0]) ? tmpl : std::string();
}

/* Names the compiler a build would come from (part of the cache key: hiprtcVersion() is the same constant for every ROCm
 * release).  The compiler process says which hipRTC / comgr files it resolves; inside this process: the hipRTC file. */
const std::string& compiler_identity()
{
	static const std::string id = []() {
		const std::string helper = helper_path();
		if (!helper.empty())
		{
			const std::string dir = temp_dir();
			if (!dir.empty())
			{
				const std::string out = dir + "/identity.txt";
				const int rc = run_helper({ helper, "--identity" }, out.c_str(), nullptr);
				std::string text = rc == 0 ? slurp(out) : std::string();
				(void)unlink(out.c_str());
				(void)rmdir(dir.c_str());
				while (!text.empty() && (text.back() == '\n' || text.back() == ' ')) text.pop_back();
				if (!text.empty()) return "process " + text;
			}
		}
		std::string text = "inprocess ";
		struct link_map* lm = nullptr;
		if (rtc().handle && dlinfo(rtc().handle, RTLD_DI_LINKMAP, &lm) == 0 && lm && lm->l_name)
		{
			struct stat st;
			text += lm->l_name;
			if (stat(lm->l_name, &st) == 0) text += " " + std::to_string((long)st.st_size);
		}
		return text;
	}();
	return id;
}

/* A value of the code object's kernel metadata (a msgpack map in the .note section: the key as a string, then an unsigned
 * integer), or -1: ".vgpr_count", ".private_segment_fixed_size". */
long metadata_value(const std::vector<char>& code, const char* key)
{
	const size_t klen = strlen(key);
	for (size_t i = 0; i + klen + 1 < code.size(); i++)
	{
		if (memcmp(&code[i], key, klen) != 0) continue;
		const uint8_t* p = reinterpret_cast<const uint8_t*>(&code[i + klen]);
		const size_t left = code.size() - (i + klen);
		if (p[0] <= 0x7f) return p[0];
		if (p[0] == 0xcc && left > 1) return p[1];
		if (p[0] == 0xcd && left > 2) return (long)p[1] << 8 | p[2];
		if (p[0] == 0xce && left > 4) return (long)p[1] << 24 | (long)p[2] << 16 | (long)p[3] << 8 | p[4];
	}
	return -1;
}

/* A build is only worth having when it keeps the occupancy the kernel is laid out for: four waves per SIMD, i.e. at most 128
 * VGPRs.  (The ROCm 7.0 compiler, met inside a PyTorch process, gives the stage functions 160 VGPRs and a scratch frame: 18 %
 * slower than the generic build of the library.)  A frame of a few bytes is accepted: some contexts (5x5 -medium) leave one
 * stage function a scalar register short, which costs one 8-byte slot per lane and still measures faster than the generic
 * build; anything bigger is a register allocation gone wrong. */
bool acceptable(const std::vector<char>& code, long* vgprs, long* scratch)
{
	*vgprs = metadata_value(code, ".vgpr_count");
	*scratch = metadata_value(code, ".private_segment_fixed_size");
	return *vgprs >= 0 && *vgprs <= 128 && *scratch >= 0 && *scratch <= JIT_MAX_SCRATCH_BYTES;
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
};
/* Never destroyed: the worker may be waiting for the compiler process when the host exits, and a host does not wait for a
 * compile -- the compiler process finishes on its own and leaves the build in the disk cache (jitc_main.cpp).  Only a compile
 * that runs INSIDE this process (no compiler process installed) is waited for at exit: LLVM must not be at work while the
 * process tears its statics down. */
std::atomic<bool> g_compiling_in_process{false};
struct ExitGuard { ~ExitGuard(); };
Queue& queue() { static Queue* q = new Queue; static ExitGuard guard; (void)guard; return *q; }
ExitGuard::~ExitGuard()
{
	Queue& q = queue();
	std::unique_lock<std::mutex> g(q.mu);
	q.quit = true;
	q.pending.clear();
	q.cv.wait(g, [] { return !g_compiling_in_process.load(); });
}

void say(const JitKernel* k, const std::string& line) { if (k->log) k->log(line.c_str()); }

/* In this process, through the hipRTC library found here (whatever compiler the host process already holds). */
bool compile_in_process(JitKernel* k, std::string& diagnostics)
{
	const Rtc& r = rtc();
	if (!r.ok) { diagnostics = "no hipRTC library"; return false; }
	std::vector<const char*> texts, names;
	for (int i = 0; i < kHeaderCount; i++) { texts.push_back(kHeaders[i].text); names.push_back(kHeaders[i].name); }
	texts.push_back(k->records.c_str()); names.push_back("fixed_contexts.inc");
	hiprtcProgram prog = nullptr;
	if (r.create(&prog, k->unit.c_str(), "astc_compress_blocks_jit.hip", (int)texts.size(), texts.data(), names.data()) != HIPRTC_SUCCESS) return false;
	std::vector<const char*> opts;
	for (const std::string& o : k->options) opts.push_back(o.c_str());
	bool ok = r.compile(prog, (int)opts.size(), opts.data()) == HIPRTC_SUCCESS;
	if (!ok)
	{
		size_t n = 0;
		if (r.log_size(prog, &n) == HIPRTC_SUCCESS && n > 1) { diagnostics.resize(n); (void)r.log(prog, &diagnostics[0]); }
	}
	size_t bytes = 0;
	if (ok) ok = r.code_size(prog, &bytes) == HIPRTC_SUCCESS && bytes > 0;
	if (ok) { k->code.resize(bytes); ok = r.code(prog, k->code.data()) == HIPRTC_SUCCESS; }
	(void)r.destroy(&prog);
	return ok;
}

/* In the compiler process (jitc_main.cpp: why): the source goes through a scratch directory. */
bool compile_in_helper(JitKernel* k, const std::string& helper, std::string& diagnostics)
{
	const std::string dir = temp_dir();
	if (dir.empty()) { diagnostics = "no scratch directory"; return false; }
	bool ok = true;
	std::vector<std::string> files;
	for (int i = 0; i < kHeaderCount && ok; i++)
	{
		files.push_back(dir + "/" + kHeaders[i].name);
		ok = spill(files.back(), kHeaders[i].text, (size_t)(kHeaders[i].end - kHeaders[i].text) - 1);      // (without the terminating zero)
	}
	files.push_back(dir + "/fixed_contexts.inc");
	ok = ok && spill(files.back(), k->records.data(), k->records.size());
	files.push_back(dir + "/unit.hip");
	ok = ok && spill(files.back(), k->unit.data(), k->unit.size());
	// With a disk cache the compiler process writes the build straight into it and removes the scratch directory itself: a
	// host that exits before the compile is done leaves the process to finish, and the next run finds the build.
	const std::string cache = cache_dir();
	if (!cache.empty()) make_dirs(cache);
	const std::string out = cache.empty() ? dir + "/out.hsaco" : cache + "/" + k->key + ".hsaco", err = dir + "/stderr.txt";
	if (ok)
	{
		std::vector<std::string> args = { helper, dir + "/unit.hip", out };
		for (const std::string& o : k->options) args.push_back(o);
		args.push_back("-I" + dir);
		ok = run_helper(args, nullptr, err.c_str(), cache.empty() ? nullptr : dir.c_str()) == 0;
		if (!ok) diagnostics = slurp(err);
	}
	if (ok)
	{
		const std::string code = slurp(out);
		k->code.assign(code.begin(), code.end());
		ok = !k->code.empty();
	}
	// (what the compiler process has not removed: everything without a cache or after a failure)
	for (const std::string& f : files) (void)unlink(f.c_str());
	if (cache.empty()) (void)unlink(out.c_str());
	(void)unlink(err.c_str());
	(void)rmdir(dir.c_str());
	return ok;
}

bool compile_now(JitKernel* k)
{
	const auto t0 = std::chrono::steady_clock::now();
	const std::string helper = helper_path();
	std::string diagnostics;
	if (helper.empty()) g_compiling_in_process.store(true);
	bool ok = helper.empty() ? compile_in_process(k, diagnostics) : compile_in_helper(k, helper, diagnostics);
	if (helper.empty()) { g_compiling_in_process.store(false); queue().cv.notify_all(); }
	k->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	if (!ok)
	{
		k->code.clear();
		say(k, "run-time build " + k->name + " failed to compile: " + diagnostics.substr(0, 600));
		return false;
	}
	long vgprs = 0, scratch = 0;
	if (!acceptable(k->code, &vgprs, &scratch))
	{
		say(k, "run-time build " + k->name + " refused: " + std::to_string(vgprs) + " VGPRs, " + std::to_string(scratch) + " bytes of scratch (compiler: " + compiler_identity() + "); the generic build stays");
		k->code.clear();
		return false;
	}
	// the disk cache (the compiler process has written it already): under a temporary name, renamed into place
	const std::string dir = cache_dir();
	if (!dir.empty() && helper.empty())
	{
		make_dirs(dir);
		const std::string path = dir + "/" + k->key + ".hsaco";
		const std::string tpath = path + ".tmp" + std::to_string((long)getpid());
		if (spill(tpath, k->code.data(), k->code.size())) (void)rename(tpath.c_str(), path.c_str());
		else (void)unlink(tpath.c_str());
	}
	char line[256];
	snprintf(line, sizeof(line), " compiled in %.1f s (%zu bytes, %ld VGPRs, %ld bytes of scratch, %s)", k->seconds, k->code.size(), vgprs, scratch, helper.empty() ? "in process" : "compiler process");
	say(k, "run-time build " + k->name + line);
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

bool jit_write_records(const char* path, const void* layout, size_t layout_bytes, const DeviceConfig& cfg_in, const TableRoot& root)
{
	if (layout_bytes != sizeof(LdsLayout)) return false;
	LdsLayout L;
	memcpy(&L, layout, sizeof(L));
	DeviceConfig cfg = cfg_in;
	cfg.debug_dup_stage = 0;
	const std::string text = records_text(L, cfg, root);
	return spill(path, text.data(), text.size());
}

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
	if (layout_bytes != sizeof(LdsLayout) || (helper_path().empty() && !rtc().ok)) return nullptr;
	LdsLayout L;
	memcpy(&L, layout, sizeof(L));
	DeviceConfig cfg = cfg_in;
	cfg.debug_dup_stage = 0;
	JitKernel* k = new JitKernel;
	k->log = log;
	k->records = records_text(L, cfg, root);
	// the translation unit: what kernel_ldr_6x6m.hip is for its context, with the records from the "header" above
	k->unit = std::string("#define ASTC_VARIANT v_jit\n") + (hdr ? "#define ASTC_ENABLE_HDR 1\n" : "#define ASTC_ENABLE_HDR 0\n") +
	          (root.texel_count <= 64 ? "#define ASTC_TEXELS_LE_64 1\n" : "#define ASTC_FIXED_OPAQUE_TEXEL_COUNT 1\n") +
	          "#define ASTC_FIXED_CONTEXT 1\n"
	          "#define ASTC_KERNEL_NAME astc_compress_blocks_jit\n#define ASTC_KERNEL_LINKAGE extern \"C\"\n#include \"kernel_device.h\"\n";
	// the numerics flags of the Makefile are part of the bit-exactness contract (wave.h)
	std::string a = arch && *arch ? arch : "gfx950";
	a = a.substr(0, a.find(':'));
	k->options = { "--offload-arch=" + a, "-O3", "-std=c++17", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-math-errno",
	               "-fno-slp-vectorize", "-Wno-unused-function" };
	// (ASTCENC_AMD_JIT_OPTIONS: further compiler options, space separated -- a debugging aid: the kernel of a context rebuilt
	//  with another optimisation level or a -D switch of the source without rebuilding the library; part of the cache key)
	if (const char* more = getenv("ASTCENC_AMD_JIT_OPTIONS"))
	{
		std::string word;
		for (const char* p = more;; p++)
		{
			if (*p == ' ' || *p == 0) { if (!word.empty()) k->options.push_back(word); word.clear(); if (!*p) break; }
			else word.push_back(*p);
		}
	}
	uint64_t h = 14695981039346656037ull;
	for (int i = 0; i < kHeaderCount; i++) h = fnv1a(h, kHeaders[i].text, (size_t)(kHeaders[i].end - kHeaders[i].text));
	h = fnv1a(h, k->records.data(), k->records.size());
	h = fnv1a(h, k->unit.data(), k->unit.size());
	for (const std::string& o : k->options) h = fnv1a(h, o.data(), o.size() + 1);
	h = fnv1a(h, compiler_identity().data(), compiler_identity().size());
	char hex[24];
	snprintf(hex, sizeof(hex), "%016llx", (unsigned long long)h);
	k->key = hex;
	k->name = std::string(JIT_ENTRY_POINT) + "_" + hex;

	Queue& q = queue();
	std::lock_guard<std::mutex> g(q.mu);
	auto it = q.by_key.find(k->key);
	if (it != q.by_key.end()) { delete k; it->second->refs++; return it->second; }
	if (load_cached(k))
	{
		// (a build in the cache that misses the occupancy -- written by a compiler process whose host was gone before it could
		//  look at the result -- is not compiled again by every process that comes by: it fails here, once per process)
		long vgprs = 0, scratch = 0;
		if (acceptable(k->code, &vgprs, &scratch)) k->state = JIT_READY;
		else
		{
			say(k, "run-time build " + k->name + " in the disk cache refused: " + std::to_string(vgprs) + " VGPRs, " + std::to_string(scratch) + " bytes of scratch; the generic build stays");
			k->code.clear();
			k->state = JIT_FAILED;
		}
	}
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
