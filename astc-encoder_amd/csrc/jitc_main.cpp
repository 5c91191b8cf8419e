// SPDX-License-Identifier: Apache-2.0
// astcenc_amd_jitc: the compiler process of the library's run-time kernel builds (kernel_jit.cpp).
//
//   astcenc_amd_jitc --identity                         prints one line naming the compiler this process would use
//   astcenc_amd_jitc <unit.hip> <out.hsaco> <option>...  compiles the translation unit with hipRTC (includes resolved through
//                                                        the -I options), writes the code object, exit status 0 / 1;
//                                                        diagnostics on stderr
//
// Why a process of its own: (1) the host application may already hold ANOTHER compiler under the same sonames -- a PyTorch
// process has the libhiprtc.so.7 / libamd_comgr.so.3 of the ROCm release PyTorch was built with, and the ROCm 7.0 compiler
// gives the stage functions of this kernel 160 VGPRs where the kernel's occupancy bound allows 128 (three waves per SIMD
// instead of four: slower than the generic build); a fresh process resolves the sonames to the system's ROCm, the one the
// library's own kernels were built with.  (2) LLVM stays out of the host's address space (comgr is 160 MB), and a host that
// exits while a compile is running does not have to wait for it: the library kills this process.
#include <hip/hiprtc.h>
#include <dlfcn.h>
#include <link.h>
#include <dirent.h>
#include <sys/stat.h>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {
template <typename F> F sym(void* h, const char* name) { return reinterpret_cast<F>(dlsym(h, name)); }
void* open_rtc()
{
	const char* names[] = { getenv("ASTCENC_AMD_HIPRTC"), "libhiprtc.so.7", "libhiprtc.so", "/opt/rocm/lib/libhiprtc.so" };
	for (const char* n : names)
		if (n && *n) if (void* h = dlopen(n, RTLD_NOW | RTLD_GLOBAL)) return h;
	return nullptr;
}
std::string path_of(void* handle)
{
	struct link_map* lm = nullptr;
	if (handle && dlinfo(handle, RTLD_DI_LINKMAP, &lm) == 0 && lm && lm->l_name) return lm->l_name;
	return "?";
}
long size_of(const std::string& path) { struct stat st; return stat(path.c_str(), &st) == 0 ? (long)st.st_size : -1; }
}

int main(int argc, char** argv)
{
	void* rtc = open_rtc();
	if (!rtc) { fprintf(stderr, "astcenc_amd_jitc: no hipRTC library (%s)\n", dlerror()); return 1; }
	auto create = sym<decltype(&hiprtcCreateProgram)>(rtc, "hiprtcCreateProgram");
	auto compile = sym<decltype(&hiprtcCompileProgram)>(rtc, "hiprtcCompileProgram");
	auto code_size = sym<decltype(&hiprtcGetCodeSize)>(rtc, "hiprtcGetCodeSize");
	auto code = sym<decltype(&hiprtcGetCode)>(rtc, "hiprtcGetCode");
	auto log_size = sym<decltype(&hiprtcGetProgramLogSize)>(rtc, "hiprtcGetProgramLogSize");
	auto log = sym<decltype(&hiprtcGetProgramLog)>(rtc, "hiprtcGetProgramLog");
	auto version = sym<decltype(&hiprtcVersion)>(rtc, "hiprtcVersion");
	if (!create || !compile || !code_size || !code || !log_size || !log) { fprintf(stderr, "astcenc_amd_jitc: hipRTC lacks a symbol\n"); return 1; }

	if (argc == 2 && strcmp(argv[1], "--identity") == 0)
	{
		// hipRTC loads its compiler (comgr) on first use: compile nothing, then name what got loaded
		hiprtcProgram p = nullptr;
		if (create(&p, "extern \"C\" __global__ void astc_identity() {}\n", "identity.hip", 0, nullptr, nullptr) == HIPRTC_SUCCESS)
		{
			const char* o[] = { "--offload-arch=gfx950" };
			(void)compile(p, 1, o);
		}
		int major = 0, minor = 0;
		if (version) (void)version(&major, &minor);
		std::string comgr = "?";
		for (const char* n : { "libamd_comgr.so.3", "libamd_comgr.so", "libamd_comgr.so.2" })
			if (void* h = dlopen(n, RTLD_NOW | RTLD_NOLOAD)) { comgr = path_of(h); break; }
		const std::string self = path_of(rtc);
		printf("hiprtc %d.%d %s %ld comgr %s %ld\n", major, minor, self.c_str(), size_of(self), comgr.c_str(), size_of(comgr));
		return 0;
	}
	if (argc < 3) { fprintf(stderr, "usage: astcenc_amd_jitc <unit.hip> <out.hsaco> <option>... | --identity\n"); return 1; }

	std::string unit;
	{
		FILE* f = fopen(argv[1], "rb");
		if (!f) { fprintf(stderr, "astcenc_amd_jitc: cannot read %s\n", argv[1]); return 1; }
		char buf[65536]; size_t n;
		while ((n = fread(buf, 1, sizeof(buf), f)) > 0) unit.append(buf, n);
		fclose(f);
	}
	hiprtcProgram p = nullptr;
	if (create(&p, unit.c_str(), "astc_compress_blocks_jit.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) { fprintf(stderr, "astcenc_amd_jitc: hiprtcCreateProgram failed\n"); return 1; }
	std::vector<const char*> opts(argv + 3, argv + argc);
	const hiprtcResult rc = compile(p, (int)opts.size(), opts.data());
	if (rc != HIPRTC_SUCCESS)
	{
		size_t n = 0;
		std::string text;
		if (log_size(p, &n) == HIPRTC_SUCCESS && n > 1) { text.resize(n); (void)log(p, &text[0]); }
		fprintf(stderr, "%.4000s\n", text.c_str());
		return 1;
	}
	size_t bytes = 0;
	if (code_size(p, &bytes) != HIPRTC_SUCCESS || bytes == 0) return 1;
	std::vector<char> out(bytes);
	if (code(p, out.data()) != HIPRTC_SUCCESS) return 1;
	// written under a temporary name and renamed into place: the output may be the library's disk cache, read by other processes
	const std::string final_path = argv[2], tmp_path = final_path + ".tmp" + std::to_string((long)getpid());
	FILE* f = fopen(tmp_path.c_str(), "wb");
	if (!f) { fprintf(stderr, "astcenc_amd_jitc: cannot write %s\n", tmp_path.c_str()); return 1; }
	const bool ok = fwrite(out.data(), 1, bytes, f) == bytes;
	if (fclose(f) != 0 || !ok || rename(tmp_path.c_str(), final_path.c_str()) != 0) { (void)unlink(tmp_path.c_str()); return 1; }
	// The library that started this process may be gone by now (a host that exits does not wait for its compile: the build is
	// for the next run): the scratch directory with the source is this process's to remove.
	if (const char* dir = getenv("ASTCENC_AMD_JITC_SCRATCH"))
	{
		if (DIR* d = opendir(dir))
		{
			while (struct dirent* e = readdir(d))
				if (strcmp(e->d_name, ".") != 0 && strcmp(e->d_name, "..") != 0) (void)unlink((std::string(dir) + "/" + e->d_name).c_str());
			closedir(d);
			(void)rmdir(dir);
		}
	}
	return 0;
}
