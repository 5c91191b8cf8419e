// SPDX-License-Identifier: Apache-2.0
// The compression kernel compiled at run time for ONE context (kernel_jit.cpp): what the three fixed-context builds of the
// library are for the BASELINE contexts -- LdsLayout, DeviceConfig and TableRoot as compile-time constants, DESIGN.md
// section 3.1 -- for any context: every preset, footprint, profile, flag and hand-edited tuning field
// (ref: astcenc_config_init treats them all alike, Source/astcenc_entry.cpp:504-724).  The library embeds its own device
// source (kernel_device.h and the wave_*.h it includes), writes the context's records as `constexpr` initializers, compiles
// the translation unit with hipRTC for the device's architecture on a background thread and keeps the code object on disk
// under a hash of (source, records, options, the compiler's files).  A host that exits while a compile is running leaves at
// once; the compiler process finishes on its own and the next run finds the build.  Until the build is there -- and whenever it cannot be made:
// no hipRTC on the box, a compile error -- the context runs the generic build of its footprint class; both give the same
// bytes (tests/test_jit.py).
#pragma once
#include "astc_tables.h"
#include <stddef.h>

namespace astcd {

struct JitKernel;      // one specialised build, shared by the contexts that ask for the same records

enum JitMode {
	JIT_OFF = 0,       // ASTCENC_AMD_JIT=off
	JIT_LAZY = 1,      // default: a build found in the disk cache is used at once; otherwise the compile is queued when the
	                   // context has compressed JIT_LAZY_BLOCKS blocks (a short-lived process never pays for it)
	JIT_EAGER = 2,     // ASTCENC_AMD_JIT=eager: queued when the context is created
	JIT_SYNC = 3       // ASTCENC_AMD_JIT=sync: compiled inside astcenc_context_alloc
};
constexpr unsigned long long JIT_LAZY_BLOCKS = 1u << 18;      // (one 3072 x 3072 texture at 6x6)

enum JitState { JIT_IDLE = 0, JIT_QUEUED, JIT_COMPILING, JIT_READY, JIT_FAILED };

JitMode jit_mode_from_environment();
/* The build for these records (created idle, or ready when the disk cache has it).  `arch`: the device's gcnArchName.
 * Returns null when run-time builds are not possible in this process (no hipRTC library). */
JitKernel* jit_acquire(const void* layout, size_t layout_bytes, const DeviceConfig& cfg, const TableRoot& root, bool hdr, const char* arch,
                       void (*log)(const char* line));
void jit_release(JitKernel* k);             // (a queued compile nobody waits for any more is dropped)
void jit_start(JitKernel* k);               // queue the compile (no-op unless idle)
JitState jit_wait(JitKernel* k);            // queue it if idle, then block until ready or failed
JitState jit_state(const JitKernel* k);
const void* jit_code(const JitKernel* k, size_t* bytes);    // the code object (ready builds)
const char* jit_kernel_name(const JitKernel* k);            // "astc_compress_blocks_jit_<hash>" (what astcenc_amd_context_kernel_name reports)
double jit_compile_seconds(const JitKernel* k);             // 0 for a build that came from the disk cache
/* The records of a context as the text a fixed-context build includes (what the run-time build is compiled with): written to
 * `path` (test infrastructure: the sequential build compiled for one context, oracle/emu/Makefile `fixed`). */
bool jit_write_records(const char* path, const void* layout, size_t layout_bytes, const DeviceConfig& cfg, const TableRoot& root);
constexpr const char* JIT_ENTRY_POINT = "astc_compress_blocks_jit";      // the kernel's symbol in every such code object

} // namespace astcd
