// SPDX-License-Identifier: Apache-2.0
// TEST INFRASTRUCTURE ONLY -- CPU "wave emulator" backend.
//
// Compiles the very same wave_*.h source that the HIP kernels are built from with ASTC_WAVE_EMU
// semantics (WV_FOR = sequential loop, WV_SYNC = no-op) so that the block compressor can be
// debugged against oracle/_ref on machines without a GPU.  It is linked only into
// oracle/emu/_build/libastcenc_emu.so and is never part of the product library.
#include "backend.h"
#include "kernel_jit.h"
#include "wave_block.h"
#include "wave_decode.h"
#include "wave_alpha.h"
#include "wave_metrics.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace astcd {

inline namespace ASTC_VARIANT { thread_local const Ctx* g_wave_ctx = nullptr; }
}
thread_local bool g_wave_one_trip_texel_loops = false;
namespace astcd {

struct Backend {
	std::vector<uint8_t> blob;      // tables + DeviceConfig + LdsLayout, like the device copy
	DeviceConfig cfg;
	LdsLayout layout;
	JitKernel* jit = nullptr;       // (backend_specialize below: compile check only)
};

Backend* backend_create(const uint8_t* blob, size_t blob_bytes, const DeviceConfig& cfg, int* status)
{
	Backend* b = new Backend;
	b->cfg = cfg;
	make_lds_layout(*reinterpret_cast<const TableRoot*>(blob), cfg, b->layout);
	// same arrangement as the device copy: [LdsLayout][DeviceConfig][table blob]
	b->blob.assign(CTX_LAYOUT_BACK + blob_bytes, 0);
	memcpy(b->blob.data(), &b->layout, sizeof(LdsLayout));
	memcpy(b->blob.data() + (CTX_LAYOUT_BACK - CTX_CONFIG_BACK), &cfg, sizeof(cfg));
	memcpy(b->blob.data() + CTX_LAYOUT_BACK, blob, blob_bytes);
	if (getenv("ASTC_EMU_DUMP_LAYOUT"))
	{
		const LdsLayout& L = b->layout;
		const TableRoot& r = *reinterpret_cast<const TableRoot*>(blob);
		fprintf(stderr, "lds layout: total %u | data %u blk %u scb %u trial %u ei_w %u ei_wes %u ptab %u candw %u | phase begin %u | search: dwi %u lowhigh %u modes %u uni %u (+%u) | refine: dtab %u ctab %u qtab %u rsc %u tsc_r %u wsc %u | part: part %u tsc_p %u part_tabs %u (chunk %u) | max_dec_table %u realign_rt %u | batch: cstate %u stride %u per batch %u / %u\n",
		        L.total, L.data, L.blk, L.scb, L.trial, L.ei_w, L.ei_wes, L.ptab, L.candw, L.dwi, L.dwi, L.lowhigh, L.modes, L.uni, L.uni_bytes,
		        L.dtab, L.ctab, L.qtab, L.rsc, L.tsc_r, L.wsc, L.part, L.tsc_p, L.part_tabs, L.part_chunk, r.max_decimation_table_bytes, r.realign_rt_floats, L.cstate, L.cstate_stride, L.bat_max[0], L.bat_max[1]);
	}
	if (const char* path = getenv("ASTC_EMU_DUMP_FIXED_CONTEXT"))
	{
		// tools/gen_fixed_contexts.py: the context's three records as the aggregate initializers of fixed_contexts.inc
		// ("path:name"; appended).  LdsLayout is 32-bit words throughout, TableRoot four bytes and then words.
		std::string spec(path);
		const size_t colon = spec.rfind(':');
		FILE* f = fopen(spec.substr(0, colon).c_str(), "a");
		if (f)
		{
			const TableRoot& r = *reinterpret_cast<const TableRoot*>(blob);
			fprintf(f, "#if defined(ASTC_FIXED_%s)\n", spec.substr(colon + 1).c_str());
			static_assert(sizeof(LdsLayout) % 4 == 0 && sizeof(TableRoot) % 4 == 0, "records are word arrays");
			fprintf(f, "constexpr LdsLayout kFixedLayout = {");
			for (size_t i = 0; i < sizeof(LdsLayout) / 4; i++) fprintf(f, "%s%uu", i ? ", " : " ", reinterpret_cast<const uint32_t*>(&b->layout)[i]);
			fprintf(f, " };\n");
			static_assert(sizeof(DeviceConfig) == 92, "DeviceConfig changed: update this dump");
			fprintf(f, "constexpr DeviceConfig kFixedConfig = { %d, %uu, { %af, %af, %af, %af }, %af, %uu, { %uu, %uu, %uu }, %uu, %uu, { %uu, %uu, %uu }, %af, %af, { %af, %af }, %af, %af, 0u };\n",
			        cfg.profile, cfg.flags, cfg.cw[0], cfg.cw[1], cfg.cw[2], cfg.cw[3], cfg.rgbm_m_scale, cfg.tune_partition_count_limit,
			        cfg.tune_partition_index_limit[0], cfg.tune_partition_index_limit[1], cfg.tune_partition_index_limit[2],
			        cfg.tune_refinement_limit, cfg.tune_candidate_limit, cfg.tune_partitioning_candidate_limit[0],
			        cfg.tune_partitioning_candidate_limit[1], cfg.tune_partitioning_candidate_limit[2], cfg.tune_db_limit, cfg.tune_mse_overshoot,
			        cfg.tune_partition_early_out_limit_factor[0], cfg.tune_partition_early_out_limit_factor[1],
			        cfg.tune_2plane_early_out_limit_correlation, cfg.tune_search_mode0_enable);
			fprintf(f, "constexpr TableRoot kFixedRoot = { %u, %u, %u, %u", r.dim_x, r.dim_y, r.texel_count, r.dim_z);
			for (size_t i = 1; i < sizeof(TableRoot) / 4; i++) fprintf(f, ", %uu", reinterpret_cast<const uint32_t*>(&r)[i]);
			fprintf(f, " };\n#endif\n");
			fclose(f);
		}
	}
	// tests/test_emu_fixed.py: the context's records as a fixed-context build includes them (what its run-time build is compiled with)
	if (const char* path = getenv("ASTC_EMU_DUMP_RECORDS")) (void)jit_write_records(path, &b->layout, sizeof(b->layout), cfg, *reinterpret_cast<const TableRoot*>(blob));
#if ASTC_FIXED
	// this library is compiled for ONE context (ASTC_FIXED_RECORDS_FILE): any other is refused, like a fixed-context kernel build does
	{
		DeviceConfig live = cfg;
		live.debug_dup_stage = kFixedConfig.debug_dup_stage;
		if (memcmp(&b->layout, &kFixedLayout, sizeof(LdsLayout)) != 0 || memcmp(&live, &kFixedConfig, sizeof(live)) != 0 || memcmp(blob, &kFixedRoot, sizeof(TableRoot)) != 0)
		{
			delete b; *status = 2; return nullptr;
		}
	}
#endif
	*status = 0;
	return b;
}

void backend_destroy(Backend* b) { if (b) jit_release(b->jit); delete b; }
int backend_device_count(const Backend*) { return 1; }
const char* backend_name() { return "emu:cpu"; }
void backend_set_log_callback(void (*)(const char*)) {}      // (the sequential build has nothing to report)
const char* backend_kernel_name(const Backend* b) { return b->jit && jit_state(b->jit) == JIT_READY ? jit_kernel_name(b->jit) : "emu"; }
/* The run-time build of the context's kernel, compiled for gfx950 on this CPU-only box (hipRTC needs no device) but never
 * launched: tests/test_jit.py checks on the CPU that the library's embedded device source compiles for a context's records. */
int backend_specialize(Backend* b)
{
	if (!b->jit)
	{
		const TableRoot* root = reinterpret_cast<const TableRoot*>(b->blob.data() + CTX_LAYOUT_BACK);
		const DeviceConfig* cfg = reinterpret_cast<const DeviceConfig*>(b->blob.data() + (CTX_LAYOUT_BACK - CTX_CONFIG_BACK));
		b->jit = jit_acquire(&b->layout, sizeof(b->layout), *cfg, *root, cfg->profile >= 2, "gfx950", [](const char* line) { fprintf(stderr, "emu jit: %s\n", line); });
	}
	return b->jit && jit_wait(b->jit) == JIT_READY ? 0 : 1;
}

int backend_compress(Backend* b, const CompressJob& job)
{
	const TableRoot* root = reinterpret_cast<const TableRoot*>(b->blob.data() + CTX_LAYOUT_BACK);
	Ctx c;
	c.tab = b->blob.data() + CTX_LAYOUT_BACK;
	c.tab_constant = false;
	c.root = root;
	c.cfg = reinterpret_cast<const DeviceConfig*>(c.tab - CTX_CONFIG_BACK);
	c.L = reinterpret_cast<const LdsLayout*>(c.tab - CTX_LAYOUT_BACK);
#if ASTC_FIXED
	c.root = &kFixedRoot; c.cfg = &kFixedConfig; c.L = &kFixedLayout;      // (as kernel_device.h does in a fixed-context build)
#endif
	// LDS starts out as garbage on the device: poison it (ASTC_EMU_POISON = byte value, or "rand")
	std::vector<uint8_t> lds(c.L->total + 64, 0xCD);
	if (const char* poison = getenv("ASTC_EMU_POISON"))
	{
		if (poison[0] == 'r') { uint32_t x = 0x2545F491u; for (auto& v : lds) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; v = (uint8_t)(x >> 11); } }
		else memset(lds.data(), atoi(poison), lds.size());
	}
	c.lds = lds.data();
	c.T = root->texel_count;
	c.Tp = (c.T + 3) & ~3;
	c.Ts = lds_row_stride(c.Tp);
	c.prof = nullptr;
	g_wave_ctx = &c;
	g_wave_one_trip_texel_loops = root->texel_count <= 64;      // what kernel_{ldr,hdr}64.hip assume (wave.h: WV_FOR_T)

	const uint32_t dim_z = job.dim_z ? job.dim_z : 1u;
	const size_t slice_bytes = (size_t)job.dim_x * job.dim_y * (job.data_type == 0 ? 4 : job.data_type == 1 ? 8 : 16);
	std::vector<uint8_t> volume;               // the slices back to back, as the device copy has them
	ImageDesc img;
	img.data = job.device_data;
	if (job.host_slices)
	{
		img.data = job.host_slices[0];
		if (dim_z > 1)
		{
			volume.resize(slice_bytes * dim_z);
			for (uint32_t z = 0; z < dim_z; z++) memcpy(volume.data() + z * slice_bytes, job.host_slices[z], slice_bytes);
			img.data = volume.data();
		}
	}
	img.dim_x = job.dim_x; img.dim_y = job.dim_y;
	img.data_type = job.data_type;
	for (int i = 0; i < 4; i++) img.swz[i] = job.swz[i];
	img.blocks_x = (job.dim_x + root->dim_x - 1) / root->dim_x;
	img.blocks_y = (job.dim_y + root->dim_y - 1) / root->dim_y;
	img.dim_z = dim_z; img.blocks_z = (dim_z + root->dim_z - 1) / root->dim_z;
	bool needs_swz = job.swz[0] != 0 || job.swz[1] != 1 || job.swz[2] != 2 || job.swz[3] != 3;
	bool hdr = b->cfg.profile >= 2;
	img.use_fast_load = (!needs_swz && !hdr && job.data_type == 0 && root->dim_z == 1) ? 1 : 0;
	img.fast_load_slice0 = job.fast_load_slice0;
	img.alpha_avg = nullptr;
	img.a_scale_radius = job.a_scale_radius;
	std::vector<float> averages;
	if (job.a_scale_radius != 0)
	{
		averages.assign((size_t)job.dim_x * job.dim_y, 0.0f);
		AlphaJob aj;
		aj.image = img.data; aj.averages = averages.data();
		aj.dim_x = job.dim_x; aj.dim_y = job.dim_y; aj.dim_z = dim_z; aj.data_type = job.data_type;
		aj.swz_a = job.swz[3]; aj.radius = job.a_scale_radius;
		const uint32_t tile = (uint32_t)alpha_tile_size(aj);
		std::vector<float> buf(alpha_scratch_floats(aj));
		for (uint32_t ty = 0; ty < (job.dim_y + tile - 1) / tile; ty++)
			for (uint32_t tx = 0; tx < (job.dim_x + tile - 1) / tile; tx++)
				alpha_average_tile(aj, tx, ty, buf.data());
		img.alpha_avg = averages.data();
	}

	uint8_t* out = job.host_out ? job.host_out : job.device_out;
#if defined(ASTC_TRACE)
	std::vector<uint32_t> trace((size_t)img.blocks_x * img.blocks_y * img.blocks_z * TRACE_WORDS_PER_BLOCK, 0u);
#endif
	const char* only = getenv("ASTC_EMU_ONLY_BLOCK");
	long only_idx = only ? atol(only) : -1;
	for (uint32_t row = 0; row < img.blocks_y * img.blocks_z; row++)
	{
		if (job.cancel_flag && job.cancel_flag->load()) break;
		const uint32_t bz = row / img.blocks_y, by = row - bz * img.blocks_y;
		for (uint32_t bx = 0; bx < img.blocks_x; bx++)
		{
			size_t idx = (size_t)row * img.blocks_x + bx;
			if (only_idx >= 0 && (long)idx != only_idx) continue;
#if defined(ASTC_TRACE)
			c.prof = reinterpret_cast<unsigned long long*>(trace.data() + idx * TRACE_WORDS_PER_BLOCK);
#endif
			if (img.alpha_avg && !block_has_visible_alpha(c, img, bx, by)) load_transparent_block(c);
			else load_block(c, img, bx, by, bz);
			c.blk().block_index = (uint32_t)idx;
			compress_block(c, out);
		}
		if (job.progress) job.progress(100.0f * (float)(row + 1) / (float)(img.blocks_y * img.blocks_z));
	}
#if defined(ASTC_TRACE)
	if (const char* path = getenv("ASTCENC_AMD_TRACE_FILE"))
		if (FILE* f = fopen(path, "wb")) { fwrite(trace.data(), sizeof(uint32_t), trace.size(), f); fclose(f); }
#endif
	if (job.kernel_ms) *job.kernel_ms = 0.0f;
	return 0;
}

int backend_decompress(Backend* b, const DecompressJob& job)
{
	const TableRoot* root = reinterpret_cast<const TableRoot*>(b->blob.data() + CTX_LAYOUT_BACK);
	const uint32_t dim_z = job.dim_z ? job.dim_z : 1u;
	const size_t slice_bytes = (size_t)job.dim_x * job.dim_y * (job.data_type == 0 ? 4 : job.data_type == 1 ? 8 : 16);
	std::vector<uint8_t> volume(dim_z > 1 ? slice_bytes * dim_z : 0);   // the slices back to back, as the device copy has them
	DecodeImage img;
	img.data = dim_z > 1 ? static_cast<void*>(volume.data()) : job.host_slices[0];
	img.dim_x = job.dim_x; img.dim_y = job.dim_y; img.dim_z = dim_z;
	img.data_type = job.data_type;
	for (int i = 0; i < 4; i++) img.swz[i] = job.swz[i];
	img.block_x = root->dim_x; img.block_y = root->dim_y; img.block_z = root->dim_z;
	img.blocks_x = (job.dim_x + root->dim_x - 1) / root->dim_x;
	img.blocks_y = (job.dim_y + root->dim_y - 1) / root->dim_y;
	img.blocks_z = (dim_z + root->dim_z - 1) / root->dim_z;
	img.profile = b->cfg.profile;
	decode_image_prepare(img);
	// the same batched routine the kernel runs (decode_row_batch): runs of DECODE_BATCH blocks of a block row
	std::vector<DecodeTables> tabs(1);
	decode_tables_build(tabs[0], (int)root->dim_x, (int)root->dim_y, (int)root->dim_z);
	img.tabs = tabs.data();
	std::vector<DecodeBatch> batch(1);
	memset(static_cast<void*>(batch.data()), 0xCD, sizeof(DecodeBatch));
	for (uint32_t bz = 0; bz < img.blocks_z; bz++)
		for (uint32_t by = 0; by < img.blocks_y; by++)
			for (uint32_t bx0 = 0; bx0 < img.blocks_x; bx0 += (uint32_t)DECODE_BATCH)
			{
				const uint32_t left = img.blocks_x - bx0;
				decode_row_batch(img, job.host_blocks, bx0, by, bz, (int)(left < (uint32_t)DECODE_BATCH ? left : (uint32_t)DECODE_BATCH), batch[0]);
			}
	if (dim_z > 1)
		for (uint32_t z = 0; z < dim_z; z++) memcpy(job.host_slices[z], volume.data() + z * slice_bytes, slice_bytes);
	return 0;
}

int backend_decompress_device(Backend* b, const DecompressDeviceJob& job)
{
	// "device" memory of the emulator is host memory: one contiguous volume, decode straight into it
	const uint32_t dim_z = job.dim_z ? job.dim_z : 1u;
	const size_t slice_bytes = (size_t)job.dim_x * job.dim_y * (job.data_type == 0 ? 4 : job.data_type == 1 ? 8 : 16);
	std::vector<void*> slices(dim_z);
	for (uint32_t z = 0; z < dim_z; z++) slices[z] = static_cast<uint8_t*>(job.device_image) + z * slice_bytes;
	DecompressJob h;
	memset(&h, 0, sizeof(h));
	h.host_blocks = job.device_blocks;
	h.host_slices = slices.data();
	h.dim_x = job.dim_x; h.dim_y = job.dim_y; h.dim_z = dim_z; h.data_type = job.data_type;
	for (int i = 0; i < 4; i++) h.swz[i] = job.swz[i];
	return backend_decompress(b, h);
}

int backend_compare(Backend*, const CompareJob& job)
{
	for (int k = 0; k < METRIC_SUMS_HOST; k++) job.sums[k] = 0.0;
	for (size_t t = 0; t < job.texels; t++)
	{
		float e[8], c1[4], c2[4];
		float m = metric_texel_terms(job.device_a, job.type_a, job.device_b, job.type_b, t, nullptr, e, c1, c2);
		for (int k = 0; k < 8; k++) job.sums[k] += (double)e[k];
		if ((double)m > job.sums[8]) job.sums[8] = (double)m;
		if (job.hdr)
		{
			float h[8];
			metric_hdr_terms(c1, c2, job.fstop_lo, job.fstop_hi, h);
			for (int k = 0; k < 8; k++) job.sums[METRIC_HDR_FIRST + k] += (double)h[k];
		}
	}
	return 0;
}

} // namespace astcd
