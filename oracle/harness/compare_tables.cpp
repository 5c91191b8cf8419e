// TEST INFRASTRUCTURE ONLY -- never linked into the product.
// Field-by-field comparison of this project's table blob (astc-encoder_amd/csrc/host_tables.cpp)
// against the reference encoder's block_size_descriptor and static data tables.  The reference is
// used where it lies: headers are included from /root/reference/Source and the objects built by
// oracle/Makefile are linked; two reference .cpp files are #included to reach file-static tables.
//
// usage: compare_tables <block_x> <block_y> <quality> [block_z]      exit code 0 = identical
#include "astcenc_internal.h"
#include "astcenc_internal_entry.h"
#include "astcenc_integer_sequence.cpp"   // file-static integer_of_trits / integer_of_quints
#include "astcenc_weight_align.cpp"       // file-static sin_table / cos_table

#include "host_tables.h"
#include <cstdio>
#include <cstdlib>

static int fails = 0;
#define CHECK(cond, ...) do { if (!(cond)) { if (fails < 20) { printf("MISMATCH: " __VA_ARGS__); printf("\n"); } fails++; } } while (0)

int main(int argc, char** argv)
{
	unsigned bx = argc > 1 ? atoi(argv[1]) : 6, by = argc > 2 ? atoi(argv[2]) : 6;
	float quality = argc > 3 ? (float)atof(argv[3]) : 60.0f;
	unsigned bz = argc > 4 ? atoi(argv[4]) : 1;

	astcenc_config cfg;
	if (astcenc_config_init(ASTCENC_PRF_LDR, bx, by, bz, quality, 0, &cfg)) { printf("config_init failed\n"); return 2; }
	astcenc_context* ctx;
	if (astcenc_context_alloc(&cfg, 1, &ctx, nullptr)) { printf("context_alloc failed\n"); return 2; }
	const block_size_descriptor& bsd = *ctx->context.bsd;
	const astcenc_config& c = ctx->context.config;

	std::vector<uint8_t> blob; astcd::HostTables host;
	if (!astcd::build_tables(bx, by, bz, c.tune_partition_count_limit, (float)c.tune_block_mode_limit / 100.0f, blob, host)) { printf("build_tables failed\n"); return 2; }
	const uint8_t* B = blob.data();
	const astcd::TableRoot& r = *(const astcd::TableRoot*)B;

	CHECK(r.texel_count == bsd.texel_count && r.dim_x == bsd.dim_x && r.dim_y == bsd.dim_y && r.dim_z == bsd.dim_z, "footprint");
	CHECK(r.block_mode_count_1plane_always == bsd.block_mode_count_1plane_always, "bm always %u %u", r.block_mode_count_1plane_always, bsd.block_mode_count_1plane_always);
	CHECK(r.block_mode_count_1plane_selected == bsd.block_mode_count_1plane_selected, "bm 1p %u %u", r.block_mode_count_1plane_selected, bsd.block_mode_count_1plane_selected);
	CHECK(r.block_mode_count_1plane_2plane_selected == bsd.block_mode_count_1plane_2plane_selected, "bm 2p");
	CHECK(r.decimation_mode_count_always == bsd.decimation_mode_count_always, "dm always");
	CHECK(r.decimation_mode_count_selected == bsd.decimation_mode_count_selected, "dm sel %u %u", r.decimation_mode_count_selected, bsd.decimation_mode_count_selected);

	const astcd::BlockMode* bms = (const astcd::BlockMode*)(B + r.off_block_modes);
	for (unsigned i = 0; i < bsd.block_mode_count_1plane_2plane_selected; i++)
	{
		const block_mode& m = bsd.block_modes[i];
		CHECK(bms[i].mode_index == m.mode_index && bms[i].decimation_mode == m.decimation_mode &&
		      bms[i].quant_mode == m.quant_mode && bms[i].weight_bits == m.weight_bits &&
		      bms[i].is_dual_plane == m.is_dual_plane, "block mode %u", i);
	}

	const astcd::DecimationMode* dms = (const astcd::DecimationMode*)(B + r.off_decimation_modes);
	const astcd::DecimationInfo* dis = (const astcd::DecimationInfo*)(B + r.off_decimation_infos);
	unsigned T = bsd.texel_count;
	for (unsigned i = 0; i < bsd.decimation_mode_count_selected; i++)
	{
		const decimation_mode& m = bsd.decimation_modes[i];
		CHECK(dms[i].maxprec_1plane == m.maxprec_1plane && dms[i].maxprec_2planes == m.maxprec_2planes, "dm prec %u", i);
		// ref bits set by unselected (4th pass) block modes only add redundant work in the reference;
		// every bit we set must be set there.
		CHECK((dms[i].refprec_1plane & ~m.refprec_1plane) == 0 && (dms[i].refprec_2planes & ~m.refprec_2planes) == 0, "dm ref %u", i);
		const decimation_info& d = bsd.decimation_tables[i];
		const astcd::DecimationInfo& e = dis[i];
		unsigned W = d.weight_count;
		CHECK(e.texel_count == d.texel_count && e.weight_count == d.weight_count && e.max_texel_weight_count == d.max_texel_weight_count &&
		      (bz > 1 || (e.weight_x == d.weight_x && e.weight_y == d.weight_y)), "di hdr %u", i);
		const uint8_t* tw = B + e.off_texel_weights; const uint8_t* tci = B + e.off_texel_contribs_int;
		const float* tcf = (const float*)(B + e.off_texel_contribs_f);
		for (unsigned t = 0; t < T; t++) for (unsigned j = 0; j < 4; j++)
		{
			// (one record per texel here, transposed in the reference)
			CHECK(tw[t * 4 + j] == d.texel_weights_tr[j][t], "di %u tw", i);
			CHECK(tci[t * 4 + j] == d.texel_weight_contribs_int_tr[j][t], "di %u tci", i);
			CHECK(tcf[t * 4 + j] == d.texel_weight_contribs_float_tr[j][t], "di %u tcf", i);
		}
		CHECK((e.off_texel_weights & 15u) == 0 && (e.off_texel_contribs_f & 15u) == 0, "di %u texel table alignment", i);
		const uint8_t* wtc = B + e.off_weight_texel_count; const uint8_t* wt = B + e.off_weight_texels;
		const float* wc = (const float*)(B + e.off_weight_contribs); const float* tcw = (const float*)(B + e.off_texel_contrib_for_weight);
		for (unsigned w = 0; w < W; w++)
		{
			CHECK(wtc[w] == d.weight_texel_count[w], "di %u wtc", i);
			for (unsigned j = 0; j < d.weight_texel_count[w]; j++)
			{
				CHECK(wt[j * W + w] == d.weight_texels_tr[j][w], "di %u wt", i);
				CHECK(wc[j * W + w] == d.weights_texel_contribs_tr[j][w], "di %u wc", i);
				CHECK(tcw[j * W + w] == d.texel_contrib_for_weight[j][w], "di %u tcw", i);
			}
		}
	}

	for (unsigned pc = 1; pc <= c.tune_partition_count_limit; pc++)
	{
		CHECK(r.partitioning_count_selected[pc - 1] == bsd.partitioning_count_selected[pc - 1], "pcount %u: %u %u", pc, r.partitioning_count_selected[pc - 1], bsd.partitioning_count_selected[pc - 1]);
		for (unsigned i = 0; i < bsd.partitioning_count_selected[pc - 1]; i++)
		{
			const partition_info& p = bsd.get_raw_partition_info(pc, i);
			const uint8_t* rec = B + r.off_partitions[pc - 1] + i * r.partition_stride;
			const astcd::PartitionHeader& h = *(const astcd::PartitionHeader*)rec;
			CHECK(h.partition_index == p.partition_index && h.partition_count == p.partition_count, "part %u/%u hdr", pc, i);
			const uint8_t* pot = rec + sizeof(astcd::PartitionHeader); const uint8_t* sorted = pot + T;
			unsigned n = 0;
			for (unsigned k = 0; k < pc; k++)
			{
				CHECK(h.texel_count[k] == p.partition_texel_count[k], "part cnt");
				for (unsigned j = 0; j < p.partition_texel_count[k]; j++) CHECK(sorted[n++] == p.texels_of_partition[k][j], "part %u/%u sorted", pc, i);
			}
			for (unsigned t = 0; t < T; t++) CHECK(pot[t] == p.partition_of_texel[t], "part pot");
			if (pc >= 2)
			{
				const uint64_t* cov = (const uint64_t*)(B + r.off_coverage[pc - 1]) + i * pc;
				const uint64_t* rc = pc == 2 ? bsd.coverage_bitmaps_2[i] : pc == 3 ? bsd.coverage_bitmaps_3[i] : bsd.coverage_bitmaps_4[i];
				for (unsigned k = 0; k < pc; k++) CHECK(cov[k] == rc[k], "coverage %u/%u", pc, i);
			}
			if (pc >= 2) CHECK(host.partition_packed_index[(pc - 2) * 1024 + p.partition_index] == i, "packed index");
		}
	}

	const uint8_t* km = B + r.off_kmeans_texels;
	for (unsigned i = 0; i < (T < 64 ? T : 64); i++) CHECK(km[i] == bsd.kmeans_texels[i], "kmeans %u", i);

	const uint8_t* cq = B + r.off_color_unquant_to_uquant; const uint8_t* cp = B + r.off_color_uquant_to_pquant;
	for (unsigned q = 0; q < 17; q++)
	{
		for (unsigned i = 0; i < 512; i++) CHECK(cq[q * 512 + i] == color_unquant_to_uquant_tables[q][i], "cq %u %u", q, i);
		// pquant table only meaningful at representable values
		for (unsigned i = 0; i < 512; i++) { unsigned v = color_unquant_to_uquant_tables[q][i]; CHECK(cp[q * 256 + v] == color_uquant_to_scrambled_pquant_tables[q][v], "cp %u %u", q, v); }
	}
	const astcd::QuantXfer* qx = (const astcd::QuantXfer*)(B + r.off_quant_xfer);
	for (unsigned q = 0; q < 12; q++)
	{
		unsigned n = get_quant_level((quant_method)q);
		for (unsigned i = 0; i < n; i++)
		{
			CHECK(qx[q].quant_to_unquant[i] == quant_and_xfer_tables[q].quant_to_unquant[i], "q2u %u %u", q, i);
			CHECK(qx[q].scramble_map[i] == quant_and_xfer_tables[q].scramble_map[i], "scr %u %u", q, i);
			unsigned v = quant_and_xfer_tables[q].quant_to_unquant[i];
			CHECK(qx[q].prev_next_values[v] == quant_and_xfer_tables[q].prev_next_values[v], "pn %u %u: %x %x", q, v, qx[q].prev_next_values[v], quant_and_xfer_tables[q].prev_next_values[v]);
		}
	}
	const int8_t* qm = (const int8_t*)(B + r.off_quant_mode_table);
	for (unsigned i = 0; i < 10; i++) for (unsigned j = 0; j < 128; j++) CHECK(qm[i * 128 + j] == quant_mode_table[i][j], "qmt %u %u: %d %d", i, j, qm[i * 128 + j], quant_mode_table[i][j]);
	// the levels never rise with the integer-pair count (format selection relies on it: wave_format.h, "stop at the first
	// level that is too low" == "skip every level that is too low")
	for (unsigned j = 0; j < 128; j++) for (unsigned i = 1; i < 10; i++) CHECK(quant_mode_table[i][j] <= quant_mode_table[i - 1][j] || i == 1, "qmt not monotone at %u %u", i, j);
	// per (partition count, block mode): the row of the mode's colour bit budget
	{
		const int8_t* ml = (const int8_t*)(B + r.off_mode_levels);
		const astcd::BlockMode* bmods = (const astcd::BlockMode*)(B + r.off_block_modes);
		const unsigned nm = r.block_mode_count_1plane_2plane_selected ? r.block_mode_count_1plane_2plane_selected : 1u;
		for (int pc = 1; pc <= 4; pc++)
			for (unsigned m = 0; m < r.block_mode_count_1plane_2plane_selected; m++)
			{
				const int free_bits[4] = { 111, 97, 94, 91 };
				const int bits = (bmods[m].is_dual_plane ? 109 : free_bits[pc - 1]) - (int)bmods[m].weight_bits;
				for (unsigned i = 0; i < 16; i++)
				{
					const int want = (bits > 0 && bits < 128 && i < 10) ? quant_mode_table[i][bits] : -1;
					CHECK(ml[((pc - 1) * nm + m) * 16 + i] == want, "mode levels pc %d mode %u pairs %u: %d %d", pc, m, i, ml[((pc - 1) * nm + m) * 16 + i], want);
				}
			}
	}
	const int8_t* qmb = (const int8_t*)(B + r.off_quant_mode_by_bits);
	for (unsigned i = 0; i < 10; i++) for (unsigned j = 0; j < 128; j++) CHECK(qmb[j * 16 + i] == quant_mode_table[i][j], "qmt by bits %u %u: %d %d", i, j, qmb[j * 16 + i], quant_mode_table[i][j]);

	const uint8_t* tr = B + r.off_integer_of_trits; const uint8_t* qu = B + r.off_integer_of_quints;
	for (unsigned a = 0; a < 3; a++) for (unsigned b = 0; b < 3; b++) for (unsigned cc = 0; cc < 3; cc++) for (unsigned d = 0; d < 3; d++) for (unsigned e = 0; e < 3; e++)
		CHECK(tr[(((a * 3 + b) * 3 + cc) * 3 + d) * 3 + e] == integer_of_trits[a][b][cc][d][e], "trits %u%u%u%u%u", a, b, cc, d, e);
	for (unsigned a = 0; a < 5; a++) for (unsigned b = 0; b < 5; b++) for (unsigned cc = 0; cc < 5; cc++)
		CHECK(qu[(a * 5 + b) * 5 + cc] == integer_of_quints[a][b][cc], "quints %u%u%u: %u %u", a, b, cc, qu[(a * 5 + b) * 5 + cc], integer_of_quints[a][b][cc]);

	const float* st = (const float*)(B + r.off_sin_table); const float* ct = (const float*)(B + r.off_cos_table);
	for (unsigned j = 0; j < 64; j++) for (unsigned i = 0; i < 32; i++)
	{
		CHECK(st[j * 32 + i] == sin_table[j][i], "sin");
		CHECK(ct[j * 32 + i] == cos_table[j][i], "cos");
		CHECK(((const float*)(B + r.off_cos_sin_table))[(j * 32 + i) * 2] == cos_table[j][i] && ((const float*)(B + r.off_cos_sin_table))[(j * 32 + i) * 2 + 1] == sin_table[j][i], "cos/sin pair");
	}

	printf("%ux%u q=%.0f: %u block modes, %u decimation modes, partitions %u/%u/%u, blob %zu bytes: %s (%d mismatches)\n",
	       bx, by, quality, r.block_mode_count_1plane_2plane_selected, r.decimation_mode_count_selected,
	       r.partitioning_count_selected[1], r.partitioning_count_selected[2], r.partitioning_count_selected[3], blob.size(), fails ? "FAIL" : "OK", fails);
	astcenc_context_free(ctx);
	return fails ? 1 : 0;
}
