// SPDX-License-Identifier: Apache-2.0
// Partition search: k-means clustering of the block, ranking of the partition table by bitmap
// mismatch, then line-fit error estimate of the best-ranked partitionings.
//   ref: kmeans_init / kmeans_assign / kmeans_update       Source/astcenc_find_best_partitioning.cpp:60-243
//        partition_mismatch{2,3,4}, count/ordering          :253-446
//        compute_kmeans_partition_ordering                  :458-501
//        insert_result, find_best_partition_candidates      :512-779
//        compute_error_squared_{rgba,rgb}                   Source/astcenc_averages_and_directions.cpp:723-946
//
// Wave mapping: lanes own texels during k-means, partition table entries during the mismatch
// count, and candidate partitionings during the line-fit scoring (each lane walks all texels of its
// candidate sequentially, which is what keeps the reference's summation order).
#pragma once
#include "wave_ctx.h"

namespace astcd { inline namespace ASTC_VARIANT {

/* Scratch of the partition search; lives in the phase-multiplexed LDS region (LdsLayout::part).
 * Fixed header followed by arrays sized for this context (see part_scratch_bytes()). */
struct PartScratch {
	uint64_t bitmaps[4];
	uint16_t mscount[64];
	int      best_count;
	int      best[MAX_PARTITIONING_CANDIDATES];
	int      n;                 // capacity of ordering / mismatch
	int      lim;               // capacity of the error arrays
	uint8_t  pad[256 - 32 - 128 - 4 - 32 - 8];
	// u16 ordering[n]; f32 uncor_err[lim]; f32 samec_err[lim]; u8 mismatch[n];
	WV_FN uint16_t* ordering() { return reinterpret_cast<uint16_t*>(reinterpret_cast<uint8_t*>(this) + 256); }
	WV_FN float* uncor_err() { return reinterpret_cast<float*>(reinterpret_cast<uint8_t*>(this) + 256 + n * 2); }
	WV_FN float* samec_err() { return uncor_err() + lim; }
	WV_FN uint8_t* mismatch() { return reinterpret_cast<uint8_t*>(samec_err() + lim); }
};
static_assert(sizeof(PartScratch) == 256, "PartScratch header");

WV_FN int mismatch2(const uint64_t* a, const uint64_t* b)
{
	int v1 = popcount64(a[0] ^ b[0]) + popcount64(a[1] ^ b[1]);
	int v2 = popcount64(a[0] ^ b[1]) + popcount64(a[1] ^ b[0]);
	return i_min(v1, v2) / 2;
}

WV_FN int mismatch3(const uint64_t* a, const uint64_t* b)
{
	int p00 = popcount64(a[0] ^ b[0]), p01 = popcount64(a[0] ^ b[1]), p02 = popcount64(a[0] ^ b[2]);
	int p10 = popcount64(a[1] ^ b[0]), p11 = popcount64(a[1] ^ b[1]), p12 = popcount64(a[1] ^ b[2]);
	int p20 = popcount64(a[2] ^ b[0]), p21 = popcount64(a[2] ^ b[1]), p22 = popcount64(a[2] ^ b[2]);
	int v0 = i_min(p11 + p22, p12 + p21) + p00;
	int v1 = i_min(p10 + p22, p12 + p20) + p01;
	int v2 = i_min(p10 + p21, p11 + p20) + p02;
	return i_min(i_min(v0, v1), v2) / 2;
}

WV_FN int mismatch4(const uint64_t* a, const uint64_t* b)
{
	int p[4][4];
	for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) p[i][j] = popcount64(a[i] ^ b[j]);
	int mx23 = i_min(p[2][2] + p[3][3], p[2][3] + p[3][2]);
	int mx13 = i_min(p[2][1] + p[3][3], p[2][3] + p[3][1]);
	int mx12 = i_min(p[2][1] + p[3][2], p[2][2] + p[3][1]);
	int mx03 = i_min(p[2][0] + p[3][3], p[2][3] + p[3][0]);
	int mx02 = i_min(p[2][0] + p[3][2], p[2][2] + p[3][0]);
	int mx01 = i_min(p[2][1] + p[3][0], p[2][0] + p[3][1]);
	int v0 = p[0][0] + i_min(i_min(p[1][1] + mx23, p[1][2] + mx13), p[1][3] + mx12);
	int v1 = p[0][1] + i_min(i_min(p[1][0] + mx23, p[1][2] + mx03), p[1][3] + mx02);
	int v2 = p[0][2] + i_min(i_min(p[1][1] + mx03, p[1][0] + mx13), p[1][3] + mx01);
	int v3 = p[0][3] + i_min(i_min(p[1][1] + mx02, p[1][2] + mx01), p[1][0] + mx12);
	return i_min(i_min(v0, v1), i_min(v2, v3)) / 2;
}

/* (ref: compute_kmeans_partition_ordering :458) -> ps.ordering[0..count) */
WV_FN int kmeans_partition_ordering(const Ctx& c, int pc, PartScratch& ps)
{
	TrialInfo& tr = c.tr();
	const BlkInfo& blk = c.blk();
	const int T = c.T;
	float* dist = c.tsc_p(0);
	float* assign = c.tsc_p(1);          // partition of texel, as float
	const f4 cw = cw4_of(blk);
	float* centers = &tr.fbox[0];      // [4][4]

	// ---- kmeans_init (ref: :60-135): sequential prefix scans, run by all lanes uniformly ----
	{
		int sample = 145897 % T;
		WV_ONE { for (int k = 0; k < 4; k++) centers[k] = c.data(k)[sample]; }
		WV_SYNC();
		int clusters_selected = 1;

		WV_FOR_T(i, T)
		{
			f4 color = mk4(c.data(0)[i], c.data(1)[i], c.data(2)[i], c.data(3)[i]);
			f4 diff = color - load4(centers);
			dist[i] = dot_s(diff * diff, cw);
		}
		WV_SYNC();

		const float cluster_cutoffs[9] = {
			0.626220f, 0.932770f, 0.275454f,
			0.318558f, 0.240113f, 0.009190f,
			0.347661f, 0.731960f, 0.156391f
		};
		int cutoff = (clusters_selected - 1) + 3 * (pc - 2);

		while (true)
		{
			// (both scans add in texel order like the reference; no branch inside: the first index at which the running
			//  sum reaches the cut-off is picked with selects)
			float distance_sum = 0.0f;
			for (int i = 0; i < T; i++) distance_sum += dist[i];

			float summa = 0.0f;
			float distance_cutoff = distance_sum * cluster_cutoffs[cutoff++];
			sample = T;
			for (int i = 0; i < T; i++)
			{
				summa += dist[i];
				sample = (sample == T && summa >= distance_cutoff) ? i : sample;
			}
			sample = i_min(sample, T - 1);

			WV_SYNC();
			WV_ONE { for (int k = 0; k < 4; k++) centers[clusters_selected * 4 + k] = c.data(k)[sample]; }
			WV_SYNC();
			clusters_selected++;
			if (clusters_selected >= pc) break;

			WV_FOR_T(i, T)
			{
				f4 color = mk4(c.data(0)[i], c.data(1)[i], c.data(2)[i], c.data(3)[i]);
				f4 diff = color - load4(&centers[(clusters_selected - 1) * 4]);
				float distance = dot_s(diff * diff, cw);
				dist[i] = f_min(distance, dist[i]);
			}
			WV_SYNC();
		}
	}

	// ---- 3 x assign, 2 x update (ref: :468-480) ----
	for (int iter = 0; iter < 3; iter++)
	{
		if (iter > 0)
		{
			// kmeans_update: per (partition, channel) sequential sums in texel order (ref: :210-243)
			WV_FOR64(k, pc * 4)
			{
				int p = k >> 2, ch = k & 3;
				const float* d = c.data(ch);
				float sum = 0.0f;
				int cnt = 0;
				for (int i = 0; i < T; i++)
				{
					// (a texel of another cluster adds +0.0, which leaves the sum -- never -0.0 -- as it is)
					const bool mine = (int)assign[i] == p;
					sum += mine ? d[i] : 0.0f;
					cnt += mine ? 1 : 0;
				}
				float scale = 1.0f / (float)cnt;
				tr.fbox[16 + k] = sum * scale;
			}
			WV_SYNC();
			WV_FOR64(k, pc * 4) { centers[k] = tr.fbox[16 + k]; }
			WV_SYNC();
		}

		// kmeans_assign (ref: :146-199)
		WV_FOR_T(i, T)
		{
			float best_distance = 3.402823466e+38f;
			int best_partition = 0;
			f4 color = mk4(c.data(0)[i], c.data(1)[i], c.data(2)[i], c.data(3)[i]);
			for (int j = 0; j < pc; j++)
			{
				f4 diff = color - load4(&centers[j * 4]);
				float distance = dot_s(diff * diff, cw);
				if (distance < best_distance)
				{
					best_distance = distance;
					best_partition = j;
				}
			}
			assign[i] = (float)best_partition;
		}
		WV_SYNC();

		// empty-cluster repair (ref: :184-198).  It only does anything when a cluster came out empty: the cluster sizes are
		// counted across the wave first (integer counts: any order), and the strictly sequential repair runs in the rare
		// case that one of them is zero.
		{
			int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
#if WV_DEVICE
			for (int base = 0; base < T; base += 64)
			{
				const int i = base + WV_LANE;
				const int a = i < T ? (int)assign[i] : -1;
				c0 += __popcll(__ballot(a == 0)); c1 += __popcll(__ballot(a == 1));
				c2 += __popcll(__ballot(a == 2)); c3 += __popcll(__ballot(a == 3));
			}
#else
			for (int i = 0; i < T; i++)
			{
				int a = (int)assign[i];
				c0 += a == 0; c1 += a == 1; c2 += a == 2; c3 += a == 3;
			}
#endif
			bool any_empty = false;
			for (int i = 0; i < pc; i++) any_empty = any_empty || (i == 0 ? c0 : i == 1 ? c1 : i == 2 ? c2 : c3) == 0;
			if (any_empty)
			{
				WV_ONE
				{
					bool problem_case;
					do
					{
						problem_case = false;
						for (int i = 0; i < pc; i++)
						{
							int ci = i == 0 ? c0 : i == 1 ? c1 : i == 2 ? c2 : c3;
							if (ci == 0)
							{
								int a = (int)assign[i];
								c0 -= a == 0; c1 -= a == 1; c2 -= a == 2; c3 -= a == 3;
								c0 += i == 0; c1 += i == 1; c2 += i == 2; c3 += i == 3;
								assign[i] = (float)i;
								problem_case = true;
							}
						}
					} while (problem_case);
				}
				WV_SYNC();
			}
		}
	}

	// ---- bitmaps over the k-means texel subset (ref: :483-490) ----
	const int texels_to_process = i_min(T, MAX_KMEANS_TEXELS);
	const uint8_t* km = c.table(c.root->off_kmeans_texels);
#if WV_DEVICE
	{
		// lane i looks at k-means texel i (at most 64 of them): a partition's bitmap is one ballot
		const int a = WV_LANE < texels_to_process ? (int)assign[km[WV_LANE]] : -1;
		for (int p = 0; p < pc; p++)
		{
			const uint64_t bm = __ballot(a == p);
			WV_ONE { ps.bitmaps[p] = bm; }
		}
	}
#else
	WV_FOR64(p, pc)
	{
		uint64_t bm = 0;
		for (int i = 0; i < texels_to_process; i++)
		{
			if ((int)assign[km[i]] == p) bm |= 1ULL << i;
		}
		ps.bitmaps[p] = bm;
	}
#endif
	WV_SYNC();

	// ---- mismatch counts against every selected partitioning (ref: :365-401) ----
	const int count = (int)c.root->partitioning_count_selected[pc - 1];
	const uint64_t* cov = reinterpret_cast<const uint64_t*>(c.table(c.root->off_coverage[pc - 1]));
	WV_FOR(i, count)
	{
		int m;
		if (pc == 2) m = mismatch2(ps.bitmaps, cov + i * 2);
		else if (pc == 3) m = mismatch3(ps.bitmaps, cov + i * 3);
		else m = mismatch4(ps.bitmaps, cov + i * 4);
		ps.mismatch()[i] = (uint8_t)m;
	}
	WV_SYNC();

	// ---- stable counting sort (ref: :412-446) ----
#if WV_DEVICE
	{
		// Elements are taken 64 at a time, one per lane.  A lane's rank among the equal keys before it in its chunk -- what
		// makes the sort stable -- comes from the mask of the lanes that hold the same key: six ballots, one per key bit
		// (a mismatch count is below 64), each narrowing the mask to the lanes that agree in that bit; the lowest lane of a
		// mask speaks for its key when the bins (64 counters in the trial's integer mailbox) are updated.  (Peeling the
		// distinct keys of a chunk off one by one, as before, costs a dozen instructions per distinct key and pass.)
		const int lane = WV_LANE;
		const uint8_t* mm = ps.mismatch();
		uint16_t* ord = ps.ordering();
		int* bins = &tr.ibox[0];
		bins[lane] = 0;
		WV_SYNC();
		auto same_key_lanes = [](int m, bool valid) -> unsigned long long
		{
			unsigned long long mask = __ballot(valid);
			#pragma unroll
			for (int k = 0; k < 6; k++)
			{
				const bool bit = ((m >> k) & 1) != 0;
				const unsigned long long with_bit = __ballot(bit);
				mask &= bit ? with_bit : ~with_bit;
			}
			return mask;
		};
		auto lanes_below = [](unsigned long long mask) -> int      // popcount(mask & lanes below this one)
		{
			return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
		};
		for (int first = 0; first < count; first += 64)
		{
			const int i = first + lane;
			const bool valid = i < count;
			const int m = valid ? (int)mm[i] : 0;
			const unsigned long long same = same_key_lanes(m, valid);
			if (valid && lanes_below(same) == 0) bins[m] += (int)__popcll(same);      // (one lane per distinct key: no two lanes write one bin)
			WV_SYNC();      // (the next chunk's lanes read the bins this chunk's rank-0 lanes wrote; a fence, no instruction)
		}
		WV_SYNC();
		// exclusive prefix over the bins (integer adds: any order is exact)
		{
			const int hist = bins[lane];
			int base = hist;
			for (int d = 1; d < 64; d <<= 1)
			{
				int up = __shfl_up(base, d);
				if (lane >= d) base += up;
			}
			bins[lane] = base - hist;
		}
		WV_SYNC();
		for (int first = 0; first < count; first += 64)
		{
			const int i = first + lane;
			const bool valid = i < count;
			const int m = valid ? (int)mm[i] : 0;
			const unsigned long long same = same_key_lanes(m, valid);
			const int rank = lanes_below(same);
			const int start = bins[m];
			if (valid) ord[start + rank] = (uint16_t)i;
			// (the wave's LDS operations complete in order: every lane has its `start` before the bin moves on)
			if (valid && rank == 0) bins[m] = start + (int)__popcll(same);
			WV_SYNC();
		}
		(void)texels_to_process;
	}
#else
	WV_ONE
	{
		for (int i = 0; i < 64; i++) ps.mscount[i] = 0;
		for (int i = 0; i < count; i++) ps.mscount[ps.mismatch()[i]]++;
		uint16_t sum = 0;
		for (int i = 0; i < texels_to_process; i++)
		{
			uint16_t cnt = ps.mscount[i];
			ps.mscount[i] = sum;
			sum = (uint16_t)(sum + cnt);
		}
		for (int i = 0; i < count; i++)
		{
			unsigned int idx = ps.mscount[ps.mismatch()[i]]++;
			ps.ordering()[idx] = (uint16_t)i;
		}
	}
#endif
	WV_SYNC();
	return count;
}

/* Line-fit error of one candidate partitioning, all work on one lane; `pv` points at the candidate's
 * record staged in LDS (a lane per candidate reading its own record from global memory would touch
 * a different cache line per lane on every access).
 * (ref: compute_avgs_and_dirs_{4_comp,3_comp_rgb} + compute_error_squared_{rgba,rgb} + :660-670 / :726-738) */
template <bool USES_ALPHA>
WV_FN void score_partitioning(const Ctx& c, int pc, const PartView& pv, float weight_imprecision_estim, float& uncor_out, float& samec_out)
{
	const BlkInfo& blk = c.blk();
	const int T = c.T, Tp = c.Tp;
	// (three or four channels: a template parameter -- as a run-time value it was a wave-uniform branch around every texel of
	//  both texel loops)
	constexpr int n = USES_ALPHA ? 4 : 3;

	// partition averages: 4-accumulator masked sums in texel order (ref: averages_and_directions.cpp:47-385), one partition
	// at a time -- sixteen accumulators live instead of forty-eight (round 5: the function fits a stage function's
	// caller-saved registers); lanes past T read the zero padding of the texel rows, which adds +0.0 like the reference's
	// masked tail lanes.  The last partition's average comes from what the others leave of the block's sums.
	float avg[4][4];
	{
		float rest[4];
		#pragma unroll
		for (int ch = 0; ch < 4; ch++) rest[ch] = blk.data_mean[ch] * (float)T;
		const uint32_t* ot4 = reinterpret_cast<const uint32_t*>(pv.of_texel);
		#pragma unroll
		for (int p = 0; p < 3; p++)
		{
			if (p >= pc - 1) break;
			float acc[4][4];
			#pragma unroll
			for (int ch = 0; ch < 4; ch++)
				#pragma unroll
				for (int l = 0; l < 4; l++) acc[ch][l] = 0.0f;
			#pragma nounroll
			for (int i = 0; i < Tp; i += 4)
			{
				const uint32_t ot = ot4[i >> 2];
				// (`acc + (in partition p ? d : 0)` as acc + d * m with m = 1.0 / 0.0: d * 1 = d, d * 0 = +0 for the
				//  non-negative texel data -- the same sum, in multiplies and adds (the fast issue class) instead of selects)
				const float m0 = (int)(ot & 0xFF) == p ? 1.0f : 0.0f;
				const float m1 = (int)((ot >> 8) & 0xFF) == p ? 1.0f : 0.0f;
				const float m2 = (int)((ot >> 16) & 0xFF) == p ? 1.0f : 0.0f;
				const float m3 = (int)(ot >> 24) == p ? 1.0f : 0.0f;
				#pragma unroll
				for (int ch = 0; ch < 4; ch++)
				{
					if (ch >= n) break;
					const float* d = c.data(ch) + i;
					acc[ch][0] = f_add_masked(acc[ch][0], d[0], m0);
					acc[ch][1] = f_add_masked(acc[ch][1], d[1], m1);
					acc[ch][2] = f_add_masked(acc[ch][2], d[2], m2);
					acc[ch][3] = f_add_masked(acc[ch][3], d[3], m3);
				}
			}
			#pragma unroll
			for (int ch = 0; ch < 4; ch++)
			{
				if (ch >= n) break;
				float total = hadd4(acc[ch][0], acc[ch][1], acc[ch][2], acc[ch][3]);
				rest[ch] = rest[ch] - total;
				avg[p][ch] = total / (float)pv.cnt(p);
			}
			if (n == 3) avg[p][3] = 0.0f;
		}
		#pragma unroll
		for (int p = 1; p < 4; p++)
		{
			if (p != pc - 1) continue;
			#pragma unroll
			for (int ch = 0; ch < 4; ch++) avg[p][ch] = ch < n ? rest[ch] / (float)pv.cnt(p) : 0.0f;
		}
	}

	float ua0 = 0.0f, ua1 = 0.0f, ua2 = 0.0f, ua3 = 0.0f;
	float sa0 = 0.0f, sa1 = 0.0f, sa2 = 0.0f, sa3 = 0.0f;
	// what each partition adds after the texel sums (ref: :660-670): formed while the partition's directions are in
	// registers, added in partition order below
	float tail_uncor[4], tail_samec[4];

	#pragma unroll
	for (int p = 0; p < 4; p++)
	{
		if (p >= pc) break;
		const uint8_t* tix = pv.sorted + pv.off(p);
		const int cnt = pv.cnt(p);
		f4 average = load4(avg[p]);

		// dominant direction (ref: :409-454)
		f4 sum[4];
		#pragma unroll
		for (int k = 0; k < 4; k++) sum[k] = splat4(0.0f);
		for (int i = 0; i < cnt; i++)
		{
			int t = tix[i];
			f4 d = mk4(c.data(0)[t], c.data(1)[t], c.data(2)[t], n == 4 ? c.data(3)[t] : 0.0f) - average;
			// (`above ? sum + d : sum` as sum + d * m, m = 1.0 / 0.0: d is finite -- every searched partitioning has texels in
			//  all its partitions, so the average is a number -- and a sum is never -0, so adding d * 0 = +-0 changes nothing)
			auto add_masked4 = [](f4 acc, f4 v, float m) { return mk4(f_add_masked(acc.x, v.x, m), f_add_masked(acc.y, v.y, m), f_add_masked(acc.z, v.z, m), f_add_masked(acc.w, v.w, m)); };
			sum[0] = add_masked4(sum[0], d, d.x > 0.0f ? 1.0f : 0.0f);
			sum[1] = add_masked4(sum[1], d, d.y > 0.0f ? 1.0f : 0.0f);
			sum[2] = add_masked4(sum[2], d, d.z > 0.0f ? 1.0f : 0.0f);
			if (n == 4) sum[3] = add_masked4(sum[3], d, d.w > 0.0f ? 1.0f : 0.0f);
		}
		f4 best_vector = sum[0];
		float best_sum = dot_s(sum[0], sum[0]);
		#pragma unroll
		for (int k = 1; k < 4; k++)
		{
			if (k >= n) break;
			float prod = dot_s(sum[k], sum[k]);
			const bool better = prod > best_sum;
			best_vector = select4(better, sum[k], best_vector);
			best_sum = better ? prod : best_sum;
		}

		f4 ub = normalize_safe4(best_vector, n == 4 ? unit4() : unit3());
		f4 sb = normalize_safe4(average, n == 4 ? unit4() : unit3());
		float dd = n == 4 ? dot_s(average, ub) : dot3_s(average, ub);
		f4 amod = average - ub * (n == 4 ? splat4(dd) : mk4(dd, dd, dd, 0.0f));

		// squared distance to both lines; accumulators run on across partitions, lane = position
		// within the partition mod 4 (ref: :778-831, :892-937)
		float lo = 1e10f, hi = -1e10f;
		auto one_texel = [&](int i, float& ua, float& sa)
		{
			int t = tix[i];
			float r = c.data(0)[t], g = c.data(1)[t], b = c.data(2)[t];
			float ue, se, uncor_param;
			if (n == 4)
			{
				float a = c.data(3)[t];
				uncor_param = (r * ub.x) + (g * ub.y) + (b * ub.z) + (a * ub.w);
				float d0 = (amod.x - r) + (uncor_param * ub.x);
				float d1 = (amod.y - g) + (uncor_param * ub.y);
				float d2 = (amod.z - b) + (uncor_param * ub.z);
				float d3 = (amod.w - a) + (uncor_param * ub.w);
				ue = (cw_of(blk, 0) * d0 * d0) + (cw_of(blk, 1) * d1 * d1) + (cw_of(blk, 2) * d2 * d2) + (cw_of(blk, 3) * d3 * d3);
				float sp = (r * sb.x) + (g * sb.y) + (b * sb.z) + (a * sb.w);
				float s0 = sp * sb.x - r, s1 = sp * sb.y - g, s2 = sp * sb.z - b, s3 = sp * sb.w - a;
				se = (cw_of(blk, 0) * s0 * s0) + (cw_of(blk, 1) * s1 * s1) + (cw_of(blk, 2) * s2 * s2) + (cw_of(blk, 3) * s3 * s3);
			}
			else
			{
				uncor_param = (r * ub.x) + (g * ub.y) + (b * ub.z);
				float d0 = (amod.x - r) + (uncor_param * ub.x);
				float d1 = (amod.y - g) + (uncor_param * ub.y);
				float d2 = (amod.z - b) + (uncor_param * ub.z);
				ue = (cw_of(blk, 0) * d0 * d0) + (cw_of(blk, 1) * d1 * d1) + (cw_of(blk, 2) * d2 * d2);
				float sp = (r * sb.x) + (g * sb.y) + (b * sb.z);
				float s0 = sp * sb.x - r, s1 = sp * sb.y - g, s2 = sp * sb.z - b;
				se = (cw_of(blk, 0) * s0 * s0) + (cw_of(blk, 1) * s1 * s1) + (cw_of(blk, 2) * s2 * s2);
			}
			// (the hardware minimum / maximum: they differ from the reference's compare-selects in the sign of a zero only
			//  -- the data are numbers -- and hi - lo goes through max(., 1e-7) below, which is blind to that)
			lo = f_run_min(uncor_param, lo);
			hi = f_run_max(uncor_param, hi);
			ua += ue; sa += se;
		};
		// accumulator lane = position mod 4: four texels per trip, each into its own pair of accumulators (one loop over i
		// with the accumulator picked by i & 3 compiles to a chain of scalar branches and register moves per texel)
		for (int i = 0; i < cnt; i += 4)
		{
			one_texel(i, ua0, sa0);
			if (i + 1 < cnt) one_texel(i + 1, ua1, sa1);
			if (i + 2 < cnt) one_texel(i + 2, ua2, sa2);
			if (i + 3 < cnt) one_texel(i + 3, ua3, sa3);
		}
		float linelen = hi - lo;
		linelen = f_max(linelen, 1e-7f);
		{
			const f4 error_weights = splat4((float)cnt * weight_imprecision_estim);
			const f4 uncor_vector = ub * linelen;
			const f4 samec_vector = sb * linelen;
			if (n == 4)
			{
				tail_uncor[p] = dot_s(uncor_vector * uncor_vector, error_weights);
				tail_samec[p] = dot_s(samec_vector * samec_vector, error_weights);
			}
			else
			{
				tail_uncor[p] = dot3_s(uncor_vector * uncor_vector, error_weights);
				tail_samec[p] = dot3_s(samec_vector * samec_vector, error_weights);
			}
		}
	}

	float uncor_error = hadd4(ua0, ua1, ua2, ua3);
	float samec_error = hadd4(sa0, sa1, sa2, sa3);
	#pragma unroll
	for (int p = 0; p < 4; p++)
	{
		if (p >= pc) break;
		uncor_error += tail_uncor[p];
		samec_error += tail_samec[p];
	}
	uncor_out = uncor_error;
	samec_out = samec_error;
}

/* (ref: insert_result :512) */
WV_FN void insert_result(int max_values, float this_error, int this_partition, float* best_errors, int* best_partitions)
{
	if (this_error >= best_errors[max_values - 1]) return;
	for (int i = 0; i < max_values; i++)
	{
		if (this_error > best_errors[i]) continue;
		for (int j = max_values - 1; j > i; j--)
		{
			best_errors[j] = best_errors[j - 1];
			best_partitions[j] = best_partitions[j - 1];
		}
		best_errors[i] = this_error;
		best_partitions[i] = this_partition;
		break;
	}
}

/* (ref: find_best_partition_candidates :551).  Returns the number of PACKED partition indices
 * written to ps.best[]. */
/* Partition search in three steps (ref: find_best_partition_candidates :551-780); each is its own
 * out-of-line stage in the kernel so that none of them needs callee-saved registers. */

/* Step 1: k-means clustering of the texels, ordering of the partitionings by how well they match it.
 * Returns the length of the ordered sequence. */
WV_FN int partition_search_order(const Ctx& c, int pc)
{
	PartScratch& ps = *reinterpret_cast<PartScratch*>(c.part());
	WV_ONE
	{
		uint32_t lim = c.cfg->tune_partition_index_limit[0];
		if (c.cfg->tune_partition_index_limit[1] > lim) lim = c.cfg->tune_partition_index_limit[1];
		if (c.cfg->tune_partition_index_limit[2] > lim) lim = c.cfg->tune_partition_index_limit[2];
		uint32_t n = (c.root->max_partitionings + 3u) & ~3u;
		ps.n = (int)n;
		ps.lim = (int)(lim < n ? lim : n);
	}
	WV_SYNC();
	return kmeans_partition_ordering(c, pc, ps);
}

/* Step 2: line-fit errors of the first partition_search_limit partitionings of the ordering. */
WV_FN void partition_search_score(const Ctx& c, int pc, int partition_search_limit)
{
	WV_LANE_SCOPE;
	PartScratch& ps = *reinterpret_cast<PartScratch*>(c.part());
	const BlkInfo& blk = c.blk();
	const int T = c.T;
	float weight_imprecision_estim = 0.055f;
	if (T <= 20) weight_imprecision_estim = 0.03f;
	else if (T <= 31) weight_imprecision_estim = 0.04f;
	else if (T <= 41) weight_imprecision_estim = 0.05f;
	weight_imprecision_estim = wv_uniform(weight_imprecision_estim * weight_imprecision_estim);

	const bool uses_alpha = wv_uniform(!(blk.data_min[3] == blk.data_max[3]));

	{ PROF_SCOPE(c, PS_PSCORE);
	const int rec_words = wv_uniform((int)c.L->part_rec_words);        // (sizeof(PartitionHeader) + 2 T + 3) / 4
	const uint32_t rec_inv24 = wv_uniform(c.L->part_rec_inv24);
	const int chunk = wv_uniform((int)c.L->part_chunk);
	uint32_t* staged = reinterpret_cast<uint32_t*>(c.lds + c.L->part_tabs);
	const uint8_t* part_base = c.table(c.root->off_partitions[pc - 1]);
	const uint32_t part_stride = wv_uniform(c.root->partition_stride);
	for (int first = 0; first < partition_search_limit; first += chunk)
	{
		const int nn = i_min(chunk, partition_search_limit - first);
		// stage the records of candidates [first, first + nn) (coalesced word copies)
		WV_FOR(k, nn * rec_words)
		{
			int sl = (int)(((uint32_t)k * rec_inv24) >> 24), w = k - sl * rec_words;
			// (word w of the record of partitioning ordering[first + sl]: scalar table base + a 32-bit byte offset)
			staged[wv_opaque(k)] = table_at_byte<uint32_t>(part_base, (uint32_t)ps.ordering()[first + sl] * part_stride + (uint32_t)w * 4u);
		}
		WV_SYNC();
		WV_FOR64(i_lane, nn)
		{
			// (opaque: the lane's record address and everything derived from it would otherwise be loop-invariant all the way
			//  out to the search loop of compress_block, hoisted there and carried -- spilled -- across every trial)
			const int i = wv_opaque(i_lane);
			const uint8_t* rec = reinterpret_cast<const uint8_t*>(staged + i * rec_words);
			PartView pv;
			pv.h = reinterpret_cast<const PartitionHeader*>(rec);
			pv.of_texel = rec + sizeof(PartitionHeader);
			pv.sorted = pv.of_texel + T;
			pv.pcount = pc;
			{
				uint32_t o = 0;
				pv.offsets = 0; pv.counts = 0;
				for (int q = 0; q < 4; q++)
				{
					uint32_t n = pv.h->texel_count[q];
					pv.offsets |= o << (8 * q); pv.counts |= n << (8 * q);
					o += n;
				}
			}
			float ue, se;
			if (uses_alpha) score_partitioning<true>(c, pc, pv, weight_imprecision_estim, ue, se);
			else score_partitioning<false>(c, pc, pv, weight_imprecision_estim, ue, se);
			ps.uncor_err()[first + i] = ue;
			ps.samec_err()[first + i] = se;
		}
		WV_SYNC();
	}
	}

}

/* Step 3: the best requested_candidates partitionings by uncorrelated and by same-chroma error,
 * interleaved and deduplicated -> ps.best[]; returns how many. */
WV_FN int partition_search_select(const Ctx& c, int partition_search_limit, int requested_candidates)
{
	PartScratch& ps = *reinterpret_cast<PartScratch*>(c.part());
#if WV_DEVICE
	// The reference inserts the candidates one by one into two sorted top-N lists (ref: :589-600, :672-673), then
	// interleaves the lists and drops repeats (ref: :745-776).  Without equal errors the lists are simply the N smallest
	// in ascending order, and a candidate's place is the number of candidates with a smaller error: one lane per
	// candidate counts that over registers (no LDS inside the loops).  With equal errors, or an error at or above the
	// lists' initial value, the insertion order decides and the sequential replay below runs instead.
	if (partition_search_limit <= 64)
	{
		const int lane = WV_LANE;
		const bool live = lane < partition_search_limit;
		const float ue = live ? ps.uncor_err()[lane] : 0.0f;
		const float se = live ? ps.samec_err()[lane] : 0.0f;
		const int part = live ? (int)ps.ordering()[lane] : -1;
		int urank = 0, srank = 0;
		bool plain = !live || (ue < ERROR_CALC_DEFAULT && se < ERROR_CALC_DEFAULT);       // (false for NaN as well)
		for (int j = 0; j < partition_search_limit; j++)
		{
			const float uj = int_as_float(__builtin_amdgcn_readlane(float_as_int(ue), j));
			const float sj = int_as_float(__builtin_amdgcn_readlane(float_as_int(se), j));
			urank += uj < ue ? 1 : 0;
			srank += sj < se ? 1 : 0;
			plain = plain && (j == lane || (uj != ue && sj != se));
		}
		if (__ballot(live && !plain) == 0ull)
		{
			// interleaved order: uncorrelated rank r at slot 2r, same-chroma rank r at slot 2r + 1; lane k picks up slot k
			int* slot_part = reinterpret_cast<int*>(ps.mscount);
			if (live && urank < requested_candidates) slot_part[2 * urank] = part;
			if (live && srank < requested_candidates) slot_part[2 * srank + 1] = part;
			WV_SYNC();
			const int nslots = 2 * requested_candidates;                              // <= 16
			const int mine = lane < nslots ? slot_part[lane] : -1;
			bool repeat = false;
			for (int j = 0; j < nslots; j++)
			{
				const int pj = __builtin_amdgcn_readlane(mine, j);
				repeat = repeat || (j < lane && pj == mine);
			}
			const unsigned long long fresh = __ballot(lane < nslots && !repeat);
			const int at = __popcll(fresh & ((1ull << lane) - 1ull));
			if (lane < nslots && !repeat && at < requested_candidates) ps.best[at] = mine;
			const int emitted = i_min(__popcll(fresh), requested_candidates);
			WV_ONE { ps.best_count = emitted; }
			WV_SYNC();
			return emitted;
		}
	}
#endif
	// sorted insertion is order dependent on ties: replay it sequentially (ref: :589-600, :672-673)
	WV_ONE
	{
		// the four top-N lists live in the (now idle) counting-sort bins rather than in private arrays
		static_assert(sizeof(ps.mscount) >= 4 * MAX_PARTITIONING_CANDIDATES * 4, "top-N lists do not fit the sort bins");
		float* uncor_best_errors = reinterpret_cast<float*>(ps.mscount);
		float* samec_best_errors = uncor_best_errors + MAX_PARTITIONING_CANDIDATES;
		int* uncor_best_partitions = reinterpret_cast<int*>(samec_best_errors + MAX_PARTITIONING_CANDIDATES);
		int* samec_best_partitions = uncor_best_partitions + MAX_PARTITIONING_CANDIDATES;
		for (int i = 0; i < MAX_PARTITIONING_CANDIDATES; i++)
		{
			uncor_best_errors[i] = ERROR_CALC_DEFAULT; samec_best_errors[i] = ERROR_CALC_DEFAULT;
			uncor_best_partitions[i] = 0; samec_best_partitions[i] = 0;
		}
		for (int i = 0; i < partition_search_limit; i++)
		{
			int partition = ps.ordering()[i];
			insert_result(requested_candidates, ps.uncor_err()[i], partition, uncor_best_errors, uncor_best_partitions);
			insert_result(requested_candidates, ps.samec_err()[i], partition, samec_best_errors, samec_best_partitions);
		}

		// interleave + dedupe (ref: :745-776); packed indices are unique per seed, so dedupe on them
		int emitted = 0;
		for (int i = 0; i < requested_candidates * 2 && emitted < requested_candidates; i++)
		{
			int partition = (i & 1) ? samec_best_partitions[i >> 1] : uncor_best_partitions[i >> 1];
			bool written = false;
			for (int j = 0; j < emitted; j++) written = written || ps.best[j] == partition;
			if (!written) ps.best[emitted++] = partition;
		}
		ps.best_count = emitted;
	}
	WV_SYNC();
	return ps.best_count;
}

} } // namespace astcd::ASTC_VARIANT
