// wave_tile_ab.hip -- the ONE A/B VERDICT r2 asked for: lane-per-pair (the product's sw_score_pk_kernel<27>, two pairs per lane in
// 16-bit halves) against a wave-per-tile anti-diagonal mapping of the same banded local score (150 bp reads, corridor 27).
//
// Wave-per-tile here: a 16-lane DPP row owns one pair, four pairs per wave.  In band coordinates (row i, band column d, window
// position j = i + d) the three predecessors of a cell are (i-1, d) diagonal, (i-1, d+1) up and (i, d-1) left, so the cells that
// can be computed together lie on 2i + d = t: every second band column.  Lane k of a row computes d = 2k on even steps and
// d = 2k + 1 on odd steps (row i = floor(t / 2) - k): 14 of 16 lanes busy for a 27-column band, 2 * rows + 27 steps per pair,
// neighbours by row_shr:1 / row_shl:1, the read row's 8-byte score table travels down the lanes, the window class up the lanes.
//
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o wave_tile_ab profiles/tools/wave_tile_ab.hip     run: ./wave_tile_ab [pairs]
#define NGM_ENGINE_KERNELS
#include "../../nextgenmap_amd/csrc/sw_device.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

namespace {

template <int C>
__global__ __launch_bounds__(256) void sw_score_wave_kernel(const uint8_t *__restrict__ ref, const uint8_t *__restrict__ qry, float *__restrict__ scores,
		int n, int q, int rl, ngm::SwConst K, int mismatch, int gap_read, int gap_ref) {
	extern __shared__ uint8_t lds[];
	__shared__ uint2 s_tab[8];
	if (threadIdx.x < 8) s_tab[threadIdx.x] = ngm::make_row_table(threadIdx.x, K);
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane >> 4, k = lane & 15;
	uint8_t *sq = lds + (size_t) (wave * 4 + sub) * (q + rl), *sr = sq + q;
	const int groups = (n + 15) / 16;  // 16 pairs per workgroup and pass
	for (int g = blockIdx.x; g < groups; g += gridDim.x) {
		const int pair = (g * 4 + wave) * 4 + sub;
		__syncthreads();
		if (pair < n) {
			for (int o = k; o < q; o += 16) sq[o] = (uint8_t) ngm::sym_class(qry[(size_t) pair * q + o]);
			for (int o = k; o < rl; o += 16) sr[o] = (uint8_t) ngm::sym_class(ref[(size_t) pair * rl + o]);
		}
		__syncthreads();
		int he = 0, ho = 0, best = 0;
		uint2 T = s_tab[6];
		uint32_t fc = (k < rl) ? sr[k] : 6u;  // window class at j = m + k (even step)
		const bool colE = 2 * k < C, colO = 2 * k + 1 < C;
		const int steps = q + (C + 1) / 2;
		for (int m = 0; m < steps; ++m) {
			const uint2 Tn = s_tab[(m < q) ? sq[m] : 6];
			T.x = (uint32_t) __builtin_amdgcn_update_dpp((int) Tn.x, (int) T.x, 0x111, 0xF, 0xF, false);  // row_shr:1, lane 0 keeps the new row's table
			T.y = (uint32_t) __builtin_amdgcn_update_dpp((int) Tn.y, (int) T.y, 0x111, 0xF, 0xF, false);
			const int i = m - k;
			const bool row_ok = i >= 0 && i < q;
			// even step: d = 2k
			{
				const int tb = (int) __builtin_amdgcn_perm(T.y, T.x, fc | 0x0C0C0C00u);
				const int left = __builtin_amdgcn_update_dpp(0, ho, 0x111, 0xF, 0xF, false);
				const int h = max(max(max(left + gap_ref, ho + gap_read), he + tb + mismatch), 0);
				he = (row_ok && colE) ? h : 0;
				best = max(best, he);
			}
			// the window class moves one lane down: j = m + k + 1
			{
				const int jn = m + 16;
				const uint32_t fnew = (jn < rl) ? sr[jn] : 6u;
				fc = (uint32_t) __builtin_amdgcn_update_dpp((int) fnew, (int) fc, 0x101, 0xF, 0xF, false);  // row_shl:1, lane 15 takes the new class
			}
			// odd step: d = 2k + 1
			{
				const int tb = (int) __builtin_amdgcn_perm(T.y, T.x, fc | 0x0C0C0C00u);
				const int up = __builtin_amdgcn_update_dpp(0, he, 0x101, 0xF, 0xF, false);
				const int h = max(max(max(he + gap_ref, up + gap_read), ho + tb + mismatch), 0);
				ho = (row_ok && colO) ? h : 0;
				best = max(best, ho);
			}
		}
#pragma unroll
		for (int sft = 8; sft >= 1; sft >>= 1) best = max(best, __shfl_xor(best, sft, 16));
		if (k == 0 && pair < n) scores[pair] = (float) best;
	}
}

}  // namespace

int main(int argc, char **argv) {
	const int n = argc > 1 ? atoi(argv[1]) : 606208, q = 152, c = 27, rl = q + c, read_len = 150;
	std::mt19937 rng(12345);
	std::vector<uint8_t> ref((size_t) n * rl), qry((size_t) n * q, 0);
	const char acgt[4] = {'A', 'C', 'G', 'T'};
	for (int p = 0; p < n; ++p) {
		uint8_t *w = &ref[(size_t) p * rl];
		for (int j = 0; j < rl; ++j) w[j] = (uint8_t) acgt[rng() & 3];
		uint8_t *r = &qry[(size_t) p * q];
		int j = c / 2 + (int) (rng() % 5) - 2;
		for (int i = 0; i < read_len; ++i, ++j) {
			const unsigned u = rng() % 1000;
			if (u < 3 && j + 1 < rl) ++j;                    // deletion
			else if (u < 6 && j > 0) --j;                   // insertion
			r[i] = (u >= 6 && u < 36) ? (uint8_t) acgt[rng() & 3] : (rng() % 400 == 0 ? 'N' : w[std::min(j, rl - 1)]);
		}
	}
	uint8_t *d_ref, *d_qry;
	CK(hipMalloc(&d_ref, ref.size())); CK(hipMalloc(&d_qry, qry.size()));
	CK(hipMemcpy(d_ref, ref.data(), ref.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(d_qry, qry.data(), qry.size(), hipMemcpyHostToDevice));
	const int RW = ngm::read_words(q), FW = ngm::ref_words(q, c), nb = (n + 63) / 64;
	uint32_t *d_packed; uint16_t *d_lens, *d_rows; float *d_sa, *d_sb;
	CK(hipMalloc(&d_packed, (size_t) nb * (RW + FW) * 64 * 4)); CK(hipMalloc(&d_lens, (size_t) nb * 64 * 2)); CK(hipMalloc(&d_rows, (size_t) nb * 2));
	CK(hipMalloc(&d_sa, (size_t) n * 4)); CK(hipMalloc(&d_sb, (size_t) n * 4));
	const int match = 10, mismatch = -15, gap_read = -20, gap_ref = -20;
	ngm::SwConst K{};
	K.tM = match - mismatch; K.tZ = -mismatch; K.gl = gap_ref; K.gu = gap_read - mismatch; K.gap_read = gap_read; K.variant = 0; K.alt = 0;
	hipEvent_t e0, e1, e2;
	CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
	const int reps = 20;
	float ms_pack = 0, ms_pk = 0, ms_wave = 0;
	const size_t lds_wave = (size_t) 16 * (q + rl);
	int dev_cus = 256;
	{ hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0)); dev_cus = pr.multiProcessorCount; }
	for (int r = -2; r < reps; ++r) {
		CK(hipEventRecord(e0, 0));
		hipLaunchKernelGGL(ngm::pack_pairs_kernel, dim3(nb), dim3(256), (size_t) 64 * (rl + q), 0, d_ref, d_qry, n, q, rl, RW, FW, d_packed, d_lens, d_rows, 0, (const uint8_t *) nullptr);
		CK(hipEventRecord(e1, 0));
		hipLaunchKernelGGL((ngm::sw_score_pk_kernel<27, false>), dim3((nb + 7) / 8), dim3(256), 0, 0, (const uint32_t *) d_packed, (const uint16_t *) d_lens, (const uint16_t *) d_rows, d_sa, n, nb, RW, K);
		CK(hipEventRecord(e2, 0));
		CK(hipEventSynchronize(e2));
		float a, b;
		CK(hipEventElapsedTime(&a, e0, e1)); CK(hipEventElapsedTime(&b, e1, e2));
		if (r >= 0) { ms_pack += a; ms_pk += b; }
		CK(hipEventRecord(e0, 0));
		hipLaunchKernelGGL((sw_score_wave_kernel<27>), dim3(dev_cus * 8), dim3(256), lds_wave, 0, (const uint8_t *) d_ref, (const uint8_t *) d_qry, d_sb, n, q, rl, K, mismatch, gap_read, gap_ref);
		CK(hipEventRecord(e1, 0));
		CK(hipEventSynchronize(e1));
		CK(hipEventElapsedTime(&a, e0, e1));
		if (r >= 0) ms_wave += a;
	}
	CK(hipGetLastError());
	std::vector<float> sa(n), sb(n);
	CK(hipMemcpy(sa.data(), d_sa, (size_t) n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(sb.data(), d_sb, (size_t) n * 4, hipMemcpyDeviceToHost));
	size_t diff = 0; double sum = 0;
	for (int p = 0; p < n; ++p) { if (sa[p] != sb[p]) { if (diff < 5) fprintf(stderr, "pair %d: lane-per-pair %.0f wave-per-tile %.0f\n", p, sa[p], sb[p]); ++diff; } sum += sa[p]; }
	const double cells = (double) n * read_len * c;
	printf("pairs %d (150 bp, corridor 27), mean score %.1f, scores differing between the two mappings: %zu\n", n, sum / n, diff);
	printf("lane-per-pair, packed 16-bit (sw_score_pk_kernel<27>): %.3f ms  -> %.2f Tcells/s   (+ pack_pairs_kernel %.3f ms: %.2f Tcells/s together)\n",
			ms_pk / reps, cells / (ms_pk / reps * 1e-3) / 1e12, ms_pack / reps, cells / ((ms_pk + ms_pack) / reps * 1e-3) / 1e12);
	printf("wave-per-tile anti-diagonal (16-lane row per pair, staging included): %.3f ms  -> %.2f Tcells/s   (%.1fx the lane-per-pair path incl. packing)\n",
			ms_wave / reps, cells / (ms_wave / reps * 1e-3) / 1e12, ms_wave / (ms_pk + ms_pack));
	return diff ? 2 : 0;
}
