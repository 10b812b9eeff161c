// align_device.h -- gfx950 device code of BatchAlign: banded DP with direction bits + argmax, and
// the backtracking walk.  Replaces NextGenMap's oclSW_Score / oclSW_ScoreGlobal
// (lib/mason/opencl/opencl/oclSwScore.cl:219-329, oclEndFreeScore.cl:206-326) and
// oclSW_Backtracking (oclSwCigar.cl:2-56).
//
// Differences in decomposition (results identical):
//   * the reference spills one byte per DP cell to a global matrix; here a cell's direction is 2 bits
//     (0 stop, 1 diagonal, 2 up = insertion, 3 left = deletion), a band row is DW = ceil(2C/32)
//     dwords per pair, stored [block][row][word][lane] so every store is a coalesced 256-byte line;
//     '=' vs 'X' is not stored at all -- it is a function of the two characters and is re-derived
//     when the CIGAR/MD strings are built (cigar_md.h);
//   * the reference emits a right-aligned (len<<4|op) short array of 2*(2q+c+1) elements per pair;
//     the walk here emits compact runs in traceback order plus a fixed 8-int record.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sw_device.h"

namespace ngm {

__host__ __device__ constexpr int dir_words(int C) { return (2 * C + 31) / 32; }
__host__ __device__ constexpr int run_stride(int q, int c) { return (q + c + 8 + 7) & ~7; }

enum { kDirStop = 0, kDirDiag = 1, kDirUp = 2, kDirLeft = 3 };
// run ops emitted by the traceback: 0 = forced mismatch column (band border in end-to-end mode, the
// reference stores CIGAR_X there: oclEndFreeScore.cl:251, :297), 1 = diagonal, 2 = insertion, 3 = deletion
enum { kRunBorderX = 0, kRunDiag = 1, kRunIns = 2, kRunDel = 3 };
enum { kRecValid = 0, kRecPos = 1, kRecQStart = 2, kRecQEnd = 3, kRecRuns = 4, kRecScore = 5, kRecBri = 6, kRecBci = 7 };

template <int C, bool ENDFREE>
__global__ __launch_bounds__(256) void sw_align_kernel(const uint32_t *__restrict__ packed,
		const uint16_t *__restrict__ lens, const uint16_t *__restrict__ blk_rows, uint32_t *__restrict__ dirs,
		int32_t *__restrict__ records, int n, int n_blocks, int RW, int q, SwConst K) {
	__shared__ uint2 s_tab[16];
	if (threadIdx.x < 16) s_tab[threadIdx.x] = make_row_table(threadIdx.x, K);
	__syncthreads();
	const int lane = threadIdx.x & 63;
	const int blk = blockIdx.x * 4 + (threadIdx.x >> 6);
	if (blk >= n_blocks) return;
	constexpr int NRG = sel_regs(C);
	constexpr int DW = dir_words(C);
	const int FW = RW + NRG / 2;
	const uint32_t *rd = packed + (size_t) blk * (RW + FW) * kSlots + lane;
	const uint32_t *fd = rd + (size_t) RW * kSlots;
	uint32_t *dout = dirs + (size_t) blk * q * DW * kSlots + lane;
	const int pair = blk * kSlots + lane;
	const int len = (pair < n) ? (int) lens[pair] : 0;  // chars before the first NUL: rows the reference visits
	const int rows = __builtin_amdgcn_readfirstlane((int) blk_rows[blk]);
	const int ngroups = (rows + 7) >> 3;

	int H[C];
#pragma unroll
	for (int d = 0; d < C; ++d) H[d] = 0;
	uint32_t RG[NRG];
#pragma unroll
	for (int r = 0; r < NRG / 2; ++r) {
		const uint32_t x = fd[(size_t) r * kSlots];
		RG[2 * r] = x & 0x0F0F0F0Fu;
		RG[2 * r + 1] = (x >> 4) & 0x0F0F0F0Fu;
	}
	int fl = K.tZ;
	int best = -1, bri = 0, bci = 0;  // oclSwScore.cl:241-242
	uint32_t rnext = (ngroups > 0) ? rd[0] : 0x66666666u;

	for (int g = 0; g < ngroups; ++g) {
		const uint32_t rx = rnext;
		rnext = (g + 1 < ngroups) ? rd[(size_t) (g + 1) * kSlots] : 0x66666666u;
		const uint32_t fx = fd[(size_t) (g + NRG / 2) * kSlots];
		const uint32_t rsel[2] = {rx & 0x0F0F0F0Fu, (rx >> 4) & 0x0F0F0F0Fu};
#pragma unroll
		for (int s = 0; s < 8; ++s) {
			const int i = g * 8 + s;
			uint32_t rc = (rsel[s >> 2] >> (8 * (s & 3))) & 0xFFu;
			rc = (i < len) ? rc : 6u;  // the reference stops at the first NUL (oclSwScore.cl:257)
			const uint2 T = s_tab[rc];
			uint32_t P[NRG];
#pragma unroll
			for (int r = 0; r < NRG; ++r) P[r] = ((s + C - 1) / 4 >= r && s / 4 <= r) ? __builtin_amdgcn_perm(T.y, T.x, RG[r]) : 0u;
			int left = ENDFREE ? (fl + kShortMin) : fl;
			uint32_t dw[DW];
#pragma unroll
			for (int w = 0; w < DW; ++w) dw[w] = 0;
			int rowkey = -1;
#pragma unroll
			for (int d = 0; d < C; ++d) {
				const int bi = s + d;
				const int t = (int) ((P[bi >> 2] >> (8 * (bi & 3))) & 0xFFu);
				const int hold = H[d];
				const int dg = hold + t;
				const int a = left + K.gl;
				int b, h;
				if (d < C - 1) b = H[d + 1] + K.gu;
				else b = ENDFREE ? (fl + kShortMin + K.gap_read) : (fl + K.gap_read);  // sentinel column
				h = max(max(a, b), dg);
				if (!ENDFREE) h = max(h, fl);
				// direction, reference priority: stop, diagonal (incl. the "== prev + mismatch" clause,
				// oclSwScore.cl:294-302 -- prev + mismatch re-based is simply the old H'), up, left
				uint32_t code = kDirLeft;
				code = (h == b) ? (uint32_t) kDirUp : code;
				code = (h == dg || h == hold) ? (uint32_t) kDirDiag : code;
				if (!ENDFREE) code = (h == fl) ? (uint32_t) kDirStop : code;
				dw[d >> 4] |= code << (2 * (d & 15));
				if (!ENDFREE) rowkey = max(rowkey, ((h - fl) << 7) | (127 - d));
				H[d] = h;
				left = h;
			}
			if (i < q) {
#pragma unroll
				for (int w = 0; w < DW; ++w) dout[((size_t) i * DW + w) * kSlots] = dw[w];
			}
			if (!ENDFREE) {
				// first strict maximum in row-major order (oclSwScore.cl:307-311)
				const int rv = rowkey >> 7;
				if (i < len && rv > best) { best = rv; bri = i; bci = 127 - (rowkey & 127); }
			}
			fl += K.tZ;
		}
#pragma unroll
		for (int r = 0; r + 2 < NRG; ++r) RG[r] = RG[r + 2];
		RG[NRG - 2] = fx & 0x0F0F0F0Fu;
		RG[NRG - 1] = (fx >> 4) & 0x0F0F0F0Fu;
	}

	if (pair < n) {
		int qend;
		if (ENDFREE) {
			// argmax over the last row, first strict maximum (oclEndFreeScore.cl:308-315); rows past the
			// read end are NUL rows, which keep both the maximum and its first position
			const int klast = -(fl - K.tZ);
			best = kShortMin;
#pragma unroll
			for (int d = 0; d < C; ++d) { const int v = H[d] + klast; if (v > best) { best = v; bci = d; } }
			bri = len - 1;
			if (len == 0) bri = (K.variant == 1) ? -1 : 0;
			qend = 0;
		} else {
			qend = len - bri - 1;
		}
		int32_t *rec = records + (size_t) pair * 8;
		rec[kRecValid] = 0;
		rec[kRecPos] = 0;
		rec[kRecQStart] = 0;
		rec[kRecQEnd] = qend;
		rec[kRecRuns] = 0;
		rec[kRecScore] = best;
		rec[kRecBri] = bri;
		rec[kRecBci] = bci;
	}
}

#ifdef NGM_ENGINE_KERNELS
// One lane per pair walks the direction bits back from the argmax (oclSwCigar.cl:13-54).
__global__ __launch_bounds__(256) void sw_traceback_kernel(const uint32_t *__restrict__ dirs,
		const uint16_t *__restrict__ lens, int32_t *__restrict__ records, uint16_t *__restrict__ runs, int n,
		int q, int C, int run_stride, int endfree) {
	const int pair = blockIdx.x * blockDim.x + threadIdx.x;
	if (pair >= n) return;
	(void) lens;
	int32_t *rec = records + (size_t) pair * 8;
	int row = rec[kRecBri], col = rec[kRecBci];
	if (row <= 0) return;  // the reference skips backtracking (oclSwCigar.cl:13); record stays invalid
	const int DW = dir_words(C);
	const uint32_t *dp = dirs + (size_t) (pair >> 6) * q * DW * kSlots + (pair & 63);
	uint16_t *out = runs + (size_t) pair * run_stride;
	int abs_ref = row + col;
	int nruns = 0, cur = -1, curlen = 0;
	for (;;) {
		int op;
		if (row < 0) break;  // matrix row 0 is all STOP
		if (col < 0 || col >= C) {
			if (!endfree) break;  // band borders are STOP in local mode (oclSwScore.cl:264, :316)
			op = kRunBorderX;     // ... and CIGAR_X in end-to-end mode
		} else {
			const uint32_t w = dp[((size_t) row * DW + (col >> 4)) * kSlots];
			op = (int) ((w >> (2 * (col & 15))) & 3u);
			if (op == kDirStop) break;
		}
		if (op == kRunDiag || op == kRunBorderX) { row -= 1; abs_ref -= 1; }
		else if (op == kRunIns) { row -= 1; col += 1; }
		else { col -= 1; abs_ref -= 1; }
		if (op == cur) curlen += 1;
		else {
			if (cur >= 0 && nruns < run_stride) out[nruns++] = (uint16_t) ((curlen << 2) | cur);
			cur = op;
			curlen = 1;
		}
	}
	if (cur >= 0 && nruns < run_stride) out[nruns++] = (uint16_t) ((curlen << 2) | cur);
	rec[kRecValid] = 1;
	rec[kRecPos] = abs_ref + 1;
	rec[kRecQStart] = row + 1;
	rec[kRecRuns] = nruns;
}

#endif  // NGM_ENGINE_KERNELS

}  // namespace ngm
