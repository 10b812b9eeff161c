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

// ---------------------------------------------------------------------------------------------
// sw_align_kernel, packed 16-bit, local mode: TWO pairs per lane (blocks 2w and 2w+1) as in sw_score_pk_kernel, every band
// value a 16-bit half of one VGPR.  Same recurrences, same direction priority (stop, diagonal incl. the "== prev + mismatch"
// clause, up, left) and the same first strict maximum in row-major order as the 32-bit kernel above; same `dirs` layout, so
// sw_traceback_kernel reads either.
//   * The direction code comes out of the packed arithmetic without compares: h is the maximum of its sources, so "h == x" is
//     min(h - x, 1) == 0 (unsigned for the old H[d], which may be above h), and
//     code = nz * (1 + nd * (1 + nb))   with nz = h above the floor, nd = not diagonal, nb = not up.
//   * Row key: (score << 5 | 31 - d) per half, the packed unsigned maximum finds the row's best score and its smallest band
//     column (scores below 2 048, at most 32 columns); WINDOW: (score - base) << 7 | 127 - d with the per-pair base that follows
//     the running row maximum, as in sw_affine_align_pk_kernel (affine_device.h) -- a row's maximum is at least the previous
//     row's minus one mismatch (the diagonal successor of that cell is in the band) and at most one match above it.
// Valid while the re-based values fit 16 bits (host-checked like the score kernel); end-to-end mode stays with the 32-bit kernel
// (its sentinels are the 16-bit minimum itself).
// ---------------------------------------------------------------------------------------------
template <int C, bool WINDOW>
__global__ __launch_bounds__(256, (C <= 32 ? 3 : 1)) void sw_align_pk_kernel(const uint32_t *__restrict__ packed,
		const uint16_t *__restrict__ lens, const uint16_t *__restrict__ blk_rows, uint32_t *__restrict__ dirs,
		int32_t *__restrict__ records, int n, int n_blocks, int RW, int q, SwConst K) {
	static_assert(WINDOW ? C <= 128 : C <= 32, "the row key keeps the band column in 7 / 5 bits");
	__shared__ uint2 s_tab[16];
	if (threadIdx.x < 16) s_tab[threadIdx.x] = make_row_table(threadIdx.x, K);
	__syncthreads();
	const int lane = threadIdx.x & 63;
	const int blkA = 2 * (blockIdx.x * 4 + (threadIdx.x >> 6));
	if (blkA >= n_blocks) return;
	const bool hasB = blkA + 1 < n_blocks;
	const int blkB = hasB ? blkA + 1 : blkA;
	constexpr int NRG = sel_regs(C);
	constexpr int DW = dir_words(C);
	const int FW = RW + NRG / 2;
	const uint32_t *rdA = packed + (size_t) blkA * (RW + FW) * kSlots + lane, *rdB = packed + (size_t) blkB * (RW + FW) * kSlots + lane;
	const uint32_t *fdA = rdA + (size_t) RW * kSlots, *fdB = rdB + (size_t) RW * kSlots;
	uint32_t *doutA = dirs + (size_t) blkA * q * DW * kSlots + lane, *doutB = dirs + (size_t) blkB * q * DW * kSlots + lane;
	const int pairA = blkA * kSlots + lane, pairB = blkB * kSlots + lane;
	const int lenA = (pairA < n) ? (int) lens[pairA] : 0, lenB = (pairB < n) ? (int) lens[pairB] : 0;
	const int rows = max(__builtin_amdgcn_readfirstlane((int) blk_rows[blkA]), __builtin_amdgcn_readfirstlane((int) blk_rows[blkB]));
	const int ngroups = (rows + 7) >> 3;

	v2s H[C];
#pragma unroll
	for (int d = 0; d < C; ++d) H[d] = pk_splat(0);
	uint32_t RGA[NRG], RGB[NRG];
#pragma unroll
	for (int r = 0; r < NRG / 2; ++r) {
		const uint32_t xa = fdA[(size_t) r * kSlots], xb = fdB[(size_t) r * kSlots];
		RGA[2 * r] = xa & 0x0F0F0F0Fu; RGA[2 * r + 1] = (xa >> 4) & 0x0F0F0F0Fu;
		RGB[2 * r] = xb & 0x0F0F0F0Fu; RGB[2 * r + 1] = (xb >> 4) & 0x0F0F0F0Fu;
	}
	int fl = K.tZ;
	int bestA = -1, briA = 0, bciA = 0, bestB = -1, briB = 0, bciB = 0;  // oclSwScore.cl:241-242
	const v2s gl2 = pk_splat(K.gl), gu2 = pk_splat(K.gu), one2 = pk_splat(1), zero2 = pk_splat(0), k32 = pk_splat(WINDOW ? 128 : 32);
	const v2s win_lo2 = pk_splat(K.tZ + 1), win_max2 = pk_splat(63);  // K.tZ = -mismatch
	v2s prevmax2 = pk_splat(0);  // the previous row's maximum of both pairs
	uint32_t rnA = (ngroups > 0) ? rdA[0] : 0x66666666u, rnB = (ngroups > 0) ? rdB[0] : 0x66666666u;

	for (int g = 0; g < ngroups; ++g) {
		const uint32_t rxA = rnA, rxB = rnB;
		rnA = (g + 1 < ngroups) ? rdA[(size_t) (g + 1) * kSlots] : 0x66666666u;
		rnB = (g + 1 < ngroups) ? rdB[(size_t) (g + 1) * kSlots] : 0x66666666u;
		const uint32_t fxA = fdA[(size_t) (g + NRG / 2) * kSlots], fxB = fdB[(size_t) (g + NRG / 2) * kSlots];
		const uint32_t rsA[2] = {rxA & 0x0F0F0F0Fu, (rxA >> 4) & 0x0F0F0F0Fu}, rsB[2] = {rxB & 0x0F0F0F0Fu, (rxB >> 4) & 0x0F0F0F0Fu};
#pragma unroll
		for (int s = 0; s < 8; ++s) {
			const int i = g * 8 + s;
			uint32_t rcA = (rsA[s >> 2] >> (8 * (s & 3))) & 0xFFu, rcB = (rsB[s >> 2] >> (8 * (s & 3))) & 0xFFu;
			rcA = (i < lenA) ? rcA : 6u;  // the reference stops at the first NUL (oclSwScore.cl:257)
			rcB = (i < lenB) ? rcB : 6u;
			const uint2 TA = s_tab[rcA], TB = s_tab[rcB];
			uint32_t PA[NRG], PB[NRG];
#pragma unroll
			for (int r = 0; r < NRG; ++r) {
				const bool used = (s + C - 1) / 4 >= r && s / 4 <= r;
				PA[r] = used ? __builtin_amdgcn_perm(TA.y, TA.x, RGA[r]) : 0u;
				PB[r] = used ? __builtin_amdgcn_perm(TB.y, TB.x, RGB[r]) : 0u;
			}
			// (plain key: + min(previous row maximum, 0) = + 0 makes the row wait for the previous one -- scheduled freely the rows of a
			// group overlap and the kernel needs 300 registers instead of 140)
			const v2s fl2 = WINDOW ? pk_splat(fl) : pk_splat(fl) + pk_min_op(prevmax2, zero2);
			const v2s sentinel2 = pk_splat(fl + K.gap_read);  // column beyond the band: 0 + gap_read, re-based
			v2s left = fl2;
			v2u rowkey = __builtin_bit_cast(v2u, zero2);
			const v2s base2 = WINDOW ? pk_max(prevmax2 - win_lo2, zero2) : zero2;
			uint32_t acc[2] = {0u, 0u};  // 8 direction codes per pair each: low halves pair A, high halves pair B
#pragma unroll
			for (int d = 0; d < C; ++d) {
				const int bi = s + d, kb = bi & 3;
				const uint32_t sel = 0x0C000C00u | (uint32_t) kb | ((uint32_t) (4 + kb) << 16);
				const v2s t = __builtin_bit_cast(v2s, __builtin_amdgcn_perm(PB[bi >> 2], PA[bi >> 2], sel));
				const v2s hold = H[d];
				const v2s dg = hold + t;
				const v2s a = left + gl2;
				const v2s b = (d < C - 1) ? H[d + 1] + gu2 : sentinel2;
				const v2s h = pk_max(pk_max(pk_max(a, b), dg), fl2);
				const v2s pos = h - fl2;                                   // the cell's score
				const v2s nz = pk_min1_op(pos);                            // 0: stop
				const v2s nd = pk_min1_u_op(pk_min_u_op(h - dg, h - hold));  // 0: h == dg or h == old H[d] (diagonal)
				const v2s nb = pk_min1_op(h - b);                          // 0: h == b (up)
				const v2s code = pk_mul(nz, pk_mad_k(nd, nb + one2, 1));   // 0 stop, 1 diagonal, 2 up, 3 left
				acc[(d >> 3) & 1] |= __builtin_bit_cast(uint32_t, code) << (2 * (d & 7));
				if (WINDOW) rowkey = __builtin_elementwise_max(rowkey, __builtin_bit_cast(v2u, pk_mad_k(pk_min(pk_max(pos - base2, zero2), win_max2), k32, 127 - d)));
				else rowkey = __builtin_elementwise_max(rowkey, __builtin_bit_cast(v2u, pk_mad_k(pos, k32, 31 - d)));
				H[d] = h;
				left = h;
				if ((d & 15) == 15 || d == C - 1) {
					if (i < q) {
						doutA[((size_t) i * DW + (d >> 4)) * kSlots] = (acc[0] & 0xFFFFu) | (acc[1] << 16);
						if (hasB) doutB[((size_t) i * DW + (d >> 4)) * kSlots] = (acc[0] >> 16) | (acc[1] & 0xFFFF0000u);
					}
					acc[0] = acc[1] = 0u;
				}
			}
			{
				// first strict maximum in row-major order (oclSwScore.cl:307-311)
				const int ka = (int) rowkey.x, kb2 = (int) rowkey.y;
				constexpr int SH = WINDOW ? 7 : 5, DM = WINDOW ? 127 : 31;
				const int rva = (ka >> SH) + (int) base2.x, rvb = (kb2 >> SH) + (int) base2.y;
				prevmax2.x = (short) rva; prevmax2.y = (short) rvb;
				if (i < lenA && rva > bestA) { bestA = rva; briA = i; bciA = DM - (ka & DM); }
				if (i < lenB && rvb > bestB) { bestB = rvb; briB = i; bciB = DM - (kb2 & DM); }
			}
			fl += K.tZ;
		}
#pragma unroll
		for (int r = 0; r + 2 < NRG; ++r) { RGA[r] = RGA[r + 2]; RGB[r] = RGB[r + 2]; }
		RGA[NRG - 2] = fxA & 0x0F0F0F0Fu; RGA[NRG - 1] = (fxA >> 4) & 0x0F0F0F0Fu;
		RGB[NRG - 2] = fxB & 0x0F0F0F0Fu; RGB[NRG - 1] = (fxB >> 4) & 0x0F0F0F0Fu;
	}
	if (pairA < n) {
		int32_t *rec = records + (size_t) pairA * 8;
		rec[kRecValid] = 0; rec[kRecPos] = 0; rec[kRecQStart] = 0; rec[kRecQEnd] = lenA - briA - 1; rec[kRecRuns] = 0;
		rec[kRecScore] = bestA; rec[kRecBri] = briA; rec[kRecBci] = bciA;
	}
	if (hasB && pairB < n) {
		int32_t *rec = records + (size_t) pairB * 8;
		rec[kRecValid] = 0; rec[kRecPos] = 0; rec[kRecQStart] = 0; rec[kRecQEnd] = lenB - briB - 1; rec[kRecRuns] = 0;
		rec[kRecScore] = bestB; rec[kRecBri] = briB; rec[kRecBci] = bciB;
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
