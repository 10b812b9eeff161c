// affine_device.h -- gfx950 kernels of the affine-gap personality (`ngm --affine`): NextGenMap's
// EndToEndAffine::BatchScore / BatchAlign, i.e. SeqAn 1.4.1's banded Gotoh alignment with band diagonals
// 0..corridor (src/seqan/EndToEndAffine.cpp:10-52, lib/seqan-library-1.4.1/include/seqan/align/
// dp_formula_affine.h:390-415, dp_formula.h:152-160, dp_scout.h:142-155, dp_traceback_impl.h:184-470).
//
// Same decomposition as the linear kernels (sw_device.h / align_device.h): one pair per lane, the band row in
// VGPRs, 8 read rows per loop trip from the interleaved packed stream, substitution score by v_perm_b32.
// Geometry: SeqAn walks columns h (reference) x rows v (read); here a DP row is a read position v and the band
// column is the diagonal d = h - v in [0, corridor] (CP = corridor + 1 columns).  Three values per cell:
// S (best), Eh (alignment ending in a reference-only column = horizontal gap), Ev (read-only column = vertical).
//   Eh(v,d) = max(Eh(v,d-1) + ext, S(v,d-1) + open)        open wins only if strictly greater
//   Ev(v,d) = max(Ev(v-1,d+1) + ext, S(v-1,d+1) + open)
//   S(v,d)  = diagonal if S(v-1,d) + sub >= max(Ev, Eh) else the gap maximum (vertical preferred on ties)
//   local:  S <= 0  ->  S = Eh = Ev = 0
// Characters compare by value in SeqAn (N matches N); on NextGenMap's alphabet (reads ACGTN, windows ACGTNx) that
// is equality of the symbol classes, with the read's NUL padding never matching.
// Values are re-based per row exactly like the linear kernels (X' = X - v * mismatch) so the table bytes are
// {0, match - mismatch}; "not reachable" is a large negative number.
// The argmax follows SeqAn's scout: first strict maximum in column-major order = smallest h, then smallest v.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sw_device.h"

namespace ngm {

struct AffConst {
	int tM;     // match - mismatch
	int tZ;     // -mismatch : per-row floor step
	int open;   // gap open (first gap character), negative
	int ext;    // gap extend, negative
	int vopen;  // open - mismatch  (vertical predecessor lives one row up)
	int vext;   // ext - mismatch
};

constexpr int kAffNeg = -(1 << 28);
__host__ __device__ constexpr int aff_dir_words(int CP) { return (CP + 3) / 4; }  // one trace byte per cell

// SeqAn's trace bits (align/dp_profile.h:116-123)
enum { kTDiag = 1, kTHori = 2, kTVert = 4, kTHoriOpen = 8, kTVertOpen = 16, kTMaxH = 32, kTMaxV = 64 };
// affine records reuse the 8-int layout of align_device.h; rec[6]/rec[7] = end cell (h, v), and
// rec[5] = best score, rec[3] bit0/bit1 = "Ev == S" / "Eh == S" at the end cell (for _correctTraceValue)

__device__ __forceinline__ uint2 aff_row_table(int rc, const AffConst &K) {
	uint32_t b[8];
	// class 4 ("other": the 'x' filler of the window decoder) never equals a read character
	for (int fc = 0; fc < 8; ++fc) b[fc] = (rc <= 5 && rc != 4 && fc == rc) ? (uint32_t) (K.tM & 0xFF) : 0u;
	uint2 r;
	r.x = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24);
	r.y = b[4] | (b[5] << 8) | (b[6] << 16) | (b[7] << 24);
	return r;
}

// CP = corridor + 1 band columns.  ALIGN: also write the trace bytes and the end-cell record.
template <int CP, bool ENDFREE, bool ALIGN>
__global__ __launch_bounds__(256) void sw_affine_kernel(const uint32_t *__restrict__ packed, const uint16_t *__restrict__ lens,
		const uint16_t *__restrict__ blk_rows, float *__restrict__ scores, uint32_t *__restrict__ dirs, int32_t *__restrict__ records,
		int n, int n_blocks, int RW, int q, AffConst K) {
	__shared__ uint2 s_tab[8];
	if (threadIdx.x < 8) s_tab[threadIdx.x] = aff_row_table(threadIdx.x, K);
	__syncthreads();
	const int lane = threadIdx.x & 63;
	const int blk = blockIdx.x * 4 + (threadIdx.x >> 6);
	if (blk >= n_blocks) return;
	constexpr int NRG = sel_regs(CP);
	constexpr int DW = aff_dir_words(CP);
	const int FW = RW + NRG / 2;
	const uint32_t *rd = packed + (size_t) blk * (RW + FW) * kSlots + lane;
	const uint32_t *fd = rd + (size_t) RW * kSlots;
	uint32_t *dout = ALIGN ? dirs + (size_t) blk * q * DW * kSlots + lane : nullptr;
	const int pair = blk * kSlots + lane;
	const int lenV = (pair < n) ? (int) lens[pair] : 0;
	const int rows = __builtin_amdgcn_readfirstlane((int) blk_rows[blk]);
	const int ngroups = (rows + 7) >> 3;

	// window length |H| = first NUL class in the window (TSequence(refSeqList[i]) is a C string); only the
	// end-to-end scout needs it (the last row's cells with h > |H| do not exist)
	int lenH = FW * 8;
	if (ENDFREE) {
		for (int m = FW - 1; m >= 0; --m) {
			const uint32_t x = fd[(size_t) m * kSlots];
			const uint32_t lo = x & 0x0F0F0F0Fu, hi = (x >> 4) & 0x0F0F0F0Fu;
#pragma unroll
			for (int j = 7; j >= 0; --j) {
				const uint32_t cls = (j < 4) ? (lo >> (8 * j)) & 15u : (hi >> (8 * (j - 4))) & 15u;
				if (cls == 6u) lenH = m * 8 + j;
			}
		}
	}

	int S[CP], Ev[CP];
#pragma unroll
	for (int d = 0; d < CP; ++d) { S[d] = 0; Ev[d] = ENDFREE ? kAffNeg : 0; }  // row v = 0
	uint32_t RG[NRG];
#pragma unroll
	for (int r = 0; r < NRG / 2; ++r) {
		const uint32_t x = fd[(size_t) r * kSlots];
		RG[2 * r] = x & 0x0F0F0F0Fu;
		RG[2 * r + 1] = (x >> 4) & 0x0F0F0F0Fu;
	}
	int fl = K.tZ;  // re-based zero of row v = 1
	// scout state: maximum is initialised by the first tracked cell (0,0) with score 0 in local mode
	int best = ENDFREE ? kAffNeg : 0, bh = 0, bv = 0, bflags = 0;
	uint32_t rnext = (ngroups > 0) ? rd[0] : 0x66666666u;

	for (int g = 0; g < ngroups; ++g) {
		const uint32_t rx = rnext;
		rnext = (g + 1 < ngroups) ? rd[(size_t) (g + 1) * kSlots] : 0x66666666u;
		const uint32_t fx = fd[(size_t) (g + NRG / 2) * kSlots];
		const uint32_t rsel[2] = {rx & 0x0F0F0F0Fu, (rx >> 4) & 0x0F0F0F0Fu};
#pragma unroll
		for (int s = 0; s < 8; ++s) {
			const int i = g * 8 + s;  // read position, v = i + 1
			uint32_t rc = (rsel[s >> 2] >> (8 * (s & 3))) & 0xFFu;
			rc = (i < lenV) ? rc : 6u;
			const uint2 T = s_tab[rc];
			uint32_t P[NRG];
#pragma unroll
			for (int r = 0; r < NRG; ++r) P[r] = ((s + CP - 1) / 4 >= r && s / 4 <= r) ? __builtin_amdgcn_perm(T.y, T.x, RG[r]) : 0u;
			int leftS = kAffNeg, leftEh = kAffNeg;  // column d = 0 has no horizontal predecessor
			uint32_t dw[DW];
#pragma unroll
			for (int w = 0; w < DW; ++w) dw[w] = 0;
			int rowkey = -1, rowflags = 0;
			uint32_t ehq[(CP + 31) / 32];
#pragma unroll
			for (int w = 0; w < (CP + 31) / 32; ++w) ehq[w] = 0;
#pragma unroll
			for (int d = 0; d < CP; ++d) {
				const int bi = s + d;
				const int t = (int) ((P[bi >> 2] >> (8 * (bi & 3))) & 0xFFu);
				const int dg = S[d] + t;
				// horizontal gap (reference base only): same row, column d - 1
				int eh = leftEh + K.ext;
				const int eho = leftS + K.open;
				const bool h_open = eh < eho;
				eh = h_open ? eho : eh;
				// vertical gap (read base only): previous row, column d + 1
				int ev, evo;
				if (d < CP - 1) { ev = Ev[d + 1] + K.vext; evo = S[d + 1] + K.vopen; }
				else { ev = kAffNeg; evo = kAffNeg; }
				const bool v_open = ev < evo;
				ev = v_open ? evo : ev;
				if (d == 0) eh = kAffNeg;
				// gap maximum: vertical unless horizontal is strictly greater; diagonal wins ties against it
				const bool from_h = (d == CP - 1) ? true : ((d == 0) ? false : (ev < eh));
				const int gapmax = from_h ? eh : ev;
				const bool diag = gapmax <= dg;
				int sc = diag ? dg : gapmax;
				bool none = false;
				if (!ENDFREE) {
					none = sc <= fl;
					sc = none ? fl : sc;
					eh = none ? fl : eh;
					ev = none ? fl : ev;
				}
				if (ALIGN) {
					uint32_t tg = 0;
					if (d > 0) tg |= h_open ? (uint32_t) kTHoriOpen : (uint32_t) kTHori;
					if (d < CP - 1) tg |= v_open ? (uint32_t) kTVertOpen : (uint32_t) kTVert;
					uint32_t tr = diag ? (tg | (uint32_t) kTDiag) : (tg | (from_h ? (uint32_t) kTMaxH : (uint32_t) kTMaxV));
					tr = none ? 0u : tr;
					dw[d >> 2] |= tr << (8 * (d & 3));
				}
				const int key = ((sc - fl) << 8) | (255 - d);  // local: sc - fl >= 0
				if (!ENDFREE) {
					if (key > rowkey) { rowkey = key; rowflags = ((ev == sc) ? 1 : 0) | ((eh == sc) ? 2 : 0); }
				}
				if (ENDFREE && ALIGN) ehq[d >> 5] |= (eh == sc) ? (1u << (d & 31)) : 0u;
				S[d] = sc;
				Ev[d] = ev;
				leftS = sc;
				leftEh = eh;
			}
			if (ALIGN && i < q) {
#pragma unroll
				for (int w = 0; w < DW; ++w) dout[((size_t) i * DW + w) * kSlots] = dw[w];
			}
			if (!ENDFREE) {
				// column-major first maximum: strictly greater, or equal with a smaller h
				const int rs = rowkey >> 8, rd_ = 255 - (rowkey & 255), rh = i + 1 + rd_;
				if (i < lenV && (rs > best || (rs == best && rs > 0 && rh < bh))) { best = rs; bh = rh; bv = i + 1; bflags = rowflags; }
			} else if (i + 1 == lenV) {
				// end-to-end: the tracked cells are the last row's, h <= |H| only, first strict maximum in h
				const int dlim = lenH - lenV;
				const int kv = -fl;  // S = S' + v * mismatch
#pragma unroll
				for (int d = 0; d < CP; ++d) {
					const int v = S[d] + kv;
					if (d <= dlim && v > best) {
						best = v; bh = lenV + d; bv = lenV;
						bflags = ((Ev[d] == S[d]) ? 1 : 0) | ((ALIGN && ((ehq[d >> 5] >> (d & 31)) & 1u)) ? 2 : 0);
					}
				}
			}
			fl += K.tZ;
		}
#pragma unroll
		for (int r = 0; r + 2 < NRG; ++r) RG[r] = RG[r + 2];
		RG[NRG - 2] = fx & 0x0F0F0F0Fu;
		RG[NRG - 1] = (fx >> 4) & 0x0F0F0F0Fu;
	}

	if (pair < n) {
		if (lenV < 1 || lenH < 1) { best = 0; bh = bv = 0; bflags = 0; }  // EndToEndAffine: empty sequence, empty alignment
		if (!ALIGN) {
			scores[pair] = (float) best;
		} else {
			int32_t *rec = records + (size_t) pair * 8;
			rec[0] = 0; rec[1] = 0; rec[2] = 0; rec[3] = bflags; rec[4] = 0; rec[5] = best; rec[6] = bh; rec[7] = bv;
		}
	}
}

// Score-only variant, packed 16-bit: two pairs per lane (blocks 2w and 2w+1), S / Eh / Ev in the 16-bit halves of one
// VGPR, v_pk_add_i16 / v_pk_max_i16 on both pairs at once.  Only the maximum is needed here (no end cell), so SeqAn's
// tie rules drop out.  Of the local clamp "S <= 0 -> S = Eh = Ev = 0" only S = max(S, 0) is executed: where a cell is clamped
// Eh, Ev <= S <= 0 already, a gap state <= 0 only produces gap states < 0 (extension costs), and a value <= 0 never wins the
// maximum of a cell whose clamped score is > 0 -- so every positive S, Eh, Ev equals SeqAn's and the non-positive ones, which
// may differ from SeqAn's 0, are never read into a result (12 packed operations per cell pair instead of 21).
// Valid while the re-based values fit 16 bits (checked by the host, which otherwise uses the 32-bit kernel).
constexpr int kAffNeg16 = -20000;

template <int CP, bool ENDFREE>
__global__ __launch_bounds__(256) void sw_affine_score_pk_kernel(const uint32_t *__restrict__ packed, const uint16_t *__restrict__ lens,
		const uint16_t *__restrict__ blk_rows, float *__restrict__ scores, int n, int n_blocks, int RW, AffConst K) {
	__shared__ uint2 s_tab[8];
	if (threadIdx.x < 8) s_tab[threadIdx.x] = aff_row_table(threadIdx.x, K);
	__syncthreads();
	const int lane = threadIdx.x & 63;
	const int blkA = 2 * (blockIdx.x * 4 + (threadIdx.x >> 6));
	if (blkA >= n_blocks) return;
	const bool hasB = blkA + 1 < n_blocks;
	const int blkB = hasB ? blkA + 1 : blkA;
	constexpr int NRG = sel_regs(CP);
	const int FW = RW + NRG / 2;
	const uint32_t *rdA = packed + (size_t) blkA * (RW + FW) * kSlots + lane, *rdB = packed + (size_t) blkB * (RW + FW) * kSlots + lane;
	const uint32_t *fdA = rdA + (size_t) RW * kSlots, *fdB = rdB + (size_t) RW * kSlots;
	const int pairA = blkA * kSlots + lane, pairB = blkB * kSlots + lane;
	const int lenVA = (pairA < n) ? (int) lens[pairA] : 0, lenVB = (pairB < n) ? (int) lens[pairB] : 0;
	const int rows = max(__builtin_amdgcn_readfirstlane((int) blk_rows[blkA]), __builtin_amdgcn_readfirstlane((int) blk_rows[blkB]));
	const int ngroups = (rows + 7) >> 3;

	auto window_len = [&](const uint32_t *fd) {  // first NUL class of the window (see sw_affine_kernel)
		int len = FW * 8;
		for (int m = FW - 1; m >= 0; --m) {
			const uint32_t x = fd[(size_t) m * kSlots];
			const uint32_t lo = x & 0x0F0F0F0Fu, hi = (x >> 4) & 0x0F0F0F0Fu;
#pragma unroll
			for (int j = 7; j >= 0; --j) {
				const uint32_t cls = (j < 4) ? (lo >> (8 * j)) & 15u : (hi >> (8 * (j - 4))) & 15u;
				if (cls == 6u) len = m * 8 + j;
			}
		}
		return len;
	};
	const int lenHA = ENDFREE ? window_len(fdA) : FW * 8, lenHB = ENDFREE ? window_len(fdB) : FW * 8;

	v2s S[CP], Ev[CP];
#pragma unroll
	for (int d = 0; d < CP; ++d) { S[d] = pk_splat(0); Ev[d] = pk_splat(ENDFREE ? kAffNeg16 : 0); }
	uint32_t RGA[NRG], RGB[NRG];
#pragma unroll
	for (int r = 0; r < NRG / 2; ++r) {
		const uint32_t xa = fdA[(size_t) r * kSlots], xb = fdB[(size_t) r * kSlots];
		RGA[2 * r] = xa & 0x0F0F0F0Fu; RGA[2 * r + 1] = (xa >> 4) & 0x0F0F0F0Fu;
		RGB[2 * r] = xb & 0x0F0F0F0Fu; RGB[2 * r + 1] = (xb >> 4) & 0x0F0F0F0Fu;
	}
	int fl = K.tZ;
	v2s best2 = pk_splat(0);                 // local: running maximum of both pairs
	int bestA = kAffNeg, bestB = kAffNeg;    // end-to-end: latched at each pair's last row
	const v2s ext2 = pk_splat(K.ext), open2 = pk_splat(K.open), vext2 = pk_splat(K.vext), vopen2 = pk_splat(K.vopen);
	const v2s neg2 = pk_splat(kAffNeg16), one2 = pk_splat(1), zero2 = pk_splat(0);
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
			rcA = (i < lenVA) ? rcA : 6u;
			rcB = (i < lenVB) ? rcB : 6u;
			const uint2 TA = s_tab[rcA], TB = s_tab[rcB];
			uint32_t PA[NRG], PB[NRG];
#pragma unroll
			for (int r = 0; r < NRG; ++r) {
				const bool used = (s + CP - 1) / 4 >= r && s / 4 <= r;
				PA[r] = used ? __builtin_amdgcn_perm(TA.y, TA.x, RGA[r]) : 0u;
				PB[r] = used ? __builtin_amdgcn_perm(TB.y, TB.x, RGB[r]) : 0u;
			}
			const v2s fl2 = pk_splat(fl);
			v2s leftS = neg2, leftEh = neg2;  // column d = 0 has no horizontal predecessor
			v2s rowmax = fl2;
#pragma unroll
			for (int d = 0; d < CP; ++d) {
				const int bi = s + d, kb = bi & 3;
				const uint32_t sel = 0x0C000C00u | (uint32_t) kb | ((uint32_t) (4 + kb) << 16);
				const v2s t = __builtin_bit_cast(v2s, __builtin_amdgcn_perm(PB[bi >> 2], PA[bi >> 2], sel));
				const v2s dg = S[d] + t;
				v2s eh = (d == 0) ? neg2 : pk_max(leftEh + ext2, leftS + open2);
				v2s ev = (d < CP - 1) ? pk_max(Ev[d + 1] + vext2, S[d + 1] + vopen2) : neg2;
				v2s sc = pk_max(pk_max(ev, eh), dg);
				if (!ENDFREE) {
					// local clamp: only S is clamped (see the note above the kernel)
					sc = pk_max(sc, fl2);
					rowmax = pk_max(rowmax, sc);
				}
				S[d] = sc;
				Ev[d] = ev;
				leftS = sc;
				leftEh = eh;
			}
			if (!ENDFREE) {
				best2 = pk_max(best2, rowmax - fl2);
			} else if (i + 1 == lenVA || i + 1 == lenVB) {
				// end-to-end: maximum of the pair's last row over the cells with h <= |H|
				const int kv = -fl;
				if (i + 1 == lenVA) {
					const int dlim = lenHA - lenVA;
#pragma unroll
					for (int d = 0; d < CP; ++d) { const int v = (int) S[d].x + kv; if (d <= dlim && v > bestA) bestA = v; }
				}
				if (i + 1 == lenVB) {
					const int dlim = lenHB - lenVB;
#pragma unroll
					for (int d = 0; d < CP; ++d) { const int v = (int) S[d].y + kv; if (d <= dlim && v > bestB) bestB = v; }
				}
			}
			fl += K.tZ;
		}
#pragma unroll
		for (int r = 0; r + 2 < NRG; ++r) { RGA[r] = RGA[r + 2]; RGB[r] = RGB[r + 2]; }
		RGA[NRG - 2] = fxA & 0x0F0F0F0Fu; RGA[NRG - 1] = (fxA >> 4) & 0x0F0F0F0Fu;
		RGB[NRG - 2] = fxB & 0x0F0F0F0Fu; RGB[NRG - 1] = (fxB >> 4) & 0x0F0F0F0Fu;
	}
	if (!ENDFREE) { bestA = best2.x; bestB = best2.y; }
	if (pairA < n) { if (lenVA < 1 || lenHA < 1) bestA = 0; scores[pairA] = (float) bestA; }
	if (hasB && pairB < n) { if (lenVB < 1 || lenHB < 1) bestB = 0; scores[pairB] = (float) bestB; }
}

// Score-only variant for WIDE bands (round 6; BASELINE config 5: 250 bp reads, corridor 80), local mode: the band's columns split over
// TWO lanes.  With 81 columns sw_affine_score_pk_kernel keeps S and Ev of the whole band in 162 VGPRs (338 with the window queue): one
// wave per SIMD, and the dependent packed operations of a row wait for each other with nothing to hide behind (3.4 Tcells/s against
// 5.1 at 28 columns).  Here lanes 2k / 2k + 1 share slot k of a block pair: the even lane owns band columns [0, CL), the odd lane
// [CL, CP), and the odd lane runs ONE ROW BEHIND, which is what makes the split legal --
//   Eh(v, d) needs (v, d - 1): the odd lane's first column of row v - 1 needs the even lane's last column of row v - 1, finished one
//   step earlier (kept in saveS / saveEh);
//   Ev(v, d) needs (v - 1, d + 1): the even lane's last column of row v needs the odd lane's first column of row v - 1, which the odd
//   lane computes first thing in the same step.
// Two DPP swaps in each direction per row; everything else is the one-lane kernel on half the columns: the odd lane's window queue
// starts CL - 1 bytes further (a multiple of eight: whole words), its read character and re-basing floor are the even lane's of the
// step before.  Step 0 has no row for the odd lane: it computes on a NUL class and is put back to the initial state afterwards (a
// wave-uniform branch, once per pair).  When CP is odd the odd lane's last column does not exist and is held at "unreachable".
// 181 VGPRs: two waves per SIMD.  (Held to 168 for three waves -- amdgpu_waves_per_eu(3, 3) -- the kernel spills 15 registers to scratch:
// alone it is faster still (282 against 335 ms per step of BASELINE config 5's shape, 5.6 against 4.6 Tcells/s), but the step is slower
// (500-512 against 466-476 ms, two alternating runs each): the search and align kernels of the other mapper instances took 1.5-5 x as long
// beside it.  profiles/r06_config5_split_kernel_ab.txt)
template <int CP>
__global__ __launch_bounds__(256) void sw_affine_score_pk_split_kernel(const uint32_t *__restrict__ packed, const uint16_t *__restrict__ lens,
		const uint16_t *__restrict__ blk_rows, float *__restrict__ scores, int n, int n_blocks, int RW, AffConst K) {
	__shared__ uint2 s_tab[8];
	if (threadIdx.x < 8) s_tab[threadIdx.x] = aff_row_table(threadIdx.x, K);
	__syncthreads();
	constexpr int CL = ((CP + 1) / 2 + 6) / 8 * 8 + 1;   // columns of the even lane: CL - 1 is a multiple of 8
	constexpr int CR = CP - CL;                          // real columns of the odd lane (the others are held unreachable)
	static_assert(CR >= 1 && CR <= CL, "band too narrow to split");
	constexpr int NRG = sel_regs(CL);
	constexpr int WOFF = (CL - 1) / 8;                   // the odd lane's window starts this many words further
	const int lane = threadIdx.x & 63;
	const bool isR = (lane & 1) != 0;
	const int unit = blockIdx.x * 4 + (threadIdx.x >> 6);   // (block pair, half of its 64 slots)
	const int blkA = 2 * (unit >> 1);
	if (blkA >= n_blocks) return;
	const bool hasB = blkA + 1 < n_blocks;
	const int blkB = hasB ? blkA + 1 : blkA;
	const int slot = (lane >> 1) + 32 * (unit & 1);
	const int FW = RW + sel_regs(CP) / 2;
	const uint32_t *rdA = packed + (size_t) blkA * (RW + FW) * kSlots + slot, *rdB = packed + (size_t) blkB * (RW + FW) * kSlots + slot;
	const uint32_t *fdA = rdA + (size_t) RW * kSlots, *fdB = rdB + (size_t) RW * kSlots;
	const int pairA = blkA * kSlots + slot, pairB = blkB * kSlots + slot;
	const int lenVA = (pairA < n) ? (int) lens[pairA] : 0, lenVB = (pairB < n) ? (int) lens[pairB] : 0;
	const int rows = max(__builtin_amdgcn_readfirstlane((int) blk_rows[blkA]), __builtin_amdgcn_readfirstlane((int) blk_rows[blkB]));
	const int ngroups = (rows + 7) >> 3, ngroups2 = (rows + 1 + 7) >> 3;   // (the odd lane's last row is one step later)
	const int woff = isR ? WOFF : 0;
	auto swap2 = [](v2s x) -> v2s { return __builtin_bit_cast(v2s, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xf, 0xf, false)); };   // quad_perm [1, 0, 3, 2]: the other lane of the pair

	v2s S[CL], Ev[CL];
#pragma unroll
	for (int d = 0; d < CL; ++d) { S[d] = pk_splat(0); Ev[d] = pk_splat(0); }
	uint32_t RGA[NRG], RGB[NRG];
#pragma unroll
	for (int r = 0; r < NRG / 2; ++r) {
		const int w = min(r + woff, FW - 1);
		const uint32_t xa = fdA[(size_t) w * kSlots], xb = fdB[(size_t) w * kSlots];
		RGA[2 * r] = xa & 0x0F0F0F0Fu; RGA[2 * r + 1] = (xa >> 4) & 0x0F0F0F0Fu;
		RGB[2 * r] = xb & 0x0F0F0F0Fu; RGB[2 * r + 1] = (xb >> 4) & 0x0F0F0F0Fu;
	}
	int fl = K.tZ;                           // re-based zero of the even lane's row
	const int rz = isR ? K.tZ : 0;           // ... the odd lane is one row behind
	v2s best2 = pk_splat(0);
	const v2s ext2 = pk_splat(K.ext), open2 = pk_splat(K.open), vext2 = pk_splat(K.vext), vopen2 = pk_splat(K.vopen);
	const v2s neg2 = pk_splat(kAffNeg16), zero2 = pk_splat(0);
	v2s saveS = neg2, saveEh = neg2;         // the even lane's last column of the step before
	uint32_t prevA = 6u, prevB = 6u;         // read classes of the step before (the odd lane's row)
	uint32_t rnA = (ngroups > 0) ? rdA[0] : 0x66666666u, rnB = (ngroups > 0) ? rdB[0] : 0x66666666u;

	for (int g = 0; g < ngroups2; ++g) {
		const uint32_t rxA = rnA, rxB = rnB;
		rnA = (g + 1 < ngroups) ? rdA[(size_t) (g + 1) * kSlots] : 0x66666666u;
		rnB = (g + 1 < ngroups) ? rdB[(size_t) (g + 1) * kSlots] : 0x66666666u;
		const int wn = min(g + NRG / 2 + woff, FW - 1);   // (beyond the window only behind the read's last row: those rows match nothing)
		const uint32_t fxA = fdA[(size_t) wn * kSlots], fxB = fdB[(size_t) wn * kSlots];
		const uint32_t rsA[2] = {rxA & 0x0F0F0F0Fu, (rxA >> 4) & 0x0F0F0F0Fu}, rsB[2] = {rxB & 0x0F0F0F0Fu, (rxB >> 4) & 0x0F0F0F0Fu};
#pragma unroll
		for (int s = 0; s < 8; ++s) {
			const int i = g * 8 + s;   // the even lane's read position; the odd lane's is i - 1
			uint32_t curA = (rsA[s >> 2] >> (8 * (s & 3))) & 0xFFu, curB = (rsB[s >> 2] >> (8 * (s & 3))) & 0xFFu;
			curA = (i < lenVA) ? curA : 6u;
			curB = (i < lenVB) ? curB : 6u;
			const uint32_t rcA = isR ? prevA : curA, rcB = isR ? prevB : curB;
			prevA = curA; prevB = curB;
			const uint2 TA = s_tab[rcA], TB = s_tab[rcB];
			uint32_t PA[NRG], PB[NRG];
#pragma unroll
			for (int r = 0; r < NRG; ++r) {
				const bool used = (s + CL - 1) / 4 >= r && s / 4 <= r;
				PA[r] = used ? __builtin_amdgcn_perm(TA.y, TA.x, RGA[r]) : 0u;
				PB[r] = used ? __builtin_amdgcn_perm(TB.y, TB.x, RGB[r]) : 0u;
			}
			const v2s fl2 = pk_splat(fl - rz);
			// the odd lane's horizontal predecessor: the even lane's last column of the same row (the step before)
			const v2s gotS = swap2(saveS), gotEh = swap2(saveEh);
			v2s leftS = isR ? gotS : neg2, leftEh = isR ? gotEh : neg2;
			v2s nxS = neg2, nxEv = neg2;   // the even lane's vertical predecessor of its last column: the odd lane's first column (below)
			v2s rowmax = fl2;
#pragma unroll
			for (int d = 0; d < CL; ++d) {
				const int bi = s + d, kb = bi & 3;
				const uint32_t sel = 0x0C000C00u | (uint32_t) kb | ((uint32_t) (4 + kb) << 16);
				const v2s t = __builtin_bit_cast(v2s, __builtin_amdgcn_perm(PB[bi >> 2], PA[bi >> 2], sel));
				const v2s dg = S[d] + t;
				const v2s eh = pk_max(leftEh + ext2, leftS + open2);
				v2s ev = (d < CL - 1) ? pk_max(Ev[d + 1] + vext2, S[d + 1] + vopen2) : pk_max(nxEv + vext2, nxS + vopen2);
				v2s sc = pk_max(pk_max(pk_max(ev, eh), dg), fl2);   // local clamp: only S (see sw_affine_score_pk_kernel)
				if (d >= CR) { sc = isR ? neg2 : sc; ev = isR ? neg2 : ev; }   // (CP odd: the odd lane's last column does not exist)
				rowmax = pk_max(rowmax, sc);
				S[d] = sc;
				Ev[d] = ev;
				leftS = sc;
				leftEh = eh;
				if (d == 0) {
					v2s xS = swap2(sc), xEv = swap2(ev);
					if (s == 0 && g == 0) { xS = zero2; xEv = zero2; }   // (wave-uniform) step 0: the odd lane's row 0 is the initial state
					nxS = isR ? neg2 : xS; nxEv = isR ? neg2 : xEv;
				}
				if (d == CL - 1) { saveS = sc; saveEh = eh; }
			}
			best2 = pk_max(best2, rowmax - fl2);
			if (s == 0 && g == 0) {   // (wave-uniform, once) the odd lane had no row in this step: back to the initial state
#pragma unroll
				for (int d = 0; d < CL; ++d) { S[d] = isR ? zero2 : S[d]; Ev[d] = isR ? zero2 : Ev[d]; }
				best2 = isR ? zero2 : best2;
			}
			fl += K.tZ;
		}
#pragma unroll
		for (int r = 0; r + 2 < NRG; ++r) { RGA[r] = RGA[r + 2]; RGB[r] = RGB[r + 2]; }
		RGA[NRG - 2] = fxA & 0x0F0F0F0Fu; RGA[NRG - 1] = (fxA >> 4) & 0x0F0F0F0Fu;
		RGB[NRG - 2] = fxB & 0x0F0F0F0Fu; RGB[NRG - 1] = (fxB >> 4) & 0x0F0F0F0Fu;
	}
	best2 = pk_max(best2, swap2(best2));
	if (isR) return;
	int bestA = best2.x, bestB = best2.y;
	if (pairA < n) { if (lenVA < 1) bestA = 0; scores[pairA] = (float) bestA; }
	if (hasB && pairB < n) { if (lenVB < 1) bestB = 0; scores[pairB] = (float) bestB; }
}

// Align variant, packed 16-bit, local mode: two pairs per lane like the score kernel above, plus what the traceback needs.
// * Trace: 4 bits per cell instead of SeqAn's 7-bit value in a byte -- "gap opened here" for the horizontal and the vertical
//   gap (bits 0, 1) and which of {none, diagonal, horizontal maximum, vertical maximum} the cell took (bits 2-3); the Hori /
//   Vert presence bits depend on the band column alone and are put back by affine_traceback_kernel (aff_trace_from_nibble).
//   The flags come out of the packed arithmetic without compares: a = max(x, y) opened / switched exactly where a - x > 0,
//   i.e. min(a - x, 1).
// * End cell: SeqAn's scout keeps the first strict maximum in column-major order; per row the packed unsigned maximum of
//   (score << 5 | 31 - d) finds the row's best score and its smallest column, the rule across rows is applied per pair as
//   in sw_affine_kernel.  Needs score < 2048 and CP <= 32 (checked by the host).
// * _correctTraceValue's "Ev == S" / "Eh == S" flags at the end cell are always clear here: with negative gap penalties a
//   gap state is strictly below some earlier S, and the end cell holds the maximum of all S (the host takes this kernel only
//   then; end-to-end alignments -- which may end in a gap -- stay with sw_affine_kernel).
__host__ __device__ constexpr int aff_nib_words(int CP) { return (CP + 7) / 8; }

// WINDOW: for bands of more than 32 columns or scores of 2 048 and more the row key is (score - base) << 7 | 127 - d with a
// per-pair base that follows the running row maximum: a row's maximum is at least the previous row's minus one mismatch (the
// diagonal successor of that cell is in the band) and at most one match above it, so the scores that can be the row maximum
// lie in a window of mismatch + match + 1 values (<= 63: host-checked) above base = max(previous row maximum - mismatch - 1, 0);
// cells below the window clamp to 0 and cannot win.
template <int CP, bool WINDOW>
__global__ __launch_bounds__(256) void sw_affine_align_pk_kernel(const uint32_t *__restrict__ packed, const uint16_t *__restrict__ lens,
		const uint16_t *__restrict__ blk_rows, uint32_t *__restrict__ dirs, int32_t *__restrict__ records, int n, int n_blocks, int RW, int q, AffConst K) {
	static_assert(WINDOW ? CP <= 128 : CP <= 32, "the row key keeps the band column in 7 / 5 bits");
	__shared__ uint2 s_tab[8];
	if (threadIdx.x < 8) s_tab[threadIdx.x] = aff_row_table(threadIdx.x, K);
	__syncthreads();
	const int lane = threadIdx.x & 63;
	const int blkA = 2 * (blockIdx.x * 4 + (threadIdx.x >> 6));
	if (blkA >= n_blocks) return;
	const bool hasB = blkA + 1 < n_blocks;
	const int blkB = hasB ? blkA + 1 : blkA;
	constexpr int NRG = sel_regs(CP);
	constexpr int DW = aff_nib_words(CP);
	const int FW = RW + NRG / 2;
	const uint32_t *rdA = packed + (size_t) blkA * (RW + FW) * kSlots + lane, *rdB = packed + (size_t) blkB * (RW + FW) * kSlots + lane;
	const uint32_t *fdA = rdA + (size_t) RW * kSlots, *fdB = rdB + (size_t) RW * kSlots;
	uint32_t *doutA = dirs + (size_t) blkA * q * DW * kSlots + lane, *doutB = dirs + (size_t) blkB * q * DW * kSlots + lane;
	const int pairA = blkA * kSlots + lane, pairB = blkB * kSlots + lane;
	const int lenVA = (pairA < n) ? (int) lens[pairA] : 0, lenVB = (pairB < n) ? (int) lens[pairB] : 0;
	const int rows = max(__builtin_amdgcn_readfirstlane((int) blk_rows[blkA]), __builtin_amdgcn_readfirstlane((int) blk_rows[blkB]));
	const int ngroups = (rows + 7) >> 3;

	v2s S[CP], Ev[CP];
#pragma unroll
	for (int d = 0; d < CP; ++d) { S[d] = pk_splat(0); Ev[d] = pk_splat(0); }
	uint32_t RGA[NRG], RGB[NRG];
#pragma unroll
	for (int r = 0; r < NRG / 2; ++r) {
		const uint32_t xa = fdA[(size_t) r * kSlots], xb = fdB[(size_t) r * kSlots];
		RGA[2 * r] = xa & 0x0F0F0F0Fu; RGA[2 * r + 1] = (xa >> 4) & 0x0F0F0F0Fu;
		RGB[2 * r] = xb & 0x0F0F0F0Fu; RGB[2 * r + 1] = (xb >> 4) & 0x0F0F0F0Fu;
	}
	int fl = K.tZ;
	int bestA = 0, bhA = 0, bvA = 0, bestB = 0, bhB = 0, bvB = 0;
	const v2s ext2 = pk_splat(K.ext), open2 = pk_splat(K.open), vext2 = pk_splat(K.vext), vopen2 = pk_splat(K.vopen);
	const v2s neg2 = pk_splat(kAffNeg16), one2 = pk_splat(1), zero2 = pk_splat(0), two2 = pk_splat(2), four2 = pk_splat(4), k32 = pk_splat(WINDOW ? 128 : 32);
	const v2s win_lo2 = pk_splat(K.tZ + 1), win_max2 = pk_splat(63);  // K.tZ = -mismatch
	v2s prevmax2 = pk_splat(0);  // WINDOW: the previous row's maximum of both pairs
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
			rcA = (i < lenVA) ? rcA : 6u;
			rcB = (i < lenVB) ? rcB : 6u;
			const uint2 TA = s_tab[rcA], TB = s_tab[rcB];
			uint32_t PA[NRG], PB[NRG];
#pragma unroll
			for (int r = 0; r < NRG; ++r) {
				const bool used = (s + CP - 1) / 4 >= r && s / 4 <= r;
				PA[r] = used ? __builtin_amdgcn_perm(TA.y, TA.x, RGA[r]) : 0u;
				PB[r] = used ? __builtin_amdgcn_perm(TB.y, TB.x, RGB[r]) : 0u;
			}
			const v2s fl2 = pk_splat(fl);
			v2s leftS = neg2, leftEh = neg2;  // column d = 0 has no horizontal predecessor
			v2u rowkey = __builtin_bit_cast(v2u, zero2);
			const v2s base2 = WINDOW ? pk_max(prevmax2 - win_lo2, zero2) : zero2;
			uint32_t acc[2] = {0u, 0u};       // 4 trace nibbles per pair each: low halves pair A, high halves pair B
#pragma unroll
			for (int d = 0; d < CP; ++d) {
				const int bi = s + d, kb = bi & 3;
				const uint32_t sel = 0x0C000C00u | (uint32_t) kb | ((uint32_t) (4 + kb) << 16);
				const v2s t = __builtin_bit_cast(v2s, __builtin_amdgcn_perm(PB[bi >> 2], PA[bi >> 2], sel));
				const v2s dg = S[d] + t;
				v2s eh = neg2, ho = zero2;   // ho / vo: 1 where the gap was opened in this cell (strictly better than extending)
				// (the flag differences saturate: "unreachable" is -20 000, and with long reads the re-based scores pass +12 767)
				if (d > 0) { const v2s e = leftEh + ext2; eh = pk_max(e, leftS + open2); ho = pk_min(pk_sub_sat(eh, e), one2); }
				v2s ev = neg2, vo = zero2;
				if (d < CP - 1) { const v2s e = Ev[d + 1] + vext2; ev = pk_max(e, S[d + 1] + vopen2); vo = pk_min(pk_sub_sat(ev, e), one2); }
				// gap maximum: vertical unless horizontal is strictly greater (fh); the diagonal wins ties against it (nd = 0)
				v2s gm, fh;
				if (d == 0) { gm = ev; fh = zero2; } else if (d == CP - 1) { gm = eh; fh = one2; } else { gm = pk_max(ev, eh); fh = pk_min(pk_sub_sat(gm, ev), one2); }
				v2s sc = pk_max(gm, dg);
				const v2s nd = pk_min(pk_sub_sat(sc, dg), one2);
				const v2s pos = pk_max(sc - fl2, zero2);     // the cell's score; 0: clamped (S = 0, no trace)
				const v2s nz = pk_min1_op(pos);
				sc = pk_max(sc, fl2);
				// Eh / Ev are left as they are where the cell is clamped (SeqAn: 0): they are <= 0 there, and gap states <= 0 never reach a
				// cell of the path -- along it every S, and every gap state it walks through, is > 0 and equals SeqAn's, and "opened
				// here" compares such a value with the extension of a predecessor that is either equal to SeqAn's or, in both, below it
				// 0 none, 1 diagonal, 2 horizontal maximum, 3 vertical maximum -- times 4; (as instructions: the compiler turns a product
				// with a 0 / 1 factor into two compares and two selects per cell pair)
				const v2s took4 = pk_mul(pk_mad(nd, two2 - fh, one2), pk_mad(nz, four2, zero2));
				const v2s nib = took4 + pk_mad(vo, two2, ho);
				acc[(d >> 2) & 1] |= __builtin_bit_cast(uint32_t, nib) << (4 * (d & 3));
				if (WINDOW) rowkey = __builtin_elementwise_max(rowkey, __builtin_bit_cast(v2u, pk_mad_k(pk_min(pk_max(pos - base2, zero2), win_max2), k32, 127 - d)));
				else rowkey = __builtin_elementwise_max(rowkey, __builtin_bit_cast(v2u, pk_mad_k(pos, k32, 31 - d)));
				S[d] = sc;
				Ev[d] = ev;
				leftS = sc;
				leftEh = eh;
				if ((d & 7) == 7 || d == CP - 1) {
					if (i < q) {
						doutA[((size_t) i * DW + (d >> 3)) * kSlots] = (acc[0] & 0xFFFFu) | (acc[1] << 16);
						if (hasB) doutB[((size_t) i * DW + (d >> 3)) * kSlots] = (acc[0] >> 16) | (acc[1] & 0xFFFF0000u);
					}
					acc[0] = acc[1] = 0u;
				}
			}
			{
				// column-major first maximum: strictly greater, or equal with a smaller h
				const int ka = (int) rowkey.x, kb2 = (int) rowkey.y;
				constexpr int SH = WINDOW ? 7 : 5, DM = WINDOW ? 127 : 31;
				const int rsa = (ka >> SH) + (int) base2.x, rha = i + 1 + (DM - (ka & DM)), rsb = (kb2 >> SH) + (int) base2.y, rhb = i + 1 + (DM - (kb2 & DM));
				if (WINDOW) { prevmax2.x = (short) rsa; prevmax2.y = (short) rsb; }
				if (i < lenVA && (rsa > bestA || (rsa == bestA && rsa > 0 && rha < bhA))) { bestA = rsa; bhA = rha; bvA = i + 1; }
				if (i < lenVB && (rsb > bestB || (rsb == bestB && rsb > 0 && rhb < bhB))) { bestB = rsb; bhB = rhb; bvB = i + 1; }
			}
			fl += K.tZ;
		}
#pragma unroll
		for (int r = 0; r + 2 < NRG; ++r) { RGA[r] = RGA[r + 2]; RGB[r] = RGB[r + 2]; }
		RGA[NRG - 2] = fxA & 0x0F0F0F0Fu; RGA[NRG - 1] = (fxA >> 4) & 0x0F0F0F0Fu;
		RGB[NRG - 2] = fxB & 0x0F0F0F0Fu; RGB[NRG - 1] = (fxB >> 4) & 0x0F0F0F0Fu;
	}
	if (pairA < n) {
		if (lenVA < 1) { bestA = 0; bhA = bvA = 0; }
		int32_t *rec = records + (size_t) pairA * 8;
		rec[0] = 0; rec[1] = 0; rec[2] = 0; rec[3] = 0; rec[4] = 0; rec[5] = bestA; rec[6] = bhA; rec[7] = bvA;
	}
	if (hasB && pairB < n) {
		if (lenVB < 1) { bestB = 0; bhB = bvB = 0; }
		int32_t *rec = records + (size_t) pairB * 8;
		rec[0] = 0; rec[1] = 0; rec[2] = 0; rec[3] = 0; rec[4] = 0; rec[5] = bestB; rec[6] = bhB; rec[7] = bvB;
	}
}

// 4-bit trace of sw_affine_align_pk_kernel -> SeqAn's trace value of the cell in band column d
__device__ __forceinline__ uint32_t aff_trace_from_nibble(uint32_t nib, int d, int CP) {
	const uint32_t took = nib >> 2;
	if (took == 0u) return 0u;
	uint32_t tg = 0;
	if (d > 0) tg |= (nib & 1u) ? (uint32_t) kTHoriOpen : (uint32_t) kTHori;
	if (d < CP - 1) tg |= (nib & 2u) ? (uint32_t) kTVertOpen : (uint32_t) kTVert;
	return took == 1u ? (tg | (uint32_t) kTDiag) : (tg | (took == 2u ? (uint32_t) kTMaxH : (uint32_t) kTMaxV));
}

#ifdef NGM_ENGINE_KERNELS
// SeqAn's single-trace, gaps-left traceback (dp_traceback_impl.h:184-470) over the stored trace bytes.
// Emits the same compact runs as the linear traceback: (len << 2) | op, op 1 = M, 2 = I, 3 = D, in traceback order.
// With `packed` (the batch the DP ran on) the kernel also counts matching / mismatching diagonal columns
// (rec[3] = matches, rec[7] = mismatches: characters compare like symbol classes on NextGenMap's alphabet), so the host
// needs neither the window nor the read to finish NM and identity.
__global__ __launch_bounds__(256) void affine_traceback_kernel(const uint32_t *__restrict__ dirs, int32_t *__restrict__ records,
		uint16_t *__restrict__ runs, int n, int q, int CP, int run_stride, const uint32_t *__restrict__ packed, int RW, int FW, int nibbles) {
	const int pair = blockIdx.x * blockDim.x + threadIdx.x;
	if (pair >= n) return;
	int32_t *rec = records + (size_t) pair * 8;
	int h = rec[6], v = rec[7];
	const int flags = rec[3];
	const int DW = nibbles ? aff_nib_words(CP) : aff_dir_words(CP);  // nibbles: the trace of sw_affine_align_pk_kernel
	const uint32_t *dp = dirs + (size_t) (pair >> 6) * q * DW * kSlots + (pair & 63);
	uint16_t *out = runs + (size_t) pair * run_stride;
	auto tv_at = [&](int hh, int vv) -> uint32_t {
		if (vv <= 0 || hh <= 0) return 0u;  // initialisation row / column: NONE
		const int d = hh - vv;
		if (d < 0 || d >= CP) return 0u;
		if (nibbles) return aff_trace_from_nibble((dp[((size_t) (vv - 1) * DW + (d >> 3)) * kSlots] >> (4 * (d & 7))) & 15u, d, CP);
		const uint32_t w = dp[((size_t) (vv - 1) * DW + (d >> 2)) * kSlots];
		return (w >> (8 * (d & 3))) & 0xFFu;
	};
	// symbol class of read base v-1 / window base h-1 of this pair in the packed batch (nibble 2k = base k, 2k+1 = base k+4)
	const uint32_t *pk = packed ? packed + (size_t) (pair >> 6) * (RW + FW) * kSlots + (pair & 63) : nullptr;
	auto cls_at = [&](int word0, int idx) -> uint32_t {
		const uint32_t x = pk[(size_t) (word0 + (idx >> 3)) * kSlots];
		const int j = idx & 7;
		return (x >> (4 * ((j & 3) * 2 + (j >> 2)))) & 15u;
	};
	int n_match = 0, n_mis = 0;
	uint32_t tv = tv_at(h, v);
	// _correctTraceValue (dp_algorithm_impl.h:1233-1250)
	if (flags & 1) tv = (tv & ~(uint32_t) kTDiag) | (uint32_t) kTMaxV;
	else if (flags & 2) tv = (tv & ~(uint32_t) kTDiag) | (uint32_t) kTMaxH;
	// _retrieveInitialTraceDirection, PreferGapsAtEnd (dp_traceback_impl.h:452-468)
	if (tv & kTMaxV) tv &= (uint32_t) (kTVert | kTVertOpen | kTMaxV);
	else if (tv & kTMaxH) tv &= (uint32_t) (kTHori | kTHoriOpen | kTMaxH);
	int nruns = 0, cur = -1, curlen = 0;
	auto emit = [&](int op) {
		if (op == cur) { ++curlen; return; }
		if (cur >= 0 && nruns < run_stride) out[nruns++] = (uint16_t) ((curlen << 2) | cur);
		cur = op; curlen = 1;
	};
	while (h > 0 && v > 0 && tv != 0u) {
		if (tv & kTDiag) {
			emit(1);
			if (pk) {
				const uint32_t rc = cls_at(0, v - 1), fc = cls_at(RW, h - 1);
				if (rc == fc && rc <= 5u && rc != 4u) ++n_match; else ++n_mis;
			}
			--h; --v; tv = tv_at(h, v);
		}
		else if ((tv & kTMaxV) && (tv & kTVert)) {
			while ((!(tv & kTVertOpen) || (tv & kTVert)) && v != 1) { emit(2); --v; tv = tv_at(h, v); }
			emit(2); --v; tv = tv_at(h, v);
		} else if ((tv & kTMaxV) && (tv & kTVertOpen)) { emit(2); --v; tv = tv_at(h, v); }
		else if ((tv & kTMaxH) && (tv & kTHori)) {
			while ((!(tv & kTHoriOpen) || (tv & kTHori)) && h != 1) { emit(3); --h; tv = tv_at(h, v); }
			emit(3); --h; tv = tv_at(h, v);
		} else if ((tv & kTMaxH) && (tv & kTHoriOpen)) { emit(3); --h; tv = tv_at(h, v); }
		else break;
	}
	if (cur >= 0 && nruns < run_stride) out[nruns++] = (uint16_t) ((curlen << 2) | cur);
	rec[0] = 1;        // an alignment (possibly empty) always exists in the reference
	rec[1] = h;        // PositionOffset: window offset of the first aligned reference base
	rec[2] = v;        // QStart
	rec[3] = n_match;
	rec[4] = nruns;
	rec[7] = n_mis;
}
#endif  // NGM_ENGINE_KERNELS

}  // namespace ngm
