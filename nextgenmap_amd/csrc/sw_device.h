// sw_device.h -- gfx950 device code of the banded Smith-Waterman score path.
//
// Replaces NextGenMap's OpenCL kernels interleaveSeq + oclSW / oclSW_Global
// (lib/mason/opencl/opencl/oclSwScore.cl:194-213, :332-377, oclEndFreeScore.cl:153-203) with a
// different decomposition, written for CDNA4:
//
//   pack kernel   ASCII pairs (the buffers IAlignment::BatchScore receives, flattened) are loaded
//                 with 16-byte coalesced reads, staged through LDS, translated to the 7 symbol
//                 classes of the reference (oclDefines.cl:64-80) and written as 4-bit codes,
//                 64-way interleaved: dword m of pair slot s of a block sits at [m][s], so the DP
//                 kernel's per-lane streams are perfectly coalesced (one 256-byte line per wave load).
//   score kernel  one pair per lane, the whole band row H[0..C) lives in VGPRs (C is a template
//                 parameter, like the reference's -D corridor_length), integer DP, no LDS traffic
//                 in the inner loop except one 8-byte per-row lookup.  The substitution score is a
//                 v_perm_b32 byte-table lookup: per read row an 8-byte table indexed by the reference
//                 symbol class gives the score of 4 band cells per instruction, which is what makes
//                 the 7-class alphabet (N, NUL padding, 'x' filler) free.
//
// Arithmetic is the reference's, re-based per row so that the table holds non-negative bytes:
//   H'(i,d) = H(i,d) - (i+1)*mismatch         (mismatch < 0, so H' >= H)
//   diag' = H'(i-1,d) + t,  t = s - mismatch in {0, -mismatch, match-mismatch}
//   up'   = H'(i-1,d+1) + (gap_read - mismatch),  left' = H'(i,d-1) + gap_ref,  floor' = -(i+1)*mismatch
// All values are exact integers (int32); results are bit-identical to the reference's short / float
// arithmetic inside its own overflow-free domain.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ngm {

constexpr int kSlots = 64;          // pairs per packed block == lanes per wave
constexpr int kShortMin = -16000;   // oclDefines.cl:28

// number of 4-byte selector registers that cover the C + 7 window bytes one 8-row group touches,
// rounded up to even (one packed dword = 8 bases = 2 registers)
__host__ __device__ constexpr int sel_regs(int C) { return (((C + 7 + 3) / 4) + 1) & ~1; }
__host__ __device__ constexpr int read_words(int q) { return (q + 7) / 8; }
__host__ __device__ constexpr int ref_words(int q, int C) { return read_words(q) + sel_regs(C) / 2; }

struct SwConst {
	int tM;        // match - mismatch : table byte of a match
	int tZ;        // -mismatch        : table byte of a zero score; also the per-row floor step
	int gl;        // gap_ref
	int gu;        // gap_read - mismatch
	int gap_read;  // raw
	int variant;   // NGM_VARIANT_*
	// strand-specific tables of the reference's __ALT_SCORING__ builds (oclDefines.cl:94-128): 0 off, 1 bisulfite, 2 SLAM-seq.
	// The pair's table (FWD / REV, the kernels' `direction` argument) travels in bit 3 of its packed READ classes, so the DP
	// kernels just index a 16-row table with the class nibble.
	int alt;
	int tMA;       // matchALT - mismatch
	int tXA;       // mismatchALT - mismatch
};

// Symbol class of a byte. oclDefines.cl:64-80: A/a 0, C/c 1, G/g 2, T/t 3, N/n 5, NUL 6, other 4.
__device__ __forceinline__ uint32_t sym_class(uint32_t ch) {
	const uint32_t u = ch & 0xDFu;  // fold case
	uint32_t k = 4;
	k = (u == 'A') ? 0u : k;
	k = (u == 'C') ? 1u : k;
	k = (u == 'G') ? 2u : k;
	k = (u == 'T') ? 3u : k;
	k = (u == 'N') ? 5u : k;
	k = (ch == 0) ? 6u : k;
	return k;
}

// Row table for read class nibble rc16 = class | table << 3: byte fc = t(rc, fc) = score(rc, fc) - mismatch.
// oclDefines.cl:85-91 ("scores"), :94-109 (bisulfite), :113-128 (SLAM-seq).
__device__ __forceinline__ uint2 make_row_table(int rc16, const SwConst &K) {
	const int rc = rc16 & 7, rev = (rc16 >> 3) & 1;
	uint32_t b[8];
	for (int fc = 0; fc < 8; ++fc) {
		uint32_t t;
		if (rc >= 6) t = K.tZ;                                  // read NUL (and the unused code 7): 0
		else if (rc == 5) t = (fc <= 3) ? K.tZ : 0;              // read N: 0 vs ACGT, mismatch otherwise
		else if (fc == 6) t = K.tZ;                              // ref NUL: 0
		else if (rc == 4) t = 0;                                 // read "other": mismatch
		else t = (fc == rc) ? K.tM : 0;                          // read ACGT
		if (K.alt == 1) {          // bisulfite: read T vs C / T (FWD), read A vs A / G (REV); the REV tables' read-N row is all zero
			if (!rev) { if (rc == 3 && fc == 1) t = K.tXA; if (rc == 3 && fc == 3) t = K.tMA; }
			else { if (rc == 0 && fc == 0) t = K.tMA; if (rc == 0 && fc == 2) t = K.tXA; if (rc == 5) t = K.tZ; }
		} else if (K.alt == 2) {   // SLAM-seq: read C vs T, read T vs T (FWD); read A vs A, read G vs A (REV)
			if (!rev) { if (rc == 1 && fc == 3) t = K.tXA; if (rc == 3 && fc == 3) t = K.tMA; }
			else { if (rc == 0 && fc == 0) t = K.tMA; if (rc == 2 && fc == 0) t = K.tXA; if (rc == 5) t = K.tZ; }
		}
		b[fc] = t & 0xFF;
	}
	uint2 r;
	r.x = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24);
	r.y = b[4] | (b[5] << 8) | (b[6] << 16) | (b[7] << 24);
	return r;
}

// 8 class codes -> one packed dword, nibble 2k = base k, nibble 2k+1 = base k+4, so that
// (x & 0x0F0F0F0F) and ((x >> 4) & 0x0F0F0F0F) are the byte-selector registers of bases 0-3 / 4-7.
__device__ __forceinline__ uint32_t pack8(const uint32_t k[8]) {
	return k[0] | (k[4] << 4) | (k[1] << 8) | (k[5] << 12) | (k[2] << 16) | (k[6] << 20) | (k[3] << 24) | (k[7] << 28);
}

#ifdef NGM_ENGINE_KERNELS  // non-template kernels are emitted by exactly one TU (ngm_hip.cpp)
// ---------------------------------------------------------------------------------------------
// pack kernel: one workgroup (256 threads) per block of 64 pairs.
//   ref  : n rows of rl = q + c bytes (flat), qry : n rows of q bytes (flat)
//   out  : block b at b * (RW + FW) * 64 dwords; read dword m at [m][slot], ref dword m at [RW + m][slot]
//   lens : per pair, number of read chars before the first NUL (what the align kernels loop over)
//   blk_rows : per block, max over its pairs of (index of last non-NUL read byte + 1)
// dynamic LDS: 64 * (rl + q) bytes
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_pairs_kernel(const uint8_t *__restrict__ ref,
		const uint8_t *__restrict__ qry, int n, int q, int rl, int RW, int FW,
		uint32_t *__restrict__ out, uint16_t *__restrict__ lens, uint16_t *__restrict__ blk_rows, int cstr_ref, const uint8_t *__restrict__ dirs) {
	extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
	__shared__ int s_first_nul[kSlots];
	__shared__ int s_rows[kSlots];
	__shared__ int s_ref_nul[kSlots];
	__shared__ int s_blk_rows;
	__shared__ uint8_t s_trans[256];
	uint8_t *lref = lds;
	uint8_t *lqry = lds + (size_t) kSlots * rl;
	const int tid = threadIdx.x;
	const int blk = blockIdx.x;
	const int p0 = blk * kSlots;
	const int np = min(kSlots, n - p0);
	if (tid < kSlots) { s_first_nul[tid] = q; s_rows[tid] = 0; s_ref_nul[tid] = rl; }
	if (tid == 0) s_blk_rows = 0;
	// the affine personality compares characters (SeqAn): lower case never equals NextGenMap's upper-case alphabet
	s_trans[tid] = (uint8_t) ((cstr_ref && tid >= 'a' && tid <= 'z') ? 4u : sym_class((uint32_t) tid));

	// phase 1: coalesced copy of the block's rows into LDS (same linear layout), zero fill the tail
	{
		const size_t gref = (size_t) p0 * rl, gqry = (size_t) p0 * q;
		const int nref = np * rl, nqry = np * q;
		const int tref = kSlots * rl, tqry = kSlots * q;
		const bool al16 = (((uintptr_t) (ref + gref) | (uintptr_t) (qry + gqry)) & 15) == 0;
		if (al16) {
			for (int o = tid * 16; o < tref; o += 256 * 16) {
				uint4 v = make_uint4(0, 0, 0, 0);
				if (o + 16 <= nref) v = *reinterpret_cast<const uint4 *>(ref + gref + o);
				*reinterpret_cast<uint4 *>(lref + o) = v;
				if (o < nref && o + 16 > nref) for (int k = 0; o + k < nref; ++k) lref[o + k] = ref[gref + o + k];
			}
			for (int o = tid * 16; o < tqry; o += 256 * 16) {
				uint4 v = make_uint4(0, 0, 0, 0);
				if (o + 16 <= nqry) v = *reinterpret_cast<const uint4 *>(qry + gqry + o);
				*reinterpret_cast<uint4 *>(lqry + o) = v;
				if (o < nqry && o + 16 > nqry) for (int k = 0; o + k < nqry; ++k) lqry[o + k] = qry[gqry + o + k];
			}
		} else {
			for (int o = tid; o < tref; o += 256) lref[o] = (o < nref) ? ref[gref + o] : 0;
			for (int o = tid; o < tqry; o += 256) lqry[o] = (o < nqry) ? qry[gqry + o] : 0;
		}
	}
	__syncthreads();

	// phase 2: lane = pair slot, four waves split the dwords; stores are 256-byte coalesced
	const int slot = tid & 63, part = tid >> 6;
	if (cstr_ref) {
		// the affine personality reads the window as a C string (TSequence(refSeqList[i]), EndToEndAffine.cpp:16):
		// nothing after the first NUL exists
		const uint8_t *row = lref + slot * rl;
		int fn = rl;
		for (int i = part; i < rl; i += 4) if (row[i] == 0) { fn = i; break; }
		atomicMin(&s_ref_nul[slot], fn);
		__syncthreads();
	}
	uint32_t *ob = out + (size_t) blk * (RW + FW) * kSlots + slot;
	const uint32_t dbit = (dirs && p0 + slot < n && dirs[p0 + slot]) ? 8u : 0u;   // the pair's score table (SwConst::alt)
	{
		const uint8_t *row = lqry + slot * q;
		int first_nul = q, rows = 0;
		for (int m = part; m < RW; m += 4) {
			uint32_t k[8];
#pragma unroll
			for (int j = 0; j < 8; ++j) {
				const int i = m * 8 + j;
				const uint32_t ch = (i < q) ? row[i] : 0u;
				k[j] = s_trans[ch] | dbit;
				if (ch == 0) first_nul = min(first_nul, i); else rows = max(rows, i + 1);
			}
			ob[(size_t) m * kSlots] = pack8(k);
		}
		first_nul = min(first_nul, q);
		atomicMin(&s_first_nul[slot], first_nul);
		atomicMax(&s_rows[slot], rows);
	}
	{
		const uint8_t *row = lref + slot * rl;
		for (int m = part; m < FW; m += 4) {
			uint32_t k[8];
#pragma unroll
			for (int j = 0; j < 8; ++j) {
				const int i = m * 8 + j;
				k[j] = s_trans[(i < s_ref_nul[slot]) ? row[i] : 0u];
			}
			ob[(size_t) (RW + m) * kSlots] = pack8(k);
		}
	}
	__syncthreads();
	if (tid < kSlots) {
		atomicMax(&s_blk_rows, s_rows[tid]);
		if (p0 + tid < n) lens[p0 + tid] = (uint16_t) s_first_nul[tid];
	}
	__syncthreads();
	if (tid == 0) blk_rows[blk] = (uint16_t) s_blk_rows;
}

#endif  // NGM_ENGINE_KERNELS

// ---------------------------------------------------------------------------------------------
// score kernel: BatchScore.  One pair per lane, one packed block per wave, 4 waves per workgroup.
// ---------------------------------------------------------------------------------------------
template <int C, bool ENDFREE>
__global__ __launch_bounds__(256) void sw_score_kernel(const uint32_t *__restrict__ packed,
		const uint16_t *__restrict__ lens, const uint16_t *__restrict__ blk_rows,
		float *__restrict__ scores, int n, int n_blocks, int RW, SwConst K) {
	__shared__ uint2 s_tab[16];
	if (threadIdx.x < 16) s_tab[threadIdx.x] = make_row_table(threadIdx.x, K);
	__syncthreads();
	const int lane = threadIdx.x & 63;
	const int blk = blockIdx.x * 4 + (threadIdx.x >> 6);
	if (blk >= n_blocks) return;
	constexpr int NRG = sel_regs(C);
	const int FW = RW + NRG / 2;
	const uint32_t *rd = packed + (size_t) blk * (RW + FW) * kSlots + lane;
	const uint32_t *fd = rd + (size_t) RW * kSlots;
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
	int fl = K.tZ;       // floor' of row 0
	int best = 0;        // virtual NUL rows contribute 0 (the reference starts at -1: see epilogue)
	uint32_t rnext = (ngroups > 0) ? rd[0] : 0x66666666u;

	for (int g = 0; g < ngroups; ++g) {
		const uint32_t rx = rnext;
		// prefetch next group's read and reference dwords (coalesced 256 B per wave)
		rnext = (g + 1 < ngroups) ? rd[(size_t) (g + 1) * kSlots] : 0x66666666u;
		const uint32_t fx = fd[(size_t) (g + NRG / 2) * kSlots];
		const uint32_t rsel[2] = {rx & 0x0F0F0F0Fu, (rx >> 4) & 0x0F0F0F0Fu};
#pragma unroll
		for (int s = 0; s < 8; ++s) {
			const uint32_t rc = (rsel[s >> 2] >> (8 * (s & 3))) & 0xFFu;
			const uint2 T = s_tab[rc];
			// scores of 4 band cells per perm: selector bytes are reference classes 0..6
			constexpr int R0 = 0;
			uint32_t P[NRG];
#pragma unroll
			for (int r = R0; r < NRG; ++r) P[r] = ((s + C - 1) / 4 >= r && s / 4 <= r) ? __builtin_amdgcn_perm(T.y, T.x, RG[r]) : 0u;
			int left = ENDFREE ? (fl + kShortMin) : fl;
			int rowmax = fl;
#pragma unroll
			for (int d = 0; d < C; ++d) {
				const int bi = s + d;
				const int t = (int) ((P[bi >> 2] >> (8 * (bi & 3))) & 0xFFu);
				const int dg = H[d] + t;
				const int a = left + K.gl;
				int h;
				if (d < C - 1) {
					const int b = H[d + 1] + K.gu;
					h = max(max(a, b), dg);
				} else if (ENDFREE) {
					const int b = fl + kShortMin + K.gap_read;  // sentinel column, oclEndFreeScore.cl:170
					h = max(max(a, b), dg);
				} else {
					h = max(a, dg);  // the 0 sentinel + gap_read never beats the floor
				}
				if (!ENDFREE) { h = max(h, fl); rowmax = max(rowmax, h); }
				H[d] = h;
				left = h;
			}
			if (!ENDFREE) best = max(best, rowmax - fl);
			fl += K.tZ;
		}
		// slide the selector window by 8 bytes
#pragma unroll
		for (int r = 0; r + 2 < NRG; ++r) RG[r] = RG[r + 2];
		RG[NRG - 2] = fx & 0x0F0F0F0Fu;
		RG[NRG - 1] = (fx >> 4) & 0x0F0F0F0Fu;
	}

	const int pair = blk * kSlots + lane;
	if (pair < n) {
		int result;
		if (ENDFREE) {
			// max over the last row and the sentinel slot (oclEndFreeScore.cl:197-201); H = H' + K_last
			const int klast = -(fl - K.tZ);
			int m = kShortMin;
#pragma unroll
			for (int d = 0; d < C; ++d) m = max(m, H[d] + klast);
			result = m;
		} else {
			result = best;
		}
		// the float4 (__CPU__) build never enters the DP for an empty read (oclSwScore.cl:124,
		// oclEndFreeScore.cl:20); the __GPU__ build does and reports 0
		if (K.variant == 1 && lens[pair] == 0) result = ENDFREE ? kShortMin : -1;
		scores[pair] = (float) result;
	}
}

// ---------------------------------------------------------------------------------------------
// score kernel, packed 16-bit: TWO pairs per lane (blocks 2w and 2w+1 of the packed batch), every band value
// is a 16-bit half of one VGPR and every DP operation a v_pk_add_i16 / v_pk_max_i16 on both pairs at once.
// Same recurrences and re-basing as sw_score_kernel; valid while the re-based values fit 16 bits
// (rows * (match - mismatch) < 2^15, checked by the host, which otherwise uses the 32-bit kernel).
// The substitution bytes of the two pairs come from two v_perm lookups and are merged into the two halves by a third.
// ---------------------------------------------------------------------------------------------
typedef short v2s __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v2s pk_splat(int x) { v2s r; r.x = (short) x; r.y = (short) x; return r; }
__device__ __forceinline__ v2s pk_max(v2s a, v2s b) { return __builtin_elementwise_max(a, b); }
typedef unsigned short v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2s pk_min(v2s a, v2s b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ v2s pk_min_op(v2s a, v2s b) {  // (opaque to the optimizer, see took4 below)
	v2s r;
	asm("v_pk_min_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
	return r;
}
__device__ __forceinline__ v2s pk_mul(v2s a, v2s b) {
	v2s r;
	asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
	return r;
}
__device__ __forceinline__ v2s pk_mad(v2s a, v2s b, v2s c) {  // a * b + c per 16-bit half, one instruction
	v2s r;
	asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
	return r;
}
__device__ __forceinline__ v2s pk_mad_k(v2s a, v2s b, int k) {  // a * b + k, k a uniform value (kept in an SGPR: no VGPR per constant)
	v2s r;
	const uint32_t kk = ((uint32_t) k & 0xFFFFu) * 0x10001u;
	asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(kk));
	return r;
}
__device__ __forceinline__ v2s pk_min1_op(v2s a) {  // min(a, 1), signed, opaque to the optimizer
	v2s r;
	asm("v_pk_min_i16 %0, %1, 1 op_sel_hi:[1,0]" : "=v"(r) : "v"(a));
	return r;
}
__device__ __forceinline__ v2s pk_min1_u_op(v2s a) {  // min(a, 1), unsigned
	v2s r;
	asm("v_pk_min_u16 %0, %1, 1 op_sel_hi:[1,0]" : "=v"(r) : "v"(a));
	return r;
}
__device__ __forceinline__ v2s pk_sub_sat(v2s a, v2s b) { return __builtin_elementwise_sub_sat(a, b); }  // v_pk_sub_i16 ... clamp
__device__ __forceinline__ v2s pk_min_u_op(v2s a, v2s b) {  // unsigned, opaque
	v2s r;
	asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
	return r;
}

template <int C, bool ENDFREE>
__global__ __launch_bounds__(256) void sw_score_pk_kernel(const uint32_t *__restrict__ packed,
		const uint16_t *__restrict__ lens, const uint16_t *__restrict__ blk_rows,
		float *__restrict__ scores, int n, int n_blocks, int RW, SwConst K) {
	__shared__ uint2 s_tab[16];
	if (threadIdx.x < 16) s_tab[threadIdx.x] = make_row_table(threadIdx.x, K);
	__syncthreads();
	const int lane = threadIdx.x & 63;
	const int blkA = 2 * (blockIdx.x * 4 + (threadIdx.x >> 6));
	if (blkA >= n_blocks) return;
	const bool hasB = blkA + 1 < n_blocks;
	const int blkB = hasB ? blkA + 1 : blkA;
	constexpr int NRG = sel_regs(C);
	const int FW = RW + NRG / 2;
	const uint32_t *rdA = packed + (size_t) blkA * (RW + FW) * kSlots + lane, *rdB = packed + (size_t) blkB * (RW + FW) * kSlots + lane;
	const uint32_t *fdA = rdA + (size_t) RW * kSlots, *fdB = rdB + (size_t) RW * kSlots;
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
	v2s best = pk_splat(0);
	const v2s gl2 = pk_splat(K.gl), gu2 = pk_splat(K.gu);
	uint32_t rnA = (ngroups > 0) ? rdA[0] : 0x66666666u, rnB = (ngroups > 0) ? rdB[0] : 0x66666666u;

	for (int g = 0; g < ngroups; ++g) {
		const uint32_t rxA = rnA, rxB = rnB;
		rnA = (g + 1 < ngroups) ? rdA[(size_t) (g + 1) * kSlots] : 0x66666666u;
		rnB = (g + 1 < ngroups) ? rdB[(size_t) (g + 1) * kSlots] : 0x66666666u;
		const uint32_t fxA = fdA[(size_t) (g + NRG / 2) * kSlots], fxB = fdB[(size_t) (g + NRG / 2) * kSlots];
		const uint32_t rsA[2] = {rxA & 0x0F0F0F0Fu, (rxA >> 4) & 0x0F0F0F0Fu}, rsB[2] = {rxB & 0x0F0F0F0Fu, (rxB >> 4) & 0x0F0F0F0Fu};
#pragma unroll
		for (int s = 0; s < 8; ++s) {
			const uint2 TA = s_tab[(rsA[s >> 2] >> (8 * (s & 3))) & 0xFFu], TB = s_tab[(rsB[s >> 2] >> (8 * (s & 3))) & 0xFFu];
			uint32_t PA[NRG], PB[NRG];
#pragma unroll
			for (int r = 0; r < NRG; ++r) {
				const bool used = (s + C - 1) / 4 >= r && s / 4 <= r;
				PA[r] = used ? __builtin_amdgcn_perm(TA.y, TA.x, RGA[r]) : 0u;
				PB[r] = used ? __builtin_amdgcn_perm(TB.y, TB.x, RGB[r]) : 0u;
			}
			const v2s fl2 = pk_splat(fl);
			v2s left = ENDFREE ? pk_splat(fl + kShortMin) : fl2;
			v2s rowmax = fl2;
#pragma unroll
			for (int d = 0; d < C; ++d) {
				const int bi = s + d, kb = bi & 3;
				// byte kb of PA -> low half, byte kb of PB -> high half, zero-extended
				const uint32_t sel = 0x0C000C00u | (uint32_t) kb | ((uint32_t) (4 + kb) << 16);
				const uint32_t tt = __builtin_amdgcn_perm(PB[bi >> 2], PA[bi >> 2], sel);
				const v2s t = __builtin_bit_cast(v2s, tt);
				const v2s dg = H[d] + t;
				const v2s a = left + gl2;
				v2s h;
				if (d < C - 1) h = pk_max(pk_max(a, H[d + 1] + gu2), dg);
				else if (ENDFREE) h = pk_max(pk_max(a, pk_splat(fl + kShortMin + K.gap_read)), dg);
				else h = pk_max(a, dg);
				if (!ENDFREE) { h = pk_max(h, fl2); rowmax = pk_max(rowmax, h); }
				H[d] = h;
				left = h;
			}
			if (!ENDFREE) best = pk_max(best, rowmax - fl2);
			fl += K.tZ;
		}
#pragma unroll
		for (int r = 0; r + 2 < NRG; ++r) { RGA[r] = RGA[r + 2]; RGB[r] = RGB[r + 2]; }
		RGA[NRG - 2] = fxA & 0x0F0F0F0Fu; RGA[NRG - 1] = (fxA >> 4) & 0x0F0F0F0Fu;
		RGB[NRG - 2] = fxB & 0x0F0F0F0Fu; RGB[NRG - 1] = (fxB >> 4) & 0x0F0F0F0Fu;
	}

	int resA, resB;
	if (ENDFREE) {
		const v2s klast = pk_splat(-(fl - K.tZ));
		v2s mx = pk_splat(kShortMin);
#pragma unroll
		for (int d = 0; d < C; ++d) mx = pk_max(mx, H[d] + klast);
		resA = mx.x; resB = mx.y;
	} else {
		resA = best.x; resB = best.y;
	}
	const int pairA = blkA * kSlots + lane, pairB = blkB * kSlots + lane;
	if (pairA < n) {
		if (K.variant == 1 && lens[pairA] == 0) resA = ENDFREE ? kShortMin : -1;
		scores[pairA] = (float) resA;
	}
	if (hasB && pairB < n) {
		if (K.variant == 1 && lens[pairB] == 0) resB = ENDFREE ? kShortMin : -1;
		scores[pairB] = (float) resB;
	}
}

}  // namespace ngm
