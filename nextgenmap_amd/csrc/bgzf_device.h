// bgzf_device.h -- BGZF blocks (RFC 1951 DEFLATE in RFC 1952 members with the BC extra field) written by the GPU (round 4).
//
// `ngm --bam` hands its records to bamtools' BgzfStream (lib/bamtools-2.3.0/src/api/internal/io/BgzfStream_p.cpp: 64 KB blocks, zlib
// level Z_DEFAULT_COMPRESSION); what a reader of the file sees is the decompressed record stream, which tests/test_gpu_bam.py compares
// with the reference's.  zlib level 6 costs ~25 core-ms per block, and the GPU boxes of this pool give a container 16 CPUs: 2 M reads/s
// however the work is spread.  BGZF blocks are independent DEFLATE streams of <= 0xFF00 input bytes -- one workgroup
// per block, the block in LDS:
//   matching   the block is cut into 8 segments of 8 192 bytes, one wave each.  A wave walks its segment 64 positions at a time:
//              every lane hashes the 4 bytes at its position, reads the most recent earlier position with that hash from the wave's
//              table (ds_max keeps the largest position: deterministic), extends that candidate and the distance-1 candidate (runs)
//              by 4-byte compares; then the wave parses the 64 positions greedily, with one step of lazy evaluation, in a scalar loop
//              over v_readlane -- bit masks of the positions that start a literal / a match, the matches (position, length, distance)
//              appended to a list in global memory;
//   CRC-32     128 bytes per thread, combined with x^(8 m) mod P from a table (the gzip trailer's CRC);
//   Huffman    histograms by LDS atomics; code lengths by rank sort (parallel) + the two-queue merge and zlib's overflow repair (one
//              thread; 286 symbols); canonical codes in parallel; the header carries all 286 + 30 lengths without run-length codes
//              (~150 bytes per block);
//   bits       every thread adds up the bits of the tokens that start in its 128 positions, a scan gives its bit offset, it writes its
//              bits (whole words plain, the two shared boundary words by atomic OR) into the LDS image of the member;
//   member     header (BSIZE), the DEFLATE bytes, CRC-32, ISIZE -> global memory, 64 KB stride; a block that does not shrink is stored.
// Ratio against zlib level 6 on BAM records: see DESIGN.md 5 (hash of 4 bytes, one candidate + runs, 8 KB windows).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace ngm {
namespace bgzf {

constexpr int kIn = 0xFF00;        // input bytes per block
constexpr int kNT = 512;
constexpr int kSegs = 8;           // waves = segments of a block
constexpr int kSeg = 8192;         // bytes per wave (128 steps of 64; the last segment is shorter)
constexpr int kChunk = 128;        // positions whose tokens one thread turns into bits (510 threads have some)
constexpr int kStride = 65536;     // bytes between the members of consecutive blocks in the strided output
constexpr int kMatCap = kSeg / 4 + 8;
constexpr int kMaskWords = kIn / 32;   // 2040
constexpr int kHdrBits = 18 * 8;   // the DEFLATE stream starts behind the 18-byte member header

struct Args {
	const uint8_t *raw;
	unsigned long long n;
	int n_blocks;
	uint8_t *out;                 // [n_blocks * kStride]
	uint32_t *sizes;              // [n_blocks]
	uint2 *scratch;               // [gridDim.x * kSegs * kMatCap]
	const uint32_t *crc_table;    // [4][256]: slicing-by-4 tables (table k: the byte followed by k zero bytes)
	const uint32_t *xpow;         // [kIn + 1]: x^(8 m) mod P, reflected
	const uint8_t *len_code;      // [256]: length - 3 -> length symbol - 257
	const uint8_t *dist_code;     // [512]: zlib's d_code table (distance - 1 < 256: [d], else [256 + (d >> 7)])
	unsigned long long *phase_cycles;   // diagnostics (NGM_HIP_BGZF_PHASES): [8] cycles of thread 0 per phase, summed over the blocks
};

__device__ __constant__ const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__device__ __constant__ const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__device__ __constant__ const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__device__ __constant__ const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__device__ __constant__ const uint8_t kPreOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

__device__ __forceinline__ uint32_t load32u(const uint32_t *w, uint32_t at) {   // four bytes at any byte address of an LDS array
	const uint32_t lo = w[at >> 2], hi = w[(at >> 2) + 1];
	return (uint32_t) ((((unsigned long long) hi << 32) | lo) >> (8u * (at & 3u)));
}
__device__ __forceinline__ uint32_t crc_mulmod(uint32_t a, uint32_t b) {   // a * b mod P, reflected (bit 31 = x^0)
	uint32_t p = 0;
#pragma unroll 4
	for (int i = 0; i < 32; ++i) {
		if (a & (0x80000000u >> i)) p ^= b;
		b = (b >> 1) ^ ((b & 1u) ? 0xedb88320u : 0u);
	}
	return p;
}

// Code lengths of a Huffman code limited to max_bits.  All threads call it; freq[n] (LDS) -> len[n] (LDS).
// tmp: 2 * n + 2 * n words of LDS (sorted symbols, weights / depths, parents) + 20 words.
__device__ inline void build_lengths(const uint32_t *freq, int n, int max_bits, uint32_t *len, uint32_t *tmp, int tid) {
	uint32_t *ssym = tmp, *w = tmp + n, *parent = tmp + 3 * n, *misc = tmp + 5 * n;   // w, parent: 2 n each
	if (tid == 0) misc[0] = 0;
	for (int s = tid; s < n; s += kNT) len[s] = 0;
	__syncthreads();
	// rank sort, ascending by (frequency, symbol), symbols in use only
	for (int s = tid; s < n; s += kNT) {
		const uint32_t f = freq[s];
		if (f == 0) continue;
		int rank = 0;
		for (int t = 0; t < n; ++t) { const uint32_t g = freq[t]; rank += (g != 0 && (g < f || (g == f && t < s))) ? 1 : 0; }
		ssym[rank] = (uint32_t) s;
		w[rank] = f;
		atomicAdd(&misc[0], 1u);
	}
	__syncthreads();
	const int m = (int) misc[0];
	uint32_t *bl = misc + 1;   // [max_bits + 2], then the number of leaves below max_bits
	if (tid <= max_bits + 2) bl[tid] = 0;
	if (m == 1) { if (tid == 0) len[ssym[0]] = 1; __syncthreads(); return; }
	if (m == 0) { __syncthreads(); return; }
	if (tid == 0) {
		// two queues: leaves [0, m) ascending, internal nodes [m, 2 m - 1) in the order they are made (ascending too); the heads in registers
		int i = 0, j = m;
		uint32_t wi = w[0], wj = 0xFFFFFFFFu;
		for (int k = m; k < 2 * m - 1; ++k) {
			uint32_t sum = 0;
#pragma unroll
			for (int r = 0; r < 2; ++r) {
				if (i < m && (j >= k || wi <= wj)) { sum += wi; parent[i] = (uint32_t) k; ++i; wi = i < m ? w[i] : 0xFFFFFFFFu; }
				else { sum += wj; parent[j] = (uint32_t) k; ++j; wj = j < k ? w[j] : 0xFFFFFFFFu; }
			}
			w[k] = sum;
			if (j == k) wj = sum;
		}
	}
	__syncthreads();
	// depth of every leaf: up the parents to the root (2 m - 2), all leaves at once
	for (int k = tid; k < m; k += kNT) {
		int d = 0;
		for (uint32_t x = (uint32_t) k; x != (uint32_t) (2 * m - 2); x = parent[x]) ++d;
		if (d > max_bits) { d = max_bits; atomicAdd(&bl[max_bits + 2], 1u); }
		atomicAdd(&bl[d], 1u);
	}
	__syncthreads();
	if (tid == 0) {
		int overflow = (int) bl[max_bits + 2];
		while (overflow > 0) {   // zlib's gen_bitlen: move one leaf down from the deepest level that has one, two of the overflowing leaves take its place
			int bits = max_bits - 1;
			while (bl[bits] == 0) --bits;
			bl[bits] -= 1; bl[bits + 1] += 2; bl[max_bits] -= 1;
			overflow -= 2;
		}
	}
	__syncthreads();
	for (int k = tid; k < m; k += kNT) {   // the rarest symbols get the longest codes
		uint32_t c = 0;
		for (int bits = max_bits; bits >= 1; --bits) { c += bl[bits]; if ((uint32_t) k < c) { len[ssym[k]] = (uint32_t) bits; break; } }
	}
	__syncthreads();
}

// canonical codes (RFC 1951 3.2.2), bit-reversed for an LSB-first stream: code[s] = reversed code | length << 16.  All threads.
__device__ inline void build_codes(const uint32_t *len, int n, uint32_t *code, uint32_t *tmp, int tid) {
	uint32_t *bl = tmp, *next = tmp + 16;
	if (tid < 16) bl[tid] = 0;
	__syncthreads();
	for (int s = tid; s < n; s += kNT) if (len[s]) atomicAdd(&bl[len[s]], 1u);
	__syncthreads();
	if (tid == 0) {
		uint32_t c = 0;
		next[0] = 0;
		for (int b = 1; b <= 15; ++b) { c = (c + bl[b - 1]) << 1; next[b] = c; }
	}
	__syncthreads();
	for (int s = tid; s < n; s += kNT) {
		const uint32_t l = len[s];
		uint32_t v = 0;
		if (l) {
			uint32_t before = 0;
			for (int t = 0; t < s; ++t) before += len[t] == l ? 1u : 0u;
			v = (__brev(next[l] + before) >> (32 - l)) | (l << 16);
		}
		code[s] = v;
	}
	__syncthreads();
}

// bits into the LDS image of the member: whole words of a thread's own range by plain stores, its first and last word by atomic OR
struct BitOut {
	uint32_t *out;
	unsigned long long acc = 0;
	uint32_t word, first;
	int fill;
	__device__ BitOut(uint32_t *o, uint32_t bit) : out(o), word(bit >> 5), first(bit >> 5), fill((int) (bit & 31u)) {}
	__device__ __forceinline__ void put(uint32_t v, int nb) {
		acc |= (unsigned long long) v << fill;
		fill += nb;
		if (fill >= 32) {
			if (word == first) atomicOr(&out[word], (uint32_t) acc); else out[word] = (uint32_t) acc;
			acc >>= 32; fill -= 32; ++word;
		}
	}
	__device__ __forceinline__ void finish() { if (fill > 0 && (uint32_t) acc != 0u) atomicOr(&out[word], (uint32_t) acc); }
};

__global__ __launch_bounds__(kNT) void deflate_kernel(Args A) {
	extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
	uint32_t *in = lds;                              // 65 536 bytes: the block + zero padding
	uint32_t *U = in + 16384;                        // 65 536 bytes: the eight hash tables, later the image of the member
	uint32_t *lit_mask = U + 16384;                  // [2048]
	uint32_t *mat_mask = lit_mask + 2048;            // [2048]
	uint32_t *hist_ll = mat_mask + 2048;             // [288]
	uint32_t *hist_d = hist_ll + 288;                // [32]
	uint32_t *len_ll = hist_d + 32;                  // [288]
	uint32_t *len_d = len_ll + 288;                  // [32]
	uint32_t *code_ll = len_d + 32;                  // [288]
	uint32_t *code_d = code_ll + 288;                // [32]
	uint32_t *hist_p = code_d + 32;                  // [32]
	uint32_t *len_p = hist_p + 32;                   // [32]
	uint32_t *code_p = len_p + 32;                   // [32]
	uint32_t *crc_t = code_p + 32;                   // [4][256]
	uint32_t *tmp = crc_t + 1024;                    // [5 * 288 + 64]
	uint32_t *sh = tmp + 5 * 288 + 64;               // [64] scalars: nmat[4], scans, CRC, bit counts
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const uint8_t *inb = (const uint8_t *) in;
	for (int j = tid; j < 1024; j += kNT) crc_t[j] = A.crc_table[j];
	unsigned long long t_prev = 0;
	auto phase = [&](int k) { if (A.phase_cycles && tid == 0) { const unsigned long long t = wall_clock64(); if (k >= 0) atomicAdd(&A.phase_cycles[k], t - t_prev); t_prev = t; } };
	for (int blk = blockIdx.x; blk < A.n_blocks; blk += gridDim.x) {
		phase(-1);
		const unsigned long long at = (unsigned long long) blk * kIn;
		const int len = (int) (A.n - at < (unsigned long long) kIn ? A.n - at : (unsigned long long) kIn);
		__syncthreads();   // (the previous block's image has been copied out)
		// ---- the block into LDS; tables, masks and histograms cleared -------------------------------------------------
		{
			const uint8_t *src = A.raw + at;
			const bool aligned = ((uintptr_t) src & 3u) == 0;
			for (int wd = tid; wd < 16384; wd += kNT) {
				const int b = wd * 4;
				uint32_t v = 0;
				if (b + 4 <= len && aligned) v = *(const uint32_t *) (src + b);
				else for (int j = 0; j < 4; ++j) if (b + j < len) v |= (uint32_t) src[b + j] << (8 * j);
				in[wd] = v;
				U[wd] = 0;
			}
			for (int wd = tid; wd < 4096; wd += kNT) lit_mask[wd] = 0;   // (both masks)
			for (int s = tid; s < 288 + 32; s += kNT) hist_ll[s] = 0;     // (and hist_d)
			if (tid < 32) hist_p[tid] = 0;
			if (tid < 64) sh[tid] = 0;
		}
		__syncthreads();
		phase(0);
		// ---- matching: one wave per segment ---------------------------------------------------------------------------
		{
			const int s0 = wv * kSeg, s1 = min(len, s0 + kSeg);
			uint32_t *tab = U + wv * 2048;
			uint2 *mlist = A.scratch + ((size_t) blockIdx.x * kSegs + wv) * kMatCap;
			int cur = s0, nm = 0;
			for (int base = s0; base < s1; base += 64) {
				const int i = base + lane;
				uint32_t mylen = 0, mydist = 0;
				const bool can = i + 4 <= s1;
				uint32_t cand = 0, v4 = 0;
				if (can) {
					v4 = load32u(in, (uint32_t) i);
					const uint32_t h = (v4 * 2654435761u) >> 21;
					cand = tab[h];
					atomicMax(&tab[h], (uint32_t) (i - s0 + 1));
				}
				// the byte in front of this lane's four: the previous lane has it (lane 0: one read)
				uint32_t before = (uint32_t) __shfl_up((int) v4, 1) & 255u;
				if (lane == 0 && i > s0) before = inb[i - 1];
				if (can && i >= cur) {
					const int maxl = min(258, s1 - i);
					// both candidates are checked on the four bytes this lane holds before any compare loop runs: most candidates end there
					auto extend = [&](int p) {   // in[p .. p + 3] == in[i .. i + 3] is known
						int l = 4;
						while (l < maxl) {
							const uint32_t x = load32u(in, (uint32_t) (p + l)) ^ load32u(in, (uint32_t) (i + l));
							if (x) { l += (__ffs((int) x) - 1) >> 3; break; }
							l += 4;
						}
						return min(l, maxl);
					};
					if (cand) {
						const int p = s0 + (int) cand - 1;
						if (load32u(in, (uint32_t) p) == v4) { mylen = (uint32_t) extend(p); mydist = (uint32_t) (i - p); }
					}
					if (i > s0 && v4 == before * 0x01010101u) {   // in[i - 1 .. i + 3] are one byte value: a run of at least four at distance 1
						const int l = extend(i - 1);
						if ((uint32_t) l >= mylen) { mylen = (uint32_t) l; mydist = 1; }
					}
				}
				// greedy parse of these 64 positions, one step of lazy evaluation -- wave-uniform: the lanes with a match as a bit mask, the
				// literals between two matches as a range of bits, v_readlane only where a match is taken or weighed
				unsigned long long litb = 0, matb = 0;
				const int nstep = min(64, s1 - base);
				const unsigned long long mm = __ballot(mylen >= 4u);
				auto below = [](int b) -> unsigned long long { return b >= 64 ? ~0ull : ((1ull << b) - 1ull); };
				while (cur < base + nstep) {
					const int k = __builtin_amdgcn_readfirstlane(cur - base);
					const unsigned long long rest = mm & ~below(k);
					if (rest == 0ull) { litb |= below(nstep) & ~below(k); cur = base + nstep; break; }
					const int k2 = __builtin_amdgcn_readfirstlane((int) __ffsll((long long) rest) - 1);
					litb |= below(k2) & ~below(k);
					const uint32_t L = (uint32_t) __builtin_amdgcn_readlane((int) mylen, k2);
					if (k2 + 1 < nstep && ((mm >> (k2 + 1)) & 1ull)) {
						const uint32_t L1 = (uint32_t) __builtin_amdgcn_readlane((int) mylen, k2 + 1);
						if (L1 > L) { litb |= 1ull << k2; cur = base + k2 + 1; continue; }
					}
					matb |= 1ull << k2;
					cur = base + k2 + (int) L;
				}
				if ((matb >> lane) & 1ull) {
					const int idx = nm + __popcll(matb & ((1ull << lane) - 1ull));
					mlist[idx] = make_uint2((uint32_t) i | ((mylen - 3u) << 16), mydist);
				}
				nm += __popcll(matb);
				if (lane == 0) {
					lit_mask[base >> 5] = (uint32_t) litb; lit_mask[(base >> 5) + 1] = (uint32_t) (litb >> 32);
					mat_mask[base >> 5] = (uint32_t) matb; mat_mask[(base >> 5) + 1] = (uint32_t) (matb >> 32);
				}
			}
			if (lane == 0) sh[wv] = (uint32_t) nm;
		}
		__threadfence_block();
		__syncthreads();
		phase(1);
		// ---- image cleared; CRC-32; histograms -------------------------------------------------------------------------
		for (int wd = tid; wd < 16384; wd += kNT) U[wd] = 0;
		{
			const int b0 = tid * kChunk, b1 = min(len, b0 + kChunk);
			uint32_t c = 0;
			// (a full chunk comes from global memory, 16 bytes at a time: in LDS the chunks of the 64 lanes start 128 bytes apart -- the same bank)
			const uint8_t *gsrc = A.raw + at + b0;
			if (b1 - b0 == kChunk && ((uintptr_t) gsrc & 15u) == 0) {
				const uint4 *g = (const uint4 *) gsrc;
#pragma unroll
				for (int q4 = 0; q4 < kChunk / 16; ++q4) {
					const uint4 v = g[q4];
					const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
					for (int e = 0; e < 4; ++e) {
						c ^= w4[e];
						c = crc_t[768 + (c & 255u)] ^ crc_t[512 + ((c >> 8) & 255u)] ^ crc_t[256 + ((c >> 16) & 255u)] ^ crc_t[c >> 24];
					}
				}
			} else for (int b = b0; b < b1; ++b) c = crc_t[(c ^ inb[b]) & 255u] ^ (c >> 8);
			uint32_t part = (b1 > b0 && c) ? crc_mulmod(c, A.xpow[len - b1]) : 0u;
			if (tid == kNT - 1) part ^= crc_mulmod(0xFFFFFFFFu, A.xpow[len]) ^ 0xFFFFFFFFu;
			for (int o = 32; o > 0; o >>= 1) part ^= (uint32_t) __shfl_xor((int) part, o);
			if (lane == 0) atomicXor(&sh[8], part);
			// literals of my positions, then my share of the matches
			if (tid < kIn / kChunk) for (int j = 0; j < kChunk / 32; ++j) {
				uint32_t m = lit_mask[tid * (kChunk / 32) + j];
				while (m) { const int bit = __ffs((int) m) - 1; m &= m - 1; atomicAdd(&hist_ll[inb[b0 + j * 32 + bit]], 1u); }
			}
			for (int s = 0; s < kSegs; ++s) {
				const uint2 *mlist = A.scratch + ((size_t) blockIdx.x * kSegs + s) * kMatCap;
				const int nm = (int) sh[s];
				for (int m = tid; m < nm; m += kNT) {
					const uint2 e = mlist[m];
					atomicAdd(&hist_ll[257 + A.len_code[(e.x >> 16) & 255u]], 1u);
					const uint32_t d = e.y - 1u;
					atomicAdd(&hist_d[d < 256u ? A.dist_code[d] : A.dist_code[256u + (d >> 7)]], 1u);
				}
			}
			if (tid == 0) atomicAdd(&hist_ll[256], 1u);
		}
		__syncthreads();
		phase(2);
		// ---- Huffman codes ----------------------------------------------------------------------------------------------
		build_lengths(hist_ll, 286, 15, len_ll, tmp, tid);
		build_lengths(hist_d, 30, 15, len_d, tmp, tid);
		for (int s = tid; s < 286 + 30; s += kNT) atomicAdd(&hist_p[s < 286 ? len_ll[s] : len_d[s - 286]], 1u);
		__syncthreads();
		build_lengths(hist_p, 19, 7, len_p, tmp, tid);
		build_codes(len_ll, 286, code_ll, tmp, tid);
		build_codes(len_d, 30, code_d, tmp, tid);
		build_codes(len_p, 19, code_p, tmp, tid);
		phase(3);
		// ---- block header (thread 0) and the bits of every thread's tokens --------------------------------------------
		// ---- block header: the fixed part by thread 0, the 286 + 30 code lengths one per thread ----------------------------
		{
			uint32_t hc = 0;   // this thread's code length symbol in the code length code: code | bits << 16
			if (tid < 286 + 30) hc = code_p[tid < 286 ? len_ll[tid] : len_d[tid - 286]];
			const uint32_t nb = hc >> 16;
			uint32_t hscan = nb;
			for (int o = 1; o < 64; o <<= 1) { const uint32_t v = (uint32_t) __shfl_up((int) hscan, o); if (lane >= o) hscan += v; }
			if (lane == 63) sh[32 + wv] = hscan;
			__syncthreads();
			uint32_t at = 3 + 14 + 57 + hscan - nb, all = 3 + 14 + 57;
			for (int w2 = 0; w2 < kSegs; ++w2) { if (w2 < wv) at += sh[32 + w2]; all += sh[32 + w2]; }
			if (tid == 0) {
				BitOut o(U, kHdrBits);
				o.put(1u, 1); o.put(2u, 2); o.put(286 - 257, 5); o.put(30 - 1, 5); o.put(19 - 4, 4);
				for (int j = 0; j < 19; ++j) o.put(len_p[kPreOrder[j]], 3);
				o.finish();
				sh[9] = all;
			}
			if (nb) { BitOut o(U, kHdrBits + at); o.put(hc & 0xFFFFu, (int) nb); o.finish(); }
		}
		// tokens that start in positions [kChunk tid, kChunk (tid + 1)): walk(emit) -> bits
		constexpr int kCW = kChunk / 32;
		uint32_t my_words_l[kCW], my_words_m[kCW];
		uint32_t nmatch_mine = 0;
		if (tid < kIn / kChunk) {
#pragma unroll
			for (int j = 0; j < kCW; ++j) { my_words_l[j] = lit_mask[tid * kCW + j]; my_words_m[j] = mat_mask[tid * kCW + j]; nmatch_mine += (uint32_t) __popc(my_words_m[j]); }
		}
		// matches before my positions (block-wide exclusive scan of the per-thread counts)
		uint32_t mscan = nmatch_mine;
		for (int o = 1; o < 64; o <<= 1) { const uint32_t v = (uint32_t) __shfl_up((int) mscan, o); if (lane >= o) mscan += v; }
		if (lane == 63) sh[16 + wv] = mscan;
		__syncthreads();
		uint32_t mbefore = mscan - nmatch_mine;
		for (int w2 = 0; w2 < wv; ++w2) mbefore += sh[16 + w2];
		auto walk = [&](auto emit) -> uint32_t {
			uint32_t bits = 0;
			if (tid >= kIn / kChunk) return 0u;
			const int p0 = tid * kChunk;
			const int seg = p0 / kSeg;   // (a thread's positions lie in one segment)
			uint32_t seg_first = 0;   // matches in the segments before `seg`
			for (int s = 0; s < seg; ++s) seg_first += sh[s];
			uint32_t midx = mbefore - seg_first;
			const uint2 *mlist = A.scratch + ((size_t) blockIdx.x * kSegs + seg) * kMatCap;
#pragma unroll 1
			for (int j = 0; j < kCW; ++j) {
				uint32_t lm = my_words_l[j], mm = my_words_m[j], both = lm | mm;
				while (both) {
					const int bit = __ffs((int) both) - 1;
					both &= both - 1;
					const int pos = p0 + j * 32 + bit;
					if ((lm >> bit) & 1u) {
						const uint32_t c = code_ll[inb[pos]];
						bits += c >> 16;
						emit(c & 0xFFFFu, (int) (c >> 16));
					} else {
						const uint2 e = mlist[midx++];
						const uint32_t l3 = (e.x >> 16) & 255u, lc = A.len_code[l3];
						const uint32_t c = code_ll[257 + lc];
						const uint32_t le = kLenExtra[lc];
						emit((c & 0xFFFFu) | ((l3 + 3u - kLenBase[lc]) << (c >> 16)), (int) (c >> 16) + (int) le);
						const uint32_t d = e.y - 1u;
						const uint32_t dc = d < 256u ? A.dist_code[d] : A.dist_code[256u + (d >> 7)];
						const uint32_t cd = code_d[dc];
						const uint32_t de = kDistExtra[dc];
						emit((cd & 0xFFFFu) | ((e.y - kDistBase[dc]) << (cd >> 16)), (int) (cd >> 16) + (int) de);
						bits += (c >> 16) + le + (cd >> 16) + de;
					}
				}
			}
			return bits;
		};
		phase(4);
		const uint32_t my_bits = walk([](uint32_t, int) {});
		uint32_t bscan = my_bits;
		for (int o = 1; o < 64; o <<= 1) { const uint32_t v = (uint32_t) __shfl_up((int) bscan, o); if (lane >= o) bscan += v; }
		if (lane == 63) sh[24 + wv] = bscan;
		__syncthreads();
		uint32_t bit0 = kHdrBits + sh[9] + bscan - my_bits;
		uint32_t total_bits = sh[9];
		for (int w2 = 0; w2 < kSegs; ++w2) { if (w2 < wv) bit0 += sh[24 + w2]; total_bits += sh[24 + w2]; }
		const uint32_t eob = code_ll[256];
		total_bits += eob >> 16;
		const uint32_t clen = (total_bits + 7u) >> 3;
		const bool stored = clen >= (uint32_t) len + 5u || clen + 26u > (uint32_t) kStride;
		uint8_t *dst = A.out + (size_t) blk * kStride;
		const uint32_t crc = sh[8];
		phase(5);
		if (!stored) {
			{
				BitOut o(U, bit0);
				(void) walk([&](uint32_t v, int nb) { o.put(v, nb); });
				if (tid == kNT - 1) o.put(eob & 0xFFFFu, (int) (eob >> 16));   // (the last thread has no positions: bit0 = the end of the tokens)
				o.finish();
			}
			const uint32_t size = 18u + clen + 8u;
			if (tid == 0) {
				atomicOr(&U[0], 0x04088b1fu); atomicOr(&U[2], 0x0006ff00u); atomicOr(&U[3], 0x00024342u); atomicOr(&U[4], (size - 1u) & 0xFFFFu);
				A.sizes[blk] = size;
			}
			__syncthreads();
			if (tid == 0) {
				const uint32_t tb = 18u + clen;
				for (int j = 0; j < 8; ++j) {
					const uint32_t byte = j < 4 ? (crc >> (8 * j)) & 255u : ((uint32_t) len >> (8 * (j - 4))) & 255u;
					const uint32_t a = tb + (uint32_t) j;
					atomicOr(&U[a >> 2], byte << (8u * (a & 3u)));
				}
			}
			__syncthreads();
			for (uint32_t wd = tid; wd < (size + 3u) / 4u; wd += kNT) ((uint32_t *) dst)[wd] = U[wd];
		} else {
			// stored block: 01, LEN, NLEN, the bytes
			const uint32_t size = 18u + 5u + (uint32_t) len + 8u;
			if (tid == 0) {
				const uint8_t head[23] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, (uint8_t) ((size - 1u) & 255u), (uint8_t) ((size - 1u) >> 8),
						1, (uint8_t) (len & 255), (uint8_t) (len >> 8), (uint8_t) (~len & 255), (uint8_t) ((~len >> 8) & 255)};
				for (int j = 0; j < 23; ++j) dst[j] = head[j];
				for (int j = 0; j < 8; ++j) dst[23 + len + j] = (uint8_t) (j < 4 ? (crc >> (8 * j)) & 255u : ((uint32_t) len >> (8 * (j - 4))) & 255u);
				A.sizes[blk] = size;
			}
			for (int b = tid; b < len; b += kNT) dst[23 + b] = inb[b];
		}
		phase(6);
	}
}

// the members of the blocks, dense: member b -> dense[offsets[b], offsets[b] + sizes[b])
__global__ __launch_bounds__(256) void gather_kernel(const uint8_t *strided, const uint32_t *sizes, const unsigned long long *offsets, uint8_t *dense, int n_blocks) {
	for (int blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
		const uint8_t *src = strided + (size_t) blk * kStride;
		uint8_t *dst = dense + offsets[blk];
		const uint32_t n = sizes[blk];
		// (the destination starts at any byte: leading bytes up to a word boundary, words, trailing bytes)
		const uint32_t lead = (uint32_t) ((4u - ((uintptr_t) dst & 3u)) & 3u);
		if (threadIdx.x < lead && threadIdx.x < n) dst[threadIdx.x] = src[threadIdx.x];
		if (n > lead) {
			const uint32_t words = (n - lead) / 4u;
			for (uint32_t wd = threadIdx.x; wd < words; wd += 256) {
				const uint32_t a = lead + 4u * wd;
				const uint32_t lo = ((const uint32_t *) src)[a >> 2], hi = ((const uint32_t *) src)[(a >> 2) + 1];
				((uint32_t *) (dst + lead))[wd] = (uint32_t) ((((unsigned long long) hi << 32) | lo) >> (8u * (a & 3u)));
			}
			for (uint32_t b = lead + 4u * words + threadIdx.x; b < n; b += 256) dst[b] = src[b];
		}
	}
}

inline size_t deflate_lds_bytes() { return (size_t) (16384 + 16384 + 4096 + 2 * (288 + 32) + (288 + 32) + 3 * 32 + 1024 + 5 * 288 + 64 + 64) * 4; }

}  // namespace bgzf
}  // namespace ngm
