// gather_device.h -- device-side gather of (read, reference window) pairs into the packed layout the DP
// kernels consume, straight from the HBM-resident genome and read batch.  Replaces the host loops of
// ScoreBuffer::DoRun / AlignmentBuffer::DoRun (src/ScoreBuffer.cpp:87-122, src/AlignmentBuffer.cpp:73-111):
// MappedRead::computeReverseSeq (src/MappedRead.cpp:53-68) and _SequenceProvider::DecodeRefSequence
// (src/SequenceProvider.cpp:382-441), including its window-geometry quirks (odd offsets yield one extra base,
// an odd decode length turns the last base into 'x', 'x' fill past the genome end, NUL tail).
// Also holds the top-1 selection / MAPQ kernel (ScoreBuffer::top1SE + computeMQ, src/ScoreBuffer.cpp:34-49, :228-277).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sw_device.h"

namespace ngm {

struct WindowGeom {
	uint64_t concat_len;   // _SequenceProvider::GetConcatRefLen
	int buffer_len;        // refMaxLen the host would allocate: ((q+c)|1)+1 for scores, (q+c)|2 for alignments
	int half_corridor;     // corridor >> 1
};

// class (A0 C1 G2 T3 x4 N5 NUL6) of window byte j for a window starting at genome position `offset`
__device__ __forceinline__ uint32_t window_class(const uint32_t *__restrict__ genome, const WindowGeom &G, uint64_t offset, int j) {
	uint64_t len = (uint64_t) G.buffer_len - 2;
	if (offset >= G.concat_len) return 5u;            // decode failed: the caller fills the row with 'N' (ScoreBuffer.cpp:113-118)
	uint64_t end = 0;
	if (offset + len > G.concat_len) { end = offset + len - G.concat_len; len -= end; }
	const uint64_t emitted = ((offset & 1) ? 1 : 0) + 2 * ((len + 1) / 2);
	const uint64_t jj = (uint64_t) j;
	if (jj < emitted) {
		if ((len & 1) && jj == emitted - 1) return 4u;  // "buffer[codedIndex - 1] = 'x'"
		const uint64_t pos = offset + jj;
		return (genome[pos >> 3] >> (4 * (pos & 7))) & 15u;
	}
	if (jj < emitted + end) return 4u;                 // 'x' fill past the genome end
	return 6u;                                         // zero fill
}

__device__ __forceinline__ uint32_t read_class_fwd(uint32_t ch) {
	uint32_t k = 5;  // reads only hold A C G T N (IParser.h:59-121); anything else is treated as N
	k = (ch == 'A') ? 0u : k;
	k = (ch == 'C') ? 1u : k;
	k = (ch == 'G') ? 2u : k;
	k = (ch == 'T') ? 3u : k;
	k = (ch == 0) ? 6u : k;
	return k;
}

// One workgroup (256 threads) per block of 64 pairs; pair p = (pair_read[p], pair_loc[p], pair_strand[p]).
__global__ __launch_bounds__(256) void gather_pairs_kernel(const uint8_t *__restrict__ reads, const uint16_t *__restrict__ read_len,
		int q, const uint32_t *__restrict__ genome, WindowGeom G, const uint32_t *__restrict__ pair_read,
		const uint32_t *__restrict__ pair_loc, const uint32_t *__restrict__ pair_sv, int n_pairs, int RW, int FW,
		uint32_t *__restrict__ out, uint16_t *__restrict__ lens, uint16_t *__restrict__ blk_rows, int alt_dir = 0) {
	__shared__ int s_rows;
	const int tid = threadIdx.x;
	const int blk = blockIdx.x;
	const int slot = tid & 63, part = tid >> 6;
	const int pair = blk * kSlots + slot;
	if (tid == 0) s_rows = 0;
	__syncthreads();
	uint32_t *ob = out + (size_t) blk * (RW + FW) * kSlots + slot;
	const bool live = pair < n_pairs;
	const uint32_t ridx = live ? pair_read[pair] : 0u;
	const bool rev = live ? (pair_sv[pair] & 1u) : false;
	const int L = live ? (int) read_len[ridx] : 0;
	const uint8_t *rp = reads + (size_t) ridx * q;
	// bisulfite / SLAM-seq: the pair's score table (the m_DirBuffer of src/ScoreBuffer.cpp:93-110, src/AlignmentBuffer.cpp:80-98) rides
	// in bit 3 of the read classes (SwConst::alt).  alt_dir: 0 off, 1 single-end (reverse strand -> 1), 2 paired (second mates flip)
	uint32_t dbit = 0;
	if (alt_dir && live) { const bool second = alt_dir == 2 && (ridx & 1u); dbit = (rev ? !second : second) ? 8u : 0u; }
	struct __attribute__((packed, aligned(1))) U32 { uint32_t v; };
	for (int m = part; m < RW; m += 4) {
		uint32_t k[8];
		// the 8 read characters of this word: two (unaligned) dword loads when they lie inside the row
		const int first = rev ? L - 8 * m - 8 : 8 * m;  // row offset of the lowest-addressed of the 8 characters
		uint32_t lo4 = 0, hi4 = 0;
		const bool fast = live && first >= 0 && first + 8 <= q;
		if (fast) { lo4 = reinterpret_cast<const U32 *>(rp + first)->v; hi4 = reinterpret_cast<const U32 *>(rp + first + 4)->v; }
#pragma unroll
		for (int j = 0; j < 8; ++j) {
			const int i = m * 8 + j;
			uint32_t c = 6u;
			if (live && i < L) {
				uint32_t ch;
				if (fast) { const int b = rev ? 7 - j : j; ch = ((b < 4 ? lo4 : hi4) >> (8 * (b & 3))) & 0xFFu; }
				else ch = rev ? rp[L - 1 - i] : rp[i];
				c = read_class_fwd(ch);
				if (rev) c = (c <= 3u) ? 3u - c : c;  // reverse complement: A<->T, C<->G, N stays (MappedRead.cpp:32-43, :53-68)
			}
			k[j] = c | dbit;
		}
		ob[(size_t) m * kSlots] = pack8(k);
	}
	const uint64_t offset = live ? (uint64_t) pair_loc[pair] - (uint64_t) G.half_corridor : 0;
	// bases [0, plain) of the window are plain genome nibbles (window_class: before the odd-length 'x' and the fill past
	// the genome end): words that lie entirely inside are cut out of two genome dwords with a funnel shift
	uint64_t plain = 0;
	if (live && offset < G.concat_len) {
		uint64_t len = (uint64_t) G.buffer_len - 2, end = 0;
		if (offset + len > G.concat_len) { end = offset + len - G.concat_len; len -= end; }
		const uint64_t emitted = ((offset & 1) ? 1 : 0) + 2 * ((len + 1) / 2);
		plain = (len & 1) ? emitted - 1 : emitted;
	}
	for (int m = part; m < FW; m += 4) {
		uint32_t word;
		if ((uint64_t) (m * 8 + 8) <= plain) {
			const uint64_t pos = offset + (uint64_t) m * 8;
			const uint32_t w0 = genome[pos >> 3], w1 = genome[(pos >> 3) + 1];
			const uint32_t r = 4u * (uint32_t) (pos & 7);
			const uint32_t v = r ? ((w0 >> r) | (w1 << (32u - r))) : w0;  // nibble j = base j
			uint32_t lo = v & 0xFFFFu, hi = v >> 16;                       // pack8: nibble 2j = base j, 2j+1 = base j+4
			lo = (lo | (lo << 8)) & 0x00FF00FFu; lo = (lo | (lo << 4)) & 0x0F0F0F0Fu;
			hi = (hi | (hi << 8)) & 0x00FF00FFu; hi = (hi | (hi << 4)) & 0x0F0F0F0Fu;
			word = lo | (hi << 4);
		} else {
			uint32_t k[8];
#pragma unroll
			for (int j = 0; j < 8; ++j) k[j] = live ? window_class(genome, G, offset, m * 8 + j) : 6u;
			word = pack8(k);
		}
		ob[(size_t) (RW + m) * kSlots] = word;
	}
	if (part == 0) {
		if (live) lens[pair] = (uint16_t) L;
		atomicMax(&s_rows, L);
	}
	__syncthreads();
	if (tid == 0) blk_rows[blk] = (uint16_t) s_rows;
}

// ScoreBuffer::top1SE + computeMQ for every read (src/ScoreBuffer.cpp:228-277, :34-49).
// winner: pair index of the best candidate (ties: smallest location, forward strand first), or 0xFFFFFFFF.
// The reference walks a read's scores in order with (best, second, number of best ones); what it ends with does not depend on the order:
//   M = the largest score.  M > 0: best = M, n_best = the scores equal to M, second = M when there are two of them, else the largest score
//   below M or 0 if that is larger (scores <= 0 never pass `s > second`); the winner is the smallest (location << 1 | strand) among the best.
//   M == 0: best = 0, n_best = the zeros, the winner the smallest key among them, MAPQ 0.  M < 0: n_best = 0 and the reference submits the
//   first candidate -- here, as before, the smallest key of all.
// A workgroup of 256 threads owns 256 consecutive reads: a thread settles its own read when it has at most kSelectSmall candidates (on a
// genome without a heavy tail: all of them), the others go to a list in LDS and are taken a WAVE at a time (coalesced loads of scores /
// pair_loc / pair_sv, one butterfly reduction per read).  Round 5's kernel walked every read with one thread: on the GRCh38-like genome
// (candidates per read: median 1, 99th percentile 1 356, maximum 9 383) the launch took as long as its longest read -- 5.2 ms per 131 072 reads.
constexpr uint32_t kSelectSmall = 8;

struct Top1State {   // of a set of candidates: its largest score, how many have it, the largest score below it, the smallest key among the best (and its index)
	float best, below;
	int num;
	uint64_t key;
	uint32_t idx;
};
__device__ __forceinline__ Top1State top1_empty() { return Top1State{-INFINITY, -INFINITY, 0, ~0ull, 0xFFFFFFFFu}; }
__device__ __forceinline__ void top1_add(Top1State &S, float s, uint64_t key, uint32_t j) {
	if (s > S.best) { S.below = S.best; S.best = s; S.num = 1; S.key = key; S.idx = j; }
	else if (s == S.best) { ++S.num; if (key < S.key || (key == S.key && j < S.idx)) { S.key = key; S.idx = j; } }
	else if (s > S.below) S.below = s;
}
__device__ __forceinline__ Top1State top1_merge(const Top1State &a, const Top1State &b) {
	if (a.best > b.best) { Top1State r = a; r.below = fmaxf(a.below, b.best); return r; }
	if (b.best > a.best) { Top1State r = b; r.below = fmaxf(b.below, a.best); return r; }
	Top1State r = a;
	r.num = a.num + b.num; r.below = fmaxf(a.below, b.below);
	if (b.key < a.key || (b.key == a.key && b.idx < a.idx)) { r.key = b.key; r.idx = b.idx; }
	return r;
}
__device__ __forceinline__ void top1_store(const Top1State &S, uint32_t all_idx, int r, const float *__restrict__ scores,
		uint32_t *__restrict__ winner, int32_t *__restrict__ mapq, int32_t *__restrict__ n_best, float *__restrict__ best_score) {
	const float M = S.best;
	uint32_t bi = S.idx;
	int num = S.num;
	float best = 0.0f, second = 0.0f;
	if (M > 0.0f) { best = M; second = num >= 2 ? M : fmaxf(S.below, 0.0f); }
	else if (M < 0.0f) { num = 0; bi = all_idx; }
	int mq = 0;
	if (best > 0 && second >= 0) mq = (int) ceilf(60.0f * (best - second) / best);  // ScoreBuffer.cpp:34-40
	winner[r] = bi;
	mapq[r] = mq;
	n_best[r] = num;
	best_score[r] = best > 0 ? best : scores[bi];
}

__global__ __launch_bounds__(256) void select_top1_kernel(int n_reads, const uint32_t *__restrict__ cand_base, const uint32_t *__restrict__ cand_count,
		const float *__restrict__ scores, const uint32_t *__restrict__ pair_loc, const uint32_t *__restrict__ pair_sv,
		uint32_t *__restrict__ winner, int32_t *__restrict__ mapq, int32_t *__restrict__ n_best, float *__restrict__ best_score) {
	__shared__ uint32_t s_big[256], s_nbig;
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	if (tid == 0) s_nbig = 0;
	__syncthreads();
	const int r = blockIdx.x * 256 + tid;
	if (r < n_reads) {
		const uint32_t b = cand_base[r], n = cand_count[r];
		if (n == 0) { winner[r] = 0xFFFFFFFFu; mapq[r] = 0; n_best[r] = 0; best_score[r] = 0.f; }
		else if (n <= kSelectSmall) {
			Top1State S = top1_empty();
			uint64_t akey = ~0ull;
			uint32_t aidx = b;
			for (uint32_t j = b; j < b + n; ++j) {
				const uint64_t key = ((uint64_t) pair_loc[j] << 1) | (pair_sv[j] & 1u);
				top1_add(S, scores[j], key, j);
				if (key < akey) { akey = key; aidx = j; }
			}
			top1_store(S, aidx, r, scores, winner, mapq, n_best, best_score);
		} else s_big[atomicAdd(&s_nbig, 1u)] = (uint32_t) tid;
	}
	__syncthreads();
	const uint32_t nbig = s_nbig;
	for (uint32_t i = (uint32_t) wv; i < nbig; i += 4u) {
		const int rr = blockIdx.x * 256 + (int) s_big[i];
		const uint32_t b = cand_base[rr], n = cand_count[rr];
		Top1State S = top1_empty();
		for (uint32_t j = b + (uint32_t) lane; j < b + n; j += 64u) top1_add(S, scores[j], ((uint64_t) pair_loc[j] << 1) | (pair_sv[j] & 1u), j);
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) {
			Top1State o;
			o.best = __shfl_xor(S.best, d); o.below = __shfl_xor(S.below, d); o.num = __shfl_xor(S.num, d);
			o.key = ((uint64_t) (uint32_t) __shfl_xor((int) (uint32_t) (S.key >> 32), d) << 32) | (uint32_t) __shfl_xor((int) (uint32_t) S.key, d);
			o.idx = (uint32_t) __shfl_xor((int) S.idx, d);
			S = top1_merge(S, o);
		}
		uint32_t aidx = S.idx;
		if (S.best < 0.0f) {   // (wave-uniform) no score above or at zero: the smallest key of all
			uint64_t akey = ~0ull;
			aidx = 0xFFFFFFFFu;
			for (uint32_t j = b + (uint32_t) lane; j < b + n; j += 64u) {
				const uint64_t key = ((uint64_t) pair_loc[j] << 1) | (pair_sv[j] & 1u);
				if (key < akey) { akey = key; aidx = j; }
			}
#pragma unroll
			for (int d = 1; d < 64; d <<= 1) {
				const uint64_t ok = ((uint64_t) (uint32_t) __shfl_xor((int) (uint32_t) (akey >> 32), d) << 32) | (uint32_t) __shfl_xor((int) (uint32_t) akey, d);
				const uint32_t oi = (uint32_t) __shfl_xor((int) aidx, d);
				if (ok < akey || (ok == akey && oi < aidx)) { akey = ok; aidx = oi; }
			}
		}
		if (lane == 0) top1_store(S, aidx, rr, scores, winner, mapq, n_best, best_score);
	}
}

// Paired-end selection, the common case on the GPU: both mates have exactly ONE candidate.  ScoreBuffer::top1PE / CheckPairs
// (src/ScoreBuffer.cpp:368-502) then has nothing to choose: the pair is taken when its insert size lies inside the window and
// its score is positive (MAPQ 60 for both mates, no equally good pair), otherwise the single-end selection of
// select_top1_kernel stands and the pair is flagged as failed.
// info[pair] >= 0: bit 0 = pair taken, bits 1.. = its insert size (the host only sums those for the running mean, ScoreBuffer.h:90);
// -1: a mate without candidates (top1SE for the other one: the host's); <= -2: a pair with choices -- entry -2 - info[pair] of
// pair_choice_kernel's output (pair_device.h): the pair is appended to the list of the small pairs (both mates at most 64
// candidates: one wave each there) or, beyond 2^30, to the list of the large ones.
__global__ void pair_simple_kernel(int n_pairs, const uint32_t *__restrict__ cand_base, const uint32_t *__restrict__ cand_count,
		const float *__restrict__ scores, const uint32_t *__restrict__ pair_loc, const uint16_t *__restrict__ read_len, int min_d, int max_d,
		int32_t *__restrict__ mapq, int32_t *__restrict__ n_best, int32_t *__restrict__ info, uint32_t *__restrict__ list_small, uint32_t *__restrict__ list_large,
		uint32_t *__restrict__ list_counts) {
	const int pi = blockIdx.x * blockDim.x + threadIdx.x;
	const bool live = pi < n_pairs;
	const int rb = 2 * pi, ra = 2 * pi + 1;   // `a` = the mate whose scores arrive last in the reference (the odd read id)
	const uint32_t ca = live ? cand_count[ra] : 1u, cb = live ? cand_count[rb] : 1u;
	{
		// the pairs with choices go to the two lists: ONE atomic per wave and list (a counter serves ~86 M returning atomics per second:
		// one per pair made this kernel 0.4 ms per 262 144 pairs)
		const bool choice = live && (ca != 1u || cb != 1u) && ca != 0u && cb != 0u && list_counts != nullptr;
		const bool small = choice && ca <= 64u && cb <= 64u, large = choice && !small;
		const unsigned long long ms = __ballot(small), ml = __ballot(large);
		const int lane = threadIdx.x & 63;
		const unsigned long long below = (1ull << lane) - 1ull;
		uint32_t bs = 0, bl = 0;
		if (ms && lane == (int) __builtin_ctzll(ms)) bs = atomicAdd(&list_counts[0], (uint32_t) __popcll(ms));
		if (ml && lane == (int) __builtin_ctzll(ml)) bl = atomicAdd(&list_counts[1], (uint32_t) __popcll(ml));
		if (ms) bs = (uint32_t) __builtin_amdgcn_readlane((int) bs, (int) __builtin_ctzll(ms));
		if (ml) bl = (uint32_t) __builtin_amdgcn_readlane((int) bl, (int) __builtin_ctzll(ml));
		if (small) { const uint32_t at = bs + (uint32_t) __popcll(ms & below); list_small[at] = (uint32_t) pi; info[pi] = -2 - (int32_t) at; }
		if (large) { const uint32_t at = bl + (uint32_t) __popcll(ml & below); list_large[at] = (uint32_t) pi; info[pi] = -2 - (int32_t) (at + (1u << 30)); }
		if (choice) return;
	}
	if (!live) return;
	if (ca != 1u || cb != 1u) { info[pi] = -1; return; }
	const uint32_t ba = cand_base[ra], bb = cand_base[rb];
	const uint64_t l1 = pair_loc[ba], l2 = pair_loc[bb];
	const int cur = (int) ((l2 > l1) ? l2 - l1 + (uint64_t) read_len[rb] : l1 - l2 + (uint64_t) read_len[ra]);
	const float ps = scores[ba] + scores[bb];
	const bool found = cur > min_d && cur < max_d && ps > 0.0f;
	if (found) { mapq[ra] = 60; mapq[rb] = 60; n_best[ra] = 0; n_best[rb] = 0; }
	info[pi] = found ? ((cur << 1) | 1) : 0;
}

// expands per-read candidate lists into pair arrays: pair j of read r gets pair_read[j] = r
__global__ void expand_pairs_kernel(int n_reads, const uint32_t *__restrict__ cand_base, const uint32_t *__restrict__ cand_count,
		uint32_t *__restrict__ pair_read) {
	const int r = blockIdx.x;
	const uint32_t b = cand_base[r], n = cand_count[r];
	for (uint32_t j = threadIdx.x; j < n; j += blockDim.x) pair_read[b + j] = (uint32_t) r;
}

// winners -> compact alignment batch
__global__ void collect_winners_kernel(int n_reads, const uint32_t *__restrict__ winner, const uint32_t *__restrict__ pair_loc,
		const uint32_t *__restrict__ pair_sv, const uint32_t *__restrict__ slot_of_read, uint32_t *__restrict__ a_read,
		uint32_t *__restrict__ a_loc, uint32_t *__restrict__ a_sv) {
	const int r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= n_reads) return;
	const uint32_t w = winner[r];
	if (w == 0xFFFFFFFFu) return;
	const uint32_t s = slot_of_read[r];
	a_read[s] = (uint32_t) r;
	a_loc[s] = pair_loc[w];
	a_sv[s] = pair_sv[w];
}

// Traceback runs live in a strided scratch (run_stride u16 per pair); alignments have a handful of runs, so
// they are compacted before the download.  rec[6] (the argmax row, no longer needed) receives the offset.
__global__ void compact_runs_kernel(int n, int32_t *__restrict__ records, const uint16_t *__restrict__ runs, int run_stride,
		uint16_t *__restrict__ out, unsigned long long *__restrict__ cursor) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	int32_t *rec = records + (size_t) i * 8;
	const int nr = rec[0] ? rec[4] : 0;
	const unsigned long long off = nr ? atomicAdd(cursor, (unsigned long long) nr) : 0ull;
	rec[6] = (int32_t) off;
	const uint16_t *src = runs + (size_t) i * run_stride;
	for (int k = 0; k < nr; ++k) out[off + k] = src[k];
}

}  // namespace ngm
