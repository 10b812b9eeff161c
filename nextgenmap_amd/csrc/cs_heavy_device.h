// cs_heavy_device.h -- candidate search for the reads the fast path hands on: thousands to tens of thousands of index hits
// (round 4).  Same semantics as every other search kernel: CS::PrefixIteration (src/CSstatic.cpp:26-76), GetRefEntry
// (src/PrefixTable.cpp:750-817), PrefixSearch / AddLocationStd (src/CS.cpp:114-213), CollectResultsStd (src/CS.cpp:263-313).
//
// Why it exists.  On a genome with a GRCh38-like k-mer spectrum (tests/humanlike.py; automatic max. k-mer frequency 1 531 instead
// of 100) a third to a half of the reads carry more hits than the fast path's bit plane and 1 024-slot table take, and the exact
// kernels that used to receive them hold ONE read per CU (a 128 KB table in LDS, one wave) or vote through L2 atomics into
// tables in global memory: 33 ms + 243 ms per 262 144 reads against 3 ms for the fast path
// (profiles/r04_heavy_tail_cs_passes.txt).  A read with H hits needs H votes, not a table of H entries:
//
//   sweep 1  every hit increments a 16-bit counter, hash(bin) -> one of NC counters in LDS (a count-min sketch with one row: a
//            counter is an UPPER bound of the votes -- both strands -- of every bin that maps to it);
//   T        from the histogram of the counter values: the smallest T for which the hits on counters >= T fit the exact table;
//   sweep 2  the hits whose counter is >= T -- all hits of their bins, or none -- go into the exact table (key = bin,
//            value = forward | reverse votes);
//   check    with M2 the largest strand count in the table: a bin outside the table has at most T - 1 votes, so if
//            T - 1 < max(kmer_min, M2 * sensitivity) (in float, as the reference computes its threshold) then M2 is the true
//            maximum, the threshold is final, and every candidate is in the table: exact.  Otherwise (a read with more
//            near-threshold bins than the table holds) the read goes on to the exact kernels as before.
// The position lists are read twice (the second time mostly from L2); the counters cost 2 bytes of LDS each, so a workgroup
// needs 36 KB (reads up to 16 384 hits: four per CU) up to 100 KB (65 535 hits: one per CU, 1 024 threads); reads with more hits, and
// the reads whose near-threshold bins outgrow the table of their class, are taken by a last class with 32-bit counters and 8 192 slots.
#pragma once

#include "cs_device.h"

namespace ngm {

// hits of the block's read, HPL consecutive ones per thread per trip (a wave covers 512 consecutive hits); f(position, list)
template <int NT, typename F>
__device__ __forceinline__ void cs_for_each_hit_block(const uint32_t *__restrict__ positions, const uint32_t *l_start, const uint32_t *l_pref,
		int n_lists, uint32_t H, int tid, F f) {
	constexpr int HPL = kCsHitsPerLane;
	for (uint32_t h0 = (uint32_t) tid * HPL; h0 < H; h0 += (uint32_t) NT * HPL) {
		int lo = 0, hi = n_lists;  // largest li with pref[li] <= h0
		while (hi - lo > 1) {
			const int mid = (lo + hi) >> 1;
			if (l_pref[mid] <= h0) lo = mid; else hi = mid;
		}
		uint32_t pos[HPL];
		int li[HPL];
#pragma unroll
		for (int j = 0; j < HPL; ++j) {
			const uint32_t h = h0 + j;
			li[j] = -1;
			pos[j] = 0;
			if (h < H) {
				while (l_pref[lo + 1] <= h) ++lo;
				li[j] = lo;
				pos[j] = positions[l_start[lo] + (h - l_pref[lo])];
			}
		}
#pragma unroll
		for (int j = 0; j < HPL; ++j) if (li[j] >= 0) f(pos[j], li[j]);
	}
}

constexpr uint32_t kCsHeavyMaxHits16 = 65535u;   // 16-bit counters: no counter can wrap below this many hits

inline size_t cs_heavy_lds_bytes(int lists_cap, int q, int log2_counters, int log2_slots, bool wide) {
	return ((size_t) lists_cap * 2 + 1 + (size_t) (q + 3) / 4 + ((size_t) 1 << (log2_counters - (wide ? 0 : 1))) + 256 + ((size_t) 2 << log2_slots)) * 4;
}

// A.read_list: the reads; A.log2_bits: log2 of the counters; A.log2_slots: log2 of the table slots; reads that cannot be certified
// are appended to A.ovf_read / A.ovf_hits (A.status[1]) for the exact kernels
// WIDE: 32-bit counters (reads of any hit count; twice the LDS per counter)
template <int NT, bool WIDE = false>
__global__ __launch_bounds__(NT) void cs_heavy_kernel(CsArgs A) {
	extern __shared__ __attribute__((aligned(16))) uint32_t cs_lds[];
	__shared__ uint32_t s_T;
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const int read = (int) A.read_list[blockIdx.x];
	const int k = A.k;
	uint32_t *l_start = cs_lds;                                  // [lists_cap]
	uint32_t *l_pref = cs_lds + A.lists_cap;                     // [lists_cap + 1]
	uint8_t *l_code = (uint8_t *) (l_pref + A.lists_cap + 1);    // [q rounded up to 4]
	uint32_t *cnt = (uint32_t *) l_code + (A.q + 3) / 4;         // [NC / 2]: two 16-bit counters per word (WIDE: [NC])
	const int log2c = A.log2_bits;
	const uint32_t n_cnt = 1u << log2c;
	const uint32_t cnt_words = WIDE ? n_cnt : (n_cnt >> 1);
	uint32_t *hist = cnt + cnt_words;                            // [256]: hits on the counters of value c (255: and above)
	uint32_t *t_keys = hist + 256;
	const int log2_slots = A.log2_slots;
	const uint32_t n_slots = 1u << log2_slots;
	uint32_t *t_votes = t_keys + n_slots;
	for (uint32_t s = tid; s < cnt_words; s += NT) cnt[s] = 0;
	for (uint32_t s = tid; s < 256u; s += NT) hist[s] = 0;
	for (uint32_t s = tid; s < n_slots; s += NT) { t_keys[s] = 0xFFFFFFFFu; t_votes[s] = 0; }
	// every wave computes the same lists (the barrier inside is the block's)
	const CsRead R = cs_prepare<false>(A, read, lane, l_start, l_pref, l_code);
	const uint32_t H = R.H;
	const int L = R.L;
	if (!WIDE && H > kCsHeavyMaxHits16) { if (wv == 0) cs_enqueue(A, read, lane, R); return; }   // (block-uniform)
	__syncthreads();
	const uint32_t cap = (n_slots * 3u) / 4u;
	auto bin_of = [&](uint32_t pos, int li) -> uint32_t {
		const int p = li >> 1;
		const uint32_t correction = (li & 1) ? (uint32_t) (L - (p + k)) : (uint32_t) p;  // CS.cpp:140-142
		return (pos - correction) >> A.bin_shift;
	};
	auto insert = [&](uint32_t bin, bool rev) {
		uint32_t slot = (bin * 0x85EBCA6Bu) >> (32 - log2_slots);
		for (;;) {
			const uint32_t prev = atomicCAS(&t_keys[slot], 0xFFFFFFFFu, bin);
			if (prev == bin || prev == 0xFFFFFFFFu) break;
			slot = (slot + 1) & (n_slots - 1);
		}
		atomicAdd(&t_votes[slot], rev ? 0x10000u : 1u);
	};
	uint32_t T = 1;
	if (H > cap) {
		cs_for_each_hit_block<NT>(A.positions, l_start, l_pref, R.n_lists, H, tid, [&](uint32_t pos, int li) {
			const uint32_t hc = (bin_of(pos, li) * 0x9E3779B1u) >> (32 - log2c);
			if (WIDE) atomicAdd(&cnt[hc], 1u); else atomicAdd(&cnt[hc >> 1], 1u << ((hc & 1u) * 16u));
		});
		__syncthreads();
		for (uint32_t i = tid; i < cnt_words; i += NT) {
			const uint32_t w = cnt[i];
			const uint32_t c0 = WIDE ? w : (w & 0xFFFFu), c1 = WIDE ? 0u : (w >> 16);
			if (c0 > 1u) atomicAdd(&hist[min(c0, 255u)], c0);   // (counters of 0 and 1 are most of them: T >= 2 here, they never matter)
			if (c1 > 1u) atomicAdd(&hist[min(c1, 255u)], c1);
		}
		__syncthreads();
		if (tid == 0) {
			uint32_t acc = 0, t = 256u;   // 256: even the counters of 255 and more carry more hits than the table takes
			for (uint32_t c = 255u; c >= 2u; --c) { acc += hist[c]; if (acc > cap) break; t = c; }
			s_T = t;
		}
		__syncthreads();
		T = s_T;
		if (T > 255u) { if (wv == 0) cs_enqueue(A, read, lane, R); return; }
		cs_for_each_hit_block<NT>(A.positions, l_start, l_pref, R.n_lists, H, tid, [&](uint32_t pos, int li) {
			const uint32_t bin = bin_of(pos, li);
			const uint32_t hc = (bin * 0x9E3779B1u) >> (32 - log2c);
			const uint32_t c = WIDE ? cnt[hc] : ((cnt[hc >> 1] >> ((hc & 1u) * 16u)) & 0xFFFFu);
			if (c >= T) insert(bin, (li & 1) != 0);
		});
	} else {
		cs_for_each_hit_block<NT>(A.positions, l_start, l_pref, R.n_lists, H, tid, [&](uint32_t pos, int li) { insert(bin_of(pos, li), (li & 1) != 0); });
	}
	__syncthreads();
	if (wv != 0) return;
	if (T > 1u) {
		int mx = 0;
		for (uint32_t s = lane; s < n_slots; s += 64) { const uint32_t v = t_votes[s]; mx = max(mx, (int) max(v & 0xFFFFu, v >> 16)); }
		mx = wave_reduce_max(mx);
		const float thresh = fmaxf(A.kmer_min, (float) mx * A.sensitivity);
		if (!((float) (T - 1u) < thresh)) { cs_enqueue(A, read, lane, R); return; }   // bins outside the table could reach the threshold
	}
	(void) cs_finish<kCsExactLds>(A, read, lane, R, t_keys, t_votes, n_slots);
}

}  // namespace ngm
