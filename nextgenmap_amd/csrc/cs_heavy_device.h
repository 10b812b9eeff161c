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


// ---- the exact search with the table in global memory, one WORKGROUP per read (round 4) -------------------------------------------
// cs_kernel<kCsExactGlobal> gives a read ONE wave: the reads that reach it on a heavy-tailed genome (17 000 - 190 000 hits, tens of
// thousands of bins with votes: 4 % of the reads there) each clear, fill and scan -- three times: maximum, count, output -- a table of
// 2^17 - 2^19 slots with 64 lanes, one L2 round trip per iteration: ~10 ms per read, 68-74 ms per 262 144 reads of the bench's
// heavy-tailed leg (70 % of its search time).  Here NT threads share the read: the votes through cs_for_each_hit_block, the three table
// passes NT slots at a time.  The candidates leave in cs_finish's order -- by (slot mod 64), then by slot -- so that nothing downstream
// can tell the kernels apart: thread t owns the slots of lane class t mod 64 in the (t / 64)-th share of the table, and the output
// offsets are a scan over the threads in (class, share) order.
template <int NT>
__global__ __launch_bounds__(NT) void cs_global_kernel(CsArgs A) {
	extern __shared__ __attribute__((aligned(16))) uint32_t cs_lds[];
	constexpr int NW = NT / 64;
	__shared__ uint32_t s_cnt[NT], s_wtot[NW], s_mx[NW], s_mxb[NW];
	__shared__ unsigned long long s_base;
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const int item = blockIdx.x;
	const int read = (int) A.read_list[item];
	const int k = A.k;
	uint32_t *l_start = cs_lds;                                  // [lists_cap]
	uint32_t *l_pref = cs_lds + A.lists_cap;                     // [lists_cap + 1]
	uint8_t *l_code = (uint8_t *) (l_pref + A.lists_cap + 1);    // [q rounded up to 4]
	const int log2_slots = (int) A.ovf_log2[item];
	const uint32_t n_slots = 1u << log2_slots;
	uint32_t *t_keys = A.gtable_keys + A.ovf_table_off[item];
	uint32_t *t_votes = A.gtable_votes + A.ovf_table_off[item];
	for (uint32_t s = tid; s < n_slots; s += NT) { t_keys[s] = 0xFFFFFFFFu; t_votes[s] = 0; }
	const CsRead R = cs_prepare<false>(A, read, lane, l_start, l_pref, l_code);   // (every wave computes the same lists; the barrier inside is the block's)
	const uint32_t H = R.H;
	const int L = R.L;
	__threadfence_block();
	__syncthreads();
	cs_for_each_hit_block<NT>(A.positions, l_start, l_pref, R.n_lists, H, tid, [&](uint32_t pos, int li) {
		const int p = li >> 1;
		const bool rev = (li & 1) != 0;
		const uint32_t correction = rev ? (uint32_t) (L - (p + k)) : (uint32_t) p;  // CS.cpp:140-142
		const uint32_t bin = (pos - correction) >> A.bin_shift;
		uint32_t slot = (bin * 2654435761u) >> (32 - log2_slots);
		for (;;) {
			const uint32_t prev = atomicCAS(&t_keys[slot], 0xFFFFFFFFu, bin);
			if (prev == bin || prev == 0xFFFFFFFFu) break;
			slot = (slot + 1) & (n_slots - 1);
		}
		atomicAdd(&t_votes[slot], rev ? 0x10000u : 1u);
	});
	__threadfence_block();
	__syncthreads();
	// thread t: lane class c = t mod 64, share sh = t / 64 of the class's slots c, c + 64, c + 128, ...
	const uint32_t per_class = n_slots >> 6;   // (n_slots >= 2^4: tables of fewer than 64 slots have per_class 0 -- such reads never get here, but stay correct below)
	const uint32_t j0 = (uint32_t) ((unsigned long long) per_class * (unsigned) wv / (unsigned) NW), j1 = (uint32_t) ((unsigned long long) per_class * (unsigned) (wv + 1) / (unsigned) NW);
	int mx = 0, mxb = 0;
	if (n_slots >= 64u) {
		for (uint32_t j = j0; j < j1; ++j) {
			const uint32_t v = cs_tload<kCsExactGlobal>(&t_votes[(j << 6) + (uint32_t) lane]);
			mx = max(mx, (int) max(v & 0xFFFFu, v >> 16));
			mxb = max(mxb, (int) ((v & 0xFFFFu) + (v >> 16)));
		}
	} else if (wv == 0 && (uint32_t) lane < n_slots) {
		const uint32_t v = cs_tload<kCsExactGlobal>(&t_votes[lane]);
		mx = (int) max(v & 0xFFFFu, v >> 16); mxb = (int) ((v & 0xFFFFu) + (v >> 16));
	}
	mx = wave_reduce_max(mx);
	mxb = wave_reduce_max(mxb);
	if (lane == 0) { s_mx[wv] = (uint32_t) mx; s_mxb[wv] = (uint32_t) mxb; }
	__syncthreads();
	mx = 0; mxb = 0;
#pragma unroll
	for (int w2 = 0; w2 < NW; ++w2) { mx = max(mx, (int) s_mx[w2]); mxb = max(mxb, (int) s_mxb[w2]); }
	const float max_hit = (float) mx;
	const float thresh = fmaxf(A.kmer_min, max_hit * A.sensitivity);
	const uint32_t region = (uint32_t) read & (kCsRegions - 1);
	if (tid == 0 && A.counters) {
		atomicAdd(&A.counters[region * kCsCursorStride], (unsigned long long) R.n_valid);
		atomicAdd(&A.counters[region * kCsCursorStride + 1], (unsigned long long) H);
	}
	auto my_slots = [&](auto f) {
		if (n_slots >= 64u) { for (uint32_t j = j0; j < j1; ++j) f((j << 6) + (uint32_t) lane); }
		else if (wv == 0 && (uint32_t) lane < n_slots) f((uint32_t) lane);
	};
	uint32_t count = 0;
	my_slots([&](uint32_t s2) {
		if (cs_tload<kCsExactGlobal>(&t_keys[s2]) != 0xFFFFFFFFu) {
			const uint32_t v = cs_tload<kCsExactGlobal>(&t_votes[s2]);
			count += ((float) (v & 0xFFFFu) >= thresh) + ((float) (v >> 16) >= thresh);
		}
	});
	// exclusive scan in (class, share) order: entry o = lane * NW + wv
	s_cnt[lane * NW + wv] = count;
	__syncthreads();
	{
		const uint32_t v = s_cnt[tid];
		const uint32_t incl = wave_inclusive_scan(v, lane);
		__syncthreads();
		s_cnt[tid] = incl - v;
		if (lane == 63) s_wtot[wv] = incl;
	}
	__syncthreads();
	const uint32_t o = (uint32_t) (lane * NW + wv);
	uint32_t before = s_cnt[o], total = 0;
#pragma unroll
	for (int w2 = 0; w2 < NW; ++w2) { if ((uint32_t) w2 < (o >> 6)) before += s_wtot[w2]; total += s_wtot[w2]; }
	if ((int64_t) total >= (int64_t) A.max_cmrs) total = 0;  // "if (index < maxScores) AllocScores" (CS.cpp:308-310)
	const bool fixed = A.fixed_base != 0u && total <= (uint32_t) kCsFixedSlots;
	if (tid == 0) {
		unsigned long long base = 0;
		if (!fixed) {
			base = total ? atomicAdd(&A.out_total[region * kCsCursorStride], (unsigned long long) total) : 0ull;
			if (base + total > A.out_capacity) { atomicExch(&A.status[0], 1u); }
		}
		s_base = base;
		A.cand_base[read] = fixed ? A.fixed_base + (uint32_t) read * (uint32_t) kCsFixedSlots : (uint32_t) (region * A.out_capacity + base);
		A.cand_count[read] = total;
		A.max_votes[read] = max_hit;
		if (A.max_both) A.max_both[read] = (float) mxb;
		A.read_len[read] = (uint16_t) R.L;
		if (A.counters && total) atomicAdd(&A.counters[region * kCsCursorStride + 2], (unsigned long long) total);
	}
	__syncthreads();
	if (total == 0) return;
	uint32_t w;
	if (fixed) w = A.fixed_base + (uint32_t) read * (uint32_t) kCsFixedSlots + before;
	else {
		const unsigned long long base = s_base;
		if (base + total > A.out_capacity) return;
		w = (uint32_t) (region * A.out_capacity + base) + before;
	}
	const uint32_t centre = A.bin_shift > 0 ? (1u << (A.bin_shift - 1)) : 0u;  // ResolveBin, CS.h:170-175
	my_slots([&](uint32_t s2) {
		const uint32_t key = cs_tload<kCsExactGlobal>(&t_keys[s2]);
		if (key != 0xFFFFFFFFu) {
			const uint32_t v = cs_tload<kCsExactGlobal>(&t_votes[s2]);
			const uint32_t f = v & 0xFFFFu, r = v >> 16;
			const uint32_t loc = (key << A.bin_shift) + centre;
			if ((float) f >= thresh) { A.out_loc[w] = loc; A.out_sv[w] = f << 1; ++w; }
			if ((float) r >= thresh) { A.out_loc[w] = loc; A.out_sv[w] = (r << 1) | 1u; ++w; }
		}
	});
}

}  // namespace ngm
