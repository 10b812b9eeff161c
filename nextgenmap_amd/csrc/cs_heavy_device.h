// cs_heavy_device.h -- candidate search for the reads the fast path hands on: thousands to tens of thousands of index hits.
// Same semantics as every other search kernel: CS::PrefixIteration (src/CSstatic.cpp:26-76), GetRefEntry
// (src/PrefixTable.cpp:750-817), PrefixSearch / AddLocationStd (src/CS.cpp:114-213), CollectResultsStd (src/CS.cpp:263-313).
//
// Why it exists.  On a genome with a GRCh38-like k-mer spectrum (tests/humanlike.py; automatic max. k-mer frequency 1 531 - 4 681
// instead of 100) a third to a half of the reads carry more hits than the fast path's bit plane and 1 024-slot table take, and the
// exact kernels that used to receive them hold ONE read per CU (a 128 KB table in LDS, one wave) or vote through L2 atomics into
// tables in global memory: 33 ms + 243 ms per 262 144 reads against 3 ms for the fast path (profiles/r04_heavy_tail_cs_passes.txt).
// A read with H hits needs H votes, not a table of H entries: cs_heavy2_kernel below (round 5; round 4's one-row kernel is gone) counts
// the hits in a sketch, filters them by it, and certifies the exact table it then builds.  cs_global_kernel is what remains for the
// reads no class can certify.
#pragma once

#include "cs_device.h"
#include "cs_queue_device.h"

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

// ---- the exact search with the table in global memory, one WORKGROUP per read (round 4) -------------------------------------------
// cs_kernel<kCsExactGlobal> gives a read ONE wave: the reads that reach it on a heavy-tailed genome (17 000 - 190 000 hits, tens of
// thousands of bins with votes: 4 % of the reads there) each clear, fill and scan -- three times: maximum, count, output -- a table of
// 2^17 - 2^19 slots with 64 lanes, one L2 round trip per iteration: ~10 ms per read, 68-74 ms per 262 144 reads of the bench's
// heavy-tailed leg (70 % of its search time).  Here NT threads share the read: the votes through cs_for_each_hit_block, the three table
// passes NT slots at a time.  The candidates leave in cs_finish's order -- by (slot mod 64), then by slot -- so that nothing downstream
// can tell the kernels apart: thread t owns the slots of lane class t mod 64 in the (t / 64)-th share of the table, and the output
// offsets are a scan over the threads in (class, share) order.
// Round 6: the number of listed reads comes from a device word (A.n_list_dev; cs_queue_device.h) and the workgroups stride over the list.
template <int NT>
__global__ __launch_bounds__(NT) void cs_global_kernel(CsArgs A) {
	extern __shared__ __attribute__((aligned(16))) uint32_t cs_lds[];
	constexpr int NW = NT / 64;
	__shared__ uint32_t s_cnt[NT], s_wtot[NW], s_mx[NW], s_mxb[NW];
	__shared__ unsigned long long s_base;
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const int n_items = A.n_list_dev ? (int) *A.n_list_dev : (int) gridDim.x;
	for (int item = blockIdx.x; item < n_items; item += (int) gridDim.x) {
	__syncthreads();   // (the previous read's shared state is no longer read)
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
	if (total == 0) continue;
	uint32_t w;
	if (fixed) w = A.fixed_base + (uint32_t) read * (uint32_t) kCsFixedSlots + before;
	else {
		const unsigned long long base = s_base;
		if (base + total > A.out_capacity) continue;
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
}

// ---- round 5: the heavy reads again -- segment-wise sweeps, a second counter row over the survivors, persistent workgroups ------------
// What round 4's kernel left (profiles/r05_heavy_tail_probe_kernel_stats_before_heavy2.csv, per 262 144 reads of the heavy-tailed
// leg): its four classes 26 ms, and cs_global_kernel 47 ms for the 10 275 reads none of them could certify -- reads with thousands of
// bins near the threshold.  Two things were wrong with it.  (1) The table had to take every HIT on a counter >= T: T was the smallest
// value whose hits fit, so a read with 3 000 bins of ~20 votes (60 000 hits on its hot counters) failed a table that its 3 000 bins
// would have filled to a half.  And a single counter row lets every background hit that shares a counter with a hot bin through --
// at 3-6 background hits per counter several times the hot bins themselves.  (2) The sweeps walked the lists hit by hit
// (cs_for_each_hit_block: a bisection per 8 hits, three dependent LDS reads and a 4-byte load per hit).
// Now:
//   sweep A   every hit increments its counter of row 1 (16-bit, hash 1);
//   T         the smallest value for which the COUNTERS >= T (about one bin each) fit the table; if even their hits fit, they go
//             straight into the table (the round-4 path: most reads);
//   sweep B   otherwise the hits on counters >= T are written (bin | strand << 31) to the workgroup's slice of a global scratch;
//   sweep C   (over the slice) row 2: hash 2, the same LDS words as row 1 -- only the survivors count, so row 2 is almost free of noise;
//   sweep D   (over the slice) the hits whose row-2 counter is >= T as well go into the exact table; entries are counted, and a table
//             filling beyond 3/4 fails the read;
//   check     a bin with v >= T votes has both its counters >= v >= T, so all its hits reach the table: as before, with M2 the
//             largest strand count in the table, T - 1 < max(kmer_min, M2 * sensitivity) certifies the result exactly.
// 16-bit counters in every class: a row is checked by its SUM (a field that wrapped into its neighbour changes the sum of the
// fields), which replaces the 32-bit class.  Work items are 8-hit segments of one list (two 16-byte loads, constant strand and
// diagonal correction; item -> list through a coarse table + a short bisection).  Workgroups are persistent (one scratch slice each)
// and draw reads from a counter.  The ORDER in which a read's candidates leave is that of the table's slots (one table), or of the list
// the table passes append their entries to (several) -- neither is fixed from run to run: two bins that hash to one slot race for it,
// and the list grows by an atomic counter.  Nothing downstream reads it: select_top1_kernel and pair_choice_kernel compare
// (location, strand) keys, the host's sorts (sort_like_reference, ngm_mapper_cs_fetch) use total orders on rank / location / strand,
// and the reference's own candidate order -- where it decides a tie -- comes from the replay kernels, never from this one.
constexpr int kCsHeavyCoarseShift = 5;   // the coarse item -> list table has an entry per 32 items (16: at GRCh38 size the middle class needed 82 KB of LDS -- one workgroup per CU instead of two)

inline size_t cs_heavy2_coarse_cap(int lists_cap, int max_kfreq) {   // items / 32 + slack: a read has at most lists_cap / 2 * max_kfreq hits
	const size_t items = ((size_t) (lists_cap / 2) * (size_t) max_kfreq) / kCsSeg + (size_t) lists_cap;
	return (items >> kCsHeavyCoarseShift) + 4;
}
inline size_t cs_heavy2_lds_bytes(int lists_cap, int q, int log2_counters, int log2_slots, size_t coarse_cap) {
	return ((size_t) lists_cap * 3 + 2 + (size_t) (q + 3) / 4 + (coarse_cap + 1) / 2 + 3 + ((size_t) 1 << (log2_counters - 1)) + 512 + ((size_t) 2 << log2_slots)) * 4;
}

// Round 6 also tried BOTH counter rows in sweep A (over all hits) with the second sweep voting straight into the table -- a hit fetched
// twice, no scratch slice: 2.0 x the algorithmic bytes instead of 2.9 x.  The rows must then be twice as large (row 2 over all hits is
// noisier than over the survivors): one workgroup of 1 024 threads per CU instead of two of 512, and although a read took 54 us
// instead of 86, a CU finished fewer of them: candidate search 252 against 188 ms per step at 3.1 Gbp (profiles/r06_cs_heavy2_two_rows.patch,
// profiles/r06_heavy_tail_two_rows_ab.txt).  The SQ counters taken afterwards say why: the kernel issues VALU instructions for 0.71 of its
// cycles (~35 lane operations per index hit over the sweeps) -- it is instruction-bound, not traffic- or latency-bound.
template <int NT>
__global__ __launch_bounds__(NT) void cs_heavy2_kernel(CsArgs A, uint32_t *__restrict__ ctl, int cls, uint32_t *__restrict__ scratch, uint32_t scratch_cap,
		uint32_t coarse_cap, uint32_t max_parts, uint32_t ent_cap, unsigned long long *__restrict__ diag) {   // ctl: the search's control block (cs_queue_device.h) -- the class's list length and work counter, the run's statistics   // diag (NGM_HIP_CS_PHASES): [0..6] 100 MHz ticks per phase of every 8th read, [8] reads sampled, [9] their hits, [10] settled without a second row, [11] survivors, [12] table passes of the reads that needed several, [13] reads sent into a second pass
	extern __shared__ __attribute__((aligned(16))) uint32_t cs_lds[];
	constexpr int NW = NT / 64;
	__shared__ uint32_t s_T, s_next, s_np, s_entries, s_fail, s_direct, s_nhot, s_nent, s_force, s_retry_ix, s_retry_T, s_short;
	__shared__ uint32_t s_red[NW], s_cnt[NT], s_wtot[NW], s_mx[NW], s_mxb[NW];
	__shared__ unsigned long long s_base;
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const int k = A.k;
	uint32_t *l_start = cs_lds;                                  // [lists_cap]
	uint32_t *l_pref = cs_lds + A.lists_cap;                     // [lists_cap + 1]
	uint8_t *l_code = (uint8_t *) (l_pref + A.lists_cap + 1);    // [q rounded up to 4]
	uint32_t *seg_pref = (uint32_t *) l_code + (A.q + 3) / 4;    // [lists_cap + 1]: 8-hit segments in front of each list
	uint16_t *coarse = (uint16_t *) (seg_pref + A.lists_cap + 1); // [coarse_cap]: the list that holds item 32 c
	uint32_t *cnt = cs_lds + (((size_t) ((uint32_t *) coarse - cs_lds) + (coarse_cap + 1) / 2 + 3) & ~(size_t) 3);  // [NC / 2]: two 16-bit counters per word (16-byte aligned: cleared with 128-bit stores)
	const int log2c = A.log2_bits;
	const uint32_t cnt_words = (1u << log2c) >> 1;
	uint32_t *hist_n = cnt + cnt_words;                          // [256]: counters of value c (255: and above)
	uint32_t *hist_h = hist_n + 256;                             // [256]: ... and the hits on them
	uint32_t *t_keys = hist_h + 256;
	const int log2_slots = A.log2_slots;
	const uint32_t n_slots = 1u << log2_slots;
	uint32_t *t_votes = t_keys + n_slots;
	const uint32_t cap = (n_slots * 3u) / 4u;
	uint32_t *my_scratch = scratch + (size_t) blockIdx.x * ((size_t) scratch_cap + 2u * (size_t) ent_cap);   // survivors, then (max_parts > 1) the entries of all parts
	uint32_t *my_ent = my_scratch + scratch_cap;
	const unsigned long long lanes_below = (1ull << lane) - 1ull;
	const uint32_t n_list = ctl[kCsqCount + cls];   // (written by cs_heavy_classify_kernel before this launch)
	uint32_t *const work_counter = ctl + kCsqWork + cls;
	// A read of the largest class (max_parts > 1) gets a SECOND pass when the first one -- T the smallest value whose bins fit ONE table -- ends
	// with T - 1 >= max(kmer_min, M2 * sensitivity): M2, exact for the bins it saw, is a lower bound of the true maximum, so the largest T' with
	// T' - 1 < that threshold is sure to certify, and the bins at or above T' are taken in as many table passes as they need (s_retry_*).
	if (tid == 0) s_retry_T = 0;
	for (;;) {
		__syncthreads();   // (the previous read's shared state is no longer read)
		if (tid == 0) {
			if (s_retry_T) { s_next = s_retry_ix; s_force = s_retry_T; s_retry_T = 0; }
			else { s_next = atomicAdd(work_counter, 1u); s_force = 0; }
		}
		__syncthreads();
		const uint32_t item_ix = s_next;
		const uint32_t T_force = s_force;
		const uint32_t parts_now = T_force ? max(max_parts, 1u) : 1u;
		if (item_ix >= n_list) return;
		const int read = (int) A.read_list[item_ix];
		const bool dg = diag != nullptr && (item_ix & 7u) == 0u && tid == 0;
		unsigned long long tk = dg ? wall_clock64() : 0ull;
		auto mark = [&](int ph) { if (dg) { const unsigned long long t2 = wall_clock64(); atomicAdd(&diag[ph], t2 - tk); tk = t2; } };
		{
			uint4 *c4 = reinterpret_cast<uint4 *>(cnt);   // counters, both histograms: zero; keys: empty; votes: zero
			for (uint32_t s = tid; s < (cnt_words + 512u) / 4u; s += NT) c4[s] = make_uint4(0u, 0u, 0u, 0u);
			uint4 *k4 = reinterpret_cast<uint4 *>(t_keys), *v4 = reinterpret_cast<uint4 *>(t_votes);
			for (uint32_t s = tid; s < n_slots / 4u; s += NT) { k4[s] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu); v4[s] = make_uint4(0u, 0u, 0u, 0u); }
		}
		if (tid == 0) { s_T = 1; s_np = 0; s_entries = 0; s_fail = 0; s_direct = 1; }
		// every wave computes the same lists (the barrier inside is the block's)
		const CsRead R = cs_prepare<false>(A, read, lane, l_start, l_pref, l_code);
		const uint32_t H = R.H;
		const int L = R.L;
		const int n_lists = R.n_lists;
		__syncthreads();
		if (wv == 0) {
			uint32_t carry = 0;
			for (int base = 0; base < n_lists; base += 64) {
				const int li = base + lane;
				const uint32_t ns = li < n_lists ? (l_pref[li + 1] - l_pref[li] + kCsSeg - 1) / kCsSeg : 0u;
				const uint32_t incl = wave_inclusive_scan(ns, lane);
				if (li < n_lists) seg_pref[li] = carry + incl - ns;
				carry += wave_last(incl);
			}
			if (lane == 0) seg_pref[n_lists] = carry;
		}
		__syncthreads();
		const uint32_t n_items = seg_pref[n_lists];
		if ((n_items >> kCsHeavyCoarseShift) + 3u > coarse_cap) { if (wv == 0) { cs_enqueue(A, read, lane, R); if (lane == 0) atomicAdd(&ctl[kCsqSentOn], 1u); } continue; }   // (block-uniform; sized from max_kfreq: not reached)
		for (int li = tid; li < n_lists; li += NT) {
			const uint32_t s0 = seg_pref[li], s1 = seg_pref[li + 1];
			constexpr uint32_t cm = (1u << kCsHeavyCoarseShift) - 1u;
			for (uint32_t c = (s0 + cm) >> kCsHeavyCoarseShift; (c << kCsHeavyCoarseShift) < s1; ++c) coarse[c] = (uint16_t) li;
		}
		if (tid == 0) {
			constexpr uint32_t cm = (1u << kCsHeavyCoarseShift) - 1u;
			coarse[((n_items + cm) >> kCsHeavyCoarseShift)] = (uint16_t) max(n_lists - 1, 0); coarse[((n_items + cm) >> kCsHeavyCoarseShift) + 1] = (uint16_t) max(n_lists - 1, 0);
		}
		__syncthreads();
		// f(position, list) for every hit: per thread one 8-hit segment at a time, the next one's loads in flight
		auto sweep = [&](auto f) {
			CsU4 cur[2], nxt[2];
			auto fetch = [&](uint32_t idx, CsU4 (&d)[2]) -> uint32_t {
				if (idx >= n_items) return 0xFFFFFFFFu;
				int lo = (int) coarse[idx >> kCsHeavyCoarseShift], hi = min((int) coarse[(idx >> kCsHeavyCoarseShift) + 1] + 1, n_lists);   // the list with seg_pref[li] <= idx < seg_pref[li + 1] (never an empty one)
				while (hi - lo > 1) {
					const int mid = (lo + hi) >> 1;
					if (seg_pref[mid] <= idx) lo = mid; else hi = mid;
				}
				const uint32_t sg = idx - seg_pref[lo];
				const CsU4 *src = reinterpret_cast<const CsU4 *>(A.positions + l_start[lo] + sg * kCsSeg);
				d[0] = src[0]; d[1] = src[1];  // the table is padded by 16 entries
				return ((uint32_t) lo << 16) | sg;
			};
			// (round 6 tried TWO segments' loads in flight behind the one being voted: 2 % -- and eight more registers, which the sweeps need)
			uint32_t item = fetch((uint32_t) tid, cur);
			for (uint32_t idx = (uint32_t) tid; idx < n_items; idx += NT) {
				const uint32_t item_n = fetch(idx + NT, nxt);
				const int li = (int) (item >> 16);
				const uint32_t sg = item & 0xFFFFu;
				const uint32_t len = l_pref[li + 1] - l_pref[li];
				const uint32_t cn = min((uint32_t) kCsSeg, len - sg * kCsSeg);
				const int p = li >> 1;
				const uint32_t correction = (li & 1) ? (uint32_t) (L - (p + k)) : (uint32_t) p;  // CS.cpp:140-142
				const uint32_t pos8[8] = {cur[0].x, cur[0].y, cur[0].z, cur[0].w, cur[1].x, cur[1].y, cur[1].z, cur[1].w};
				f(pos8, cn, correction, (li & 1) != 0, len);
				item = item_n; cur[0] = nxt[0]; cur[1] = nxt[1];
			}
		};
		auto insert = [&](uint32_t bin, bool rev) {
			uint32_t slot = (bin * 0x85EBCA6Bu) >> (32 - log2_slots);
			for (;;) {
				const uint32_t prev = atomicCAS(&t_keys[slot], 0xFFFFFFFFu, bin);
				if (prev == bin) break;
				if (prev == 0xFFFFFFFFu) { if (atomicAdd(&s_entries, 1u) >= cap) atomicExch(&s_fail, 1u); break; }
				slot = (slot + 1) & (n_slots - 1);
			}
			atomicAdd(&t_votes[slot], rev ? 0x10000u : 1u);
		};
		auto counter_of = [&](uint32_t hc) -> uint32_t { return (cnt[hc >> 1] >> ((hc & 1u) * 16u)) & 0xFFFFu; };
		// histogram of the row's counters of at least `from` (hist_n: how many of each value, 255: and above; with_hits: hist_h, the hits on
		// them) and the sum of ALL its 16-bit fields, which must be `want`: a field that wrapped into its neighbour changes the sum
		auto row_hist = [&](const uint32_t *row, uint32_t want, uint32_t from, bool with_hits) -> bool {
			uint32_t sm = 0;
			for (uint32_t i = tid; i < cnt_words; i += NT) {
				const uint32_t w = row[i];
				const uint32_t c0 = w & 0xFFFFu, c1 = w >> 16;
				sm += c0 + c1;
				if (c0 >= from) { atomicAdd(&hist_n[min(c0, 255u)], 1u); if (with_hits) atomicAdd(&hist_h[min(c0, 255u)], c0); }   // (counters of 0 and 1 are most of them: T >= 2, they never matter)
				if (c1 >= from) { atomicAdd(&hist_n[min(c1, 255u)], 1u); if (with_hits) atomicAdd(&hist_h[min(c1, 255u)], c1); }
			}
			sm = wave_last(wave_inclusive_scan(sm, lane));
			if (lane == 0) s_red[wv] = sm;
			__syncthreads();
			uint32_t tot = 0;
#pragma unroll
			for (int w2 = 0; w2 < NW; ++w2) tot += s_red[w2];
			return tot == want;
		};
		// Round 6: a LOWER bound of the read's maximum before the threshold is picked.  Sweep A also votes the hits of the SHORT lists -- the
		// lists of at most `n_short` hits, as many of them as half the table takes -- into the exact table: the largest strand count L found
		// there is made of real votes of one bin, so the read's maximum is at least L and its final threshold at least max(kmer_min, L *
		// sensitivity).  Every T with T - 1 below THAT is certain to certify (the bin behind L has at least L >= T votes: it reaches the
		// table, M2 >= L), so T starts there instead of at the smallest value the table's room allows: on a GRCh38-sized index, where a
		// k-mer has ~15 chance occurrences, half of a read's lists are short, L is about half the true maximum, and the survivors of
		// row 1 fall from ~40 % of the hits (T = 3-5: counters with a few chance hits) to the hits of bins with real votes.
		if (H > cap && wv == 0) {
			const uint32_t budget = n_slots / 2u;
			for (int base = 0; base < n_lists; base += 64) {
				const int li = base + lane;
				const uint32_t len = li < n_lists ? l_pref[li + 1] - l_pref[li] : 0u;
				if (len >= 1u && len <= 64u) atomicAdd(&hist_h[len], len);   // (hist_h is zero here; entries 1..64 are zeroed again below)
			}
			const uint32_t mine = hist_h[lane + 1];
			const uint32_t incl = wave_inclusive_scan(mine, lane);
			const uint32_t ns = (uint32_t) __popcll(__ballot(incl <= budget));   // lists of up to ns hits fit (the sums grow with the length)
			hist_h[lane + 1] = 0;
			if (lane == 0) s_short = (T_force || A.fast_items < 0) ? 0u : ns;   // (a second pass has its T from the first pass's exact maximum; fast_items < 0: A/B runs without the bound)
		}
		uint32_t T = 1;
		bool failed = false;
		uint32_t why = 0;   // (diagnostics) what sent the read on: 1 a counter row wrapped, 2 no T <= 255 fits, 3 more survivors than the slice, 4 the table (or the entry list) overflowed, 5 T - 1 not below the threshold
		mark(0);
		if (H > cap) {
			// sweep A
			__syncthreads();
			const uint32_t n_short = s_short;
			sweep([&](const uint32_t (&pos8)[8], uint32_t cn, uint32_t correction, bool rev, uint32_t len) {
				uint32_t bin[kCsSeg];
#pragma unroll
				for (int j = 0; j < kCsSeg; ++j) {
					// (no branch per hit: a hit beyond the segment's end adds 0 -- the SQ counters have this kernel at 0.71 of the VALU issue slots
					// with a scalar branch pair around every hit)
					bin[j] = (pos8[j] - correction) >> A.bin_shift;
					const uint32_t hc = (bin[j] * 0x9E3779B1u) >> (32 - log2c);
					atomicAdd(&cnt[hc >> 1], (uint32_t) j < cn ? 1u << ((hc & 1u) * 16u) : 0u);
				}
				if (len <= n_short) {
					// the segment's first probes in flight together (a returning LDS atomic per hit, one after the other, made this sweep three
					// times as long); no entry count: the budget keeps the table below half full, and it is emptied again below
					uint32_t slot[kCsSeg], prev[kCsSeg];
#pragma unroll
					for (int j = 0; j < kCsSeg; ++j) {
						slot[j] = (bin[j] * 0x85EBCA6Bu) >> (32 - log2_slots);
						prev[j] = (uint32_t) j < cn ? atomicCAS(&t_keys[slot[j]], 0xFFFFFFFFu, bin[j]) : 0u;
					}
#pragma unroll
					for (int j = 0; j < kCsSeg; ++j) if ((uint32_t) j < cn) {
						uint32_t sl = slot[j], pv = prev[j];
						while (pv != bin[j] && pv != 0xFFFFFFFFu) { sl = (sl + 1) & (n_slots - 1); pv = atomicCAS(&t_keys[sl], 0xFFFFFFFFu, bin[j]); }
						atomicAdd(&t_votes[sl], rev ? 0x10000u : 1u);
					}
				}
			});
			__syncthreads();
			mark(1);
			uint32_t T_low = 2;
			if (n_short) {   // (block-uniform) L from the table, then the table is empty again
				int lm = 0;
				uint4 *k4 = reinterpret_cast<uint4 *>(t_keys), *v4 = reinterpret_cast<uint4 *>(t_votes);
				for (uint32_t s2 = tid; s2 < n_slots / 4u; s2 += NT) {
					const uint4 v = v4[s2];
					lm = max(lm, (int) max(max(max(v.x & 0xFFFFu, v.x >> 16), max(v.y & 0xFFFFu, v.y >> 16)), max(max(v.z & 0xFFFFu, v.z >> 16), max(v.w & 0xFFFFu, v.w >> 16))));
					k4[s2] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu); v4[s2] = make_uint4(0u, 0u, 0u, 0u);
				}
				lm = wave_reduce_max(lm);
				if (lane == 0) s_mx[wv] = (uint32_t) lm;
				__syncthreads();
				lm = 0;
#pragma unroll
				for (int w2 = 0; w2 < NW; ++w2) lm = max(lm, (int) s_mx[w2]);
				T_low = min(255u, max(2u, (uint32_t) ceilf(fmaxf(A.kmer_min, (float) lm * A.sensitivity))));
				if (dg) atomicAdd(&diag[7], (unsigned long long) T_low);
			}
			// histogram of the counter values -- and their sum: a 16-bit field that wrapped into its neighbour changes it
			if (!row_hist(cnt, H, 2u, true)) { failed = true; why = 1; }   // (block-uniform)
			if (!failed) {
				if (wv == 0) {
					// the smallest T whose counters fit the table with a quarter of it to spare (one bin per counter, and what slips through
					// both rows) and whose hits fit the scratch slice; when even the hits fit the table, no second row is needed.  Both
					// conditions are monotone in T: lane l looks at the values 4 l .. 4 l + 3, suffix sums from a wave scan.
					const uint32_t room = ((cap * 3u) / 4u) * parts_now;   // (a read's second pass takes the bins in several parts: below)
					const int t_min = (int) max(max(2u, T_force), T_low);
					uint32_t n4[4], h4[4], sn = 0, sh = 0;
#pragma unroll
					for (int j = 0; j < 4; ++j) { n4[j] = hist_n[4 * lane + j]; h4[j] = hist_h[4 * lane + j]; sn += n4[j]; sh += h4[j]; }
					const uint32_t in_n = wave_inclusive_scan(sn, lane), in_h = wave_inclusive_scan(sh, lane);
					const uint32_t tot_n = wave_last(in_n), tot_h = wave_last(in_h);
					uint32_t below_n = in_n - sn, below_h = in_h - sh;   // counters / hits of the values below 4 l
					int my_t = 256;
					uint32_t my_h = 0;
#pragma unroll
					for (int j = 0; j < 4; ++j) {
						const uint32_t suf_n = tot_n - below_n, suf_h = tot_h - below_h;   // values >= 4 l + j
						if (my_t == 256 && 4 * lane + j >= t_min && suf_n <= room && suf_h <= scratch_cap) { my_t = 4 * lane + j; my_h = suf_h; }
						below_n += n4[j]; below_h += h4[j];
					}
					const int t = wave_reduce_min(my_t);
					const bool direct = __ballot(my_t == t && t < 256 && my_h <= cap) != 0ull;
					if (lane == 0) { s_T = (uint32_t) t; s_direct = direct ? 1u : 0u; }
				}
				__syncthreads();
				T = s_T;
				if (T > 255u) { failed = true; why = 2; }
			}
			mark(2);
			if (!failed && s_direct) {
				sweep([&](const uint32_t (&pos8)[8], uint32_t cn, uint32_t correction, bool rev, uint32_t) {
					uint32_t bin[8], hc[8], cw[8];
#pragma unroll
					for (int j = 0; j < kCsSeg; ++j) {   // (the eight counter words in flight together, as in sweep B)
						bin[j] = (pos8[j] - correction) >> A.bin_shift;
						hc[j] = (bin[j] * 0x9E3779B1u) >> (32 - log2c);
						cw[j] = cnt[hc[j] >> 1];
					}
#pragma unroll
					for (int j = 0; j < kCsSeg; ++j) if ((uint32_t) j < cn && ((cw[j] >> ((hc[j] & 1u) * 16u)) & 0xFFFFu) >= T) insert(bin[j], rev);
				});
			} else if (!failed) {
				// sweep B: the survivors of row 1 -> scratch slice (one slot request per wave and trip)
				sweep([&](const uint32_t (&pos8)[8], uint32_t cn, uint32_t correction, bool rev, uint32_t) {
					uint32_t keep = 0, nk = 0;
					uint32_t e[8];
					uint32_t hc[8], cw[8];
#pragma unroll
					for (int j = 0; j < kCsSeg; ++j) {   // the eight counter words in flight together (round 5: a branch and an LDS round trip per hit)
						const uint32_t bin = (pos8[j] - correction) >> A.bin_shift;
						e[j] = (bin & 0x3FFFFFFFu) | (rev ? 0x80000000u : 0u);
						hc[j] = (bin * 0x9E3779B1u) >> (32 - log2c);
						cw[j] = cnt[hc[j] >> 1];
					}
#pragma unroll
					for (int j = 0; j < kCsSeg; ++j) {
						const uint32_t k1 = ((uint32_t) j < cn && ((cw[j] >> ((hc[j] & 1u) * 16u)) & 0xFFFFu) >= T) ? 1u : 0u;
						keep |= k1 << j; nk += k1;
					}
					// (lanes that left the loop do not take part: the prefix runs over the active ones)
					const unsigned long long act = __ballot(true);
					uint32_t pre = 0, tot = 0;
#pragma unroll
					for (int b = 0; b < 4; ++b) {
						const unsigned long long mb = __ballot((nk >> b) & 1u);
						pre += (uint32_t) __popcll(mb & lanes_below) << b;
						tot += (uint32_t) __popcll(mb) << b;
					}
					uint32_t base = 0;
					if (tot) {
						const int leader = (int) __builtin_ctzll(act);
						if (lane == leader) base = atomicAdd(&s_np, tot);
						base = (uint32_t) __builtin_amdgcn_readlane((int) base, leader);
					}
					uint32_t w = base + pre;
#pragma unroll
					for (int j = 0; j < kCsSeg; ++j) if ((keep >> j) & 1u) { if (w < scratch_cap) my_scratch[w] = e[j]; ++w; }
				});
				__threadfence_block();
				__syncthreads();
				mark(3);
				const uint32_t np = s_np;
				if (dg) atomicAdd(&diag[11], (unsigned long long) np);
				if (np > scratch_cap) { failed = true; why = 3; }
				// the survivors, K at a time per thread: their loads (L2) are in flight together
				constexpr int KS = 8;
				auto for_survivors = [&](uint32_t np_, auto f) {
					for (uint32_t x0 = (uint32_t) tid; x0 < np_; x0 += (uint32_t) NT * KS) {
						uint32_t e[KS];
#pragma unroll
						for (int j = 0; j < KS; ++j) { const uint32_t x = x0 + (uint32_t) j * NT; e[j] = x < np_ ? my_scratch[x] : 0xFFFFFFFFu; }
#pragma unroll
						for (int j = 0; j < KS; ++j) if (x0 + (uint32_t) j * NT < np_) f(e[j]);
					}
				};
				if (!failed) {
					{
						uint4 *c4 = reinterpret_cast<uint4 *>(cnt);
						for (uint32_t s = tid; s < (cnt_words + 256u) / 4u; s += NT) c4[s] = make_uint4(0u, 0u, 0u, 0u);   // (the counters and hist_n)
					}
					__syncthreads();
					// sweep C: row 2 over the survivors
					for_survivors(np, [&](uint32_t e) {
						const uint32_t hc = ((e & 0x3FFFFFFFu) * 0xC2B2AE35u + 0x27D4EB2Fu) >> (32 - log2c);
						atomicAdd(&cnt[hc >> 1], 1u << ((hc & 1u) * 16u));
					});
					__syncthreads();
					// row 2 is almost free of noise: its counters >= t are the bins with >= t votes -- the final T is the smallest one
					// (not below row 1's) whose bins leave the table a quarter of its room
					if (!row_hist(cnt, np, T, false)) { failed = true; why = 1; }
				}
				if (!failed) {
					if (wv == 0) {
						const uint32_t room = ((cap * 3u) / 4u) * parts_now;
						uint32_t n4[4], sn = 0;
#pragma unroll
						for (int j = 0; j < 4; ++j) { n4[j] = hist_n[4 * lane + j]; sn += n4[j]; }
						const uint32_t in_n = wave_inclusive_scan(sn, lane);
						const uint32_t tot_n = wave_last(in_n);
						uint32_t below_n = in_n - sn;
						int my_t = 256;
						uint32_t my_n = 0;
#pragma unroll
						for (int j = 0; j < 4; ++j) {
							if (my_t == 256 && (uint32_t) (4 * lane + j) >= T && tot_n - below_n <= room) { my_t = 4 * lane + j; my_n = tot_n - below_n; }
							below_n += n4[j];
						}
						const int t = wave_reduce_min(my_t);
						const unsigned long long own = __ballot(my_t == t && t < 256);
						const uint32_t nh = own ? (uint32_t) __builtin_amdgcn_readlane((int) my_n, (int) __builtin_ctzll(own)) : 0u;
						if (lane == 0) { s_T = (uint32_t) t; s_nhot = nh; }
					}
					__syncthreads();
					T = s_T;
					if (T > 255u) { failed = true; why = 2; }
				}
				mark(4);
				// sweep D: KS survivors per thread and trip -- their counter reads, then their first probes, are in flight together; part `pt` of
				// `parts` (a third hash of the bin): the bins the table takes in this pass
				auto sweep_d = [&](uint32_t pt, uint32_t parts) {
					for (uint32_t x0 = (uint32_t) tid; x0 < np; x0 += (uint32_t) NT * KS) {
						if (*(volatile uint32_t *) &s_fail) break;
						uint32_t e[KS], cv[KS], slot[KS], prev[KS];
#pragma unroll
						for (int j = 0; j < KS; ++j) { const uint32_t x = x0 + (uint32_t) j * NT; e[j] = x < np ? my_scratch[x] : 0xFFFFFFFFu; }
#pragma unroll
						for (int j = 0; j < KS; ++j) {
							cv[j] = e[j] != 0xFFFFFFFFu ? counter_of(((e[j] & 0x3FFFFFFFu) * 0xC2B2AE35u + 0x27D4EB2Fu) >> (32 - log2c)) : 0u;
							if (parts > 1u && __umulhi((e[j] & 0x3FFFFFFFu) * 0x7FEB352Du, parts) != pt) cv[j] = 0u;
						}
#pragma unroll
						for (int j = 0; j < KS; ++j) {
							slot[j] = ((e[j] & 0x3FFFFFFFu) * 0x85EBCA6Bu) >> (32 - log2_slots);
							prev[j] = 0;
							if (cv[j] >= T) prev[j] = atomicCAS(&t_keys[slot[j]], 0xFFFFFFFFu, e[j] & 0x3FFFFFFFu);
						}
#pragma unroll
						for (int j = 0; j < KS; ++j) if (cv[j] >= T) {
							const uint32_t bin = e[j] & 0x3FFFFFFFu;
							uint32_t sl = slot[j], pv = prev[j];
							for (uint32_t probes = 0; pv != bin && pv != 0xFFFFFFFFu && probes < n_slots; ++probes) {
								sl = (sl + 1) & (n_slots - 1);
								pv = atomicCAS(&t_keys[sl], 0xFFFFFFFFu, bin);
							}
							if (pv == 0xFFFFFFFFu) { if (atomicAdd(&s_entries, 1u) >= cap) atomicExch(&s_fail, 1u); }
							else if (pv != bin) { atomicExch(&s_fail, 1u); continue; }   // (the table is full: not reached, the entry count fails the read first)
							atomicAdd(&t_votes[sl], (e[j] >> 31) ? 0x10000u : 1u);
						}
					}
				};
				const uint32_t room1 = (cap * 3u) / 4u;
				uint32_t parts = (failed || s_nhot <= room1) ? 1u : min(parts_now, (s_nhot + s_nhot / 8u + room1 - 1u) / room1);
				if (!failed && parts == 1u && parts_now == 1u) sweep_d(0u, 1u);
				else if (!failed) {
					// More bins at or above T than the table holds: the table takes them in `parts` passes over the survivors and hands its
					// entries (bin, votes) to a list in the scratch slice; maximum, threshold and candidates then come from that list.
					// s_nhot counts COUNTERS: with more such bins than the row has counters (a read whose every k-mer is in a family of
					// tens of thousands of copies) it is far too small, a pass overflows, and the read used to leave for cs_global_kernel --
					// 1 000 reads per 262 144 at 3.1 Gbp, 15 % of that leg's GPU time.  Now the passes start over with twice as many parts.
					int pmx = 0, pmxb = 0;
					for (bool again = false;; again = true) {
					__syncthreads();
					if (tid == 0) s_nent = 0;
					pmx = 0; pmxb = 0;
					for (uint32_t pt = 0; pt < parts; ++pt) {
						__syncthreads();
						if (pt > 0u || again) {
							uint4 *k4 = reinterpret_cast<uint4 *>(t_keys), *v4 = reinterpret_cast<uint4 *>(t_votes);
							for (uint32_t s2 = tid; s2 < n_slots / 4u; s2 += NT) { k4[s2] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu); v4[s2] = make_uint4(0u, 0u, 0u, 0u); }
							if (tid == 0) s_entries = 0;
							__syncthreads();
						}
						sweep_d(pt, parts);
						__syncthreads();
						if (s_fail) break;   // (block-uniform)
						for (uint32_t s2 = tid; s2 < n_slots; s2 += NT) {
							const uint32_t key = t_keys[s2];
							if (key == 0xFFFFFFFFu) continue;
							const uint32_t v = t_votes[s2];
							pmx = max(pmx, (int) max(v & 0xFFFFu, v >> 16));
							pmxb = max(pmxb, (int) ((v & 0xFFFFu) + (v >> 16)));
							const uint32_t at = atomicAdd(&s_nent, 1u);
							if (at < ent_cap) { my_ent[2u * at] = key; my_ent[2u * at + 1u] = v; }
						}
					}
					__syncthreads();
					if (!s_fail || parts >= parts_now) break;   // (block-uniform)
					parts = min(parts_now, parts * 2u);
					__syncthreads();
					if (tid == 0) { s_fail = 0; atomicAdd(&ctl[kCsqRestart], 1u); }
					}
					mark(5);
					if (dg) { atomicAdd(&diag[8], 1ull); atomicAdd(&diag[9], (unsigned long long) H); atomicAdd(&diag[12], (unsigned long long) parts); }
					const uint32_t n_ent = s_nent;
					if (s_fail || n_ent > ent_cap) { if (diag && tid == 0) atomicAdd(&diag[15], 1ull); if (wv == 0) { cs_enqueue(A, read, lane, R); if (lane == 0) atomicAdd(&ctl[kCsqSentOn], 1u); } continue; }
					pmx = wave_reduce_max(pmx); pmxb = wave_reduce_max(pmxb);
					if (lane == 0) { s_mx[wv] = (uint32_t) pmx; s_mxb[wv] = (uint32_t) pmxb; }
					__syncthreads();
					pmx = 0; pmxb = 0;
#pragma unroll
					for (int w2 = 0; w2 < NW; ++w2) { pmx = max(pmx, (int) s_mx[w2]); pmxb = max(pmxb, (int) s_mxb[w2]); }
					const float max_hit_p = (float) pmx;
					const float thresh_p = fmaxf(A.kmer_min, max_hit_p * A.sensitivity);
					if (!((float) (T - 1u) < thresh_p)) { if (diag && tid == 0) atomicAdd(&diag[15], 1ull << 32); if (wv == 0) { cs_enqueue(A, read, lane, R); if (lane == 0) atomicAdd(&ctl[kCsqSentOn], 1u); } continue; }   // bins below T could reach the threshold (a second pass: T was forced as low as the first pass's maximum asks for; the table did not take it)
					const uint32_t region_p = (uint32_t) read & (kCsRegions - 1);
					if (tid == 0 && A.counters) {
						atomicAdd(&A.counters[region_p * kCsCursorStride], (unsigned long long) R.n_valid);
						atomicAdd(&A.counters[region_p * kCsCursorStride + 1], (unsigned long long) H);
					}
					uint32_t cnt_p = 0;
					for (uint32_t x = tid; x < n_ent; x += NT) { const uint32_t v = my_ent[2u * x + 1u]; cnt_p += ((float) (v & 0xFFFFu) >= thresh_p) + ((float) (v >> 16) >= thresh_p); }
					const uint32_t incl_p = wave_inclusive_scan(cnt_p, lane);
					if (lane == 63) s_wtot[wv] = incl_p;
					__syncthreads();
					uint32_t before_p = incl_p - cnt_p, total_p = 0;
#pragma unroll
					for (int w2 = 0; w2 < NW; ++w2) { if (w2 < wv) before_p += s_wtot[w2]; total_p += s_wtot[w2]; }
					if ((int64_t) total_p >= (int64_t) A.max_cmrs) total_p = 0;  // "if (index < maxScores) AllocScores" (CS.cpp:308-310)
					const bool fixed_p = A.fixed_base != 0u && total_p <= (uint32_t) kCsFixedSlots;
					if (tid == 0) {
						unsigned long long base = 0;
						if (!fixed_p) {
							base = total_p ? atomicAdd(&A.out_total[region_p * kCsCursorStride], (unsigned long long) total_p) : 0ull;
							if (base + total_p > A.out_capacity) { atomicExch(&A.status[0], 1u); }
						}
						s_base = base;
						A.cand_base[read] = fixed_p ? A.fixed_base + (uint32_t) read * (uint32_t) kCsFixedSlots : (uint32_t) (region_p * A.out_capacity + base);
						A.cand_count[read] = total_p;
						A.max_votes[read] = max_hit_p;
						if (A.max_both) A.max_both[read] = (float) pmxb;
						A.read_len[read] = (uint16_t) R.L;
						if (A.counters && total_p) atomicAdd(&A.counters[region_p * kCsCursorStride + 2], (unsigned long long) total_p);
					}
					__syncthreads();
					if (total_p != 0u) {
						uint32_t w = 0;
						bool room_ok = true;
						if (fixed_p) w = A.fixed_base + (uint32_t) read * (uint32_t) kCsFixedSlots + before_p;
						else { const unsigned long long base = s_base; room_ok = base + total_p <= A.out_capacity; w = (uint32_t) (region_p * A.out_capacity + base) + before_p; }
						const uint32_t centre_p = A.bin_shift > 0 ? (1u << (A.bin_shift - 1)) : 0u;  // ResolveBin, CS.h:170-175
						if (room_ok) for (uint32_t x = tid; x < n_ent; x += NT) {
							const uint32_t key = my_ent[2u * x], v = my_ent[2u * x + 1u];
							const uint32_t f = v & 0xFFFFu, r = v >> 16;
							const uint32_t loc = (key << A.bin_shift) + centre_p;
							if ((float) f >= thresh_p) { A.out_loc[w] = loc; A.out_sv[w] = f << 1; ++w; }
							if ((float) r >= thresh_p) { A.out_loc[w] = loc; A.out_sv[w] = (r << 1) | 1u; ++w; }
						}
					}
					mark(6);
					continue;
				}
			}
		} else {
			sweep([&](const uint32_t (&pos8)[8], uint32_t cn, uint32_t correction, bool rev, uint32_t) {
#pragma unroll
				for (int j = 0; j < kCsSeg; ++j) if ((uint32_t) j < cn) insert((pos8[j] - correction) >> A.bin_shift, rev);
			});
		}
		__syncthreads();
		mark(5);
		if (dg) { atomicAdd(&diag[8], 1ull); atomicAdd(&diag[9], (unsigned long long) H); if (H > cap && s_direct) atomicAdd(&diag[10], 1ull); }
		if (failed || s_fail) {
			if (!failed && !T_force && max_parts > 1u && T > 1u) { if (tid == 0) { s_retry_ix = item_ix; s_retry_T = T; atomicAdd(&ctl[kCsqSecond], 1u); } if (dg) atomicAdd(&diag[13], 1ull); continue; }   // the table overflowed: the same T in several passes
			if (diag && tid == 0) { if (failed && why <= 2u) atomicAdd(&diag[14], why == 2u ? 1ull << 32 : 1ull); else atomicAdd(&diag[15], 1ull); }
			if (wv == 0) { cs_enqueue(A, read, lane, R); if (lane == 0) atomicAdd(&ctl[kCsqSentOn], 1u); }
			continue;
		}
		// the table: maximum, candidates (cs_global_kernel's order: thread t owns the slots of lane class t mod 64 in the (t / 64)-th share)
		const uint32_t per_class = n_slots >> 6;
		const uint32_t j0 = (uint32_t) ((unsigned long long) per_class * (unsigned) wv / (unsigned) NW), j1 = (uint32_t) ((unsigned long long) per_class * (unsigned) (wv + 1) / (unsigned) NW);
		int mx = 0, mxb = 0;
		for (uint32_t j = j0; j < j1; ++j) {
			const uint32_t v = t_votes[(j << 6) + (uint32_t) lane];
			mx = max(mx, (int) max(v & 0xFFFFu, v >> 16));
			mxb = max(mxb, (int) ((v & 0xFFFFu) + (v >> 16)));
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
		if (T > 1u && !((float) (T - 1u) < thresh)) {   // bins outside the table could reach the threshold
			if (!T_force && max_parts > 1u) { if (tid == 0) { s_retry_ix = item_ix; s_retry_T = max(2u, (uint32_t) ceilf(thresh)); atomicAdd(&ctl[kCsqSecond], 1u); } if (dg) atomicAdd(&diag[13], 1ull); continue; }   // once more, from the T this maximum asks for
			if (diag && tid == 0) atomicAdd(&diag[15], 1ull << 32);
			if (wv == 0) { cs_enqueue(A, read, lane, R); if (lane == 0) atomicAdd(&ctl[kCsqSentOn], 1u); }
			continue;
		}
		const uint32_t region = (uint32_t) read & (kCsRegions - 1);
		if (tid == 0 && A.counters) {
			atomicAdd(&A.counters[region * kCsCursorStride], (unsigned long long) R.n_valid);
			atomicAdd(&A.counters[region * kCsCursorStride + 1], (unsigned long long) H);
		}
		uint32_t count = 0;
		for (uint32_t j = j0; j < j1; ++j) {
			const uint32_t s2 = (j << 6) + (uint32_t) lane;
			if (t_keys[s2] != 0xFFFFFFFFu) {
				const uint32_t v = t_votes[s2];
				count += ((float) (v & 0xFFFFu) >= thresh) + ((float) (v >> 16) >= thresh);
			}
		}
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
		if (total == 0) continue;
		uint32_t w;
		if (fixed) w = A.fixed_base + (uint32_t) read * (uint32_t) kCsFixedSlots + before;
		else {
			const unsigned long long base = s_base;
			if (base + total > A.out_capacity) continue;
			w = (uint32_t) (region * A.out_capacity + base) + before;
		}
		const uint32_t centre = A.bin_shift > 0 ? (1u << (A.bin_shift - 1)) : 0u;  // ResolveBin, CS.h:170-175
		for (uint32_t j = j0; j < j1; ++j) {
			const uint32_t s2 = (j << 6) + (uint32_t) lane;
			const uint32_t key = t_keys[s2];
			if (key != 0xFFFFFFFFu) {
				const uint32_t v = t_votes[s2];
				const uint32_t f = v & 0xFFFFu, r = v >> 16;
				const uint32_t loc = (key << A.bin_shift) + centre;
				if ((float) f >= thresh) { A.out_loc[w] = loc; A.out_sv[w] = f << 1; ++w; }
				if ((float) r >= thresh) { A.out_loc[w] = loc; A.out_sv[w] = (r << 1) | 1u; ++w; }
			}
		}
		mark(6);
	}
}

}  // namespace ngm
