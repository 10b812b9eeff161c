// cs_slam_device.h -- the WEIGHTED candidate search of `--slam-seq` with bit 2 set (round 4).
//
// Reference: CS::PrefixMutateSearch's slamSeq branch and PrefixMutateSearchSlamSeq (src/CS.cpp:57-92), PrefixSearch (:114-160),
// AddLocationStd (:162-213), CollectResultsStd (:263-313).  Every read k-mer is looked up as it is (weight 1) and then once per
// convertible base with that base converted alone (C > T, second mates G > A; weight 1 / (convertible bases + 1), a float).  The
// votes of a bin are FLOAT sums in the order the hits arrive (fScore += freq), the running maximum and the rList threshold are
// floats too -- and float addition is not associative, so the only way to the reference's bits is the reference's order: this
// kernel IS the sequential loop of AddLocationStd, 64 hits per trip.  Per trip: the hits of 64 consecutive times; lanes that hit
// the same (bin, strand) add their weights one after the other in lane order (rounds), the running maximum is an inclusive prefix
// maximum over the lanes (positive floats order like their bit patterns), a bin enters rList at its first hit with
// score >= maximum * sensitivity.  The candidates leave in rList order, so their index inside the read IS the reference's
// candidate order (no separate order replay for these runs).
//
// One wave per read; the read's table (key, forward sum, reverse sum, rList rank per slot) and rList live in a slice of global
// memory: persistent workgroups own a slice each and draw reads from a counter; a read whose hits outgrow the slice is queued and
// re-run with a slice of its own.  Every access to a slice comes from one wave, so workgroup-scope ordering is enough.
#pragma once

#include "cs_device.h"

namespace ngm {

constexpr uint32_t kCsSlamEmpty = 0xFFFFFFFFu;

// words a read with `hits` index hits needs: 4 per slot of a table with 2^l >= 1.5 * hits slots, plus rList
__host__ __device__ inline uint32_t cs_slam_log2_slots(uint32_t hits) {
	uint32_t l = 6;
	while ((1ull << l) * 2ull < 3ull * (unsigned long long) hits && l < 28) ++l;
	return l;
}
__host__ __device__ inline unsigned long long cs_slam_words(uint32_t hits) { return (4ull << cs_slam_log2_slots(hits)) + hits + 64ull; }

__global__ __launch_bounds__(64) void cs_slam_kernel(CsArgs A) {
	extern __shared__ __attribute__((aligned(16))) uint32_t cs_lds[];
	const int lane = threadIdx.x;
	const int k = A.k;
	uint32_t *l_start = cs_lds;                        // [lists_cap]
	uint32_t *l_pref = cs_lds + A.lists_cap;           // [lists_cap + 1]
	uint8_t *l_code = (uint8_t *) (l_pref + A.lists_cap + 1);
	uint32_t *l_vbase = (uint32_t *) l_code + (A.q + 3) / 4;   // [q + 1]
	uint16_t *l_vpos = (uint16_t *) (l_vbase + A.q + 1);       // [kCsBsChunk]: read position | weight divisor << 10
	const bool persistent = A.read_list == nullptr;
	const unsigned long long lanes_below = (1ull << lane) - 1ull;
	auto wave_sync = [] { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_wave_barrier(); };
	for (int item = (int) blockIdx.x;; ) {
		int read;
		if (persistent) {
			read = item;
			if (read >= A.n) break;
		} else {
			if (item != (int) blockIdx.x) break;
			read = (int) A.read_list[item];
		}
		uint32_t *slice = A.gtable_keys + (persistent ? (unsigned long long) blockIdx.x * A.slam_slice_words : A.ovf_table_off[blockIdx.x]);
		__syncthreads();   // (the LDS rows are reused)
		const CsBsRead B = cs_bs_scan(A, read, lane, l_code, l_vbase);
		uint32_t Hs = 0;   // counting pass: the table is sized from the read's hits
		for (uint32_t v0 = 0; v0 < B.V; v0 += kCsBsChunk) {
			Hs += cs_bs_chunk<false>(A, B, lane, l_code, l_vbase, v0, min(B.V, v0 + (uint32_t) kCsBsChunk), 0u, l_start, l_pref, l_vpos);
			__syncthreads();
		}
		CsRead R;
		R.L = B.L; R.n_lists = 0; R.H = Hs; R.n_valid = B.n_valid; R.n_items = 0;
		const int L = B.L;
		// next read of a persistent workgroup (drawn now: the atomic's latency hides behind the read)
		int next_item = item + 1;
		if (persistent) {
			uint32_t d = 0;
			if (lane == 0) d = atomicAdd(&A.status[2], 1u);
			next_item = (int) gridDim.x + (int) wave_first(d);
		}
		if (persistent && cs_slam_words(Hs) > A.slam_slice_words) { cs_enqueue(A, read, lane, R); item = next_item; continue; }
		const uint32_t log2_slots = persistent ? cs_slam_log2_slots(Hs) : A.ovf_log2[blockIdx.x];
		const uint32_t n_slots = 1u << log2_slots;
		uint32_t *t_key = slice, *t_f = slice + n_slots, *t_r = t_f + n_slots, *t_rank = t_r + n_slots, *rlist = t_rank + n_slots;
		for (uint32_t s = lane; s < n_slots; s += 64) { t_key[s] = kCsSlamEmpty; t_f[s] = 0u; t_r[s] = 0u; t_rank[s] = kCsSlamEmpty; }
		wave_sync();
		float max_hit = 0.0f;
		uint32_t n_r = 0;
		for (uint32_t v0 = 0; v0 < B.V; v0 += kCsBsChunk) {
			const uint32_t v1 = min(B.V, v0 + (uint32_t) kCsBsChunk);
			const uint32_t hc = cs_bs_chunk<false>(A, B, lane, l_code, l_vbase, v0, v1, 0u, l_start, l_pref, l_vpos);
			__syncthreads();
			const int n_lists = (int) (2u * (v1 - v0));
			for (uint32_t t0 = 0; t0 < hc; t0 += 64) {
				const uint32_t t = t0 + (uint32_t) lane;
				const bool act = t < hc;
				uint32_t bin = 0, slot = 0;
				bool rev = false;
				float w = 0.0f;
				if (act) {
					int lo = 0, hi = n_lists;   // the list with l_pref[li] <= t < l_pref[li + 1] (never an empty one)
					while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (l_pref[mid] <= t) lo = mid; else hi = mid; }
					while (l_pref[lo + 1] <= t) ++lo;
					const uint32_t pos = A.positions[l_start[lo] + (t - l_pref[lo])];
					const uint32_t vp = l_vpos[lo >> 1];
					const int p = (int) (vp & 0x3FFu);
					const uint32_t wdiv = vp >> 10;
					rev = (lo & 1) != 0;
					w = wdiv <= 1u ? 1.0f : 1.0f / (float) wdiv;   // CS.cpp:133-138: 1.0f / m_CurrentMutLocs
					const uint32_t correction = rev ? (uint32_t) (L - (p + k)) : (uint32_t) p;  // CS.cpp:140-142
					bin = (pos - correction) >> A.bin_shift;
					slot = (bin * 2654435761u) >> (32u - log2_slots);
					for (;;) {
						const uint32_t prev = atomicCAS(&t_key[slot], kCsSlamEmpty, bin);
						if (prev == bin || prev == kCsSlamEmpty) break;
						slot = (slot + 1u) & (n_slots - 1u);
					}
				}
				wave_sync();
				// lanes of this trip on the same bin / the same (bin, strand)
				const unsigned long long active = __ballot(act);
				unsigned long long same_bin = active;
				for (uint32_t b = 0; b < log2_slots; ++b) {
					const bool bit = (slot >> b) & 1u;
					const unsigned long long bb = __ballot(act && bit);
					same_bin &= bit ? bb : ~bb;
				}
				const unsigned long long rev_lanes = __ballot(act && rev);
				const unsigned long long same_key = same_bin & (rev ? rev_lanes : (active & ~rev_lanes));
				const uint32_t ord = act ? (uint32_t) __popcll(same_key & lanes_below) : 0u;
				const uint32_t rounds = (uint32_t) wave_reduce_max((int) ord) + 1u;
				float score = 0.0f;
				uint32_t *cell = rev ? &t_r[slot] : &t_f[slot];
				for (uint32_t r = 0; r < rounds; ++r) {   // fScore += freq, one hit after the other (CS.cpp:180-194)
					if (act && ord == r) { score = __uint_as_float(*cell) + w; *cell = __float_as_uint(score); }
					wave_sync();
				}
				// running maximum (CS.cpp:197-202) and rList (:205-208), in lane = time order
				const uint32_t mx_bits = max(wave_inclusive_max(act ? __float_as_uint(score) : 0u), __float_as_uint(max_hit));
				const float mx = __uint_as_float(mx_bits);
				const bool enters = act && score >= mx * A.sensitivity && t_rank[slot] == kCsSlamEmpty;
				const unsigned long long entering = __ballot(enters);
				const bool first = enters && (entering & same_bin & lanes_below) == 0ull;
				const unsigned long long firsts = __ballot(first);
				if (first) { const uint32_t rk = n_r + (uint32_t) __popcll(firsts & lanes_below); t_rank[slot] = rk; rlist[rk] = slot; }
				n_r += (uint32_t) __popcll(firsts);
				max_hit = __uint_as_float(wave_last(mx_bits));
				wave_sync();
			}
			__syncthreads();
		}
		// CollectResultsStd (CS.cpp:263-313): rList in order, forward before reverse of an entry
		const float thresh = fmaxf(A.kmer_min, max_hit * A.sensitivity);
		uint32_t total = 0;
		for (uint32_t i0 = 0; i0 < n_r; i0 += 64) {
			const uint32_t i = i0 + (uint32_t) lane;
			uint32_t c = 0;
			if (i < n_r) { const uint32_t s = rlist[i]; c = (__uint_as_float(t_f[s]) >= thresh) + (__uint_as_float(t_r[s]) >= thresh); }
			total += wave_last(wave_inclusive_scan(c, lane));
		}
		if ((int64_t) total >= (int64_t) A.max_cmrs) total = 0;  // "if (index < maxScores) AllocScores" (CS.cpp:308-310)
		const uint32_t region = (uint32_t) read & (kCsRegions - 1);
		const bool fixed = A.fixed_base != 0u && total <= (uint32_t) kCsFixedSlots;
		unsigned long long base = 0;
		if (lane == 0) {
			if (!fixed) {
				base = total ? atomicAdd(&A.out_total[region * kCsCursorStride], (unsigned long long) total) : 0ull;
				if (base + total > A.out_capacity) atomicExch(&A.status[0], 1u);
			}
			A.cand_base[read] = fixed ? A.fixed_base + (uint32_t) read * (uint32_t) kCsFixedSlots : (uint32_t) (region * A.out_capacity + base);
			A.cand_count[read] = total;
			A.max_votes[read] = max_hit;
			if (A.max_both) A.max_both[read] = max_hit;
			A.read_len[read] = (uint16_t) B.L;
			if (A.counters) {
				atomicAdd(&A.counters[region * kCsCursorStride], (unsigned long long) B.n_valid);
				atomicAdd(&A.counters[region * kCsCursorStride + 1], (unsigned long long) Hs);
				if (total) atomicAdd(&A.counters[region * kCsCursorStride + 2], (unsigned long long) total);
			}
		}
		if (total != 0u) {
			bool ok = true;
			uint32_t w0;
			if (fixed) w0 = A.fixed_base + (uint32_t) read * (uint32_t) kCsFixedSlots;
			else {
				base = wave_first((uint32_t) base) | ((unsigned long long) wave_first((uint32_t) (base >> 32)) << 32);
				ok = base + total <= A.out_capacity;
				w0 = (uint32_t) (region * A.out_capacity + base);
			}
			if (ok) {
				const uint32_t centre = A.bin_shift > 0 ? (1u << (A.bin_shift - 1)) : 0u;  // ResolveBin, CS.h:170-175
				uint32_t done = 0;
				for (uint32_t i0 = 0; i0 < n_r; i0 += 64) {
					const uint32_t i = i0 + (uint32_t) lane;
					uint32_t c = 0, s = 0;
					float f = 0.f, r = 0.f;
					if (i < n_r) { s = rlist[i]; f = __uint_as_float(t_f[s]); r = __uint_as_float(t_r[s]); c = (f >= thresh) + (r >= thresh); }
					const uint32_t incl = wave_inclusive_scan(c, lane);
					uint32_t w = w0 + done + incl - c;
					if (c) {
						const uint32_t loc = (t_key[s] << A.bin_shift) + centre;
						if (f >= thresh) { A.out_loc[w] = loc; A.out_sv[w] = (uint32_t) (f + 0.5f) << 1; ++w; }   // (the vote count of a candidate is only reported, never compared)
						if (r >= thresh) { A.out_loc[w] = loc; A.out_sv[w] = ((uint32_t) (r + 0.5f) << 1) | 1u; ++w; }
					}
					done += wave_last(incl);
				}
			}
		}
		item = next_item;
	}
}

}  // namespace ngm
