// cs_device.h -- candidate search on the GPU: k-mer lookups into the HBM-resident index and binned
// diagonal votes.  Replaces NextGenMap's CS::PrefixIteration (read side, src/CSstatic.cpp:26-76),
// CompactPrefixTable::GetRefEntry (src/PrefixTable.cpp:750-817), CS::PrefixSearch / AddLocationStd
// (src/CS.cpp:114-213) and CS::CollectResultsStd (src/CS.cpp:263-313).
//
// Decomposition (one 64-lane wave per read):
//   1. lanes own k-mer start positions; each valid k-mer costs two 8-byte index reads (forward k-mer,
//      reverse-complement k-mer) -> up to 2*(L-k+1) position lists, skipped when fwd+rev >= max_kfreq;
//   2. the lists are flattened: a lane takes 8 consecutive hits (one binary search over the prefix sums kept
//      in LDS, eight independent loads), a wave 512 consecutive hits -> coalesced gathers with 8 loads in
//      flight per lane;
//   3. votes.  Against a GRCh38-sized index a 150 bp read collects ~4 300 hits, almost all of them single
//      background hits that can never reach the threshold.  FAST path: two bit planes in LDS record "bin seen"
//      and "bin seen twice" (atomicOr, no probing); a second sweep over the hits inserts only those whose bin was
//      seen twice into a small exact table (key = bin, value = fwd votes | rev votes << 16).  This is exact
//      whenever the final threshold exceeds 1 vote -- every candidate bin then has >= 2 votes, and all votes of
//      a bin share one bit -- and needs ~20 KB of LDS instead of 64 KB, so 3x more reads are in flight per CU.
//      Reads for which it is not provably exact (threshold <= 1, or the small table fills up) are queued and
//      re-run by the EXACT path: every hit goes into an open-addressing table in LDS, or in global memory when
//      the read has more hits than the largest LDS table holds;
//   4. max votes -> threshold max(kmer_min, max * sensitivity), in float exactly as the reference computes it;
//      table entries at or above it are the candidate mapping regions (bin centre, strand, votes).
// The reference walks hits sequentially and remembers the order in which bins first crossed the running
// threshold; the SET of candidates does not depend on that order (the running threshold never exceeds the final
// one), only ties between equally scoring loci do.  The selection stage breaks such ties by position.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ngm {

enum { kCsFast = 0, kCsExactLds = 1, kCsExactGlobal = 2 };

struct CsArgs {
	const uint8_t *reads;       // n rows of q bytes
	const uint32_t *read_list;  // optional: workgroup i handles read read_list[i] (re-run of queued reads)
	int n;
	int q;
	int k;
	int bin_shift;
	int max_kfreq;
	float sensitivity;
	float kmer_min;
	int max_cmrs;
	const uint2 *index;
	const uint32_t *positions;
	int lists_cap;          // LDS capacity for lists (>= 2*(q-k+1))
	int log2_slots;         // exact table slots (power of two) in LDS
	int log2_bits;          // FAST: bits per plane
	uint32_t hit_cap;       // reads with more hits than this are queued for the next path
	// outputs
	uint16_t *read_len;     // [n]
	uint32_t *cand_base;    // [n]
	uint32_t *cand_count;   // [n]
	float *max_votes;       // [n]
	float *max_both;        // [n] optional: max over bins of forward + reverse votes (sensitivity estimation)
	uint32_t *out_loc;      // candidate bin centres (concatenated coordinates)
	uint32_t *out_sv;       // votes << 1 | strand
	unsigned long long *out_total;  // allocation cursor
	unsigned long long out_capacity;
	uint32_t *status;       // [0] output overflow flag, [1] number of queued reads
	unsigned long long *counters;  // [0] k-mers looked up, [1] hits voted (algorithmic-bytes accounting)
	uint32_t *ovf_read;     // [n] queue written by this pass
	uint32_t *ovf_hits;     // [n]
	// kCsExactGlobal
	const uint64_t *ovf_table_off;  // per queued read: offset (in slots) into gtable_*
	const uint32_t *ovf_log2;       // per queued read: log2 slots
	uint32_t *gtable_keys;
	uint32_t *gtable_votes;
};

__device__ __forceinline__ uint32_t cs_revcomp(uint32_t prefix, int k) {  // PrefixTable.cpp:94-108
	const int shift = 32 - 2 * k;
	uint32_t c = (prefix ^ 0xAAAAAAAAu) << shift;
	c = (c & 0xFFFF0000u) >> 16 | (c & 0x0000FFFFu) << 16;
	c = (c & 0xFF00FF00u) >> 8 | (c & 0x00FF00FFu) << 8;
	c = (c & 0xF0F0F0F0u) >> 4 | (c & 0x0F0F0F0Fu) << 4;
	c = (c & 0xCCCCCCCCu) >> 2 | (c & 0x33333333u) << 2;
	return c;
}

__device__ __forceinline__ int wave_reduce_max(int v) {
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
	return v;
}
__device__ __forceinline__ int wave_reduce_min(int v) {
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
	return v;
}
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v, int lane) {
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		const uint32_t t = __shfl_up(v, o);
		if (lane >= o) v += t;
	}
	return v;
}

// table reads after the voting phase: a global-memory table was updated by L2 atomics, so bypass L1
template <int MODE>
__device__ __forceinline__ uint32_t cs_tload(const uint32_t *p) {
	if (MODE == kCsExactGlobal) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	return *p;
}

// Enumerates the read's hits, HPL consecutive hits per lane per trip (HPL independent loads in flight per lane,
// a wave covers 64*HPL consecutive hits); f(position, list index).
constexpr int kCsHitsPerLane = 8;
template <typename F>
__device__ __forceinline__ void cs_for_each_hit(const uint32_t *__restrict__ positions, const uint32_t *l_start, const uint32_t *l_pref,
		int n_lists, uint32_t H, int lane, F f) {
	constexpr int HPL = kCsHitsPerLane;
	for (uint32_t h0 = (uint32_t) lane * HPL; h0 < H; h0 += 64u * HPL) {
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

template <int MODE>
__global__ __launch_bounds__(64) void cs_kernel(CsArgs A) {
	extern __shared__ __attribute__((aligned(16))) uint32_t cs_lds[];
	__shared__ uint32_t s_flags[2];  // [0] distinct bins in the small table (FAST), [1] abort
	const int lane = threadIdx.x;
	const int item = blockIdx.x;
	const int read = A.read_list ? (int) A.read_list[item] : item;
	const int k = A.k;
	uint32_t *l_start = cs_lds;                        // [lists_cap]
	uint32_t *l_pref = cs_lds + A.lists_cap;           // [lists_cap + 1]
	uint8_t *l_code = (uint8_t *) (l_pref + A.lists_cap + 1);  // [q rounded up to 4]
	const int code_words = (A.q + 3) / 4;
	uint32_t *plane1 = (uint32_t *) l_code + code_words;  // FAST only
	const uint32_t plane_words = (MODE == kCsFast) ? (1u << (A.log2_bits - 5)) : 0u;
	uint32_t *plane2 = plane1 + plane_words;
	uint32_t *t_keys, *t_votes;
	int log2_slots;
	if (MODE == kCsExactGlobal) {
		log2_slots = (int) A.ovf_log2[item];
		t_keys = A.gtable_keys + A.ovf_table_off[item];
		t_votes = A.gtable_votes + A.ovf_table_off[item];
	} else {
		log2_slots = A.log2_slots;
		t_keys = plane2 + plane_words;
		t_votes = t_keys + (1u << log2_slots);
	}
	uint32_t n_slots = 1u << log2_slots;
	if (lane < 2) s_flags[lane] = 0;

	// ---- 1. read -> 2-bit codes (A0 C1 T2 G3, CSstatic.cpp:20-22), N = 4, past the end = 255 ----------------
	const uint8_t *rp = A.reads + (size_t) read * A.q;
	int first_nul = A.q;
	for (int i = lane; i < A.q; i += 64) {
		const uint32_t ch = rp[i];
		uint8_t code;
		if (ch == 0) { code = 255; first_nul = min(first_nul, i); }
		else if (ch == 'N') code = 4;
		else code = (uint8_t) ((ch >> 1) & 3u);
		l_code[i] = code;
	}
	const int L = wave_reduce_min(first_nul);  // MappedRead::length
	__syncthreads();

	// ---- 2. k-mers and their two position lists (lane = k-mer, lists 2p = forward, 2p+1 = reverse complement) --
	const int n_kmers = L - k + 1;
	const int n_lists = n_kmers > 0 ? 2 * n_kmers : 0;
	uint32_t carry = 0, n_valid = 0;
	for (int base = 0; base < n_kmers; base += 64) {
		const int p = base + lane;
		uint32_t cf = 0, cr = 0, sf = 0, sr = 0;
		bool counted = false;
		if (p < n_kmers) {
			bool valid = true;
			uint32_t kmer = 0;
			for (int j = 0; j < k; ++j) {
				const uint32_t c = l_code[p + j];
				valid = valid && (c < 4);
				kmer = (kmer << 2) | (c & 3u);
			}
			// CSstatic.cpp:30-41: a k-mer that starts right after a restart-position N run and ends exactly at
			// the read end is never visited
			if (valid && p + k == L && p >= 1 && l_code[p - 1] == 4 && (p == 1 || l_code[p - 2] == 4)) valid = false;
			if (valid) {
				const uint2 ef = A.index[kmer];
				const uint2 er = A.index[cs_revcomp(kmer, k)];
				if ((int) (ef.y + er.y) < A.max_kfreq) { cf = ef.y; sf = ef.x; cr = er.y; sr = er.x; }  // CS.cpp:122
			}
			counted = valid;
		}
		n_valid += __popcll(__ballot(counted));
		const uint32_t both = cf + cr;
		const uint32_t incl = wave_inclusive_scan(both, lane);
		if (p < n_kmers) {
			const uint32_t b0 = carry + incl - both;
			l_start[2 * p] = sf; l_pref[2 * p] = b0;
			l_start[2 * p + 1] = sr; l_pref[2 * p + 1] = b0 + cf;
		}
		carry += __shfl(incl, 63);
	}
	if (lane == 0) l_pref[n_lists] = carry;
	const uint32_t H = carry;

	auto enqueue = [&]() {  // hand the read to the next, more general path
		if (lane == 0) {
			const uint32_t slot = atomicAdd(&A.status[1], 1u);
			A.ovf_read[slot] = (uint32_t) read;
			A.ovf_hits[slot] = H;
			A.read_len[read] = (uint16_t) L;
		}
	};
	if (MODE != kCsExactGlobal && H > A.hit_cap) { enqueue(); return; }

	// ---- 3. votes ---------------------------------------------------------------------------------------
	auto bin_of = [&](uint32_t pos, int li) -> uint32_t {
		const int p = li >> 1;
		const uint32_t correction = (li & 1) ? (uint32_t) (L - (p + k)) : (uint32_t) p;  // CS.cpp:140-142
		return (pos - correction) >> A.bin_shift;
	};
	if (MODE == kCsExactLds) {
		// the table in use is sized to this read's hit count (power of two >= 2H): clearing and scanning it cost
		// what the read needs, not what the allocation allows
		int need = 8;
		while ((1u << need) < 2u * H && need < log2_slots) ++need;
		log2_slots = need;
		n_slots = 1u << log2_slots;
		t_votes = t_keys + n_slots;
	}
	for (uint32_t s = lane; s < n_slots; s += 64) { t_keys[s] = 0xFFFFFFFFu; t_votes[s] = 0; }
	if (MODE == kCsFast) for (uint32_t s = lane; s < 2 * plane_words; s += 64) plane1[s] = 0;
	__syncthreads();
	if (MODE == kCsExactGlobal) __threadfence_block();

	auto insert = [&](uint32_t bin, bool rev) -> bool {
		uint32_t slot = (bin * 2654435761u) >> (32 - log2_slots);
		for (;;) {
			const uint32_t prev = atomicCAS(&t_keys[slot], 0xFFFFFFFFu, bin);
			if (prev == bin) break;
			if (prev == 0xFFFFFFFFu) {
				if (MODE == kCsFast && atomicAdd(&s_flags[0], 1u) > (n_slots * 3u) / 4u) return false;
				break;
			}
			slot = (slot + 1) & (n_slots - 1);
		}
		atomicAdd(&t_votes[slot], rev ? 0x10000u : 1u);
		return true;
	};

	if (MODE == kCsFast) {
		const int sh = 32 - A.log2_bits;
		cs_for_each_hit(A.positions, l_start, l_pref, n_lists, H, lane, [&](uint32_t pos, int li) {
			const uint32_t b = (bin_of(pos, li) * 0x9E3779B1u) >> sh;
			const uint32_t m = 1u << (b & 31);
			const uint32_t old = atomicOr(&plane1[b >> 5], m);
			if (old & m) atomicOr(&plane2[b >> 5], m);
		});
		__syncthreads();
		cs_for_each_hit(A.positions, l_start, l_pref, n_lists, H, lane, [&](uint32_t pos, int li) {
			const uint32_t bin = bin_of(pos, li);
			const uint32_t b = (bin * 0x9E3779B1u) >> sh;
			if ((plane2[b >> 5] >> (b & 31)) & 1u) {
				if (!insert(bin, li & 1)) s_flags[1] = 1;
			}
		});
	} else {
		cs_for_each_hit(A.positions, l_start, l_pref, n_lists, H, lane, [&](uint32_t pos, int li) { insert(bin_of(pos, li), li & 1); });
	}
	__syncthreads();
	if (MODE == kCsExactGlobal) __threadfence_block();
	if (MODE == kCsFast && s_flags[1]) { enqueue(); return; }  // small table full: not provably exact

	// ---- 4. threshold and candidates (CS.cpp:201-205, :263-313) -------------------------------------------
	int mx = 0, mxb = 0;
	for (uint32_t s = lane; s < n_slots; s += 64) {
		const uint32_t v = cs_tload<MODE>(&t_votes[s]);
		mx = max(mx, (int) max(v & 0xFFFFu, v >> 16));
		mxb = max(mxb, (int) ((v & 0xFFFFu) + (v >> 16)));
	}
	mx = wave_reduce_max(mx);
	mxb = wave_reduce_max(mxb);
	if (MODE == kCsFast && H > 0 && mx < 2) mx = 1;  // only single votes survived the filter: the true maximum is 1
	if (MODE == kCsFast && H > 0 && mxb < 2) mxb = 1;
	const float max_hit = (float) mx;
	const float thresh = fmaxf(A.kmer_min, max_hit * A.sensitivity);
	// the filter dropped bins with a single vote: exact only if those cannot be candidates
	if (MODE == kCsFast && H > 0 && !(thresh > 1.0f)) { enqueue(); return; }
	if (lane == 0 && A.counters) { atomicAdd(&A.counters[0], (unsigned long long) n_valid); atomicAdd(&A.counters[1], (unsigned long long) H); }
	uint32_t count = 0;
	for (uint32_t s = lane; s < n_slots; s += 64) {
		if (cs_tload<MODE>(&t_keys[s]) != 0xFFFFFFFFu) {
			const uint32_t v = cs_tload<MODE>(&t_votes[s]);
			count += ((float) (v & 0xFFFFu) >= thresh) + ((float) (v >> 16) >= thresh);
		}
	}
	const uint32_t incl = wave_inclusive_scan(count, lane);
	uint32_t total = __shfl(incl, 63);
	if ((int64_t) total >= (int64_t) A.max_cmrs) total = 0;  // "if (index < maxScores) AllocScores" (CS.cpp:308-310)
	unsigned long long base = 0;
	if (lane == 0) {
		base = total ? atomicAdd(A.out_total, (unsigned long long) total) : 0ull;
		if (base + total > A.out_capacity) { atomicExch(&A.status[0], 1u); }
		A.cand_base[read] = (uint32_t) base;
		A.cand_count[read] = total;
		A.max_votes[read] = max_hit;
		if (A.max_both) A.max_both[read] = (float) mxb;
		A.read_len[read] = (uint16_t) L;
	}
	base = __shfl((uint32_t) base, 0) | ((unsigned long long) __shfl((uint32_t) (base >> 32), 0) << 32);
	if (total == 0 || base + total > A.out_capacity) return;
	uint32_t w = (uint32_t) base + (incl - count);
	const uint32_t centre = A.bin_shift > 0 ? (1u << (A.bin_shift - 1)) : 0u;  // ResolveBin, CS.h:170-175
	for (uint32_t s = lane; s < n_slots; s += 64) {
		const uint32_t key = cs_tload<MODE>(&t_keys[s]);
		if (key != 0xFFFFFFFFu) {
			const uint32_t v = cs_tload<MODE>(&t_votes[s]);
			const uint32_t f = v & 0xFFFFu, r = v >> 16;
			const uint32_t loc = (key << A.bin_shift) + centre;
			if ((float) f >= thresh) { A.out_loc[w] = loc; A.out_sv[w] = f << 1; ++w; }
			if ((float) r >= thresh) { A.out_loc[w] = loc; A.out_sv[w] = (r << 1) | 1u; ++w; }
		}
	}
}

}  // namespace ngm
