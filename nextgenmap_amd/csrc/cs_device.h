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
constexpr int kCsRegions = 256, kCsCursorStride = 16;

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
	const uint32_t *buckets;   // FAST: one bucket of 1 << bucket_log2_words dwords per k-mer (refindex.h)
	int bucket_log2_words;
	uint32_t ovf_items;        // FAST: LDS capacity for 8-hit segments of lists longer than a bucket
	int lists_cap;          // LDS capacity for lists (>= 2*(q-k+1))
	int log2_slots;         // exact table slots (power of two) in LDS
	int log2_bits;          // (unused by the kernels; kept for diagnostics)
	uint32_t plane_bits;    // FAST: bits of the plane, a power of two
	uint32_t hit_cap;       // reads with more hits than this are queued for the next path
	// outputs
	uint16_t *read_len;     // [n]
	uint32_t *cand_base;    // [n]
	uint32_t *cand_count;   // [n]
	float *max_votes;       // [n]
	float *max_both;        // [n] optional: max over bins of forward + reverse votes (sensitivity estimation)
	uint32_t *out_loc;      // candidate bin centres (concatenated coordinates)
	uint32_t *out_sv;       // votes << 1 | strand
	// candidate output: kCsRegions independent regions of region_capacity entries, each with its own cursor (a single
	// cursor would make every read of the batch wait on one L2 atomic); compacted afterwards in read order
	unsigned long long *out_total;  // [kCsRegions * kCsCursorStride]
	unsigned long long out_capacity;  // entries per region
	uint32_t *status;       // [0] output overflow flag, [1] number of queued reads
	unsigned long long *counters;  // per region, stride kCsCursorStride: [0] k-mers looked up, [1] hits voted (algorithmic-bytes accounting)
	uint32_t *order_scratch;    // cs_order_kernel: time lines in global memory for reads with more hits than LDS holds
	uint32_t order_gcap;        // ... entries per workgroup
	unsigned long long *phase_cycles;  // optional diagnostics (fast path): [0] lists [1] sweep 1 [2] sweep 2 [3] candidates
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

// exclusive prefix sum over the lanes of a small per-lane count (< 2^BITS), bit-sliced: one ballot + mbcnt per bit, no
// cross-lane data movement; total = sum over all lanes (wave-uniform)
template <int BITS>
__device__ __forceinline__ uint32_t wave_prefix_small(uint32_t v, uint32_t &total) {
	uint32_t pre = 0;
	total = 0;
#pragma unroll
	for (int b = 0; b < BITS; ++b) {
		const unsigned long long m = __ballot((v >> b) & 1u);
		pre += __builtin_amdgcn_mbcnt_hi((uint32_t) (m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) m, 0u)) << b;
		total += (uint32_t) __popcll(m) << b;
	}
	return pre;
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
		for (int j = 0; j < HPL; ++j) if (li[j] >= 0) f(pos[j], li[j], h0 + (uint32_t) j);
	}
}

// ---- shared phases -------------------------------------------------------------------------------------------
struct CsRead {
	int L;            // MappedRead::length
	int n_lists;      // 2 * k-mers
	uint32_t H;       // hits of all lists
	uint32_t n_valid; // k-mers looked up
	uint32_t n_items; // ITEMS: 16-hit segments of all lists
};
constexpr int kCsSeg = 8;   // hits per work item of the fast path

// 1. read -> 2-bit codes (A0 C1 T2 G3, CSstatic.cpp:20-22), N = 4, past the end = 255;
// 2. k-mers and their two position lists (lane = k-mer, lists 2p = forward, 2p+1 = reverse complement).
// The index reads of up to four 64-k-mer rounds are issued before any of them is consumed.
// ITEMS (fast path, order replay): l_pref holds per list (length | time of its first hit << 16) instead of the prefix sums
// (both < 65536 for every read these paths accept), and every list is cut into segments of kCsSeg hits, enumerated in
// l_items as (list << 16 | segment).
// work item encodings: 32-bit (list << 16 | segment) or, when every list index is < 512 and every list has at most
// 128 segments, 16-bit (list << 7 | segment), which halves the item list in LDS
template <typename ItemT> struct CsItem;
template <> struct CsItem<uint32_t> {
	static __device__ __forceinline__ uint32_t make(uint32_t li, uint32_t sg) { return (li << 16) | sg; }
	static __device__ __forceinline__ uint32_t list(uint32_t it) { return it >> 16; }
	static __device__ __forceinline__ uint32_t seg(uint32_t it) { return it & 0xFFFFu; }
};
template <> struct CsItem<uint16_t> {
	static __device__ __forceinline__ uint16_t make(uint32_t li, uint32_t sg) { return (uint16_t) ((li << 7) | sg); }
	static __device__ __forceinline__ uint32_t list(uint32_t it) { return it >> 7; }
	static __device__ __forceinline__ uint32_t seg(uint32_t it) { return it & 0x7Fu; }
};

template <bool ITEMS, typename ItemT = uint32_t>
__device__ __forceinline__ CsRead cs_prepare(const CsArgs &A, int read, int lane, uint32_t *l_start, uint32_t *l_pref, uint8_t *l_code,
		ItemT *l_items = nullptr, uint32_t items_cap = 0) {
	const int k = A.k;
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
	CsRead R;
	R.L = wave_reduce_min(first_nul);
	__syncthreads();
	const int L = R.L;
	const int n_kmers = L - k + 1;
	R.n_lists = n_kmers > 0 ? 2 * n_kmers : 0;
	uint32_t carry = 0, n_valid = 0, carry_s = 0;
	constexpr int RB = 4;
	for (int base = 0; base < n_kmers; base += 64 * RB) {
		uint2 ef[RB], er[RB];
		bool valid[RB];
#pragma unroll
		for (int r = 0; r < RB; ++r) {
			const int p = base + r * 64 + lane;
			valid[r] = false;
			ef[r] = make_uint2(0, 0); er[r] = make_uint2(0, 0);
			if (p < n_kmers) {
				bool v = true;
				uint32_t kmer = 0;
				for (int j = 0; j < k; ++j) {
					const uint32_t c = l_code[p + j];
					v = v && (c < 4);
					kmer = (kmer << 2) | (c & 3u);
				}
				// CSstatic.cpp:30-41: a k-mer that starts right after a restart-position N run and ends exactly at
				// the read end is never visited
				if (v && p + k == L && p >= 1 && l_code[p - 1] == 4 && (p == 1 || l_code[p - 2] == 4)) v = false;
				valid[r] = v;
				if (v) { ef[r] = A.index[kmer]; er[r] = A.index[cs_revcomp(kmer, k)]; }
			}
		}
#pragma unroll
		for (int r = 0; r < RB; ++r) {
			if (base + r * 64 >= n_kmers) break;
			const int p = base + r * 64 + lane;
			uint32_t cf = 0, cr = 0, sf = 0, sr = 0;
			if (valid[r] && (int) (ef[r].y + er[r].y) < A.max_kfreq) { cf = ef[r].y; sf = ef[r].x; cr = er[r].y; sr = er[r].x; }  // CS.cpp:122
			n_valid += __popcll(__ballot(valid[r]));
			const uint32_t both = cf + cr;
			const uint32_t incl = wave_inclusive_scan(both, lane);
			if (!ITEMS) {
				if (p < n_kmers) {
					const uint32_t b0 = carry + incl - both;
					l_start[2 * p] = sf; l_pref[2 * p] = b0;
					l_start[2 * p + 1] = sr; l_pref[2 * p + 1] = b0 + cf;
				}
			} else {
				const uint32_t nsf = (cf + kCsSeg - 1) / kCsSeg, nsr = (cr + kCsSeg - 1) / kCsSeg;
				const uint32_t incl_s = wave_inclusive_scan(nsf + nsr, lane);
				if (p < n_kmers) {
					const uint32_t b0 = carry + incl - both;  // time (flattened hit index) of the forward list's first hit
					l_start[2 * p] = sf; l_pref[2 * p] = (cf & 0xFFFFu) | (b0 << 16);
					l_start[2 * p + 1] = sr; l_pref[2 * p + 1] = (cr & 0xFFFFu) | ((b0 + cf) << 16);
					uint32_t o = carry_s + incl_s - (nsf + nsr);
					for (uint32_t sg = 0; sg < nsf; ++sg, ++o) if (o < items_cap) l_items[o] = CsItem<ItemT>::make((uint32_t) (2 * p), sg);
					for (uint32_t sg = 0; sg < nsr; ++sg, ++o) if (o < items_cap) l_items[o] = CsItem<ItemT>::make((uint32_t) (2 * p + 1), sg);
				}
				carry_s += __shfl(incl_s, 63);
			}
			carry += __shfl(incl, 63);
		}
	}
	if (!ITEMS && lane == 0) l_pref[R.n_lists] = carry;
	R.H = carry;
	R.n_valid = n_valid;
	R.n_items = carry_s;
	return R;
}

__device__ __forceinline__ void cs_enqueue(const CsArgs &A, int read, int lane, const CsRead &R) {  // hand the read to the next path
	if (lane == 0) {
		const uint32_t slot = atomicAdd(&A.status[1], 1u);
		A.ovf_read[slot] = (uint32_t) read;
		A.ovf_hits[slot] = R.H;
		A.read_len[read] = (uint16_t) R.L;
	}
}

// 4. threshold and candidates (CS.cpp:201-205, :263-313).  Returns false when the FAST path cannot certify the
// result (final threshold <= 1 vote): the caller queues the read for the exact path.
template <int MODE>
__device__ __forceinline__ bool cs_finish(const CsArgs &A, int read, int lane, const CsRead &R, const uint32_t *t_keys, const uint32_t *t_votes,
		uint32_t n_slots) {
	const uint32_t H = R.H;
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
	if (MODE == kCsFast && H > 0 && !(thresh > 1.0f)) return false;
	const uint32_t region = (uint32_t) read & (kCsRegions - 1);
	if (lane == 0 && A.counters) {
		atomicAdd(&A.counters[region * kCsCursorStride], (unsigned long long) R.n_valid);
		atomicAdd(&A.counters[region * kCsCursorStride + 1], (unsigned long long) H);
	}
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
		base = total ? atomicAdd(&A.out_total[region * kCsCursorStride], (unsigned long long) total) : 0ull;
		if (base + total > A.out_capacity) { atomicExch(&A.status[0], 1u); }
		A.cand_base[read] = (uint32_t) (region * A.out_capacity + base);
		A.cand_count[read] = total;
		A.max_votes[read] = max_hit;
		if (A.max_both) A.max_both[read] = (float) mxb;
		A.read_len[read] = (uint16_t) R.L;
	}
	base = __shfl((uint32_t) base, 0) | ((unsigned long long) __shfl((uint32_t) (base >> 32), 0) << 32);
	if (total == 0 || base + total > A.out_capacity) return true;
	uint32_t w = (uint32_t) (region * A.out_capacity + base) + (incl - count);
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
	return true;
}

// ---- FAST path ---------------------------------------------------------------------------------------------------
// T waves per read (workgroup = 64 T lanes).
// Phase 1: lanes own k-mer start positions and write the bucket number of every list (2p = forward k-mer, 2p+1 = reverse
// complement) to LDS -- no memory access, no dependent index read.
// Sweep 1: the lists are taken in rounds; in a round a wave loads 64/LPB whole buckets, LPB = W/4 lanes x 16 bytes each,
// i.e. ONE aligned request per bucket (the random-request rate, ~50 G/s on MI355X, is the memory-side bound of this
// kernel: profiles/r02_gather_calibration.txt), kCsBucketDepth rounds in flight.  A lane gets 4 bucket words; word 0 of
// the bucket carries the list length, the length of the other strand's list of the same k-mer (CS.cpp:122 needs the
// sum) and the "does not fit" flag, and is broadcast inside the lane group.  Slots past the list end vote with an
// all-zero mask.  A vote is two LDS atomics and no branch: atomicOr into plane 1 ("bin seen"); the returned word tells
// whether the bit was already set, and that bit is OR-ed into plane 2 ("bin seen again").  The bins stay in registers.
// Lists longer than a bucket are collected in LDS and voted from d_positions in 8-hit segments by a run-time loop.
// Sweep 2 (registers + LDS only): every hit whose plane-2 bit is set is appended to its lane's private queue (branch
// free: always store, advance the cursor on a hit) and then inserted into the small exact table (key = bin, value =
// forward | reverse votes).  A bin with >= 2 votes has set its plane-2 bit, so ALL its hits are counted: exact; bins
// with a single vote are dropped (never candidates when the final threshold exceeds 1); bit collisions only cost
// spurious table entries with their exact counts.
constexpr int kCsBucketDepth = 8;      // bucket rounds in flight per wave (one 16-byte load per lane each)
constexpr int kCsOvfLists = 64;        // lists longer than a bucket, per read, that the fast path takes
constexpr uint32_t kCsEmptySlot = 0u;  // register entry of a slot without a hit (a hit is bin | 1 << 30 | strand << 31)

struct __attribute__((packed, aligned(4))) CsU4 { uint32_t x, y, z, w; };

// bucket word 0 (written by fill_buckets_kernel, refindex.cpp)
constexpr uint32_t kCsHdrCountMask = 0x3FFFu;   // bits 0-13 own list length (<= 9900), bits 14-27 the other strand's
constexpr uint32_t kCsHdrOverflow = 0x80000000u;

// LDS words of the fast path: plane 1 first (its word address is a bit field of the bin: no base to add); the lane
// queues of sweep 2 reuse plane 1
__host__ __device__ inline uint32_t cs_fast_lds_words(int q, int lists_cap, uint32_t plane_bits, int log2_slots, uint32_t ovf_items) {
	return (plane_bits >> 5) + (plane_bits >> 7) + (2u << log2_slots) + (uint32_t) ((q + 3) / 4) + (uint32_t) lists_cap + 1u + 2u * kCsOvfLists + ovf_items + 16u;
}

template <int ROUNDS, int T>
__global__ __launch_bounds__(64 * T) void cs_bucket_kernel(CsArgs A) {
	extern __shared__ __attribute__((aligned(16))) uint32_t cs_lds[];
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int read = blockIdx.x;
	const int k = A.k, q = A.q;
	uint32_t *plane1 = cs_lds;                                   // [plane_bits / 32]
	const uint32_t p1_words = A.plane_bits >> 5, p2_words = A.plane_bits >> 7;
	uint32_t *plane2 = plane1 + p1_words;                        // [plane_bits / 128]: a quarter of plane 1, same bit-in-word
	uint32_t *t_keys = plane2 + p2_words;
	const int log2_slots = A.log2_slots;
	const uint32_t n_slots = 1u << log2_slots;
	uint32_t *t_votes = t_keys + n_slots;
	uint8_t *l_code = (uint8_t *) (t_votes + n_slots);
	uint32_t *l_list = t_votes + n_slots + (q + 3) / 4;          // [lists_cap + 1] bucket number per list; the last entry = the all-zero bucket
	uint32_t *l_ovf = l_list + A.lists_cap + 1;                  // [2 * kCsOvfLists] start, count << 11 | list
	uint32_t *l_items = l_ovf + 2 * kCsOvfLists;                 // [ovf_items] list slot << 16 | segment
	uint32_t *s_misc = l_items + A.ovf_items;                    // [0] read length [1] abort [2] overflow lists [3] hits [4] k-mers [5] overflow items
	for (uint32_t s = tid; s < p1_words + p2_words; s += 64 * T) plane1[s] = 0;
	for (uint32_t s = tid; s < n_slots; s += 64 * T) { t_keys[s] = 0xFFFFFFFFu; t_votes[s] = 0; }
	if (tid < 16) s_misc[tid] = tid == 0 ? (uint32_t) q : 0u;
	__syncthreads();
	const bool diag = A.phase_cycles && (read & 255) == 0;  // sampled: the global atomics would serialise otherwise
	const unsigned long long c0 = diag ? wall_clock64() : 0ull;

	// read -> 2-bit codes (A0 C1 T2 G3, CSstatic.cpp:20-22), N = 4, past the end = 255
	{
		const uint8_t *rp = A.reads + (size_t) read * q;
		int first_nul = q;
		for (int i = tid; i < q; i += 64 * T) {
			const uint32_t ch = rp[i];
			uint8_t code;
			if (ch == 0) { code = 255; first_nul = min(first_nul, i); }
			else if (ch == 'N') code = 4;
			else code = (uint8_t) ((ch >> 1) & 3u);
			l_code[i] = code;
		}
		first_nul = wave_reduce_min(first_nul);
		if (lane == 0 && first_nul < q) atomicMin(&s_misc[0], (uint32_t) first_nul);
	}
	__syncthreads();
	const int L = (int) s_misc[0];
	const int n_kmers = L - k + 1;
	const int n_lists = n_kmers > 0 ? 2 * n_kmers : 0;
	const uint32_t zero_bucket = 1u << (2 * k);   // one bucket past the last k-mer: all zero
	{
		uint32_t nv = 0;
		for (int p = tid; p < n_kmers; p += 64 * T) {
			bool v = true;
			uint32_t kmer = 0;
			for (int j = 0; j < k; ++j) {
				const uint32_t c = l_code[p + j];
				v = v && (c < 4);
				kmer = (kmer << 2) | (c & 3u);
			}
			// CSstatic.cpp:30-41: a k-mer that starts right after a restart-position N run and ends exactly at the
			// read end is never visited
			if (v && p + k == L && p >= 1 && l_code[p - 1] == 4 && (p == 1 || l_code[p - 2] == 4)) v = false;
			l_list[2 * p] = v ? kmer : zero_bucket;
			l_list[2 * p + 1] = v ? cs_revcomp(kmer, k) : zero_bucket;
			nv += v ? 1u : 0u;
		}
		for (int i = n_lists + tid; i <= A.lists_cap; i += 64 * T) l_list[i] = zero_bucket;
		{ uint32_t total; (void) wave_prefix_small<5>(nv, total); if (lane == 0 && total) atomicAdd(&s_misc[4], total); }  // nv <= 16 (q <= 1024)
	}
	__syncthreads();
	const unsigned long long c1 = diag ? wall_clock64() : 0ull;

	const int bw = A.bucket_log2_words;      // bucket = 1 << bw dwords
	const int ls = bw - 2;                   // lanes per bucket = 1 << ls
	const uint32_t lpb = 1u << ls;
	const int bpr = 64 >> ls;                // buckets per wave round
	const uint32_t sub = (uint32_t) lane & (lpb - 1u);
	const int hs = 32 - log2_slots;
	const uint32_t a1_mask = (p1_words - 1u) << 2, a2_mask = (p2_words - 1u) << 2;   // byte address of the plane word = (bin >> 3) & mask
	const int bin_shift = A.bin_shift;
	// this lane's list in round r is li = (r T + wave) bpr + (lane >> ls): its strand and the step of its diagonal
	// correction are the same in every round (bpr is even)
	const uint32_t strand = ((uint32_t) lane >> ls) & 1u;
	const uint32_t tag = (strand << 31) | 0x40000000u;                 // register entry of a hit = bin | tag
	const int li0 = wave * bpr + (lane >> ls);
	const int p0 = li0 >> 1;
	// diagonal of the hit (CS.cpp:140-142): forward lists p, reverse-complement lists L - (p + k)
	uint32_t corr = strand ? (uint32_t) (L - (p0 + k)) : (uint32_t) p0;
	const uint32_t corr_step = strand ? (uint32_t) -(T * bpr / 2) : (uint32_t) (T * bpr / 2);
	const uint32_t w0 = sub * 4u - 1u;   // bucket word of slot j is 4 sub + j; it holds position (4 sub + j - 1) of the list

	uint32_t hits = 0;
	bool abort_fast = false;
	// one vote: returns the register entry (bin | tag, or kCsEmptySlot).  The LDS atomics run under the execution mask of the
	// lanes that have a hit: the kernel is bound by LDS bank cycles (64 random addresses on 32 banks), and lanes that are
	// switched off cost none -- half of the bucket slots are empty, and the second atomic is needed by 2-3 % of the hits
	auto vote = [&](uint32_t pos, uint32_t cr, bool valid, uint32_t tg) -> uint32_t {
		uint32_t out = kCsEmptySlot;
		if (valid) {
			const uint32_t bin = __builtin_amdgcn_ubfe(pos - cr, (uint32_t) bin_shift, 30u);
			const uint32_t msk = 1u << (bin & 31u);
			const uint32_t a = bin >> 3;
			const uint32_t old = atomicOr((uint32_t *) ((char *) plane1 + (a & a1_mask)), msk);
			if (old & msk) atomicOr((uint32_t *) ((char *) plane2 + (a & a2_mask)), msk);
			out = bin | tg;
		}
		return out;
	};

	// one bucket round of this wave: round r covers lists [(r T + wave) bpr, +bpr).  The load is unconditional (lanes without
	// a list read the all-zero bucket): a load under a divergent branch makes the compiler wait for it right there (the merge
	// copies the loaded registers), which would serialise the rounds on the memory latency.
	auto fetch = [&](int r, CsU4 &d) {
		const int li = (r * T + wave) * bpr + (lane >> ls);
		const uint32_t id = l_list[min(li, A.lists_cap)];
		d = *reinterpret_cast<const CsU4 *>(A.buckets + ((size_t) id << bw) + sub * 4u);
	};

	uint32_t bins[ROUNDS * 4];
	constexpr int DEPTH = kCsBucketDepth < ROUNDS ? kCsBucketDepth : ROUNDS;
	CsU4 ring[DEPTH];
#pragma unroll
	for (int d = 0; d < DEPTH - 1; ++d) fetch(d, ring[d]);
#pragma unroll
	for (int r = 0; r < ROUNDS; ++r) {
		// the prefetch is issued on every path: loads under a branch make the compiler's wait-count bookkeeping give up one
		// round of pipelining per merge point (rounds past the last list read the zero bucket and are skipped below)
		if (r + DEPTH - 1 < ROUNDS) fetch(r + DEPTH - 1, ring[(r + DEPTH - 1) % DEPTH]);
		const CsU4 cur = ring[r % DEPTH];
		const uint32_t cr = corr;
		corr += corr_step;
		if ((r * T + wave) * bpr >= n_lists) {  // wave-uniform
#pragma unroll
			for (int j = 0; j < 4; ++j) bins[r * 4 + j] = kCsEmptySlot;
			continue;
		}
		const uint32_t hdr = (uint32_t) __shfl((int) cur.x, lane & ~(int) (lpb - 1u));
		const uint32_t n_own = hdr & kCsHdrCountMask, n_other = (hdr >> 14) & kCsHdrCountMask;
		const uint32_t n_used = ((int) (n_own + n_other) < A.max_kfreq) ? n_own : 0u;   // CS.cpp:122
		hits += n_used;   // every lane of the group counts it: divided by the group size at the end
		const bool ovf = (hdr & kCsHdrOverflow) != 0u;
		if (ovf && n_used && sub == 0u) {   // rare
			const uint32_t slot = atomicAdd(&s_misc[2], 1u);
			const uint32_t li = (uint32_t) ((r * T + wave) * bpr + (lane >> ls));
			if (slot < (uint32_t) kCsOvfLists) { l_ovf[2 * slot] = cur.y; l_ovf[2 * slot + 1] = (n_used << 11) | li; }
		}
		const uint32_t n_inl = ovf ? 0u : n_used;
		const uint32_t pos[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
		for (int j = 0; j < 4; ++j) bins[r * 4 + j] = vote(pos[j], cr, (w0 + (uint32_t) j) < n_inl, tag);
	}
	{ uint32_t h = hits; for (int o = 32; o > 0; o >>= 1) h += __shfl_xor((int) h, o); h >>= ls; if (lane == 0 && h) atomicAdd(&s_misc[3], h); }
	__syncthreads();

	// lists longer than their bucket: 8-hit segments straight from d_positions, run-time loop, all waves
	const uint32_t n_ovf = s_misc[2];
	if (n_ovf > 0u && n_ovf <= (uint32_t) kCsOvfLists) {
		if (wave == 0) {
			const uint32_t nseg = (uint32_t) lane < n_ovf ? ((l_ovf[2 * lane + 1] >> 11) + kCsSeg - 1) / kCsSeg : 0u;
			const uint32_t incl = wave_inclusive_scan(nseg, lane);
			uint32_t o = incl - nseg;
			for (uint32_t sg = 0; sg < nseg; ++sg, ++o) if (o < A.ovf_items) l_items[o] = ((uint32_t) lane << 16) | sg;
			if (lane == 63) s_misc[5] = incl;
		}
		__syncthreads();
	}
	const uint32_t n_items = s_misc[5];
	if (n_ovf > (uint32_t) kCsOvfLists || n_items > A.ovf_items) abort_fast = true;
	// item -> its up to 8 positions, diagonal correction and tag
	auto ovf_item = [&](uint32_t idx, uint32_t (&pos8)[8], uint32_t &cnt, uint32_t &cr, uint32_t &tg) {
		const uint32_t item = l_items[idx];
		const uint32_t lo = item >> 16, sg = item & 0xFFFFu;
		const uint32_t meta = l_ovf[2 * lo + 1], li = meta & 0x7FFu;
		cnt = min((uint32_t) kCsSeg, (meta >> 11) - sg * kCsSeg);
		const CsU4 *src = reinterpret_cast<const CsU4 *>(A.positions + l_ovf[2 * lo] + sg * kCsSeg);  // the table is padded by 16 entries
		const CsU4 a = src[0], b = src[1];
		pos8[0] = a.x; pos8[1] = a.y; pos8[2] = a.z; pos8[3] = a.w; pos8[4] = b.x; pos8[5] = b.y; pos8[6] = b.z; pos8[7] = b.w;
		const uint32_t p = li >> 1;
		tg = ((li & 1u) << 31) | 0x40000000u;
		cr = (li & 1u) ? (uint32_t) (L - ((int) p + k)) : p;
	};
	if (!abort_fast) for (uint32_t idx = (uint32_t) tid; idx < n_items; idx += 64u * T) {
		uint32_t pos8[8], cnt, cr, tg;
		ovf_item(idx, pos8, cnt, cr, tg);
#pragma unroll
		for (int j = 0; j < kCsSeg; ++j) (void) vote(pos8[j], cr, (uint32_t) j < cnt, tg);
	}
	if (abort_fast) s_misc[1] = 1u;
	__syncthreads();
	const unsigned long long c2 = diag ? wall_clock64() : 0ull;
	CsRead R;
	R.L = L; R.n_lists = n_lists; R.H = s_misc[3]; R.n_valid = s_misc[4]; R.n_items = 0;
	if (s_misc[1] != 0u || R.H > A.hit_cap) {  // not provably exact here
		if (A.phase_cycles && tid == 0) atomicAdd(&A.phase_cycles[R.H > A.hit_cap ? 8 : 9], 1ull);
		if (wave == 0) cs_enqueue(A, read, lane, R);
		return;
	}

	// sweep 2.  Lane queues in what was plane 1: entry i of lane l of wave w at [(w cap1 + i) * 64 + l], the last row takes
	// the stores of full queues
	const uint32_t cap = p1_words / (64u * T) - 1u;
	uint32_t *qrow = plane1 + (uint32_t) wave * (cap + 1u) * 64u + (uint32_t) lane;
	uint32_t cnt = 0;
#pragma unroll
	for (int r = 0; r < ROUNDS; ++r) {
		if ((r * T + wave) * bpr >= n_lists) continue;  // wave-uniform
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			const uint32_t e = bins[r * 4 + j];
			if (e != kCsEmptySlot) {
				const uint32_t w = *(const uint32_t *) ((const char *) plane2 + ((e >> 3) & a2_mask));
				if ((w >> (e & 31u)) & 1u) { qrow[min(cnt, cap) * 64u] = e; ++cnt; }
			}
		}
	}
	// (plane 1 is only written above by its own wave's lanes: no barrier needed before the queues are read back)
	auto insert = [&](uint32_t e) -> bool {
		const uint32_t bin = e & 0x3FFFFFFFu;
		uint32_t slot = (bin * 2654435761u) >> hs;
		for (uint32_t probes = 0; probes < n_slots; ++probes) {
			const uint32_t prev = atomicCAS(&t_keys[slot], 0xFFFFFFFFu, bin);
			if (prev == bin || prev == 0xFFFFFFFFu) { atomicAdd(&t_votes[slot], (e & 0x80000000u) ? 0x10000u : 1u); return true; }
			slot = (slot + 1) & (n_slots - 1);
		}
		return false;
	};
	bool lost = cnt > cap;
	if (A.phase_cycles) {  // diagnostics: [4] queue entries [5] reads with a full lane queue
		uint32_t t = cnt; for (int o = 32; o > 0; o >>= 1) t += __shfl_xor((int) t, o);
		if (lane == 0) { atomicAdd(&A.phase_cycles[4], (unsigned long long) t); if (__ballot(cnt > cap)) atomicAdd(&A.phase_cycles[5], 1ull); }
	}
	{
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		const uint32_t mx = (uint32_t) wave_reduce_max((int) min(cnt, cap));
		for (uint32_t i = 0; i < mx; ++i) if (i < cnt) { if (!insert(qrow[i * 64u])) lost = true; }
	}
	for (uint32_t idx = (uint32_t) tid; idx < n_items; idx += 64u * T) {  // the overflow lists again (L2 / Infinity Cache hits)
		uint32_t pos8[8], n8, cr, tg;
		ovf_item(idx, pos8, n8, cr, tg);
		for (int j = 0; j < kCsSeg; ++j) if ((uint32_t) j < n8) {
			const uint32_t bin = __builtin_amdgcn_ubfe(pos8[j] - cr, (uint32_t) bin_shift, 30u);
			const uint32_t w = *(const uint32_t *) ((const char *) plane2 + ((bin >> 3) & a2_mask));
			if ((w >> (bin & 31u)) & 1u) { if (!insert(bin | tg)) lost = true; }
		}
	}
	if (__ballot(lost)) s_misc[1] = 1u;
	__syncthreads();
	const unsigned long long c3 = diag ? wall_clock64() : 0ull;
	if (wave != 0) return;
	if (s_misc[1] != 0u) { cs_enqueue(A, read, lane, R); return; }
	{
		// more than 3/4 full: too many spurious entries for this table -- the exact path has the room
		uint32_t keys = 0;
		for (uint32_t s2 = lane; s2 < n_slots; s2 += 64) keys += t_keys[s2] != 0xFFFFFFFFu;
		for (int o = 32; o > 0; o >>= 1) keys += __shfl_xor((int) keys, o);
		if (A.phase_cycles && lane == 0) { atomicAdd(&A.phase_cycles[6], (unsigned long long) keys); if (keys > (n_slots * 3u) / 4u) atomicAdd(&A.phase_cycles[7], 1ull); }
		if (keys > (n_slots * 3u) / 4u) { cs_enqueue(A, read, lane, R); return; }
	}
	if (!cs_finish<kCsFast>(A, read, lane, R, t_keys, t_votes, n_slots)) cs_enqueue(A, read, lane, R);
	if (diag && lane == 0) {  // diagnostics: 100 MHz ticks spent per phase, summed over the sampled reads
		atomicAdd(&A.phase_cycles[0], c1 - c0); atomicAdd(&A.phase_cycles[1], c2 - c1); atomicAdd(&A.phase_cycles[2], c3 - c2);
		atomicAdd(&A.phase_cycles[3], wall_clock64() - c3);
	}
}

// ---- EXACT paths: every hit goes into an open-addressing table (LDS, or global memory for very repetitive reads) --
template <int MODE>
__global__ __launch_bounds__(64) void cs_kernel(CsArgs A) {
	extern __shared__ __attribute__((aligned(16))) uint32_t cs_lds[];
	const int lane = threadIdx.x;
	const int item = blockIdx.x;
	const int read = A.read_list ? (int) A.read_list[item] : item;
	const int k = A.k;
	uint32_t *l_start = cs_lds;                        // [lists_cap]
	uint32_t *l_pref = cs_lds + A.lists_cap;           // [lists_cap + 1]
	uint8_t *l_code = (uint8_t *) (l_pref + A.lists_cap + 1);  // [q rounded up to 4]
	uint32_t *t_keys, *t_votes;
	int log2_slots;
	if (MODE == kCsExactGlobal) {
		log2_slots = (int) A.ovf_log2[item];
		t_keys = A.gtable_keys + A.ovf_table_off[item];
		t_votes = A.gtable_votes + A.ovf_table_off[item];
	} else {
		log2_slots = A.log2_slots;
		t_keys = (uint32_t *) l_code + (A.q + 3) / 4;
		t_votes = t_keys + (1u << log2_slots);
	}
	uint32_t n_slots = 1u << log2_slots;

	const CsRead R = cs_prepare<false>(A, read, lane, l_start, l_pref, l_code);
	const uint32_t H = R.H;
	const int L = R.L;
	if (MODE != kCsExactGlobal && H > A.hit_cap) { cs_enqueue(A, read, lane, R); return; }

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
	__syncthreads();
	if (MODE == kCsExactGlobal) __threadfence_block();

	cs_for_each_hit(A.positions, l_start, l_pref, R.n_lists, H, lane, [&](uint32_t pos, int li, uint32_t) {
		const int p = li >> 1;
		const uint32_t correction = (li & 1) ? (uint32_t) (L - (p + k)) : (uint32_t) p;  // CS.cpp:140-142
		const uint32_t bin = (pos - correction) >> A.bin_shift;
		uint32_t slot = (bin * 2654435761u) >> (32 - log2_slots);
		for (;;) {
			const uint32_t prev = atomicCAS(&t_keys[slot], 0xFFFFFFFFu, bin);
			if (prev == bin || prev == 0xFFFFFFFFu) break;
			slot = (slot + 1) & (n_slots - 1);
		}
		atomicAdd(&t_votes[slot], (li & 1) ? 0x10000u : 1u);
	});
	__syncthreads();
	if (MODE == kCsExactGlobal) __threadfence_block();
	(void) cs_finish<MODE>(A, read, lane, R, t_keys, t_votes, n_slots);
}

// ---- candidate ORDER --------------------------------------------------------------------------------------------
// The reference lists a read's candidates in the order in which their bins first reached the running threshold
// (rList, CS::AddLocationStd, src/CS.cpp:196-211), and ScoreBuffer::top1SE keeps the FIRST of several equally scoring
// candidates.  The set of candidates does not depend on that order, so the search kernels above do not track it; for
// the reads where it matters (equal scores among the candidates) this kernel replays the votes of one read in exactly
// the reference's order -- k-mers left to right, forward list then reverse-complement list, list entries in index
// order -- and records when each bin entered rList.  Only bins with >= 2 votes can move the running maximum beyond 1
// or become candidates, so the replay is restricted to them: sweep A finds those bins (bit plane + exact table, as in
// the fast path), sweep B counts their votes and marks their hits on a time line in LDS, then the marked hits
// (a few hundred of ~4 300) are replayed in time order by the whole wave in lock step.
// out: cand_rank[c] = 2 * (rList position among the tracked bins) + strand for every candidate c of the read, i.e. its
// relative order in CollectResultsStd's output (forward before reverse of one bin, src/CS.cpp:289-304).
constexpr int kCsOrderLog2Slots = 10;       // tracked bins (>= 2 votes, plus bit collisions): 1024 slots
constexpr uint32_t kCsOrderMaxHits = 6144;  // time line entries in LDS; reads with more hits keep the position order
constexpr uint32_t kCsOrderUnknown = 0xFFFFFFFFu;
constexpr uint32_t kCsOrderItemCap = 1280; // 8-hit list segments of such a read

__global__ __launch_bounds__(64) void cs_order_kernel(CsArgs A, const uint32_t *__restrict__ cand_loc, const uint32_t *__restrict__ cand_sv,
		uint32_t *__restrict__ cand_rank) {
	extern __shared__ __attribute__((aligned(16))) uint32_t cs_lds[];
	__shared__ uint32_t s_keys;  // distinct tracked bins
	const int lane = threadIdx.x;
	const int read = (int) A.read_list[blockIdx.x];
	const int k = A.k;
	uint32_t *l_start = cs_lds;
	uint32_t *l_pref = cs_lds + A.lists_cap;
	uint8_t *l_code = (uint8_t *) (l_pref + A.lists_cap + 1);
	uint32_t *plane = (uint32_t *) l_code + (A.q + 3) / 4;
	constexpr uint32_t plane_words = 2048, n_slots = 1u << kCsOrderLog2Slots;
	uint32_t *t_keys = plane + plane_words;
	uint32_t *t_votes = t_keys + n_slots;   // final votes: forward | reverse << 16
	uint32_t *t_run = t_votes + n_slots;    // votes so far during the replay
	uint32_t *t_rank = t_run + n_slots;
	uint32_t *ev_at = t_rank + n_slots;     // [kCsOrderMaxHits]: slot | strand << 31 of the hit at that time, or empty (reads with more hits: global memory)
	for (uint32_t s = lane; s < plane_words; s += 64) plane[s] = 0;
	for (uint32_t s = lane; s < n_slots; s += 64) { t_keys[s] = 0xFFFFFFFFu; t_votes[s] = 0; t_run[s] = 0; t_rank[s] = kCsOrderUnknown; }
	if (lane == 0) s_keys = 0;
	uint32_t *l_items = ev_at + kCsOrderMaxHits;  // [kCsOrderItemCap]
	const CsRead R = cs_prepare<true, uint32_t>(A, read, lane, l_start, l_pref, l_code, l_items, kCsOrderItemCap);
	const uint32_t H = R.H;
	const int L = R.L;
	const uint32_t cb = A.cand_base[read], cn = A.cand_count[read];
	auto give_up = [&]() { for (uint32_t c = lane; c < cn; c += 64) cand_rank[cb + c] = kCsOrderUnknown; };
	// very repetitive reads: the time line moves to global memory and the lists are walked hit by hit (rare, slow, exact);
	// the 16-bit list offsets of l_pref bound that at 65 535 hits
	const bool big = H > kCsOrderMaxHits || R.n_items > kCsOrderItemCap;
	if (big) {
		if (!A.order_scratch || H > A.order_gcap || H >= 65536u) { give_up(); return; }
		ev_at = A.order_scratch + (size_t) blockIdx.x * A.order_gcap;
	}
	__syncthreads();
	auto bin_of = [&](uint32_t pos, int li) -> uint32_t {
		const int p = li >> 1;
		const uint32_t correction = (li & 1) ? (uint32_t) (L - (p + k)) : (uint32_t) p;  // CS.cpp:140-142
		return ((pos - correction) >> A.bin_shift) & 0x3FFFFFFFu;
	};
	// sweep A (the only pass over the position lists): every hit is written to the time line (bin | strand << 31);
	// bins hit at least twice (or colliding on a plane bit) become tracked keys
	auto vote = [&](uint32_t pos, int li, uint32_t t) {
		const uint32_t bin = bin_of(pos, li);
		ev_at[t] = bin | ((li & 1) ? 0x80000000u : 0u);
		const uint32_t b = (bin * 0x9E3779B1u) >> 16;
		const uint32_t msk = 1u << (b & 31);
		if (atomicOr(&plane[b >> 5], msk) & msk) {
			uint32_t slot = (bin * 2654435761u) >> (32 - kCsOrderLog2Slots);
			for (uint32_t probes = 0; probes < n_slots; ++probes) {
				const uint32_t prev = atomicCAS(&t_keys[slot], 0xFFFFFFFFu, bin);
				if (prev == bin) break;
				if (prev == 0xFFFFFFFFu) { atomicAdd(&s_keys, 1u); break; }
				slot = (slot + 1) & (n_slots - 1);
			}
		}
	};
	if (big) {
		const int n_lists = R.n_lists;
		for (uint32_t h = (uint32_t) lane; h < H; h += 64) {
			int lo = 0, hi = n_lists;  // largest list whose first hit is at or before h (empty lists share their successor's offset)
			while (hi - lo > 1) {
				const int mid = (lo + hi) >> 1;
				if ((l_pref[mid] >> 16) <= h) lo = mid; else hi = mid;
			}
			while ((l_pref[lo] & 0xFFFFu) == 0u || h - (l_pref[lo] >> 16) >= (l_pref[lo] & 0xFFFFu)) --lo;  // skip empty lists that start at the same offset
			vote(A.positions[l_start[lo] + (h - (l_pref[lo] >> 16))], lo, h);
		}
	} else {
		// work item = 8 consecutive hits of one list (two 16-byte loads), the next item's loads in flight
		CsU4 cur[2], nxt[2];
		auto fetch = [&](uint32_t idx, CsU4 (&d)[2]) -> uint32_t {
			if (idx >= R.n_items) return 0xFFFFFFFFu;
			const uint32_t item = l_items[idx];
			const uint32_t li = item >> 16, sg = item & 0xFFFFu;
			const CsU4 *src = reinterpret_cast<const CsU4 *>(A.positions + l_start[li] + sg * kCsSeg);
			d[0] = src[0]; d[1] = src[1];  // the table is padded by 16 entries
			return item;
		};
		uint32_t item = fetch((uint32_t) lane, cur);
		for (uint32_t idx = (uint32_t) lane; idx < R.n_items; idx += 64) {
			const uint32_t item_n = fetch(idx + 64, nxt);
			const uint32_t li = item >> 16, sg = item & 0xFFFFu;
			const uint32_t meta = l_pref[li];
			const uint32_t cnt = min((uint32_t) kCsSeg, (meta & 0xFFFFu) - sg * kCsSeg), t0 = (meta >> 16) + sg * kCsSeg;
			const uint32_t pos8[8] = {cur[0].x, cur[0].y, cur[0].z, cur[0].w, cur[1].x, cur[1].y, cur[1].z, cur[1].w};
#pragma unroll
			for (int j = 0; j < kCsSeg; ++j) if ((uint32_t) j < cnt) vote(pos8[j], (int) li, t0 + (uint32_t) j);
			item = item_n; cur[0] = nxt[0]; cur[1] = nxt[1];
		}
	}
	__syncthreads();
	if (s_keys > (n_slots * 3u) / 4u) { give_up(); return; }
	// sweep B (LDS only): exact votes of the tracked bins; their time line entries become slot | strand << 31, all others
	// empty.  The plane is rebuilt as a bit set of the tracked keys first, so that the ~90 % untracked hits cost one read.
	for (uint32_t s2 = lane; s2 < plane_words; s2 += 64) plane[s2] = 0;
	__syncthreads();
	for (uint32_t s2 = lane; s2 < n_slots; s2 += 64) {
		const uint32_t key = t_keys[s2];
		if (key != 0xFFFFFFFFu) { const uint32_t b = (key * 0x9E3779B1u) >> 16; atomicOr(&plane[b >> 5], 1u << (b & 31)); }
	}
	__syncthreads();
	for (uint32_t t = lane; t < H; t += 64) {
		const uint32_t e = ev_at[t];
		const uint32_t bin = e & 0x3FFFFFFFu;
		const uint32_t b = (bin * 0x9E3779B1u) >> 16;
		uint32_t out = 0xFFFFFFFFu;
		if ((plane[b >> 5] >> (b & 31)) & 1u) {
			uint32_t slot = (bin * 2654435761u) >> (32 - kCsOrderLog2Slots);
			for (;;) {
				const uint32_t key = t_keys[slot];
				if (key == bin) { atomicAdd(&t_votes[slot], (e & 0x80000000u) ? 0x10000u : 1u); out = slot | (e & 0x80000000u); break; }
				if (key == 0xFFFFFFFFu) break;
				slot = (slot + 1) & (n_slots - 1);
			}
		}
		ev_at[t] = out;
	}
	__syncthreads();
	// replay in time order.  A bin with one vote never moves the maximum beyond 1 and is never a candidate: the very
	// first hit of the read already sets the maximum to 1 (CS.cpp:197-202), so only bins with >= 2 votes are replayed.
	float max_hit = H > 0 ? 1.0f : 0.0f, thresh = max_hit * A.sensitivity;
	uint32_t next_rank = 0;
	for (uint32_t t0 = 0; t0 < H; t0 += 64) {
		const uint32_t t = t0 + (uint32_t) lane;
		uint32_t e = (t < H) ? ev_at[t] : 0xFFFFFFFFu;
		if (e != 0xFFFFFFFFu) {
			const uint32_t v = t_votes[e & 0x7FFFFFFFu];
			if ((v & 0xFFFFu) + (v >> 16) < 2u) e = 0xFFFFFFFFu;
		}
		unsigned long long todo = __ballot(e != 0xFFFFFFFFu);
		while (todo) {
			const int i = __ffsll((long long) todo) - 1;
			todo &= todo - 1;
			const uint32_t ev = (uint32_t) __shfl((int) e, i);
			const uint32_t slot = ev & 0x7FFFFFFFu;
			uint32_t run = t_run[slot];
			uint32_t score;
			if (ev & 0x80000000u) { run += 0x10000u; score = run >> 16; } else { run += 1u; score = run & 0xFFFFu; }
			if (lane == 0) t_run[slot] = run;
			if ((float) score > max_hit) { max_hit = (float) score; thresh = max_hit * A.sensitivity; }      // CS.cpp:197-202
			if (t_rank[slot] == kCsOrderUnknown && (float) score >= thresh) { if (lane == 0) t_rank[slot] = next_rank; ++next_rank; }  // CS.cpp:205-208
			__builtin_amdgcn_wave_barrier();  // one wave: LDS operations complete in program order
		}
	}
	__syncthreads();
	const uint32_t centre = A.bin_shift > 0 ? (1u << (A.bin_shift - 1)) : 0u;
	for (uint32_t c = lane; c < cn; c += 64) {
		const uint32_t bin = ((cand_loc[cb + c] - centre) >> A.bin_shift) & 0x3FFFFFFFu;
		uint32_t slot = (bin * 2654435761u) >> (32 - kCsOrderLog2Slots);
		uint32_t rank = kCsOrderUnknown;
		for (uint32_t probes = 0; probes < n_slots; ++probes) {
			const uint32_t key = t_keys[slot];
			if (key == bin) { if (t_rank[slot] != kCsOrderUnknown) rank = 2u * t_rank[slot] + (cand_sv[cb + c] & 1u); break; }
			if (key == 0xFFFFFFFFu) break;
			slot = (slot + 1) & (n_slots - 1);
		}
		// a candidate with a single vote (possible only when the final threshold is <= 1) entered rList at its only hit:
		// not tracked here, its order stays unknown and the caller falls back to the position order for that read
		cand_rank[cb + c] = rank;
	}
}

}  // namespace ngm
