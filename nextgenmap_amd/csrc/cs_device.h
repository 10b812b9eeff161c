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
	const uint32_t *n_list_dev; // optional (cs_kernel, cs_global_kernel): the number of listed reads lives in this device word and the workgroups stride over the list (cs_queue_device.h); null: one workgroup per item of the grid
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
	const uint32_t *buckets;   // FAST: the bucketed index followed by a copy of the position table (refindex.h), one array
	int bucket_log2_words;     // FAST: bucket = 1 << this dwords
	uint32_t pos_base;         // FAST: word offset of the position table copy inside `buckets`
	int lists_cap;          // LDS capacity for lists (>= 2*(q-k+1))
	int log2_slots;         // exact table slots (power of two) in LDS
	int log2_bits;          // (unused by the kernels; kept for diagnostics)
	uint32_t plane_bits;    // FAST: bits of the plane, a multiple of 2048 (any size: the hash is reduced with a multiply-high)
	int fast_items;         // FAST: items per lane of the kernel instantiation in use
	int items16;            // FAST: 16-bit work items
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
	// ... except for reads with at most kCsFixedSlots candidates (nearly all): those have their own slots behind the regions, so
	// that their workgroup does not wait for a returning L2 atomic.  0: off
	uint32_t fixed_base;
	uint32_t *status;       // [0] output overflow flag, [1] number of queued reads
	unsigned long long *counters;  // per region, stride kCsCursorStride: [0] k-mers looked up, [1] hits voted (algorithmic-bytes accounting), [2] candidates
	uint32_t *order_scratch;    // cs_order_kernel: time lines in global memory for reads with more hits than LDS holds
	uint32_t order_gcap;        // ... entries per workgroup
	uint32_t order_max_hits;    // cs_order_kernel: time line entries in LDS (sized by the host from the expected hits per read)
	uint32_t *order_info;       // cs_order_kernel: per listed read {index hits, 0 = replayed | reason it was left to the exact kernel (1 k-mer variants, 2 time line, 3 tracked bins) | tracked bins << 8}
	unsigned long long *phase_cycles;  // optional diagnostics (fast path): [0] lists [1] sweep 1 [2] sweep 2 [3] candidates
	int read_lo;                       // cs_canon_kernel: first read of this launch (a batch may be searched in several launches: reads [read_lo, n))
	int reads_per_wg;                  // cs_canon_kernel: > 0: workgroup b maps reads [b, b + 1) * reads_per_wg (as many workgroups as that takes); 0: persistent workgroups that draw reads from status[2]
	int debug_stop;                    // diagnostics (NGM_HIP_CS_STOP, cs_canon_kernel only): 1-3 = leave a read after that phase with no candidates -- instruction counts per phase by difference
	uint32_t *ovf_read;     // [n] queue written by this pass
	uint32_t *ovf_hits;     // [n]
	// kCsExactGlobal
	// bisulfite mapping (CS::PrefixMutateSearch, src/CS.cpp:54-112): every T (second mate of a pair: A) of a read k-mer may be a
	// converted C (G): all 2^m combinations are looked up when the k-mer has at most bs_cutoff of them, none otherwise; the read is
	// walked with bs_read_skip (the "kmer_skip" of a --bs-mapping run applies to the READ, src/CS.cpp:556-560, the index is built
	// with skip 0, src/PrefixTable.cpp:199-207).  Exact paths only.
	int bs;                 // 0 off, 1 on, 2: `--slam-seq` with bit 2 (weighted search, cs_slam_device.h): every read k-mer and its SINGLE C > T (second mates: G > A) conversions, no cut-off, read walked with skip 0 (src/CS.cpp:57-92)
	int bs_cutoff;          // Config "bs_cutoff" (6)
	int bs_read_skip;       // k-mers skipped between two looked-up ones inside an N-free stretch of the read
	int bs_paired;          // reads 2i + 1 are second mates: A -> G instead of T -> C (src/CS.cpp:356-376)
	const uint64_t *ovf_table_off;  // per queued read: offset (in slots) into gtable_*
	const uint32_t *ovf_log2;       // per queued read: log2 slots
	uint32_t *gtable_keys;
	uint32_t *gtable_votes;
	// cs_slam_kernel: per-workgroup slices of gtable_keys (persistent workgroups: slice blockIdx.x of slam_slice_words words; with a
	// read_list: ovf_table_off / ovf_log2 per listed read)
	unsigned long long slam_slice_words;
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

// Cross-lane steps as DPP modifiers (row shifts / mirrors inside the 16-lane rows, row_bcast15 / row_bcast31 across them:
// gfx9 has both) instead of ds_bpermute round trips through the LDS pipe.  All 64 lanes must be active at the call.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int wave_dpp(int identity, int v) { return __builtin_amdgcn_update_dpp(identity, v, CTRL, ROW_MASK, 0xf, false); }
constexpr int kDppQuadSwap1 = 0xB1, kDppQuadSwap2 = 0x4E, kDppRowHalfMirror = 0x141, kDppRowMirror = 0x140, kDppRowBcast15 = 0x142, kDppRowBcast31 = 0x143;
constexpr int kDppRowShr1 = 0x111, kDppRowShr2 = 0x112, kDppRowShr4 = 0x114, kDppRowShr8 = 0x118;

__device__ __forceinline__ int wave_reduce_max(int v) {
	v = max(v, wave_dpp<kDppQuadSwap1, 0xf>(v, v));
	v = max(v, wave_dpp<kDppQuadSwap2, 0xf>(v, v));
	v = max(v, wave_dpp<kDppRowHalfMirror, 0xf>(v, v));
	v = max(v, wave_dpp<kDppRowMirror, 0xf>(v, v));       // every lane: maximum of its row
	v = max(v, wave_dpp<kDppRowBcast15, 0xa>(v, v));      // rows 1, 3: with the row before
	v = max(v, wave_dpp<kDppRowBcast31, 0xc>(v, v));      // rows 2, 3: with rows 0-1
	return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int wave_reduce_min(int v) {
	v = min(v, wave_dpp<kDppQuadSwap1, 0xf>(v, v));
	v = min(v, wave_dpp<kDppQuadSwap2, 0xf>(v, v));
	v = min(v, wave_dpp<kDppRowHalfMirror, 0xf>(v, v));
	v = min(v, wave_dpp<kDppRowMirror, 0xf>(v, v));
	v = min(v, wave_dpp<kDppRowBcast15, 0xa>(v, v));
	v = min(v, wave_dpp<kDppRowBcast31, 0xc>(v, v));
	return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v, int) {
	v += (uint32_t) wave_dpp<kDppRowShr1, 0xf>(0, (int) v);   // lanes without a source inside their row add the identity
	v += (uint32_t) wave_dpp<kDppRowShr2, 0xf>(0, (int) v);
	v += (uint32_t) wave_dpp<kDppRowShr4, 0xf>(0, (int) v);
	v += (uint32_t) wave_dpp<kDppRowShr8, 0xf>(0, (int) v);   // inclusive prefix inside each row
	v += (uint32_t) wave_dpp<kDppRowBcast15, 0xa>(0, (int) v);
	v += (uint32_t) wave_dpp<kDppRowBcast31, 0xc>(0, (int) v);
	return v;
}
__device__ __forceinline__ uint32_t wave_inclusive_max(uint32_t v) {
	v = max(v, (uint32_t) wave_dpp<kDppRowShr1, 0xf>(0, (int) v));
	v = max(v, (uint32_t) wave_dpp<kDppRowShr2, 0xf>(0, (int) v));
	v = max(v, (uint32_t) wave_dpp<kDppRowShr4, 0xf>(0, (int) v));
	v = max(v, (uint32_t) wave_dpp<kDppRowShr8, 0xf>(0, (int) v));
	v = max(v, (uint32_t) wave_dpp<kDppRowBcast15, 0xa>(0, (int) v));
	v = max(v, (uint32_t) wave_dpp<kDppRowBcast31, 0xc>(0, (int) v));
	return v;
}
__device__ __forceinline__ uint32_t wave_last(uint32_t v) { return (uint32_t) __builtin_amdgcn_readlane((int) v, 63); }
__device__ __forceinline__ uint32_t wave_first(uint32_t v) { return (uint32_t) __builtin_amdgcn_readlane((int) v, 0); }

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

// bucket word 0 (written by fill_buckets_kernel, refindex.cpp)
constexpr uint32_t kCsHdrCountMask = 0x3FFFu;   // bits 0-13 own list length (<= 9900), bits 14-27 the other strand's
constexpr uint32_t kCsHdrOverflow = 0x80000000u;
constexpr int kCsFixedSlots = 4;   // candidate slots every read owns (CsArgs::fixed_base)

// BUCKETS: the lists come from the bucketed index: one 8-byte read of bucket words 0 and 1 per list -- the header (length,
// "does not fit" flag) and either the first position or, for a list that does not fit, its start in the position table --
// which also pulls the first 64-byte sector of the bucket (15 positions) towards the L2; {start, count} is then the same
// pair the plain index holds, with `start` a word offset into A.buckets.
// PrefT: uint32_t keeps (length | time of the list's first hit << 16) per list (order replay), uint16_t the length only (the
// fast path: 560 bytes of LDS less per read, which is what lets a tenth read fit a CU)
template <bool ITEMS, typename ItemT = uint32_t, bool BUCKETS = false, typename PrefT = uint32_t>
__device__ __forceinline__ CsRead cs_prepare(const CsArgs &A, int read, int lane, uint32_t *l_start, PrefT *l_pref, uint8_t *l_code,
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
				if (v && !BUCKETS) { ef[r] = A.index[kmer]; er[r] = A.index[cs_revcomp(kmer, k)]; }
				if (v && BUCKETS) {
					const uint32_t kf = kmer, kr = cs_revcomp(kmer, k);
					const uint2 hf = *reinterpret_cast<const uint2 *>(A.buckets + ((size_t) kf << A.bucket_log2_words));
					const uint2 hr = *reinterpret_cast<const uint2 *>(A.buckets + ((size_t) kr << A.bucket_log2_words));
					ef[r] = make_uint2((hf.x & kCsHdrOverflow) ? A.pos_base + hf.y : (kf << A.bucket_log2_words) + 1u, hf.x & kCsHdrCountMask);
					er[r] = make_uint2((hr.x & kCsHdrOverflow) ? A.pos_base + hr.y : (kr << A.bucket_log2_words) + 1u, hr.x & kCsHdrCountMask);
				}
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
					l_start[2 * p] = sf; l_pref[2 * p] = (PrefT) b0;
					l_start[2 * p + 1] = sr; l_pref[2 * p + 1] = (PrefT) (b0 + cf);
				}
			} else {
				const uint32_t nsf = (cf + kCsSeg - 1) / kCsSeg, nsr = (cr + kCsSeg - 1) / kCsSeg;
				const uint32_t incl_s = wave_inclusive_scan(nsf + nsr, lane);
				if (p < n_kmers) {
					const uint32_t b0 = carry + incl - both;  // time (flattened hit index) of the forward list's first hit
					l_start[2 * p] = sf; l_pref[2 * p] = (PrefT) ((cf & 0xFFFFu) | (b0 << 16));
					l_start[2 * p + 1] = sr; l_pref[2 * p + 1] = (PrefT) ((cr & 0xFFFFu) | ((b0 + cf) << 16));
					uint32_t o = carry_s + incl_s - (nsf + nsr);
					for (uint32_t sg = 0; sg < nsf; ++sg, ++o) if (o < items_cap) l_items[o] = CsItem<ItemT>::make((uint32_t) (2 * p), sg);
					for (uint32_t sg = 0; sg < nsr; ++sg, ++o) if (o < items_cap) l_items[o] = CsItem<ItemT>::make((uint32_t) (2 * p + 1), sg);
				}
				carry_s += wave_last(incl_s);
			}
			carry += wave_last(incl);
		}
	}
	if (!ITEMS && lane == 0) l_pref[R.n_lists] = (PrefT) carry;
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

// ---- bisulfite mapping: which k-mers of the read are looked up, and as which variants ---------------------------------
// One wave.  Codes as in cs_prepare; l_vbase[p] = number of k-mer variants in front of k-mer p (l_vbase[n_kmers] = all of them):
// a k-mer contributes 2^m variants (m = its bases equal to `from`, m <= bs_cutoff), 0 when it is not looked up at all -- N inside,
// more than bs_cutoff convertible bases, or not on the read-side stride: PrefixIteration visits the first k-mer of every N-free
// stretch and then every (skip + 1)-th (CSstatic.cpp:57-75; a stretch begins after an N, the counter restarts with it).
struct CsBsRead { int L; int n_kmers; uint32_t V; uint32_t n_valid; uint32_t from, to; };
constexpr int kCsBsChunk = 1024;   // variants whose lists are held in LDS at a time

__device__ __forceinline__ CsBsRead cs_bs_scan(const CsArgs &A, int read, int lane, uint8_t *l_code, uint32_t *l_vbase) {
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
	CsBsRead R;
	R.L = wave_reduce_min(first_nul);
	__syncthreads();
	const int L = R.L;
	R.n_kmers = max(L - k + 1, 0);
	const bool second = A.bs_paired && (read & 1);
	const bool slam = A.bs == 2;
	R.from = slam ? (second ? 3u : 1u) : (second ? 0u : 2u);   // bisulfite: A -> G for the second mate, T -> C otherwise; SLAM-seq: G -> A / C -> T (codes A0 C1 T2 G3, CS.cpp:356-376)
	R.to = slam ? (second ? 0u : 2u) : (second ? 3u : 1u);
	uint32_t carry = 0, n_valid = 0;
	int last_n = -1;   // position of the last N in front of the current round (wave-uniform)
	if (lane == 0) l_vbase[0] = 0;
	for (int base = 0; base < R.n_kmers; base += 64) {
		const int p = base + lane;
		// the N-free stretch k-mer p lies in starts behind the last N at or before p: prefix maximum over the positions of the round
		const int here = (p < L && l_code[p] == 4) ? p : -1;
		const int seg_n = max((int) wave_inclusive_max((uint32_t) (here + 1)) - 1, last_n);   // last N at or before p (-1: none)
		uint32_t nvar = 0;
		bool looked = false;
		if (p < R.n_kmers) {
			bool v = true;
			uint32_t m = 0;
			for (int j = 0; j < k; ++j) {
				const uint32_t c = l_code[p + j];
				v = v && (c < 4);
				m += (c == R.from) ? 1u : 0u;
			}
			if (v && p + k == L && p >= 1 && l_code[p - 1] == 4 && (p == 1 || l_code[p - 2] == 4)) v = false;  // CSstatic.cpp:30-41, see cs_prepare
			// (a valid k-mer holds no N: the last N at or before p lies in front of it)
			// (SLAM-seq: the k-mer itself and its m single conversions, whatever m -- CS.cpp:69-75, :80-92)
			if (v && ((p - (seg_n + 1)) % (A.bs_read_skip + 1)) == 0) { looked = true; if (slam) nvar = 1u + m; else if ((int) m <= A.bs_cutoff) nvar = 1u << m; }
		}
		n_valid += (uint32_t) __popcll(__ballot(looked));
		const uint32_t incl = wave_inclusive_scan(nvar, lane);
		if (p < R.n_kmers) l_vbase[p + 1] = carry + incl;
		carry += wave_last(incl);
		last_n = max(last_n, (int) wave_last((uint32_t) (seg_n + 1)) - 1);
	}
	R.V = carry; R.n_valid = n_valid;
	__syncthreads();
	return R;
}

// The variants [v0, v1) of the read, in the reference's order: k-mers left to right; per k-mer the recursion of
// CS::PrefixMutateSearchEx (src/CS.cpp:97-112) -- the k-mer itself, then for every convertible base i (from the LAST base of the
// k-mer towards the first) the k-mer with base i converted followed by all further conversions of bases beyond i -- i.e. the
// subsets of the convertible bases in lexicographic order of their sorted element lists.  Per variant two lists (forward k-mer,
// reverse complement), dropped together when they hold max_kfreq hits or more (CS.cpp:122).
// l_start[2j], l_start[2j + 1]: position-table offsets of variant v0 + j; l_pref: prefix sums of the list lengths (TIMES: length |
// time of the list's first hit << 16, times counted from t_base); l_vpos[j]: the k-mer's position in the read.
// Returns the hits of the chunk; *segments (optional) += their 8-hit segments.  One wave.
template <bool TIMES>
__device__ __forceinline__ uint32_t cs_bs_chunk(const CsArgs &A, const CsBsRead &R, int lane, const uint8_t *l_code, const uint32_t *l_vbase, uint32_t v0, uint32_t v1,
		uint32_t t_base, uint32_t *l_start, uint32_t *l_pref, uint16_t *l_vpos, uint32_t *segments = nullptr) {
	const int k = A.k;
	uint32_t carry = 0, carry_s = 0;
	for (uint32_t vb = v0; vb < v1; vb += 64) {
		const uint32_t v = vb + (uint32_t) lane;
		uint32_t cf = 0, cr = 0, sf = 0, sr = 0, wdiv_out = 0;
		int p = 0;
		if (v < v1) {
			int lo = 0, hi = R.n_kmers;  // the k-mer with l_vbase[p] <= v < l_vbase[p + 1]
			while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (l_vbase[mid] <= v) lo = mid; else hi = mid; }
			p = lo;
			uint32_t r = v - l_vbase[p];
			uint32_t kmer = 0, conv = 0;   // conv: bit i set = base i (counted from the k-mer's last base) is convertible
			for (int j = 0; j < k; ++j) {
				const uint32_t c = l_code[p + j] & 3u;
				kmer = (kmer << 2) | c;
				conv = (conv << 1) | (c == R.from ? 1u : 0u);
			}
			// unrank r in the recursion order: rank 0 = nothing converted; below a node whose last converted base is the e-th convertible one
			// lie 2^(m - 1 - e) nodes
			const int m = __popc(conv);
			uint32_t chosen = 0;   // bit e: the e-th convertible base (in increasing i) is converted
			int e = 0;
			// SLAM-seq (PrefixMutateSearchSlamSeq, CS.cpp:80-92): variant 0 is the k-mer itself (weight 1), variant r > 0 converts the
			// (r - 1)-th convertible base counted from the k-mer's last base, alone (weight 1 / (m + 1): the divisor travels in l_vpos)
			const uint32_t wdiv = (A.bs == 2) ? (r == 0u ? 1u : (uint32_t) m + 1u) : 0u;
			if (A.bs == 2) { chosen = r ? (1u << (r - 1u)) : 0u; r = 0; }
			wdiv_out = wdiv;
			while (r > 0) {
				r -= 1;
				for (;; ++e) {
					const uint32_t size = 1u << (m - 1 - e);
					if (r < size) { chosen |= 1u << e; ++e; break; }
					r -= size;
				}
			}
			int idx = 0;
			for (int i = 0; i < k; ++i) if ((conv >> i) & 1u) { if ((chosen >> idx) & 1u) kmer = (kmer & ~(3u << (2 * i))) | (R.to << (2 * i)); ++idx; }
			const uint2 ef = A.index[kmer], er = A.index[cs_revcomp(kmer, k)];
			if ((int) (ef.y + er.y) < A.max_kfreq) { cf = ef.y; sf = ef.x; cr = er.y; sr = er.x; }  // CS.cpp:122
		}
		const uint32_t both = cf + cr;
		const uint32_t incl = wave_inclusive_scan(both, lane);
		const uint32_t nseg = (cf + kCsSeg - 1) / kCsSeg + (cr + kCsSeg - 1) / kCsSeg;
		const uint32_t incl_s = wave_inclusive_scan(nseg, lane);
		if (v < v1) {
			const uint32_t j = v - v0, b0 = carry + incl - both;
			l_start[2 * j] = sf; l_start[2 * j + 1] = sr;
			if (TIMES) { l_pref[2 * j] = (cf & 0xFFFFu) | ((t_base + b0) << 16); l_pref[2 * j + 1] = (cr & 0xFFFFu) | ((t_base + b0 + cf) << 16); }
			else { l_pref[2 * j] = b0; l_pref[2 * j + 1] = b0 + cf; }
			l_vpos[j] = (uint16_t) ((uint32_t) p | (wdiv_out << 10));   // (read positions are below 1 024)
		}
		carry += wave_last(incl);
		carry_s += wave_last(incl_s);
	}
	if (!TIMES && lane == 0) l_pref[2 * (v1 - v0)] = carry;
	if (segments) *segments += carry_s;
	return carry;
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
	uint32_t total = wave_last(incl);
	if ((int64_t) total >= (int64_t) A.max_cmrs) total = 0;  // "if (index < maxScores) AllocScores" (CS.cpp:308-310)
	const bool fixed = A.fixed_base != 0u && total <= (uint32_t) kCsFixedSlots;  // wave-uniform
	unsigned long long base = 0;
	if (lane == 0) {
		if (!fixed) {
			base = total ? atomicAdd(&A.out_total[region * kCsCursorStride], (unsigned long long) total) : 0ull;
			if (base + total > A.out_capacity) { atomicExch(&A.status[0], 1u); }
		}
		A.cand_base[read] = fixed ? A.fixed_base + (uint32_t) read * (uint32_t) kCsFixedSlots : (uint32_t) (region * A.out_capacity + base);
		A.cand_count[read] = total;
		A.max_votes[read] = max_hit;
		if (A.max_both) A.max_both[read] = (float) mxb;
		A.read_len[read] = (uint16_t) R.L;
		if (A.counters && total) atomicAdd(&A.counters[region * kCsCursorStride + 2], (unsigned long long) total);
	}
	if (total == 0) return true;
	uint32_t w;
	if (fixed) w = A.fixed_base + (uint32_t) read * (uint32_t) kCsFixedSlots + (incl - count);
	else {
		base = wave_first((uint32_t) base) | ((unsigned long long) wave_first((uint32_t) (base >> 32)) << 32);
		if (base + total > A.out_capacity) return true;
		w = (uint32_t) (region * A.out_capacity + base) + (incl - count);
	}
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
// Work item = one segment of up to 8 consecutive hits of ONE position list (constant diagonal correction and strand,
// two 16-byte loads, no per-hit list walking); a lane owns up to kCsFastItems items, the next item's loads are in
// flight while the current one votes.  The bins stay in registers, so the lists are read from HBM exactly once.
// Sweep 1: atomicOr into a "bin seen" bit plane; a hit that finds its bit already set is a repeat and goes (through a
// small LDS queue, inserted by the whole wave) into the small exact table.  Sweep 2 (registers only): every hit that
// was the first on its bit adds its vote if -- and only if -- its bin made it into the table (pre-filtered through a
// bit plane of the table keys that reuses the plane memory).  A bin with >= 2 votes has all but its first vote
// inserted in sweep 1 and the first one added in sweep 2: exact; bins with a single vote are dropped (never
// candidates when the final threshold exceeds 1), bit collisions only cost a spurious 1-vote entry.
// items per lane (template parameter of the kernel): 12 -> up to 768 segments (~4 900 typical hits, 150 bp reads vs a
// human-size index), 24 -> 1 536 segments (250 bp reads)
constexpr int kCsFastItemsShort = 12, kCsFastItemsLong = 24;
constexpr int kCsFastDepth = 2;    // segments in flight per lane
#ifndef NGM_CS_FAST2_DEPTH
#define NGM_CS_FAST2_DEPTH 3
#endif
constexpr int kCsFast2Depth = NGM_CS_FAST2_DEPTH;  // ... of the two-wave kernel (fewer registers per lane: room for one more)
// LDS queue: as many entries as the table may hold keys (3/4 of its slots) -- sweep 2 queues at most one hit per key;
// sweep 1 flushes whenever more than 96 repeats are waiting

struct __attribute__((packed, aligned(4))) CsU4 { uint32_t x, y, z, w; };

template <int kCsFastItems, typename ItemT>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 3))) void cs_fast_kernel(CsArgs A) {
	constexpr uint32_t kCsFastItemCap = (uint32_t) kCsFastItems * 64u;
	extern __shared__ __attribute__((aligned(16))) uint32_t cs_lds[];
	const int lane = threadIdx.x;
	const int read = blockIdx.x;
	const int k = A.k;
	uint32_t *l_start = cs_lds;
	uint16_t *l_len = (uint16_t *) (cs_lds + A.lists_cap);   // [lists_cap] (lists_cap is even)
	uint8_t *l_code = (uint8_t *) (l_len + A.lists_cap);
	ItemT *l_items = (ItemT *) ((uint32_t *) l_code + (A.q + 3) / 4);
	uint32_t *plane = (uint32_t *) (l_items + kCsFastItemCap);  // kCsFastItemCap is a multiple of 64: stays 4-byte aligned
	const uint32_t plane_words = A.plane_bits >> 5;
	uint32_t *t_keys = plane + plane_words;
	const int log2_slots = A.log2_slots;
	const uint32_t n_slots = 1u << log2_slots;
	uint32_t *t_votes = t_keys + n_slots;
	uint32_t *s_queue = t_votes + n_slots;
	const uint32_t kCsFastQueue = (n_slots * 3u) / 4u;
	for (uint32_t s = lane; s < n_slots; s += 64) { t_keys[s] = 0xFFFFFFFFu; t_votes[s] = 0; }
	for (uint32_t s = lane; s < plane_words; s += 64) plane[s] = 0;

	const bool diag = A.phase_cycles && (read & 255) == 0;  // sampled: the global atomics would serialise otherwise
	const unsigned long long c0 = diag ? wall_clock64() : 0ull;
	const CsRead R = cs_prepare<true, ItemT, true, uint16_t>(A, read, lane, l_start, l_len, l_code, l_items, kCsFastItemCap);
	const uint32_t H = R.H;
	const int L = R.L;
	if (H > A.hit_cap || R.n_items > kCsFastItemCap) { cs_enqueue(A, read, lane, R); return; }
	__syncthreads();
	const unsigned long long c1 = diag ? wall_clock64() : 0ull;

	const uint32_t pbits = A.plane_bits;
	const int hs = 32 - log2_slots;
	const uint32_t n_items = R.n_items;

	// wave-uniform bookkeeping lives in registers: queue length, distinct keys in the table, abort flag
	uint32_t q_len = 0, n_keys = 0;
	bool abort_fast = false;
	// queue slots for this lane's `mine` entries: exclusive prefix over the lanes (no LDS counter, no same-address atomics)
	auto reserve = [&](uint32_t mine) -> uint32_t {  // mine <= 8 * kCsFastItems <= 192
		uint32_t total;
		const uint32_t base = q_len + wave_prefix_small<8>(mine, total);
		q_len += total;
		return base;
	};
	// inserts the queued entries (bin | strand << 31), one per lane per round
	auto flush_inserts = [&]() {
		__syncthreads();
		const uint32_t nq = min(q_len, kCsFastQueue);
		if (n_keys + nq >= n_slots) abort_fast = true;  // could fill the table completely: leave the read to the exact path
		uint32_t fresh = 0;
		if (!abort_fast) for (uint32_t i = lane; i < nq; i += 64) {
			const uint32_t e = s_queue[i];
			const uint32_t bin = e & 0x3FFFFFFFu;
			uint32_t slot = (bin * 2654435761u) >> hs;
			for (;;) {
				const uint32_t prev = atomicCAS(&t_keys[slot], 0xFFFFFFFFu, bin);
				if (prev == bin) break;
				if (prev == 0xFFFFFFFFu) { ++fresh; break; }
				slot = (slot + 1) & (n_slots - 1);
			}
			atomicAdd(&t_votes[slot], (e & 0x80000000u) ? 0x10000u : 1u);
		}
		{ uint32_t total; (void) wave_prefix_small<4>(fresh, total); n_keys += total; }  // fresh <= 12 per lane
		if (n_keys > (n_slots * 3u) / 4u) abort_fast = true;  // probing gets slow and the spurious entries too many
		__syncthreads();
		q_len = 0;
	};

	// item -> (hit count << 16 | correction, strand in bit 31) and its positions
	auto fetch = [&](int it, CsU4 (&d)[kCsSeg / 4]) -> uint32_t {
		const uint32_t idx = (uint32_t) it * 64u + (uint32_t) lane;
		uint32_t meta = 0;
		if (idx < n_items) {
			const uint32_t item = l_items[idx];
			const uint32_t li = CsItem<ItemT>::list(item), sg = CsItem<ItemT>::seg(item);
			const uint32_t cnt = min((uint32_t) kCsSeg, (uint32_t) l_len[li] - sg * kCsSeg);
			const CsU4 *src = reinterpret_cast<const CsU4 *>(A.buckets + l_start[li] + sg * kCsSeg);
#pragma unroll
			for (int v = 0; v < kCsSeg / 4; ++v) if ((uint32_t) (4 * v) < cnt) d[v] = src[v];
			const int p = (int) (li >> 1);
			// diagonal of the hit (CS.cpp:140-142); bit 31 = reverse-complement list
			meta = (cnt << 16) | ((li & 1u) ? ((uint32_t) (L - (p + k)) | 0x80000000u) : (uint32_t) p);
		}
		return meta;
	};

	uint32_t bins[kCsFastItems * kCsSeg];  // bin | first-on-its-bit << 30 | reverse strand << 31 ; 0 = empty slot
	// ring of kCsFastDepth items in flight per lane: the position loads of items it+1 .. it+depth-1 are outstanding
	// while item it votes (the loop is fully unrolled, so the ring index is a compile-time constant)
	constexpr int DEPTH = kCsFastDepth;
	CsU4 ring[DEPTH][kCsSeg / 4];
	uint32_t rmeta[DEPTH];
#pragma unroll
	for (int d = 0; d < DEPTH - 1; ++d) rmeta[d] = fetch(d, ring[d]);
#pragma unroll
	for (int it = 0; it < kCsFastItems; ++it) {
		if ((uint32_t) it * 64u >= n_items) {  // wave-uniform
#pragma unroll
			for (int j = 0; j < kCsSeg; ++j) bins[it * kCsSeg + j] = 0;
			continue;
		}
		if (it + DEPTH - 1 < kCsFastItems) rmeta[(it + DEPTH - 1) % DEPTH] = fetch(it + DEPTH - 1, ring[(it + DEPTH - 1) % DEPTH]);
		const uint32_t meta = rmeta[it % DEPTH];
		CsU4 (&cur)[kCsSeg / 4] = ring[it % DEPTH];
		const uint32_t cnt = (meta >> 16) & 0x1Fu, corr = meta & 0xFFFFu, rev = meta & 0x80000000u;
		// branch-free per hit (the loop is unrolled 12 x 8 times; the kernel has to stay small enough for the instruction
		// cache): empty slots vote with an all-zero mask (a no-op on whatever plane word their garbage position selects)
		uint32_t old[kCsSeg], msk[kCsSeg], ent[kCsSeg];
#pragma unroll
		for (int j = 0; j < kCsSeg; ++j) {
			const uint32_t pos = (j & 3) == 0 ? cur[j >> 2].x : (j & 3) == 1 ? cur[j >> 2].y : (j & 3) == 2 ? cur[j >> 2].z : cur[j >> 2].w;
			const bool valid = (uint32_t) j < cnt;
			const uint32_t bin = ((pos - corr) >> A.bin_shift) & 0x3FFFFFFFu;
			const uint32_t b = __umulhi(bin * 0x9E3779B1u, pbits);
			msk[j] = valid ? (1u << (b & 31)) : 0u;
			old[j] = atomicOr(&plane[b >> 5], msk[j]);
			ent[j] = bin | rev;
		}
		uint32_t ndup = 0;
#pragma unroll
		for (int j = 0; j < kCsSeg; ++j) ndup += (old[j] & msk[j]) ? 1u : 0u;
		uint32_t qb;
		{ uint32_t total; qb = q_len + wave_prefix_small<4>(ndup, total); q_len += total; }  // ndup <= 8
#pragma unroll
		for (int j = 0; j < kCsSeg; ++j) {
			const bool dup = (old[j] & msk[j]) != 0u;           // implies a valid slot
			const bool first = msk[j] != 0u && !dup;            // valid and first on its bit
			if (dup) { if (qb < kCsFastQueue) s_queue[qb] = ent[j]; ++qb; }
			bins[it * kCsSeg + j] = first ? (ent[j] | 0x40000000u) : 0u;  // repeats voted in sweep 1: nothing left to do
		}
		const uint32_t fill = q_len;
		if (fill > kCsFastQueue) abort_fast = true;  // more repeats than the queue holds: leave it to the exact path
		// insert when the next item round (typically ~100 repeats) might not fit any more, and after the last one
		if (fill > 96u || (uint32_t) (it + 1) * 64u >= n_items || it + 1 == kCsFastItems) flush_inserts();  // small batches: the table-capacity guard of flush_inserts stays loose
	}
	const unsigned long long c2 = diag ? wall_clock64() : 0ull;
	if (abort_fast) { cs_enqueue(A, read, lane, R); return; }  // not provably exact here

	// sweep 2: plane := bits of the bins that are in the table; first-on-bit hits whose bit is set are queued, then added
	for (uint32_t s = lane; s < plane_words; s += 64) plane[s] = 0;
	__syncthreads();
	for (uint32_t s = lane; s < n_slots; s += 64) {
		const uint32_t key = t_keys[s];
		if (key != 0xFFFFFFFFu) {
			const uint32_t b = __umulhi(key * 0x9E3779B1u, pbits);
			atomicOr(&plane[b >> 5], 1u << (b & 31));
		}
	}
	__syncthreads();
	uint32_t nhit = 0;
	uint32_t wmask[kCsFastItems];
#pragma unroll
	for (int it = 0; it < kCsFastItems; ++it) {
		wmask[it] = 0;
		if ((uint32_t) it * 64u >= n_items) continue;
#pragma unroll
		for (int j = 0; j < kCsSeg; ++j) {
			const uint32_t e = bins[it * kCsSeg + j];
			const uint32_t b = __umulhi((e & 0x3FFFFFFFu) * 0x9E3779B1u, pbits);
			const uint32_t w = (plane[b >> 5] >> (b & 31)) & (e >> 30) & 1u;  // bit 30 = first on its bit (0 for empty slots)
			wmask[it] |= w << j;
		}
		nhit += __popc(wmask[it]);
	}
	uint32_t qb = reserve(nhit);
#pragma unroll
	for (int it = 0; it < kCsFastItems; ++it) {
		if (wmask[it])
#pragma unroll
			for (int j = 0; j < kCsSeg; ++j) if ((wmask[it] >> j) & 1u) { if (qb < kCsFastQueue) s_queue[qb] = bins[it * kCsSeg + j]; ++qb; }
	}
	__syncthreads();
	if (q_len > kCsFastQueue) abort_fast = true;
	{
		// add the votes of the queued first hits whose bin is in the table (a set bit may also be a collision)
		const uint32_t nq = min(q_len, kCsFastQueue);
		for (uint32_t i = lane; i < nq; i += 64) {
			const uint32_t e = s_queue[i];
			const uint32_t bin = e & 0x3FFFFFFFu;
			uint32_t slot = (bin * 2654435761u) >> hs;
			for (;;) {
				const uint32_t key = t_keys[slot];
				if (key == bin) { atomicAdd(&t_votes[slot], (e & 0x80000000u) ? 0x10000u : 1u); break; }
				if (key == 0xFFFFFFFFu) break;
				slot = (slot + 1) & (n_slots - 1);
			}
		}
	}
	__syncthreads();
	const unsigned long long c3 = diag ? wall_clock64() : 0ull;
	if (abort_fast) { cs_enqueue(A, read, lane, R); return; }
	if (!cs_finish<kCsFast>(A, read, lane, R, t_keys, t_votes, n_slots)) cs_enqueue(A, read, lane, R);
	if (diag && lane == 0) {  // diagnostics: 100 MHz ticks spent per phase, summed over the sampled reads
		atomicAdd(&A.phase_cycles[0], c1 - c0); atomicAdd(&A.phase_cycles[1], c2 - c1); atomicAdd(&A.phase_cycles[2], c3 - c2);
		atomicAdd(&A.phase_cycles[3], wall_clock64() - c3);
	}
}

// ---- FAST path, T waves per read --------------------------------------------------------------------------------------
// The same algorithm as cs_fast_kernel with the read's k-mers and work items dealt to the lanes of T waves (2-4) that share
// the plane and the table (LDS atomics do not care which wave votes): half the dependent chain per wave and twice the waves
// per CU for the same LDS footprint -- the kernel is bound by per-read latency x reads in flight.  Each wave has
// its own half of the repeat queue and inserts it on its own (wave-level ordering only), so sweep 1 runs without block barriers.
template <int T, int kItems, typename ItemT>
__global__ __launch_bounds__(T * 64) __attribute__((amdgpu_waves_per_eu((T == 3 && kItems <= 4) ? 8 : 1))) void cs_fast2_kernel(CsArgs A) {
	constexpr int NT = T * 64;
	constexpr uint32_t kItemCap = (uint32_t) kItems * (uint32_t) NT;
	constexpr int RB = T >= 3 ? 1 : 2;  // 64-k-mer chunks per wave per trip of the lists phase (a trip covers T * RB * 64 k-mers)
	extern __shared__ __attribute__((aligned(16))) uint32_t cs_lds[];
	__shared__ uint32_t s_tot[T * RB][3];  // per chunk: hits, segments, k-mers looked up
	__shared__ int s_len[T];
	__shared__ uint32_t s_abort, s_nkeys;
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const int read = blockIdx.x;
	const int k = A.k;
	uint32_t *l_start = cs_lds;
	uint16_t *l_len = (uint16_t *) (cs_lds + A.lists_cap);
	uint8_t *l_code = (uint8_t *) (l_len + A.lists_cap);
	ItemT *l_items = (ItemT *) ((uint32_t *) l_code + (A.q + 3) / 4);
	uint32_t *plane = (uint32_t *) (l_items + kItemCap);
	const uint32_t plane_words = A.plane_bits >> 5;
	uint32_t *t_keys = plane + plane_words;
	const int log2_slots = A.log2_slots;
	const uint32_t n_slots = 1u << log2_slots;
	uint32_t *t_votes = t_keys + n_slots;
	const uint32_t q_cap = ((n_slots * 3u) / 4u) / (uint32_t) T;  // per wave
	uint32_t *my_queue = t_votes + n_slots + (uint32_t) wv * q_cap;
	const uint8_t *rp = A.reads + (size_t) read * A.q;
	const uint32_t ch0 = tid < A.q ? (uint32_t) rp[tid] : 0u;  // on its way while the table is cleared
	for (uint32_t s = tid; s < n_slots; s += NT) { t_keys[s] = 0xFFFFFFFFu; t_votes[s] = 0; }
	for (uint32_t s = tid; s < plane_words; s += NT) plane[s] = 0;
	if (tid == 0) { s_abort = 0; s_nkeys = 0; }
	const bool diag = A.phase_cycles && (read & 255) == 0;
	const unsigned long long c0 = diag ? wall_clock64() : 0ull;

	// 1. codes, k-mers, lists (cs_prepare's steps; the prefix sums cross the two waves through s_tot)
	CsRead R;
	{
		int first_nul = A.q;
		for (int i = tid; i < A.q; i += NT) {
			const uint32_t ch = i == tid ? ch0 : (uint32_t) rp[i];
			uint8_t code;
			if (ch == 0) { code = 255; first_nul = min(first_nul, i); }
			else if (ch == 'N') code = 4;
			else code = (uint8_t) ((ch >> 1) & 3u);
			l_code[i] = code;
		}
		first_nul = wave_reduce_min(first_nul);
		if (lane == 0) s_len[wv] = first_nul;
		__syncthreads();
		R.L = s_len[0];
#pragma unroll
		for (int w2 = 1; w2 < T; ++w2) R.L = min(R.L, s_len[w2]);
		const int L = R.L, n_kmers = L - k + 1;
		R.n_lists = n_kmers > 0 ? 2 * n_kmers : 0;
		uint32_t carry = 0, carry_s = 0, n_valid = 0;
		for (int base = 0; base < n_kmers; base += NT * RB) {
			uint2 hf[RB], hr[RB];
			uint32_t kf[RB], kr[RB];
			bool valid[RB];
#pragma unroll
			for (int r = 0; r < RB; ++r) {
				const int p = base + (r * T + wv) * 64 + lane;
				valid[r] = false;
				hf[r] = make_uint2(0, 0); hr[r] = make_uint2(0, 0); kf[r] = 0; kr[r] = 0;
				if (p < n_kmers) {
					bool v = true;
					uint32_t kmer = 0;
					for (int j = 0; j < k; ++j) {
						const uint32_t c = l_code[p + j];
						v = v && (c < 4);
						kmer = (kmer << 2) | (c & 3u);
					}
					if (v && p + k == L && p >= 1 && l_code[p - 1] == 4 && (p == 1 || l_code[p - 2] == 4)) v = false;  // CSstatic.cpp:30-41, see cs_prepare
					valid[r] = v;
					if (v) {
						kf[r] = kmer; kr[r] = cs_revcomp(kmer, k);
						hf[r] = *reinterpret_cast<const uint2 *>(A.buckets + ((size_t) kf[r] << A.bucket_log2_words));
						hr[r] = *reinterpret_cast<const uint2 *>(A.buckets + ((size_t) kr[r] << A.bucket_log2_words));
					}
				}
			}
			uint32_t cf[RB], cr[RB], sf[RB], sr[RB], ex[RB], ex_s[RB];
#pragma unroll
			for (int r = 0; r < RB; ++r) {
				cf[r] = cr[r] = sf[r] = sr[r] = 0;
				const uint32_t nf = hf[r].x & kCsHdrCountMask, nr = hr[r].x & kCsHdrCountMask;
				if (valid[r] && (int) (nf + nr) < A.max_kfreq) {  // CS.cpp:122
					cf[r] = nf; sf[r] = (hf[r].x & kCsHdrOverflow) ? A.pos_base + hf[r].y : (kf[r] << A.bucket_log2_words) + 1u;
					cr[r] = nr; sr[r] = (hr[r].x & kCsHdrOverflow) ? A.pos_base + hr[r].y : (kr[r] << A.bucket_log2_words) + 1u;
				}
				const uint32_t both = cf[r] + cr[r], segs = (cf[r] + kCsSeg - 1) / kCsSeg + (cr[r] + kCsSeg - 1) / kCsSeg;
				const uint32_t incl = wave_inclusive_scan(both, lane), incl_s = wave_inclusive_scan(segs, lane);
				ex[r] = incl - both; ex_s[r] = incl_s - segs;
				const uint32_t nv = (uint32_t) __popcll(__ballot(valid[r]));
				if (lane == 63) { s_tot[r * T + wv][0] = incl; s_tot[r * T + wv][1] = incl_s; s_tot[r * T + wv][2] = nv; }
			}
			__syncthreads();
#pragma unroll
			for (int r = 0; r < RB; ++r) {
				const int c = r * T + wv, p = base + c * 64 + lane;
				uint32_t o = carry_s + ex_s[r];
				for (int c2 = 0; c2 < c; ++c2) o += s_tot[c2][1];
				if (p < n_kmers) {
					l_start[2 * p] = sf[r]; l_len[2 * p] = (uint16_t) cf[r];
					l_start[2 * p + 1] = sr[r]; l_len[2 * p + 1] = (uint16_t) cr[r];
					const uint32_t nsf = (cf[r] + kCsSeg - 1) / kCsSeg, nsr = (cr[r] + kCsSeg - 1) / kCsSeg;
					for (uint32_t sg = 0; sg < nsf; ++sg, ++o) if (o < kItemCap) l_items[o] = CsItem<ItemT>::make((uint32_t) (2 * p), sg);
					for (uint32_t sg = 0; sg < nsr; ++sg, ++o) if (o < kItemCap) l_items[o] = CsItem<ItemT>::make((uint32_t) (2 * p + 1), sg);
				}
			}
#pragma unroll
			for (int c = 0; c < T * RB; ++c) { carry += s_tot[c][0]; carry_s += s_tot[c][1]; n_valid += s_tot[c][2]; }
			__syncthreads();
		}
		R.H = carry; R.n_valid = n_valid; R.n_items = carry_s;
	}
	const uint32_t H = R.H;
	const int L = R.L;
	if (H > A.hit_cap || R.n_items > kItemCap) { if (wv == 0) cs_enqueue(A, read, lane, R); return; }
	const unsigned long long c1 = diag ? wall_clock64() : 0ull;

	const uint32_t pbits = A.plane_bits;
	const int hs = 32 - log2_slots;
	const uint32_t n_items = R.n_items;
	// LDS operations of one wave complete in program order: the queue hand-over inside a wave needs no hardware barrier, only
	// the compiler kept from moving the accesses
	auto wave_sync = [] { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); };
	uint32_t q_len = 0;        // this wave's queue (wave-uniform)
	bool abort_fast = false;
	auto flush_inserts = [&]() {
		wave_sync();
		const uint32_t nq = min(q_len, q_cap);
		uint32_t fresh = 0;
		if (!abort_fast) for (uint32_t i = lane; i < nq; i += 64) {
			const uint32_t e = my_queue[i];
			const uint32_t bin = e & 0x3FFFFFFFu;
			uint32_t slot = (bin * 2654435761u) >> hs;
			for (uint32_t probes = 0;; ++probes) {
				if (probes >= n_slots) { slot = 0xFFFFFFFFu; break; }  // the other wave filled the table meanwhile
				const uint32_t prev = atomicCAS(&t_keys[slot], 0xFFFFFFFFu, bin);
				if (prev == bin) break;
				if (prev == 0xFFFFFFFFu) { ++fresh; break; }
				slot = (slot + 1) & (n_slots - 1);
			}
			if (slot != 0xFFFFFFFFu) atomicAdd(&t_votes[slot], (e & 0x80000000u) ? 0x10000u : 1u); else s_abort = 1u;
		}
		uint32_t total;
		(void) wave_prefix_small<4>(fresh, total);  // fresh <= q_cap / 64 < 16
		uint32_t before = 0;
		if (lane == 0 && total) before = atomicAdd(&s_nkeys, total);
		before = wave_first(before);
		if (before + total > (n_slots * 3u) / 4u) { abort_fast = true; if (lane == 0) s_abort = 1u; }  // probing gets slow, the spurious entries too many
		wave_sync();
		q_len = 0;
	};
	auto fetch = [&](int it, CsU4 (&d)[kCsSeg / 4]) -> uint32_t {
		const uint32_t idx = (uint32_t) it * (uint32_t) NT + (uint32_t) tid;
		uint32_t meta = 0;
		if (idx < n_items) {
			const uint32_t item = l_items[idx];
			const uint32_t li = CsItem<ItemT>::list(item), sg = CsItem<ItemT>::seg(item);
			const uint32_t cnt = min((uint32_t) kCsSeg, (uint32_t) l_len[li] - sg * kCsSeg);
			const CsU4 *src = reinterpret_cast<const CsU4 *>(A.buckets + l_start[li] + sg * kCsSeg);
#pragma unroll
			for (int v = 0; v < kCsSeg / 4; ++v) if ((uint32_t) (4 * v) < cnt) d[v] = src[v];
			const int p = (int) (li >> 1);
			meta = (cnt << 16) | ((li & 1u) ? ((uint32_t) (L - (p + k)) | 0x80000000u) : (uint32_t) p);  // CS.cpp:140-142
		}
		return meta;
	};

	// sweep 1 (see cs_fast_kernel)
	uint32_t bins[kItems * kCsSeg];
	constexpr int DEPTH = kCsFast2Depth;
	CsU4 ring[DEPTH][kCsSeg / 4];
	uint32_t rmeta[DEPTH];
#pragma unroll
	for (int d = 0; d < DEPTH - 1; ++d) rmeta[d] = fetch(d, ring[d]);
#pragma unroll
	for (int it = 0; it < kItems; ++it) {
		if ((uint32_t) it * (uint32_t) NT >= n_items) {  // block-uniform
#pragma unroll
			for (int j = 0; j < kCsSeg; ++j) bins[it * kCsSeg + j] = 0;
			continue;
		}
		if (it + DEPTH - 1 < kItems) rmeta[(it + DEPTH - 1) % DEPTH] = fetch(it + DEPTH - 1, ring[(it + DEPTH - 1) % DEPTH]);
		const uint32_t meta = rmeta[it % DEPTH];
		CsU4 (&cur)[kCsSeg / 4] = ring[it % DEPTH];
		const uint32_t cnt = (meta >> 16) & 0x1Fu, corr = meta & 0xFFFFu, rev = meta & 0x80000000u;
		uint32_t old[kCsSeg], msk[kCsSeg], ent[kCsSeg];
#pragma unroll
		for (int j = 0; j < kCsSeg; ++j) {
			const uint32_t pos = (j & 3) == 0 ? cur[j >> 2].x : (j & 3) == 1 ? cur[j >> 2].y : (j & 3) == 2 ? cur[j >> 2].z : cur[j >> 2].w;
			const bool valid = (uint32_t) j < cnt;
			const uint32_t bin = ((pos - corr) >> A.bin_shift) & 0x3FFFFFFFu;
			const uint32_t b = __umulhi(bin * 0x9E3779B1u, pbits);
			msk[j] = valid ? (1u << (b & 31)) : 0u;
			old[j] = atomicOr(&plane[b >> 5], msk[j]);
			ent[j] = bin | rev;
		}
		uint32_t ndup = 0;
#pragma unroll
		for (int j = 0; j < kCsSeg; ++j) ndup += (old[j] & msk[j]) ? 1u : 0u;
		uint32_t qb;
		{ uint32_t total; qb = q_len + wave_prefix_small<4>(ndup, total); q_len += total; }
#pragma unroll
		for (int j = 0; j < kCsSeg; ++j) {
			const bool first = msk[j] != 0u && (old[j] & msk[j]) == 0u;  // valid and first on its bit
			bins[it * kCsSeg + j] = first ? (ent[j] | 0x40000000u) : 0u;    // repeats vote in sweep 1: nothing left to do for them
		}
		// the repeats go through the queue: inserted when a good batch is waiting (a round adds ~40 per wave) and after the last
		// round; when one round brings more than the queue holds (repetitive reads) it is filled and emptied window by window
		const bool last_round = (uint32_t) (it + 1) * (uint32_t) NT >= n_items || it + 1 == kItems;
		uint32_t window = 0;  // queue coordinates [window, window + q_cap) are in the queue right now
		for (;;) {
			uint32_t at = qb;
#pragma unroll
			for (int j = 0; j < kCsSeg; ++j) if ((old[j] & msk[j]) != 0u) { if (at - window < q_cap) my_queue[at - window] = ent[j]; ++at; }
			const bool more = q_len - window > q_cap;
			if (more || q_len - window > (T >= 4 ? 0u : q_cap / 2u) || last_round) {
				const uint32_t all = q_len;
				q_len = min(all - window, q_cap);
				flush_inserts();  // leaves q_len = 0
				if (!more) break;
				q_len = all; window += q_cap;
				continue;
			}
			q_len -= window;
			break;
		}
	}
	__syncthreads();
	const unsigned long long c2 = diag ? wall_clock64() : 0ull;
	if (s_abort) { if (wv == 0) cs_enqueue(A, read, lane, R); return; }  // not provably exact here

	// sweep 2
	for (uint32_t s = tid; s < plane_words; s += NT) plane[s] = 0;
	__syncthreads();
	for (uint32_t s = tid; s < n_slots; s += NT) {
		const uint32_t key = t_keys[s];
		if (key != 0xFFFFFFFFu) {
			const uint32_t b = __umulhi(key * 0x9E3779B1u, pbits);
			atomicOr(&plane[b >> 5], 1u << (b & 31));
		}
	}
	__syncthreads();
	uint32_t nhit = 0;
	uint32_t wmask[kItems];
#pragma unroll
	for (int it = 0; it < kItems; ++it) {
		wmask[it] = 0;
		if ((uint32_t) it * (uint32_t) NT >= n_items) continue;
#pragma unroll
		for (int j = 0; j < kCsSeg; ++j) {
			const uint32_t e = bins[it * kCsSeg + j];
			const uint32_t b = __umulhi((e & 0x3FFFFFFFu) * 0x9E3779B1u, pbits);
			const uint32_t w = (plane[b >> 5] >> (b & 31)) & (e >> 30) & 1u;
			wmask[it] |= w << j;
		}
		nhit += __popc(wmask[it]);
	}
	{
		// the marked hits go through this wave's queue to be spread over its lanes: add the vote where the bin is in the table (a
		// set bit may also be a collision).  One hit per table key plus the collisions: more than the per-wave queue holds for the
		// reads with the fullest tables, so window by window again
		uint32_t total;
		const uint32_t qb = wave_prefix_small<8>(nhit, total);  // nhit <= 8 * kItems <= 96
		for (uint32_t window = 0; window < total; window += q_cap) {
			uint32_t at = qb;
#pragma unroll
			for (int it = 0; it < kItems; ++it) {
				if (wmask[it])
#pragma unroll
					for (int j = 0; j < kCsSeg; ++j) if ((wmask[it] >> j) & 1u) { if (at - window < q_cap) my_queue[at - window] = bins[it * kCsSeg + j]; ++at; }
			}
			wave_sync();
			const uint32_t nq = min(total - window, q_cap);
			for (uint32_t i = lane; i < nq; i += 64) {
				const uint32_t e = my_queue[i];
				const uint32_t bin = e & 0x3FFFFFFFu;
				uint32_t slot = (bin * 2654435761u) >> hs;
				for (;;) {  // the table is at most 3/4 full here
					const uint32_t key = t_keys[slot];
					if (key == bin) { atomicAdd(&t_votes[slot], (e & 0x80000000u) ? 0x10000u : 1u); break; }
					if (key == 0xFFFFFFFFu) break;
					slot = (slot + 1) & (n_slots - 1);
				}
			}
			wave_sync();
		}
	}
	__syncthreads();
	const unsigned long long c3 = diag ? wall_clock64() : 0ull;
	if (wv != 0) return;
	if (s_abort) { cs_enqueue(A, read, lane, R); return; }
	if (!cs_finish<kCsFast>(A, read, lane, R, t_keys, t_votes, n_slots)) cs_enqueue(A, read, lane, R);
	if (diag && lane == 0) {
		atomicAdd(&A.phase_cycles[0], c1 - c0); atomicAdd(&A.phase_cycles[1], c2 - c1); atomicAdd(&A.phase_cycles[2], c3 - c2);
		atomicAdd(&A.phase_cycles[3], wall_clock64() - c3);
	}
}

// ---- EXACT paths: every hit goes into an open-addressing table (LDS, or global memory for very repetitive reads) --
template <int MODE>
__device__ __forceinline__ void cs_exact_read(const CsArgs &A, const int item, const int lane) {
	extern __shared__ __attribute__((aligned(16))) uint32_t cs_lds[];
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

	// bisulfite mapping: the lists of the k-mer VARIANTS, kCsBsChunk variants at a time (votes commute); behind the codes in LDS:
	// l_vbase [q + 1], l_vpos [kCsBsChunk]
	uint32_t *l_vbase = nullptr;
	uint16_t *l_vpos = nullptr;
	CsBsRead B{};
	CsRead R;
	if (A.bs) {
		l_vbase = (uint32_t *) l_code + (A.q + 3) / 4;
		l_vpos = (uint16_t *) (l_vbase + A.q + 1);
		uint32_t *after = (uint32_t *) (l_vpos + kCsBsChunk);
		if (MODE != kCsExactGlobal) { t_keys = after; t_votes = t_keys + (1u << log2_slots); }
		B = cs_bs_scan(A, read, lane, l_code, l_vbase);
		uint32_t Hs = 0;   // counting pass: the table is sized from the read's hits
		for (uint32_t v0 = 0; v0 < B.V; v0 += kCsBsChunk) {
			Hs += cs_bs_chunk<false>(A, B, lane, l_code, l_vbase, v0, min(B.V, v0 + (uint32_t) kCsBsChunk), 0u, l_start, l_pref, l_vpos);
			__syncthreads();
		}
		R.L = B.L; R.n_lists = 0; R.H = Hs; R.n_valid = B.n_valid; R.n_items = 0;
	} else R = cs_prepare<false>(A, read, lane, l_start, l_pref, l_code);
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

	auto vote = [&](uint32_t pos, int p, bool rev) {
		const uint32_t correction = rev ? (uint32_t) (L - (p + k)) : (uint32_t) p;  // CS.cpp:140-142
		const uint32_t bin = (pos - correction) >> A.bin_shift;
		uint32_t slot = (bin * 2654435761u) >> (32 - log2_slots);
		for (;;) {
			const uint32_t prev = atomicCAS(&t_keys[slot], 0xFFFFFFFFu, bin);
			if (prev == bin || prev == 0xFFFFFFFFu) break;
			slot = (slot + 1) & (n_slots - 1);
		}
		atomicAdd(&t_votes[slot], rev ? 0x10000u : 1u);
	};
	if (A.bs) {
		for (uint32_t v0 = 0; v0 < B.V; v0 += kCsBsChunk) {
			const uint32_t v1 = min(B.V, v0 + (uint32_t) kCsBsChunk);
			const uint32_t hc = cs_bs_chunk<false>(A, B, lane, l_code, l_vbase, v0, v1, 0u, l_start, l_pref, l_vpos);
			__syncthreads();
			cs_for_each_hit(A.positions, l_start, l_pref, (int) (2 * (v1 - v0)), hc, lane, [&](uint32_t pos, int li, uint32_t) { vote(pos, (int) l_vpos[li >> 1], (li & 1) != 0); });
			__syncthreads();
		}
	} else {
		cs_for_each_hit(A.positions, l_start, l_pref, R.n_lists, H, lane, [&](uint32_t pos, int li, uint32_t) { vote(pos, li >> 1, (li & 1) != 0); });
	}
	__syncthreads();
	if (MODE == kCsExactGlobal) __threadfence_block();
	(void) cs_finish<MODE>(A, read, lane, R, t_keys, t_votes, n_slots);
}
template <int MODE>
__global__ __launch_bounds__(64) void cs_kernel(CsArgs A) {
	const int lane = threadIdx.x;
	if (!A.n_list_dev) { cs_exact_read<MODE>(A, (int) blockIdx.x, lane); return; }
	const int n_items = (int) *A.n_list_dev;
	for (int item = blockIdx.x; item < n_items; item += (int) gridDim.x) {
		cs_exact_read<MODE>(A, item, lane);
		__syncthreads();
	}
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
constexpr uint32_t kCsOrderMaxHits = 4096;  // least time line entries in LDS (CsArgs::order_max_hits); reads with more hits use a slice of global memory
constexpr uint32_t kCsOrderUnknown = 0xFFFFFFFFu;
constexpr int kCsOrderThreads = 256;
__host__ __device__ inline uint32_t cs_order_tau(int lists_cap) { return (uint32_t) lists_cap / 2u + 64u; }   // most votes of one (bin, strand) the replay follows: a list per k-mer and strand (+ the odd second hit of one list in a bin)

// GLOBAL (the exact fall-back for the reads the LDS replay leaves: more hits than its time line, more repeated bins than its
// 1 024-slot table -- common on a genome with a heavy-tailed k-mer spectrum): the same replay with the time line AND the table of
// tracked bins in a per-read slice of global memory sized from the read's hits (A.ovf_table_off / A.ovf_log2 / A.gtable_keys), and
// 32-bit hit times (l_time) instead of the 16-bit ones packed into l_pref.  Nothing is given up but a bisulfite read with more
// k-mer variants than the list rows hold.
constexpr uint32_t kCsOrderStage = 2048;      // entries of a wave's staging buffer (GLOBAL)
constexpr int kCsOrderThreadsGlobal = 1024;   // the exact replay in global memory: every step of it waits for L2 -- four times the waves per read
template <bool GLOBAL>
__global__ __launch_bounds__(GLOBAL ? kCsOrderThreadsGlobal : kCsOrderThreads) void cs_order_kernel(CsArgs A, const uint32_t *__restrict__ cand_loc, const uint32_t *__restrict__ cand_sv,
		uint32_t *__restrict__ cand_rank) {
	extern __shared__ __attribute__((aligned(16))) uint32_t cs_lds[];
	__shared__ uint32_t s_keys;  // distinct tracked bins
	const unsigned long long t_block = A.phase_cycles ? wall_clock64() : 0ull;
	// four waves: the sweeps over the hits (most of the time) are spread over all of them, the sequential parts -- the time-ordered
	// compaction and the replay -- stay with wave 0; cs_prepare is run by every wave (same values, its barrier is the block's)
	constexpr int NT = GLOBAL ? kCsOrderThreadsGlobal : kCsOrderThreads;
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	auto og = [](const uint32_t *p) -> uint32_t { return GLOBAL ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p; };   // words updated by L2 atomics: not through L1
	const int read = (int) A.read_list[blockIdx.x];
	const int k = A.k;
	uint32_t *l_start = cs_lds;
	uint32_t *l_pref = cs_lds + A.lists_cap;
	uint8_t *l_code = (uint8_t *) (l_pref + A.lists_cap + 1);
	uint32_t *plane = (uint32_t *) l_code + (A.q + 3) / 4;
	constexpr uint32_t plane_words = 2048;
	const int log2_slots = GLOBAL ? (int) A.ovf_log2[blockIdx.x] : kCsOrderLog2Slots;
	const uint32_t n_slots = 1u << log2_slots;
	uint32_t *t_keys = GLOBAL ? A.gtable_keys + A.ovf_table_off[blockIdx.x] : plane + plane_words;
	uint32_t *t_votes = t_keys + n_slots;   // final votes: forward | reverse << 16
	uint32_t *t_run = t_votes + n_slots;    // votes so far during the replay
	uint32_t *t_rank = t_run + n_slots;
	uint32_t *t_cand = t_rank + n_slots;    // 1: the bin is one of the read's candidates (tracked even with a single vote)
	uint32_t *ev_at = GLOBAL ? t_cand + n_slots : t_cand + n_slots;     // [order_max_hits]: slot | strand << 31 of the hit at that time, or empty (LDS; reads with more hits: global memory)
	for (uint32_t s = tid; s < plane_words; s += NT) plane[s] = 0;
	for (uint32_t s = tid; s < n_slots; s += NT) { t_keys[s] = 0xFFFFFFFFu; t_votes[s] = 0; t_cand[s] = 0; }   // (t_run / t_rank: written for every slot in use before they are read, below)
	if (tid == 0) s_keys = 0;
	uint32_t *tm = ev_at + A.order_max_hits;   // [order_max_hits]: hit times by (slot, strand) (LDS; GLOBAL / big reads: behind the time line in global memory, below)
	uint32_t *seg_pref = GLOBAL ? plane + plane_words : tm + A.order_max_hits;
	uint32_t *l_time = seg_pref + A.lists_cap + 1 + (A.bs ? (size_t) A.q + 1 + A.lists_cap / 4 + 1 : 0);   // GLOBAL: [lists_cap] time of every list's first hit
	uint32_t *tau = l_time + (GLOBAL ? A.lists_cap : 0);   // [cs_order_tau(lists_cap)]
	const uint32_t n_tau = cs_order_tau(A.lists_cap);
	// GLOBAL: per wave a staging buffer for the hit times of 64 slots at a time (their segments are contiguous in tm: one coalesced copy
	// in, the segments sorted in LDS, one coalesced copy out -- a load per segment from L2 was 1 of the 1.5 ms of the replay's last phase)
	const uint32_t stage = GLOBAL ? A.order_gcap : 0u;   // (GLOBAL: order_gcap carries the staging entries per wave; 0: none -- bisulfite runs, whose list rows take the LDS)
	uint32_t *wbuf = tau + n_tau + (size_t) (threadIdx.x >> 6) * stage;
	const bool diag = A.phase_cycles && (blockIdx.x & 63) == 0;
	unsigned long long ck[6] = {0, 0, 0, 0, 0, 0};
	if (diag) ck[0] = wall_clock64();  // [lists_cap + 1]: work items (8-hit list segments) in front of each list
	const uint32_t cb = A.cand_base[read], cn = A.cand_count[read];
	auto give_up = [&](uint32_t why, uint32_t hits, uint32_t keys) {
		for (uint32_t c = tid; c < cn; c += NT) cand_rank[cb + c] = kCsOrderUnknown;
		if (tid == 0 && A.order_info) { A.order_info[2 * blockIdx.x] = hits; A.order_info[2 * blockIdx.x + 1] = why | (keys << 8); }
	};
	CsRead R;
	const uint16_t *l_vpos = nullptr;   // bisulfite mapping: the read position of the k-mer behind list pair j
	if (A.bs) {
		// the lists of ALL k-mer variants at once, in the reference's order (their hits form one time line); every wave computes
		// the same values.  Reads with more variants than the LDS rows hold keep the position order.
		uint32_t *l_vbase = seg_pref + A.lists_cap + 1;
		uint16_t *vp = (uint16_t *) (l_vbase + A.q + 1);
		const CsBsRead B = cs_bs_scan(A, read, lane, l_code, l_vbase);
		if (2u * B.V > (uint32_t) A.lists_cap) { give_up(1u, 0u, 0u); return; }   // (block-uniform)
		uint32_t segs = 0;
		const uint32_t Hb = cs_bs_chunk<true>(A, B, lane, l_code, l_vbase, 0u, B.V, 0u, l_start, l_pref, vp, &segs);
		R.L = B.L; R.n_lists = (int) (2u * B.V); R.H = Hb; R.n_valid = B.n_valid; R.n_items = segs;
		l_vpos = vp;
		__syncthreads();
	} else R = cs_prepare<true, uint32_t>(A, read, lane, l_start, l_pref, l_code, (uint32_t *) nullptr, 0);
	const uint32_t H = R.H;
	const int L = R.L;
	if (diag) ck[1] = wall_clock64();
	// very repetitive reads: the time line moves to global memory; the 16-bit list offsets of l_pref bound that at 65 535 hits
	const bool big = !GLOBAL && H > A.order_max_hits;
	if (big) {
		if (!A.order_scratch || H > A.order_gcap || H >= 65536u) { give_up(2u, H, 0u); return; }
		ev_at = A.order_scratch + (size_t) blockIdx.x * 2u * A.order_gcap;
		tm = ev_at + A.order_gcap;
	}
	if (GLOBAL) tm = ev_at + ((H + 63u) & ~63u);   // (the slice holds the table, the time line, the sorted times and the list of the slots in use: 6 * slots + 2 * (hits + 64) words)
	uint32_t *t_list = GLOBAL ? tm + ((H + 63u) & ~63u) : nullptr;   // GLOBAL: the slots in use, in no particular order (the table is a few per cent full: the passes below walk this list, not the table)
	__syncthreads();
	if (GLOBAL && wv == 1) {
		uint32_t carry = 0;
		for (int base = 0; base < R.n_lists; base += 64) {
			const int li = base + lane;
			const uint32_t cnt = li < R.n_lists ? (l_pref[li] & 0xFFFFu) : 0u;
			const uint32_t incl = wave_inclusive_scan(cnt, lane);
			if (li < R.n_lists) l_time[li] = carry + incl - cnt;
			carry += wave_last(incl);
		}
	}
	if (wv == 0) {
		uint32_t carry = 0;
		for (int base = 0; base < R.n_lists; base += 64) {
			const int li = base + lane;
			const uint32_t ns = li < R.n_lists ? ((l_pref[li] & 0xFFFFu) + kCsSeg - 1) / kCsSeg : 0u;
			const uint32_t incl = wave_inclusive_scan(ns, lane);
			if (li < R.n_lists) seg_pref[li] = carry + incl - ns;
			carry += wave_last(incl);
		}
		if (lane == 0) seg_pref[R.n_lists] = carry;
	}
	// the candidates themselves are tracked whatever their votes: with a final threshold <= 1 (few votes: sensitive settings,
	// diverged reads) single-vote bins are candidates too, and entered rList at their only hit
	{
		const uint32_t centre0 = A.bin_shift > 0 ? (1u << (A.bin_shift - 1)) : 0u;
		for (uint32_t c = tid; c < cn; c += NT) {
			const uint32_t bin = ((cand_loc[cb + c] - centre0) >> A.bin_shift) & 0x3FFFFFFFu;
			uint32_t slot = (bin * 2654435761u) >> (32 - log2_slots);
			for (uint32_t probes = 0; probes < n_slots; ++probes) {
				const uint32_t prev = atomicCAS(&t_keys[slot], 0xFFFFFFFFu, bin);
				if (prev == bin) { t_cand[slot] = 1u; break; }
				if (prev == 0xFFFFFFFFu) { const uint32_t at = atomicAdd(&s_keys, 1u); if (GLOBAL) t_list[at] = slot; t_cand[slot] = 1u; break; }
				slot = (slot + 1) & (n_slots - 1);
			}
		}
	}
	__syncthreads();
	auto bin_of = [&](uint32_t pos, int li) -> uint32_t {
		const int p = l_vpos ? (int) l_vpos[li >> 1] : li >> 1;
		const uint32_t correction = (li & 1) ? (uint32_t) (L - (p + k)) : (uint32_t) p;  // CS.cpp:140-142
		return ((pos - correction) >> A.bin_shift) & 0x3FFFFFFFu;
	};
	// sweep A (the only pass over the position lists): every hit is written to the time line (bin | strand << 31);
	// bins hit at least twice (or colliding on a plane bit) become tracked keys
	// (round 5: the eight hits of a work item together -- their plane updates, then the first probes of those that need the table, are in
	// flight at the same time; one hit after the other, every probe of a table in global memory was a round trip to L2 on its own)
	auto vote8 = [&](const uint32_t (&pos8)[8], uint32_t cnt, int li, uint32_t t0) {
		uint32_t bin[kCsSeg], slot[kCsSeg], prev[kCsSeg];
		bool need[kCsSeg];
		const uint32_t sbit = (li & 1) ? 0x80000000u : 0u;
#pragma unroll
		for (int j = 0; j < kCsSeg; ++j) {
			need[j] = false;
			if ((uint32_t) j < cnt) {
				bin[j] = bin_of(pos8[j], li);
				ev_at[t0 + (uint32_t) j] = bin[j] | sbit;
				const uint32_t b = (bin[j] * 0x9E3779B1u) >> 16;
				const uint32_t msk = 1u << (b & 31);
				need[j] = (atomicOr(&plane[b >> 5], msk) & msk) != 0u;
			}
		}
		// (a read whose repeated bins outgrow the LDS table is given up below: stop inserting as soon as that is certain -- with the table
		// nearly full every further hit would walk hundreds of slots)
		if (!GLOBAL && *(volatile uint32_t *) &s_keys > (n_slots * 3u) / 4u) return;
#pragma unroll
		for (int j = 0; j < kCsSeg; ++j) if (need[j]) { slot[j] = (bin[j] * 2654435761u) >> (32 - log2_slots); prev[j] = atomicCAS(&t_keys[slot[j]], 0xFFFFFFFFu, bin[j]); }
#pragma unroll
		for (int j = 0; j < kCsSeg; ++j) if (need[j]) {
			uint32_t sl = slot[j], pv = prev[j];
			for (uint32_t probes = 1; pv != bin[j] && pv != 0xFFFFFFFFu && probes < n_slots; ++probes) {
				sl = (sl + 1) & (n_slots - 1);
				pv = atomicCAS(&t_keys[sl], 0xFFFFFFFFu, bin[j]);
			}
			if (pv == 0xFFFFFFFFu) { const uint32_t at = atomicAdd(&s_keys, 1u); if (GLOBAL) t_list[at] = sl; }
		}
	};
	{
		// work item = 8 consecutive hits of one list (two 16-byte loads), the next item's loads in flight; item -> (list,
		// segment) by bisection of seg_pref, so that no item list has to fit anywhere
		const int n_lists = R.n_lists;
		CsU4 cur[2], nxt[2];
		auto fetch = [&](uint32_t idx, CsU4 (&d)[2]) -> uint32_t {
			if (idx >= R.n_items) return 0xFFFFFFFFu;
			int lo = 0, hi = n_lists;  // the list with seg_pref[li] <= idx < seg_pref[li + 1] (never an empty one)
			while (hi - lo > 1) {
				const int mid = (lo + hi) >> 1;
				if (seg_pref[mid] <= idx) lo = mid; else hi = mid;
			}
			const uint32_t li = (uint32_t) lo, sg = idx - seg_pref[lo];
			const CsU4 *src = reinterpret_cast<const CsU4 *>(A.positions + l_start[li] + sg * kCsSeg);
			d[0] = src[0]; d[1] = src[1];  // the table is padded by 16 entries
			return (li << 16) | sg;
		};
		uint32_t item = fetch((uint32_t) tid, cur);
		for (uint32_t idx = (uint32_t) tid; idx < R.n_items; idx += NT) {
			const uint32_t item_n = fetch(idx + NT, nxt);
			const uint32_t li = item >> 16, sg = item & 0xFFFFu;
			const uint32_t meta = l_pref[li];
			const uint32_t cnt = min((uint32_t) kCsSeg, (meta & 0xFFFFu) - sg * kCsSeg), t0 = (GLOBAL ? l_time[li] : (meta >> 16)) + sg * kCsSeg;
			const uint32_t pos8[8] = {cur[0].x, cur[0].y, cur[0].z, cur[0].w, cur[1].x, cur[1].y, cur[1].z, cur[1].w};
			vote8(pos8, cnt, (int) li, t0);
			item = item_n; cur[0] = nxt[0]; cur[1] = nxt[1];
		}
	}
	__syncthreads();
	if (diag) ck[2] = wall_clock64();
	// (GLOBAL: every access to the read's slice comes from this workgroup -- one CU, one L1 -- so workgroup scope is enough; an
	// agent-scope fence writes this XCD's L2 back, and one per replay trip made the kernel 50 x slower)
	if (GLOBAL) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
	if (!GLOBAL && s_keys > (n_slots * 3u) / 4u) { give_up(3u, H, s_keys); if (diag && tid == 0) atomicAdd(&A.phase_cycles[13], 1ull); return; }
	// sweep B (LDS only): exact votes of the tracked bins; their time line entries become slot | strand << 31, all others
	// empty.  The plane is rebuilt as a bit set of the tracked keys first, so that the ~90 % untracked hits cost one read.
	for (uint32_t s2 = tid; s2 < plane_words; s2 += NT) plane[s2] = 0;
	__syncthreads();
	// the slots in use: entry i of the list (GLOBAL) or slot i of the table
	const uint32_t n_ent = GLOBAL ? s_keys : n_slots;
	auto slot_of = [&](uint32_t i) -> uint32_t { return GLOBAL ? t_list[i] : i; };
	for (uint32_t i = tid; i < n_ent; i += NT) {
		const uint32_t key = og(&t_keys[slot_of(i)]);
		if (key != 0xFFFFFFFFu) { const uint32_t b = (key * 0x9E3779B1u) >> 16; atomicOr(&plane[b >> 5], 1u << (b & 31)); }
	}
	__syncthreads();
	constexpr int KB = 8;   // time line entries per thread and trip: their loads and first probes are in flight together
	for (uint32_t t0 = (uint32_t) tid; t0 < H; t0 += (uint32_t) NT * KB) {
		uint32_t e[KB], slot[KB], key[KB];
		bool in[KB];
#pragma unroll
		for (int j = 0; j < KB; ++j) { const uint32_t t = t0 + (uint32_t) j * NT; e[j] = t < H ? ev_at[t] : 0u; }
#pragma unroll
		for (int j = 0; j < KB; ++j) {
			const uint32_t bin = e[j] & 0x3FFFFFFFu;
			const uint32_t b = (bin * 0x9E3779B1u) >> 16;
			in[j] = t0 + (uint32_t) j * NT < H && ((plane[b >> 5] >> (b & 31)) & 1u) != 0u;
			slot[j] = (bin * 2654435761u) >> (32 - log2_slots);
			key[j] = in[j] ? og(&t_keys[slot[j]]) : 0xFFFFFFFFu;
		}
#pragma unroll
		for (int j = 0; j < KB; ++j) {
			const uint32_t t = t0 + (uint32_t) j * NT;
			if (t >= H) continue;
			const uint32_t bin = e[j] & 0x3FFFFFFFu;
			uint32_t out = 0xFFFFFFFFu, sl = slot[j], ky = key[j];
			while (ky != bin && ky != 0xFFFFFFFFu) { sl = (sl + 1) & (n_slots - 1); ky = og(&t_keys[sl]); }
			if (in[j] && ky == bin) { atomicAdd(&t_votes[sl], (e[j] & 0x80000000u) ? 0x10000u : 1u); out = sl | (e[j] & 0x80000000u); }
			ev_at[t] = out;
		}
	}
	__syncthreads();
	if (diag) ck[3] = wall_clock64();
	// The replay, without its loop (round 5).  What the sequential loop of CS::AddLocationStd carries from hit to hit is (a) the votes of
	// the hit's bin and strand so far and (b) the running maximum M (CS.cpp:197-202); a bin enters rList at its first hit with
	// score >= M * sensitivity (CS.cpp:205-208).  Both follow from the TIMES of the hits of every tracked (bin, strand), sorted:
	//   tau[v]  = the earliest time at which any (bin, strand) has v votes = min over them of the time of their v-th hit
	//             (tau is increasing in v; tau[1] = 0: the first hit of the read);
	//   M(t)    = the number of v with tau[v] <= t;
	//   a candidate's bin enters at the earliest time t_j of a j-th hit of one of its strands with (float) j >= (float) M(t_j) * sensitivity.
	// A bin with one vote never moves the maximum beyond 1 and is never a candidate unless the final threshold is <= 1 (those are
	// tracked as t_cand), so only the hits of bins with >= 2 votes and of the candidates take part.  Every step is parallel over the
	// slots or the hits: counting sort of the hit times by (slot, strand) -- offsets from a block scan over the votes, one atomic per
	// hit for its place, an insertion sort per (short, nearly sorted) segment -- then atomicMin into tau (LDS) and one pass over the
	// candidates' segments.  The sequential replay took 64 hits per trip with the table's latency on every trip: 20-30 ms for a read
	// with 300 000 tracked hits in global memory (profiles/r05_heavy_tail_3100mbp_first.txt: 1.1 s of replay per 1 M reads).
	// cand_rank[c] = 2 * (the time its bin entered rList) + strand: the same order as the rList positions, which is all its users compare.
	{
		uint32_t *t_off = t_run;    // offset of the slot's segment (forward hits, then reverse hits) in tm
		uint32_t *t_fill = t_rank;  // hits placed so far (forward | reverse << 16); afterwards: the time the bin entered rList
		__shared__ uint32_t s_scan[NT / 64], s_bad;
		if (tid == 0) s_bad = 0;
		for (uint32_t v = tid; v < n_tau; v += NT) tau[v] = v == 1u && H > 0u ? 0u : 0xFFFFFFFFu;
		auto kept = [&](uint32_t s2, uint32_t &nf, uint32_t &nr) -> bool {   // does slot s2 take part, and with how many hits
			nf = nr = 0;
			if (!GLOBAL && t_keys[s2] == 0xFFFFFFFFu) return false;
			const uint32_t v = og(&t_votes[s2]);
			nf = v & 0xFFFFu; nr = v >> 16;
			return nf + nr >= 2u || t_cand[s2] != 0u;
		};
		// (a) offsets: every thread a contiguous run of entries; a slot that does not take part is marked
		const uint32_t per = (n_ent + NT - 1) / NT, s_lo = min(n_ent, (uint32_t) tid * per), s_hi = min(n_ent, s_lo + per);
		uint32_t mine = 0;
		for (uint32_t i = s_lo; i < s_hi; ++i) { uint32_t nf, nr; if (kept(slot_of(i), nf, nr)) mine += nf + nr; }
		const uint32_t incl = wave_inclusive_scan(mine, lane);
		if (lane == 63) s_scan[wv] = incl;
		__syncthreads();
		uint32_t run = incl - mine;
		for (int w2 = 0; w2 < wv; ++w2) run += s_scan[w2];
		for (uint32_t i = s_lo; i < s_hi; ++i) {
			const uint32_t s2 = slot_of(i);
			uint32_t nf, nr;
			const bool k2 = kept(s2, nf, nr);
			t_off[s2] = k2 ? run : 0xFFFFFFFFu; t_fill[s2] = 0;
			if (k2) run += nf + nr;
		}
		if (GLOBAL) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
		__syncthreads();
		// (b) every tracked hit to its segment (KB entries of the time line per thread and trip)
		for (uint32_t t0 = (uint32_t) tid; t0 < H; t0 += (uint32_t) NT * KB) {
			uint32_t e[KB], off[KB], nfw[KB], kk[KB];
#pragma unroll
			for (int j = 0; j < KB; ++j) { const uint32_t t = t0 + (uint32_t) j * NT; e[j] = t < H ? ev_at[t] : 0xFFFFFFFFu; }
#pragma unroll
			for (int j = 0; j < KB; ++j) off[j] = e[j] != 0xFFFFFFFFu ? t_off[e[j] & 0x7FFFFFFFu] : 0xFFFFFFFFu;
#pragma unroll
			for (int j = 0; j < KB; ++j) {
				const bool rev = (e[j] >> 31) != 0u;
				nfw[j] = (off[j] != 0xFFFFFFFFu && rev) ? (og(&t_votes[e[j] & 0x7FFFFFFFu]) & 0xFFFFu) : 0u;
				kk[j] = off[j] != 0xFFFFFFFFu ? atomicAdd(&t_fill[e[j] & 0x7FFFFFFFu], rev ? 0x10000u : 1u) : 0u;
			}
#pragma unroll
			for (int j = 0; j < KB; ++j) if (off[j] != 0xFFFFFFFFu) {
				const bool rev = (e[j] >> 31) != 0u;
				tm[off[j] + (rev ? nfw[j] + (kk[j] >> 16) : (kk[j] & 0xFFFFu))] = t0 + (uint32_t) j * NT;
			}
		}
		if (GLOBAL) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
		__syncthreads();
		// (c) segments sorted by time; tau.  A WAVE per segment (round 5, second version: a thread per slot walked its segments with one
		// dependent load per element -- 1.5 of the 2.3 ms of an average read replayed in global memory): the lanes of a wave take the
		// meta data of 64 slots, then the wave goes through their segments one by one -- one coalesced load (the next segment's already
		// in flight), every element's rank by counting (times are distinct), the element stored at its rank, atomicMin into tau.
		// Segments of more than 64 hits (a bin with more votes than that on one strand) stay with the lane that owns the slot.
		auto serial_segment = [&](uint32_t *g, uint32_t n2) {
			for (uint32_t x = 1; x < n2; ++x) {
				const uint32_t key = g[x];
				uint32_t y = x;
				while (y > 0 && g[y - 1] > key) { g[y] = g[y - 1]; --y; }
				g[y] = key;
			}
			if (n2 >= n_tau) { atomicExch(&s_bad, 1u); return; }
			for (uint32_t j = 1; j <= n2; ++j) atomicMin(&tau[j], g[j - 1]);
		};
		for (uint32_t i0 = (uint32_t) wv * 64u; i0 < n_ent; i0 += (uint32_t) (NT / 64) * 64u) {
			const uint32_t i = i0 + (uint32_t) lane;
			const uint32_t s2 = i < n_ent ? slot_of(i) : 0u;
			uint32_t off = i < n_ent ? t_off[s2] : 0xFFFFFFFFu, nf = 0, nr = 0;
			if (i < n_ent) t_fill[s2] = kCsOrderUnknown;
			if (off != 0xFFFFFFFFu) (void) kept(s2, nf, nr);
			// segments of more than 64 hits (a bin that nearly every k-mer of the read votes for: satellite arrays) -- up to 320 by the
			// whole wave, five elements per lane, ranks by counting over all chunks; beyond that the owning lane, serially.  (Left to the
			// owning lanes they were most of this phase: one dependent load per element and an insertion sort in global memory.)
			{
				const bool bigseg = off != 0xFFFFFFFFu && (nf > 64u || nr > 64u);
				if (bigseg && (nf > 320u || nr > 320u)) { serial_segment(tm + off, nf); serial_segment(tm + off + nf, nr); }
				unsigned long long bm = __ballot(bigseg && nf <= 320u && nr <= 320u);
				while (bm) {
					const int kb = (int) __builtin_ctzll(bm);
					bm &= bm - 1ull;
					const uint32_t o = (uint32_t) __builtin_amdgcn_readlane((int) off, kb), f = (uint32_t) __builtin_amdgcn_readlane((int) nf, kb), r = (uint32_t) __builtin_amdgcn_readlane((int) nr, kb);
					for (int st2 = 0; st2 < 2; ++st2) {
						const uint32_t n2 = st2 ? r : f;
						uint32_t *g = tm + o + (st2 ? f : 0u);
						if (n2 == 0u) continue;
						uint32_t ev[5], rk[5];
#pragma unroll
						for (int c2 = 0; c2 < 5; ++c2) { const uint32_t x = (uint32_t) c2 * 64u + (uint32_t) lane; ev[c2] = x < n2 ? g[x] : 0xFFFFFFFFu; rk[c2] = 0; }
#pragma unroll
						for (int cm = 0; cm < 5; ++cm) {
							if ((uint32_t) cm * 64u >= n2) break;
							const uint32_t lim = min(64u, n2 - (uint32_t) cm * 64u);
							for (uint32_t mm = 0; mm < lim; ++mm) {
								const uint32_t vk = (uint32_t) __builtin_amdgcn_readlane((int) ev[cm], (int) mm);
#pragma unroll
								for (int c2 = 0; c2 < 5; ++c2) rk[c2] += vk < ev[c2] ? 1u : 0u;
							}
						}
#pragma unroll
						for (int c2 = 0; c2 < 5; ++c2) if ((uint32_t) c2 * 64u + (uint32_t) lane < n2) {
							g[rk[c2]] = ev[c2];
							if (rk[c2] + 1u < n_tau) atomicMin(&tau[rk[c2] + 1u], ev[c2]); else atomicExch(&s_bad, 1u);
						}
					}
				}
				if (bigseg) off = 0xFFFFFFFFu;
			}
			const unsigned long long todo = __ballot(off != 0xFFFFFFFFu);
			// GLOBAL: the 64 slots' segments are one contiguous range of tm (offsets were handed out in list order): through the staging buffer
			uint32_t *src = tm;
			uint32_t r_lo = 0, r_hi = 0, sub = 0;
			bool staged = false;
			if (GLOBAL && todo) {
				r_lo = (uint32_t) wave_reduce_min(off != 0xFFFFFFFFu ? (int) off : 0x7FFFFFFF);
				r_hi = (uint32_t) wave_reduce_max(off != 0xFFFFFFFFu ? (int) (off + nf + nr) : 0);
				if (r_hi - r_lo <= stage) {
					for (uint32_t x = (uint32_t) lane; x < r_hi - r_lo; x += 64u) wbuf[x] = tm[r_lo + x];
					__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
					src = wbuf; sub = r_lo;   // (indexed with the same offsets, less the range's start)
					staged = true;
				}
			}
			// the segments of these slots, forward then reverse, one after the other; the next one's elements are loaded before the current one is ranked
			uint32_t e_cur = 0xFFFFFFFFu, n_cur = 0, base_cur = 0;
			unsigned long long left = todo;
			int kk = left ? (int) __builtin_ctzll(left) : 64, st = 0;
			auto next_segment = [&](uint32_t &e, uint32_t &n2, uint32_t &base) {   // advances (kk, st) to the next non-empty segment and issues its load
				n2 = 0; e = 0xFFFFFFFFu; base = 0;
				while (kk < 64) {
					const uint32_t o = (uint32_t) __builtin_amdgcn_readlane((int) off, kk), f = (uint32_t) __builtin_amdgcn_readlane((int) nf, kk), r = (uint32_t) __builtin_amdgcn_readlane((int) nr, kk);
					const uint32_t n = st ? r : f, b2 = o + (st ? f : 0u);
					if (st == 0) st = 1; else { st = 0; left &= left - 1ull; kk = left ? (int) __builtin_ctzll(left) : 64; }
					if (n) { n2 = n; base = b2 - sub; e = (uint32_t) lane < n ? src[base + (uint32_t) lane] : 0xFFFFFFFFu; return; }
				}
			};
			next_segment(e_cur, n_cur, base_cur);
			while (n_cur) {
				uint32_t e_nx, n_nx, base_nx;
				next_segment(e_nx, n_nx, base_nx);
				uint32_t rank = 0;
				for (uint32_t mm = 0; mm < n_cur; ++mm) rank += (uint32_t) __builtin_amdgcn_readlane((int) e_cur, (int) mm) < e_cur ? 1u : 0u;
				if ((uint32_t) lane < n_cur) {
					src[base_cur + rank] = e_cur;
					if (rank + 1u < n_tau) atomicMin(&tau[rank + 1u], e_cur); else atomicExch(&s_bad, 1u);
				}
				e_cur = e_nx; n_cur = n_nx; base_cur = base_nx;
			}
			if (staged) {
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
				for (uint32_t x = (uint32_t) lane; x < r_hi - r_lo; x += 64u) tm[r_lo + x] = wbuf[x];
			}
		}
		if (GLOBAL) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
		__syncthreads();
		if (s_bad) { give_up(4u, H, s_keys); return; }   // (more votes for one bin and strand than the read has k-mers + 64: not reached)
		// (d) when does every candidate's bin enter rList: a wave per candidate slot, lane j - 1 looks at the j-th hit of a strand
		for (uint32_t i0 = (uint32_t) wv * 64u; i0 < n_ent; i0 += (uint32_t) (NT / 64) * 64u) {
			const uint32_t i = i0 + (uint32_t) lane;
			const uint32_t s2 = i < n_ent ? slot_of(i) : 0u;
			uint32_t off = (i < n_ent && t_cand[s2]) ? t_off[s2] : 0xFFFFFFFFu, nf = 0, nr = 0;
			if (off != 0xFFFFFFFFu) (void) kept(s2, nf, nr);
			uint32_t enter_mine = kCsOrderUnknown;
			{
				// a candidate's segments of more than 64 hits: the wave looks at them 64 hits at a time (times increase with j: the first chunk
				// with a qualifying hit ends the strand)
				const bool bigseg = off != 0xFFFFFFFFu && (nf > 64u || nr > 64u);
				unsigned long long bm = __ballot(bigseg);
				while (bm) {
					const int kb = (int) __builtin_ctzll(bm);
					bm &= bm - 1ull;
					const uint32_t o = (uint32_t) __builtin_amdgcn_readlane((int) off, kb), f = (uint32_t) __builtin_amdgcn_readlane((int) nf, kb), r = (uint32_t) __builtin_amdgcn_readlane((int) nr, kb);
					uint32_t enter = kCsOrderUnknown;
					for (int st2 = 0; st2 < 2; ++st2) {
						const uint32_t n2 = st2 ? r : f;
						const uint32_t *g = tm + o + (st2 ? f : 0u);
						for (uint32_t j0 = 0; j0 < n2; j0 += 64u) {
							const uint32_t j = j0 + (uint32_t) lane + 1u;
							const uint32_t t = j <= n2 ? g[j - 1u] : 0xFFFFFFFFu;
							uint32_t lo = 1, hi = n_tau;
							while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (tau[mid] <= t) lo = mid; else hi = mid; }
							const unsigned long long mk = __ballot(t != 0xFFFFFFFFu && (float) j >= (float) lo * A.sensitivity);
							if (mk) { enter = min(enter, (uint32_t) __builtin_amdgcn_readlane((int) t, (int) __builtin_ctzll(mk))); break; }
						}
					}
					if (lane == kb) enter_mine = enter;
				}
				if (bigseg) { t_fill[s2] = enter_mine; off = 0xFFFFFFFFu; }
			}
			unsigned long long left = __ballot(off != 0xFFFFFFFFu);
			const uint32_t *src = tm;
			uint32_t sub = 0;
			if (GLOBAL && left) {   // the candidates' segments of these 64 slots: staged when they lie close together
				const uint32_t r_lo = (uint32_t) wave_reduce_min(off != 0xFFFFFFFFu ? (int) off : 0x7FFFFFFF), r_hi = (uint32_t) wave_reduce_max(off != 0xFFFFFFFFu ? (int) (off + nf + nr) : 0);
				if (r_hi - r_lo <= stage) {
					for (uint32_t x = (uint32_t) lane; x < r_hi - r_lo; x += 64u) wbuf[x] = tm[r_lo + x];
					__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
					src = wbuf; sub = r_lo;
				}
			}
			while (left) {
				const int kk = (int) __builtin_ctzll(left);
				left &= left - 1ull;
				const uint32_t o = (uint32_t) __builtin_amdgcn_readlane((int) off, kk), f = (uint32_t) __builtin_amdgcn_readlane((int) nf, kk), r = (uint32_t) __builtin_amdgcn_readlane((int) nr, kk);
				const uint32_t tf = (uint32_t) lane < f ? src[o - sub + (uint32_t) lane] : 0xFFFFFFFFu, tr = (uint32_t) lane < r ? src[o - sub + f + (uint32_t) lane] : 0xFFFFFFFFu;
				uint32_t enter = kCsOrderUnknown;
#pragma unroll
				for (int st2 = 0; st2 < 2; ++st2) {
					const uint32_t t = st2 ? tr : tf;
					uint32_t lo = 1, hi = n_tau;   // M(t): the largest v with tau[v] <= t (tau[1] = 0 <= t)
					while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (tau[mid] <= t) lo = mid; else hi = mid; }
					const bool in = t != 0xFFFFFFFFu && (float) (lane + 1) >= (float) lo * A.sensitivity;
					const unsigned long long mk = __ballot(in);
					if (mk) enter = min(enter, (uint32_t) __builtin_amdgcn_readlane((int) t, (int) __builtin_ctzll(mk)));   // the first j that qualifies (times increase with j)
				}
				if (lane == kk) enter_mine = enter;
			}
			if (off != 0xFFFFFFFFu) t_fill[s2] = enter_mine;
		}
		if (GLOBAL) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
	}
	__syncthreads();
	if (diag) ck[4] = wall_clock64();
	if (diag && tid == 0) {  // 100 MHz ticks: lists, sweep A, sweep B, compaction + replay; sampled reads, big ones, hits, replayed hits
		atomicAdd(&A.phase_cycles[8], ck[1] - ck[0]); atomicAdd(&A.phase_cycles[9], ck[2] - ck[1]); atomicAdd(&A.phase_cycles[10], ck[3] - ck[2]);
		atomicAdd(&A.phase_cycles[11], ck[4] - ck[3]); atomicAdd(&A.phase_cycles[12], 1ull); 
		atomicAdd(&A.phase_cycles[14], (unsigned long long) H); atomicAdd(&A.phase_cycles[15], (unsigned long long) s_keys);
	}
	const uint32_t centre = A.bin_shift > 0 ? (1u << (A.bin_shift - 1)) : 0u;
	for (uint32_t c = tid; c < cn; c += NT) {
		const uint32_t bin = ((cand_loc[cb + c] - centre) >> A.bin_shift) & 0x3FFFFFFFu;
		uint32_t slot = (bin * 2654435761u) >> (32 - log2_slots);
		uint32_t rank = kCsOrderUnknown;
		for (uint32_t probes = 0; probes < n_slots; ++probes) {
			const uint32_t key = og(&t_keys[slot]);
			if (key == bin) { if (t_rank[slot] != kCsOrderUnknown) rank = 2u * t_rank[slot] + (cand_sv[cb + c] & 1u); break; }   // (t_rank: the time the bin entered rList)
			if (key == 0xFFFFFFFFu) break;
			slot = (slot + 1) & (n_slots - 1);
		}
		cand_rank[cb + c] = rank;
	}
	if (tid == 0 && A.order_info && !GLOBAL) { A.order_info[2 * blockIdx.x] = H; A.order_info[2 * blockIdx.x + 1] = s_keys << 8; }
	if (A.phase_cycles && threadIdx.x == 0) {
		const unsigned long long dt = wall_clock64() - t_block;
		atomicAdd(&A.phase_cycles[13], dt << 8);  // whole-block time of every workgroup (ticks << 8 above the give-up count)
		atomicMax(&A.phase_cycles[16], dt);
		if (dt > 100000ull) { atomicAdd(&A.phase_cycles[17], 1ull); atomicMax(&A.phase_cycles[18], (unsigned long long) H); atomicMax(&A.phase_cycles[19], (unsigned long long) s_keys); }  // > 1 ms
	}
}

}  // namespace ngm
