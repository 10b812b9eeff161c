// cs_order_bucket_device.h -- the candidate ORDER of reads with more hits than the LDS replay's time line holds (cs_order_kernel<false>,
// cs_device.h), without a table in global memory.
//
// What the replay needs (cs_device.h, "The replay, without its loop"): for every (bin, strand) the TIMES of its hits in order -- the
// time of a hit is its index in the reference's visiting order: k-mers left to right, forward list then reverse-complement list, list
// entries in index order (CS::AddLocationStd is called in that order, src/CS.cpp:112-160) -- then
//   tau[v] = the earliest time at which any (bin, strand) has v votes,   M(t) = the largest v with tau[v] <= t   (CS.cpp:197-202),
//   a candidate's bin enters rList at the earliest j-th hit of one of its strands with (float) j >= (float) M(t_j) * sensitivity (CS.cpp:205-208).
// cs_order_kernel<true> gets the hits of a bin together through a hash table in a per-read slice of global memory: a handful of
// dependent L2 round trips per hit (2.3 ms for a read of 70 000 hits, profiles/r05_heavy_tail_probe_*).  Here the hits are dealt
// into 2^b BUCKETS by a hash of their bin -- count (LDS), scan, scatter of (bin | strand << 31, time) pairs into the workgroup's slice
// of a scratch: one 8-byte store per hit, nothing else leaves the CU -- and a bucket (8 hits on average, every hit of a bin in the
// same one) is a few lanes of a wave: a hit's v is 1 + the number of hits of its (bin, strand) in the bucket with a smaller time.
// tau[v] = the minimum over the hits (LDS); then, tau complete, a lane per candidate looks through the candidate's bucket for the
// earliest hit of its bin that qualifies.
// Workgroups are persistent (one slice each) and draw reads from a counter; the host lists the reads by decreasing hits.
// Left to cs_order_kernel<true>: bisulfite runs (lists per k-mer variant), reads of more than 2^20 hits.
#pragma once
#include "cs_device.h"
#include "cs_heavy_device.h"

namespace ngm {

constexpr int kCsOrderBucketThreads = 512;     // (two or three workgroups per CU: while one waits at a barrier the others sweep)
constexpr int kCsOrderBucketLog2Max = 14;      // most buckets of a read (one LDS word each; the host lowers it when the LDS does not hold them beside the lists: CsArgs::log2_bits).
// Round 5 had 2^13 buckets of ~8 hits; 2^14 of ~4 (stress sub-leg of the bench, 66 000 hits per replayed read: `v + tau` 416 -> 365 us per read, scatter 75 -> 80; with 2^15
// 350 and 85-100, and 128 KB of LDS that the search kernels of the other instances then lack: profiles/r06_bucket_fill_ab.txt) -- the walk's trip count is the
// longest bucket of a window, and in reads from repeats that is a bin with many hits of its own, not a crowded bucket
constexpr uint32_t kCsOrderBucketMaxHits = 1u << 20;   // a hit's time and its v share a word (20 + 12 bits)
constexpr uint32_t kCsOrderBucketMaxTau = 1u << 12;

inline size_t cs_order_bucket_lds_bytes(int lists_cap, int q, size_t coarse_cap, int log2_buckets) {
	return ((size_t) lists_cap * 3 + 2 + (size_t) (q + 3) / 4 + (coarse_cap + 1) / 2 + 3 + cs_order_tau(lists_cap) + ((size_t) 1 << log2_buckets) + 1 + 4) * 4;
}

// info[2 * i + 1] of listed read i: 0 = order determined | 2 more hits than the slice (or than 2^20) | 4 more votes than tau holds | 6 item table
template <int NT>
__device__ __forceinline__ void cs_order_bucket_body(const CsArgs &A, uint32_t n_list, uint32_t *__restrict__ work_counter, uint2 *__restrict__ scratch, uint32_t scratch_cap,
		uint32_t coarse_cap, const uint32_t *__restrict__ cand_loc, const uint32_t *__restrict__ cand_sv, uint32_t *__restrict__ cand_rank, uint32_t *__restrict__ info,
		unsigned long long *__restrict__ diag) {   // diag (NGM_HIP_CS_PHASES): [0..6] 100 MHz ticks per phase of every 8th read, [8] reads sampled, [9] their hits, [10] their candidates, [11] reads left to the table kernel
	extern __shared__ __attribute__((aligned(16))) uint32_t cs_lds[];
	constexpr int NW = NT / 64;
	__shared__ uint32_t s_next, s_bad, s_scan[NW], s_mlo[257];
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const int k = A.k;
	uint32_t *l_start = cs_lds;                                   // [lists_cap]
	uint32_t *l_pref = cs_lds + A.lists_cap;                      // [lists_cap + 1]: hits in front of each list = the time of its first hit
	uint8_t *l_code = (uint8_t *) (l_pref + A.lists_cap + 1);     // [q rounded up to 4]
	uint32_t *seg_pref = (uint32_t *) l_code + (A.q + 3) / 4;     // [lists_cap + 1]: 8-hit segments in front of each list
	uint16_t *coarse = (uint16_t *) (seg_pref + A.lists_cap + 1); // [coarse_cap]: the list that holds item 32 c
	uint32_t *tau = cs_lds + (((size_t) ((uint32_t *) coarse - cs_lds) + (coarse_cap + 1) / 2 + 3) & ~(size_t) 3);
	const uint32_t n_tau = cs_order_tau(A.lists_cap);
	uint32_t *bk = tau + n_tau;                                   // [buckets + 1]: bucket b is my[bk[b] .. bk[b + 1]) (bk[0] = 0; bk[b + 1]: its count, then its cursor)
	uint2 *my = scratch + (size_t) blockIdx.x * scratch_cap;
	const uint32_t centre = A.bin_shift > 0 ? (1u << (A.bin_shift - 1)) : 0u;
	for (;;) {
		__syncthreads();   // (the previous read's shared state is no longer read)
		if (tid == 0) { s_next = atomicAdd(work_counter, 1u); s_bad = 0; }
		__syncthreads();
		const uint32_t item_ix = s_next;
		if (item_ix >= n_list) return;
		const int read = (int) A.read_list[item_ix];
		const bool dg = diag != nullptr && (item_ix & 7u) == 0u && tid == 0;
		unsigned long long tk = dg ? wall_clock64() : 0ull;
		auto mark = [&](int ph) { if (dg) { const unsigned long long t2 = wall_clock64(); atomicAdd(&diag[ph], t2 - tk); tk = t2; } };
		const uint32_t cb = A.cand_base[read], cn = A.cand_count[read];
		auto give_up = [&](uint32_t why, uint32_t hits) {   // (block-uniform)
			for (uint32_t c = tid; c < cn; c += NT) cand_rank[cb + c] = kCsOrderUnknown;
			if (tid == 0) { info[2 * item_ix] = hits; info[2 * item_ix + 1] = why; if (diag) atomicAdd(&diag[11], 1ull); }
		};
		// every wave computes the same lists (the barrier inside is the block's)
		const CsRead R = cs_prepare<false>(A, read, lane, l_start, l_pref, l_code);
		const uint32_t H = R.H;
		const int L = R.L;
		const int n_lists = R.n_lists;
		__syncthreads();
		if (H > scratch_cap || H > kCsOrderBucketMaxHits) { give_up(2u, H); continue; }
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
		int log2_nb = 6;
		const int log2_nb_max = min(kCsOrderBucketLog2Max, max(6, A.log2_bits));   // (A.log2_bits: the host's limit -- kCsOrderBucketLog2Max unless a test asks for crowded buckets)
		const uint32_t fill = A.order_gcap ? A.order_gcap : 4u;   // hits per bucket aimed at
		while (log2_nb < log2_nb_max && (fill << log2_nb) < H) ++log2_nb;
		const uint32_t nb = 1u << log2_nb;
		for (uint32_t b = tid; b <= nb; b += NT) bk[b] = 0;
		for (uint32_t v = tid; v < n_tau; v += NT) tau[v] = 0xFFFFFFFFu;
		__syncthreads();
		const uint32_t n_items = seg_pref[n_lists];
		if ((n_items >> kCsHeavyCoarseShift) + 3u > coarse_cap) { give_up(6u, H); continue; }   // (sized from max_kfreq: not reached)
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
		mark(0);
		// f(positions, count, key bits of the strand, diagonal correction, time of the first) for every 8-hit segment of a list: per thread one
		// segment at a time, the next one's loads in flight (cs_heavy2_kernel's sweep)
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
			CsU4 nx2[2];   // (two segments' loads in flight behind the one in hand, as in cs_heavy2_kernel)
			uint32_t item = fetch((uint32_t) tid, cur), item_1 = fetch((uint32_t) tid + NT, nxt);
			for (uint32_t idx = (uint32_t) tid; idx < n_items; idx += NT) {
				const uint32_t item_n = item_1;
				item_1 = fetch(idx + 2u * NT, nx2);
				const int li = (int) (item >> 16);
				const uint32_t sg = item & 0xFFFFu;
				const uint32_t first = l_pref[li], len = l_pref[li + 1] - first;
				const uint32_t cnt = min((uint32_t) kCsSeg, len - sg * kCsSeg);
				const int p = li >> 1;
				const uint32_t correction = (li & 1) ? (uint32_t) (L - (p + k)) : (uint32_t) p;  // CS.cpp:140-142
				const uint32_t pos8[8] = {cur[0].x, cur[0].y, cur[0].z, cur[0].w, cur[1].x, cur[1].y, cur[1].z, cur[1].w};
				f(pos8, cnt, (li & 1) ? 0x80000000u : 0u, correction, first + sg * kCsSeg);
				item = item_n; cur[0] = nxt[0]; cur[1] = nxt[1]; nxt[0] = nx2[0]; nxt[1] = nx2[1];
			}
		};
		auto bucket_of = [&](uint32_t bin) -> uint32_t { return (bin * 2654435761u) >> (32 - log2_nb); };
		// 1. hits per bucket
		sweep([&](const uint32_t (&pos8)[8], uint32_t cnt, uint32_t, uint32_t correction, uint32_t) {
#pragma unroll
			for (int j = 0; j < kCsSeg; ++j) if ((uint32_t) j < cnt) atomicAdd(&bk[bucket_of(((pos8[j] - correction) >> A.bin_shift) & 0x3FFFFFFFu) + 1u], 1u);
		});
		__syncthreads();
		mark(1);
		// 2. where every bucket starts in the slice (a contiguous run of buckets per thread): bk[b + 1] = start of bucket b
		{
			const uint32_t per = (nb + NT - 1) / NT, lo = min(nb, (uint32_t) tid * per), hi = min(nb, lo + per);
			uint32_t mine = 0;
			for (uint32_t b = lo; b < hi; ++b) mine += bk[b + 1];
			const uint32_t incl = wave_inclusive_scan(mine, lane);
			if (lane == 63) s_scan[wv] = incl;
			__syncthreads();
			uint32_t run = incl - mine;
			for (int w2 = 0; w2 < wv; ++w2) run += s_scan[w2];
			for (uint32_t b = lo; b < hi; ++b) { const uint32_t c = bk[b + 1]; bk[b + 1] = run; run += c; }
		}
		__syncthreads();
		// 3. every hit to its bucket; afterwards bk[b + 1] is the END of bucket b = the start of bucket b + 1
		sweep([&](const uint32_t (&pos8)[8], uint32_t cnt, uint32_t sbit, uint32_t correction, uint32_t t0) {
			uint32_t bin[kCsSeg], at[kCsSeg];
#pragma unroll
			for (int j = 0; j < kCsSeg; ++j) if ((uint32_t) j < cnt) { bin[j] = ((pos8[j] - correction) >> A.bin_shift) & 0x3FFFFFFFu; at[j] = atomicAdd(&bk[bucket_of(bin[j]) + 1u], 1u); }
#pragma unroll
			for (int j = 0; j < kCsSeg; ++j) if ((uint32_t) j < cnt) my[at[j]] = make_uint2(bin[j] | sbit, t0 + (uint32_t) j);
		});
		// (every access to the slice comes from this workgroup -- one CU, one L1: workgroup scope)
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
		__syncthreads();
		mark(2);
		// 4. v of every hit = 1 + the hits of its (bin, strand) with a smaller time; tau.  A wave streams through its share of the slice in
		// WINDOWS of whole buckets, at most 64 hits: one coalesced load (the next window's is in flight), every lane then walks the lanes of its
		// own bucket (ds_bpermute: ~8 of them).  A bucket of more than 64 hits is a window of its own, 64 of its hits at a time against all of
		// them.  The hit's word 1 becomes v << 20 | time.
		const uint2 none = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);   // (no key looks like this: bit 30 of a key is 0)
		{
			const uint32_t b_lo = (uint32_t) wv * (nb / NW), b_hi = b_lo + nb / NW;   // (nb >= 64: a multiple of the waves)
			// the window that starts with bucket b (at hit p): buckets b .. b2 - 1, hits p .. p + n - 1; n > 64: the one bucket b
			auto window = [&](uint32_t b, uint32_t p, uint32_t &b2, uint32_t &n) -> uint2 {
				// (everything here is the same in every lane; what comes out of LDS is made uniform by hand, or the loops over it run under exec masks)
				p = (uint32_t) __builtin_amdgcn_readfirstlane((int) p);
				while (b < b_hi && (uint32_t) __builtin_amdgcn_readfirstlane((int) bk[b + 1]) == p) ++b;   // (empty buckets)
				if (b >= b_hi) { b2 = b_hi; n = 0; return none; }
				const uint32_t idx = b + 1u + (uint32_t) lane;
				const uint32_t fits = (uint32_t) __popcll(__ballot(idx <= b_hi && bk[idx] <= p + 64u));   // (bk increases: a run of lanes from 0)
				b2 = b + max(fits, 1u);
				n = (uint32_t) __builtin_amdgcn_readfirstlane((int) (bk[b2] - p));
				return (uint32_t) lane < n ? my[p + (uint32_t) lane] : none;
			};
			// returns v of this lane's hit of a window of at most 64 hits (the caller stores it); a larger bucket is finished here
			auto process = [&](const uint2 &e_c, uint32_t p_c, uint32_t n_c) -> uint32_t {
				if (n_c <= 64u) {
					// A hit's v = 1 + the hits of its (bin, strand) with a smaller time; those can only sit in its own BUCKET, whose bounds are in LDS.
					// Round 5 compared every hit of the window with every other through scalar registers (64 trips of two v_readlane, two compares, an
					// AND and an add: ~5 600 cycles per window, 290 of the 650 us of a 66 000-hit read).  Round 6: every lane walks its own bucket
					// through ds_bpermute, EIGHT trips' permutes in flight (the walk round 5 tried waited for every LDS round trip in turn: "no
					// faster"); the trip count is the window's longest bucket -- ~8 in a window of chance hits, 64 when one repeat-family bin fills it.
					const uint32_t key = e_c.x, t = e_c.y;
					const bool live = key != 0xFFFFFFFFu;
					const uint32_t bkt = live ? bucket_of(key & 0x3FFFFFFFu) : 0u;
					const uint32_t bs = live ? bk[bkt] - p_c : (uint32_t) lane, be = live ? bk[bkt + 1u] - p_c : (uint32_t) lane;   // this lane's bucket inside the window
					const uint32_t longest = (uint32_t) wave_reduce_max((int) (be - bs));
					uint32_t v = 1;
					// (no bound on the walk: a lane beyond the bucket's end -- the permute takes the lane number modulo 64, and fewer than 64 steps
					// never come round to the own bucket again -- holds a hit of ANOTHER bucket, whose key differs, or nothing)
					const uint32_t a0 = bs << 2;
					for (uint32_t j0 = 0; j0 < longest; j0 += 8u) {
						uint32_t k2[8], t2[8];
						const uint32_t a1 = a0 + (j0 << 2);
#pragma unroll
						for (int u = 0; u < 8; ++u) {
							k2[u] = (uint32_t) __builtin_amdgcn_ds_bpermute((int) (a1 + 4u * (uint32_t) u), (int) key);
							t2[u] = (uint32_t) __builtin_amdgcn_ds_bpermute((int) (a1 + 4u * (uint32_t) u), (int) t);
						}
#pragma unroll
						for (int u = 0; u < 8; ++u) v += (k2[u] == key && t2[u] < t) ? 1u : 0u;
					}
					if (live) {
						if (v >= n_tau) atomicExch(&s_bad, 4u); else if (t < tau[v]) atomicMin(&tau[v], t);
					}
					return v;
				} else {
					const uint32_t n_b = (uint32_t) __builtin_amdgcn_readfirstlane((int) n_c);
					for (uint32_t c0 = 0; c0 < n_b; c0 += 64u) {
						const uint2 mine = c0 + (uint32_t) lane < n_b ? my[p_c + c0 + (uint32_t) lane] : none;
						uint32_t v = 1;
						for (uint32_t d0 = 0; d0 < n_b; d0 += 64u) {
							const uint2 other = d0 + (uint32_t) lane < n_b ? my[p_c + d0 + (uint32_t) lane] : none;
							const uint32_t ot = other.y & 0xFFFFFu;   // (the hits in front have their v in the upper bits already)
							// every hit of this chunk against every hit of the other one: the other chunk's lanes by ds_bpermute, rotating, eight trips'
							// permutes in flight (round 5: through scalar registers, two v_readlane per trip)
							for (uint32_t j0 = 0; j0 < 64u; j0 += 8u) {
								uint32_t k2[8], t2[8];
#pragma unroll
								for (int u = 0; u < 8; ++u) {
									const uint32_t src = ((uint32_t) lane + j0 + (uint32_t) u) & 63u;
									k2[u] = (uint32_t) __builtin_amdgcn_ds_bpermute((int) (src << 2), (int) other.x);
									t2[u] = (uint32_t) __builtin_amdgcn_ds_bpermute((int) (src << 2), (int) ot);
								}
#pragma unroll
								for (int u = 0; u < 8; ++u) v += (k2[u] == mine.x && t2[u] < mine.y) ? 1u : 0u;
							}
						}
						if (mine.x != 0xFFFFFFFFu) {
							if (v >= n_tau) atomicExch(&s_bad, 4u); else if (mine.y < tau[v]) atomicMin(&tau[v], mine.y);
						}
						// (the chunks after this one read these words again: masked to the time, the same value before and after this store)
						if (mine.x != 0xFFFFFFFFu) my[p_c + c0 + (uint32_t) lane].y = (min(v, kCsOrderBucketMaxTau - 1u) << 20) | mine.y;
					}
					return 0u;
				}
			};
			// four windows at a time: their bounds come from LDS, so the four loads are in flight together; and the NEXT four are issued before this
			// batch's v are stored -- a wave's loads and stores complete in order, and behind its own stores every batch waited ~10 us
			// (measured: 2.6 us per window, the same with one and with three workgroups per CU)
			constexpr int KW = 4;
			uint32_t b_next = b_lo;
			unsigned long long tw = 0, tp = 0, nwin = 0;   // (diagnostics: wave 0 of the sampled reads)
			uint32_t pW[KW], nW[KW];
			uint2 eW[KW];
			auto issue = [&](uint32_t (&pp)[KW], uint32_t (&nn)[KW], uint2 (&ee)[KW]) {
#pragma unroll
				for (int w = 0; w < KW; ++w) {
					uint32_t b2 = b_hi, n = 0;
					ee[w] = window(b_next, bk[b_next], b2, n);
					nn[w] = n; pp[w] = (uint32_t) __builtin_amdgcn_readfirstlane((int) (bk[b2] - n)); b_next = b2;
				}
			};
			issue(pW, nW, eW);
			while (nW[0]) {
				const unsigned long long c0 = dg ? wall_clock64() : 0ull;
				uint32_t pN[KW], nN[KW], vW[KW];
				uint2 eN[KW];
				issue(pN, nN, eN);
				const unsigned long long c1 = dg ? wall_clock64() : 0ull;
#pragma unroll
				for (int w = 0; w < KW; ++w) { vW[w] = nW[w] ? process(eW[w], pW[w], nW[w]) : 0u; nwin += nW[w] ? 1u : 0u; }
#pragma unroll
				for (int w = 0; w < KW; ++w) if (nW[w] && nW[w] <= 64u && eW[w].x != 0xFFFFFFFFu) my[pW[w] + (uint32_t) lane].y = (vW[w] << 20) | eW[w].y;
				if (dg) { const unsigned long long c2 = wall_clock64(); tw += c1 - c0; tp += c2 - c1; }
#pragma unroll
				for (int w = 0; w < KW; ++w) { pW[w] = pN[w]; nW[w] = nN[w]; eW[w] = eN[w]; }
			}
			if (dg) { atomicAdd(&diag[12], tw); atomicAdd(&diag[13], tp); atomicAdd(&diag[14], nwin); }
		}
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
		__syncthreads();
		mark(3);
		if (s_bad) { give_up(s_bad, H); continue; }
		// 5. tau is complete.  M(t) -- the largest v with tau[v] <= t (tau[1] = 0 <= t) -- from a table of M at 256 evenly spaced times and a step
		// or two along tau, not a bisection per hit
		int m_shift = 0;
		while ((H >> m_shift) > 256u) ++m_shift;
		if (tid < 257) {
			const uint32_t t = (uint32_t) tid << m_shift;
			uint32_t lo = 1, hi = n_tau;
			while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (tau[mid] <= t) lo = mid; else hi = mid; }
			s_mlo[tid] = lo;
		}
		__syncthreads();
		mark(4);
		// the time of the hit if it is one of `bin` (either strand) and qualifies -- (float) v >= (float) M(t) * sensitivity (CS.cpp:205) -- else "never"
		auto qualifies = [&](const uint2 &e, uint32_t bin) -> uint32_t {
			if ((e.x & 0x7FFFFFFFu) != bin) return kCsOrderUnknown;   // (an empty lane's key has bit 30 set)
			const uint32_t v = e.y >> 20, t = e.y & 0xFFFFFu;
			uint32_t mt = s_mlo[t >> m_shift];
			while (mt + 1u < n_tau && tau[mt + 1u] <= t) ++mt;
			return (float) v >= (float) mt * A.sensitivity ? t : kCsOrderUnknown;
		};
		// 6. the earliest qualifying hit of every candidate's bin, in the bin's bucket: a lane per candidate, 64 candidates of a wave at a time;
		// the buckets of more than 64 hits among them are then scanned by the whole wave, one after the other
		for (uint32_t c0 = (uint32_t) wv * 64u; c0 < cn; c0 += (uint32_t) NT) {
			const uint32_t c = c0 + (uint32_t) lane;
			uint32_t bin = 0, s0 = 0, s1 = 0;
			if (c < cn) { bin = ((cand_loc[cb + c] - centre) >> A.bin_shift) & 0x3FFFFFFFu; const uint32_t b = bucket_of(bin); s0 = bk[b]; s1 = bk[b + 1]; }
			const bool big = s1 - s0 > 64u;
			uint32_t enter = kCsOrderUnknown;
			if (!big) for (uint32_t i = s0; i < s1; i += 4u) {
				uint2 e[4];
#pragma unroll
				for (int j = 0; j < 4; ++j) e[j] = i + (uint32_t) j < s1 ? my[i + (uint32_t) j] : none;
#pragma unroll
				for (int j = 0; j < 4; ++j) enter = min(enter, qualifies(e[j], bin));
			}
			// (the first 128 hits of the next such bucket are loaded before this one's are looked at)
			unsigned long long bm = __ballot(big);
			auto start = [&](int kb, uint32_t &bin2, uint32_t &t0, uint32_t &t1, uint2 &h0, uint2 &h1) {
				bin2 = (uint32_t) __builtin_amdgcn_readlane((int) bin, kb); t0 = (uint32_t) __builtin_amdgcn_readlane((int) s0, kb); t1 = (uint32_t) __builtin_amdgcn_readlane((int) s1, kb);
				h0 = t0 + (uint32_t) lane < t1 ? my[t0 + (uint32_t) lane] : none;
				h1 = t0 + 64u + (uint32_t) lane < t1 ? my[t0 + 64u + (uint32_t) lane] : none;
			};
			int ck = -1;
			uint32_t cbin = 0, ct0 = 0, ct1 = 0;
			uint2 ch0 = none, ch1 = none;
			if (bm) { ck = (int) __builtin_ctzll(bm); bm &= bm - 1ull; start(ck, cbin, ct0, ct1, ch0, ch1); }
			while (ck >= 0) {
				int nk = -1;
				uint32_t nbin = 0, nt0 = 0, nt1 = 0;
				uint2 nh0 = none, nh1 = none;
				if (bm) { nk = (int) __builtin_ctzll(bm); bm &= bm - 1ull; start(nk, nbin, nt0, nt1, nh0, nh1); }
				uint32_t en = min(qualifies(ch0, cbin), qualifies(ch1, cbin));
				for (uint32_t i = ct0 + 128u + (uint32_t) lane; i < ct1; i += 64u) en = min(en, qualifies(my[i], cbin));
				en = (uint32_t) wave_reduce_min((int) min(en, 0x7FFFFFFFu));
				if (lane == ck) enter = en == 0x7FFFFFFFu ? kCsOrderUnknown : en;
				ck = nk; cbin = nbin; ct0 = nt0; ct1 = nt1; ch0 = nh0; ch1 = nh1;
			}
			if (c < cn) cand_rank[cb + c] = enter == kCsOrderUnknown ? kCsOrderUnknown : 2u * enter + (cand_sv[cb + c] & 1u);
		}
		if (tid == 0) { info[2 * item_ix] = H; info[2 * item_ix + 1] = 0u; }
		mark(5);
		if (dg) { atomicAdd(&diag[8], 1ull); atomicAdd(&diag[9], (unsigned long long) H); atomicAdd(&diag[10], (unsigned long long) cn); }
	}
}

template <int NT>
__global__ __launch_bounds__(NT) void cs_order_bucket_kernel(CsArgs A, uint32_t n_list, uint32_t *work_counter, uint2 *scratch, uint32_t scratch_cap, uint32_t coarse_cap,
		const uint32_t *cand_loc, const uint32_t *cand_sv, uint32_t *cand_rank, uint32_t *info, unsigned long long *diag) {
	cs_order_bucket_body<NT>(A, n_list, work_counter, scratch, scratch_cap, coarse_cap, cand_loc, cand_sv, cand_rank, info, diag);
}

}  // namespace ngm
