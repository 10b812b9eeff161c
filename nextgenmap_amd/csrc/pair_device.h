// pair_device.h -- paired-end selection of the pairs WITH choices on the GPU (round 5): what ScoreBuffer::top1PE + CheckPairs
// (src/ScoreBuffer.cpp:368-502) decide from the scores alone.
//
// pair_simple_kernel (gather_device.h) settles the pairs whose mates have one candidate each.  Every other pair with candidates
// on both sides used to be walked by the host (select_pair, mapper.cpp): two sorts and na x nb insert-size checks per pair -- with
// the hundreds of candidates per mate of a repeat-rich genome 16-20 ms of a 64-thread pool per 262 144 reads, on a 16-CPU quota.
// Here one workgroup per such pair computes everything that does not depend on the reference's sequential state:
//   * per mate the best and second-best score (computeMQ, :34-49) and the candidates at or above best * pair_score_cutoff (:382-396);
//   * every combination of those inside the insert-size window (CheckPairs, :463-502): its pair score, insert size, candidates;
//   * no two in-window combinations with the same pair score  ->  the winner is the unique best one whatever the running mean
//     insert size and whatever the candidate order: settled here (found / not found, winners, MAPQs, insert size);
//   * otherwise ("tied")  ->  what the host's sequential passes need: found, whether two combinations also share the insert size
//     (`dup`: the candidate order decides), the best-scoring combinations (up to 8) and the range of their insert sizes.
// Nothing here depends on the order of the candidates: the sets above are order-free, and a result is only called settled when it
// is unique.  Pairs beyond the kernel's limits (more than kPairCap candidates above the cut-off on one side; a negative best score
// shared by several candidates, where the reference keeps "the first") are flagged for the host's walk.
#pragma once

#include <stdint.h>
#include <hip/hip_runtime.h>

#include "cs_device.h"

namespace ngm {

constexpr int kPairThreads = 256, kPairCap = 2048, kPairCapHuge = 8192, kPairCombos = 64;
inline size_t pair_choice_lds_bytes(int cap) { return (size_t) cap * 12; }   // per mate: locations (32-bit) and candidate indices (16-bit) of the candidates above the cut-off
enum : int32_t { kPairFound = 1, kPairTied = 2, kPairDup = 4, kPairHost = 8 };

// flags: kPair* | mq_a << 8 | mq_b << 16 | n_top << 24 (n_top <= 8: the listed best-scoring combinations; 15: more than 8; 0 with more than
// kPairCombos combinations inside the window: none listed)
struct PairOut { int32_t flags, wa, wb, dist, dmin, dmax, tied_ix, pair; };   // pair: the pair this entry belongs to
struct PairTop { int32_t d[8], a[8], b[8]; };

__device__ __forceinline__ int pair_f2o(float f) { const int b = __float_as_int(f); return b ^ ((b >> 31) & 0x7FFFFFFF); }   // float order as signed-int order
__device__ __forceinline__ float pair_o2f(int o) { return __int_as_float(o ^ ((o >> 31) & 0x7FFFFFFF)); }

// list[0 .. *list_count): the pairs pair_simple_kernel left (both mates have candidates, at least one of them several); out[i] belongs to
// list[i] (PairOut::pair names it); persistent workgroups walk the list with the grid's stride.
// NT threads per pair, at most CAP candidates above the cut-off per mate: <64, 64> (one wave) for the pairs whose mates have up to 64
// candidates each -- nearly all of them --, <kPairThreads, kPairCap> for the others (pair_simple_kernel sorts them into the two lists).
// Round 6: a pair with more than CAP candidates above the cut-off on one side used to go back to the host's walk (~1 000 pairs per 131 072 reads of the
// bench's stress sub-leg, 100-400 ms of its pool per batch).  The kernel now hands such a pair to a list (`huge_list`: the pair's entry number) that a third
// launch with CAP = kPairCapHuge works through (`indirect`: the list then holds entry numbers, the pairs are indirect[entry]) and writes into the same entries;
// pairs of repeat families have far more than kPairCombos combinations inside the window, so that launch learns "found, tied, order decides" after a
// few trips of its combination loop and stops.  The candidate lists live in dynamic LDS (pair_choice_lds_bytes).
template <int NT, int CAP>
__global__ __launch_bounds__(NT) void pair_choice_kernel(const uint32_t *__restrict__ list, const uint32_t *__restrict__ list_count, const uint32_t *__restrict__ cand_base,
		const uint32_t *__restrict__ cand_count, const float *__restrict__ scores, const uint32_t *__restrict__ pair_loc, const uint16_t *__restrict__ read_len, int min_d, int max_d,
		float cutoff, PairOut *__restrict__ out, PairTop *__restrict__ tops, uint32_t *__restrict__ tied_count, uint32_t tied_cap,
		uint32_t *__restrict__ huge_list, uint32_t *__restrict__ huge_count, const uint32_t *__restrict__ indirect) {
	constexpr int NW = NT / 64;
	extern __shared__ __attribute__((aligned(16))) uint32_t pair_lds[];
	uint32_t (*s_loc)[CAP] = reinterpret_cast<uint32_t (*)[CAP]>(pair_lds);
	uint16_t (*s_ix)[CAP] = reinterpret_cast<uint16_t (*)[CAP]>(pair_lds + 2 * CAP);
	__shared__ int s_red[3][NW];
	__shared__ uint32_t s_n[2], s_ncombo;
	__shared__ int s_top;   // bits of the largest positive pair score inside the window (0: none)
	__shared__ float s_cps[kPairCombos];
	__shared__ int s_cd[kPairCombos], s_ca[kPairCombos], s_cb[kPairCombos];
	const uint32_t n_list = *list_count;
	for (uint32_t item_l = blockIdx.x; item_l < n_list; item_l += gridDim.x) {
	__syncthreads();   // (the previous pair's shared state is no longer read)
	const uint32_t item = indirect ? list[item_l] : item_l;   // the pair's entry in `out`
	const int pi = (int) (indirect ? indirect[item] : list[item_l]);
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const int rb = 2 * pi, ra = 2 * pi + 1;   // `a` = the mate whose scores arrive last in the reference (the odd read id): the outer loop of top1PE
	const uint32_t cnt[2] = {cand_count[ra], cand_count[rb]}, base[2] = {cand_base[ra], cand_base[rb]};
	const int len_a = (int) read_len[ra], len_b = (int) read_len[rb];
	if (tid == 0) { s_n[0] = s_n[1] = 0; s_ncombo = 0; s_top = 0; }
	__syncthreads();
	PairOut po{0, -1, -1, 0, 0, 0, -1, pi};
	bool to_host = false;
	int mq[2];
	for (int side = 0; side < 2; ++side) {
		// best score, how many candidates share it, the best of the others
		const uint32_t b0 = base[side], c = cnt[side];
		int mx = INT_MIN;
		for (uint32_t j = tid; j < c; j += NT) mx = max(mx, pair_f2o(scores[b0 + j]));
		mx = wave_reduce_max(mx);
		if (lane == 0) s_red[0][wv] = mx;
		__syncthreads();
		mx = s_red[0][0];
#pragma unroll
		for (int w = 1; w < NW; ++w) mx = max(mx, s_red[0][w]);
		int nb = 0, sec = INT_MIN;
		for (uint32_t j = tid; j < c; j += NT) { const int o = pair_f2o(scores[b0 + j]); if (o == mx) ++nb; else sec = max(sec, o); }
		nb = (int) wave_last(wave_inclusive_scan((uint32_t) nb, lane));
		sec = wave_reduce_max(sec);
		if (lane == 0) { s_red[1][wv] = nb; s_red[2][wv] = sec; }
		__syncthreads();
		nb = 0; sec = INT_MIN;
#pragma unroll
		for (int w = 0; w < NW; ++w) { nb += s_red[1][w]; sec = max(sec, s_red[2][w]); }
		const float best = pair_o2f(mx);
		const float second = nb > 1 ? best : (sec == INT_MIN ? 0.0f : pair_o2f(sec));
		int q = 60;   // computeMQ(MappedRead*), ScoreBuffer.cpp:42-49
		if (c > 1u) { q = 0; if (best > 0.0f && second >= 0.0f) q = (int) ceilf(60.0f * (best - second) / best); }
		mq[side] = q;
		// the candidates the pairing looks at: sorted by score, "while (n < numScores && min <= Scores[n])" from n = 1 (:382-396)
		const float mn = best * cutoff;
		const bool head_only = !(mn <= best);   // negative best score (or a cut-off above 1): only Scores[0] -- "the first" of the best
		if (head_only && nb > 1) to_host = true;
		for (uint32_t j = tid; j < c; j += NT) {
			const float s = scores[b0 + j];
			if (head_only ? (pair_f2o(s) == mx) : (mn <= s)) {
				const uint32_t at = atomicAdd(&s_n[side], 1u);
				if (at < (uint32_t) CAP) { s_loc[side][at] = pair_loc[b0 + j]; s_ix[side][at] = (uint16_t) min(j, 65535u); }
			}
		}
		if (c > 65535u) to_host = true;
		__syncthreads();
	}
	const uint32_t na = s_n[0], nbb = s_n[1];
	const bool over_cap = !to_host && (na > (uint32_t) CAP || nbb > (uint32_t) CAP);
	if (over_cap) to_host = true;
	if (to_host) {
		if (tid == 0) {
			po.flags = kPairHost | (mq[0] << 8) | (mq[1] << 16); out[item] = po;
			if (over_cap && huge_list) huge_list[atomicAdd(huge_count, 1u)] = item;   // (the launch with the larger lists overwrites the entry)
		}
		continue;
	}
	// every combination inside the insert-size window; the larger side across the lanes
	{
		const int inner = nbb >= na ? 1 : 0, outer = 1 - inner;
		const uint32_t n_out = s_n[outer], n_in = s_n[inner];
		for (uint32_t i = wv; i < n_out; i += NW) {
			if (*(volatile uint32_t *) &s_ncombo > (uint32_t) kPairCombos && *(volatile int *) &s_top > 0) break;   // tied beyond listing, and found: nothing more to learn
			const uint32_t lo = s_loc[outer][i];
			for (uint32_t j0 = 0; j0 < n_in; j0 += 64) {
				const uint32_t j = j0 + (uint32_t) lane;
				if (j >= n_in) continue;
				const uint32_t li = s_loc[inner][j];
				const uint32_t l1 = outer == 0 ? lo : li, l2 = outer == 0 ? li : lo;   // l1: mate a (ls1), l2: mate b (ls2)
				const uint64_t v = (l2 > l1) ? (uint64_t) (l2 - l1) + (uint64_t) len_b : (uint64_t) (l1 - l2) + (uint64_t) len_a;
				const int cur = (int) (uint32_t) v;
				if (cur > min_d && cur < max_d) {
					const int ia = (int) (base[0] + (outer == 0 ? s_ix[0][i] : s_ix[0][j])), ib = (int) (base[1] + (outer == 0 ? s_ix[1][j] : s_ix[1][i]));
					const float ps = scores[ia] + scores[ib];
					const uint32_t at = atomicAdd(&s_ncombo, 1u);
					if (at < (uint32_t) kPairCombos) { s_cps[at] = ps; s_cd[at] = cur; s_ca[at] = ia; s_cb[at] = ib; }
					if (ps > 0.0f) atomicMax(&s_top, __float_as_int(ps));
				}
			}
		}
	}
	__syncthreads();
	if (wv != 0) continue;
	const uint32_t nc = s_ncombo;
	const float top = __int_as_float(s_top);   // 0.0f: no positive pair score
	const bool found = s_top > 0;
	int flags = (found ? kPairFound : 0) | (mq[0] << 8) | (mq[1] << 16);
	int n_top = 0;
	if (nc > (uint32_t) kPairCombos) {
		flags |= kPairTied | kPairDup;
		po.dmin = min_d; po.dmax = max_d;
	} else if (nc > 0u) {
		const bool act = (uint32_t) lane < nc;
		const float ps = act ? s_cps[lane] : 0.0f;
		const int d = act ? s_cd[lane] : 0;
		bool eq = false, dp = false;
		for (uint32_t x = 0; x + 1 < nc; ++x) {
			const float px = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ps), (int) x));
			const int dx = __builtin_amdgcn_readlane(d, (int) x);
			if (act && (uint32_t) lane > x && ps == px) { eq = true; if (d == dx) dp = true; }
		}
		const bool tied = __ballot(eq) != 0ull;
		const bool dup = __ballot(dp) != 0ull;
		const bool is_top = act && found && ps == top;
		const unsigned long long tops_m = __ballot(is_top);
		n_top = (int) __popcll(tops_m);
		if (!tied) {
			if (found) {   // exactly one lane holds the best pair score
				const int w = (int) __builtin_ctzll(tops_m);
				po.wa = s_ca[w]; po.wb = s_cb[w]; po.dist = s_cd[w];
			}
		} else {
			flags |= kPairTied | (dup || n_top > 8 ? kPairDup : 0);
			int dmn = is_top ? d : INT_MAX, dmx = is_top ? d : 0;
			dmn = wave_reduce_min(dmn); dmx = wave_reduce_max(dmx);
			if (found) { po.dmin = dmn; po.dmax = dmx; }
		}
	}
	if (flags & kPairTied) {
		uint32_t ix = 0;
		if (lane == 0) ix = atomicAdd(tied_count, 1u);
		ix = wave_first(ix);
		if (ix < tied_cap) {
			po.tied_ix = (int32_t) ix;
			// the best-scoring combinations, in any order (the host asks which of them is closest to the running mean, and only
			// trusts a unique answer)
			if (nc <= (uint32_t) kPairCombos && found) {
				const bool act = (uint32_t) lane < nc;
				const bool is_top = act && s_cps[lane] == top;
				const unsigned long long tops_m = __ballot(is_top);
				if (is_top) {
					const int at = (int) __popcll(tops_m & ((1ull << lane) - 1ull));
					if (at < 8) { tops[ix].d[at] = s_cd[lane]; tops[ix].a[at] = s_ca[lane]; tops[ix].b[at] = s_cb[lane]; }
				}
			}
		} else flags = (flags & ~(kPairTied | kPairDup)) | kPairHost;   // (the list is sized for every pair: not reached)
	}
	if (lane == 0) { po.flags = flags | ((nc > (uint32_t) kPairCombos ? 0 : min(n_top, 15)) << 24); out[item] = po; }
	}
}

}  // namespace ngm
