// cs_canon_device.h -- candidate search over the CANONICAL bucket layout (refindex.h, round 3).
//
// Same semantics as cs_fast2_kernel (cs_device.h): CS::PrefixIteration (src/CSstatic.cpp:26-76), GetRefEntry
// (src/PrefixTable.cpp:750-817), PrefixSearch / AddLocationStd (src/CS.cpp:114-213), CollectResultsStd (src/CS.cpp:263-313);
// what changes is how the index is read and how a read's dependent steps are ordered.  The search always needs the list of a
// k-mer AND of its reverse complement (CS.cpp:116-160), and profiles/r02_gather_calibration.txt says a random request costs the
// same from 4 to 128 bytes, so:
//   * the two lists of a pair live side by side in ONE bucket (refindex.h): half the lookups;
//   * k-mer -> bucket address is arithmetic (the k-mer with the strand bit of its middle base removed), so the first 128-byte
//     line of every bucket of the read is requested BLIND, all of them back to back, lane groups of 8 x 16 bytes -- one full-line
//     request each, no index entry -> list dependency; the header word arrives with the first 31 positions;
//   * only the pairs with more than 31 positions (~45 % at GRCh38 size) cost a second request: those 16-byte chunks -- and the
//     chunks of the rare lists that do not fit a bucket -- are enumerated in LDS from the headers and requested BEFORE the first
//     lines vote, so that their latency hides behind those votes;
//   * workgroups are persistent: the characters of a workgroup's next read are requested while the current one is processed.
// 138 + ~62 requests per 150 bp read instead of 357, and one exposed memory round trip per read instead of three to five.
//
// Votes as in cs_fast2_kernel: sweep 1 = atomicOr into a "bin seen" bit plane; a hit that finds its bit set is a repeat and goes
// through a per-wave queue into the small exact table.  What is new is the end: a table entry holds all votes of its bin but
// possibly the first one (the hit that set the bit).  Instead of looking every first-on-its-bit hit up again (sweep 2 of
// cs_fast2_kernel: a rebuilt plane, 32 random LDS reads per lane, a second queue), only the entries that can still matter are
// completed: with m = the largest count in the table, the true maximum is >= m, the final threshold >= m * sensitivity, and an
// entry with count c has at most c + 1 votes -- so entries with c + 1 < m * sensitivity are neither candidates nor the maximum.
// The few that remain (the read's true locus and its close repeats; up to kCsCanonRel) are compared against the bins kept in
// registers, which adds the missing first votes exactly; reads with more such entries take the general sweep 2.
#pragma once

#include <type_traits>

#include "cs_device.h"

namespace ngm {

constexpr int kCsCanonFirstLineWords = 32;   // words of a bucket fetched blind (one 128-byte line)
constexpr int kCsCanonRel = 8;               // table entries completed by comparison (more: general sweep 2)
constexpr int kCsCanonDraw = 4;              // reads a persistent workgroup draws from the launch's read counter at a time

// SH2: the bins are four bases wide (bin_shift 2, the default): a plane word's byte address is a mask of the diagonal.
// T waves per read; R1 rounds of blind first-line loads (a round covers T * 64 >> glog buckets, glog = log2 of the lanes per
// first line: 3 for buckets of 32 words and more); R2 rounds of 16-byte chunk items.  R1 and R2 are even: votes are cast in
// steps of 8 slots (two rounds), the shape the queue bookkeeping of cs_fast2_kernel was tuned for.
template <int T, int R1, int R2, int CH = 1, int WPE = 8, bool SH2 = false>
__global__ __launch_bounds__(T * 64) __attribute__((amdgpu_waves_per_eu(T == 3 ? WPE : 1, T == 3 ? WPE : 8))) void cs_canon_kernel(CsArgs A) {
	static_assert(R1 % 2 == 0 && R2 % 2 == 0, "steps of two rounds");
	constexpr int NT = T * 64;
	constexpr int S1 = R1 / 2, S2 = R2 / 2, NS = S1 + S2;
	constexpr uint32_t kItemCap = (uint32_t) R2 * (uint32_t) NT;
	extern __shared__ __attribute__((aligned(16))) uint32_t cs_lds[];
	// the workgroup's few shared variables live BEHIND the dynamic arrays (no static LDS): the bit plane -- the address every vote
	// computes -- then starts at LDS address 0, and its word address needs no base added (one VALU instruction less per slot)
	struct Shared {
		uint32_t tot[T][3];   // per wave: chunk items, hits, k-mers looked up
		int len[T];
		uint32_t abort, nkeys, nrel;
		int next_read, draw_next, draw_left;
		uint32_t mx[T][2];
		uint32_t rel_key[kCsCanonRel], rel_slot[kCsCanonRel];
	};
	const int tid0 = threadIdx.x;
	const int k = A.k;
	const int kcap = A.lists_cap >> 1;                      // k-mers a read can have
	// (the bit plane first: its word addresses are then a constant away from the hash -- one add less per vote)
	uint32_t *plane = cs_lds;
	typedef __attribute__((address_space(3))) uint32_t lds_u32;
	lds_u32 *const plane0 = (lds_u32 *) (uint32_t) 0;   // == plane: the kernel has no static LDS, the dynamic array starts at 0
	const uint32_t plane_words = A.plane_bits >> 5;
	uint32_t *l_kinfo = plane + plane_words;                 // [kcap] bucket number | pair member << 30 | valid << 31
	uint32_t *l_hdr = l_kinfo + kcap;                        // [kcap] bucket header of the k-mers in use, else 0
	uint8_t *l_code = (uint8_t *) (l_hdr + kcap);
	uint16_t *l_items = (uint16_t *) ((uint32_t *) l_code + (A.q + 3) / 4);   // [kItemCap] k-mer << 8 | chunk
	uint32_t *t_keys = (uint32_t *) (l_items + kItemCap);
	const int log2_slots = A.log2_slots;
	const uint32_t n_slots = 1u << log2_slots;
	uint32_t *t_votes = t_keys + n_slots;
	const uint32_t q_cap = ((n_slots * 3u) / 4u) / (uint32_t) T;  // per wave
	Shared &S = *reinterpret_cast<Shared *>(t_votes + n_slots + (n_slots * 3u) / 4u);
	const int lw = A.bucket_log2_words;
	const int glog = min(lw, 5) - 2;                       // lanes per first line: 1, 2, 4, 8
	const int bpr = NT >> glog;                            // buckets per round
	const uint32_t flw = min(1u << lw, (uint32_t) kCsCanonFirstLineWords);
	// the plane holds a power of two of bits and is indexed by the low bits of the bin itself: genome positions of background hits are
	// uniform, so no hash is needed (two multiplies less per vote); bins that collide are a multiple of plane_bits x 4 bp apart on the
	// same diagonal -- rare, and a collision only sends a hit through the exact table
	const uint32_t pmask = A.plane_bits - 1u;
	const uint32_t wmask4 = ((A.plane_bits >> 5) - 1u) << 2;   // byte address of a plane word from (bin << 2)
	const int wbits = 31 - __clz((int) (A.plane_bits >> 5));   // log2 of the plane's words
	const int hs = 32 - log2_slots;
	const int cb = 2 * (k >> 1) + 1;  // the bit that tells the two k-mers of a pair apart (refindex.h)
	// LDS operations of one wave complete in program order: the queue hand-over inside a wave needs no hardware barrier, only
	// the compiler kept from moving the accesses
	auto wave_sync = [] { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); };

	// persistent workgroups: reads b and b + grid are this workgroup's by position, every further one is drawn from a counter
	// (status[2]) -- a workgroup that only becomes resident when others have finished (the occupancy the API promises is not
	// always what the hardware admits) then finds little left instead of a full static share, and slow CUs do fewer reads
	// ... or (reads_per_wg > 0) many short-lived workgroups, each with its run of consecutive reads and the same prefetch inside the
	// run: no counter at all, the hardware balances the load, and -- unlike workgroups that live as long as the kernel -- a kernel
	// of another stream (the order replay of the other mapper instance, on a high-priority stream) gets onto the CUs as runs end
	if (tid0 == 0) { S.draw_next = 0; S.draw_left = 0; }   // (only thread 0 reads them)
	const int run = A.reads_per_wg;
	int read = A.read_lo + (run > 0 ? (int) blockIdx.x * run : (int) blockIdx.x), read_next = run > 0 ? read + 1 : read + (int) gridDim.x;
	const int read_end = run > 0 ? min(A.n, read + run) : A.n;
	uint32_t ch_next = (read < read_end && tid0 < A.q) ? (uint32_t) A.reads[(size_t) read * A.q + tid0] : 0u;
	for (; read < read_end; read = read_next, read_next = S.next_read) {
		// everything derived from the thread index is recomputed per read from a value the compiler cannot see through: hoisted out
		// of this loop those values would each hold a register for the whole kernel (78 spilled registers instead of 7)
		const bool diag = A.phase_cycles && (read & 255) == 0;
		const unsigned long long c_top = diag ? wall_clock64() : 0ull;
		int tid = tid0;
		asm volatile("" : "+v"(tid));
		const int lane = tid & 63, wv = tid >> 6;
		uint32_t *my_queue = t_votes + n_slots + (uint32_t) wv * q_cap;
		const uint32_t sub = (uint32_t) tid & ((1u << glog) - 1u);
		const int b0 = tid >> glog;
		const uint8_t *rp = A.reads + (size_t) read * A.q;
		const uint32_t ch0 = ch_next;
		// this workgroup's next read: its characters travel while this one is processed; the one after that is drawn now (a
		// returning L2 atomic: its latency hides behind the read, too)
		// (requested after the vote phase, not here: across the votes the compiler spills the register, and a spilled load is waited for
		// where it is issued)
		auto next_chars = [&]() { ch_next = (read_next < read_end && tid < A.q) ? (uint32_t) A.reads[(size_t) read_next * A.q + tid] : 0u; };
		// (kCsCanonDraw reads per draw: ONE counter for the whole launch serves ~86 M returning atomics per second -- measured with
		// NGM_HIP_CS_STOP: a launch that leaves every read after its setup phase took as long as the whole kernel, 6.1 ms per
		// 524 288 reads, because every read drew its successor separately)
		int drawn = 0;
		if (tid == 0) {
			if (run > 0) drawn = read_next + 1;
			else if (S.draw_left > 0) { drawn = S.draw_next; S.draw_next = drawn + 1; S.draw_left -= 1; }
			else { drawn = A.read_lo + (int) (2u * gridDim.x + atomicAdd(&A.status[2], (uint32_t) kCsCanonDraw)); S.draw_next = drawn + 1; S.draw_left = kCsCanonDraw - 1; }
		}
		for (uint32_t s = tid; s < n_slots; s += NT) { t_keys[s] = 0xFFFFFFFFu; t_votes[s] = 0; }
		for (uint32_t s = tid; s < plane_words; s += NT) plane[s] = 0;
		const unsigned long long c0 = diag ? wall_clock64() : 0ull;

		// 0. codes (A0 C1 T2 G3, CSstatic.cpp:20-22; N = 4; past the end = 255), read length
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
			if (lane == 0) S.len[wv] = first_nul;
			__syncthreads();
			// (reset here, not at the top: a wave that is still deciding on the previous read's S.abort has not passed the barrier above)
			if (tid == 0) { S.abort = 0; S.nkeys = 0; S.nrel = 0; }
			R.L = S.len[0];
#pragma unroll
			for (int w2 = 1; w2 < T; ++w2) R.L = min(R.L, S.len[w2]);
		}
		const int L = R.L, n_kmers = max(L - k + 1, 0);
		R.n_lists = 2 * n_kmers;
		// k-mer -> bucket of its pair, one k-mer per lane
		for (int p = tid; p < n_kmers; p += NT) {
			bool v = true;
			uint32_t kmer = 0;
			for (int j = 0; j < k; ++j) {
				const uint32_t c = l_code[p + j];
				v = v && (c < 4);
				kmer = (kmer << 2) | (c & 3u);
			}
			if (v && p + k == L && p >= 1 && l_code[p - 1] == 4 && (p == 1 || l_code[p - 2] == 4)) v = false;  // CSstatic.cpp:30-41, see cs_prepare
			uint32_t info = 0;
			if (v) {
				const uint32_t member = (kmer >> cb) & 1u;               // 1: the k-mer is the reverse complement of its pair's canonical one
				const uint32_t y = member ? cs_revcomp(kmer, k) : kmer;
				info = ((y >> (cb + 1)) << cb) | (y & ((1u << cb) - 1u)) | (member << 30) | 0x80000000u;
			}
			l_kinfo[p] = info;
			l_hdr[p] = 0;
		}
		__syncthreads();

		// 1. the first line of every bucket, blind
		CsU4 d[R1];
#pragma unroll
		for (int r = 0; r < R1; ++r) {
			d[r] = CsU4{0u, 0u, 0u, 0u};
			const int b = r * bpr + b0;
			if (b < n_kmers) {
				const uint32_t info = l_kinfo[b];
				if (info >> 31) d[r] = *reinterpret_cast<const CsU4 *>(A.buckets + ((size_t) (info & 0x3FFFFFFFu) << lw) + sub * 4u);
			}
		}
		const unsigned long long c1 = diag ? wall_clock64() : 0ull;
		// the headers: which k-mers are in use (CS.cpp:122), how much of their lists lies beyond the first line
#pragma unroll
		for (int r = 0; r < R1; ++r) {
			const int b = r * bpr + b0;
			if (sub == 0u && b < n_kmers) {
				const uint32_t hdr = d[r].x;
				const uint32_t ntot = (hdr & kCsHdrCountMask) + ((hdr >> 14) & kCsHdrCountMask);
				if ((int) ntot < A.max_kfreq) l_hdr[b] = hdr;   // (k-mers without a bucket read 0)
			}
		}
		if (tid == 0) S.next_read = drawn;   // (the atomic was issued before the loads above: it has returned with them)
		__syncthreads();
		const unsigned long long c1a = diag ? wall_clock64() : 0ull;

		// 2. chunk items: the part of a pair's lists beyond the first line (16-byte chunks of the bucket's further words), or both
		// lists of a pair that does not fit its bucket (16-byte chunks of the position table copy behind the buckets)
		{
			uint32_t nch = 0, hits = 0, looked = 0;
			if (tid < n_kmers) {
				const uint32_t hdr = l_hdr[tid];
				const uint32_t na = hdr & kCsHdrCountMask, nb = (hdr >> 14) & kCsHdrCountMask, ntot = na + nb;
				hits = ntot;
				looked = l_kinfo[tid] >> 31;
				if (hdr & kCsHdrOverflow) nch = (na + 3u) / 4u + (nb + 3u) / 4u;
				else if (ntot > flw - 1u) nch = (ntot - (flw - 1u) + 3u) / 4u;
			}
			const uint32_t incl = wave_inclusive_scan(nch, lane);
			const uint32_t hsum = wave_last(wave_inclusive_scan(hits, lane));
			const uint32_t nv = (uint32_t) __popcll(__ballot(looked != 0u));
			if (lane == 63) { S.tot[wv][0] = incl; S.tot[wv][1] = hsum; S.tot[wv][2] = nv; }
			__syncthreads();
			uint32_t o = incl - nch, n_items = 0, H = 0, n_valid = 0;
#pragma unroll
			for (int w2 = 0; w2 < T; ++w2) { if (w2 < wv) o += S.tot[w2][0]; n_items += S.tot[w2][0]; H += S.tot[w2][1]; n_valid += S.tot[w2][2]; }
			for (uint32_t c = 0; c < nch; ++c, ++o) if (o < kItemCap) l_items[o] = (uint16_t) (((uint32_t) tid << 8) | c);
			R.H = H; R.n_valid = n_valid; R.n_items = n_items;
			__syncthreads();
		}
		const unsigned long long c1b = diag ? wall_clock64() : 0ull;
		const uint32_t H = R.H;
		const uint32_t n_items = R.n_items;
		auto stop_here = [&]() { if (tid == 0) { A.cand_base[read] = 0; A.cand_count[read] = 0; A.max_votes[read] = 0.f; A.read_len[read] = (uint16_t) R.L; } };
		if (A.debug_stop == 1) { stop_here(); next_chars(); continue; }
		if (H > A.hit_cap || n_items > kItemCap || n_kmers > NT || n_kmers > R1 * bpr) { if (wv == 0) cs_enqueue(A, read, lane, R); next_chars(); continue; }

		// item -> address of its 16 bytes; its slots are positions [first, first + 4) of the bucket's (or the pair's) hit numbering
		auto item_meta = [&](uint32_t idx, uint32_t &p, uint32_t &first, uint32_t &lim, uint32_t &na) -> size_t {
			const uint32_t item = l_items[idx];
			p = item >> 8;
			const uint32_t c = item & 255u;
			const uint32_t hdr = l_hdr[p];
			const uint32_t nb = (hdr >> 14) & kCsHdrCountMask;
			na = hdr & kCsHdrCountMask;
			const size_t bucket = (size_t) (l_kinfo[p] & 0x3FFFFFFFu) << lw;
			if (!(hdr & kCsHdrOverflow)) { first = flw + 4u * c - 1u; lim = na + nb; return bucket + flw + 4u * c; }
			const uint32_t ca = (na + 3u) / 4u;
			const bool in_b = c >= ca;
			const uint32_t cc = in_b ? c - ca : c;
			first = (in_b ? na : 0u) + 4u * cc;
			lim = in_b ? na + nb : na;
			return (size_t) A.pos_base + A.buckets[bucket + (in_b ? 2u : 1u)] + 4u * cc;
		};
		CsU4 d2[R2];
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
					if (probes >= n_slots) { slot = 0xFFFFFFFFu; break; }  // the other waves filled the table meanwhile
					const uint32_t prev = atomicCAS(&t_keys[slot], 0xFFFFFFFFu, bin);
					if (prev == bin) break;
					if (prev == 0xFFFFFFFFu) { ++fresh; break; }
					slot = (slot + 1) & (n_slots - 1);
				}
				if (slot != 0xFFFFFFFFu) atomicAdd(&t_votes[slot], (e & 0x80000000u) ? 0x10000u : 1u); else S.abort = 1u;
			}
			uint32_t total;
			(void) wave_prefix_small<4>(fresh, total);  // fresh <= q_cap / 64 < 16
			uint32_t before = 0;
			if (lane == 0 && total) before = atomicAdd(&S.nkeys, total);
			before = wave_first(before);
			if (before + total > (n_slots * 3u) / 4u) { abort_fast = true; if (lane == 0) S.abort = 1u; }  // probing gets slow, the spurious entries too many
			wave_sync();
			q_len = 0;
		};

		uint32_t bins[NS * kCsSeg];   // bin | first-on-its-bit << 30 | reverse strand << 31 ; 0 = empty slot
		// one step: 8 slots per lane vote (sweep 1 of cs_fast2_kernel): positions, validity, diagonal correction and strand per slot
		// (revf: the hit's strand in bit 31 and the "first on its bit" flag, bit 30, that the entries kept in `bins` carry; the queue ignores bit 30)
		auto vote_step = [&](auto nslots, const int step, const uint32_t (&pos)[kCsSeg], const bool (&valid)[kCsSeg], const uint32_t (&corr)[kCsSeg], const uint32_t (&revf)[kCsSeg], const bool last_step) {
			constexpr int NSL = decltype(nslots)::value;   // 8, or 4 when the step's second round is empty for the whole workgroup
			uint32_t dup[kCsSeg], msk[kCsSeg], ent[kCsSeg];
#pragma unroll
			for (int j = 0; j < NSL; ++j) {
				// plane bit of a bin: word = its low bits, bit = the five bits above them.  With bins of four bases (bin_shift 2, the
				// default) the word's BYTE address is a mask of the diagonal itself -- (t >> 2) & (W - 1) words = t & ((W - 1) << 2) bytes --
				// one instruction instead of shift + mask (round 4).
				// (the plane starts at LDS address 0 -- see Shared above -- and is addressed as such: through the symbol of the dynamic array the
				// compiler keeps an `add 0` per vote, the symbol's address being a link-time constant)
				const uint32_t t = pos[j] - corr[j];
				const uint32_t bin = (t >> A.bin_shift) & 0x3FFFFFFFu;
				const uint32_t wbyte = SH2 ? (t & wmask4) : ((t >> (A.bin_shift - 2)) & wmask4);   // (the fast path is not used with bins below four bases)
				msk[j] = valid[j] ? (1u << ((bin >> wbits) & 31)) : 0u;      // empty slots vote with an all-zero mask: branch-free
				dup[j] = __hip_atomic_fetch_or((lds_u32 *) (uintptr_t) wbyte, msk[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) & msk[j];   // != 0: a repeat on its bit
				ent[j] = bin | revf[j];
			}
			uint32_t ndup = 0;
#pragma unroll
			for (int j = 0; j < NSL; ++j) ndup += dup[j] ? 1u : 0u;
			uint32_t qb;
			{ uint32_t total; qb = q_len + wave_prefix_small<4>(ndup, total); q_len += total; }
#pragma unroll
			for (int j = 0; j < kCsSeg; ++j) bins[step * kCsSeg + j] = (j < NSL && (msk[j] ^ dup[j])) ? ent[j] : 0u;  // valid and first on its bit; repeats vote in sweep 1: nothing left to do for them
			// the repeats go through this wave's queue: inserted when a good batch is waiting and after the last step; when one step
			// brings more than the queue holds (repetitive reads) it is filled and emptied window by window
			uint32_t window = 0;
			for (;;) {
				uint32_t at = qb;
#pragma unroll
				for (int j = 0; j < NSL; ++j) if (dup[j] != 0u) { if (at - window < q_cap) my_queue[at - window] = ent[j]; ++at; }
				const bool more = q_len - window > q_cap;
				if (more || q_len - window > (T >= 4 ? 0u : q_cap / 2u) || last_step) {
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
		};

		auto issue_chunks = [&]() {
#pragma unroll
			for (int r = 0; r < R2; ++r) {
				d2[r] = CsU4{0u, 0u, 0u, 0u};
				const uint32_t idx = (uint32_t) r * (uint32_t) NT + (uint32_t) tid;
				if (idx < n_items) {
					uint32_t p, first, lim, na;
					const size_t at = item_meta(idx, p, first, lim, na);
					d2[r] = *reinterpret_cast<const CsU4 *>(A.buckets + at);
				}
			}
		};
		if (CH == 0) issue_chunks();
		// votes of the first lines; the chunk loads are issued after step CH - 1 and travel under the remaining steps
#pragma unroll
		for (int s = 0; s < S1; ++s) {
			if (CH > 0 && s == (CH < S1 ? CH : S1 - 1) && CH < S1) issue_chunks();
			if ((uint32_t) (2 * s) * (uint32_t) bpr >= (uint32_t) n_kmers) {  // block-uniform: nothing in these rounds
#pragma unroll
				for (int j = 0; j < kCsSeg; ++j) bins[s * kCsSeg + j] = 0;
				continue;
			}
			uint32_t pos[kCsSeg], corr[kCsSeg], rev[kCsSeg];
			bool valid[kCsSeg];
#pragma unroll
			for (int h = 0; h < 2; ++h) {
				const int r = 2 * s + h;
				const int b = r * bpr + b0;
				const uint32_t info = b < n_kmers ? l_kinfo[b] : 0u;
				const uint32_t hdr = b < n_kmers ? l_hdr[b] : 0u;          // 0 for k-mers that are not in use
				const uint32_t na = hdr & kCsHdrCountMask, ntot = na + ((hdr >> 14) & kCsHdrCountMask);
				const bool member = ((info >> 30) & 1u) != 0u;
				const uint32_t lim = (hdr & kCsHdrOverflow) ? 0u : ntot;
				const uint32_t cf = (uint32_t) b, cr = (uint32_t) (L - (b + k));   // CS.cpp:140-142
				// first list of the bucket = the canonical k-mer's, second = its reverse complement's; `member` says which of the two the read has
				const uint32_t c1 = member ? cr : cf, c2 = member ? cf : cr, f1 = member ? 0xC0000000u : 0x40000000u, f2 = member ? 0x40000000u : 0xC0000000u;
				const uint32_t w4[4] = {d[r].x, d[r].y, d[r].z, d[r].w};
#pragma unroll
				for (int e = 0; e < 4; ++e) {
					const uint32_t idx = sub * 4u + (uint32_t) e - 1u;    // position index inside the bucket (word 0 is the header: wraps to "invalid")
					const bool second = idx >= na;
					valid[h * 4 + e] = idx < lim;
					rev[h * 4 + e] = second ? f2 : f1;
					corr[h * 4 + e] = second ? c2 : c1;
					pos[h * 4 + e] = w4[e];
				}
			}
			vote_step(std::integral_constant<int, kCsSeg>{}, s, pos, valid, corr, rev, n_items == 0u && ((uint32_t) (2 * s + 2) * (uint32_t) bpr >= (uint32_t) n_kmers || s + 1 == S1));
		}
		const unsigned long long c1c = diag ? wall_clock64() : 0ull;
		if (CH >= S1) issue_chunks();
		// votes of the chunks
#pragma unroll
		for (int s = 0; s < S2; ++s) {
			if ((uint32_t) (2 * s) * (uint32_t) NT >= n_items) {  // block-uniform
#pragma unroll
				for (int j = 0; j < kCsSeg; ++j) bins[(S1 + s) * kCsSeg + j] = 0;
				continue;
			}
			uint32_t pos[kCsSeg], corr[kCsSeg], rev[kCsSeg];
			bool valid[kCsSeg];
#pragma unroll
			for (int h = 0; h < 2; ++h) {
				const int r = 2 * s + h;
				const uint32_t idx = (uint32_t) r * (uint32_t) NT + (uint32_t) tid;
				uint32_t p = 0, first = 0, lim = 0, na = 0;
				bool member = false;
				if (idx < n_items) { (void) item_meta(idx, p, first, lim, na); member = ((l_kinfo[p] >> 30) & 1u) != 0u; }
				const uint32_t cf = p, cr = (uint32_t) (L - ((int) p + k));
				const uint32_t c1 = member ? cr : cf, c2 = member ? cf : cr, f1 = member ? 0xC0000000u : 0x40000000u, f2 = member ? 0x40000000u : 0xC0000000u;
				const uint32_t w4[4] = {d2[r].x, d2[r].y, d2[r].z, d2[r].w};
#pragma unroll
				for (int e = 0; e < 4; ++e) {
					const uint32_t i2 = first + (uint32_t) e;
					const bool second = i2 >= na;
					valid[h * 4 + e] = i2 < lim;
					rev[h * 4 + e] = second ? f2 : f1;
					corr[h * 4 + e] = second ? c2 : c1;
					pos[h * 4 + e] = w4[e];
				}
			}
			// (most reads have fewer chunk items than lanes: the step's second round is then empty for every lane, and half a step is saved)
			if ((uint32_t) (2 * s + 1) * (uint32_t) NT >= n_items) vote_step(std::integral_constant<int, kCsSeg / 2>{}, S1 + s, pos, valid, corr, rev, true);
			else vote_step(std::integral_constant<int, kCsSeg>{}, S1 + s, pos, valid, corr, rev, (uint32_t) (2 * s + 2) * (uint32_t) NT >= n_items || s + 1 == S2);
		}
		__syncthreads();
		const unsigned long long c2 = diag ? wall_clock64() : 0ull;
		next_chars();
		if (A.debug_stop == 2) { stop_here(); continue; }
		if (S.abort) { if (wv == 0) cs_enqueue(A, read, lane, R); continue; }  // not provably exact here

		// 3. complete the entries that can still matter (see the header): largest count, then the entries within reach of it
		{
			uint32_t mx = 0, mxb = 0;
			for (uint32_t s = tid; s < n_slots; s += NT) {
				const uint32_t v = t_votes[s];
				mx = max(mx, max(v & 0xFFFFu, v >> 16));
				mxb = max(mxb, (v & 0xFFFFu) + (v >> 16));
			}
			mx = (uint32_t) wave_reduce_max((int) mx);
			mxb = (uint32_t) wave_reduce_max((int) mxb);
			if (lane == 0) { S.mx[wv][0] = mx; S.mx[wv][1] = mxb; }
		}
		__syncthreads();
		uint32_t mx_lo = 0, mxb_lo = 0;
#pragma unroll
		for (int w2 = 0; w2 < T; ++w2) { mx_lo = max(mx_lo, S.mx[w2][0]); mxb_lo = max(mxb_lo, S.mx[w2][1]); }
		{
			// count + 1 >= m * sensitivity, in integers: the left side is one (counts are below 65 536: exact in a float)
			const uint32_t reach = (uint32_t) ceilf((float) mx_lo * A.sensitivity);
			const bool both = A.max_both != nullptr;
			for (uint32_t s = tid; s < n_slots; s += NT) {
				const uint32_t key = t_keys[s];
				if (key == 0xFFFFFFFFu) continue;
				const uint32_t v = t_votes[s];
				const uint32_t f = v & 0xFFFFu, r = v >> 16;
				if (max(f, r) + 1u >= reach || (both && f + r + 1u >= mxb_lo)) {
					const uint32_t at = atomicAdd(&S.nrel, 1u);
					if (at < (uint32_t) kCsCanonRel) { S.rel_key[at] = key | 0x40000000u; S.rel_slot[at] = s; }
				}
			}
		}
		__syncthreads();
		const uint32_t n_rel = S.nrel;
		if (n_rel <= (uint32_t) kCsCanonRel) {
			for (uint32_t j = 0; j < n_rel; ++j) {   // (typically one or two entries: the keys stay scalar)
				const uint32_t key = (uint32_t) __builtin_amdgcn_readfirstlane((int) S.rel_key[j]);
				uint32_t hit = 0;   // bin | first-on-its-bit flag (| strand << 31): at most one such hit per bin in the whole workgroup; empty slots are 0, keys are not
#pragma unroll
				for (int i = 0; i < NS * kCsSeg; ++i) {
					const uint32_t e = bins[i];
					hit = ((e ^ key) << 1) == 0u ? e : hit;
				}
				if (hit) atomicAdd(&t_votes[S.rel_slot[j]], (hit >> 31) ? 0x10000u : 1u);
			}
			__syncthreads();
			if (A.debug_stop == 3) { stop_here(); continue; }
			const unsigned long long c3 = diag ? wall_clock64() : 0ull;
			// 4. threshold and candidates (cs_finish, CS.cpp:201-205, :263-313) over the completed entries -- all others are out of reach
			if (wv == 0) {
				const bool mine = (uint32_t) lane < n_rel;
				const uint32_t slot = mine ? S.rel_slot[lane] : 0u;
				const uint32_t v = mine ? t_votes[slot] : 0u;
				const uint32_t f = v & 0xFFFFu, r = v >> 16;
				int mxi = wave_reduce_max((int) max(f, r)), mxbi = wave_reduce_max((int) (f + r));
				mxi = max(mxi, (int) mx_lo); mxbi = max(mxbi, (int) mxb_lo);   // (n_rel = 0: an empty table)
				if (H > 0 && mxi < 2) mxi = 1;   // only single votes survived the filter: the true maximum is 1
				if (H > 0 && mxbi < 2) mxbi = 1;
				const float max_hit = (float) mxi;
				const float thresh = fmaxf(A.kmer_min, max_hit * A.sensitivity);
				if (H > 0 && !(thresh > 1.0f)) cs_enqueue(A, read, lane, R);   // the filter dropped bins with a single vote: exact only if those cannot be candidates
				else {
					const uint32_t region = (uint32_t) read & (kCsRegions - 1);
					if (lane == 0 && A.counters) {
						atomicAdd(&A.counters[region * kCsCursorStride], (unsigned long long) R.n_valid);
						atomicAdd(&A.counters[region * kCsCursorStride + 1], (unsigned long long) H);
					}
					const uint32_t count = mine ? ((float) f >= thresh) + ((float) r >= thresh) : 0u;
					// output order of cs_finish: by table slot, lane-major (slot mod 64, then slot / 64)
					const uint32_t okey = mine ? ((slot & 63u) << 16) | (slot >> 6) : 0xFFFFFFFFu;
					uint32_t before = 0, total = 0;
#pragma unroll
					for (int j = 0; j < kCsCanonRel; ++j) {
						const uint32_t cj = (uint32_t) __builtin_amdgcn_readlane((int) count, j), kj = (uint32_t) __builtin_amdgcn_readlane((int) okey, j);
						total += cj;
						if (kj < okey) before += cj;
					}
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
						if (A.max_both) A.max_both[read] = (float) mxbi;
						A.read_len[read] = (uint16_t) R.L;
						if (A.counters && total) atomicAdd(&A.counters[region * kCsCursorStride + 2], (unsigned long long) total);
					}
					if (total != 0u) {
						bool ok = true;
						uint32_t w;
						if (fixed) w = A.fixed_base + (uint32_t) read * (uint32_t) kCsFixedSlots + before;
						else {
							base = wave_first((uint32_t) base) | ((unsigned long long) wave_first((uint32_t) (base >> 32)) << 32);
							ok = base + total <= A.out_capacity;
							w = (uint32_t) (region * A.out_capacity + base) + before;
						}
						if (ok && mine) {
							const uint32_t centre = A.bin_shift > 0 ? (1u << (A.bin_shift - 1)) : 0u;  // ResolveBin, CS.h:170-175
							const uint32_t loc = ((t_keys[slot]) << A.bin_shift) + centre;
							if ((float) f >= thresh) { A.out_loc[w] = loc; A.out_sv[w] = f << 1; ++w; }
							if ((float) r >= thresh) { A.out_loc[w] = loc; A.out_sv[w] = (r << 1) | 1u; ++w; }
						}
					}
				}
				if (diag && lane == 0) {
					atomicAdd(&A.phase_cycles[0], c1 - c0); atomicAdd(&A.phase_cycles[1], c2 - c1); atomicAdd(&A.phase_cycles[2], c3 - c2);
					atomicAdd(&A.phase_cycles[3], wall_clock64() - c3);
					atomicAdd(&A.phase_cycles[8], c0 - c_top);   // the resets and the prefetch in front of the timed phases
					atomicAdd(&A.phase_cycles[4], c1a - c1); atomicAdd(&A.phase_cycles[5], c1b - c1a); atomicAdd(&A.phase_cycles[6], c1c - c1b); atomicAdd(&A.phase_cycles[7], c2 - c1c);
				}
			}
			__syncthreads();   // the table is reused by the next read
			continue;
		}

		// general sweep 2 (cs_fast2_kernel): plane := bits of the bins in the table; first-on-bit hits whose bit is set add their vote
		for (uint32_t s = tid; s < plane_words; s += NT) plane[s] = 0;
		__syncthreads();
		for (uint32_t s = tid; s < n_slots; s += NT) {
			const uint32_t key = t_keys[s];
			if (key != 0xFFFFFFFFu) {
				const uint32_t b = key & pmask;
				atomicOr(&plane[b >> 5], 1u << (b & 31));
			}
		}
		__syncthreads();
		{
			unsigned long long wmask = 0ull;   // NS * 8 <= 48 slots per lane
#pragma unroll
			for (int i = 0; i < NS * kCsSeg; ++i) {
				const uint32_t e = bins[i];
				const uint32_t b = e & 0x3FFFFFFFu & pmask;
				const uint32_t w = (plane[b >> 5] >> (b & 31)) & (e >> 30) & 1u;
				wmask |= (unsigned long long) w << i;
			}
			const uint32_t nhit = (uint32_t) __popcll(wmask);
			uint32_t total;
			const uint32_t qb = wave_prefix_small<8>(nhit, total);  // nhit <= 48
			for (uint32_t window = 0; window < total; window += q_cap) {
				uint32_t at = qb;
#pragma unroll
				for (int i = 0; i < NS * kCsSeg; ++i) if ((wmask >> i) & 1ull) { if (at - window < q_cap) my_queue[at - window] = bins[i]; ++at; }
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
		if (wv == 0) {
			if (!cs_finish<kCsFast>(A, read, lane, R, t_keys, t_votes, n_slots)) cs_enqueue(A, read, lane, R);
		}
		__syncthreads();
	}
}

}  // namespace ngm
