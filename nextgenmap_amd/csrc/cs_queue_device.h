// cs_queue_device.h -- the queue of reads between the passes of a candidate search, kept on the device (round 6).
//
// A search pass hands the reads it cannot certify to the next one through a queue in HBM (cs_enqueue: read, index hits; the count in a
// status word).  Until round 5 the host downloaded that queue after every pass, dealt the reads into the classes of cs_heavy2_kernel,
// sized the tables of cs_global_kernel and uploaded the lists again: five to six stream synchronisations per batch on a GRCh38-like
// genome, each with the GPU idle while the turn was held.  The kernels here do those steps where the queue lives; every later pass is
// launched with a fixed grid of persistent (or grid-striding) workgroups that read their item count from a device word, so that the
// whole search -- fast path, heavy classes, second round, exact LDS tables, exact global tables, compaction -- is ONE enqueue with ONE
// synchronisation at its end.  Nothing here changes what a pass computes (src/CS.cpp:341-436 is the reference's own overflow handling).
#pragma once

#include "cs_device.h"

namespace ngm {

// words of the per-mapper control block (ngm_mapper::d_heavy_ctr)
constexpr int kCsqWork = 0;       // [0..2]  work counters of the three heavy classes (persistent workgroups draw reads from them)
constexpr int kCsqCount = 4;      // [4..6]  reads in the class lists
constexpr int kCsqRun = 8;        // [8..9]  reads run by the heavy kernels in round 0 / round 1 (statistics)
constexpr int kCsqSecond = 16;    // reads given a second pass (T from the first pass's maximum)
constexpr int kCsqRestart = 17;   // table passes started over with twice the parts
constexpr int kCsqSentOn = 18;    // reads a heavy class could not certify (queued again)
constexpr int kCsqWords = 32;

// words of the status block (ngm_mapper::d_status, 16 words): a pass writes the block it is given as CsArgs::status ([0] candidate
// output overflowed, [1] reads queued)
constexpr int kCsqStatusMain = 0;     // fast path and heavy classes; the exact LDS pass reads its list from this queue
constexpr int kCsqStatusExact = 4;    // written by the exact LDS pass (reads beyond its table: to the global-memory tables)
constexpr int kCsqPoolFlag = 12;      // 1: the tables of the queued reads need more slots than the pool holds (the host grows it and runs that pass again)
constexpr int kCsqGlobalCount = 13;   // reads given to cs_global_kernel
constexpr int kCsqPoolNeed = 14;      // [14..15] slots the tables need (64-bit)

// Deals the queue into the class lists.  One workgroup (the queue is a few thousand to a few hundred thousand entries; in-place
// compaction of the entries that stay needs no second buffer this way).  A read with h index hits goes to class 0 when h < b0, else to
// class 1 when h < b1, else to class 2 when h < b2, else it STAYS queued (b0 <= b1 <= b2; a bound equal to the one before it closes the
// class).  The host picks the bounds per round:
//   round 0: the classes' hit limits (a class that is not run this round takes nothing);
//   round 1: what a smaller class sent on goes to class 2 (the largest table) once more; what class 2 itself has seen stays queued --
//            the same kernel would fail the same way -- for the exact kernels.
// lists: three rows of `stride` entries.  status[1] becomes the number of entries that stay.
__global__ __launch_bounds__(1024) void cs_heavy_classify_kernel(uint32_t *__restrict__ status, uint32_t *q_read, uint32_t *q_hits, uint32_t *__restrict__ lists, uint32_t stride,
		uint32_t *__restrict__ ctr, unsigned long long b0, unsigned long long b1, unsigned long long b2, int round) {
	__shared__ uint32_t s_cnt[4];
	const int tid = threadIdx.x, lane = tid & 63;
	const uint32_t n = status[1];
	if (tid < 4) s_cnt[tid] = 0;
	__syncthreads();
	const unsigned long long below = (1ull << lane) - 1ull;
	for (uint32_t base = 0; base < n; base += 1024u) {
		const uint32_t i = base + (uint32_t) tid;
		const bool valid = i < n;
		const uint32_t r = valid ? q_read[i] : 0u, h = valid ? q_hits[i] : 0u;
		int cls = -1;
		if (valid) cls = h < b0 ? 0 : h < b1 ? 1 : h < b2 ? 2 : 3;
		uint32_t pos = 0;
#pragma unroll
		for (int c = 0; c < 4; ++c) {
			const unsigned long long mk = __ballot(cls == c);
			if (mk) {
				const int leader = (int) __builtin_ctzll(mk);
				uint32_t b = 0;
				if (lane == leader) b = atomicAdd(&s_cnt[c], (uint32_t) __popcll(mk));
				b = (uint32_t) __builtin_amdgcn_readlane((int) b, leader);
				if (cls == c) pos = b + (uint32_t) __popcll(mk & below);
			}
		}
		__syncthreads();   // (every entry of this chunk has been read: the entries that stay may now overwrite the front of the queue)
		if (cls >= 0 && cls < 3) lists[(size_t) cls * stride + pos] = r;
		else if (cls == 3) { q_read[pos] = r; q_hits[pos] = h; }
		__syncthreads();
	}
	if (tid == 0) {
		status[1] = s_cnt[3];
		ctr[kCsqWork] = 0; ctr[kCsqWork + 1] = 0; ctr[kCsqWork + 2] = 0;
		ctr[kCsqCount] = s_cnt[0]; ctr[kCsqCount + 1] = s_cnt[1]; ctr[kCsqCount + 2] = s_cnt[2];
		ctr[kCsqRun + (round ? 1 : 0)] = s_cnt[0] + s_cnt[1] + s_cnt[2];
	}
}

// Sizes the global-memory vote tables of the reads the exact LDS pass queued (status block kCsqStatusExact): per read a table of
// 2^l >= 2 * hits slots (l >= 4), offsets by a running sum in queue order.  When the pool holds them all the reads are released to
// cs_global_kernel (status[kCsqGlobalCount]); otherwise the flag is raised and none is (the host grows the pool and runs this again).
__global__ __launch_bounds__(1024) void cs_global_prepare_kernel(uint32_t *__restrict__ status, const uint32_t *__restrict__ q_hits, uint64_t *__restrict__ table_off, uint32_t *__restrict__ table_log2,
		unsigned long long pool_slots) {
	__shared__ unsigned long long s_wsum[16], s_carry;
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const uint32_t n = status[kCsqStatusExact + 1];
	if (tid == 0) s_carry = 0ull;
	__syncthreads();
	for (uint32_t base = 0; base < n; base += 1024u) {
		const uint32_t i = base + (uint32_t) tid;
		uint32_t l = 0;
		unsigned long long slots = 0ull;
		if (i < n) {
			const uint32_t h = q_hits[i];
			l = 4;
			while ((1ull << l) < 2ull * h) ++l;
			slots = 1ull << l;
		}
		// inclusive scan over the wave (64-bit: two 32-bit halves would need a carry), then over the waves
		unsigned long long incl = slots;
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) {
			const unsigned long long o = ((unsigned long long) (uint32_t) __shfl_up((int) (uint32_t) (incl >> 32), d) << 32) | (uint32_t) __shfl_up((int) (uint32_t) incl, d);
			if (lane >= d) incl += o;
		}
		if (lane == 63) s_wsum[wv] = incl;
		__syncthreads();
		unsigned long long before = s_carry;
		for (int w = 0; w < wv; ++w) before += s_wsum[w];
		if (i < n) { table_off[i] = before + incl - slots; table_log2[i] = l; }
		__syncthreads();
		if (tid == 1023) s_carry = before + incl;
		__syncthreads();
	}
	if (tid == 0) {
		const unsigned long long need = s_carry;
		const bool fits = need <= pool_slots;
		status[kCsqPoolFlag] = fits ? 0u : 1u;
		status[kCsqGlobalCount] = fits ? n : 0u;
		status[kCsqPoolNeed] = (uint32_t) need; status[kCsqPoolNeed + 1] = (uint32_t) (need >> 32);
	}
}

// candidate regions -> one dense array in read order (new_base = exclusive prefix sum of cand_count): a thread copies its own read's
// (few) candidates, a read with more than kCompactSmall of them is copied by a wave (round 5: one thread per read, 2.8 ms per
// 131 072 reads of the GRCh38-like genome -- candidates per read: median 1, 99th percentile 1 356, maximum 9 383)
// The compaction is enqueued behind the passes without the host having seen their status: when a pass ran out of candidate room, or the
// reads queued for the global-memory tables were not run (pool too small), bases and counts are not all valid -- the kernel then does
// nothing (the host repeats the batch, or that pass and the compaction).
constexpr uint32_t kCompactSmall = 8;
__global__ __launch_bounds__(256) void compact_candidates_kernel(int n_reads, const uint32_t *__restrict__ status, const uint32_t *__restrict__ old_base, const uint32_t *__restrict__ new_base,
		const uint32_t *__restrict__ cand_count, const uint32_t *__restrict__ loc_in, const uint32_t *__restrict__ sv_in,
		uint32_t *__restrict__ loc_out, uint32_t *__restrict__ sv_out) {
	__shared__ uint32_t s_big[256], s_nbig;
	if (status[kCsqStatusMain] | status[kCsqStatusExact] | status[8] | status[kCsqPoolFlag]) return;   // (uniform)
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	if (tid == 0) s_nbig = 0;
	__syncthreads();
	const int r = blockIdx.x * 256 + tid;
	if (r < n_reads) {
		const uint32_t n = cand_count[r];
		if (n > kCompactSmall) s_big[atomicAdd(&s_nbig, 1u)] = (uint32_t) tid;
		else if (n) {
			const uint32_t ob = old_base[r], nb = new_base[r];
			for (uint32_t j = 0; j < n; ++j) { loc_out[nb + j] = loc_in[ob + j]; sv_out[nb + j] = sv_in[ob + j]; }
		}
	}
	__syncthreads();
	const uint32_t nbig = s_nbig;
	for (uint32_t i = (uint32_t) wv; i < nbig; i += 4u) {
		const int rr = blockIdx.x * 256 + (int) s_big[i];
		const uint32_t ob = old_base[rr], nb = new_base[rr], n = cand_count[rr];
		for (uint32_t j = (uint32_t) lane; j < n; j += 64u) { loc_out[nb + j] = loc_in[ob + j]; sv_out[nb + j] = sv_in[ob + j]; }
	}
}

}  // namespace ngm
