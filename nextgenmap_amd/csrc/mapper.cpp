// mapper.cpp -- the device-resident mapping path above IAlignment (include/ngm_pipeline.h):
//   candidate search -> window gather -> BatchScore -> top-1 selection / MAPQ -> window gather -> BatchAlign.
// One ngm_mapper is what one NextGenMap CS thread owns (CS + ScoreBuffer + AlignmentBuffer + IAlignment,
// src/CS.cpp:455-461); everything between the read upload and the traceback download stays in HBM.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <condition_variable>
#include <mutex>
#include <numeric>
#include <thread>
#include <vector>

#include <string.h>
#include <ctype.h>
#include <sched.h>
#include <rocprim/rocprim.hpp>

#include "refindex.h"
#include "engine_internal.h"
#include "align_device.h"
#include "cigar_md.h"
#include "cigar_device.h"
#include "cs_device.h"
#include "cs_canon_device.h"
#include "cs_heavy_device.h"
#include "cs_order_bucket_device.h"
#include "cs_slam_device.h"
#define NGM_SAM_KERNELS
#include "sam_device.h"
#include "gather_device.h"
#include "pair_device.h"
#include "thread_pool.h"

#define MAP_HIP_TRY(expr)                                                                      \
	do {                                                                                       \
		hipError_t e_ = (expr);                                                                \
		if (e_ != hipSuccess) {                                                                \
			ngm::pipeline_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
			return -5;                                                                         \
		}                                                                                      \
	} while (0)

// ScoreBuffer's running insert-size sum / count (src/ScoreBuffer.h:90) when several mapper instances work on one input:
// batches carry their input-order number and the order-dependent part of the selection takes turns in that order, so
// every batch starts from the state the reference's single CS thread would have at its first pair.
struct ngm_pair_state {
	std::mutex mu;
	std::condition_variable cv;
	uint64_t next = 0;                 // sequence number of the batch whose turn it is
	long dist_count = 1, dist_sum = 0;
	uint64_t scores_so_far = 0;        // candidates of the pairs of the reference's current CS batch: where its score buffer would stand
	uint64_t reads_so_far = 0;         // reads of all batches so far (position inside the reference's CS batches)
};

struct ngm_mapper {
	const ngm_ref *ref = nullptr;
	ngm_pair_state *ps = nullptr;      // shared paired-end state (null: the mapper's own)
	uint64_t batch_seq = 0;            // ... and the input-order number of the next paired-end batch
	int fast_pairing = 0;              // Config "fast_pairing": top1SE for both mates instead of top1PE (src/ScoreBuffer.cpp:203-216)
	ngm_mapper_params prm{};
	ngm_hip_ctx *eng = nullptr;
	hipStream_t st = nullptr;
	int max_kfreq = 0;
	int cs_log2_slots = 14;   // large LDS vote table: 2^14 slots * 8 B = 128 KB (2^13 when the lists of very long reads need the room)
	int cs_log2_small = 10;   // fast path: small exact table ...
	uint32_t cs_plane_bits = 65536;
	uint32_t cs_plane_bits0 = 65536;  // ... before it was trimmed to the LDS granule
	int cs_log2_bits = 16;    // ... behind two bit planes of this many bits; both picked from the index density
	uint32_t cs_queued_exact = 0;
	int cs_fast_items = ngm::kCsFastItemsShort;
	size_t cs_region_cap = 0; // candidate slots of all output regions together (grows when a batch overflows)
	double cs_hexp = 4096;    // expected index hits per read
	int cs_waves = 3;         // waves per read of the fast path (cs_fast2_kernel; 1: cs_fast_kernel, NGM_HIP_CS_WAVES)
	bool cs_paired = false;   // the batch being searched holds pairs (bisulfite mapping: second mates are searched A>G)
	int cs_canon_wpe = 7;      // 72 VGPRs: with 64 the sweeps spill (scratch round trips inside the vote loop cost more than the tenth read per CU brings)
	int cs_canon_ch = 1;      // canonical path, shape 2: chunk loads issued after this vote step (NGM_HIP_CS_CANON_CH: 0, 1, 3)
	int cs_canon = 0;         // 0: fast path over one bucket per k-mer (cs_fast2_kernel); 1-3: over canonical pair buckets, cs_canon_kernel<3,4,2> / <3,6,2> / <4,8,4>
	long pair_dist_count = 1, pair_dist_sum = 0;  // ScoreBuffer.h:90
	uint64_t scores_so_far = 0, reads_so_far = 0;  // see ngm_pair_state
	int ref_cs_batch = 0;             // reads per CS batch of the reference (1 800 000 / average read length, CS.cpp:26, :542): its score buffer is flushed there
	int ref_score_buffer = 0;         // entries of the reference's score buffer (IAlignment::GetScoreBatchSize there); 0: pairs are never lost (ngm_mapper_set_reference_score_buffer)
	uint64_t lost_pairs = 0;          // ngm_mapper_lost_pairs
	// ngm_mapper_path_counters: reads searched, candidates, reads re-run by the exact LDS / exact global-memory search, reads whose
	// candidate order was replayed, of those beyond the LDS replay's limits (replayed by the exact global-memory kernel), left undetermined
	uint64_t st_heavy = 0, st_reads = 0, st_cands = 0, st_exact_lds = 0, st_exact_global = 0, st_order_reads = 0, st_order_big = 0, st_order_unknown = 0, st_order_table = 0;
	ngm::CsArgs last_cs{};                          // arguments of the last candidate search (for the order replay)
	hipStream_t st_hi = nullptr;                    // high-priority stream: the (small) order replay runs outside the stage lock
	hipEvent_t turn_ev[16] = {};                     // GpuStage: the events that end this instance's turns
	unsigned turn_next = 0;
	hipStream_t st_copy = nullptr;                  // the per-read arrays of a search travel to the host beside the score stage's kernels, not in front of them
	hipEvent_t ev_cs_done = nullptr, ev_cs_copied = nullptr;
	bool cs_copy_pending = false;
	ngm::DevBuf<uint32_t> d_order_list, d_cand_rank, d_order_scratch, d_order_info, d_order_big, d_order_gt, d_order_log2;
	ngm::DevBuf<uint64_t> d_order_off;
	ngm::CsArgs order_args{};                       // arguments of the replay in flight
	ngm::PinnedBuf<uint32_t> p_rank, p_order_info;
	std::vector<uint32_t> order_pending;            // the reads of the replay in flight (candidate_order_finish accounts for them)
	// pinned staging for the per-batch downloads
	ngm::PinnedBuf<uint32_t> p_winner, p_loc, p_sv;
	ngm::PinnedBuf<int32_t> p_mapq, p_nbest, p_rec;
	ngm::PinnedBuf<float> p_best, p_scores;
	ngm::PinnedBuf<uint16_t> p_runs;
	// batch state in HBM
	ngm::DevBuf<uint8_t> d_reads;
	ngm::DevBuf<uint16_t> d_read_len;
	ngm::DevBuf<uint32_t> d_cand_base, d_cand_count, d_out_loc, d_out_sv, d_status, d_ovf_read, d_ovf_read2, d_ovf_hits, d_ovf_log2;
	ngm::DevBuf<uint64_t> d_ovf_off;
	ngm::DevBuf<uint32_t> d_gt_keys, d_gt_votes, d_heavy_list, d_heavy_ctr;
	ngm::DevBuf<float> d_max_votes, d_max_both, d_scores, d_best;
	ngm::DevBuf<unsigned long long> d_total, d_counters, d_heavy_diag;
	ngm::DevBuf<uint32_t> d_out_loc2, d_out_sv2, d_new_base;
	ngm::DevBuf<uint8_t> d_scan_tmp;
	unsigned long long cs_kmers = 0, cs_hits = 0;
	float cs_kernel_ms = 0.f;
	hipEvent_t cev[6] = {};
	hipEvent_t oev[4] = {};            // around the order replay's launches (its own stream)
	float order_ms = 0.f;              // GPU time of the order replays of the last batch (ngm_mapper_last_order_replay_ms)
	ngm::DevBuf<uint32_t> d_pair_read, d_winner, d_a_read, d_a_loc, d_a_sv;
	ngm::DevBuf<int32_t> d_mapq, d_nbest, d_records, d_pair_info;
	ngm::PinnedBuf<int32_t> p_pair_info;
	ngm::DevBuf<ngm::PairOut> d_pair_out;      // pair_choice_kernel (pair_device.h): per pair, and the best-scoring combinations of the tied ones
	ngm::DevBuf<ngm::PairTop> d_pair_top;
	ngm::DevBuf<uint32_t> d_pair_tied_n, d_pair_list;
	ngm::PinnedBuf<ngm::PairOut> p_pair_out;
	ngm::PinnedBuf<ngm::PairTop> p_pair_top;
	ngm::PinnedBuf<uint32_t> p_pair_tied_n;
	ngm::DevBuf<uint16_t> d_runs, d_runs_c;
	ngm::DevBuf<char> d_str;   // CIGAR / MD on the device: the compact byte stream
	ngm::DevBuf<ngm::CigarDevOut> d_cigout;
	ngm::PinnedBuf<ngm::CigarDevOut> p_cigout;
	ngm::PinnedBuf<char> p_str;
	// SAM text on the GPU (sam_device.h)
	ngm_sam_options sam_opt{};
	ngm_bgzf *bz = nullptr;   // sam_opt.bam: the BGZF compressor of this mapper's BAM records
	bool sam_ready = false;
	std::string sam_rg;
	ngm::DevBuf<char> d_sam_contig_names, d_sam_rg, d_sam_names, d_sam_text;
	ngm::DevBuf<uint32_t> d_sam_contig_off, d_sam_len, d_sam_off;
	ngm::DevBuf<uint64_t> d_sam_contig_start;
	ngm::DevBuf<uint8_t> d_sam_quals;
	ngm::DevBuf<ngm::SamMeta> d_sam_meta;
	ngm::DevBuf<ngm::SamRef> d_sam_refs;
	ngm::DevBuf<ngm_hit> d_sam_hits;
	ngm::PinnedBuf<ngm_hit> p_sam_hits;
	ngm::PinnedBuf<ngm::SamRef> p_sam_refs;
	ngm::PinnedBuf<char> p_sam_extra;
	uint64_t sam_text_bytes = 0;      // of the last batch (still in d_sam_text)
	uint64_t pair_stats[3] = {0, 0, 0};   // ngm_mapper_last_pair_stats
	// last CS result on the host
	int n_reads = 0;
	// per-read candidate offsets / counts / best vote counts of the last search, downloaded into pinned memory
	template <typename T> struct HostArr {
		ngm::PinnedBuf<T> b;
		T &operator[](size_t i) { return b.p[i]; }
		const T &operator[](size_t i) const { return b.p[i]; }
		T *data() { return b.p; }
	};
	HostArr<uint32_t> h_base, h_count;
	HostArr<float> h_maxv;
	uint64_t n_cand = 0;
	hipEvent_t ev[10] = {};   // [8]: behind the last kernel of the align stage
	float ms[8] = {};
};

namespace {

struct DevGuard {
	int prev = -1;
	explicit DevGuard(int d) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; if (prev != d) (void) hipSetDevice(d); else prev = -1; }
	~DevGuard() { if (prev >= 0) (void) hipSetDevice(prev); }
};

// canonical fast path: waves per read and chunk-item rounds of the kernel shapes (cs_canon_device.h)
constexpr int kCanonT[4] = {0, 3, 3, 4}, kCanonR1[4] = {0, 4, 6, 8}, kCanonR2[4] = {0, 2, 2, 4};
constexpr int kCsCanonMode = 3;
size_t cs_canon_lds_bytes(const ngm::CsArgs &A, int shape) {  // k-mer info + headers, codes, chunk items (16-bit), plane, table, queue (+ the kernel's static variables)
	const size_t w = (size_t) A.lists_cap + (A.q + 3) / 4 + (size_t) kCanonR2[shape] * kCanonT[shape] * 64 / 2 + ((size_t) A.plane_bits >> 5) + ((size_t) 2 << A.log2_slots) +
			((size_t) 3 << A.log2_slots) / 4 + 96;  // (+ slack: the per-wave k-mer rows round up, the static variables)
	return w * 4;
}

// shape 1-3; shape 2 exists in variants (experiments: NGM_HIP_CS_CANON_CH = the vote step after which the chunk loads are issued,
// NGM_HIP_CS_CANON_WPE = waves per SIMD the register allocation aims at)
const void *cs_canon_fn(int shape, int ch, int wpe, int bin_shift = 0) {
	if (shape == 2 && ch == 1 && wpe == 7 && bin_shift == 2) return (const void *) ngm::cs_canon_kernel<3, 6, 2, 1, 7, true>;   // the default
	if (shape == 1) return (const void *) ngm::cs_canon_kernel<3, 4, 2, 1>;
	if (shape == 3) return (const void *) ngm::cs_canon_kernel<4, 8, 4, 1>;
	if (wpe <= 5) return (const void *) ngm::cs_canon_kernel<3, 6, 2, 1, 5>;   // experiments: 86 VGPRs, no scratch, 5 waves per SIMD
	if (wpe == 6) return (const void *) ngm::cs_canon_kernel<3, 6, 2, 1, 6>;   // 80 VGPRs, 5 spilled dwords
	if (wpe <= 7) return ch == 0 ? (const void *) ngm::cs_canon_kernel<3, 6, 2, 0, 7> : (const void *) ngm::cs_canon_kernel<3, 6, 2, 1, 7>;
	return ch == 0 ? (const void *) ngm::cs_canon_kernel<3, 6, 2, 0> : ch >= 3 ? (const void *) ngm::cs_canon_kernel<3, 6, 2, 3> : (const void *) ngm::cs_canon_kernel<3, 6, 2, 1>;
}

size_t cs_lds_bytes(const ngm::CsArgs &A, int mode) {
	size_t w = (size_t) A.lists_cap * 2 + 1 + (A.q + 3) / 4;
	if (mode == ngm::kCsFast)  // list starts (32-bit) + lengths (16-bit), codes, plane, items, table, queue
		w = (size_t) A.lists_cap + (size_t) A.lists_cap / 2 + (A.q + 3) / 4 + ((size_t) A.plane_bits >> 5) + (size_t) A.fast_items * 64 / (A.items16 ? 2 : 1) +
				((size_t) 3 << A.log2_slots) / 4 + 32;  // + the kernels' static variables (<= 128 bytes): this is what the occupancy math sees
	if (mode != ngm::kCsFast && A.bs) w += (size_t) A.q + 1 + ngm::kCsBsChunk / 2;  // l_vbase, l_vpos
	if (mode != ngm::kCsExactGlobal) w += (size_t) 2 << A.log2_slots;
	return w * 4;
}

// ---- whose kernels run now ------------------------------------------------------------------------------------------------
// GPU stages (candidate search + score, align, SAM text: each from its first launch to the end of its last kernel) of the mapper
// instances of one process take turns: kernels of different instances then do not slow each other down, while the host stages of one
// instance -- and, since round 4, the downloads behind a stage's last kernel -- overlap the GPU stages of the others.
// NGM_HIP_GPU_STAGE_LOCK: 0 no turns (streams share the GPU), 1 one lock per device (default), 2 one lock per stage kind (search + score
// | align + SAM text).
// NGM_HIP_STAGE_CHAIN=1 (experiment, round 4): the turn passed ON THE GPU -- the host mutex held only while kernels are enqueued, the stream
// made to wait for the event behind the previous holder's kernels (hipStreamWaitEvent), every host synchronisation inside a stage ending
// the turn so that another instance's kernels fill the gap.  Measured on one box, 20 steps, twice each: 51.6 / 47.3 M reads/s chained
// against 54.8 / 51.3 with the host lock (three and four instances chained: 47.0 / 49.2): the cross-stream waits and the longer way of
// a batch through finer turns cost more than the idle time they remove.  Not the default.
struct StageChain { std::mutex mu; hipEvent_t last = nullptr; };
StageChain g_chain[16][2];
std::atomic<long long> g_stage_hold_us[3], g_stage_wait_us[3];   // diagnostics (NGM_HIP_HOST_TIMING): turn held / waited for on the host, per stage kind (0 search + score, 1 align, 2 SAM text)
struct GpuStage {
	ngm_mapper *m;
	int kind, slot;
	bool held = false;
	StageChain *ch = nullptr;
	std::chrono::steady_clock::time_point t_acq;
	static int mode() { static const int v = getenv("NGM_HIP_GPU_STAGE_LOCK") ? atoi(getenv("NGM_HIP_GPU_STAGE_LOCK")) : 1; return v; }
	static bool chained() { static const bool v = getenv("NGM_HIP_STAGE_CHAIN") && atoi(getenv("NGM_HIP_STAGE_CHAIN")) != 0; return v; }
	GpuStage(ngm_mapper *m_, int kind_ = 0, bool now = true, int slot_ = -1) : m(m_), kind(kind_), slot(slot_ < 0 ? kind_ : slot_) {
		ch = &g_chain[(unsigned) m->ref->device & 15u][mode() == 2 ? kind : 0];
		if (now) acquire();
	}
	~GpuStage() { release(); }
	void acquire() {   // before kernels are enqueued
		if (mode() == 0 || held) return;
		const auto t0 = std::chrono::steady_clock::now();
		ch->mu.lock();
		held = true;
		t_acq = std::chrono::steady_clock::now();
		g_stage_wait_us[slot] += std::chrono::duration_cast<std::chrono::microseconds>(t_acq - t0).count();
		if (chained() && ch->last) (void) hipStreamWaitEvent(m->st, ch->last, 0);
	}
	void release() {   // the kernels of this turn have been enqueued (chained) / have finished (host lock)
		if (!held) return;
		if (chained()) {
			hipEvent_t e = m->turn_ev[m->turn_next++ & 15u];
			if (e && hipEventRecord(e, m->st) == hipSuccess) ch->last = e;
		}
		g_stage_hold_us[slot] += std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t_acq).count();
		held = false;
		ch->mu.unlock();
	}
	// the stage's last kernel has been enqueued and `ev` recorded behind it; copies to the host follow: chained, the turn ends here
	void kernels_done() { if (chained()) release(); }
	// ... host lock: it ends when that kernel has finished, not when the copies have (the next instance's kernels run under them)
	void done_after(hipEvent_t ev) { if (held && !chained()) (void) hipEventSynchronize(ev); release(); }
	void done() { release(); }
	// a host synchronisation inside a stage: chained, the turn is passed on first (and taken again by the next acquire)
	void before_sync() { if (chained()) release(); }
};

// candidate search for the reads already in m->d_reads; leaves per-read base/count/max votes and the
// candidate arrays in HBM (and base/count/max votes on the host)
int run_cs(ngm_mapper *m, int n, GpuStage *stage = nullptr) {
	const ngm_ref *r = m->ref;
	auto hold = [&] { if (stage) stage->acquire(); };       // before kernels are enqueued
	auto yield = [&] { if (stage) stage->before_sync(); };   // before the host waits for the stream
	const int q = m->prm.qry_max_len;
	if (n <= 0) { m->n_reads = 0; m->n_cand = 0; return 0; }
	if (m->d_read_len.reserve(n) || m->d_cand_base.reserve(n) || m->d_cand_count.reserve(n) || m->d_max_votes.reserve(n) || m->d_max_both.reserve(n) ||
			m->d_status.reserve(4) || m->d_total.reserve(ngm::kCsRegions * ngm::kCsCursorStride + 16) || m->d_counters.reserve(ngm::kCsRegions * ngm::kCsCursorStride + 32) || m->d_new_base.reserve(n) || m->d_ovf_read.reserve(n) || m->d_ovf_read2.reserve(n) ||
			m->d_ovf_hits.reserve(n)) {
		ngm::pipeline_set_error("out of device memory (candidate search, %d reads)", n);
		return -12;
	}
	const size_t ctr_words = (size_t) ngm::kCsRegions * ngm::kCsCursorStride;
	size_t cap = std::max<size_t>(m->cs_region_cap, (size_t) n * 4 + 64 * ngm::kCsRegions);
	cap = (cap + ngm::kCsRegions - 1) / ngm::kCsRegions * ngm::kCsRegions;
	const size_t fixed_slots = getenv("NGM_HIP_CS_NO_FIXED_SLOTS") ? 0 : (size_t) n * ngm::kCsFixedSlots;
	for (int attempt = 0; attempt < 8; ++attempt) {
		// candidate offsets are 32-bit (base = region * capacity + cursor; the prefix sums over the counts)
		if (cap + fixed_slots >= 0xFFFFFFFFull) { ngm::pipeline_set_error("more than 2^32 candidate slots needed for %d reads: use smaller batches or a higher sensitivity", n); return -75; }
		if (m->d_out_loc.reserve(cap + fixed_slots) || m->d_out_sv.reserve(cap + fixed_slots) || m->d_out_loc2.reserve(cap + fixed_slots) || m->d_out_sv2.reserve(cap + fixed_slots)) { ngm::pipeline_set_error("out of device memory (candidates)"); return -12; }
		hold();
		MAP_HIP_TRY(hipMemsetAsync(m->d_status.p, 0, 16, m->st));
		MAP_HIP_TRY(hipMemsetAsync(m->d_total.p, 0, (ctr_words + 16) * 8, m->st));
		MAP_HIP_TRY(hipMemsetAsync(m->d_counters.p, 0, (ctr_words + 32) * 8, m->st));
		ngm::CsArgs A{};
		A.reads = m->d_reads.p; A.n = n; A.q = q; A.k = r->prm.kmer; A.bin_shift = r->prm.bin_size;
		A.max_kfreq = m->max_kfreq; A.sensitivity = m->prm.sensitivity; A.kmer_min = m->prm.kmer_min; A.max_cmrs = m->prm.max_cmrs;
		A.index = r->d_index; A.positions = r->d_positions;
		A.lists_cap = 2 * std::max(1, q - r->prm.kmer + 1);
		A.read_len = m->d_read_len.p; A.cand_base = m->d_cand_base.p; A.cand_count = m->d_cand_count.p; A.max_votes = m->d_max_votes.p; A.max_both = m->d_max_both.p;
		A.out_loc = m->d_out_loc.p; A.out_sv = m->d_out_sv.p; A.out_total = m->d_total.p; A.out_capacity = cap / ngm::kCsRegions;
		A.fixed_base = fixed_slots ? (uint32_t) cap : 0u;
		A.status = m->d_status.p; A.ovf_read = m->d_ovf_read.p; A.ovf_hits = m->d_ovf_hits.p; A.counters = m->d_counters.p;
		A.phase_cycles = getenv("NGM_HIP_CS_PHASES") ? m->d_counters.p + ctr_words : nullptr;
		A.debug_stop = getenv("NGM_HIP_CS_STOP") ? atoi(getenv("NGM_HIP_CS_STOP")) : 0;
		uint32_t status[4];
		m->cs_kernel_ms = 0;
		float pass_ms[3] = {0, 0, 0};
		auto timed = [&](int e) { float t = 0; if (hipEventElapsedTime(&t, m->cev[e], m->cev[e + 1]) == hipSuccess) { m->cs_kernel_ms += t; pass_ms[e / 2] = t; } };

		const bool bs = m->prm.bs_mapping != 0;
		A.bs = bs ? 1 : 0; A.bs_cutoff = m->prm.bs_cutoff; A.bs_read_skip = std::max(0, m->prm.bs_read_skip); A.bs_paired = m->cs_paired ? 1 : 0;
		if (bs) A.lists_cap = 2 * ngm::kCsBsChunk;  // the exact kernels hold the lists of kCsBsChunk k-mer variants at a time
		// pass 1 -- FAST path for every read (bit-plane filter + small exact table, many workgroups per CU)
		A.log2_bits = m->cs_log2_bits; A.plane_bits = m->cs_plane_bits; A.log2_slots = m->cs_log2_small; A.fast_items = m->cs_fast_items;
		A.buckets = r->d_buckets; A.bucket_log2_words = r->bucket_log2_words; A.pos_base = r->bucket_pos_base;
		A.hit_cap = m->cs_plane_bits / 6u;
		if (A.bin_shift < 2) A.hit_cap = 0;  // the register encoding of the fast path keeps bins in 30 bits
		MAP_HIP_TRY(hipEventRecord(m->cev[0], m->st));
		const bool slamw = (m->prm.slam_seq & 4) != 0;
		if (slamw) {
			// `--slam-seq` with bit 2: the weighted search (cs_slam_device.h) -- float votes in the reference's order, one wave per read,
			// tables in slices of global memory.  Persistent workgroups with a slice each; reads whose hits outgrow it are queued and re-run
			// with a slice of their own.
			A.bs = 2; A.bs_cutoff = 0; A.bs_read_skip = 0; A.bs_paired = m->cs_paired ? 1 : 0;
			A.lists_cap = 2 * ngm::kCsBsChunk;
			const size_t lds = cs_lds_bytes(A, ngm::kCsExactGlobal);
			const int grid = std::min(n, 2048);
			A.slam_slice_words = ngm::cs_slam_words(49152u);
			if (m->d_gt_keys.reserve((size_t) grid * A.slam_slice_words)) { ngm::pipeline_set_error("out of device memory (weighted SLAM-seq search)"); return -12; }
			A.gtable_keys = m->d_gt_keys.p;
			hipLaunchKernelGGL(ngm::cs_slam_kernel, dim3(grid), dim3(64), lds, m->st, A);
			MAP_HIP_TRY(hipGetLastError());
			yield();
			MAP_HIP_TRY(hipMemcpyAsync(status, m->d_status.p, 16, hipMemcpyDeviceToHost, m->st));
			MAP_HIP_TRY(hipStreamSynchronize(m->st));
			if (status[1] > 0) {
				const uint32_t no = status[1];
				std::vector<uint32_t> qr(no), qh(no), lg(no);
				std::vector<uint64_t> off(no);
				MAP_HIP_TRY(hipMemcpy(qr.data(), m->d_ovf_read.p, (size_t) no * 4, hipMemcpyDeviceToHost));
				MAP_HIP_TRY(hipMemcpy(qh.data(), m->d_ovf_hits.p, (size_t) no * 4, hipMemcpyDeviceToHost));
				if (m->d_ovf_off.reserve(no) || m->d_ovf_log2.reserve(no) || m->d_ovf_read2.reserve(no)) { ngm::pipeline_set_error("out of device memory (weighted SLAM-seq search)"); return -12; }
				constexpr uint64_t kPoolWords = 1ull << 30;
				for (uint32_t j0 = 0; j0 < no;) {
					uint64_t total = 0;
					uint32_t j1 = j0;
					while (j1 < no && (j1 == j0 || total + ngm::cs_slam_words(qh[j1]) <= kPoolWords)) { off[j1] = total; lg[j1] = ngm::cs_slam_log2_slots(qh[j1]); total += ngm::cs_slam_words(qh[j1]); ++j1; }
					if (m->d_gt_keys.reserve(total)) { ngm::pipeline_set_error("out of device memory (weighted SLAM-seq search, %llu words)", (unsigned long long) total); return -12; }
					hold();
					MAP_HIP_TRY(hipMemcpyAsync(m->d_ovf_read2.p, qr.data() + j0, (size_t) (j1 - j0) * 4, hipMemcpyHostToDevice, m->st));
					MAP_HIP_TRY(hipMemcpyAsync(m->d_ovf_log2.p, lg.data() + j0, (size_t) (j1 - j0) * 4, hipMemcpyHostToDevice, m->st));
					MAP_HIP_TRY(hipMemcpyAsync(m->d_ovf_off.p, off.data() + j0, (size_t) (j1 - j0) * 8, hipMemcpyHostToDevice, m->st));
					ngm::CsArgs Q = A;
					Q.read_list = m->d_ovf_read2.p; Q.ovf_log2 = m->d_ovf_log2.p; Q.ovf_table_off = m->d_ovf_off.p; Q.gtable_keys = m->d_gt_keys.p;
					hipLaunchKernelGGL(ngm::cs_slam_kernel, dim3(j1 - j0), dim3(64), lds, m->st, Q);
					MAP_HIP_TRY(hipGetLastError());
					yield();
					MAP_HIP_TRY(hipStreamSynchronize(m->st));
					j0 = j1;
				}
				MAP_HIP_TRY(hipMemcpy(status, m->d_status.p, 16, hipMemcpyDeviceToHost));
			}
			MAP_HIP_TRY(hipEventRecord(m->cev[1], m->st));
			MAP_HIP_TRY(hipStreamSynchronize(m->st));
			timed(0);
			status[1] = 0;
		} else if (bs) {
			// bisulfite mapping: no fast path (a read looks up ~10 variants of every k-mer: exact tables only); every read starts in pass 2
			MAP_HIP_TRY(hipEventRecord(m->cev[1], m->st));
			status[0] = 0; status[1] = (uint32_t) n; status[2] = status[3] = 0;
		} else {
		// 16-bit work items when every list index fits 9 bits and no used list can have more than 128 segments
		A.items16 = (A.lists_cap <= 512 && m->max_kfreq <= 128 * ngm::kCsSeg) ? 1 : 0;
		if (!A.items16) A.fast_items = ngm::kCsFastItemsLong;  // the 32-bit item list only exists in the large size
		if (m->cs_canon) {
			A.buckets = r->d_cbuckets; A.bucket_log2_words = r->cbucket_log2_words; A.pos_base = r->cbucket_pos_base;
			const size_t lds = cs_canon_lds_bytes(A, m->cs_canon) - 96;  // (the kernel has no static LDS: its shared variables are the last 160 bytes of this)
			// persistent workgroups: as many as the GPU holds at once, each walking the reads with that stride
			const void *fn = cs_canon_fn(m->cs_canon, m->cs_canon_ch, m->cs_canon_wpe, A.bin_shift);
			int per_cu = 0, cus = 0;
			if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, kCanonT[m->cs_canon] * 64, lds) != hipSuccess || per_cu < 1) per_cu = 1;
			if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, r->device) != hipSuccess || cus < 1) cus = 256;
			int grid = std::min(n, per_cu * cus);
			if (const char *e = getenv("NGM_HIP_CS_GRID_PER_CU")) grid = std::min(n, std::max(1, atoi(e)) * cus);  // experiments
			// (experiment, NGM_HIP_CS_READS_PER_WG=16..128: 5.55 / 5.74 / 6.44 ms per 524 288 reads with runs of 32 / 64 / 128 reads against 5.41 with
			// persistent workgroups on the same box, and the other instance's order replay waits as long either way: not the default)
			static const int run_env = getenv("NGM_HIP_CS_READS_PER_WG") ? atoi(getenv("NGM_HIP_CS_READS_PER_WG")) : 0;
			A.reads_per_wg = std::max(0, run_env);
			if (A.reads_per_wg > 0) grid = (n + A.reads_per_wg - 1) / A.reads_per_wg;
			// NGM_HIP_CS_SPLIT=k: the batch in k launches -- persistent workgroups hold every CU until their launch ends, and a kernel of
			// another stream (the other mapper instance's order replay, on a high-priority stream) only gets in between launches
			static const int split = std::max(1, getenv("NGM_HIP_CS_SPLIT") ? atoi(getenv("NGM_HIP_CS_SPLIT")) : 1);
			const int n_all = A.n;
			for (int part = 0; part < split; ++part) {
				ngm::CsArgs P = A;
				P.read_lo = (int) ((long long) n_all * part / split);
				P.n = (int) ((long long) n_all * (part + 1) / split);
				if (P.n <= P.read_lo) continue;
				const int cnt = P.n - P.read_lo;
				const int g = A.reads_per_wg > 0 ? (cnt + A.reads_per_wg - 1) / A.reads_per_wg : std::min(cnt, grid);
				if (part > 0) (void) hipMemsetAsync(m->d_status.p + 2, 0, 4, m->st);
				void *kargs[] = {(void *) &P};
				(void) hipLaunchKernel(fn, dim3(g), dim3(kCanonT[m->cs_canon] * 64), kargs, lds, m->st);
			}
			if (A.phase_cycles)
				fprintf(stderr, "[ngm-hip] cs canonical path (shape %d): %zu bytes of LDS per read, %d reads resident per CU (grid %d), bucket 2^%d words\n", m->cs_canon, lds, per_cu, grid, A.bucket_log2_words);
		}
		else if (m->cs_waves >= 2 && A.items16) {  // T waves per read: the same 768 / 1 536 segments, dealt to T * 64 lanes
			const bool shrt = A.fast_items == ngm::kCsFastItemsShort;
			const size_t lds = cs_lds_bytes(A, ngm::kCsFast) - 128;  // the kernel's static variables take the rest
#define NGM_CS_LAUNCH_T(T) \
			do { if (shrt) hipLaunchKernelGGL((ngm::cs_fast2_kernel<T, ngm::kCsFastItemsShort / T, uint16_t>), dim3(n), dim3(T * 64), lds, m->st, A); \
				else hipLaunchKernelGGL((ngm::cs_fast2_kernel<T, ngm::kCsFastItemsLong / T, uint16_t>), dim3(n), dim3(T * 64), lds, m->st, A); } while (0)
			if (m->cs_waves == 2) NGM_CS_LAUNCH_T(2); else if (m->cs_waves == 3) NGM_CS_LAUNCH_T(3); else NGM_CS_LAUNCH_T(4);
			if (A.phase_cycles && m->cs_waves == 3 && shrt) {  // diagnostics: reads resident per CU
				int blocks = 0;
				(void) hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, (const void *) ngm::cs_fast2_kernel<3, ngm::kCsFastItemsShort / 3, uint16_t>, 192, lds);
				fprintf(stderr, "[ngm-hip] cs fast path: %zu bytes of LDS per read, %d reads resident per CU\n", lds, blocks);
			}
#undef NGM_CS_LAUNCH_T
		}
		else if (A.fast_items == ngm::kCsFastItemsShort && A.items16) hipLaunchKernelGGL((ngm::cs_fast_kernel<ngm::kCsFastItemsShort, uint16_t>), dim3(n), dim3(64), cs_lds_bytes(A, ngm::kCsFast), m->st, A);
		else if (A.items16) hipLaunchKernelGGL((ngm::cs_fast_kernel<ngm::kCsFastItemsLong, uint16_t>), dim3(n), dim3(64), cs_lds_bytes(A, ngm::kCsFast), m->st, A);
		else hipLaunchKernelGGL((ngm::cs_fast_kernel<ngm::kCsFastItemsLong, uint32_t>), dim3(n), dim3(64), cs_lds_bytes(A, ngm::kCsFast), m->st, A);
		MAP_HIP_TRY(hipGetLastError());
		MAP_HIP_TRY(hipEventRecord(m->cev[1], m->st));
		yield();
		MAP_HIP_TRY(hipMemcpyAsync(status, m->d_status.p, 16, hipMemcpyDeviceToHost, m->st));
		MAP_HIP_TRY(hipStreamSynchronize(m->st));
		timed(0);
		}
		uint32_t n_heavy = 0;
		static const bool heavy_on = !getenv("NGM_HIP_CS_NO_HEAVY");
		static const bool heavy_v1 = getenv("NGM_HIP_CS_HEAVY_V1") != nullptr;   // round 4's kernel (cs_heavy_kernel), for A/B runs
		if (!bs && heavy_on && !heavy_v1 && A.bin_shift >= 2 && status[1] > 0) {
			// pass 1b -- the reads with more hits than the fast path takes (cs_heavy2_kernel, cs_heavy_device.h): two rows of sketch counters +
			// an exact table in LDS, by hit count in three classes of persistent workgroups; pass 1c -- what the two smaller classes
			// cannot certify, once more in the largest; what is left after that is queued for the exact kernels below
			struct HeavyClass { uint32_t max_hits; int log2c, log2s, nt; uint32_t scratch_cap; const void *fn; uint32_t max_parts = 1; };   // max_parts: table passes a read may take (the largest class)
			// (class limits measured on the heavy-tailed probe, per 262 144 reads: 16 384 / 32 768 / rest 18.3 ms; 16 384 / 65 536 / rest 16.1; 16 384 / all the
			// rest in the middle class -- two workgroups per CU -- and the largest class only for what that cannot certify: 15.1)
			static HeavyClass classes[3] = {{16384u, 13, 11, 256, 16384u, (const void *) ngm::cs_heavy2_kernel<256>}, {0xFFFFFFFEu, 14, 12, 512, 262144u, (const void *) ngm::cs_heavy2_kernel<512>},
					{0xFFFFFFFFu, 15, 13, 1024, 1u << 20, (const void *) ngm::cs_heavy2_kernel<1024>, 32u}};
			static const bool parts_env = [] { if (const char *e = getenv("NGM_HIP_HEAVY_PARTS")) classes[2].max_parts = (uint32_t) std::max(1, std::min(256, atoi(e))); return true; }();   // experiments: table passes of the largest class
			(void) parts_env;
			static const bool classes_env = [] {   // experiments: NGM_HIP_HEAVY_CLASSES=max0,max1 (hits up to which a read starts in class 0 / class 1)
				if (const char *e = getenv("NGM_HIP_HEAVY_CLASSES")) {
					unsigned long a = 0, b = 0;
					if (sscanf(e, "%lu,%lu", &a, &b) == 2 && a > 0 && b >= a) {
						classes[0].max_hits = (uint32_t) std::min<unsigned long>(a, 0xFFFFFFFEul); classes[1].max_hits = (uint32_t) std::min<unsigned long>(b, 0xFFFFFFFEul);
						classes[0].scratch_cap = std::min<uint32_t>(classes[0].max_hits, 262144u); classes[1].scratch_cap = std::min<uint32_t>(classes[1].max_hits, 262144u);
					}
				}
				return true; }();
			(void) classes_env;
			const uint32_t coarse_cap = (uint32_t) ngm::cs_heavy2_coarse_cap(A.lists_cap, m->max_kfreq);
			n_heavy = status[1];
			float t_heavy[2] = {0, 0};
			uint32_t in_round[2] = {0, 0};
			int cus = 0;
			if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, r->device) != hipSuccess || cus < 1) cus = 256;
			if (m->d_heavy_ctr.reserve(8)) { ngm::pipeline_set_error("out of device memory (candidate search)"); return -12; }
			for (int round = 0; round < 2 && status[1] > 0; ++round) {
				const uint32_t no = status[1];
				std::vector<uint32_t> qr(no), qh(no), lists[3], keep_r, keep_h;
				MAP_HIP_TRY(hipMemcpyAsync(qr.data(), m->d_ovf_read.p, (size_t) no * 4, hipMemcpyDeviceToHost, m->st));
				MAP_HIP_TRY(hipMemcpyAsync(qh.data(), m->d_ovf_hits.p, (size_t) no * 4, hipMemcpyDeviceToHost, m->st));
				MAP_HIP_TRY(hipStreamSynchronize(m->st));
				for (uint32_t i = 0; i < no; ++i) {
					if (round == 0) lists[qh[i] <= classes[0].max_hits ? 0 : qh[i] <= classes[1].max_hits ? 1 : 2].push_back(qr[i]);
					else if (qh[i] <= classes[1].max_hits) lists[2].push_back(qr[i]);   // failed in a smaller class: once more with the largest table
					else { keep_r.push_back(qr[i]); keep_h.push_back(qh[i]); }          // (the largest class has seen it: the same kernel would fail the same way)
				}
				const uint32_t n_run = (uint32_t) (lists[0].size() + lists[1].size() + lists[2].size());
				in_round[round] = n_run;
				if (n_run == 0) break;
				if (m->d_heavy_list.reserve(no)) { ngm::pipeline_set_error("out of device memory (candidate search)"); return -12; }
				int grid[3] = {0, 0, 0};
				size_t lds[3] = {0, 0, 0}, scratch_words = 0;
				auto ent_cap_of = [&](int c) -> uint32_t { return classes[c].max_parts > 1 ? classes[c].max_parts * ((3u << classes[c].log2s) / 4u) : 0u; };   // (bin, votes) entries of all table passes
				for (int c = 0; c < 3; ++c) {
					if (lists[c].empty()) continue;
					lds[c] = ngm::cs_heavy2_lds_bytes(A.lists_cap, A.q, classes[c].log2c, classes[c].log2s, coarse_cap);
					int per_cu = 0;
					if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, classes[c].fn, classes[c].nt, lds[c]) != hipSuccess || per_cu < 1) per_cu = 1;
					grid[c] = (int) std::min<size_t>(lists[c].size(), (size_t) per_cu * cus);
					scratch_words += (size_t) grid[c] * ((size_t) classes[c].scratch_cap + 2 * (size_t) ent_cap_of(c));
				}
				if (m->d_gt_keys.reserve(scratch_words)) { ngm::pipeline_set_error("out of device memory (candidate search scratch, %zu words)", scratch_words); return -12; }
				hold();
				// the queue restarts with the reads this round does not run again
				const uint32_t n_keep = (uint32_t) keep_r.size();
				MAP_HIP_TRY(hipMemcpyAsync(m->d_status.p + 1, &n_keep, 4, hipMemcpyHostToDevice, m->st));
				if (n_keep) {
					MAP_HIP_TRY(hipMemcpyAsync(m->d_ovf_read.p, keep_r.data(), (size_t) n_keep * 4, hipMemcpyHostToDevice, m->st));
					MAP_HIP_TRY(hipMemcpyAsync(m->d_ovf_hits.p, keep_h.data(), (size_t) n_keep * 4, hipMemcpyHostToDevice, m->st));
				}
				MAP_HIP_TRY(hipMemsetAsync(m->d_heavy_ctr.p, 0, 32, m->st));
				if (A.phase_cycles) { if (m->d_heavy_diag.reserve(64)) return -12; MAP_HIP_TRY(hipMemsetAsync(m->d_heavy_diag.p, 0, 64 * 8, m->st)); }
				MAP_HIP_TRY(hipEventRecord(m->cev[2], m->st));
				uint32_t off = 0;
				size_t soff = 0;
				for (int c = 2; c >= 0; --c) {   // (the largest reads first: their workgroups are the long ones)
					const uint32_t cnt = (uint32_t) lists[c].size();
					if (cnt == 0) continue;
					MAP_HIP_TRY(hipMemcpyAsync(m->d_heavy_list.p + off, lists[c].data(), (size_t) cnt * 4, hipMemcpyHostToDevice, m->st));
					ngm::CsArgs Hv = A;
					Hv.read_list = m->d_heavy_list.p + off; Hv.log2_bits = classes[c].log2c; Hv.log2_slots = classes[c].log2s;
					uint32_t n_list = cnt, scap = classes[c].scratch_cap, ccap = coarse_cap, mparts = classes[c].max_parts, ecap = ent_cap_of(c);
					uint32_t *ctr = m->d_heavy_ctr.p + c, *scr = m->d_gt_keys.p + soff;
					unsigned long long *dg = A.phase_cycles ? m->d_heavy_diag.p + 16 * c : nullptr;
					void *kargs[] = {(void *) &Hv, (void *) &n_list, (void *) &ctr, (void *) &scr, (void *) &scap, (void *) &ccap, (void *) &mparts, (void *) &ecap, (void *) &dg};
					MAP_HIP_TRY(hipLaunchKernel(classes[c].fn, dim3(grid[c]), dim3(classes[c].nt), kargs, lds[c], m->st));
					off += cnt;
					soff += (size_t) grid[c] * ((size_t) classes[c].scratch_cap + 2 * (size_t) ecap);
				}
				MAP_HIP_TRY(hipEventRecord(m->cev[3], m->st));
				yield();
				MAP_HIP_TRY(hipMemcpyAsync(status, m->d_status.p, 16, hipMemcpyDeviceToHost, m->st));
				MAP_HIP_TRY(hipStreamSynchronize(m->st));   // (the lists live until here)
				if (hipEventElapsedTime(&t_heavy[round], m->cev[2], m->cev[3]) == hipSuccess) m->cs_kernel_ms += t_heavy[round];
				if (A.phase_cycles) {
					unsigned long long dg[64];
					MAP_HIP_TRY(hipMemcpy(dg, m->d_heavy_diag.p, sizeof(dg), hipMemcpyDeviceToHost));
					for (int c = 0; c < 3; ++c) if (dg[16 * c + 8]) {
						const double ns = (double) dg[16 * c + 8];
						fprintf(stderr, "[ngm-hip] heavy class %d (round %d, %zu reads, grid %d): us per sampled read: setup %.1f | sweep A %.1f | sum + T %.1f | insert / sweep B %.1f | row 2 %.1f | sweep D %.1f | candidates %.1f; hits %.0f, survivors %.0f, %.0f %% without a second row; second passes %.0f %% of the reads, table passes of the partitioned reads %.1f\n",
								c, round, lists[c].size(), grid[c], dg[16 * c] / ns / 100.0, dg[16 * c + 1] / ns / 100.0, dg[16 * c + 2] / ns / 100.0, dg[16 * c + 3] / ns / 100.0, dg[16 * c + 4] / ns / 100.0,
								dg[16 * c + 5] / ns / 100.0, dg[16 * c + 6] / ns / 100.0, dg[16 * c + 9] / ns, dg[16 * c + 11] / ns, 100.0 * dg[16 * c + 10] / ns, 100.0 * dg[16 * c + 13] / ns, (double) dg[16 * c + 12]);
						const unsigned long long w = dg[16 * c + 14], x = dg[16 * c + 15];
						if (w | x) fprintf(stderr, "[ngm-hip] heavy class %d sent on: %llu reads with a wrapped counter row, %llu without a T <= 255 that fits, %llu with more survivors than the slice or an overflowing table / entry list, %llu with T - 1 not below the threshold\n",
								c, w & 0xFFFFFFFFull, w >> 32, x & 0xFFFFFFFFull, x >> 32);
					}
				}
			}
			if (getenv("NGM_HIP_HOST_TIMING")) fprintf(stderr, "[ngm-hip] candidate search pass 1b (heavy reads): %.2f ms for %u reads; pass 1c (the largest class once more): %.2f ms for %u reads; %u left for the exact kernels\n",
					t_heavy[0], in_round[0], t_heavy[1], in_round[1], status[1]);
		} else
		if (!bs && heavy_on && status[1] > 0) {
			// pass 1b -- the reads with more hits than the fast path takes (cs_heavy_device.h): sketch counters + exact table in LDS, by
			// hit count in three classes of workgroups; pass 1c -- what those cannot certify (more near-threshold bins than their table
			// holds) and the reads beyond 65 535 hits: the same with 32-bit counters and 8 192 slots; what is left after that is queued
			// for the exact kernels below
			struct HeavyClass { uint32_t max_hits; int log2c, log2s, nt; bool wide; const void *fn; };
			static const HeavyClass classes[4] = {{16384u, 13, 11, 256, false, (const void *) ngm::cs_heavy_kernel<256>}, {32768u, 14, 12, 512, false, (const void *) ngm::cs_heavy_kernel<512>},
					{ngm::kCsHeavyMaxHits16, 15, 12, 1024, false, (const void *) ngm::cs_heavy_kernel<1024>}, {0xFFFFFFFFu, 14, 13, 1024, true, (const void *) ngm::cs_heavy_kernel<1024, true>}};
			n_heavy = status[1];
			float t_heavy[2] = {0, 0};
			uint32_t in_round[2] = {0, 0};
			for (int round = 0; round < 2 && status[1] > 0; ++round) {
				const uint32_t no = status[1];
				in_round[round] = no;
				std::vector<uint32_t> qr(no), qh(no), lists[4];
				MAP_HIP_TRY(hipMemcpyAsync(qr.data(), m->d_ovf_read.p, (size_t) no * 4, hipMemcpyDeviceToHost, m->st));
				MAP_HIP_TRY(hipMemcpyAsync(qh.data(), m->d_ovf_hits.p, (size_t) no * 4, hipMemcpyDeviceToHost, m->st));
				MAP_HIP_TRY(hipStreamSynchronize(m->st));
				for (uint32_t i = 0; i < no; ++i) lists[round == 1 ? 3 : qh[i] <= classes[0].max_hits ? 0 : qh[i] <= classes[1].max_hits ? 1 : qh[i] <= classes[2].max_hits ? 2 : 3].push_back(qr[i]);
				if (m->d_heavy_list.reserve(no)) { ngm::pipeline_set_error("out of device memory (candidate search)"); return -12; }
				hold();
				MAP_HIP_TRY(hipMemsetAsync(m->d_status.p + 1, 0, 4, m->st));
				MAP_HIP_TRY(hipEventRecord(m->cev[2], m->st));
				uint32_t off = 0;
				for (int c = 0; c < 4; ++c) {
					const uint32_t cnt = (uint32_t) lists[c].size();
					if (cnt == 0) continue;
					MAP_HIP_TRY(hipMemcpyAsync(m->d_heavy_list.p + off, lists[c].data(), (size_t) cnt * 4, hipMemcpyHostToDevice, m->st));
					ngm::CsArgs Hv = A;
					Hv.read_list = m->d_heavy_list.p + off; Hv.log2_bits = classes[c].log2c; Hv.log2_slots = classes[c].log2s;
					const size_t lds = ngm::cs_heavy_lds_bytes(Hv.lists_cap, Hv.q, Hv.log2_bits, Hv.log2_slots, classes[c].wide);
					void *kargs[] = {(void *) &Hv};
					MAP_HIP_TRY(hipLaunchKernel(classes[c].fn, dim3(cnt), dim3(classes[c].nt), kargs, lds, m->st));
					off += cnt;
				}
				MAP_HIP_TRY(hipEventRecord(m->cev[3], m->st));
				yield();
				MAP_HIP_TRY(hipMemcpyAsync(status, m->d_status.p, 16, hipMemcpyDeviceToHost, m->st));
				MAP_HIP_TRY(hipStreamSynchronize(m->st));   // (the lists live until here)
				if (hipEventElapsedTime(&t_heavy[round], m->cev[2], m->cev[3]) == hipSuccess) m->cs_kernel_ms += t_heavy[round];
			}
			if (getenv("NGM_HIP_HOST_TIMING")) fprintf(stderr, "[ngm-hip] candidate search pass 1b (heavy reads): %.2f ms for %u reads; pass 1c (32-bit counters, 8 192 slots): %.2f ms for %u reads; %u left for the exact kernels\n",
					t_heavy[0], in_round[0], t_heavy[1], in_round[1], status[1]);
		}
		m->cs_queued_exact = status[1];
		const uint32_t n_exact_lds = status[1];
		uint32_t n_exact_global = 0;
		auto dump_queue = [&](int pass, uint32_t cnt) {   // diagnostics (NGM_HIP_DUMP_OVF=file): index hits of the reads queued for `pass`
			const char *fn = getenv("NGM_HIP_DUMP_OVF");
			if (!fn || cnt == 0) return;
			std::vector<uint32_t> hh(cnt);
			if (hipMemcpy(hh.data(), m->d_ovf_hits.p, (size_t) cnt * 4, hipMemcpyDeviceToHost) != hipSuccess) return;
			if (FILE *f = fopen(fn, "a")) { for (uint32_t x : hh) fprintf(f, "%d %u\n", pass, x); fclose(f); }
		};
		dump_queue(2, status[1]);
		if (getenv("NGM_HIP_HOST_TIMING")) fprintf(stderr, "[ngm-hip] candidate search pass 1 (fast path): %.2f ms for %d reads, %u queued for the exact path\n", pass_ms[0], n, status[1]);
		if (status[1] > 0) {
			// pass 2 -- EXACT path, table in LDS, for the reads the fast path could not certify
			const uint32_t no = status[1];
			hold();
			if (!bs) MAP_HIP_TRY(hipMemcpyAsync(m->d_ovf_read2.p, m->d_ovf_read.p, (size_t) no * 4, hipMemcpyDeviceToDevice, m->st));
			MAP_HIP_TRY(hipMemsetAsync(m->d_status.p + 1, 0, 4, m->st));
			ngm::CsArgs B = A;
			B.log2_slots = bs ? std::min(m->cs_log2_slots, 13) : m->cs_log2_slots;
			B.hit_cap = (uint32_t) ((1u << B.log2_slots) * 0.66f);
			B.read_list = bs ? nullptr : m->d_ovf_read2.p;
			MAP_HIP_TRY(hipEventRecord(m->cev[2], m->st));
			hipLaunchKernelGGL(ngm::cs_kernel<ngm::kCsExactLds>, dim3(no), dim3(64), cs_lds_bytes(B, ngm::kCsExactLds), m->st, B);
			MAP_HIP_TRY(hipGetLastError());
			MAP_HIP_TRY(hipEventRecord(m->cev[3], m->st));
			yield();
			MAP_HIP_TRY(hipMemcpyAsync(status, m->d_status.p, 16, hipMemcpyDeviceToHost, m->st));
			MAP_HIP_TRY(hipStreamSynchronize(m->st));
			timed(2);
		}
		if (status[1] > 0) {
			// pass 3 -- EXACT path with per-read tables in global memory (reads with more hits than LDS holds)
			const uint32_t no = status[1];
			n_exact_global = no;
			dump_queue(3, no);
			if (getenv("NGM_HIP_HOST_TIMING")) fprintf(stderr, "[ngm-hip] candidate search pass 2 (exact, LDS table): %.2f ms for %u reads, %u queued for the global-memory tables\n", pass_ms[1], n_exact_lds, no);
			std::vector<uint32_t> hits(no), lg(no);
			std::vector<uint64_t> off(no);
			MAP_HIP_TRY(hipMemcpy(hits.data(), m->d_ovf_hits.p, no * 4, hipMemcpyDeviceToHost));
			uint64_t total_slots = 0;
			for (uint32_t i = 0; i < no; ++i) {
				uint32_t l = 4;
				while ((1ull << l) < 2ull * hits[i]) ++l;
				lg[i] = l;
				off[i] = total_slots;
				total_slots += 1ull << l;
			}
			if (m->d_gt_keys.reserve(total_slots) || m->d_gt_votes.reserve(total_slots) || m->d_ovf_off.reserve(no) || m->d_ovf_log2.reserve(no)) {
				ngm::pipeline_set_error("out of device memory (overflow vote tables, %llu slots)", (unsigned long long) total_slots);
				return -12;
			}
			hold();
			MAP_HIP_TRY(hipMemcpyAsync(m->d_ovf_off.p, off.data(), no * 8, hipMemcpyHostToDevice, m->st));
			MAP_HIP_TRY(hipMemcpyAsync(m->d_ovf_log2.p, lg.data(), no * 4, hipMemcpyHostToDevice, m->st));
			MAP_HIP_TRY(hipMemcpyAsync(m->d_ovf_read2.p, m->d_ovf_read.p, (size_t) no * 4, hipMemcpyDeviceToDevice, m->st));
			ngm::CsArgs G = A;
			G.read_list = m->d_ovf_read2.p;
			G.ovf_table_off = m->d_ovf_off.p; G.ovf_log2 = m->d_ovf_log2.p; G.gtable_keys = m->d_gt_keys.p; G.gtable_votes = m->d_gt_votes.p;
			MAP_HIP_TRY(hipEventRecord(m->cev[4], m->st));
			// one workgroup per read (cs_global_kernel, cs_heavy_device.h); bisulfite runs keep the one-wave kernel (their lists come in chunks of variants)
			static const int global_nt = getenv("NGM_HIP_CS_GLOBAL_THREADS") ? atoi(getenv("NGM_HIP_CS_GLOBAL_THREADS")) : 512;   // (64: the one-wave kernel of rounds 1-3)
			if (G.bs || global_nt <= 64) hipLaunchKernelGGL(ngm::cs_kernel<ngm::kCsExactGlobal>, dim3(no), dim3(64), cs_lds_bytes(G, ngm::kCsExactGlobal), m->st, G);
			else if (global_nt >= 1024) hipLaunchKernelGGL(ngm::cs_global_kernel<1024>, dim3(no), dim3(1024), cs_lds_bytes(G, ngm::kCsExactGlobal), m->st, G);
			else if (global_nt >= 512) hipLaunchKernelGGL(ngm::cs_global_kernel<512>, dim3(no), dim3(512), cs_lds_bytes(G, ngm::kCsExactGlobal), m->st, G);
			else hipLaunchKernelGGL(ngm::cs_global_kernel<256>, dim3(no), dim3(256), cs_lds_bytes(G, ngm::kCsExactGlobal), m->st, G);
			MAP_HIP_TRY(hipGetLastError());
			MAP_HIP_TRY(hipEventRecord(m->cev[5], m->st));
			yield();
			MAP_HIP_TRY(hipMemcpyAsync(status, m->d_status.p, 16, hipMemcpyDeviceToHost, m->st));
			MAP_HIP_TRY(hipStreamSynchronize(m->st));
			timed(4);
			if (getenv("NGM_HIP_HOST_TIMING")) fprintf(stderr, "[ngm-hip] candidate search pass 3 (exact, global-memory tables): %.2f ms for %u reads\n", pass_ms[2], no);
		}
		if (status[0] == 0) {
			// regions -> one dense candidate array in read order
			hold();
			size_t tmp_bytes = 0;
			(void) rocprim::exclusive_scan(nullptr, tmp_bytes, m->d_cand_count.p, m->d_new_base.p, 0u, (size_t) n, rocprim::plus<uint32_t>(), m->st);
			if (m->d_scan_tmp.reserve(tmp_bytes + 16)) { ngm::pipeline_set_error("out of device memory (scan)"); return -12; }
			MAP_HIP_TRY(rocprim::exclusive_scan(m->d_scan_tmp.p, tmp_bytes, m->d_cand_count.p, m->d_new_base.p, 0u, (size_t) n, rocprim::plus<uint32_t>(), m->st));
			hipLaunchKernelGGL(ngm::compact_candidates_kernel, dim3((n + 255) / 256), dim3(256), 0, m->st, n, m->d_cand_base.p, m->d_new_base.p, m->d_cand_count.p,
					m->d_out_loc.p, m->d_out_sv.p, m->d_out_loc2.p, m->d_out_sv2.p);
			MAP_HIP_TRY(hipGetLastError());
			std::swap(m->d_out_loc, m->d_out_loc2); std::swap(m->d_out_sv, m->d_out_sv2); std::swap(m->d_cand_base, m->d_new_base);
			yield();
			m->last_cs = A;
			m->cs_region_cap = cap;
			m->n_reads = n;
			std::vector<unsigned long long> ctr(ctr_words + 16);
			MAP_HIP_TRY(hipMemcpyAsync(ctr.data(), m->d_counters.p, ctr.size() * 8, hipMemcpyDeviceToHost, m->st));
			if (m->h_base.b.reserve(n) || m->h_count.b.reserve(n) || m->h_maxv.b.reserve(n)) { ngm::pipeline_set_error("out of pinned host memory"); return -12; }
			// the host needs the number of candidates now (it sizes the score stage); the per-read arrays (12 bytes per read) only after the
			// score stage (cs_host_arrays): an experiment lets them travel on a stream of their own under its kernels
			uint32_t last[2] = {0, 0};
			MAP_HIP_TRY(hipMemcpyAsync(&last[0], m->d_cand_base.p + (n - 1), 4, hipMemcpyDeviceToHost, m->st));
			MAP_HIP_TRY(hipMemcpyAsync(&last[1], m->d_cand_count.p + (n - 1), 4, hipMemcpyDeviceToHost, m->st));
			// (NGM_HIP_CS_COPY_SIDE_STREAM=1: measured on one box, three runs each -- 49.8 / 52.0 / 51.5 M reads/s with the side stream against
			// 55.0 / 52.1 / 54.8 without: the copies are blit kernels either way, and a second stream only adds their scheduling: not the default)
			static const bool side = getenv("NGM_HIP_CS_COPY_SIDE_STREAM") != nullptr;
			hipStream_t cst = (side && m->st_copy && m->ev_cs_done && m->ev_cs_copied) ? m->st_copy : m->st;
			if (cst != m->st) { MAP_HIP_TRY(hipEventRecord(m->ev_cs_done, m->st)); MAP_HIP_TRY(hipStreamWaitEvent(cst, m->ev_cs_done, 0)); }
			MAP_HIP_TRY(hipMemcpyAsync(m->h_base.data(), m->d_cand_base.p, (size_t) n * 4, hipMemcpyDeviceToHost, cst));
			MAP_HIP_TRY(hipMemcpyAsync(m->h_count.data(), m->d_cand_count.p, (size_t) n * 4, hipMemcpyDeviceToHost, cst));
			MAP_HIP_TRY(hipMemcpyAsync(m->h_maxv.data(), m->d_max_votes.p, (size_t) n * 4, hipMemcpyDeviceToHost, cst));
			if (cst != m->st) { MAP_HIP_TRY(hipEventRecord(m->ev_cs_copied, cst)); m->cs_copy_pending = true; }
			MAP_HIP_TRY(hipStreamSynchronize(m->st));
			m->n_cand = (uint64_t) last[0] + last[1];
			{
				// the 32-bit prefix sums wrap silently: cross-check the total against the 64-bit candidate counters
				unsigned long long sum = 0;
				for (int g = 0; g < ngm::kCsRegions; ++g) sum += ctr[(size_t) g * ngm::kCsCursorStride + 2];
				if (sum != m->n_cand) { ngm::pipeline_set_error("%llu candidates in one batch of %d reads exceed the 32-bit candidate index: use smaller batches", sum, n); return -75; }
			}
			m->st_heavy += n_heavy; m->st_reads += (uint64_t) n; m->st_cands += m->n_cand; m->st_exact_lds += n_exact_lds; m->st_exact_global += n_exact_global;
			m->cs_kmers = m->cs_hits = 0;
			for (int g = 0; g < ngm::kCsRegions; ++g) { m->cs_kmers += ctr[(size_t) g * ngm::kCsCursorStride]; m->cs_hits += ctr[(size_t) g * ngm::kCsCursorStride + 1]; }
			const unsigned long long *ph = ctr.data() + ctr_words;
			if (A.phase_cycles)
				fprintf(stderr, "[ngm-hip] cs fast path, 100 MHz ticks per read: lists %.1f sweep1 %.1f sweep2 %.1f candidates %.1f; %u of %d reads re-run by the exact path; kernels %.2f + %.2f + %.2f ms\n",
						(double) ph[0] * 256 / n, (double) ph[1] * 256 / n, (double) ph[2] * 256 / n, (double) ph[3] * 256 / n, m->cs_queued_exact, n, pass_ms[0], pass_ms[1], pass_ms[2]);
			if (A.phase_cycles && m->cs_canon)
				fprintf(stderr, "[ngm-hip] cs canonical path, inside sweep 1: first lines arrived %.1f | chunk items %.1f | first-line votes %.1f | chunk votes %.1f; in front of the phases (resets, prefetch) %.1f\n",
						(double) ph[4] * 256 / n, (double) ph[5] * 256 / n, (double) ph[6] * 256 / n, (double) ph[7] * 256 / n, (double) ph[8] * 256 / n);
			return 0;
		}
		cap *= 4;  // candidate buffer too small: grow and redo the batch
	}
	ngm::pipeline_set_error("candidate buffer overflow persists");
	return -75;
}

// h_base / h_count / h_maxv of the last search are complete (see the end of run_cs)
int cs_host_arrays(ngm_mapper *m) {
	if (!m->cs_copy_pending) return 0;
	MAP_HIP_TRY(hipEventSynchronize(m->ev_cs_copied));
	m->cs_copy_pending = false;
	return 0;
}

int upload_reads(ngm_mapper *m, int n, const char *reads) {
	const size_t bytes = (size_t) n * m->prm.qry_max_len;
	if (m->d_reads.reserve(bytes)) { ngm::pipeline_set_error("out of device memory (reads)"); return -12; }
	MAP_HIP_TRY(hipMemcpyAsync(m->d_reads.p, reads, bytes, hipMemcpyHostToDevice, m->st));
	return 0;
}

// the host tail of a batch (CIGAR / MD strings, coordinate conversion) is embarrassingly parallel over reads;
// NextGenMap does it on its CS threads, here a batch is fanned out over the host cores
std::atomic<int> g_live_mappers{0};  // mapper instances share the host cores
// GPU stages (candidate search / score / align, each from its first launch to its stream sync) of the mapper instances of
// one process take turns: kernels of different instances then do not slow each other down, while the host stages of one
// instance still overlap the GPU stages of the others (NGM_HIP_GPU_STAGE_LOCK=0: let the streams share the GPU)
// NGM_HIP_GPU_STAGE_LOCK=2: one lock per stage KIND -- the align stage of one instance (1 wave per SIMD, hardly any LDS) may then
// run under the search stage of another (LDS-bound at 10 waves per CU), only stages of the same kind take turns.
// per-read host loops run on the process-wide persistent pool (thread_pool.h): shared by the mapper instances, sized
// to this rank's share of the host cores; no threads are started per call
template <typename F>
void parallel_for(int n, F f, int min_grain = 0) {
	ngm::ThreadPool::instance().parallel_for(n, f, min_grain > 0 ? min_grain : 2048);
}

char class_char(uint8_t c) {
	static const char t[8] = {'A', 'C', 'G', 'T', 'x', 'N', 0, 0};
	return t[c & 7];
}

// host twin of window_class (gather_device.h) for the CIGAR/MD pass: DecodeRefSequence into ASCII
void host_window(const ngm_ref *r, uint64_t offset, int buffer_len, int want, char *out) {
	const uint64_t concat_len = r->n_bases - 1;
	uint64_t len = (uint64_t) buffer_len - 2;
	if (offset >= concat_len) { memset(out, 'N', want); return; }
	uint64_t end = 0;
	if (offset + len > concat_len) { end = offset + len - concat_len; len -= end; }
	const uint64_t emitted = ((offset & 1) ? 1 : 0) + 2 * ((len + 1) / 2);
	for (int j = 0; j < want; ++j) {
		const uint64_t jj = (uint64_t) j;
		char ch;
		if (jj < emitted) ch = ((len & 1) && jj == emitted - 1) ? 'x' : class_char(r->host_cls[offset + jj]);
		else if (jj < emitted + end) ch = 'x';
		else ch = 0;
		out[j] = ch;
	}
}

}  // namespace

extern "C" {

int ngm_ref_decode(const ngm_ref *r, uint64_t offset, int buffer_len, char *out) {
	// runs the device window function over one window so that tests exercise the HBM copy
	DevGuard g(r->device);
	if (buffer_len < 2) return -22;
	const uint64_t concat_len = r->n_bases - 1;
	if (offset >= concat_len) { memset(out, 0, buffer_len); return 0; }  // DecodeRefSequence returns false
	// reuse the gather kernel: one pair, read of length 0, corridor 0, q = buffer_len rounded up
	const int q = 8, c = buffer_len;  // window = q + c >= buffer_len bytes
	const int RW = ngm::read_words(q), FW = (q + c + 7) / 8 + 1;
	uint32_t *d_out; uint16_t *d_lens, *d_rows, *d_rl; uint8_t *d_reads; uint32_t *d_pr, *d_pl, *d_ps;
	MAP_HIP_TRY(hipMalloc(&d_out, (size_t) (RW + FW) * 64 * 4)); MAP_HIP_TRY(hipMalloc(&d_lens, 128)); MAP_HIP_TRY(hipMalloc(&d_rows, 8));
	MAP_HIP_TRY(hipMalloc(&d_rl, 8)); MAP_HIP_TRY(hipMalloc(&d_reads, q)); MAP_HIP_TRY(hipMalloc(&d_pr, 4)); MAP_HIP_TRY(hipMalloc(&d_pl, 4)); MAP_HIP_TRY(hipMalloc(&d_ps, 4));
	MAP_HIP_TRY(hipMemset(d_rl, 0, 8)); MAP_HIP_TRY(hipMemset(d_reads, 0, q)); MAP_HIP_TRY(hipMemset(d_pr, 0, 4)); MAP_HIP_TRY(hipMemset(d_ps, 0, 4));
	const uint32_t loc = (uint32_t) offset;
	MAP_HIP_TRY(hipMemcpy(d_pl, &loc, 4, hipMemcpyHostToDevice));
	ngm::WindowGeom G{concat_len, buffer_len, 0};
	hipLaunchKernelGGL(ngm::gather_pairs_kernel, dim3(1), dim3(256), 0, 0, d_reads, d_rl, q, r->d_genome, G, d_pr, d_pl, d_ps, 1, RW, FW, d_out, d_lens, d_rows);
	MAP_HIP_TRY(hipGetLastError());
	std::vector<uint32_t> h((size_t) (RW + FW) * 64);
	MAP_HIP_TRY(hipMemcpy(h.data(), d_out, h.size() * 4, hipMemcpyDeviceToHost));
	for (int j = 0; j < buffer_len; ++j) {
		const uint32_t w = h[(size_t) (RW + j / 8) * 64];
		const int b = j & 7;
		const uint32_t cls = (b < 4) ? (w >> (8 * b)) & 15u : (w >> (8 * (b - 4) + 4)) & 15u;
		out[j] = class_char((uint8_t) cls);
	}
	(void) hipFree(d_out); (void) hipFree(d_lens); (void) hipFree(d_rows); (void) hipFree(d_rl); (void) hipFree(d_reads); (void) hipFree(d_pr); (void) hipFree(d_pl); (void) hipFree(d_ps);
	return 1;
}

ngm_mapper *ngm_mapper_create(const ngm_ref *ref, const ngm_mapper_params *p) {
	if (!ref || !p) { ngm::pipeline_set_error("ngm_mapper_create: null argument"); return nullptr; }
	DevGuard g(ref->device);
	if (p->qry_max_len < ref->prm.kmer + 1 || p->qry_max_len > 1024) { ngm::pipeline_set_error("ngm_mapper_create: qry_max_len %d out of range", p->qry_max_len); return nullptr; }
	ngm_hip_params ep{};
	ep.abi_version = NGM_HIP_ABI_VERSION;
	ep.qry_max_len = p->qry_max_len; ep.corridor = p->corridor;
	ep.match_bonus = p->match_bonus; ep.mismatch_penalty = p->mismatch_penalty; ep.gap_read_penalty = p->gap_read_penalty; ep.gap_ref_penalty = p->gap_ref_penalty;
	ep.variant = p->variant; ep.hard_clip = p->hard_clip; ep.silent_clip = p->silent_clip; ep.max_batch = 0;
	ep.personality = p->personality; ep.gap_extend_penalty = p->gap_extend_penalty;
	if (p->bs_mapping) {
		if (ref->prm.kmer_skip != 0) { ngm::pipeline_set_error("ngm_mapper_create: bisulfite mapping needs a reference index built with kmer_skip 0 (src/PrefixTable.cpp:199-207)"); return nullptr; }
		if (p->mode != 0) { ngm::pipeline_set_error("ngm_mapper_create: '--bs-mapping' and '--end-to-end' can't be used at the same time"); return nullptr; }   // Config.cpp:448-470 (-n is allowed there: ScoreBuffer::topNSE)
		ep.alt_scoring = ep.alt_cigar = NGM_ALT_BISULFITE; ep.match_bonus_tt = p->match_bonus_tt; ep.match_bonus_tc = p->match_bonus_tc;
	}
	if (p->slam_seq) {
		if (p->bs_mapping) { ngm::pipeline_set_error("ngm_mapper_create: '--bs-mapping' and '--slam-seq' can't be used at the same time!"); return nullptr; }  // Config.cpp:454-457
		if (p->personality != NGM_PERSONALITY_LINEAR) { ngm::pipeline_set_error("ngm_mapper_create: '--slam-seq' needs the default (linear-gap) personality: EndToEndAffine produces no per-base records (Align::ExtendedData)"); return nullptr; }
		ep.alt_cigar = NGM_ALT_SLAMSEQ;
		if (p->slam_seq & 2) { ep.alt_scoring = NGM_ALT_SLAMSEQ; ep.match_bonus_tt = p->match_bonus_tt; ep.match_bonus_tc = p->match_bonus_tc; }
	}
	ngm_hip_ctx *eng = ngm_hip_create(ref->device, &ep);
	if (!eng) { ngm::pipeline_set_error("%s", ngm_hip_last_error(nullptr)); return nullptr; }
	ngm_mapper *m = new ngm_mapper();
	++g_live_mappers;
	m->ref = ref; m->prm = *p; m->eng = eng; m->st = eng->stream;
	{
		int lo = 0, hi = 0;
		(void) hipDeviceGetStreamPriorityRange(&lo, &hi);  // hi = numerically smallest = greatest priority
		// (the order replay's stream: greatest priority by default -- persistent search workgroups hold every CU until their launch ends, and
		// the replay gets in as they leave; NGM_HIP_ORDER_PRIORITY=low / normal: experiments on workloads whose replays are long)
		int prio = hi;
		if (const char *e = getenv("NGM_HIP_ORDER_PRIORITY")) prio = !strcmp(e, "low") ? lo : !strcmp(e, "normal") ? (lo + hi) / 2 : hi;
		if (hipStreamCreateWithPriority(&m->st_hi, hipStreamNonBlocking, prio) != hipSuccess) m->st_hi = nullptr;
		if (hipStreamCreateWithFlags(&m->st_copy, hipStreamNonBlocking) != hipSuccess) m->st_copy = nullptr;
		for (auto &e : m->turn_ev) if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) e = nullptr;
		if (hipEventCreateWithFlags(&m->ev_cs_done, hipEventDisableTiming) != hipSuccess) m->ev_cs_done = nullptr;
		if (hipEventCreateWithFlags(&m->ev_cs_copied, hipEventDisableTiming) != hipSuccess) m->ev_cs_copied = nullptr;
	}
	m->max_kfreq = p->max_kfreq > 0 ? p->max_kfreq : ref->auto_max_kfreq;
	for (auto &e : m->ev) (void) hipEventCreate(&e);
	for (auto &e : m->cev) (void) hipEventCreate(&e);
	for (auto &e : m->oev) (void) hipEventCreate(&e);
	// fast-path geometry from the expected hits per read H = 2 (q - k) lists x average list length:
	// bit planes >= 12 H bits (6-8 % of the single background hits collide and survive the filter),
	// small exact table for the survivors + the real signal with headroom
	{
		const double avg_list = (double) ref->n_entries / (double) (1ull << (2 * ref->prm.kmer));
		const double hexp = std::max(64.0, 2.0 * std::max(1, p->qry_max_len - ref->prm.kmer) * avg_list);
		m->cs_hexp = hexp;
		int lb = 12;
		while ((double) (1u << lb) < 12.0 * hexp && lb < 17) ++lb;
		m->cs_log2_bits = lb;
		// plane of P bits (any multiple of 2048 from 12 bits per expected hit up to the next power of two) and table of
		// 2^ls slots, 3/4 of which may fill: entries = hits that find their bit already set -- H^2 / (2 P) by collision
		// -- plus the real repeats, with headroom.  Take the pair that needs the least LDS.
		size_t best_bytes = ~(size_t) 0;
		const double p_lo = std::min(131072.0, std::max(4096.0, ceil(12.0 * hexp / 2048.0) * 2048.0)), p_hi = (double) (1u << lb);
		for (double P = p_lo; P <= p_hi; P += 2048.0) {
			int ls = 8;
			while (0.75 * (double) (1u << ls) < 1.3 * hexp * hexp / (2.0 * P) + 0.02 * hexp + 100.0 && ls < 12) ++ls;
			const size_t bytes = (size_t) P / 8 + ((size_t) 8 << ls) + ((size_t) 3 << ls);  // plane + keys/votes + queue
			if (bytes < best_bytes) { best_bytes = bytes; m->cs_plane_bits = (uint32_t) P; m->cs_log2_small = ls; }
		}
		// expected 8-hit segments per read: every list contributes its hits / 8 plus, on average, 7/16 of a segment of slack
		const double segs = hexp / 8.0 + 0.44 * 2.0 * std::max(1, p->qry_max_len - ref->prm.kmer);
		m->cs_fast_items = segs * 1.10 > 64.0 * ngm::kCsFastItemsShort ? ngm::kCsFastItemsLong : ngm::kCsFastItemsShort;
		if (const char *e = getenv("NGM_HIP_CS_FAST_ITEMS")) m->cs_fast_items = atoi(e) > ngm::kCsFastItemsShort ? ngm::kCsFastItemsLong : ngm::kCsFastItemsShort;  // tests
		const uint32_t plane_untrimmed = m->cs_plane_bits;
		m->cs_plane_bits0 = plane_untrimmed;
		// The kernel is bound by reads in flight per CU, and those by LDS, which gfx950 hands out in granules of
		// 1 280 bytes (160 KB / 128: hipOccupancyMaxActiveBlocksPerMultiprocessor reports 9 workgroups of 16 328 bytes per CU and 10
		// of 15 360).  The plane is sized generously (12 bits per expected hit): when giving up at most a fifth of it (never below
		// 10 bits per hit -- the spurious table entries H^2 / 2P stay far from the table's capacity) lets one more read in, do it.
		{
			ngm::CsArgs G{};
			G.lists_cap = 2 * std::max(1, p->qry_max_len - ref->prm.kmer + 1); G.q = p->qry_max_len; G.log2_slots = m->cs_log2_small;
			G.fast_items = m->cs_fast_items; G.items16 = (G.lists_cap <= 512) ? 1 : 0; G.plane_bits = m->cs_plane_bits;
			const size_t granule = 1280, lds = 160 * 1024;
			const size_t bytes = (cs_lds_bytes(G, ngm::kCsFast) + granule - 1) / granule * granule;
			const size_t per_cu = lds / std::max<size_t>(bytes, 1);
			if (per_cu >= 1 && per_cu < 10) {
				const size_t target = lds / (per_cu + 1) / granule * granule;  // bytes that would let one more read in
				const size_t have = cs_lds_bytes(G, ngm::kCsFast);
				if (have > target) {
					const uint32_t cut_bits = (uint32_t) (((have - target) * 8 + 31) / 32 * 32);
					if (cut_bits <= m->cs_plane_bits / 5 && (double) (m->cs_plane_bits - cut_bits) >= 10.0 * hexp) m->cs_plane_bits -= cut_bits;
				}
			}
		}
	}
	// which index layout the fast path gathers from: canonical pair buckets (odd k, up to 256 k-mers per read, k-mer pairs in
	// use shorter than 1 000 hits: the chunk items are 16-bit) unless NGM_HIP_CS_PLAIN_BUCKETS asks for one bucket per k-mer
	{
		const int n_kmers = std::max(1, p->qry_max_len - ref->prm.kmer + 1);
		const bool canon_ok = (ref->prm.kmer & 1) && n_kmers <= 256 && m->max_kfreq <= 1000 && !getenv("NGM_HIP_CS_PLAIN_BUCKETS") && !p->bs_mapping;
		if (!p->bs_mapping && ngm_ref_ensure_buckets(ref, canon_ok ? 1 : 0) != 0) { ngm_mapper_destroy(m); return nullptr; }   // (bisulfite mapping: exact paths only, no buckets)
		if (canon_ok) {
			const int glog = std::min(ref->cbucket_log2_words, 5) - 2;
			int shape = 1;
			while (shape < 3 && (n_kmers > kCanonT[shape] * 64 || n_kmers > kCanonR1[shape] * ((kCanonT[shape] * 64) >> glog))) ++shape;
			if (const char *e = getenv("NGM_HIP_CS_CANON_SHAPE")) shape = std::max(shape, std::min(3, atoi(e)));  // tests: a larger shape than needed
			m->cs_canon = shape;
			if (const char *e = getenv("NGM_HIP_CS_CANON_CH")) m->cs_canon_ch = atoi(e);
			if (const char *e = getenv("NGM_HIP_CS_CANON_WPE")) m->cs_canon_wpe = atoi(e);
			// the canonical kernel indexes its plane with the low bits of the bin: a power of two of bits, at least 10 per expected hit
			// (150 bp reads at GRCh38 size: 65 536 bits; with the rest of a read's LDS 16 KB -> 13 granules of 1 280 bytes, nine reads per CU)
			uint32_t pb = 4096;
			while ((double) pb < 10.0 * m->cs_hexp && pb < 131072u) pb <<= 1;
			m->cs_plane_bits = pb;
		}
	}
	ngm::CsArgs A{}; A.lists_cap = 2 * std::max(1, p->qry_max_len - ref->prm.kmer + 1); A.q = p->qry_max_len;
	A.log2_slots = m->cs_log2_slots;
	if (cs_lds_bytes(A, ngm::kCsExactLds) > 158 * 1024) { m->cs_log2_slots = 13; A.log2_slots = 13; } A.log2_bits = 17; A.plane_bits = 131072;
	A.fast_items = ngm::kCsFastItemsLong;
	A.items16 = 0;
	const int log2_exact = A.log2_slots;
	A.log2_slots = 12;  // the largest table the fast path picks (above)
	(void) hipFuncSetAttribute((const void *) ngm::cs_fast_kernel<ngm::kCsFastItemsShort, uint16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) cs_lds_bytes(A, ngm::kCsFast));
	(void) hipFuncSetAttribute((const void *) ngm::cs_fast_kernel<ngm::kCsFastItemsLong, uint16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) cs_lds_bytes(A, ngm::kCsFast));
	(void) hipFuncSetAttribute((const void *) ngm::cs_fast_kernel<ngm::kCsFastItemsLong, uint32_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) cs_lds_bytes(A, ngm::kCsFast));
#define NGM_CS_ATTR_T(T) \
	(void) hipFuncSetAttribute((const void *) ngm::cs_fast2_kernel<T, ngm::kCsFastItemsShort / T, uint16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) cs_lds_bytes(A, ngm::kCsFast)); \
	(void) hipFuncSetAttribute((const void *) ngm::cs_fast2_kernel<T, ngm::kCsFastItemsLong / T, uint16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) cs_lds_bytes(A, ngm::kCsFast))
	NGM_CS_ATTR_T(2); NGM_CS_ATTR_T(3); NGM_CS_ATTR_T(4);
#undef NGM_CS_ATTR_T
	for (int shape = 1; shape <= 3; ++shape) for (int ch : {0, 1, 3}) for (int wpe : {5, 6, 7, 8}) for (int bs : {0, 2})
		(void) hipFuncSetAttribute(cs_canon_fn(shape, ch, wpe, bs), hipFuncAttributeMaxDynamicSharedMemorySize, (int) cs_canon_lds_bytes(A, shape));
	// three waves per read for the 768-segment size (150 bp reads), four for the 1 536-segment one (250 bp: 12.3 instead of 14.7 ms
	// per 524 288 reads -- with twice the work items per read the fourth wave pays for the seventh-of-a-CU it costs)
	m->cs_waves = m->cs_fast_items == ngm::kCsFastItemsLong ? 4 : 3;
	if (const char *e = getenv("NGM_HIP_CS_WAVES")) m->cs_waves = std::min(4, std::max(1, atoi(e)));
	A.log2_slots = log2_exact;
	(void) hipFuncSetAttribute((const void *) ngm::cs_heavy_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
	(void) hipFuncSetAttribute((const void *) ngm::cs_heavy_kernel<512>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
	(void) hipFuncSetAttribute((const void *) ngm::cs_heavy_kernel<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
	(void) hipFuncSetAttribute((const void *) ngm::cs_heavy_kernel<1024, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
	(void) hipFuncSetAttribute((const void *) ngm::cs_heavy2_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
	(void) hipFuncSetAttribute((const void *) ngm::cs_heavy2_kernel<512>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
	(void) hipFuncSetAttribute((const void *) ngm::cs_heavy2_kernel<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
	(void) hipFuncSetAttribute((const void *) ngm::cs_order_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);  // per device
	(void) hipFuncSetAttribute((const void *) ngm::cs_order_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
	if (p->bs_mapping) { A.bs = 1; A.lists_cap = 2 * ngm::kCsBsChunk; A.log2_slots = std::min(A.log2_slots, 13); }
	(void) hipFuncSetAttribute((const void *) ngm::cs_kernel<ngm::kCsExactLds>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) cs_lds_bytes(A, ngm::kCsExactLds));
	(void) hipFuncSetAttribute((const void *) ngm::cs_kernel<ngm::kCsExactGlobal>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) cs_lds_bytes(A, ngm::kCsExactGlobal));
	(void) hipGetLastError();  // a refused attribute shows up as a launch failure where it matters, not as a stale error at the next check
	return m;
}

void ngm_mapper_destroy(ngm_mapper *m) {
	if (!m) return;
	if (--g_live_mappers == 0 && getenv("NGM_HIP_HOST_TIMING"))
		fprintf(stderr, "[ngm-hip] GPU stage lock, ms summed over all mappers: search + score held %.1f (waited %.1f) | align held %.1f (waited %.1f) | SAM text held %.1f (waited %.1f)\n",
				g_stage_hold_us[0] / 1e3, g_stage_wait_us[0] / 1e3, g_stage_hold_us[1] / 1e3, g_stage_wait_us[1] / 1e3, g_stage_hold_us[2] / 1e3, g_stage_wait_us[2] / 1e3);
	DevGuard g(m->ref->device);
	(void) hipStreamSynchronize(m->st);
	ngm_bgzf_destroy(m->bz);
	if (m->st_hi) { (void) hipStreamSynchronize(m->st_hi); (void) hipStreamDestroy(m->st_hi); }
	if (m->st_copy) { (void) hipStreamSynchronize(m->st_copy); (void) hipStreamDestroy(m->st_copy); }
	// (no other instance may be left waiting for an event of this one: its kernels have finished -- the stream was synchronised above)
	for (int k2 = 0; k2 < 2; ++k2) {
		StageChain &c = g_chain[(unsigned) m->ref->device & 15u][k2];
		std::lock_guard<std::mutex> lk(c.mu);
		for (auto &e : m->turn_ev) if (e && c.last == e) c.last = nullptr;
	}
	for (auto &e : m->turn_ev) if (e) (void) hipEventDestroy(e);
	if (m->ev_cs_done) (void) hipEventDestroy(m->ev_cs_done);
	if (m->ev_cs_copied) (void) hipEventDestroy(m->ev_cs_copied);
	m->d_reads.release(); m->d_read_len.release(); m->d_cand_base.release(); m->d_cand_count.release(); m->d_out_loc.release(); m->d_out_sv.release();
	m->d_status.release(); m->d_ovf_read.release(); m->d_ovf_read2.release(); m->d_ovf_hits.release(); m->d_ovf_log2.release(); m->d_ovf_off.release(); m->d_gt_keys.release();
	m->d_gt_votes.release(); m->d_heavy_list.release(); m->d_heavy_ctr.release(); m->d_max_votes.release(); m->d_max_both.release(); m->d_scores.release(); m->d_best.release(); m->d_total.release(); m->d_pair_read.release();
	m->d_winner.release(); m->d_a_read.release(); m->d_a_loc.release(); m->d_a_sv.release(); m->d_mapq.release(); m->d_nbest.release();
	m->d_records.release(); m->d_runs.release(); m->d_runs_c.release();
	m->d_pair_info.release(); m->p_pair_info.release(); m->d_sam_contig_start.release();
	m->d_pair_out.release(); m->d_pair_top.release(); m->d_pair_tied_n.release(); m->d_pair_list.release(); m->p_pair_out.release(); m->p_pair_top.release(); m->p_pair_tied_n.release();
	m->d_sam_contig_names.release(); m->d_sam_rg.release(); m->d_sam_names.release(); m->d_sam_text.release(); m->d_sam_contig_off.release(); m->d_sam_len.release(); m->d_sam_off.release();
	m->d_sam_quals.release(); m->d_sam_meta.release(); m->d_sam_refs.release(); m->d_sam_hits.release(); m->p_sam_hits.release(); m->p_sam_refs.release(); m->p_sam_extra.release();
	for (auto &e : m->ev) if (e) (void) hipEventDestroy(e);
	for (auto &e : m->cev) if (e) (void) hipEventDestroy(e);
	for (auto &e : m->oev) if (e) (void) hipEventDestroy(e);
	m->d_counters.release(); m->d_heavy_diag.release(); m->d_order_list.release(); m->d_cand_rank.release(); m->d_order_scratch.release(); m->d_order_info.release(); m->p_order_info.release(); m->d_order_big.release(); m->d_order_gt.release(); m->d_order_log2.release(); m->d_order_off.release(); m->p_rank.release(); m->h_base.b.release(); m->h_count.b.release(); m->h_maxv.b.release(); m->d_out_loc2.release(); m->d_out_sv2.release(); m->d_new_base.release(); m->d_scan_tmp.release();
	m->p_winner.release(); m->p_loc.release(); m->p_sv.release(); m->p_mapq.release(); m->p_nbest.release(); m->p_rec.release();
	m->p_best.release(); m->p_scores.release(); m->p_runs.release();
	m->d_str.release(); m->d_cigout.release(); m->p_cigout.release(); m->p_str.release();
	ngm_hip_destroy(m->eng);
	delete m;
}

int ngm_mapper_cs(ngm_mapper *m, int n, const char *reads, uint32_t *cand_offsets, float *max_votes) {
	if (!m || n < 0) return -22;
	DevGuard g(m->ref->device);
	if (n == 0) { cand_offsets[0] = 0; m->n_reads = 0; m->n_cand = 0; return 0; }
	if (int r = upload_reads(m, n, reads)) return r;
	m->cs_paired = false;
	if (int r = run_cs(m, n)) return r;
	if (int r = cs_host_arrays(m)) return r;
	uint32_t acc = 0;
	for (int i = 0; i < n; ++i) { cand_offsets[i] = acc; acc += m->h_count[i]; max_votes[i] = m->h_maxv[i]; }
	cand_offsets[n] = acc;
	return 0;
}

int ngm_mapper_cs_fetch(ngm_mapper *m, uint64_t *loc, uint8_t *strand, float *votes) {
	if (!m) return -22;
	DevGuard g(m->ref->device);
	if (m->n_cand == 0) return 0;
	std::vector<uint32_t> hl(m->n_cand), hs(m->n_cand);
	MAP_HIP_TRY(hipMemcpy(hl.data(), m->d_out_loc.p, m->n_cand * 4, hipMemcpyDeviceToHost));
	MAP_HIP_TRY(hipMemcpy(hs.data(), m->d_out_sv.p, m->n_cand * 4, hipMemcpyDeviceToHost));
	size_t w = 0;
	std::vector<uint32_t> order;
	for (int i = 0; i < m->n_reads; ++i) {
		const uint32_t b = m->h_base[i], c = m->h_count[i];
		order.resize(c);
		std::iota(order.begin(), order.end(), b);
		std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
			const uint64_t kx = ((uint64_t) hl[x] << 1) | (hs[x] & 1), ky = ((uint64_t) hl[y] << 1) | (hs[y] & 1);
			return kx < ky;
		});
		for (uint32_t j : order) { loc[w] = hl[j]; strand[w] = (uint8_t) (hs[j] & 1); votes[w] = (float) (hs[j] >> 1); ++w; }
	}
	return 0;
}

// the SAM stage of a call (ngm_mapper_map_sam): inputs the records need beyond the reads, where the text goes
struct SamCall { const char *quals; const char *names; size_t names_bytes; const ngm::SamMeta *meta; char *out; size_t out_cap; uint64_t *stats; long long text_bytes; float kernel_ms; };
static int map_impl(ngm_mapper *m, int n, const char *reads, const void *d_reads_ext, ngm_hit *hits, char *cigars, char *mds, bool paired, SamCall *sam = nullptr);

// Reference order of the candidates of the listed reads (cs_order_kernel): h_rank[c] for every candidate c of those
// reads, kCsOrderUnknown where it could not be determined.  Only called for reads where the order decides something.
static int candidate_order_finish(ngm_mapper *m, hipStream_t ost, uint64_t np);
static int candidate_order_wait(ngm_mapper *m, uint32_t **h_rank) {
	static const bool order_on_main = getenv("NGM_HIP_ORDER_ON_MAIN_STREAM") != nullptr;
	if (int rc = candidate_order_finish(m, (m->st_hi && !order_on_main) ? m->st_hi : m->st, m->n_cand)) return rc;
	*h_rank = m->p_rank.p;
	return 0;
}
// After the LDS replay: wait for it, account for the reads it left to the exact kernel (more hits than its time line, more repeated
// bins than its table: CsArgs::order_info) and replay those exactly in global memory.  No read keeps an undetermined order silently.
static int candidate_order_finish(ngm_mapper *m, hipStream_t ost, uint64_t np) {
	static const bool trace = getenv("NGM_HIP_ORDER_TRACE") != nullptr;   // (diagnostics: where the time of a replay goes, stage by stage)
	const auto t_trace = std::chrono::steady_clock::now();
	auto tr = [&](const char *what, unsigned long long a = 0, unsigned long long b = 0, unsigned long long c = 0) {
		if (trace) fprintf(stderr, "[ngm-hip] order trace %p +%.1f ms: %s %llu %llu %llu\n", (void *) m, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_trace).count(), what, a, b, c);
	};
	tr("wait for the LDS replay", m->order_pending.size());
	MAP_HIP_TRY(hipStreamSynchronize(ost));
	tr("LDS replay done");
	{ float t = 0; if (hipEventElapsedTime(&t, m->oev[0], m->oev[1]) == hipSuccess) m->order_ms += t; }
	const uint32_t nl = (uint32_t) m->order_pending.size();
	m->st_order_reads += nl;
	std::vector<uint32_t> big;
	for (uint32_t i = 0; i < nl; ++i) if (m->p_order_info.p[2 * i + 1] & 0xFFu) big.push_back(i);
	if (const char *hf = getenv("NGM_HIP_ORDER_HIST")) {   // diagnostics: hits / tracked bins / outcome of every replayed read
		if (FILE *f = fopen(hf, "a")) { for (uint32_t i = 0; i < nl; ++i) fprintf(f, "%u %u %u\n", m->p_order_info.p[2 * i], m->p_order_info.p[2 * i + 1] & 0xFFu, m->p_order_info.p[2 * i + 1] >> 8); fclose(f); }
	}
	m->st_order_big += big.size();
	const std::vector<uint32_t> beyond_lds = big;
	// The reads beyond the LDS replay: hits dealt into buckets (cs_order_bucket_kernel -- no table in global memory); what that kernel
	// leaves (bisulfite runs, a bucket of more than 256 hits) goes on to the replay with a table in global memory below.
	static const bool buckets_on = getenv("NGM_HIP_ORDER_NO_BUCKETS") == nullptr;
	if (!big.empty() && buckets_on && !m->order_args.bs && ngm::cs_order_tau(m->order_args.lists_cap) <= ngm::kCsOrderBucketMaxTau) {
		const uint32_t nb = (uint32_t) big.size();
		std::sort(big.begin(), big.end(), [&](uint32_t a, uint32_t b) { const uint32_t ha = m->p_order_info.p[2 * a], hb = m->p_order_info.p[2 * b]; return ha != hb ? ha > hb : a < b; });   // the longest first: they end the launch
		std::vector<uint32_t> reads(nb);
		for (uint32_t j = 0; j < nb; ++j) reads[j] = m->order_pending[big[j]];
		ngm::CsArgs B = m->order_args;
		B.order_info = nullptr; B.order_scratch = nullptr; B.order_max_hits = 0;
		B.order_gcap = 0;
		const size_t coarse_cap = ngm::cs_heavy2_coarse_cap(B.lists_cap, B.max_kfreq);
		const size_t lds = ngm::cs_order_bucket_lds_bytes(B.lists_cap, B.q, coarse_cap);
		static const bool two_per_cu = getenv("NGM_HIP_ORDER_BUCKET_W8") != nullptr;   // (experiments)
		auto kern = two_per_cu ? ngm::cs_order_bucket_kernel_w8<ngm::kCsOrderBucketThreads> : ngm::cs_order_bucket_kernel<ngm::kCsOrderBucketThreads>;
		int per_cu = 0, cus = 0;
		if (lds > 64 * 1024) (void) hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
		if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, ngm::kCsOrderBucketThreads, lds) != hipSuccess || per_cu < 1) per_cu = 1;
		if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, m->ref->device) != hipSuccess || cus < 1) cus = 256;
		// ONE workgroup per CU unless told otherwise: its eight waves leave the CU's other wave slots and LDS to the search kernels of the other
		// mapper instances, which run at the same time (measured at 3.1 Gbp, four instances: 2.63 M reads/s with one, 2.54 M with the two that fit)
		static const int per_cu_env = getenv("NGM_HIP_ORDER_BUCKET_PER_CU") ? atoi(getenv("NGM_HIP_ORDER_BUCKET_PER_CU")) : 1;
		if (per_cu_env > 0) per_cu = std::min(per_cu, per_cu_env);
		// elements of a workgroup's slice: the most hits a read of this run can have (a list per k-mer and strand, none longer than max_kfreq) --
		// not the most of THIS list: every new maximum would be a hipFree + hipMalloc, two device-wide synchronisations, in the middle of the run
		uint64_t cap = std::min<uint64_t>(ngm::kCsOrderBucketMaxHits, (((uint64_t) (B.lists_cap / 2) * (uint64_t) std::max(B.max_kfreq, 1)) + 63) & ~63ull);
		uint32_t grid = (uint32_t) per_cu * (uint32_t) cus;
		{
			size_t free_b = 0, total_b = 0;
			uint64_t room = 4ull << 30;
			if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) room = std::min<uint64_t>(room, ((uint64_t) free_b + (uint64_t) m->d_order_gt.cap * 4) / 2);
			cap = std::min<uint64_t>(cap, room / 8);                       // (a read with more hits than that goes on to the table kernel)
			grid = (uint32_t) std::max<uint64_t>(1, std::min<uint64_t>(grid, room / 8 / std::max<uint64_t>(cap, 1)));
		}
		const size_t slice_words = (size_t) grid * cap * 2;
		grid = std::min<uint32_t>(grid, nb);
		tr("bucket stage: reads, grid, slice", nb, grid, cap);
		bool ok = cap >= 64 && !m->d_order_gt.reserve(slice_words) && !m->d_order_big.reserve(nb) && !m->d_order_log2.reserve(nb + 1) && !m->d_order_info.reserve(2 * (size_t) nb);
		if (ok) {
			MAP_HIP_TRY(hipMemcpyAsync(m->d_order_big.p, reads.data(), (size_t) nb * 4, hipMemcpyHostToDevice, ost));
			MAP_HIP_TRY(hipMemsetAsync(m->d_order_log2.p, 0, 4, ost));
			MAP_HIP_TRY(hipMemsetAsync(m->d_order_info.p, 0xFF, 2 * (size_t) nb * 4, ost));
			unsigned long long *diag = getenv("NGM_HIP_CS_PHASES") ? m->d_counters.p + (size_t) ngm::kCsRegions * ngm::kCsCursorStride + 8 : nullptr;
			if (diag) MAP_HIP_TRY(hipMemsetAsync(diag, 0, 16 * 8, ost));
			B.read_list = m->d_order_big.p;
			MAP_HIP_TRY(hipEventRecord(m->oev[2], ost));
			hipLaunchKernelGGL(kern, dim3(grid), dim3(ngm::kCsOrderBucketThreads), lds, ost, B, nb, m->d_order_log2.p, (uint2 *) m->d_order_gt.p, (uint32_t) cap, (uint32_t) coarse_cap,
					(const uint32_t *) m->d_out_loc.p, (const uint32_t *) m->d_out_sv.p, m->d_cand_rank.p, m->d_order_info.p, diag);
			MAP_HIP_TRY(hipGetLastError());
			MAP_HIP_TRY(hipEventRecord(m->oev[3], ost));
			tr("bucket kernel launched");
			std::vector<uint32_t> binfo(2 * (size_t) nb);
			MAP_HIP_TRY(hipMemcpyAsync(binfo.data(), m->d_order_info.p, binfo.size() * 4, hipMemcpyDeviceToHost, ost));
			MAP_HIP_TRY(hipStreamSynchronize(ost));
			{ float t = 0; if (hipEventElapsedTime(&t, m->oev[2], m->oev[3]) == hipSuccess) m->order_ms += t; }
			tr("bucket kernel done");
			if (diag) {
				unsigned long long ph[16];
				MAP_HIP_TRY(hipMemcpy(ph, diag, sizeof(ph), hipMemcpyDeviceToHost));
				const double ns = (double) std::max(1ull, ph[8]);
				fprintf(stderr, "[ngm-hip] order replay through buckets (%u reads, grid %u, %d per CU, slice %llu hits), us per sampled read: lists %.1f | count %.1f | scan + scatter %.1f | v + tau %.1f (wave 0: %.0f windows, bounds + loads issued %.1f, counted + stored %.1f) | table of M %.1f | candidates %.1f; hits %.0f, candidates %.0f per read; %llu left to the table kernel\n",
						nb, grid, per_cu, (unsigned long long) cap, ph[0] / ns / 100.0, ph[1] / ns / 100.0, ph[2] / ns / 100.0, ph[3] / ns / 100.0, ph[14] / ns, ph[12] / ns / 100.0, ph[13] / ns / 100.0, ph[4] / ns / 100.0, ph[5] / ns / 100.0, ph[9] / ns, ph[10] / ns, ph[11]);
			}
			std::vector<uint32_t> left;
			for (uint32_t j = 0; j < nb; ++j) if (binfo[2 * j + 1] != 0u) left.push_back(big[j]);
			std::sort(left.begin(), left.end());
			m->st_order_table += left.size();
			big.swap(left);
		} else m->st_order_table += big.size();   // (no room for the slices: all of them to the table kernel)
	}
	if (!big.empty() && (!buckets_on || m->order_args.bs || ngm::cs_order_tau(m->order_args.lists_cap) > ngm::kCsOrderBucketMaxTau)) m->st_order_table += big.size();
	if (!big.empty()) {
		// exact replay in global memory (cs_order_kernel<true>): per read a table of 2^l >= 2 (hits + candidates) slots x 5 words and a
		// time line of `hits` words; launches of as many reads as fit a scratch pool of 8 GB
		const uint32_t nb = (uint32_t) big.size();
		std::vector<uint32_t> reads(nb), lg(nb);
		std::vector<uint64_t> off(nb), words(nb);
		for (uint32_t j = 0; j < nb; ++j) {
			const uint32_t i = big[j], rd = m->order_pending[i];
			const uint64_t hits = m->p_order_info.p[2 * i], want = 2ull * (hits + m->h_count[rd]);
			uint32_t l = 11;
			while ((1ull << l) < want && l < 30) ++l;
			reads[j] = rd; lg[j] = l;
			words[j] = (6ull << l) + 2 * (hits + 64) + 64;   // table (5 words per slot), time line, hit times by (slot, strand), the slots in use
		}
		if (m->d_order_big.reserve(nb) || m->d_order_log2.reserve(nb) || m->d_order_off.reserve(nb)) { ngm::pipeline_set_error("out of device memory (exact candidate order)"); return -12; }
		ngm::CsArgs G = m->order_args;
		G.order_info = nullptr; G.order_scratch = nullptr; G.order_max_hits = 0;
		G.order_gcap = G.bs ? 0u : ngm::kCsOrderStage;   // (cs_order_kernel<true>: staging entries per wave)
		G.phase_cycles = getenv("NGM_HIP_CS_PHASES") ? m->d_counters.p + (size_t) ngm::kCsRegions * ngm::kCsCursorStride : nullptr;   // diagnostics: phases of every 64th workgroup
		if (G.phase_cycles) MAP_HIP_TRY(hipMemsetAsync(G.phase_cycles + 8, 0, 12 * 8, ost));
		const size_t lds = ((size_t) G.lists_cap * 4 + 4 + (G.q + 3) / 4 + 2048 + ngm::cs_order_tau(G.lists_cap) + (size_t) (ngm::kCsOrderThreadsGlobal / 64) * G.order_gcap + (G.bs ? (size_t) G.q + 1 + G.lists_cap / 4 + 1 : 0)) * 4;
		// (8 GB per launch: a read with 50 000 hits takes 2.6 MB of table and time line, and with the 1.5 GB pool of the first version the
		// 5 500 such reads of a heavy-tailed batch went through ten launches of ~570 workgroups each -- two per CU, 118 ms of waiting per batch)
		// (the pool never asks for more than half of what the device has free, a read that needs more than the pool -- or a pool
		// that cannot be had -- keeps an UNDETERMINED order, which the run reports (st_order_unknown) instead of dying: ties then resolve by position)
		uint64_t pool_words = 2048ull << 20;
		{
			size_t free_b = 0, total_b = 0;
			if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) pool_words = std::max<uint64_t>(std::min<uint64_t>(pool_words, ((uint64_t) free_b + (uint64_t) m->d_order_gt.cap * 4) / 8), 1ull << 20);
		}
		for (uint32_t j0 = 0; j0 < nb;) {
			if (words[j0] > pool_words) { ++j0; continue; }   // (its candidates keep kCsOrderUnknown from the LDS replay's give-up)
			uint64_t total = 0;
			uint32_t j1 = j0;
			while (j1 < nb && total + words[j1] <= pool_words) { off[j1] = total; total += words[j1]; ++j1; }
			if (m->d_order_gt.reserve(total)) {
				if (pool_words > (1ull << 22)) { pool_words /= 2; continue; }   // a smaller pool, more launches
				break;                                                        // no memory at all: the remaining reads stay undetermined
			}
			tr("table kernel: reads, words", j1 - j0, total, pool_words);
			MAP_HIP_TRY(hipMemcpyAsync(m->d_order_big.p + j0, reads.data() + j0, (size_t) (j1 - j0) * 4, hipMemcpyHostToDevice, ost));
			MAP_HIP_TRY(hipMemcpyAsync(m->d_order_log2.p + j0, lg.data() + j0, (size_t) (j1 - j0) * 4, hipMemcpyHostToDevice, ost));
			MAP_HIP_TRY(hipMemcpyAsync(m->d_order_off.p + j0, off.data() + j0, (size_t) (j1 - j0) * 8, hipMemcpyHostToDevice, ost));
			G.read_list = m->d_order_big.p + j0; G.ovf_log2 = m->d_order_log2.p + j0; G.ovf_table_off = m->d_order_off.p + j0; G.gtable_keys = m->d_order_gt.p;
			MAP_HIP_TRY(hipEventRecord(m->oev[2], ost));
			hipLaunchKernelGGL(ngm::cs_order_kernel<true>, dim3(j1 - j0), dim3(ngm::kCsOrderThreadsGlobal), lds, ost, G, (const uint32_t *) m->d_out_loc.p, (const uint32_t *) m->d_out_sv.p, m->d_cand_rank.p);
			MAP_HIP_TRY(hipGetLastError());
			MAP_HIP_TRY(hipEventRecord(m->oev[3], ost));
			MAP_HIP_TRY(hipStreamSynchronize(ost));   // (reads, lg, off of this launch are consumed; the pool is reused by the next one)
			{ float t = 0; if (hipEventElapsedTime(&t, m->oev[2], m->oev[3]) == hipSuccess) m->order_ms += t; }
			j0 = j1;
		}
		if (G.phase_cycles) {
			MAP_HIP_TRY(hipStreamSynchronize(ost));
			unsigned long long ph[12];
			MAP_HIP_TRY(hipMemcpy(ph, G.phase_cycles + 8, sizeof(ph), hipMemcpyDeviceToHost));
			const double ns = (double) std::max(1ull, ph[4]);
			fprintf(stderr, "[ngm-hip] exact order replay in global memory (%u reads), us per sampled read: lists %.1f | sweep A %.1f | sweep B %.1f | times + tau + entering %.1f; hits %.0f, slots in use %.0f per read; workgroups start to end %.1f us on average, the slowest %.1f us\n",
					nb, ph[0] / ns / 100.0, ph[1] / ns / 100.0, ph[2] / ns / 100.0, ph[3] / ns / 100.0, ph[6] / ns, ph[7] / ns, (double) (ph[5] >> 8) / 100.0 / std::max(1u, nb), ph[8] / 100.0);
		}
	}
	if (!beyond_lds.empty()) {
		tr("ranks");
		MAP_HIP_TRY(hipMemcpyAsync(m->p_rank.p, m->d_cand_rank.p, np * 4, hipMemcpyDeviceToHost, ost));
		MAP_HIP_TRY(hipStreamSynchronize(ost));
		tr("done");
		for (uint32_t i : beyond_lds) {
			const uint32_t rd = m->order_pending[i], b = m->h_base[rd], c = m->h_count[rd];
			bool unknown = false;
			for (uint32_t x = 0; x < c && !unknown; ++x) unknown = m->p_rank.p[b + x] == ngm::kCsOrderUnknown;
			m->st_order_unknown += unknown ? 1 : 0;
		}
	}
	m->order_pending.clear();
	return 0;
}
// wait = false: only enqueue (the list must stay alive until candidate_order_wait)
static int candidate_order(ngm_mapper *m, const std::vector<uint32_t> &list, uint64_t np, uint32_t **h_rank, bool wait = true) {
	const uint32_t nl = (uint32_t) list.size();
	if (m->prm.slam_seq & 4) {
		// the weighted SLAM-seq search replays the votes in the reference's order anyway: its candidates leave in rList order
		if (m->p_rank.reserve(np + 1)) { ngm::pipeline_set_error("out of memory (candidate order)"); return -12; }
		for (uint32_t rd : list) { const uint32_t b = m->h_base[rd], c = m->h_count[rd]; for (uint32_t x = 0; x < c; ++x) m->p_rank.p[b + x] = x; }
		m->st_order_reads += nl;
		m->order_pending.clear();
		*h_rank = m->p_rank.p;
		(void) wait;
		return 0;
	}
	static const bool order_on_main = getenv("NGM_HIP_ORDER_ON_MAIN_STREAM") != nullptr;  // diagnostics
	hipStream_t ost = (m->st_hi && !order_on_main) ? m->st_hi : m->st;  // everything this depends on has been synchronised by the caller
	const auto t_begin = std::chrono::steady_clock::now();
	if (m->d_order_list.reserve(nl) || m->d_cand_rank.reserve(np + 1) || m->p_rank.reserve(np + 1) || m->d_order_info.reserve(2 * (size_t) nl) || m->p_order_info.reserve(2 * (size_t) nl)) { ngm::pipeline_set_error("out of memory (candidate order)"); return -12; }
	m->order_pending = list;
	MAP_HIP_TRY(hipMemcpyAsync(m->d_order_list.p, list.data(), (size_t) nl * 4, hipMemcpyHostToDevice, ost));
	ngm::CsArgs A = m->last_cs;
	A.read_list = m->d_order_list.p;
	A.cand_base = m->d_cand_base.p; A.cand_count = m->d_cand_count.p;
	A.counters = nullptr;
	const size_t ctr_words_o = (size_t) ngm::kCsRegions * ngm::kCsCursorStride;
	A.phase_cycles = getenv("NGM_HIP_CS_PHASES") ? m->d_counters.p + ctr_words_o : nullptr;
	if (A.phase_cycles) MAP_HIP_TRY(hipMemsetAsync(A.phase_cycles + 8, 0, 12 * 8, ost));
	// the time line takes what is left of 80 KB of LDS (reads with more hits walk a slice of global memory, ~10 x slower): the
	// size that lets TWO workgroups share a CU: measured on MI355X, this kernel with 88 KB of LDS has 26
	// workgroups in flight instead of 232 (NGM_HIP_CS_PHASES=1 prints the summed workgroup time; a plain spinning kernel of the
	// same LDS size does reach 232, profiles/tools/lds_occupancy_calib.hip) -- 1.7 s instead of 0.1 s for config 5's 256 k tied reads
	static const size_t lds_budget_kb = getenv("NGM_HIP_ORDER_LDS_KB") ? (size_t) atoi(getenv("NGM_HIP_ORDER_LDS_KB")) : 80;  // (tuning)
	const size_t lds_budget = lds_budget_kb * 1024;
	if (A.bs) A.lists_cap = 2 * 3072;   // bisulfite mapping: the lists of all k-mer variants of a read (more: that read keeps the position order)
	const size_t lds_fixed = ((size_t) A.lists_cap * 3 + 2 + (A.q + 3) / 4 + 2048 + ngm::cs_order_tau(A.lists_cap) + ((size_t) 5 << ngm::kCsOrderLog2Slots) + (A.bs ? (size_t) A.q + 1 + A.lists_cap / 4 + 1 : 0)) * 4;
	// (two arrays of that many entries: the time line and the hit times sorted by bin and strand)
	const size_t hits_room = lds_fixed + 8 * (size_t) ngm::kCsOrderMaxHits < lds_budget ? (lds_budget - 64 - lds_fixed) / 8 : (size_t) ngm::kCsOrderMaxHits;
	A.order_max_hits = (uint32_t) std::max<size_t>(ngm::kCsOrderMaxHits, std::min<size_t>(hits_room, 65535));  // (all of the budget: two workgroups per CU either way)
	const size_t lds = lds_fixed + (size_t) A.order_max_hits * 8;
	// reads with more hits than the LDS time line holds use a slice of a global scratch: launches of at most 4096 reads
	constexpr uint32_t kChunk = 4096, kGcap = 49152;
	// reads with more hits than the LDS time line holds: to the bucket kernel (cs_order_bucket_kernel) -- the LDS replay with its time line in a
	// slice of global memory is what bisulfite runs (no bucket kernel) and NGM_HIP_ORDER_LDS_BIG=1 / NGM_HIP_ORDER_NO_BUCKETS=1 still use
	// (measured at 3.1 Gbp, half of the reads from repeats: 0.745 M reads/s without it, 0.669 M with it)
	static const bool lds_big_env = getenv("NGM_HIP_ORDER_LDS_BIG") != nullptr || getenv("NGM_HIP_ORDER_NO_BUCKETS") != nullptr;
	const bool no_lds_big = !lds_big_env && !A.bs && ngm::cs_order_tau(A.lists_cap) <= ngm::kCsOrderBucketMaxTau;
	if (no_lds_big || m->d_order_scratch.reserve((size_t) std::min(nl, kChunk) * kGcap * 2)) { A.order_scratch = nullptr; A.order_gcap = 0; }   // (time line + hit times per workgroup)
	else { A.order_scratch = m->d_order_scratch.p; A.order_gcap = kGcap; }
	MAP_HIP_TRY(hipEventRecord(m->oev[0], ost));
	for (uint32_t off = 0; off < nl; off += kChunk) {
		A.read_list = m->d_order_list.p + off;
		A.order_info = m->d_order_info.p + 2 * (size_t) off;
		hipLaunchKernelGGL(ngm::cs_order_kernel<false>, dim3(std::min(kChunk, nl - off)), dim3(ngm::kCsOrderThreads), lds, ost, A, (const uint32_t *) m->d_out_loc.p, (const uint32_t *) m->d_out_sv.p, m->d_cand_rank.p);
		MAP_HIP_TRY(hipGetLastError());
	}
	MAP_HIP_TRY(hipEventRecord(m->oev[1], ost));
	MAP_HIP_TRY(hipMemcpyAsync(m->p_rank.p, m->d_cand_rank.p, np * 4, hipMemcpyDeviceToHost, ost));
	MAP_HIP_TRY(hipMemcpyAsync(m->p_order_info.p, m->d_order_info.p, 2 * (size_t) nl * 4, hipMemcpyDeviceToHost, ost));
	m->order_args = A;
	if (!wait) return 0;
	if (int rc = candidate_order_finish(m, ost, np)) return rc;
	*h_rank = m->p_rank.p;
	if (A.phase_cycles) {
		unsigned long long ph[12];
		MAP_HIP_TRY(hipMemcpy(ph, A.phase_cycles + 8, sizeof(ph), hipMemcpyDeviceToHost));
		fprintf(stderr, "[ngm-hip] order replay: slowest workgroup %.1f us; %llu workgroups above 1 ms (most hits among them %llu, most tracked bins %llu)\n", ph[8] / 100.0, ph[9], ph[10], ph[11]);
		const double ns = (double) std::max(1ull, ph[4]);
		fprintf(stderr, "[ngm-hip] order replay, us per sampled read: lists %.1f | sweep A %.1f | sweep B %.1f | compaction + replay %.1f; %llu sampled, %llu gave up, %llu on the global time line; hits %.0f, replayed %.0f per read\n",
				ph[0] / ns / 100.0, ph[1] / ns / 100.0, ph[2] / ns / 100.0, ph[3] / ns / 100.0, ph[4], ph[5] & 0xFFull, 0ull, ph[6] / ns, ph[7] / ns);
		fprintf(stderr, "[ngm-hip] order replay: %.1f us per workgroup start to end, summed %.1f ms over %u workgroups\n", (double) (ph[5] >> 8) / 100.0 / std::max(1u, nl), (double) (ph[5] >> 8) / 1e5, nl);
	}
	if (getenv("NGM_HIP_HOST_TIMING"))
		fprintf(stderr, "[ngm-hip] candidate order replay: %u reads, %.2f ms\n", nl, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
	return 0;
}

int ngm_mapper_map_se(ngm_mapper *m, int n, const char *reads, ngm_hit *hits, char *cigars, char *mds) {
	return map_impl(m, n, reads, nullptr, hits, cigars, mds, false);
}
int ngm_mapper_map_se_resident(ngm_mapper *m, int n, const char *reads, const void *d_reads_ext, ngm_hit *hits, char *cigars, char *mds) {
	return map_impl(m, n, reads, d_reads_ext, hits, cigars, mds, false);
}
int ngm_mapper_map_pe(ngm_mapper *m, int n, const char *reads, ngm_hit *hits, char *cigars, char *mds) {
	return map_impl(m, n, reads, nullptr, hits, cigars, mds, true);
}
int ngm_mapper_map_pe_resident(ngm_mapper *m, int n, const char *reads, const void *d_reads_ext, ngm_hit *hits, char *cigars, char *mds) {
	return map_impl(m, n, reads, d_reads_ext, hits, cigars, mds, true);
}

int ngm_mapper_set_sam_options(ngm_mapper *m, const ngm_sam_options *o) {
	if (!m || !o) return -22;
	DevGuard g(m->ref->device);
	if (o->bam && o->slam_seq) { ngm::pipeline_set_error("ngm_mapper_set_sam_options: BAM records with SLAM-seq tags are formatted by the caller"); return -22; }
	m->sam_opt = *o;
	m->sam_opt.rg_id = nullptr;
	if (o->bam && !m->bz) { m->bz = ngm_bgzf_create(m->ref->device); if (!m->bz) return -12; }
	m->sam_rg = o->rg_id ? o->rg_id : "";
	const ngm_ref *r = m->ref;
	std::string names;
	std::vector<uint32_t> off(r->contigs.size() + 1, 0);
	for (size_t i = 0; i < r->contigs.size(); ++i) { off[i] = (uint32_t) names.size(); names += r->contigs[i].name; }
	off[r->contigs.size()] = (uint32_t) names.size();
	if (m->d_sam_contig_names.reserve(names.size() + 16) || m->d_sam_contig_off.reserve(off.size()) || m->d_sam_rg.reserve(m->sam_rg.size() + 16)) { ngm::pipeline_set_error("out of device memory (SAM options)"); return -12; }
	MAP_HIP_TRY(hipMemcpy(m->d_sam_contig_names.p, names.data(), names.size(), hipMemcpyHostToDevice));
	MAP_HIP_TRY(hipMemcpy(m->d_sam_contig_off.p, off.data(), off.size() * 4, hipMemcpyHostToDevice));
	if (!m->sam_rg.empty()) MAP_HIP_TRY(hipMemcpy(m->d_sam_rg.p, m->sam_rg.data(), m->sam_rg.size(), hipMemcpyHostToDevice));
	{
		std::vector<uint64_t> starts(r->contigs.size() + 1, 0);
		for (size_t i = 0; i < r->contigs.size(); ++i) starts[i] = r->contigs[i].start;
		if (m->d_sam_contig_start.reserve(starts.size())) { ngm::pipeline_set_error("out of device memory (SAM options)"); return -12; }
		MAP_HIP_TRY(hipMemcpy(m->d_sam_contig_start.p, starts.data(), starts.size() * 8, hipMemcpyHostToDevice));
	}
	m->sam_ready = true;
	return 0;
}

static_assert(sizeof(ngm_sam_read) == sizeof(ngm::SamMeta) && sizeof(ngm_sam_read) == 8, "ngm_sam_read is the device's per-read record");

long long ngm_mapper_map_sam(ngm_mapper *m, int n, const char *reads, const char *quals, const char *names, size_t names_bytes, const ngm_sam_read *meta,
		char *out, size_t out_cap, uint64_t stats[3], float *kernel_ms) {
	if (!m || n < 0 || (n > 0 && (!reads || !quals || !meta))) return -22;
	SamCall sc{quals, names, names_bytes, reinterpret_cast<const ngm::SamMeta *>(meta), out, out_cap, stats, 0, 0.f};
	const int rc = map_impl(m, n, reads, nullptr, nullptr, nullptr, nullptr, m->sam_opt.paired != 0, &sc);
	if (rc < 0) return rc;
	if (kernel_ms) *kernel_ms = sc.kernel_ms;
	if (n == 0 && stats) stats[0] = stats[1] = stats[2] = 0;
	return sc.text_bytes;
}

int ngm_mapper_sam_fetch(ngm_mapper *m, char *out, size_t out_cap) {
	if (!m || !out || out_cap < m->sam_text_bytes) return -22;
	DevGuard g(m->ref->device);
	if (m->sam_opt.bam) {   // the last call's records are still in HBM: their BGZF blocks into the larger buffer; returns their length
		if (!m->sam_text_bytes) return 0;
		const long long zlen = ngm_bgzf_compress_device(m->bz, m->d_sam_text.p, (size_t) m->sam_text_bytes, out, out_cap);
		if (zlen < 0 || zlen > 0x7fffffffll) return zlen < 0 ? (int) zlen : -75;
		m->sam_text_bytes = 0;
		return (int) zlen;
	}
	if (m->sam_text_bytes) MAP_HIP_TRY(hipMemcpy(out, m->d_sam_text.p, (size_t) m->sam_text_bytes, hipMemcpyDeviceToHost));
	return 0;
}

// ScoreBuffer::top1PE's std::sort(Scores, sortLocationScore) (src/ScoreBuffer.cpp:373-376) over one read's candidates.
static void sort_like_reference(uint32_t *v, uint32_t base, uint32_t cnt, const uint32_t *loc, const uint32_t *sv, const float *score, const uint32_t *rank,
	bool *ranked_out = nullptr) {
	std::iota(v, v + cnt, base);
	// The reference sorts its candidate list (CollectResultsStd's order) with std::sort(sortLocationScore): an insertion
	// sort -- stable -- up to 16 elements, an unstable introsort above, so there do exactly that on the same sequence.
	// Without the candidate order (first pass; only pairs whose result does not depend on it are kept) any deterministic
	// order will do.
	auto by_place = [&](uint32_t x, uint32_t y) { return loc[x] != loc[y] ? loc[x] < loc[y] : (sv[x] & 1u) < (sv[y] & 1u); };
	bool ranked = rank != nullptr;
	for (uint32_t x = base; ranked && x < base + cnt; ++x) ranked = rank[x] != ngm::kCsOrderUnknown;
	if (ranked_out) *ranked_out = ranked;
	if (!ranked) { std::sort(v, v + cnt, [&](uint32_t x, uint32_t y) { return score[x] != score[y] ? score[x] > score[y] : by_place(x, y); }); return; }
	auto by_rank = [&](uint32_t x, uint32_t y) { return rank[x] != rank[y] ? rank[x] < rank[y] : by_place(x, y); };
	if (cnt <= 16) { std::sort(v, v + cnt, [&](uint32_t x, uint32_t y) { return score[x] != score[y] ? score[x] > score[y] : by_rank(x, y); }); return; }
	std::sort(v, v + cnt, by_rank);
	std::sort(v, v + cnt, [&](uint32_t x, uint32_t y) { return score[x] > score[y]; });
}

// What select_pair learns about the equally scoring pairs of one read pair (see the comment at the end of select_pair).
struct PairTies {
	PairTies() {}  // the arrays stay uninitialised: this is constructed once per pair
	bool equal_scores = false;   // two in-window pairs share a pair score: the running mean insert size is consulted
	bool dup = false;            // ... and two of them also share the insert size: the candidate order decides, NH/X0 counts them
	bool unique_closest = true;  // at the mean passed in, exactly one best-scoring pair is closest to it
	int dmin_top = 0, dmax_top = 0;  // range of the insert sizes of the best-scoring pairs
	int n_top = 0;               // the best-scoring pairs themselves (insert size, candidates), at most 8 unless `dup`
	int top_d[8], top_a[8], top_b[8];
};

extern "C++" {
// The part of ScoreBuffer::top1PE (src/ScoreBuffer.cpp:368-413) that does not depend on the running mean insert size: both candidate
// arrays sorted like the reference sorts them, the MAPQs, the candidates at or above best * pair_score_cutoff, and -- f(pair score,
// insert size, candidate of a, candidate of b) -- every combination inside the insert-size window in the order of the reference's
// double loop.  `a` = the mate whose scores arrive last (the odd read id: "read"), `b` = its mate.
template <typename F>
static void walk_pair(ngm_mapper *m, uint32_t base_a, uint32_t cnt_a, int len_a, uint32_t base_b, uint32_t cnt_b, int len_b,
		const uint32_t *loc, const uint32_t *sv, const float *score, const uint32_t *rank, int *mq_a, int *mq_b, F &&f) {
	auto mq_of = [&](const uint32_t *v, uint32_t cnt) {  // computeMQ(MappedRead*), ScoreBuffer.cpp:42-49
		if (cnt <= 1) return 60;
		const float best = score[v[0]], second = score[v[1]];
		int mq = 0;
		if (best > 0 && second >= 0) mq = (int) ceilf(60.0f * (best - second) / best);
		return mq;
	};
	uint32_t small_a[32], small_b[32];  // nearly always enough; no allocation then
	std::vector<uint32_t> big_a, big_b;
	if (cnt_a > 32) big_a.resize(cnt_a);
	if (cnt_b > 32) big_b.resize(cnt_b);
	uint32_t *A = cnt_a > 32 ? big_a.data() : small_a, *B = cnt_b > 32 ? big_b.data() : small_b;
	sort_like_reference(A, base_a, cnt_a, loc, sv, score, rank);
	sort_like_reference(B, base_b, cnt_b, loc, sv, score, rank);
	*mq_a = mq_of(A, cnt_a); *mq_b = mq_of(B, cnt_b);
	const float cutoff = m->prm.pair_score_cutoff > 0 ? m->prm.pair_score_cutoff : 0.9f;
	const float min_a = score[A[0]] * cutoff, min_b = score[B[0]] * cutoff;
	size_t na = 1, nb = 1;
	while (na < cnt_a && min_a <= score[A[na]]) ++na;
	while (nb < cnt_b && min_b <= score[B[nb]]) ++nb;
	const int min_d = m->prm.min_insert_size, max_d = m->prm.max_insert_size > 0 ? m->prm.max_insert_size : INT_MAX;
	// Mates with hundreds of candidates each (repeat families of a GRCh38-like genome): CheckPairs walks all na x nb combinations,
	// but only those inside the insert-size window do anything -- B's candidates sorted by position, per candidate of A the ones
	// within max_d, visited in increasing j like the reference's inner loop: the same sequence of in-window pairs, so every
	// order-dependent outcome (first best, the equal counter, the recorded combinations) is unchanged.
	const bool windowed = na * nb > 1024 && max_d < (1 << 28);
	std::vector<std::pair<uint32_t, uint32_t>> b_by_loc;
	std::vector<uint32_t> js;
	if (windowed) {
		b_by_loc.resize(nb);
		for (size_t j = 0; j < nb; ++j) b_by_loc[j] = {loc[B[j]], (uint32_t) j};
		std::sort(b_by_loc.begin(), b_by_loc.end());
	}
	for (size_t i = 0; i < na; ++i) {
		size_t n_inner = nb;
		if (windowed) {
			const uint64_t l1w = loc[A[i]];
			const uint32_t lo_loc = l1w > (uint64_t) max_d ? (uint32_t) (l1w - (uint64_t) max_d) : 0u;
			const uint64_t hi64 = l1w + (uint64_t) max_d;
			const uint32_t hi_loc = hi64 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t) hi64;
			js.clear();
			for (auto it = std::lower_bound(b_by_loc.begin(), b_by_loc.end(), std::make_pair(lo_loc, 0u)); it != b_by_loc.end() && it->first <= hi_loc; ++it) js.push_back(it->second);
			std::sort(js.begin(), js.end());
			n_inner = js.size();
		}
		for (size_t jj = 0; jj < n_inner; ++jj) {
			const size_t j = windowed ? js[jj] : jj;
			const uint64_t l1 = loc[A[i]], l2 = loc[B[j]];
			const int cur = (int) ((l2 > l1) ? l2 - l1 + (uint64_t) len_b : l1 - l2 + (uint64_t) len_a);
			if (cur > min_d && cur < max_d) f(score[A[i]] + score[B[j]], cur, (int) A[i], (int) B[j]);
		}
	}
}

// One pair's in-window combinations in the reference's order (walk_pair), kept: ScoreBuffer::CheckPairs at a given running mean
// is then a scan over them -- the sequential pass evaluates an open pair in microseconds instead of sorting and windowing again.
struct PairSeq {
	std::vector<float> ps;
	std::vector<int> d, a, b;
	int mq_a = 0, mq_b = 0;
};
struct PairOutcome { int wa, wb, mqa, mqb, equal, dist; bool found; };
// the double loop of top1PE over CheckPairs (src/ScoreBuffer.cpp:405-413, :463-502) at running mean `avg`
static PairOutcome eval_pair_seq(const PairSeq &q, int avg) {
	float top = 0.0f;
	int distance = 0, equal = 0, ta = -1, tb = -1;
	const size_t nq = q.ps.size();
	for (size_t x = 0; x < nq; ++x) {
		const float ps = q.ps[x];
		const int cur = q.d[x];
		bool take = false;
		if (ps > top * 1.00f) { top = ps; distance = cur; take = true; }
		else if (ps == top) {
			if (abs(distance - avg) > abs(cur - avg)) { top = ps; distance = cur; take = true; }
			else if (abs(distance) == abs(cur)) equal += 1;
		}
		if (take) { ta = q.a[x]; tb = q.b[x]; }
	}
	PairOutcome o{-1, -1, q.mq_a, q.mq_b, 0, 0, top > 0.0f};
	if (o.found) { o.wa = ta; o.wb = tb; o.equal = equal; o.dist = distance; }
	return o;
}

}  // extern "C++"

// ScoreBuffer::top1PE + CheckPairs (src/ScoreBuffer.cpp:368-502) for one pair; `a` = the mate whose scores arrive last
// (the odd read id: "read"), `b` = its mate.  Candidates are (pair index) lists into loc/score.
static void select_pair(ngm_mapper *m, const long dist_sum, const long dist_count, int *dist_out, uint32_t base_a, uint32_t cnt_a, int len_a, uint32_t base_b, uint32_t cnt_b, int len_b,
		const uint32_t *loc, const uint32_t *sv, const float *score, const uint32_t *rank, int *win_a, int *win_b, int *mq_a, int *mq_b, int *equal_out, bool *found,
		PairTies *ties = nullptr) {
	if (ties) *ties = PairTies{};
	if (cnt_a == 1 && cnt_b == 1) {  // the common case: one candidate per mate
		*mq_a = *mq_b = 60;
		const uint64_t l1 = loc[base_a], l2 = loc[base_b];
		const int cur = (int) ((l2 > l1) ? l2 - l1 + (uint64_t) len_b : l1 - l2 + (uint64_t) len_a);
		const int min_d1 = m->prm.min_insert_size, max_d1 = m->prm.max_insert_size > 0 ? m->prm.max_insert_size : INT_MAX;
		const float ps = score[base_a] + score[base_b];
		*found = cur > min_d1 && cur < max_d1 && ps > 0.0f;
		if (*found) { *dist_out = cur; *win_a = (int) base_a; *win_b = (int) base_b; *equal_out = 0; }
		return;
	}
	const int min_d = m->prm.min_insert_size, max_d = m->prm.max_insert_size > 0 ? m->prm.max_insert_size : INT_MAX;
	float top = 0.0f;
	int distance = 0, equal = 0, ta = -1, tb = -1, n_combo = 0;
	float combo_s[64];  // pair score, insert size and candidates of every pair inside the insert-size window
	int combo_d[64], combo_a[64], combo_b[64];
	const int avg = (int) (dist_sum / std::max(1L, dist_count));
	walk_pair(m, base_a, cnt_a, len_a, base_b, cnt_b, len_b, loc, sv, score, rank, mq_a, mq_b, [&](float ps, int cur, int ia, int ib) {
		if (n_combo < 64) { combo_s[n_combo] = ps; combo_d[n_combo] = cur; combo_a[n_combo] = ia; combo_b[n_combo] = ib; }
		++n_combo;
		bool take = false;
		if (ps > top * 1.00f) { top = ps; distance = cur; take = true; }
		else if (ps == top) {
			if (abs(distance - avg) > abs(cur - avg)) { top = ps; distance = cur; take = true; }
			else if (abs(distance) == abs(cur)) equal += 1;
		}
		if (take) { ta = ia; tb = ib; }
	});
	*found = top > 0.0f;
	if (*found) { *dist_out = distance; *win_a = ta; *win_b = tb; *equal_out = equal; }
	// Which state does the outcome depend on besides the scores?  CheckPairs consults the running mean insert size (the
	// sequential state of the reference's CS thread) only when a pair's score equals the best so far, keeping the pair
	// closer to the mean; and it counts a pair as "equal" (NH / X0, never reset) when it ties with the current best in
	// score AND insert size.  So: no two in-window pairs of equal score -> the result is fixed.  Otherwise the winner is
	// the best-scoring pair closest to the mean, whatever the order, provided no two pairs share score and insert size
	// (`dup`: the first one in candidate order wins, later ones are counted) and the closest one is unique.
	if (ties && n_combo > 1) {
		if (n_combo > 64) { ties->equal_scores = ties->dup = true; ties->unique_closest = false; ties->dmin_top = min_d; ties->dmax_top = max_d; return; }
		int best_c = INT_MAX, n_best_c = 0, dmin = INT_MAX, dmax = 0;
		for (int x = 0; x < n_combo; ++x) {
			if (combo_s[x] == top) {
				const int cx = abs(combo_d[x] - avg);
				if (cx < best_c) { best_c = cx; n_best_c = 1; } else if (cx == best_c) ++n_best_c;
				dmin = std::min(dmin, combo_d[x]); dmax = std::max(dmax, combo_d[x]);
				if (*found) {
					if (ties->n_top < 8) { ties->top_d[ties->n_top] = combo_d[x]; ties->top_a[ties->n_top] = combo_a[x]; ties->top_b[ties->n_top] = combo_b[x]; }
					++ties->n_top;
				}
			}
			for (int y = x + 1; y < n_combo; ++y) if (combo_s[x] == combo_s[y]) { ties->equal_scores = true; if (combo_d[x] == combo_d[y]) ties->dup = true; }
		}
		ties->unique_closest = n_best_c <= 1;
		if (*found) { ties->dmin_top = dmin; ties->dmax_top = dmax; }
		if (ties->n_top > 8) ties->dup = true;  // too many to list: left to the exact sequential pass
	}
}

static int map_impl(ngm_mapper *m, int n, const char *reads, const void *d_reads_ext, ngm_hit *hits, char *cigars, char *mds, bool paired, SamCall *sam) {
	if (!m) return -22;
	// shared paired-end state: wait for this batch's turn before the running mean is read, pass it on when the batch is
	// done with it (also on EVERY early return, the argument checks below included: with a shared ngm_pair_state a batch that
	// never passes its turn on would block all later ones)
	struct PairTurn {
		ngm_mapper *m; bool active, held = false;
		void acquire() {
			if (!active || held) return;
			std::unique_lock<std::mutex> lk(m->ps->mu);
			m->ps->cv.wait(lk, [&] { return m->ps->next == m->batch_seq; });
			m->pair_dist_sum = m->ps->dist_sum; m->pair_dist_count = m->ps->dist_count; m->scores_so_far = m->ps->scores_so_far; m->reads_so_far = m->ps->reads_so_far;
			held = true;
		}
		void release() {  // the running mean is final for this batch: the next batch may read it (align + CIGAR of this one go on)
			if (!active || released) return;
			acquire();
			{ std::lock_guard<std::mutex> lk(m->ps->mu); m->ps->dist_sum = m->pair_dist_sum; m->ps->dist_count = m->pair_dist_count; m->ps->scores_so_far = m->scores_so_far; m->ps->reads_so_far = m->reads_so_far; m->ps->next = m->batch_seq + 1; }
			m->ps->cv.notify_all();
			released = true;
		}
		bool released = false;
		~PairTurn() { release(); }
	} pair_turn{m, paired && m->ps != nullptr};
	if (n < 0) return -22;
	if (paired && m->prm.topn > 1) { ngm::pipeline_set_error("Paired end mode with topn > 1 not yet supported."); return -38; }  // ScoreBuffer::topNPE
	if (paired && (n & 1)) { ngm::pipeline_set_error("paired-end batches need an even number of reads"); return -22; }
	if (n == 0) return 0;
	const ngm_ref *r = m->ref;
	DevGuard g(r->device);
	ngm_hip_ctx *eng = m->eng;
	const int q = m->prm.qry_max_len, c = m->prm.corridor, mode = m->prm.mode;
	const size_t str_stride = (size_t) 4 * std::max(1, q);
	for (auto &x : m->ms) x = 0.f;
	m->order_ms = 0.f;

	// reads already in HBM: alias them as the batch (no copy); otherwise upload
	ngm::DevBuf<uint8_t> own = m->d_reads;
	struct Restore { ngm_mapper *m; ngm::DevBuf<uint8_t> own; bool on; ~Restore() { if (on) m->d_reads = own; } } restore{m, own, d_reads_ext != nullptr};
	if (d_reads_ext) { m->d_reads.p = (uint8_t *) d_reads_ext; m->d_reads.cap = (size_t) n * q; }
	const bool host_timing = getenv("NGM_HIP_HOST_TIMING") != nullptr;
	const int alt_cigar = m->prm.bs_mapping ? NGM_ALT_BISULFITE : (m->prm.slam_seq ? NGM_ALT_SLAMSEQ : NGM_ALT_NONE);
	const int alt_dir = alt_cigar ? (paired ? 2 : 1) : 0;   // the pairs' direction bits (gather_pairs_kernel): score tables, conversion rule
	auto now = [] { return std::chrono::steady_clock::now(); };
	auto tp0 = now();
	double t_stage[6] = {0, 0, 0, 0, 0, 0};
	auto lap = [&](int k) { auto t = now(); t_stage[k] += std::chrono::duration<double, std::milli>(t - tp0).count(); tp0 = t; };
	MAP_HIP_TRY(hipEventRecord(m->ev[0], m->st));
	if (!d_reads_ext) if (int rc = upload_reads(m, n, reads)) return rc;
	if (sam) {
		// what the SAM records need beyond the reads: qualities, names (travel while the search runs)
		if (!m->sam_ready || (!paired) != (!m->sam_opt.paired) || m->prm.topn > 1) { ngm::pipeline_set_error("ngm_mapper_map_sam: call ngm_mapper_set_sam_options first (single alignments only; paired as configured)"); return -22; }
		if ((unsigned long long) n * (unsigned long long) (2 * q + 1024) + sam->names_bytes >= 0xFFFFFFFFull) { ngm::pipeline_set_error("ngm_mapper_map_sam: the text of %d reads may exceed the 32-bit offsets of a batch: use smaller batches", n); return -75; }
		if (m->d_sam_quals.reserve((size_t) n * q) || m->d_sam_names.reserve(sam->names_bytes + 16) || m->d_sam_meta.reserve(n) || m->p_sam_hits.reserve(n) || m->p_sam_refs.reserve(n) ||
				m->d_sam_hits.reserve(n) || m->d_sam_refs.reserve(n)) { ngm::pipeline_set_error("out of memory (SAM stage)"); return -12; }
		MAP_HIP_TRY(hipMemcpyAsync(m->d_sam_quals.p, sam->quals, (size_t) n * q, hipMemcpyHostToDevice, m->st));
		if (sam->names_bytes) MAP_HIP_TRY(hipMemcpyAsync(m->d_sam_names.p, sam->names, sam->names_bytes, hipMemcpyHostToDevice, m->st));
		MAP_HIP_TRY(hipMemcpyAsync(m->d_sam_meta.p, sam->meta, (size_t) n * sizeof(ngm::SamMeta), hipMemcpyHostToDevice, m->st));
		hits = m->p_sam_hits.p;
	}
	GpuStage stage_cs(m);
	m->cs_paired = paired;
	if (int rc = run_cs(m, n, &stage_cs)) return rc;
	MAP_HIP_TRY(hipEventRecord(m->ev[1], m->st));
	const uint64_t np = m->n_cand;
	lap(0);
	if (const char *dump = getenv("NGM_HIP_DUMP_COUNTS")) {  // diagnostics: candidates per read, appended batch after batch
		if (int rc = cs_host_arrays(m)) return rc;
		if (FILE *f = fopen(dump, "ab")) { fwrite(m->h_count.data(), 4, (size_t) n, f); fclose(f); }
	}

	if (m->p_winner.reserve(n) || m->p_mapq.reserve(n) || m->p_nbest.reserve(n) || m->p_best.reserve(n) || m->p_loc.reserve(np + 1) ||
			m->p_sv.reserve(np + 1) || (paired && m->p_scores.reserve(np + 1))) { ngm::pipeline_set_error("out of pinned host memory"); return -12; }
	uint32_t *h_winner = m->p_winner.p, *h_loc = m->p_loc.p, *h_sv = m->p_sv.p;
	int32_t *h_mapq = m->p_mapq.p, *h_nbest = m->p_nbest.p;
	float *h_best = m->p_best.p, *h_scores = m->p_scores.p;
	if (np == 0) { if (int rc = cs_host_arrays(m)) return rc; for (int i = 0; i < n; ++i) { h_winner[i] = 0xFFFFFFFFu; h_mapq[i] = 0; h_nbest[i] = 0; h_best[i] = 0.f; } }
	std::vector<int> pair_flags(n, 0);
	auto reference_buffer_walk = [&]() {   // inside the batch's turn (sequential state)
		// A pair the reference LOSES (round 5; the "early top1SE" earlier rounds described does not exist -- MappedRead::Calculated
		// starts at -1, src/MappedRead.cpp:14, so ScoreBuffer.cpp:196 is false while the mate has not been searched).  CS::RunBatch
		// hands a read to its ScoreBuffer right after the search (CS.cpp:436); when the last score of a pair's FIRST mate fills the
		// buffer exactly (ScoreBuffer.cpp:519-523: DoRun), its scores are complete but the mate's Calculated is still -1: nothing is
		// selected.  If the mate then turns out to have NO candidates it goes to the writer alone (CS.cpp:326-329) and no later
		// DoRun ever looks at the first mate again: neither read is written ("(2 discarded)" in the reference's summary;
		// profiles/r05_reference_lost_pair_experiment.txt).  Deterministic at -t 1 for a known buffer size (SeqAn personality:
		// 1 024, src/seqan/EndToEndAffine.h:44-46); mirrored when the caller names that size (ngm_mapper_set_reference_score_buffer).
		// Sequential like the running mean: part of the batch's turn.
		uint64_t tot = m->scores_so_far, at = m->reads_so_far;
		// (the next flush position is carried along: a 64-bit remainder per pair made this loop the longest part of the turn)
		const uint64_t rb = m->ref_cs_batch > 0 ? (uint64_t) m->ref_cs_batch : 0;
		const uint64_t sb = m->ref_score_buffer > 0 ? (uint64_t) m->ref_score_buffer : 0;
		uint64_t next_flush = rb ? (at + rb - 1) / rb * rb : ~0ull;
		uint64_t fill = sb ? tot % sb : 0;   // entries in the reference's score buffer
		for (int pi = 0; pi < n / 2; ++pi, at += 2) {
			while (at > next_flush) next_flush += rb;   // (an odd batch size: flush positions between two pairs never match `at`)
			if (at == next_flush) { tot = 0; fill = 0; next_flush += rb; }  // the reference flushes its score buffer at the end of a CS batch (CS.cpp:488-500)
			const uint32_t c1 = m->h_count[2 * pi], c2 = m->h_count[2 * pi + 1];
			tot += (uint64_t) c1 + c2;
			if (!sb) continue;
			fill += c1;
			const bool full_at_first_mate = c1 > 0 && fill % sb == 0;
			fill = (fill + c2) % sb;
			if (full_at_first_mate && c2 == 0) {
				++m->lost_pairs;
				pair_flags[2 * pi] = pair_flags[2 * pi + 1] = NGM_PAIR_LOST;
				h_winner[2 * pi] = h_winner[2 * pi + 1] = 0xFFFFFFFFu;
			}
		}
		m->scores_so_far = tot; m->reads_so_far = at;
	};
	if (np > 0) {
		// ---- score stage: all candidates of the batch in one BatchScore -------------------------------------
		if (m->d_pair_read.reserve(np) || m->d_scores.reserve(np) || m->d_winner.reserve(n) || m->d_mapq.reserve(n) || m->d_nbest.reserve(n) ||
				m->d_best.reserve(n)) { ngm::pipeline_set_error("out of device memory (score stage)"); return -12; }
		if (int rc = ngm::engine_reserve(eng, (int) np)) { ngm::pipeline_set_error("%s", ngm_hip_last_error(eng)); return rc; }
		stage_cs.acquire();
		hipLaunchKernelGGL(ngm::expand_pairs_kernel, dim3(n), dim3(64), 0, m->st, n, m->d_cand_base.p, m->d_cand_count.p, m->d_pair_read.p);
		const int nb = (int) ((np + ngm::kSlots - 1) / ngm::kSlots);
		ngm::WindowGeom Gs{r->n_bases - 1, ((q + c) | 1) + 1, c >> 1};  // refMaxLen of ScoreBuffer.h:112
		hipLaunchKernelGGL(ngm::gather_pairs_kernel, dim3(nb), dim3(256), 0, m->st, m->d_reads.p, m->d_read_len.p, q, r->d_genome, Gs,
				m->d_pair_read.p, m->d_out_loc.p, m->d_out_sv.p, (int) np, eng->RW, eng->FW, eng->packed.p, eng->lens.p, eng->blk_rows.p, alt_dir);
		MAP_HIP_TRY(hipGetLastError());
		MAP_HIP_TRY(hipEventRecord(m->ev[2], m->st));
		if (int rc = ngm::engine_score_packed(eng, mode, (int) np, m->d_scores.p, m->st)) { ngm::pipeline_set_error("%s", ngm_hip_last_error(eng)); return rc; }
		MAP_HIP_TRY(hipEventRecord(m->ev[3], m->st));
		hipLaunchKernelGGL(ngm::select_top1_kernel, dim3((n + 255) / 256), dim3(256), 0, m->st, n, m->d_cand_base.p, m->d_cand_count.p,
				m->d_scores.p, m->d_out_loc.p, m->d_out_sv.p, m->d_winner.p, m->d_mapq.p, m->d_nbest.p, m->d_best.p);
		MAP_HIP_TRY(hipGetLastError());
		static const bool pair_gpu = !getenv("NGM_HIP_HOST_PAIR_PASS1");
		const bool pe_select = paired && !m->fast_pairing;   // --fast-pairing: the mates are selected single-end, the writer checks the pair (AlignmentBuffer.cpp:176-199)
		const bool simple_on_gpu = pe_select && pair_gpu && m->prm.strata == 0;   // (--strata touches NH of every pair: host)
		// ... and the pairs with choices: everything the scores alone decide (pair_device.h); NGM_HIP_HOST_PAIR_CHOICE=1 keeps the host's walk
		// (which side walks the pairs with choices depends on the workload: with ~1.2 candidates per read -- a genome without a heavy tail --
		// the pairs with choices have two or three candidates, the host's pass 1 is 1.4 ms on the pool beside the other instance's kernels,
		// and the two extra kernel launches cost the step more than they save: 50.0-50.3 M reads/s against 51.8 M on one box,
		// profiles/r05_main_leg_ab_vs_r04.txt; from 3 candidates per read on the GPU takes them.  NGM_HIP_HOST_PAIR_CHOICE=1 / NGM_HIP_GPU_PAIR_CHOICE=1 force a side)
		static const bool pair_choice_host = getenv("NGM_HIP_HOST_PAIR_CHOICE") != nullptr, pair_choice_force = getenv("NGM_HIP_GPU_PAIR_CHOICE") != nullptr;
		const bool choice_on_gpu = simple_on_gpu && !pair_choice_host && (pair_choice_force || np >= 3ull * (uint64_t) n);
		if (simple_on_gpu) {
			// pairs whose mates have one candidate each (most): settled here, the host only sums their insert sizes; the others are sorted
			// into two lists for pair_choice_kernel (mates with up to 64 candidates each: one wave per pair; the rest: a workgroup)
			const size_t npairs = (size_t) n / 2;
			if (m->d_pair_info.reserve(npairs + 1) || m->p_pair_info.reserve(npairs + 1)) { ngm::pipeline_set_error("out of memory (pair selection)"); return -12; }
			if (choice_on_gpu) {
				if (m->d_pair_out.reserve(2 * npairs + 2) || m->d_pair_top.reserve(npairs + 1) || m->d_pair_tied_n.reserve(4) || m->p_pair_tied_n.reserve(4) || m->d_pair_list.reserve(2 * npairs + 2)) {
					ngm::pipeline_set_error("out of memory (pair selection)"); return -12; }
				MAP_HIP_TRY(hipMemsetAsync(m->d_pair_tied_n.p, 0, 16, m->st));   // [0] tied pairs, [1] small pairs, [2] large pairs
			}
			uint32_t *counts = choice_on_gpu ? m->d_pair_tied_n.p + 1 : nullptr;
			hipLaunchKernelGGL(ngm::pair_simple_kernel, dim3((n / 2 + 255) / 256), dim3(256), 0, m->st, n / 2, m->d_cand_base.p, m->d_cand_count.p, m->d_scores.p,
					m->d_out_loc.p, m->d_read_len.p, m->prm.min_insert_size, m->prm.max_insert_size > 0 ? m->prm.max_insert_size : INT_MAX, m->d_mapq.p, m->d_nbest.p, m->d_pair_info.p,
					m->d_pair_list.p, m->d_pair_list.p + npairs, counts);
			MAP_HIP_TRY(hipGetLastError());
			if (choice_on_gpu) {
				// persistent workgroups over the two lists (their lengths stay on the device); entries of the small pairs at out[0 ..), of the large
				// ones at out[npairs ..) (2^30 + ... in the numbering pair_simple_kernel puts into d_pair_info)
				const int min_d = m->prm.min_insert_size, max_d = m->prm.max_insert_size > 0 ? m->prm.max_insert_size : INT_MAX;
				const float cutoff = m->prm.pair_score_cutoff > 0 ? m->prm.pair_score_cutoff : 0.9f;
				hipLaunchKernelGGL((ngm::pair_choice_kernel<64, 64>), dim3((unsigned) std::min<size_t>(npairs, 8192)), dim3(64), 0, m->st, (const uint32_t *) m->d_pair_list.p, (const uint32_t *) counts,
						m->d_cand_base.p, m->d_cand_count.p, m->d_scores.p, m->d_out_loc.p, m->d_read_len.p, min_d, max_d, cutoff, m->d_pair_out.p, m->d_pair_top.p, m->d_pair_tied_n.p, (uint32_t) npairs);
				hipLaunchKernelGGL((ngm::pair_choice_kernel<ngm::kPairThreads, ngm::kPairCap>), dim3((unsigned) std::min<size_t>(npairs, 1024)), dim3(ngm::kPairThreads), 0, m->st,
						(const uint32_t *) (m->d_pair_list.p + npairs), (const uint32_t *) (counts + 1), m->d_cand_base.p, m->d_cand_count.p, m->d_scores.p, m->d_out_loc.p, m->d_read_len.p, min_d, max_d, cutoff,
						m->d_pair_out.p + npairs, m->d_pair_top.p, m->d_pair_tied_n.p, (uint32_t) npairs);
				MAP_HIP_TRY(hipGetLastError());
				MAP_HIP_TRY(hipMemcpyAsync(m->p_pair_tied_n.p, m->d_pair_tied_n.p, 16, hipMemcpyDeviceToHost, m->st));
			}
			MAP_HIP_TRY(hipMemcpyAsync(m->p_pair_info.p, m->d_pair_info.p, (size_t) (n / 2) * 4, hipMemcpyDeviceToHost, m->st));
		}
		MAP_HIP_TRY(hipEventRecord(m->ev[4], m->st));
		stage_cs.kernels_done();
		MAP_HIP_TRY(hipMemcpyAsync(h_winner, m->d_winner.p, (size_t) n * 4, hipMemcpyDeviceToHost, m->st));
		MAP_HIP_TRY(hipMemcpyAsync(h_mapq, m->d_mapq.p, (size_t) n * 4, hipMemcpyDeviceToHost, m->st));
		MAP_HIP_TRY(hipMemcpyAsync(h_nbest, m->d_nbest.p, (size_t) n * 4, hipMemcpyDeviceToHost, m->st));
		MAP_HIP_TRY(hipMemcpyAsync(h_best, m->d_best.p, (size_t) n * 4, hipMemcpyDeviceToHost, m->st));
		MAP_HIP_TRY(hipMemcpyAsync(h_loc, m->d_out_loc.p, np * 4, hipMemcpyDeviceToHost, m->st));
		MAP_HIP_TRY(hipMemcpyAsync(h_sv, m->d_out_sv.p, np * 4, hipMemcpyDeviceToHost, m->st));
		if (paired) MAP_HIP_TRY(hipMemcpyAsync(h_scores, m->d_scores.p, np * 4, hipMemcpyDeviceToHost, m->st));
		stage_cs.done_after(m->ev[4]);
		MAP_HIP_TRY(hipStreamSynchronize(m->st));
		if (int rc = cs_host_arrays(m)) return rc;
		if (choice_on_gpu) {   // pair_choice_kernel's entries (the pairs with choices only) and the tied pairs' best-scoring combinations
			const size_t npairs = (size_t) n / 2;
			const size_t n_small = std::min<size_t>(m->p_pair_tied_n.p[1], npairs), n_large = std::min<size_t>(m->p_pair_tied_n.p[2], npairs), nt = std::min<size_t>(m->p_pair_tied_n.p[0], npairs);
			if (m->p_pair_out.reserve(2 * npairs + 2) || m->p_pair_top.reserve(nt + 1)) { ngm::pipeline_set_error("out of pinned host memory"); return -12; }
			if (n_small) MAP_HIP_TRY(hipMemcpyAsync(m->p_pair_out.p, m->d_pair_out.p, n_small * sizeof(ngm::PairOut), hipMemcpyDeviceToHost, m->st));
			if (n_large) MAP_HIP_TRY(hipMemcpyAsync(m->p_pair_out.p + npairs, m->d_pair_out.p + npairs, n_large * sizeof(ngm::PairOut), hipMemcpyDeviceToHost, m->st));
			if (nt) MAP_HIP_TRY(hipMemcpyAsync(m->p_pair_top.p, m->d_pair_top.p, nt * sizeof(ngm::PairTop), hipMemcpyDeviceToHost, m->st));
			MAP_HIP_TRY(hipStreamSynchronize(m->st));
		}
		lap(1);
		static const bool position_order = getenv("NGM_HIP_POSITION_ORDER") != nullptr;
		if ((!paired || m->fast_pairing) && m->prm.topn <= 1 && !position_order) {
			// several candidates share the best score: the reference keeps the first one in ITS candidate order
			// (ScoreBuffer::top1SE over CollectResultsStd's rList order); replay the votes of just those reads
			std::vector<uint32_t> tied;
			for (int i = 0; i < n; ++i) if (h_nbest[i] != 1 && m->h_count[i] > 1) tied.push_back((uint32_t) i);  // 0: no positive score, the first candidate is kept
			if (!tied.empty()) {
				uint32_t *h_rank = nullptr;
				if (int rc = candidate_order(m, tied, np, &h_rank)) return rc;
				if (m->p_scores.reserve(np + 1)) { ngm::pipeline_set_error("out of pinned host memory"); return -12; }
				h_scores = m->p_scores.p;
				MAP_HIP_TRY(hipMemcpy(h_scores, m->d_scores.p, np * 4, hipMemcpyDeviceToHost));
				for (uint32_t i : tied) {
					const uint32_t b = m->h_base[i], cnt = m->h_count[i];
					float best = h_scores[b];
					for (uint32_t c = 1; c < cnt; ++c) best = std::max(best, h_scores[b + c]);
					uint32_t pick = 0xFFFFFFFFu, pick_rank = ngm::kCsOrderUnknown;
					bool known = true;
					for (uint32_t c = 0; c < cnt; ++c) if (h_scores[b + c] == best || !(best > 0.0f)) {
						if (h_rank[b + c] == ngm::kCsOrderUnknown) { known = false; break; }
						if (h_rank[b + c] < pick_rank) { pick_rank = h_rank[b + c]; pick = b + c; }
					}
					// (without a positive score top1SE keeps the FIRST candidate, whatever its score: AS:i is that candidate's -- end-to-end mode)
					if (known && pick != 0xFFFFFFFFu) { h_winner[i] = pick; h_best[i] = h_scores[pick]; }
				}
			}
		}
		if (pe_select) {
			// Pairs in input order.  The running mean insert size (tie-break between equally scoring pairs only) is sequential state
			// of one CS thread in the reference (pairDistSum / pairDistCount, ScoreBuffer.h:90).  Everything that does not depend on
			// it happens outside this batch's turn for that state -- on the GPU (pair_simple_kernel, pair_choice_kernel) or in
			// parallel on the host -- and the turn itself only scans what is left (NGM_HIP_HOST_THREADS=1: strictly sequential).
			const bool pe_strata = m->prm.strata != 0;
			auto commit = [&](int ra, int rb, bool found, int wa, int wb, int mqa, int mqb, int equal) {
				if (found && pe_strata && equal > 0) {  // "To many equal scoring positions": both mates unmapped (ScoreBuffer.cpp:437-446)
					h_winner[ra] = h_winner[rb] = 0xFFFFFFFFu;
					h_mapq[ra] = h_mapq[rb] = 0;
					pair_flags[ra] = pair_flags[rb] = NGM_PAIR_SELECTED;
				} else if (found) {
					h_winner[ra] = (uint32_t) wa; h_winner[rb] = (uint32_t) wb;
					h_mapq[ra] = mqa; h_mapq[rb] = mqb;
					h_nbest[ra] = h_nbest[rb] = equal;
					h_best[ra] = h_scores[wa]; h_best[rb] = h_scores[wb];
					pair_flags[ra] = pair_flags[rb] = NGM_PAIR_SELECTED;
				} else {
					pair_flags[ra] = pair_flags[rb] = NGM_PAIR_FAILED;  // no pair inside the window: single-end selection stands
				}
			};
			auto len_of = [&](int rd) { return (int) strnlen(reads + (size_t) rd * q, q); };
			auto run_pair = [&](int pi, long sum, long cnt, const uint32_t *rank, int *wa, int *wb, int *mqa, int *mqb, int *equal, int *dist, bool *found, PairTies *ties) {
				const int rb = 2 * pi, ra = 2 * pi + 1;
				select_pair(m, sum, cnt, dist, m->h_base[ra], m->h_count[ra], len_of(ra), m->h_base[rb], m->h_count[rb], len_of(rb), h_loc, h_sv, h_scores, rank, wa, wb, mqa, mqb, equal, found, ties);
			};
			// Pass 1: every pair whose result depends on the scores alone -- nearly all of them -- is settled.  The others ("tied":
			// equally scoring pairs inside the window) depend on the running mean insert size, and some also on the candidate order.
			// gap_*: insert sizes / number of the pairs selected since the previous tied pair; dist: this pair's once it is closed;
			// seq: its in-window combinations in the reference's order (pass 3), or -1
			struct Tied { int pi; bool found, dup, open; int dmin, dmax, mqa, mqb; long avg_lo, avg_hi; int n_top, top_d[8], top_a[8], top_b[8]; long gap_sum, gap_cnt; int dist; int seq; };
			auto tq0 = now(); double tq[5] = {0, 0, 0, 0, 0};
			auto qlap = [&](int k) { auto t = now(); tq[k] += std::chrono::duration<double, std::milli>(t - tq0).count(); tq0 = t; };
			std::mutex tied_mu;
			std::vector<Tied> tied;
			std::vector<uint32_t> se_tied;  // mates selected single-end (no pair in the window / mate without candidates) whose best score is shared
			struct Chunk { int plo; std::vector<Tied> tied; std::vector<uint32_t> se; long tail_sum, tail_cnt; };
			std::vector<Chunk> chunks;
			auto se_check = [&](std::vector<uint32_t> &out, int i) { if (h_nbest[i] != 1 && m->h_count[i] > 1) out.push_back((uint32_t) i); };
			const ngm::PairOut *h_po = choice_on_gpu ? m->p_pair_out.p : nullptr;
			const ngm::PairTop *h_pt = choice_on_gpu ? m->p_pair_top.p : nullptr;
			std::atomic<long> n_host_walk{0};
			parallel_for(n / 2, [&](int plo, int phi) {
				std::vector<Tied> local;
				std::vector<uint32_t> local_se;
				long gsum = 0, gcnt = 0, walked = 0;
				for (int pi = plo; pi < phi; ++pi) {
					const int rb = 2 * pi, ra = 2 * pi + 1;
					const int pinfo = simple_on_gpu ? m->p_pair_info.p[pi] : -1;
					if (simple_on_gpu && pinfo >= 0) {   // one candidate per mate: pair_simple_kernel has settled it
						const int info = m->p_pair_info.p[pi];
						if (info & 1) { pair_flags[ra] = pair_flags[rb] = NGM_PAIR_SELECTED; gsum += info >> 1; ++gcnt; }
						else pair_flags[ra] = pair_flags[rb] = NGM_PAIR_FAILED;
						continue;
					}
					if (m->h_count[ra] == 0 || m->h_count[rb] == 0) { se_check(local_se, rb); se_check(local_se, ra); continue; }  // top1SE for the mate that has candidates (ScoreBuffer.cpp:204-209)
					const ngm::PairOut *pe = nullptr;
					if (h_po && pinfo <= -2) { const uint32_t e = (uint32_t) (-2 - (long long) pinfo); pe = &h_po[e >= (1u << 30) ? (size_t) (n / 2) + (e - (1u << 30)) : e]; }
					if (pe && !(pe->flags & ngm::kPairHost)) {   // pair_choice_kernel: settled, or what the sequential passes need
						const ngm::PairOut &po = *pe;
						const bool found = (po.flags & ngm::kPairFound) != 0;
						const int mqa = (po.flags >> 8) & 255, mqb = (po.flags >> 16) & 255;
						if (po.flags & ngm::kPairTied) {
							Tied t{pi, found, (po.flags & ngm::kPairDup) != 0, true, po.dmin, po.dmax, mqa, mqb, 0, 0, std::min((po.flags >> 24) & 15, 8), {}, {}, {}, gsum, gcnt, 0, -1};
							gsum = gcnt = 0;
							const ngm::PairTop &pt = h_pt[po.tied_ix];
							for (int x = 0; x < t.n_top; ++x) { t.top_d[x] = pt.d[x]; t.top_a[x] = pt.a[x]; t.top_b[x] = pt.b[x]; }
							local.push_back(t);
							continue;
						}
						commit(ra, rb, found, po.wa, po.wb, mqa, mqb, 0);
						if (found) { gsum += po.dist; ++gcnt; } else { se_check(local_se, rb); se_check(local_se, ra); }
						continue;
					}
					++walked;
					int wa = -1, wb = -1, mqa = 0, mqb = 0, equal = 0, dist = 0;
					bool found = false;
					PairTies ties;
					run_pair(pi, 0, 1, nullptr, &wa, &wb, &mqa, &mqb, &equal, &dist, &found, &ties);
					if (ties.equal_scores) {
						Tied t{pi, found, ties.dup || pe_strata, true, ties.dmin_top, ties.dmax_top, mqa, mqb, 0, 0, std::min(ties.n_top, 8), {}, {}, {}, gsum, gcnt, 0, -1};
						gsum = gcnt = 0;
						for (int x = 0; x < t.n_top; ++x) { t.top_d[x] = ties.top_d[x]; t.top_a[x] = ties.top_a[x]; t.top_b[x] = ties.top_b[x]; }
						local.push_back(t);
						continue;
					}
					commit(ra, rb, found, wa, wb, mqa, mqb, equal);
					if (found) { gsum += dist; ++gcnt; } else { se_check(local_se, rb); se_check(local_se, ra); }
				}
				n_host_walk += walked;
				{ std::lock_guard<std::mutex> lk(tied_mu); chunks.push_back(Chunk{plo, std::move(local), std::move(local_se), gsum, gcnt}); }
			});
			std::sort(chunks.begin(), chunks.end(), [](const Chunk &x, const Chunk &y) { return x.plo < y.plo; });  // pairs in input order again
			long carry_sum = 0, carry_cnt = 0;  // selected pairs after the last tied pair so far
			for (Chunk &c : chunks) {
				if (!c.tied.empty()) { c.tied[0].gap_sum += carry_sum; c.tied[0].gap_cnt += carry_cnt; carry_sum = carry_cnt = 0; }
				carry_sum += c.tail_sum; carry_cnt += c.tail_cnt;
				tied.insert(tied.end(), c.tied.begin(), c.tied.end());
				se_tied.insert(se_tied.end(), c.se.begin(), c.se.end());
			}
			// a tied pair without a positive pair score fails whatever the mean: single-end selection for both mates
			for (Tied &t : tied) if (!t.found) {
				commit(2 * t.pi + 1, 2 * t.pi, false, -1, -1, 0, 0, 0);
				se_check(se_tied, 2 * t.pi); se_check(se_tied, 2 * t.pi + 1);
				t.open = false;
			}
			// Which tied pairs will stay open in the sequential pass?  Those whose equally scoring pairs also share the insert size
			// (`dup`: the candidate order decides) -- and those whose best-scoring pair closest to the mean is not the same unique one
			// over the range the mean can have when their turn comes.  That range is only known inside the turn (pass 2 carries exact
			// bounds); the mean of thousands of insert sizes hardly moves, so outside the turn pass 2 runs SPECULATIVELY from the last
			// PUBLISHED mean +- 3 -- bounds that contain the exact ones leave a superset of the exact pass's pairs open -- and the pairs
			// it leaves open have their candidate order replayed and their combinations listed (pass 3) before the turn begins.  A
			// pair that the exact pass 2 leaves open without having been picked here (the first batches of a run, while the mean still
			// moves by more than 3) is handled inside the turn, as every open pair was before round 5.
			auto closest_top = [](const Tied &t, long avg, bool *unique) {
				int best = 0, n_best = 0; long best_c = LONG_MAX;
				for (int x = 0; x < t.n_top; ++x) {
					const long cx = labs((long) t.top_d[x] - avg);
					if (cx < best_c) { best_c = cx; best = x; n_best = 1; } else if (cx == best_c) ++n_best;
				}
				*unique = n_best == 1;
				return best;
			};
			auto closed_between = [&](const Tied &t, long lo, long hi, int *which) {
				if (t.dup || t.n_top <= 0) return false;
				bool u_lo = false, u_hi = false;
				const int x_lo = closest_top(t, lo, &u_lo), x_hi = closest_top(t, hi, &u_hi);
				*which = x_lo;
				return x_lo == x_hi && u_lo && u_hi;
			};
			// pass 2 itself (see below), from a given state of the running mean: exact -- it commits what it closes -- or speculative
			auto pass2 = [&](long sum_lo, long sum_hi, long cnt, bool exact, std::vector<int> &left_open) {
				for (size_t x = 0; x < tied.size(); ++x) {
					Tied &t = tied[x];
					sum_lo += t.gap_sum; sum_hi += t.gap_sum; cnt += t.gap_cnt;
					const long a_lo = sum_lo / std::max(1L, cnt), a_hi = sum_hi / std::max(1L, cnt);
					if (exact) { t.avg_lo = a_lo; t.avg_hi = a_hi; }
					if (!t.found) continue;
					int which = 0;
					if (closed_between(t, a_lo, a_hi, &which)) {
						if (exact) {
							commit(2 * t.pi + 1, 2 * t.pi, true, t.top_a[which], t.top_b[which], t.mqa, t.mqb, 0);
							t.dist = t.top_d[which];
							t.open = false;
						}
						sum_lo += t.top_d[which]; sum_hi += t.top_d[which]; ++cnt;
						continue;
					}
					left_open.push_back((int) x);
					sum_lo += t.dmin; sum_hi += t.dmax; ++cnt;
				}
			};
			long pub_sum = m->pair_dist_sum, pub_cnt = m->pair_dist_count;
			if (m->ps && !pair_turn.held) { std::lock_guard<std::mutex> lk(m->ps->mu); pub_sum = m->ps->dist_sum; pub_cnt = m->ps->dist_count; }
			std::vector<int> picked;   // indices into `tied`
			if (pe_strata) { for (size_t x = 0; x < tied.size(); ++x) if (tied[x].found) picked.push_back((int) x); }
			else pass2(pub_sum - 3 * pub_cnt, pub_sum + 3 * pub_cnt, pub_cnt, false, picked);   // (bounds that contain the exact ones leave a superset of the exact pass's pairs open)
			std::vector<uint32_t> need;
			need.reserve(2 * picked.size() + se_tied.size());
			for (int x : picked) { need.push_back((uint32_t) (2 * tied[x].pi)); need.push_back((uint32_t) (2 * tied[x].pi + 1)); }
			need.insert(need.end(), se_tied.begin(), se_tied.end());
			qlap(0);
			uint32_t *h_rank_pe = nullptr;
			if (!need.empty() && !position_order) if (int rc = candidate_order(m, need, np, &h_rank_pe)) return rc;
			qlap(1);
			// Pass 3 (parallel, outside the turn): the in-window combinations of the picked pairs in the reference's order
			std::vector<PairSeq> seqs(picked.size());
			auto build_seq = [&](PairSeq &sq, int pi) {
				const int rb = 2 * pi, ra = 2 * pi + 1;
				walk_pair(m, m->h_base[ra], m->h_count[ra], len_of(ra), m->h_base[rb], m->h_count[rb], len_of(rb), h_loc, h_sv, h_scores, h_rank_pe, &sq.mq_a, &sq.mq_b,
						[&](float ps, int cur, int ia, int ib) { sq.ps.push_back(ps); sq.d.push_back(cur); sq.a.push_back(ia); sq.b.push_back(ib); });
			};
			parallel_for((int) picked.size(), [&](int lo, int hi) { for (int x = lo; x < hi; ++x) { build_seq(seqs[x], tied[picked[x]].pi); tied[picked[x]].seq = x; } }, 8);
			qlap(2);
			auto first_best = [&](uint32_t i) {  // ScoreBuffer::top1SE keeps the first of the equally best candidates
				const uint32_t b = m->h_base[i], cnt = m->h_count[i];
				if (!h_rank_pe) return;
				float best = h_scores[b];
				for (uint32_t c2 = 1; c2 < cnt; ++c2) best = std::max(best, h_scores[b + c2]);
				uint32_t pick = 0xFFFFFFFFu, pick_rank = ngm::kCsOrderUnknown;
				for (uint32_t c2 = 0; c2 < cnt; ++c2) if (h_scores[b + c2] == best || !(best > 0.0f)) {
					if (h_rank_pe[b + c2] == ngm::kCsOrderUnknown) return;
					if (h_rank_pe[b + c2] < pick_rank) { pick_rank = h_rank_pe[b + c2]; pick = b + c2; }
				}
				if (pick != 0xFFFFFFFFu) { h_winner[i] = pick; h_best[i] = h_scores[pick]; }
			};
			// ... but when both mates have candidates top1PE has already SORTED the arrays before it falls back to
			// top1SE (ScoreBuffer.cpp:373-376, 449-455): the first of the best is the head of that (unstable) sort
			auto first_sorted = [&](uint32_t i) {
				if (!h_rank_pe) return;
				// (also for the <= 16 candidates of a stable insertion sort: the head of the sorted array is the BEST score's first candidate --
				// without a positive score top1SE then keeps it, not the first candidate of the unsorted list: end-to-end mode)
				bool ranked = false;
				std::vector<uint32_t> v(m->h_count[i]);
				sort_like_reference(v.data(), m->h_base[i], m->h_count[i], h_loc, h_sv, h_scores, h_rank_pe, &ranked);
				if (ranked) { h_winner[i] = v[0]; h_best[i] = h_scores[v[0]]; }
			};
			// the single-end ties do not touch the mean: settled here, in parallel
			parallel_for((int) se_tied.size(), [&](int lo, int hi) { for (int x = lo; x < hi; ++x) { const uint32_t i = se_tied[x]; if (m->h_count[i ^ 1u] > 0) first_sorted(i); else first_best(i); } }, 64);
			// ---- this batch's turn for the running mean ----------------------------------------------------------------------
			pair_turn.acquire();
			reference_buffer_walk();
			// Pass 2 (sequential, cheap): the running mean at every tied pair, as bounds -- a tied pair that stays open contributes one
			// of the insert sizes of its best-scoring pairs.  Without pairs of equal score AND insert size the winner is the
			// best-scoring pair closest to the mean: the same unique winner at both bounds is the winner for every mean in between,
			// and its insert size keeps the bounds exact.  (With --strata a tied pair may contribute nothing at all; then every tied
			// pair simply waits for pass 4.)
			std::vector<int> late;   // left open without having been picked above
			if (!pe_strata) {
				std::vector<int> left_open;
				pass2(m->pair_dist_sum, m->pair_dist_sum, m->pair_dist_count, true, left_open);
				for (int x : left_open) if (tied[x].seq < 0) late.push_back(x);
			}
			qlap(3);
			size_t n_open = 0;
			for (const Tied &t : tied) n_open += t.open;
			if (!late.empty()) {
				std::vector<uint32_t> need_late;
				for (int x : late) { need_late.push_back((uint32_t) (2 * tied[x].pi)); need_late.push_back((uint32_t) (2 * tied[x].pi + 1)); }
				if (!position_order) if (int rc = candidate_order(m, need_late, np, &h_rank_pe)) return rc;
				const size_t s0 = seqs.size();
				seqs.resize(s0 + late.size());
				parallel_for((int) late.size(), [&](int lo, int hi) { for (int x = lo; x < hi; ++x) { build_seq(seqs[s0 + x], tied[late[x]].pi); tied[late[x]].seq = (int) (s0 + x); } }, 8);
			}
			// Pass 4 (sequential): the running mean in input order; the open pairs see exactly the reference's value
			for (const Tied &t : tied) {
				const int pi = t.pi;
				m->pair_dist_sum += t.gap_sum; m->pair_dist_count += t.gap_cnt;
				if (!t.open) { if (t.dist) { m->pair_dist_sum += t.dist; m->pair_dist_count += 1; } continue; }  // closed by pass 2 (or not found)
				const PairOutcome o = eval_pair_seq(seqs[t.seq], (int) (m->pair_dist_sum / std::max(1L, m->pair_dist_count)));
				const int rb = 2 * pi, ra = 2 * pi + 1;
				commit(ra, rb, o.found, o.wa, o.wb, o.mqa, o.mqb, o.equal);
				if (o.found && !(pe_strata && o.equal > 0)) { m->pair_dist_sum += o.dist; m->pair_dist_count += 1; }
				if (!o.found) { if (h_nbest[ra] != 1 && m->h_count[ra] > 1) first_sorted((uint32_t) ra); if (h_nbest[rb] != 1 && m->h_count[rb] > 1) first_sorted((uint32_t) rb); }
			}
			m->pair_dist_sum += carry_sum; m->pair_dist_count += carry_cnt;
			pair_turn.release();
			qlap(4);
			if (host_timing) fprintf(stderr, "[ngm-hip] pair selection: %zu tied pairs, %zu picked for the order replay + %zu single-end ties, %zu open in the turn, %zu of them late; %ld pairs walked on the host; "
					"ms: pass 1 %.2f | order replay %.2f | pass 3 %.2f | turn: pass 2 %.2f, late + pass 4 %.2f\n", tied.size(), picked.size(), se_tied.size(), n_open, late.size(), (long) n_host_walk, tq[0], tq[1], tq[2], tq[3], tq[4]);
		}
	}
	if (paired && (np == 0 || m->fast_pairing)) { pair_turn.acquire(); reference_buffer_walk(); pair_turn.release(); }   // (--fast-pairing / a batch without candidates: no top1PE turn above)
	if (paired && np > 0 && m->prm.strata)  // mates selected single-end (top1SE): several equally best candidates -> unmapped
		parallel_for(n, [&](int lo, int hi) { for (int i = lo; i < hi; ++i) if (!(pair_flags[i] & NGM_PAIR_SELECTED) && h_nbest[i] > 1) { h_winner[i] = 0xFFFFFFFFu; h_mapq[i] = 0; } }, 16384);
	const int topn = (!paired && m->prm.topn > 1) ? m->prm.topn : 1;
	std::vector<uint32_t> tn_pairs;  // topn > 1: per output entry the candidate (pair index) to align, or none
	if (!paired && np > 0 && (topn > 1 || m->prm.strata)) {
		if (topn == 1) {  // top1SE with strata: several equally best candidates -> unmapped (ScoreBuffer.cpp:259-276)
			for (int i = 0; i < n; ++i) if (h_nbest[i] > 1) { h_winner[i] = 0xFFFFFFFFu; h_mapq[i] = 0; }
		} else {
			// ScoreBuffer::topNSE: sort by score, report min(topn, candidates) (strata: the equally best ones only)
			if (m->p_scores.reserve(np + 1)) { ngm::pipeline_set_error("out of pinned host memory"); return -12; }
			h_scores = m->p_scores.p;
			MAP_HIP_TRY(hipMemcpy(h_scores, m->d_scores.p, np * 4, hipMemcpyDeviceToHost));
			tn_pairs.assign((size_t) n * topn, 0xFFFFFFFFu);
			// equally scoring candidates keep the reference's candidate order (the cut at -n is order dependent)
			uint32_t *h_rank_tn = nullptr;
			if (!getenv("NGM_HIP_POSITION_ORDER")) {
				std::vector<uint32_t> need;
				for (int i = 0; i < n; ++i) {
					const uint32_t b = m->h_base[i], cnt = m->h_count[i];
					bool eq = false;
					for (uint32_t x = 0; x + 1 < cnt && !eq; ++x) for (uint32_t y = x + 1; y < cnt; ++y) if (h_scores[b + x] == h_scores[b + y]) { eq = true; break; }
					if (eq) need.push_back((uint32_t) i);
				}
				if (!need.empty()) if (int rc = candidate_order(m, need, np, &h_rank_tn)) return rc;
			}
			parallel_for(n, [&](int lo, int hi) {
				std::vector<uint32_t> v;
				for (int i = lo; i < hi; ++i) {
					const uint32_t b = m->h_base[i], cnt = m->h_count[i];
					if (cnt == 0) continue;
					v.resize(cnt);
					std::iota(v.begin(), v.end(), b);
					// std::sort(sortLocationScore) over the reference's candidate order, as in select_pair
					auto by_place = [&](uint32_t x, uint32_t y) { return h_loc[x] != h_loc[y] ? h_loc[x] < h_loc[y] : (h_sv[x] & 1u) < (h_sv[y] & 1u); };
					bool ranked = h_rank_tn != nullptr;
					for (uint32_t x = b; ranked && x < b + cnt; ++x) ranked = h_rank_tn[x] != ngm::kCsOrderUnknown;
					if (ranked) {
						std::sort(v.begin(), v.end(), [&](uint32_t x, uint32_t y) { return h_rank_tn[x] != h_rank_tn[y] ? h_rank_tn[x] < h_rank_tn[y] : by_place(x, y); });
						std::sort(v.begin(), v.end(), [&](uint32_t x, uint32_t y) { return h_scores[x] > h_scores[y]; });
					} else {
						std::sort(v.begin(), v.end(), by_place);
						std::stable_sort(v.begin(), v.end(), [&](uint32_t x, uint32_t y) { return h_scores[x] > h_scores[y]; });
					}
					int ntop = 1;
					while (ntop < (int) cnt && h_scores[v[0]] == h_scores[v[ntop]]) ++ntop;
					h_nbest[i] = ntop;
					int ns = 0;
					if (ntop <= topn || !m->prm.strata) {
						ns = m->prm.strata ? ntop : std::min<int>((int) cnt, topn);
						int mq = 60;  // computeMQ(MappedRead*)
						if (cnt > 1) { const float bs = h_scores[v[0]], s2 = h_scores[v[1]]; mq = (bs > 0 && s2 >= 0) ? (int) ceilf(60.0f * (bs - s2) / bs) : 0; }
						h_mapq[i] = mq;
					} else {
						h_mapq[i] = 0;
					}
					for (int t = 0; t < ns; ++t) tn_pairs[(size_t) i * topn + t] = v[t];
					h_winner[i] = ns > 0 ? v[0] : 0xFFFFFFFFu;
					h_best[i] = h_scores[v[0]];
				}
			});
		}
	}
	stage_cs.done();
	lap(2);
	// ---- alignment stage: one pair per read that has a winner (AlignmentBuffer::DoRun) --------------------
	std::vector<uint32_t> a_read((size_t) n * topn), a_loc((size_t) n * topn), a_sv((size_t) n * topn), a_out((size_t) n * topn), a_pair((size_t) n * topn);
	int na = 0;
	if (topn == 1 || np == 0) {
		// winners, compacted in read order: count per slice, prefix, fill (the gathers through h_winner miss the caches)
		const int slices = std::max(1, std::min(256, n / 4096));
		std::vector<int> first(slices + 1, 0);
		parallel_for(slices, [&](int lo, int hi) {
			for (int s2 = lo; s2 < hi; ++s2) {
				const int i0 = (int) ((long long) n * s2 / slices), i1 = (int) ((long long) n * (s2 + 1) / slices);
				int cnt = 0;
				for (int i = i0; i < i1; ++i) cnt += h_winner[i] != 0xFFFFFFFFu;
				first[s2 + 1] = cnt;
			}
		}, 1);
		for (int s2 = 0; s2 < slices; ++s2) first[s2 + 1] += first[s2];
		na = first[slices];
		parallel_for(slices, [&](int lo, int hi) {
			for (int s2 = lo; s2 < hi; ++s2) {
				const int i0 = (int) ((long long) n * s2 / slices), i1 = (int) ((long long) n * (s2 + 1) / slices);
				int at = first[s2];
				for (int i = i0; i < i1; ++i) if (h_winner[i] != 0xFFFFFFFFu) {
					a_read[at] = (uint32_t) i; a_out[at] = (uint32_t) i; a_pair[at] = h_winner[i]; a_loc[at] = h_loc[h_winner[i]]; a_sv[at] = h_sv[h_winner[i]]; ++at;
				}
			}
		}, 1);
	} else {
		for (int i = 0; i < n; ++i) for (int t = 0; t < topn; ++t) {
			const uint32_t w = tn_pairs[(size_t) i * topn + t];
			if (w == 0xFFFFFFFFu) break;
			a_read[na] = i; a_out[na] = (uint32_t) (i * topn + t); a_pair[na] = w; a_loc[na] = h_loc[w]; a_sv[na] = h_sv[w]; ++na;
		}
	}
	const int rs = ngm::run_stride(q, c);
	if (m->p_rec.reserve((size_t) na * 8 + 8)) { ngm::pipeline_set_error("out of pinned host memory"); return -12; }
	int32_t *h_rec = m->p_rec.p;
	uint16_t *h_runs = nullptr;
	const int align_buf_len = (q + c) | 2;  // AlignmentBuffer.h:67: (qry_max_len + corridor) | 1 + 1
	static const bool dev_strings = !getenv("NGM_HIP_HOST_CIGAR");
	uint64_t str_base = 0;   // bytes of the device's CIGAR / MD stream (host-built strings of the SAM stage go behind them)
	GpuStage stage_align(m, 1, false);
	if (na > 0) {
		if (m->d_a_read.reserve(na) || m->d_a_loc.reserve(na) || m->d_a_sv.reserve(na) || m->d_records.reserve((size_t) na * 8) ||
				m->d_runs.reserve((size_t) na * rs)) { ngm::pipeline_set_error("out of device memory (align stage)"); return -12; }
		if (int rc = ngm::engine_reserve(eng, na)) { ngm::pipeline_set_error("%s", ngm_hip_last_error(eng)); return rc; }
		MAP_HIP_TRY(hipMemcpyAsync(m->d_a_read.p, a_read.data(), (size_t) na * 4, hipMemcpyHostToDevice, m->st));
		MAP_HIP_TRY(hipMemcpyAsync(m->d_a_loc.p, a_loc.data(), (size_t) na * 4, hipMemcpyHostToDevice, m->st));
		MAP_HIP_TRY(hipMemcpyAsync(m->d_a_sv.p, a_sv.data(), (size_t) na * 4, hipMemcpyHostToDevice, m->st));
		MAP_HIP_TRY(hipStreamSynchronize(m->st));   // (the uploads travel outside the stage lock)
		stage_align.acquire();
		MAP_HIP_TRY(hipEventRecord(m->ev[5], m->st));
		ngm::WindowGeom Ga{r->n_bases - 1, align_buf_len, c >> 1};
		hipLaunchKernelGGL(ngm::gather_pairs_kernel, dim3((na + ngm::kSlots - 1) / ngm::kSlots), dim3(256), 0, m->st, m->d_reads.p, m->d_read_len.p, q,
				r->d_genome, Ga, m->d_a_read.p, m->d_a_loc.p, m->d_a_sv.p, na, eng->RW, eng->FW, eng->packed.p, eng->lens.p, eng->blk_rows.p, alt_dir);
		MAP_HIP_TRY(hipGetLastError());
		MAP_HIP_TRY(hipEventRecord(m->ev[6], m->st));
		const bool was_prof = eng->profiling;
		eng->profiling = true;  // brackets DP vs traceback with eng->ev[2]
		if (int rc = ngm::engine_align_packed(eng, mode, na, m->d_records.p, m->d_runs.p, rs, m->st)) { eng->profiling = was_prof; ngm::pipeline_set_error("%s", ngm_hip_last_error(eng)); return rc; }
		eng->profiling = was_prof;
		MAP_HIP_TRY(hipEventRecord(m->ev[7], m->st));
		if (m->d_runs_c.reserve((size_t) na * rs)) { ngm::pipeline_set_error("out of device memory (runs)"); return -12; }
		MAP_HIP_TRY(hipMemsetAsync(m->d_total.p, 0, 8, m->st));
		hipLaunchKernelGGL(ngm::compact_runs_kernel, dim3((na + 255) / 256), dim3(256), 0, m->st, na, m->d_records.p, m->d_runs.p, rs,
				m->d_runs_c.p, m->d_total.p);
		MAP_HIP_TRY(hipGetLastError());
		unsigned long long n_runs_total = 0, n_str_total = 0;
		if (!dev_strings) { MAP_HIP_TRY(hipEventRecord(m->ev[8], m->st)); stage_align.kernels_done(); }   // (behind the stage's last kernel)
		// CIGAR / MD / NM / identity on the GPU (cigar_device.h); NGM_HIP_HOST_CIGAR=1 keeps the host builders (tests)
		if (dev_strings) {
			const unsigned long long scap = (unsigned long long) na * 96ull + 4096ull;
			if (m->d_cigout.reserve(na) || m->d_str.reserve(scap) || m->p_cigout.reserve(na)) { ngm::pipeline_set_error("out of memory (CIGAR strings)"); return -12; }
			MAP_HIP_TRY(hipMemsetAsync(m->d_total.p + 8, 0, 8, m->st));
			const bool affine = m->prm.personality == NGM_PERSONALITY_AFFINE;
			if (affine) hipLaunchKernelGGL(ngm::cigar_strings_kernel<true>, dim3((na + 255) / 256), dim3(256), 0, m->st, na, m->d_records.p, m->d_runs_c.p, eng->packed.p, eng->RW, eng->FW,
					m->d_read_len.p, m->d_a_read.p, m->prm.variant == NGM_VARIANT_OCL_CPU ? 1 : 0, m->prm.hard_clip, m->prm.silent_clip, m->d_cigout.p, m->d_str.p, scap,
					(unsigned long long *) (m->d_total.p + 8), 0);
			else hipLaunchKernelGGL(ngm::cigar_strings_kernel<false>, dim3((na + 255) / 256), dim3(256), 0, m->st, na, m->d_records.p, m->d_runs_c.p, eng->packed.p, eng->RW, eng->FW,
					m->d_read_len.p, m->d_a_read.p, m->prm.variant == NGM_VARIANT_OCL_CPU ? 1 : 0, m->prm.hard_clip, m->prm.silent_clip, m->d_cigout.p, m->d_str.p, scap,
					(unsigned long long *) (m->d_total.p + 8), alt_cigar);
			MAP_HIP_TRY(hipGetLastError());
			MAP_HIP_TRY(hipEventRecord(m->ev[8], m->st));
			stage_align.kernels_done();
			MAP_HIP_TRY(hipMemcpyAsync(m->p_cigout.p, m->d_cigout.p, (size_t) na * sizeof(ngm::CigarDevOut), hipMemcpyDeviceToHost, m->st));
			MAP_HIP_TRY(hipMemcpyAsync(&n_str_total, m->d_total.p + 8, 8, hipMemcpyDeviceToHost, m->st));
		}
		MAP_HIP_TRY(hipMemcpyAsync(h_rec, m->d_records.p, (size_t) na * 8 * 4, hipMemcpyDeviceToHost, m->st));
		MAP_HIP_TRY(hipMemcpyAsync(&n_runs_total, m->d_total.p, 8, hipMemcpyDeviceToHost, m->st));
		stage_align.done_after(m->ev[8]);
		MAP_HIP_TRY(hipStreamSynchronize(m->st));
		if (m->p_runs.reserve(n_runs_total + 1)) { ngm::pipeline_set_error("out of pinned host memory"); return -12; }
		h_runs = m->p_runs.p;
		MAP_HIP_TRY(hipMemcpy(h_runs, m->d_runs_c.p, n_runs_total * 2, hipMemcpyDeviceToHost));
		if (dev_strings) {
			n_str_total = std::min<unsigned long long>(n_str_total, (unsigned long long) na * 96ull + 4096ull);
			str_base = n_str_total;
			if (m->p_str.reserve(n_str_total + 1)) { ngm::pipeline_set_error("out of pinned host memory"); return -12; }
			if (n_str_total && !sam) MAP_HIP_TRY(hipMemcpy(m->p_str.p, m->d_str.p, n_str_total, hipMemcpyDeviceToHost));
		}
	}
	stage_align.done();

	lap(3);
	// ---- host: CIGAR / MD, final positions --------------------------------------------------------------
	// (SAM stage: the strings stay in the device's byte stream and the records point into it; only strings the device could
	// not build are made here and appended to that stream)
	ngm::SamRef *sam_refs = sam ? m->p_sam_refs.p : nullptr;
	std::mutex extra_mu;
	std::vector<char> extra;
	parallel_for(n, [&](int lo, int hi) {
		for (int i = lo; i < hi; ++i) for (int t = 0; t < topn; ++t) {
			const size_t o = (size_t) i * topn + t;
			ngm_hit &h = hits[o];
			memset(&h, 0, sizeof(h));
			h.n_candidates = (int) m->h_count[i];
			h.max_votes = m->h_maxv[i];
			h.mapq = h_mapq[i];
			h.n_best = h_nbest[i];
			h.score = h_best[i];
			h.pair_flags = pair_flags[i];
			if (!sam) { cigars[o * str_stride] = 0; mds[o * str_stride] = 0; }
			else sam_refs[o] = ngm::SamRef{0, 0, 0, 0};
		}
	});
	ngm::CigarParams cp{m->prm.match_bonus, -m->prm.mismatch_penalty, m->prm.variant, m->prm.hard_clip, m->prm.silent_clip, alt_cigar};
	parallel_for(na, [&](int lo, int hi) {
		std::vector<char> win((size_t) q + c + 8), qry((size_t) q + 8), scr(sam ? 2 * str_stride : 0);
		for (int j = lo; j < hi; ++j) {
			const int i = (int) a_read[j];
			const size_t o = a_out[j];
			ngm_hit &h = hits[o];
			if (topn > 1) h.score = h_scores[a_pair[j]];  // AS:i of this candidate
			const bool rev = a_sv[j] & 1u;
			h.reverse = rev;
			const char *rd = reads + (size_t) i * q;
			// (the read's length is only needed where the strings are built here: the rows of a batch -- 160 MB per million reads -- are
			// not touched on the host when the GPU has built CIGAR and MD)
			int L_cached = -1;
			auto read_length = [&]() { if (L_cached < 0) L_cached = (int) strnlen(rd, q); return L_cached; };
			ngm_hip_align_out ao{};
			ao.cigar = sam ? scr.data() : cigars + o * str_stride;
			ao.md = sam ? scr.data() + str_stride : mds + o * str_stride;
			bool host_strings = true;
			const ngm::CigarDevOut *dv = dev_strings ? &m->p_cigout.p[j] : nullptr;
			if (dv && (dv->flags & 1)) {  // built on the GPU: copy the two strings and the numbers
				if (!(dv->flags & 2)) { h.mapped = 0; continue; }  // no alignment could be built
				if (sam) { sam_refs[o] = ngm::SamRef{dv->cig_off, dv->md_off, dv->cig_len, dv->md_len}; host_strings = false; }
				else {
					memcpy(ao.cigar, m->p_str.p + dv->cig_off, dv->cig_len); ao.cigar[dv->cig_len] = 0;
					memcpy(ao.md, m->p_str.p + dv->md_off, dv->md_len); ao.md[dv->md_len] = 0;
				}
				ao.identity = dv->identity; ao.nm = dv->nm; ao.qstart = dv->qstart; ao.qend = dv->qend; ao.position_offset = dv->position_offset;
				ao.score_token = dv->score_token;
			} else if (m->prm.personality == NGM_PERSONALITY_AFFINE) {
				// matches / mismatches were counted by the traceback kernel: no window decode, no reverse complement here.
				// EndToEndAffine never touches pBuffer2: the SAM record carries AlignmentBuffer's "!!!" (AlignmentBuffer.cpp:109)
				ngm::build_cigar_affine(&h_rec[(size_t) j * 8], &h_runs[(size_t) (uint32_t) h_rec[(size_t) j * 8 + 6]], nullptr, nullptr, q, &ao, read_length());
				memcpy(ao.md, "!!!", 4);
			} else {
				const uint64_t offset = (uint64_t) a_loc[j] - (uint64_t) (c >> 1);
				host_window(r, offset, align_buf_len, q + c, win.data());
				memset(qry.data(), 0, qry.size());
				const int L = read_length();
				if (!rev) memcpy(qry.data(), rd, L);
				else for (int t = 0; t < L; ++t) {
					const char ch = rd[L - 1 - t];
					qry[t] = ch == 'A' ? 'T' : ch == 'T' ? 'A' : ch == 'C' ? 'G' : ch == 'G' ? 'C' : ch;
				}
				const bool second = paired && (i & 1);
				ngm::build_cigar_md(cp, &h_rec[(size_t) j * 8], &h_runs[(size_t) (uint32_t) h_rec[(size_t) j * 8 + 6]], win.data(), qry.data(), &ao, cp.alt ? ((rev ? !second : second) ? 1 : 0) : 0);
				if (ao.score_token < 0) { h.mapped = 0; continue; }  // no alignment could be built
			}
			if (sam && host_strings) {
				const size_t cl = strlen(ao.cigar), ml = strlen(ao.md);
				std::lock_guard<std::mutex> lk(extra_mu);
				const size_t at = extra.size();
				extra.insert(extra.end(), ao.cigar, ao.cigar + cl);
				extra.insert(extra.end(), ao.md, ao.md + ml);
				sam_refs[o] = ngm::SamRef{(uint32_t) (str_base + at), (uint32_t) (str_base + at + cl), (uint16_t) cl, (uint16_t) ml};
			}
			h.identity = ao.identity; h.nm = ao.nm; h.qstart = ao.qstart; h.qend = ao.qend;
			// AlignmentBuffer.cpp:129 then SequenceProvider.convert (AlignmentBuffer.cpp:173)
			const uint64_t final_loc = (uint64_t) a_loc[j] + (uint64_t) (int64_t) ao.position_offset - (uint64_t) (c >> 1);
			int contig = 0; uint64_t cpos = 0;
			if (!ngm_ref_convert(r, final_loc, &contig, &cpos)) { h.mapped = 0; continue; }
			h.mapped = 1; h.contig = contig; h.pos = cpos;
		}
	});

	lap(4);
	if (sam) {
		// ---- SAM text on the GPU: lengths per unit, exclusive prefix sum, bytes (sam_device.h) ------------------------------
		GpuStage stage_sam(m, 1, false, 2);
		const int units = paired ? n / 2 : n;
		if (!extra.empty()) {
			// the byte stream grows by the host-built strings (rare: strings beyond the device's scratch rows)
			const size_t need = (size_t) str_base + extra.size();
			if (need > m->d_str.cap) {
				ngm::DevBuf<char> bigger;
				if (bigger.reserve(need + 4096)) { ngm::pipeline_set_error("out of device memory (SAM strings)"); return -12; }
				if (str_base) MAP_HIP_TRY(hipMemcpyAsync(bigger.p, m->d_str.p, (size_t) str_base, hipMemcpyDeviceToDevice, m->st));
				MAP_HIP_TRY(hipStreamSynchronize(m->st));
				std::swap(bigger, m->d_str);
				bigger.release();
			}
			if (m->p_sam_extra.reserve(extra.size())) { ngm::pipeline_set_error("out of pinned host memory"); return -12; }
			memcpy(m->p_sam_extra.p, extra.data(), extra.size());
			MAP_HIP_TRY(hipMemcpyAsync(m->d_str.p + str_base, m->p_sam_extra.p, extra.size(), hipMemcpyHostToDevice, m->st));
		}
		if (m->d_sam_len.reserve((size_t) units + 1) || m->d_sam_off.reserve((size_t) units + 1)) { ngm::pipeline_set_error("out of device memory (SAM stage)"); return -12; }
		MAP_HIP_TRY(hipMemcpyAsync(m->d_sam_hits.p, hits, (size_t) n * sizeof(ngm_hit), hipMemcpyHostToDevice, m->st));
		MAP_HIP_TRY(hipMemcpyAsync(m->d_sam_refs.p, sam_refs, (size_t) n * sizeof(ngm::SamRef), hipMemcpyHostToDevice, m->st));
		MAP_HIP_TRY(hipMemsetAsync(m->d_total.p + 16, 0, 64, m->st));
		MAP_HIP_TRY(hipStreamSynchronize(m->st));   // (the uploads -- 64 bytes per read -- travel outside the stage lock)
		stage_sam.acquire();
		ngm::SamArgs S{};
		S.n = n; S.q = q; S.paired = paired ? 1 : 0;
		S.unit_len_bound = (uint32_t) ((paired ? 2 : 1) * (2 * q + 1024));
		S.reads = m->d_reads.p; S.quals = m->d_sam_quals.p; S.names = m->d_sam_names.p; S.meta = m->d_sam_meta.p; S.hits = m->d_sam_hits.p; S.refs = m->d_sam_refs.p;
		S.str = m->d_str.p; S.contig_names = m->d_sam_contig_names.p; S.contig_name_off = m->d_sam_contig_off.p;
		S.min_insert = m->sam_opt.min_insert_size; S.max_insert = m->sam_opt.max_insert_size > 0 ? m->sam_opt.max_insert_size : 2147483647;
		S.min_mq = m->sam_opt.min_mq; S.no_unal = m->sam_opt.no_unal; S.hard_clip = m->prm.hard_clip; S.silent_clip = m->prm.silent_clip;
		S.min_identity = m->sam_opt.min_identity; S.min_residues = m->sam_opt.min_residues;
		S.rg = m->sam_rg.empty() ? nullptr : m->d_sam_rg.p; S.rg_len = (int) m->sam_rg.size(); S.bs_mapping = m->sam_opt.bs_mapping;
		S.slam_seq = m->sam_opt.slam_seq; S.variant_cpu = m->prm.variant == NGM_VARIANT_OCL_CPU ? 1 : 0; S.alt_scoring = (m->prm.bs_mapping || (m->prm.slam_seq & 2)) ? 1 : 0;
		S.genome = r->d_genome; S.contig_start = m->d_sam_contig_start.p;
		S.unit_len = m->d_sam_len.p; S.unit_off = m->d_sam_off.p; S.counters = m->d_total.p + 16;
		S.bam = m->sam_opt.bam ? 1 : 0;
		hipEvent_t e0 = m->cev[0], e1 = m->cev[1];
		MAP_HIP_TRY(hipEventRecord(e0, m->st));
		unsigned long long total = 0;
		if (units > 0) {
			hipLaunchKernelGGL(ngm::sam_lengths_kernel, dim3((units + 255) / 256), dim3(256), 0, m->st, S, units);
			MAP_HIP_TRY(hipGetLastError());
			size_t tmp_bytes = 0;
			(void) rocprim::exclusive_scan(nullptr, tmp_bytes, m->d_sam_len.p, m->d_sam_off.p, 0u, (size_t) units + 1, rocprim::plus<uint32_t>(), m->st);
			if (m->d_scan_tmp.reserve(tmp_bytes + 16)) { ngm::pipeline_set_error("out of device memory (scan)"); return -12; }
			MAP_HIP_TRY(hipMemsetAsync(m->d_sam_len.p + units, 0, 4, m->st));
			MAP_HIP_TRY(rocprim::exclusive_scan(m->d_scan_tmp.p, tmp_bytes, m->d_sam_len.p, m->d_sam_off.p, 0u, (size_t) units + 1, rocprim::plus<uint32_t>(), m->st));
			uint32_t total32 = 0;
			unsigned long long total64 = 0;
			stage_sam.before_sync();
			MAP_HIP_TRY(hipMemcpyAsync(&total32, m->d_sam_off.p + units, 4, hipMemcpyDeviceToHost, m->st));
			MAP_HIP_TRY(hipMemcpyAsync(&total64, m->d_total.p + 19, 8, hipMemcpyDeviceToHost, m->st));
			MAP_HIP_TRY(hipStreamSynchronize(m->st));
			if (total64 != (unsigned long long) total32) {   // the 32-bit prefix sums have wrapped (detected directly, not through a per-record bound)
				ngm::pipeline_set_error("ngm_mapper_map_sam: the text of this batch of %d reads is %llu bytes, beyond the 32-bit offsets of a batch: use smaller batches", n, total64);
				return -75;
			}
			total = total32;
			if (m->d_sam_text.reserve((size_t) total + 16)) { ngm::pipeline_set_error("out of device memory (SAM text)"); return -12; }
			S.out = m->d_sam_text.p;
			stage_sam.acquire();
			hipLaunchKernelGGL(ngm::sam_write_kernel, dim3((units + 255) / 256), dim3(256), 0, m->st, S, units);
			MAP_HIP_TRY(hipGetLastError());
		}
		MAP_HIP_TRY(hipEventRecord(e1, m->st));
		stage_sam.kernels_done();
		unsigned long long ctr[7] = {0, 0, 0, 0, 0, 0, 0};
		MAP_HIP_TRY(hipMemcpyAsync(ctr, m->d_total.p + 16, 56, hipMemcpyDeviceToHost, m->st));
		m->sam_text_bytes = total;
		const bool bam = m->sam_opt.bam != 0;
		if (!bam && total <= sam->out_cap && total > 0) MAP_HIP_TRY(hipMemcpyAsync(sam->out, m->d_sam_text.p, (size_t) total, hipMemcpyDeviceToHost, m->st));
		stage_sam.done_after(e1);   // the text (~420 bytes per read) travels while the next instance's kernels run
		MAP_HIP_TRY(hipStreamSynchronize(m->st));
		sam->text_bytes = (long long) total;
		float bgzf_ms = 0.f;
		if (bam && total > 0) {
			// the records stay in HBM: their BGZF blocks are written there too (bgzf_device.h, the compressor's own stream -- beside the next
			// instance's kernels), and only those travel
			if (sam->out_cap < ngm_bgzf_bound((size_t) total)) sam->text_bytes = (long long) ngm_bgzf_bound((size_t) total);   // (> out_cap: ngm_mapper_sam_fetch with a buffer of that size)
			else {
				const long long zlen = ngm_bgzf_compress_device(m->bz, m->d_sam_text.p, (size_t) total, sam->out, sam->out_cap);
				if (zlen < 0) return (int) zlen;
				sam->text_bytes = zlen;
				m->sam_text_bytes = 0;
				bgzf_ms = ngm_bgzf_last_kernel_ms(m->bz);
			}
		}
		if (sam->stats) { sam->stats[0] = ctr[0]; sam->stats[1] = ctr[1]; sam->stats[2] = ctr[2]; }
		m->pair_stats[0] = ctr[4]; m->pair_stats[1] = ctr[5]; m->pair_stats[2] = ctr[6];
		float t = 0;
		sam->kernel_ms = (hipEventElapsedTime(&t, e0, e1) == hipSuccess ? t : 0.f) + bgzf_ms;
		lap(5);
	}
	if (host_timing)
		fprintf(stderr, "[ngm-hip] host wall ms: candidate search %.1f | score stage + downloads %.1f | pair selection %.1f | align stage + downloads %.1f | CIGAR/positions %.1f\n",
				t_stage[0], t_stage[1], t_stage[2], t_stage[3], t_stage[4]);
	// kernel times
	auto et = [&](int a, int b) { float t = 0; if (hipEventElapsedTime(&t, m->ev[a], m->ev[b]) != hipSuccess) t = 0; return t; };
	m->ms[0] = m->cs_kernel_ms;  // sum of the candidate-search kernel launches only
	m->ms[7] = et(0, 1);          // ... and the whole CS stage including the host round trips between passes
	if (np > 0) { m->ms[1] = et(1, 2); m->ms[2] = et(2, 3); m->ms[3] = et(3, 4); }
	if (na > 0) {
		m->ms[4] = et(5, 6);
		float t = 0;
		if (hipEventElapsedTime(&t, m->ev[6], eng->ev[2]) == hipSuccess) m->ms[5] = t;
		if (hipEventElapsedTime(&t, eng->ev[2], m->ev[7]) == hipSuccess) m->ms[6] = t;
	}
	return n;
}

ngm_pair_state *ngm_pair_state_create(void) { return new ngm_pair_state(); }
void ngm_pair_state_destroy(ngm_pair_state *ps) { delete ps; }
int ngm_mapper_set_pair_state(ngm_mapper *m, ngm_pair_state *ps) { if (!m) return -22; m->ps = ps; return 0; }
int ngm_mapper_set_batch_seq(ngm_mapper *m, uint64_t seq) { if (!m) return -22; m->batch_seq = seq; return 0; }
int ngm_mapper_set_fast_pairing(ngm_mapper *m, int on) { if (!m) return -22; m->fast_pairing = on ? 1 : 0; return 0; }

void *ngm_host_alloc(size_t bytes) {
	void *p = nullptr;
	if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { ngm::pipeline_set_error("out of pinned host memory (%zu bytes)", bytes); return nullptr; }
	return p;
}
void ngm_host_free(void *p) { if (p) (void) hipHostFree(p); }

int ngm_host_pin_to_device_node(int device) {
	if (getenv("NGM_HIP_NO_NUMA_PIN")) return 0;
	char bdf[64] = {0};
	if (hipDeviceGetPCIBusId(bdf, (int) sizeof(bdf), device) != hipSuccess) { ngm::pipeline_set_error("no PCI bus id for device %d", device); return -19; }
	for (char *c = bdf; *c; ++c) *c = (char) tolower((unsigned char) *c);
	char path[256], line[4096] = {0};
	snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bdf);
	int node = -1;
	if (FILE *f = fopen(path, "r")) { if (fscanf(f, "%d", &node) != 1) node = -1; fclose(f); }
	if (node < 0) return 0;
	snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
	FILE *f = fopen(path, "r");
	if (!f) return 0;
	const bool got = fgets(line, sizeof(line), f) != nullptr;
	fclose(f);
	if (!got) return 0;
	cpu_set_t set;
	CPU_ZERO(&set);
	int n = 0;
	for (const char *c = line; *c;) {  // "0-63,128-191"
		char *end = nullptr;
		const long a = strtol(c, &end, 10);
		if (end == c) break;
		long b = a;
		c = end;
		if (*c == '-') { b = strtol(c + 1, &end, 10); if (end == c + 1) break; c = end; }
		for (long x = a; x <= b && x < CPU_SETSIZE; ++x) if (x >= 0) { CPU_SET((int) x, &set); ++n; }
		if (*c == ',') ++c; else break;
	}
	if (n == 0) return 0;
	if (sched_setaffinity(0, sizeof(set), &set) != 0) return 0;  // (a container may forbid it: not an error)
	return n;
}

int ngm_mapper_cs_max_combined(ngm_mapper *m, float *out) {
	if (!m) return -22;
	DevGuard g(m->ref->device);
	if (m->n_reads > 0) MAP_HIP_TRY(hipMemcpy(out, m->d_max_both.p, (size_t) m->n_reads * 4, hipMemcpyDeviceToHost));
	return 0;
}

int ngm_mapper_set_reference_cs_batch(ngm_mapper *m, int reads) {
	if (!m || reads < 0) return -22;
	m->ref_cs_batch = reads & ~1;
	return 0;
}

int ngm_mapper_set_reference_score_buffer(ngm_mapper *m, int entries) {
	if (!m || entries < 0) return -22;
	m->ref_score_buffer = entries;
	return 0;
}

int ngm_mapper_lost_pairs(ngm_mapper *m, uint64_t *out) {
	if (!m || !out) return -22;
	*out = m->lost_pairs;
	return 0;
}

int ngm_mapper_path_counters(ngm_mapper *m, uint64_t out[8]) {
	if (!m || !out) return -22;
	out[0] = m->st_reads; out[1] = m->st_cands; out[2] = m->st_exact_lds; out[3] = m->st_exact_global;
	out[4] = m->st_order_reads; out[5] = m->st_order_big; out[6] = m->st_order_unknown; out[7] = m->st_heavy;
	return 0;
}

int ngm_mapper_order_table_reads(ngm_mapper *m, uint64_t *out) {
	if (!m || !out) return -22;
	*out = m->st_order_table;
	return 0;
}

int ngm_mapper_cs_counters(ngm_mapper *m, uint64_t out[3]) {
	if (!m) return -22;
	out[0] = m->cs_kmers; out[1] = m->cs_hits; out[2] = m->n_cand;
	return 0;
}

float ngm_mapper_last_order_replay_ms(ngm_mapper *m) { return m ? m->order_ms : 0.f; }

int ngm_mapper_last_pair_stats(ngm_mapper *m, uint64_t out[3]) {
	if (!m || !out) return -22;
	out[0] = m->pair_stats[0]; out[1] = m->pair_stats[1]; out[2] = m->pair_stats[2];
	return 0;
}

int ngm_mapper_last_kernel_ms(ngm_mapper *m, float ms[8]) {
	if (!m) return -22;
	for (int i = 0; i < 8; ++i) ms[i] = m->ms[i];
	return 0;
}

}  // extern "C"
