// mapper_internal.h -- what the translation units of the mapping path share (mapper.cpp: score / select / align / SAM stages and the C
// entry points; mapper_search.cpp: candidate search and the candidate-order replay): the mapper's state, the stage lock, error plumbing.
#pragma once

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
#include "../../include/ngm_pipeline.h"
#include "refindex.h"
#include "engine_internal.h"
#include "cigar_md.h"
#include "cigar_device.h"
#include "cs_device.h"
#include "sam_device.h"
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
	int cs_canon = 0;         // 0: fast path over one bucket per k-mer (cs_fast2_kernel); 1-3: over canonical pair buckets, cs_canon_kernel<3,4,2> / <3,6,2> / <4,8,4>
	long pair_dist_count = 1, pair_dist_sum = 0;  // ScoreBuffer.h:90
	uint64_t scores_so_far = 0, reads_so_far = 0;  // see ngm_pair_state
	int ref_cs_batch = 0;             // reads per CS batch of the reference (1 800 000 / average read length, CS.cpp:26, :542): its score buffer is flushed there
	int ref_score_buffer = 0;         // entries of the reference's score buffer (IAlignment::GetScoreBatchSize there); 0: pairs are never lost (ngm_mapper_set_reference_score_buffer)
	uint64_t lost_pairs = 0;          // ngm_mapper_lost_pairs
	// ngm_mapper_path_counters: reads searched, candidates, reads re-run by the exact LDS / exact global-memory search, reads whose
	// candidate order was replayed, of those beyond the LDS replay's limits (replayed by the exact global-memory kernel), left undetermined
	uint64_t st_heavy_second = 0, st_heavy_restart = 0, st_heavy_sent_on = 0, st_pool_regrown = 0;   // ngm_mapper_heavy_counters: second passes, table passes started over, reads a heavy class sent on, regrown table pools
	int cus = 256;                    // compute units of the device
	int heavy_per_cu[3] = {0, 0, 0};  // workgroups of each heavy class a CU holds (for the LDS size in heavy_lds)
	size_t heavy_lds[3] = {0, 0, 0};
	uint64_t st_heavy = 0, st_reads = 0, st_cands = 0, st_exact_lds = 0, st_exact_global = 0, st_order_reads = 0, st_order_big = 0, st_order_unknown = 0, st_order_table = 0;
	ngm::CsArgs last_cs{};                          // arguments of the last candidate search (for the order replay)
	hipStream_t st_hi = nullptr;                    // high-priority stream: the (small) order replay runs outside the stage lock
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
	ngm::DevBuf<uint32_t> d_cand_base, d_cand_count, d_out_loc, d_out_sv, d_status, d_ovf_read, d_ovf_read2, d_ovf_hits, d_ovf_hits2, d_ovf_log2;
	ngm::PinnedBuf<uint32_t> p_cs_status;   // the status and control blocks of a search, as downloaded (cs_queue_device.h)
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

namespace ngm {

struct DevGuard {
	int prev = -1;
	explicit DevGuard(int d) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; if (prev != d) (void) hipSetDevice(d); else prev = -1; }
	~DevGuard() { if (prev >= 0) (void) hipSetDevice(prev); }
};

// ---- whose kernels run now ------------------------------------------------------------------------------------------------
// GPU stages (candidate search + score, align, SAM text: each from its first launch to the end of its last kernel) of the mapper
// instances of one process take turns: kernels of different instances then do not slow each other down, while the host stages of one
// instance -- and the downloads behind a stage's last kernel -- overlap the GPU stages of the others.
// NGM_HIP_GPU_STAGE_LOCK: 0 no turns (streams share the GPU), 1 one lock per device (default), 2 one lock per stage kind (search + score
// | align + SAM text).  (Round 4 also tried passing the turn ON THE GPU, with events between the streams: 51.6 / 47.3 M reads/s against
// 54.8 / 51.3 with the host lock -- removed in round 6.)
struct StageLock { std::mutex mu; };
extern StageLock g_stage_lock[16][2];
extern std::atomic<long long> g_stage_hold_us[3], g_stage_wait_us[3];   // diagnostics (NGM_HIP_HOST_TIMING): turn held / waited for on the host, per stage kind (0 search + score, 1 align, 2 SAM text)
struct GpuStage {
	ngm_mapper *m;
	int kind, slot;
	bool held = false;
	StageLock *ch = nullptr;
	std::chrono::steady_clock::time_point t_acq;
	static int mode() { static const int v = getenv("NGM_HIP_GPU_STAGE_LOCK") ? atoi(getenv("NGM_HIP_GPU_STAGE_LOCK")) : 1; return v; }
	GpuStage(ngm_mapper *m_, int kind_ = 0, bool now = true, int slot_ = -1) : m(m_), kind(kind_), slot(slot_ < 0 ? kind_ : slot_) {
		ch = &g_stage_lock[(unsigned) m->ref->device & 15u][mode() == 2 ? kind : 0];
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
	}
	void release() {   // the kernels of this turn have finished
		if (!held) return;
		g_stage_hold_us[slot] += std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t_acq).count();
		held = false;
		ch->mu.unlock();
	}
	// the stage's last kernel has been enqueued and `ev` recorded behind it; copies to the host follow: the turn ends when that kernel has
	// finished, not when the copies have (the next instance's kernels run under them)
	void done_after(hipEvent_t ev) { if (held) (void) hipEventSynchronize(ev); release(); }
	void done() { release(); }
};

// ---- mapper_search.cpp ----------------------------------------------------------------------------------------------------------
// fast-path geometry, kernel attributes (part of ngm_mapper_create)
int cs_configure(ngm_mapper *m, const ngm_mapper_params *p);
// candidate search for the reads already in m->d_reads; leaves per-read base/count/max votes and the candidate arrays in HBM (and
// base/count/max votes on the host)
int run_cs(ngm_mapper *m, int n, GpuStage *stage = nullptr);
int cs_host_arrays(ngm_mapper *m);
// Reference order of the candidates of the listed reads (cs_order_kernel): h_rank[c] for every candidate c of those
// reads, kCsOrderUnknown where it could not be determined.  Only called for reads where the order decides something.
// wait = false: only enqueue (the list must stay alive until candidate_order_wait)
int candidate_order(ngm_mapper *m, const std::vector<uint32_t> &list, uint64_t np, uint32_t **h_rank, bool wait = true);
int candidate_order_wait(ngm_mapper *m, uint32_t **h_rank);
void cs_release(ngm_mapper *m);   // the search side's buffers (ngm_mapper_destroy)

}  // namespace ngm
