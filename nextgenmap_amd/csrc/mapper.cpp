// mapper.cpp -- the device-resident mapping path above IAlignment (include/ngm_pipeline.h):
//   candidate search -> window gather -> BatchScore -> top-1 selection / MAPQ -> window gather -> BatchAlign.
// One ngm_mapper is what one NextGenMap CS thread owns (CS + ScoreBuffer + AlignmentBuffer + IAlignment,
// src/CS.cpp:455-461); everything between the read upload and the traceback download stays in HBM.
// This file: the mapper's life cycle, the score / select / align / SAM stages and the C entry points; the candidate-search stage and the
// candidate-order replay live in mapper_search.cpp (mapper_internal.h is what the two share).
#define NGM_SAM_KERNELS
#include "mapper_internal.h"
#include <rocprim/rocprim.hpp>
#include "align_device.h"
#include "gather_device.h"

using ngm::DevGuard;
using ngm::GpuStage;
using ngm::run_cs;
using ngm::cs_host_arrays;
using ngm::candidate_order;
using ngm::candidate_order_wait;

namespace ngm {
StageLock g_stage_lock[16][2];
std::atomic<long long> g_stage_hold_us[3], g_stage_wait_us[3];
}
using ngm::g_stage_hold_us;
using ngm::g_stage_wait_us;

namespace {

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
		// (the order replay's stream: greatest priority -- persistent search workgroups hold every CU until their launch ends, and the replay
		// gets in as they leave)
		if (hipStreamCreateWithPriority(&m->st_hi, hipStreamNonBlocking, hi) != hipSuccess) m->st_hi = nullptr;
	}
	m->max_kfreq = p->max_kfreq > 0 ? p->max_kfreq : ref->auto_max_kfreq;
	for (auto &e : m->ev) (void) hipEventCreate(&e);
	for (auto &e : m->cev) (void) hipEventCreate(&e);
	for (auto &e : m->oev) (void) hipEventCreate(&e);
	if (ngm::cs_configure(m, p) != 0) { ngm_mapper_destroy(m); return nullptr; }
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
	ngm::cs_release(m);
	m->d_reads.release(); m->d_scores.release(); m->d_best.release(); m->d_pair_read.release();
	m->d_winner.release(); m->d_a_read.release(); m->d_a_loc.release(); m->d_a_sv.release(); m->d_mapq.release(); m->d_nbest.release();
	m->d_records.release(); m->d_runs.release(); m->d_runs_c.release();
	m->d_pair_info.release(); m->p_pair_info.release(); m->d_sam_contig_start.release();
	m->d_pair_out.release(); m->d_pair_top.release(); m->d_pair_tied_n.release(); m->d_pair_list.release(); m->p_pair_out.release(); m->p_pair_top.release(); m->p_pair_tied_n.release();
	m->d_sam_contig_names.release(); m->d_sam_rg.release(); m->d_sam_names.release(); m->d_sam_text.release(); m->d_sam_contig_off.release(); m->d_sam_len.release(); m->d_sam_off.release();
	m->d_sam_quals.release(); m->d_sam_meta.release(); m->d_sam_refs.release(); m->d_sam_hits.release(); m->p_sam_hits.release(); m->p_sam_refs.release(); m->p_sam_extra.release();
	for (auto &e : m->ev) if (e) (void) hipEventDestroy(e);
	for (auto &e : m->cev) if (e) (void) hipEventDestroy(e);
	for (auto &e : m->oev) if (e) (void) hipEventDestroy(e);
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
	if (!ranked) {
		if (cnt > 64) {   // (a total order: any algorithm gives the same result -- sorted as records, not through the index arrays)
			struct Rec { float s; uint32_t l, st, i; };
			thread_local std::vector<Rec> recs;
			recs.resize(cnt);
			for (uint32_t x = 0; x < cnt; ++x) { const uint32_t i = base + x; recs[x] = Rec{score[i], loc[i], sv[i] & 1u, i}; }
			std::sort(recs.begin(), recs.end(), [](const Rec &a, const Rec &b) { return a.s != b.s ? a.s > b.s : a.l != b.l ? a.l < b.l : a.st < b.st; });
			for (uint32_t x = 0; x < cnt; ++x) v[x] = recs[x].i;
			return;
		}
		std::sort(v, v + cnt, [&](uint32_t x, uint32_t y) { return score[x] != score[y] ? score[x] > score[y] : by_place(x, y); });
		return;
	}
	auto by_rank = [&](uint32_t x, uint32_t y) { return rank[x] != rank[y] ? rank[x] < rank[y] : by_place(x, y); };
	if (cnt <= 16) { std::sort(v, v + cnt, [&](uint32_t x, uint32_t y) { return score[x] != score[y] ? score[x] > score[y] : by_rank(x, y); }); return; }
	// Round 6: the same two sorts on VALUES instead of through the index arrays (a read of a repeat family has thousands of candidates, and
	// the stress sub-leg of the bench sorts ~10 000 such lists per batch on a 16-CPU quota: every comparison was two or four cache misses).
	// (1) the reference's candidate order: ranks are distinct (2 x the entering time of the bin + strand), so (rank << 32 | index) sorts as
	// integers -- should two ranks ever be equal, the comparator path below decides as before; (2) std::sort by score on (score, index)
	// records in that order: the same algorithm asked the same questions in the same sequence moves the same elements.
	{
		thread_local std::vector<uint64_t> keys;
		struct Rec { float s; uint32_t i; };
		thread_local std::vector<Rec> recs;
		keys.resize(cnt);
		uint32_t rank_or = 0;
		for (uint32_t x = 0; x < cnt; ++x) { keys[x] = ((uint64_t) rank[base + x] << 32) | (uint64_t) (base + x); rank_or |= rank[base + x]; }
		if (cnt < 256) std::sort(keys.begin(), keys.end());
		else {
			// ranks are 2 x a hit time + strand: 21-23 bits -- two or three stable passes of 11 bits (equal ranks, should there be any, keep the
			// index order they were written in: what the integer sort of (rank, index) gives)
			thread_local std::vector<uint64_t> tmp;
			tmp.resize(cnt);
			uint64_t *src = keys.data(), *dst = tmp.data();
			for (int sh = 32; sh < 64 && (rank_or >> (sh - 32)) != 0u; sh += 11) {
				uint32_t hist[2048] = {0};
				for (uint32_t x = 0; x < cnt; ++x) ++hist[(src[x] >> sh) & 2047u];
				uint32_t run = 0;
				for (uint32_t h = 0; h < 2048; ++h) { const uint32_t c = hist[h]; hist[h] = run; run += c; }
				for (uint32_t x = 0; x < cnt; ++x) dst[hist[(src[x] >> sh) & 2047u]++] = src[x];
				std::swap(src, dst);
			}
			if (src != keys.data()) std::copy(src, src + cnt, keys.data());
		}
		bool distinct = true;
		for (uint32_t x = 1; x < cnt && distinct; ++x) distinct = (keys[x] >> 32) != (keys[x - 1] >> 32);
		if (distinct) {
			recs.resize(cnt);
			for (uint32_t x = 0; x < cnt; ++x) { const uint32_t i = (uint32_t) keys[x]; recs[x] = Rec{score[i], i}; }
			std::sort(recs.begin(), recs.end(), [](const Rec &a, const Rec &b) { return a.s > b.s; });
			for (uint32_t x = 0; x < cnt; ++x) v[x] = recs[x].i;
			return;
		}
	}
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
// (NGM_HIP_HOST_TIMING: where pass 3's CPU time goes -- [0] ns in the two sorts, [1] ns in the rest of the walk, [2] candidates, [3] heads, [4] combinations looked at, [5] pairs)
static std::atomic<bool> g_walk_probe{false};
static std::atomic<uint64_t> g_walk_ns[6];
template <typename F>
static void walk_pair(const ngm_mapper_params &prm, uint32_t base_a, uint32_t cnt_a, int len_a, uint32_t base_b, uint32_t cnt_b, int len_b,
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
	const bool tm = g_walk_probe.load(std::memory_order_relaxed);
	const auto t_0 = tm ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
	sort_like_reference(A, base_a, cnt_a, loc, sv, score, rank);
	sort_like_reference(B, base_b, cnt_b, loc, sv, score, rank);
	const auto t_1 = tm ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
	struct WalkProbe { bool on; std::chrono::steady_clock::time_point t1; uint64_t cnt, *na, *nb, *nv; ~WalkProbe() { if (!on) return;
		g_walk_ns[1] += (uint64_t) std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t1).count(); g_walk_ns[2] += cnt; g_walk_ns[3] += *na + *nb; g_walk_ns[4] += *nv; g_walk_ns[5] += 1; } };
	if (tm) g_walk_ns[0] += (uint64_t) std::chrono::duration_cast<std::chrono::nanoseconds>(t_1 - t_0).count();
	*mq_a = mq_of(A, cnt_a); *mq_b = mq_of(B, cnt_b);
	const float cutoff = prm.pair_score_cutoff > 0 ? prm.pair_score_cutoff : 0.9f;
	const float min_a = score[A[0]] * cutoff, min_b = score[B[0]] * cutoff;
	size_t na = 1, nb = 1;
	while (na < cnt_a && min_a <= score[A[na]]) ++na;
	while (nb < cnt_b && min_b <= score[B[nb]]) ++nb;
	uint64_t pna = na, pnb = nb, n_visits = 0;
	WalkProbe probe{tm, t_1, (uint64_t) cnt_a + cnt_b, &pna, &pnb, &n_visits};
	const int min_d = prm.min_insert_size, max_d = prm.max_insert_size > 0 ? prm.max_insert_size : INT_MAX;
	// Mates with hundreds of candidates each (repeat families of a GRCh38-like genome): CheckPairs walks all na x nb combinations,
	// but only those inside the insert-size window do anything -- B's candidates sorted by position, per candidate of A the ones
	// within max_d, visited in increasing j like the reference's inner loop: the same sequence of in-window pairs, so every
	// order-dependent outcome (first best, the equal counter, the recorded combinations) is unchanged.
	const bool windowed = na * nb > 1024 && max_d < (1 << 28);
	auto visit = [&](size_t i, size_t j) {
		++n_visits;
		const uint64_t l1 = loc[A[i]], l2 = loc[B[j]];
		const int cur = (int) ((l2 > l1) ? l2 - l1 + (uint64_t) len_b : l1 - l2 + (uint64_t) len_a);
		if (cur > min_d && cur < max_d) f(score[A[i]] + score[B[j]], cur, (int) A[i], (int) B[j]);
	};
	// (the reference's insert size is an `int` made from a 64-bit difference, ScoreBuffer.cpp:467-473: two locations at opposite ends of the
	// 32-bit range come out as a small number.  No genome ngm-hip accepts puts candidates there, but the window by location would miss what
	// the double loop counts: such lists take the double loop)
	bool wraps = false;
	if (windowed) {
		uint32_t l_min = 0xFFFFFFFFu, l_max = 0;
		for (size_t i = 0; i < na; ++i) { l_min = std::min(l_min, loc[A[i]]); l_max = std::max(l_max, loc[A[i]]); }
		for (size_t j = 0; j < nb; ++j) { l_min = std::min(l_min, loc[B[j]]); l_max = std::max(l_max, loc[B[j]]); }
		wraps = (uint64_t) (l_max - l_min) + (uint64_t) std::max(len_a, len_b) >= (1ull << 32);
	}
	if (!windowed || wraps) {
		for (size_t i = 0; i < na; ++i) for (size_t j = 0; j < nb; ++j) visit(i, j);
		return;
	}
	// Round 6 (the stress sub-leg of the bench: ~1 800 candidates above the cut-off per tied pair, 5 600 such pairs per batch): both heads
	// sorted by location, ONE sweep with two pointers gives every candidate of A its run of B's candidates within max_d -- the window's
	// bounds grow with the location -- and the double loop's order (i major, j ascending) comes from a bit per j: set for the run, read
	// back lowest first.  (Round 5: a binary search and a sort of the run per candidate of A -- half of pass 3's CPU time; sorting all
	// matches at once was tried and is slower: satellite arrays put hundreds of B's candidates into every window.)
	thread_local std::vector<uint64_t> a_by_loc, b_by_loc, bits;
	thread_local std::vector<uint32_t> run_lo, run_hi;
	a_by_loc.resize(na); b_by_loc.resize(nb); run_lo.resize(na); run_hi.resize(na);
	bits.assign((nb + 63) / 64, 0ull);
	for (size_t i = 0; i < na; ++i) a_by_loc[i] = ((uint64_t) loc[A[i]] << 32) | (uint64_t) i;
	for (size_t j = 0; j < nb; ++j) b_by_loc[j] = ((uint64_t) loc[B[j]] << 32) | (uint64_t) j;
	std::sort(a_by_loc.begin(), a_by_loc.end());
	std::sort(b_by_loc.begin(), b_by_loc.end());
	size_t lo = 0, hi = 0;
	for (size_t x = 0; x < na; ++x) {
		const uint64_t l1w = a_by_loc[x] >> 32;
		const uint64_t lo_loc = l1w > (uint64_t) max_d ? l1w - (uint64_t) max_d : 0u;
		const uint64_t hi_loc = std::min<uint64_t>(l1w + (uint64_t) max_d, 0xFFFFFFFFull);
		while (lo < nb && (b_by_loc[lo] >> 32) < lo_loc) ++lo;
		if (hi < lo) hi = lo;
		while (hi < nb && (b_by_loc[hi] >> 32) <= hi_loc) ++hi;
		const size_t i = (size_t) (a_by_loc[x] & 0xFFFFFFFFull);
		run_lo[i] = (uint32_t) lo; run_hi[i] = (uint32_t) hi;
	}
	for (size_t i = 0; i < na; ++i) {
		const uint32_t r0 = run_lo[i], r1 = run_hi[i];
		if (r1 - r0 == 1u) { visit(i, (size_t) (b_by_loc[r0] & 0xFFFFFFFFull)); continue; }
		if (r1 == r0) continue;
		uint32_t w_min = 0xFFFFFFFFu, w_max = 0;
		for (uint32_t y = r0; y < r1; ++y) {
			const uint32_t j = (uint32_t) b_by_loc[y], w = j >> 6;
			bits[w] |= 1ull << (j & 63u);
			w_min = std::min(w_min, w); w_max = std::max(w_max, w);
		}
		for (uint32_t w = w_min; w <= w_max; ++w) {
			uint64_t word = bits[w];
			bits[w] = 0ull;
			while (word) { visit(i, (size_t) w * 64u + (size_t) __builtin_ctzll(word)); word &= word - 1ull; }
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
// what build_seq hands walk_pair: a combination is kept unless eval_pair_seq passes over it whatever the running mean -- its `top` is the
// maximum of the pair scores so far, starting at 0, and a combination below it takes neither branch (a pair of two satellite-array
// mates has hundreds of thousands of those)
struct PairSeqKeeper {
	PairSeq &sq;
	float top = 0.0f;
	void operator()(float ps, int cur, int ia, int ib) {
		if (ps < top) return;
		top = ps;
		sq.ps.push_back(ps); sq.d.push_back(cur); sq.a.push_back(ia); sq.b.push_back(ib);
	}
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
	walk_pair(m->prm, base_a, cnt_a, len_a, base_b, cnt_b, len_b, loc, sv, score, rank, mq_a, mq_b, [&](float ps, int cur, int ia, int ib) {
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
		const bool pe_select = paired && !m->fast_pairing;   // --fast-pairing: the mates are selected single-end, the writer checks the pair (AlignmentBuffer.cpp:176-199)
		const bool simple_on_gpu = pe_select && m->prm.strata == 0;   // (--strata touches NH of every pair: host)
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
				if (m->d_pair_out.reserve(2 * npairs + 2) || m->d_pair_top.reserve(npairs + 1) || m->d_pair_tied_n.reserve(4) || m->p_pair_tied_n.reserve(4) || m->d_pair_list.reserve(3 * npairs + 2)) {
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
				uint32_t *const huge_list = m->d_pair_list.p + 2 * npairs, *const huge_count = m->d_pair_tied_n.p + 3;   // entries of the large pairs with more than kPairCap candidates above the cut-off
				hipLaunchKernelGGL((ngm::pair_choice_kernel<64, 64>), dim3((unsigned) std::min<size_t>(npairs, 8192)), dim3(64), ngm::pair_choice_lds_bytes(64), m->st, (const uint32_t *) m->d_pair_list.p, (const uint32_t *) counts,
						m->d_cand_base.p, m->d_cand_count.p, m->d_scores.p, m->d_out_loc.p, m->d_read_len.p, min_d, max_d, cutoff, m->d_pair_out.p, m->d_pair_top.p, m->d_pair_tied_n.p, (uint32_t) npairs,
						(uint32_t *) nullptr, (uint32_t *) nullptr, (const uint32_t *) nullptr);
				hipLaunchKernelGGL((ngm::pair_choice_kernel<ngm::kPairThreads, ngm::kPairCap>), dim3((unsigned) std::min<size_t>(npairs, 1024)), dim3(ngm::kPairThreads), ngm::pair_choice_lds_bytes(ngm::kPairCap), m->st,
						(const uint32_t *) (m->d_pair_list.p + npairs), (const uint32_t *) (counts + 1), m->d_cand_base.p, m->d_cand_count.p, m->d_scores.p, m->d_out_loc.p, m->d_read_len.p, min_d, max_d, cutoff,
						m->d_pair_out.p + npairs, m->d_pair_top.p, m->d_pair_tied_n.p, (uint32_t) npairs, huge_list, huge_count, (const uint32_t *) nullptr);
				// ... and what outgrew those lists once more with kPairCapHuge of them (96 KB of LDS: one workgroup per CU; a few hundred pairs of repeat families per batch)
				static const bool huge_attr = [] { (void) hipFuncSetAttribute((const void *) ngm::pair_choice_kernel<1024, ngm::kPairCapHuge>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) ngm::pair_choice_lds_bytes(ngm::kPairCapHuge)); return true; }();
				(void) huge_attr;
				hipLaunchKernelGGL((ngm::pair_choice_kernel<1024, ngm::kPairCapHuge>), dim3((unsigned) std::min<size_t>(npairs, 256)), dim3(1024), ngm::pair_choice_lds_bytes(ngm::kPairCapHuge), m->st,   // (one workgroup per CU either way: sixteen waves share a pair)
						(const uint32_t *) huge_list, (const uint32_t *) huge_count, m->d_cand_base.p, m->d_cand_count.p, m->d_scores.p, m->d_out_loc.p, m->d_read_len.p, min_d, max_d, cutoff,
						m->d_pair_out.p + npairs, m->d_pair_top.p, m->d_pair_tied_n.p, (uint32_t) npairs, (uint32_t *) nullptr, (uint32_t *) nullptr, (const uint32_t *) (m->d_pair_list.p + npairs));
				MAP_HIP_TRY(hipGetLastError());
				MAP_HIP_TRY(hipMemcpyAsync(m->p_pair_tied_n.p, m->d_pair_tied_n.p, 16, hipMemcpyDeviceToHost, m->st));
			}
			MAP_HIP_TRY(hipMemcpyAsync(m->p_pair_info.p, m->d_pair_info.p, (size_t) (n / 2) * 4, hipMemcpyDeviceToHost, m->st));
		}
		MAP_HIP_TRY(hipEventRecord(m->ev[4], m->st));
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
		if ((!paired || m->fast_pairing) && m->prm.topn <= 1) {
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
			if (!need.empty()) if (int rc = candidate_order(m, need, np, &h_rank_pe)) return rc;
			qlap(1);
			// Pass 3 (parallel, outside the turn): the in-window combinations of the picked pairs in the reference's order
			std::vector<PairSeq> seqs(picked.size());
			auto build_seq = [&](PairSeq &sq, int pi) {
				const int rb = 2 * pi, ra = 2 * pi + 1;
				// (only the combinations that reach the running maximum of the pair score are kept: eval_pair_seq does nothing at the others
				// whatever the mean -- its `top` is the maximum so far, starting at 0 -- and a pair of two satellite-array mates has
				// hundreds of thousands of them)
				walk_pair(m->prm, m->h_base[ra], m->h_count[ra], len_of(ra), m->h_base[rb], m->h_count[rb], len_of(rb), h_loc, h_sv, h_scores, h_rank_pe, &sq.mq_a, &sq.mq_b,
						PairSeqKeeper{sq});
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
				if (int rc = candidate_order(m, need_late, np, &h_rank_pe)) return rc;
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
			if (host_timing) {
				g_walk_probe = true;
				const double pn = (double) std::max<uint64_t>(1, g_walk_ns[5].load());
				fprintf(stderr, "[ngm-hip] pair walks so far: %.0f pairs; per pair: sorts %.1f us, rest of the walk %.1f us, %.0f candidates, %.0f above the cut-off, %.0f combinations looked at\n", pn,
						g_walk_ns[0].load() / pn / 1e3, g_walk_ns[1].load() / pn / 1e3, g_walk_ns[2].load() / pn, g_walk_ns[3].load() / pn, g_walk_ns[4].load() / pn);
			}
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
			{
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
		if (!dev_strings) { MAP_HIP_TRY(hipEventRecord(m->ev[8], m->st)); }   // (behind the stage's last kernel)
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

// Debug / test entry, host only (no GPU): the part of top1PE that pass 3 of the pair selection runs per tied pair -- both candidate lists
// sorted as the reference sorts them (candidate order from `rank`, then std::sort by score), the MAPQs, and the in-window combinations of
// the candidates above the cut-off in the order of the reference's double loop.  out_a / out_b: the sorted lists (indices into loc / sv /
// score / rank; cnt_a and cnt_b entries); combo_*: at most `cap` combinations, *n_combo their number (all of them counted).
// tests/test_pair_walk.py compares it with a restatement of libstdc++'s std::sort and the plain double loop.
int ngm_debug_pair_walk(uint32_t cnt_a, int len_a, uint32_t cnt_b, int len_b, const uint32_t *loc, const uint32_t *sv, const float *score, const uint32_t *rank,
		float cutoff, int min_insert, int max_insert, uint32_t *out_a, uint32_t *out_b, int *mq_a, int *mq_b, uint64_t cap, float *combo_score, int *combo_dist, int *combo_a, int *combo_b,
		uint64_t *n_combo) {
	if (!cnt_a || !cnt_b || !loc || !sv || !score || !out_a || !out_b || !mq_a || !mq_b || !n_combo) { ngm::pipeline_set_error("ngm_debug_pair_walk: bad arguments"); return -1; }
	ngm_mapper_params prm{};
	prm.pair_score_cutoff = cutoff; prm.min_insert_size = min_insert; prm.max_insert_size = max_insert;
	sort_like_reference(out_a, 0, cnt_a, loc, sv, score, rank);
	sort_like_reference(out_b, cnt_a, cnt_b, loc, sv, score, rank);
	uint64_t n = 0;
	walk_pair(prm, 0, cnt_a, len_a, cnt_a, cnt_b, len_b, loc, sv, score, rank, mq_a, mq_b, [&](float ps, int cur, int ia, int ib) {
		if (n < cap && combo_score && combo_dist && combo_a && combo_b) { combo_score[n] = ps; combo_dist[n] = cur; combo_a[n] = ia; combo_b[n] = ib; }
		++n;
	});
	*n_combo = n;
	return 0;
}

// Debug / test entry, host only: the double loop of top1PE over CheckPairs at running mean `avg` (eval_pair_seq) on a given sequence of
// in-window combinations -- once on all of them, once on what PairSeqKeeper keeps.  out[6]: found, winner of a, winner of b, pairs of
// equal score and insert size, insert size, combinations evaluated.
int ngm_debug_pair_eval(uint64_t n, const float *pair_score, const int *dist, const int *ia, const int *ib, int avg, int out_all[6], int out_kept[6]) {
	if ((n && (!pair_score || !dist || !ia || !ib)) || !out_all || !out_kept) { ngm::pipeline_set_error("ngm_debug_pair_eval: bad arguments"); return -1; }
	PairSeq all, kept;
	PairSeqKeeper keep{kept};
	for (uint64_t x = 0; x < n; ++x) {
		all.ps.push_back(pair_score[x]); all.d.push_back(dist[x]); all.a.push_back(ia[x]); all.b.push_back(ib[x]);
		keep(pair_score[x], dist[x], ia[x], ib[x]);
	}
	auto put = [](const PairSeq &q, int avg_, int *o) {
		const PairOutcome r = eval_pair_seq(q, avg_);
		o[0] = r.found ? 1 : 0; o[1] = r.wa; o[2] = r.wb; o[3] = r.equal; o[4] = r.dist; o[5] = (int) q.ps.size();
	};
	put(all, avg, out_all); put(kept, avg, out_kept);
	return 0;
}

int ngm_debug_select_top1(int device, int n_reads, const uint32_t *base, const uint32_t *count, uint64_t n_cand, const float *scores, const uint32_t *loc,
		const uint32_t *strand_votes, uint32_t *winner, int32_t *mapq, int32_t *n_best, float *best_score) {
	if (n_reads <= 0 || !base || !count || !winner || !mapq || !n_best || !best_score) return -22;
	DevGuard g(device);
	ngm::DevBuf<uint32_t> d_base, d_count, d_loc, d_sv, d_win;
	ngm::DevBuf<int32_t> d_mq, d_nb;
	ngm::DevBuf<float> d_sc, d_best;
	const size_t nc = (size_t) std::max<uint64_t>(n_cand, 1);
	if (d_base.reserve(n_reads) || d_count.reserve(n_reads) || d_loc.reserve(nc) || d_sv.reserve(nc) || d_sc.reserve(nc) || d_win.reserve(n_reads) || d_mq.reserve(n_reads) ||
			d_nb.reserve(n_reads) || d_best.reserve(n_reads)) { ngm::pipeline_set_error("out of device memory (ngm_debug_select_top1)"); return -12; }
	MAP_HIP_TRY(hipMemcpy(d_base.p, base, (size_t) n_reads * 4, hipMemcpyHostToDevice));
	MAP_HIP_TRY(hipMemcpy(d_count.p, count, (size_t) n_reads * 4, hipMemcpyHostToDevice));
	if (n_cand) {
		MAP_HIP_TRY(hipMemcpy(d_sc.p, scores, (size_t) n_cand * 4, hipMemcpyHostToDevice));
		MAP_HIP_TRY(hipMemcpy(d_loc.p, loc, (size_t) n_cand * 4, hipMemcpyHostToDevice));
		MAP_HIP_TRY(hipMemcpy(d_sv.p, strand_votes, (size_t) n_cand * 4, hipMemcpyHostToDevice));
	}
	hipLaunchKernelGGL(ngm::select_top1_kernel, dim3((n_reads + 255) / 256), dim3(256), 0, 0, n_reads, d_base.p, d_count.p, d_sc.p, d_loc.p, d_sv.p, d_win.p, d_mq.p, d_nb.p, d_best.p);
	MAP_HIP_TRY(hipGetLastError());
	MAP_HIP_TRY(hipDeviceSynchronize());
	MAP_HIP_TRY(hipMemcpy(winner, d_win.p, (size_t) n_reads * 4, hipMemcpyDeviceToHost));
	MAP_HIP_TRY(hipMemcpy(mapq, d_mq.p, (size_t) n_reads * 4, hipMemcpyDeviceToHost));
	MAP_HIP_TRY(hipMemcpy(n_best, d_nb.p, (size_t) n_reads * 4, hipMemcpyDeviceToHost));
	MAP_HIP_TRY(hipMemcpy(best_score, d_best.p, (size_t) n_reads * 4, hipMemcpyDeviceToHost));
	d_base.release(); d_count.release(); d_loc.release(); d_sv.release(); d_win.release(); d_mq.release(); d_nb.release(); d_sc.release(); d_best.release();
	return 0;
}

int ngm_mapper_heavy_counters(ngm_mapper *m, uint64_t out[4]) {
	if (!m || !out) return -22;
	out[0] = m->st_heavy_second; out[1] = m->st_heavy_restart; out[2] = m->st_heavy_sent_on; out[3] = m->st_pool_regrown;
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
