// mapper_search.cpp -- the candidate-search stage of a mapper (CS::RunBatch, src/CS.cpp:341-436) and the replay of the reference's
// candidate order (CS::AddLocationStd's rList, src/CS.cpp:196-211) for the reads where it decides a tie.
//
// One search = ONE enqueue (round 6): fast path for every read -> the queue of the reads it hands on is dealt into the classes of
// cs_heavy2_kernel ON THE DEVICE (cs_queue_device.h) -> the classes, from device-side lists -> what they cannot certify once more in
// the largest class -> exact search with the table in LDS -> exact search with tables in a pool of global memory -> compaction of the
// candidate regions; every pass is launched with a fixed grid and reads its item count from a device word, and the host synchronises
// once, at the end.  (Until round 5 the host downloaded the queue after every pass: five to six synchronisations per batch on a
// GRCh38-like genome, each with the GPU idle while the instance held its turn.)  Bisulfite and weighted SLAM-seq searches keep their
// host-driven passes (they have no fast path and their queues are sized per k-mer variant).
#include "mapper_internal.h"
#include <rocprim/rocprim.hpp>
#define NGM_CS_KERNELS
#include "cs_canon_device.h"
#include "cs_heavy_device.h"
#include "cs_order_bucket_device.h"
#include "cs_slam_device.h"
#include "cs_queue_device.h"

namespace ngm {

namespace {

// canonical fast path: waves per read and chunk-item rounds of the kernel shapes (cs_canon_device.h)
constexpr int kCanonT[4] = {0, 3, 3, 4}, kCanonR1[4] = {0, 4, 6, 8}, kCanonR2[4] = {0, 2, 2, 4};
size_t cs_canon_lds_bytes(const CsArgs &A, int shape) {  // k-mer info + headers, codes, chunk items (16-bit), plane, table, queue (+ the kernel's static variables)
	const size_t w = (size_t) A.lists_cap + (A.q + 3) / 4 + (size_t) kCanonR2[shape] * kCanonT[shape] * 64 / 2 + ((size_t) A.plane_bits >> 5) + ((size_t) 2 << A.log2_slots) +
			((size_t) 3 << A.log2_slots) / 4 + 96;  // (+ slack: the per-wave k-mer rows round up, the static variables)
	return w * 4;
}

// shape 1-3 (waves per read by the number of k-mers); shape 2 -- 150 bp reads -- with the plane word's byte address as one mask of the
// diagonal when the bins are four bases wide.  (The variants of rounds 3-4 -- chunk loads issued after another vote step, 5 / 6 / 8
// waves per SIMD -- were all measured slower and are gone: DESIGN.md 4.)
const void *cs_canon_fn(int shape, int bin_shift) {
	if (shape == 1) return (const void *) cs_canon_kernel<3, 4, 2, 1>;
	if (shape == 3) return (const void *) cs_canon_kernel<4, 8, 4, 1>;
	return bin_shift == 2 ? (const void *) cs_canon_kernel<3, 6, 2, 1, 7, true> : (const void *) cs_canon_kernel<3, 6, 2, 1, 7>;
}

size_t cs_lds_bytes(const CsArgs &A, int mode) {
	size_t w = (size_t) A.lists_cap * 2 + 1 + (A.q + 3) / 4;
	if (mode == kCsFast)  // list starts (32-bit) + lengths (16-bit), codes, plane, items, table, queue
		w = (size_t) A.lists_cap + (size_t) A.lists_cap / 2 + (A.q + 3) / 4 + ((size_t) A.plane_bits >> 5) + (size_t) A.fast_items * 64 / (A.items16 ? 2 : 1) +
				((size_t) 3 << A.log2_slots) / 4 + 32;  // + the kernels' static variables (<= 128 bytes): this is what the occupancy math sees
	if (mode != kCsFast && A.bs) w += (size_t) A.q + 1 + kCsBsChunk / 2;  // l_vbase, l_vpos
	if (mode != kCsExactGlobal) w += (size_t) 2 << A.log2_slots;
	return w * 4;
}

constexpr size_t kLdsLimit = 160 * 1024;      // of a CU (gfx950)
constexpr int kLdsAttr = 150 * 1024;          // what hipFuncAttributeMaxDynamicSharedMemorySize is set to for the kernels sized at run time

// NGM_HIP_TEST_LIMITS="key=value,key=value": small tables / pools so that the paths a GRCh38-sized genome takes -- table passes that
// start over, reads that reach cs_global_kernel, crowded buckets in the order replay, a pool too small for a read -- run on a test
// genome (tests/test_gpu_humanlike.py).  Never set in production; every key only makes a limit smaller.
long test_limit(const char *key, long dflt) {
	static const char *env = getenv("NGM_HIP_TEST_LIMITS");
	if (!env) return dflt;
	const size_t kl = strlen(key);
	for (const char *c = env; *c;) {
		if (!strncmp(c, key, kl) && c[kl] == '=') return atol(c + kl + 1);
		c = strchr(c, ',');
		if (!c) break;
		++c;
	}
	return dflt;
}

// the classes of cs_heavy2_kernel: reads of up to max_hits index hits start in the class; counters, table slots, threads, survivors
// the scratch slice holds, table passes a read may take (the largest class)
struct HeavyClass { uint32_t max_hits; int log2c, log2s, nt; uint32_t scratch_cap; const void *fn; uint32_t max_parts; };
// (limits measured on the heavy-tailed probe, per 262 144 reads: 16 384 / 32 768 / rest 18.3 ms; 16 384 / 65 536 / rest 16.1; 16 384 / all the
// rest in the middle class -- two workgroups per CU -- and the largest class only for what that cannot certify: 15.1)
const HeavyClass *heavy_classes() {
	static HeavyClass cl[3] = {{16384u, 13, 11, 256, 16384u, (const void *) cs_heavy2_kernel<256>, 1u}, {0xFFFFFFFEu, 14, 12, 512, 262144u, (const void *) cs_heavy2_kernel<512>, 1u},
			{0xFFFFFFFFu, 15, 13, 1024, 1u << 20, (const void *) cs_heavy2_kernel<1024>, 32u}};
	static const bool once = [] {
		const long shrink_c = test_limit("heavy_log2c", 0), shrink_s = test_limit("heavy_log2s", 0);   // log2 of the SMALLEST class's counters / slots; the others follow
		if (shrink_c > 0) for (int c = 0; c < 3; ++c) cl[c].log2c = (int) std::min<long>(cl[c].log2c, std::max<long>(6, shrink_c + c));
		if (shrink_s > 0) for (int c = 0; c < 3; ++c) cl[c].log2s = (int) std::min<long>(cl[c].log2s, std::max<long>(6, shrink_s + c));
		const long m0 = test_limit("heavy_max0", 0), m1 = test_limit("heavy_max1", 0);
		if (m0 > 0) { cl[0].max_hits = (uint32_t) m0; cl[0].scratch_cap = std::min<uint32_t>(cl[0].scratch_cap, (uint32_t) m0); }
		if (m1 > 0 && m1 >= (long) cl[0].max_hits) cl[1].max_hits = (uint32_t) m1;
		const long c1c = test_limit("heavy_c1_log2c", 0), c1s = test_limit("heavy_c1_log2s", 0);   // (experiments: the middle class alone)
		if (c1c > 0) cl[1].log2c = (int) std::min<long>(cl[1].log2c, std::max<long>(6, c1c));
		if (c1s > 0) cl[1].log2s = (int) std::min<long>(cl[1].log2s, std::max<long>(6, c1s));
		const long sc = test_limit("heavy_scratch", 0);
		if (sc > 0) for (int c = 1; c < 3; ++c) cl[c].scratch_cap = std::min<uint32_t>(cl[c].scratch_cap, (uint32_t) sc << (c - 1));
		return true; }();
	(void) once;
	return cl;
}

}  // namespace

// ---- part of ngm_mapper_create: geometry of the fast path, kernel attributes --------------------------------------------------------
int cs_configure(ngm_mapper *m, const ngm_mapper_params *p) {
	const ngm_ref *ref = m->ref;
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
		m->cs_fast_items = segs * 1.10 > 64.0 * kCsFastItemsShort ? kCsFastItemsLong : kCsFastItemsShort;
		if (const char *e = getenv("NGM_HIP_CS_FAST_ITEMS")) m->cs_fast_items = atoi(e) > kCsFastItemsShort ? kCsFastItemsLong : kCsFastItemsShort;  // tests
		m->cs_plane_bits0 = m->cs_plane_bits;
		// The kernel is bound by reads in flight per CU, and those by LDS, which gfx950 hands out in granules of
		// 1 280 bytes (160 KB / 128: hipOccupancyMaxActiveBlocksPerMultiprocessor reports 9 workgroups of 16 328 bytes per CU and 10
		// of 15 360).  The plane is sized generously (12 bits per expected hit): when giving up at most a fifth of it (never below
		// 10 bits per hit -- the spurious table entries H^2 / 2P stay far from the table's capacity) lets one more read in, do it.
		{
			CsArgs G{};
			G.lists_cap = 2 * std::max(1, p->qry_max_len - ref->prm.kmer + 1); G.q = p->qry_max_len; G.log2_slots = m->cs_log2_small;
			G.fast_items = m->cs_fast_items; G.items16 = (G.lists_cap <= 512) ? 1 : 0; G.plane_bits = m->cs_plane_bits;
			const size_t granule = 1280, lds = kLdsLimit;
			const size_t bytes = (cs_lds_bytes(G, kCsFast) + granule - 1) / granule * granule;
			const size_t per_cu = lds / std::max<size_t>(bytes, 1);
			if (per_cu >= 1 && per_cu < 10) {
				const size_t target = lds / (per_cu + 1) / granule * granule;  // bytes that would let one more read in
				const size_t have = cs_lds_bytes(G, kCsFast);
				if (have > target) {
					const uint32_t cut_bits = (uint32_t) (((have - target) * 8 + 31) / 32 * 32);
					if (cut_bits <= m->cs_plane_bits / 5 && (double) (m->cs_plane_bits - cut_bits) >= 10.0 * hexp) m->cs_plane_bits -= cut_bits;
				}
			}
		}
	}
	// which index layout the fast path gathers from: canonical pair buckets (odd k, up to 256 k-mers per read, k-mer pairs in
	// use shorter than 1 000 hits: the chunk items are 16-bit), else one bucket per k-mer
	{
		const int n_kmers = std::max(1, p->qry_max_len - ref->prm.kmer + 1);
		const bool canon_ok = (ref->prm.kmer & 1) && n_kmers <= 256 && m->max_kfreq <= 1000 && !p->bs_mapping;
		if (!p->bs_mapping && ngm_ref_ensure_buckets(ref, canon_ok ? 1 : 0) != 0) return -12;   // (bisulfite mapping: exact paths only, no buckets)
		if (canon_ok) {
			const int glog = std::min(ref->cbucket_log2_words, 5) - 2;
			int shape = 1;
			while (shape < 3 && (n_kmers > kCanonT[shape] * 64 || n_kmers > kCanonR1[shape] * ((kCanonT[shape] * 64) >> glog))) ++shape;
			m->cs_canon = shape;
			// the canonical kernel indexes its plane with the low bits of the bin: a power of two of bits, at least 10 per expected hit
			// (150 bp reads at GRCh38 size: 65 536 bits; with the rest of a read's LDS 16 KB -> 13 granules of 1 280 bytes, nine reads per CU)
			uint32_t pb = 4096;
			while ((double) pb < 10.0 * m->cs_hexp && pb < 131072u) pb <<= 1;
			m->cs_plane_bits = pb;
		}
	}
	CsArgs A{}; A.lists_cap = 2 * std::max(1, p->qry_max_len - ref->prm.kmer + 1); A.q = p->qry_max_len;
	A.log2_slots = m->cs_log2_slots;
	if (cs_lds_bytes(A, kCsExactLds) > 158 * 1024) { m->cs_log2_slots = 13; A.log2_slots = 13; } A.log2_bits = 17; A.plane_bits = 131072;
	A.fast_items = kCsFastItemsLong;
	A.items16 = 0;
	const int log2_exact = A.log2_slots;
	A.log2_slots = 12;  // the largest table the fast path picks (above)
	(void) hipFuncSetAttribute((const void *) cs_fast_kernel<kCsFastItemsShort, uint16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) cs_lds_bytes(A, kCsFast));
	(void) hipFuncSetAttribute((const void *) cs_fast_kernel<kCsFastItemsLong, uint16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) cs_lds_bytes(A, kCsFast));
	(void) hipFuncSetAttribute((const void *) cs_fast_kernel<kCsFastItemsLong, uint32_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) cs_lds_bytes(A, kCsFast));
#define NGM_CS_ATTR_T(T) \
	(void) hipFuncSetAttribute((const void *) cs_fast2_kernel<T, kCsFastItemsShort / T, uint16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) cs_lds_bytes(A, kCsFast)); \
	(void) hipFuncSetAttribute((const void *) cs_fast2_kernel<T, kCsFastItemsLong / T, uint16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) cs_lds_bytes(A, kCsFast))
	NGM_CS_ATTR_T(2); NGM_CS_ATTR_T(3); NGM_CS_ATTR_T(4);
#undef NGM_CS_ATTR_T
	for (int shape = 1; shape <= 3; ++shape) for (int bs : {0, 2})
		(void) hipFuncSetAttribute(cs_canon_fn(shape, bs), hipFuncAttributeMaxDynamicSharedMemorySize, (int) cs_canon_lds_bytes(A, shape));
	// three waves per read for the 768-segment size (150 bp reads), four for the 1 536-segment one (250 bp: 12.3 instead of 14.7 ms
	// per 524 288 reads -- with twice the work items per read the fourth wave pays for the seventh-of-a-CU it costs)
	m->cs_waves = m->cs_fast_items == kCsFastItemsLong ? 4 : 3;
	if (const char *e = getenv("NGM_HIP_CS_WAVES")) m->cs_waves = std::min(4, std::max(1, atoi(e)));
	A.log2_slots = log2_exact;
	(void) hipFuncSetAttribute((const void *) cs_heavy2_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsAttr);
	(void) hipFuncSetAttribute((const void *) cs_heavy2_kernel<512>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsAttr);
	(void) hipFuncSetAttribute((const void *) cs_heavy2_kernel<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsAttr);
	(void) hipFuncSetAttribute((const void *) cs_order_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsAttr);  // per device
	(void) hipFuncSetAttribute((const void *) cs_order_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsAttr);
	if (p->bs_mapping) { A.bs = 1; A.lists_cap = 2 * kCsBsChunk; A.log2_slots = std::min(A.log2_slots, 13); }
	(void) hipFuncSetAttribute((const void *) cs_kernel<kCsExactLds>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) cs_lds_bytes(A, kCsExactLds));
	(void) hipFuncSetAttribute((const void *) cs_kernel<kCsExactGlobal>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) cs_lds_bytes(A, kCsExactGlobal));
	(void) hipFuncSetAttribute((const void *) cs_global_kernel<512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) cs_lds_bytes(A, kCsExactGlobal));
	(void) hipGetLastError();  // a refused attribute shows up as a launch failure where it matters, not as a stale error at the next check
	if (hipDeviceGetAttribute(&m->cus, hipDeviceAttributeMultiprocessorCount, ref->device) != hipSuccess || m->cus < 1) m->cus = 256;
	return 0;
}

void cs_release(ngm_mapper *m) {
	m->d_read_len.release(); m->d_cand_base.release(); m->d_cand_count.release(); m->d_out_loc.release(); m->d_out_sv.release();
	m->d_status.release(); m->d_ovf_read.release(); m->d_ovf_read2.release(); m->d_ovf_hits.release(); m->d_ovf_hits2.release(); m->d_ovf_log2.release(); m->d_ovf_off.release(); m->d_gt_keys.release();
	m->d_gt_votes.release(); m->d_heavy_list.release(); m->d_heavy_ctr.release(); m->d_max_votes.release(); m->d_max_both.release(); m->d_total.release();
	m->d_counters.release(); m->d_heavy_diag.release(); m->d_order_list.release(); m->d_cand_rank.release(); m->d_order_scratch.release(); m->d_order_info.release(); m->p_order_info.release();
	m->d_order_big.release(); m->d_order_gt.release(); m->d_order_log2.release(); m->d_order_off.release(); m->p_rank.release(); m->h_base.b.release(); m->h_count.b.release(); m->h_maxv.b.release();
	m->d_out_loc2.release(); m->d_out_sv2.release(); m->d_new_base.release(); m->d_scan_tmp.release(); m->p_cs_status.release();
}

// candidate search for the reads already in m->d_reads; leaves per-read base/count/max votes and the
// candidate arrays in HBM (and base/count/max votes on the host)
int run_cs(ngm_mapper *m, int n, GpuStage *stage) {
	const ngm_ref *r = m->ref;
	auto hold = [&] { if (stage) stage->acquire(); };       // before kernels are enqueued
	const int q = m->prm.qry_max_len;
	if (n <= 0) { m->n_reads = 0; m->n_cand = 0; return 0; }
	const size_t ctr_words = (size_t) kCsRegions * kCsCursorStride;
	if (m->d_read_len.reserve(n) || m->d_cand_base.reserve(n) || m->d_cand_count.reserve(n) || m->d_max_votes.reserve(n) || m->d_max_both.reserve(n) ||
			m->d_status.reserve(16) || m->d_total.reserve(ctr_words + 16) || m->d_counters.reserve(ctr_words + 32) || m->d_new_base.reserve(n) || m->d_ovf_read.reserve(n) || m->d_ovf_read2.reserve(n) ||
			m->d_ovf_hits.reserve(n) || m->d_ovf_hits2.reserve(n) || m->d_heavy_ctr.reserve(kCsqWords) || m->p_cs_status.reserve(16 + kCsqWords + 2) ||
			m->h_base.b.reserve(n) || m->h_count.b.reserve(n) || m->h_maxv.b.reserve(n)) {
		pipeline_set_error("out of memory (candidate search, %d reads)", n);
		return -12;
	}
	size_t cap = std::max<size_t>(m->cs_region_cap, (size_t) n * 4 + 64 * kCsRegions);
	cap = (cap + kCsRegions - 1) / kCsRegions * kCsRegions;
	const size_t fixed_slots = getenv("NGM_HIP_CS_NO_FIXED_SLOTS") ? 0 : (size_t) n * kCsFixedSlots;
	static const bool host_timing = getenv("NGM_HIP_HOST_TIMING") != nullptr;
	static const bool phases = getenv("NGM_HIP_CS_PHASES") != nullptr;
	const bool bs = m->prm.bs_mapping != 0;
	const bool slamw = (m->prm.slam_seq & 4) != 0;
	for (int attempt = 0; attempt < 8; ++attempt) {
		// candidate offsets are 32-bit (base = region * capacity + cursor; the prefix sums over the counts)
		if (cap + fixed_slots >= 0xFFFFFFFFull) { pipeline_set_error("more than 2^32 candidate slots needed for %d reads: use smaller batches or a higher sensitivity", n); return -75; }
		if (m->d_out_loc.reserve(cap + fixed_slots) || m->d_out_sv.reserve(cap + fixed_slots) || m->d_out_loc2.reserve(cap + fixed_slots) || m->d_out_sv2.reserve(cap + fixed_slots)) { pipeline_set_error("out of device memory (candidates)"); return -12; }
		size_t tmp_bytes = 0;
		(void) rocprim::exclusive_scan(nullptr, tmp_bytes, m->d_cand_count.p, m->d_new_base.p, 0u, (size_t) n, rocprim::plus<uint32_t>(), m->st);
		if (m->d_scan_tmp.reserve(tmp_bytes + 16)) { pipeline_set_error("out of device memory (scan)"); return -12; }
		hold();
		MAP_HIP_TRY(hipMemsetAsync(m->d_status.p, 0, 64, m->st));
		MAP_HIP_TRY(hipMemsetAsync(m->d_total.p, 0, (ctr_words + 16) * 8, m->st));
		MAP_HIP_TRY(hipMemsetAsync(m->d_counters.p, 0, (ctr_words + 32) * 8, m->st));
		MAP_HIP_TRY(hipMemsetAsync(m->d_heavy_ctr.p, 0, kCsqWords * 4, m->st));
		MAP_HIP_TRY(hipMemsetAsync(m->d_cand_count.p, 0, (size_t) n * 4, m->st));   // (a read no pass has finished -- a pass skipped for want of pool room -- has no candidates yet)
		CsArgs A{};
		A.reads = m->d_reads.p; A.n = n; A.q = q; A.k = r->prm.kmer; A.bin_shift = r->prm.bin_size;
		A.max_kfreq = m->max_kfreq; A.sensitivity = m->prm.sensitivity; A.kmer_min = m->prm.kmer_min; A.max_cmrs = m->prm.max_cmrs;
		A.index = r->d_index; A.positions = r->d_positions;
		A.lists_cap = 2 * std::max(1, q - r->prm.kmer + 1);
		A.read_len = m->d_read_len.p; A.cand_base = m->d_cand_base.p; A.cand_count = m->d_cand_count.p; A.max_votes = m->d_max_votes.p; A.max_both = m->d_max_both.p;
		A.out_loc = m->d_out_loc.p; A.out_sv = m->d_out_sv.p; A.out_total = m->d_total.p; A.out_capacity = cap / kCsRegions;
		A.fixed_base = fixed_slots ? (uint32_t) cap : 0u;
		A.status = m->d_status.p; A.ovf_read = m->d_ovf_read.p; A.ovf_hits = m->d_ovf_hits.p; A.counters = m->d_counters.p;
		A.phase_cycles = phases ? m->d_counters.p + ctr_words : nullptr;
		uint32_t *const status = m->p_cs_status.p;           // [0..15] the status block, [16..] the control block, as downloaded
		uint32_t *const ctl = status + 16;
		memset(status, 0, (16 + kCsqWords + 2) * 4);
		m->cs_kernel_ms = 0;
		float pass_ms[3] = {0, 0, 0};
		auto timed = [&](int e) { float t = 0; if (hipEventElapsedTime(&t, m->cev[e], m->cev[e + 1]) == hipSuccess) { m->cs_kernel_ms += t; pass_ms[e / 2] = t; } };
		auto fetch_status = [&]() -> int {   // the status block -> host (synchronises the stream)
			MAP_HIP_TRY(hipMemcpyAsync(status, m->d_status.p, 64, hipMemcpyDeviceToHost, m->st));
			MAP_HIP_TRY(hipStreamSynchronize(m->st));
			return 0;
		};
		// exact search with per-read tables in a pool of global memory, for the reads of the queue behind status block `blk` (their
		// hits in `q_hits`, the reads in `q_read`): tables sized on the device; a pool too small is reported in the status block
		// (the pool starts small -- on most genomes no read ever gets here -- and keeps the size the largest batch so far needed)
		unsigned long long pool_slots = std::max<unsigned long long>(m->d_gt_votes.cap, 1ull << std::max<long>(10, test_limit("gtable_pool_log2", 22)));
		auto enqueue_global = [&](const uint32_t *q_read, const uint32_t *q_hits) -> int {
			if (m->d_gt_keys.reserve(pool_slots) || m->d_gt_votes.reserve(pool_slots) || m->d_ovf_off.reserve(n) || m->d_ovf_log2.reserve(n)) {
				pipeline_set_error("out of device memory (overflow vote tables, %llu slots)", pool_slots);
				return -12;
			}
			hipLaunchKernelGGL(cs_global_prepare_kernel, dim3(1), dim3(1024), 0, m->st, m->d_status.p, q_hits, m->d_ovf_off.p, m->d_ovf_log2.p, pool_slots);
			CsArgs G = A;
			G.read_list = q_read; G.n_list_dev = m->d_status.p + kCsqGlobalCount;
			G.ovf_table_off = m->d_ovf_off.p; G.ovf_log2 = m->d_ovf_log2.p; G.gtable_keys = m->d_gt_keys.p; G.gtable_votes = m->d_gt_votes.p;
			G.status = m->d_status.p + 8;   // (a block of its own: it queues nothing, only the output-overflow flag)
			// one workgroup per read (cs_global_kernel, cs_heavy_device.h), striding over the list
			hipLaunchKernelGGL(cs_global_kernel<512>, dim3(std::min(n, 4 * m->cus)), dim3(512), cs_lds_bytes(G, kCsExactGlobal), m->st, G);
			MAP_HIP_TRY(hipGetLastError());
			return 0;
		};
		// regions -> one dense candidate array in read order, the per-read arrays and the counters -> host
		std::vector<unsigned long long> ctr(ctr_words + 16);
		uint32_t last[2] = {0, 0};
		auto enqueue_finish = [&]() -> int {
			MAP_HIP_TRY(rocprim::exclusive_scan(m->d_scan_tmp.p, tmp_bytes, m->d_cand_count.p, m->d_new_base.p, 0u, (size_t) n, rocprim::plus<uint32_t>(), m->st));
			hipLaunchKernelGGL(compact_candidates_kernel, dim3((n + 255) / 256), dim3(256), 0, m->st, n, (const uint32_t *) m->d_status.p, m->d_cand_base.p, m->d_new_base.p, m->d_cand_count.p,
					m->d_out_loc.p, m->d_out_sv.p, m->d_out_loc2.p, m->d_out_sv2.p);
			MAP_HIP_TRY(hipGetLastError());
			MAP_HIP_TRY(hipMemcpyAsync(status, m->d_status.p, 64, hipMemcpyDeviceToHost, m->st));
			MAP_HIP_TRY(hipMemcpyAsync(ctl, m->d_heavy_ctr.p, kCsqWords * 4, hipMemcpyDeviceToHost, m->st));
			MAP_HIP_TRY(hipMemcpyAsync(ctr.data(), m->d_counters.p, ctr.size() * 8, hipMemcpyDeviceToHost, m->st));
			// the host needs the number of candidates now (it sizes the score stage); the per-read arrays (12 bytes per read) only after the score stage
			MAP_HIP_TRY(hipMemcpyAsync(&last[0], m->d_new_base.p + (n - 1), 4, hipMemcpyDeviceToHost, m->st));
			MAP_HIP_TRY(hipMemcpyAsync(&last[1], m->d_cand_count.p + (n - 1), 4, hipMemcpyDeviceToHost, m->st));
			MAP_HIP_TRY(hipMemcpyAsync(m->h_base.data(), m->d_new_base.p, (size_t) n * 4, hipMemcpyDeviceToHost, m->st));
			MAP_HIP_TRY(hipMemcpyAsync(m->h_count.data(), m->d_cand_count.p, (size_t) n * 4, hipMemcpyDeviceToHost, m->st));
			MAP_HIP_TRY(hipMemcpyAsync(m->h_maxv.data(), m->d_max_votes.p, (size_t) n * 4, hipMemcpyDeviceToHost, m->st));
			return 0;
		};
		uint32_t n_heavy = 0, n_exact_lds = 0, n_exact_global = 0;

		A.bs = bs ? 1 : 0; A.bs_cutoff = m->prm.bs_cutoff; A.bs_read_skip = std::max(0, m->prm.bs_read_skip); A.bs_paired = m->cs_paired ? 1 : 0;
		if (bs) A.lists_cap = 2 * kCsBsChunk;  // the exact kernels hold the lists of kCsBsChunk k-mer variants at a time
		A.log2_bits = m->cs_log2_bits; A.plane_bits = m->cs_plane_bits; A.log2_slots = m->cs_log2_small; A.fast_items = m->cs_fast_items;
		A.buckets = r->d_buckets; A.bucket_log2_words = r->bucket_log2_words; A.pos_base = r->bucket_pos_base;
		A.hit_cap = m->cs_plane_bits / 6u;
		if (A.bin_shift < 2) A.hit_cap = 0;  // the register encoding of the fast path keeps bins in 30 bits
		MAP_HIP_TRY(hipEventRecord(m->cev[0], m->st));
		if (slamw) {
			// `--slam-seq` with bit 2: the weighted search (cs_slam_device.h) -- float votes in the reference's order, one wave per read,
			// tables in slices of global memory.  Persistent workgroups with a slice each; reads whose hits outgrow it are queued and re-run
			// with a slice of their own.
			A.bs = 2; A.bs_cutoff = 0; A.bs_read_skip = 0; A.bs_paired = m->cs_paired ? 1 : 0;
			A.lists_cap = 2 * kCsBsChunk;
			const size_t lds = cs_lds_bytes(A, kCsExactGlobal);
			const int grid = std::min(n, 2048);
			A.slam_slice_words = cs_slam_words(49152u);
			if (m->d_gt_keys.reserve((size_t) grid * A.slam_slice_words)) { pipeline_set_error("out of device memory (weighted SLAM-seq search)"); return -12; }
			A.gtable_keys = m->d_gt_keys.p;
			hipLaunchKernelGGL(cs_slam_kernel, dim3(grid), dim3(64), lds, m->st, A);
			MAP_HIP_TRY(hipGetLastError());
			if (int rc = fetch_status()) return rc;
			if (status[1] > 0) {
				const uint32_t no = status[1];
				std::vector<uint32_t> qr(no), qh(no), lg(no);
				std::vector<uint64_t> off(no);
				MAP_HIP_TRY(hipMemcpy(qr.data(), m->d_ovf_read.p, (size_t) no * 4, hipMemcpyDeviceToHost));
				MAP_HIP_TRY(hipMemcpy(qh.data(), m->d_ovf_hits.p, (size_t) no * 4, hipMemcpyDeviceToHost));
				if (m->d_ovf_off.reserve(no) || m->d_ovf_log2.reserve(no)) { pipeline_set_error("out of device memory (weighted SLAM-seq search)"); return -12; }
				constexpr uint64_t kPoolWords = 1ull << 30;
				for (uint32_t j0 = 0; j0 < no;) {
					uint64_t total = 0;
					uint32_t j1 = j0;
					while (j1 < no && (j1 == j0 || total + cs_slam_words(qh[j1]) <= kPoolWords)) { off[j1] = total; lg[j1] = cs_slam_log2_slots(qh[j1]); total += cs_slam_words(qh[j1]); ++j1; }
					if (m->d_gt_keys.reserve(total)) { pipeline_set_error("out of device memory (weighted SLAM-seq search, %llu words)", (unsigned long long) total); return -12; }
					MAP_HIP_TRY(hipMemcpyAsync(m->d_ovf_read2.p, qr.data() + j0, (size_t) (j1 - j0) * 4, hipMemcpyHostToDevice, m->st));
					MAP_HIP_TRY(hipMemcpyAsync(m->d_ovf_log2.p, lg.data() + j0, (size_t) (j1 - j0) * 4, hipMemcpyHostToDevice, m->st));
					MAP_HIP_TRY(hipMemcpyAsync(m->d_ovf_off.p, off.data() + j0, (size_t) (j1 - j0) * 8, hipMemcpyHostToDevice, m->st));
					CsArgs Q = A;
					Q.read_list = m->d_ovf_read2.p; Q.ovf_log2 = m->d_ovf_log2.p; Q.ovf_table_off = m->d_ovf_off.p; Q.gtable_keys = m->d_gt_keys.p;
					hipLaunchKernelGGL(cs_slam_kernel, dim3(j1 - j0), dim3(64), lds, m->st, Q);
					MAP_HIP_TRY(hipGetLastError());
					MAP_HIP_TRY(hipStreamSynchronize(m->st));
					j0 = j1;
				}
			}
			MAP_HIP_TRY(hipEventRecord(m->cev[1], m->st));
			if (int rc = enqueue_finish()) return rc;
			MAP_HIP_TRY(hipStreamSynchronize(m->st));
			timed(0);
		} else if (bs) {
			// bisulfite mapping: no fast path (a read looks up ~10 variants of every k-mer: exact tables only): every read goes through the
			// exact kernel with the table in LDS; reads with more hits than that takes are queued for the tables in global memory
			MAP_HIP_TRY(hipEventRecord(m->cev[1], m->st));
			CsArgs B = A;
			B.log2_slots = std::min(m->cs_log2_slots, 13);
			B.hit_cap = (uint32_t) ((1u << B.log2_slots) * 0.66f);
			B.read_list = nullptr;
			B.status = m->d_status.p + kCsqStatusExact; B.ovf_read = m->d_ovf_read2.p; B.ovf_hits = m->d_ovf_hits2.p;
			MAP_HIP_TRY(hipEventRecord(m->cev[2], m->st));
			hipLaunchKernelGGL(cs_kernel<kCsExactLds>, dim3(n), dim3(64), cs_lds_bytes(B, kCsExactLds), m->st, B);
			MAP_HIP_TRY(hipGetLastError());
			MAP_HIP_TRY(hipEventRecord(m->cev[3], m->st));
			n_exact_lds = (uint32_t) n;
			if (int rc = fetch_status()) return rc;
			if (status[kCsqStatusExact + 1] > 0) {
				// (their lists come in chunks of variants: the one-wave kernel; tables sized by the host)
				const uint32_t no = status[kCsqStatusExact + 1];
				n_exact_global = no;
				std::vector<uint32_t> hits(no), lg(no);
				std::vector<uint64_t> off(no);
				MAP_HIP_TRY(hipMemcpy(hits.data(), m->d_ovf_hits2.p, (size_t) no * 4, hipMemcpyDeviceToHost));
				uint64_t total_slots = 0;
				for (uint32_t i = 0; i < no; ++i) {
					uint32_t l = 4;
					while ((1ull << l) < 2ull * hits[i]) ++l;
					lg[i] = l; off[i] = total_slots; total_slots += 1ull << l;
				}
				if (m->d_gt_keys.reserve(total_slots) || m->d_gt_votes.reserve(total_slots) || m->d_ovf_off.reserve(no) || m->d_ovf_log2.reserve(no)) {
					pipeline_set_error("out of device memory (overflow vote tables, %llu slots)", (unsigned long long) total_slots);
					return -12;
				}
				MAP_HIP_TRY(hipMemcpyAsync(m->d_ovf_off.p, off.data(), (size_t) no * 8, hipMemcpyHostToDevice, m->st));
				MAP_HIP_TRY(hipMemcpyAsync(m->d_ovf_log2.p, lg.data(), (size_t) no * 4, hipMemcpyHostToDevice, m->st));
				CsArgs G = A;
				G.read_list = m->d_ovf_read2.p; G.status = m->d_status.p + 8;
				G.ovf_table_off = m->d_ovf_off.p; G.ovf_log2 = m->d_ovf_log2.p; G.gtable_keys = m->d_gt_keys.p; G.gtable_votes = m->d_gt_votes.p;
				MAP_HIP_TRY(hipEventRecord(m->cev[4], m->st));
				hipLaunchKernelGGL(cs_kernel<kCsExactGlobal>, dim3(no), dim3(64), cs_lds_bytes(G, kCsExactGlobal), m->st, G);
				MAP_HIP_TRY(hipGetLastError());
				MAP_HIP_TRY(hipEventRecord(m->cev[5], m->st));
				MAP_HIP_TRY(hipStreamSynchronize(m->st));   // (off, lg live until here)
			}
			if (int rc = enqueue_finish()) return rc;
			MAP_HIP_TRY(hipStreamSynchronize(m->st));
			timed(2);
			if (n_exact_global) timed(4);
		} else {
			// pass 1 -- FAST path for every read (bit-plane filter + small exact table, many workgroups per CU)
			// 16-bit work items when every list index fits 9 bits and no used list can have more than 128 segments
			A.items16 = (A.lists_cap <= 512 && m->max_kfreq <= 128 * kCsSeg) ? 1 : 0;
			if (!A.items16) A.fast_items = kCsFastItemsLong;  // the 32-bit item list only exists in the large size
			if (m->cs_canon) {
				A.buckets = r->d_cbuckets; A.bucket_log2_words = r->cbucket_log2_words; A.pos_base = r->cbucket_pos_base;
				const size_t lds = cs_canon_lds_bytes(A, m->cs_canon) - 96;  // (the kernel has no static LDS: its shared variables are the last 160 bytes of this)
				// persistent workgroups: as many as the GPU holds at once, each walking the reads with that stride
				const void *fn = cs_canon_fn(m->cs_canon, A.bin_shift);
				int per_cu = 0;
				if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, kCanonT[m->cs_canon] * 64, lds) != hipSuccess || per_cu < 1) per_cu = 1;
				const int grid = std::min(n, per_cu * m->cus);
				void *kargs[] = {(void *) &A};
				MAP_HIP_TRY(hipLaunchKernel(fn, dim3(grid), dim3(kCanonT[m->cs_canon] * 64), kargs, lds, m->st));
				if (A.phase_cycles)
					fprintf(stderr, "[ngm-hip] cs canonical path (shape %d): %zu bytes of LDS per read, %d reads resident per CU (grid %d), bucket 2^%d words\n", m->cs_canon, lds, per_cu, grid, A.bucket_log2_words);
			}
			else if (m->cs_waves >= 2 && A.items16) {  // T waves per read: the same 768 / 1 536 segments, dealt to T * 64 lanes
				const bool shrt = A.fast_items == kCsFastItemsShort;
				const size_t lds = cs_lds_bytes(A, kCsFast) - 128;  // the kernel's static variables take the rest
#define NGM_CS_LAUNCH_T(T) \
				do { if (shrt) hipLaunchKernelGGL((cs_fast2_kernel<T, kCsFastItemsShort / T, uint16_t>), dim3(n), dim3(T * 64), lds, m->st, A); \
					else hipLaunchKernelGGL((cs_fast2_kernel<T, kCsFastItemsLong / T, uint16_t>), dim3(n), dim3(T * 64), lds, m->st, A); } while (0)
				if (m->cs_waves == 2) NGM_CS_LAUNCH_T(2); else if (m->cs_waves == 3) NGM_CS_LAUNCH_T(3); else NGM_CS_LAUNCH_T(4);
#undef NGM_CS_LAUNCH_T
			}
			else if (A.fast_items == kCsFastItemsShort && A.items16) hipLaunchKernelGGL((cs_fast_kernel<kCsFastItemsShort, uint16_t>), dim3(n), dim3(64), cs_lds_bytes(A, kCsFast), m->st, A);
			else if (A.items16) hipLaunchKernelGGL((cs_fast_kernel<kCsFastItemsLong, uint16_t>), dim3(n), dim3(64), cs_lds_bytes(A, kCsFast), m->st, A);
			else hipLaunchKernelGGL((cs_fast_kernel<kCsFastItemsLong, uint32_t>), dim3(n), dim3(64), cs_lds_bytes(A, kCsFast), m->st, A);
			MAP_HIP_TRY(hipGetLastError());
			MAP_HIP_TRY(hipEventRecord(m->cev[1], m->st));
			// pass 1b -- the reads with more hits than the fast path takes (cs_heavy2_kernel, cs_heavy_device.h): two rows of sketch counters +
			// an exact table in LDS, by hit count in three classes of persistent workgroups; pass 1c -- what the two smaller classes
			// cannot certify, once more in the largest; what is left after that is queued for the exact kernels below.  The queue is dealt
			// into the class lists on the device; a class whose list is empty costs one empty launch.
			MAP_HIP_TRY(hipEventRecord(m->cev[2], m->st));
			const HeavyClass *classes = heavy_classes();
			const bool heavy_on = A.bin_shift >= 2;   // (the kernel keeps bins in 30 bits)
			if (heavy_on) {
				const uint32_t coarse_cap = (uint32_t) cs_heavy2_coarse_cap(A.lists_cap, m->max_kfreq);
				if (m->d_heavy_list.reserve((size_t) 3 * n)) { pipeline_set_error("out of device memory (candidate search)"); return -12; }
				auto ent_cap_of = [&](int c) -> uint32_t { return classes[c].max_parts > 1 ? classes[c].max_parts * ((3u << classes[c].log2s) / 4u) : 0u; };   // (bin, votes) entries of all table passes
				int grid[3] = {0, 0, 0};
				size_t lds[3] = {0, 0, 0}, words[3] = {0, 0, 0};
				for (int c = 0; c < 3; ++c) {
					lds[c] = cs_heavy2_lds_bytes(A.lists_cap, A.q, classes[c].log2c, classes[c].log2s, coarse_cap);
					if (lds[c] > (size_t) kLdsAttr) { grid[c] = 0; continue; }   // (very long reads: the class's lists do not fit beside its table -- its reads go on to the exact kernels)
					int &per_cu = m->heavy_per_cu[c];
					if (per_cu <= 0 || m->heavy_lds[c] != lds[c]) { if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, classes[c].fn, classes[c].nt, lds[c]) != hipSuccess || per_cu < 1) per_cu = 1; m->heavy_lds[c] = lds[c]; }
					grid[c] = (int) std::min<size_t>((size_t) n, (size_t) per_cu * m->cus);
					words[c] = (size_t) grid[c] * ((size_t) classes[c].scratch_cap + 2 * (size_t) ent_cap_of(c));
				}
				// (round 0 runs classes 0 and 1 -- and 2, when the class limits leave it reads --, round 1 class 2 alone: they share the scratch)
				const bool c2_round0 = classes[1].max_hits < 0xFFFFFFFEu;
				const size_t scratch_words = std::max(words[0] + words[1] + (c2_round0 ? words[2] : 0), words[2]);
				if (m->d_gt_keys.reserve(scratch_words)) { pipeline_set_error("out of device memory (candidate search scratch, %zu words)", scratch_words); return -12; }
				if (A.phase_cycles) { if (m->d_heavy_diag.reserve(64)) return -12; MAP_HIP_TRY(hipMemsetAsync(m->d_heavy_diag.p, 0, 64 * 8, m->st)); }
				for (int round = 0; round < 2; ++round) {
					// (exclusive bounds of the classes' hit counts, see the kernel; a class without a grid -- its LDS does not fit -- takes nothing)
					unsigned long long b0 = 0, b1 = 0, b2 = 0;
					if (round == 0) {
						b0 = grid[0] ? (unsigned long long) classes[0].max_hits + 1 : 0ull;
						b1 = std::max(b0, grid[1] ? (unsigned long long) classes[1].max_hits + 1 : 0ull);
						b2 = std::max(b1, (c2_round0 && grid[2]) ? 1ull << 32 : 0ull);
					} else b2 = grid[2] ? (c2_round0 ? (unsigned long long) classes[1].max_hits + 1 : 1ull << 32) : 0ull;
					hipLaunchKernelGGL(cs_heavy_classify_kernel, dim3(1), dim3(1024), 0, m->st, m->d_status.p, m->d_ovf_read.p, m->d_ovf_hits.p, m->d_heavy_list.p, (uint32_t) n,
							m->d_heavy_ctr.p, b0, b1, b2, round);
					size_t soff = 0;
					for (int c = 2; c >= 0; --c) {   // (the largest reads first: their workgroups are the long ones)
						if (round == 0 ? (c == 2 && !c2_round0) : c != 2) continue;
						if (grid[c] == 0) continue;
						CsArgs Hv = A;
						Hv.read_list = m->d_heavy_list.p + (size_t) c * n; Hv.log2_bits = classes[c].log2c; Hv.log2_slots = classes[c].log2s;
						if (test_limit("heavy_no_short", 0)) Hv.fast_items = -1;   // (A/B: T without the lower bound from the short lists)
						uint32_t scap = classes[c].scratch_cap, ccap = coarse_cap, mparts = classes[c].max_parts, ecap = ent_cap_of(c);
						uint32_t *ctl_d = m->d_heavy_ctr.p, *scr = m->d_gt_keys.p + soff;
						int cls = c;
						unsigned long long *dg = A.phase_cycles ? m->d_heavy_diag.p + 16 * c : nullptr;
						void *kargs[] = {(void *) &Hv, (void *) &ctl_d, (void *) &cls, (void *) &scr, (void *) &scap, (void *) &ccap, (void *) &mparts, (void *) &ecap, (void *) &dg};
						MAP_HIP_TRY(hipLaunchKernel(classes[c].fn, dim3(grid[c]), dim3(classes[c].nt), kargs, lds[c], m->st));
						soff += words[c];
					}
				}
			}
			MAP_HIP_TRY(hipEventRecord(m->cev[3], m->st));
			// pass 2 -- EXACT path, table in LDS, for the reads still queued (a few per batch); what outgrows that table is queued behind the
			// second status block for pass 3 -- EXACT path with per-read tables in global memory
			MAP_HIP_TRY(hipEventRecord(m->cev[4], m->st));
			{
				CsArgs B = A;
				B.log2_slots = m->cs_log2_slots;
				B.hit_cap = (uint32_t) ((1u << B.log2_slots) * 0.66f);
				B.read_list = m->d_ovf_read.p; B.n_list_dev = m->d_status.p + 1;
				B.status = m->d_status.p + kCsqStatusExact; B.ovf_read = m->d_ovf_read2.p; B.ovf_hits = m->d_ovf_hits2.p;
				hipLaunchKernelGGL(cs_kernel<kCsExactLds>, dim3(std::min(n, 8 * m->cus)), dim3(64), cs_lds_bytes(B, kCsExactLds), m->st, B);
				MAP_HIP_TRY(hipGetLastError());
			}
			if (int rc = enqueue_global(m->d_ovf_read2.p, m->d_ovf_hits2.p)) return rc;
			MAP_HIP_TRY(hipEventRecord(m->cev[5], m->st));
			if (int rc = enqueue_finish()) return rc;
			MAP_HIP_TRY(hipStreamSynchronize(m->st));   // the ONE synchronisation of a search
			if (status[kCsqPoolFlag] && !(status[0] | status[kCsqStatusExact] | status[8])) {
				// the tables of the reads queued for global memory outgrew the pool (not seen on GRCh38-sized runs; NGM_HIP_TEST_LIMITS): a larger
				// pool, that pass and the compaction once more
				const unsigned long long need = (unsigned long long) status[kCsqPoolNeed] | ((unsigned long long) status[kCsqPoolNeed + 1] << 32);
				++m->st_pool_regrown;
				pool_slots = 1024;
				while (pool_slots < need + need / 2) pool_slots <<= 1;
				if (int rc = enqueue_global(m->d_ovf_read2.p, m->d_ovf_hits2.p)) return rc;
				if (int rc = enqueue_finish()) return rc;
				MAP_HIP_TRY(hipStreamSynchronize(m->st));
				if (status[kCsqPoolFlag]) { pipeline_set_error("candidate search: the pool of global-memory vote tables could not be sized (%llu slots)", need); return -12; }
			}
			timed(0); timed(2); timed(4);
			n_heavy = ctl[kCsqRun]; n_exact_lds = status[1]; n_exact_global = status[kCsqGlobalCount];
			m->st_heavy_second += ctl[kCsqSecond]; m->st_heavy_restart += ctl[kCsqRestart]; m->st_heavy_sent_on += ctl[kCsqSentOn];
			if (A.phase_cycles && heavy_on) {
				unsigned long long dg[64];
				MAP_HIP_TRY(hipMemcpy(dg, m->d_heavy_diag.p, sizeof(dg), hipMemcpyDeviceToHost));
				for (int c = 0; c < 3; ++c) if (dg[16 * c + 8]) {
					const double ns = (double) dg[16 * c + 8];
					fprintf(stderr, "[ngm-hip] heavy class %d (%u reads): us per sampled read: setup %.1f | sweep A %.1f | sum + T %.1f | insert / sweep B %.1f | row 2 %.1f | sweep D %.1f | candidates %.1f; hits %.0f, survivors %.0f, %.0f %% without a second row; T from the short lists %.1f; second passes %.0f %% of the reads, table passes of the partitioned reads %.1f\n",
							c, c == 2 ? ctl[kCsqRun + 1] : ctl[kCsqCount + c], dg[16 * c] / ns / 100.0, dg[16 * c + 1] / ns / 100.0, dg[16 * c + 2] / ns / 100.0, dg[16 * c + 3] / ns / 100.0, dg[16 * c + 4] / ns / 100.0,
							dg[16 * c + 5] / ns / 100.0, dg[16 * c + 6] / ns / 100.0, dg[16 * c + 9] / ns, dg[16 * c + 11] / ns, 100.0 * dg[16 * c + 10] / ns, dg[16 * c + 7] / ns, 100.0 * dg[16 * c + 13] / ns, (double) dg[16 * c + 12]);
					const unsigned long long w = dg[16 * c + 14], x = dg[16 * c + 15];
					if (w | x) fprintf(stderr, "[ngm-hip] heavy class %d sent on: %llu reads with a wrapped counter row, %llu without a T <= 255 that fits, %llu with more survivors than the slice or an overflowing table / entry list, %llu with T - 1 not below the threshold\n",
							c, w & 0xFFFFFFFFull, w >> 32, x & 0xFFFFFFFFull, x >> 32);
				}
			}
			if (host_timing) fprintf(stderr, "[ngm-hip] candidate search: fast path %.2f ms for %d reads; heavy classes %.2f ms for %u reads (+ %u once more in the largest class; second passes %u, restarts %u, sent on %u); exact kernels %.2f ms (LDS table %u reads, global-memory tables %u)\n",
					pass_ms[0], n, pass_ms[1], ctl[kCsqRun], ctl[kCsqRun + 1], ctl[kCsqSecond], ctl[kCsqRestart], ctl[kCsqSentOn], pass_ms[2], n_exact_lds, n_exact_global);
		}
		m->cs_queued_exact = n_exact_lds;
		if ((status[0] | status[kCsqStatusExact] | status[8]) == 0) {
			std::swap(m->d_out_loc, m->d_out_loc2); std::swap(m->d_out_sv, m->d_out_sv2); std::swap(m->d_cand_base, m->d_new_base);
			m->last_cs = A;
			m->last_cs.n_list_dev = nullptr;
			m->cs_region_cap = cap;
			m->n_reads = n;
			m->n_cand = (uint64_t) last[0] + last[1];
			{
				// the 32-bit prefix sums wrap silently: cross-check the total against the 64-bit candidate counters
				unsigned long long sum = 0;
				for (int g = 0; g < kCsRegions; ++g) sum += ctr[(size_t) g * kCsCursorStride + 2];
				if (sum != m->n_cand) { pipeline_set_error("%llu candidates in one batch of %d reads exceed the 32-bit candidate index: use smaller batches", sum, n); return -75; }
			}
			m->st_heavy += n_heavy; m->st_reads += (uint64_t) n; m->st_cands += m->n_cand; m->st_exact_lds += n_exact_lds; m->st_exact_global += n_exact_global;
			m->cs_kmers = m->cs_hits = 0;
			for (int g = 0; g < kCsRegions; ++g) { m->cs_kmers += ctr[(size_t) g * kCsCursorStride]; m->cs_hits += ctr[(size_t) g * kCsCursorStride + 1]; }
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
	pipeline_set_error("candidate buffer overflow persists");
	return -75;
}

// h_base / h_count / h_maxv of the last search are complete when run_cs returns (they travel with its one synchronisation)
int cs_host_arrays(ngm_mapper *) { return 0; }

// ---- candidate ORDER ----------------------------------------------------------------------------------------------------------------
namespace {
// After the LDS replay: wait for it, account for the reads it left to the exact kernel (more hits than its time line, more repeated
// bins than its table: CsArgs::order_info) and replay those exactly in global memory.  No read keeps an undetermined order silently.
int candidate_order_finish(ngm_mapper *m, hipStream_t ost, uint64_t np) {
	static const bool phases = getenv("NGM_HIP_CS_PHASES") != nullptr;
	MAP_HIP_TRY(hipStreamSynchronize(ost));
	{ float t = 0; if (hipEventElapsedTime(&t, m->oev[0], m->oev[1]) == hipSuccess) m->order_ms += t; }
	const uint32_t nl = (uint32_t) m->order_pending.size();
	m->st_order_reads += nl;
	std::vector<uint32_t> big;
	for (uint32_t i = 0; i < nl; ++i) if (m->p_order_info.p[2 * i + 1] & 0xFFu) big.push_back(i);
	m->st_order_big += big.size();
	const std::vector<uint32_t> beyond_lds = big;
	// memory the replay's scratch may take: half of what the device has free, less when a test says so
	auto scratch_room = [&](uint64_t want_bytes) -> uint64_t {
		size_t free_b = 0, total_b = 0;
		uint64_t room = want_bytes;
		if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) room = std::min<uint64_t>(room, ((uint64_t) free_b + (uint64_t) m->d_order_gt.cap * 4) / 2);
		const long kb = test_limit("order_pool_kb", 0);
		if (kb > 0) room = std::min<uint64_t>(room, (uint64_t) kb << 10);
		return room;
	};
	// The reads beyond the LDS replay: hits dealt into buckets (cs_order_bucket_kernel -- no table in global memory); what that kernel
	// leaves (bisulfite runs, a read with more hits than a slice) goes on to the replay with a table in global memory below.
	static const bool buckets_on = getenv("NGM_HIP_ORDER_NO_BUCKETS") == nullptr;
	const size_t bucket_coarse_cap = cs_heavy2_coarse_cap(m->order_args.lists_cap, m->order_args.max_kfreq);
	int bucket_log2 = (int) test_limit("order_buckets_log2", kCsOrderBucketLog2Max);   // (most buckets of a read: what the LDS holds beside the lists)
	while (bucket_log2 > 13 && cs_order_bucket_lds_bytes(m->order_args.lists_cap, m->order_args.q, bucket_coarse_cap, bucket_log2) > (size_t) kLdsAttr) --bucket_log2;
	const size_t bucket_lds = cs_order_bucket_lds_bytes(m->order_args.lists_cap, m->order_args.q, bucket_coarse_cap, std::max(bucket_log2, 6));
	const bool buckets_fit = buckets_on && !m->order_args.bs && cs_order_tau(m->order_args.lists_cap) <= kCsOrderBucketMaxTau && bucket_lds <= (size_t) kLdsAttr;
	if (!big.empty() && buckets_fit) {
		const uint32_t nb = (uint32_t) big.size();
		std::sort(big.begin(), big.end(), [&](uint32_t a, uint32_t b) { const uint32_t ha = m->p_order_info.p[2 * a], hb = m->p_order_info.p[2 * b]; return ha != hb ? ha > hb : a < b; });   // the longest first: they end the launch
		std::vector<uint32_t> reads(nb);
		for (uint32_t j = 0; j < nb; ++j) reads[j] = m->order_pending[big[j]];
		CsArgs B = m->order_args;
		B.order_info = nullptr; B.order_scratch = nullptr; B.order_max_hits = 0;
		B.order_gcap = (uint32_t) test_limit("order_bucket_fill", 4);   // (cs_order_bucket_kernel reads it as the hits per bucket it aims at)
		B.log2_bits = bucket_log2;
		auto kern = cs_order_bucket_kernel<kCsOrderBucketThreads>;
		if (bucket_lds > 64 * 1024) (void) hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int) bucket_lds);
		// ONE workgroup per CU: its eight waves leave the CU's other wave slots and LDS to the search kernels of the other mapper instances,
		// which run at the same time (measured at 3.1 Gbp, four instances: 2.63 M reads/s with one, 2.54 M with the two that fit)
		// (round 6, four instances, same box: reads drawn uniformly 3.49 / 3.48 M with one / two, stress sub-leg 1.31 / 1.24 M)
		// elements of a workgroup's slice: the most hits a read of this run can have (a list per k-mer and strand, none longer than max_kfreq) --
		// not the most of THIS list: every new maximum would be a hipFree + hipMalloc, two device-wide synchronisations, in the middle of the run
		uint64_t cap = std::min<uint64_t>(kCsOrderBucketMaxHits, (((uint64_t) (B.lists_cap / 2) * (uint64_t) std::max(B.max_kfreq, 1)) + 63) & ~63ull);
		uint32_t grid = (uint32_t) m->cus;
		{
			const uint64_t room = scratch_room(4ull << 30);
			cap = std::min<uint64_t>(cap, room / 8);                       // (a read with more hits than that goes on to the table kernel)
			grid = (uint32_t) std::max<uint64_t>(1, std::min<uint64_t>(grid, room / 8 / std::max<uint64_t>(cap, 1)));
		}
		const size_t slice_words = (size_t) grid * cap * 2;
		grid = std::min<uint32_t>(grid, nb);
		bool ok = cap >= 64 && !m->d_order_gt.reserve(slice_words) && !m->d_order_big.reserve(nb) && !m->d_order_log2.reserve(nb + 1) && !m->d_order_info.reserve(2 * (size_t) nb);
		if (ok) {
			MAP_HIP_TRY(hipMemcpyAsync(m->d_order_big.p, reads.data(), (size_t) nb * 4, hipMemcpyHostToDevice, ost));
			MAP_HIP_TRY(hipMemsetAsync(m->d_order_log2.p, 0, 4, ost));
			MAP_HIP_TRY(hipMemsetAsync(m->d_order_info.p, 0xFF, 2 * (size_t) nb * 4, ost));
			unsigned long long *diag = phases ? m->d_counters.p + (size_t) kCsRegions * kCsCursorStride + 8 : nullptr;
			if (diag) MAP_HIP_TRY(hipMemsetAsync(diag, 0, 16 * 8, ost));
			B.read_list = m->d_order_big.p;
			MAP_HIP_TRY(hipEventRecord(m->oev[2], ost));
			hipLaunchKernelGGL(kern, dim3(grid), dim3(kCsOrderBucketThreads), bucket_lds, ost, B, nb, m->d_order_log2.p, (uint2 *) m->d_order_gt.p, (uint32_t) cap, (uint32_t) bucket_coarse_cap,
					(const uint32_t *) m->d_out_loc.p, (const uint32_t *) m->d_out_sv.p, m->d_cand_rank.p, m->d_order_info.p, diag);
			MAP_HIP_TRY(hipGetLastError());
			MAP_HIP_TRY(hipEventRecord(m->oev[3], ost));
			std::vector<uint32_t> binfo(2 * (size_t) nb);
			MAP_HIP_TRY(hipMemcpyAsync(binfo.data(), m->d_order_info.p, binfo.size() * 4, hipMemcpyDeviceToHost, ost));
			MAP_HIP_TRY(hipStreamSynchronize(ost));
			{ float t = 0; if (hipEventElapsedTime(&t, m->oev[2], m->oev[3]) == hipSuccess) m->order_ms += t; }
			if (diag) {
				unsigned long long ph[16];
				MAP_HIP_TRY(hipMemcpy(ph, diag, sizeof(ph), hipMemcpyDeviceToHost));
				const double ns = (double) std::max(1ull, ph[8]);
				fprintf(stderr, "[ngm-hip] order replay through buckets (%u reads, grid %u, slice %llu hits), us per sampled read: lists %.1f | count %.1f | scan + scatter %.1f | v + tau %.1f (wave 0: %.0f windows, bounds + loads issued %.1f, counted + stored %.1f) | table of M %.1f | candidates %.1f; hits %.0f, candidates %.0f per read; %llu left to the table kernel\n",
						nb, grid, (unsigned long long) cap, ph[0] / ns / 100.0, ph[1] / ns / 100.0, ph[2] / ns / 100.0, ph[3] / ns / 100.0, ph[14] / ns, ph[12] / ns / 100.0, ph[13] / ns / 100.0, ph[4] / ns / 100.0, ph[5] / ns / 100.0, ph[9] / ns, ph[10] / ns, ph[11]);
			}
			std::vector<uint32_t> left;
			for (uint32_t j = 0; j < nb; ++j) if (binfo[2 * j + 1] != 0u) left.push_back(big[j]);
			std::sort(left.begin(), left.end());
			m->st_order_table += left.size();
			big.swap(left);
		} else m->st_order_table += big.size();   // (no room for the slices: all of them to the table kernel)
	} else if (!big.empty()) m->st_order_table += big.size();
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
		if (m->d_order_big.reserve(nb) || m->d_order_log2.reserve(nb) || m->d_order_off.reserve(nb)) { pipeline_set_error("out of device memory (exact candidate order)"); return -12; }
		CsArgs G = m->order_args;
		G.order_info = nullptr; G.order_scratch = nullptr; G.order_max_hits = 0;
		// (cs_order_kernel<true>: staging entries per wave -- what the LDS leaves beside the list rows, the plane and tau (with the
		// 2 048 entries of round 5 a read of more than ~390 bases asked for more LDS than the attribute allows and the launch failed); none
		// for bisulfite runs, whose list rows take the room)
		const size_t lds_fixed = ((size_t) G.lists_cap * 4 + 4 + (G.q + 3) / 4 + 2048 + cs_order_tau(G.lists_cap) + (G.bs ? (size_t) G.q + 1 + G.lists_cap / 4 + 1 : 0)) * 4;
		G.order_gcap = 0;
		if (!G.bs && lds_fixed + 4096 < (size_t) kLdsAttr) {
			const size_t per_wave = ((size_t) kLdsAttr - lds_fixed) / 4 / (kCsOrderThreadsGlobal / 64);
			G.order_gcap = (uint32_t) std::min<size_t>(kCsOrderStage, per_wave & ~(size_t) 63);
			if (G.order_gcap < 256u) G.order_gcap = 0;
		}
		if (lds_fixed > (size_t) kLdsAttr) {   // (reads too long for this kernel's rows: their candidates keep kCsOrderUnknown, counted below)
			big.clear();
		}
		G.phase_cycles = phases ? m->d_counters.p + (size_t) kCsRegions * kCsCursorStride : nullptr;   // diagnostics: phases of every 64th workgroup
		if (G.phase_cycles) MAP_HIP_TRY(hipMemsetAsync(G.phase_cycles + 8, 0, 12 * 8, ost));
		const size_t lds = lds_fixed + (size_t) (kCsOrderThreadsGlobal / 64) * G.order_gcap * 4;
		// (8 GB per launch: a read with 50 000 hits takes 2.6 MB of table and time line, and with the 1.5 GB pool of the first version the
		// 5 500 such reads of a heavy-tailed batch went through ten launches of ~570 workgroups each -- two per CU, 118 ms of waiting per batch)
		// (the pool never asks for more than half of what the device has free, a read that needs more than the pool -- or a pool
		// that cannot be had -- keeps an UNDETERMINED order, which the run reports (st_order_unknown) instead of dying: ties then resolve by position)
		uint64_t pool_words = std::max<uint64_t>(scratch_room(8ull << 30) / 4, test_limit("order_pool_kb", 0) > 0 ? 1024ull : 1ull << 18);
		for (uint32_t j0 = 0; j0 < (uint32_t) big.size();) {
			if (words[j0] > pool_words) { ++j0; continue; }   // (its candidates keep kCsOrderUnknown from the LDS replay's give-up)
			uint64_t total = 0;
			uint32_t j1 = j0;
			while (j1 < nb && total + words[j1] <= pool_words) { off[j1] = total; total += words[j1]; ++j1; }
			if (m->d_order_gt.reserve(total)) {
				if (pool_words > (1ull << 22)) { pool_words /= 2; continue; }   // a smaller pool, more launches
				break;                                                        // no memory at all: the remaining reads stay undetermined
			}
			MAP_HIP_TRY(hipMemcpyAsync(m->d_order_big.p + j0, reads.data() + j0, (size_t) (j1 - j0) * 4, hipMemcpyHostToDevice, ost));
			MAP_HIP_TRY(hipMemcpyAsync(m->d_order_log2.p + j0, lg.data() + j0, (size_t) (j1 - j0) * 4, hipMemcpyHostToDevice, ost));
			MAP_HIP_TRY(hipMemcpyAsync(m->d_order_off.p + j0, off.data() + j0, (size_t) (j1 - j0) * 8, hipMemcpyHostToDevice, ost));
			G.read_list = m->d_order_big.p + j0; G.ovf_log2 = m->d_order_log2.p + j0; G.ovf_table_off = m->d_order_off.p + j0; G.gtable_keys = m->d_order_gt.p;
			MAP_HIP_TRY(hipEventRecord(m->oev[2], ost));
			hipLaunchKernelGGL(cs_order_kernel<true>, dim3(j1 - j0), dim3(kCsOrderThreadsGlobal), lds, ost, G, (const uint32_t *) m->d_out_loc.p, (const uint32_t *) m->d_out_sv.p, m->d_cand_rank.p);
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
		MAP_HIP_TRY(hipMemcpyAsync(m->p_rank.p, m->d_cand_rank.p, np * 4, hipMemcpyDeviceToHost, ost));
		MAP_HIP_TRY(hipStreamSynchronize(ost));
		for (uint32_t i : beyond_lds) {
			const uint32_t rd = m->order_pending[i], b = m->h_base[rd], c = m->h_count[rd];
			bool unknown = false;
			for (uint32_t x = 0; x < c && !unknown; ++x) unknown = m->p_rank.p[b + x] == kCsOrderUnknown;
			m->st_order_unknown += unknown ? 1 : 0;
		}
	}
	m->order_pending.clear();
	return 0;
}
}  // namespace

int candidate_order_wait(ngm_mapper *m, uint32_t **h_rank) {
	if (int rc = candidate_order_finish(m, m->st_hi ? m->st_hi : m->st, m->n_cand)) return rc;
	*h_rank = m->p_rank.p;
	return 0;
}

int candidate_order(ngm_mapper *m, const std::vector<uint32_t> &list, uint64_t np, uint32_t **h_rank, bool wait) {
	const uint32_t nl = (uint32_t) list.size();
	if (m->prm.slam_seq & 4) {
		// the weighted SLAM-seq search replays the votes in the reference's order anyway: its candidates leave in rList order
		if (m->p_rank.reserve(np + 1)) { pipeline_set_error("out of memory (candidate order)"); return -12; }
		for (uint32_t rd : list) { const uint32_t b = m->h_base[rd], c = m->h_count[rd]; for (uint32_t x = 0; x < c; ++x) m->p_rank.p[b + x] = x; }
		m->st_order_reads += nl;
		m->order_pending.clear();
		*h_rank = m->p_rank.p;
		(void) wait;
		return 0;
	}
	static const bool phases = getenv("NGM_HIP_CS_PHASES") != nullptr;
	hipStream_t ost = m->st_hi ? m->st_hi : m->st;  // everything this depends on has been synchronised by the caller
	const auto t_begin = std::chrono::steady_clock::now();
	if (m->d_order_list.reserve(nl) || m->d_cand_rank.reserve(np + 1) || m->p_rank.reserve(np + 1) || m->d_order_info.reserve(2 * (size_t) nl) || m->p_order_info.reserve(2 * (size_t) nl)) { pipeline_set_error("out of memory (candidate order)"); return -12; }
	m->order_pending = list;
	MAP_HIP_TRY(hipMemcpyAsync(m->d_order_list.p, list.data(), (size_t) nl * 4, hipMemcpyHostToDevice, ost));
	CsArgs A = m->last_cs;
	A.read_list = m->d_order_list.p;
	A.cand_base = m->d_cand_base.p; A.cand_count = m->d_cand_count.p;
	A.counters = nullptr;
	const size_t ctr_words_o = (size_t) kCsRegions * kCsCursorStride;
	A.phase_cycles = phases ? m->d_counters.p + ctr_words_o : nullptr;
	if (A.phase_cycles) MAP_HIP_TRY(hipMemsetAsync(A.phase_cycles + 8, 0, 12 * 8, ost));
	// the time line takes what is left of 80 KB of LDS (reads with more hits walk a slice of global memory, ~10 x slower): the
	// size that lets TWO workgroups share a CU: measured on MI355X, this kernel with 88 KB of LDS has 26
	// workgroups in flight instead of 232 (NGM_HIP_CS_PHASES=1 prints the summed workgroup time; a plain spinning kernel of the
	// same LDS size does reach 232, profiles/tools/lds_occupancy_calib.hip) -- 1.7 s instead of 0.1 s for config 5's 256 k tied reads
	const size_t lds_budget = 80 * 1024;
	if (A.bs) A.lists_cap = 2 * 3072;   // bisulfite mapping: the lists of all k-mer variants of a read (more: that read keeps the position order)
	const size_t lds_fixed = ((size_t) A.lists_cap * 3 + 2 + (A.q + 3) / 4 + 2048 + cs_order_tau(A.lists_cap) + ((size_t) 5 << kCsOrderLog2Slots) + (A.bs ? (size_t) A.q + 1 + A.lists_cap / 4 + 1 : 0)) * 4;
	// (two arrays of that many entries: the time line and the hit times sorted by bin and strand)
	const size_t hits_room = lds_fixed + 8 * (size_t) kCsOrderMaxHits < lds_budget ? (lds_budget - 64 - lds_fixed) / 8 : (size_t) kCsOrderMaxHits;
	A.order_max_hits = (uint32_t) std::max<size_t>(kCsOrderMaxHits, std::min<size_t>(hits_room, 65535));  // (all of the budget: two workgroups per CU either way)
	const size_t lds = lds_fixed + (size_t) A.order_max_hits * 8;
	// reads with more hits than the LDS time line holds use a slice of a global scratch: launches of at most 4096 reads
	constexpr uint32_t kChunk = 4096, kGcap = 49152;
	// reads with more hits than the LDS time line holds: to the bucket kernel (cs_order_bucket_kernel) -- the LDS replay with its time line in a
	// slice of global memory is what bisulfite runs (no bucket kernel) and NGM_HIP_ORDER_LDS_BIG=1 / NGM_HIP_ORDER_NO_BUCKETS=1 still use
	// (measured at 3.1 Gbp, half of the reads from repeats: 0.745 M reads/s without it, 0.669 M with it)
	static const bool lds_big_env = getenv("NGM_HIP_ORDER_LDS_BIG") != nullptr || getenv("NGM_HIP_ORDER_NO_BUCKETS") != nullptr;
	const bool no_lds_big = !lds_big_env && !A.bs && cs_order_tau(A.lists_cap) <= kCsOrderBucketMaxTau;
	if (no_lds_big || m->d_order_scratch.reserve((size_t) std::min(nl, kChunk) * kGcap * 2)) { A.order_scratch = nullptr; A.order_gcap = 0; }   // (time line + hit times per workgroup)
	else { A.order_scratch = m->d_order_scratch.p; A.order_gcap = kGcap; }
	MAP_HIP_TRY(hipEventRecord(m->oev[0], ost));
	for (uint32_t off = 0; off < nl; off += kChunk) {
		A.read_list = m->d_order_list.p + off;
		A.order_info = m->d_order_info.p + 2 * (size_t) off;
		hipLaunchKernelGGL(cs_order_kernel<false>, dim3(std::min(kChunk, nl - off)), dim3(kCsOrderThreads), lds, ost, A, (const uint32_t *) m->d_out_loc.p, (const uint32_t *) m->d_out_sv.p, m->d_cand_rank.p);
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
		fprintf(stderr, "[ngm-hip] order replay, us per sampled read: lists %.1f | sweep A %.1f | sweep B %.1f | compaction + replay %.1f; %llu sampled, %llu gave up; hits %.0f, replayed %.0f per read\n",
				ph[0] / ns / 100.0, ph[1] / ns / 100.0, ph[2] / ns / 100.0, ph[3] / ns / 100.0, ph[4], ph[5] & 0xFFull, ph[6] / ns, ph[7] / ns);
	}
	static const bool host_timing = getenv("NGM_HIP_HOST_TIMING") != nullptr;
	if (host_timing)
		fprintf(stderr, "[ngm-hip] candidate order replay: %u reads, %.2f ms\n", nl, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
	return 0;
}

}  // namespace ngm
