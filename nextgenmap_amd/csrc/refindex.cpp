// refindex.cpp -- reference encoding and k-mer index construction (host walk + GPU sort).
// See refindex.h for what it replaces and for the HBM layout.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <functional>
#include <mutex>
#include <string.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "refindex.h"
#include "thread_pool.h"

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <zlib.h>

namespace {
thread_local std::string g_pipeline_error;

#define REF_HIP_TRY(expr)                                                                      \
	do {                                                                                       \
		hipError_t e_ = (expr);                                                                \
		if (e_ != hipSuccess) {                                                                \
			ngm::pipeline_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
			return -5;                                                                         \
		}                                                                                      \
	} while (0)

inline uint8_t class_of_base(uint8_t c) {  // enc4 semantics (SequenceProvider.cpp:72-85): non-ACGT -> N
	switch (c & 0xDF) {
	case 'A': return 0;
	case 'C': return 1;
	case 'G': return 2;
	case 'T': return 3;
	default: return 5;
	}
}

// ---- the reference's k-mer enumeration, restated ---------------------------------------------------
// CS::PrefixIteration (src/CSstatic.cpp:26-76) with prefixskip = kmer_skip over one contig, feeding
// CompactPrefixTable::CountKmer / BuildPrefixTable (src/PrefixTable.cpp:641-709): emits the (k-mer,
// position) pairs that end up in the index, in genome order.
struct KmerWalker {
	int k, skip, bin_shift;
	uint32_t mask;
	uint32_t last_prefix;
	int64_t last_bin;
	std::vector<uint32_t> *keys, *vals;

	void fire(uint32_t prefix, uint64_t pos) {
		// same k-mer as the previous indexed one AND same bin as the previous same-k-mer hit -> dropped
		if (prefix == last_prefix) {
			const int64_t bin = (int64_t) (pos >> bin_shift);
			if (bin != last_bin || last_bin == -1) { keys->push_back(prefix); vals->push_back((uint32_t) pos); }
			last_bin = bin;
		} else {
			last_bin = -1;
			keys->push_back(prefix);
			vals->push_back((uint32_t) pos);
		}
		last_prefix = prefix;
	}

	// seq: class per base, the reference's quirks already applied (see walk_contig); N = class 5
	void iterate(const uint8_t *seq, uint64_t length, uint64_t offset) {
		for (;;) {  // the reference recurses after every 'N'; this loop is that recursion
			if (length < (uint64_t) k) return;
			if (seq[0] == 5) {
				uint64_t n_skip = 1;
				while (n_skip < length && seq[n_skip] == 5) ++n_skip;
				seq += n_skip;
				if (n_skip >= (length - k)) return;  // CSstatic.cpp:37 (drops a tail of exactly k bases too)
				length -= n_skip;
				offset += n_skip;
			}
			uint32_t prefix = 0;
			bool restart = false;
			for (uint64_t i = 0; i < (uint64_t) k - 1; ++i) {
				if (seq[i] == 5) { seq += i + 1; length -= i + 1; offset += i + 1; restart = true; break; }
				prefix = (prefix << 2) | ngm::kmer_code_of_class(seq[i]);
			}
			if (restart) continue;
			int skipcount = skip;
			for (uint64_t i = k - 1; i < length; ++i) {
				if (seq[i] == 5) { seq += i + 1; length -= i + 1; offset += i + 1; restart = true; break; }
				prefix = ((prefix << 2) | ngm::kmer_code_of_class(seq[i])) & mask;
				if (skipcount == skip) { fire(prefix, offset + i + 1 - k); skipcount = 0; }
				else ++skipcount;
			}
			if (!restart) return;
		}
	}
};

__global__ void mark_runs_kernel(const uint32_t *__restrict__ keys, uint64_t n, uint32_t *__restrict__ starts, uint32_t *__restrict__ raw) {
	const uint64_t j = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= n) return;
	const uint32_t kj = keys[j];
	if (j == 0 || keys[j - 1] != kj) starts[kj] = (uint32_t) j;
	if (j + 1 == n || keys[j + 1] != kj) raw[kj] = (uint32_t) (j + 1);  // end; turned into a length below
}

__global__ void finish_index_kernel(uint32_t n_kmers, int k, const uint32_t *__restrict__ starts, uint32_t *__restrict__ raw,
		uint2 *__restrict__ index) {
	const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
	if (p >= n_kmers) return;
	const uint32_t end = raw[p];
	const uint32_t cnt = end ? end - starts[p] : 0u;  // raw[] holds end+0 only for present k-mers
	raw[p] = cnt;
	index[p] = make_uint2(starts[p], cnt);
}

__device__ __forceinline__ uint32_t d_revcomp(uint32_t prefix, int k) {
	const int shift = 32 - 2 * k;
	uint32_t c = (prefix ^ 0xAAAAAAAAu) << shift;
	c = (c & 0xFFFF0000u) >> 16 | (c & 0x0000FFFFu) << 16;
	c = (c & 0xFF00FF00u) >> 8 | (c & 0x00FF00FFu) << 8;
	c = (c & 0xF0F0F0F0u) >> 4 | (c & 0x0F0F0F0Fu) << 4;
	c = (c & 0xCCCCCCCCu) >> 2 | (c & 0x33333333u) << 2;
	return c;
}

// the reference marks a k-mer unused when (10000 - min(total,10000)) * 100 / 10000 truncates to 0, i.e.
// total = own + reverse-complement occurrences >= 9901 (PrefixTable.cpp:468-478); lookups then see an empty list
__global__ void apply_usage_rule_kernel(uint32_t n_kmers, int k, const uint32_t *__restrict__ raw, uint2 *__restrict__ index) {
	const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
	if (p >= n_kmers) return;
	const uint32_t total = raw[p] + raw[d_revcomp(p, k)];
	if (total > 9900u) index[p].y = 0;
}

void index_stats(ngm_ref *r, const uint32_t *raw, uint32_t n_kmers);
int upload_staged(void *dst, const void *src, size_t bytes);

// ---- the same enumeration on the GPU ------------------------------------------------------------------------
// KmerWalker above, position-parallel.  Which k-mers the reference visits (CS::PrefixIteration, src/CSstatic.cpp:26-76, with
// prefixskip = kmer_skip) follows from the N-free runs of a contig alone: in a maximal N-free run [a, b) the k-mers start at
// a, a + (skip+1), ... while start + k <= b (the skip counter restarts at every restart), except the run of exactly k bases at
// the contig end behind an N run that the walker meets at a restart position (CSstatic.cpp:37).  Which of them are stored
// (CompactPrefixTable::CountKmer / BuildPrefixTable, src/PrefixTable.cpp:641-709: a k-mer equal to the previously visited one is
// dropped when it falls into the same bin as that one, from the third of such a run on) depends on the two visited k-mers before
// it only.  So: N marks -> inclusive max-scan = start of the current run -> visit flags -> compaction -> keys -> store flags ->
// compaction; all in HBM (12.4 GB of scratch for a GRCh38-size genome).
__device__ __forceinline__ uint32_t d_class_at(const uint32_t *g, uint64_t i) { return (g[i >> 3] >> (4 * (int) (i & 7))) & 15u; }

__global__ void walk_mark_n_kernel(const uint32_t *__restrict__ genome, uint64_t n, uint32_t *__restrict__ v) {
	const uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) v[i] = d_class_at(genome, i) == 5u ? (uint32_t) i + 1u : 0u;
}
// the last two bases of every contig act as 'A' for the walk (see build_index below)
__global__ void walk_unmark_tails_kernel(const uint64_t *__restrict__ cend, int n_contigs, uint32_t *__restrict__ v) {
	const int c = blockIdx.x * blockDim.x + threadIdx.x;
	if (c < n_contigs) { v[cend[c] - 1] = 0; v[cend[c] - 2] = 0; }
}
__device__ __forceinline__ int walk_contig_of(const uint64_t *cstart, int n_contigs, uint64_t pos) {  // last contig with start <= pos
	int lo = 0, hi = n_contigs;
	while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (cstart[mid] <= pos) lo = mid; else hi = mid; }
	return lo;
}
__global__ void walk_visit_flags_kernel(const uint32_t *__restrict__ run_start, const uint32_t *__restrict__ genome, uint64_t n, int k, int step,
		const uint64_t *__restrict__ cstart, const uint64_t *__restrict__ cend, int n_contigs, uint8_t *__restrict__ flag) {
	const uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	uint8_t f = 0;
	if (i + k <= n) {
		const uint64_t a = run_start[i + k - 1];  // index after the last N at or before i + k - 1
		if (a <= i && (i - a) % (uint64_t) step == 0) {
			f = 1;
			if (i == a) {  // CSstatic.cpp:37: exactly k bases left behind an N run met at a restart position
				const int c = walk_contig_of(cstart, n_contigs, i);
				const uint64_t p = i - cstart[c];
				if (i + k == cend[c] && p >= 1 && d_class_at(genome, i - 1) == 5u && (p == 1 || d_class_at(genome, i - 2) == 5u)) f = 0;
			}
		}
	}
	flag[i] = f;
}
__global__ void walk_keys_kernel(const uint32_t *__restrict__ pos, uint64_t m, const uint32_t *__restrict__ genome, int k, const uint64_t *__restrict__ cstart,
		const uint64_t *__restrict__ cend, int n_contigs, uint32_t *__restrict__ keys) {
	const uint64_t j = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= m) return;
	const uint64_t p = pos[j];
	const uint64_t tail = cend[walk_contig_of(cstart, n_contigs, p)] - 2;  // from here on the walk sees 'A'
	uint32_t key = 0;
	for (int t = 0; t < k; ++t) {
		const uint32_t cls = (p + t >= tail) ? 0u : d_class_at(genome, p + t);
		key = (key << 2) | (cls == 2u ? 3u : (cls == 3u ? 2u : cls));  // A0 C1 T2 G3 (CSstatic.cpp:20-22)
	}
	keys[j] = key;
}
// PrefixTable.cpp:641-709 (fire() of the host walker): stored unless it repeats the previous visited k-mer of its contig in the
// same bin, where the bin memory starts with the second k-mer of such a run
__global__ void walk_store_flags_kernel(const uint32_t *__restrict__ pos, const uint32_t *__restrict__ keys, uint64_t m, int bin_shift,
		const uint64_t *__restrict__ cstart, int n_contigs, uint8_t *__restrict__ flag) {
	const uint64_t j = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= m) return;
	const uint64_t cs = cstart[walk_contig_of(cstart, n_contigs, pos[j])];
	const bool has1 = j >= 1 && pos[j - 1] >= cs, has2 = j >= 2 && pos[j - 2] >= cs;
	const uint32_t prev = has1 ? keys[j - 1] : 111111u;        // PrefixTable.cpp:335-336: lastPrefix starts as 111111 per contig
	bool store = true;
	if (keys[j] == prev) {
		// the bin remembered when k-mer j is looked at: none after a k-mer that differed from ITS predecessor
		bool have_bin = false;
		if (has1) { const uint32_t prev2 = has2 ? keys[j - 2] : 111111u; have_bin = keys[j - 1] == prev2; }
		if (have_bin && (pos[j] >> bin_shift) == (pos[j - 1] >> bin_shift)) store = false;
	}
	flag[j] = store ? 1 : 0;
}

// -> d_keys / d_vals (device, n entries, genome order).  Returns 0 or a negative error.
int gpu_kmer_walk(ngm_ref *r, uint32_t **d_keys_out, uint32_t **d_vals_out, uint64_t *n_out) {
	const uint64_t n = r->n_bases;
	const int k = r->prm.kmer, nc = (int) r->contigs.size();
	std::vector<uint64_t> hs(nc), he(nc);
	for (int c = 0; c < nc; ++c) { hs[c] = r->contigs[c].start; he[c] = r->contigs[c].start + r->contigs[c].len; }
	uint64_t *d_cs = nullptr, *d_ce = nullptr;
	uint32_t *d_run = nullptr, *d_pos = nullptr, *d_keys = nullptr, *d_keys2 = nullptr, *d_pos2 = nullptr;
	uint8_t *d_flag = nullptr;
	uint64_t *d_count = nullptr;
	void *d_tmp = nullptr;
	auto cleanup = [&]() { (void) hipFree(d_cs); (void) hipFree(d_ce); (void) hipFree(d_run); (void) hipFree(d_flag); (void) hipFree(d_count); (void) hipFree(d_tmp);
		d_cs = d_ce = nullptr; d_run = nullptr; d_flag = nullptr; d_count = nullptr; d_tmp = nullptr; };
	// every error return below frees what the walk holds (up to ~12 GB of scratch for a human-size genome)
	bool done = false;
	struct Guard { std::function<void()> f; bool *done; ~Guard() { if (!*done) f(); } } guard{[&]() { cleanup(); (void) hipFree(d_pos); (void) hipFree(d_keys); (void) hipFree(d_keys2); (void) hipFree(d_pos2); }, &done};
	REF_HIP_TRY(hipMalloc(&d_cs, std::max(1, nc) * 8)); REF_HIP_TRY(hipMalloc(&d_ce, std::max(1, nc) * 8));
	REF_HIP_TRY(hipMemcpy(d_cs, hs.data(), nc * 8, hipMemcpyHostToDevice)); REF_HIP_TRY(hipMemcpy(d_ce, he.data(), nc * 8, hipMemcpyHostToDevice));
	REF_HIP_TRY(hipMalloc(&d_run, std::max<uint64_t>(n, 1) * 4)); REF_HIP_TRY(hipMalloc(&d_flag, std::max<uint64_t>(n, 1))); REF_HIP_TRY(hipMalloc(&d_count, 8));
	const unsigned nb = (unsigned) ((n + 255) / 256);
	hipLaunchKernelGGL(walk_mark_n_kernel, dim3(nb), dim3(256), 0, 0, r->d_genome, n, d_run);
	if (nc > 0) hipLaunchKernelGGL(walk_unmark_tails_kernel, dim3((nc + 255) / 256), dim3(256), 0, 0, d_ce, nc, d_run);
	size_t tb = 0, tb2 = 0;
	(void) rocprim::inclusive_scan(nullptr, tb, d_run, d_run, (size_t) n, rocprim::maximum<uint32_t>());
	(void) rocprim::select(nullptr, tb2, rocprim::counting_iterator<uint32_t>(0), d_flag, (uint32_t *) nullptr, d_count, (size_t) n);
	size_t tb3 = 0;
	(void) rocprim::select(nullptr, tb3, (uint32_t *) nullptr, d_flag, (uint32_t *) nullptr, d_count, (size_t) n);
	tb = std::max(tb, std::max(tb2, tb3)) + 256;
	REF_HIP_TRY(hipMalloc(&d_tmp, tb));
	REF_HIP_TRY(rocprim::inclusive_scan(d_tmp, tb, d_run, d_run, (size_t) n, rocprim::maximum<uint32_t>()));
	hipLaunchKernelGGL(walk_visit_flags_kernel, dim3(nb), dim3(256), 0, 0, d_run, r->d_genome, n, k, r->prm.kmer_skip + 1, d_cs, d_ce, nc, d_flag);
	REF_HIP_TRY(hipGetLastError());
	// upper bound of visited k-mers: every (skip+1)-th position plus one per run
	const uint64_t cap = n / (uint64_t) (r->prm.kmer_skip + 1) + n / 1000 + 4096;
	REF_HIP_TRY(hipMalloc(&d_pos, cap * 4));
	REF_HIP_TRY(rocprim::select(d_tmp, tb, rocprim::counting_iterator<uint32_t>(0), d_flag, d_pos, d_count, (size_t) n));
	uint64_t m = 0;
	REF_HIP_TRY(hipMemcpy(&m, d_count, 8, hipMemcpyDeviceToHost));
	if (m > cap) { ngm::pipeline_set_error("k-mer walk: more visited k-mers than expected"); return -75; }
	(void) hipFree(d_run); d_run = nullptr;
	REF_HIP_TRY(hipMalloc(&d_keys, std::max<uint64_t>(m, 1) * 4)); REF_HIP_TRY(hipMalloc(&d_keys2, std::max<uint64_t>(m, 1) * 4)); REF_HIP_TRY(hipMalloc(&d_pos2, std::max<uint64_t>(m, 1) * 4));
	uint64_t kept = 0;
	if (m > 0) {
		const unsigned mb = (unsigned) ((m + 255) / 256);
		hipLaunchKernelGGL(walk_keys_kernel, dim3(mb), dim3(256), 0, 0, d_pos, m, r->d_genome, k, d_cs, d_ce, nc, d_keys);
		hipLaunchKernelGGL(walk_store_flags_kernel, dim3(mb), dim3(256), 0, 0, d_pos, d_keys, m, r->prm.bin_size, d_cs, nc, d_flag);
		REF_HIP_TRY(hipGetLastError());
		REF_HIP_TRY(rocprim::select(d_tmp, tb, d_keys, d_flag, d_keys2, d_count, (size_t) m));
		REF_HIP_TRY(rocprim::select(d_tmp, tb, d_pos, d_flag, d_pos2, d_count, (size_t) m));
		REF_HIP_TRY(hipMemcpy(&kept, d_count, 8, hipMemcpyDeviceToHost));
	}
	cleanup();
	(void) hipFree(d_pos); (void) hipFree(d_keys);
	done = true;
	*d_keys_out = d_keys2; *d_vals_out = d_pos2; *n_out = kept;
	return 0;
}

// bucket of k-mer p: word 0 = own list length (bits 0-13; 0: unused, or more than 9900 occurrences) | list length of the
// reverse-complement k-mer << 14 (the search needs the sum of both, CS.cpp:122) | 1 << 31 when the list does not fit;
// words 1.. = the positions, or word 1 = start in d_positions for a list that does not fit.  One bucket more than there
// are k-mers: the last one stays zero (what lanes without a list read).  One thread per bucket word: coalesced stores.
__global__ void fill_buckets_kernel(uint32_t n_kmers, int k, int log2_w, const uint2 *__restrict__ index, const uint32_t *__restrict__ positions,
		uint32_t *__restrict__ buckets) {
	const uint64_t g = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t p = (uint32_t) (g >> log2_w), w = (uint32_t) g & ((1u << log2_w) - 1u);
	if (p > n_kmers) return;
	if (p == n_kmers) { buckets[g] = 0; return; }
	const uint2 e = index[p];
	const bool inl = e.y < (1u << log2_w);
	uint32_t v = 0;
	if (w == 0) v = e.y | (index[d_revcomp(p, k)].y << 14) | (inl ? 0u : 0x80000000u);
	else if (inl) v = (w <= e.y) ? positions[e.x + (w - 1)] : 0u;
	else if (w == 1) v = e.x;
	buckets[g] = v;
}

// canonical bucket c (refindex.h): the pair {X = kmer_of_canon_id(c), revcomp(X)}
__global__ void fill_cbuckets_kernel(uint32_t n_pairs, int k, int log2_w, const uint2 *__restrict__ index, const uint32_t *__restrict__ positions,
		uint32_t *__restrict__ buckets) {
	const uint64_t g = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t c = (uint32_t) (g >> log2_w), w = (uint32_t) g & ((1u << log2_w) - 1u);
	if (c >= n_pairs) return;
	const int b = 2 * (k / 2) + 1;
	const uint32_t x = ((c >> b) << (b + 1)) | (c & ((1u << b) - 1u));
	const uint2 ea = index[x], eb = index[d_revcomp(x, k)];
	const uint32_t na = ea.y, nb = eb.y;
	const bool inl = na + nb < (1u << log2_w);
	uint32_t v = 0;
	if (w == 0) v = na | (nb << 14) | (inl ? 0u : 0x80000000u);
	else if (inl) v = (w <= na) ? positions[ea.x + (w - 1)] : (w <= na + nb) ? positions[eb.x + (w - 1 - na)] : 0u;
	else if (w == 1) v = ea.x;
	else if (w == 2) v = eb.x;
	buckets[g] = v;
}

std::mutex g_bucket_mu;

int build_buckets_kind(ngm_ref *r, int kind) {
	const int k = r->prm.kmer;
	const uint32_t n_kmers = 1u << (2 * k);
	if (kind == 1) {
		if (r->d_cbuckets) return 0;
		if ((k & 1) == 0) { ngm::pipeline_set_error("canonical buckets need an odd k-mer length"); return -22; }
		const uint32_t n_pairs = n_kmers / 2;
		const double mu = (double) r->n_entries / (double) n_pairs;
		int lw = 2;
		while (lw < 6 && (double) ((1 << lw) - 1) < mu + 4.0 * sqrt(mu) + 0.5) ++lw;
		while (lw > 2 && ((uint64_t) n_pairs << lw) + r->n_entries + 16 >= 0xFFFFFFFFull) --lw;
		const uint64_t words = (uint64_t) n_pairs << lw;
		uint32_t *d = nullptr;
		REF_HIP_TRY(hipMalloc(&d, (words + r->n_entries + 16) * 4));
		hipLaunchKernelGGL(fill_cbuckets_kernel, dim3((unsigned) ((words + 255) / 256)), dim3(256), 0, 0, n_pairs, k, lw, r->d_index, r->d_positions, d);
		if (hipGetLastError() != hipSuccess || hipMemcpy(d + words, r->d_positions, (r->n_entries + 16) * 4, hipMemcpyDeviceToDevice) != hipSuccess ||
				hipDeviceSynchronize() != hipSuccess) { (void) hipFree(d); ngm::pipeline_set_error("building the canonical buckets failed"); return -5; }
		r->cbucket_log2_words = lw;
		r->cbucket_pos_base = (uint32_t) words;
		r->d_cbuckets = d;
		return 0;
	}
	if (r->d_buckets) return 0;
	// W - 1 >= mean + 4 standard deviations of a Poisson list length (real genomes have a heavy tail on top: those
	// lists overflow into d_positions, which costs them one more request)
	const double mu = (double) r->n_entries / (double) n_kmers;
	int lw = 2;
	while (lw < 5 && (double) ((1 << lw) - 1) < mu + 4.0 * sqrt(mu) + 0.5) ++lw;
	// the buckets are followed by a copy of the position table, so that one 32-bit word offset addresses an inline list and a
	// list that did not fit alike; both must stay below 2^32 words
	while (lw > 2 && (((uint64_t) n_kmers + 1) << lw) + r->n_entries + 16 >= 0xFFFFFFFFull) --lw;
	const uint64_t words = ((uint64_t) n_kmers + 1) << lw;
	uint32_t *d = nullptr;
	REF_HIP_TRY(hipMalloc(&d, (words + r->n_entries + 16) * 4));
	hipLaunchKernelGGL(fill_buckets_kernel, dim3((unsigned) ((words + 255) / 256)), dim3(256), 0, 0, n_kmers, k, lw, r->d_index, r->d_positions, d);
	if (hipGetLastError() != hipSuccess || hipMemcpy(d + words, r->d_positions, (r->n_entries + 16) * 4, hipMemcpyDeviceToDevice) != hipSuccess ||
			hipDeviceSynchronize() != hipSuccess) { (void) hipFree(d); ngm::pipeline_set_error("building the buckets failed"); return -5; }
	r->bucket_log2_words = lw;
	r->bucket_pos_base = (uint32_t) words;
	r->d_buckets = d;
	return 0;
}

int build_index(ngm_ref *r) {
	const int k = r->prm.kmer;
	const uint32_t n_kmers = 1u << (2 * k);
	uint32_t *d_keys = nullptr, *d_vals = nullptr, *d_keys2 = nullptr, *d_starts = nullptr;
	uint64_t n = 0;
	{
		// the walk on the GPU (CountKmerFreq decodes a contig with bufferLength = len and DecodeRefSequence emits len - 2 bases,
		// 'x' / NUL after that, which encode() maps to 0: the last two bases of a contig act as 'A' -- handled in the kernels)
		if (gpu_kmer_walk(r, &d_keys, &d_vals, &n) != 0) { d_keys = d_vals = nullptr; n = 0; (void) hipGetLastError(); }  // e.g. out of memory on a shared GPU: the host walk needs no scratch
	}
	if (!d_keys) {
		std::vector<uint32_t> keys, vals;
		keys.reserve(r->n_bases / (r->prm.kmer_skip + 1) + 16);
		vals.reserve(r->n_bases / (r->prm.kmer_skip + 1) + 16);
		KmerWalker w{k, r->prm.kmer_skip, r->prm.bin_size, (k == 16) ? 0xFFFFFFFFu : ((1u << (2 * k)) - 1u), 111111u, -1, &keys, &vals};
		std::vector<uint8_t> tmp;
		for (const NgmContig &c : r->contigs) {
			w.last_prefix = 111111u;  // PrefixTable.cpp:335-336
			w.last_bin = -1;
			// CountKmerFreq decodes the contig with bufferLength = len, and DecodeRefSequence emits bufferLength-2
			// bases ('x' / NUL after that, SequenceProvider.cpp:384, :424-439); PrefixIteration then still walks
			// all len characters and encode() maps both fillers to 0: the last two bases act as 'A'.
			tmp.assign(r->host_cls.begin() + c.start, r->host_cls.begin() + c.start + c.len);
			if (c.len >= 2) { tmp[c.len - 2] = 0; tmp[c.len - 1] = 0; }
			w.iterate(tmp.data(), c.len, c.start);
		}
		n = keys.size();
		REF_HIP_TRY(hipMalloc(&d_keys, std::max<uint64_t>(n, 1) * 4));
		REF_HIP_TRY(hipMalloc(&d_vals, std::max<uint64_t>(n, 1) * 4));
		REF_HIP_TRY(hipMemcpy(d_keys, keys.data(), n * 4, hipMemcpyHostToDevice));
		REF_HIP_TRY(hipMemcpy(d_vals, vals.data(), n * 4, hipMemcpyHostToDevice));
	}
	r->n_entries = n;
	REF_HIP_TRY(hipMalloc(&d_keys2, std::max<uint64_t>(n, 1) * 4));
	REF_HIP_TRY(hipMalloc(&r->d_positions, (std::max<uint64_t>(n, 1) + 16) * 4));  // the search reads 16-entry segments
	REF_HIP_TRY(hipMemset(r->d_positions, 0, (std::max<uint64_t>(n, 1) + 16) * 4));
	REF_HIP_TRY(hipMalloc(&d_starts, (size_t) n_kmers * 4));
	REF_HIP_TRY(hipMalloc(&r->d_raw_counts, (size_t) n_kmers * 4));
	REF_HIP_TRY(hipMalloc(&r->d_index, (size_t) n_kmers * sizeof(uint2)));
	REF_HIP_TRY(hipMemset(d_starts, 0, (size_t) n_kmers * 4));
	REF_HIP_TRY(hipMemset(r->d_raw_counts, 0, (size_t) n_kmers * 4));
	if (n > 0) {
		// stable LSD radix sort on the 2k key bits: positions stay ascending inside each k-mer group,
		// which is the order BuildPrefixTable appends them in (PrefixTable.cpp:729-748)
		size_t tmp_bytes = 0;
		REF_HIP_TRY(rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_keys, d_keys2, d_vals, r->d_positions, n, 0, 2 * k));
		void *d_tmp = nullptr;
		REF_HIP_TRY(hipMalloc(&d_tmp, tmp_bytes));
		REF_HIP_TRY(rocprim::radix_sort_pairs(d_tmp, tmp_bytes, d_keys, d_keys2, d_vals, r->d_positions, n, 0, 2 * k));
		hipLaunchKernelGGL(mark_runs_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, 0, d_keys2, n, d_starts, r->d_raw_counts);
		REF_HIP_TRY(hipGetLastError());
		REF_HIP_TRY(hipDeviceSynchronize());
		(void) hipFree(d_tmp);
	}
	hipLaunchKernelGGL(finish_index_kernel, dim3((n_kmers + 255) / 256), dim3(256), 0, 0, n_kmers, k, d_starts, r->d_raw_counts, r->d_index);
	hipLaunchKernelGGL(apply_usage_rule_kernel, dim3((n_kmers + 255) / 256), dim3(256), 0, 0, n_kmers, k, r->d_raw_counts, r->d_index);
	REF_HIP_TRY(hipGetLastError());
	REF_HIP_TRY(hipDeviceSynchronize());
	(void) hipFree(d_keys); (void) hipFree(d_vals); (void) hipFree(d_keys2); (void) hipFree(d_starts);

	std::vector<uint32_t> raw(n_kmers);
	REF_HIP_TRY(hipMemcpy(raw.data(), r->d_raw_counts, (size_t) n_kmers * 4, hipMemcpyDeviceToHost));
	index_stats(r, raw.data(), (uint32_t) raw.size());
	return 0;   // the bucket layouts are built when the first mapper asks for one (ngm_ref_ensure_buckets): a bisulfite run needs none
}

// CompactPrefixTable::stats (PrefixTable.cpp:150-194): integer sums are exact in double, order-free
void index_stats(ngm_ref *r, const uint32_t *raw, uint32_t n_kmers) {
	// (sums of integers below 2^53: exact in doubles whatever the order, so the pool threads may split them)
	std::mutex mu;
	double sum = 0.0, sum2 = 0.0;
	ngm::ThreadPool::instance().parallel_for((int) ((n_kmers + (1u << 18) - 1) >> 18), [&](int lo, int hi) {
		double s1 = 0.0, s2 = 0.0;
		for (uint32_t j = (uint32_t) lo << 18, e = (uint32_t) std::min<uint64_t>(n_kmers, (uint64_t) hi << 18); j < e; ++j) { s1 += raw[j]; s2 += (double) raw[j] * (double) raw[j]; }
		std::lock_guard<std::mutex> lk(mu);
		sum += s1; sum2 += s2;
	}, 1);
	const double len = (double) n_kmers;
	const double avg = sum / len;
	const double stdev = sqrt(sum2 / (len - 1) - 2.0 * avg * (sum / (len - 1)) + ((len * avg * avg) / (len - 1)));
	r->auto_max_kfreq = (int) ceil(std::max(100.0, avg + 5 * stdev));
}


// host -> device for the multi-GB arrays of the reference (pageable or file-mapped memory): the pool threads copy 32 MB pieces
// into two page-locked buffers, each piece travels while the next is copied
int upload_staged(void *dst, const void *src, size_t bytes) {
	constexpr size_t kPiece = (size_t) 32 << 20;
	if (bytes <= kPiece) { REF_HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice)); return 0; }
	void *stage[2] = {nullptr, nullptr};
	hipEvent_t done[2] = {nullptr, nullptr};
	hipStream_t st = nullptr;
	int rc = 0;
	auto cleanup = [&] { for (int i = 0; i < 2; ++i) { if (stage[i]) (void) hipHostFree(stage[i]); if (done[i]) (void) hipEventDestroy(done[i]); } if (st) (void) hipStreamDestroy(st); };
	if (hipHostMalloc(&stage[0], kPiece, hipHostMallocDefault) != hipSuccess || hipHostMalloc(&stage[1], kPiece, hipHostMallocDefault) != hipSuccess ||
			hipEventCreateWithFlags(&done[0], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&done[1], hipEventDisableTiming) != hipSuccess ||
			hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) {
		cleanup();
		REF_HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));   // (no page-locked memory left: the plain copy)
		return 0;
	}
	ngm::ThreadPool &pool = ngm::ThreadPool::instance();
	int turn = 0;
	for (size_t off = 0; off < bytes && rc == 0; off += kPiece, turn ^= 1) {
		const size_t nb = std::min(kPiece, bytes - off);
		if (off >= 2 * kPiece && hipEventSynchronize(done[turn]) != hipSuccess) { rc = -5; break; }
		const char *from = (const char *) src + off;
		char *to = (char *) stage[turn];
		const int parts = (int) ((nb + (1u << 20) - 1) >> 20);
		pool.parallel_for(parts, [&](int lo, int hi) { const size_t a = (size_t) lo << 20, b = std::min(nb, (size_t) hi << 20); memcpy(to + a, from + a, b - a); }, 1);
		if (hipMemcpyAsync((char *) dst + off, to, nb, hipMemcpyHostToDevice, st) != hipSuccess || hipEventRecord(done[turn], st) != hipSuccess) rc = -5;
	}
	if (hipStreamSynchronize(st) != hipSuccess) rc = -5;
	cleanup();
	if (rc) ngm::pipeline_set_error("copying the reference to the GPU failed");
	return rc;
}

int upload_genome(ngm_ref *r) {
	// artificial upper bound for positions on the last contig (SequenceProvider.cpp:378)
	r->start_pos.clear();
	for (const NgmContig &c : r->contigs) r->start_pos.push_back(c.start);
	if (!r->contigs.empty()) r->start_pos.push_back(r->contigs.back().start + r->contigs.back().len + 1000);
	// upload the genome as packed nibbles (+ one guard word so window reads never run off the end)
	r->genome_words = (r->n_bases + 7) / 8 + 64;
	std::vector<uint32_t, ngm::default_init_allocator<uint32_t>> packed(r->genome_words);
	{
		const uint64_t nb = r->n_bases, nw = (nb + 7) / 8;
		for (uint64_t wi = nw; wi < r->genome_words; ++wi) packed[wi] = 0x66666666u;  // NUL class beyond the end
		const uint8_t *cls = r->host_cls.data();
		uint32_t *out = packed.data();
		const int blocks = (int) ((nw + (1u << 17) - 1) >> 17);  // 128 Kword blocks on the pool threads
		ngm::ThreadPool::instance().parallel_for(blocks, [&](int lo, int hi) {
			for (uint64_t wi = (uint64_t) lo << 17, e = std::min<uint64_t>(nw, (uint64_t) hi << 17); wi < e; ++wi) {
				uint32_t w = 0x66666666u;
				for (int b = 0; b < 8; ++b) {
					const uint64_t i = wi * 8 + (uint64_t) b;
					if (i < nb) w = (w & ~(0xFu << (4 * b))) | ((uint32_t) cls[i] << (4 * b));
				}
				out[wi] = w;
			}
		}, 1);
	}
	REF_HIP_TRY(hipMalloc(&r->d_genome, r->genome_words * 4));
	return upload_staged(r->d_genome, packed.data(), r->genome_words * 4);
}

int finish_ref(ngm_ref *r) {
	if (int rc = upload_genome(r)) return rc;
	return build_index(r);
}

void append_spacer(ngm::ByteVec &g) { g.insert(g.end(), 1000, (uint8_t) 5); }

void append_contig(ngm_ref *r, const std::string &name, const uint8_t *seq, uint64_t len) {
	if (len <= 10) return;  // minRefSeqLen, SequenceProvider.h:71 / .cpp:300
	NgmContig c;
	c.name = name.substr(0, 100);
	c.start = r->host_cls.size();
	c.len = len;
	r->host_cls.reserve(r->host_cls.size() + len + 1002);
	r->host_cls.resize(c.start + len);
	{
		uint8_t *dst = r->host_cls.data() + c.start;
		const int blocks = (int) ((len + (1u << 20) - 1) >> 20);  // 1 Mbase blocks on the pool threads
		ngm::ThreadPool::instance().parallel_for(blocks, [&](int lo, int hi) {
			for (uint64_t i = (uint64_t) lo << 20, e = std::min<uint64_t>(len, (uint64_t) hi << 20); i < e; ++i) dst[i] = class_of_base(seq[i]);
		}, 1);
	}
	if (len & 1) r->host_cls.push_back(5);  // odd contig: the second nibble of the last byte is an N
	append_spacer(r->host_cls);
	r->contigs.push_back(c);
}

ngm_ref *new_ref(int device, const ngm_ref_params *p) {
	if (!p || p->kmer < 4 || p->kmer > 15 || p->kmer_skip < 0 || p->bin_size < 0 || p->bin_size > 8) {
		ngm::pipeline_set_error("ngm_ref_create: bad parameters (kmer 4..15, kmer_skip >= 0, bin_size 0..8)");
		return nullptr;
	}
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
		ngm::pipeline_set_error("ngm_ref_create: no HIP device available (this library has no CPU fallback)");
		return nullptr;
	}
	if (device < 0 || device >= ndev || hipSetDevice(device) != hipSuccess) {
		ngm::pipeline_set_error("ngm_ref_create: cannot select device %d", device);
		return nullptr;
	}
	ngm_ref *r = new ngm_ref();
	r->device = device;
	r->prm = *p;
	append_spacer(r->host_cls);  // padding so that windows never start at a negative position
	return r;
}

ngm_ref *seal(ngm_ref *r) {
	r->n_bases = r->host_cls.size();
	if (r->contigs.empty()) { ngm::pipeline_set_error("ngm_ref_create: no usable contig"); ngm_ref_destroy(r); return nullptr; }
	if (r->n_bases >= 0xFFFFFFFFull) { ngm::pipeline_set_error("ngm_ref_create: references >= 4 Gbp (multi-unit index) are not supported yet"); ngm_ref_destroy(r); return nullptr; }
	if (finish_ref(r) != 0) { ngm_ref_destroy(r); return nullptr; }
	return r;
}
}  // namespace

namespace ngm {
void pipeline_set_error(const char *fmt, ...) {
	char buf[1024];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof(buf), fmt, ap);
	va_end(ap);
	g_pipeline_error = buf;
}
}  // namespace ngm

extern "C" {

const char *ngm_pipeline_last_error(void) { return g_pipeline_error.c_str(); }

ngm_ref *ngm_ref_create(int device, const ngm_ref_params *p, int n_contigs, const char *const *names,
		const uint8_t *const *seqs, const uint64_t *lens) {
	ngm_ref *r = new_ref(device, p);
	if (!r) return nullptr;
	for (int i = 0; i < n_contigs; ++i) append_contig(r, names[i], seqs[i], lens[i]);
	return seal(r);
}

// NextGenMap's own cache files next to the FASTA: <ref>-enc.2.ngm (SequenceProvider.cpp:189-208, :228-262) and
// <ref>-ht-<k>-<skip>.3.ngm (PrefixTable.cpp:819-855, :857-930).  nullptr (with the reason as last error) when they are
// absent, were written with other parameters or are not single-unit indexes.
namespace {
// a cache file mapped read-only (the kernel's page cache is the only copy of its bytes on the host)
struct MappedCache {
	const uint8_t *p = nullptr;
	size_t n = 0;
	int fd = -1;
	bool open(const char *path) {
		fd = ::open(path, O_RDONLY);
		if (fd < 0) return false;
		struct stat st;
		if (fstat(fd, &st) != 0 || st.st_size <= 0) return false;
		n = (size_t) st.st_size;
		void *m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
		if (m == MAP_FAILED) return false;
		p = (const uint8_t *) m;
		return true;
	}
	~MappedCache() { if (p) munmap((void *) p, n); if (fd >= 0) close(fd); }
};
}  // namespace

// NextGenMap's cache files next to the FASTA: <fasta>-enc.2.ngm (SequenceProvider.cpp:189-208, :264-330) and
// <fasta>-ht-<k>-<skip>.3.ngm (PrefixTable.cpp:819-903).  Both files are mapped; everything that touches the 1.5 + 4.4 GB of a
// GRCh38-size reference (decoding, checks, the copies to the GPU) runs on all pool threads.
ngm_ref *ngm_ref_create_from_cache(int device, const ngm_ref_params *p, const char *fasta_path) {
	if (!p || !fasta_path) { ngm::pipeline_set_error("ngm_ref_create_from_cache: null argument"); return nullptr; }
	const std::string enc_fn = std::string(fasta_path) + "-enc.2.ngm";
	const std::string ht_fn = std::string(fasta_path) + "-ht-" + std::to_string(p->kmer) + "-" + std::to_string(p->kmer_skip) + ".3.ngm";
	MappedCache fe, fh;
	if (!fe.open(enc_fn.c_str()) || !fh.open(ht_fn.c_str())) { ngm::pipeline_set_error("no index cache next to %s", fasta_path); return nullptr; }
	const bool timing = getenv("NGM_HIP_HOST_TIMING") != nullptr;
	auto t0 = std::chrono::steady_clock::now();
	auto lap = [&](const char *what) {
		if (!timing) return;
		const auto t1 = std::chrono::steady_clock::now();
		fprintf(stderr, "[ngm-hip] index cache: %s %.3f s\n", what, std::chrono::duration<double>(t1 - t0).count());
		t0 = t1;
	};
	ngm_ref *r = new_ref(device, p);
	if (!r) return nullptr;
	auto fail = [&](const char *why) -> ngm_ref * { ngm::pipeline_set_error("index cache of %s: %s", fasta_path, why); ngm_ref_destroy(r); return nullptr; };
	ngm::ThreadPool &pool = ngm::ThreadPool::instance();
	struct RefIdx { uint32_t SeqId, Flags; uint64_t SeqStart; uint32_t SeqLen, NameLen; char name[100]; uint32_t pad; };
	static_assert(sizeof(RefIdx) == 128, "RefIdx layout");
	size_t at = 0;
	auto take = [](const MappedCache &f, size_t &o, void *dst, size_t bytes) -> bool { if (o + bytes > f.n) return false; memcpy(dst, f.p + o, bytes); o += bytes; return true; };
	uint32_t cookie = 0, ref_count = 0;
	uint64_t bin_ref_index = 0, enc_size = 0;
	if (!take(fe, at, &cookie, 4) || !take(fe, at, &ref_count, 4) || !take(fe, at, &bin_ref_index, 8) || !take(fe, at, &enc_size, 8) ||
			cookie != 0x74656 || ref_count == 0 || bin_ref_index >= 0xFFFFFFFFull || enc_size < bin_ref_index / 2) return fail("bad header of the encoded reference");
	for (uint32_t i = 0; i < ref_count; ++i) {
		RefIdx x;
		if (!take(fe, at, &x, sizeof(x))) return fail("truncated contig table");
		NgmContig c;
		c.name.assign(x.name, std::min<uint32_t>(x.NameLen, 100));
		c.start = x.SeqStart; c.len = x.SeqLen;
		r->contigs.push_back(c);
	}
	if (at + enc_size > fe.n) return fail("truncated sequence data");
	{
		// 4 bits per base, first base in the high nibble, A0 T1 G2 C3 N4 (SequenceProvider.cpp:72-85) -> classes A0 C1 G2 T3 N5
		const uint8_t *data = fe.p + at;
		r->host_cls.resize(bin_ref_index);
		uint8_t *cls = r->host_cls.data();
		const uint64_t n_bytes = (bin_ref_index + 1) / 2;
		pool.parallel_for((int) ((n_bytes + (1u << 20) - 1) >> 20), [&](int lo, int hi) {
			static const uint8_t dec[16] = {0, 3, 2, 1, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5};
			for (uint64_t b = (uint64_t) lo << 20, e = std::min<uint64_t>(n_bytes, (uint64_t) hi << 20); b < e; ++b) {
				const uint8_t x = data[b];
				cls[2 * b] = dec[x >> 4];
				if (2 * b + 1 < bin_ref_index) cls[2 * b + 1] = dec[x & 15];
			}
		}, 1);
	}
	r->n_bases = bin_ref_index;
	lap("sequence decoded");
	uint32_t kk = 0, skip = 0, units = 0, index_size = 0, table_len = 0;
	cookie = 0;
	at = 0;
	if (!take(fh, at, &cookie, 4) || !take(fh, at, &kk, 4) || !take(fh, at, &skip, 4) || !take(fh, at, &units, 4) || !take(fh, at, &index_size, 4) ||
			!take(fh, at, &table_len, 4) || cookie != 0x74656) return fail("bad header of the k-mer table");
	const uint32_t n_kmers = 1u << (2 * p->kmer);
	if ((int) kk != p->kmer || (int) skip != p->kmer_skip || index_size != n_kmers + 1) return fail("k-mer table was built with other parameters");
	if (units != 1) return fail("multi-unit k-mer tables (references >= 4 Gbp) are not supported yet");
	const size_t ib_bytes = (size_t) index_size * 5, pos_bytes = (size_t) table_len * 4;
	if (at + ib_bytes + pos_bytes + 12 > fh.n) return fail("truncated k-mer table");
	const uint8_t *ib = fh.p + at;
	const uint8_t *pos = ib + ib_bytes;
	{
		// trailer: offset of the unit + signature (PrefixTable.cpp:838-846, checked by the reference at :892-903)
		size_t o = at + ib_bytes + pos_bytes;
		uint64_t unit_offset = 0;
		uint32_t signature = 0;
		if (!take(fh, o, &unit_offset, 8) || !take(fh, o, &signature, 4) || signature != cookie + kk + skip + units + index_size)
			return fail("k-mer table without a valid signature (truncated or written by another version)");
	}
	// a corrupt file must not turn into out-of-bounds reads on the GPU: contigs inside the encoded genome, index offsets
	// monotonic and inside the position table
	for (const NgmContig &c : r->contigs) if (c.start + c.len > bin_ref_index || c.len == 0) return fail("contig table does not match the sequence data");
	std::vector<uint32_t, ngm::default_init_allocator<uint32_t>> raw(n_kmers);
	std::vector<uint2, ngm::default_init_allocator<uint2>> idx(n_kmers);
	{
		std::atomic<bool> corrupt{false};
		auto entry = [&](uint32_t q) { uint32_t t; memcpy(&t, ib + (size_t) q * 5, 4); return t; };
		pool.parallel_for((int) ((n_kmers + (1u << 16) - 1) >> 16), [&](int lo, int hi) {
			const uint32_t a = (uint32_t) lo << 16, e = (uint32_t) std::min<uint64_t>(n_kmers, (uint64_t) hi << 16);
			uint32_t t0 = entry(a);
			if (t0 < 1u) { corrupt = true; return; }
			for (uint32_t q = a; q < e; ++q) {
				const uint32_t t1 = entry(q + 1);
				if (t1 < t0 || t1 > table_len + 1) { corrupt = true; return; }
				raw[q] = t1 - t0;
				// the usage byte doubles as "skip this k-mer" (PrefixTable.cpp:468-478, :771-775)
				idx[q] = make_uint2(t0 - 1, ib[(size_t) q * 5 + 4] ? t1 - t0 : 0u);
				t0 = t1;
			}
		}, 1);
		if (corrupt) return fail("corrupt k-mer table index");
	}
	r->n_entries = table_len;
	lap("index entries checked");
	if (upload_genome(r) != 0) { ngm_ref_destroy(r); return nullptr; }
	lap("sequence packed and copied");
	const size_t pos_alloc = ((size_t) table_len + 16) * 4;
	if (hipMalloc((void **) &r->d_index, (size_t) n_kmers * 8) != hipSuccess || hipMalloc((void **) &r->d_raw_counts, (size_t) n_kmers * 4) != hipSuccess ||
			hipMalloc((void **) &r->d_positions, pos_alloc) != hipSuccess || hipMemset((char *) r->d_positions + pos_bytes, 0, pos_alloc - pos_bytes) != hipSuccess) {
		ngm::pipeline_set_error("out of device memory loading the index cache");
		ngm_ref_destroy(r);
		return nullptr;
	}
	if (upload_staged(r->d_index, idx.data(), (size_t) n_kmers * 8) != 0 || upload_staged(r->d_raw_counts, raw.data(), (size_t) n_kmers * 4) != 0 ||
			upload_staged(r->d_positions, pos, pos_bytes) != 0) { ngm_ref_destroy(r); return nullptr; }
	lap("k-mer table copied");
	index_stats(r, raw.data(), n_kmers);
	r->from_cache = true;
	return r;
}

ngm_ref *ngm_ref_create_from_fasta(int device, const ngm_ref_params *p, const char *path) {
	// like the reference, an index cache next to the FASTA is used instead of rebuilding (NGM_HIP_NO_CACHE=1: always rebuild)
	if (!getenv("NGM_HIP_NO_CACHE")) {
		if (ngm_ref *cached = ngm_ref_create_from_cache(device, p, path)) return cached;
	}
	gzFile f = gzopen(path, "rb");
	if (!f) { ngm::pipeline_set_error("cannot open reference %s", path); return nullptr; }
	ngm_ref *r = new_ref(device, p);
	if (!r) { gzclose(f); return nullptr; }
	std::string name, seq, line;
	bool have = false;
	std::vector<char> buf(1 << 20);
	std::string carry;
	auto flush = [&]() { if (have) append_contig(r, name, (const uint8_t *) seq.data(), seq.size()); seq.clear(); };
	auto handle_line = [&](const char *s, size_t n) {
		while (n && (s[n - 1] == '\r' || s[n - 1] == '\n')) --n;
		if (n && s[0] == '>') {
			flush();
			size_t e = 1;
			while (e < n && s[e] != ' ' && s[e] != '\t') ++e;  // kseq: name = up to the first whitespace
			name.assign(s + 1, e - 1);
			have = true;
		} else if (have) {
			for (size_t i = 0; i < n; ++i) if (s[i] != ' ' && s[i] != '\t') seq.push_back(s[i]);
		}
	};
	int got;
	while ((got = gzread(f, buf.data(), (unsigned) buf.size())) > 0) {
		size_t b = 0;
		for (size_t i = 0; i < (size_t) got; ++i) {
			if (buf[i] == '\n') {
				if (!carry.empty()) { carry.append(buf.data() + b, i - b); handle_line(carry.data(), carry.size()); carry.clear(); }
				else handle_line(buf.data() + b, i - b);
				b = i + 1;
			}
		}
		carry.append(buf.data() + b, (size_t) got - b);
	}
	if (!carry.empty()) handle_line(carry.data(), carry.size());
	flush();
	gzclose(f);
	return seal(r);
}

void ngm_ref_destroy(ngm_ref *r) {
	if (!r) return;
	(void) hipSetDevice(r->device);
	if (r->d_genome) (void) hipFree(r->d_genome);
	if (r->d_index) (void) hipFree(r->d_index);
	if (r->d_raw_counts) (void) hipFree(r->d_raw_counts);
	if (r->d_positions) (void) hipFree(r->d_positions);
	if (r->d_buckets) (void) hipFree(r->d_buckets);
	if (r->d_cbuckets) (void) hipFree(r->d_cbuckets);
	delete r;
}

}  // extern "C"

int ngm_ref_ensure_buckets(const ngm_ref *r, int kind) {
	std::lock_guard<std::mutex> lk(g_bucket_mu);
	int prev = -1;
	(void) hipGetDevice(&prev);
	(void) hipSetDevice(r->device);
	const int rc = build_buckets_kind(const_cast<ngm_ref *>(r), kind);
	if (prev >= 0 && prev != r->device) (void) hipSetDevice(prev);
	return rc;
}

extern "C" {

int ngm_ref_prepare_search(ngm_ref *r, int bs_mapping) {
	if (!r) return -22;
	if (bs_mapping) return 0;
	const bool canon = (r->prm.kmer & 1) != 0;   // (mapper.cpp: reads of more than 256 k-mers fall back to one bucket per k-mer)
	return ngm_ref_ensure_buckets(r, canon ? 1 : 0);
}

int ngm_ref_host_classes(const ngm_ref *r, uint64_t pos, int n, uint8_t *out) {
	if (!r || !out || n < 0) return -22;
	int k = 0;
	for (; k < n && pos + (uint64_t) k < r->host_cls.size(); ++k) out[k] = r->host_cls[pos + (uint64_t) k];
	return k;
}

int ngm_ref_contig_count(const ngm_ref *r) { return (int) r->contigs.size(); }
const char *ngm_ref_contig_name(const ngm_ref *r, int i) { return r->contigs[i].name.c_str(); }
uint64_t ngm_ref_contig_start(const ngm_ref *r, int i) { return r->contigs[i].start; }
uint64_t ngm_ref_contig_len(const ngm_ref *r, int i) { return r->contigs[i].len; }
uint64_t ngm_ref_concat_len(const ngm_ref *r) { return r->n_bases - 1; }  // SequenceProvider.cpp:454-456
int ngm_ref_auto_max_kfreq(const ngm_ref *r) { return r->auto_max_kfreq; }
int ngm_ref_loaded_from_cache(const ngm_ref *r) { return r->from_cache ? 1 : 0; }
uint64_t ngm_ref_index_entries(const ngm_ref *r) { return r->n_entries; }

int ngm_ref_index_copy(const ngm_ref *r, uint32_t *counts, uint32_t *raw_counts, uint32_t *positions) {
	(void) hipSetDevice(r->device);
	const uint32_t n_kmers = 1u << (2 * r->prm.kmer);
	if (counts) {
		std::vector<uint2> idx(n_kmers);
		REF_HIP_TRY(hipMemcpy(idx.data(), r->d_index, (size_t) n_kmers * sizeof(uint2), hipMemcpyDeviceToHost));
		for (uint32_t i = 0; i < n_kmers; ++i) counts[i] = idx[i].y;
	}
	if (raw_counts) REF_HIP_TRY(hipMemcpy(raw_counts, r->d_raw_counts, (size_t) n_kmers * 4, hipMemcpyDeviceToHost));
	if (positions && r->n_entries) REF_HIP_TRY(hipMemcpy(positions, r->d_positions, r->n_entries * 4, hipMemcpyDeviceToHost));
	return 0;
}

// A cache file appears under its name complete or not at all: it is written to <name>.tmp.<pid>, every write is checked, and then
// renamed (another process -- a sibling shard, another run -- may be mapping the old file while this one is written)
namespace {
struct CacheFile {
	std::string fn, tmp;
	FILE *fp = nullptr;
	bool ok = true;
	explicit CacheFile(const std::string &name) : fn(name), tmp(name + ".tmp." + std::to_string((long) getpid())) { fp = fopen(tmp.c_str(), "wb"); ok = fp != nullptr; }
	void put(const void *p, size_t size, size_t n) { if (ok && n && fwrite(p, size, n, fp) != n) ok = false; }
	bool commit() {
		if (fp && fclose(fp) != 0) ok = false;
		fp = nullptr;
		if (ok && rename(tmp.c_str(), fn.c_str()) != 0) ok = false;
		if (!ok) (void) unlink(tmp.c_str());
		return ok;
	}
	~CacheFile() { if (fp) { fclose(fp); (void) unlink(tmp.c_str()); } }
};
}  // namespace

int ngm_ref_write_ngm_cache(const ngm_ref *r, const char *fasta_path) {
	(void) hipSetDevice(r->device);
	const int k = r->prm.kmer;
	const uint32_t n_kmers = 1u << (2 * k);
	// ---- <ref>-enc.2.ngm -------------------------------------------------------------------------------
	{
		const std::string fn = std::string(fasta_path) + "-enc.2.ngm";
		CacheFile cf(fn);
		if (!cf.ok) { ngm::pipeline_set_error("cannot write %s", fn.c_str()); return -13; }
		const uint32_t cookie = 0x74656, ref_count = (uint32_t) r->contigs.size();
		uint64_t size = 1000;  // getSize(): SequenceProvider.cpp:210-226
		for (const NgmContig &c : r->contigs) size += ((c.len | 1) + 1) + 1000;
		const uint64_t bin_ref_index = r->n_bases, enc_size = ((size / 2) | 1) + 1;
		cf.put(&cookie, 4, 1); cf.put(&ref_count, 4, 1); cf.put(&bin_ref_index, 8, 1); cf.put(&enc_size, 8, 1);
		struct RefIdx { uint32_t SeqId, Flags; uint64_t SeqStart; uint32_t SeqLen, NameLen; char name[100]; uint32_t pad; };
		static_assert(sizeof(RefIdx) == 128, "RefIdx layout");
		for (size_t i = 0; i < r->contigs.size(); ++i) {
			RefIdx x{};
			x.SeqId = (uint32_t) i; x.SeqStart = r->contigs[i].start; x.SeqLen = (uint32_t) r->contigs[i].len;
			x.NameLen = (uint32_t) std::min<size_t>(100, r->contigs[i].name.size());
			memcpy(x.name, r->contigs[i].name.data(), x.NameLen);
			cf.put(&x, sizeof(x), 1);
		}
		// 4 bits per base, first base in the high nibble, A0 T1 G2 C3 N4 (SequenceProvider.cpp:72-85, :310-313)
		static const uint8_t enc[8] = {0, 3, 2, 1, 4, 4, 4, 4};
		std::vector<uint8_t> data(enc_size, 0);
		for (uint64_t i = 0; i + 1 < r->n_bases; i += 2) data[i >> 1] = (uint8_t) ((enc[r->host_cls[i] & 7] << 4) | enc[r->host_cls[i + 1] & 7]);
		cf.put(data.data(), 1, enc_size);
		if (!cf.commit()) { ngm::pipeline_set_error("write error on %s", fn.c_str()); return -5; }
	}
	// ---- <ref>-ht-<k>-<skip>.3.ngm ---------------------------------------------------------------------
	{
		const std::string fn = std::string(fasta_path) + "-ht-" + std::to_string(k) + "-" + std::to_string(r->prm.kmer_skip) + ".3.ngm";
		CacheFile cf(fn);
		if (!cf.ok) { ngm::pipeline_set_error("cannot write %s", fn.c_str()); return -13; }
		const uint32_t cookie = 0x74656, kk = (uint32_t) k, skip = (uint32_t) r->prm.kmer_skip, units = 1, index_size = n_kmers + 1;
		cf.put(&cookie, 4, 1); cf.put(&kk, 4, 1); cf.put(&skip, 4, 1); cf.put(&units, 4, 1); cf.put(&index_size, 4, 1);
		std::vector<uint32_t> raw(n_kmers);
		std::vector<uint2> idx(n_kmers);
		REF_HIP_TRY(hipMemcpy(raw.data(), r->d_raw_counts, (size_t) n_kmers * 4, hipMemcpyDeviceToHost));
		REF_HIP_TRY(hipMemcpy(idx.data(), r->d_index, (size_t) n_kmers * 8, hipMemcpyDeviceToHost));
		std::vector<uint32_t> pos(r->n_entries + 1, 0);
		if (r->n_entries) REF_HIP_TRY(hipMemcpy(pos.data(), r->d_positions, r->n_entries * 4, hipMemcpyDeviceToHost));
		const uint32_t table_len = (uint32_t) r->n_entries;
		cf.put(&table_len, 4, 1);
		// Index { uint m_TabIndex; char m_RevCompIndex; } packed to 5 bytes (PrefixTable.h:18-61)
		std::vector<uint8_t> ib((size_t) index_size * 5, 0);
		uint32_t next = 0;
		for (uint32_t p = 0; p < n_kmers; ++p) {
			const uint32_t tab = next + 1;
			memcpy(&ib[(size_t) p * 5], &tab, 4);
			if (raw[p] > 0) {
				const uint32_t total = raw[p] + raw[ngm::kmer_revcomp(p, k)];
				const int dummy = 10000;
				ib[(size_t) p * 5 + 4] = (uint8_t) (char) ((dummy - (int) std::min<uint32_t>(total, dummy)) * 100.0f / dummy);
				if (idx[p].y == 0) for (uint32_t j = 0; j < raw[p]; ++j) pos[next + j] = 0;  // unused k-mers keep empty slots
			}
			next += raw[p];
		}
		const uint32_t tab = next + 1;
		memcpy(&ib[(size_t) n_kmers * 5], &tab, 4);
		cf.put(ib.data(), 1, ib.size());
		cf.put(pos.data(), 4, table_len);
		const uint64_t offset = 0;
		cf.put(&offset, 8, 1);
		const uint32_t signature = cookie + kk + skip + units + index_size;
		cf.put(&signature, 4, 1);
		if (!cf.commit()) { ngm::pipeline_set_error("write error on %s", fn.c_str()); return -5; }
	}
	return 0;
}

// _SequenceProvider::convert (SequenceProvider.cpp:111-141)
int ngm_ref_convert(const ngm_ref *r, uint64_t pos, int *contig, uint64_t *contig_pos) {
	auto upper = std::upper_bound(r->start_pos.begin(), r->start_pos.end(), pos);
	if (upper == r->start_pos.end() || upper == r->start_pos.begin()) return 0;
	if ((*upper - pos) < 1000) return 0;  // inside the spacer in front of the next contig
	*contig = (int) ((upper - 1) - r->start_pos.begin());
	*contig_pos = pos - *(upper - 1);
	return 1;
}

}  // extern "C"
