// refindex.h -- encoded reference + k-mer index, host description shared by the pipeline TUs.
//
// Replaces NextGenMap's _SequenceProvider (src/SequenceProvider.cpp) and CompactPrefixTable
// (src/PrefixTable.cpp).  Coordinates are the reference's: all contigs concatenated, 1000 'N' before the
// first, between and after contigs, every contig starting at an even position (SequenceProvider.cpp:289-326).
//
// HBM layout (MI355X: everything replicated per GPU, 288 GB is plenty for GRCh38):
//   d_genome    : 4-bit symbol classes (A0 C1 G2 T3 N5, the DP kernels' alphabet), 8 bases per dword,
//                 base i in bits 4*(i%8)                                   -- 1.55 GB for GRCh38
//   d_index     : 4^k entries {start, count}: one 8-byte random read per k-mer lookup (the reference reads two
//                 5-byte packed entries, PrefixTable.h:27-61); count is already 0 for k-mers that the reference
//                 treats as unused (>= 9901 occurrences fwd+revcomp, PrefixTable.cpp:468-478)  -- 537 MB
//   d_positions : one uint32 per indexed k-mer occurrence, grouped by k-mer, ascending        -- ~4.1 GB
//   d_buckets   : the layout the candidate search actually gathers from: one aligned bucket of W = 4..32 dwords per
//                 k-mer, {count, position 0 .. W-2}, so that a lookup is ONE memory request instead of index entry ->
//                 position list (measured on MI355X, profiles/r02_gather_calibration.txt: random gathers are bound by
//                 ~50 G requests/s whatever their size up to 128 B, where they reach the 6.2 TB/s streaming ceiling).
//                 Lists longer than W-1 keep {count | 1<<31, offset into the position table}; a copy of that table follows the
//                 buckets in the same allocation, so one 32-bit word offset addresses either kind of list.  W from the mean list length
//                 (GRCh38 size: 15.4 -> W = 32, 128 B, 99.98 % of the lists inline)                -- 8.6 GB
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "../../include/ngm_pipeline.h"

namespace ngm {
// std::vector whose resize() leaves new elements uninitialised: the multi-GB host arrays of the reference are filled by all pool
// threads right after they are sized -- value-initialising them first is a second, single-threaded pass over the same memory
template <class T> struct default_init_allocator : std::allocator<T> {
	template <class U> struct rebind { using other = default_init_allocator<U>; };
	template <class U, class... A> void construct(U *p, A &&...a) {
		if constexpr (sizeof...(A) == 0) ::new ((void *) p) U; else ::new ((void *) p) U(std::forward<A>(a)...);
	}
};
using ByteVec = std::vector<uint8_t, default_init_allocator<uint8_t>>;
}  // namespace ngm

struct NgmContig {
	std::string name;
	uint64_t start;  // concatenated coordinate of base 0
	uint64_t len;
};

struct ngm_ref {
	int device = 0;
	ngm_ref_params prm{};
	std::vector<NgmContig> contigs;
	std::vector<uint64_t> start_pos;  // contig starts + one artificial upper bound (SequenceProvider.cpp:370-378)
	uint64_t n_bases = 0;             // nibbles in the encoded genome (binRefIndex after doubling)
	ngm::ByteVec host_cls;            // one class per base (kept for the host-side helpers)
	uint64_t n_entries = 0;
	int auto_max_kfreq = 100;
	bool from_cache = false;          // loaded from NextGenMap's cache files instead of being built
	// device
	uint32_t *d_genome = nullptr;
	uint64_t genome_words = 0;
	uint2 *d_index = nullptr;
	uint32_t *d_raw_counts = nullptr;
	uint32_t *d_positions = nullptr;
	uint32_t *d_buckets = nullptr;
	int bucket_log2_words = 2;        // W = 1 << bucket_log2_words dwords per k-mer bucket
	uint32_t bucket_pos_base = 0;     // word offset of the position table copy that follows the buckets in d_buckets
	double overflow_hit_share = 0.0;  // share of all index entries that live in lists longer than W-1
	// canonical buckets (round 3, odd k): ONE bucket per {k-mer, reverse complement} pair -- the search always needs both lists
	// (CS.cpp:116-160), so they live side by side: word 0 = length of the canonical k-mer's list | the other one's << 14 | 1 << 31
	// when the two do not fit, words 1.. = the canonical k-mer's positions followed by its reverse complement's (a pair that does
	// not fit keeps the two list starts in the position table in words 1 and 2).  The canonical one of a pair is the k-mer whose
	// MIDDLE base is A or C (bit 1 of its 2-bit code clear: complementing flips exactly that bit, and the middle base of an odd
	// k-mer stays in the middle), its bucket number the k-mer with that bit removed: a bijection onto [0, 4^k / 2), no table.
	// W = 4..64 dwords from the mean length of both lists (GRCh38 size: 30.8 -> 64 dwords = two 128-byte lines, the second one
	// only read for the ~45 % of the pairs with more than 31 positions).  Same allocation scheme: a copy of the position table
	// follows the buckets.
	uint32_t *d_cbuckets = nullptr;
	int cbucket_log2_words = 2;
	uint32_t cbucket_pos_base = 0;
};

// builds the bucket layout `kind` (0: one bucket per k-mer, 1: canonical pairs; odd k only) if it does not exist yet; 0 or -errno
int ngm_ref_ensure_buckets(const ngm_ref *r, int kind);

namespace ngm {
void pipeline_set_error(const char *fmt, ...);
// k-mer integer as the reference builds it: 2 bits per base, A0 C1 T2 G3 ((c >> 1) & 3, CSstatic.cpp:20-22)
inline uint32_t kmer_code_of_class(uint32_t cls) { return cls == 2 ? 3u : (cls == 3 ? 2u : cls); }
// reverse complement of a k-mer integer (PrefixTable.cpp:94-108), valid for 2k <= 32
inline uint32_t kmer_revcomp(uint32_t prefix, int k);
// canonical pairs (odd k): bit 2*(k/2)+1 of a k-mer -- the high bit of its middle base -- tells the two k-mers of a pair apart
inline int kmer_canon_bit(int k) { return 2 * (k / 2) + 1; }
inline uint32_t kmer_canon_id(uint32_t canon_kmer, int k) { const int b = kmer_canon_bit(k); return ((canon_kmer >> (b + 1)) << b) | (canon_kmer & ((1u << b) - 1u)); }
inline uint32_t kmer_of_canon_id(uint32_t id, int k) { const int b = kmer_canon_bit(k); return ((id >> b) << (b + 1)) | (id & ((1u << b) - 1u)); }
inline uint32_t kmer_revcomp(uint32_t prefix, int k) {
	const int shift = 32 - 2 * k;
	uint32_t c = (prefix ^ 0xAAAAAAAAu) << shift;
	c = (c & 0xFFFF0000u) >> 16 | (c & 0x0000FFFFu) << 16;
	c = (c & 0xFF00FF00u) >> 8 | (c & 0x00FF00FFu) << 8;
	c = (c & 0xF0F0F0F0u) >> 4 | (c & 0x0F0F0F0Fu) << 4;
	c = (c & 0xCCCCCCCCu) >> 2 | (c & 0x33333333u) << 2;
	return c;
}
}  // namespace ngm
