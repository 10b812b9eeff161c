/*
 * ngm_pipeline.h -- flat C ABI of the device-resident mapping path that sits ABOVE IAlignment in
 * NextGenMap: encoded reference + k-mer index in HBM, candidate search, window gather, score,
 * candidate selection / MAPQ, alignment.  Rows a1-a5, a7, a8 of SURVEY.md section 8.
 *
 * Reference interfaces replaced (paths relative to the NextGenMap tree):
 *   ngm_ref_*        <- _SequenceProvider (src/SequenceProvider.cpp:228-441: 4-bit concatenated genome with
 *                       1000-N spacers, DecodeRefSequence, convert) and CompactPrefixTable
 *                       (src/PrefixTable.cpp:328-498, :641-817: k-mer index, GetRefEntry, stats/max_kfreq)
 *   ngm_mapper_cs    <- CS::RunBatch / PrefixSearch / AddLocationStd / CollectResultsStd (src/CS.cpp:114-313)
 *   ngm_mapper_map   <- ScoreBuffer::DoRun + top1SE + computeMQ (src/ScoreBuffer.cpp:34-49, :80-277) and
 *                       AlignmentBuffer::DoRun (src/AlignmentBuffer.cpp:64-147) around BatchScore / BatchAlign
 * Plain pointers and sizes only.  All functions return >= 0 on success, a negative errno-style value on
 * failure (ngm_pipeline_last_error() describes it).  No CPU fallback: a HIP device is required.
 */
#ifndef NGM_PIPELINE_H
#define NGM_PIPELINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ngm_ref ngm_ref;
typedef struct ngm_mapper ngm_mapper;

typedef struct ngm_ref_params {
	int kmer;        /* Config "kmer", default 13 (src/config/Config.cpp:383); 4..15 supported here */
	int kmer_skip;   /* Config "kmer_skip", default 2: every (skip+1)-th reference k-mer is indexed */
	int bin_size;    /* Config "bin_size", default 2: candidate bins are 2^bin_size bases wide */
} ngm_ref_params;

const char *ngm_pipeline_last_error(void);

/* Encode a reference (contigs in order; names NUL-terminated; sequences ASCII, any case, non-ACGT -> N;
 * contigs of length <= 10 are skipped like the reference does, SequenceProvider.h:71) and build its
 * k-mer index on `device`. */
ngm_ref *ngm_ref_create(int device, const ngm_ref_params *p, int n_contigs, const char *const *names,
		const uint8_t *const *seqs, const uint64_t *lens);
/* Same, reading a (optionally gzip-compressed) FASTA file. */
ngm_ref *ngm_ref_create_from_fasta(int device, const ngm_ref_params *p, const char *path);
/* Loads NextGenMap's own cache files next to fasta_path -- <fasta>-enc.2.ngm (SequenceProvider.cpp:189-262) and
 * <fasta>-ht-<kmer>-<kmer_skip>.3.ngm (PrefixTable.cpp:819-930) -- written by `ngm` or by ngm_ref_write_ngm_cache, so
 * an existing NextGenMap index drops in.  NULL when absent / built with other parameters / multi-unit.
 * ngm_ref_create_from_fasta tries this first, like the reference (NGM_HIP_NO_CACHE=1 forces a rebuild). */
ngm_ref *ngm_ref_create_from_cache(int device, const ngm_ref_params *params, const char *fasta_path);
/* 1 when the reference came from those cache files (a corrupt / mismatching cache falls back to a build: the caller rewrites it) */
int ngm_ref_loaded_from_cache(const ngm_ref *r);
void ngm_ref_destroy(ngm_ref *r);

int ngm_ref_contig_count(const ngm_ref *r);
const char *ngm_ref_contig_name(const ngm_ref *r, int i);
uint64_t ngm_ref_contig_start(const ngm_ref *r, int i); /* position in concatenated coordinates */
uint64_t ngm_ref_contig_len(const ngm_ref *r, int i);
uint64_t ngm_ref_concat_len(const ngm_ref *r);          /* _SequenceProvider::GetConcatRefLen */
int ngm_ref_auto_max_kfreq(const ngm_ref *r);           /* ceil(max(100, avg + 5 sigma)), PrefixTable.cpp:150-194 */
uint64_t ngm_ref_index_entries(const ngm_ref *r);       /* number of stored k-mer positions */
/* copy the index out (tests): counts[4^k] = list length per k-mer as a lookup sees it (0 for k-mers
 * disabled by the 9900-occurrence rule, PrefixTable.cpp:468-478), raw_counts[4^k] = before that rule,
 * positions[index_entries] grouped by k-mer in k-mer order, ascending inside a group. NULL = skip. */
int ngm_ref_index_copy(const ngm_ref *r, uint32_t *counts, uint32_t *raw_counts, uint32_t *positions);
/* _SequenceProvider::DecodeRefSequence(buffer, 0, offset, buffer_len) executed from the HBM copy. */
int ngm_ref_decode(const ngm_ref *r, uint64_t offset, int buffer_len, char *out);
/* _SequenceProvider::convert: concatenated position -> (contig, 0-based position); returns 0 when the
 * position lies in a spacer (reported unmapped), 1 otherwise. */
int ngm_ref_convert(const ngm_ref *r, uint64_t pos, int *contig, uint64_t *contig_pos);

/* n symbol classes (A0 C1 G2 T3 x4 N5) of the encoded genome from concatenated position `pos` on, from the host copy (what a
 * record formatter needs to label the columns of an alignment, e.g. for the SLAM-seq tags); returns the number copied. */
int ngm_ref_host_classes(const ngm_ref *r, uint64_t pos, int n, uint8_t *out);

/* Builds the index layout the candidate search gathers from (one bucket per k-mer pair for odd k) now instead
 * of when the first mapper is created: part of preparing the reference, like loading NextGenMap's cache files
 * (src/PrefixTable.cpp:232-262).  bs_mapping != 0: nothing to build (that mode runs the exact search).  0, or -errno. */
int ngm_ref_prepare_search(ngm_ref *ref, int bs_mapping);

/* Write NextGenMap's own cache files next to `fasta_path` so that the reference program loads this encoded
 * genome and index instead of rebuilding them: <fasta_path>-enc.2.ngm (src/SequenceProvider.cpp:189-208) and
 * <fasta_path>-ht-<k>-<skip>.3.ngm (src/PrefixTable.cpp:819-855).  Content is what NGM itself would write. */
int ngm_ref_write_ngm_cache(const ngm_ref *r, const char *fasta_path);

typedef struct ngm_mapper_params {
	int qry_max_len;       /* bytes per read row */
	int corridor;
	int match_bonus, mismatch_penalty, gap_read_penalty, gap_ref_penalty;
	int mode;              /* 0 local, 1 end-to-end */
	int variant;           /* NGM_VARIANT_* of ngm_hip.h */
	float sensitivity;     /* Config "sensitivity" (-s) */
	float kmer_min;        /* Config "kmer_min" */
	int max_cmrs;          /* Config "max_cmrs" (INT_MAX = unlimited) */
	int max_kfreq;         /* <= 0: use ngm_ref_auto_max_kfreq */
	int hard_clip, silent_clip;
	int personality;       /* NGM_PERSONALITY_* of ngm_hip.h: 1 = `--affine` (EndToEndAffine / SeqAn scoring and CIGARs) */
	int gap_extend_penalty; /* Config "gap_extend_penalty" (affine personality) */
	/* paired-end selection (ScoreBuffer::top1PE / CheckPairs, src/ScoreBuffer.cpp:368-502) */
	int min_insert_size;   /* Config "min_insert_size" (-I), default 0 */
	int max_insert_size;   /* Config "max_insert_size" (-X), default 1000; <= 0: unlimited (src/NGM.cpp:38-41) */
	float pair_score_cutoff; /* Config "pair_score_cutoff"; <= 0: 0.9 */
	/* ScoreBuffer::topNSE (src/ScoreBuffer.cpp:279-327), single-end only as in the reference */
	int topn;              /* Config "topn" (-n): alignments reported per read; <= 1: one */
	int strata;            /* Config "strata": only the equally best ones, none if there are more than topn */
	/* `--bs-mapping` (src/CS.cpp:54-112, :340-376, :553-560; lib/mason/opencl/SWOcl.cpp:225-232): candidate search looks every read
	 * k-mer up in all its T>C (second mates: A>G) conversions, scoring uses the strand-specific tables.  The reference index must
	 * have been built with kmer_skip 0 (src/PrefixTable.cpp:199-207); the run's "kmer_skip" applies to the read instead. */
	int bs_mapping;        /* Config "bs_mapping" */
	int bs_cutoff;         /* Config "bs_cutoff" (6): k-mers with more convertible bases are not looked up */
	int bs_read_skip;      /* Config "kmer_skip" of the run (2) */
	int match_bonus_tt, match_bonus_tc;   /* Config MATCH_BONUS_TT / MATCH_BONUS_TC (4 / 4) */
	/* `--slam-seq <n>` (Config SLAM_SEQ): any value makes computeCigarMD count T>C (reverse strand: A>G) columns as matches and the
	 * writer add TC / RA / MP tags; bit 1 (2) also switches the score tables (scoresSlamSeqFWD / REV, match_bonus_tt / -match_bonus_tc);
	 * bit 2 (4): the weighted k-mer mutation search (src/CS.cpp:57-92, :133-138) -- every read k-mer and its single C > T (second mates
	 * G > A) conversions, float votes of 1 / (convertible bases + 1) summed in the reference's order (csrc/cs_slam_device.h). */
	int slam_seq;
} ngm_mapper_params;

ngm_mapper *ngm_mapper_create(const ngm_ref *ref, const ngm_mapper_params *p);
void ngm_mapper_destroy(ngm_mapper *m);

/* Candidate search for n reads (reads: n rows of qry_max_len bytes, upper-case ACGTN, NUL padded, host
 * memory).  Outputs: cand_offsets[n+1] (prefix sums), max_votes[n] (read->s, SAM XE:i); the candidates
 * themselves are fetched with ngm_mapper_cs_fetch: loc = bin centre in concatenated coordinates
 * (ResolveBin), strand 0/1, votes.  Candidate order inside a read is (loc, strand) ascending -- the
 * reference's order is first-threshold-crossing order, which only matters for exact score ties. */
int ngm_mapper_cs(ngm_mapper *m, int n, const char *reads, uint32_t *cand_offsets, float *max_votes);
int ngm_mapper_cs_fetch(ngm_mapper *m, uint64_t *loc, uint8_t *strand, float *votes);

/* after ngm_mapper_cs: per read, the largest forward + reverse vote sum of any bin -- what the reference's
 * sensitivity estimate is built from (ReadProvider.cpp:57-77, :79-124). out: n floats. */
int ngm_mapper_cs_max_combined(ngm_mapper *m, float *out);

/* Per-read mapping result (single-end, topn = 1). */
typedef struct ngm_hit {
	int mapped;            /* 0: no candidate / no alignment */
	int contig;            /* index into the reference contigs */
	uint64_t pos;          /* 0-based leftmost position on the contig */
	int reverse;           /* 1: read aligned as reverse complement */
	int mapq;
	float score;           /* AS:i (from the score stage, like the reference: SAMWriter.cpp:169) */
	float identity;        /* XI:f */
	int nm;
	int qstart, qend;
	int n_candidates;      /* CMRs scored for this read */
	int n_best;            /* NH:i / X0:i : candidates sharing the best score */
	float max_votes;       /* XE:i */
	int pair_flags;        /* paired-end runs: NGM_PAIR_* */
} ngm_hit;

#define NGM_PAIR_SELECTED 1  /* top1PE found a pair inside the insert-size window; n_best = pairs sharing its score and distance */
#define NGM_PAIR_FAILED 2    /* both mates had candidates but no such pair: NGMNames::PairedFail, mates selected single-end */
#define NGM_PAIR_LOST 4      /* the reference never writes this pair (ngm_mapper_set_reference_score_buffer): no record for either mate */

/* Full single-end path for n reads; cigars/mds: n rows of 4*qry_max_len bytes (NUL-terminated strings).
 * With topn > 1 every read owns topn consecutive entries of hits / rows of cigars and mds (entry k = k-th best
 * candidate, unused entries have mapped = 0 and n_candidates as usual); mapq and n_best are the read's. */
int ngm_mapper_map_se(ngm_mapper *m, int n, const char *reads, ngm_hit *hits, char *cigars, char *mds);
/* Same, with the read batch already resident in HBM (d_reads: n rows of qry_max_len bytes, device memory on
 * the mapper's GPU); `reads` is the host copy the CIGAR/MD pass consults.  What bench.py times. */
int ngm_mapper_map_se_resident(ngm_mapper *m, int n, const char *reads, const void *d_reads, ngm_hit *hits, char *cigars,
		char *mds);

/* Paired-end path: reads 2i and 2i+1 are mates (ReadProvider::GenerateRead, src/ReadProvider.cpp:526-584).
 * Candidate search, scoring and alignment are per read as above; the selection is ScoreBuffer::top1PE: among the
 * candidates within pair_score_cutoff of each mate's best, the best-scoring pair whose insert size lies inside
 * (min_insert_size, max_insert_size), ties by closeness to the running mean insert size (which is carried across
 * calls, like the reference's per-thread state).  MAPQ of a mate = its own best vs second-best candidate.
 * What the writer derives from both mates (proper-pair check, TLEN, flags) is the caller's: ngm-hip does it. */
int ngm_mapper_map_pe(ngm_mapper *m, int n, const char *reads, ngm_hit *hits, char *cigars, char *mds);
int ngm_mapper_map_pe_resident(ngm_mapper *m, int n, const char *reads, const void *d_reads, ngm_hit *hits, char *cigars,
		char *mds);

/* The SAM text of a batch, assembled on the GPU (csrc/sam_device.h): what GenericReadWriter::WriteRead / WritePair
 * (src/writer/GenericReadWriter.h:190-304: min_identity / min_residues / min_mq filters), AlignmentBuffer::WriteRead's
 * proper-pair check (src/AlignmentBuffer.cpp:175-199) and SAMWriter::DoWriteReadGeneric / DoWriteUnmappedReadGeneric /
 * DoWritePair (src/writer/SAMWriter.cpp:98-372) produce for the reads of one ngm_mapper_map_* call, records in input order
 * (the two records of a pair: mate 2 first, as the reference writes them).  Single alignments per read (topn <= 1). */
typedef struct ngm_sam_options {
	int paired;                 /* the calls map pairs (reads 2i, 2i + 1) */
	int min_insert_size, max_insert_size;   /* the writer's proper-pair window (max <= 0: unlimited) */
	int min_mq;                 /* Config "min_mq" */
	float min_identity, min_residues;       /* Config "min_identity" (0.65), "min_residues" (0.5; <= 1: share of the read length) */
	int no_unal;                /* Config "no_unal": unmapped reads are not written */
	const char *rg_id;          /* read group id for the RG:Z tag, or NULL */
	int bs_mapping;             /* Config "bs_mapping": the ZS:Z tag (src/writer/SAMWriter.cpp:173-187) */
	int slam_seq;               /* Config SLAM_SEQ != 0: the TC:i / RA:Z / MP:Z tags (src/writer/SAMWriter.cpp:203-221, GenericReadWriter.h:87-186) */
	int bam;                    /* Config "bam": the records as BAM (src/writer/BAMWriter.cpp:147-375), in BGZF blocks written by the GPU -- what
	                             * ngm_mapper_map_sam returns is then a piece of the BAM file (whole BGZF members); not with slam_seq */
} ngm_sam_options;
int ngm_mapper_set_sam_options(ngm_mapper *m, const ngm_sam_options *o);
typedef struct ngm_sam_read {   /* per read: where its name is, how long its quality string is */
	uint32_t name_off;          /* into `names` */
	uint16_t name_len;
	uint16_t qual_len;          /* bytes of the quality string the parser holds (the row holds the first qry_max_len - 1 of them); 0: none ('*');
	                             * bit 15 set: the read has no sequence and is discarded (NGMNames::Empty, GenericReadWriter.h:245-252) */
} ngm_sam_read;
/* Maps the batch like ngm_mapper_map_se / _pe and formats it.  reads, quals: n rows of qry_max_len bytes (page-locked memory
 * makes the copies asynchronous); names: names_bytes bytes; out: out_cap bytes for the text.  Returns the length of the text
 * (> out_cap: nothing was copied -- call ngm_mapper_sam_fetch with a larger buffer), or < 0.  stats: reads counted, reads
 * mapped, lines written.  kernel_ms (optional): GPU time of the formatting kernels. */
long long ngm_mapper_map_sam(ngm_mapper *m, int n, const char *reads, const char *quals, const char *names, size_t names_bytes,
		const ngm_sam_read *meta, char *out, size_t out_cap, uint64_t stats[3], float *kernel_ms);
int ngm_mapper_sam_fetch(ngm_mapper *m, char *out, size_t out_cap);

/* Several mappers on ONE input (ngm-hip hands batches to a mapper per worker thread / per GPU, like NextGenMap hands them to
 * its CS threads, src/NGM.cpp:232-279, src/CS.cpp:440-456).  top1PE's tie-break reads the running mean insert size of the
 * pairs selected so far (ScoreBuffer.h:90, ScoreBuffer.cpp:420-422, :487-488) -- sequential state.  Mappers that share an
 * ngm_pair_state take turns for that part of the selection in batch order (ngm_mapper_set_batch_seq before every
 * ngm_mapper_map_pe*, numbers 0, 1, 2 ... without gaps), so the result equals one mapper seeing the batches in order,
 * i.e. `ngm -t 1`.  Everything else of a batch (search, scoring, the order-free part of the selection, alignment)
 * overlaps freely. */
typedef struct ngm_pair_state ngm_pair_state;
ngm_pair_state *ngm_pair_state_create(void);
void ngm_pair_state_destroy(ngm_pair_state *ps);
int ngm_mapper_set_pair_state(ngm_mapper *m, ngm_pair_state *ps);
int ngm_mapper_set_batch_seq(ngm_mapper *m, uint64_t seq);
/* Config "fast_pairing" (--fast-pairing, src/ScoreBuffer.cpp:203-216): paired-end batches select both mates with top1SE instead of
 * top1PE / CheckPairs; whether the two winners form a pair is decided where the records are written (src/AlignmentBuffer.cpp:176-199) */
int ngm_mapper_set_fast_pairing(ngm_mapper *m, int on);

/* BGZF blocks written by the GPU (csrc/bgzf_device.h): what bamtools' BgzfStream does for `ngm --bam`
 * (lib/bamtools-2.3.0/src/api/internal/io/BgzfStream_p.cpp: DeflateBlock per 64 KB, zlib level 6) -- every 0xFF00 input bytes become one
 * BGZF member (its own DEFLATE stream with dynamic Huffman codes, CRC-32, ISIZE), the members one after the other in `out`.
 * raw / out: host memory (page-locked memory of ngm_host_alloc copies at PCIe rate); out_cap >= ngm_bgzf_bound(n).
 * Returns the bytes written, < 0 on error (ngm_pipeline_last_error).  One call at a time per object; objects are independent. */
typedef struct ngm_bgzf ngm_bgzf;
ngm_bgzf *ngm_bgzf_create(int device);
void ngm_bgzf_destroy(ngm_bgzf *z);
size_t ngm_bgzf_bound(size_t n);
long long ngm_bgzf_compress(ngm_bgzf *z, const void *raw, size_t n, void *out, size_t out_cap);
long long ngm_bgzf_compress_device(ngm_bgzf *z, const void *d_raw, size_t n, void *out, size_t out_cap);   /* raw bytes already in the device's memory */
float ngm_bgzf_last_kernel_ms(const ngm_bgzf *z);   /* HIP-event time of the last call's compression kernel */

/* page-locked host memory for read batches (the H2D copy then runs at PCIe rate without a staging copy) */
void *ngm_host_alloc(size_t bytes);
void ngm_host_free(void *p);

/* Host threads next to the GPU.  A two-socket host runs the host stages (read parsing, pair selection, SAM text) at half the
 * rate when their threads and buffers are spread over both sockets (measured with ngm-hip).  This pins the
 * CALLING thread -- and with it every thread it creates afterwards, the library's thread pool included -- to the CPUs of
 * the NUMA node the device hangs on (/sys/bus/pci/devices/<bdf>/numa_node, .../node<N>/cpulist).  Call it first thing, before
 * any other entry point.  Returns the number of CPUs in the set, 0 when nothing was changed (no NUMA information, or
 * NGM_HIP_NO_NUMA_PIN is set), < 0 on error.  (NextGenMap itself leaves placement to the OS; this replaces nothing there.) */
int ngm_host_pin_to_device_node(int device);

/* A reference artefact that IS mirrored on request: NextGenMap hands a read to its ScoreBuffer right after the search
 * (src/CS.cpp:436); when the last score of a pair's first mate fills the score buffer exactly (src/ScoreBuffer.cpp:519-523; the
 * buffer holds IAlignment::GetScoreBatchSize() entries: 1 024 for the SeqAn personality, src/seqan/EndToEndAffine.h:44-46), DoRun
 * sees the mate's Calculated == -1 (src/MappedRead.cpp:14, src/ScoreBuffer.cpp:196) and selects nothing; if the mate then has no
 * candidates it goes to the writer alone (src/CS.cpp:326-329) and the pair is never written -- "(2 discarded)" in the reference's
 * summary.  With entries > 0 the mapper follows the reference's buffer through its batches (sequential state, like the running mean:
 * exact for `ngm -t 1`) and flags such pairs NGM_PAIR_LOST: no alignment, no record.  0 (default): no pair is lost. */
int ngm_mapper_set_reference_score_buffer(ngm_mapper *m, int entries);
/* pairs flagged NGM_PAIR_LOST by this mapper so far */
int ngm_mapper_lost_pairs(ngm_mapper *m, uint64_t *out);
/* ... the reference flushes that buffer at the end of every CS batch of 1 800 000 / qry_avg_len reads (src/CS.cpp:26, :542-543,
 * even for paired input: src/NGM.cpp:238-243); tell the mapper that number (0: never flushed) so that the walk follows it */
int ngm_mapper_set_reference_cs_batch(ngm_mapper *m, int reads);

/* which paths the reads took, summed over all batches of this mapper: [0] reads searched, [1] candidates, [2] reads re-run by the exact
 * search with its table in LDS, [3] ... in global memory, [4] reads whose candidate order (rList, src/CS.cpp:196-211) was replayed because
 * it decides a tie, [5] of those beyond the limits of the LDS replay (replayed exactly through buckets / a table in global memory), [6] reads whose order was
 * left undetermined (ties then resolve by position), [7] reads searched by the heavy-read kernel (more index hits than the fast path takes) */
int ngm_mapper_path_counters(ngm_mapper *m, uint64_t out[8]);

/* of path counter [5] (reads beyond the LDS replay): the reads the bucket replay (csrc/cs_order_bucket_device.h) left to the replay with a
 * table in global memory (cs_order_kernel<true>) -- bisulfite runs, a read with a bucket of more than 256 hits, NGM_HIP_ORDER_NO_BUCKETS */
int ngm_mapper_order_table_reads(ngm_mapper *m, uint64_t *out);

/* Host-only debug / test entry (no GPU needed): what pass 3 of the pair selection does per tied pair -- ScoreBuffer::top1PE's two sorts
 * (src/ScoreBuffer.cpp:373-376: std::sort(sortLocationScore) on the candidate lists in CollectResultsStd's order, here given by `rank`),
 * computeMQ (:34-49), and the CheckPairs double loop (:405-413, :463-502) over the candidates at or above best * cutoff, restricted to
 * the insert-size window.  Candidates of mate a are entries [0, cnt_a) of loc / sv / score / rank, those of mate b [cnt_a, cnt_a + cnt_b).
 * rank may be NULL (no candidate order: any deterministic order).  Returns 0, or a negative error code. */
int ngm_debug_pair_walk(uint32_t cnt_a, int len_a, uint32_t cnt_b, int len_b, const uint32_t *loc, const uint32_t *sv, const float *score, const uint32_t *rank,
		float cutoff, int min_insert, int max_insert, uint32_t *out_a, uint32_t *out_b, int *mq_a, int *mq_b, uint64_t cap, float *combo_score, int *combo_dist, int *combo_a, int *combo_b,
		uint64_t *n_combo);

/* Host-only debug / test entry: the double loop of ScoreBuffer::top1PE over CheckPairs (src/ScoreBuffer.cpp:405-413, :463-502) at the running
 * mean insert size `avg`, on a given sequence of in-window combinations (pair score, insert size, candidate of a, candidate of b) -- on all
 * of them (out_all) and on the ones the product keeps for its sequential pass (out_kept: those that reach the running maximum of the pair
 * score).  out[6] = {found, winner of a, winner of b, equal-score-and-insert-size count, insert size, combinations evaluated}. */
int ngm_debug_pair_eval(uint64_t n, const float *pair_score, const int *dist, const int *ia, const int *ib, int avg, int out_all[6], int out_kept[6]);

/* test hook: ScoreBuffer::top1SE + computeMQ (src/ScoreBuffer.cpp:228-277, :34-49) as the score stage runs it (select_top1_kernel) over
 * host arrays: read i owns candidates [base[i], base[i] + count[i]); out: winner (candidate index, 0xFFFFFFFF: none), MAPQ, number of
 * best-scoring candidates, best score.  tests/test_gpu_select.py compares it with the reference's sequential loop. */
int ngm_debug_select_top1(int device, int n_reads, const uint32_t *base, const uint32_t *count, uint64_t n_cand, const float *scores, const uint32_t *loc,
		const uint32_t *strand_votes, uint32_t *winner, int32_t *mapq, int32_t *n_best, float *best_score);

/* of path counter [7] (reads searched by the heavy-read kernel, csrc/cs_heavy_device.h), summed over all batches: [0] reads given a second
 * pass (T from the first pass's maximum), [1] table passes started over with twice the parts, [2] reads a class could not certify and
 * queued again, [3] times the pool of global-memory vote tables had to grow (one more synchronisation in that batch) */
int ngm_mapper_heavy_counters(ngm_mapper *m, uint64_t out[4]);

/* work counters of the last candidate search: [0] k-mers looked up, [1] index hits voted, [2] candidates emitted
 * (algorithmic bytes of the search = 20 * kmers + 4 * hits + 16 * candidates) */
int ngm_mapper_cs_counters(ngm_mapper *m, uint64_t out[3]);

/* ---- the one collective of the path: mapping statistics summed over the shard processes --------------------------
 * Reads shard across GPUs with nothing shared on the data path (`ngm-hip -g a,b,... --shard-output`: one process per GPU, the genome and
 * the index replicated); what the reference keeps in process-wide counters that all its CS threads add to (src/NGM.cpp:172-200
 * AddMappedRead / AddUnmappedRead / AddWrittenRead / AddReadRead; the pair counters of src/AlignmentBuffer.cpp:175-199) is, across
 * processes, the vector {reads, mapped, unmapped, written, pairs_total, pairs_broken, insert_sum, insert_cnt} summed with ONE
 * ncclAllReduce (RCCL over xGMI; librccl is loaded at run time).  ngm_stats_unique_id: a new communicator id as 256 hex digits + NUL
 * (the parent makes it and hands it to the shard processes); ngm_stats_comm_create: rank `rank` of `world` on `device` joins (about a
 * second: call it beside the load of the index); ngm_stats_allreduce: v becomes the sum over all ranks.  All return < 0 / NULL with
 * ngm_pipeline_last_error() set when librccl is missing or a call fails -- the parent's sum over the shards' pipes stands then. */
typedef struct ngm_stats_comm ngm_stats_comm;
int ngm_stats_unique_id(char hex[257]);
ngm_stats_comm *ngm_stats_comm_create(int device, int rank, int world, const char *hex);
int ngm_stats_allreduce(ngm_stats_comm *c, int64_t v[8]);
void ngm_stats_comm_destroy(ngm_stats_comm *c);
/* pairs with both mates mapped, those of them the writer flags as broken (other contig / insert size outside the window / same strand),
 * and the sum of the others' insert sizes, of the last ngm_mapper_map_sam call (src/AlignmentBuffer.cpp:175-199 pairInsertCount,
 * brokenPairs, pairInsertSum) */
int ngm_mapper_last_pair_stats(ngm_mapper *m, uint64_t out[3]);

/* kernel wall-clock of the last ngm_mapper_* call, HIP events on the launch stream, ms:
 * [0] candidate search kernels  [1] gather+pack  [2] score  [3] select  [4] gather+pack (align)  [5] align DP
 * [6] traceback  [7] candidate-search stage including host round trips between its passes */
int ngm_mapper_last_kernel_ms(ngm_mapper *m, float ms[8]);
/* ... and of the candidate-order replays of that call (cs_order_kernel: runs on a stream of its own beside the stages above) */
float ngm_mapper_last_order_replay_ms(ngm_mapper *m);

#ifdef __cplusplus
}
#endif
#endif
