/*
 * ngm_oracle.h -- CPU restatement of NextGenMap's score/align hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (nextgenmap_amd/, include/)
 * may include, link or call this.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, and only as the checker.
 *
 * Every function cites the reference file:line it restates (paths relative to
 * /root/reference).  Pinning: see oracle/README.md -- the restatement is checked
 * against the reference's own OpenCL kernels compiled (unmodified, by the ROCm
 * OpenCL toolchain) for gfx950 and run on the MI355X (code objects under oracle/_ref), and
 * against golden vectors captured from those runs (tests/golden/).
 */
#ifndef NGM_ORACLE_H
#define NGM_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* Scores as the reference kernels see them (already negated penalties,
 * lib/mason/opencl/SWOcl.cpp:208-213). */
typedef struct ngm_oracle_scoring {
	int match;     /* > 0 */
	int mismatch;  /* < 0 */
	int gap_read;  /* < 0, charged per read base consumed without a ref base (CIGAR I) */
	int gap_ref;   /* < 0, charged per ref base consumed without a read base (CIGAR D) */
	/* -D__ALT_SCORING__ builds (lib/mason/opencl/SWOcl.cpp:225-242): 0 = off (tables "scores"), 1 = bisulfite
	 * (scoresBsFWD / scoresBsREV, oclDefines.cl:94-109), 2 = SLAM-seq (scoresSlamSeqFWD / REV, oclDefines.cl:113-128) */
	int alt;
	int match_alt;     /* -D matchALT    = Config MATCH_BONUS_TT */
	int mismatch_alt;  /* -D mismatchALT = Config MATCH_BONUS_TC (bisulfite) or its negation (SLAM-seq) */
	int dir;           /* the pair's entry of the kernels' `direction` argument (extData, src/ScoreBuffer.cpp:93-110): 0 = FWD table */
} ngm_oracle_scoring;

/* Which build of the reference kernels is being restated where they differ
 * (SURVEY.md Appendix A "variant equivalence"). */
enum {
	NGM_ORACLE_VARIANT_GPU = 0, /* -D__GPU__ : short arithmetic, '='/'X' by char equality */
	NGM_ORACLE_VARIANT_CPU = 1  /* -D__CPU__ : float4 lanes,     '='/'X' by score == match */
};

/* CIGAR op codes handed to the kernels (lib/mason/opencl/SWOclCigar.cpp:35). */
enum {
	NGM_OP_M = 0, NGM_OP_I = 1, NGM_OP_D = 2, NGM_OP_N = 3, NGM_OP_S = 4,
	NGM_OP_H = 5, NGM_OP_P = 6, NGM_OP_EQ = 7, NGM_OP_X = 8, NGM_OP_STOP = 10
};

/* Symbol class 0..6 (A C G T other N NUL). oclDefines.cl:64-80 */
int ngm_oracle_sym_class(unsigned char ch);
/* Substitution score for (read class, ref class). oclDefines.cl:85-91 */
int ngm_oracle_pair_score(const ngm_oracle_scoring *sc, int read_class, int ref_class);

/* BatchScore, local mode: kernel oclSW. oclSwScore.cl:332-377 (GPU) / :111-154 (CPU).
 * ref: q+c bytes readable, qry: q bytes readable (NUL padded). */
int ngm_oracle_score_local(const char *ref, const char *qry, int q, int c,
		const ngm_oracle_scoring *sc, int variant);

/* BatchScore, end-to-end mode: kernel oclSW_Global. oclEndFreeScore.cl:153-203 / :5-55 */
int ngm_oracle_score_endfree(const char *ref, const char *qry, int q, int c,
		const ngm_oracle_scoring *sc, int variant);

/* Raw output of BatchAlign's two kernels for one pair. */
typedef struct ngm_oracle_trace {
	int valid;            /* 0 when the reference skips backtracking (best_read_index <= 0) */
	int best_read_index;  /* pass-1 argmax row   (result[0] before backtracking) */
	int best_ref_index;   /* pass-1 argmax band column (result[1] before backtracking) */
	int ref_position;     /* window offset of the first aligned base (result[0] after) */
	int qstart;           /* result[1] after backtracking */
	int qend;             /* result[2] */
	int alignment_offset; /* result[3]: index of the leading-clip element in rle[] */
	int best_score;       /* pass-1 maximum (not exported by the reference; handy for tests) */
} ngm_oracle_trace;

/* BatchAlign pass 1 + pass 2.  mode 0 = local (oclSW_Score, oclSwScore.cl:219-329 /
 * :4-107), mode 1 = end-to-end (oclSW_ScoreGlobal, oclEndFreeScore.cl:206-326 / :58-146);
 * backtracking oclSwCigar.cl:2-56 / :60-125.
 * rle: caller buffer of 2*(2q+c+1) shorts; elements [alignment_offset, 2q+c] are written
 * exactly as the reference writes them ((len<<4)|op, right-aligned). */
void ngm_oracle_align_trace(int mode, const char *ref, const char *qry, int q, int c,
		const ngm_oracle_scoring *sc, int variant, ngm_oracle_trace *out, short *rle);

/* Result of the host post-processing, mirrors struct Align (include/IAlignment.h:14-29). */
typedef struct ngm_oracle_align {
	int ok;              /* 0: computeCigarMD returned false (Score = -1 in the reference) */
	int position_offset;
	int qstart, qend;
	int nm;
	float identity;
	float score_token;   /* "Score" abused as final read_index, SWOclCigar.cpp:613 */
} ngm_oracle_align;

/* computeCigarMD. lib/mason/opencl/SWOclCigar.cpp:430-615 (bs_mapping / slam_seq off).
 * cigar, md: caller buffers (4*q bytes as the reference allocates, AlignmentBuffer.cpp:106-109). */
void ngm_oracle_cigar_md(const ngm_oracle_trace *tr, const short *rle, const char *ref,
		const char *qry, int q, int c, int hard_clip, int silent_clip,
		ngm_oracle_align *out, char *cigar, char *md);

/* ... with bs_mapping (alt 1) / slam_seq (alt 2) on: a mismatch column whose read base is bsFrom and whose reference base is
 * bsTo counts as a match, every other one as a mismatch (NM) -- SWOclCigar.cpp:300-317 (bsFrom / bsTo from the pair's
 * direction), :496-520. */
void ngm_oracle_cigar_md_alt(const ngm_oracle_trace *tr, const short *rle, const char *ref,
		const char *qry, int q, int c, int hard_clip, int silent_clip, int alt, int dir,
		ngm_oracle_align *out, char *cigar, char *md);

/* Convenience: full BatchAlign for one pair (trace + cigar/md). */
void ngm_oracle_align_pair(int mode, const char *ref, const char *qry, int q, int c,
		const ngm_oracle_scoring *sc, int variant, int hard_clip, int silent_clip,
		ngm_oracle_align *out, char *cigar, char *md);

/* Batch drivers over flat buffers (ref_stride/qry_stride bytes apart); nthreads <= 1 is serial. */
void ngm_oracle_batch_score(int mode, int n, const char *ref, long ref_stride, const char *qry,
		long qry_stride, int q, int c, const ngm_oracle_scoring *sc, int variant,
		float *scores, int nthreads);
void ngm_oracle_batch_align(int mode, int n, const char *ref, long ref_stride, const char *qry,
		long qry_stride, int q, int c, const ngm_oracle_scoring *sc, int variant,
		int hard_clip, int silent_clip, ngm_oracle_align *out, char *cigars, char *mds,
		long str_stride, int nthreads);

/* the same with the per-pair `direction` bytes of the __ALT_SCORING__ builds (dirs == NULL: all 0) */
void ngm_oracle_batch_score_alt(int mode, int n, const char *ref, long ref_stride, const char *qry,
		long qry_stride, int q, int c, const ngm_oracle_scoring *sc, int variant, const char *dirs,
		float *scores, int nthreads);
void ngm_oracle_batch_align_alt(int mode, int n, const char *ref, long ref_stride, const char *qry,
		long qry_stride, int q, int c, const ngm_oracle_scoring *sc, int variant, const char *dirs,
		int hard_clip, int silent_clip, ngm_oracle_align *out, char *cigars, char *mds,
		long str_stride, int nthreads);

/* ---- affine-gap personality (`--affine`): EndToEndAffine over SeqAn 1.4.1, see ngm_affine_oracle.c ---------- */
typedef struct ngm_oracle_affine_scoring {
	int match;       /* Config match_bonus */
	int mismatch;    /* -mismatch_penalty */
	int gap_open;    /* -gap_read_penalty: score of the FIRST gap character (SeqAn convention) */
	int gap_extend;  /* -gap_extend_penalty: every further gap character */
} ngm_oracle_affine_scoring;

/* EndToEndAffine::BatchScore for one pair (src/seqan/EndToEndAffine.cpp:10-28); mode 0 local, 1 end-to-end. */
int ngm_oracle_affine_score(int mode, const char *ref, const char *qry, int q, int c, const ngm_oracle_affine_scoring *sc);
/* EndToEndAffine::BatchAlign + convertToCIGAR (:30-155): CIGAR (S/M/I/D), PositionOffset, QStart, QEnd,
 * NM = mismatching columns, Identity = matches / (columns + gap runs).  MD is not produced by the reference. */
void ngm_oracle_affine_align(int mode, const char *ref, const char *qry, int q, int c, const ngm_oracle_affine_scoring *sc,
		ngm_oracle_align *out, char *cigar);
void ngm_oracle_affine_batch(int mode, int n, const char *ref, long ref_stride, const char *qry, long qry_stride, int q, int c,
		const ngm_oracle_affine_scoring *sc, float *scores, ngm_oracle_align *out, char *cigars, long str_stride, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
