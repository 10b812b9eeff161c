/*
 * ngm_hip.h -- flat C ABI of the MI355X (gfx950) score/align engine.
 *
 * This is the drop-in boundary for NextGenMap's IAlignment plugin path: the entry points below are
 * what a cgo/JNI/ctypes/C++ binding of that path binds.  Plain pointers and sizes only; no C++ or
 * torch types.  The C++ adapter that presents these as NGM's `class IAlignment` + the plugin exports
 * (SetLog/SetConfig/Cookie/IsAvailable/CreateAlignment/DeleteAlignment/ExternalDeleteString) is
 * declared in ngm_ialignment.h.
 *
 * Reference interfaces replaced (paths relative to the NextGenMap tree):
 *   ngm_hip_create / ngm_hip_destroy   <- CreateAlignment / DeleteAlignment
 *                                         lib/mason/opencl/SWOcl_export.cpp:37-83, src/NGM.cpp:388-437
 *                                         (JIT-time -D constants: lib/mason/opencl/SWOcl.cpp:206-242)
 *   ngm_hip_score_batch_size           <- IAlignment::GetScoreBatchSize  include/IAlignment.h:56,
 *                                         lib/mason/opencl/SWOcl.cpp:384-386
 *   ngm_hip_align_batch_size           <- IAlignment::GetAlignBatchSize  include/IAlignment.h:57
 *   ngm_hip_batch_score                <- IAlignment::BatchScore         include/IAlignment.h:59-63,
 *                                         lib/mason/opencl/SWOcl.cpp:33-162
 *   ngm_hip_batch_align                <- IAlignment::BatchAlign         include/IAlignment.h:65-69,
 *                                         lib/mason/opencl/SWOclCigar.cpp:104-370 (+ computeCigarMD :430-615)
 *   (personality AFFINE: the same two entry points replace EndToEndAffine::BatchScore / BatchAlign,
 *                                         src/seqan/EndToEndAffine.cpp:10-28, :30-155)
 *   ngm_hip_score_device / _align_device : same operations on batches already resident in HBM
 *                                         (no reference counterpart; what bench.py times).
 *
 * Error behaviour mirrors the reference: Batch* return the number of pairs processed (== n on
 * success; the callers only compare with n, src/ScoreBuffer.cpp:131-132); a negative value is a
 * hard failure (-errno style) and ngm_hip_last_error() describes it.  A pair whose alignment cannot
 * be produced gets score_token = -1 (SWOclCigar.cpp:322-327).
 */
#ifndef NGM_HIP_H
#define NGM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NGM_HIP_ABI_VERSION 3

/* mode argument of batch_score / batch_align: include/IAlignment.h:33-48 */
#define NGM_MODE_LOCAL 0      /* Smith-Waterman, kernels oclSW / oclSW_Score        */
#define NGM_MODE_END_TO_END 1 /* read global, reference ends free: oclSW_Global ... */
#define NGM_MODE_ALIGN_MASK 0xFF

/* Which build of the reference's kernels is reproduced where the two differ:
 * the __GPU__ build labels '='/'X' by character equality and scores an empty read 0, the __CPU__
 * (float4) build labels by score == match and scores an empty read -1. */
#define NGM_VARIANT_OCL_GPU 0
#define NGM_VARIANT_OCL_CPU 1

/* Which IAlignment implementation of the reference is reproduced (src/NGM.cpp:388-437):
 * LINEAR = the OpenCL plugin (default; linear gap costs gap_read / gap_ref, CIGAR + MD),
 * AFFINE = `ngm --affine`: src/seqan/EndToEndAffine.cpp over SeqAn 1.4.1's banded Gotoh alignment
 *          (gap open = gap_read_penalty, gap extend = gap_extend_penalty, band diagonals 0..corridor,
 *          CIGAR with S/M/I/D only, NM = mismatches, no MD string: pBuffer2 is left untouched). */
#define NGM_PERSONALITY_LINEAR 0
#define NGM_PERSONALITY_AFFINE 1

typedef struct ngm_hip_params {
	int abi_version;   /* NGM_HIP_ABI_VERSION */
	int qry_max_len;   /* Config "qry_max_len": bytes per read row, read length <= qry_max_len - 1 */
	int corridor;      /* Config "corridor": band columns (2..200); reference window = qry_max_len + corridor bytes */
	/* NGM config values (positive penalties), src/config/Config.cpp:440-444 */
	int match_bonus;
	int mismatch_penalty;
	int gap_read_penalty;
	int gap_ref_penalty;
	int variant;       /* NGM_VARIANT_* */
	int hard_clip;     /* Config "hard_clip"   (SWOclCigar.cpp:450) */
	int silent_clip;   /* Config "silent_clip" (SWOclCigar.cpp:454) */
	int max_batch;     /* largest n the caller will pass (0 = default 1<<20); sizes the HBM workspace */
	int personality;   /* NGM_PERSONALITY_* */
	int gap_extend_penalty; /* Config "gap_extend_penalty" (affine personality only), src/config/Config.cpp:444 */
	/* ABI 3: the strand-specific score tables of `--bs-mapping` / `--slam-seq` (the reference builds its kernels with
	 * -D__ALT_SCORING__, lib/mason/opencl/SWOcl.cpp:225-242, oclDefines.cl:94-128); linear personality only, like the reference
	 * (src/config/Config.cpp:448-460).  The per-pair table choice arrives through the `dir` argument of batch_score / batch_align. */
	int alt_scoring;        /* NGM_ALT_* */
	int match_bonus_tt;     /* Config MATCH_BONUS_TT (-D matchALT) */
	int match_bonus_tc;     /* Config MATCH_BONUS_TC (-D mismatchALT; SLAM-seq: its negation) */
	int alt_cigar;          /* NGM_ALT_*: computeCigarMD counts the strand's conversion as a match for NM / Identity (SWOclCigar.cpp:300-317,
	                         * :496-520): bs_mapping, or ANY slam_seq value -- also one that leaves the score tables alone (slam_seq & 2 == 0) */
} ngm_hip_params;
#define NGM_ALT_NONE 0
#define NGM_ALT_BISULFITE 1   /* Config "bs_mapping" == 1: tables scoresBsFWD / scoresBsREV, bsFrom/bsTo T>C (dir 0), A>G (dir 1) */
#define NGM_ALT_SLAMSEQ 2     /* Config "slam_seq" & 2: tables scoresSlamSeqFWD / REV, C>T (dir 0), G>A (dir 1) */

/* Mirrors struct Align (include/IAlignment.h:14-29); buffers are caller-owned. */
typedef struct ngm_hip_align_out {
	char *cigar;          /* >= 4*qry_max_len bytes, receives NUL-terminated SAM CIGAR (Align.pBuffer1) */
	char *md;             /* >= 4*qry_max_len bytes, receives NUL-terminated MD string  (Align.pBuffer2) */
	int position_offset;  /* window offset of the first aligned reference base */
	int qstart;           /* clipped read bases at the start */
	int qend;             /* clipped read bases at the end */
	float score_token;    /* Align.Score: final read index, or -1 when no alignment could be built */
	float identity;
	int nm;
} ngm_hip_align_out;

typedef struct ngm_hip_ctx ngm_hip_ctx;

/* NULL on failure (see ngm_hip_last_error(NULL)). device = HIP ordinal. */
ngm_hip_ctx *ngm_hip_create(int device, const ngm_hip_params *params);
void ngm_hip_destroy(ngm_hip_ctx *ctx);
const char *ngm_hip_last_error(const ngm_hip_ctx *ctx);
int ngm_hip_device_count(void);

int ngm_hip_score_batch_size(const ngm_hip_ctx *ctx);
int ngm_hip_align_batch_size(const ngm_hip_ctx *ctx);

/* Host-pointer drop-in.  ref[i] -> qry_max_len + corridor readable bytes, qry[i] -> qry_max_len
 * bytes (NUL padded); scores[i] receives the integer score as float.  dir: with alt_scoring, n bytes -- 0 selects the FWD
 * score table for the pair, anything else the REV one (what ScoreBuffer / AlignmentBuffer pass as extData,
 * src/ScoreBuffer.cpp:93-127); NULL = all 0; ignored without alt_scoring. */
int ngm_hip_batch_score(ngm_hip_ctx *ctx, int mode, int n, const char *const *ref, const char *const *qry,
		float *scores, const char *dir);
int ngm_hip_batch_align(ngm_hip_ctx *ctx, int mode, int n, const char *const *ref, const char *const *qry,
		ngm_hip_align_out *out, const char *dir);

/* Device-resident batches: d_ref = n rows of (qry_max_len + corridor) bytes, d_qry = n rows of
 * qry_max_len bytes, both flat, in HBM; d_scores = n floats in HBM.  Work is enqueued on `stream`
 * (a hipStream_t, NULL = the context's own stream) and is asynchronous. */
int ngm_hip_score_device(ngm_hip_ctx *ctx, int mode, int n, const void *d_ref, const void *d_qry,
		float *d_scores, void *stream);

/* alt_scoring with device-resident batches: d_dir = n bytes in HBM (0: FWD table), read by the next *_device call; NULL: all 0. */
int ngm_hip_set_pair_directions(ngm_hip_ctx *ctx, const void *d_dir);

/* Raw per-pair traceback record produced on the device (8 ints):
 *   [0] valid  [1] position_offset  [2] qstart  [3] qend  [4] n_runs  [5] best_score
 *   [6] best_read_index  [7] best_ref_index
 * d_runs: n rows of run_stride uint16, runs in traceback order (last alignment column first),
 * each (len << 2) | op with op 1 = diagonal (match or mismatch), 2 = insertion (read base only),
 * 3 = deletion (reference base only).  run_stride >= ngm_hip_align_run_stride(ctx). */
int ngm_hip_align_run_stride(const ngm_hip_ctx *ctx);
int ngm_hip_align_device(ngm_hip_ctx *ctx, int mode, int n, const void *d_ref, const void *d_qry,
		int32_t *d_records, uint16_t *d_runs, int run_stride, void *stream);

/* Wall-clock of the GPU kernels of the last *_device / batch_* call on this context, measured with
 * HIP events on the launch stream (call after synchronising): [0] pack, [1] DP, [2] traceback, ms. */
int ngm_hip_last_kernel_ms(ngm_hip_ctx *ctx, float ms[3]);
/* Enable/disable the event bracketing above (off by default: no events are recorded). */
void ngm_hip_set_profiling(ngm_hip_ctx *ctx, int enabled);

/* The band width is a compile-time shape of the DP kernels, as in the reference (its OpenCL kernels are JIT-compiled
 * with -D corridor_length, lib/mason/opencl/SWOcl.cpp:206-217).  Corridors 8 12 19 20 27 42 80 are built ahead of
 * time; any other width (<= 200) is compiled by ngm_hip_create with hiprtc, once per process.  This diagnostic compiles
 * the kernels for `corridor` without touching a device: returns the code-object size in bytes, or -1 with the
 * compiler log in msg. */
long ngm_hip_jit_selftest(int corridor, char *msg, int msg_len);

#ifdef __cplusplus
}
#endif
#endif
