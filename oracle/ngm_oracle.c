/*
 * ngm_oracle.c -- CPU restatement of NextGenMap's BatchScore / BatchAlign arithmetic
 * (linear-gap "OpenCL" personality, the default IAlignment of NGM 0.5.5).
 *
 * TEST INFRASTRUCTURE ONLY -- see ngm_oracle.h.  Plain scalar C, one pair at a time,
 * written for clarity, not speed.  All arithmetic is in int; the reference's GPU build
 * uses short and its CPU build uses float holding small integers, all three agree while
 * |values| < 2^15 (q * max(|mismatch|,|gap|) + 16000 < 32768).
 *
 * Notation: read row i (0-based), band column d in [0,c), window index j = i + d.
 */
#include "ngm_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define SHORT_MIN_SENTINEL (-16000) /* oclDefines.cl:28 */

static inline int imax(int a, int b) { return a > b ? a : b; }

/* oclDefines.cl:64-80 -- A/a 0, C/c 1, G/g 2, T/t 3, N/n 5, NUL 6, anything else 4. */
int ngm_oracle_sym_class(unsigned char ch) {
	switch (ch) {
	case 'A': case 'a': return 0;
	case 'C': case 'c': return 1;
	case 'G': case 'g': return 2;
	case 'T': case 't': return 3;
	case 'N': case 'n': return 5;
	case 0: return 6;
	default: return 4;
	}
}

/* oclDefines.cl:85-91, table "scores": rows = read class, columns = ref class; with sc->alt the tables of the
 * __ALT_SCORING__ builds (oclDefines.cl:94-128), which differ from "scores" in two rows each -- and in the read-N row of the
 * REV tables, which is all zero there (oclDefines.cl:108, :127). */
int ngm_oracle_pair_score(const ngm_oracle_scoring *sc, int rc, int fc) {
	if (sc->alt == 1) {          /* bisulfite */
		if (sc->dir == 0) { if (rc == 3 && fc == 1) return sc->mismatch_alt; if (rc == 3 && fc == 3) return sc->match_alt; }  /* read T vs C / T */
		else { if (rc == 0 && fc == 0) return sc->match_alt; if (rc == 0 && fc == 2) return sc->mismatch_alt; if (rc == 5) return 0; }  /* read A vs A / G */
	} else if (sc->alt == 2) {   /* SLAM-seq */
		if (sc->dir == 0) { if (rc == 1 && fc == 3) return sc->mismatch_alt; if (rc == 3 && fc == 3) return sc->match_alt; }  /* read C vs T, read T vs T */
		else { if (rc == 0 && fc == 0) return sc->match_alt; if (rc == 2 && fc == 0) return sc->mismatch_alt; if (rc == 5) return 0; }  /* read A vs A, read G vs A */
	}
	if (rc == 6) return 0;                       /* read NUL: every column 0 */
	if (rc == 5) return fc <= 3 ? 0 : sc->mismatch; /* read N: free vs ACGT, else mismatch */
	if (fc == 6) return 0;                       /* ref NUL vs read ACGT/other: 0 */
	if (rc == 4) return sc->mismatch;            /* read "other" */
	return rc == fc ? sc->match : sc->mismatch;  /* read ACGT */
}

/* oclSW: oclSwScore.cl:332-377 (GPU) / :111-154 (CPU). */
int ngm_oracle_score_local(const char *ref, const char *qry, int q, int c,
		const ngm_oracle_scoring *sc, int variant) {
	int best = -1;
	/* CPU build skips the whole DP for an empty read (oclSwScore.cl:124). */
	if (variant == NGM_ORACLE_VARIANT_CPU && qry[0] == 0) return best;
	int *H = (int *) calloc((size_t) c + 1, sizeof(int)); /* H[c] is the never-written 0 sentinel */
	for (int i = 0; i < q; ++i) { /* all q rows; rows past the read end are NUL rows */
		int rc = ngm_oracle_sym_class((unsigned char) qry[i]);
		int left = 0;
		for (int d = 0; d < c; ++d) {
			int diag = H[d] + ngm_oracle_pair_score(sc, rc, ngm_oracle_sym_class((unsigned char) ref[i + d]));
			left = imax(0, left + sc->gap_ref);
			left = imax(H[d + 1] + sc->gap_read, left);
			left = imax(diag, left);
			best = imax(best, left);
			H[d] = left;
		}
	}
	free(H);
	return best;
}

/* oclSW_Global: oclEndFreeScore.cl:153-203 (GPU) / :5-55 (CPU). */
int ngm_oracle_score_endfree(const char *ref, const char *qry, int q, int c,
		const ngm_oracle_scoring *sc, int variant) {
	int best = SHORT_MIN_SENTINEL;
	if (variant == NGM_ORACLE_VARIANT_CPU && qry[0] == 0) return best; /* oclEndFreeScore.cl:20 */
	int *H = (int *) calloc((size_t) c + 1, sizeof(int));
	H[c] = SHORT_MIN_SENTINEL; /* :170 / :26 */
	for (int i = 0; i < q; ++i) {
		int rc = ngm_oracle_sym_class((unsigned char) qry[i]);
		int left = SHORT_MIN_SENTINEL;
		for (int d = 0; d < c; ++d) {
			int diag = H[d] + ngm_oracle_pair_score(sc, rc, ngm_oracle_sym_class((unsigned char) ref[i + d]));
			left = imax(H[d + 1] + sc->gap_read, left + sc->gap_ref);
			left = imax(diag, left);
			H[d] = left;
		}
	}
	for (int d = 0; d <= c; ++d) best = imax(best, H[d]); /* includes the sentinel slot */
	free(H);
	return best;
}

/*
 * BatchAlign pass 1 (direction matrix + argmax) and pass 2 (backtracking to RLE).
 *   local:      oclSW_Score        oclSwScore.cl:219-329 (GPU) / :4-107 (CPU)
 *   end-to-end: oclSW_ScoreGlobal  oclEndFreeScore.cl:206-326 (GPU) / :58-146 (CPU)
 *   backtrack:  oclSW_Backtracking oclSwCigar.cl:2-56 (GPU) / :60-125 (CPU)
 * Matrix rows have c+2 entries: [0] and [c+1] are borders (STOP in local mode, X in
 * end-to-end mode), row 0 is all STOP.
 */
void ngm_oracle_align_trace(int mode, const char *ref, const char *qry, int q, int c,
		const ngm_oracle_scoring *sc, int variant, ngm_oracle_trace *out, short *rle) {
	const int stride = c + 2;
	const int alignment_length = 2 * q + c + 1;
	const int local = (mode == 0);
	unsigned char *M = (unsigned char *) malloc((size_t) stride * (size_t) (q + 1));
	int *H = (int *) calloc((size_t) c + 1, sizeof(int));
	memset(out, 0, sizeof(*out));

	for (int k = 0; k < stride; ++k) M[k] = NGM_OP_STOP;
	if (!local) H[c] = SHORT_MIN_SENTINEL;

	int best = local ? -1 : SHORT_MIN_SENTINEL;
	int bri = 0, bci = 0;
	int L = 0;
	const unsigned char border = local ? NGM_OP_STOP : NGM_OP_X;
	for (; L < q && qry[L] != 0; ++L) {
		const int i = L;
		const int rc = ngm_oracle_sym_class((unsigned char) qry[i]);
		unsigned char *row = M + (size_t) (i + 1) * stride;
		row[0] = border;
		int left = local ? 0 : SHORT_MIN_SENTINEL;
		for (int d = 0; d < c; ++d) {
			const int s = ngm_oracle_pair_score(sc, rc, ngm_oracle_sym_class((unsigned char) ref[i + d]));
			const int prev = H[d];
			const int diag = prev + s;
			const int up = H[d + 1] + sc->gap_read;
			left += sc->gap_ref;
			int m = local ? imax(0, left) : left;
			m = imax(diag, m);
			m = imax(up, m);
			/* '=' / 'X': __GPU__ by character equality; __CPU__ by score == match, or -- __ALT_SCORING__ -- by class equality
			 * (oclSwScore.cl:65 / :69, oclEndFreeScore.cl alike) */
			const int is_eq = (variant == NGM_ORACLE_VARIANT_CPU) ? (sc->alt ? (rc == ngm_oracle_sym_class((unsigned char) ref[i + d])) : (s == sc->match)) : (qry[i] == ref[i + d]);
			unsigned char ptr;
			if (local && m <= 0) ptr = NGM_OP_STOP;
			else if (m == diag || m == prev + sc->mismatch) ptr = is_eq ? NGM_OP_EQ : NGM_OP_X;
			else if (m == up) ptr = NGM_OP_I;
			else ptr = NGM_OP_D;
			row[d + 1] = ptr;
			if (local && m > best) { best = m; bri = i; bci = d; }
			left = m;
			H[d] = m;
		}
		row[c + 1] = border;
	}
	if (!local) {
		/* argmax over the last row, first strict maximum, sentinel slot included (:308-315 / :135-140) */
		for (int d = 0; d <= c; ++d) if (H[d] > best) { best = H[d]; bri = L - 1; bci = d; }
		/* empty read: CPU build stores read_index-1 = -1 (:144); GPU build zeroes it (:319-321) */
		if (L == 0) bri = (variant == NGM_ORACLE_VARIANT_CPU) ? -1 : 0;
	}
	out->best_read_index = bri;
	out->best_ref_index = bci;
	out->best_score = best;
	/* (an empty read: the float4 build never enters its DP and stores qend = 0, oclSwScore.cl:38, :101-106; the __GPU__ build computes
	 * read_index - best_read_index - 1 = -1, :325) */
	out->qend = local ? ((variant == NGM_ORACLE_VARIANT_CPU && L == 0) ? 0 : (L - bri - 1)) : 0;

	/* pass 2 -- skipped by the reference when best_read_index <= 0, leaving its outputs
	 * undefined; the oracle reports that as valid = 0. */
	if (bri > 0) {
		int row = bri, col = bci;      /* col = band column; matrix entry is col+1 */
		int abs_ref = bci + bri;
		int idx = alignment_length - 1;
		int run_op = NGM_OP_S, run_len = out->qend;
		for (;;) {
			const int ptr = M[(size_t) (row + 1) * stride + (col + 1)];
			if (ptr == NGM_OP_STOP) break;
			if (ptr == NGM_OP_X || ptr == NGM_OP_EQ) { row -= 1; abs_ref -= 1; }
			else if (ptr == NGM_OP_I) { row -= 1; col += 1; }
			else { col -= 1; abs_ref -= 1; }
			if (ptr == run_op) run_len += 1;
			else { rle[idx--] = (short) ((run_len << 4) | run_op); run_op = ptr; run_len = 1; }
		}
		rle[idx--] = (short) ((run_len << 4) | run_op);
		rle[idx] = (short) (((row + 1) << 4) | NGM_OP_S);
		out->valid = 1;
		out->ref_position = abs_ref + 1;
		out->qstart = row + 1;
		out->alignment_offset = idx;
	}
	free(H);
	free(M);
}

/* computeCigarMD: lib/mason/opencl/SWOclCigar.cpp:430-615. */
void ngm_oracle_cigar_md(const ngm_oracle_trace *tr, const short *rle, const char *ref,
		const char *qry, int q, int c, int hard_clip, int silent_clip,
		ngm_oracle_align *out, char *cigar, char *md) {
	ngm_oracle_cigar_md_alt(tr, rle, ref, qry, q, c, hard_clip, silent_clip, 0, 0, out, cigar, md);
}

void ngm_oracle_cigar_md_alt(const ngm_oracle_trace *tr, const short *rle, const char *ref,
		const char *qry, int q, int c, int hard_clip, int silent_clip, int alt, int dir,
		ngm_oracle_align *out, char *cigar, char *md) {
	/* SWOclCigar.cpp:300-317 */
	char bs_from = '0', bs_to = '0';
	if (alt == 1) { if (dir == 1) { bs_from = 'A'; bs_to = 'G'; } else { bs_from = 'T'; bs_to = 'C'; } }
	if (alt == 2) { if (dir == 1) { bs_from = 'G'; bs_to = 'A'; } else { bs_from = 'C'; bs_to = 'T'; } }
	const int alignment_length = 2 * q + c + 1;
	memset(out, 0, sizeof(*out));
	cigar[0] = 0;
	md[0] = 0;
	if (!tr->valid) return;
	const char *refseq = ref + tr->ref_position; /* SWOclCigar.cpp:324 */
	int co = 0, mo = 0;
	const int off = tr->alignment_offset;
	const int lead = rle[off] >> 4;
	if (lead > 0) {
		if (hard_clip == 1) co += sprintf(cigar + co, "%dH", lead);
		else if (silent_clip != 1) co += sprintf(cigar + co, "%dS", lead);
		out->qstart = lead;
	}
	int match = 0, mismatch = 0, total = 0, m_len = 0, md_eq = 0, ref_i = 0, read_i = out->qstart;
	int ok = 1;
	for (int j = off + 1; j < alignment_length - 1 && ok; ++j) {
		const int op = rle[j] & 15, len = rle[j] >> 4;
		total += len;
		switch (op) {
		case NGM_OP_X:
			m_len += len;
			if (!alt) mismatch += len;
			mo += sprintf(md + mo, "%d", md_eq);
			for (int k = 0; k < len; ++k) {
				if (alt) { if (qry[read_i] == bs_from && refseq[ref_i] == bs_to) match += 1; else mismatch += 1; }  /* :507-514 */
				md[mo++] = refseq[ref_i++]; read_i += 1;
			}
			md_eq = 0;
			break;
		case NGM_OP_EQ:
			match += len; m_len += len; md_eq += len; ref_i += len; read_i += len;
			break;
		case NGM_OP_D:
			if (m_len > 0) { co += sprintf(cigar + co, "%dM", m_len); m_len = 0; }
			co += sprintf(cigar + co, "%dD", len);
			mo += sprintf(md + mo, "%d", md_eq);
			md_eq = 0;
			md[mo++] = '^';
			for (int k = 0; k < len; ++k) md[mo++] = refseq[ref_i++];
			mismatch += len;
			break;
		case NGM_OP_I:
			if (m_len > 0) { co += sprintf(cigar + co, "%dM", m_len); m_len = 0; }
			co += sprintf(cigar + co, "%dI", len);
			read_i += len;
			mismatch += len;
			break;
		default:
			ok = 0; /* "This alignment will be discarded" :571-576 */
			break;
		}
	}
	if (!ok) { out->ok = 0; return; }
	mo += sprintf(md + mo, "%d", md_eq);
	if (m_len > 0) co += sprintf(cigar + co, "%dM", m_len);
	const int trail = rle[alignment_length - 1] >> 4;
	if (trail > 0) {
		if (hard_clip == 1) co += sprintf(cigar + co, "%dH", trail);
		else if (silent_clip != 1) co += sprintf(cigar + co, "%dS", trail);
		out->qend = trail;
	}
	cigar[co] = 0;
	md[mo] = 0;
	out->ok = 1;
	out->identity = match * 1.0f / total;
	out->nm = mismatch;
	out->score_token = (float) read_i;
	out->position_offset = tr->ref_position; /* SWOclCigar.cpp:328 */
}

void ngm_oracle_align_pair(int mode, const char *ref, const char *qry, int q, int c,
		const ngm_oracle_scoring *sc, int variant, int hard_clip, int silent_clip,
		ngm_oracle_align *out, char *cigar, char *md) {
	ngm_oracle_trace tr;
	short *rle = (short *) malloc(sizeof(short) * 2 * (size_t) (2 * q + c + 1));
	ngm_oracle_align_trace(mode, ref, qry, q, c, sc, variant, &tr, rle);
	ngm_oracle_cigar_md_alt(&tr, rle, ref, qry, q, c, hard_clip, silent_clip, sc->alt, sc->dir, out, cigar, md);
	free(rle);
}

void ngm_oracle_batch_score(int mode, int n, const char *ref, long ref_stride, const char *qry,
		long qry_stride, int q, int c, const ngm_oracle_scoring *sc, int variant,
		float *scores, int nthreads) {
	(void) nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
#endif
	for (int i = 0; i < n; ++i) {
		const char *r = ref + (long) i * ref_stride, *s = qry + (long) i * qry_stride;
		scores[i] = (float) ((mode & 0xFF) == 0 ? ngm_oracle_score_local(r, s, q, c, sc, variant)
				: ngm_oracle_score_endfree(r, s, q, c, sc, variant));
	}
}

void ngm_oracle_batch_align(int mode, int n, const char *ref, long ref_stride, const char *qry,
		long qry_stride, int q, int c, const ngm_oracle_scoring *sc, int variant,
		int hard_clip, int silent_clip, ngm_oracle_align *out, char *cigars, char *mds,
		long str_stride, int nthreads) {
	(void) nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
#endif
	for (int i = 0; i < n; ++i) {
		ngm_oracle_align_pair(mode & 0xFF, ref + (long) i * ref_stride, qry + (long) i * qry_stride,
				q, c, sc, variant, hard_clip, silent_clip, out + i, cigars + (long) i * str_stride,
				mds + (long) i * str_stride);
	}
}

void ngm_oracle_batch_score_alt(int mode, int n, const char *ref, long ref_stride, const char *qry,
		long qry_stride, int q, int c, const ngm_oracle_scoring *sc, int variant, const char *dirs,
		float *scores, int nthreads) {
	(void) nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
#endif
	for (int i = 0; i < n; ++i) {
		ngm_oracle_scoring s1 = *sc;
		s1.dir = dirs ? (dirs[i] != 0) : 0;
		const char *r = ref + (long) i * ref_stride, *s = qry + (long) i * qry_stride;
		scores[i] = (float) ((mode & 0xFF) == 0 ? ngm_oracle_score_local(r, s, q, c, &s1, variant)
				: ngm_oracle_score_endfree(r, s, q, c, &s1, variant));
	}
}

void ngm_oracle_batch_align_alt(int mode, int n, const char *ref, long ref_stride, const char *qry,
		long qry_stride, int q, int c, const ngm_oracle_scoring *sc, int variant, const char *dirs,
		int hard_clip, int silent_clip, ngm_oracle_align *out, char *cigars, char *mds,
		long str_stride, int nthreads) {
	(void) nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
#endif
	for (int i = 0; i < n; ++i) {
		ngm_oracle_scoring s1 = *sc;
		s1.dir = dirs ? (dirs[i] != 0) : 0;
		ngm_oracle_align_pair(mode & 0xFF, ref + (long) i * ref_stride, qry + (long) i * qry_stride,
				q, c, &s1, variant, hard_clip, silent_clip, out + i, cigars + (long) i * str_stride,
				mds + (long) i * str_stride);
	}
}
