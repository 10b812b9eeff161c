/*
 * ngm_affine_oracle.c -- CPU restatement of NextGenMap's affine-gap IAlignment (`--affine`):
 * EndToEndAffine::BatchScore / BatchAlign (src/seqan/EndToEndAffine.cpp:10-155), i.e. SeqAn 1.4.1's banded
 * Gotoh alignment with band diagonals 0..corridor, single trace, gaps left
 * (lib/seqan-library-1.4.1/include/seqan/align/: dp_formula_affine.h:390-415 recurrence and tie rules,
 * dp_formula.h:152-160 local clamp, dp_scout.h:142-155 first strict maximum in column-major order,
 * dp_algorithm_impl.h:1233-1250 _correctTraceValue, dp_traceback_impl.h:184-470 traceback,
 * dp_setup.h:646 TracebackConfig_<SingleTrace, GapsLeft>).
 *
 * TEST INFRASTRUCTURE ONLY (see ngm_oracle.h).  Pinned against the reference's own code: oracle/_ref/ngm/
 * ngm_affine_ref is EndToEndAffine.cpp + SeqAn compiled from /root/reference behind a small driver
 * (oracle/affine_ref_main.cpp); tests/test_affine_oracle.py diffs the two on seeded pairs.
 *
 * Geometry: H = reference window up to its first NUL (horizontal), V = read (vertical); DP cell (h, v),
 * 0 <= h <= |H|, 0 <= v <= |V|, exists iff 0 <= h - v <= corridor.  Scores are small integers (the reference
 * keeps them in float).
 */
#include "ngm_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

enum { T_NONE = 0, T_DIAG = 1, T_HORI = 2, T_VERT = 4, T_HORI_OPEN = 8, T_VERT_OPEN = 16, T_MAX_H = 32, T_MAX_V = 64 };
#define NEG_INF (-1000000000)

typedef struct { int s, h, v; } cell_t;

/* out_ops: alignment operations from the END of the alignment to its start, one char per column: 'M' 'I' 'D'. */
static int affine_dp(int mode, const char *ref, const char *qry, int q, int c, const ngm_oracle_affine_scoring *sc,
		int *h_end, int *v_end, int *h_beg, int *v_beg, char *ops, int *n_ops, int want_trace) {
	const int lenH = (int) strnlen(ref, (size_t) q + c), lenV = (int) strnlen(qry, (size_t) q);
	const int local = (mode == 0);
	*n_ops = 0; *h_end = *v_end = *h_beg = *v_beg = 0;
	if (lenH < 1 || lenV < 1) return 0;
	const int W = c + 1;  /* band slots per column: slot = v - (h - c) */
	cell_t *col = (cell_t *) malloc(sizeof(cell_t) * (size_t) (W + 2) * 2);
	cell_t *prev = col, *cur = col + (W + 2);
	unsigned char *T = want_trace ? (unsigned char *) calloc((size_t) (lenH + 1) * (size_t) W, 1) : NULL;
#define SLOT(h, v) ((v) - ((h) - c))
	for (int k = 0; k < W + 2; ++k) { prev[k].s = prev[k].h = prev[k].v = NEG_INF; cur[k] = prev[k]; }
	int best = NEG_INF, bh = 0, bv = 0;
	cell_t bestc = {NEG_INF, NEG_INF, NEG_INF};
	for (int h = 0; h <= lenH; ++h) {
		const int v0 = h - c > 0 ? h - c : 0, v1 = h < lenV ? h : lenV;
		for (int k = 0; k < W + 2; ++k) { cur[k].s = cur[k].h = cur[k].v = NEG_INF; }
		for (int v = v0; v <= v1; ++v) {
			cell_t a;
			unsigned char tr = T_NONE;
			if (v == 0) {
				/* initialisation row: free leading reference in both modes (local, or AlignConfig<true,...>) */
				a.s = 0;
				a.h = a.v = local ? 0 : NEG_INF;
			} else if (h == 0) {
				a.s = a.h = a.v = NEG_INF;  /* not reachable with lower diagonal 0 */
			} else {
				const int sub = (ref[h - 1] == qry[v - 1]) ? sc->match : sc->mismatch;
				const cell_t *pd = &prev[SLOT(h - 1, v - 1)];
				const int has_h = (h - v >= 1), has_v = (h - v <= c - 1);
				unsigned char tg = 0, tmax;
				a.h = a.v = NEG_INF;
				if (has_h) {
					const cell_t *ph = &prev[SLOT(h - 1, v)];
					a.h = ph->h + sc->gap_extend;
					const int t = ph->s + sc->gap_open;
					if (a.h < t) { a.h = t; tg |= T_HORI_OPEN; } else tg |= T_HORI;
				}
				if (has_v) {
					const cell_t *pv = &cur[SLOT(h, v - 1)];
					a.v = pv->v + sc->gap_extend;
					const int t = pv->s + sc->gap_open;
					if (a.v < t) { a.v = t; tg |= T_VERT_OPEN; } else tg |= T_VERT;
				}
				if (has_h && has_v) {
					a.s = a.v;
					if (a.s < a.h) { a.s = a.h; tmax = T_MAX_H; } else tmax = T_MAX_V;
				} else if (has_h) { a.s = a.h; tmax = T_MAX_H; }
				else { a.s = a.v; tmax = T_MAX_V; }
				const int d = pd->s + sub;
				if (a.s <= d) { a.s = d; tr = (unsigned char) (T_DIAG | tg); }
				else tr = (unsigned char) (tg | tmax);
				if (local && a.s <= 0) { a.s = a.h = a.v = 0; tr = T_NONE; }
			}
			cur[SLOT(h, v)] = a;
			if (T) T[(size_t) h * W + SLOT(h, v)] = tr;
			const int tracked = local ? 1 : (v == lenV);
			if (tracked && a.s > best) { best = a.s; bh = h; bv = v; bestc = a; }
		}
		cell_t *t = prev; prev = cur; cur = t;
	}
	*h_end = bh; *v_end = bv;
	if (want_trace && T) {
		int h = bh, v = bv;
		unsigned char tv = T[(size_t) h * W + SLOT(h, v)];
		/* _correctTraceValue */
		if (bestc.v == bestc.s) { tv = (unsigned char) ((tv & ~T_DIAG) | T_MAX_V); }
		else if (bestc.h == bestc.s) { tv = (unsigned char) ((tv & ~T_DIAG) | T_MAX_H); }
		/* _retrieveInitialTraceDirection (PreferGapsAtEnd for affine gaps) */
		if (tv & T_MAX_V) tv &= (T_VERT | T_VERT_OPEN | T_MAX_V);
		else if (tv & T_MAX_H) tv &= (T_HORI | T_HORI_OPEN | T_MAX_H);
		int n = 0;
#define TV(h, v) T[(size_t) (h) * W + SLOT(h, v)]
		while (h > 0 && v > 0 && tv != T_NONE) {
			if (tv & T_DIAG) { ops[n++] = 'M'; --h; --v; tv = TV(h, v); }
			else if ((tv & T_MAX_V) && (tv & T_VERT)) {
				while ((!(tv & T_VERT_OPEN) || (tv & T_VERT)) && v != 1) { ops[n++] = 'I'; --v; tv = TV(h, v); }
				ops[n++] = 'I'; --v; tv = TV(h, v);
			} else if ((tv & T_MAX_V) && (tv & T_VERT_OPEN)) { ops[n++] = 'I'; --v; tv = TV(h, v); }
			else if ((tv & T_MAX_H) && (tv & T_HORI)) {
				while ((!(tv & T_HORI_OPEN) || (tv & T_HORI)) && h != 1) { ops[n++] = 'D'; --h; tv = TV(h, v); }
				ops[n++] = 'D'; --h; tv = TV(h, v);
			} else if ((tv & T_MAX_H) && (tv & T_HORI_OPEN)) { ops[n++] = 'D'; --h; tv = TV(h, v); }
			else break;
		}
		*n_ops = n;
		*h_beg = h; *v_beg = v;
	}
	free(col);
	if (T) free(T);
	return best;
}

int ngm_oracle_affine_score(int mode, const char *ref, const char *qry, int q, int c, const ngm_oracle_affine_scoring *sc) {
	int he, ve, hb, vb, n;
	return affine_dp(mode, ref, qry, q, c, sc, &he, &ve, &hb, &vb, NULL, &n, 0);
}

/* EndToEndAffine::BatchAlign + convertToCIGAR (EndToEndAffine.cpp:30-155) for one pair. */
void ngm_oracle_affine_align(int mode, const char *ref, const char *qry, int q, int c, const ngm_oracle_affine_scoring *sc,
		ngm_oracle_align *out, char *cigar) {
	const int lenV = (int) strnlen(qry, (size_t) q);
	char *ops = (char *) malloc((size_t) 2 * (q + c) + 8);
	int he, ve, hb, vb, n;
	memset(out, 0, sizeof(*out));
	cigar[0] = 0;
	const int score = affine_dp(mode, ref, qry, q, c, sc, &he, &ve, &hb, &vb, ops, &n, 1);
	(void) score;
	/* local: the alignment spans pattern [vb, ve), text [hb, he); end-to-end: leading/trailing pattern gaps are
	 * stripped by convertToCIGAR, which leaves the same spans (vb is 0 and ve is |V| there) */
	int co = 0, match = 0, mismatch = 0, total = 0;
	out->qstart = vb;
	out->position_offset = hb;
	if (out->qstart > 0) co += sprintf(cigar + co, "%dS", out->qstart);
	int h = hb, v = vb, pattern_chars = 0;
	for (int k = n - 1; k >= 0;) {
		const char op = ops[k];
		int run = 0;
		while (k >= 0 && ops[k] == op) { ++run; --k; }
		if (op == 'M') {
			for (int t = 0; t < run; ++t) { if (ref[h + t] == qry[v + t]) ++match; else ++mismatch; }
			total += run; h += run; v += run; pattern_chars += run;
			co += sprintf(cigar + co, "%dM", run);
		} else if (op == 'I') {
			co += sprintf(cigar + co, "%dI", run);
			total += 1; v += run; pattern_chars += run;
		} else {
			co += sprintf(cigar + co, "%dD", run);
			total += 1; h += run;
		}
	}
	out->qend = lenV - (pattern_chars + out->qstart);
	if (out->qend > 0) co += sprintf(cigar + co, "%dS", out->qend);
	cigar[co] = 0;
	out->ok = 1;
	out->identity = match * 1.0f / total;
	out->nm = mismatch;
	free(ops);
}

void ngm_oracle_affine_batch(int mode, int n, const char *ref, long ref_stride, const char *qry, long qry_stride, int q, int c,
		const ngm_oracle_affine_scoring *sc, float *scores, ngm_oracle_align *out, char *cigars, long str_stride, int nthreads) {
	(void) nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
#endif
	for (int i = 0; i < n; ++i) {
		const char *r = ref + (long) i * ref_stride, *s = qry + (long) i * qry_stride;
		if (scores) scores[i] = (float) ngm_oracle_affine_score(mode & 0xFF, r, s, q, c, sc);
		if (out) ngm_oracle_affine_align(mode & 0xFF, r, s, q, c, sc, out + i, cigars + (long) i * str_stride);
	}
}
