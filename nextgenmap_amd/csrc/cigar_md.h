// cigar_md.h -- host post-processing of a traceback record into SAM CIGAR / MD / NM / identity.
// Produces what NextGenMap's SWOclCigar::computeCigarMD produces
// (lib/mason/opencl/SWOclCigar.cpp:430-615, bisulfite / SLAM-seq branches excluded), but from the
// compact device runs of align_device.h instead of the reference's (len<<4|op) short array: the
// '=' / 'X' split of a diagonal run is re-derived here from the two sequences.
#pragma once

#include <cstdint>
#include <cstdio>

#include "../../include/ngm_hip.h"

namespace ngm {

struct CigarParams {
	int match;     // > 0
	int mismatch;  // < 0
	int variant;   // NGM_VARIANT_*
	int hard_clip;
	int silent_clip;
	int alt = 0;   // NGM_ALT_*: bs_mapping / slam_seq (SWOclCigar.cpp:300-317, :496-520)
};

// symbol classes of the reference's kernels (oclDefines.cl:64-80)
inline int host_sym_class(unsigned char ch) {
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

// Is a diagonal column labelled '=' ?  The __GPU__ build compares the characters
// (oclSwScore.cl:275), the __CPU__ build tests score == match (oclSwScore.cl:64).
inline bool column_is_eq(const CigarParams &p, unsigned char r, unsigned char f) {
	if (p.variant == NGM_VARIANT_OCL_GPU) return r == f;
	const int rc = host_sym_class(r), fc = host_sym_class(f);
	if (p.alt) return rc == fc;   // the __ALT_SCORING__ build of the float4 kernels compares the classes (oclSwScore.cl:69)
	return rc <= 3 && rc == fc;
}

// "%d" without stdio (this runs once per CIGAR / MD element of every read on the host threads)
inline int put_num(char *dst, int v) {
	char b[12];
	int n = 0, i = 0;
	unsigned u = v < 0 ? 0u - (unsigned) v : (unsigned) v;
	if (v < 0) dst[i++] = '-';
	do { b[n++] = (char) ('0' + u % 10u); u /= 10u; } while (u);
	while (n) dst[i++] = b[--n];
	dst[i] = 0;
	return i;
}
inline int put_op(char *dst, int v, char op) { const int n = put_num(dst, v); dst[n] = op; dst[n + 1] = 0; return n + 1; }

// rec: the 8-int record, runs: rec[4] entries in traceback order.
// dir: the pair's entry of extData (bs_mapping / slam_seq: which conversion counts as a match)
inline void build_cigar_md(const CigarParams &p, const int32_t *rec, const uint16_t *runs, const char *ref,
		const char *qry, ngm_hip_align_out *out, int dir = 0) {
	char bs_from = '0', bs_to = '0';  // SWOclCigar.cpp:300-317
	if (p.alt == NGM_ALT_BISULFITE) { bs_from = dir ? 'A' : 'T'; bs_to = dir ? 'G' : 'C'; }
	if (p.alt == NGM_ALT_SLAMSEQ) { bs_from = dir ? 'G' : 'C'; bs_to = dir ? 'A' : 'T'; }
	out->position_offset = 0;
	out->qstart = 0;
	out->qend = 0;
	out->identity = 0.f;
	out->nm = 0;
	if (!rec[0]) {  // no alignment could be built: the reference reports Score = -1 (SWOclCigar.cpp:322-327)
		out->score_token = -1.0f;
		return;
	}
	char *cigar = out->cigar, *md = out->md;
	int co = 0, mo = 0;
	const int lead = rec[2], trail = rec[3], nruns = rec[4];
	const char *refseq = ref + rec[1];
	if (lead > 0) {
		if (p.hard_clip == 1) co += put_op(cigar + co, lead, 'H');
		else if (p.silent_clip != 1) co += put_op(cigar + co, lead, 'S');
		out->qstart = lead;
	}
	int match = 0, mismatch = 0, total = 0, m_len = 0, md_eq = 0, ref_i = 0, read_i = out->qstart;
	bool in_x_run = false;  // the reference merges adjacent X columns into one element (one MD prefix)
	for (int j = nruns - 1; j >= 0; --j) {
		const int op = runs[j] & 3, len = runs[j] >> 2;
		total += len;
		if (op == 1 || op == 0) {
			for (int k = 0; k < len; ++k) {
				const bool eq = (op == 1) && column_is_eq(p, (unsigned char) qry[read_i], (unsigned char) refseq[ref_i]);
				if (eq) {
					match += 1; md_eq += 1; in_x_run = false;
				} else {
					// bs_mapping / slam_seq: the conversion is a match for identity and NM, not for CIGAR / MD (SWOclCigar.cpp:507-514)
					if (p.alt && qry[read_i] == bs_from && refseq[ref_i] == bs_to) match += 1; else mismatch += 1;
					if (!in_x_run) { mo += put_num(md + mo, md_eq); md_eq = 0; in_x_run = true; }
					md[mo++] = refseq[ref_i];
				}
				m_len += 1; ref_i += 1; read_i += 1;
			}
		} else if (op == 3) {  // deletion: reference bases only
			in_x_run = false;
			if (m_len > 0) { co += put_op(cigar + co, m_len, 'M'); m_len = 0; }
			co += put_op(cigar + co, len, 'D');
			mo += put_num(md + mo, md_eq);
			md_eq = 0;
			md[mo++] = '^';
			for (int k = 0; k < len; ++k) md[mo++] = refseq[ref_i++];
			mismatch += len;
		} else {  // insertion: read bases only
			in_x_run = false;
			if (m_len > 0) { co += put_op(cigar + co, m_len, 'M'); m_len = 0; }
			co += put_op(cigar + co, len, 'I');
			read_i += len;
			mismatch += len;
		}
	}
	mo += put_num(md + mo, md_eq);
	if (m_len > 0) co += put_op(cigar + co, m_len, 'M');
	if (trail > 0) {
		if (p.hard_clip == 1) co += put_op(cigar + co, trail, 'H');
		else if (p.silent_clip != 1) co += put_op(cigar + co, trail, 'S');
		out->qend = trail;
	}
	cigar[co] = 0;
	md[mo] = 0;
	out->identity = match * 1.0f / total;
	out->nm = mismatch;
	out->score_token = (float) read_i;
	out->position_offset = rec[1];
}

// Affine personality: EndToEndAffine::convertToCIGAR (src/seqan/EndToEndAffine.cpp:52-155) from the device runs.
// CIGAR uses S / M / I / D only, Identity = matches / (diagonal columns + number of gap runs), NM = mismatches,
// and neither Align.Score nor pBuffer2 (MD) is written.
// rec[1] = window offset of the first aligned reference base, rec[2] = first aligned read base, runs in traceback order.
// ref / qry may be null: the diagonal columns' match / mismatch counts then come from the device (rec[3], rec[7]) and
// read_len is the read's length.
inline void build_cigar_affine(const int32_t *rec, const uint16_t *runs, const char *ref, const char *qry, int qry_max_len,
		ngm_hip_align_out *out, int read_len = -1) {
	int len_v = 0;
	if (qry) { while (len_v < qry_max_len && qry[len_v]) ++len_v; }
	else len_v = read_len;
	char *cigar = out->cigar;
	int co = 0, match = 0, mismatch = 0, total = 0, pattern_chars = 0;
	int h = rec[1], v = rec[2];
	out->position_offset = h;
	out->qstart = v;
	if (v > 0) { co += put_num(cigar + co, v); cigar[co++] = 'S'; }
	for (int k = rec[4] - 1; k >= 0; --k) {
		const int op = runs[k] & 3, run = runs[k] >> 2;
		co += put_num(cigar + co, run);
		if (op == 1) {
			if (ref && qry) for (int t = 0; t < run; ++t) { if (ref[h + t] == qry[v + t]) ++match; else ++mismatch; }
			total += run; h += run; v += run; pattern_chars += run;
			cigar[co++] = 'M';
		} else if (op == 2) {
			total += 1; v += run; pattern_chars += run;
			cigar[co++] = 'I';
		} else {
			total += 1; h += run;
			cigar[co++] = 'D';
		}
	}
	out->qend = len_v - (pattern_chars + out->qstart);
	if (out->qend > 0) { co += put_num(cigar + co, out->qend); cigar[co++] = 'S'; }
	cigar[co] = 0;
	if (!(ref && qry)) { match = rec[3]; mismatch = rec[7]; }
	out->identity = match * 1.0f / total;
	out->nm = mismatch;
}

}  // namespace ngm
