// sam_device.h -- the SAM text of a batch, assembled on the GPU.
//
// Replaces, for the plain-SAM output of `ngm-hip`, the host formatter that mirrors GenericReadWriter::WriteRead / WritePair
// (src/writer/GenericReadWriter.h:190-304: the filters), AlignmentBuffer::WriteRead's proper-pair check
// (src/AlignmentBuffer.cpp:175-199), SAMWriter::DoWriteReadGeneric / DoWriteUnmappedReadGeneric / DoWritePair
// (src/writer/SAMWriter.cpp:98-372).  Everything a record needs is in HBM after the align stage -- the read rows, the CIGAR / MD
// byte stream, the per-read results -- or cheap to upload (names, qualities): formatting 422 bytes per read costs the host
// 2.2 us of CPU time per read on 64 threads (22 of the 20 CPU-seconds of a 10 M read run); the GPU does it in two passes
// over the batch: lengths per unit (a read, or a pair: the two records of a pair are written together, mate 2 first),
// exclusive prefix sum, bytes.  The host only copies the finished text to the output file.
// The host formatter in ngm_cli.cpp stays: BAM output, -n > 1, and as the twin the tests compare this one with.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ngm_pipeline.h"

namespace ngm {

struct SamMeta {            // per read
	uint32_t name_off;      // into the batch's name bytes
	uint16_t name_len;
	uint16_t qual_len;      // bytes of the quality string as the parser holds it (0: none, the record prints '*'); bit 15: the read has no sequence (discarded)
};
struct SamRef { uint32_t cig_off, md_off; uint16_t cig_len, md_len; };   // the read's CIGAR / MD in the byte stream

struct SamArgs {
	uint32_t unit_len_bound;   // (unused since round 5: the 32-bit offsets are checked against the 64-bit sum of the lengths, counters[3])
	int n, q, paired;
	const uint8_t *reads;       // n rows of q bytes
	const uint8_t *quals;       // n rows of q bytes
	const char *names;
	const SamMeta *meta;
	const ngm_hit *hits;
	const SamRef *refs;
	const char *str;            // CIGAR / MD byte stream
	const char *contig_names;   // concatenated
	const uint32_t *contig_name_off;  // [contigs + 1]
	int min_insert, max_insert, min_mq, no_unal, hard_clip, silent_clip;
	float min_identity, min_residues;
	const char *rg;             // read group id (rg_len bytes) or null
	int rg_len;
	int bs_mapping;             // ZS:Z tag
	int slam_seq;               // TC:i / RA:Z / MP:Z tags
	int variant_cpu, alt_scoring;   // how a column was labelled '=' (SwConst::variant, ::alt)
	const uint32_t *genome;     // 4-bit symbol classes, 8 per dword (refindex.h)
	const uint64_t *contig_start;
	uint32_t *unit_len;         // [units] pass 1
	const uint32_t *unit_off;   // [units] exclusive prefix sums
	char *out;
	unsigned long long *counters;  // [0] reads counted [1] reads mapped [2] lines written
	int bam;                    // the records as BAM (binary, uncompressed: BAMWriter.cpp:147-375 over bamtools' BamWriter_p.cpp) instead of SAM text
};

struct SamCountSink {
	uint32_t n = 0;
	__device__ __forceinline__ void put(char) { ++n; }
	__device__ __forceinline__ void put32(uint32_t) { n += 4; }
	__device__ __forceinline__ void bytes(const char *, uint32_t len) { n += len; }
	__device__ __forceinline__ void seq_fwd(const uint8_t *, int len) { n += (uint32_t) len; }
	__device__ __forceinline__ void seq_rc(const uint8_t *, int, int len) { n += (uint32_t) len; }
	__device__ __forceinline__ void rev(const uint8_t *, int, int len) { n += (uint32_t) len; }
};
struct SamWriteSink {
	char *p;
	__device__ __forceinline__ void put(char c) { *p++ = c; }
	__device__ __forceinline__ void put32(uint32_t v) { p[0] = (char) v; p[1] = (char) (v >> 8); p[2] = (char) (v >> 16); p[3] = (char) (v >> 24); p += 4; }
	__device__ __forceinline__ void bytes(const char *s, uint32_t len) { for (uint32_t i = 0; i < len; ++i) p[i] = s[i]; p += len; }
	__device__ __forceinline__ void seq_fwd(const uint8_t *s, int len) { for (int i = 0; i < len; ++i) p[i] = (char) s[i]; p += len; }
	// len characters: complement of s[last], s[last - 1], ...
	__device__ __forceinline__ void seq_rc(const uint8_t *s, int last, int len) {
		for (int t = 0; t < len; ++t) { const char ch = (char) s[last - t]; p[t] = ch == 'A' ? 'T' : ch == 'T' ? 'A' : ch == 'C' ? 'G' : ch == 'G' ? 'C' : ch; }
		p += len;
	}
	__device__ __forceinline__ void rev(const uint8_t *s, int last, int len) { for (int t = 0; t < len; ++t) p[t] = (char) s[last - t]; p += len; }
};

template <typename Sink> __device__ __forceinline__ void sam_u64(Sink &s, unsigned long long v) {
	char b[24];
	int i = 24;
	do { b[--i] = (char) ('0' + (int) (v % 10ull)); v /= 10ull; } while (v);
	for (; i < 24; ++i) s.put(b[i]);
}
template <typename Sink> __device__ __forceinline__ void sam_i64(Sink &s, long long v) {
	if (v < 0) { s.put('-'); sam_u64(s, (unsigned long long) (-(v + 1)) + 1ull); } else sam_u64(s, (unsigned long long) v);
}
template <typename Sink> __device__ __forceinline__ void sam_lit(Sink &s, const char *lit) { for (; *lit; ++lit) s.put(*lit); }
// printf("%g") of roundf(identity * 10000) / 10000 (SAMWriter.cpp:163): at most four decimals, trailing zeros dropped
template <typename Sink> __device__ __forceinline__ void sam_identity(Sink &s, float identity) {
	const float r = roundf(identity * 10000.0f);
	if (!(r >= 0.0f && r <= 10000.0f)) {  // not a ratio of counts: what glibc prints for the non-finite values (never seen from an alignment)
		if (r != r) { if (__float_as_uint(r) >> 31) s.put('-'); sam_lit(s, "nan"); }
		else if (r < 0.0f && r < -3.0e38f) sam_lit(s, "-inf");
		else if (r > 3.0e38f) sam_lit(s, "inf");
		else sam_lit(s, "0");
		return;
	}
	const int iv = (int) r;
	if (iv == 10000) { s.put('1'); return; }
	if (iv == 0) { s.put('0'); return; }
	const char b[6] = {'0', '.', (char) ('0' + iv / 1000), (char) ('0' + iv / 100 % 10), (char) ('0' + iv / 10 % 10), (char) ('0' + iv % 10)};
	int k = 6;
	while (b[k - 1] == '0') --k;
	for (int i = 0; i < k; ++i) s.put(b[i]);
}

struct SamView { int i; const ngm_hit *h; const uint8_t *row, *qual; int L; SamMeta m; };

__device__ __forceinline__ SamView sam_view(const SamArgs &A, int i) {
	SamView v;
	v.i = i; v.h = A.hits + i; v.row = A.reads + (size_t) i * A.q; v.qual = A.quals + (size_t) i * A.q; v.m = A.meta[i];
	int L = 0;
	while (L < A.q && v.row[L] != 0) ++L;
	v.L = L;
	return v;
}
__device__ __forceinline__ bool sam_passes(const SamArgs &A, const SamView &v) {  // GenericReadWriter.h:205-215, :262-273
	float min_res = A.min_residues;
	if (min_res <= 1.0f) min_res = v.L * min_res;
	return v.h->mapped && v.h->mapq >= A.min_mq && v.h->identity >= A.min_identity && (float) (v.L - v.h->qstart - v.h->qend) >= min_res;
}
template <typename Sink> __device__ __forceinline__ void sam_contig(const SamArgs &A, Sink &s, int contig) {
	const uint32_t o = A.contig_name_off[contig];
	s.bytes(A.contig_names + o, A.contig_name_off[contig + 1] - o);
}


// SLAM-seq tags (src/writer/SAMWriter.cpp:203-221 over GenericReadWriter::computeSlaSeqTags, GenericReadWriter.h:87-186, fed by the
// per-column records computeCigarMD leaves in Align::ExtendedData, SWOclCigar.cpp:484-540): every aligned column has a type
// 5 * ref + read (A0 C1 G2 T3 other 4); RA = the 25 type counts, MP = "type:read position:ref position" (1-based, from the start of
// the aligned read / of the alignment) of the columns the kernel labelled 'X', TC = the T>C columns (reverse strand: A>G).
template <typename Sink>
__device__ __forceinline__ void sam_slam_tags(const SamArgs &A, Sink &s, const SamView &v, const SamRef &rf) {
	const ngm_hit &h = *v.h;
	const int L = v.L;
	const uint64_t g0 = A.contig_start[h.contig] + h.pos;
	auto ref_class = [&](int i) -> uint32_t { const uint64_t p = g0 + (uint64_t) i; return (A.genome[p >> 3] >> (4 * (p & 7))) & 15u; };   // A0 C1 G2 T3 x4 N5
	auto read_char = [&](int i) -> char {
		if (!h.reverse) return (char) v.row[i];
		const char ch = (char) v.row[L - 1 - i];
		return ch == 'A' ? 'T' : ch == 'T' ? 'A' : ch == 'C' ? 'G' : ch == 'G' ? 'C' : ch;
	};
	auto read_class = [](char ch) -> uint32_t { return ch == 'A' ? 0u : ch == 'C' ? 1u : ch == 'G' ? 2u : ch == 'T' ? 3u : ch == 'N' ? 5u : 4u; };
	int rates[25];
	for (int i = 0; i < 25; ++i) rates[i] = 0;
	for (int pass = 0; pass < 2; ++pass) {   // pass 0: counts (TC and RA come first in the record), pass 1: the MP list
		if (pass == 1) {
			const int tc = h.reverse ? rates[0 * 5 + 2] : rates[3 * 5 + 1];
			sam_lit(s, "\tTC:i:"); sam_i64(s, tc); sam_lit(s, "\tRA:Z:");
			for (int i = 0; i < 25; ++i) { if (i) s.put(','); sam_i64(s, rates[i]); }
		}
		int read_i = h.qstart, ref_i = 0, num = 0;
		bool any = false;
		for (uint32_t c = 0; c < rf.cig_len; ++c) {
			const char ch = A.str[rf.cig_off + c];
			if (ch >= '0' && ch <= '9') { num = num * 10 + (ch - '0'); continue; }
			if (ch == 'M') {
				for (int k2 = 0; k2 < num; ++k2) {
					const uint32_t fc = ref_class(ref_i + k2);
					const char rch = read_char(read_i + k2);
					const uint32_t rc = read_class(rch);
					const int type = 5 * (int) (fc <= 3u ? fc : 4u) + (int) (rc <= 3u ? rc : 4u);
					if (pass == 0) { rates[type] += 1; continue; }
					// '=' as the kernel labelled it: the __GPU__ build compares the characters, the float4 build the score / the classes
					const char fch = fc == 0u ? 'A' : fc == 1u ? 'C' : fc == 2u ? 'G' : fc == 3u ? 'T' : fc == 4u ? 'x' : 'N';
					const bool eq = !A.variant_cpu ? (rch == fch) : (A.alt_scoring ? rc == fc : (rc <= 3u && rc == fc));
					if (!eq) {
						sam_lit(s, any ? "," : "\tMP:Z:"); any = true;
						sam_i64(s, type); s.put(':'); sam_i64(s, read_i + k2 + 1); s.put(':'); sam_i64(s, ref_i + k2 + 1);
					}
				}
				read_i += num; ref_i += num;
			} else if (ch == 'I') read_i += num;
			else if (ch == 'D') ref_i += num;
			num = 0;   // (S / H: the clipped bases are not columns; the leading ones are h.qstart)
		}
	}
}


// ---- BAM records (`ngm --bam`): BAMWriter::DoWriteReadGeneric / DoWriteUnmappedReadGeneric (src/writer/BAMWriter.cpp:147-375) over bamtools'
// BamWriterPrivate::WriteAlignment (lib/bamtools-2.3.0/src/api/internal/bam/BamWriter_p.cpp:180-300): bin, packed CIGAR, 4-bit bases,
// phred values, typed tags.  The host twin is bam_writer.h's put_record with the tags ngm_cli.cpp adds; tests/test_gpu_bam.py decodes
// the file and compares it with the reference's own --bam output.
__device__ __forceinline__ uint32_t bam_min_bin(int begin, int end) {   // BamWriterPrivate::CalculateMinimumBin (BamWriter_p.cpp:33-41)
	--end;
	if ((begin >> 14) == (end >> 14)) return (uint32_t) (4681 + (begin >> 14));
	if ((begin >> 17) == (end >> 17)) return (uint32_t) (585 + (begin >> 17));
	if ((begin >> 20) == (end >> 20)) return (uint32_t) (73 + (begin >> 20));
	if ((begin >> 23) == (end >> 23)) return (uint32_t) (9 + (begin >> 23));
	if ((begin >> 26) == (end >> 26)) return (uint32_t) (1 + (begin >> 26));
	return 0;
}
__device__ __forceinline__ uint32_t bam_base_code(char c) {   // "=ACMGRSVTWYHKDBN"
	switch (c) {
	case '=': return 0; case 'A': return 1; case 'C': return 2; case 'M': return 3; case 'G': return 4; case 'R': return 5;
	case 'S': return 6; case 'V': return 7; case 'T': return 8; case 'W': return 9; case 'Y': return 10; case 'H': return 11;
	case 'K': return 12; case 'D': return 13; case 'B': return 14; default: return 15;
	}
}
template <typename Sink> __device__ __forceinline__ void bam_tag_int(Sink &s, char a, char b, int v) { s.put(a); s.put(b); s.put('i'); s.put32((uint32_t) v); }
template <typename Sink> __device__ __forceinline__ void bam_tag_str(Sink &s, char a, char b, const char *v, uint32_t len) { s.put(a); s.put(b); s.put('Z'); s.bytes(v, len); s.put('\0'); }
// the operations of a CIGAR text: f(length, operation code); returns the reference span (BAMWriter.cpp:207-215, bam_writer.h put_record)
template <typename F> __device__ __forceinline__ int bam_cigar_ops(const char *cig, uint32_t len, F f) {
	int span = 0;
	uint32_t num = 0;
	for (uint32_t c = 0; c < len; ++c) {
		const char ch = cig[c];
		if (ch >= '0' && ch <= '9') { num = num * 10u + (uint32_t) (ch - '0'); continue; }
		uint32_t op;
		switch (ch) {
		case 'M': op = 0; span += (int) num; break;
		case 'I': op = 1; break;
		case 'D': op = 2; span += (int) num; break;
		case 'N': op = 3; span += (int) num; break;
		case 'S': op = 4; break;
		case 'H': op = 5; break;
		case 'P': op = 6; break;
		case '=': op = 7; span += (int) num; break;
		default: op = 8; span += (int) num; break;
		}
		f(num, op);
		num = 0;
	}
	return span;
}
template <typename Sink>
__device__ __forceinline__ void bam_mapped(const SamArgs &A, Sink &s, const SamView &v, int flags, int mate_ref, long long mate_pos0, long long tlen) {
	const ngm_hit &h = *v.h;
	const int L = v.L;
	const int qlen = v.m.qual_len & 0x7FFF;
	const bool noq = qlen == 0;
	const bool clip = A.hard_clip || A.silent_clip;
	const int s0 = clip ? h.qstart : 0, sl = clip ? L - h.qstart - h.qend : L;
	const int n = max(0, min(sl, 1000));
	const int QL = min(qlen, L);
	const SamRef rf = A.refs[v.i];
	uint32_t n_ops = 0;
	const int span = bam_cigar_ops(A.str + rf.cig_off, rf.cig_len, [&](uint32_t, uint32_t) { if (n_ops < 512u) ++n_ops; });
	const uint32_t l_name = (uint32_t) v.m.name_len + 1u;
	const uint32_t tags = 21u + (A.bs_mapping ? 6u : 0u) + 7u + 21u + (3u + rf.md_len + 1u) + (A.rg_len > 0 ? 3u + (uint32_t) A.rg_len + 1u : 0u);
	s.put32(32u + l_name + 4u * n_ops + (uint32_t) ((n + 1) / 2) + (uint32_t) n + tags);
	s.put32((uint32_t) h.contig);
	s.put32((uint32_t) (int) h.pos);
	s.put32((bam_min_bin((int) h.pos, (int) h.pos + span) << 16) | ((uint32_t) h.mapq << 8) | l_name);
	s.put32(((uint32_t) flags << 16) | n_ops);
	s.put32((uint32_t) n);
	s.put32((uint32_t) mate_ref);
	s.put32((uint32_t) (int) mate_pos0);
	s.put32((uint32_t) (int) tlen);
	s.bytes(A.names + v.m.name_off, v.m.name_len); s.put('\0');
	{
		uint32_t k = 0;
		(void) bam_cigar_ops(A.str + rf.cig_off, rf.cig_len, [&](uint32_t len, uint32_t op) { if (k < 512u) { s.put32((len << 4) | op); ++k; } });
	}
	auto base = [&](int t) -> char {
		if (!h.reverse) return (char) v.row[s0 + t];
		const char ch = (char) v.row[L - 1 - (s0 + t)];
		return ch == 'A' ? 'T' : ch == 'T' ? 'A' : ch == 'C' ? 'G' : ch == 'G' ? 'C' : ch;
	};
	for (int t = 0; t < n; t += 2) s.put((char) ((bam_base_code(base(t)) << 4) | (t + 1 < n ? bam_base_code(base(t + 1)) : 0u)));
	for (int t = 0; t < n; ++t) {
		char qc = ':';
		if (!noq) {
			if (!h.reverse) { if (s0 + t < QL) qc = (char) v.qual[s0 + t]; }
			else { const int at = QL - 1 - (s0 + t); if (at >= 0) qc = (char) v.qual[at]; }
		}
		s.put((char) (qc - 33));
	}
	bam_tag_int(s, 'A', 'S', (int) h.score); bam_tag_int(s, 'N', 'M', h.nm); bam_tag_int(s, 'N', 'H', h.n_best);
	if (A.bs_mapping) {   // BAMWriter.cpp:240-254
		const bool second = A.paired && (flags & 0x80);
		bam_tag_str(s, 'Z', 'S', second ? (h.reverse ? "+-" : "--") : (h.reverse ? "-+" : "++"), 2u);
	}
	s.put('X'); s.put('I'); s.put('f'); s.put32(__float_as_uint(roundf(h.identity * 10000.0f) / 10000.0f));
	bam_tag_int(s, 'X', '0', h.n_best); bam_tag_int(s, 'X', 'E', (int) h.max_votes); bam_tag_int(s, 'X', 'R', L - h.qstart - h.qend);
	bam_tag_str(s, 'M', 'D', A.str + rf.md_off, rf.md_len);
	if (A.rg_len > 0) bam_tag_str(s, 'R', 'G', A.rg, (uint32_t) A.rg_len);
}
template <typename Sink>
__device__ __forceinline__ void bam_unmapped(const SamArgs &A, Sink &s, const SamView &v, int flags, int contig, unsigned long long pos1) {
	const int qlen = v.m.qual_len & 0x7FFF;
	const int n = min(v.L, 1000), QL = min(qlen, v.L);
	const int ref = contig >= 0 ? contig : -1, p0 = contig >= 0 ? (int) pos1 - 1 : -1;
	const uint32_t l_name = (uint32_t) v.m.name_len + 1u;
	const uint32_t tags = A.rg_len > 0 ? 3u + (uint32_t) A.rg_len + 1u : 0u;
	s.put32(32u + l_name + (uint32_t) ((n + 1) / 2) + (uint32_t) n + tags);
	s.put32((uint32_t) ref);
	s.put32((uint32_t) p0);
	s.put32((bam_min_bin(p0, p0) << 16) | l_name);
	s.put32((uint32_t) (flags | 0x4) << 16);
	s.put32((uint32_t) n);
	s.put32((uint32_t) ref);
	s.put32((uint32_t) p0);
	s.put32(0u);
	s.bytes(A.names + v.m.name_off, v.m.name_len); s.put('\0');
	for (int t = 0; t < n; t += 2) s.put((char) ((bam_base_code((char) v.row[t]) << 4) | (t + 1 < n ? bam_base_code((char) v.row[t + 1]) : 0u)));
	for (int t = 0; t < n; ++t) s.put((char) (((qlen != 0 && t < QL) ? (char) v.qual[t] : ':') - 33));
	if (A.rg_len > 0) bam_tag_str(s, 'R', 'G', A.rg, (uint32_t) A.rg_len);
}

// SAMWriter::DoWriteReadGeneric (SAMWriter.cpp:98-228).  rnext: 0 '*', 1 '=', 2 the name of contig rnext_contig; bm_*: what BAMWriter::DoWritePair
// passes on instead (mate reference, 0-based mate position, its own TLEN rule)
template <typename Sink>
__device__ __forceinline__ void sam_mapped(const SamArgs &A, Sink &s, const SamView &v, int flags, int rnext, int rnext_contig, unsigned long long pnext, long long tlen,
		int bm_ref = -1, long long bm_pos0 = -1, long long bm_tlen = 0) {
	const ngm_hit &h = *v.h;
	const int L = v.L;
	if (A.bam) { bam_mapped(A, s, v, flags | (h.reverse ? 0x10 : 0), bm_ref, bm_pos0, bm_tlen); return; }
	const int qlen = v.m.qual_len & 0x7FFF;
	const bool noq = qlen == 0;
	if (h.reverse) flags |= 0x10;
	const bool clip = A.hard_clip || A.silent_clip;
	const int s0 = clip ? h.qstart : 0, sl = clip ? L - h.qstart - h.qend : L;
	s.bytes(A.names + v.m.name_off, v.m.name_len); s.put('\t'); sam_u64(s, (unsigned) flags); s.put('\t');
	sam_contig(A, s, h.contig); s.put('\t'); sam_u64(s, (unsigned long long) h.pos + 1ull); s.put('\t'); sam_i64(s, h.mapq); s.put('\t');
	const SamRef rf = A.refs[v.i];
	s.bytes(A.str + rf.cig_off, rf.cig_len); s.put('\t');
	if (rnext == 0) s.put('*'); else if (rnext == 1) s.put('='); else sam_contig(A, s, rnext_contig);
	s.put('\t'); sam_u64(s, pnext); s.put('\t'); sam_i64(s, tlen); s.put('\t');
	if (sl > 0) { if (!h.reverse) s.seq_fwd(v.row + s0, sl); else s.seq_rc(v.row, L - 1 - s0, sl); }
	s.put('\t');
	if (noq) s.put('*');
	else if (sl > 0) {
		// the quality string as the reference holds it: the first L characters (IParser.h copies qry_max_len - 1 at most)
		const int QL = min(qlen, L);
		const int take = max(0, min(sl, QL - s0));
		if (!h.reverse) s.seq_fwd(v.qual + s0, take); else s.rev(v.qual, QL - 1 - s0, take);
	}
	s.put('\t');
	if (A.rg_len > 0) { sam_lit(s, "RG:Z:"); s.bytes(A.rg, (uint32_t) A.rg_len); s.put('\t'); }
	sam_lit(s, "AS:i:"); sam_i64(s, (int) h.score); sam_lit(s, "\tNM:i:"); sam_i64(s, h.nm); sam_lit(s, "\tNH:i:"); sam_i64(s, h.n_best);
	if (A.bs_mapping) {  // SAMWriter.cpp:173-187
		const bool second = A.paired && (flags & 0x80);
		sam_lit(s, second ? (h.reverse ? "\tZS:Z:+-" : "\tZS:Z:--") : (h.reverse ? "\tZS:Z:-+" : "\tZS:Z:++"));
	}
	sam_lit(s, "\tXI:f:"); sam_identity(s, h.identity);
	sam_lit(s, "\tX0:i:"); sam_i64(s, h.n_best); sam_lit(s, "\tXE:i:"); sam_i64(s, (int) h.max_votes); sam_lit(s, "\tXR:i:"); sam_i64(s, L - h.qstart - h.qend);
	sam_lit(s, "\tMD:Z:"); s.bytes(A.str + rf.md_off, rf.md_len);
	if (A.slam_seq) sam_slam_tags(A, s, v, rf);
	s.put('\n');
}
// SAMWriter::DoWriteUnmappedReadGeneric (SAMWriter.cpp:311-372): contig < 0 prints '*'
template <typename Sink>
__device__ __forceinline__ void sam_unmapped(const SamArgs &A, Sink &s, const SamView &v, int flags, int contig, unsigned long long pos1, char rnext, unsigned long long pnext1) {
	const int qlen = v.m.qual_len & 0x7FFF;
	if (A.bam) { bam_unmapped(A, s, v, flags, contig, pos1); return; }
	s.bytes(A.names + v.m.name_off, v.m.name_len); s.put('\t'); sam_u64(s, (unsigned) (flags | 0x4)); s.put('\t');
	if (contig >= 0) sam_contig(A, s, contig); else s.put('*');
	s.put('\t'); sam_u64(s, pos1); sam_lit(s, "\t0\t*\t"); s.put(rnext); s.put('\t'); sam_u64(s, pnext1); sam_lit(s, "\t0\t");
	s.seq_fwd(v.row, v.L); s.put('\t');
	if (qlen == 0) s.put('*'); else s.seq_fwd(v.qual, min(qlen, v.L));
	if (A.rg_len > 0) { sam_lit(s, "\tRG:Z:"); s.bytes(A.rg, (uint32_t) A.rg_len); }
	s.put('\n');
}

// one unit: read `unit` (single-end) or reads 2 * unit, 2 * unit + 1 (paired).  cnt: reads counted / mapped, lines written; pairs with
// both mates mapped, of those broken, insert size of the others (AlignmentBuffer.cpp:175-199 pairInsertCount / brokenPairs / pairInsertSum)
template <typename Sink>
__device__ __forceinline__ void sam_unit(const SamArgs &A, int unit, Sink &s, uint32_t (&cnt)[6]) {
	if (!A.paired) {
		const SamView v = sam_view(A, unit);
		if (v.m.qual_len & 0x8000u) return;  // NGMNames::Empty reads are discarded (GenericReadWriter.h:245-247)
		++cnt[0];
		if (!sam_passes(A, v)) { if (!A.no_unal) { sam_unmapped(A, s, v, 0, -1, 0, '*', 0); ++cnt[2]; } return; }
		++cnt[1];
		sam_mapped(A, s, v, 0, 0, 0, 0, 0); ++cnt[2];
		return;
	}
	// read1 = the first mate (even ReadId), written second by AlignmentBuffer::WriteRead; read2 = its mate
	const SamView v1 = sam_view(A, 2 * unit), v2 = sam_view(A, 2 * unit + 1);
	if ((v1.m.qual_len & 0x8000u) || (v2.m.qual_len & 0x8000u)) return;  // GenericReadWriter.h:250-252
	const ngm_hit &h1 = *v1.h, &h2 = *v2.h;
	if ((h1.pair_flags | h2.pair_flags) & NGM_PAIR_LOST) return;   // the reference never writes this pair (ngm_mapper_set_reference_score_buffer)
	cnt[0] += 2;
	// AlignmentBuffer::WriteRead (AlignmentBuffer.cpp:175-199): is the pair consistent?
	bool paired_fail = (h1.pair_flags & NGM_PAIR_FAILED) || (h2.pair_flags & NGM_PAIR_FAILED);
	if (h1.mapped && h2.mapped) {
		const long long distance = (h2.pos > h1.pos) ? (long long) (h2.pos - h1.pos) + v1.L : (long long) (h1.pos - h2.pos) + v2.L;
		++cnt[3];
		if (h1.contig != h2.contig || distance < A.min_insert || distance > A.max_insert || h1.reverse == h2.reverse) { paired_fail = true; ++cnt[4]; }
		else cnt[5] += (uint32_t) distance;
	}
	const bool m1 = sam_passes(A, v1), m2 = sam_passes(A, v2);  // GenericReadWriter::WritePair
	cnt[1] += (m1 ? 1u : 0u) + (m2 ? 1u : 0u);
	const int f1 = 0x1 | 0x40, f2 = 0x1 | 0x80;  // SAMWriter::DoWritePair (SAMWriter.cpp:230-310)
	const unsigned long long p1 = h1.pos + 1, p2 = h2.pos + 1;
	const uint32_t unal = A.no_unal ? 0u : 1u;
	if (!m1 && !m2) {
		if (unal) { sam_unmapped(A, s, v2, f2 | 0x8, -1, 0, '*', 0); sam_unmapped(A, s, v1, f1 | 0x8, -1, 0, '*', 0); }
		cnt[2] += 2 * unal;
	} else if (!m1) {
		sam_mapped(A, s, v2, f2 | 0x8, 1, 0, p2, 0, h2.contig, (long long) h2.pos, 0);
		if (unal) sam_unmapped(A, s, v1, f1, h2.contig, p2, '=', p2);
		cnt[2] += 1 + unal;
	} else if (!m2) {
		if (unal) sam_unmapped(A, s, v2, f2, h1.contig, p1, '=', p1);
		sam_mapped(A, s, v1, f1 | 0x8, 1, 0, p1, 0, h1.contig, (long long) h1.pos, 0);
		cnt[2] += 1 + unal;
	} else if (!paired_fail) {
		if (!h1.reverse) {
			const long long d = ((long long) h2.pos + v2.L - h2.qstart - h2.qend) - (long long) h1.pos;
			const long long db = (long long) h2.pos + v2.L - (long long) h1.pos;   // BAMWriter.cpp:428-433: the whole read length
			sam_mapped(A, s, v2, f2 | 0x2, 1, 0, p1, -d, h2.contig, (long long) h1.pos, -db);
			sam_mapped(A, s, v1, f1 | 0x2 | 0x20, 1, 0, p2, d, h2.contig, (long long) h2.pos, db);
			cnt[2] += 2;
		} else if (!h2.reverse) {
			const long long d = ((long long) h1.pos + v1.L - h1.qstart - h1.qend) - (long long) h2.pos;
			const long long db = (long long) h1.pos + v1.L - (long long) h2.pos;
			sam_mapped(A, s, v2, f2 | 0x2 | 0x20, 1, 0, p1, d, h2.contig, (long long) h1.pos, db);
			sam_mapped(A, s, v1, f1 | 0x2, 1, 0, p2, -d, h2.contig, (long long) h2.pos, -db);
			cnt[2] += 2;
		}
	} else {
		sam_mapped(A, s, v2, f2 | (h1.reverse ? 0x20 : 0), 2, h1.contig, p1, 0, h1.contig, (long long) h1.pos, 0);
		sam_mapped(A, s, v1, f1 | (h2.reverse ? 0x20 : 0), 2, h2.contig, p2, 0, h2.contig, (long long) h2.pos, 0);
		cnt[2] += 2;
	}
}

#ifdef NGM_SAM_KERNELS
__global__ __launch_bounds__(256) void sam_lengths_kernel(SamArgs A, int units) {
	__shared__ unsigned long long s_sum;
	if (threadIdx.x == 0) s_sum = 0;
	__syncthreads();
	const int u = blockIdx.x * blockDim.x + threadIdx.x;
	uint32_t len = 0;
	if (u < units) {
		SamCountSink s;
		uint32_t cnt[6] = {0, 0, 0, 0, 0, 0};
		sam_unit(A, u, s, cnt);
		A.unit_len[u] = len = s.n;
	}
	// the batch's offsets are 32-bit prefix sums: their total is checked against this 64-bit sum of the same lengths (a wrap is
	// detected directly, whatever the length of a single record -- MP:Z / RA:Z of a SLAM-seq run can outgrow any per-read estimate)
	if (len) atomicAdd(&s_sum, (unsigned long long) len);
	__syncthreads();
	if (threadIdx.x == 0 && s_sum) atomicAdd(&A.counters[3], s_sum);
}
__global__ __launch_bounds__(256) void sam_write_kernel(SamArgs A, int units) {
	__shared__ unsigned long long s_cnt[6];
	if (threadIdx.x < 6) s_cnt[threadIdx.x] = 0;
	__syncthreads();
	const int u = blockIdx.x * blockDim.x + threadIdx.x;
	uint32_t cnt[6] = {0, 0, 0, 0, 0, 0};
	if (u < units) {
		SamWriteSink s{A.out + A.unit_off[u]};
		sam_unit(A, u, s, cnt);
	}
	for (int k = 0; k < 6; ++k) if (cnt[k]) atomicAdd(&s_cnt[k], (unsigned long long) cnt[k]);
	__syncthreads();
	// counters: [0..2] reads counted / mapped, lines written; [3] the 64-bit sum of the unit lengths (sam_lengths_kernel); [4..6] the pair counters
	if (threadIdx.x < 6 && s_cnt[threadIdx.x]) atomicAdd(&A.counters[threadIdx.x < 3 ? threadIdx.x : threadIdx.x + 1], s_cnt[threadIdx.x]);
}
#endif

}  // namespace ngm
