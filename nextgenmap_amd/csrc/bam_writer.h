// bam_writer.h -- BAM records and BGZF blocks for `ngm-hip --bam`.
// Replaces NextGenMap's BAMWriter (src/writer/BAMWriter.cpp:147-460) on top of bamtools 2.3.0's BamWriter
// (lib/bamtools-2.3.0/src/api/internal/bam/BamWriter_p.cpp:33-300: bin, packed CIGAR, 4-bit sequence, phred + tag
// encoding; internal/io/BgzfStream_p.cpp: 64 KB blocks).  Records are appended to a byte string per formatter chunk and
// compressed chunk by chunk -- BGZF blocks are independent deflate streams, so the chunks of a batch compress in
// parallel and concatenate into a valid file.  What the decoded file must equal is the reference's own `--bam` output
// (tests/test_gpu_bam.py decodes both).
#pragma once

#include <zlib.h>

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace ngm {
namespace bam {

inline void put32(std::string &s, uint32_t v) { s.append((const char *) &v, 4); }  // little-endian host (x86-64)

// BamWriterPrivate::CalculateMinimumBin (BamWriter_p.cpp:33-41), signed arithmetic as there (unmapped reads: begin = end = -1)
inline uint32_t min_bin(int begin, int end) {
	--end;
	if ((begin >> 14) == (end >> 14)) return (uint32_t) (4681 + (begin >> 14));
	if ((begin >> 17) == (end >> 17)) return (uint32_t) (585 + (begin >> 17));
	if ((begin >> 20) == (end >> 20)) return (uint32_t) (73 + (begin >> 20));
	if ((begin >> 23) == (end >> 23)) return (uint32_t) (9 + (begin >> 23));
	if ((begin >> 26) == (end >> 26)) return (uint32_t) (1 + (begin >> 26));
	return 0;
}

inline uint8_t base_code(char c) {  // BamWriterPrivate::EncodeQuerySequence: "=ACMGRSVTWYHKDBN"
	switch (c) {
	case '=': return 0; case 'A': return 1; case 'C': return 2; case 'M': return 3; case 'G': return 4; case 'R': return 5;
	case 'S': return 6; case 'V': return 7; case 'T': return 8; case 'W': return 9; case 'Y': return 10; case 'H': return 11;
	case 'K': return 12; case 'D': return 13; case 'B': return 14; default: return 15;
	}
}

// magic, header text, reference dictionary (BamWriterPrivate::Open, BamWriter_p.cpp:155-175, :424-470)
inline void put_header(std::string &raw, const std::string &text, const std::vector<std::string> &names, const std::vector<uint64_t> &lens) {
	raw.append("BAM\1", 4);
	put32(raw, (uint32_t) text.size());
	raw += text;
	put32(raw, (uint32_t) names.size());
	for (size_t i = 0; i < names.size(); ++i) {
		put32(raw, (uint32_t) names[i].size() + 1);
		raw.append(names[i].c_str(), names[i].size() + 1);
		put32(raw, (uint32_t) lens[i]);
	}
}

struct Tags {
	std::string data;
	void add_int(const char *tag, int32_t v) { data.append(tag, 2); data.push_back('i'); data.append((const char *) &v, 4); }  // AddTag(tag, "i", int)
	void add_float(const char *tag, float v) { data.append(tag, 2); data.push_back('f'); data.append((const char *) &v, 4); }
	void add_string(const char *tag, const char *v, size_t n) { data.append(tag, 2); data.push_back('Z'); data.append(v, n); data.push_back('\0'); }
};

// One alignment record (BamWriterPrivate::WriteAlignment).  cigar: SAM text ("12S88M", empty / null for none); seq: the bases as
// written (already reverse-complemented / clipped); qual: ASCII phred+33 of the same length, or null -> ':' like BAMWriter.cpp:223-228
inline void put_record(std::string &raw, const char *name, size_t name_len, uint32_t flag, int ref_id, int pos0, int mapq, const char *cigar,
		const char *seq, size_t seq_len, const char *qual, int mate_ref, int mate_pos0, int tlen, const Tags &tags) {
	uint32_t ops[512];
	int n_ops = 0, ref_span = 0;
	if (cigar) {
		// BAMWriter.cpp:207-215: atoi of the digits in front of every operation character
		for (const char *c = cigar; *c;) {
			uint32_t len = 0;
			while (*c >= '0' && *c <= '9') len = len * 10 + (uint32_t) (*c++ - '0');
			if (!*c) break;
			uint32_t op;
			switch (*c) {
			case 'M': op = 0; ref_span += (int) len; break;
			case 'I': op = 1; break;
			case 'D': op = 2; ref_span += (int) len; break;
			case 'N': op = 3; ref_span += (int) len; break;
			case 'S': op = 4; break;
			case 'H': op = 5; break;
			case 'P': op = 6; break;
			case '=': op = 7; ref_span += (int) len; break;
			default: op = 8; ref_span += (int) len; break;  // 'X'
			}
			if (n_ops < 512) ops[n_ops++] = (len << 4) | op;
			++c;
		}
	}
	const uint32_t l_name = (uint32_t) name_len + 1;
	const uint32_t block = 32 + l_name + 4 * (uint32_t) n_ops + (uint32_t) ((seq_len + 1) / 2) + (uint32_t) seq_len + (uint32_t) tags.data.size();
	put32(raw, block);
	put32(raw, (uint32_t) ref_id);
	put32(raw, (uint32_t) pos0);
	put32(raw, (min_bin(pos0, pos0 + ref_span) << 16) | ((uint32_t) mapq << 8) | l_name);
	put32(raw, (flag << 16) | (uint32_t) n_ops);
	put32(raw, (uint32_t) seq_len);
	put32(raw, (uint32_t) mate_ref);
	put32(raw, (uint32_t) mate_pos0);
	put32(raw, (uint32_t) tlen);
	raw.append(name, name_len);
	raw.push_back('\0');
	raw.append((const char *) ops, 4 * (size_t) n_ops);
	const size_t at = raw.size();
	raw.resize(at + (seq_len + 1) / 2 + seq_len);
	uint8_t *enc = (uint8_t *) &raw[at];
	for (size_t i = 0; i < seq_len; i += 2) enc[i / 2] = (uint8_t) ((base_code(seq[i]) << 4) | (i + 1 < seq_len ? base_code(seq[i + 1]) : 0));
	uint8_t *q = enc + (seq_len + 1) / 2;
	for (size_t i = 0; i < seq_len; ++i) q[i] = (uint8_t) ((qual ? qual[i] : ':') - 33);
	raw += tags.data;
}

// appends `raw` as complete BGZF blocks (<= 0xFF00 input bytes each, zlib level 6 = Z_DEFAULT_COMPRESSION like bamtools).
// One deflate state per thread, reset per block: deflateInit2 / deflateEnd allocate and clear ~260 KB per call, and with every pool
// thread doing that once per 64 KB the allocator, not deflate, set the pace (24 effective threads of 128).
struct BgzfState {
	z_stream zs;
	bool ok = false;
	unsigned char buf[0x10000 + 1024];
	BgzfState() { memset(&zs, 0, sizeof(zs)); ok = deflateInit2(&zs, Z_DEFAULT_COMPRESSION, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) == Z_OK; }
	~BgzfState() { if (ok) deflateEnd(&zs); }
};
inline bool bgzf_compress(const char *raw, size_t n, std::string &out) {
	const size_t kIn = 0xFF00;
	static thread_local BgzfState st;
	if (!st.ok) return false;
	for (size_t at = 0; at < n || (n == 0 && at == 0); at += kIn) {
		const size_t len = n ? std::min(kIn, n - at) : 0;
		z_stream &zs = st.zs;
		if (deflateReset(&zs) != Z_OK) return false;
		zs.next_in = (Bytef *) (raw + at);
		zs.avail_in = (uInt) len;
		zs.next_out = st.buf;
		zs.avail_out = (uInt) sizeof(st.buf);
		const int rc = deflate(&zs, Z_FINISH);
		const size_t clen = zs.total_out;
		if (rc != Z_STREAM_END || clen + 26 > 0x10000) return false;  // (0xFF00 input bytes always fit)
		const uint32_t crc = (uint32_t) crc32(crc32(0L, Z_NULL, 0), (const Bytef *) (raw + at), (uInt) len);
		const uint16_t bsize = (uint16_t) (clen + 25);
		static const unsigned char head[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
		out.append((const char *) head, 16);
		out.append((const char *) &bsize, 2);
		out.append((const char *) st.buf, clen);
		out.append((const char *) &crc, 4);
		const uint32_t isize = (uint32_t) len;
		out.append((const char *) &isize, 4);
		if (n == 0) break;
	}
	return true;
}
inline void bgzf_eof(std::string &out) { (void) bgzf_compress(nullptr, 0, out); }  // the empty block that marks the end of the file

}  // namespace bam
}  // namespace ngm
