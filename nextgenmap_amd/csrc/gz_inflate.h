// gz_inflate.h -- a .gz file inflated into memory in one go (round 4).
//
// NextGenMap reads .gz input through zlib's gzFile (src/parser/ReadProvider.cpp:38-41, :240-262: kseq over gzread), one stream per file;
// its pre-mapping estimate needs the first reads of file 1 before a single read can be mapped.  ngm-hip inflates a .gz input ONCE into
// memory and then treats it like a mapped plain file (ngm_cli.cpp, MappedFile) -- so the time from the first input byte to the first
// mapped read IS the inflate, and zlib's inflate() runs at ~350 MB/s of text on this host.  This is a DEFLATE decoder (RFC 1951, gzip
// framing RFC 1952) written for that one use: whole input mapped, whole output in one reserved range, so that
//   * the bit buffer is refilled with one unaligned 8-byte load and no per-byte bounds check (the mapping ends >= 16 bytes behind
//     the hot loop's reads),
//   * a literal/length symbol costs ONE table lookup for every code of up to 11 bits (distance: 8 bits; longer codes take a second
//     lookup in a sub-table), the entry carrying the base value and the number of extra bits,
//   * matches are copied 8 bytes at a time straight from the output itself (the window is the output), over-running by up to 7
//     bytes into space that is written next anyway,
//   * the CRC-32 of the text (the gzip trailer's) is computed by a helper thread that trails the decoder, and the first touch of
//     the output pages is taken by another that runs ahead of it.
// Anything this decoder does not like (a damaged stream, an output beyond the limit) makes inflate_file() return false and the caller
// goes back to zlib's reader, whose error messages the user then sees.
#pragma once

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

namespace ngm {
namespace gz {

constexpr int kLitRoot = 11, kDistRoot = 8, kPreRoot = 7;
constexpr uint32_t kKind = 0x30u, kSym = 0x00u, kLit = 0x10u, kExcept = 0x20u, kSub = 0x30u;   // entry kinds
constexpr int kLitCap = (1 << kLitRoot) + 288 * 16, kDistCap = (1 << kDistRoot) + 32 * 128, kPreCap = 1 << kPreRoot;

// entry: bits 0-3 bits to drop | bits 4-5 kind | the rest by kind:
//   kLit    bits 6-7 = literals - 1, bits 8-31 = up to three literal bytes, first one lowest (root entries whose remaining bits decode
//           to further literals carry those too: a FASTQ text is mostly literals with codes of 2 to 5 bits)
//   kSym    a length or distance symbol: bits 8-12 = extra bits, bits 16-31 = base value
//   kExcept bit 16 clear = end of block, set = not a valid symbol
//   kSub    bits 0-3 = root bits, bits 8-12 = bits of the sub-table, bits 16-31 = its first entry
struct Tables {
	uint32_t lit[kLitCap];
	uint32_t dist[kDistCap];
	uint32_t pre[kPreCap];
};

inline uint32_t reverse_bits(uint32_t v, int n) {
	uint32_t r = 0;
	for (int i = 0; i < n; ++i) { r = (r << 1) | (v & 1u); v >>= 1; }
	return r;
}

// canonical Huffman code (RFC 1951 3.2.2) -> lookup table indexed by the next `root` bits of the stream (LSB first).
// entry_of(symbol) gives the entry without its bit count.  Returns false for an over-subscribed code; an incomplete code is legal only
// where the format allows it (a single distance code): the unused entries then read "not a valid symbol".
template <typename F>
inline bool build_table(const uint8_t *lens, int n, uint32_t *table, int root, int cap, F entry_of, bool code_lengths = false) {
	int count[16] = {0};
	for (int i = 0; i < n; ++i) count[lens[i]]++;
	count[0] = 0;
	uint32_t next_code[17];
	uint32_t code = 0;
	long left = 1;
	for (int l = 1; l <= 15; ++l) {
		left = (left << 1) - count[l];
		if (left < 0) return false;   // over-subscribed
		code = (code + (uint32_t) count[l - 1]) << 1;
		next_code[l] = code;
	}
	// an incomplete code is an error, as for zlib's inflate_table: only a literal/length or distance alphabet of ONE code of one bit
	// (or of no code at all) may leave code space unused
	int longest = 15;
	while (longest >= 1 && count[longest] == 0) --longest;
	if (left > 0 && longest != 0 && (code_lengths || longest != 1)) return false;
	const uint32_t invalid = kExcept | (1u << 16) | 1u;   // (the decoder stops on it)
	const int root_size = 1 << root;
	for (int i = 0; i < root_size; ++i) table[i] = invalid;
	// the longest code under each root prefix decides the size of that prefix's sub-table
	uint8_t sub_bits[1 << kLitRoot];
	bool any_long = false;
	for (int l = root + 1; l <= 15; ++l) any_long = any_long || count[l] != 0;
	if (any_long) {
		memset(sub_bits, 0, (size_t) root_size);
		uint32_t nc[17];
		memcpy(nc, next_code, sizeof(nc));
		for (int s = 0; s < n; ++s) {
			const int l = lens[s];
			if (l == 0) continue;
			const uint32_t c = nc[l]++;
			if (l > root) {
				const uint32_t prefix = reverse_bits(c, l) & (uint32_t) (root_size - 1);
				if (l - root > sub_bits[prefix]) sub_bits[prefix] = (uint8_t) (l - root);
			}
		}
	}
	int used = root_size;
	if (any_long) for (int p = 0; p < root_size; ++p) if (sub_bits[p]) {
		const int size = 1 << sub_bits[p];
		if (used + size > cap) return false;
		table[p] = kSub | ((uint32_t) sub_bits[p] << 8) | (uint32_t) root | ((uint32_t) used << 16);
		for (int i = 0; i < size; ++i) table[used + i] = invalid;
		used += size;
	}
	for (int s = 0; s < n; ++s) {
		const int l = lens[s];
		if (l == 0) continue;
		const uint32_t c = reverse_bits(next_code[l]++, l);
		const uint32_t e = entry_of(s);
		if (l <= root) {
			for (uint32_t i = c; i < (uint32_t) root_size; i += 1u << l) table[i] = e | (uint32_t) l;
		} else {
			const uint32_t ptr = table[c & (uint32_t) (root_size - 1)];
			const int sb = (int) ((ptr >> 8) & 31u);
			uint32_t *sub = table + (ptr >> 16);
			for (uint32_t i = c >> root; i < (1u << sb); i += 1u << (l - root)) sub[i] = e | (uint32_t) (l - root);
		}
	}
	return true;
}

// root entries of the literal/length table: a literal whose code leaves room for one or two more literals in the root bits takes them along
inline void pack_literals(uint32_t *table, int root) {
	const int n = 1 << root;
	uint32_t base[1 << kLitRoot];
	memcpy(base, table, (size_t) n * 4);
	for (int i = 0; i < n; ++i) {
		const uint32_t e1 = base[i];
		if ((e1 & kKind) != kLit) continue;
		const int l1 = (int) (e1 & 15u);
		const uint32_t e2 = base[i >> l1];
		const int l2 = (int) (e2 & 15u);
		if ((e2 & kKind) != kLit || l1 + l2 > root) continue;
		uint32_t e = kLit | (uint32_t) (l1 + l2) | (1u << 6) | (e1 & 0xFF00u) | ((e2 & 0xFF00u) << 8);
		const uint32_t e3 = base[i >> (l1 + l2)];
		const int l3 = (int) (e3 & 15u);
		if ((e3 & kKind) == kLit && l1 + l2 + l3 <= root) e = kLit | (uint32_t) (l1 + l2 + l3) | (2u << 6) | (e1 & 0xFF00u) | ((e2 & 0xFF00u) << 8) | ((e3 & 0xFF00u) << 16);
		table[i] = e;
	}
}

inline uint32_t litlen_entry(int s) {
	static const uint16_t base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
	static const uint8_t extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
	if (s < 256) return kLit | ((uint32_t) s << 8);
	if (s == 256) return kExcept;
	if (s > 285) return kExcept | (1u << 16);
	return ((uint32_t) base[s - 257] << 16) | ((uint32_t) extra[s - 257] << 8);
}
inline uint32_t dist_entry(int s) {
	static const uint16_t base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
	static const uint8_t extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
	if (s > 29) return kExcept | (1u << 16);
	return ((uint32_t) base[s] << 16) | ((uint32_t) extra[s] << 8);
}

struct BitReader {
	const uint8_t *in, *in_end;
	uint64_t buf = 0;
	int cnt = 0;   // valid bits in buf (above them: zeros, or bits of the bytes at `in` -- the same bits a later refill ORs in)
	// >= 56 bits when 8 bytes can be read at `in` (one unaligned load; `in` advances by the whole bytes that now count as buffered)
	inline void refill_fast() {
		uint64_t w;
		memcpy(&w, in, 8);
		buf |= w << cnt;
		in += (63 - cnt) >> 3;
		cnt |= 56;
	}
	inline void refill_safe() {
		if (in_end - in >= 8) { refill_fast(); return; }
		while (cnt <= 56 && in < in_end) { buf |= (uint64_t) *in++ << cnt; cnt += 8; }
	}
	inline uint32_t peek(int n) const { return (uint32_t) (buf & ((1ull << n) - 1ull)); }
	inline void drop(int n) { buf >>= n; cnt -= n; }
	inline uint32_t take(int n) { const uint32_t v = peek(n); drop(n); return v; }
};

// one DEFLATE stream from r.in; text appended at out (the window is [out_begin, out)); returns the new end of the text, nullptr on error
inline uint8_t *inflate_stream(BitReader &r, Tables &T, uint8_t *out_begin, uint8_t *out, uint8_t *out_limit, std::atomic<size_t> *progress, const uint8_t *progress_base) {
	static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
	int fixed_built = 0;
	for (;;) {
		r.refill_safe();
		if (r.cnt < 3) return nullptr;
		const uint32_t final_block = r.take(1), type = r.take(2);
		if (type == 0) {
			// stored: back to a byte boundary; the bytes still in the bit buffer go back to the input
			r.drop(r.cnt & 7);
			r.in -= r.cnt >> 3;
			r.buf = 0; r.cnt = 0;
			if (r.in_end - r.in < 4) return nullptr;
			const uint32_t len = (uint32_t) r.in[0] | ((uint32_t) r.in[1] << 8), nlen = (uint32_t) r.in[2] | ((uint32_t) r.in[3] << 8);
			r.in += 4;
			if ((len ^ nlen) != 0xFFFFu || (size_t) (r.in_end - r.in) < len || (size_t) (out_limit - out) < len) return nullptr;
			memcpy(out, r.in, len);
			out += len; r.in += len;
		} else if (type == 1 || type == 2) {
			if (type == 1) {
				if (fixed_built != 1) {
					uint8_t l[288 + 32];
					for (int i = 0; i < 144; ++i) l[i] = 8;
					for (int i = 144; i < 256; ++i) l[i] = 9;
					for (int i = 256; i < 280; ++i) l[i] = 7;
					for (int i = 280; i < 288; ++i) l[i] = 8;
					for (int i = 0; i < 32; ++i) l[288 + i] = 5;
					if (!build_table(l, 288, T.lit, kLitRoot, kLitCap, litlen_entry) || !build_table(l + 288, 32, T.dist, kDistRoot, kDistCap, dist_entry)) return nullptr;
					pack_literals(T.lit, kLitRoot);
					fixed_built = 1;
				}
			} else {
				fixed_built = 0;
				r.refill_safe();
				if (r.cnt < 14) return nullptr;
				const int hlit = (int) r.take(5) + 257, hdist = (int) r.take(5) + 1, hclen = (int) r.take(4) + 4;
				if (hlit > 286 || hdist > 30) return nullptr;
				uint8_t pl[19] = {0};
				for (int i = 0; i < hclen; ++i) {
					if (r.cnt < 3) { r.refill_safe(); if (r.cnt < 3) return nullptr; }
					pl[order[i]] = (uint8_t) r.take(3);
				}
				if (!build_table(pl, 19, T.pre, kPreRoot, kPreCap, [](int s) { return (uint32_t) s << 16; }, true)) return nullptr;
				uint8_t l[286 + 30 + 140];
				int i = 0;
				const int total = hlit + hdist;
				while (i < total) {
					if (r.cnt < 16) { r.refill_safe(); }
					const uint32_t e = T.pre[r.peek(kPreRoot)];
					if ((e & kKind) == kExcept) return nullptr;
					if ((int) (e & 15u) > r.cnt) return nullptr;
					r.drop((int) (e & 15u));
					const int s = (int) (e >> 16);
					if (s < 16) { l[i++] = (uint8_t) s; continue; }
					int rep;
					uint8_t v = 0;
					if (s == 16) { if (i == 0 || r.cnt < 2) return nullptr; v = l[i - 1]; rep = 3 + (int) r.take(2); }
					else if (s == 17) { if (r.cnt < 3) return nullptr; rep = 3 + (int) r.take(3); }
					else { if (r.cnt < 7) return nullptr; rep = 11 + (int) r.take(7); }
					if (i + rep > total) return nullptr;
					memset(l + i, v, (size_t) rep);
					i += rep;
				}
				if (l[256] == 0) return nullptr;   // no end-of-block code
				if (!build_table(l, hlit, T.lit, kLitRoot, kLitCap, litlen_entry)) return nullptr;
				pack_literals(T.lit, kLitRoot);
				// (one distance code of one bit, or none at all -- a block of literals only -- is an incomplete code the format allows)
				if (!build_table(l + hlit, hdist, T.dist, kDistRoot, kDistCap, dist_entry)) return nullptr;
			}
			// ---- the symbols of the block ----
			const uint32_t lmask = (1u << kLitRoot) - 1u, dmask = (1u << kDistRoot) - 1u;
			bool end_of_block = false;
			while (!end_of_block) {
				// hot loop: 16 input bytes and 320 output bytes of room -- no checks inside
				if (r.in_end - r.in >= 16 && out_limit - out >= 320) {
					for (;;) {
						r.refill_fast();   // >= 56 bits
						uint32_t e = T.lit[r.buf & lmask];
						if ((e & kKind) == kSub) { r.drop(kLitRoot); e = T.lit[(e >> 16) + r.peek((int) ((e >> 8) & 31u))]; }
						r.drop((int) (e & 15u));
						if (e & kLit) {   // (kSub is resolved: bit 4 alone says literal)
							uint32_t v = e >> 8;
							memcpy(out, &v, 4);
							out += 1u + ((e >> 6) & 3u);
							// two more root entries on the bits that are left (>= 56 - 15 - 11 for the second)
							e = T.lit[r.buf & lmask];
							if ((e & kKind) == kLit) {
								r.drop((int) (e & 15u));
								v = e >> 8;
								memcpy(out, &v, 4);
								out += 1u + ((e >> 6) & 3u);
								e = T.lit[r.buf & lmask];
								if ((e & kKind) == kLit) {
									r.drop((int) (e & 15u));
									v = e >> 8;
									memcpy(out, &v, 4);
									out += 1u + ((e >> 6) & 3u);
								}
							}
							if (r.in_end - r.in < 16 || out_limit - out < 320) break;
							continue;
						}
						if (e & kExcept) { if (e >> 16) return nullptr; end_of_block = true; break; }
						// a match: <= 5 extra bits of the length, then <= 15 + 13 bits of distance -- 41 bits are left at least
						const uint32_t len = (e >> 16) + r.take((int) ((e >> 8) & 31u));
						uint32_t d = T.dist[r.buf & dmask];
						if ((d & kKind) == kSub) { r.drop(kDistRoot); d = T.dist[(d >> 16) + r.peek((int) ((d >> 8) & 31u))]; }
						if (d & kExcept) return nullptr;
						r.drop((int) (d & 15u));
						const uint32_t dist = (d >> 16) + r.take((int) ((d >> 8) & 31u));
						if ((size_t) dist > (size_t) (out - out_begin)) return nullptr;
						const uint8_t *src = out - dist;
						uint8_t *const end = out + len;
						if (dist >= 8) {
							do { uint64_t w; memcpy(&w, src, 8); memcpy(out, &w, 8); out += 8; src += 8; } while (out < end);
						} else if (dist == 1) {
							memset(out, *src, len);
						} else {
							do { *out++ = *src++; } while (out < end);
						}
						out = end;
						if (r.in_end - r.in < 16 || out_limit - out < 320) break;
					}
					continue;
				}
				// careful path (the last bytes of the input, the last bytes of the output range)
				r.refill_safe();
				uint32_t e = T.lit[r.buf & lmask];
				if ((e & kKind) == kSub) { r.drop(kLitRoot); e = T.lit[(e >> 16) + r.peek((int) ((e >> 8) & 31u))]; }
				r.drop((int) (e & 15u));
				if (r.cnt < 0) return nullptr;
				if (e & kLit) {
					const uint32_t nl = 1u + ((e >> 6) & 3u);
					if ((size_t) (out_limit - out) < nl) return nullptr;
					for (uint32_t i = 0; i < nl; ++i) *out++ = (uint8_t) (e >> (8 + 8 * i));
					continue;
				}
				if (e & kExcept) { if (e >> 16) return nullptr; end_of_block = true; break; }
				const uint32_t len = (e >> 16) + r.take((int) ((e >> 8) & 31u));
				if (r.cnt < 0) return nullptr;
				r.refill_safe();
				uint32_t d = T.dist[r.buf & dmask];
				if ((d & kKind) == kSub) { r.drop(kDistRoot); d = T.dist[(d >> 16) + r.peek((int) ((d >> 8) & 31u))]; }
				if (d & kExcept) return nullptr;
				r.drop((int) (d & 15u));
				const uint32_t dist = (d >> 16) + r.take((int) ((d >> 8) & 31u));
				if (r.cnt < 0) return nullptr;
				if ((size_t) dist > (size_t) (out - out_begin) || (size_t) (out_limit - out) < len) return nullptr;
				const uint8_t *src = out - dist;
				for (uint32_t i = 0; i < len; ++i) out[i] = src[i];
				out += len;
			}
		} else {
			return nullptr;
		}
		if (progress) progress->store((size_t) (out - progress_base), std::memory_order_release);
		if (final_block) return out;
	}
}

// The text of a .gz file (every member, as gzread does).  *text is a private anonymous mapping of *reserved bytes (munmap it);
// limit: the most text taken.  check_crc: the trailer's CRC-32 and length of every member are verified (a helper thread, trailing).
inline bool inflate_file(const char *path, char **text, size_t *len, size_t *reserved, size_t limit, bool check_crc = true) {
	const int fd = ::open(path, O_RDONLY);
	if (fd < 0) return false;
	struct stat st;
	if (fstat(fd, &st) != 0 || st.st_size < 18) { close(fd); return false; }
	const size_t zn = (size_t) st.st_size;
	void *zm = mmap(nullptr, zn, PROT_READ, MAP_PRIVATE, fd, 0);
	close(fd);
	if (zm == MAP_FAILED) return false;
	madvise(zm, zn, MADV_SEQUENTIAL);
	// DEFLATE cannot expand beyond 1032 : 1; address space is reserved, memory is only what the text touches
	const size_t want = std::min(limit, zn > ((size_t) 1 << 40) ? limit : zn * 1032 + 4096);
	const size_t res = ((want + 320 + 4095) & ~(size_t) 4095) + ((size_t) 2 << 20);
	void *om = mmap(nullptr, res, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
	if (om == MAP_FAILED) { munmap(zm, zn); return false; }
	// (no MADV_HUGEPAGE: with it the first touch of a 2 MB range compacts and clears it under the decoder -- 1.2 to 3.6 s for the same
	// 321 MB file from run to run on the development host, 1.1 to 1.3 s without)
	uint8_t *const out_begin = (uint8_t *) om, *const out_limit = out_begin + want;
	const uint8_t *z = (const uint8_t *) zm, *const z_end = z + zn;

	struct Member { size_t start, end; uint32_t crc; };
	std::mutex mu;
	std::vector<Member> members;
	std::atomic<size_t> produced{0};
	std::atomic<int> state{0};   // 1: all members listed, 2: stop
	std::atomic<bool> crc_bad{false};
	// first touch of the output pages: a thread that stays up to 64 MB ahead of the decoder takes the page faults off its path.  The
	// pages are populated by the kernel (MADV_POPULATE_WRITE, Linux 5.14): no store of this thread ever lands in memory the decoder
	// writes.  Where the kernel lacks it, the decoder takes its own faults.
	std::thread toucher([&] {
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
		size_t touched = 0;
		while (state.load(std::memory_order_acquire) == 0) {
			const size_t target = std::min(want, produced.load(std::memory_order_acquire) + ((size_t) 64 << 20)) & ~(size_t) 4095;
			if (touched >= target) { std::this_thread::sleep_for(std::chrono::microseconds(50)); continue; }
			const size_t stop = std::min(target, touched + ((size_t) 4 << 20));
			if (madvise(out_begin + touched, stop - touched, MADV_POPULATE_WRITE) != 0) return;
			touched = stop;
		}
	});
	std::thread helper;
	if (check_crc) helper = std::thread([&] {
		size_t pos = 0, mi = 0;
		uLong c = crc32(0L, Z_NULL, 0);
		for (;;) {
			const int s = state.load(std::memory_order_acquire);
			const size_t lim = produced.load(std::memory_order_acquire);
			Member m{0, 0, 0};
			bool have = false;
			{ std::lock_guard<std::mutex> g(mu); if (mi < members.size()) { m = members[mi]; have = true; } }
			const size_t stop = have ? m.end : lim;
			if (pos < stop) {
				const size_t chunk = std::min<size_t>(stop - pos, (size_t) 4 << 20);
				c = crc32(c, out_begin + pos, (uInt) chunk);
				pos += chunk;
				continue;
			}
			if (have) { if ((uint32_t) c != m.crc) crc_bad = true; c = crc32(0L, Z_NULL, 0); ++mi; continue; }
			if (s != 0) break;
			std::this_thread::sleep_for(std::chrono::microseconds(100));
		}
	});
	auto *T = new Tables;
	bool ok = true;
	uint8_t *out = out_begin;
	bool any = false;
	while (ok && z < z_end) {
		// member header (RFC 1952); zero bytes or garbage after the last member end the file as they do for gzread
		if (z_end - z < 18 || z[0] != 0x1f || z[1] != 0x8b) { if (!any) ok = false; break; }
		if (z[2] != 8 || (z[3] & 0xE0)) { ok = false; break; }
		const uint8_t flg = z[3];
		const uint8_t *p = z + 10;
		if (flg & 4) { if (z_end - p < 2) { ok = false; break; } const size_t xl = (size_t) p[0] | ((size_t) p[1] << 8); p += 2; if ((size_t) (z_end - p) < xl) { ok = false; break; } p += xl; }
		if (flg & 8) { while (p < z_end && *p) ++p; if (p >= z_end) { ok = false; break; } ++p; }
		if (flg & 16) { while (p < z_end && *p) ++p; if (p >= z_end) { ok = false; break; } ++p; }
		if (flg & 2) { if (z_end - p < 2) { ok = false; break; } p += 2; }
		BitReader r;
		r.in = p; r.in_end = z_end;
		uint8_t *const member_begin = out;
		// (the window of a member starts with the member: a distance must not reach into the previous member's text)
		uint8_t *e = inflate_stream(r, *T, member_begin, out, out_limit, &produced, out_begin);
		if (!e) { ok = false; break; }
		out = e;
		// trailer: the bytes still in the bit buffer go back first
		r.drop(r.cnt & 7);
		const uint8_t *t = r.in - (r.cnt >> 3);
		if (z_end - t < 8) { ok = false; break; }
		const uint32_t crc = (uint32_t) t[0] | ((uint32_t) t[1] << 8) | ((uint32_t) t[2] << 16) | ((uint32_t) t[3] << 24);
		const uint32_t isize = (uint32_t) t[4] | ((uint32_t) t[5] << 8) | ((uint32_t) t[6] << 16) | ((uint32_t) t[7] << 24);
		if (isize != (uint32_t) (size_t) (out - member_begin)) { ok = false; break; }
		if (check_crc) { std::lock_guard<std::mutex> g(mu); members.push_back(Member{(size_t) (member_begin - out_begin), (size_t) (out - out_begin), crc}); }
		z = t + 8;
		any = true;
	}
	delete T;
	state.store(ok ? 1 : 2, std::memory_order_release);
	if (helper.joinable()) helper.join();
	toucher.join();
	munmap(zm, zn);
	if (!ok || crc_bad.load() || out == out_begin) { munmap(om, res); return false; }
	*text = (char *) om; *len = (size_t) (out - out_begin); *reserved = res;
	return true;
}

}  // namespace gz
}  // namespace ngm
