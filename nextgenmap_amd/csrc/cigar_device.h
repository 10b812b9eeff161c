// cigar_device.h -- CIGAR / MD / NM / Identity of a batch of alignments on the GPU.
// Device twin of cigar_md.h: SWOclCigar::computeCigarMD (lib/mason/opencl/SWOclCigar.cpp:430-615) for the default
// personality, EndToEndAffine::convertToCIGAR (src/seqan/EndToEndAffine.cpp:52-155) for `--affine`.  The host versions
// stay as the reference for tests and as the fall-back for strings that do not fit the scratch rows; what moves here is the
// part that hurt on the host: the MD string needs the reference window, which is already in HBM (the packed pairs of the
// align stage) but costs three cache-missing reads into a 3 GB array per read on the host.
// One lane per alignment; the packed pairs are 64-way interleaved, so a wave's lanes read consecutive dwords.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sw_device.h"

namespace ngm {

struct CigarDevOut {   // per alignment
	uint32_t cig_off, md_off;     // into the compact byte stream
	uint16_t cig_len, md_len;     // bytes, without the terminating NUL
	int32_t position_offset, qstart, qend, nm;
	float identity, score_token;
	int32_t flags;                // bit 0: strings valid (else: host fall-back), bit 1: alignment valid (linear: rec[0])
};

__device__ __forceinline__ uint32_t packed_class(const uint32_t *base, int m_base, int j) {  // symbol j of a 64-way interleaved row
	const uint32_t w = base[(size_t) (m_base + (j >> 3)) * kSlots];
	const int b = j & 7;
	return (b < 4) ? (w >> (8 * b)) & 15u : (w >> (8 * (b - 4) + 4)) & 15u;
}
__device__ __forceinline__ char class_to_char(uint32_t c) {
	switch (c) { case 0: return 'A'; case 1: return 'C'; case 2: return 'G'; case 3: return 'T'; case 4: return 'x'; case 5: return 'N'; default: return 0; }
}
__device__ __forceinline__ int dev_put_num(char *dst, int v) {
	char b[12];
	int n = 0, i = 0;
	unsigned u = v < 0 ? 0u - (unsigned) v : (unsigned) v;
	if (v < 0) dst[i++] = '-';
	do { b[n++] = (char) ('0' + u % 10u); u /= 10u; } while (u);
	while (n) dst[i++] = b[--n];
	return i;
}

// cig / md: scratch rows of `stride` bytes per alignment.  read_len: per READ (affine: QEnd); a_read: read of alignment j.
// The strings are built in LDS rows (kCigarRow bytes for the CIGAR, as many for MD -- byte stores to 608-byte-strided global rows
// cost one cache line per lane and store) and leave the workgroup as one contiguous piece of the compact byte stream: block
// prefix sum of the lengths, ONE atomic on the stream cursor per workgroup.
constexpr int kCigarRow = 96;

template <bool AFFINE>
__global__ __launch_bounds__(256) void cigar_strings_kernel(int n, const int32_t *__restrict__ records, const uint16_t *__restrict__ runs_c,
		const uint32_t *__restrict__ packed, int RW, int FW, const uint16_t *__restrict__ read_len, const uint32_t *__restrict__ a_read, int variant_cpu,
		int hard_clip, int silent_clip, CigarDevOut *__restrict__ out, char *__restrict__ bytes, unsigned long long capacity, unsigned long long *__restrict__ cursor, int alt = 0) {
	__shared__ char s_rows[256 * 2 * kCigarRow];
	__shared__ uint32_t s_wave[4];
	__shared__ unsigned long long s_base;
	const int j = blockIdx.x * blockDim.x + threadIdx.x;
	const bool live = j < n;
	constexpr int stride = kCigarRow;
	CigarDevOut o{};
	char *cg = s_rows + (size_t) threadIdx.x * 2 * kCigarRow, *mdp = cg + kCigarRow;
	const int lim = stride - 16;  // room for one more element; longer strings go to the host fall-back
	bool fits = true;
	int co = 0, mo = 0;
	if (live) {
	const int32_t *rec = records + (size_t) j * 8;
	const uint16_t *runs = runs_c + (uint32_t) rec[6];
	if (AFFINE) {
		int h = rec[1], v = rec[2], total = 0, pattern_chars = 0;
		o.position_offset = h; o.qstart = v;
		if (v > 0) { co += dev_put_num(cg + co, v); cg[co++] = 'S'; }
		for (int k = rec[4] - 1; k >= 0; --k) {
			const int op = runs[k] & 3, run = runs[k] >> 2;
			if (co > lim) { fits = false; break; }
			co += dev_put_num(cg + co, run);
			if (op == 1) { total += run; h += run; v += run; pattern_chars += run; cg[co++] = 'M'; }
			else if (op == 2) { total += 1; v += run; pattern_chars += run; cg[co++] = 'I'; }
			else { total += 1; h += run; cg[co++] = 'D'; }
		}
		const int len_v = (int) read_len[a_read[j]];
		o.qend = len_v - (pattern_chars + o.qstart);
		if (co > lim) fits = false;
		if (fits && o.qend > 0) { co += dev_put_num(cg + co, o.qend); cg[co++] = 'S'; }
		o.identity = (float) rec[3] * 1.0f / (float) total;
		o.nm = rec[7];
		o.score_token = 0.f;
		mdp[0] = '!'; mdp[1] = '!'; mdp[2] = '!'; mo = 3;  // EndToEndAffine never touches pBuffer2 (AlignmentBuffer.cpp:109)
		o.flags = (fits ? 1 : 0) | 2;
	} else {
		if (!rec[0]) {  // no alignment could be built: Score = -1 (SWOclCigar.cpp:322-327)
			o.score_token = -1.0f; o.flags = 1;
		} else {
			const uint32_t *pb = packed + (size_t) (j >> 6) * (RW + FW) * kSlots + (j & 63);
			const int lead = rec[2], trail = rec[3], nruns = rec[4], ref0 = rec[1];
			if (lead > 0) {
				if (hard_clip == 1) { co += dev_put_num(cg + co, lead); cg[co++] = 'H'; }
				else if (silent_clip != 1) { co += dev_put_num(cg + co, lead); cg[co++] = 'S'; }
				o.qstart = lead;
			}
			// symbol classes of read position j / window position j: eight per dword, the dword kept while it lasts
			uint32_t rw_at = 0xFFFFFFFFu, rw = 0, fw_at = 0xFFFFFFFFu, fw = 0;
			auto nib = [](uint32_t w, int b) -> uint32_t { return (b < 4) ? (w >> (8 * b)) & 15u : (w >> (8 * (b - 4) + 4)) & 15u; };
			// (bit 3 of a read class = the pair's score table, SwConst::alt)
			auto rclass = [&](int j) -> uint32_t { const uint32_t wi = (uint32_t) j >> 3; if (wi != rw_at) { rw = pb[(size_t) wi * kSlots]; rw_at = wi; } return nib(rw, j & 7) & 7u; };
			const uint32_t dir = alt ? ((pb[0] >> 3) & 1u) : 0u;
			// bs_mapping / slam_seq: the conversion that counts as a match (SWOclCigar.cpp:300-317), as symbol classes A0 C1 G2 T3
			const uint32_t bs_from = alt == 1 ? (dir ? 0u : 3u) : (dir ? 2u : 1u), bs_to = alt == 1 ? (dir ? 2u : 1u) : (dir ? 0u : 3u);
			auto fclass = [&](int j) -> uint32_t { const uint32_t wi = (uint32_t) j >> 3; if (wi != fw_at) { fw = pb[(size_t) (RW + wi) * kSlots]; fw_at = wi; } return nib(fw, j & 7); };
			int match = 0, mismatch = 0, total = 0, m_len = 0, md_eq = 0, ref_i = 0, read_i = o.qstart;
			bool in_x_run = false, odd_symbol = false;
			for (int k = nruns - 1; k >= 0 && fits; --k) {
				const int op = runs[k] & 3, len = runs[k] >> 2;
				total += len;
				if (co > lim || mo > lim) { fits = false; break; }
				if (op == 1 || op == 0) {
					for (int t = 0; t < len; ++t) {
						const uint32_t rc = rclass(read_i), fc = fclass(ref0 + ref_i);
						if (rc == 4u) odd_symbol = true;  // a read symbol outside ACGTN: characters and classes may disagree -> host
						const bool eq = (op == 1) && ((variant_cpu && !alt) ? (rc <= 3u && rc == fc) : (rc == fc));
						if (eq) { match += 1; md_eq += 1; in_x_run = false; }
						else {
							if (alt && rc == bs_from && fc == bs_to) match += 1; else mismatch += 1;   // SWOclCigar.cpp:507-514
							if (mo > lim) { fits = false; break; }
							if (!in_x_run) { mo += dev_put_num(mdp + mo, md_eq); md_eq = 0; in_x_run = true; }
							mdp[mo++] = class_to_char(fc);
						}
						m_len += 1; ref_i += 1; read_i += 1;
					}
				} else if (op == 3) {
					in_x_run = false;
					if (m_len > 0) { co += dev_put_num(cg + co, m_len); cg[co++] = 'M'; m_len = 0; }
					co += dev_put_num(cg + co, len); cg[co++] = 'D';
					mo += dev_put_num(mdp + mo, md_eq); md_eq = 0;
					mdp[mo++] = '^';
					for (int t = 0; t < len; ++t) { if (mo > lim) { fits = false; break; } mdp[mo++] = class_to_char(fclass(ref0 + ref_i)); ref_i += 1; }
					mismatch += len;
				} else {
					in_x_run = false;
					if (m_len > 0) { co += dev_put_num(cg + co, m_len); cg[co++] = 'M'; m_len = 0; }
					co += dev_put_num(cg + co, len); cg[co++] = 'I';
					read_i += len;
					mismatch += len;
				}
			}
			if (co > lim || mo > lim) fits = false;  // the last run may have used the head room the tail below needs: host fall-back
			if (fits) {
				mo += dev_put_num(mdp + mo, md_eq);
				if (m_len > 0) { co += dev_put_num(cg + co, m_len); cg[co++] = 'M'; }
				if (trail > 0) {
					if (hard_clip == 1) { co += dev_put_num(cg + co, trail); cg[co++] = 'H'; }
					else if (silent_clip != 1) { co += dev_put_num(cg + co, trail); cg[co++] = 'S'; }
					o.qend = trail;
				}
			}
			o.identity = (float) match * 1.0f / (float) total;
			o.nm = mismatch;
			o.score_token = (float) read_i;
			o.position_offset = ref0;
			o.flags = ((fits && !odd_symbol) ? 1 : 0) | 2;
		}
	}
	}  // live
	o.cig_len = (uint16_t) co; o.md_len = (uint16_t) mo;
	// this workgroup's piece of the stream
	const uint32_t need = (live && (o.flags & 1)) ? (uint32_t) (co + mo) : 0u;
	uint32_t incl = need;
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(incl, d); if (lane >= d) incl += t; }
	if (lane == 63) s_wave[wv] = incl;
	__syncthreads();
	uint32_t before = incl - need, total = 0;
#pragma unroll
	for (int w2 = 0; w2 < 4; ++w2) { if (w2 < wv) before += s_wave[w2]; total += s_wave[w2]; }
	if (threadIdx.x == 0) s_base = total ? atomicAdd(cursor, (unsigned long long) total) : 0ull;
	__syncthreads();
	const unsigned long long base = s_base;
	if (!live) return;
	if (need) {
		if (base + total > capacity) o.flags &= ~1;  // stream full: built on the host
		else {
			const unsigned long long off = base + before;
			o.cig_off = (uint32_t) off; o.md_off = (uint32_t) off + o.cig_len;
			for (int t = 0; t < co; ++t) bytes[off + t] = cg[t];
			for (int t = 0; t < mo; ++t) bytes[off + co + t] = mdp[t];
		}
	}
	out[j] = o;
}

}  // namespace ngm
