// ngm_hip.cpp -- host side of the flat C ABI (include/ngm_hip.h): context, HBM workspace, kernel
// dispatch.  Compiled by hipcc for gfx950 only.  There is deliberately no CPU fallback: without a
// HIP device every entry point fails loudly.
#include "../../include/ngm_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#define NGM_ENGINE_KERNELS
#include "engine_internal.h"
#include "align_device.h"
#include "affine_device.h"
#include "cigar_md.h"
#include "jit.h"

namespace {

thread_local std::string g_last_error;

void set_error(ngm_hip_ctx *ctx, const char *fmt, ...);

#define HIP_TRY(ctx, expr)                                                                         \
	do {                                                                                           \
		hipError_t e_ = (expr);                                                                    \
		if (e_ != hipSuccess) {                                                                    \
			set_error(ctx, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
			return -5;                                                                             \
		}                                                                                          \
	} while (0)

}  // namespace


namespace {

void set_error(ngm_hip_ctx *ctx, const char *fmt, ...) {
	char buf[1024];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof(buf), fmt, ap);
	va_end(ap);
	g_last_error = buf;
	if (ctx) ctx->error = buf;
}

// ---- kernel dispatch over the compiled corridor widths ------------------------------------------
// The band width is a compile-time shape exactly as in the reference (its kernels are JIT-compiled
// with -D corridor_length, lib/mason/opencl/SWOcl.cpp:206-217).  These are the shapes built ahead
// of time; NGM derives corridor = int(5 + 0.15 * avg_read_len) (src/ReadProvider.cpp:301-305) or
// 2 * max-consec-indels (src/config/Config.cpp:540-557).
#define NGM_CORRIDORS(X) X(8) X(12) X(19) X(20) X(27) X(42) X(80)

// A DP kernel is either one of the ahead-of-time instantiations (host stub) or a function of a run-time compiled module.
struct KernelRef {
	const void *aot = nullptr;
	hipFunction_t jit = nullptr;
	explicit operator bool() const { return aot || jit; }
};

// kind: 0/1 linear score local/end-to-end, 2/3 linear align, 4/5 affine score, 6/7 affine align (jit.h)
const void *find_aot_kernel(int c, int kind) {
#define X(C) if (c == C) switch (kind) { \
		case 0: return (const void *) ngm::sw_score_kernel<C, false>; case 1: return (const void *) ngm::sw_score_kernel<C, true>; \
		case 2: return (const void *) ngm::sw_align_kernel<C, false>; case 3: return (const void *) ngm::sw_align_kernel<C, true>; \
		case 4: return (const void *) ngm::sw_affine_kernel<C + 1, false, false>; case 5: return (const void *) ngm::sw_affine_kernel<C + 1, true, false>; \
		case 6: return (const void *) ngm::sw_affine_kernel<C + 1, false, true>; default: return (const void *) ngm::sw_affine_kernel<C + 1, true, true>; }
	NGM_CORRIDORS(X)
#undef X
	return nullptr;
}

// packed 16-bit linear score kernels (two pairs per lane) of the ahead-of-time corridors
const void *find_pk_score_kernel(int c, bool endfree) {
#define X(C) if (c == C) return endfree ? (const void *) ngm::sw_score_pk_kernel<C, true> : (const void *) ngm::sw_score_pk_kernel<C, false>;
	NGM_CORRIDORS(X)
#undef X
	return nullptr;
}

const void *find_pk_affine_score_kernel(int c, bool endfree) {
#define X(C) if (c == C) return endfree ? (const void *) ngm::sw_affine_score_pk_kernel<C + 1, true> : (const void *) ngm::sw_affine_score_pk_kernel<C + 1, false>;
	NGM_CORRIDORS(X)
#undef X
	return nullptr;
}

// wide bands, local mode: the band's columns split over two lanes (sw_affine_score_pk_split_kernel; three waves per SIMD instead of one)
const void *find_pk_affine_score_split_kernel(int c) {
	if (c == 80) return (const void *) ngm::sw_affine_score_pk_split_kernel<81>;
	return nullptr;
}

// window = false: the plain row key (<= 32 band columns, scores below 2 048); true: the windowed one (<= 128 columns)
template <int CP> const void *pk_affine_align_ptr(bool window) {
	if (!window) { if constexpr (CP <= 32) return (const void *) ngm::sw_affine_align_pk_kernel<CP, false>; else return nullptr; }
	if constexpr (CP <= 128) return (const void *) ngm::sw_affine_align_pk_kernel<CP, true>; else return nullptr;
}
const void *find_pk_affine_align_kernel(int c, bool window) {
#define X(C) if (c == C) return pk_affine_align_ptr<C + 1>(window);
	NGM_CORRIDORS(X)
#undef X
	return nullptr;
}

// packed 16-bit linear align kernels (local mode): plain row key (<= 32 band columns, scores below 2 048) or the windowed one
template <int C> const void *pk_align_ptr(bool window) {
	if (!window) { if constexpr (C <= 32) return (const void *) ngm::sw_align_pk_kernel<C, false>; else return nullptr; }
	if constexpr (C <= 128) return (const void *) ngm::sw_align_pk_kernel<C, true>; else return nullptr;
}
const void *find_pk_align_kernel(int c, bool window) {
#define X(C) if (c == C) return pk_align_ptr<C>(window);
	NGM_CORRIDORS(X)
#undef X
	return nullptr;
}

KernelRef find_kernel(ngm_hip_ctx *ctx, int kind) {
	KernelRef k;
	k.aot = find_aot_kernel(ctx->c, kind);
	if (!k.aot && ctx->jit) k.jit = ctx->jit->fn[kind];
	return k;
}

template <typename... Args>
hipError_t launch_kernel(const KernelRef &k, dim3 grid, dim3 block, hipStream_t st, Args... args) {
	void *argv[] = {(void *) &args...};
	if (k.aot) return hipLaunchKernel(k.aot, grid, block, argv, 0, st);
	return hipModuleLaunchKernel(k.jit, grid.x, grid.y, grid.z, block.x, block.y, block.z, 0, st, argv, nullptr);
}

int n_blocks_of(int n) { return (n + ngm::kSlots - 1) / ngm::kSlots; }

}  // namespace
namespace ngm {
int engine_reserve(ngm_hip_ctx *ctx, int n) {
	const size_t nb = (size_t) n_blocks_of(n);
	if (ctx->packed.reserve(nb * (size_t) (ctx->RW + ctx->FW) * ngm::kSlots) || ctx->lens.reserve(nb * ngm::kSlots) ||
			ctx->blk_rows.reserve(nb)) {
		set_error(ctx, "out of device memory reserving the packed workspace for %d pairs", n);
		return -12;
	}
	return 0;
}

int engine_score_packed(ngm_hip_ctx *ctx, int mode, int n, float *d_scores, hipStream_t st) {
	const int nb = n_blocks_of(n);
	static const bool force32 = getenv("NGM_HIP_SCORE_32BIT") != nullptr;  // NGM_HIP_SCORE_32BIT=1: never use the packed 16-bit kernels
	if (ctx->prm.personality == NGM_PERSONALITY_AFFINE) {
		const bool ef = (mode & NGM_MODE_ALIGN_MASK) == NGM_MODE_END_TO_END;
		if (!force32 && (long) ctx->q * ctx->KA.tM < 30000 && ctx->prm.gap_read_penalty + (long) ctx->q * (ctx->prm.gap_extend_penalty + ctx->KA.tZ) < 19000) {
			if (const void *sp = ef ? nullptr : find_pk_affine_score_split_kernel(ctx->c)) {
				KernelRef kp; kp.aot = sp;
				const int units = 2 * ((nb + 1) / 2);   // (block pair, half of its slots) per wave
				HIP_TRY(ctx, launch_kernel(kp, dim3((units + 3) / 4), dim3(256), st, (const uint32_t *) ctx->packed.p, (const uint16_t *) ctx->lens.p,
						(const uint16_t *) ctx->blk_rows.p, d_scores, n, nb, ctx->RW, ctx->KA));
				return 0;
			}
			if (const void *pk = find_pk_affine_score_kernel(ctx->c, ef)) {
				KernelRef kp; kp.aot = pk;
				HIP_TRY(ctx, launch_kernel(kp, dim3((nb + 7) / 8), dim3(256), st, (const uint32_t *) ctx->packed.p, (const uint16_t *) ctx->lens.p,
						(const uint16_t *) ctx->blk_rows.p, d_scores, n, nb, ctx->RW, ctx->KA));
				return 0;
			}
		}
		const KernelRef ka = find_kernel(ctx, 4 + ((mode & NGM_MODE_ALIGN_MASK) == NGM_MODE_END_TO_END ? 1 : 0));
		HIP_TRY(ctx, launch_kernel(ka, dim3((nb + 3) / 4), dim3(256), st, (const uint32_t *) ctx->packed.p, (const uint16_t *) ctx->lens.p,
				(const uint16_t *) ctx->blk_rows.p, d_scores, (uint32_t *) nullptr, (int32_t *) nullptr, n, nb, ctx->RW, ctx->q, ctx->KA));
		return 0;
	}
	const bool endfree = (mode & NGM_MODE_ALIGN_MASK) == NGM_MODE_END_TO_END;
	// two pairs per lane in 16-bit halves while every re-based value fits (rows * (match - mismatch) and the end-to-end
	// sentinel stay inside int16)
	if (!force32 && (long) ctx->q * std::max(ctx->K.tM, ctx->K.alt ? std::max(ctx->K.tMA, ctx->K.tXA) : 0) < 30000 && (long) ctx->q * ctx->K.tZ < 14000) {
		if (const void *pk = find_pk_score_kernel(ctx->c, endfree)) {
			KernelRef kp; kp.aot = pk;
			HIP_TRY(ctx, launch_kernel(kp, dim3((nb + 7) / 8), dim3(256), st, (const uint32_t *) ctx->packed.p, (const uint16_t *) ctx->lens.p,
					(const uint16_t *) ctx->blk_rows.p, d_scores, n, nb, ctx->RW, ctx->K));
			return 0;
		}
	}
	const KernelRef k = find_kernel(ctx, endfree ? 1 : 0);
	HIP_TRY(ctx, launch_kernel(k, dim3((nb + 3) / 4), dim3(256), st, (const uint32_t *) ctx->packed.p, (const uint16_t *) ctx->lens.p,
			(const uint16_t *) ctx->blk_rows.p, d_scores, n, nb, ctx->RW, ctx->K));
	return 0;
}

int engine_align_packed(ngm_hip_ctx *ctx, int mode, int n, int32_t *d_records, uint16_t *d_runs, int run_stride, hipStream_t st) {
	const int am = mode & NGM_MODE_ALIGN_MASK;
	const int nb = n_blocks_of(n);
	if (ctx->prm.personality == NGM_PERSONALITY_AFFINE) {
		const int CP = ctx->c + 1, ADW = ngm::aff_dir_words(CP);
		if (ctx->dirs.reserve((size_t) nb * ctx->q * ADW * ngm::kSlots)) { set_error(ctx, "out of device memory for the trace matrix"); return -12; }
		// local alignments of short reads: two pairs per lane in 16-bit halves, 4-bit trace (see sw_affine_align_pk_kernel for the
		// conditions: scores below 2 048, at most 32 band columns, negative gap penalties, the 16-bit range of the score kernel)
		static const bool force32 = getenv("NGM_HIP_ALIGN_32BIT") != nullptr;
		const void *pk = nullptr;
		if (!force32 && am != NGM_MODE_END_TO_END && ctx->KA.open < 0 && ctx->KA.ext < 0 && (long) ctx->q * ctx->KA.tM < 30000 &&
				ctx->prm.gap_read_penalty + (long) ctx->q * (ctx->prm.gap_extend_penalty + ctx->KA.tZ) < 19000) {
			const bool plain_key = (long) ctx->q * ctx->prm.match_bonus < 2048 && ctx->c + 1 <= 32;
			if (plain_key) pk = find_pk_affine_align_kernel(ctx->c, false);
			else if (ctx->KA.tZ > 0 && ctx->KA.tM > 0 && ctx->KA.tM + 1 <= 63) pk = find_pk_affine_align_kernel(ctx->c, true);  // window: mismatch + match + 1 values
		}
		if (pk) {
			KernelRef kp; kp.aot = pk;
			HIP_TRY(ctx, launch_kernel(kp, dim3((nb + 7) / 8), dim3(256), st, (const uint32_t *) ctx->packed.p, (const uint16_t *) ctx->lens.p,
					(const uint16_t *) ctx->blk_rows.p, ctx->dirs.p, d_records, n, nb, ctx->RW, ctx->q, ctx->KA));
		} else {
			const KernelRef ka = find_kernel(ctx, 6 + (am == NGM_MODE_END_TO_END ? 1 : 0));
			HIP_TRY(ctx, launch_kernel(ka, dim3((nb + 3) / 4), dim3(256), st, (const uint32_t *) ctx->packed.p, (const uint16_t *) ctx->lens.p,
					(const uint16_t *) ctx->blk_rows.p, (float *) nullptr, ctx->dirs.p, d_records, n, nb, ctx->RW, ctx->q, ctx->KA));
		}
		if (ctx->profiling) HIP_TRY(ctx, hipEventRecord(ctx->ev[2], st));
		hipLaunchKernelGGL(ngm::affine_traceback_kernel, dim3((n + 255) / 256), dim3(256), 0, st, ctx->dirs.p, d_records, d_runs, n,
				ctx->q, CP, run_stride, (const uint32_t *) ctx->packed.p, ctx->RW, ctx->FW, pk ? 1 : 0);
		HIP_TRY(ctx, hipGetLastError());
		return 0;
	}
	const int DW = ngm::dir_words(ctx->c);
	if (ctx->dirs.reserve((size_t) nb * ctx->q * DW * ngm::kSlots)) { set_error(ctx, "out of device memory for the direction matrix"); return -12; }
	// local alignments whose re-based values fit 16 bits (the score kernel's condition): two pairs per lane, see sw_align_pk_kernel
	static const bool force32 = getenv("NGM_HIP_ALIGN_32BIT") != nullptr;
	const void *pk = nullptr;
	const int max_gain = std::max(ctx->K.tM, ctx->K.alt ? std::max(ctx->K.tMA, ctx->K.tXA) : 0);  // largest table byte: best column score - mismatch
	if (!force32 && am != NGM_MODE_END_TO_END && (long) ctx->q * max_gain < 30000 && (long) ctx->q * ctx->K.tZ < 14000 && ctx->K.gl < 0 && ctx->K.gap_read < 0) {
		const bool plain_key = (long) ctx->q * (max_gain - ctx->K.tZ) < 2048 && ctx->c <= 32;
		if (plain_key) pk = find_pk_align_kernel(ctx->c, false);
		else if (ctx->K.tZ > 0 && max_gain + 1 <= 63) pk = find_pk_align_kernel(ctx->c, true);  // window: mismatch + best column score + 1 values
	}
	if (pk) {
		KernelRef kp; kp.aot = pk;
		HIP_TRY(ctx, launch_kernel(kp, dim3((nb + 7) / 8), dim3(256), st, (const uint32_t *) ctx->packed.p, (const uint16_t *) ctx->lens.p,
				(const uint16_t *) ctx->blk_rows.p, ctx->dirs.p, d_records, n, nb, ctx->RW, ctx->q, ctx->K));
	} else {
		const KernelRef k = find_kernel(ctx, 2 + (am == NGM_MODE_END_TO_END ? 1 : 0));
		HIP_TRY(ctx, launch_kernel(k, dim3((nb + 3) / 4), dim3(256), st, (const uint32_t *) ctx->packed.p, (const uint16_t *) ctx->lens.p,
				(const uint16_t *) ctx->blk_rows.p, ctx->dirs.p, d_records, n, nb, ctx->RW, ctx->q, ctx->K));
	}
	if (ctx->profiling) HIP_TRY(ctx, hipEventRecord(ctx->ev[2], st));
	hipLaunchKernelGGL(ngm::sw_traceback_kernel, dim3((n + 255) / 256), dim3(256), 0, st, ctx->dirs.p, ctx->lens.p, d_records,
			d_runs, n, ctx->q, ctx->c, run_stride, am == NGM_MODE_END_TO_END ? 1 : 0);
	HIP_TRY(ctx, hipGetLastError());
	return 0;
}
}  // namespace ngm
namespace {
using ngm::engine_reserve;

struct ScopedDevice {
	int prev = -1;
	explicit ScopedDevice(int dev) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; if (prev != dev) (void) hipSetDevice(dev); else prev = -1; }
	~ScopedDevice() { if (prev >= 0) (void) hipSetDevice(prev); }
};

int launch_pack(ngm_hip_ctx *ctx, int n, const void *d_ref, const void *d_qry, hipStream_t st) {
	const int nb = n_blocks_of(n);
	const size_t lds = (size_t) ngm::kSlots * (ctx->rl + ctx->q);
	hipLaunchKernelGGL(ngm::pack_pairs_kernel, dim3(nb), dim3(256), lds, st, (const uint8_t *) d_ref,
			(const uint8_t *) d_qry, n, ctx->q, ctx->rl, ctx->RW, ctx->FW, ctx->packed.p, ctx->lens.p, ctx->blk_rows.p,
			ctx->prm.personality == NGM_PERSONALITY_AFFINE ? 1 : 0, (ctx->K.alt || ctx->prm.alt_cigar) ? ctx->pair_dir : nullptr);
	HIP_TRY(ctx, hipGetLastError());
	return 0;
}

// the `dir` bytes of a host-pointer call -> device (null: all pairs use the FWD table)
int stage_dirs(ngm_hip_ctx *ctx, int n, const char *dir) {
	ctx->pair_dir = nullptr;
	if ((!ctx->K.alt && !ctx->prm.alt_cigar) || !dir) return 0;
	if (ctx->d_pair_dir.reserve(n)) { set_error(ctx, "out of memory staging %d pairs", n); return -12; }
	HIP_TRY(ctx, hipMemcpyAsync(ctx->d_pair_dir.p, dir, (size_t) n, hipMemcpyHostToDevice, ctx->stream));
	ctx->pair_dir = ctx->d_pair_dir.p;
	return 0;
}

}  // namespace

extern "C" {

const char *ngm_hip_last_error(const ngm_hip_ctx *ctx) { return ctx ? ctx->error.c_str() : g_last_error.c_str(); }

int ngm_hip_device_count(void) {
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

ngm_hip_ctx *ngm_hip_create(int device, const ngm_hip_params *p) {
	if (!p || p->abi_version != NGM_HIP_ABI_VERSION) { set_error(nullptr, "ngm_hip_create: bad params / ABI version"); return nullptr; }
	if (p->qry_max_len < 2 || p->qry_max_len > 4096 || p->corridor < 2) { set_error(nullptr, "ngm_hip_create: qry_max_len=%d corridor=%d out of range", p->qry_max_len, p->corridor); return nullptr; }
	if (p->match_bonus <= 0 || p->mismatch_penalty <= 0 || p->gap_read_penalty <= 0 || p->gap_ref_penalty <= 0) {
		set_error(nullptr, "ngm_hip_create: scores must be positive integers (match %d mismatch %d gap_read %d gap_ref %d)", p->match_bonus, p->mismatch_penalty, p->gap_read_penalty, p->gap_ref_penalty);
		return nullptr;
	}
	if (p->personality != NGM_PERSONALITY_LINEAR && p->personality != NGM_PERSONALITY_AFFINE) { set_error(nullptr, "ngm_hip_create: unknown personality %d", p->personality); return nullptr; }
	if (p->personality == NGM_PERSONALITY_AFFINE && p->gap_extend_penalty <= 0) { set_error(nullptr, "ngm_hip_create: gap_extend_penalty must be a positive integer"); return nullptr; }
	if (p->match_bonus + p->mismatch_penalty > 255) { set_error(nullptr, "ngm_hip_create: match_bonus + mismatch_penalty must be <= 255"); return nullptr; }
	if (p->corridor > 200) { set_error(nullptr, "ngm_hip_create: corridor %d too wide (the band row lives in registers)", p->corridor); return nullptr; }
	if (p->alt_cigar != NGM_ALT_NONE && p->alt_cigar != NGM_ALT_BISULFITE && p->alt_cigar != NGM_ALT_SLAMSEQ) { set_error(nullptr, "ngm_hip_create: unknown alt_cigar %d", p->alt_cigar); return nullptr; }
	if (p->alt_cigar != NGM_ALT_NONE && p->personality != NGM_PERSONALITY_LINEAR) { set_error(nullptr, "ngm_hip_create: bisulfite / SLAM-seq and the affine personality can't be used at the same time"); return nullptr; }
	if (p->alt_scoring != NGM_ALT_NONE) {
		// Config.cpp:448-460: bs-mapping and affine exclude each other; the tables hold (score - mismatch) as unsigned bytes
		if (p->alt_scoring != NGM_ALT_BISULFITE && p->alt_scoring != NGM_ALT_SLAMSEQ) { set_error(nullptr, "ngm_hip_create: unknown alt_scoring %d", p->alt_scoring); return nullptr; }
		if (p->personality != NGM_PERSONALITY_LINEAR) { set_error(nullptr, "ngm_hip_create: bisulfite / SLAM-seq scoring and the affine personality can't be used at the same time"); return nullptr; }
		const int m_alt = p->match_bonus_tt, x_alt = p->alt_scoring == NGM_ALT_SLAMSEQ ? -p->match_bonus_tc : p->match_bonus_tc;
		if (m_alt + p->mismatch_penalty < 0 || m_alt + p->mismatch_penalty > 255 || x_alt + p->mismatch_penalty < 0 || x_alt + p->mismatch_penalty > 255) {
			set_error(nullptr, "ngm_hip_create: match_bonus_tt %d / match_bonus_tc %d out of range for mismatch_penalty %d", p->match_bonus_tt, p->match_bonus_tc, p->mismatch_penalty);
			return nullptr;
		}
	}
	int ndev = ngm_hip_device_count();
	if (ndev <= 0) { set_error(nullptr, "ngm_hip_create: no HIP device available (this library has no CPU fallback)"); return nullptr; }
	if (device < 0 || device >= ndev) { set_error(nullptr, "ngm_hip_create: device %d out of range (%d devices)", device, ndev); return nullptr; }
	ScopedDevice sd(device);
	ngm_hip_ctx *ctx = new ngm_hip_ctx();
	ctx->device = device;
	ctx->prm = *p;
	ctx->q = p->qry_max_len;
	ctx->c = p->corridor;
	ctx->rl = ctx->q + ctx->c;
	ctx->RW = ngm::read_words(ctx->q);
	// the affine band has corridor + 1 diagonals (lDiag = 0 .. uDiag = corridor, EndToEndAffine.h:40-41)
	ctx->FW = ngm::ref_words(ctx->q, p->personality == NGM_PERSONALITY_AFFINE ? ctx->c + 1 : ctx->c);
	ctx->max_batch = p->max_batch > 0 ? p->max_batch : (1 << 20);
	if (!find_aot_kernel(ctx->c, 0)) {
		// no ahead-of-time build for this band width: compile the kernels now, like the reference's OpenCL JIT
		std::string err;
		ctx->jit = ngm::jit_kernels_for_corridor(ctx->c, &err);
		if (!ctx->jit) { set_error(nullptr, "ngm_hip_create: %s", err.c_str()); delete ctx; return nullptr; }
	}
	const int match = p->match_bonus, mismatch = -p->mismatch_penalty, gap_read = -p->gap_read_penalty, gap_ref = -p->gap_ref_penalty;
	ctx->K.tM = match - mismatch;
	ctx->K.tZ = -mismatch;
	ctx->K.gl = gap_ref;
	ctx->K.gu = gap_read - mismatch;
	ctx->K.gap_read = gap_read;
	ctx->K.variant = p->variant;
	ctx->K.alt = p->alt_scoring;
	ctx->K.tMA = p->match_bonus_tt - mismatch;                                                            // SWOcl.cpp:230-237
	ctx->K.tXA = (p->alt_scoring == NGM_ALT_SLAMSEQ ? -p->match_bonus_tc : p->match_bonus_tc) - mismatch;
	// Score<float, Simple>(match, -mismatch, -gap_extend, -gap_read): extend = gap_extend, open = gap_read (EndToEndAffine.h:37)
	ctx->KA.tM = match - mismatch;
	ctx->KA.tZ = -mismatch;
	ctx->KA.open = gap_read;
	ctx->KA.ext = -p->gap_extend_penalty;
	ctx->KA.vopen = gap_read - mismatch;
	ctx->KA.vext = -p->gap_extend_penalty - mismatch;
	if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { set_error(nullptr, "ngm_hip_create: hipStreamCreate failed"); delete ctx; return nullptr; }
	for (auto &e : ctx->ev) if (hipEventCreate(&e) != hipSuccess) { set_error(nullptr, "ngm_hip_create: hipEventCreate failed"); ngm_hip_destroy(ctx); return nullptr; }
	// the pack kernel stages 64 raw pairs in LDS
	const size_t lds = (size_t) ngm::kSlots * (ctx->rl + ctx->q);
	if (lds > 160 * 1024 - 2048) { set_error(nullptr, "ngm_hip_create: qry_max_len too large for the LDS staging tile"); ngm_hip_destroy(ctx); return nullptr; }
	if (lds > 48 * 1024) (void) hipFuncSetAttribute((const void *) ngm::pack_pairs_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
	return ctx;
}

void ngm_hip_destroy(ngm_hip_ctx *ctx) {
	if (!ctx) return;
	ScopedDevice sd(ctx->device);
	if (ctx->stream) (void) hipStreamSynchronize(ctx->stream);
	ctx->packed.release(); ctx->lens.release(); ctx->blk_rows.release(); ctx->d_ref.release(); ctx->d_qry.release();
	ctx->d_scores.release(); ctx->dirs.release(); ctx->d_pair_dir.release(); ctx->d_records.release(); ctx->d_runs.release();
	ctx->h_ref.release(); ctx->h_qry.release(); ctx->h_scores.release(); ctx->h_records.release(); ctx->h_runs.release();
	for (auto &e : ctx->ev) if (e) (void) hipEventDestroy(e);
	if (ctx->stream) (void) hipStreamDestroy(ctx->stream);
	delete ctx;
}

// The reference sizes its batches from the device (CUs * block_multiplier * ..., SWOcl.cpp:558-590);
// callers allocate their buffers from these once (ScoreBuffer.h:92, AlignmentBuffer.h:63).
int ngm_hip_score_batch_size(const ngm_hip_ctx *ctx) { return ctx ? ctx->max_batch : 0; }
int ngm_hip_align_batch_size(const ngm_hip_ctx *ctx) { return ctx ? ctx->max_batch : 0; }

void ngm_hip_set_profiling(ngm_hip_ctx *ctx, int enabled) { if (ctx) ctx->profiling = enabled != 0; }

int ngm_hip_last_kernel_ms(ngm_hip_ctx *ctx, float ms[3]) {
	if (!ctx) return -22;
	ScopedDevice sd(ctx->device);
	for (int i = 0; i < 3; ++i) {
		ms[i] = 0.f;
		if (ctx->ev_valid[i]) {
			HIP_TRY(ctx, hipEventSynchronize(ctx->ev[i + 1]));
			HIP_TRY(ctx, hipEventElapsedTime(&ms[i], ctx->ev[i], ctx->ev[i + 1]));
		}
	}
	return 0;
}

int ngm_hip_score_device(ngm_hip_ctx *ctx, int mode, int n, const void *d_ref, const void *d_qry,
		float *d_scores, void *stream) {
	if (!ctx) return -22;
	if (n <= 0) return 0;
	const int am = mode & NGM_MODE_ALIGN_MASK;
	if (am != NGM_MODE_LOCAL && am != NGM_MODE_END_TO_END) { set_error(ctx, "unsupported alignment mode %d", am); return -22; }
	ScopedDevice sd(ctx->device);
	hipStream_t st = stream ? (hipStream_t) stream : ctx->stream;
	if (int r = engine_reserve(ctx, n)) return r;
	const int nb = n_blocks_of(n);
	ctx->ev_valid[0] = ctx->ev_valid[1] = ctx->ev_valid[2] = false;
	if (ctx->profiling) HIP_TRY(ctx, hipEventRecord(ctx->ev[0], st));
	if (int r = launch_pack(ctx, n, d_ref, d_qry, st)) return r;
	if (ctx->profiling) HIP_TRY(ctx, hipEventRecord(ctx->ev[1], st));
	if (int r = ngm::engine_score_packed(ctx, mode, n, d_scores, st)) return r;
	if (ctx->profiling) { HIP_TRY(ctx, hipEventRecord(ctx->ev[2], st)); ctx->ev_valid[0] = ctx->ev_valid[1] = true; }
	return n;
}

int ngm_hip_batch_score(ngm_hip_ctx *ctx, int mode, int n, const char *const *ref, const char *const *qry,
		float *scores, const char *dir) {
	if (!ctx) return -22;
	if (n <= 0) return 0;  // SWOcl.cpp:39-42
	ScopedDevice sd(ctx->device);
	if (int r0 = stage_dirs(ctx, n, dir)) return r0;
	const size_t rl = ctx->rl, q = ctx->q;
	if (ctx->h_ref.reserve((size_t) n * rl) || ctx->h_qry.reserve((size_t) n * q) || ctx->h_scores.reserve(n) ||
			ctx->d_ref.reserve((size_t) n * rl) || ctx->d_qry.reserve((size_t) n * q) || ctx->d_scores.reserve(n)) {
		set_error(ctx, "out of memory staging %d pairs", n);
		return -12;
	}
	// flatten exactly the bytes the reference copies (SWOcl.cpp:545-548): q+c of the window, q of the read
	for (int i = 0; i < n; ++i) {
		memcpy(ctx->h_ref.p + (size_t) i * rl, ref[i], rl);
		memcpy(ctx->h_qry.p + (size_t) i * q, qry[i], q);
	}
	HIP_TRY(ctx, hipMemcpyAsync(ctx->d_ref.p, ctx->h_ref.p, (size_t) n * rl, hipMemcpyHostToDevice, ctx->stream));
	HIP_TRY(ctx, hipMemcpyAsync(ctx->d_qry.p, ctx->h_qry.p, (size_t) n * q, hipMemcpyHostToDevice, ctx->stream));
	int r = ngm_hip_score_device(ctx, mode, n, ctx->d_ref.p, ctx->d_qry.p, ctx->d_scores.p, ctx->stream);
	if (r < 0) return r;
	HIP_TRY(ctx, hipMemcpyAsync(ctx->h_scores.p, ctx->d_scores.p, sizeof(float) * (size_t) n, hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	memcpy(scores, ctx->h_scores.p, sizeof(float) * (size_t) n);
	return n;
}

int ngm_hip_set_pair_directions(ngm_hip_ctx *ctx, const void *d_dir) {
	if (!ctx) return -22;
	ctx->pair_dir = (const uint8_t *) d_dir;
	return 0;
}

int ngm_hip_align_run_stride(const ngm_hip_ctx *ctx) { return ctx ? ngm::run_stride(ctx->q, ctx->c) : 0; }

int ngm_hip_align_device(ngm_hip_ctx *ctx, int mode, int n, const void *d_ref, const void *d_qry,
		int32_t *d_records, uint16_t *d_runs, int run_stride, void *stream) {
	if (!ctx) return -22;
	if (n <= 0) return 0;
	const int am = mode & NGM_MODE_ALIGN_MASK;
	if (am != NGM_MODE_LOCAL && am != NGM_MODE_END_TO_END) { set_error(ctx, "unsupported alignment mode %d", am); return -22; }
	if (run_stride < ngm::run_stride(ctx->q, ctx->c)) { set_error(ctx, "run_stride %d too small", run_stride); return -22; }
	ScopedDevice sd(ctx->device);
	hipStream_t st = stream ? (hipStream_t) stream : ctx->stream;
	if (int r = engine_reserve(ctx, n)) return r;
	const int nb = n_blocks_of(n);
	ctx->ev_valid[0] = ctx->ev_valid[1] = ctx->ev_valid[2] = false;
	if (ctx->profiling) HIP_TRY(ctx, hipEventRecord(ctx->ev[0], st));
	if (int r = launch_pack(ctx, n, d_ref, d_qry, st)) return r;
	if (ctx->profiling) HIP_TRY(ctx, hipEventRecord(ctx->ev[1], st));
	if (int r = ngm::engine_align_packed(ctx, mode, n, d_records, d_runs, run_stride, st)) return r;
	if (ctx->profiling) { HIP_TRY(ctx, hipEventRecord(ctx->ev[3], st)); ctx->ev_valid[0] = ctx->ev_valid[1] = ctx->ev_valid[2] = true; }
	return n;
}

int ngm_hip_batch_align(ngm_hip_ctx *ctx, int mode, int n, const char *const *ref, const char *const *qry,
		ngm_hip_align_out *out, const char *dir) {
	if (!ctx) return -22;
	if (n <= 0) return 0;  // SWOclCigar.cpp:109-112
	ScopedDevice sd(ctx->device);
	if (int r0 = stage_dirs(ctx, n, dir)) return r0;
	const size_t rl = ctx->rl, q = ctx->q;
	const int rs = ngm::run_stride(ctx->q, ctx->c);
	if (ctx->h_ref.reserve((size_t) n * rl) || ctx->h_qry.reserve((size_t) n * q) || ctx->d_ref.reserve((size_t) n * rl) ||
			ctx->d_qry.reserve((size_t) n * q) || ctx->d_records.reserve((size_t) n * 8) || ctx->d_runs.reserve((size_t) n * rs) ||
			ctx->h_records.reserve((size_t) n * 8) || ctx->h_runs.reserve((size_t) n * rs)) {
		set_error(ctx, "out of memory staging %d pairs", n);
		return -12;
	}
	for (int i = 0; i < n; ++i) {
		memcpy(ctx->h_ref.p + (size_t) i * rl, ref[i], rl);
		memcpy(ctx->h_qry.p + (size_t) i * q, qry[i], q);
	}
	HIP_TRY(ctx, hipMemcpyAsync(ctx->d_ref.p, ctx->h_ref.p, (size_t) n * rl, hipMemcpyHostToDevice, ctx->stream));
	HIP_TRY(ctx, hipMemcpyAsync(ctx->d_qry.p, ctx->h_qry.p, (size_t) n * q, hipMemcpyHostToDevice, ctx->stream));
	int r = ngm_hip_align_device(ctx, mode, n, ctx->d_ref.p, ctx->d_qry.p, ctx->d_records.p, ctx->d_runs.p, rs, ctx->stream);
	if (r < 0) return r;
	HIP_TRY(ctx, hipMemcpyAsync(ctx->h_records.p, ctx->d_records.p, sizeof(int32_t) * 8 * (size_t) n, hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(ctx, hipMemcpyAsync(ctx->h_runs.p, ctx->d_runs.p, sizeof(uint16_t) * (size_t) rs * n, hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	ngm::CigarParams cp;
	cp.match = ctx->prm.match_bonus;
	cp.mismatch = -ctx->prm.mismatch_penalty;
	cp.variant = ctx->prm.variant;
	cp.hard_clip = ctx->prm.hard_clip;
	cp.silent_clip = ctx->prm.silent_clip;
	cp.alt = ctx->prm.alt_cigar;
	if (ctx->prm.personality == NGM_PERSONALITY_AFFINE) {
		for (int i = 0; i < n; ++i)
			ngm::build_cigar_affine(ctx->h_records.p + (size_t) i * 8, ctx->h_runs.p + (size_t) i * rs, ref[i], qry[i], ctx->q, &out[i]);
		return n;
	}
	for (int i = 0; i < n; ++i) {
		ngm::build_cigar_md(cp, ctx->h_records.p + (size_t) i * 8, ctx->h_runs.p + (size_t) i * rs, ref[i], qry[i], &out[i], (cp.alt && dir && dir[i]) ? 1 : 0);
	}
	return n;
}

}  // extern "C"
