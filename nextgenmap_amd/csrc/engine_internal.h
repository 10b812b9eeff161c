// engine_internal.h -- private to the library: context layout shared by the C-ABI TU (ngm_hip.cpp) and
// the mapping pipeline (mapper.cpp), plus the entry points that run the DP kernels on an already packed batch.
#pragma once

#include <algorithm>

#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

#include "../../include/ngm_hip.h"
#include "sw_device.h"
#include "affine_device.h"
#include "jit.h"

namespace ngm {
template <typename T>
struct DevBuf {
	T *p = nullptr;
	size_t cap = 0;  // elements
	int reserve(size_t n) {
		if (n <= cap) return 0;
		if (p) (void) hipFree(p);
		p = nullptr;
		cap = 0;
		// (an eighth more than asked for: batches of a real run differ by a few per cent, and every new maximum would otherwise be a
		// hipFree + hipMalloc -- two device-wide synchronisations -- in the middle of the mapping pass)
		const size_t want = n + std::min<size_t>(n / 8, ((size_t) 64 << 20) / sizeof(T));   // (at most 64 MB more: the index arrays come through here too)
		if (hipMalloc(&p, want * sizeof(T)) == hipSuccess) { cap = want; return 0; }
		if (hipMalloc(&p, n * sizeof(T)) != hipSuccess) return -1;
		cap = n;
		return 0;
	}
	void release() { if (p) (void) hipFree(p); p = nullptr; cap = 0; }
};

template <typename T>
struct PinnedBuf {
	T *p = nullptr;
	size_t cap = 0;
	int reserve(size_t n) {
		if (n <= cap) return 0;
		if (p) (void) hipHostFree(p);
		p = nullptr;
		cap = 0;
		const size_t want = n + std::min<size_t>(n / 8, ((size_t) 64 << 20) / sizeof(T));
		if (hipHostMalloc(&p, want * sizeof(T), hipHostMallocDefault) == hipSuccess) { cap = want; return 0; }
		if (hipHostMalloc(&p, n * sizeof(T), hipHostMallocDefault) != hipSuccess) return -1;
		cap = n;
		return 0;
	}
	void release() { if (p) (void) hipHostFree(p); p = nullptr; cap = 0; }
};

}  // namespace ngm

struct ngm_hip_ctx {
	int device = 0;
	ngm_hip_params prm{};
	ngm::SwConst K{};
	ngm::AffConst KA{};
	const ngm::JitKernels *jit = nullptr;  // run-time compiled DP kernels when the corridor has no ahead-of-time build
	int q = 0, c = 0, rl = 0, RW = 0, FW = 0;
	int max_batch = 0;
	hipStream_t stream = nullptr;
	// HBM workspace
	ngm::DevBuf<uint32_t> packed;
	ngm::DevBuf<uint16_t> lens, blk_rows;
	ngm::DevBuf<uint8_t> d_ref, d_qry, d_pair_dir;   // d_pair_dir: the `dir` bytes of the host-pointer entry points (alt_scoring)
	const uint8_t *pair_dir = nullptr;               // ... what the next pack reads: a device array of n bytes, or null (all 0)
	ngm::DevBuf<float> d_scores;
	ngm::DevBuf<uint32_t> dirs;
	ngm::DevBuf<int32_t> d_records;
	ngm::DevBuf<uint16_t> d_runs;
	// pinned staging for the host-pointer entry points
	ngm::PinnedBuf<uint8_t> h_ref, h_qry;
	ngm::PinnedBuf<float> h_scores;
	ngm::PinnedBuf<int32_t> h_records;
	ngm::PinnedBuf<uint16_t> h_runs;
	// profiling
	bool profiling = false;
	hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
	bool ev_valid[3] = {false, false, false};
	std::string error;
};

namespace ngm {
// workspace for n pairs (packed batch, lens, blk_rows); returns 0 or a negative error
int engine_reserve(ngm_hip_ctx *ctx, int n);
// DP over ctx->packed / ctx->lens / ctx->blk_rows (filled by pack_pairs_kernel or gather_pairs_kernel)
int engine_score_packed(ngm_hip_ctx *ctx, int mode, int n, float *d_scores, hipStream_t st);
int engine_align_packed(ngm_hip_ctx *ctx, int mode, int n, int32_t *d_records, uint16_t *d_runs, int run_stride, hipStream_t st);
}  // namespace ngm
