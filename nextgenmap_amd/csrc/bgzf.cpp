// bgzf.cpp -- host side of the GPU BGZF compressor (bgzf_device.h): ngm_bgzf_create / _compress / _destroy (include/ngm_pipeline.h).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/ngm_pipeline.h"
#include "bgzf_device.h"
#include "refindex.h"

#define BGZF_HIP_TRY(expr)                                                                      \
	do {                                                                                        \
		hipError_t e_ = (expr);                                                                 \
		if (e_ != hipSuccess) {                                                                 \
			ngm::pipeline_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
			return -5;                                                                          \
		}                                                                                       \
	} while (0)

struct ngm_bgzf {
	int device = 0;
	hipStream_t st = nullptr;
	uint8_t *d_raw = nullptr, *d_out = nullptr, *d_dense = nullptr, *d_tables = nullptr;
	uint32_t *d_sizes = nullptr;
	unsigned long long *d_offsets = nullptr;
	uint2 *d_scratch = nullptr;
	unsigned long long *d_phases = nullptr;
	size_t raw_cap = 0, blocks_cap = 0, dense_cap = 0;
	int grid = 0;
	uint32_t *h_sizes = nullptr;              // page-locked
	unsigned long long *h_offsets = nullptr;  // page-locked
	size_t h_cap = 0;
	float last_ms = 0.f;
	hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

namespace {
constexpr size_t kTabCrc = 0, kTabXpow = 4096, kTabLen = kTabXpow + 4 * (size_t) (ngm::bgzf::kIn + 1), kTabDist = kTabLen + 256, kTabBytes = kTabDist + 512;

uint32_t mulmod(uint32_t a, uint32_t b) {
	uint32_t p = 0;
	for (int i = 0; i < 32; ++i) {
		if (a & (0x80000000u >> i)) p ^= b;
		b = (b >> 1) ^ ((b & 1u) ? 0xedb88320u : 0u);
	}
	return p;
}
void fill_tables(std::vector<uint8_t> &t) {
	t.assign(kTabBytes, 0);
	uint32_t *crc = (uint32_t *) (t.data() + kTabCrc);
	for (uint32_t i = 0; i < 256; ++i) {
		uint32_t c = i;
		for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ 0xedb88320u : c >> 1;
		crc[i] = c;
	}
	for (uint32_t i = 0; i < 256; ++i) for (int k = 1; k < 4; ++k) crc[k * 256 + i] = crc[crc[(k - 1) * 256 + i] & 255u] ^ (crc[(k - 1) * 256 + i] >> 8);   // slicing-by-4
	uint32_t *xp = (uint32_t *) (t.data() + kTabXpow);
	xp[0] = 0x80000000u;   // x^0
	for (int m = 1; m <= ngm::bgzf::kIn; ++m) xp[m] = mulmod(xp[m - 1], 0x00800000u);   // * x^8
	static const int lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
	static const int dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
	uint8_t *lc = t.data() + kTabLen, *dc = t.data() + kTabDist;
	for (int l = 3; l <= 258; ++l) { int c = 28; while (lbase[c] > l) --c; lc[l - 3] = (uint8_t) c; }
	auto dcode = [&](int d) { int c = 29; while (dbase[c] > d) --c; return (uint8_t) c; };
	for (int d = 0; d < 256; ++d) dc[d] = dcode(d + 1);
	for (int i = 2; i < 256; ++i) dc[256 + i] = dcode((i << 7) + 1);   // distance - 1 = i * 128 + (0..127): one code for all of them from 257 on
}
}  // namespace

extern "C" ngm_bgzf *ngm_bgzf_create(int device) {
	if (hipSetDevice(device) != hipSuccess) { ngm::pipeline_set_error("hipSetDevice(%d) failed", device); return nullptr; }
	ngm_bgzf *z = new ngm_bgzf();
	z->device = device;
	std::vector<uint8_t> t;
	fill_tables(t);
	hipDeviceProp_t prop;
	bool ok = hipGetDeviceProperties(&prop, device) == hipSuccess;
	z->grid = ok ? prop.multiProcessorCount : 256;
	ok = ok && hipStreamCreateWithFlags(&z->st, hipStreamNonBlocking) == hipSuccess;
	ok = ok && hipMalloc(&z->d_tables, kTabBytes) == hipSuccess && hipMemcpy(z->d_tables, t.data(), kTabBytes, hipMemcpyHostToDevice) == hipSuccess;
	ok = ok && hipMalloc(&z->d_scratch, (size_t) z->grid * ngm::bgzf::kSegs * ngm::bgzf::kMatCap * sizeof(uint2)) == hipSuccess;
	if (getenv("NGM_HIP_CS_PHASES")) ok = ok && hipMalloc(&z->d_phases, 64) == hipSuccess;   // (the diagnostics switch of the search kernels: phases of the deflate kernel too)
	ok = ok && hipEventCreate(&z->ev0) == hipSuccess && hipEventCreate(&z->ev1) == hipSuccess;
	ok = ok && hipFuncSetAttribute((const void *) ngm::bgzf::deflate_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int) ngm::bgzf::deflate_lds_bytes()) == hipSuccess;
	if (!ok) { ngm::pipeline_set_error("GPU BGZF compressor: set-up failed on device %d (%s)", device, hipGetErrorString(hipGetLastError())); ngm_bgzf_destroy(z); return nullptr; }
	return z;
}

extern "C" void ngm_bgzf_destroy(ngm_bgzf *z) {
	if (!z) return;
	(void) hipSetDevice(z->device);
	if (z->st) (void) hipStreamSynchronize(z->st);
	(void) hipFree(z->d_raw); (void) hipFree(z->d_out); (void) hipFree(z->d_dense); (void) hipFree(z->d_tables); (void) hipFree(z->d_sizes);
	(void) hipFree(z->d_offsets); (void) hipFree(z->d_scratch); (void) hipFree(z->d_phases);
	if (z->h_sizes) (void) hipHostFree(z->h_sizes);
	if (z->h_offsets) (void) hipHostFree(z->h_offsets);
	if (z->ev0) (void) hipEventDestroy(z->ev0);
	if (z->ev1) (void) hipEventDestroy(z->ev1);
	if (z->st) (void) hipStreamDestroy(z->st);
	delete z;
}

extern "C" size_t ngm_bgzf_bound(size_t n) { return n + 64 * ((n + ngm::bgzf::kIn - 1) / ngm::bgzf::kIn) + 64; }

extern "C" float ngm_bgzf_last_kernel_ms(const ngm_bgzf *z) { return z ? z->last_ms : 0.f; }

static long long compress_on_device(ngm_bgzf *z, const uint8_t *d_raw, size_t n, void *out, size_t out_cap);

extern "C" long long ngm_bgzf_compress(ngm_bgzf *z, const void *raw, size_t n, void *out, size_t out_cap) {
	if (!z || (!raw && n) || !out) { ngm::pipeline_set_error("ngm_bgzf_compress: bad arguments"); return -22; }
	if (n == 0) return 0;
	BGZF_HIP_TRY(hipSetDevice(z->device));
	if (n + 16 > z->raw_cap) {
		(void) hipFree(z->d_raw); z->d_raw = nullptr; z->raw_cap = 0;
		const size_t cap = n + n / 4 + 4096;
		BGZF_HIP_TRY(hipMalloc(&z->d_raw, cap));
		z->raw_cap = cap;
	}
	BGZF_HIP_TRY(hipMemcpyAsync(z->d_raw, raw, n, hipMemcpyHostToDevice, z->st));
	return compress_on_device(z, z->d_raw, n, out, out_cap);
}

// the same for bytes that already are in this device's memory (complete: the caller has synchronised the stream that wrote them)
extern "C" long long ngm_bgzf_compress_device(ngm_bgzf *z, const void *d_raw, size_t n, void *out, size_t out_cap) {
	if (!z || (!d_raw && n) || !out) { ngm::pipeline_set_error("ngm_bgzf_compress_device: bad arguments"); return -22; }
	if (n == 0) return 0;
	BGZF_HIP_TRY(hipSetDevice(z->device));
	return compress_on_device(z, (const uint8_t *) d_raw, n, out, out_cap);
}

static long long compress_on_device(ngm_bgzf *z, const uint8_t *d_raw, size_t n, void *out, size_t out_cap) {
	if (out_cap < ngm_bgzf_bound(n)) { ngm::pipeline_set_error("ngm_bgzf_compress: the output buffer holds %zu bytes, %zu may be needed", out_cap, ngm_bgzf_bound(n)); return -22; }
	const size_t nb = (n + ngm::bgzf::kIn - 1) / ngm::bgzf::kIn;
	if (nb > 0x7fffffffull) { ngm::pipeline_set_error("ngm_bgzf_compress: too much input for one call"); return -22; }
	if (nb > z->blocks_cap) {
		(void) hipFree(z->d_out); (void) hipFree(z->d_sizes); (void) hipFree(z->d_offsets);
		z->d_out = nullptr; z->d_sizes = nullptr; z->d_offsets = nullptr; z->blocks_cap = 0;
		const size_t cap = nb + nb / 4 + 16;
		BGZF_HIP_TRY(hipMalloc(&z->d_out, cap * ngm::bgzf::kStride));
		BGZF_HIP_TRY(hipMalloc(&z->d_sizes, cap * 4));
		BGZF_HIP_TRY(hipMalloc(&z->d_offsets, cap * 8));
		z->blocks_cap = cap;
	}
	if (nb > z->h_cap) {
		if (z->h_sizes) (void) hipHostFree(z->h_sizes);
		if (z->h_offsets) (void) hipHostFree(z->h_offsets);
		z->h_sizes = nullptr; z->h_offsets = nullptr; z->h_cap = 0;
		const size_t cap = nb + nb / 4 + 16;
		BGZF_HIP_TRY(hipHostMalloc(&z->h_sizes, cap * 4, hipHostMallocDefault));
		BGZF_HIP_TRY(hipHostMalloc(&z->h_offsets, cap * 8, hipHostMallocDefault));
		z->h_cap = cap;
	}
	ngm::bgzf::Args A{};
	A.raw = d_raw; A.n = n; A.n_blocks = (int) nb; A.out = z->d_out; A.sizes = z->d_sizes; A.scratch = z->d_scratch;
	A.crc_table = (const uint32_t *) (z->d_tables + kTabCrc); A.xpow = (const uint32_t *) (z->d_tables + kTabXpow);
	A.len_code = z->d_tables + kTabLen; A.dist_code = z->d_tables + kTabDist;
	A.phase_cycles = z->d_phases;
	if (z->d_phases) BGZF_HIP_TRY(hipMemsetAsync(z->d_phases, 0, 64, z->st));
	BGZF_HIP_TRY(hipEventRecord(z->ev0, z->st));
	hipLaunchKernelGGL(ngm::bgzf::deflate_kernel, dim3((unsigned) std::min<size_t>(nb, (size_t) z->grid)), dim3(ngm::bgzf::kNT), ngm::bgzf::deflate_lds_bytes(), z->st, A);
	BGZF_HIP_TRY(hipGetLastError());
	BGZF_HIP_TRY(hipEventRecord(z->ev1, z->st));
	BGZF_HIP_TRY(hipMemcpyAsync(z->h_sizes, z->d_sizes, nb * 4, hipMemcpyDeviceToHost, z->st));
	BGZF_HIP_TRY(hipStreamSynchronize(z->st));
	unsigned long long total = 0;
	for (size_t b = 0; b < nb; ++b) {
		if (z->h_sizes[b] < 26u + 2u || z->h_sizes[b] > (uint32_t) ngm::bgzf::kStride) { ngm::pipeline_set_error("GPU BGZF compressor: block %zu has %u bytes", b, z->h_sizes[b]); return -5; }
		z->h_offsets[b] = total;
		total += z->h_sizes[b];
	}
	if (total > out_cap) { ngm::pipeline_set_error("GPU BGZF compressor: %llu bytes for a buffer of %zu", total, out_cap); return -5; }
	if (total + 16 > z->dense_cap) {
		(void) hipFree(z->d_dense); z->d_dense = nullptr; z->dense_cap = 0;
		const size_t cap = (size_t) total + (size_t) total / 4 + 4096;
		BGZF_HIP_TRY(hipMalloc(&z->d_dense, cap));
		z->dense_cap = cap;
	}
	BGZF_HIP_TRY(hipMemcpyAsync(z->d_offsets, z->h_offsets, nb * 8, hipMemcpyHostToDevice, z->st));
	hipLaunchKernelGGL(ngm::bgzf::gather_kernel, dim3((unsigned) std::min<size_t>(nb, (size_t) z->grid * 8)), dim3(256), 0, z->st, (const uint8_t *) z->d_out, (const uint32_t *) z->d_sizes,
			(const unsigned long long *) z->d_offsets, z->d_dense, (int) nb);
	BGZF_HIP_TRY(hipGetLastError());
	BGZF_HIP_TRY(hipMemcpyAsync(out, z->d_dense, (size_t) total, hipMemcpyDeviceToHost, z->st));
	BGZF_HIP_TRY(hipStreamSynchronize(z->st));
	if (z->d_phases) {
		unsigned long long ph[8];
		BGZF_HIP_TRY(hipMemcpy(ph, z->d_phases, 64, hipMemcpyDeviceToHost));
		fprintf(stderr, "[ngm-hip] BGZF kernel, us of thread 0 per block (100 MHz clock): load %.1f | matching %.1f | crc + histograms %.1f | code lengths + codes %.1f | header + match scan %.1f | token bits %.1f | emit + copy out %.1f\n",
				ph[0] / 100.0 / nb, ph[1] / 100.0 / nb, ph[2] / 100.0 / nb, ph[3] / 100.0 / nb, ph[4] / 100.0 / nb, ph[5] / 100.0 / nb, ph[6] / 100.0 / nb);
	}
	float ms = 0.f;
	if (hipEventElapsedTime(&ms, z->ev0, z->ev1) == hipSuccess) z->last_ms = ms;
	return (long long) total;
}
