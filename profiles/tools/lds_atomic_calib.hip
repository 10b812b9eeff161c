// lds_atomic_calib.hip -- what an LDS atomic costs on MI355X when its 64 addresses are random (the vote of candidate search):
// LDS-array cycles per wave instruction for ds_or_rtn_b32 / ds_or_b32 / ds_read_b32, all lanes vs half the lanes active,
// independent vs dependent, at the occupancy the search kernel runs at.
//   hipcc --offload-arch=gfx950 -O3 lds_atomic_calib.hip -o lds_atomic_calib && ./lds_atomic_calib
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t xs(uint32_t &s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }

// MODE 0 or_rtn all lanes | 1 or_rtn half of the lanes (exec mask) | 2 or (no return) | 3 read | 4 or_rtn, address depends on the result
template <int MODE, int ILP>
__global__ __launch_bounds__(64) void lds_kernel(int iters, uint32_t words_mask, uint32_t *out) {
	extern __shared__ uint32_t lds[];
	for (uint32_t i = threadIdx.x; i <= words_mask; i += 64) lds[i] = 0;
	__syncthreads();
	uint32_t s = (blockIdx.x * 64 + threadIdx.x) * 2654435761u + 12345u, acc = 0;
	for (int it = 0; it < iters; it += ILP) {
		uint32_t r[ILP], v[ILP];
#pragma unroll
		for (int u = 0; u < ILP; ++u) r[u] = xs(s) ^ (MODE == 4 ? acc : 0u);
#pragma unroll
		for (int u = 0; u < ILP; ++u) {
			uint32_t *p = &lds[(r[u] >> 5) & words_mask];
			const uint32_t m = 1u << (r[u] & 31);
			v[u] = 0;
			if (MODE == 0 || MODE == 4) v[u] = atomicOr(p, m);
			else if (MODE == 1) { if (r[u] & 0x80000000u) v[u] = atomicOr(p, m); }
			else if (MODE == 2) __hip_atomic_fetch_or(p, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			else v[u] = *(volatile uint32_t *) p;
		}
#pragma unroll
		for (int u = 0; u < ILP; ++u) acc += v[u];
	}
	if (acc == 0x12345u) out[0] = acc;
}

template <int MODE, int ILP>
static void run(const char *label, int lds_kb, uint32_t *out) {
	const int blocks = 256 * 64, iters = 4096;
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	CHECK(hipFuncSetAttribute((const void *) lds_kernel<MODE, ILP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_kb * 1024));
	float best = 1e30f;
	for (int rep = 0; rep < 3; ++rep) {
		CHECK(hipEventRecord(e0));
		hipLaunchKernelGGL((lds_kernel<MODE, ILP>), dim3(blocks), dim3(64), lds_kb * 1024, 0, iters, 2047u, out);
		CHECK(hipEventRecord(e1));
		CHECK(hipEventSynchronize(e1));
		float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
		if (ms < best) best = ms;
	}
	// CU-cycles per wave instruction at 2.4 GHz, 256 CUs
	const double cyc = best * 1e-3 * 2.4e9 * 256.0 / ((double) blocks * iters);
	printf("%-34s ILP %d  %2d KB LDS/wave (%2d waves/CU)  %7.3f ms  %6.2f CU-cycles per wave instruction\n", label, ILP, lds_kb, 160 / lds_kb, best, cyc);
}

int main() {
	uint32_t *out; CHECK(hipMalloc(&out, 64));
	for (int kb : {20, 10, 5}) {
		run<0, 4>("ds_or_rtn_b32, 64 random lanes", kb, out);
		run<1, 4>("ds_or_rtn_b32, ~32 random lanes", kb, out);
		run<2, 4>("ds_or_b32 (no return), 64 lanes", kb, out);
		run<3, 4>("ds_read_b32, 64 random lanes", kb, out);
		run<4, 1>("ds_or_rtn_b32, dependent chain", kb, out);
		run<0, 1>("ds_or_rtn_b32, 64 lanes, no ILP", kb, out);
		run<0, 8>("ds_or_rtn_b32, 64 lanes", kb, out);
	}
	return 0;
}
