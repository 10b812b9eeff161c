// valu_rate_calib.hip -- issue cost of the VALU / LDS instructions the candidate search and the DP kernels are made of, on gfx950.
// One workgroup of 256 threads per CU x WPS (waves per SIMD), every wave runs N dependent-free chains (ILP 4) of one instruction;
// reported: SIMD cycles per wave instruction = elapsed cycles * resident waves per SIMD / instructions per wave.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o valu_rate_calib profiles/tools/valu_rate_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kIters = 4096, kUnroll = 16;  // instructions per wave = kIters * kUnroll * 4 chains

#define BODY4(INS) \
	asm volatile(INS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));

#define KERNEL(NAME, INS) \
__global__ __launch_bounds__(256) void NAME(uint32_t *out, uint32_t b, uint32_t c, unsigned long long *cyc) { \
	__shared__ uint32_t lds[4096]; \
	for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i; \
	__syncthreads(); \
	uint32_t a0 = threadIdx.x * 4u, a1 = a0 + 1024u, a2 = a0 + 2048u, a3 = a0 + 3072u; \
	(void) lds; \
	const unsigned long long t0 = __builtin_readcyclecounter(); \
	for (int it = 0; it < kIters; ++it) { \
		_Pragma("unroll") for (int u = 0; u < kUnroll; ++u) { BODY4(INS) } \
	} \
	const unsigned long long t1 = __builtin_readcyclecounter(); \
	out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3; \
	if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0; \
}

KERNEL(k_add, "v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4")
KERNEL(k_mul_lo, "v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4")
KERNEL(k_mul_hi, "v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4")
KERNEL(k_mul_u24, "v_mul_u32_u24 %0, %0, %4\n v_mul_u32_u24 %1, %1, %4\n v_mul_u32_u24 %2, %2, %4\n v_mul_u32_u24 %3, %3, %4")
KERNEL(k_mad_u24, "v_mad_u32_u24 %0, %0, %4, %5\n v_mad_u32_u24 %1, %1, %4, %5\n v_mad_u32_u24 %2, %2, %4, %5\n v_mad_u32_u24 %3, %3, %4, %5")
KERNEL(k_perm, "v_perm_b32 %0, %0, %4, %5\n v_perm_b32 %1, %1, %4, %5\n v_perm_b32 %2, %2, %4, %5\n v_perm_b32 %3, %3, %4, %5")
KERNEL(k_pk_add, "v_pk_add_u16 %0, %0, %4\n v_pk_add_u16 %1, %1, %4\n v_pk_add_u16 %2, %2, %4\n v_pk_add_u16 %3, %3, %4")
KERNEL(k_pk_max, "v_pk_max_i16 %0, %0, %4\n v_pk_max_i16 %1, %1, %4\n v_pk_max_i16 %2, %2, %4\n v_pk_max_i16 %3, %3, %4")
KERNEL(k_pk_mul, "v_pk_mul_lo_u16 %0, %0, %4\n v_pk_mul_lo_u16 %1, %1, %4\n v_pk_mul_lo_u16 %2, %2, %4\n v_pk_mul_lo_u16 %3, %3, %4")
KERNEL(k_pk_mad, "v_pk_mad_u16 %0, %0, %4, %5\n v_pk_mad_u16 %1, %1, %4, %5\n v_pk_mad_u16 %2, %2, %4, %5\n v_pk_mad_u16 %3, %3, %4, %5")
KERNEL(k_max3, "v_max3_i32 %0, %0, %4, %5\n v_max3_i32 %1, %1, %4, %5\n v_max3_i32 %2, %2, %4, %5\n v_max3_i32 %3, %3, %4, %5")
KERNEL(k_lshl_or, "v_lshl_or_b32 %0, %0, 1, %5\n v_lshl_or_b32 %1, %1, 1, %5\n v_lshl_or_b32 %2, %2, 1, %5\n v_lshl_or_b32 %3, %3, 1, %5")
KERNEL(k_dpp, "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf")
// LDS: addresses stay inside the 16 KB array (and 0x3FFC); returning atomics wait for their result (lgkmcnt) once per four
KERNEL(k_ds_or_rtn, "v_and_b32 %0, 0x3ffc, %0\n v_and_b32 %1, 0x3ffc, %1\n v_and_b32 %2, 0x3ffc, %2\n v_and_b32 %3, 0x3ffc, %3\n ds_or_rtn_b32 %0, %0, %4\n ds_or_rtn_b32 %1, %1, %4\n ds_or_rtn_b32 %2, %2, %4\n ds_or_rtn_b32 %3, %3, %4\n s_waitcnt lgkmcnt(0)")
KERNEL(k_ds_read, "v_and_b32 %0, 0x3ffc, %0\n v_and_b32 %1, 0x3ffc, %1\n v_and_b32 %2, 0x3ffc, %2\n v_and_b32 %3, 0x3ffc, %3\n ds_read_b32 %0, %0\n ds_read_b32 %1, %1\n ds_read_b32 %2, %2\n ds_read_b32 %3, %3\n s_waitcnt lgkmcnt(0)")
KERNEL(k_ds_add, "v_and_b32 %0, 0x3ffc, %0\n v_and_b32 %1, 0x3ffc, %1\n v_and_b32 %2, 0x3ffc, %2\n v_and_b32 %3, 0x3ffc, %3\n ds_add_u32 %0, %4\n ds_add_u32 %1, %4\n ds_add_u32 %2, %4\n ds_add_u32 %3, %4\n v_add_u32 %0, %0, %5\n v_add_u32 %1, %1, %5\n v_add_u32 %2, %2, %5\n v_add_u32 %3, %3, %5\n s_waitcnt lgkmcnt(0)")

template <typename K>
void run(const char *name, K kern, int cus, int wps, double extra_valu) {
	uint32_t *out; unsigned long long *cyc;
	const int blocks = cus * wps;
	CK(hipMalloc(&out, (size_t) blocks * 256 * 4)); CK(hipMalloc(&cyc, (size_t) blocks * 8));
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 0x9E3779B1u, 12345u, cyc);
	CK(hipEventRecord(e0, 0));
	hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 0x9E3779B1u, 12345u, cyc);
	CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
	float ms; CK(hipEventElapsedTime(&ms, e0, e1));
	const double instr = (double) kIters * kUnroll * 4;   // of the measured kind, per wave
	int clk_khz = 0; CK(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0));
	const double simd_cycles = ms * 1e-3 * clk_khz * 1e3;
	printf("%-14s %d wave(s)/SIMD: %.3f ms, %.2f SIMD cycles per wave instruction at the %d MHz the API reports (%.2f if the %s VALU ops next to it cost 4 cycles each)\n",
			name, wps, ms, simd_cycles / (instr * wps), clk_khz / 1000, (simd_cycles / wps - extra_valu * instr * 4) / instr, extra_valu ? "address" : "no");
	CK(hipFree(out)); CK(hipFree(cyc));
}

int main() {
	hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
	const int cus = pr.multiProcessorCount;
	printf("%s, %d CUs\n", pr.gcnArchName, cus);
	for (int wps : {1, 2, 4}) {
		run("v_add_u32", k_add, cus, wps, 0);
		run("v_mul_lo_u32", k_mul_lo, cus, wps, 0);
		run("v_mul_hi_u32", k_mul_hi, cus, wps, 0);
		run("v_mul_u32_u24", k_mul_u24, cus, wps, 0);
		run("v_mad_u32_u24", k_mad_u24, cus, wps, 0);
		run("v_perm_b32", k_perm, cus, wps, 0);
		run("v_pk_add_u16", k_pk_add, cus, wps, 0);
		run("v_pk_max_i16", k_pk_max, cus, wps, 0);
		run("v_pk_mul_lo_u16", k_pk_mul, cus, wps, 0);
		run("v_pk_mad_u16", k_pk_mad, cus, wps, 0);
		run("v_max3_i32", k_max3, cus, wps, 0);
		run("v_lshl_or_b32", k_lshl_or, cus, wps, 0);
		run("v_mov_dpp", k_dpp, cus, wps, 0);
		run("ds_or_rtn_b32", k_ds_or_rtn, cus, wps, 1);
		run("ds_read_b32", k_ds_read, cus, wps, 1);
		run("ds_add_u32", k_ds_add, cus, wps, 2);
	}
	return 0;
}
