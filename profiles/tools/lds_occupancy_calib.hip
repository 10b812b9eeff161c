// How many single-wave workgroups run at once as a function of their dynamic LDS size?  (MI355X: 160 KB of LDS per CU.)
// Every workgroup touches its LDS and then waits ~100 us on the wall clock; concurrency = blocks * 100 us / kernel time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ __launch_bounds__(64) void spin_kernel(unsigned *out, unsigned lds_words, unsigned long long ticks) {
	extern __shared__ unsigned lds[];
	for (unsigned i = threadIdx.x; i < lds_words; i += 64) lds[i] = i;
	__syncthreads();
	const unsigned long long t0 = wall_clock64();
	while (wall_clock64() - t0 < ticks) {}
	if (threadIdx.x == 0) out[blockIdx.x] = lds[(blockIdx.x * 7u) % lds_words];
}
int main() {
	unsigned *d_out;
	const int blocks = 8192;
	hipMalloc(&d_out, blocks * 4);
	hipFuncSetAttribute((const void *) spin_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	const int kb[] = {4, 15, 30, 60, 64, 65, 80, 88, 128, 150};
	for (int threads : {64, 256}) for (int k : kb) {
		const size_t bytes = (size_t) k * 1024;
		for (int rep = 0; rep < 2; ++rep) {
			hipEventRecord(e0);
			hipLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(threads), bytes, 0, d_out, (unsigned) (bytes / 4), 10000ull /* 100 us at 100 MHz */);
			hipEventRecord(e1);
			hipEventSynchronize(e1);
			float ms = 0; hipEventElapsedTime(&ms, e0, e1);
			if (rep == 1) printf("threads %3d  LDS %3d KB: %8.3f ms  -> %6.1f workgroups in flight (%s)\n", threads, k, ms, blocks * 0.1 / ms, hipGetErrorString(hipGetLastError()));
		}
	}
	return 0;
}
