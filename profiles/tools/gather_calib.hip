// gather_calib.hip -- what the MI355X memory system delivers for RANDOM gathers of G contiguous bytes out of a buffer
// far larger than the 256 MiB Infinity Cache, and what rocprofv3's FETCH_SIZE / TCC_EA0_RDREQ report for them.
// This is the access pattern of candidate search (k-mer index entry -> position list); the numbers size the index
// buckets (csrc/refindex.cpp) and calibrate `roofline.traffic` (bench.py) for gathers, for which the guide's
// "FETCH_SIZE counts half" rule (coalesced streaming reads) is not established.
//
//   hipcc --offload-arch=gfx950 -O3 gather_calib.hip -o gather_calib
//   ./gather_calib [buffer GiB = 8] [chained = 0]
// prints one line per granule: bytes/item, items, ms, G items/s, useful GB/s.  Under
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace -- ./gather_calib     (and TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum)
// the per-dispatch counter values divide by the known item counts printed here.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {  // splitmix64
	x += 0x9E3779B97F4A7C15ull;
	x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
	x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
	return x ^ (x >> 31);
}

struct __attribute__((aligned(16))) U4 { uint32_t x, y, z, w; };

// G bytes per item, LPI = lanes that share an item (each loads G / LPI bytes: 4, 8 or 16), UNR independent items in
// flight per lane group.  CHAINED: the address of item j+1 depends on the data of item j (index entry -> list).
template <int G, int UNR, bool CHAINED>
__global__ __launch_bounds__(256) void gather_kernel(const uint8_t *__restrict__ buf, uint64_t n_slots, uint64_t n_items, uint32_t *out, uint64_t seed) {
	constexpr int W = G >= 16 ? 16 : G;           // bytes per lane
	constexpr int LPI = G / W;
	const uint64_t gid = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
	const uint64_t group = gid / LPI, sub = gid % LPI;
	const uint64_t n_groups = ((uint64_t) gridDim.x * blockDim.x) / LPI;
	uint32_t acc = 0;
	for (uint64_t it0 = group * UNR; it0 < n_items; it0 += n_groups * UNR) {
		U4 v[UNR];
		uint64_t carry = 0;
#pragma unroll
		for (int u = 0; u < UNR; ++u) {
			const uint64_t slot = mix((it0 + u) ^ seed ^ (CHAINED ? carry : 0)) % n_slots;
			const uint8_t *p = buf + slot * G + sub * W;
			if (W == 16) v[u] = *reinterpret_cast<const U4 *>(p);
			else if (W == 8) { const uint2 t = *reinterpret_cast<const uint2 *>(p); v[u] = U4{t.x, t.y, 0, 0}; }
			else v[u] = U4{*reinterpret_cast<const uint32_t *>(p), 0, 0, 0};
			if (CHAINED) carry = v[u].x;
		}
#pragma unroll
		for (int u = 0; u < UNR; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
	}
	if (acc == 0x12345u) out[0] = acc;  // keeps the loads alive
}

template <int G, bool CHAINED>
static void run(const uint8_t *buf, uint64_t buf_bytes, uint32_t *out, uint64_t n_items, const char *label) {
	constexpr int UNR = CHAINED ? 2 : 8;
	const uint64_t n_slots = buf_bytes / G;
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	const int blocks = 256 * 8 * 2;  // 2 x 8 waves-of-256-threads per CU
	float best = 1e30f;
	for (int rep = 0; rep < 3; ++rep) {
		CHECK(hipEventRecord(e0));
		hipLaunchKernelGGL((gather_kernel<G, UNR, CHAINED>), dim3(blocks), dim3(256), 0, 0, buf, n_slots, n_items, out, 0x5EEDull + rep);
		CHECK(hipEventRecord(e1));
		CHECK(hipEventSynchronize(e1));
		float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
		if (ms < best) best = ms;
	}
	printf("%-8s G=%4d B  items=%11llu  ms=%8.3f  Gitems/s=%7.2f  useful_GB/s=%8.1f\n", label, G, (unsigned long long) n_items, best,
			n_items / best / 1e6, (double) n_items * G / best / 1e6);
}


// ---- wave-per-read bucket access patterns (one 64-lane workgroup per "read", K = 276 random 128-byte buckets each) ----
// mode 0 "blind"  : 8 lanes x 16 B per bucket, every bucket read once in full (35 rounds of 8 buckets)
// mode 1 "touch+reread": 2 lanes per bucket touch dwords 0 and 16 (pulls the whole line with one request), then a lane per
//                   32-byte segment re-reads segments 0..nseg-1 (nseg from the first dword, here: hash -> 2.4 on average)
// mode 2 "header+segs": 1 lane per bucket reads dword 0 only (64-byte request), then the segments as in mode 1
//                   (the second half of ~48 % of the buckets is a second request)
template <int MODE>
__global__ __launch_bounds__(64) void bucket_kernel(const uint8_t *__restrict__ buf, uint64_t n_buckets, int K, uint32_t *out, uint64_t seed) {
	extern __shared__ uint32_t lds[];   // sized by the host to bound the workgroups per CU like the real kernel (17 KB -> 9)
	const int lane = threadIdx.x;
	const uint64_t read = blockIdx.x;
	uint32_t acc = 0;
	auto bucket_of = [&](int j) -> const uint8_t * { return buf + (mix((read * 1024 + (uint64_t) j) ^ seed) % n_buckets) * 128; };
	if (MODE == 0) {
		for (int j0 = 0; j0 < K; j0 += 32) {   // 4 rounds of 8 buckets in flight
			U4 v[4];
#pragma unroll
			for (int u = 0; u < 4; ++u) {
				const int j = j0 + u * 8 + (lane >> 3);
				v[u] = U4{0, 0, 0, 0};
				if (j < K) v[u] = *reinterpret_cast<const U4 *>(bucket_of(j) + (lane & 7) * 16);
			}
#pragma unroll
			for (int u = 0; u < 4; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
		}
	} else {
		// phase A: counts
		for (int j0 = 0; j0 < K; j0 += (MODE == 1 ? 32 : 64)) {
			const int j = j0 + (MODE == 1 ? (lane >> 1) : lane);
			if (j < K) {
				const uint32_t w = *reinterpret_cast<const uint32_t *>(bucket_of(j) + (MODE == 1 ? (lane & 1) * 64 : 0));
				if (MODE == 2 || (lane & 1) == 0) lds[j] = w;
			}
		}
		__syncthreads();
		// phase B: 32-byte segments; bucket j has 1..4 segments: 1 + (hash % 100 < 95) + (hash % 100 < 48) + (hash % 100 < 5)
		for (int i0 = 0; i0 < K * 4; i0 += 128) {
			U4 v[4];
#pragma unroll
			for (int u = 0; u < 2; ++u) {
				const int i = i0 + u * 64 + lane;
				const int j = i >> 2, sg = i & 3;
				v[2 * u] = U4{0, 0, 0, 0}; v[2 * u + 1] = U4{0, 0, 0, 0};
				if (j < K) {
					const uint32_t h = (uint32_t) ((mix(read * 1024 + j) + (lds[j] == 0x77777777u ? 1u : 0u)) % 100);  // real dependency on phase A
					const int nseg = 1 + (h < 95) + (h < 48) + (h < 5);
					if (sg < nseg) {
						const U4 *src = reinterpret_cast<const U4 *>(bucket_of(j) + sg * 32);
						v[2 * u] = src[0]; v[2 * u + 1] = src[1];
					}
				}
			}
#pragma unroll
			for (int u = 0; u < 4; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
		}
	}
	if (acc == 0x12345u) out[0] = acc;
}

template <int MODE>
static void run_bucket(const uint8_t *buf, uint64_t buf_bytes, uint32_t *out, const char *label) {
	const int n_reads = 524288, K = 276;
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	CHECK(hipFuncSetAttribute((const void *) bucket_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 17 * 1024));
	float best = 1e30f;
	for (int rep = 0; rep < 3; ++rep) {
		CHECK(hipEventRecord(e0));
		hipLaunchKernelGGL((bucket_kernel<MODE>), dim3(n_reads), dim3(64), 17 * 1024, 0, buf, buf_bytes / 128, K, out, 0xB0C4ull + rep);
		CHECK(hipEventRecord(e1));
		CHECK(hipEventSynchronize(e1));
		float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
		if (ms < best) best = ms;
	}
	printf("bucket %-14s reads=%d K=%d  ms=%8.3f  Gbuckets/s=%7.2f\n", label, n_reads, K, best, (double) n_reads * K / best / 1e6);
}

int main(int argc, char **argv) {
	const double gib = argc > 1 ? atof(argv[1]) : 8.0;
	const uint64_t buf_bytes = (uint64_t) (gib * 1024.0 * 1024.0 * 1024.0) / 4096 * 4096;
	uint8_t *buf; uint32_t *out;
	CHECK(hipMalloc(&buf, buf_bytes)); CHECK(hipMalloc(&out, 64));
	CHECK(hipMemset(buf, 0x5A, buf_bytes));
	CHECK(hipDeviceSynchronize());
	printf("buffer %.1f GiB\n", gib);
	const uint64_t N = 1ull << 28;
	run<4, false>(buf, buf_bytes, out, N, "indep");
	run<8, false>(buf, buf_bytes, out, N, "indep");
	run<16, false>(buf, buf_bytes, out, N, "indep");
	run<32, false>(buf, buf_bytes, out, N / 2, "indep");
	run<64, false>(buf, buf_bytes, out, N / 2, "indep");
	run<128, false>(buf, buf_bytes, out, N / 4, "indep");
	run<256, false>(buf, buf_bytes, out, N / 8, "indep");
	run<512, false>(buf, buf_bytes, out, N / 16, "indep");
	run<8, true>(buf, buf_bytes, out, N / 4, "chained");
	run<64, true>(buf, buf_bytes, out, N / 4, "chained");
	run<128, true>(buf, buf_bytes, out, N / 8, "chained");
	run_bucket<0>(buf, buf_bytes, out, "blind");
	run_bucket<1>(buf, buf_bytes, out, "touch+reread");
	run_bucket<2>(buf, buf_bytes, out, "header+segs");
	return 0;
}
