// ref_runner.cpp -- TEST INFRASTRUCTURE.  Loads a code object built by oracle/build_ref.sh
// (the reference's own OpenCL kernels, unmodified, compiled for gfx950) through the HIP module
// API and drives it the way NGM's OpenCL host does:
//   BatchScore : flatten -> interleaveSeq -> oclSW | oclSW_Global     (lib/mason/opencl/SWOcl.cpp:409-452)
//   BatchAlign : flatten -> interleaveSeq -> oclSW_Score | oclSW_ScoreGlobal -> oclSW_Backtracking
//                                                                  (lib/mason/opencl/SWOclCigar.cpp:52-102)
// so that the C restatement (ngm_oracle.c) and the HIP product kernels can be compared against
// the real reference arithmetic on the MI355X.  Not part of the product.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
	fprintf(stderr, "ref_runner: %s failed: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); return -1; } } while (0)

namespace {
struct Launch {
	hipModule_t mod = nullptr;
	int load(const char *path) { CK(hipModuleLoad(&mod, path)); return 0; }
	int run(const char *name, unsigned global, unsigned local, std::vector<void *> bufs, float *ms) {
		hipFunction_t f;
		CK(hipModuleGetFunction(&f, mod, name));
		std::vector<void *> args;
		for (auto &b : bufs) args.push_back(&b);
		hipEvent_t a, b;
		CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
		CK(hipEventRecord(a, nullptr));
		CK(hipModuleLaunchKernel(f, global / local, 1, 1, local, 1, 1, 0, nullptr, args.data(), nullptr));
		CK(hipEventRecord(b, nullptr));
		CK(hipEventSynchronize(b));
		float t = 0; CK(hipEventElapsedTime(&t, a, b));
		if (ms) *ms += t;
		(void) hipEventDestroy(a); (void) hipEventDestroy(b);
		return 0;
	}
	~Launch() { if (mod) (void) hipModuleUnload(mod); }
};
}  // namespace

// variant 0 = __GPU__ build (256 work-items per group, refs interleaved 256-way),
// variant 1 = __CPU__ build (4 pairs per work-item, local size 1, no interleave),
// variant 2 = __GPU__ build with threads_per_block = interleave_number = 1 (local size 1, flat refs).
// ref_flat: n rows of (q+c) bytes, qry_flat: n rows of q bytes (NUL padded).
// dirs != null: a code object built with -D__ALT_SCORING__ (build_ref.sh ... bs|slam): the score / align kernels take the per-pair
// `direction` bytes as one more argument (SWOcl.cpp:123-137, SWOclCigar.cpp:252-267)
extern "C" int ngm_ref_run_score_alt(const char *co_path, int variant, int mode, int n, const char *ref_flat,
		const char *qry_flat, int q, int c, const char *dirs, float *scores, float *kernel_ms);
extern "C" int ngm_ref_run_score(const char *co_path, int variant, int mode, int n, const char *ref_flat,
		const char *qry_flat, int q, int c, float *scores, float *kernel_ms) {
	return ngm_ref_run_score_alt(co_path, variant, mode, n, ref_flat, qry_flat, q, c, nullptr, scores, kernel_ms);
}
extern "C" int ngm_ref_run_score_alt(const char *co_path, int variant, int mode, int n, const char *ref_flat,
		const char *qry_flat, int q, int c, const char *dirs, float *scores, float *kernel_ms) {
	const int rl = q + c;
	const int group = variant == 0 ? 256 : (variant == 1 ? 4 : 1);
	const int np = (n + group - 1) / group * group;
	std::vector<char> href((size_t) np * rl, 0), hqry((size_t) np * q, 0);
	memcpy(href.data(), ref_flat, (size_t) n * rl);
	memcpy(hqry.data(), qry_flat, (size_t) n * q);
	if (variant == 1) for (int i = n; i < np; ++i) { // SWOcl.cpp:71-80: pad with copies of pair 0
		memcpy(&href[(size_t) i * rl], ref_flat, rl); memcpy(&hqry[(size_t) i * q], qry_flat, q); }
	Launch L;
	if (L.load(co_path)) return -1;
	char *dref, *dref_il, *dqry; float *dres;
	CK(hipMalloc(&dref, href.size())); CK(hipMalloc(&dref_il, href.size()));
	CK(hipMalloc(&dqry, hqry.size())); CK(hipMalloc(&dres, sizeof(float) * np));
	CK(hipMemcpy(dref, href.data(), href.size(), hipMemcpyHostToDevice));
	CK(hipMemcpy(dqry, hqry.data(), hqry.size(), hipMemcpyHostToDevice));
	CK(hipMemset(dres, 0, sizeof(float) * np));
	char *ddir = nullptr;
	if (dirs) {
		std::vector<char> hd(np, 0);
		memcpy(hd.data(), dirs, n);
		CK(hipMalloc(&ddir, np));
		CK(hipMemcpy(ddir, hd.data(), np, hipMemcpyHostToDevice));
	}
	auto with_dir = [&](std::vector<void *> v) { if (ddir) v.push_back(ddir); return v; };
	float ms = 0;
	const char *kname = (mode & 0xFF) == 0 ? "oclSW" : "oclSW_Global";
	if (variant == 0) {
		if (L.run("interleaveSeq", np, 256, {dref, dref_il}, nullptr)) return -1;
		if (L.run(kname, np, 256, with_dir({dref_il, dqry, dres}), &ms)) return -1;
	} else if (variant == 1) {
		if (L.run(kname, np / 4, 1, with_dir({dref, dqry, dres}), &ms)) return -1;
	} else {
		if (L.run(kname, np, 1, with_dir({dref, dqry, dres}), &ms)) return -1;
	}
	if (ddir) (void) hipFree(ddir);
	std::vector<float> h(np);
	CK(hipMemcpy(h.data(), dres, sizeof(float) * np, hipMemcpyDeviceToHost));
	memcpy(scores, h.data(), sizeof(float) * n);
	if (kernel_ms) *kernel_ms = ms;
	(void) hipFree(dref); (void) hipFree(dref_il); (void) hipFree(dqry); (void) hipFree(dres);
	return n;
}

// results4: n x 4 shorts  (ref_position, qstart, qend, alignment_offset after backtracking;
//                          best_read_index, best_ref_index in [0],[1] when backtracking was skipped)
// rle     : n x 2*(2q+c+1) shorts, device buffer pre-filled with 0
extern "C" int ngm_ref_run_align_alt(const char *co_path, int variant, int mode, int n, const char *ref_flat,
		const char *qry_flat, int q, int c, const char *dirs, short *results4, short *rle, float *kernel_ms);
extern "C" int ngm_ref_run_align(const char *co_path, int variant, int mode, int n, const char *ref_flat,
		const char *qry_flat, int q, int c, short *results4, short *rle, float *kernel_ms) {
	return ngm_ref_run_align_alt(co_path, variant, mode, n, ref_flat, qry_flat, q, c, nullptr, results4, rle, kernel_ms);
}
extern "C" int ngm_ref_run_align_alt(const char *co_path, int variant, int mode, int n, const char *ref_flat,
		const char *qry_flat, int q, int c, const char *dirs, short *results4, short *rle, float *kernel_ms) {
	const int rl = q + c;
	const int al = 2 * q + c + 1;
	const int group = variant == 0 ? 256 : (variant == 1 ? 4 : 1);
	const int np = (n + group - 1) / group * group;
	std::vector<char> href((size_t) np * rl, 0), hqry((size_t) np * q, 0);
	memcpy(href.data(), ref_flat, (size_t) n * rl);
	memcpy(hqry.data(), qry_flat, (size_t) n * q);
	if (variant == 1) for (int i = n; i < np; ++i) {
		memcpy(&href[(size_t) i * rl], ref_flat, rl); memcpy(&hqry[(size_t) i * q], qry_flat, q); }
	Launch L;
	if (L.load(co_path)) return -1;
	char *dref, *dref_il, *dqry, *dmat; short *dres, *drle;
	const size_t matrix_bytes = (size_t) np * (c + 2) * (q + 1); // SWOclCigar.cpp:234-236
	CK(hipMalloc(&dref, href.size())); CK(hipMalloc(&dref_il, href.size()));
	CK(hipMalloc(&dqry, hqry.size())); CK(hipMalloc(&dmat, matrix_bytes));
	CK(hipMalloc(&dres, sizeof(short) * 4 * np)); CK(hipMalloc(&drle, sizeof(short) * 2 * al * (size_t) np));
	CK(hipMemcpy(dref, href.data(), href.size(), hipMemcpyHostToDevice));
	CK(hipMemcpy(dqry, hqry.data(), hqry.size(), hipMemcpyHostToDevice));
	CK(hipMemset(dres, 0, sizeof(short) * 4 * np));
	CK(hipMemset(drle, 0, sizeof(short) * 2 * al * (size_t) np));
	CK(hipMemset(dmat, 0, matrix_bytes));
	char *ddir = nullptr;
	if (dirs) {
		std::vector<char> hd(np, 0);
		memcpy(hd.data(), dirs, n);
		CK(hipMalloc(&ddir, np));
		CK(hipMemcpy(ddir, hd.data(), np, hipMemcpyHostToDevice));
	}
	auto with_dir = [&](std::vector<void *> v) { if (ddir) v.push_back(ddir); return v; };
	float ms = 0;
	const char *kname = (mode & 0xFF) == 0 ? "oclSW_Score" : "oclSW_ScoreGlobal";
	if (variant == 0) {
		if (L.run("interleaveSeq", np, 256, {dref, dref_il}, nullptr)) return -1;
		if (L.run(kname, np, 256, with_dir({dref_il, dqry, dres, dmat}), &ms)) return -1;
		if (L.run("oclSW_Backtracking", np, 256, {dref_il, dqry, dres, dmat, drle}, &ms)) return -1;
	} else if (variant == 1) {
		if (L.run(kname, np / 4, 1, with_dir({dref, dqry, dres, dmat}), &ms)) return -1;
		if (L.run("oclSW_Backtracking", np / 4, 1, {dref, dqry, dres, dmat, drle}, &ms)) return -1;
	} else {
		if (L.run(kname, np, 1, with_dir({dref, dqry, dres, dmat}), &ms)) return -1;
		if (L.run("oclSW_Backtracking", np, 1, {dref, dqry, dres, dmat, drle}, &ms)) return -1;
	}
	if (ddir) (void) hipFree(ddir);
	std::vector<short> hres((size_t) 4 * np), hrle((size_t) 2 * al * np);
	CK(hipMemcpy(hres.data(), dres, sizeof(short) * hres.size(), hipMemcpyDeviceToHost));
	CK(hipMemcpy(hrle.data(), drle, sizeof(short) * hrle.size(), hipMemcpyDeviceToHost));
	if (variant != 1) {
		memcpy(results4, hres.data(), sizeof(short) * 4 * (size_t) n);
	} else { // CPU build stores results as 4 groups of short4: [param][k] per work-item (oclSwScore.cl:104-106)
		for (int i = 0; i < n; ++i) for (int p = 0; p < 4; ++p)
			results4[(size_t) i * 4 + p] = hres[(size_t) (i / 4) * 16 + p * 4 + (i % 4)];
	}
	memcpy(rle, hrle.data(), sizeof(short) * 2 * al * (size_t) n);
	if (kernel_ms) *kernel_ms = ms;
	(void) hipFree(dref); (void) hipFree(dref_il); (void) hipFree(dqry); (void) hipFree(dmat); (void) hipFree(dres); (void) hipFree(drle);
	return n;
}
