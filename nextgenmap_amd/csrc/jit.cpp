// jit.cpp -- run-time compilation of the DP kernels for a corridor width that has no ahead-of-time build.
// The band width is a compile-time shape of the kernels (C columns of the band row live in registers), exactly as in
// the reference, whose OpenCL kernels are JIT-compiled with -D corridor_length (lib/mason/opencl/SWOcl.cpp:206-217).
// The common widths are instantiated by hipcc when the library is built; any other width is compiled here with hiprtc
// from the same headers (embedded at build time: csrc/jit_sources.inc) and cached for the life of the process.
#include "jit.h"

#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>

#include <map>
#include <mutex>
#include <vector>

#include "jit_sources.inc"

namespace ngm {
namespace {
std::mutex g_mu;
std::map<int, JitKernels> g_cache;

std::string name_of(int kind, int c) {
	char b[128];
	switch (kind) {
	case 0: snprintf(b, sizeof(b), "ngm::sw_score_kernel<%d, false>", c); break;
	case 1: snprintf(b, sizeof(b), "ngm::sw_score_kernel<%d, true>", c); break;
	case 2: snprintf(b, sizeof(b), "ngm::sw_align_kernel<%d, false>", c); break;
	case 3: snprintf(b, sizeof(b), "ngm::sw_align_kernel<%d, true>", c); break;
	case 4: snprintf(b, sizeof(b), "ngm::sw_affine_kernel<%d, false, false>", c + 1); break;
	case 5: snprintf(b, sizeof(b), "ngm::sw_affine_kernel<%d, true, false>", c + 1); break;
	case 6: snprintf(b, sizeof(b), "ngm::sw_affine_kernel<%d, false, true>", c + 1); break;
	default: snprintf(b, sizeof(b), "ngm::sw_affine_kernel<%d, true, true>", c + 1); break;
	}
	return b;
}
}  // namespace

int jit_compile_only(int corridor, std::vector<char> *code, std::vector<std::string> *lowered, std::string *err) {
	// hiprtc has no <stdint.h>: the fixed-width names the headers use
	std::string src = "typedef unsigned char uint8_t; typedef unsigned short uint16_t; typedef unsigned int uint32_t; typedef int int32_t;\n"
	                  "typedef unsigned long long uint64_t; typedef long long int64_t;\n";
	for (const char *chunk : kJitSourceChunks) src += chunk;
	hiprtcProgram prog;
	if (hiprtcCreateProgram(&prog, src.c_str(), "ngm_dp_kernels.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) { *err = "hiprtcCreateProgram failed"; return -1; }
	std::vector<std::string> names;
	for (int kind = 0; kind < 8; ++kind) {
		names.push_back(name_of(kind, corridor));
		if (hiprtcAddNameExpression(prog, names.back().c_str()) != HIPRTC_SUCCESS) { *err = "hiprtcAddNameExpression failed"; hiprtcDestroyProgram(&prog); return -1; }
	}
	const char *opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17"};
	const hiprtcResult rc = hiprtcCompileProgram(prog, 3, opts);
	if (rc != HIPRTC_SUCCESS) {
		size_t n = 0;
		hiprtcGetProgramLogSize(prog, &n);
		std::string log(n, 0);
		if (n) hiprtcGetProgramLog(prog, &log[0]);
		*err = "hiprtc could not compile the DP kernels for corridor " + std::to_string(corridor) + ": " + log.substr(0, 1500);
		hiprtcDestroyProgram(&prog);
		return -1;
	}
	size_t sz = 0;
	hiprtcGetCodeSize(prog, &sz);
	code->resize(sz);
	hiprtcGetCode(prog, code->data());
	lowered->clear();
	for (const std::string &nm : names) {
		const char *low = nullptr;
		if (hiprtcGetLoweredName(prog, nm.c_str(), &low) != HIPRTC_SUCCESS || !low) { *err = "hiprtcGetLoweredName failed for " + nm; hiprtcDestroyProgram(&prog); return -1; }
		lowered->push_back(low);
	}
	hiprtcDestroyProgram(&prog);
	return 0;
}

const JitKernels *jit_kernels_for_corridor(int corridor, std::string *err) {
	std::lock_guard<std::mutex> lk(g_mu);
	auto it = g_cache.find(corridor);
	if (it != g_cache.end()) return &it->second;
	std::vector<char> code;
	std::vector<std::string> lowered;
	if (jit_compile_only(corridor, &code, &lowered, err)) return nullptr;
	JitKernels k{};
	if (hipModuleLoadData(&k.module, code.data()) != hipSuccess) { *err = "hipModuleLoadData failed for the run-time compiled DP kernels"; return nullptr; }
	for (int kind = 0; kind < 8; ++kind)
		if (hipModuleGetFunction(&k.fn[kind], k.module, lowered[kind].c_str()) != hipSuccess) { *err = "hipModuleGetFunction failed: " + lowered[kind]; return nullptr; }
	return &(g_cache[corridor] = k);
}

}  // namespace ngm

// test hook: compile (no device needed) and report the code object size; < 0 on failure
extern "C" long ngm_hip_jit_selftest(int corridor, char *msg, int msg_len) {
	std::vector<char> code;
	std::vector<std::string> lowered;
	std::string err;
	const int rc = ngm::jit_compile_only(corridor, &code, &lowered, &err);
	if (msg && msg_len > 0) snprintf(msg, msg_len, "%s", rc ? err.c_str() : lowered[0].c_str());
	return rc ? -1 : (long) code.size();
}
