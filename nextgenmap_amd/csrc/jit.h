// jit.h -- DP kernels compiled at run time for corridor widths without an ahead-of-time build (see jit.cpp).
#pragma once

#include <hip/hip_runtime.h>

#include <string>
#include <vector>

namespace ngm {

// fn[kind]: 0/1 linear score local/end-to-end, 2/3 linear align, 4/5 affine score, 6/7 affine align
struct JitKernels {
	hipModule_t module;
	hipFunction_t fn[8];
};

const JitKernels *jit_kernels_for_corridor(int corridor, std::string *err);
int jit_compile_only(int corridor, std::vector<char> *code, std::vector<std::string> *lowered, std::string *err);

}  // namespace ngm
