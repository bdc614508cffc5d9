// experiment build of translation unit 1 (real kernels, Gauss 3M, split-K): includes THIS directory's gemm_bf16_impl.h
#define GEMM_BF16_TU 1
#include "gemm_bf16_impl.h"
