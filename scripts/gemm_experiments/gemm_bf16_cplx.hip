// experiment build of translation unit 2 (complex kernels): includes THIS directory's gemm_bf16_impl.h
#define GEMM_BF16_TU 2
#include "gemm_bf16_impl.h"
