// Complex (4M) bf16 MFMA GEMM: translation unit 2 of gemm_bf16_impl.h.
#define GEMM_BF16_TU 2
#include "gemm_bf16_impl.h"
