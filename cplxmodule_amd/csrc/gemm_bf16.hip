// Real bf16 MFMA GEMM (256x256 tile), the Gauss 3M driver and the split-K helpers: translation unit 1
// of gemm_bf16_impl.h (the complex kernels are instantiated in gemm_bf16_cplx.hip so that the two
// halves compile in parallel).
#define GEMM_BF16_TU 1
#include "gemm_bf16_impl.h"
