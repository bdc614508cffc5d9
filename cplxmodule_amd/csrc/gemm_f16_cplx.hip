// Complex (4M) MFMA GEMM on IEEE-half operands, float32 output (see gemm_f16.hip).
#define CPLXAMD_GEMM_F16 1
#define launch_gemm_bf16 launch_gemm_f16
#define launch_gemm_bf16_w4 launch_gemm_f16_w4
#define gemm_bf16_kernel gemm_f16_kernel
#define gemm_bf16_persist_kernel gemm_f16_persist_kernel
#define GEMM_BF16_TU 2
#include "gemm_bf16_impl.h"
