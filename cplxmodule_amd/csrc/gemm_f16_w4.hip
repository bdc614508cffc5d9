// One-wave-per-SIMD GEMM family on IEEE-half operands, float32 output: gemm_bf16_w4.hip compiled with the half matrix
// instruction (gemm.h: CPLXAMD_MFMA16) in its own namespace (see gemm_f16.hip).
#define CPLXAMD_GEMM_F16 1
#define w4 w4h
#define launch_gemm_bf16_w4 launch_gemm_f16_w4
#define launch_gemm_bf16 launch_gemm_f16
#include "gemm_bf16_w4.hip"
