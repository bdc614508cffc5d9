// Generic complex / real GEMM on the exact-f32 matrix cores (v_mfma_f32_32x32x2_f32):
//   C[m,n] = sum_k A[m,k] * op(B[n,k]) (+ bias[n]),   any element strides, any M, N, K,
//   float32 or bf16 inputs (bf16 is widened exactly), float32 accumulation.
// This is the parity path (fp32 results == an fmaf chain in k order) and the fallback for
// shapes / layouts the bf16 fast path (gemm_bf16.hip) does not take.
//
// Reference arithmetic: cplx.linear_naive cplx.py:634-648, Cplx.__matmul__ cplx.py:167-174,
// F.linear in the LRT variance term nn/relevance/complex/base.py:50-54.
//
// Tiling: 64x64 outputs per 256-thread block (4 waves as 2x2, one 32x32 MFMA tile each),
// BK = 16, operands staged k-major in LDS so that every ds_read_b32 is conflict-free.
#include "gemm.h"

namespace cplxamd {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int GBM = 64, GBN = 64, GBK = 16, GLD = GBM + 1;

template <typename T>
__device__ __forceinline__ float ldg(const void* p, int64_t off) {
  return io<T>::ld(reinterpret_cast<const T*>(p) + off);
}

// stage one [rows x GBK] operand tile (rows = 64) into LDS as dst[k][row]
template <typename TIN>
__device__ __forceinline__ void stage_tile(float (*dst)[GLD], const void* src, int64_t rs,
                                           int64_t cs, int row0, int k0, int rows, int K) {
  const int t = threadIdx.x;
  if (cs == 1 || rs != 1) {  // k fastest across threads
    const int k = t & 15, rb = t >> 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = rb + 16 * j;
      const int gr = row0 + r, gk = k0 + k;
      float v = 0.0f;
      if (gr < rows && gk < K) v = ldg<TIN>(src, (int64_t)gr * rs + (int64_t)gk * cs);
      dst[k][r] = v;
    }
  } else {  // rows fastest across threads
    const int r = t & 63, kb = t >> 6;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = kb + 4 * j;
      const int gr = row0 + r, gk = k0 + k;
      float v = 0.0f;
      if (gr < rows && gk < K) v = ldg<TIN>(src, (int64_t)gr * rs + (int64_t)gk * cs);
      dst[k][r] = v;
    }
  }
}

template <typename TIN, typename TOUT, bool CPLX>
__global__ __launch_bounds__(256) void gemm_generic_kernel(GemmArgs g) {
  __shared__ float As_r[GBK][GLD], Bs_r[GBK][GLD];
  __shared__ float As_i[CPLX ? GBK : 1][GLD], Bs_i[CPLX ? GBK : 1][GLD];

  const int m0 = blockIdx.y * GBM, n0 = blockIdx.x * GBN;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int wm = (wid >> 1) * 32, wn = (wid & 1) * 32;
  const int l31 = lane & 31, lk = lane >> 5;
  const float sgn = g.conj_b ? -1.0f : 1.0f;

  f32x16 acc_r = {0}, acc_i = {0};

  for (int k0 = 0; k0 < g.K; k0 += GBK) {
    stage_tile<TIN>(As_r, g.a_r, g.a_rs, g.a_cs, m0, k0, g.M, g.K);
    stage_tile<TIN>(Bs_r, g.b_r, g.b_rs, g.b_cs, n0, k0, g.N, g.K);
    if (CPLX) {
      stage_tile<TIN>(As_i, g.a_i, g.a_rs, g.a_cs, m0, k0, g.M, g.K);
      stage_tile<TIN>(Bs_i, g.b_i, g.b_rs, g.b_cs, n0, k0, g.N, g.K);
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < GBK; kk += 2) {
      const float ar = As_r[kk + lk][wm + l31];
      const float br = Bs_r[kk + lk][wn + l31];
      acc_r = __builtin_amdgcn_mfma_f32_32x32x2f32(ar, br, acc_r, 0, 0, 0);
      if (CPLX) {
        const float ai = As_i[kk + lk][wm + l31];
        const float bi = sgn * Bs_i[kk + lk][wn + l31];
        acc_r = __builtin_amdgcn_mfma_f32_32x32x2f32(-ai, bi, acc_r, 0, 0, 0);
        acc_i = __builtin_amdgcn_mfma_f32_32x32x2f32(ar, bi, acc_i, 0, 0, 0);
        acc_i = __builtin_amdgcn_mfma_f32_32x32x2f32(ai, br, acc_i, 0, 0, 0);
      }
    }
    __syncthreads();
  }

  // C/D layout of the 32x32 tile: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  const int col = n0 + wn + l31;
  if (col >= g.N) return;
  const float b_r = g.bias_r ? g.bias_r[col] : 0.0f;
  const float b_i = (CPLX && g.bias_i) ? g.bias_i[col] : 0.0f;
  TOUT* cr = reinterpret_cast<TOUT*>(g.c_r);
  TOUT* ci = reinterpret_cast<TOUT*>(g.c_i);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * lk;
    if (row >= g.M) continue;
    const int64_t o = (int64_t)row * g.ldc + col;
    float vr = acc_r[r] + b_r;
    if (g.emul) vr *= g.emul[o];
    if (g.accumulate) vr += io<TOUT>::ld(cr + o);
    io<TOUT>::st(cr + o, vr);
    if (CPLX) {
      float vi = acc_i[r] + b_i;
      if (g.accumulate) vi += io<TOUT>::ld(ci + o);
      io<TOUT>::st(ci + o, vi);
    }
  }
}

template <bool CPLX>
int launch_gemm_generic(const GemmArgs& g, int in_dtype, int out_dtype, hipStream_t st) {
  if (g.M <= 0 || g.N <= 0) return 0;
  dim3 grid((g.N + GBN - 1) / GBN, (g.M + GBM - 1) / GBM);
  if (in_dtype == CPLXAMD_F32 && out_dtype == CPLXAMD_F32)
    gemm_generic_kernel<float, float, CPLX><<<grid, 256, 0, st>>>(g);
  else if (in_dtype == CPLXAMD_BF16 && out_dtype == CPLXAMD_F32)
    gemm_generic_kernel<bf16_t, float, CPLX><<<grid, 256, 0, st>>>(g);
  else if (in_dtype == CPLXAMD_BF16 && out_dtype == CPLXAMD_BF16)
    gemm_generic_kernel<bf16_t, bf16_t, CPLX><<<grid, 256, 0, st>>>(g);
  else if (in_dtype == CPLXAMD_F32 && out_dtype == CPLXAMD_BF16)
    gemm_generic_kernel<float, bf16_t, CPLX><<<grid, 256, 0, st>>>(g);
  else
    return CPLXAMD_EINVAL;
  CPLXAMD_CHECK_LAUNCH();
  return 0;
}

template int launch_gemm_generic<true>(const GemmArgs&, int, int, hipStream_t);
template int launch_gemm_generic<false>(const GemmArgs&, int, int, hipStream_t);

}  // namespace cplxamd
