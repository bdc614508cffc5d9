// the bilinear layer family (file:line under /root/reference/cplxmodule):
//   cplx.bilinear          cplx.py:1062-1087      y[b,o] = sum_ij conj?(x1[b,i]) W[o,i,j] x2[b,j] + bias[o]
//   CplxBilinearGaussian   nn/relevance/complex/base.py:59-84   s2 = bilinear(|x1|^2, |x2|^2, exp(log_sigma2))
//   BilinearGaussian       nn/relevance/real/base.py:52-77      the same for real tensors
//
// The contraction over j is a GEMM with the weight read as stored, T[b, (o,i)] = x2[b,:] . W[(o,i),:]
// (gemm*.hip: N = O*I1, K = I2); what is left is the reduction over i below, an HBM-bound pass over
// T (8 B per complex float32 element of T, read once):
//   fwd   y[b,o]   = sum_i u[b,i] T[b,o,i] + bias[o]            u = conj?(x1)
//   bwd   dT[b,o,i] = g[b,o] conj(u[b,i])                       (real: g u)
//         du[b,i]   = sum_o g[b,o] conj(T[b,o,i])               (real: sum_o g T);  dx1 = conj?(du)
// bwd reads T once and writes dT once (16 B per complex float32 element of T); dT then goes through
// the dgrad / wgrad GEMMs.  Real tensors pass NULL imaginary planes.
#include "common.h"

namespace cplxamd {

constexpr int kBT = 256;

// GS consecutive lanes reduce one (b, o) row of T; rows are contiguous, so a wave reads 64 / GS
// consecutive rows = one contiguous run of memory.
template <typename T, bool CPLX, int GS>
__global__ __launch_bounds__(kBT) void bilinear_reduce_fwd_kernel(
    const T* __restrict__ ur, const T* __restrict__ ui, const T* __restrict__ tr, const T* __restrict__ ti,
    const float* __restrict__ bias_r, const float* __restrict__ bias_i, T* __restrict__ yr,
    T* __restrict__ yi, int64_t rows, int O, int I1, float usign) {
  const int sub = threadIdx.x % GS;
  const int64_t row0 = ((int64_t)blockIdx.x * kBT + threadIdx.x) / GS;
  const int64_t rstride = (int64_t)gridDim.x * (kBT / GS);
  // every lane of a group runs the same number of trips (row0 is group-uniform): shuffles stay converged
  for (int64_t row = row0; row < rows; row += rstride) {
    const int64_t b = row / O;
    const T* u_r = ur + b * I1;
    const T* t_r = tr + row * I1;
    float ar = 0.0f, ai = 0.0f;
    if (CPLX) {
      const T* u_i = ui + b * I1;
      const T* t_i = ti + row * I1;
#pragma unroll 4
      for (int i = sub; i < I1; i += GS) {
        const float a = io<T>::ld(u_r + i), c = usign * io<T>::ld(u_i + i);
        const float p = io<T>::ld(t_r + i), q = io<T>::ld(t_i + i);
        ar += a * p - c * q;
        ai += a * q + c * p;
      }
    } else {
#pragma unroll 4
      for (int i = sub; i < I1; i += GS) ar += io<T>::ld(u_r + i) * io<T>::ld(t_r + i);
    }
#pragma unroll
    for (int m = GS >> 1; m > 0; m >>= 1) {
      ar += __shfl_xor(ar, m, 64);
      if (CPLX) ai += __shfl_xor(ai, m, 64);
    }
    if (sub == 0) {
      const int o = (int)(row - b * O);
      io<T>::st(yr + row, ar + (bias_r ? bias_r[o] : 0.0f));
      if (CPLX) io<T>::st(yi + row, ai + (bias_i ? bias_i[o] : 0.0f));
    }
  }
}

// one thread per (b, i): walks o, reading T[b,o,i] (coalesced over i), writing dT[b,o,i], summing du
template <typename T, bool CPLX>
__global__ __launch_bounds__(kBT) void bilinear_reduce_bwd_kernel(
    const T* __restrict__ ur, const T* __restrict__ ui, const T* __restrict__ tr, const T* __restrict__ ti,
    const T* __restrict__ gr, const T* __restrict__ gi, T* __restrict__ dur, T* __restrict__ dui,
    T* __restrict__ dtr, T* __restrict__ dti, int64_t BI, int O, int I1, float usign) {
  const int64_t p = (int64_t)blockIdx.x * kBT + threadIdx.x;
  if (p >= BI) return;
  const int64_t b = p / I1;
  const int i = (int)(p - b * I1);
  const float a = io<T>::ld(ur + p), c = CPLX ? usign * io<T>::ld(ui + p) : 0.0f;
  const T* g_r = gr + b * O;
  const T* g_i = CPLX ? gi + b * O : nullptr;
  const int64_t base = b * O * I1 + i;
  float sr = 0.0f, si = 0.0f;
#pragma unroll 4
  for (int o = 0; o < O; ++o) {
    const int64_t e = base + (int64_t)o * I1;
    const float x = io<T>::ld(g_r + o);
    if (CPLX) {
      const float y = io<T>::ld(g_i + o);
      if (tr) {
        const float m = io<T>::ld(tr + e), n = io<T>::ld(ti + e);
        sr += x * m + y * n;          // g conj(t)
        si += y * m - x * n;
      }
      if (dtr) {
        io<T>::st(dtr + e, x * a + y * c);  // g conj(u)
        io<T>::st(dti + e, y * a - x * c);
      }
    } else {
      if (tr) sr += x * io<T>::ld(tr + e);
      if (dtr) io<T>::st(dtr + e, x * a);
    }
  }
  if (dur) {
    io<T>::st(dur + p, sr);
    if (CPLX) io<T>::st(dui + p, usign * si);   // x1 = conj?(u)
  }
}

template <typename T, bool CPLX>
static int launch_fwd(const void* ur, const void* ui, const void* tr, const void* ti, const float* br,
                      const float* bi, void* yr, void* yi, int64_t rows, int O, int I1, float usign,
                      hipStream_t st) {
  // group size: the power of two that gives every lane about 4 elements, 4..64
  int gs = 4;
  while (gs < 64 && gs * 4 < I1) gs <<= 1;
#define GO(GS)                                                                                        \
  bilinear_reduce_fwd_kernel<T, CPLX, GS><<<stream_grid(rows * GS, kBT), kBT, 0, st>>>(                \
      (const T*)ur, (const T*)ui, (const T*)tr, (const T*)ti, br, bi, (T*)yr, (T*)yi, rows, O, I1, usign)
  switch (gs) {
    case 4: GO(4); break;
    case 8: GO(8); break;
    case 16: GO(16); break;
    case 32: GO(32); break;
    default: GO(64); break;
  }
#undef GO
  CPLXAMD_CHECK_LAUNCH();
  return 0;
}

}  // namespace cplxamd

using namespace cplxamd;

extern "C" {

int cplxamd_bilinear_reduce_fwd(const void* ur, const void* ui, const void* tr, const void* ti,
                                const float* bias_r, const float* bias_i, void* yr, void* yi, int64_t B,
                                int O, int I1, int conj_u, int dtype, void* stream) {
  const bool cplx = ui != nullptr;
  if (!ur || !tr || !yr || B < 0 || O <= 0 || I1 <= 0) return CPLXAMD_EINVAL;
  if (cplx != (ti != nullptr) || cplx != (yi != nullptr) || (!cplx && bias_i)) return CPLXAMD_EINVAL;
  if (cplx && (bias_r != nullptr) != (bias_i != nullptr)) return CPLXAMD_EINVAL;
  if (B == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const float us = (cplx && conj_u) ? -1.0f : 1.0f;
  const int64_t rows = B * O;
  if (dtype == CPLXAMD_F32)
    return cplx ? launch_fwd<float, true>(ur, ui, tr, ti, bias_r, bias_i, yr, yi, rows, O, I1, us, st)
                : launch_fwd<float, false>(ur, ui, tr, ti, bias_r, bias_i, yr, yi, rows, O, I1, us, st);
  if (dtype == CPLXAMD_BF16)
    return cplx ? launch_fwd<bf16_t, true>(ur, ui, tr, ti, bias_r, bias_i, yr, yi, rows, O, I1, us, st)
                : launch_fwd<bf16_t, false>(ur, ui, tr, ti, bias_r, bias_i, yr, yi, rows, O, I1, us, st);
  return CPLXAMD_EINVAL;
}

int cplxamd_bilinear_reduce_bwd(const void* ur, const void* ui, const void* tr, const void* ti,
                                const void* gr, const void* gi, void* dur, void* dui, void* dtr, void* dti,
                                int64_t B, int O, int I1, int conj_u, int dtype, void* stream) {
  const bool cplx = ui != nullptr;
  if (!ur || !gr || B < 0 || O <= 0 || I1 <= 0) return CPLXAMD_EINVAL;
  if (cplx != (gi != nullptr)) return CPLXAMD_EINVAL;
  if ((dur != nullptr) && !tr) return CPLXAMD_EINVAL;              // du needs T
  if (cplx && ((tr != nullptr) != (ti != nullptr) || (dur != nullptr) != (dui != nullptr) ||
               (dtr != nullptr) != (dti != nullptr)))
    return CPLXAMD_EINVAL;
  if (!dur) tr = ti = nullptr;                                     // T is only read for du
  if (B == 0 || (!dur && !dtr)) return 0;
  hipStream_t st = (hipStream_t)stream;
  const float us = (cplx && conj_u) ? -1.0f : 1.0f;
  const int64_t BI = B * I1;
  const int grid = (int)((BI + kBT - 1) / kBT);
#define GO(T, C)                                                                                     \
  bilinear_reduce_bwd_kernel<T, C><<<grid, kBT, 0, st>>>((const T*)ur, (const T*)ui, (const T*)tr,    \
                                                         (const T*)ti, (const T*)gr, (const T*)gi,   \
                                                         (T*)dur, (T*)dui, (T*)dtr, (T*)dti, BI, O, I1, us)
  if (dtype == CPLXAMD_F32) { if (cplx) GO(float, true); else GO(float, false); }
  else if (dtype == CPLXAMD_BF16) { if (cplx) GO(bf16_t, true); else GO(bf16_t, false); }
  else return CPLXAMD_EINVAL;
#undef GO
  CPLXAMD_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
