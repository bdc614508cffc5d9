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
#include <initializer_list>

#include "common.h"

namespace cplxamd {

constexpr int kBT = 256;

// V elements per lane and load (4 when I1 % 4 == 0: 16-B float / 8-B bf16 loads, else 1)
template <typename T, int V> struct vecio;
template <typename T> struct vecio<T, 1> {
  struct type { float v[1]; };
  static __device__ __forceinline__ type ld(const T* p) { return type{{io<T>::ld(p)}}; }
  static __device__ __forceinline__ void st(T* p, const type& a) { io<T>::st(p, a.v[0]); }
};
template <typename T> struct vecio<T, 4> {
  using type = f4;
  static __device__ __forceinline__ type ld(const T* p) { return ld4(p); }
  static __device__ __forceinline__ void st(T* p, const type& a) { st4(p, a); }
};

// GS consecutive lanes reduce one (b, o) row of T; rows are contiguous, so a wave reads 64 / GS
// consecutive rows = one contiguous run of memory.
template <typename T, bool CPLX, int GS, int V>
__global__ __launch_bounds__(kBT) void bilinear_reduce_fwd_kernel(
    const T* __restrict__ ur, const T* __restrict__ ui, const T* __restrict__ tr, const T* __restrict__ ti,
    const float* __restrict__ bias_r, const float* __restrict__ bias_i, T* __restrict__ yr,
    T* __restrict__ yi, int64_t rows, int O, int I1, float usign) {
  using IO = vecio<T, V>;
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
#pragma unroll 2
      for (int i = sub * V; i < I1; i += GS * V) {
        const auto a = IO::ld(u_r + i), c = IO::ld(u_i + i), p = IO::ld(t_r + i), q = IO::ld(t_i + i);
#pragma unroll
        for (int e = 0; e < V; ++e) {
          const float cs = usign * c.v[e];
          ar += a.v[e] * p.v[e] - cs * q.v[e];
          ai += a.v[e] * q.v[e] + cs * p.v[e];
        }
      }
    } else {
#pragma unroll 2
      for (int i = sub * V; i < I1; i += GS * V) {
        const auto a = IO::ld(u_r + i), p = IO::ld(t_r + i);
#pragma unroll
        for (int e = 0; e < V; ++e) ar += a.v[e] * p.v[e];
      }
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

// one thread per (b, V consecutive i): walks o, reading T[b,o,i..] (coalesced over i), writing
// dT[b,o,i..], summing du
template <typename T, bool CPLX, int V>
__global__ __launch_bounds__(kBT) void bilinear_reduce_bwd_kernel(
    const T* __restrict__ ur, const T* __restrict__ ui, const T* __restrict__ tr, const T* __restrict__ ti,
    const T* __restrict__ gr, const T* __restrict__ gi, T* __restrict__ dur, T* __restrict__ dui,
    T* __restrict__ dtr, T* __restrict__ dti, int64_t BI, int O, int I1, float usign) {
  using IO = vecio<T, V>;
  const int64_t p = ((int64_t)blockIdx.x * kBT + threadIdx.x) * V;
  if (p >= BI) return;
  const int64_t b = p / I1;
  const int i = (int)(p - b * I1);
  const auto a = IO::ld(ur + p);
  auto c = a;
  if (CPLX) {
    c = IO::ld(ui + p);
#pragma unroll
    for (int e = 0; e < V; ++e) c.v[e] *= usign;
  }
  const T* g_r = gr + b * O;
  const T* g_i = CPLX ? gi + b * O : nullptr;
  const int64_t base = b * O * I1 + i;
  float sr[V], si[V];
#pragma unroll
  for (int e = 0; e < V; ++e) sr[e] = si[e] = 0.0f;
#pragma unroll 4
  for (int o = 0; o < O; ++o) {
    const int64_t e0 = base + (int64_t)o * I1;
    const float x = io<T>::ld(g_r + o);
    const float y = CPLX ? io<T>::ld(g_i + o) : 0.0f;
    if (tr) {
      const auto m = IO::ld(tr + e0);
      if (CPLX) {
        const auto n = IO::ld(ti + e0);
#pragma unroll
        for (int e = 0; e < V; ++e) {
          sr[e] += x * m.v[e] + y * n.v[e];      // g conj(t)
          si[e] += y * m.v[e] - x * n.v[e];
        }
      } else {
#pragma unroll
        for (int e = 0; e < V; ++e) sr[e] += x * m.v[e];
      }
    }
    if (dtr) {
      auto dr = a, di = a;
#pragma unroll
      for (int e = 0; e < V; ++e) {
        dr.v[e] = CPLX ? x * a.v[e] + y * c.v[e] : x * a.v[e];   // g conj(u)
        di.v[e] = y * a.v[e] - x * c.v[e];
      }
      IO::st(dtr + e0, dr);
      if (CPLX) IO::st(dti + e0, di);
    }
  }
  if (dur) {
    auto o_r = a, o_i = a;
#pragma unroll
    for (int e = 0; e < V; ++e) { o_r.v[e] = sr[e]; o_i.v[e] = usign * si[e]; }   // x1 = conj?(u)
    IO::st(dur + p, o_r);
    if (CPLX) IO::st(dui + p, o_i);
  }
}

template <typename T>
static bool vec4_ok(int I1, std::initializer_list<const void*> ptrs) {
  if (I1 & 3) return false;
  for (const void* q : ptrs)
    if (q && (reinterpret_cast<uintptr_t>(q) & (4 * sizeof(T) - 1))) return false;
  return true;
}

template <typename T, bool CPLX>
static int launch_fwd(const void* ur, const void* ui, const void* tr, const void* ti, const float* br,
                      const float* bi, void* yr, void* yi, int64_t rows, int O, int I1, float usign,
                      hipStream_t st) {
  const bool v4 = vec4_ok<T>(I1, {ur, ui, tr, ti});
  const int V = v4 ? 4 : 1;
  // group size: the power of two that gives every lane about two loads, 4..64
  int gs = 4;
  while (gs < 64 && gs * V * 2 < I1) gs <<= 1;
#define GO(GS, VV)                                                                                    \
  bilinear_reduce_fwd_kernel<T, CPLX, GS, VV><<<stream_grid(rows * GS, kBT), kBT, 0, st>>>(            \
      (const T*)ur, (const T*)ui, (const T*)tr, (const T*)ti, br, bi, (T*)yr, (T*)yi, rows, O, I1, usign)
#define GOV(GS) do { if (v4) GO(GS, 4); else GO(GS, 1); } while (0)
  switch (gs) {
    case 4: GOV(4); break;
    case 8: GOV(8); break;
    case 16: GOV(16); break;
    case 32: GOV(32); break;
    default: GOV(64); break;
  }
#undef GOV
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
  // the o loop is serial per thread: 4 elements per thread only when that still leaves >= 4096 blocks
  const bool wide = BI >= ((int64_t)1 << 22);
#define GO(T, C, V)                                                                                  \
  bilinear_reduce_bwd_kernel<T, C, V><<<(int)((BI / V + kBT - 1) / kBT), kBT, 0, st>>>(               \
      (const T*)ur, (const T*)ui, (const T*)tr, (const T*)ti, (const T*)gr, (const T*)gi, (T*)dur,   \
      (T*)dui, (T*)dtr, (T*)dti, BI, O, I1, us)
#define GOV(T, C)                                                                                    \
  do {                                                                                               \
    if (wide && vec4_ok<T>(I1, {ur, ui, tr, ti, dur, dui, dtr, dti})) GO(T, C, 4); else GO(T, C, 1);         \
  } while (0)
  if (dtype == CPLXAMD_F32) { if (cplx) GOV(float, true); else GOV(float, false); }
  else if (dtype == CPLXAMD_BF16) { if (cplx) GOV(bf16_t, true); else GOV(bf16_t, false); }
  else return CPLXAMD_EINVAL;
#undef GOV
#undef GO
  CPLXAMD_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
