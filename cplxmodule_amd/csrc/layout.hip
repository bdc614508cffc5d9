// the steps either side of the GEMM / conv path in the reference's
// example models, as single-pass HBM-bound kernels (file:line under /root/reference/cplxmodule):
//   * (de)interleave            cplx.py:451-470  from / to_interleaved_real  (complex_view + clone /
//                               stack + flatten: two strided passes each in the reference)
//   * modReLU                   cplx.py:565-616  z * relu(1 - tau / clamp(|z|, 1e-5)) and its backward
//   * complex dropout           nn/modules/extra.py:7-25  one Bernoulli draw per COMPLEX element
// Traffic per complex element (fp32): interleave / deinterleave 16 B; modReLU fwd 16 B, bwd 32 B
// (+4 B when the threshold gradient is wanted); dropout fwd / bwd 16 B (the mask is regenerated from
// the Philox counter, never stored).
#include "common.h"

#pragma clang fp contract(off)      // modReLU parity: one rounding per reference op

namespace cplxamd {

constexpr int kLT = 256;

// x[2 i], x[2 i + 1] -> re[i], im[i]     (4 complex elements per thread and iteration)
template <typename T>
__global__ __launch_bounds__(kLT) void deinterleave_kernel(const T* x, T* re, T* im, int64_t n) {
  const int64_t n4 = n >> 2, stride = (int64_t)gridDim.x * kLT;
  for (int64_t i = (int64_t)blockIdx.x * kLT + threadIdx.x; i < n4; i += stride) {
    const f8 v = ld8(x + 8 * i);
    st4(re + 4 * i, f4{{v.h[0].v[0], v.h[0].v[2], v.h[1].v[0], v.h[1].v[2]}});
    st4(im + 4 * i, f4{{v.h[0].v[1], v.h[0].v[3], v.h[1].v[1], v.h[1].v[3]}});
  }
  if (blockIdx.x == 0) {
    const int64_t e = (n4 << 2) + threadIdx.x;
    if (e < n) { re[e] = x[2 * e]; im[e] = x[2 * e + 1]; }
  }
}

template <typename T>
__global__ __launch_bounds__(kLT) void interleave_kernel(const T* re, const T* im, T* out, int64_t n) {
  const int64_t n4 = n >> 2, stride = (int64_t)gridDim.x * kLT;
  for (int64_t i = (int64_t)blockIdx.x * kLT + threadIdx.x; i < n4; i += stride) {
    const f4 a = ld4(re + 4 * i), b = ld4(im + 4 * i);
    st8(out + 8 * i, f8{{f4{{a.v[0], b.v[0], a.v[1], b.v[1]}}, f4{{a.v[2], b.v[2], a.v[3], b.v[3]}}}});
  }
  if (blockIdx.x == 0) {
    const int64_t e = (n4 << 2) + threadIdx.x;
    if (e < n) { out[2 * e] = re[e]; out[2 * e + 1] = im[e]; }
  }
}

// Elementwise complex product / quotient (Cplx.__mul__ / __truediv__, cplxmodule/cplx.py:135-165) in ONE launch, with the
// reference's operation order (every product, sum and quotient rounded on its own: no fused multiply-add), so values
// are bit-identical to its 6 / 12 elementwise torch kernels:
//   mul:  re = a c - b d;  im = b c + a d                       (conj_b: d -> -d)
//   div:  den = c c + d d;  p = c / den, q = -d / den;  re = a p - b q;  im = b p + a q      (conj_b: d -> -d)
// `neg` negates the result (the quotient's gradient with respect to the divisor).  (fp contraction is off file-wide.)
template <typename T, bool DIV>
__global__ __launch_bounds__(kLT) void cplx_mul_kernel(const T* ar, const T* ai, const T* br, const T* bi, T* or_, T* oi,
                                                       int64_t n, float sd, float sg) {
  const int64_t n4 = n >> 2, stride = (int64_t)gridDim.x * kLT;
  auto f = [&](float a, float b, float c, float d, float& re, float& im) __attribute__((always_inline)) {
    d = sd * d;
    if (DIV) {
      const float den = c * c + d * d;
      c = c / den;
      d = -d / den;
    }
    const float t0 = a * c, t1 = b * d, t2 = b * c, t3 = a * d;
    re = sg * (t0 - t1);
    im = sg * (t2 + t3);
  };
  for (int64_t i = (int64_t)blockIdx.x * kLT + threadIdx.x; i < n4; i += stride) {
    const f4 a = ld4(ar + 4 * i), b = ld4(ai + 4 * i), c = ld4(br + 4 * i), d = ld4(bi + 4 * i);
    f4 x, y;
#pragma unroll
    for (int j = 0; j < 4; ++j) f(a.v[j], b.v[j], c.v[j], d.v[j], x.v[j], y.v[j]);
    st4(or_ + 4 * i, x);
    st4(oi + 4 * i, y);
  }
  if (blockIdx.x == 0) {
    const int64_t e = (n4 << 2) + threadIdx.x;
    if (e < n) {
      float x, y;
      f(io<T>::ld(ar + e), io<T>::ld(ai + e), io<T>::ld(br + e), io<T>::ld(bi + e), x, y);
      io<T>::st(or_ + e, x);
      io<T>::st(oi + e, y);
    }
  }
}
// ReLU on both planes of a complex tensor (CplxToCplx[torch.nn.ReLU], cplxmodule/nn/modules/base.py:167-199) in one
// launch, forward and backward: torch's semantics (NaN passes; the backward masks on the saved OUTPUT, y <= 0 -> 0).
template <typename T, bool BWD>
__global__ __launch_bounds__(kLT) void split_relu_kernel(const T* ar, const T* ai, const T* gr, const T* gi, T* or_, T* oi,
                                                         int64_t n) {
  const int64_t n4 = n >> 2, stride = (int64_t)gridDim.x * kLT;
  auto f = [](float a, float g) __attribute__((always_inline)) -> float {
    if (BWD) return a <= 0.f ? 0.f : g;
    return (a > 0.f || a != a) ? a : 0.f;
  };
  for (int64_t i = (int64_t)blockIdx.x * kLT + threadIdx.x; i < n4; i += stride) {
    const f4 a = ld4(ar + 4 * i), b = ld4(ai + 4 * i);
    f4 u = a, v = b, x, y;
    if (BWD) { u = ld4(gr + 4 * i); v = ld4(gi + 4 * i); }
#pragma unroll
    for (int j = 0; j < 4; ++j) { x.v[j] = f(a.v[j], u.v[j]); y.v[j] = f(b.v[j], v.v[j]); }
    st4(or_ + 4 * i, x);
    st4(oi + 4 * i, y);
  }
  if (blockIdx.x == 0) {
    const int64_t e = (n4 << 2) + threadIdx.x;
    if (e < n) {
      const float a = io<T>::ld(ar + e), b = io<T>::ld(ai + e);
      io<T>::st(or_ + e, f(a, BWD ? io<T>::ld(gr + e) : a));
      io<T>::st(oi + e, f(b, BWD ? io<T>::ld(gi + e) : b));
    }
  }
}

// modReLU with the reference's op order: m = max(|z|, 1e-5) (|z| = sqrt(fma(zi, zi, zr zr)), the CPU
// kernel's form), s = max(1 - tau / m, 0), y = z s.  tau: scalar value, or a tensor of n elements.
struct ModRelu {
  float s, m, az;
  __device__ __forceinline__ ModRelu(float zr, float zi, float tau) {
    az = rn_sqrt(fmaf(zi, zi, zr * zr));
    m = fmaxf(az, 1e-5f);
    s = fmaxf(1.0f - tau / m, 0.0f);
  }
};

template <typename T, bool TVEC>
__global__ __launch_bounds__(kLT) void modrelu_fwd_kernel(const T* zr, const T* zi, const float* tau,
                                                          float tau0, T* yr, T* yi, int64_t n) {
  const int64_t n4 = n >> 2, stride = (int64_t)gridDim.x * kLT;
  if (!TVEC && tau) tau0 = tau[0];                       // 1-element device tensor: no host sync
  for (int64_t i = (int64_t)blockIdx.x * kLT + threadIdx.x; i < n4; i += stride) {
    const f4 a = ld4(zr + 4 * i), b = ld4(zi + 4 * i);
    f4 t, or_, oi;
    if (TVEC) t = ld4(tau + 4 * i);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const ModRelu r(a.v[j], b.v[j], TVEC ? t.v[j] : tau0);
      or_.v[j] = a.v[j] * r.s;
      oi.v[j] = b.v[j] * r.s;
    }
    st4(yr + 4 * i, or_);
    st4(yi + 4 * i, oi);
  }
  if (blockIdx.x == 0) {
    const int64_t e = (n4 << 2) + threadIdx.x;
    if (e < n) {
      const float a = io<T>::ld(zr + e), b = io<T>::ld(zi + e);
      const ModRelu r(a, b, TVEC ? tau[e] : tau0);
      io<T>::st(yr + e, a * r.s);
      io<T>::st(yi + e, b * r.s);
    }
  }
}

// dz = g s + z (g . z) tau / m^3 on the active branch with |z| above the clamp;  dtau = -(g . z) / m
__device__ __forceinline__ void modrelu_grad(float a, float b, float tau, float gr, float gi, float& dr,
                                             float& di, float& dt) {
  const ModRelu r(a, b, tau);
  const float dot = gr * a + gi * b;
  const bool active = (1.0f - tau / r.m) > 0.0f;
  const float k = (active && r.az >= 1e-5f) ? dot * tau / (r.m * r.m * r.m) : 0.0f;
  dr = gr * r.s + a * k;
  di = gi * r.s + b * k;
  dt = active ? -dot / r.m : 0.0f;
}

template <typename T, bool TVEC>
__global__ __launch_bounds__(kLT) void modrelu_bwd_kernel(const T* zr, const T* zi, const float* tau,
                                                          float tau0, const T* gr, const T* gi, T* dzr,
                                                          T* dzi, float* dtau, int64_t n) {
  const int64_t n4 = n >> 2, stride = (int64_t)gridDim.x * kLT;
  if (!TVEC && tau) tau0 = tau[0];
  for (int64_t i = (int64_t)blockIdx.x * kLT + threadIdx.x; i < n4; i += stride) {
    const f4 a = ld4(zr + 4 * i), b = ld4(zi + 4 * i), u = ld4(gr + 4 * i), v = ld4(gi + 4 * i);
    f4 t, dr, di, dt;
    if (TVEC) t = ld4(tau + 4 * i);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      modrelu_grad(a.v[j], b.v[j], TVEC ? t.v[j] : tau0, u.v[j], v.v[j], dr.v[j], di.v[j], dt.v[j]);
    st4(dzr + 4 * i, dr);
    st4(dzi + 4 * i, di);
    if (dtau) st4(dtau + 4 * i, dt);
  }
  if (blockIdx.x == 0) {
    const int64_t e = (n4 << 2) + threadIdx.x;
    if (e < n) {
      float dr, di, dt;
      modrelu_grad(io<T>::ld(zr + e), io<T>::ld(zi + e), TVEC ? tau[e] : tau0, io<T>::ld(gr + e),
                   io<T>::ld(gi + e), dr, di, dt);
      io<T>::st(dzr + e, dr);
      io<T>::st(dzi + e, di);
      if (dtau) dtau[e] = dt;
    }
  }
}

// Philox4x32-7 (common.h: kPhiloxRounds), counter (group_lo, group_hi, offset_lo, offset_hi), key seed: 4 uniform words per
// group of 4 complex elements; element e is kept iff word (e & 3) of group e >> 2 is >= p * 2^32.
__device__ __forceinline__ void philox4(uint64_t ctr_lo, uint64_t ctr_hi, uint64_t key, uint32_t (&o)[4]) {
  uint32_t c0 = (uint32_t)ctr_lo, c1 = (uint32_t)(ctr_lo >> 32);
  uint32_t c2 = (uint32_t)ctr_hi, c3 = (uint32_t)(ctr_hi >> 32);
  uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
#pragma unroll
  for (int r = 0; r < kPhiloxRounds; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = __builtin_amdgcn_bitop3_b32((uint32_t)(p1 >> 32), c1, k0, 0x96);
    const uint32_t n2 = __builtin_amdgcn_bitop3_b32((uint32_t)(p0 >> 32), c3, k1, 0x96);
    c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

// y = x * keep / (1 - p) on both planes (forward on activations, backward on gradients: same kernel)
template <typename T>
__global__ __launch_bounds__(kLT) void cplx_dropout_kernel(const T* xr, const T* xi, T* yr, T* yi,
                                                           uint32_t thresh, float scale, uint64_t seed,
                                                           uint64_t offset, const uint64_t* state,
                                                           int64_t n) {
  if (state) { seed = state[0]; offset = state[1]; }
  const int64_t n4 = (n + 3) >> 2, stride = (int64_t)gridDim.x * kLT;
  for (int64_t i = (int64_t)blockIdx.x * kLT + threadIdx.x; i < n4; i += stride) {
    uint32_t w[4];
    philox4((uint64_t)i, offset, seed, w);
    if (4 * i + 3 < n) {
      const f4 a = ld4(xr + 4 * i), b = ld4(xi + 4 * i);
      f4 or_, oi;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float k = w[j] >= thresh ? scale : 0.0f;
        or_.v[j] = a.v[j] * k;
        oi.v[j] = b.v[j] * k;
      }
      st4(yr + 4 * i, or_);
      st4(yi + 4 * i, oi);
    } else {
      for (int j = 0; j < 4 && 4 * i + j < n; ++j) {
        const float k = w[j] >= thresh ? scale : 0.0f;
        io<T>::st(yr + 4 * i + j, io<T>::ld(xr + 4 * i + j) * k);
        io<T>::st(yi + 4 * i + j, io<T>::ld(xi + 4 * i + j) * k);
      }
    }
  }
}

static bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace cplxamd

using namespace cplxamd;

extern "C" {

int cplxamd_deinterleave(const void* x, void* re, void* im, int64_t n, int dtype, void* stream) {
  if (!x || !re || !im || n < 0) return CPLXAMD_EINVAL;
  if (!al16(x) || !al16(re) || !al16(im)) return CPLXAMD_EALIGN;
  if (n == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int grid = stream_grid((n >> 2) + 1, kLT);
  if (dtype == CPLXAMD_F32) deinterleave_kernel<float><<<grid, kLT, 0, st>>>((const float*)x, (float*)re, (float*)im, n);
  else if (dtype == CPLXAMD_BF16) deinterleave_kernel<bf16_t><<<grid, kLT, 0, st>>>((const bf16_t*)x, (bf16_t*)re, (bf16_t*)im, n);
  else return CPLXAMD_EINVAL;
  CPLXAMD_CHECK_LAUNCH();
  return 0;
}

int cplxamd_interleave(const void* re, const void* im, void* out, int64_t n, int dtype, void* stream) {
  if (!out || !re || !im || n < 0) return CPLXAMD_EINVAL;
  if (!al16(out) || !al16(re) || !al16(im)) return CPLXAMD_EALIGN;
  if (n == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int grid = stream_grid((n >> 2) + 1, kLT);
  if (dtype == CPLXAMD_F32) interleave_kernel<float><<<grid, kLT, 0, st>>>((const float*)re, (const float*)im, (float*)out, n);
  else if (dtype == CPLXAMD_BF16) interleave_kernel<bf16_t><<<grid, kLT, 0, st>>>((const bf16_t*)re, (const bf16_t*)im, (bf16_t*)out, n);
  else return CPLXAMD_EINVAL;
  CPLXAMD_CHECK_LAUNCH();
  return 0;
}

/* o = a b (div == 0) or a / b (div != 0), b conjugated first when conj_b != 0, negated when neg != 0; planes of n elements,
 * 16-byte aligned.  Operation order of cplxmodule/cplx.py:135-165 (bit-identical values). */
int cplxamd_cplx_mul(const void* a_r, const void* a_i, const void* b_r, const void* b_i, void* o_r, void* o_i, int64_t n,
                     int div, int conj_b, int neg, int dtype, void* stream) {
  if (!a_r || !a_i || !b_r || !b_i || !o_r || !o_i || n < 0) return CPLXAMD_EINVAL;
  if (!al16(a_r) || !al16(a_i) || !al16(b_r) || !al16(b_i) || !al16(o_r) || !al16(o_i)) return CPLXAMD_EALIGN;
  if (n == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int grid = stream_grid((n >> 2) + 1, kLT);
  const float sd = conj_b ? -1.f : 1.f, sg = neg ? -1.f : 1.f;
#define CM(T, D) cplx_mul_kernel<T, D><<<grid, kLT, 0, st>>>((const T*)a_r, (const T*)a_i, (const T*)b_r, (const T*)b_i, (T*)o_r, (T*)o_i, n, sd, sg)
  if (dtype == CPLXAMD_F32) { if (div) CM(float, true); else CM(float, false); }
  else if (dtype == CPLXAMD_BF16) { if (div) CM(bf16_t, true); else CM(bf16_t, false); }
  else return CPLXAMD_EINVAL;
#undef CM
  CPLXAMD_CHECK_LAUNCH();
  return 0;
}

/* y = relu(x) on both planes (bwd == 0), or dx = (y <= 0 ? 0 : g) from the saved outputs (bwd != 0: a_* = y, g_* = g). */
int cplxamd_split_relu(const void* a_r, const void* a_i, const void* g_r, const void* g_i, void* o_r, void* o_i, int64_t n,
                       int bwd, int dtype, void* stream) {
  if (!a_r || !a_i || !o_r || !o_i || n < 0 || (bwd && (!g_r || !g_i))) return CPLXAMD_EINVAL;
  if (!al16(a_r) || !al16(a_i) || !al16(o_r) || !al16(o_i) || (bwd && (!al16(g_r) || !al16(g_i)))) return CPLXAMD_EALIGN;
  if (n == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int grid = stream_grid((n >> 2) + 1, kLT);
#define SR(T, B) split_relu_kernel<T, B><<<grid, kLT, 0, st>>>((const T*)a_r, (const T*)a_i, (const T*)g_r, (const T*)g_i, (T*)o_r, (T*)o_i, n)
  if (dtype == CPLXAMD_F32) { if (bwd) SR(float, true); else SR(float, false); }
  else if (dtype == CPLXAMD_BF16) { if (bwd) SR(bf16_t, true); else SR(bf16_t, false); }
  else return CPLXAMD_EINVAL;
#undef SR
  CPLXAMD_CHECK_LAUNCH();
  return 0;
}

int cplxamd_modrelu_fwd(const void* zr, const void* zi, const float* tau, float tau_value, int tau_numel,
                        void* yr, void* yi, int64_t n, int dtype, void* stream) {
  if (!zr || !zi || !yr || !yi || n < 0) return CPLXAMD_EINVAL;
  if (tau_numel != 0 && tau_numel != 1 && tau_numel != n) return CPLXAMD_ESHAPE;
  if ((tau_numel != 0) != (tau != nullptr)) return CPLXAMD_EINVAL;
  if (!al16(zr) || !al16(zi) || !al16(yr) || !al16(yi) || (tau_numel > 1 && !al16(tau))) return CPLXAMD_EALIGN;
  if (n == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int grid = stream_grid((n >> 2) + 1, kLT);
  const bool tv = tau_numel > 1;
#define MR(T, V) modrelu_fwd_kernel<T, V><<<grid, kLT, 0, st>>>((const T*)zr, (const T*)zi, tau, tau_value, (T*)yr, (T*)yi, n)
  if (dtype == CPLXAMD_F32) { if (tv) MR(float, true); else MR(float, false); }
  else if (dtype == CPLXAMD_BF16) { if (tv) MR(bf16_t, true); else MR(bf16_t, false); }
  else return CPLXAMD_EINVAL;
#undef MR
  CPLXAMD_CHECK_LAUNCH();
  return 0;
}

int cplxamd_modrelu_bwd(const void* zr, const void* zi, const float* tau, float tau_value, int tau_numel,
                        const void* gr, const void* gi, void* dzr, void* dzi, float* dtau, int64_t n,
                        int dtype, void* stream) {
  if (!zr || !zi || !gr || !gi || !dzr || !dzi || n < 0) return CPLXAMD_EINVAL;
  if (tau_numel != 0 && tau_numel != 1 && tau_numel != n) return CPLXAMD_ESHAPE;
  if ((tau_numel != 0) != (tau != nullptr)) return CPLXAMD_EINVAL;
  if (!al16(zr) || !al16(zi) || !al16(gr) || !al16(gi) || !al16(dzr) || !al16(dzi) || !al16(dtau) ||
      (tau_numel > 1 && !al16(tau)))
    return CPLXAMD_EALIGN;
  if (n == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int grid = stream_grid((n >> 2) + 1, kLT);
  const bool tv = tau_numel > 1;
#define MR(T, V) modrelu_bwd_kernel<T, V><<<grid, kLT, 0, st>>>((const T*)zr, (const T*)zi, tau, tau_value, (const T*)gr, (const T*)gi, (T*)dzr, (T*)dzi, dtau, n)
  if (dtype == CPLXAMD_F32) { if (tv) MR(float, true); else MR(float, false); }
  else if (dtype == CPLXAMD_BF16) { if (tv) MR(bf16_t, true); else MR(bf16_t, false); }
  else return CPLXAMD_EINVAL;
#undef MR
  CPLXAMD_CHECK_LAUNCH();
  return 0;
}

int cplxamd_cplx_dropout(const void* xr, const void* xi, void* yr, void* yi, double p, uint64_t seed,
                         uint64_t offset, const uint64_t* state, int64_t n, int dtype, void* stream) {
  if (!xr || !xi || !yr || !yi || n < 0 || !(p >= 0.0) || !(p < 1.0)) return CPLXAMD_EINVAL;
  if (!al16(xr) || !al16(xi) || !al16(yr) || !al16(yi)) return CPLXAMD_EALIGN;
  if (n == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const double t = p * 4294967296.0;
  const uint32_t thresh = t >= 4294967295.0 ? 0xffffffffu : (uint32_t)t;
  const float scale = (float)(1.0 / (1.0 - p));
  const int grid = stream_grid((n >> 2) + 1, kLT);
  if (dtype == CPLXAMD_F32)
    cplx_dropout_kernel<float><<<grid, kLT, 0, st>>>((const float*)xr, (const float*)xi, (float*)yr, (float*)yi, thresh, scale, seed, offset, state, n);
  else if (dtype == CPLXAMD_BF16)
    cplx_dropout_kernel<bf16_t><<<grid, kLT, 0, st>>>((const bf16_t*)xr, (const bf16_t*)xi, (bf16_t*)yr, (bf16_t*)yi, thresh, scale, seed, offset, state, n);
  else return CPLXAMD_EINVAL;
  CPLXAMD_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
