// K6/K7: log-alpha -> KL penalty (+ wavefront-shuffle reduction, + gradients) and masks.
//
// One pass over (wr, wi, log_sigma2): 12 B/element read for the complex kinds (8 B real),
// + 12 B/element written when gradients are produced.  HBM-bound; the arithmetic (one fp64
// log only in the mask/log-alpha kernels, fp32 otherwise) stays well under the VALU budget.
//
// Reference arithmetic restated here (file:line under /root/reference/cplxmodule):
//   log_alpha  nn/relevance/complex/base.py:27-31, nn/relevance/real/base.py:23-26
//   abs(Cplx)  cplx.py:183-192
//   penalties  nn/relevance/real/vd.py:54-76, real/ard.py:10-39, complex/vd.py:95-99,
//              complex/ard.py:9-39;  Ei backward complex/vd.py:39-41
//   masks      nn/relevance/real/vd.py:16-19, complex/vd.py:50-53
#include "common.h"

// The reference issues separate torch ops (one rounding each); keep the compiler from
// fusing a*b+c into fma so that parity-mode results are bit-identical.  Explicit fmaf()
// calls below are deliberate.
#pragma clang fp contract(off)

namespace cplxamd {

constexpr int kKlThreads = 256;
constexpr int kKlMaxBlocks = 2048;
#ifndef KL_UNROLL
#define KL_UNROLL 1     // (2 and 4 measured 2-4 % slower on the fused kernel: profiles/r06_kl_pmc.txt)
#endif
#ifndef KL_NT
#define KL_NT 3       // bit 0: nontemporal gradient stores, bit 1: nontemporal operand loads (read once, written once: +3-8 % on the fused kernel, +10 % on the value kernel; profiles/r06_kl_pmc.txt)
#endif

constexpr float kEulerGamma = 0.57721566490153286f;
constexpr float kK1 = 0.63576f, kK2 = 1.87320f, kK3 = 1.48695f;

// |w| exactly as torch's CPU 2-norm kernel rounds it: sqrt(fma(wi, wi, rn(wr*wr)))
// (bit-for-bit on 2^20 samples, DESIGN.md "mask exactness").
template <bool CPLX, bool EXACT = true>
__device__ __forceinline__ float weight_abs(float wr, float wi) {
  // EXACT (masks / log_alpha): correctly rounded sqrt.  KL values and slopes only need ~1e-6
  // relative accuracy: the native v_sqrt_f32 (1 ulp) saves the Newton fix-up.
  if (CPLX) return EXACT ? rn_sqrt(fmaf(wi, wi, wr * wr)) : __builtin_amdgcn_sqrtf(fmaf(wi, wi, wr * wr));
  return fabsf(wr);
}

// EXACT: the log is evaluated in fp64 and rounded once, i.e. correctly rounded, which is
// what the reference's libm delivers on > 99.99 % of inputs -> masks come out bit-identical.
template <bool CPLX, bool EXACT>
__device__ __forceinline__ float log_alpha_of(float ls2, float wr, float wi, float& theta) {
  theta = weight_abs<CPLX, EXACT>(wr, wi);
  const float u = theta + 1e-12f;
  const float l = EXACT ? exact_logf(u) : logf(u);
  return ls2 - 2.0f * l;
}

// log1p(e) = log(u) * e / (u - 1), u = fl(1 + e): the rounding error of u cancels in the
// ratio (Kahan / HP-15C trick), ~1 ulp, one log + one reciprocal instead of ocml's log1pf.
__device__ __forceinline__ float log1p_pos(float e) {
  const float u = 1.0f + e;
  const float d = u - 1.0f;
  return d == 0.0f ? e : logf(u) * e * __builtin_amdgcn_rcpf(d);
}
__device__ __forceinline__ float softplus_f(float t) {
  return t > 20.0f ? t : log1p_pos(expf(t));
}
__device__ __forceinline__ float sigmoid_f(float t) { return 1.0f / (1.0f + expf(-t)); }
__device__ __forceinline__ float softplus_grad_f(float t) {
  return t > 20.0f ? 1.0f : sigmoid_f(t);
}

// f(t) = gamma + t - Ei(-e^t) = gamma + ln x + E1(x), x = e^t.
//  x <= 1: the power series of E1 cancels gamma + ln x analytically:
//          f = -sum_{k>=1} (-x)^k / (k k!)                       (11 terms, rel err 2e-10)
//  x  > 1: E1(x) = e^-x / x * P4(x)/Q4(x), Abramowitz & Stegun 5.1.56 (|err| < 2e-8)
__device__ __forceinline__ float cplx_vd_value(float t) {
  const float x = expf(t);
  if (x <= 1.0f) {
    float s = 2.2774860e-9f;           // +1/(11*11!)
    s = fmaf(s, x, -2.7557319e-8f);    // -1/(10*10!)
    s = fmaf(s, x, 3.0619244e-7f);     // +1/(9*9!)
    s = fmaf(s, x, -3.1001984e-6f);    // -1/(8*8!)
    s = fmaf(s, x, 2.8344671e-5f);     // +1/(7*7!)
    s = fmaf(s, x, -2.3148148e-4f);    // -1/(6*6!)
    s = fmaf(s, x, 1.6666667e-3f);     // +1/(5*5!)
    s = fmaf(s, x, -1.0416667e-2f);    // -1/(4*4!)
    s = fmaf(s, x, 5.5555556e-2f);     // +1/(3*3!)
    s = fmaf(s, x, -0.25f);            // -1/(2*2!)
    s = fmaf(s, x, 1.0f);              // +1/(1*1!)
    return s * x;
  }
  float e1 = 0.0f;
  if (x < 104.0f) {
    const float num = fmaf(fmaf(fmaf(fmaf(1.0f, x, 8.5733287401f), x, 18.0590169730f), x,
                                8.6347608925f), x, 0.2677737343f);
    const float den = fmaf(fmaf(fmaf(fmaf(1.0f, x, 9.5733223454f), x, 25.6329561486f), x,
                                21.0996530827f), x, 3.9584969228f);
    e1 = expf(-x) * num * __builtin_amdgcn_rcpf(x * den);   // one reciprocal (1 ulp) for both divisions
  }
  return kEulerGamma + t + e1;
}

// 1 - exp(-x) for x = e^t >= 0 (the slope of the exact complex KL, complex/vd.py:95-99 differentiated): alternating series
// below 1/4 (8 terms: the first omitted one is x^9 / 9! < 1e-11 x), 1 - expf(-x) above (absolute error of expf <= 6e-8 on
// a result >= 0.22): relative error < 3e-7 everywhere, half the instructions of ocml's expm1f (the fused KL kernel is
// co-limited by its ~140 lane-operations per element, profiles/r03_kl_pmc.txt).  Both arms are evaluated, one select.
__device__ __forceinline__ float one_minus_exp_neg(float x) {
  float s = -2.4801587e-5f;            // -1/8!
  s = fmaf(s, x, 1.9841270e-4f);       // +1/7!
  s = fmaf(s, x, -1.3888889e-3f);      // -1/6!
  s = fmaf(s, x, 8.3333333e-3f);       // +1/5!
  s = fmaf(s, x, -4.1666667e-2f);      // -1/4!
  s = fmaf(s, x, 1.6666667e-1f);       // +1/3!
  s = fmaf(s, x, -0.5f);
  s = fmaf(s, x, 1.0f);
  const float big = 1.0f - expf(-x);
  return x < 0.25f ? s * x : big;
}

template <int KIND>
__device__ __forceinline__ float kl_value(float t) {
  if (KIND == CPLXAMD_KL_REAL_VD)
    return fmaf(0.5f, softplus_f(t), kK1 * sigmoid_f(fmaf(kK3, t, -kK2)));
  if (KIND == CPLXAMD_KL_REAL_ARD) return 0.5f * softplus_f(t);
  if (KIND == CPLXAMD_KL_CPLX_VD) return cplx_vd_value(t);
  // extensions/complex.py:113-117: softplus(t) + 0.57810 sigmoid(1.36526 t - 1.45926)
  if (KIND == CPLXAMD_KL_CPLX_VD_APPROX)
    return fmaf(0.57810f, sigmoid_f(fmaf(1.36526f, t, -1.45926f)), softplus_f(t));
  // extensions/complex.py:43-46: log|w| - ls2 - Ei(-e^t)/2 = (f_vd(t) - gamma)/2 - ls2/2; the
  // -ls2/2 term is added by the kernel (kl_ls2_term)
  if (KIND == CPLXAMD_KL_CPLX_VD_SCALEFREE) return 0.5f * (cplx_vd_value(t) - kEulerGamma);
  // extensions/complex.py:142-160: the Ei term reads as 0 in the value, the slope is the exact one
  if (KIND == CPLXAMD_KL_CPLX_VD_BOGUS) return t;
  return softplus_f(t);
}
// part of the penalty that depends on log_sigma2 directly (not through t): c * ls2
template <int KIND> __device__ __forceinline__ constexpr float kl_ls2_term() {
  return KIND == CPLXAMD_KL_CPLX_VD_SCALEFREE ? -0.5f : 0.0f;
}

// f'(t)
template <int KIND>
__device__ __forceinline__ float kl_slope(float t) {
  if (KIND == CPLXAMD_KL_REAL_VD) {
    const float su = sigmoid_f(fmaf(kK3, t, -kK2));
    return fmaf(0.5f, softplus_grad_f(t), kK1 * kK3 * su * (1.0f - su));
  }
  if (KIND == CPLXAMD_KL_REAL_ARD) return 0.5f * softplus_grad_f(t);
  if (KIND == CPLXAMD_KL_CPLX_VD || KIND == CPLXAMD_KL_CPLX_VD_BOGUS) return one_minus_exp_neg(expf(t));  // 1 - exp(-e^t)
  if (KIND == CPLXAMD_KL_CPLX_VD_APPROX) {
    const float su = sigmoid_f(fmaf(1.36526f, t, -1.45926f));
    return fmaf(0.57810f * 1.36526f, su * (1.0f - su), softplus_grad_f(t));
  }
  if (KIND == CPLXAMD_KL_CPLX_VD_SCALEFREE) return 0.5f * one_minus_exp_neg(expf(t));
  return softplus_grad_f(t);
}

template <bool CPLX>
__device__ __forceinline__ void weight_grad(float fp, float wr, float wi, float theta,
                                            float& gwr, float& gwi) {
  // d(-log_alpha)/dw = 2 w / (theta (theta + 1e-12)); 0 at theta == 0 (torch subgradient)
  if (theta > 0.0f) {
    if (CPLX) {
      // (one hardware reciprocal, 1 ulp, instead of the correctly rounded division's ~10 instructions)
      const float c = 2.0f * fp * __builtin_amdgcn_rcpf(theta * (theta + 1e-12f));
      gwr = c * wr;
      gwi = c * wi;
    } else {
      // sign(w) * 2 fp / (|w| + 1e-12): fp carries the upstream gradient and may be negative
      gwr = (wr < 0.0f ? -2.0f : 2.0f) * fp / (theta + 1e-12f);
      gwi = 0.0f;
    }
  } else {
    gwr = 0.0f;
    gwi = 0.0f;
  }
}

// ---- exact complex KL (CPLXAMD_KL_CPLX_VD), fused value + slope, two elements per instruction -----------------------------
// The fused forward + backward kernel moves 24 B per element and was co-limited by ~140 VALU lane-operations per element
// (profiles/r03_kl_pmc.txt: 74 % VALU busy at 57 % of the HBM peak).  This form needs about a third of that:
//   * x = e^t straight from the operands: x = |w|^2 exp(-log_sigma2) -- no square root, no logarithm on the way to the
//     argument (t = 2 log(|w| + 1e-12) - log_sigma2; for |w|^2 >= 1e-8 the 1e-12 shifts t by < 2e-8, and the rare element
//     below that takes the original formulas);
//   * the value for x <= 1 is the series in x (needs no t), for x > 1 gamma + ln x + E1(x) with ONE v_log_f32;
//   * the slope 1 - exp(-x) shares exp(-x) with E1; below 2^-6 it is x - x^2/2 + x^3/6 (next term 1.6e-7 relative);
//   * d/dw = 2 f' w / |w|^2: one reciprocal of the |w|^2 that is there already;
//   * v_exp_f32 / v_log_f32 directly (base 2; the product with log2(e) carried to double-float accuracy where the exponent
//     argument is large), and every polynomial on float2 operands (v_pk_fma_f32: two elements per instruction).
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 sp2(float v) { return f2{v, v}; }
__device__ __forceinline__ f2 sel2(bool c0, bool c1, f2 a, f2 b) { return f2{c0 ? a.x : b.x, c1 ? a.y : b.y}; }

// exp(-ls) = 2^(-ls log2 e): hi + lo product, so that |ls| ~ 30 costs no accuracy (one v_exp_f32 + three fma / mul)
__device__ __forceinline__ float exp_neg(float ls) {
  const float yh = -ls * 1.44269504089f;
  const float yl = fmaf(-ls, 1.44269504089f, -yh);                 // exact remainder of the product
  const float e = __builtin_amdgcn_exp2f(yh);
  return fmaf(e, yl * 0.69314718056f, e);
}

// value f(x) = gamma + ln x + E1(x) and slope 1 - exp(-x) for x = e^t >= 0, two elements
__device__ __forceinline__ void cplx_vd_pair(f2 x, f2& val, f2& slope) {
  f2 s = sp2(3.0619244e-7f);                 // +1/(9*9!)   (first omitted term 2.8e-8 x^10)
  s = fma2(s, x, sp2(-3.1001984e-6f));       // -1/(8*8!)
  s = fma2(s, x, sp2(2.8344671e-5f));        // +1/(7*7!)
  s = fma2(s, x, sp2(-2.3148148e-4f));       // -1/(6*6!)
  s = fma2(s, x, sp2(1.6666667e-3f));        // +1/(5*5!)
  s = fma2(s, x, sp2(-1.0416667e-2f));       // -1/(4*4!)
  s = fma2(s, x, sp2(5.5555556e-2f));        // +1/(3*3!)
  s = fma2(s, x, sp2(-0.25f));               // -1/(2*2!)
  s = fma2(s, x, sp2(1.0f));
  const f2 small = s * x;
  // x > 1: E1(x) = e^-x / x * P4(x) / Q4(x)   (Abramowitz & Stegun 5.1.56, |err| < 2e-8)
  const f2 num = fma2(fma2(fma2(x + sp2(8.5733287401f), x, sp2(18.0590169730f)), x, sp2(8.6347608925f)), x, sp2(0.2677737343f));
  const f2 den = fma2(fma2(fma2(x + sp2(9.5733223454f), x, sp2(25.6329561486f)), x, sp2(21.0996530827f)), x, sp2(3.9584969228f));
  const f2 xd = x * den;
  const f2 nx = x * sp2(-1.44269504089f);
  const f2 e = f2{__builtin_amdgcn_exp2f(nx.x), __builtin_amdgcn_exp2f(nx.y)};                 // exp(-x)
  const f2 rc = f2{__builtin_amdgcn_rcpf(xd.x), __builtin_amdgcn_rcpf(xd.y)};
  const f2 lg = f2{__builtin_amdgcn_logf(x.x), __builtin_amdgcn_logf(x.y)} * sp2(0.69314718056f);   // ln x
  f2 e1 = e * num * rc;
  e1 = sel2(x.x < 104.0f, x.y < 104.0f, e1, sp2(0.0f));           // (beyond: 0 * inf)
  const f2 large = sp2(kEulerGamma) + lg + e1;
  val = sel2(x.x <= 1.0f, x.y <= 1.0f, small, large);
  const f2 ss = fma2(fma2(x, sp2(1.6666667e-1f), sp2(-0.5f)), x, sp2(1.0f)) * x;
  slope = sel2(x.x < 0.015625f, x.y < 0.015625f, ss, sp2(1.0f) - e);
}

struct KlArgs {
  const float* wr;
  const float* wi;
  const float* ls2;
  const float* g_elem;    // upstream per-element gradient (nullable)
  const float* g_scalar;  // upstream scalar gradient on device (nullable)
  float gscale;           // host scalar multiplier (used when both above are null)
  float* out_elem;
  double* partial;        // per-block partial sums (nullable)
  float* g_ls2;
  float* g_wr;
  float* g_wi;
  int64_t n;
  // fused per-step operand preparation of the bf16 LRT layers (all nullable; n % 4 == 0 required):
  // bf16 copies of the weight planes and exp(log_sigma2) for the mean / variance GEMMs
  bf16_t* wr_b = nullptr;
  bf16_t* wi_b = nullptr;
  bf16_t* s_b = nullptr;
};

__device__ __forceinline__ void st4g(float* p, const f4& v) {
#if KL_NT & 1
  typedef float f4v __attribute__((ext_vector_type(4)));
  __builtin_nontemporal_store(f4v{v.v[0], v.v[1], v.v[2], v.v[3]}, reinterpret_cast<f4v*>(p));
#else
  st4(p, v);
#endif
}
__device__ __forceinline__ f4 ld4g(const float* p) {
#if KL_NT & 2
  typedef float f4v __attribute__((ext_vector_type(4)));
  const f4v t = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p));
  return f4{{t.x, t.y, t.z, t.w}};
#else
  return ld4(p);
#endif
}

template <int KIND, bool VALUE, bool GRAD>
__global__ __launch_bounds__(kKlThreads) void kl_kernel(KlArgs a) {
  constexpr bool CPLX = KIND >= CPLXAMD_KL_CPLX_VD;
  constexpr float kLs = kl_ls2_term<KIND>();
  __shared__ double red[kKlThreads / 64];
  const int64_t n4 = a.n >> 2;
  const int64_t stride = (int64_t)gridDim.x * kKlThreads;
  float gs = a.gscale;
  if (GRAD && a.g_scalar) gs *= *a.g_scalar;
  double acc = 0.0;

  // one step of the grid-stride loop: the 4 elements at vector index i (operands already in registers)
  auto body = [&](int64_t i, const f4& wr, const f4& ls, const f4& wi, const f4& ge) __attribute__((always_inline)) {
    f4 val, d_ls, d_wr, d_wi;
    float part = 0.0f;
    bool done = false;
    if constexpr (KIND == CPLXAMD_KL_CPLX_VD) {
      // the two-elements-per-instruction form (cplx_vd_pair); a lane holding a weight with |w|^2 < 1e-8 -- where the
      // 1e-12 of log(|w| + 1e-12) matters, zero weights included -- redoes its four elements below
      float q[4];
      bool tiny = false;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        q[j] = fmaf(wi.v[j], wi.v[j], wr.v[j] * wr.v[j]);
        tiny |= !(q[j] >= 1e-8f);
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const f2 x = f2{q[2 * h] * exp_neg(ls.v[2 * h]), q[2 * h + 1] * exp_neg(ls.v[2 * h + 1])};
        f2 v, sl;
        cplx_vd_pair(x, v, sl);
        if (VALUE) { val.v[2 * h] = v.x; val.v[2 * h + 1] = v.y; }
        if (GRAD) {
          const f2 fp = sl * f2{ge.v[2 * h], ge.v[2 * h + 1]} * sp2(gs);
          const f2 c = (fp + fp) * f2{__builtin_amdgcn_rcpf(q[2 * h]), __builtin_amdgcn_rcpf(q[2 * h + 1])};
          d_ls.v[2 * h] = -fp.x; d_ls.v[2 * h + 1] = -fp.y;
          d_wr.v[2 * h] = c.x * wr.v[2 * h]; d_wr.v[2 * h + 1] = c.y * wr.v[2 * h + 1];
          d_wi.v[2 * h] = c.x * wi.v[2 * h]; d_wi.v[2 * h + 1] = c.y * wi.v[2 * h + 1];
        }
      }
      done = !tiny;
      if (VALUE && done) part = (val.v[0] + val.v[1]) + (val.v[2] + val.v[3]);
    }
    if (!done) {
      part = 0.0f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float theta;
        const float t = -log_alpha_of<CPLX, false>(ls.v[j], wr.v[j], wi.v[j], theta);
        if (VALUE) {
          val.v[j] = kl_value<KIND>(t) + kLs * ls.v[j];
          part += val.v[j];
        }
        if (GRAD) {
          const float fp = kl_slope<KIND>(t) * ge.v[j] * gs;
          d_ls.v[j] = kLs * ge.v[j] * gs - fp;
          weight_grad<CPLX>(fp, wr.v[j], wi.v[j], theta, d_wr.v[j], d_wi.v[j]);
        }
      }
    }
    if (VALUE) {
      acc += (double)part;
      if (a.out_elem) st4(a.out_elem + 4 * i, val);
    }
    if (GRAD) {
      if (a.g_ls2) st4g(a.g_ls2 + 4 * i, d_ls);
      if (a.g_wr) st4g(a.g_wr + 4 * i, d_wr);
      if (CPLX && a.g_wi) st4g(a.g_wi + 4 * i, d_wi);
    }
    if (a.wr_b) st4(a.wr_b + 4 * i, wr);
    if (CPLX && a.wi_b) st4(a.wi_b + 4 * i, wi);
    if (a.s_b) {
      f4 e;
#pragma unroll
      for (int j = 0; j < 4; ++j) e.v[j] = expf(ls.v[j]);
      st4(a.s_b + 4 * i, e);
    }
  };
  // KL_UNROLL vector indices per thread and iteration, every load issued before the first result is needed (the loop
  // is latency-bound otherwise: three dependent 16-byte loads per thread in flight against ~2 us of HBM latency)
  for (int64_t i0 = (int64_t)blockIdx.x * kKlThreads + threadIdx.x; i0 < n4; i0 += KL_UNROLL * stride) {
    f4 wr[KL_UNROLL], ls[KL_UNROLL], wi[KL_UNROLL], ge[KL_UNROLL];
#pragma unroll
    for (int u = 0; u < KL_UNROLL; ++u) {
      const int64_t i = i0 + u * stride;
      wi[u] = f4{{0.f, 0.f, 0.f, 0.f}};
      ge[u] = f4{{1.f, 1.f, 1.f, 1.f}};
      if (i < n4) {
        wr[u] = ld4g(a.wr + 4 * i);
        ls[u] = ld4g(a.ls2 + 4 * i);
        if (CPLX) wi[u] = ld4g(a.wi + 4 * i);
        if (GRAD && a.g_elem) ge[u] = ld4(a.g_elem + 4 * i);
      }
    }
#pragma unroll
    for (int u = 0; u < KL_UNROLL; ++u) {
      const int64_t i = i0 + u * stride;
      if (i < n4) body(i, wr[u], ls[u], wi[u], ge[u]);
    }
  }
  // scalar tail (n % 4 elements), handled by block 0
  if (blockIdx.x == 0) {
    const int64_t i = (n4 << 2) + threadIdx.x;
    if (i < a.n) {
      const float wr = a.wr[i], ls = a.ls2[i];
      const float wi = CPLX ? a.wi[i] : 0.0f;
      float theta;
      const float t = -log_alpha_of<CPLX, false>(ls, wr, wi, theta);
      if (VALUE) {
        const float v = kl_value<KIND>(t) + kLs * ls;
        acc += (double)v;
        if (a.out_elem) a.out_elem[i] = v;
      }
      if (GRAD) {
        const float ge = a.g_elem ? a.g_elem[i] : 1.0f;
        const float fp = kl_slope<KIND>(t) * ge * gs;
        float gwr, gwi;
        weight_grad<CPLX>(fp, wr, wi, theta, gwr, gwi);
        if (a.g_ls2) a.g_ls2[i] = kLs * ge * gs - fp;
        if (a.g_wr) a.g_wr[i] = gwr;
        if (CPLX && a.g_wi) a.g_wi[i] = gwi;
      }
    }
  }
  if (VALUE && a.partial) {
    const double s = block_sum<double, kKlThreads>(acc, red);
    if (threadIdx.x == 0) a.partial[blockIdx.x] = s;
  }
}

// one block: out = sum(partial[0..m))
__global__ __launch_bounds__(kKlThreads) void kl_final_kernel(const double* partial, int m,
                                                              float* out) {
  __shared__ double red[kKlThreads / 64];
  double acc = 0.0;
  for (int i = threadIdx.x; i < m; i += kKlThreads) acc += partial[i];
  const double s = block_sum<double, kKlThreads>(acc, red);
  if (threadIdx.x == 0) *out = (float)s;
}

template <bool VALUE, bool GRAD>
static int launch_kl(int kind, const KlArgs& a, int grid, hipStream_t st) {
  switch (kind) {
    case CPLXAMD_KL_REAL_VD:
      kl_kernel<CPLXAMD_KL_REAL_VD, VALUE, GRAD><<<grid, kKlThreads, 0, st>>>(a);
      break;
    case CPLXAMD_KL_REAL_ARD:
      kl_kernel<CPLXAMD_KL_REAL_ARD, VALUE, GRAD><<<grid, kKlThreads, 0, st>>>(a);
      break;
    case CPLXAMD_KL_CPLX_VD:
      kl_kernel<CPLXAMD_KL_CPLX_VD, VALUE, GRAD><<<grid, kKlThreads, 0, st>>>(a);
      break;
    case CPLXAMD_KL_CPLX_ARD:
      kl_kernel<CPLXAMD_KL_CPLX_ARD, VALUE, GRAD><<<grid, kKlThreads, 0, st>>>(a);
      break;
    case CPLXAMD_KL_CPLX_VD_APPROX:
      kl_kernel<CPLXAMD_KL_CPLX_VD_APPROX, VALUE, GRAD><<<grid, kKlThreads, 0, st>>>(a);
      break;
    case CPLXAMD_KL_CPLX_VD_SCALEFREE:
      kl_kernel<CPLXAMD_KL_CPLX_VD_SCALEFREE, VALUE, GRAD><<<grid, kKlThreads, 0, st>>>(a);
      break;
    case CPLXAMD_KL_CPLX_VD_BOGUS:
      kl_kernel<CPLXAMD_KL_CPLX_VD_BOGUS, VALUE, GRAD><<<grid, kKlThreads, 0, st>>>(a);
      break;
    default:
      return CPLXAMD_EINVAL;
  }
  CPLXAMD_CHECK_LAUNCH();
  return 0;
}

// operand preparation without the KL: bf16(wr), bf16(wi), bf16(exp(log_sigma2)) in one pass
__global__ __launch_bounds__(kKlThreads) void prep_kernel(const float* wr, const float* wi, const float* ls2,
                                                          bf16_t* wr_b, bf16_t* wi_b, bf16_t* s_b, int64_t n) {
  const int64_t n4 = n >> 2, stride = (int64_t)gridDim.x * kKlThreads;
  for (int64_t i = (int64_t)blockIdx.x * kKlThreads + threadIdx.x; i < n4; i += stride) {
    if (wr_b) st4(wr_b + 4 * i, ld4(wr + 4 * i));
    if (wi_b) st4(wi_b + 4 * i, ld4(wi + 4 * i));
    if (s_b) {
      f4 e = ld4(ls2 + 4 * i);
#pragma unroll
      for (int j = 0; j < 4; ++j) e.v[j] = expf(e.v[j]);
      st4(s_b + 4 * i, e);
    }
  }
}

static bool kl_args_ok(const float* wr, const float* wi, const float* ls2, int kind, int64_t n) {
  if (!wr || !ls2 || n < 0 || kind < 0 || kind > CPLXAMD_KL_CPLX_VD_BOGUS) return false;
  const bool cplx = kind >= CPLXAMD_KL_CPLX_VD;
  return cplx ? wi != nullptr : true;
}

// ---- log-alpha / mask ---------------------------------------------------------------------
template <bool CPLX, bool MASK>
__global__ __launch_bounds__(kKlThreads) void log_alpha_kernel(const float* wr, const float* wi,
                                                               const float* ls2, float thr,
                                                               float* out, double* partial,
                                                               int64_t n) {
  __shared__ double red[kKlThreads / 64];
  const int64_t n4 = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * kKlThreads;
  double cnt = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * kKlThreads + threadIdx.x; i < n4; i += stride) {
    const f4 a = ld4(wr + 4 * i), l = ld4(ls2 + 4 * i);
    f4 b = {{0.f, 0.f, 0.f, 0.f}};
    if (CPLX) b = ld4(wi + 4 * i);
    f4 o;
    float c = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float theta;
      const float la = log_alpha_of<CPLX, true>(l.v[j], a.v[j], b.v[j], theta);
      o.v[j] = MASK ? (la <= thr ? 1.0f : 0.0f) : la;
      c += MASK ? o.v[j] : 0.0f;
    }
    st4(out + 4 * i, o);
    cnt += (double)c;
  }
  if (blockIdx.x == 0) {
    const int64_t i = (n4 << 2) + threadIdx.x;
    if (i < n) {
      float theta;
      const float la = log_alpha_of<CPLX, true>(ls2[i], wr[i], CPLX ? wi[i] : 0.0f, theta);
      const float o = MASK ? (la <= thr ? 1.0f : 0.0f) : la;
      out[i] = o;
      if (MASK) cnt += (double)o;
    }
  }
  if (MASK && partial) {
    const double s = block_sum<double, kKlThreads>(cnt, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
  }
}

// gradient of log_alpha wrt the weight: d/dw (ls2 - 2 log(|w| + 1e-12)) * g  (d/d ls2 = g itself)
template <bool CPLX>
__global__ __launch_bounds__(kKlThreads) void log_alpha_bwd_kernel(const float* g, const float* wr,
                                                                   const float* wi, float* g_wr, float* g_wi,
                                                                   int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * kKlThreads;
  for (int64_t i = (int64_t)blockIdx.x * kKlThreads + threadIdx.x; i < n; i += stride) {
    const float a = wr[i], b = CPLX ? wi[i] : 0.0f;
    const float theta = weight_abs<CPLX, true>(a, b);
    float ga, gb;
    weight_grad<CPLX>(-g[i], a, b, theta, ga, gb);
    g_wr[i] = ga;
    if (CPLX) g_wi[i] = gb;
  }
}

__global__ __launch_bounds__(kKlThreads) void count_final_kernel(const double* partial, int m,
                                                                 int64_t* out) {
  __shared__ double red[kKlThreads / 64];
  double acc = 0.0;
  for (int i = threadIdx.x; i < m; i += kKlThreads) acc += partial[i];
  const double s = block_sum<double, kKlThreads>(acc, red);
  if (threadIdx.x == 0) *out = (int64_t)(s + 0.5);
}

// ---- Ei(x) for the torch_expi seam (both signs) ----------------------------------------------
// x < 0: Ei(x) = -E1(-x).  x > 0: power series gamma + ln x + sum x^k/(k k!) for x <= 40 (in
// fp64: the series terms alternate in magnitude but not in sign, no cancellation), asymptotic
// e^x/x * sum k!/x^k beyond.  Matches scipy (double compute, one rounding) to ~1e-7 relative.
__device__ __forceinline__ float expi_f(float xf) {
  const double x = (double)xf;
  if (xf == 0.0f) return -INFINITY;
  if (xf < 0.0f) {
    const double y = -x;
    if (y <= 1.0) {
      double s = 0.0, term = 1.0;
      for (int k = 1; k <= 20; ++k) {
        term *= -y / k;
        s += term / k;
      }
      return (float)(0.57721566490153286 + log(y) + s);
    }
    if (y > 745.0) return -0.0f;
    // continued fraction for E1 (modified Lentz), converges fast for y > 1
    double b = y + 1.0, c = 1e300, d = 1.0 / b, h = d;
    for (int i = 1; i <= 60; ++i) {
      const double an = -(double)i * i;
      b += 2.0;
      d = 1.0 / (an * d + b);
      c = b + an / c;
      const double del = c * d;
      h *= del;
      if (fabs(del - 1.0) < 1e-15) break;
    }
    return (float)(-h * exp(-y));
  }
  if (x <= 40.0) {
    double s = 0.0, term = 1.0;
    for (int k = 1; k <= 200; ++k) {
      term *= x / k;
      const double add = term / k;
      s += add;
      if (add < s * 1e-17) break;
    }
    return (float)(0.57721566490153286 + log(x) + s);
  }
  double s = 1.0, term = 1.0;
  for (int k = 1; k <= 40; ++k) {
    const double nt = term * k / x;
    if (nt > term) break;
    term = nt;
    s += term;
  }
  return (float)(exp(x) / x * s);
}

__global__ __launch_bounds__(kKlThreads) void expi_fwd_kernel(const float* x, float* y, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * kKlThreads;
  for (int64_t i = (int64_t)blockIdx.x * kKlThreads + threadIdx.x; i < n; i += stride)
    y[i] = expi_f(x[i]);
}
__global__ __launch_bounds__(kKlThreads) void expi_bwd_kernel(const float* g, const float* x,
                                                              float* gx, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * kKlThreads;
  for (int64_t i = (int64_t)blockIdx.x * kKlThreads + threadIdx.x; i < n; i += stride)
    gx[i] = g[i] * expf(x[i]) / x[i];  // nn/relevance/complex/vd.py:39-41
}

}  // namespace cplxamd

using namespace cplxamd;

extern "C" {

int cplxamd_abi_version(void) { return CPLXAMD_ABI_VERSION; }

int64_t cplxamd_vd_kl_ws_bytes(void) { return (int64_t)kKlMaxBlocks * sizeof(double); }

int cplxamd_vd_kl_fwd(const float* wr, const float* wi, const float* log_sigma2, int kind,
                      float* out_elem, float* out_sum, void* ws, int64_t n, void* stream) {
  if (!kl_args_ok(wr, wi, log_sigma2, kind, n) || (out_sum && !ws)) return CPLXAMD_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int grid = stream_grid(n >> 2, kKlThreads);
  KlArgs a{wr, wi, log_sigma2, nullptr, nullptr, 1.0f, out_elem,
           out_sum ? (double*)ws : nullptr, nullptr, nullptr, nullptr, n};
  int rc = launch_kl<true, false>(kind, a, grid, st);
  if (rc) return rc;
  if (out_sum) {
    kl_final_kernel<<<1, kKlThreads, 0, st>>>((const double*)ws, grid, out_sum);
    CPLXAMD_CHECK_LAUNCH();
  }
  return 0;
}

int cplxamd_vd_kl_bwd(const float* wr, const float* wi, const float* log_sigma2, int kind,
                      const float* g_elem, const float* g_scalar, float* g_log_sigma2,
                      float* g_wr, float* g_wi, int64_t n, void* stream) {
  if (!kl_args_ok(wr, wi, log_sigma2, kind, n)) return CPLXAMD_EINVAL;
  const int grid = stream_grid(n >> 2, kKlThreads);
  KlArgs a{wr, wi, log_sigma2, g_elem, g_scalar, 1.0f, nullptr, nullptr,
           g_log_sigma2, g_wr, g_wi, n};
  return launch_kl<false, true>(kind, a, grid, (hipStream_t)stream);
}

int cplxamd_vd_kl_fwd_bwd(const float* wr, const float* wi, const float* log_sigma2, int kind,
                          float gscale, float* out_sum, float* g_log_sigma2, float* g_wr,
                          float* g_wi, void* ws, int64_t n, void* stream) {
  if (!kl_args_ok(wr, wi, log_sigma2, kind, n) || !out_sum || !ws) return CPLXAMD_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int grid = stream_grid(n >> 2, kKlThreads);
  KlArgs a{wr, wi, log_sigma2, nullptr, nullptr, gscale, nullptr, (double*)ws,
           g_log_sigma2, g_wr, g_wi, n};
  int rc = launch_kl<true, true>(kind, a, grid, st);
  if (rc) return rc;
  kl_final_kernel<<<1, kKlThreads, 0, st>>>((const double*)ws, grid, out_sum);
  CPLXAMD_CHECK_LAUNCH();
  return 0;
}

int cplxamd_vd_prep_kl(const float* wr, const float* wi, const float* log_sigma2, int kind, int with_kl,
                       void* wr_bf16, void* wi_bf16, void* s_bf16, float* out_sum, float* g_log_sigma2,
                       float* g_wr, float* g_wi, void* ws, int64_t n, void* stream) {
  if (n < 0 || (n & 3)) return CPLXAMD_ESHAPE;
  if ((wr_bf16 && !wr) || (wi_bf16 && !wi) || (s_bf16 && !log_sigma2)) return CPLXAMD_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int grid = stream_grid(n >> 2, kKlThreads);
  if (!with_kl) {
    prep_kernel<<<grid, kKlThreads, 0, st>>>(wr, wi, log_sigma2, (bf16_t*)wr_bf16, (bf16_t*)wi_bf16,
                                             (bf16_t*)s_bf16, n);
    CPLXAMD_CHECK_LAUNCH();
    return 0;
  }
  if (!kl_args_ok(wr, wi, log_sigma2, kind, n) || !out_sum || !ws) return CPLXAMD_EINVAL;
  KlArgs a{wr, wi, log_sigma2, nullptr, nullptr, 1.0f, nullptr, (double*)ws,
           g_log_sigma2, g_wr, g_wi, n};
  a.wr_b = (bf16_t*)wr_bf16; a.wi_b = (bf16_t*)wi_bf16; a.s_b = (bf16_t*)s_bf16;
  int rc = launch_kl<true, true>(kind, a, grid, st);
  if (rc) return rc;
  kl_final_kernel<<<1, kKlThreads, 0, st>>>((const double*)ws, grid, out_sum);
  CPLXAMD_CHECK_LAUNCH();
  return 0;
}

int cplxamd_vd_log_alpha(const float* wr, const float* wi, const float* log_sigma2, float* out,
                         int64_t n, void* stream) {
  if (!wr || !log_sigma2 || !out || n < 0) return CPLXAMD_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int grid = stream_grid(n >> 2, kKlThreads);
  if (wi)
    log_alpha_kernel<true, false><<<grid, kKlThreads, 0, st>>>(wr, wi, log_sigma2, 0.f, out,
                                                               nullptr, n);
  else
    log_alpha_kernel<false, false><<<grid, kKlThreads, 0, st>>>(wr, wi, log_sigma2, 0.f, out,
                                                                nullptr, n);
  CPLXAMD_CHECK_LAUNCH();
  return 0;
}

int cplxamd_vd_log_alpha_bwd(const float* g, const float* wr, const float* wi, float* g_wr, float* g_wi,
                             int64_t n, void* stream) {
  if (!g || !wr || !g_wr || n < 0 || ((wi == nullptr) != (g_wi == nullptr))) return CPLXAMD_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int grid = stream_grid(n, kKlThreads);
  if (wi) log_alpha_bwd_kernel<true><<<grid, kKlThreads, 0, st>>>(g, wr, wi, g_wr, g_wi, n);
  else log_alpha_bwd_kernel<false><<<grid, kKlThreads, 0, st>>>(g, wr, wi, g_wr, g_wi, n);
  CPLXAMD_CHECK_LAUNCH();
  return 0;
}

int cplxamd_vd_mask(const float* wr, const float* wi, const float* log_sigma2, float threshold,
                    float* mask, int64_t* count, void* ws, int64_t n, void* stream) {
  if (!wr || !log_sigma2 || !mask || n < 0 || (count && !ws)) return CPLXAMD_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int grid = stream_grid(n >> 2, kKlThreads);
  double* partial = count ? (double*)ws : nullptr;
  if (wi)
    log_alpha_kernel<true, true><<<grid, kKlThreads, 0, st>>>(wr, wi, log_sigma2, threshold,
                                                              mask, partial, n);
  else
    log_alpha_kernel<false, true><<<grid, kKlThreads, 0, st>>>(wr, wi, log_sigma2, threshold,
                                                               mask, partial, n);
  CPLXAMD_CHECK_LAUNCH();
  if (count) {
    count_final_kernel<<<1, kKlThreads, 0, st>>>(partial, grid, count);
    CPLXAMD_CHECK_LAUNCH();
  }
  return 0;
}

int cplxamd_expi_fwd(const float* x, float* y, int64_t n, void* stream) {
  if (!x || !y || n < 0) return CPLXAMD_EINVAL;
  expi_fwd_kernel<<<stream_grid(n, kKlThreads), kKlThreads, 0, (hipStream_t)stream>>>(x, y, n);
  CPLXAMD_CHECK_LAUNCH();
  return 0;
}

int cplxamd_expi_bwd(const float* g, const float* x, float* gx, int64_t n, void* stream) {
  if (!g || !x || !gx || n < 0) return CPLXAMD_EINVAL;
  expi_bwd_kernel<<<stream_grid(n, kKlThreads), kKlThreads, 0, (hipStream_t)stream>>>(g, x, gx, n);
  CPLXAMD_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
