// float64 contractions of the path (the reference's `.double()` models: /root/reference/tests/test_modules.py:88-127,
// tests/test_cplx.py:58-61): complex / real GEMM on strided operands, complex / real 2-d convolution with its two
// gradients, and the exponential integral.  A PARITY mode, not a tuned one: v_fma_f64 loops through LDS tiles (the float64
// MFMA and the vector pipe of gfx950 peak at the same 78.6 TFLOP/s, so plain FMA code gives nothing away structurally), one
// thread per output element for the convolutions.  The elementwise algebra of the float64 layers is torch's, under autograd,
// exactly as in the reference (cplxmodule_amd/f64.py); what runs here is what torch would hand to a vendor library.
//   GEMM        cplx.py:634-648 (linear_naive), :167-174 (__matmul__) and their autograd (A.1)
//   convolution cplx.py:717-838 (convnd_naive / convnd_quick: cross-correlation without conjugation) and its autograd
//   Ei          nn/relevance/complex/vd.py:15-44 (scipy.special.expi on the host there)
#include "common.h"

namespace cplxamd {
namespace f64k {

constexpr int T = 32;      // output tile edge; 256 threads, 2 x 2 outputs each

struct GemmArgs {
  const double* a_r; const double* a_i; int64_t a_rs, a_cs, a_bs;
  const double* b_r; const double* b_i; int64_t b_rs, b_cs, b_bs;
  const double* bias_r; const double* bias_i;
  double* c_r; double* c_i; int64_t ldc, c_bs;
  int M, N, K, conj_b;
};

// C[m, n] = sum_k A[m, k] op(B[n, k]) (+ bias[n]); planes a_i / b_i / c_i nullptr: real
template <bool CPLX>
__global__ __launch_bounds__(256) void gemm_f64_kernel(GemmArgs g) {
  __shared__ double sa[CPLX ? 2 : 1][T][T + 1], sb[CPLX ? 2 : 1][T][T + 1];
  const int z = blockIdx.z;
  const double* ar = g.a_r + z * g.a_bs;
  const double* br = g.b_r + z * g.b_bs;
  const double* ai = CPLX ? g.a_i + z * g.a_bs : nullptr;
  const double* bi = CPLX ? g.b_i + z * g.b_bs : nullptr;
  const int m0 = blockIdx.y * T, n0 = blockIdx.x * T;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;      // outputs (m0 + ty + 16 i, n0 + tx + 16 j)
  double cr[2][2] = {{0, 0}, {0, 0}}, ci[2][2] = {{0, 0}, {0, 0}};
  for (int k0 = 0; k0 < g.K; k0 += T) {
    for (int e = threadIdx.x; e < T * T; e += 256) {
      const int r = e / T, k = e % T;
      const bool ka = k0 + k < g.K;
      const bool oa = ka && m0 + r < g.M, ob = ka && n0 + r < g.N;
      const int64_t ia = (int64_t)(m0 + r) * g.a_rs + (int64_t)(k0 + k) * g.a_cs;
      const int64_t ib = (int64_t)(n0 + r) * g.b_rs + (int64_t)(k0 + k) * g.b_cs;
      sa[0][r][k] = oa ? ar[ia] : 0.0;
      sb[0][r][k] = ob ? br[ib] : 0.0;
      if (CPLX) {
        sa[1][r][k] = oa ? ai[ia] : 0.0;
        sb[1][r][k] = ob ? (g.conj_b ? -bi[ib] : bi[ib]) : 0.0;
      }
    }
    __syncthreads();
#pragma unroll 4
    for (int k = 0; k < T; ++k)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const double xr = sa[0][ty + 16 * i][k], wr = sb[0][tx + 16 * j][k];
          if (CPLX) {
            const double xi = sa[1][ty + 16 * i][k], wi = sb[1][tx + 16 * j][k];
            cr[i][j] = fma(xr, wr, cr[i][j]); cr[i][j] = fma(-xi, wi, cr[i][j]);
            ci[i][j] = fma(xr, wi, ci[i][j]); ci[i][j] = fma(xi, wr, ci[i][j]);
          } else {
            cr[i][j] = fma(xr, wr, cr[i][j]);
          }
        }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int m = m0 + ty + 16 * i, n = n0 + tx + 16 * j;
      if (m < g.M && n < g.N) {
        const int64_t o = z * g.c_bs + (int64_t)m * g.ldc + n;
        g.c_r[o] = cr[i][j] + (g.bias_r ? g.bias_r[n] : 0.0);
        if (CPLX) g.c_i[o] = ci[i][j] + (g.bias_i ? g.bias_i[n] : 0.0);
      }
    }
}

struct ConvP { int B, Ci, Co, H, W, KH, KW, sh, sw, ph, pw, dh, dw, groups, Ho, Wo; };

// mode 0: y[b, co, ho, wo] = sum x[b, ci, ho sh - ph + kh dh, wo sw - pw + kw dw] w[co, ci', kh, kw] (+ bias)
// mode 1: dx[b, ci, h, w]  = sum g[b, co, ho, wo] conj(w[co, ci', kh, kw])  over the taps that reach (h, w)
// mode 2: dw[co, ci', kh, kw] = sum g[b, co, ho, wo] conj(x[b, ci, h, w])
// one thread per output element; imaginary planes nullptr: real
template <int MODE>
__global__ __launch_bounds__(256) void conv_f64_kernel(const double* pr, const double* pi, const double* qr, const double* qi,
                                                       const double* bias_r, const double* bias_i, double* or_, double* oi,
                                                       ConvP p, int64_t total) {
  const bool cplx = pi != nullptr;
  const int cig = p.Ci / p.groups, cog = p.Co / p.groups;
  for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (int64_t)gridDim.x * 256) {
    double sr = 0.0, si = 0.0;
    if (MODE == 0) {            // p = x, q = w
      int64_t t = o;
      const int wo = (int)(t % p.Wo); t /= p.Wo;
      const int ho = (int)(t % p.Ho); t /= p.Ho;
      const int co = (int)(t % p.Co); const int b = (int)(t / p.Co);
      const int grp = co / cog;
      for (int c = 0; c < cig; ++c)
        for (int kh = 0; kh < p.KH; ++kh) {
          const int h = ho * p.sh - p.ph + kh * p.dh;
          if (h < 0 || h >= p.H) continue;
          for (int kw = 0; kw < p.KW; ++kw) {
            const int w = wo * p.sw - p.pw + kw * p.dw;
            if (w < 0 || w >= p.W) continue;
            const int64_t ix = (((int64_t)b * p.Ci + grp * cig + c) * p.H + h) * p.W + w;
            const int64_t iw = (((int64_t)co * cig + c) * p.KH + kh) * p.KW + kw;
            const double xr = pr[ix], wr = qr[iw];
            if (cplx) {
              const double xi = pi[ix], wi = qi[iw];
              sr = fma(xr, wr, sr); sr = fma(-xi, wi, sr);
              si = fma(xr, wi, si); si = fma(xi, wr, si);
            } else {
              sr = fma(xr, wr, sr);
            }
          }
        }
      if (bias_r) { sr += bias_r[co]; if (cplx) si += bias_i[co]; }
    } else if (MODE == 1) {     // p = g, q = w
      int64_t t = o;
      const int w = (int)(t % p.W); t /= p.W;
      const int h = (int)(t % p.H); t /= p.H;
      const int ci = (int)(t % p.Ci); const int b = (int)(t / p.Ci);
      const int grp = ci / cig, c = ci % cig;
      for (int kh = 0; kh < p.KH; ++kh) {
        const int hh = h + p.ph - kh * p.dh;
        if (hh < 0 || hh % p.sh) continue;
        const int ho = hh / p.sh;
        if (ho >= p.Ho) continue;
        for (int kw = 0; kw < p.KW; ++kw) {
          const int ww = w + p.pw - kw * p.dw;
          if (ww < 0 || ww % p.sw) continue;
          const int wo = ww / p.sw;
          if (wo >= p.Wo) continue;
          for (int k = 0; k < cog; ++k) {
            const int co = grp * cog + k;
            const int64_t ig = (((int64_t)b * p.Co + co) * p.Ho + ho) * p.Wo + wo;
            const int64_t iw = (((int64_t)co * cig + c) * p.KH + kh) * p.KW + kw;
            const double gr = pr[ig], wr = qr[iw];
            if (cplx) {       // g conj(w)
              const double gi = pi[ig], wi = qi[iw];
              sr = fma(gr, wr, sr); sr = fma(gi, wi, sr);
              si = fma(gi, wr, si); si = fma(-gr, wi, si);
            } else {
              sr = fma(gr, wr, sr);
            }
          }
        }
      }
    } else {                    // p = g, q = x
      int64_t t = o;
      const int kw = (int)(t % p.KW); t /= p.KW;
      const int kh = (int)(t % p.KH); t /= p.KH;
      const int c = (int)(t % cig); const int co = (int)(t / cig);
      const int grp = co / cog;
      for (int b = 0; b < p.B; ++b)
        for (int ho = 0; ho < p.Ho; ++ho) {
          const int h = ho * p.sh - p.ph + kh * p.dh;
          if (h < 0 || h >= p.H) continue;
          for (int wo = 0; wo < p.Wo; ++wo) {
            const int w = wo * p.sw - p.pw + kw * p.dw;
            if (w < 0 || w >= p.W) continue;
            const int64_t ig = (((int64_t)b * p.Co + co) * p.Ho + ho) * p.Wo + wo;
            const int64_t ix = (((int64_t)b * p.Ci + grp * cig + c) * p.H + h) * p.W + w;
            const double gr = pr[ig], xr = qr[ix];
            if (cplx) {       // g conj(x)
              const double gi = pi[ig], xi = qi[ix];
              sr = fma(gr, xr, sr); sr = fma(gi, xi, sr);
              si = fma(gi, xr, si); si = fma(-gr, xi, si);
            } else {
              sr = fma(gr, xr, sr);
            }
          }
        }
    }
    or_[o] = sr;
    if (cplx) oi[o] = si;
  }
}

// Ei(x), both signs, in float64: x < 0: -E1(-x) (power series up to 1, modified-Lentz continued fraction beyond); x > 0:
// power series to 40, asymptotic series beyond.  Agrees with scipy.special.expi to ~1e-15 relative.
__device__ double expi_d(double x) {
  if (x == 0.0) return -INFINITY;
  if (x != x) return x;
  if (x < 0.0) {
    const double y = -x;
    if (y <= 1.0) {
      double s = 0.0, term = 1.0;
      for (int k = 1; k <= 30; ++k) {
        term *= -y / k;
        s += term / k;
      }
      return 0.57721566490153286 + log(y) + s;
    }
    if (y > 745.0) return -0.0;
    double b = y + 1.0, c = 1e300, d = 1.0 / b, h = d;
    for (int i = 1; i <= 200; ++i) {
      const double an = -(double)i * i;
      b += 2.0;
      d = 1.0 / (an * d + b);
      c = b + an / c;
      const double del = c * d;
      h *= del;
      if (fabs(del - 1.0) < 1e-16) break;
    }
    return -h * exp(-y);
  }
  if (x <= 40.0) {
    double s = 0.0, term = 1.0;
    for (int k = 1; k <= 300; ++k) {
      term *= x / k;
      const double add = term / k;
      s += add;
      if (add < s * 1e-18) break;
    }
    return 0.57721566490153286 + log(x) + s;
  }
  double s = 1.0, term = 1.0;
  for (int k = 1; k <= 60; ++k) {
    const double nt = term * k / x;
    if (nt > term) break;
    term = nt;
    s += term;
  }
  return exp(x) / x * s;
}

__global__ __launch_bounds__(256) void expi_f64_kernel(const double* x, double* y, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) y[i] = expi_d(x[i]);
}

}  // namespace f64k
}  // namespace cplxamd

using namespace cplxamd;

extern "C" {

int cplxamd_gemm_f64(const double* a_r, const double* a_i, int64_t a_rs, int64_t a_cs, int64_t a_bs,
                     const double* b_r, const double* b_i, int64_t b_rs, int64_t b_cs, int64_t b_bs,
                     const double* bias_r, const double* bias_i, double* c_r, double* c_i, int64_t ldc, int64_t c_bs,
                     int batch, int M, int N, int K, int conj_b, void* stream) {
  if (!a_r || !b_r || !c_r || batch < 0 || M < 0 || N < 0 || K < 0 || ldc < N) return CPLXAMD_EINVAL;
  const bool cplx = a_i != nullptr;
  if (cplx != (b_i != nullptr) || cplx != (c_i != nullptr) || (bias_i && !cplx)) return CPLXAMD_EINVAL;
  if (batch == 0 || M == 0 || N == 0) return 0;
  if (batch > 65535) return CPLXAMD_ESHAPE;
  f64k::GemmArgs g{a_r, a_i, a_rs, a_cs, a_bs, b_r, b_i, b_rs, b_cs, b_bs, bias_r, bias_i, c_r, c_i, ldc, c_bs, M, N, K,
                   conj_b ? 1 : 0};
  const dim3 grid((N + f64k::T - 1) / f64k::T, (M + f64k::T - 1) / f64k::T, batch);
  if (cplx) f64k::gemm_f64_kernel<true><<<grid, 256, 0, (hipStream_t)stream>>>(g);
  else f64k::gemm_f64_kernel<false><<<grid, 256, 0, (hipStream_t)stream>>>(g);
  CPLXAMD_CHECK_LAUNCH();
  return 0;
}

/* mode 0 forward (p = x, q = w, out = y), 1 data gradient (p = g, q = w, out = dx), 2 weight gradient (p = g, q = x,
 * out = dw); geom as cplxamd_conv2d_fwd: {B, Ci, Co, H, W, KH, KW, sh, sw, ph, pw, dh, dw, groups}. */
int cplxamd_conv2d_f64(const double* p_r, const double* p_i, const double* q_r, const double* q_i, const double* bias_r,
                       const double* bias_i, double* out_r, double* out_i, const int* geom, int mode, void* stream) {
  if (!p_r || !q_r || !out_r || !geom || mode < 0 || mode > 2) return CPLXAMD_EINVAL;
  const bool cplx = p_i != nullptr;
  if (cplx != (q_i != nullptr) || cplx != (out_i != nullptr)) return CPLXAMD_EINVAL;
  f64k::ConvP p{geom[0], geom[1], geom[2], geom[3], geom[4], geom[5], geom[6], geom[7], geom[8], geom[9], geom[10], geom[11],
                geom[12], geom[13], 0, 0};
  if (p.B < 0 || p.Ci <= 0 || p.Co <= 0 || p.groups <= 0 || p.Ci % p.groups || p.Co % p.groups || p.sh <= 0 || p.sw <= 0 ||
      p.dh <= 0 || p.dw <= 0 || p.KH <= 0 || p.KW <= 0)
    return CPLXAMD_EINVAL;
  p.Ho = (p.H + 2 * p.ph - p.dh * (p.KH - 1) - 1) / p.sh + 1;
  p.Wo = (p.W + 2 * p.pw - p.dw * (p.KW - 1) - 1) / p.sw + 1;
  if (p.Ho <= 0 || p.Wo <= 0) return CPLXAMD_ESHAPE;
  const int64_t total = mode == 0 ? (int64_t)p.B * p.Co * p.Ho * p.Wo
                        : mode == 1 ? (int64_t)p.B * p.Ci * p.H * p.W
                                    : (int64_t)p.Co * (p.Ci / p.groups) * p.KH * p.KW;
  if (total == 0) return 0;
  const int grid = stream_grid(total, 256);
  hipStream_t st = (hipStream_t)stream;
  if (mode == 0) f64k::conv_f64_kernel<0><<<grid, 256, 0, st>>>(p_r, p_i, q_r, q_i, bias_r, bias_i, out_r, out_i, p, total);
  else if (mode == 1) f64k::conv_f64_kernel<1><<<grid, 256, 0, st>>>(p_r, p_i, q_r, q_i, nullptr, nullptr, out_r, out_i, p, total);
  else f64k::conv_f64_kernel<2><<<grid, 256, 0, st>>>(p_r, p_i, q_r, q_i, nullptr, nullptr, out_r, out_i, p, total);
  CPLXAMD_CHECK_LAUNCH();
  return 0;
}

int cplxamd_expi_f64(const double* x, double* y, int64_t n, void* stream) {
  if (!x || !y || n < 0) return CPLXAMD_EINVAL;
  if (n == 0) return 0;
  f64k::expi_f64_kernel<<<stream_grid(n, 256), 256, 0, (hipStream_t)stream>>>(x, y, n);
  CPLXAMD_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
