// K2: complex / real 2-d convolution (cross-correlation, NCHW, groups, stride, padding,
// dilation) forward, dgrad and wgrad as implicit GEMMs on the exact-f32 matrix cores:
// operand tiles are gathered on the fly (im2col is never materialised) into LDS and
// contracted with v_mfma_f32_32x32x2_f32; complex = 4 MFMA chains sharing one K-loop.
//
//   FWD    Y[b,co,oh,ow]  = sum_{ci,kh,kw} X[b,ci,oh*s-p+kh*d, ..] * W[co,ci,kh,kw] (+ bias)
//          GEMM: M = Co/g, N = B*Ho*Wo (pixels, coalesced stores), K = Ci/g*KH*KW
//   DGRAD  dX[b,ci,ih,iw] = sum_{co,kh,kw} G[b,co,oh,ow] * conj(W[co,ci,kh,kw]),  oh = (ih+p-kh*d)/s
//          GEMM: M = Ci/g, N = B*H*W, K = Co/g*KH*KW
//   WGRAD  dW[co,ci,kh,kw] = sum_{b,oh,ow} G[b,co,oh,ow] * conj(X[b,ci,oh*s-p+kh*d, ..])
//          GEMM: M = Co/g, N = Ci/g*KH*KW, K = B*Ho*Wo, split-K into fp32 partial slabs that a
//          second kernel sums (deterministic, no atomics).
//
// Reference: cplx.convnd / convnd_quick / convnd_naive, cplxmodule/cplx.py:717-800 (no conjugation
// in the forward), and the LRT variance conv nn/relevance/complex/base.py:125-133.
// This is the parity-first kernel (round 1); the bf16-MFMA tiling is the next step.
#include "common.h"

namespace cplxamd {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int CBM = 64, CBN = 64, CBK = 16, CLD = CBM + 1;

struct ConvP {
  int B, Ci, Co, H, W, KH, KW, Ho, Wo, sh, sw, ph, pw, dh, dw, G;
  int Cg, Cog;   // channels per group (in / out)
};

enum { MODE_FWD = 0, MODE_DGRAD = 1, MODE_WGRAD = 2 };

struct ConvArgs {
  const void* xr; const void* xi;     // FWD: input      DGRAD: grad out   WGRAD: grad out
  const void* wr; const void* wi;     // FWD: weight     DGRAD: weight     WGRAD: input
  const float* bias_r; const float* bias_i;
  void* yr; void* yi;                 // output (WGRAD: fp32 partial slabs [splits][Co*Cg*KH*KW])
  ConvP p;
  int64_t M, N, K;                    // GEMM dims per group
  int splits;                         // WGRAD split-K factor (1 otherwise)
  int64_t kchunk;                     // K elements per split
};

template <typename T>
__device__ __forceinline__ float ldv(const void* p, int64_t off) {
  return io<T>::ld(reinterpret_cast<const T*>(p) + off);
}

// ---- operand fetchers: return the plane offset of element (row, k) or -1 if it is padding --
template <int MODE>
__device__ __forceinline__ int64_t a_offset(const ConvP& p, int g, int64_t m, int64_t k) {
  if (MODE == MODE_FWD) {          // W[(g*Cog+m), k]
    return ((int64_t)g * p.Cog + m) * ((int64_t)p.Cg * p.KH * p.KW) + k;
  } else if (MODE == MODE_DGRAD) { // W[(g*Cog+co), ci=m, kh, kw], k = (co, kh, kw)
    const int khw = p.KH * p.KW;
    const int co = (int)(k / khw), r = (int)(k - (int64_t)co * khw);
    return (((int64_t)g * p.Cog + co) * p.Cg + m) * khw + r;
  } else {                         // G[b, g*Cog+m, oh, ow], k = (b, oh, ow)
    const int64_t hw = (int64_t)p.Ho * p.Wo;
    const int64_t b = k / hw, r = k - b * hw;
    return ((b * p.Co + (int64_t)g * p.Cog + m) * hw) + r;
  }
}

template <int MODE>
__device__ __forceinline__ int64_t b_offset(const ConvP& p, int g, int64_t n, int64_t k) {
  if (MODE == MODE_FWD) {          // n = (b, oh, ow), k = (ci, kh, kw) -> X
    const int64_t hw = (int64_t)p.Ho * p.Wo;
    const int64_t b = n / hw;
    const int r = (int)(n - b * hw);
    const int oh = r / p.Wo, ow = r - oh * p.Wo;
    const int khw = p.KH * p.KW;
    const int ci = (int)(k / khw), rk = (int)(k - (int64_t)ci * khw);
    const int kh = rk / p.KW, kw = rk - kh * p.KW;
    const int ih = oh * p.sh - p.ph + kh * p.dh, iw = ow * p.sw - p.pw + kw * p.dw;
    if (ih < 0 || ih >= p.H || iw < 0 || iw >= p.W) return -1;
    return ((b * p.Ci + (int64_t)g * p.Cg + ci) * p.H + ih) * p.W + iw;
  } else if (MODE == MODE_DGRAD) { // n = (b, ih, iw), k = (co, kh, kw) -> G
    const int64_t hw = (int64_t)p.H * p.W;
    const int64_t b = n / hw;
    const int r = (int)(n - b * hw);
    const int ih = r / p.W, iw = r - ih * p.W;
    const int khw = p.KH * p.KW;
    const int co = (int)(k / khw), rk = (int)(k - (int64_t)co * khw);
    const int kh = rk / p.KW, kw = rk - kh * p.KW;
    const int th = ih + p.ph - kh * p.dh, tw = iw + p.pw - kw * p.dw;
    if (th < 0 || tw < 0 || th % p.sh || tw % p.sw) return -1;
    const int oh = th / p.sh, ow = tw / p.sw;
    if (oh >= p.Ho || ow >= p.Wo) return -1;
    return ((b * p.Co + (int64_t)g * p.Cog + co) * p.Ho + oh) * p.Wo + ow;
  } else {                         // n = (ci, kh, kw), k = (b, oh, ow) -> X
    const int khw = p.KH * p.KW;
    const int ci = (int)(n / khw), rk = (int)(n - (int64_t)ci * khw);
    const int kh = rk / p.KW, kw = rk - kh * p.KW;
    const int64_t hw = (int64_t)p.Ho * p.Wo;
    const int64_t b = k / hw;
    const int r = (int)(k - b * hw);
    const int oh = r / p.Wo, ow = r - oh * p.Wo;
    const int ih = oh * p.sh - p.ph + kh * p.dh, iw = ow * p.sw - p.pw + kw * p.dw;
    if (ih < 0 || ih >= p.H || iw < 0 || iw >= p.W) return -1;
    return ((b * p.Ci + (int64_t)g * p.Cg + ci) * p.H + ih) * p.W + iw;
  }
}

// One thread's 4 elements of a [64 x CBK] operand tile: gather global -> registers (fetch), then
// registers -> LDS as d[k][r] (commit).  Split so that the gathers of tile t+1 are in flight while
// the MFMAs of tile t run (the K loop is double-buffered in LDS).
struct ConvRegs { float r[4], i[4]; };

template <typename T, bool CPLX, int MODE, bool IS_A, bool KFAST>
__device__ __forceinline__ ConvRegs fetch(const void* sr, const void* si, const ConvP& p, int g,
                                          int64_t row0, int64_t rows, int64_t k0, int64_t kend) {
  const int t = threadIdx.x;
  ConvRegs o;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int r, k;
    if (KFAST) { k = t & 15; r = (t >> 4) + 16 * j; }
    else { r = t & 63; k = (t >> 6) + 4 * j; }
    const int64_t gr = row0 + r, gk = k0 + k;
    float vr = 0.f, vi = 0.f;
    if (gr < rows && gk < kend) {
      const int64_t off = IS_A ? a_offset<MODE>(p, g, gr, gk) : b_offset<MODE>(p, g, gr, gk);
      if (off >= 0) {
        vr = ldv<T>(sr, off);
        if (CPLX) vi = ldv<T>(si, off);
      }
    }
    o.r[j] = vr;
    o.i[j] = vi;
  }
  return o;
}

template <bool CPLX, bool KFAST>
__device__ __forceinline__ void commit(float (*dr)[CLD], float (*di)[CLD], const ConvRegs& o) {
  const int t = threadIdx.x;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int r, k;
    if (KFAST) { k = t & 15; r = (t >> 4) + 16 * j; }
    else { r = t & 63; k = (t >> 6) + 4 * j; }
    dr[k][r] = o.r[j];
    if (CPLX) di[k][r] = o.i[j];
  }
}

// T: element type of the activations / gradients; TW: element type of the weight operand
template <typename T, bool CPLX, int MODE>
__global__ __launch_bounds__(256) void conv_kernel(ConvArgs a) {
  __shared__ float As_r[2][CBK][CLD], Bs_r[2][CBK][CLD];
  __shared__ float As_i[CPLX ? 2 : 1][CPLX ? CBK : 1][CLD], Bs_i[CPLX ? 2 : 1][CPLX ? CBK : 1][CLD];
  const ConvP& p = a.p;
  const int g = blockIdx.z / a.splits, split = blockIdx.z % a.splits;
  const int64_t m0 = (int64_t)blockIdx.y * CBM, n0 = (int64_t)blockIdx.x * CBN;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int wm = (wid >> 1) * 32, wn = (wid & 1) * 32, l31 = lane & 31, lk = lane >> 5;
  const int64_t kbeg = (int64_t)split * a.kchunk;
  int64_t kend = kbeg + a.kchunk;
  if (kend > a.K) kend = a.K;
  // conjugation: DGRAD conj(W) is the A operand, WGRAD conj(X) is the B operand
  const float sa = (MODE == MODE_DGRAD) ? -1.f : 1.f, sb = (MODE == MODE_WGRAD) ? -1.f : 1.f;
  constexpr bool BK_FAST = MODE == MODE_WGRAD;   // B: pixels fastest (FWD, DGRAD) or k (= pixels) fastest

  f32x16 acc_r = {0}, acc_i = {0};
  ConvRegs ra, rb;
  auto fetch_both = [&](int64_t k0) {
    // A: weights (FWD: k contiguous; DGRAD: gathered) / grad-out (WGRAD: k contiguous)
    ra = fetch<T, CPLX, MODE, true, true>(MODE == MODE_WGRAD ? a.xr : a.wr, MODE == MODE_WGRAD ? a.xi : a.wi,
                                          p, g, m0, a.M, k0, kend);
    if (MODE == MODE_WGRAD) rb = fetch<T, CPLX, MODE, false, true>(a.wr, a.wi, p, g, n0, a.N, k0, kend);
    else rb = fetch<T, CPLX, MODE, false, false>(a.xr, a.xi, p, g, n0, a.N, k0, kend);
  };
  if (kbeg < kend) fetch_both(kbeg);
  int buf = 0;
  for (int64_t k0 = kbeg; k0 < kend; k0 += CBK, buf ^= 1) {
    commit<CPLX, true>(As_r[buf], As_i[CPLX ? buf : 0], ra);
    commit<CPLX, BK_FAST>(Bs_r[buf], Bs_i[CPLX ? buf : 0], rb);
    __syncthreads();                         // tile visible; the other buffer is free again
    if (k0 + CBK < kend) fetch_both(k0 + CBK);
#pragma unroll
    for (int kk = 0; kk < CBK; kk += 2) {
      const float ar = As_r[buf][kk + lk][wm + l31];
      const float br = Bs_r[buf][kk + lk][wn + l31];
      acc_r = __builtin_amdgcn_mfma_f32_32x32x2f32(ar, br, acc_r, 0, 0, 0);
      if (CPLX) {
        const float ai = sa * As_i[buf][kk + lk][wm + l31];
        const float bi = sb * Bs_i[buf][kk + lk][wn + l31];
        acc_r = __builtin_amdgcn_mfma_f32_32x32x2f32(-ai, bi, acc_r, 0, 0, 0);
        acc_i = __builtin_amdgcn_mfma_f32_32x32x2f32(ar, bi, acc_i, 0, 0, 0);
        acc_i = __builtin_amdgcn_mfma_f32_32x32x2f32(ai, br, acc_i, 0, 0, 0);
      }
    }
  }

  // epilogue: col = lane & 31 runs along N, rows (M) across registers
  const int64_t n = n0 + wn + l31;
  if (n >= a.N) return;
  int64_t out_base, out_mstride;
  if (MODE == MODE_FWD) {
    const int64_t hw = (int64_t)p.Ho * p.Wo;
    const int64_t b = n / hw, r = n - b * hw;
    out_base = (b * p.Co + (int64_t)g * p.Cog) * hw + r;
    out_mstride = hw;
  } else if (MODE == MODE_DGRAD) {
    const int64_t hw = (int64_t)p.H * p.W;
    const int64_t b = n / hw, r = n - b * hw;
    out_base = (b * p.Ci + (int64_t)g * p.Cg) * hw + r;
    out_mstride = hw;
  } else {
    const int64_t wsz = (int64_t)p.Co * p.Cg * p.KH * p.KW;
    out_base = (int64_t)split * wsz + (int64_t)g * p.Cog * a.N + n;
    out_mstride = a.N;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int64_t m = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * lk;
    if (m >= a.M) continue;
    const int64_t o = out_base + m * out_mstride;
    float vr = acc_r[r], vi = acc_i[r];
    if (MODE == MODE_FWD && a.bias_r) {
      vr += a.bias_r[g * p.Cog + m];
      if (CPLX) vi += a.bias_i[g * p.Cog + m];
    }
    if (MODE == MODE_WGRAD) {
      reinterpret_cast<float*>(a.yr)[o] = vr;
      if (CPLX) reinterpret_cast<float*>(a.yi)[o] = vi;
    } else {
      io<T>::st(reinterpret_cast<T*>(a.yr) + o, vr);
      if (CPLX) io<T>::st(reinterpret_cast<T*>(a.yi) + o, vi);
    }
  }
}

// out[i] = (sum_s slab[s][i]) (* emul[i])
// out[i] = (sum_s slab[s][i]) (* emul[i]).  Block = 64 consecutive elements x 16 split lanes (one wave
// per split lane, coalesced 256-B reads, 4 loads in flight per thread), fixed summation order.
__global__ __launch_bounds__(1024) void slab_sum_kernel(const float* slabs, int splits, int64_t n,
                                                        const float* emul, float* out) {
  __shared__ float red[16][64];
  const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 64 + lane;
  float a[4] = {0.f, 0.f, 0.f, 0.f};
  if (i < n) {
    int s = sl;
    for (; s + 48 < splits; s += 64) {
#pragma unroll
      for (int u = 0; u < 4; ++u) a[u] += slabs[(int64_t)(s + 16 * u) * n + i];
    }
    for (; s < splits; s += 16) a[0] += slabs[(int64_t)s * n + i];
  }
  red[sl][lane] = (a[0] + a[1]) + (a[2] + a[3]);
  __syncthreads();
  if (sl == 0 && i < n) {
    float acc = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) acc += red[w][lane];
    out[i] = emul ? acc * emul[i] : acc;
  }
}

// per-channel sum over (b, spatial) of an NCHW tensor: stage 1 partials, stage 2 final
template <typename T>
__global__ __launch_bounds__(256) void chansum_partial(const T* x, int64_t B, int C, int64_t S,
                                                       double* partial) {
  __shared__ double red[4];
  const int c = blockIdx.x;
  double acc = 0.0;
  // 8 elements per lane and load (16-B for bf16, 2 x 16-B for f32; planes are only element-aligned,
  // the packed struct makes the compiler emit unaligned-tolerant dwordx4 loads); float partial of
  // the 8, double across iterations
  struct __attribute__((packed, aligned(sizeof(T)))) V8 { T v[8]; };
  for (int64_t b = blockIdx.y; b < B; b += gridDim.y) {
    const T* pl = x + (b * C + c) * S;
    const int64_t S8 = S & ~(int64_t)7;
    for (int64_t s = (int64_t)threadIdx.x * 8; s < S8; s += 256 * 8) {
      const V8 v = *reinterpret_cast<const V8*>(pl + s);
      float p = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) p += io<T>::ld(&v.v[e]);
      acc += (double)p;
    }
    for (int64_t s = S8 + threadIdx.x; s < S; s += 256) acc += (double)io<T>::ld(pl + s);
  }
  const double t = block_sum<double, 256>(acc, red);
  if (threadIdx.x == 0) partial[(int64_t)blockIdx.y * C + c] = t;
}
// one wave per channel: lanes stride over the chunk partials, then a wave reduction
__global__ __launch_bounds__(64) void chansum_final(const double* partial, int chunks, int C, float* out) {
  const int c = blockIdx.x;
  double acc = 0.0;
  for (int j = threadIdx.x; j < chunks; j += 64) acc += partial[(int64_t)j * C + c];
  acc = wave_sum(acc);
  if (threadIdx.x == 0) out[c] = (float)acc;
}

static bool conv_geom_ok(const ConvP& p) {
  return p.B >= 0 && p.Ci > 0 && p.Co > 0 && p.G > 0 && p.Ci % p.G == 0 && p.Co % p.G == 0 &&
         p.KH > 0 && p.KW > 0 && p.sh > 0 && p.sw > 0 && p.dh > 0 && p.dw > 0 && p.Ho > 0 &&
         p.Wo > 0 && p.Cg == p.Ci / p.G && p.Cog == p.Co / p.G;
}

template <typename T, int MODE>
static int conv_launch(ConvArgs& a, bool cplx, hipStream_t st) {
  dim3 grid((unsigned)((a.N + CBN - 1) / CBN), (unsigned)((a.M + CBM - 1) / CBM),
            (unsigned)(a.p.G * a.splits));
  if (grid.y > 65535 || grid.z > 65535) return CPLXAMD_ESHAPE;
  if (cplx) conv_kernel<T, true, MODE><<<grid, 256, 0, st>>>(a);
  else conv_kernel<T, false, MODE><<<grid, 256, 0, st>>>(a);
  CPLXAMD_CHECK_LAUNCH();
  return 0;
}

}  // namespace cplxamd

using namespace cplxamd;

extern "C" {

static int fill_geom(const int* g, ConvP& p) {
  // g = {B, Ci, Co, H, W, KH, KW, sh, sw, ph, pw, dh, dw, groups}
  p.B = g[0]; p.Ci = g[1]; p.Co = g[2]; p.H = g[3]; p.W = g[4]; p.KH = g[5]; p.KW = g[6];
  p.sh = g[7]; p.sw = g[8]; p.ph = g[9]; p.pw = g[10]; p.dh = g[11]; p.dw = g[12]; p.G = g[13];
  if (p.G <= 0 || p.sh <= 0 || p.sw <= 0) return CPLXAMD_EINVAL;
  p.Ho = (p.H + 2 * p.ph - p.dh * (p.KH - 1) - 1) / p.sh + 1;
  p.Wo = (p.W + 2 * p.pw - p.dw * (p.KW - 1) - 1) / p.sw + 1;
  p.Cg = p.Ci / p.G; p.Cog = p.Co / p.G;
  return conv_geom_ok(p) ? 0 : CPLXAMD_ESHAPE;
}

int cplxamd_conv2d_out_shape(const int* geom, int* ho, int* wo) {
  ConvP p;
  const int rc = fill_geom(geom, p);
  if (rc) return rc;
  *ho = p.Ho; *wo = p.Wo;
  return 0;
}

int cplxamd_conv2d_fwd(const void* xr, const void* xi, const void* wr, const void* wi,
                       const float* bias_r, const float* bias_i, void* yr, void* yi,
                       const int* geom, int dtype, void* stream) {
  if (!xr || !wr || !yr || !geom) return CPLXAMD_EINVAL;
  const bool cplx = xi != nullptr;
  if (cplx && (!wi || !yi)) return CPLXAMD_EINVAL;
  ConvArgs a{};
  int rc = fill_geom(geom, a.p);
  if (rc) return rc;
  a.xr = xr; a.xi = xi; a.wr = wr; a.wi = wi; a.bias_r = bias_r; a.bias_i = bias_i;
  a.yr = yr; a.yi = yi;
  if (a.p.B == 0) return 0;                                   // empty batch: nothing to write
  a.M = a.p.Cog; a.N = (int64_t)a.p.B * a.p.Ho * a.p.Wo; a.K = (int64_t)a.p.Cg * a.p.KH * a.p.KW;
  a.splits = 1; a.kchunk = a.K;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CPLXAMD_F32) return conv_launch<float, MODE_FWD>(a, cplx, st);
  if (dtype == CPLXAMD_BF16) return conv_launch<bf16_t, MODE_FWD>(a, cplx, st);
  return CPLXAMD_EINVAL;
}

int cplxamd_conv2d_dgrad(const void* gr, const void* gi, const void* wr, const void* wi,
                         void* dxr, void* dxi, const int* geom, int dtype, void* stream) {
  if (!gr || !wr || !dxr || !geom) return CPLXAMD_EINVAL;
  const bool cplx = gi != nullptr;
  if (cplx && (!wi || !dxi)) return CPLXAMD_EINVAL;
  ConvArgs a{};
  int rc = fill_geom(geom, a.p);
  if (rc) return rc;
  a.xr = gr; a.xi = gi; a.wr = wr; a.wi = wi; a.yr = dxr; a.yi = dxi;
  if (a.p.B == 0) return 0;
  a.M = a.p.Cg; a.N = (int64_t)a.p.B * a.p.H * a.p.W; a.K = (int64_t)a.p.Cog * a.p.KH * a.p.KW;
  a.splits = 1; a.kchunk = a.K;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CPLXAMD_F32) return conv_launch<float, MODE_DGRAD>(a, cplx, st);
  if (dtype == CPLXAMD_BF16) return conv_launch<bf16_t, MODE_DGRAD>(a, cplx, st);
  return CPLXAMD_EINVAL;
}

/* split-K factor the wgrad will use and the scratch it needs */
int cplxamd_conv2d_wgrad_splits(const int* geom) {
  ConvP p;
  if (fill_geom(geom, p)) return 0;
  const int64_t K = (int64_t)p.B * p.Ho * p.Wo;
  const int64_t tiles = (int64_t)((p.Cog + CBM - 1) / CBM) * (((int64_t)p.Cg * p.KH * p.KW + CBN - 1) / CBN) * p.G;
  int64_t s = (768 + tiles - 1) / tiles;            // ~3 workgroups per CU; more only adds slab traffic
  const int64_t maxs = (K + 4 * CBK - 1) / (4 * CBK);
  if (s > maxs) s = maxs;
  if (s < 1) s = 1;
  if (s * p.G > 65535) s = 65535 / p.G;
  return (int)s;
}

int64_t cplxamd_conv2d_wgrad_ws_bytes(const int* geom, int cplx) {
  ConvP p;
  if (fill_geom(geom, p)) return -1;
  const int64_t wsz = (int64_t)p.Co * p.Cg * p.KH * p.KW;
  return (int64_t)cplxamd_conv2d_wgrad_splits(geom) * wsz * sizeof(float) * (cplx ? 2 : 1);
}

int cplxamd_conv2d_wgrad(const void* gr, const void* gi, const void* xr, const void* xi,
                         const float* emul, float* dwr, float* dwi, const int* geom, int dtype,
                         void* ws, int64_t ws_bytes, void* stream) {
  if (!gr || !xr || !dwr || !geom || !ws) return CPLXAMD_EINVAL;
  const bool cplx = gi != nullptr;
  if (cplx && (!xi || !dwi)) return CPLXAMD_EINVAL;
  ConvArgs a{};
  int rc = fill_geom(geom, a.p);
  if (rc) return rc;
  const int64_t wsz = (int64_t)a.p.Co * a.p.Cg * a.p.KH * a.p.KW;
  if (a.p.B == 0) {                                            // empty batch: the gradient is zero
    hipError_t e = hipMemsetAsync(dwr, 0, wsz * sizeof(float), (hipStream_t)stream);
    if (e == hipSuccess && cplx) e = hipMemsetAsync(dwi, 0, wsz * sizeof(float), (hipStream_t)stream);
    return (int)e;
  }
  if (ws_bytes < cplxamd_conv2d_wgrad_ws_bytes(geom, cplx)) return CPLXAMD_EWS;
  a.splits = cplxamd_conv2d_wgrad_splits(geom);
  a.xr = gr; a.xi = gi; a.wr = xr; a.wi = xi;
  a.yr = ws; a.yi = (float*)ws + (int64_t)a.splits * wsz;
  a.M = a.p.Cog; a.N = (int64_t)a.p.Cg * a.p.KH * a.p.KW; a.K = (int64_t)a.p.B * a.p.Ho * a.p.Wo;
  a.kchunk = ((a.K + a.splits - 1) / a.splits + CBK - 1) / CBK * CBK;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CPLXAMD_F32) rc = conv_launch<float, MODE_WGRAD>(a, cplx, st);
  else if (dtype == CPLXAMD_BF16) rc = conv_launch<bf16_t, MODE_WGRAD>(a, cplx, st);
  else return CPLXAMD_EINVAL;
  if (rc) return rc;
  const int sgrid = (int)((wsz + 63) / 64);
  slab_sum_kernel<<<sgrid, 1024, 0, st>>>((const float*)a.yr, a.splits, wsz, emul, dwr);
  CPLXAMD_CHECK_LAUNCH();
  if (cplx) {
    slab_sum_kernel<<<sgrid, 1024, 0, st>>>((const float*)a.yi, a.splits, wsz, nullptr, dwi);
    CPLXAMD_CHECK_LAUNCH();
  }
  return 0;
}

/* out[c] = sum over (b, spatial) of an NCHW tensor (conv bias gradient); ws >= 64*C doubles */
int cplxamd_chansum(const void* x, float* out, int64_t B, int C, int64_t S, int dtype, void* ws,
                    void* stream) {
  if (!x || !out || !ws || B <= 0 || C <= 0 || S <= 0) return CPLXAMD_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int chunks = (int)(B < 64 ? B : 64);
  dim3 grid(C, chunks);
  if (dtype == CPLXAMD_F32)
    chansum_partial<float><<<grid, 256, 0, st>>>((const float*)x, B, C, S, (double*)ws);
  else if (dtype == CPLXAMD_BF16)
    chansum_partial<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)x, B, C, S, (double*)ws);
  else
    return CPLXAMD_EINVAL;
  CPLXAMD_CHECK_LAUNCH();
  chansum_final<<<C, 64, 0, st>>>((const double*)ws, chunks, C, out);
  CPLXAMD_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
