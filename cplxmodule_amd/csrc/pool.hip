// Complex abs-max pooling (reference: cplx.max_poolnd, cplxmodule/cplx.py:1114-1175
// = F.max_pool(abs(z), return_indices) + two gathers): the element of largest modulus in each window
// keeps BOTH its parts.  One pass: |z| is computed in registers (sqrt(fma(zi, zi, zr zr)), the
// CPU kernel's form), the argmax is torch's (first maximum in row-major window order, NaN wins),
// the flat index is kept for the backward.  Backward is a deterministic gather: every input
// position sums the gradients of the output windows that selected it (no atomics).
#include "common.h"

#pragma clang fp contract(off)

namespace cplxamd {

struct PoolP {
  int B, C, H, W, Ho, Wo, kh, kw, sh, sw, ph, pw, dh, dw;
};

// CL: channels-last storage ([B][H][W][C], consecutive threads = consecutive channels of one output pixel); the kept
// index is the pixel h * W + w either way.
template <typename T, bool CL>
__global__ __launch_bounds__(256) void cplx_maxpool_fwd_kernel(const T* zr, const T* zi, T* yr, T* yi,
                                                               int32_t* idx, PoolP p, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < n; o += stride) {
    const int64_t op = CL ? o / p.C : o;                       // output pixel index (x channel for planar)
    const int ow = (int)(op % p.Wo), oh = (int)((op / p.Wo) % p.Ho);
    const int64_t bc = op / ((int64_t)p.Wo * p.Ho);            // CL: image b; planar: b * C + c
    const int64_t ps = CL ? p.C : 1;                           // pixel stride
    const T* pr = zr + (CL ? bc * p.H * p.W * p.C + (o - op * p.C) : bc * p.H * p.W);
    const T* pi = zi + (CL ? bc * p.H * p.W * p.C + (o - op * p.C) : bc * p.H * p.W);
    float best = -INFINITY, br = 0.f, bi = 0.f;
    int bidx = -1;
    for (int i = 0; i < p.kh; ++i) {
      const int h = oh * p.sh - p.ph + i * p.dh;
      if (h < 0 || h >= p.H) continue;
      for (int j = 0; j < p.kw; ++j) {
        const int w = ow * p.sw - p.pw + j * p.dw;
        if (w < 0 || w >= p.W) continue;
        const float a = io<T>::ld(pr + (int64_t)(h * p.W + w) * ps), b = io<T>::ld(pi + (int64_t)(h * p.W + w) * ps);
        const float m = rn_sqrt(fmaf(b, b, a * a));
        if (bidx < 0 || m > best || m != m) {
          best = m; br = a; bi = b; bidx = h * p.W + w;
        }
      }
    }
    io<T>::st(yr + o, br);
    io<T>::st(yi + o, bi);
    idx[o] = bidx;
  }
}

__device__ __forceinline__ int cdiv_floor(int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); }

template <typename T, bool CL>
__global__ __launch_bounds__(256) void cplx_maxpool_bwd_kernel(const T* gr, const T* gi,
                                                               const int32_t* idx, T* dzr, T* dzi,
                                                               PoolP p, int64_t n_in) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n_in; e += stride) {
    const int64_t ep = CL ? e / p.C : e;
    const int w = (int)(ep % p.W), h = (int)((ep / p.W) % p.H);
    const int64_t bc = ep / ((int64_t)p.W * p.H);
    const int64_t ps = CL ? p.C : 1, ch = CL ? e - ep * p.C : 0;
    const int me = h * p.W + w;
    // output rows whose window can contain h: oh*sh - ph <= h <= oh*sh - ph + (kh-1)*dh
    int oh0 = cdiv_floor(h + p.ph - (p.kh - 1) * p.dh + p.sh - 1, p.sh), oh1 = cdiv_floor(h + p.ph, p.sh);
    int ow0 = cdiv_floor(w + p.pw - (p.kw - 1) * p.dw + p.sw - 1, p.sw), ow1 = cdiv_floor(w + p.pw, p.sw);
    oh0 = oh0 < 0 ? 0 : oh0; ow0 = ow0 < 0 ? 0 : ow0;
    oh1 = oh1 >= p.Ho ? p.Ho - 1 : oh1; ow1 = ow1 >= p.Wo ? p.Wo - 1 : ow1;
    float ar = 0.f, ai = 0.f;
    const int64_t ob = bc * p.Ho * p.Wo;
    for (int oh = oh0; oh <= oh1; ++oh)
      for (int ow = ow0; ow <= ow1; ++ow) {
        const int64_t o = (ob + (int64_t)oh * p.Wo + ow) * ps + ch;
        if (idx[o] == me) { ar += io<T>::ld(gr + o); ai += io<T>::ld(gi + o); }
      }
    io<T>::st(dzr + e, ar);
    io<T>::st(dzi + e, ai);
  }
}

}  // namespace cplxamd

using namespace cplxamd;

extern "C" {

// pool = int[14]: B, C, H, W, Ho, Wo, kh, kw, sh, sw, ph, pw, dh, dw  (Ho / Wo as torch computes them)
static int maxpool_fwd(const void* zr, const void* zi, void* yr, void* yi, int32_t* idx, const int* pool, int dtype,
                       bool cl, void* stream) {
  if (!zr || !zi || !yr || !yi || !idx || !pool) return CPLXAMD_EINVAL;
  PoolP p{pool[0], pool[1], pool[2], pool[3], pool[4], pool[5], pool[6], pool[7], pool[8], pool[9],
          pool[10], pool[11], pool[12], pool[13]};
  if (p.B < 0 || p.C <= 0 || p.H <= 0 || p.W <= 0 || p.Ho <= 0 || p.Wo <= 0 || p.kh <= 0 || p.kw <= 0 ||
      p.sh <= 0 || p.sw <= 0 || p.ph < 0 || p.pw < 0 || p.dh <= 0 || p.dw <= 0)
    return CPLXAMD_EINVAL;
  if ((int64_t)p.H * p.W >= ((int64_t)1 << 31)) return CPLXAMD_ESHAPE;
  const int64_t n = (int64_t)p.B * p.C * p.Ho * p.Wo;
  if (n == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int grid = stream_grid(n, 256);
#define MP(T, L) cplx_maxpool_fwd_kernel<T, L><<<grid, 256, 0, st>>>((const T*)zr, (const T*)zi, (T*)yr, (T*)yi, idx, p, n)
  if (dtype == CPLXAMD_F32) { if (cl) MP(float, true); else MP(float, false); }
  else if (dtype == CPLXAMD_BF16) { if (cl) MP(bf16_t, true); else MP(bf16_t, false); }
  else return CPLXAMD_EINVAL;
#undef MP
  CPLXAMD_CHECK_LAUNCH();
  return 0;
}

int cplxamd_cplx_maxpool2d_fwd(const void* zr, const void* zi, void* yr, void* yi, int32_t* idx,
                               const int* pool, int dtype, void* stream) {
  return maxpool_fwd(zr, zi, yr, yi, idx, pool, dtype, false, stream);
}
/* the same on channels-last planes ([B][H][W][C] in, [B][Ho][Wo][C] out, idx too) */
int cplxamd_cplx_maxpool2d_fwd_cl(const void* zr, const void* zi, void* yr, void* yi, int32_t* idx,
                                  const int* pool, int dtype, void* stream) {
  return maxpool_fwd(zr, zi, yr, yi, idx, pool, dtype, true, stream);
}

static int maxpool_bwd(const void* gr, const void* gi, const int32_t* idx, void* dzr, void* dzi, const int* pool, int dtype,
                       bool cl, void* stream) {
  if (!gr || !gi || !dzr || !dzi || !idx || !pool) return CPLXAMD_EINVAL;
  PoolP p{pool[0], pool[1], pool[2], pool[3], pool[4], pool[5], pool[6], pool[7], pool[8], pool[9],
          pool[10], pool[11], pool[12], pool[13]};
  if (p.sh <= 0 || p.sw <= 0 || p.dh <= 0 || p.dw <= 0 || p.kh <= 0 || p.kw <= 0) return CPLXAMD_EINVAL;
  const int64_t n = (int64_t)p.B * p.C * p.H * p.W;
  if (n <= 0) return n == 0 ? 0 : CPLXAMD_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int grid = stream_grid(n, 256);
#define MP(T, L) cplx_maxpool_bwd_kernel<T, L><<<grid, 256, 0, st>>>((const T*)gr, (const T*)gi, idx, (T*)dzr, (T*)dzi, p, n)
  if (dtype == CPLXAMD_F32) { if (cl) MP(float, true); else MP(float, false); }
  else if (dtype == CPLXAMD_BF16) { if (cl) MP(bf16_t, true); else MP(bf16_t, false); }
  else return CPLXAMD_EINVAL;
#undef MP
  CPLXAMD_CHECK_LAUNCH();
  return 0;
}

int cplxamd_cplx_maxpool2d_bwd(const void* gr, const void* gi, const int32_t* idx, void* dzr, void* dzi,
                               const int* pool, int dtype, void* stream) {
  return maxpool_bwd(gr, gi, idx, dzr, dzi, pool, dtype, false, stream);
}
int cplxamd_cplx_maxpool2d_bwd_cl(const void* gr, const void* gi, const int32_t* idx, void* dzr, void* dzi,
                                  const int* pool, int dtype, void* stream) {
  return maxpool_bwd(gr, gi, idx, dzr, dzi, pool, dtype, true, stream);
}

}  // extern "C"
