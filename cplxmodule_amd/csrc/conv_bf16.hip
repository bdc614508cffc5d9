// K2 fast path: complex / real 2-d convolution with bf16 activations on the bf16 matrix cores
// (v_mfma_f32_32x32x16_bf16), implicit GEMM, im2col never materialised.
//
// Same three GEMM views as conv.hip (FWD / DGRAD / WGRAD) and the same 4-chain complex core as
// gemm_bf16_impl.h, but the operand tiles are gathered through registers:
//   * index decode is hoisted: the (ci, kh, kw) -> (offset, dh, dw) map is a small table built on
//     the host side (ktab), the pixel -> (b, oh, ow) decode happens once per thread per K tile;
//   * "row-fast" operands (pixels along lanes): each thread fetches the 8 consecutive k of one
//     16-B LDS chunk with 8 coalesced 2-byte loads and writes it with ONE ds_write_b128;
//   * K-contiguous operands (weights; host-repacked for DGRAD) move as 16-B global loads;
//   * LDS rows are 80 B (64 B of k + 16 B pad): every ds_read_b128 lane group hits 16 distinct
//     16-B slots.
// Tile: 64 x 64 outputs per 256-thread block (4 waves, one 32x32 MFMA tile x {re, im} each),
// BK = 32, single LDS stage (20 KiB): ~8 blocks per CU hide the gather latency.  (A register-
// prefetching, double-buffered variant at 4 blocks per CU measured 12 % SLOWER on cfg3: the
// kernel is bound by gather-instruction issue, not by latency.)
//
// Reference semantics: cplx.convnd (cplxmodule/cplx.py:717-800), zero padding, no conjugation
// in the forward; DGRAD / WGRAD conjugate the weight / input.
#include <stdlib.h>

#include "common.h"

namespace cplxamd {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int FBM = 64, FBN = 64, FBK = 32, FROW = 80;   // FROW: LDS row pitch in bytes
constexpr int FPLANE = FBM * FROW;                       // 5120 B per operand plane

struct ConvFP {
  int B, Ci, Co, H, W, KH, KW, Ho, Wo, sh, sw, ph, pw, dh, dw, G, Cg, Cog;
};

enum { FMODE_FWD = 0, FMODE_DGRAD = 1, FMODE_WGRAD = 2, FMODE_WGRAD_ROWS = 3 };

struct ConvFArgs {
  const bf16_t* ar; const bf16_t* ai;   // A operand: FWD weight [Co][K]; DGRAD repacked weight
                                        // [G][Cg][Cog*KH*KW]; WGRAD grad-out (gathered)
  const bf16_t* br; const bf16_t* bi;   // B operand: FWD x; DGRAD grad-out; WGRAD x
  const int* ktab;                      // [3][T] offset, dh, dw per (ci|co, kh, kw) index
  int ktab_n;
  const float* bias_r; const float* bias_i;
  void* yr; void* yi;
  ConvFP p;
  int64_t M, N, K;
  int splits; int64_t kchunk;
};

__device__ __forceinline__ uint32_t pack2(bf16_t lo, bf16_t hi) {
  return (uint32_t)lo | ((uint32_t)hi << 16);
}

// ---- K-contiguous operand tile [64 rows][32 k]: one 16-B chunk per thread per plane --------
template <bool CPLX>
__device__ __forceinline__ void stage_kvec(char* lds_r, char* lds_i, const bf16_t* sr,
                                           const bf16_t* si, int64_t row0, int64_t rows,
                                           int64_t ld, int64_t k0) {
  const int t = threadIdx.x;
  const int r = t >> 2, c = t & 3;
  uint4 vr = make_uint4(0, 0, 0, 0), vi = vr;
  if (row0 + r < rows) {
    const int64_t off = (row0 + r) * ld + k0 + c * 8;
    vr = *reinterpret_cast<const uint4*>(sr + off);
    if (CPLX) vi = *reinterpret_cast<const uint4*>(si + off);
  }
  *reinterpret_cast<uint4*>(lds_r + r * FROW + c * 16) = vr;
  if (CPLX) *reinterpret_cast<uint4*>(lds_i + r * FROW + c * 16) = vi;
}

// ---- row-fast gathered tile: rows = pixels along lanes, k uniform per wave ------------------
// base / h0 / w0 describe this thread's row (pixel); element k adds ktab offsets.
template <bool CPLX, bool CHECK>
__device__ __forceinline__ void stage_rowfast(char* lds_r, char* lds_i, const bf16_t* sr,
                                              const bf16_t* si, bool row_ok, int64_t base, int h0,
                                              int w0, int Hl, int Wl, const int* ktab, int T,
                                              int64_t k0) {
  const int t = threadIdx.x;
  const int r = t & 63;
  const int kq = __builtin_amdgcn_readfirstlane(t >> 6);   // which 8-k chunk (wave-uniform)
  bf16_t vr[8], vi[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = (int)k0 + kq * 8 + j;
    const int off = ktab[k], dhk = ktab[T + k], dwk = ktab[2 * T + k];
    bool ok = row_ok;
    if (CHECK) ok = ok && (unsigned)(h0 + dhk) < (unsigned)Hl && (unsigned)(w0 + dwk) < (unsigned)Wl;
    vr[j] = 0; vi[j] = 0;
    if (ok) {
      vr[j] = sr[base + off];
      if (CPLX) vi[j] = si[base + off];
    }
  }
  const uint4 pr = make_uint4(pack2(vr[0], vr[1]), pack2(vr[2], vr[3]), pack2(vr[4], vr[5]), pack2(vr[6], vr[7]));
  *reinterpret_cast<uint4*>(lds_r + r * FROW + kq * 16) = pr;
  if (CPLX) {
    const uint4 pi = make_uint4(pack2(vi[0], vi[1]), pack2(vi[2], vi[3]), pack2(vi[4], vi[5]), pack2(vi[6], vi[7]));
    *reinterpret_cast<uint4*>(lds_i + r * FROW + kq * 16) = pi;
  }
}

// 8 consecutive bf16 from a 2-byte-aligned address (one global_load_dwordx4 on gfx950), the
// first `valid` of them kept, the rest zero.
struct __attribute__((packed, aligned(2))) U16x8 { uint16_t v[8]; };
__device__ __forceinline__ uint4 load8_masked(const bf16_t* p, int valid) {
  uint4 r = make_uint4(0, 0, 0, 0);
  if (valid >= 8) {
    const U16x8 u = *reinterpret_cast<const U16x8*>(p);
    __builtin_memcpy(&r, &u, 16);
  } else if (valid > 0) {
    uint16_t e[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) e[j] = j < valid ? p[j] : (uint16_t)0;
    r = make_uint4(pack2(e[0], e[1]), pack2(e[2], e[3]), pack2(e[4], e[5]), pack2(e[6], e[7]));
  }
  return r;
}

__device__ __forceinline__ bf16x8 frag(const char* plane, int row, int kc) {
  return *reinterpret_cast<const bf16x8*>(plane + row * FROW + kc * 16);
}
__device__ __forceinline__ bf16x8 negf(bf16x8 v) {
  uint4 u = __builtin_bit_cast(uint4, v);
  u.x ^= 0x80008000u; u.y ^= 0x80008000u; u.z ^= 0x80008000u; u.w ^= 0x80008000u;
  return __builtin_bit_cast(bf16x8, u);
}

template <bool CPLX, int MODE, bool CHECK>
__global__ __launch_bounds__(256) void conv_bf16_kernel(ConvFArgs a) {
  __shared__ __attribute__((aligned(16))) char lds[(CPLX ? 4 : 2) * FPLANE];
  char* sAr = lds; char* sBr = lds + FPLANE;
  char* sAi = lds + 2 * FPLANE; char* sBi = lds + 3 * FPLANE;
  const ConvFP& p = a.p;
  const int g = blockIdx.z / a.splits, split = blockIdx.z % a.splits;
  const int64_t m0 = (int64_t)blockIdx.y * FBM, n0 = (int64_t)blockIdx.x * FBN;
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int wm = (wid >> 1) * 32, wn = (wid & 1) * 32, l31 = lane & 31, lk = lane >> 5;
  const int64_t kbeg = (int64_t)split * a.kchunk;
  int64_t kend = kbeg + a.kchunk;
  if (kend > a.K) kend = a.K;
  const int khw = p.KH * p.KW;
  const int64_t HW = (int64_t)p.H * p.W, HoWo = (int64_t)p.Ho * p.Wo;

  // ---- per-thread row decode, hoisted out of the K loop ----
  // FWD / DGRAD: this thread's B row (pixel) for the row-fast gather
  bool b_ok = false; int64_t b_base = 0; int bh0 = 0, bw0 = 0;
  if (MODE == FMODE_FWD || MODE == FMODE_DGRAD) {
    const int64_t n = n0 + (t & 63);
    b_ok = n < a.N;
    if (b_ok) {
      if (MODE == FMODE_FWD) {
        const int64_t b = n / HoWo; const int r = (int)(n - b * HoWo);
        const int oh = r / p.Wo, ow = r - oh * p.Wo;
        bh0 = oh * p.sh - p.ph; bw0 = ow * p.sw - p.pw;
        b_base = (b * p.Ci + (int64_t)g * p.Cg) * HW + (int64_t)bh0 * p.W + bw0;
      } else {  // DGRAD (stride 1): pixel of dx
        const int64_t b = n / HW; const int r = (int)(n - b * HW);
        const int ih = r / p.W, iw = r - ih * p.W;
        bh0 = ih + p.ph; bw0 = iw + p.pw;
        b_base = (b * p.Co + (int64_t)g * p.Cog) * HoWo + (int64_t)bh0 * p.Wo + bw0;
      }
    }
  }
  const int Hl = (MODE == FMODE_DGRAD) ? p.Ho : p.H, Wl = (MODE == FMODE_DGRAD) ? p.Wo : p.W;

  // WGRAD_ROWS (stride 1, no padding): a K tile is 32 consecutive output pixels of ONE output
  // row (b, oh); both operand rows are then contiguous in memory and move as 16-B loads.
  // K counts such tiles; this thread owns LDS chunk (row r, 8-pixel group q) of both operands.
  int wr_off = 0;
  const int wr_r = t >> 2, wr_q = t & 3;
  const int chunks_w = (p.Wo + FBK - 1) / FBK;
  if (MODE == FMODE_WGRAD_ROWS) {
    const int64_t n = n0 + wr_r;
    wr_off = n < a.N ? a.ktab[n] : -1;   // ci*H*W + kh*dh*W + kw*dw, hoisted for the whole loop
  }

  f32x16 acc_r = {0}, acc_i = {0};
  for (int64_t k0 = kbeg; k0 < kend; k0 += (MODE == FMODE_WGRAD_ROWS ? 1 : FBK)) {
    if (MODE == FMODE_WGRAD_ROWS) {
      const int64_t bo = k0 / chunks_w;                 // (b, oh) index
      const int c = (int)(k0 - bo * chunks_w);
      const int64_t b = bo / p.Ho; const int oh = (int)(bo - b * p.Ho);
      const int ow = c * FBK + wr_q * 8;
      const int valid = p.Wo - ow;
      uint4 gr4 = make_uint4(0, 0, 0, 0), gi4 = gr4, xr4 = gr4, xi4 = gr4;
      if (m0 + wr_r < a.M) {
        const int64_t off = ((b * p.Co + (int64_t)g * p.Cog + m0 + wr_r) * p.Ho + oh) * p.Wo + ow;
        gr4 = load8_masked(a.ar + off, valid);
        if (CPLX) gi4 = load8_masked(a.ai + off, valid);
      }
      if (wr_off >= 0) {
        const int64_t off = (b * p.Ci + (int64_t)g * p.Cg) * HW + wr_off + (int64_t)oh * p.W + ow;
        xr4 = load8_masked(a.br + off, valid);
        if (CPLX) xi4 = load8_masked(a.bi + off, valid);
      }
      *reinterpret_cast<uint4*>(sAr + wr_r * FROW + wr_q * 16) = gr4;
      *reinterpret_cast<uint4*>(sBr + wr_r * FROW + wr_q * 16) = xr4;
      if (CPLX) {
        *reinterpret_cast<uint4*>(sAi + wr_r * FROW + wr_q * 16) = gi4;
        *reinterpret_cast<uint4*>(sBi + wr_r * FROW + wr_q * 16) = xi4;
      }
    } else if (MODE == FMODE_FWD) {
      const int64_t Kw = (int64_t)p.Cg * khw;
      stage_kvec<CPLX>(sAr, sAi, a.ar + (int64_t)g * p.Cog * Kw, a.ai + (int64_t)g * p.Cog * Kw, m0,
                       a.M, Kw, k0);
      stage_rowfast<CPLX, CHECK>(sBr, sBi, a.br, a.bi, b_ok, b_base, bh0, bw0, Hl, Wl, a.ktab,
                                 a.ktab_n, k0);
    } else if (MODE == FMODE_DGRAD) {
      const int64_t Kw = (int64_t)p.Cog * khw;
      stage_kvec<CPLX>(sAr, sAi, a.ar + (int64_t)g * p.Cg * Kw, a.ai + (int64_t)g * p.Cg * Kw, m0,
                       a.M, Kw, k0);
      stage_rowfast<CPLX, true>(sBr, sBi, a.br, a.bi, b_ok, b_base, bh0, bw0, Hl, Wl, a.ktab,
                                a.ktab_n, k0);
    } else {
      // WGRAD: k = (b, oh, ow) runs along lanes for both operands
      const int kk = t & 31, rq = t >> 5;
      const int64_t k = k0 + kk;
      const bool k_ok = k < kend;
      int64_t gbase = 0, xbase = 0; int h0 = 0, w0 = 0;
      if (k_ok) {
        const int64_t b = k / HoWo; const int r = (int)(k - b * HoWo);
        const int oh = r / p.Wo, ow = r - oh * p.Wo;
        gbase = (b * p.Co + (int64_t)g * p.Cog) * HoWo + r;
        h0 = oh * p.sh - p.ph; w0 = ow * p.sw - p.pw;
        xbase = (b * p.Ci + (int64_t)g * p.Cg) * HW + (int64_t)h0 * p.W + w0;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int r = rq + 8 * j;
        bf16_t vr = 0, vi = 0;
        if (k_ok && m0 + r < a.M) {
          const int64_t off = gbase + (m0 + r) * HoWo;
          vr = a.ar[off];
          if (CPLX) vi = a.ai[off];
        }
        *reinterpret_cast<bf16_t*>(sAr + r * FROW + kk * 2) = vr;
        if (CPLX) *reinterpret_cast<bf16_t*>(sAi + r * FROW + kk * 2) = vi;
        bf16_t wr = 0, wi = 0;
        const int64_t n = n0 + r;
        if (k_ok && n < a.N) {
          const int off = a.ktab[n], dhk = a.ktab[a.ktab_n + n], dwk = a.ktab[2 * a.ktab_n + n];
          bool ok = true;
          if (CHECK) ok = (unsigned)(h0 + dhk) < (unsigned)p.H && (unsigned)(w0 + dwk) < (unsigned)p.W;
          if (ok) {
            wr = a.br[xbase + off];
            if (CPLX) wi = a.bi[xbase + off];
          }
        }
        *reinterpret_cast<bf16_t*>(sBr + r * FROW + kk * 2) = wr;
        if (CPLX) *reinterpret_cast<bf16_t*>(sBi + r * FROW + kk * 2) = wi;
      }
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int kc = ks * 2 + lk;
      const bf16x8 ar = frag(sAr, wm + l31, kc), br = frag(sBr, wn + l31, kc);
      acc_r = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar, br, acc_r, 0, 0, 0);
      if (CPLX) {
        const bf16x8 ai = frag(sAi, wm + l31, kc), bi = frag(sBi, wn + l31, kc);
        // FWD: (Ar + iAi)(Br + iBi); DGRAD: conj(A) B; WGRAD: A conj(B)
        if (MODE == FMODE_FWD) {
          acc_r = __builtin_amdgcn_mfma_f32_32x32x16_bf16(negf(ai), bi, acc_r, 0, 0, 0);
          acc_i = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar, bi, acc_i, 0, 0, 0);
          acc_i = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ai, br, acc_i, 0, 0, 0);
        } else if (MODE == FMODE_DGRAD) {
          acc_r = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ai, bi, acc_r, 0, 0, 0);
          acc_i = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar, bi, acc_i, 0, 0, 0);
          acc_i = __builtin_amdgcn_mfma_f32_32x32x16_bf16(negf(ai), br, acc_i, 0, 0, 0);
        } else {   // both WGRAD flavours: A conj(B)
          acc_r = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ai, bi, acc_r, 0, 0, 0);
          acc_i = __builtin_amdgcn_mfma_f32_32x32x16_bf16(negf(ar), bi, acc_i, 0, 0, 0);
          acc_i = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ai, br, acc_i, 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }

  // epilogue: col = lane & 31 runs along N, rows (M) across registers
  const int64_t n = n0 + wn + l31;
  if (n >= a.N) return;
  int64_t out_base, out_mstride;
  if (MODE == FMODE_FWD) {
    const int64_t b = n / HoWo, r = n - b * HoWo;
    out_base = (b * p.Co + (int64_t)g * p.Cog) * HoWo + r; out_mstride = HoWo;
  } else if (MODE == FMODE_DGRAD) {
    const int64_t b = n / HW, r = n - b * HW;
    out_base = (b * p.Ci + (int64_t)g * p.Cg) * HW + r; out_mstride = HW;
  } else {
    const int64_t wsz = (int64_t)p.Co * p.Cg * khw;
    out_base = (int64_t)split * wsz + (int64_t)g * p.Cog * a.N + n; out_mstride = a.N;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int64_t m = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * lk;
    if (m >= a.M) continue;
    const int64_t o = out_base + m * out_mstride;
    float vr = acc_r[r], vi = acc_i[r];
    if (MODE == FMODE_FWD && a.bias_r) {
      vr += a.bias_r[g * p.Cog + m];
      if (CPLX) vi += a.bias_i[g * p.Cog + m];
    }
    if (MODE == FMODE_WGRAD || MODE == FMODE_WGRAD_ROWS) {
      reinterpret_cast<float*>(a.yr)[o] = vr;
      if (CPLX) reinterpret_cast<float*>(a.yi)[o] = vi;
    } else {
      io<bf16_t>::st(reinterpret_cast<bf16_t*>(a.yr) + o, vr);
      if (CPLX) io<bf16_t>::st(reinterpret_cast<bf16_t*>(a.yi) + o, vi);
    }
  }
}

template <int MODE>
int conv_bf16_launch(const ConvFArgs& a, bool cplx, bool check, hipStream_t st) {
  dim3 grid((unsigned)((a.N + FBN - 1) / FBN), (unsigned)((a.M + FBM - 1) / FBM),
            (unsigned)(a.p.G * a.splits));
  if (grid.y > 65535 || grid.z > 65535) return CPLXAMD_ESHAPE;
  if (cplx) {
    if (check) conv_bf16_kernel<true, MODE, true><<<grid, 256, 0, st>>>(a);
    else conv_bf16_kernel<true, MODE, false><<<grid, 256, 0, st>>>(a);
  } else {
    if (check) conv_bf16_kernel<false, MODE, true><<<grid, 256, 0, st>>>(a);
    else conv_bf16_kernel<false, MODE, false><<<grid, 256, 0, st>>>(a);
  }
  CPLXAMD_CHECK_LAUNCH();
  return 0;
}

// out[i] = (sum_s slab[s][i]) (* emul[i]).  Block = 64 consecutive elements x 16 split lanes (one wave
// per split lane, coalesced 256-B reads, 4 loads in flight per thread), fixed summation order.
__global__ __launch_bounds__(1024) void slab_sum2_kernel(const float* slabs, int splits, int64_t n,
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

static int fill_fp(const int* g, ConvFP& p) {
  p.B = g[0]; p.Ci = g[1]; p.Co = g[2]; p.H = g[3]; p.W = g[4]; p.KH = g[5]; p.KW = g[6];
  p.sh = g[7]; p.sw = g[8]; p.ph = g[9]; p.pw = g[10]; p.dh = g[11]; p.dw = g[12]; p.G = g[13];
  if (p.G <= 0 || p.sh <= 0 || p.sw <= 0 || p.dh <= 0 || p.dw <= 0 || p.B <= 0) return CPLXAMD_EINVAL;
  if (p.Ci % p.G || p.Co % p.G) return CPLXAMD_ESHAPE;
  p.Ho = (p.H + 2 * p.ph - p.dh * (p.KH - 1) - 1) / p.sh + 1;
  p.Wo = (p.W + 2 * p.pw - p.dw * (p.KW - 1) - 1) / p.sw + 1;
  p.Cg = p.Ci / p.G; p.Cog = p.Co / p.G;
  return (p.Ho > 0 && p.Wo > 0) ? 0 : CPLXAMD_ESHAPE;
}

static int wgrad_splits(const ConvFP& p) {
  const int64_t K = (int64_t)p.B * p.Ho * p.Wo;
  const int64_t tiles = (int64_t)((p.Cog + FBM - 1) / FBM) *
                        (((int64_t)p.Cg * p.KH * p.KW + FBN - 1) / FBN) * p.G;
  int64_t s = (768 + tiles - 1) / tiles;            // ~3 workgroups per CU; more only adds slab traffic
  const int64_t maxs = (K + 8 * FBK - 1) / (8 * FBK);
  if (s > maxs) s = maxs;
  if (s < 1) s = 1;
  if (s * p.G > 65535) s = 65535 / p.G;
  return (int)s;
}

}  // namespace cplxamd

using namespace cplxamd;

extern "C" {

/* ---- host-side helpers (plain CPU code) -------------------------------------------------- */
int cplxamd_conv2d_ktab_size(const int* geom, int mode) {
  ConvFP p;
  if (!geom || fill_fp(geom, p)) return -1;
  const int T = (mode == FMODE_DGRAD ? p.Cog : p.Cg) * p.KH * p.KW;
  return 3 * T;
}

/* out[0..T) = element offset, out[T..2T) = dh, out[2T..3T) = dw of (channel, kh, kw) index t */
int cplxamd_conv2d_ktab_fill(const int* geom, int mode, int* out) {
  ConvFP p;
  if (!geom || !out) return CPLXAMD_EINVAL;
  const int rc = fill_fp(geom, p);
  if (rc) return rc;
  const int khw = p.KH * p.KW;
  const int T = (mode == FMODE_DGRAD ? p.Cog : p.Cg) * khw;
  for (int t = 0; t < T; ++t) {
    const int c = t / khw, r = t % khw, kh = r / p.KW, kw = r % p.KW;
    if (mode == FMODE_DGRAD) {
      out[t] = c * p.Ho * p.Wo - kh * p.dh * p.Wo - kw * p.dw;
      out[T + t] = -kh * p.dh; out[2 * T + t] = -kw * p.dw;
    } else {
      out[t] = c * p.H * p.W + kh * p.dh * p.W + kw * p.dw;
      out[T + t] = kh * p.dh; out[2 * T + t] = kw * p.dw;
    }
  }
  return 0;
}

static bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

int cplxamd_conv2d_bf16_fwd(const void* xr, const void* xi, const void* wr, const void* wi,
                            const float* bias_r, const float* bias_i, void* yr, void* yi,
                            const int* geom, const int* ktab, void* stream) {
  if (!xr || !wr || !yr || !geom || !ktab) return CPLXAMD_EINVAL;
  const bool cplx = xi != nullptr;
  if (cplx && (!wi || !yi)) return CPLXAMD_EINVAL;
  ConvFArgs a{};
  int rc = fill_fp(geom, a.p);
  if (rc) return rc;
  const ConvFP& p = a.p;
  a.K = (int64_t)p.Cg * p.KH * p.KW;
  if (a.K % FBK || !al16(wr) || (cplx && !al16(wi))) return CPLXAMD_ESHAPE;
  a.ar = (const bf16_t*)wr; a.ai = (const bf16_t*)wi; a.br = (const bf16_t*)xr; a.bi = (const bf16_t*)xi;
  a.ktab = ktab; a.ktab_n = (int)a.K; a.bias_r = bias_r; a.bias_i = bias_i; a.yr = yr; a.yi = yi;
  a.M = p.Cog; a.N = (int64_t)p.B * p.Ho * p.Wo; a.splits = 1; a.kchunk = a.K;
  const bool check = p.ph > 0 || p.pw > 0;
  return conv_bf16_launch<FMODE_FWD>(a, cplx, check, (hipStream_t)stream);
}

/* wtr / wti: weight repacked to [groups][Ci/g][Co/g * KH * KW] (K-contiguous for the dgrad GEMM) */
int cplxamd_conv2d_bf16_dgrad(const void* gr, const void* gi, const void* wtr, const void* wti,
                              void* dxr, void* dxi, const int* geom, const int* ktab,
                              void* stream) {
  if (!gr || !wtr || !dxr || !geom || !ktab) return CPLXAMD_EINVAL;
  const bool cplx = gi != nullptr;
  if (cplx && (!wti || !dxi)) return CPLXAMD_EINVAL;
  ConvFArgs a{};
  int rc = fill_fp(geom, a.p);
  if (rc) return rc;
  const ConvFP& p = a.p;
  if (p.sh != 1 || p.sw != 1) return CPLXAMD_ESHAPE;       // strided dgrad: generic kernel
  a.K = (int64_t)p.Cog * p.KH * p.KW;
  if (a.K % FBK || !al16(wtr) || (cplx && !al16(wti))) return CPLXAMD_ESHAPE;
  a.ar = (const bf16_t*)wtr; a.ai = (const bf16_t*)wti; a.br = (const bf16_t*)gr; a.bi = (const bf16_t*)gi;
  a.ktab = ktab; a.ktab_n = (int)a.K; a.yr = dxr; a.yi = dxi;
  a.M = p.Cg; a.N = (int64_t)p.B * p.H * p.W; a.splits = 1; a.kchunk = a.K;
  return conv_bf16_launch<FMODE_DGRAD>(a, cplx, true, (hipStream_t)stream);
}

int64_t cplxamd_conv2d_bf16_wgrad_ws_bytes(const int* geom, int cplx) {
  ConvFP p;
  if (!geom || fill_fp(geom, p)) return -1;
  const int64_t wsz = (int64_t)p.Co * p.Cg * p.KH * p.KW;
  return (int64_t)wgrad_splits(p) * wsz * sizeof(float) * (cplx ? 2 : 1);
}

int cplxamd_conv2d_bf16_wgrad(const void* gr, const void* gi, const void* xr, const void* xi,
                              const float* emul, float* dwr, float* dwi, const int* geom,
                              const int* ktab, void* ws, int64_t ws_bytes, void* stream) {
  if (!gr || !xr || !dwr || !geom || !ktab || !ws) return CPLXAMD_EINVAL;
  const bool cplx = gi != nullptr;
  if (cplx && (!xi || !dwi)) return CPLXAMD_EINVAL;
  ConvFArgs a{};
  int rc = fill_fp(geom, a.p);
  if (rc) return rc;
  const ConvFP& p = a.p;
  if (ws_bytes < cplxamd_conv2d_bf16_wgrad_ws_bytes(geom, cplx)) return CPLXAMD_EWS;
  const int64_t wsz = (int64_t)p.Co * p.Cg * p.KH * p.KW;
  a.splits = wgrad_splits(p);
  a.ar = (const bf16_t*)gr; a.ai = (const bf16_t*)gi; a.br = (const bf16_t*)xr; a.bi = (const bf16_t*)xi;
  a.ktab = ktab; a.ktab_n = p.Cg * p.KH * p.KW;
  a.yr = ws; a.yi = (float*)ws + (int64_t)a.splits * wsz;
  a.M = p.Cog; a.N = (int64_t)p.Cg * p.KH * p.KW; a.K = (int64_t)p.B * p.Ho * p.Wo;
  hipStream_t st = (hipStream_t)stream;
  const bool check = p.ph > 0 || p.pw > 0;
  if (p.sh == 1 && p.sw == 1 && !check && p.Wo >= 16) {
    // contiguous-row formulation: K = number of (b, oh, 32-pixel chunk) tiles
    a.K = (int64_t)p.B * p.Ho * ((p.Wo + FBK - 1) / FBK);
    a.kchunk = (a.K + a.splits - 1) / a.splits;
    rc = conv_bf16_launch<FMODE_WGRAD_ROWS>(a, cplx, false, st);
  } else {
    a.kchunk = ((a.K + a.splits - 1) / a.splits + FBK - 1) / FBK * FBK;
    rc = conv_bf16_launch<FMODE_WGRAD>(a, cplx, check, st);
  }
  if (rc) return rc;
  const int sgrid = (int)((wsz + 63) / 64);
  slab_sum2_kernel<<<sgrid, 1024, 0, st>>>((const float*)a.yr, a.splits, wsz, emul, dwr);
  CPLXAMD_CHECK_LAUNCH();
  if (cplx) {
    slab_sum2_kernel<<<sgrid, 1024, 0, st>>>((const float*)a.yi, a.splits, wsz, nullptr, dwi);
    CPLXAMD_CHECK_LAUNCH();
  }
  return 0;
}

}  // extern "C"
