// K2: complex / real 2-d convolution (cross-correlation, NCHW, groups, stride, padding,
// dilation) forward, dgrad and wgrad as implicit GEMMs on the exact-f32 matrix cores:
// operand tiles are gathered on the fly (im2col is never materialised) into LDS and
// contracted with v_mfma_f32_32x32x2_f32 (v_mfma_f32_4x4x1_16b_f32 for layers of <= 16 output rows); complex = 4 MFMA
// chains sharing one K-loop.
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
// This is the parity path (exact float32) and the kernel of every geometry the channels-last bf16 kernels (conv_cl*.hip,
// conv_nhwc*.hip) do not take -- strides, groups, odd channel counts, small layers.  Round 3 made it fast on small,
// launch-bound models (BASELINE configs[4]): multiply-shift index math, narrow tiles, 4 x 4 x 1 MFMAs, stride phases for
// the data gradient, the bias gradient as a column of the weight-gradient GEMM.
#include "common.h"

namespace cplxamd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int CBM = 64, CBN = 64, CBK = 16, KH16 = CBK / 16;   // (K step 32 measured no faster on the narrow layers of cfg5: 1.32 vs 1.29 ms per step)

// Division by a launch-constant divisor of indices below 2^31: q = (x * mul) >> (31 + s), s = ceil(log2 d),
// mul = floor(2^(31 + s) / d) + 1 (Granlund-Montgomery round-up multiplier, exact for x < 2^31); d = 1 passes x through.
// The gathers below split pixel / tap indices with these instead of 64-bit divisions (which were 3/4 of the
// kernel's instructions: every operand element paid two of them).
struct FastDiv { uint32_t mul, sh, d; };

static FastDiv make_fastdiv(int64_t d64) {
  FastDiv f{0u, 0u, (uint32_t)d64};
  if (d64 <= 1) { f.d = 1; return f; }
  uint32_t s = 0;
  while ((1ull << s) < (uint64_t)d64) ++s;
  f.mul = (uint32_t)(((1ull << (31 + s)) / (uint64_t)d64) + 1);
  f.sh = s - 1;
  return f;
}

__device__ __forceinline__ int fdiv(int x, const FastDiv& f) {      // x >= 0
  return f.d == 1 ? x : (int)(__umulhi((uint32_t)x, f.mul) >> f.sh);
}

struct ConvP {
  int B, Ci, Co, H, W, KH, KW, Ho, Wo, sh, sw, ph, pw, dh, dw, G;
  int Cg, Cog;   // channels per group (in / out)
  FastDiv f_khw, f_kw, f_howo, f_wo, f_hw, f_w, f_sh, f_sw;
};

enum { MODE_FWD = 0, MODE_DGRAD = 1, MODE_WGRAD = 2 };

// Strided data gradient by PHASES.  Input pixel (ih, iw) receives tap (kh, kw) only if (ih + ph - kh dh) is a multiple of
// sh (and likewise in w): with stride 2 three taps of four multiply zeros in the plain implicit GEMM.  The input pixels
// are therefore split into sh x sw classes by (ih + ph) mod sh, (iw + pw) mod sw; inside a class the valid taps are the same
// for every pixel -- an arithmetic progression kh = kh0 + a khs, a < nh -- and the class is its own GEMM: N = pixels of the
// class, K = Cout x nh x nw.  blockIdx.z carries the class.  (Strides up to 2 per dimension; larger ones take the plain path.)
struct DgradPhase {
  int ih0, iw0, Hp, Wp;               // first row / column of the class, rows / columns in it
  int kh0, kw0, nh, nw;               // first valid tap and number of valid taps per dimension
  int N, K;                           // GEMM dims of the class
  FastDiv f_hpwp, f_wp, f_nhw, f_nw;
};
constexpr int kMaxPhases = 4;

struct ConvArgs {
  const void* xr; const void* xi;     // FWD: input      DGRAD: grad out   WGRAD: grad out
  const void* wr; const void* wi;     // FWD: weight     DGRAD: weight     WGRAD: input
  const float* bias_r; const float* bias_i;
  void* yr; void* yi;                 // output (WGRAD: fp32 partial slabs [splits][Co*Cg*KH*KW])
  ConvP p;
  int M, N, K;                        // GEMM dims per group (all below 2^31: checked at launch)
  int splits;                         // WGRAD split-K factor (1 otherwise)
  int kchunk;                         // K elements per split
  int nph;                            // DGRAD: number of phases (0 = plain path)
  int khs, kws;                       // DGRAD phases: tap steps
  DgradPhase phase[kMaxPhases];
  int bias_col;                       // WGRAD: 1 = one more GEMM column n == N whose B entries are (1, 0): its outputs are
                                      // the row sums of the A operand = the BIAS gradient (sum of G over batch and pixels),
                                      // written behind the weight slab of each split (slab stride wsz + Co)
};

template <typename T>
__device__ __forceinline__ float ldv(const void* p, int64_t off) {
  return io<T>::ld(reinterpret_cast<const T*>(p) + off);
}

// Implicit-GEMM tile: TM x 64 outputs per workgroup of 4 waves, K in steps of CBK = 16, double-buffered in LDS; the gathers
// of tile t + 1 are in flight while the MFMAs of tile t run.
//   TM = 64: waves 2 (M) x 2 (N), each a 32 x 32 MFMA block over the whole K step.
//   TM = 32 (M <= 32: narrow layers), TN = 64: waves 2 (N) x 2 (K halves of each step), partial sums joined through LDS
//            at the end -- twice the workgroups and half the MFMA chain per wave of a 32 x 128 tile.
//   TM = 32, TN = 32 (narrow layers with few pixels: fewer than two 32 x 64 tiles per CU): the four waves take a quarter
//            of each K step -- with one workgroup per CU a wave has its SIMD to itself, and its gather arithmetic, LDS
//            traffic and MFMAs simply add up (PMC: VALU 9 %, MFMA 10 % busy on cfg5's 7 x 7 layers); half the tile is half
//            of all three per wave, on twice the workgroups.
// Operand elements per thread and K step: A (TM x CBK): k = (t & 15) + 16 h, rows (t >> 4) + 16 j; B (TN x CBK): pixels
// fastest (FWD, DGRAD: row t % TN fixed, k = t / TN + (256 / TN) j) or k fastest (WGRAD: k = (t & 15) + 16 h, rows
// (t >> 4) + 16 j).  Whatever is fixed per thread is decomposed ONCE in front of the K loop.
// NARROW = 1 / 2 (TM = 32, M <= 8 / M <= 16: the first layers of small-width models, and every data gradient into them):
//   the products run on v_mfma_f32_4x4x1_16b_f32 -- 16 independent 4 x 4 blocks per instruction, here 8 pixel quads x 2
//   channel quads, i.e. 32 pixels x 8 channels x 1 k in 8 cycles -- once per 8-channel group, instead of on 32 x 32 x 2
//   tiles whose 32 rows an 8-channel operand fills to a quarter (PMC / ablation: the MFMAs were 34 of the 61 us of cfg5's
//   layer-2 data gradient).  Lane l: A = channel 4 (l >> 5) + (l & 3), B = pixel 4 ((l >> 2) & 7) + (l & 3), D register r =
//   channel 4 (l >> 5) + r of that pixel.
template <typename T, bool CPLX, int MODE, int TM, int TN, int NARROW>
__global__ __launch_bounds__(256) void conv_kernel(ConvArgs a) {
  static_assert((TM == 64 && TN == 64) || (TM == 32 && (TN == 64 || TN == 32)), "tile shapes");
  static_assert(NARROW == 0 || (TM == 32 && NARROW <= 2), "4 x 4 x 1 path: one or two 8-channel groups of a 32-row tile");
  constexpr int ALD = TM + 1, BLD = TN + 1, AJ = TM / 16;
  constexpr int WN = TN / 32, KS = 4 / ((TM / 32) * WN);       // waves along N; K parts of a step (1, 2 or 4)
  constexpr bool KSPLIT = KS > 1;
  constexpr int KBS = 256 / TN, BJ = CBK / KBS, WJ = TN / 16;  // B: k stride / elements per thread (pixels fastest); rows (k fastest)
  // LDS: operand rings (statically addressed: the MFMA loop's reads fold their offsets into the instructions); the
  // K-split's hand-over buffers (2 waves x 16 x 64 floats per plane) reuse the B rings once the K loop is done -- 25 KiB
  // per workgroup instead of 41: more workgroups per CU, which is what hides the load latencies of the short (1-5 K step)
  // workgroups of wide, shallow layers
  constexpr int NPL = CPLX ? 2 : 1;
  __shared__ float As_r[2][CBK][ALD], Bs_r[2][CBK][BLD];
  __shared__ float As_i[CPLX ? 2 : 1][CPLX ? CBK : 1][ALD], Bs_i[CPLX ? 2 : 1][CPLX ? CBK : 1][BLD];
  __shared__ float sbias[2][64];                               // this tile's bias (FWD)
  static_assert(KS != 2 || 2 * 16 * 64 <= 2 * CBK * BLD, "hand-over buffer must fit in a B ring");
  __shared__ float red4[KS == 4 ? 3 * NPL * 16 * 64 : 1];     // (three handing waves: does not fit in the 32-wide rings)
  float (*red_r)[16][64] = reinterpret_cast<float (*)[16][64]>(KS == 4 ? &red4[0] : &Bs_r[0][0][0]);
  float (*red_i)[16][64] = reinterpret_cast<float (*)[16][64]>(KS == 4 ? &red4[CPLX ? 3 * 16 * 64 : 0] : &Bs_i[0][0][0]);
  const ConvP& p = a.p;
  const int t = threadIdx.x;
  const bool phased = MODE == MODE_DGRAD && a.nph > 0;
  const int zz = phased ? blockIdx.z / a.nph : blockIdx.z;
  const DgradPhase& P = a.phase[phased ? blockIdx.z % a.nph : 0];
  const int g = zz / a.splits, split = zz % a.splits;
  const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;
  const int Nn = phased ? P.N : a.N;                           // this launch's (class's) pixels
  if (phased && n0 >= Nn) return;                              // (the grid is sized for the largest class)
  const int lane = t & 63, wid = t >> 6;
  const int wm = KSPLIT ? 0 : (wid >> 1) * 32, wn = (wid % WN) * 32, l31 = lane & 31, lk = lane >> 5;
  const int kq = KSPLIT ? wid / WN : 0;                        // which K part of each step
  const int kbeg = split * a.kchunk;
  int kend = kbeg + a.kchunk;
  if (kend > a.K) kend = a.K;
  if (phased) kend = P.K;
  // conjugation: DGRAD conj(W) is the A operand, WGRAD conj(X) is the B operand
  const float sa = (MODE == MODE_DGRAD) ? -1.f : 1.f, sb = (MODE == MODE_WGRAD) ? -1.f : 1.f;
  const int khw = p.KH * p.KW, howo = p.Ho * p.Wo;
  const void* a_r = MODE == MODE_WGRAD ? a.xr : a.wr;
  const void* a_i = MODE == MODE_WGRAD ? a.xi : a.wi;
  const void* b_r = MODE == MODE_WGRAD ? a.wr : a.xr;
  const void* b_i = MODE == MODE_WGRAD ? a.wi : a.xi;

  // ---- per-thread constants of the gathers
  const int ka = t & 15, ra0 = t >> 4;                         // A: k lane, first row
  const int rb = t % TN, kb0 = t / TN;                         // B (pixels fastest): row, first k
  // B row = output pixel (FWD) / input pixel (DGRAD): image, and the window origin
  bool b_ok = false;
  int b_h0 = 0, b_w0 = 0;
  int64_t b_base = 0;
  if (MODE != MODE_WGRAD) {
    const int n = n0 + rb;
    b_ok = n < Nn;
    if (b_ok) {
      if (phased) {
        const int b = fdiv(n, P.f_hpwp), r = n - b * (P.Hp * P.Wp);
        const int tq = fdiv(r, P.f_wp), uq = r - tq * P.Wp;
        b_h0 = P.ih0 + tq * p.sh + p.ph; b_w0 = P.iw0 + uq * p.sw + p.pw;
        b_base = ((int64_t)b * p.Co + (int64_t)g * p.Cog) * howo;
      } else if (MODE == MODE_FWD) {
        const int b = fdiv(n, p.f_howo), r = n - b * howo;
        const int oh = fdiv(r, p.f_wo), ow = r - oh * p.Wo;
        b_h0 = oh * p.sh - p.ph; b_w0 = ow * p.sw - p.pw;
        b_base = ((int64_t)b * p.Ci + (int64_t)g * p.Cg) * p.H * p.W;
      } else {
        const int b = fdiv(n, p.f_hw), r = n - b * (p.H * p.W);
        const int ih = fdiv(r, p.f_w), iw = r - ih * p.W;
        b_h0 = ih + p.ph; b_w0 = iw + p.pw;
        b_base = ((int64_t)b * p.Co + (int64_t)g * p.Cog) * howo;
      }
    }
  }
  // WGRAD: B rows = (ci, kh, kw) taps
  int w_ci[WJ], w_dh[WJ], w_dw[WJ];
#pragma unroll
  for (int j = 0; j < WJ; ++j) {
    w_ci[j] = -1; w_dh[j] = w_dw[j] = 0;
    if (MODE == MODE_WGRAD) {
      const int n = n0 + ra0 + 16 * j;
      if (n < a.N) {
        const int ci = fdiv(n, p.f_khw), rk = n - ci * khw;
        const int kh = fdiv(rk, p.f_kw), kw = rk - kh * p.KW;
        w_ci[j] = ci; w_dh[j] = kh * p.dh - p.ph; w_dw[j] = kw * p.dw - p.pw;
      } else if (n == a.N && a.bias_col) {
        w_ci[j] = -2;                                            // the ones column
      }
    }
  }

  constexpr int NB = MODE == MODE_WGRAD ? KH16 * WJ : BJ;      // B elements per thread and K step
  float ar_[KH16][AJ], ai_[KH16][AJ], br_[NB], bi_[NB];
  auto fetch_both = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
    for (int h = 0; h < KH16; ++h) {
      // ---- A: weights (FWD: k contiguous; DGRAD: gathered) / grad-out (WGRAD: k = pixels contiguous)
      const int gk = k0 + ka + 16 * h;
      const bool kok = gk < kend;
      int64_t abase = 0, astride = 0;                            // element (row m, this k) = abase + m * astride
      int wb = 0, woh = 0, wow = 0;                              // WGRAD: pixel of this k
      if (MODE == MODE_FWD) {
        abase = (int64_t)g * p.Cog * a.K + gk; astride = a.K;
      } else if (MODE == MODE_DGRAD) {
        int co, r;
        if (phased) {
          co = kok ? fdiv(gk, P.f_nhw) : 0;
          const int rk = gk - co * (P.nh * P.nw), ta = kok ? fdiv(rk, P.f_nw) : 0;
          r = (P.kh0 + ta * a.khs) * p.KW + P.kw0 + (rk - ta * P.nw) * a.kws;
        } else {
          co = kok ? fdiv(gk, p.f_khw) : 0; r = gk - co * khw;
        }
        abase = ((int64_t)g * p.Cog + co) * p.Cg * khw + r; astride = khw;
      } else {
        wb = kok ? fdiv(gk, p.f_howo) : 0;
        const int r = gk - wb * howo;
        woh = kok ? fdiv(r, p.f_wo) : 0; wow = r - woh * p.Wo;
        abase = ((int64_t)wb * p.Co + (int64_t)g * p.Cog) * howo + r; astride = howo;
      }
#pragma unroll
      for (int j = 0; j < AJ; ++j) {
        const int m = m0 + ra0 + 16 * j;
        float vr = 0.f, vi = 0.f;
        if (kok && m < a.M) {
          const int64_t off = abase + (int64_t)m * astride;
          vr = ldv<T>(a_r, off);
          if (CPLX) vi = ldv<T>(a_i, off);
        }
        ar_[h][j] = vr; ai_[h][j] = vi;
      }
      // ---- B, k fastest (WGRAD): the same k, rows = taps
      if (MODE == MODE_WGRAD) {
        const int ihb = woh * p.sh, iwb = wow * p.sw;
        const int64_t xb = ((int64_t)wb * p.Ci + (int64_t)g * p.Cg) * p.H * p.W;
#pragma unroll
        for (int j = 0; j < WJ; ++j) {
          float vr = 0.f, vi = 0.f;
          const int ih = ihb + w_dh[j], iw = iwb + w_dw[j];
          if (kok && w_ci[j] >= 0 && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W) {
            const int64_t off = xb + ((int64_t)w_ci[j] * p.H + ih) * p.W + iw;
            vr = ldv<T>(b_r, off);
            if (CPLX) vi = ldv<T>(b_i, off);
          } else if (kok && w_ci[j] == -2) {
            vr = 1.f;
          }
          br_[h * WJ + j] = vr; bi_[h * WJ + j] = vi;
        }
      }
    }
    // ---- B, pixels fastest (FWD, DGRAD)
    if (MODE != MODE_WGRAD) {
#pragma unroll
      for (int j = 0; j < BJ; ++j) {
        // (64-pixel tiles: kb0 is the wave index, so k -- and its split into channel and tap -- is the same for the 64 lanes
        //  and runs on the scalar unit; PMC: the vector ALUs are the busiest unit of the wide layers' launches)
        const int k = TN == 64 ? __builtin_amdgcn_readfirstlane(k0 + kb0 + KBS * j) : k0 + kb0 + KBS * j;
        float vr = 0.f, vi = 0.f;
        if (b_ok && k < kend) {
          int c, kh, kw;                                         // c: input channel (FWD) / output channel (DGRAD)
          if (phased) {
            c = fdiv(k, P.f_nhw);
            const int rk = k - c * (P.nh * P.nw), ta = fdiv(rk, P.f_nw);
            kh = P.kh0 + ta * a.khs; kw = P.kw0 + (rk - ta * P.nw) * a.kws;
          } else {
            c = fdiv(k, p.f_khw);
            const int rk = k - c * khw;
            kh = fdiv(rk, p.f_kw); kw = rk - kh * p.KW;
          }
          if (MODE == MODE_FWD) {
            const int ih = b_h0 + kh * p.dh, iw = b_w0 + kw * p.dw;
            if (ih >= 0 && ih < p.H && iw >= 0 && iw < p.W) {
              const int64_t off = b_base + ((int64_t)c * p.H + ih) * p.W + iw;
              vr = ldv<T>(b_r, off);
              if (CPLX) vi = ldv<T>(b_i, off);
            }
          } else {
            const int th = b_h0 - kh * p.dh, tw = b_w0 - kw * p.dw;
            if (th >= 0 && tw >= 0) {
              const int oh = fdiv(th, p.f_sh), ow = fdiv(tw, p.f_sw);
              if (oh * p.sh == th && ow * p.sw == tw && oh < p.Ho && ow < p.Wo) {
                const int64_t off = b_base + ((int64_t)c * p.Ho + oh) * p.Wo + ow;
                vr = ldv<T>(b_r, off);
                if (CPLX) vi = ldv<T>(b_i, off);
              }
            }
          }
        }
        br_[j] = vr; bi_[j] = vi;
      }
    }
  };
  auto commit = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int h = 0; h < KH16; ++h)
#pragma unroll
      for (int j = 0; j < AJ; ++j) {
        As_r[buf][ka + 16 * h][ra0 + 16 * j] = ar_[h][j];
        if (CPLX) As_i[buf][ka + 16 * h][ra0 + 16 * j] = ai_[h][j];
      }
    if (MODE == MODE_WGRAD) {
#pragma unroll
      for (int h = 0; h < KH16; ++h)
#pragma unroll
        for (int j = 0; j < WJ; ++j) {
          Bs_r[buf][ka + 16 * h][ra0 + 16 * j] = br_[h * WJ + j];
          if (CPLX) Bs_i[buf][ka + 16 * h][ra0 + 16 * j] = bi_[h * WJ + j];
        }
    } else {
#pragma unroll
      for (int j = 0; j < BJ; ++j) {
        Bs_r[buf][kb0 + KBS * j][rb] = br_[j];
        if (CPLX) Bs_i[buf][kb0 + KBS * j][rb] = bi_[j];
      }
    }
  };

  f32x16 acc_r = {0}, acc_i = {0};
  constexpr int NCG = NARROW ? NARROW : 1;
  f32x4 nr[NCG], ni[NCG];
#pragma unroll
  for (int c = 0; c < NCG; ++c) { nr[c] = f32x4{0.f, 0.f, 0.f, 0.f}; ni[c] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  const int chl = (lane >> 5) * 4 + (lane & 3), pxl = ((lane >> 2) & 7) * 4 + (lane & 3);   // NARROW: this lane's channel / pixel
  if (MODE == MODE_FWD && a.bias_r && t < 2 * TM) {            // the tile's bias -> LDS now, not a dependent load in the epilogue
    const int pl = t / TM, m = m0 + (t - pl * TM);
    if (pl == 0 || CPLX) sbias[pl][t - pl * TM] = m < a.M ? (pl ? a.bias_i : a.bias_r)[g * p.Cog + m] : 0.f;
  }
  if (kbeg < kend) fetch_both(kbeg);
  int buf = 0;
  for (int k0 = kbeg; k0 < kend; k0 += CBK, buf ^= 1) {
    commit(buf);
    __syncthreads();                         // tile visible; the other buffer is free again
    if (k0 + CBK < kend) fetch_both(k0 + CBK);
    constexpr int KKN = CBK / KS;
    if (NARROW) {
#pragma unroll
      for (int kk = 0; kk < KKN; ++kk) {
        const int kr = kq * KKN + kk;
        const float br = Bs_r[buf][kr][wn + pxl];
        const float bi = CPLX ? sb * Bs_i[buf][kr][wn + pxl] : 0.f;
#pragma unroll
        for (int c = 0; c < NCG; ++c) {
          const float ar = As_r[buf][kr][c * 8 + chl];
          nr[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(ar, br, nr[c], 0, 0, 0);
          if (CPLX) {
            const float ai = sa * As_i[buf][kr][c * 8 + chl];
            nr[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(-ai, bi, nr[c], 0, 0, 0);
            ni[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(ar, bi, ni[c], 0, 0, 0);
            ni[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(ai, br, ni[c], 0, 0, 0);
          }
        }
      }
      continue;
    }
#pragma unroll
    for (int kk = 0; kk < KKN; kk += 2) {
      const int kr = kq * KKN + kk + lk;
      const float ar = As_r[buf][kr][wm + l31];
      const float br = Bs_r[buf][kr][wn + l31];
      acc_r = __builtin_amdgcn_mfma_f32_32x32x2f32(ar, br, acc_r, 0, 0, 0);
      if (CPLX) {
        const float ai = sa * As_i[buf][kr][wm + l31];
        const float bi = sb * Bs_i[buf][kr][wn + l31];
        acc_r = __builtin_amdgcn_mfma_f32_32x32x2f32(-ai, bi, acc_r, 0, 0, 0);
        acc_i = __builtin_amdgcn_mfma_f32_32x32x2f32(ar, bi, acc_i, 0, 0, 0);
        acc_i = __builtin_amdgcn_mfma_f32_32x32x2f32(ai, br, acc_i, 0, 0, 0);
      }
    }
  }
  constexpr int NR = NARROW ? 4 * NARROW : 16;                 // result registers per lane and plane
  if (NARROW) {
#pragma unroll
    for (int c = 0; c < NCG; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) { acc_r[c * 4 + r] = nr[c][r]; acc_i[c * 4 + r] = ni[c][r]; }
  }
  if (KSPLIT) {                              // join the K parts: the waves with kq > 0 hand theirs to the kq == 0 waves
    __syncthreads();                         // (every wave is done with the B rings the hand-over buffer may live in)
    if (kq > 0) {
      const int slot = (kq - 1) * WN + (wid % WN);
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        red_r[slot][r][lane] = acc_r[r];
        if (CPLX) red_i[slot][r][lane] = acc_i[r];
      }
    }
    __syncthreads();
    if (kq > 0) return;
#pragma unroll
    for (int q = 0; q < KS - 1; ++q)
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        acc_r[r] += red_r[q * WN + (wid % WN)][r][lane];
        if (CPLX) acc_i[r] += red_i[q * WN + (wid % WN)][r][lane];
      }
  }

  // epilogue: col = lane & 31 (NARROW: the lane's pixel) runs along N, rows (M) across registers
  const int n = n0 + wn + (NARROW ? pxl : l31);
  if (n >= Nn + (MODE == MODE_WGRAD ? a.bias_col : 0)) return;
  int64_t out_base, out_mstride;
  if (MODE == MODE_FWD) {
    const int b = fdiv(n, p.f_howo), r = n - b * howo;
    out_base = ((int64_t)b * p.Co + (int64_t)g * p.Cog) * howo + r;
    out_mstride = howo;
  } else if (MODE == MODE_DGRAD) {
    const int hw = p.H * p.W;
    int b, r;
    if (phased) {
      b = fdiv(n, P.f_hpwp);
      const int rr = n - b * (P.Hp * P.Wp), tq = fdiv(rr, P.f_wp);
      r = (P.ih0 + tq * p.sh) * p.W + P.iw0 + (rr - tq * P.Wp) * p.sw;
    } else {
      b = fdiv(n, p.f_hw); r = n - b * hw;
    }
    out_base = ((int64_t)b * p.Ci + (int64_t)g * p.Cg) * hw + r;
    out_mstride = hw;
  } else {
    const int64_t wsz = (int64_t)p.Co * p.Cg * p.KH * p.KW, slab = wsz + (a.bias_col ? p.Co : 0);
    if (n < a.N) { out_base = (int64_t)split * slab + (int64_t)g * p.Cog * a.N + n; out_mstride = a.N; }
    else { out_base = (int64_t)split * slab + wsz + (int64_t)g * p.Cog; out_mstride = 1; }
  }
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int m = NARROW ? m0 + (r >> 2) * 8 + (lane >> 5) * 4 + (r & 3) : m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * lk;
    if (m >= a.M) continue;
    const int64_t o = out_base + (int64_t)m * out_mstride;
    float vr = acc_r[r], vi = acc_i[r];
    if (MODE == MODE_FWD && a.bias_r) {
      vr += sbias[0][m - m0];
      if (CPLX) vi += sbias[1][m - m0];
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

// out[i] = (sum_s slab[s][i]) (* emul[i]).  Block = 64 consecutive elements x 16 split lanes (one wave
// per split lane, coalesced 256-B reads, 4 loads in flight per thread), fixed summation order.  blockIdx.y = plane
// (real / imaginary slabs and outputs; the multiplier applies to plane 0: the real-valued variance path).
__global__ __launch_bounds__(1024) void slab_sum_kernel(const float* slabs0, const float* slabs1, int splits, int64_t n,
                                                        const float* emul, float* out0, float* out1, int64_t nw,
                                                        float* outb0, float* outb1) {
  // elements [0, nw) of a slab -> out (weights), [nw, n) -> outb (the fused bias-gradient column, conv_kernel WGRAD)
  __shared__ float red[16][64];
  const float* slabs = blockIdx.y ? slabs1 : slabs0;
  float* out = blockIdx.y ? out1 : out0;
  float* outb = blockIdx.y ? outb1 : outb0;
  if (blockIdx.y) emul = nullptr;
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
    if (i < nw) out[i] = emul ? acc * emul[i] : acc;
    else outb[i - nw] = acc;
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
// both planes of a complex tensor per launch (the complex bias gradient): grid (C, chunks), partial[plane][chunk][c];
// chansum_final then runs over 2 C "channels".  (Finishing in the same launch -- the last block of a channel to take a
// ticket adds up the partials -- was measured SLOWER on this chip: device-scope visibility between the XCDs' L2s costs
// more than the ~5 us of a second launch, 16 us against 2 x 4.9; with __threadfence() 78 us.)
template <typename T>
__global__ __launch_bounds__(256) void chansum2_partial(const T* xr, const T* xi, int64_t B, int C, int64_t S, double* partial) {
  __shared__ double red[4];
  const int c = blockIdx.x, chunks = gridDim.y;
  struct __attribute__((packed, aligned(sizeof(T)))) V8 { T v[8]; };
  double acc[2] = {0.0, 0.0};
  const int64_t S8 = S & ~(int64_t)7;
  for (int64_t b = blockIdx.y; b < B; b += chunks) {
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
      const T* p = (pl ? xi : xr) + (b * C + c) * S;
      for (int64_t s = (int64_t)threadIdx.x * 8; s < S8; s += 256 * 8) {
        const V8 v = *reinterpret_cast<const V8*>(p + s);
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) q += io<T>::ld(&v.v[e]);
        acc[pl] += (double)q;
      }
      for (int64_t s = S8 + threadIdx.x; s < S; s += 256) acc[pl] += (double)io<T>::ld(p + s);
    }
  }
#pragma unroll
  for (int pl = 0; pl < 2; ++pl) {
    const double t = block_sum<double, 256>(acc[pl], red);
    // laid out as chansum_final reads it for 2 C channels: [chunk][plane * C + c]
    if (threadIdx.x == 0) partial[(int64_t)blockIdx.y * 2 * C + (int64_t)pl * C + c] = t;
  }
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

static int gcd_i(int x, int y) { while (y) { const int t = x % y; x = y; y = t; } return x; }

// DGRAD: the phase plan (DgradPhase) for strides up to 2 per dimension; leaves a.nph = 0 (plain path) for unit stride,
// larger strides, or when a phase's pixel count would not fit the 32-bit index math
static void plan_dgrad_phases(ConvArgs& a) {
  const ConvP& p = a.p;
  a.nph = 0;
  if ((p.sh == 1 && p.sw == 1) || p.sh > 2 || p.sw > 2) return;
  a.khs = p.sh / gcd_i(p.dh, p.sh);
  a.kws = p.sw / gcd_i(p.dw, p.sw);
  int n = 0;
  for (int rh = 0; rh < p.sh; ++rh)
    for (int rw = 0; rw < p.sw; ++rw) {
      DgradPhase& q = a.phase[n++];
      q.ih0 = ((rh - p.ph) % p.sh + p.sh) % p.sh;
      q.iw0 = ((rw - p.pw) % p.sw + p.sw) % p.sw;
      q.Hp = q.ih0 < p.H ? (p.H - q.ih0 + p.sh - 1) / p.sh : 0;
      q.Wp = q.iw0 < p.W ? (p.W - q.iw0 + p.sw - 1) / p.sw : 0;
      q.kh0 = q.kw0 = 0; q.nh = q.nw = 0;
      for (int kh = 0; kh < p.KH; ++kh)
        if ((kh * p.dh) % p.sh == rh) { if (!q.nh) q.kh0 = kh; ++q.nh; }
      for (int kw = 0; kw < p.KW; ++kw)
        if ((kw * p.dw) % p.sw == rw) { if (!q.nw) q.kw0 = kw; ++q.nw; }
      q.N = p.B * q.Hp * q.Wp;
      q.K = p.Cog * q.nh * q.nw;
      q.f_hpwp = make_fastdiv((int64_t)q.Hp * q.Wp); q.f_wp = make_fastdiv(q.Wp);
      q.f_nhw = make_fastdiv((int64_t)q.nh * q.nw); q.f_nw = make_fastdiv(q.nw);
    }
  a.nph = n;
}

static int conv_tile_m(int64_t M) { return M <= 32 ? 32 : CBM; }
// narrow layers with few pixels: 32-pixel tiles when 64-pixel ones would leave the chip with fewer than two per CU
static int conv_tile_n(int64_t M, int64_t N, int64_t zdim) {
  return (M <= 32 && ((N + CBN - 1) / CBN) * ((M + 31) / 32) * zdim < 512) ? 32 : CBN;
}

template <typename T, int MODE>
static int conv_launch(ConvArgs& a, bool cplx, hipStream_t st) {
  int64_t nmax = a.N;                                          // DGRAD by phases: the largest class sizes the grid
  if (MODE == MODE_DGRAD && a.nph > 0) {
    nmax = 0;
    for (int i = 0; i < a.nph; ++i) nmax = a.phase[i].N > nmax ? a.phase[i].N : nmax;
  }
  const int zmul = (MODE == MODE_DGRAD && a.nph > 0) ? a.nph : 1;
  const int tm = conv_tile_m(a.M), tn = conv_tile_n(a.M, nmax, (int64_t)a.p.G * a.splits * zmul);
  dim3 grid((unsigned)((nmax + a.bias_col + tn - 1) / tn), (unsigned)((a.M + tm - 1) / tm),
            (unsigned)(a.p.G * a.splits * zmul));
  if (grid.y > 65535 || grid.z > 65535) return CPLXAMD_ESHAPE;
#define CONV_GO(TM_, TN_, NW_)                                                              \
  do {                                                                                      \
    if (cplx) conv_kernel<T, true, MODE, TM_, TN_, NW_><<<grid, 256, 0, st>>>(a);           \
    else conv_kernel<T, false, MODE, TM_, TN_, NW_><<<grid, 256, 0, st>>>(a);               \
  } while (0)
  const int nw = a.M <= 8 ? 1 : (a.M <= 16 ? 2 : 0);           // 8-channel groups on the 4 x 4 x 1 path
  if (tm == 64) CONV_GO(64, 64, 0);
  else if (tn == 64) { if (nw == 1) CONV_GO(32, 64, 1); else if (nw == 2) CONV_GO(32, 64, 2); else CONV_GO(32, 64, 0); }
  else { if (nw == 1) CONV_GO(32, 32, 1); else if (nw == 2) CONV_GO(32, 32, 2); else CONV_GO(32, 32, 0); }
#undef CONV_GO
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
  if (!conv_geom_ok(p)) return CPLXAMD_ESHAPE;
  // the gathers index pixels and taps in 32 bits (offsets into the tensors stay 64-bit)
  const int64_t lim = 0x7fe00000;   // (room for the split-K chunk rounding)
  if ((int64_t)p.B * p.Ho * p.Wo > lim || (int64_t)p.B * p.H * p.W > lim || (int64_t)p.Cg * p.KH * p.KW > lim ||
      (int64_t)p.Cog * p.KH * p.KW > lim)
    return CPLXAMD_ESHAPE;
  p.f_khw = make_fastdiv((int64_t)p.KH * p.KW); p.f_kw = make_fastdiv(p.KW);
  p.f_howo = make_fastdiv((int64_t)p.Ho * p.Wo); p.f_wo = make_fastdiv(p.Wo);
  p.f_hw = make_fastdiv((int64_t)p.H * p.W); p.f_w = make_fastdiv(p.W);
  p.f_sh = make_fastdiv(p.sh); p.f_sw = make_fastdiv(p.sw);
  return 0;
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
  a.M = a.p.Cog; a.N = a.p.B * a.p.Ho * a.p.Wo; a.K = a.p.Cg * a.p.KH * a.p.KW;
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
  a.M = a.p.Cg; a.N = a.p.B * a.p.H * a.p.W; a.K = a.p.Cog * a.p.KH * a.p.KW;
  a.splits = 1; a.kchunk = a.K;
  plan_dgrad_phases(a);
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
  const int tm = conv_tile_m(p.Cog);
  const int64_t tiles = (int64_t)((p.Cog + tm - 1) / tm) * (((int64_t)p.Cg * p.KH * p.KW + CBN - 1) / CBN) * p.G;   // (64-wide: the split plan comes first)
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
  // [splits][planes] slabs of wsz + Co floats (the fused bias-gradient column), then the bias fallback's partial sums
  return ((int64_t)cplxamd_conv2d_wgrad_splits(geom) * (wsz + p.Co) * sizeof(float) * (cplx ? 2 : 1) + 255) / 256 * 256 +
         (int64_t)2 * 64 * p.Co * sizeof(double);
}

int cplxamd_chansum(const void* x, float* out, int64_t B, int C, int64_t S, int dtype, void* ws, void* stream);
int cplxamd_chansum2(const void* xr, const void* xi, float* out_r, float* out_i, int64_t B, int C, int64_t S, int dtype,
                     void* ws, void* stream);

int cplxamd_conv2d_wgrad_bias(const void* gr, const void* gi, const void* xr, const void* xi,
                              const float* emul, float* dwr, float* dwi, float* dbr, float* dbi, const int* geom,
                              int dtype, void* ws, int64_t ws_bytes, void* stream) {
  if (!gr || !xr || !dwr || !geom || !ws) return CPLXAMD_EINVAL;
  const bool cplx = gi != nullptr;
  if (cplx && (!xi || !dwi)) return CPLXAMD_EINVAL;
  const bool want_b = dbr != nullptr;
  if (want_b && cplx && dbi != dbr + geom[2]) return CPLXAMD_EINVAL;       // one [2][Co] array
  ConvArgs a{};
  int rc = fill_geom(geom, a.p);
  if (rc) return rc;
  const int64_t wsz = (int64_t)a.p.Co * a.p.Cg * a.p.KH * a.p.KW;
  if (a.p.B == 0) {                                            // empty batch: the gradient is zero
    hipError_t e = hipMemsetAsync(dwr, 0, wsz * sizeof(float), (hipStream_t)stream);
    if (e == hipSuccess && cplx) e = hipMemsetAsync(dwi, 0, wsz * sizeof(float), (hipStream_t)stream);
    if (e == hipSuccess && want_b) e = hipMemsetAsync(dbr, 0, (cplx ? 2 : 1) * a.p.Co * sizeof(float), (hipStream_t)stream);
    return (int)e;
  }
  if (ws_bytes < cplxamd_conv2d_wgrad_ws_bytes(geom, cplx)) return CPLXAMD_EWS;
  a.splits = cplxamd_conv2d_wgrad_splits(geom);
  a.M = a.p.Cog; a.N = a.p.Cg * a.p.KH * a.p.KW; a.K = a.p.B * a.p.Ho * a.p.Wo;
  // the bias gradient rides as one more column of the GEMM where that column does not start a tile of its own
  const int tn = conv_tile_n(a.M, a.N, (int64_t)a.p.G * a.splits);
  a.bias_col = (want_b && a.N % tn != 0) ? 1 : 0;
  const int64_t slab = wsz + (a.bias_col ? a.p.Co : 0);
  a.xr = gr; a.xi = gi; a.wr = xr; a.wi = xi;
  a.yr = ws; a.yi = (float*)ws + (int64_t)a.splits * slab;
  a.kchunk = (int)((((int64_t)a.K + a.splits - 1) / a.splits + CBK - 1) / CBK * CBK);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CPLXAMD_F32) rc = conv_launch<float, MODE_WGRAD>(a, cplx, st);
  else if (dtype == CPLXAMD_BF16) rc = conv_launch<bf16_t, MODE_WGRAD>(a, cplx, st);
  else return CPLXAMD_EINVAL;
  if (rc) return rc;
  const int sgrid = (int)((slab + 63) / 64);
  slab_sum_kernel<<<dim3(sgrid, cplx ? 2 : 1), 1024, 0, st>>>((const float*)a.yr, (const float*)a.yi, a.splits, slab, emul,
                                                              dwr, dwi, wsz, dbr, dbi);
  CPLXAMD_CHECK_LAUNCH();
  if (want_b && !a.bias_col) {                                 // (the column would have cost a tile column: two small launches)
    void* cws = (char*)ws + ((int64_t)a.splits * (wsz + a.p.Co) * sizeof(float) * (cplx ? 2 : 1) + 255) / 256 * 256;
    const int64_t S = (int64_t)a.p.Ho * a.p.Wo;
    return cplx ? cplxamd_chansum2(gr, gi, dbr, dbi, a.p.B, a.p.Co, S, dtype, cws, stream)
                : cplxamd_chansum(gr, dbr, a.p.B, a.p.Co, S, dtype, cws, stream);
  }
  return 0;
}

int cplxamd_conv2d_wgrad(const void* gr, const void* gi, const void* xr, const void* xi,
                         const float* emul, float* dwr, float* dwi, const int* geom, int dtype,
                         void* ws, int64_t ws_bytes, void* stream) {
  return cplxamd_conv2d_wgrad_bias(gr, gi, xr, xi, emul, dwr, dwi, nullptr, nullptr, geom, dtype, ws, ws_bytes, stream);
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

/* Complex bias gradient, both planes per launch: out_r[c], out_i[c] = sums over (b, spatial) of the two NCHW planes.
 * ws >= 2 * 64 * C doubles.  out_r and out_i must be the two halves of one [2][C] array (out_i == out_r + C). */
int cplxamd_chansum2(const void* xr, const void* xi, float* out_r, float* out_i, int64_t B, int C, int64_t S, int dtype,
                     void* ws, void* stream) {
  if (!xr || !xi || !out_r || !out_i || !ws || B <= 0 || C <= 0 || S <= 0) return CPLXAMD_EINVAL;
  if (out_i != out_r + C) return CPLXAMD_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int chunks = (int)(B < 64 ? B : 64);
  dim3 grid(C, chunks);
  if (dtype == CPLXAMD_F32)
    chansum2_partial<float><<<grid, 256, 0, st>>>((const float*)xr, (const float*)xi, B, C, S, (double*)ws);
  else if (dtype == CPLXAMD_BF16)
    chansum2_partial<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)xr, (const bf16_t*)xi, B, C, S, (double*)ws);
  else
    return CPLXAMD_EINVAL;
  CPLXAMD_CHECK_LAUNCH();
  chansum_final<<<2 * C, 64, 0, st>>>((const double*)ws, chunks, 2 * C, out_r);
  CPLXAMD_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
