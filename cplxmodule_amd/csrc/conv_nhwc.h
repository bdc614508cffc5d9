// Shared pieces of the channels-last convolution kernels (conv_nhwc.hip: bf16, conv_nhwc_f32.hip:
// exact float32): launch geometry, kernel arguments, and the epilogue that turns the wave tiles'
// accumulators into contiguous NCHW runs through LDS.
#pragma once
#include "common.h"

namespace cplxamd {

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace cn {

constexpr int BM = 256, BN = 64, NT = BM;
constexpr int MAX_EXTRA = 32;                       // (KW-1)*dil_w rows of halo at most
constexpr int OUT_LD = BM + 8;                      // epilogue image [channel][pixel], padded rows

struct Args {
  const void* x_r; const void* x_i;       // channels-last grid (bf16 or float32, per kernel)
  const void* w_r; const void* w_i;       // packed weights (same element type)
  const float* bias_r; const float* bias_i;
  void* y_r; void* y_i;
  int64_t rows;                 // B * Hp * Wp
  int B, Hp, Wp, C, Cout, KH, KW, dil_h, dil_w, Ho, Wo;   // Ho x Wo: extent of the output image
  int64_t row_bias;             // added to every input row index (<= 0: the data gradient reads backwards)
  int oh, ow;                   // grid position of output pixel (0, 0)
  int srows;                    // staged input rows per tile: BM + (KW-1)*dil_w
  int npieces;                  // LDS-DMA pieces per stage that carry data (the rest go to a dump slot)
  int dbg;                      // ablation bit (CPLXAMD_CONV_DBG): 4 no stores
};

struct __attribute__((packed, aligned(2))) bf8_t { uint4 v; };    // 16 B at 2-byte alignment

// accumulators (4 waves x 64 x 64, C/D layout of the 32x32 MFMA with the pixel operand first: column
// = channel = lane & 31, rows = pixels 8 q + 4 (lane >> 5) + e) -> LDS image [plane][64 channels]
// [256 pixels] (bf16 / f32 as TOUT) -> each thread stores 8 consecutive pixels of one channel, a wave
// 2 x 256 contiguous pixels.  The caller has drained its LDS-DMA (vmcnt(0)).
template <typename TOUT, bool CPLX>
__device__ __forceinline__ void store_tile(const Args& g, char* smem, const f32x16 (&acc_r)[2][2],
                                           const f32x16 (&acc_i)[2][2], int64_t m0, int n0) {
  constexpr int NP = CPLX ? 2 : 1;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid * 64, l31 = lane & 31, lk = lane >> 5;
  if (g.dbg & 4) {
    if (acc_r[0][0][0] == 123.456f) reinterpret_cast<TOUT*>(g.y_r)[0] = (TOUT)1;   // keep live
    return;
  }
  TOUT* yr = reinterpret_cast<TOUT*>(g.y_r);
  TOUT* yi = reinterpret_cast<TOUT*>(g.y_i);
  const int64_t plane_sz = (int64_t)g.Ho * g.Wo;
  // this thread's 8 pixels (same for every channel it stores)
  constexpr int MT = BM / 8, NR = NT / MT;                      // threads per pixel row, channels per sweep
  const int m8 = (tid % MT) * 8;
  int64_t off[8];
  bool ok[8];
  {
    const uint32_t img = (uint32_t)g.Hp * (uint32_t)g.Wp;
    const int64_t m = m0 + m8;
    uint32_t b = (uint32_t)(m < g.rows ? m : 0) / img;          // rows < 2^31 (checked by the host)
    const uint32_t rem = (uint32_t)(m < g.rows ? m : 0) - b * img;
    uint32_t hp = rem / (uint32_t)g.Wp, wp = rem - hp * (uint32_t)g.Wp;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const uint32_t ho = hp - (uint32_t)g.oh, wo = wp - (uint32_t)g.ow;   // wraps when before the image
      ok[e] = m + e < g.rows && b < (uint32_t)g.B && ho < (uint32_t)g.Ho && wo < (uint32_t)g.Wo;
      off[e] = (int64_t)b * g.Cout * plane_sz + (int64_t)ho * g.Wo + wo;
      if (++wp >= (uint32_t)g.Wp) { wp = 0; if (++hp >= (uint32_t)g.Hp) { hp = 0; ++b; } }
    }
  }
  const bool run8 = ok[0] && ok[7] && off[7] == off[0] + 7;
  constexpr int PASSES = sizeof(TOUT) == 2 ? 1 : 2;            // f32: the image is 2 x 64 KiB, one plane at a time
#pragma unroll
  for (int pass = 0; pass < (CPLX ? PASSES : 1); ++pass) {
    __syncthreads();                                            // ring (or previous pass) no longer read
    TOUT* img_lds = reinterpret_cast<TOUT*>(smem);
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) {
      if (PASSES == 2 && pl != pass) continue;
      TOUT* dst = img_lds + (PASSES == 2 ? 0 : pl) * BN * OUT_LD;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int n = j * 32 + l31;
          const int gn = n0 + n;
          const float bias = pl ? ((g.bias_i && gn < g.Cout) ? g.bias_i[gn] : 0.f)
                                : ((g.bias_r && gn < g.Cout) ? g.bias_r[gn] : 0.f);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            f4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v.v[e] = (pl ? acc_i[i][j][4 * q + e] : acc_r[i][j][4 * q + e]) + bias;
            st4(dst + n * OUT_LD + wm + i * 32 + 8 * q + 4 * lk, v);
          }
        }
    }
    __syncthreads();
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) {
      if (PASSES == 2 && pl != pass) continue;
      const TOUT* src = img_lds + (PASSES == 2 ? 0 : pl) * BN * OUT_LD;
      TOUT* out = pl ? yi : yr;
#pragma unroll
      for (int r = 0; r < BN / NR; ++r) {
        const int n = tid / MT + NR * r;
        if (n0 + n >= g.Cout) continue;
        const TOUT* p = src + n * OUT_LD + m8;
        TOUT* o = out + (int64_t)(n0 + n) * plane_sz;
        if (run8) {
          if (sizeof(TOUT) == 2) {
            *reinterpret_cast<bf8_t*>(o + off[0]) = bf8_t{*reinterpret_cast<const uint4*>(p)};
          } else {
            *reinterpret_cast<bf8_t*>(o + off[0]) = bf8_t{*reinterpret_cast<const uint4*>(p)};
            *reinterpret_cast<bf8_t*>(o + off[0] + 4) = bf8_t{*reinterpret_cast<const uint4*>(p + 4)};
          }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (ok[e]) o[off[e]] = p[e];
        }
      }
    }
  }
}

}  // namespace cn
}  // namespace cplxamd
