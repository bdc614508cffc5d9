// Exact-float32 counterpart of conv_nhwc.hip: complex / real 2-d convolution (stride 1, groups 1)
// as a shifted-row GEMM over a zero-padded channels-last float32 grid, on v_mfma_f32_32x32x2_f32
// (float32 products and accumulation: the parity path for float32 models, 157 TFLOP/s peak).
//
// Same structure as the bf16 kernel -- 256-row x 64-channel tile, 4 waves x (64 x 64), stage =
// (kernel row kh, 16 channels) = 64-B rows moved by LDS-DMA, the KW taps of a kernel row read one
// staged tile at LDS row offset kw*dil_w, weights from global memory, LDS-transposed epilogue, the
// data gradient = the same kernel reading the gradient grid backwards (conv_nhwc.h) -- with the
// fragment handling of the k = 2 MFMA: one ds_read_b128 (a row's 4 consecutive channels) feeds two
// MFMA steps, lane half lk = lane >> 5 taking channels {lk, 2 + lk}; weights are pre-packed
// [KH][KW][C/4][Cout][4] so that one 16-B load does the same for the other operand.
// A stage carries 384 MFMAs of 64 cycles per wave, so loads and LDS latency disappear behind the
// matrix pipe; the exact-f32 gather kernel (conv.hip) stays for strides, groups and odd channels.
//
// Reference semantics: cplx.convnd (cplxmodule/cplx.py:717-838) and its autograd backward.
#include <stdlib.h>

#include <type_traits>

#include "conv_nhwc.h"

namespace cplxamd {
namespace cn {

constexpr int BKF = 16;                               // float32 channels per stage (64-B rows)

// 4 consecutive channels of LDS row `row` (16-B chunk ch of 4), slots XOR-swizzled per row group
__device__ __forceinline__ float4 frag4(const char* plane, int row, int ch) {
  return *reinterpret_cast<const float4*>(plane + row * 64 + ((ch ^ ((row >> 2) & 3)) << 4));
}

template <typename TOUT, bool CPLX, bool CONJ>
__global__ __launch_bounds__(NT, 2) void conv_nhwc_f32_kernel(Args g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NP = CPLX ? 2 : 1;
  constexpr int NPC = NP * (BM + MAX_EXTRA) * 4 / NT + 1;       // LDS-DMA pieces per stage (fixed: 9 / 5)
  const int tiles_n = (g.Cout + BN - 1) / BN;
  const int bn = blockIdx.x % tiles_n;
  const int64_t m0 = (int64_t)(blockIdx.x / tiles_n) * BM;
  const int n0 = bn * BN;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid * 64, l31 = lane & 31, lk = lane >> 5;
  const int wave_chunk = wid * 64;
  const int cpt = g.C / BKF;                      // channel chunks per kernel row
  const int nk = g.KH * cpt;                      // stages: (kh, channel chunk)
  const int nA = g.srows * 4;                     // 16-B chunks per plane
  const int plane_bytes = nA * 16;
  const int stage_bytes = g.npieces * NT * 16;
  char* const dump = smem + 2 * stage_bytes;
  const float* xr = (const float*)g.x_r;
  const float* xi = (const float*)g.x_i;
  const float* wr = (const float*)g.w_r;
  const float* wi = (const float*)g.w_i;

  f32x16 acc_r[2][2], acc_i[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      acc_r[i][j] = f32x16{0};
      acc_i[i][j] = f32x16{0};
    }

  auto stage_q = [&](int buf, int kt, int q) {
    kt = kt < nk ? kt : nk - 1;
    const int kh = kt / cpt, c0 = (kt - kh * cpt) * BKF;
    int tid_ = tid;
    asm volatile("" : "+v"(tid_));                // recompute the offsets, do not hoist 18 registers
    int c = q * NT + tid_;
    c = c < NP * nA ? c : 0;
    const int plane = c >= nA;
    c -= plane * nA;
    const int row = c >> 2;
    int64_t grow = m0 + row + g.row_bias + (int64_t)kh * g.dil_h * g.Wp;
    grow = grow < g.rows ? grow : g.rows - 1;
    const float* base = plane ? xi : xr;
    lds_dma16(base + grow * g.C + c0 + (((c & 3) ^ ((row >> 2) & 3)) << 2),
              q < g.npieces ? smem + buf * stage_bytes + (q * NT + wave_chunk) * 16 : dump + wave_chunk * 16);
  };

  // weight chunk (4 channels) of (stage kt, tap kw, chunk ch) for this lane's two output channels
  int nrow[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + j * 32 + l31;
    nrow[j] = n < g.Cout ? n : g.Cout - 1;
  }
  const int c4 = g.C >> 2;
  auto load_b = [&](int kt, int kw, int ch, float4 (&br)[2], float4 (&bi)[2]) {
    kt = kt < nk ? kt : nk - 1;
    const int kh = kt / cpt, cc = (kt - kh * cpt) * 4 + ch;
    const int64_t blk = ((int64_t)(kh * g.KW + kw) * c4 + cc) * g.Cout;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int64_t o = (blk + nrow[j]) * 4;
      br[j] = *reinterpret_cast<const float4*>(wr + o);
      if (CPLX) bi[j] = *reinterpret_cast<const float4*>(wi + o);
    }
  };

  float4 br[2][2], bi[2][2];                      // [ping-pong set][j]
  load_b(0, 0, 0, br[0], bi[0]);
#pragma unroll
  for (int q = 0; q < NPC; ++q) stage_q(0, 0, q);

  // one chunk (4 channels = 2 MFMA k-steps) of tap kw of stage t
  auto chunk = [&](auto stage_tag, int t, int kw, int ch, const char* sA, const char* sAi) {
    constexpr bool STAGE = decltype(stage_tag)::value;
    const int r0 = wm + l31 + kw * g.dil_w;
    const int set = ch & 1;
    // the next chunk's weights (next chunk, else next tap, else next stage) into the other set
    {
      const bool last_ch = ch == 3, last_kw = kw + 1 == g.KW;
      const int chn = last_ch ? 0 : ch + 1;
      const int kwn = last_ch ? (last_kw ? 0 : kw + 1) : kw;
      const int tn = (last_ch && last_kw) ? t + 1 : t;
      load_b(tn, kwn, chn, br[set ^ 1], bi[set ^ 1]);
    }
    float4 ar[2], ai[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      ar[i] = frag4(sA, r0 + i * 32, ch);
      if (CPLX) ai[i] = frag4(sAi, r0 + i * 32, ch);
    }
    int q = 0;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      float a_r[2], a_i[2], b_r[2], b_i[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a_r[i] = lk ? (s ? ar[i].w : ar[i].y) : (s ? ar[i].z : ar[i].x);
        a_i[i] = CPLX ? (lk ? (s ? ai[i].w : ai[i].y) : (s ? ai[i].z : ai[i].x)) : 0.f;
        b_r[i] = lk ? (s ? br[set][i].w : br[set][i].y) : (s ? br[set][i].z : br[set][i].x);
        b_i[i] = CPLX ? (lk ? (s ? bi[set][i].w : bi[set][i].y) : (s ? bi[set][i].z : bi[set][i].x)) : 0.f;
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc_r[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_r[i], b_r[j], acc_r[i][j], 0, 0, 0);
          if (CPLX) {
            // y = x w: re -= xi wi, im += xr wi ;  y = x conj(w): re += xi wi, im -= xr wi
            acc_i[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_i[i], b_r[j], acc_i[i][j], 0, 0, 0);
            if (CONJ) {
              acc_r[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_i[i], b_i[j], acc_r[i][j], 0, 0, 0);
              acc_i[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(-a_r[i], b_i[j], acc_i[i][j], 0, 0, 0);
            } else {
              acc_r[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(-a_i[i], b_i[j], acc_r[i][j], 0, 0, 0);
              acc_i[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_r[i], b_i[j], acc_i[i][j], 0, 0, 0);
            }
          }
          if (STAGE) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
              if (q < NPC) {
                __builtin_amdgcn_sched_barrier(0);
                stage_q((t + 1) & 1, t + 1, q);
                __builtin_amdgcn_sched_barrier(0);
                ++q;
              }
          }
        }
    }
  };

  for (int t = 0; t < nk; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // own pieces of stage t landed ...
    __builtin_amdgcn_s_barrier();                       // ... everyone's did; slot (t+1)&1 is free
    const char* sA = smem + (t & 1) * stage_bytes;
    const char* sAi = sA + plane_bytes;
    // all NPC pieces of stage t+1 go out during the first chunk (8 MFMA groups x 2)
    chunk(std::true_type{}, t, 0, 0, sA, sAi);
    chunk(std::false_type{}, t, 0, 1, sA, sAi);
    chunk(std::false_type{}, t, 0, 2, sA, sAi);
    chunk(std::false_type{}, t, 0, 3, sA, sAi);
    for (int kw = 1; kw < g.KW; ++kw) {
      chunk(std::false_type{}, t, kw, 0, sA, sAi);
      chunk(std::false_type{}, t, kw, 1, sA, sAi);
      chunk(std::false_type{}, t, kw, 2, sA, sAi);
      chunk(std::false_type{}, t, kw, 3, sA, sAi);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  store_tile<TOUT, CPLX>(g, smem, acc_r, acc_i, m0, n0);
}

// planar NCHW float32 -> zero-padded channels-last float32 [B][Hp][Wp][C] (see nhwc_pad_kernel)
constexpr int TPF = 64;
struct __attribute__((packed, aligned(4))) f4u_t { float v[4]; };

__global__ __launch_bounds__(256) void nhwc_pad_f32_kernel(const float* __restrict__ x,
                                                           float* __restrict__ out, int B, int C, int H,
                                                           int W, int ph, int pw, int Hp, int Wp) {
  __shared__ float tile[TPF][TPF + 1];
  const int wt = blockIdx.x, hp = blockIdx.y;
  const int ctiles = (C + TPF - 1) / TPF;
  const int b = blockIdx.z / ctiles, c0 = (blockIdx.z - b * ctiles) * TPF;
  const int h = hp - ph;
  const int tid = threadIdx.x;
  const bool row_in = h >= 0 && h < H;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int idx = r * 256 + tid;
    const int c = idx >> 4, w4 = (idx & 15) * 4;
    const int wsrc = wt * TPF + w4 - pw;
    f4u_t v{{0.f, 0.f, 0.f, 0.f}};
    if (row_in && c0 + c < C) {
      const float* src = x + (((int64_t)b * C + c0 + c) * H + h) * W + wsrc;
      if (wsrc >= 0 && wsrc + 3 < W) {
        v = *reinterpret_cast<const f4u_t*>(src);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (wsrc + e >= 0 && wsrc + e < W) v.v[e] = src[e];
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) tile[c][w4 + e] = v.v[e];
  }
  __syncthreads();
  // store: lane -> (pixel = idx / 16, 4 consecutive channels) = one 16-B chunk
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int idx = r * 256 + tid;
    const int w = idx >> 4, cc = (idx & 15) * 4;
    const int wp = wt * TPF + w;
    if (wp >= Wp || c0 + cc >= C) continue;
    *reinterpret_cast<float4*>(out + (((int64_t)b * Hp + hp) * Wp + wp) * C + c0 + cc) =
        make_float4(tile[cc][w], tile[cc + 1][w], tile[cc + 2][w], tile[cc + 3][w]);
  }
}

template <typename TOUT, bool CPLX>
static int launch_f32(const Args& g0, bool conj, hipStream_t st) {
  Args g = g0;
  static const int dbg = getenv("CPLXAMD_CONV_DBG") ? atoi(getenv("CPLXAMD_CONV_DBG")) : 0;
  g.dbg = dbg;
  g.npieces = ((CPLX ? 2 : 1) * g.srows * 4 + NT - 1) / NT;
  int smem = (2 * g.npieces + 1) * NT * 16;
  const int out_img = BN * OUT_LD * (int)sizeof(TOUT) * (sizeof(TOUT) == 2 && CPLX ? 2 : 1);
  smem = smem > out_img ? smem : out_img;
  const int64_t tiles = ((g.rows + BM - 1) / BM) * ((g.Cout + BN - 1) / BN);
  if (tiles > 0x7fffffff) return CPLXAMD_ESHAPE;
  auto go = [&](auto kern) -> int {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       80 * 1024);
    if (e != hipSuccess) return (int)e;
    kern<<<dim3((unsigned)tiles), NT, smem, st>>>(g);
    CPLXAMD_CHECK_LAUNCH();
    return 0;
  };
  if constexpr (CPLX) {
    if (conj) return go(conv_nhwc_f32_kernel<TOUT, true, true>);
  }
  return go(conv_nhwc_f32_kernel<TOUT, CPLX, false>);
}

}  // namespace cn
}  // namespace cplxamd

using namespace cplxamd;

extern "C" {

int cplxamd_nhwc_pad_f32(const void* x, void* out, int B, int C, int H, int W, int pad_h, int pad_w,
                         int Hp, int Wp, void* stream) {
  if (!x || !out || B < 0 || C <= 0 || H <= 0 || W <= 0 || pad_h < 0 || pad_w < 0 ||
      Hp < H + pad_h || Wp < W + pad_w)
    return CPLXAMD_EINVAL;
  if (C % 4) return CPLXAMD_ESHAPE;
  if ((reinterpret_cast<uintptr_t>(out) & 15) != 0) return CPLXAMD_EALIGN;
  if (B == 0) return 0;
  const int ctiles = (C + cn::TPF - 1) / cn::TPF;
  if ((int64_t)B * ctiles > 65535 || Hp > 65535) return CPLXAMD_ESHAPE;
  dim3 grid((Wp + cn::TPF - 1) / cn::TPF, Hp, B * ctiles);
  cn::nhwc_pad_f32_kernel<<<grid, 256, 0, (hipStream_t)stream>>>((const float*)x, (float*)out, B, C, H, W,
                                                               pad_h, pad_w, Hp, Wp);
  CPLXAMD_CHECK_LAUNCH();
  return 0;
}

int cplxamd_conv2d_nhwc_f32(const void* xp_r, const void* xp_i, const void* w_r, const void* w_i,
                            const float* bias_r, const float* bias_i, void* y_r, void* y_i, int B,
                            int Hp, int Wp, int C, int Cout, int KH, int KW, int dil_h, int dil_w,
                            int conj_w, int64_t row_bias, int oh, int ow, int Hout, int Wout,
                            void* stream) {
  if (!xp_r || !w_r || !y_r) return CPLXAMD_EINVAL;
  const bool cplx = xp_i != nullptr;
  if (cplx && (!w_i || !y_i)) return CPLXAMD_EINVAL;
  if (B < 0 || Hp <= 0 || Wp <= 0 || C <= 0 || Cout <= 0 || KH <= 0 || KW <= 0 || dil_h <= 0 ||
      dil_w <= 0)
    return CPLXAMD_EINVAL;
  if (C % cn::BKF || (KW - 1) * dil_w > cn::MAX_EXTRA) return CPLXAMD_ESHAPE;
  if (Hout <= 0 || Wout <= 0 || oh < 0 || ow < 0 || oh + Hout > Hp || ow + Wout > Wp || row_bias > 0)
    return CPLXAMD_EINVAL;
  const int64_t rows = (int64_t)B * Hp * Wp;
  if (rows >= ((int64_t)1 << 31) || (int64_t)Hp * Wp >= ((int64_t)1 << 31)) return CPLXAMD_ESHAPE;
  auto a16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  if (!a16(xp_r) || !a16(w_r) || (cplx && (!a16(xp_i) || !a16(w_i)))) return CPLXAMD_EALIGN;
  if (B == 0) return 0;
  cn::Args g{xp_r, xp_i, w_r, w_i, bias_r, bias_i, y_r, y_i, rows, B, Hp, Wp, C, Cout, KH, KW, dil_h, dil_w,
             Hout, Wout, row_bias, oh, ow, cn::BM + (KW - 1) * dil_w, 0, 0};
  hipStream_t st = (hipStream_t)stream;
  if (cplx) return cn::launch_f32<float, true>(g, conj_w != 0, st);
  return cn::launch_f32<float, false>(g, false, st);
}

}  // extern "C"
