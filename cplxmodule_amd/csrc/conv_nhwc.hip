// bf16 complex / real 2-d convolution (stride 1, groups 1) as ONE shifted-row MFMA GEMM over a
// zero-padded channels-last copy of the input.
//
// Reference semantics: cplx.convnd (cplxmodule/cplx.py:717-838, four real cross-correlations
// combined) and its autograd backward; the data gradient is the same kernel on the padded output
// gradient with the spatially flipped, channel-swapped, conjugated weight.
//
//   xp [B][Hp][Wp][C]            bf16, padded (nhwc_pad_kernel below), rows r = (b, hp, wp) flattened
//   wq [KH][KW][C/16][Cout][16]  bf16 (one MFMA B fragment = 16 contiguous bytes per lane)
//   y[b, co, ho, wo] = sum_{kh, kw, c} xp[r(b, ho, wo) + kh*dil_h*Wp + kw*dil_w][c] * w[co, c, kh, kw]
// The shift does not depend on the row, so a tap of the implicit GEMM is a plain dense A tile at a
// displaced base pointer: no gathers, no index tables.  One staged A tile serves all KW taps of a
// kernel row: a stage = (kernel row kh, 32 channels) holds rows [m0 + kh*dil_h*Wp, + 256 +
// (KW-1)*dil_w) of the input, moved by global_load_lds_dwordx4; tap kw reads its A fragments at LDS
// row offset kw*dil_w (64-B rows, so the offset keeps every ds_read_b128 aligned).  The weight
// fragments are tiny and identical for every wave: they come straight from global / L1 into
// registers, one K sub-step ahead.  All Hp*Wp positions of an image are computed; the (KH-1)*dil
// rows and (KW-1)*dil columns whose window leaves the image are dropped at the store (1.6 % extra
// MFMA work for 3x3 on 256x256).
//
// The K loop of a convolution is short (KH * C / 32 = 6 stages for 3x3 x 64 channels), so the
// per-tile prologue (first stage latency) and epilogue weigh as much as the main loop.  Hence:
//  * 256-row x 64-channel tiles, 4 waves x (64 x 64), 2-stage ring = 76 KiB LDS, so TWO workgroups
//    share a CU (128-row tiles with four workgroups per CU measured 6 % slower) and one's epilogue / prologue overlaps the other's MFMAs;
//  * the epilogue transposes the accumulators through LDS and writes each channel's 256 pixels as
//    one contiguous run (the direct MFMA C layout would scatter 16-B pieces over 32 channel planes:
//    measured 1.06 ms of a 3.06 ms kernel).
// Measured history (cfg3, B=64, forward): gather kernel (conv_bf16.hip) 3.82 ms; 512-row tiles with
// one stage per tap 3.35 ms; kernel-row stages 2.34 ms; with weights staged through LDS at BK = 16
// and a 3-stage ring 3.04 ms (32-B rows halve the LDS-DMA efficiency).  On the final structure
// (1.65 ms): 128 x 32 wave tiles (half the weight-fragment loads) 1.68 ms; A fragments prefetched one
// tap ahead 2.26 ms (32 spilled registers) -- neither is kept.  A single-stage variant with four
// workgroups per CU (latency covered by occupancy instead of the 2-stage ring) does not exist for the
// complex kernel: 128 accumulators + 126 other registers = 254 per lane, i.e. 2 waves per SIMD at most
// (forcing 128 registers spills 750+).
// Ablation of the final kernel (-DCONV_ABL=1 / 2 / 3, CPLXAMD_CONV_DBG=4; same box, cfg3 B=64 forward,
// profiles/r01_conv_ablation.md): full 1.675 ms; without the epilogue stores 1.357; without LDS-DMA after
// the prologue 1.397; without weight re-loads 1.506; with neither (MFMA + ds_read + barriers) 1.227, and
// 0.856 ms when the stores are dropped too -- the MFMA floor is 0.59 ms at the power-limited clock.
#include <stdlib.h>

#include <type_traits>

#include "conv_nhwc.h"

namespace cplxamd {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace cn {

constexpr int BK = 32;
#ifndef CONV_ABL
#define CONV_ABL 0      // ablation builds: bit 1 no LDS-DMA after the prologue, bit 2 no weight re-loads
#endif

__device__ __forceinline__ void glds16(const void* gsrc, char* lds_wave_base) {
  lds_dma16(gsrc, lds_wave_base);     // common.h: inline asm, invisible to the compiler's waitcnt pass
}

// 8 consecutive k of LDS row `row` (16-B chunk kc of 4); chunk slots XOR-swizzled per row group
__device__ __forceinline__ bf16x8 frag(const char* plane, int row, int kc) {
  return *reinterpret_cast<const bf16x8*>(plane + row * 64 + ((kc ^ ((row >> 2) & 3)) << 4));
}

__device__ __forceinline__ bf16x8 neg(bf16x8 v) {
  uint4 u = __builtin_bit_cast(uint4, v);
  u.x ^= 0x80008000u; u.y ^= 0x80008000u; u.z ^= 0x80008000u; u.w ^= 0x80008000u;
  return __builtin_bit_cast(bf16x8, u);
}


template <typename TOUT, bool CPLX, bool CONJ>
__global__ __launch_bounds__(NT, 2) void conv_nhwc_kernel(Args g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NP = CPLX ? 2 : 1;
  // every plane of a stage is PP whole LDS-DMA pieces (5 x 256 chunks = 320 rows >= BM + MAX_EXTRA), so a
  // piece never straddles planes: its source is a SCALAR base (plane, tile row, kernel row, channel
  // chunk) plus one per-lane offset that is the same for all pieces and all stages
  constexpr int PP = ((BM + MAX_EXTRA) * 4 + NT - 1) / NT;
  constexpr int NPC = NP * PP;                       // pieces per stage (10 / 5)
  constexpr int PLANE = PP * NT * 16;                // bytes per staged plane (20 KiB)
  const int tiles_n = (g.Cout + BN - 1) / BN;
  const int bn = blockIdx.x % tiles_n;
  const int64_t m0 = (int64_t)(blockIdx.x / tiles_n) * BM;
  const int n0 = bn * BN;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid * 64, l31 = lane & 31, lk = lane >> 5;
  const int wave_chunk = wid * 64;
  const int cpt = g.C / BK;                       // channel chunks per kernel row
  const int nk = g.KH * cpt;                      // stages: (kh, channel chunk)
  constexpr int plane_bytes = PLANE;
  constexpr int stage_bytes = NP * PLANE;         // [A_r | A_i]

  f32x16 acc_r[2][2], acc_i[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      acc_r[i][j] = f32x16{0};
      acc_i[i][j] = f32x16{0};
    }

  // LDS-DMA piece q = (plane, 64-row group j) of stage kt into ring slot buf.  Rows past the end of the
  // grid are NOT clamped: the caller allocates 320 + (KH-1)*dil_h*Wp + (KW-1)*dil_w readable rows
  // behind it (cplxamd.h); they only feed outputs that are dropped at the store.
  const uint32_t smem_off = lds_offset_of(smem);
  const uint32_t wave_lds = (uint32_t)__builtin_amdgcn_readfirstlane(tid >> 6) * 1024u;
  const uint32_t voff0 = (uint32_t)(((tid >> 2) * g.C + (((tid & 3) ^ ((tid >> 4) & 3)) << 3)) * 2);
  auto stage_q = [&](int buf, int kt, int q) {
    kt = kt < nk ? kt : nk - 1;
    const int kh = kt / cpt, c0 = (kt - kh * cpt) * BK;
    const int pl = q / PP, j = q - pl * PP;
    const int64_t row = m0 + j * 64 + g.row_bias + (int64_t)kh * g.dil_h * g.Wp;
    const bf16_t* base = (const bf16_t*)(pl ? g.x_i : g.x_r) + row * g.C + c0;
    lds_dma16_sv(base, voff0, smem_off + (uint32_t)(buf * stage_bytes + pl * PLANE + j * (NT * 16)) + wave_lds);
  };

  // weight fragments of (stage kt, tap kw, K sub-step ks) straight from global memory
  int nrow[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + j * 32 + l31;
    nrow[j] = n < g.Cout ? n : g.Cout - 1;
  }
  const int c16 = g.C >> 4;
  auto load_b = [&](int kt, int kw, int ks, bf16x8 (&br)[2], bf16x8 (&bi)[2]) {
    kt = kt < nk ? kt : nk - 1;                                  // the prefetch past the end is unused
    const int kh = kt / cpt, cc = (kt - kh * cpt) * 2 + ks;
    const int64_t blk = ((int64_t)(kh * g.KW + kw) * c16 + cc) * g.Cout;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int64_t o = (blk + nrow[j]) * 16 + lk * 8;
      br[j] = *reinterpret_cast<const bf16x8*>((const bf16_t*)g.w_r + o);
      if (CPLX) bi[j] = *reinterpret_cast<const bf16x8*>((const bf16_t*)g.w_i + o);
    }
  };

  // weight fragments [ks][j]: each K sub-step's set is re-loaded for the next tap as soon as its
  // MFMAs are issued (16 MFMAs of lead; these loads come from L1 / L2)
  bf16x8 br[2][2], bi[2][2];
  load_b(0, 0, 0, br[0], bi[0]);
  load_b(0, 0, 1, br[1], bi[1]);
#pragma unroll
  for (int q = 0; q < NPC; ++q) stage_q(0, 0, q);

  // one tap (32 MFMAs) of stage t; STAGE: the LDS-DMA pieces of stage t+1 go out between the MFMA
  // groups (a body without branches, so that the compiler can interleave reads, loads and MFMAs)
  auto tap = [&](auto stage_tag, int t, int kw, const char* sA, const char* sAi) {
    constexpr bool STAGE = decltype(stage_tag)::value;
    const int r0 = wm + l31 + kw * g.dil_w;
    const bool last_kw = kw + 1 == g.KW;
    const int tn = last_kw ? t + 1 : t, kwn = last_kw ? 0 : kw + 1;
    bf16x8 ar[2][2], ai[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        ar[ks][i] = frag(sA, r0 + i * 32, ks * 2 + lk);
        if (CPLX) ai[ks][i] = frag(sAi, r0 + i * 32, ks * 2 + lk);
      }
    if (STAGE && !(CONV_ABL & 1)) {
      // the whole next stage goes out NOW, ahead of this tap's weight re-loads: vmcnt counts in order,
      // so the first weight fragment consumed in the next tap forces everything older to have landed --
      // issued here the pieces get a full tap, interleaved further down they got half of one
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < NPC; ++q) stage_q((t + 1) & 1, t + 1, q);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 na[2];
      if (CPLX) {
        // y = x w: re -= xi wi, im += xr wi ;  y = x conj(w): re += xi wi, im -= xr wi
#pragma unroll
        for (int i = 0; i < 2; ++i) na[i] = neg(CONJ ? ar[ks][i] : ai[ks][i]);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const bf16x8 wr_ = br[ks][j], wi_ = bi[ks][j];
          acc_r[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[ks][i], wr_, acc_r[i][j], 0, 0, 0);
          if (CPLX) {
            acc_i[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ai[ks][i], wr_, acc_i[i][j], 0, 0, 0);
            if (CONJ) {
              acc_r[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ai[ks][i], wi_, acc_r[i][j], 0, 0, 0);
              acc_i[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(na[i], wi_, acc_i[i][j], 0, 0, 0);
            } else {
              acc_r[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(na[i], wi_, acc_r[i][j], 0, 0, 0);
              acc_i[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[ks][i], wi_, acc_i[i][j], 0, 0, 0);
            }
          }
        }
      __builtin_amdgcn_sched_barrier(0);
      if (!(CONV_ABL & 2)) load_b(tn, kwn, ks, br[ks], bi[ks]);           // this sub-step's set, for the next tap
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  for (int t = 0; t < nk; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // own pieces of stage t landed ...
    __builtin_amdgcn_s_barrier();                       // ... everyone's did; slot (t+1)&1 is free
    const char* sA = smem + (t & 1) * stage_bytes;
    const char* sAi = sA + plane_bytes;
    // (after the last stage this re-loads stage nk-1 into the free slot: harmless, no branch)
    tap(std::true_type{}, t, 0, sA, sAi);
    for (int kw = 1; kw < g.KW; ++kw) tap(std::false_type{}, t, kw, sA, sAi);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  store_tile<TOUT, CPLX>(g, smem, acc_r, acc_i, m0, n0);
}

// planar NCHW bf16 -> zero-padded channels-last bf16 [B][Hp][Wp][C]; one block moves a
// 64-channel x 64-pixel tile of one padded row through LDS.
constexpr int TP = 64;
struct __attribute__((packed, aligned(2))) bf4_t { bf16_t v[4]; };

// (ph, pw) = rows / columns of zeros before the image; the rest of the Hp x Wp grid after it is
// zero too.
__global__ __launch_bounds__(256) void nhwc_pad_kernel(const bf16_t* __restrict__ x,
                                                       bf16_t* __restrict__ out, int B, int C, int H,
                                                       int W, int ph, int pw, int Hp, int Wp) {
  __shared__ bf16_t tile[TP][TP + 4];
  const int wt = blockIdx.x, hp = blockIdx.y;
  const int ctiles = (C + TP - 1) / TP;
  const int b = blockIdx.z / ctiles, c0 = (blockIdx.z - b * ctiles) * TP;
  const int h = hp - ph;
  const int tid = threadIdx.x;
  const bool row_in = h >= 0 && h < H;
  // load: lane -> (channel = idx / 16, 4 consecutive pixels)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int idx = r * 256 + tid;
    const int c = idx >> 4, w4 = (idx & 15) * 4;
    const int wsrc = wt * TP + w4 - pw;
    bf4_t v{{0, 0, 0, 0}};
    if (row_in && c0 + c < C) {
      const bf16_t* src = x + (((int64_t)b * C + c0 + c) * H + h) * W + wsrc;
      if (wsrc >= 0 && wsrc + 3 < W) {
        v = *reinterpret_cast<const bf4_t*>(src);
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
  // store: lane -> (pixel = idx / 8, 8 consecutive channels) = one 16-B chunk
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int idx = r * 256 + tid;
    const int w = idx >> 3, cc = (idx & 7) * 8;
    const int wp = wt * TP + w;
    if (wp >= Wp || c0 + cc >= C) continue;
    uint32_t pk[4];
#pragma unroll
    for (int e = 0; e < 4; ++e)
      pk[e] = (uint32_t)tile[cc + 2 * e][w] | ((uint32_t)tile[cc + 2 * e + 1][w] << 16);
    *reinterpret_cast<uint4*>(out + (((int64_t)b * Hp + hp) * Wp + wp) * C + c0 + cc) =
        uint4{pk[0], pk[1], pk[2], pk[3]};
  }
}

template <typename TOUT, bool CPLX>
static int launch_conj(const Args& g0, bool conj, hipStream_t st) {
  Args g = g0;
  static const int dbg = getenv("CPLXAMD_CONV_DBG") ? atoi(getenv("CPLXAMD_CONV_DBG")) : 0;
  g.dbg = dbg;
  g.npieces = 0;   // (field of the float32 kernel; the bf16 stage is 5 pieces per plane, always)
  int smem = 2 * (CPLX ? 2 : 1) * 5 * NT * 16;
  const int out_img = (sizeof(TOUT) == 2 ? (CPLX ? 2 : 1) : 1) * BN * OUT_LD * (int)sizeof(TOUT);
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
    if (conj) return go(conv_nhwc_kernel<TOUT, true, true>);
  }
  return go(conv_nhwc_kernel<TOUT, CPLX, false>);
}

}  // namespace cn
}  // namespace cplxamd

using namespace cplxamd;

extern "C" {

int cplxamd_nhwc_pad(const void* x, void* out, int B, int C, int H, int W, int pad_h, int pad_w,
                     int Hp, int Wp, void* stream) {
  if (!x || !out || B < 0 || C <= 0 || H <= 0 || W <= 0 || pad_h < 0 || pad_w < 0 ||
      Hp < H + pad_h || Wp < W + pad_w)
    return CPLXAMD_EINVAL;
  if (C % 8) return CPLXAMD_ESHAPE;
  if ((reinterpret_cast<uintptr_t>(out) & 15) != 0) return CPLXAMD_EALIGN;
  if (B == 0) return 0;
  const int ctiles = (C + cn::TP - 1) / cn::TP;
  if ((int64_t)B * ctiles > 65535 || Hp > 65535) return CPLXAMD_ESHAPE;
  dim3 grid((Wp + cn::TP - 1) / cn::TP, Hp, B * ctiles);
  cn::nhwc_pad_kernel<<<grid, 256, 0, (hipStream_t)stream>>>((const bf16_t*)x, (bf16_t*)out, B, C, H,
                                                           W, pad_h, pad_w, Hp, Wp);
  CPLXAMD_CHECK_LAUNCH();
  return 0;
}

int cplxamd_conv2d_nhwc(const void* xp_r, const void* xp_i, const void* w_r, const void* w_i,
                        const float* bias_r, const float* bias_i, void* y_r, void* y_i, int B,
                        int Hp, int Wp, int C, int Cout, int KH, int KW, int dil_h, int dil_w,
                        int conj_w, int64_t row_bias, int oh, int ow, int Hout, int Wout,
                        int out_dtype, void* stream) {
  if (!xp_r || !w_r || !y_r) return CPLXAMD_EINVAL;
  const bool cplx = xp_i != nullptr;
  if (cplx && (!w_i || !y_i)) return CPLXAMD_EINVAL;
  if (B < 0 || Hp <= 0 || Wp <= 0 || C <= 0 || Cout <= 0 || KH <= 0 || KW <= 0 || dil_h <= 0 ||
      dil_w <= 0)
    return CPLXAMD_EINVAL;
  if (out_dtype != CPLXAMD_BF16 && out_dtype != CPLXAMD_F32) return CPLXAMD_EINVAL;
  if (C % 32 || (KW - 1) * dil_w > cn::MAX_EXTRA) return CPLXAMD_ESHAPE;
  if (Hout <= 0 || Wout <= 0 || oh < 0 || ow < 0 || oh + Hout > Hp || ow + Wout > Wp || row_bias > 0)
    return CPLXAMD_EINVAL;
  const int Ho = Hout, Wo = Wout;
  const int64_t rows = (int64_t)B * Hp * Wp;
  if (rows >= ((int64_t)1 << 31) || (int64_t)Hp * Wp >= ((int64_t)1 << 31)) return CPLXAMD_ESHAPE;
  auto a16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  if (!a16(xp_r) || !a16(w_r) || (cplx && (!a16(xp_i) || !a16(w_i)))) return CPLXAMD_EALIGN;
  if (B == 0) return 0;
  cn::Args g{xp_r, xp_i, w_r, w_i,
             bias_r, bias_i, y_r, y_i, rows, B, Hp, Wp, C, Cout, KH, KW, dil_h, dil_w, Ho, Wo,
             row_bias, oh, ow, cn::BM + (KW - 1) * dil_w, 0, 0};
  hipStream_t st = (hipStream_t)stream;
  const bool f32 = out_dtype == CPLXAMD_F32;
  if (cplx)
    return f32 ? cn::launch_conj<float, true>(g, conj_w != 0, st)
               : cn::launch_conj<bf16_t, true>(g, conj_w != 0, st);
  return f32 ? cn::launch_conj<float, false>(g, false, st) : cn::launch_conj<bf16_t, false>(g, false, st);
}

}  // extern "C"
