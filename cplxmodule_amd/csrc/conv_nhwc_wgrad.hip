// bf16 weight gradient of the stride-1 / groups-1 convolution on the channels-last copies the
// shifted-row kernels (conv_nhwc.hip) already use:
//
//   dW[co, ci, kh, kw] = sum_r  Gp[r][co] * conj(Xp[r + kh*dil_h*Wp + kw*dil_w][ci])
//
// over the rows r = (b, hp, wp) of the padded input grid; Gp is the output gradient laid on that
// grid (top-left aligned, zero where the window leaves the image: cplxamd_nhwc_pad with asymmetric
// padding), Xp the padded input.  Reference: autograd of cplx.convnd (cplxmodule/cplx.py:717-838);
// dW = G^H-correlation per SURVEY A.1.
//
// Per tap this is a (T, T) GEMM with M = Cout, N = Cin and K = rows: both operands are K-major as
// stored, so the fragments come from ds_read_b64_tr_b16 exactly as in the wgrad of the linear
// layer.  One workgroup = one kernel row kh x a range of rows (split-K); its KW waves each own one
// tap kw and a full 64 x 64 (x re / im) accumulator tile.  A stage holds 32 rows of Gp and the
// 32 + (KW-1)*dil_w matching rows of Xp; tap kw reads its X fragments at row offset kw*dil_w, so the
// KW taps share every staged byte.  3-stage LDS-DMA ring, counted vmcnt.  Partial tiles go to fp32
// slabs [split][kh][kw][plane][64][64]; a small kernel sums them into dW[Cout][Cin][KH][KW].
#include <stdlib.h>

#include "common.h"

namespace cplxamd {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace cw {

constexpr int KR = 32;            // rows (GEMM k) per stage
constexpr int TC = 64;            // channels per tile side
constexpr int STAGES = 3;
constexpr int MAXP = 8;           // LDS-DMA pieces per thread per stage at most
#ifndef CW_ABL
#define CW_ABL 0                  // ablation builds (bf16 kernel): bit 1 no LDS-DMA after the prologue, 2 no MFMA
#endif

struct Args {
  const bf16_t* g_r; const bf16_t* g_i;     // Gp [rows][Co]
  const bf16_t* x_r; const bf16_t* x_i;     // Xp [rows][Ci]
  float* ws;                                // [splits][KH][KW][planes][tiles_co][tiles_ci][64][64]
  int64_t rows;
  int Wp, Co, Ci, KH, KW, dil_h, dil_w;
  int splits, tiles_per_split;              // K tiles (of KR rows) per split
  int xrows;                                // KR + (KW-1)*dil_w
  int npieces;
  int dbg;
};

__device__ __forceinline__ void glds16(const void* gsrc, char* lds_wave_base) {
  lds_dma16(gsrc, lds_wave_base);     // common.h: inline asm, invisible to the compiler's waitcnt pass
}

// [k][64 channels] image, 128-B rows; the 64-B half is swapped on every other PAIR of k rows so
// that the 4 k rows one 16-lane group touches fall into distinct banks.
__device__ __forceinline__ int img_off(int k, int chunk) {           // chunk: 16-B index 0..7
  return k * 128 + ((chunk ^ (((k >> 1) & 1) << 2)) << 4);
}

// 8 consecutive k (starting at kb) of channel block rb..rb+15 as an MFMA fragment:
// two hardware-transposed 4 x 16 reads (lane m of a 16-lane group: see gemm_bf16_impl.h)
__device__ __forceinline__ bf16x8 frag_t(const char* img, int rb, int kb, int m) {
  const int r = rb + 4 * (m & 3);
  s16x4 v[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int k = kb + 4 * h + (m >> 2);
    v[h] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)(img + img_off(k, r >> 3) + (r & 7) * 2));
  }
  const s16x8 both = __builtin_shufflevector(v[0], v[1], 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, both);
}

__device__ __forceinline__ bf16x8 neg(bf16x8 v) {
  uint4 u = __builtin_bit_cast(uint4, v);
  u.x ^= 0x80008000u; u.y ^= 0x80008000u; u.z ^= 0x80008000u; u.w ^= 0x80008000u;
  return __builtin_bit_cast(bf16x8, u);
}

__device__ __forceinline__ void wait_vmcnt_rt(int n) {   // wave-uniform n
  switch (n) {
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

// grid: x = split * KH + kh, y = co tile, z = ci tile; block = 64 * KW threads
template <bool CPLX>
__global__ __launch_bounds__(256, 2) void conv_wgrad_nhwc_kernel(Args g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NP = CPLX ? 2 : 1;
  const int NT = blockDim.x;
  const int split = blockIdx.x / g.KH, kh = blockIdx.x - split * g.KH;
  const int co0 = blockIdx.y * TC, ci0 = blockIdx.z * TC;
  const int tid = threadIdx.x, lane = tid & 63, kw = tid >> 6;
  const int l31 = lane & 31, lk = lane >> 5, l15 = lane & 15, lg = (lane >> 4) & 1;
  const int wave_chunk = kw * 64;

  // stage image: [G_r | G_i | X_r | X_i] as a flat list of 16-B chunks (8 per row), whole pieces
  const int nG = KR * 8, nX = g.xrows * 8;
  const int stage_bytes = g.npieces * NT * 16;
  const int64_t t_begin = (int64_t)split * g.tiles_per_split;
  const int64_t t_all = (g.rows + KR - 1) / KR;
  int64_t t_end = t_begin + g.tiles_per_split;
  t_end = t_end < t_all ? t_end : t_all;
  const int nt = (int)(t_end - t_begin);
  const int64_t xshift = (int64_t)kh * g.dil_h * g.Wp;

  f32x16 acc_r[2][2], acc_i[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      acc_r[i][j] = f32x16{0};
      acc_i[i][j] = f32x16{0};
    }

  // LDS-DMA piece q of K tile t into ring slot buf.  Both channels-last buffers are allocated with
  // zero tail rows (Gp: up to a multiple of KR; Xp: KR + the largest tap shift), so no row needs a
  // clamp and the source address of piece q is a fixed per-thread pointer for tile t_begin plus
  // (t - t_begin) * step: 2 + 1 registers per piece instead of ~45 VALU ops per issue.
  const uint64_t g_base = (uint64_t)g.g_r, x_base = (uint64_t)g.x_r;
  const uint64_t g_delta = CPLX ? (uint64_t)g.g_i - g_base : 0, x_delta = CPLX ? (uint64_t)g.x_i - x_base : 0;
  uint64_t ptr0[MAXP];
  uint32_t xbits = 0;                    // bit q: piece q reads Xp (step KR*Ci) rather than Gp (KR*Co)
#pragma unroll
  for (int q = 0; q < MAXP; ++q) {
    int c = q * NT + tid;
    c = c < NP * (nG + nX) ? c : 0;
    const bool isx = c >= NP * nG;
    c -= isx ? NP * nG : 0;
    const int npl = isx ? nX : nG;
    const bool plane = c >= npl;
    c -= plane ? npl : 0;
    const int k = c >> 3, ch = (c & 7) ^ (((k >> 1) & 1) << 2);
    const int64_t row = t_begin * KR + k + (isx ? xshift : 0);
    const int C_ = isx ? g.Ci : g.Co;
    int col = (isx ? ci0 : co0) + ch * 8;
    col = col + 8 <= C_ ? col : C_ - 8;
    // mask arithmetic instead of ?: on these four (selects of captured variables become a load
    // through a selected ADDRESS, i.e. scratch traffic)
    const uint64_t mx = 0 - (uint64_t)isx, mp = 0 - (uint64_t)plane;
    const uint64_t base = ((x_base & mx) | (g_base & ~mx)) + (((x_delta & mx) | (g_delta & ~mx)) & mp);
    ptr0[q] = base + 2 * (uint64_t)(row * C_ + col);
    xbits |= (uint32_t)isx << q;
  }
  const uint32_t step_g = KR * 2 * g.Co, step_x = KR * 2 * g.Ci;
  // Every thread issues all MAXP pieces of every tile, tiles past the end re-load the last one, and
  // pieces past npieces land in a dump slot behind the ring: the K loop has no run-time branch (one
  // basic block), so the compiler's s_waitcnt placement is exact and the counted vmcnt is a constant.
  const uint32_t smem_off = lds_offset_of(smem);
  const uint32_t wave_lds = (uint32_t)__builtin_amdgcn_readfirstlane(kw) * 1024u;
  const uint32_t dump_off = (uint32_t)(STAGES * stage_bytes);
  auto stage_q = [&](int buf, int trel, int q) {     // trel = t - t_begin; q is a compile-time index
    trel = trel < nt ? trel : nt - 1;
    const uint32_t step = ((xbits >> q) & 1) ? step_x : step_g;
    const uint32_t dst = q < g.npieces ? (uint32_t)(buf * stage_bytes + q * NT * 16) : dump_off;
    lds_dma16_at(reinterpret_cast<const void*>(ptr0[q] + (uint64_t)trel * step), smem_off + dst + wave_lds);
  };
  auto stage_all = [&](int buf, int trel) {
#pragma unroll
    for (int q = 0; q < MAXP; ++q) stage_q(buf, trel, q);
  };

  auto compute = [&](int buf, int nbuf, int tnext) {
    const char* sGr = smem + buf * stage_bytes;
    const char* sGi = sGr + nG * 16;
    const char* sXr = sGr + NP * nG * 16;
    const char* sXi = sXr + nX * 16;
    const int xk = kw * g.dil_w;
    bf16x8 ar[2][2], br[2][2], ai[2][2], bi[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        ar[ks][i] = frag_t(sGr, i * 32 + 16 * lg, ks * 16 + 8 * lk, l15);
        br[ks][i] = frag_t(sXr, i * 32 + 16 * lg, xk + ks * 16 + 8 * lk, l15);
        if (CPLX) {
          ai[ks][i] = frag_t(sGi, i * 32 + 16 * lg, ks * 16 + 8 * lk, l15);
          bi[ks][i] = frag_t(sXi, i * 32 + 16 * lg, xk + ks * 16 + 8 * lk, l15);
        }
      }
    int q = 0;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 nar[2];
      if (CPLX) {
#pragma unroll
        for (int i = 0; i < 2; ++i) nar[i] = neg(ar[ks][i]);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          // X fragment first: the accumulator holds the tile transposed, row (co) = lane & 31,
          // 4 consecutive ci per register group.  G conj(X): re = gr xr + gi xi, im = gi xr - gr xi
          if (!(CW_ABL & 2))
          acc_r[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(br[ks][j], ar[ks][i], acc_r[i][j], 0, 0, 0);
          if (CPLX && !(CW_ABL & 2)) {
            acc_i[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(br[ks][j], ai[ks][i], acc_i[i][j], 0, 0, 0);
            acc_r[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bi[ks][j], ai[ks][i], acc_r[i][j], 0, 0, 0);
            acc_i[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bi[ks][j], nar[i], acc_i[i][j], 0, 0, 0);
          }
          if (q < MAXP) {
            __builtin_amdgcn_sched_barrier(0);
            if (!(CW_ABL & 1)) stage_q(nbuf, tnext, q);
            __builtin_amdgcn_sched_barrier(0);
            ++q;
          }
        }
    }
  };

  if (nt > 0) {
    stage_all(0, 0);
    stage_all(1, 1);
    int cur = 0;
    for (int t = 0; t < nt; ++t) {
      if (CW_ABL & 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MAXP) : "memory");   // tile t landed; tile t+1 may be in flight
      __builtin_amdgcn_s_barrier();
      int nxt = cur + 2; nxt = nxt >= STAGES ? nxt - STAGES : nxt;
      compute(cur, nxt, t + 2);
      cur = cur + 1 == STAGES ? 0 : cur + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the clamped re-loads must land before the LDS is released
  }

  // slab [split][kh][kw][plane][tile_co][tile_ci][64][64]
  const int64_t tile = ((((int64_t)(split * g.KH + kh) * g.KW + kw) * NP) * gridDim.y + blockIdx.y) *
                       gridDim.z + blockIdx.z;
  float* out_r = g.ws + tile * (TC * TC);
  float* out_i = out_r + (int64_t)gridDim.y * gridDim.z * (TC * TC);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int o = (i * 32 + l31) * TC + j * 32 + 8 * q + 4 * lk;
        f4 vr, vi;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          vr.v[e] = acc_r[i][j][4 * q + e];
          vi.v[e] = CPLX ? acc_i[i][j][4 * q + e] : 0.f;
        }
        st4(out_r + o, vr);
        if (CPLX) st4(out_i + o, vi);
      }
}

// dW[co][ci][kh][kw] (plane pl) = sum_split slab[split][kh][kw][pl][tco][tci][co % 64][ci % 64] (* emul)
// one thread per slab element (coalesced reads of every split), scattered 4-B write into dW
__global__ __launch_bounds__(64) void wgrad_slab_reduce_kernel(const float* ws, int splits, int KH,
                                                                int KW, int planes, int pl, int Co,
                                                                int Ci, int tco, int tci,
                                                                const float* emul, float* dw) {
  const int64_t per_plane = (int64_t)tco * tci * (TC * TC);
  const int64_t n = (int64_t)KH * KW * per_plane;
  const int64_t j = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (j >= n) return;
  const int tap = (int)(j / per_plane);
  int64_t r = j - (int64_t)tap * per_plane;
  const int t = (int)(r / (TC * TC));
  r -= (int64_t)t * (TC * TC);
  const int co = (t / tci) * TC + (int)(r / TC), ci = (t % tci) * TC + (int)(r % TC);
  if (co >= Co || ci >= Ci) return;
  const int64_t per_split = (int64_t)KH * KW * planes * per_plane;
  const int64_t o = ((int64_t)tap * planes + pl) * per_plane + (int64_t)t * (TC * TC) + r;
  // fixed summation order (deterministic), 8 loads in flight per thread
  float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int s = 0;
  for (; s + 8 <= splits; s += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) a8[u] += ws[(int64_t)(s + u) * per_split + o];
  }
  for (; s < splits; ++s) a8[0] += ws[(int64_t)s * per_split + o];
  const float acc = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
  const int64_t i = ((int64_t)co * Ci + ci) * KH * KW + tap;        // tap = kh * KW + kw
  dw[i] = emul ? acc * emul[i] : acc;
}

static int plan_splits(int64_t rows, int KH, int tiles) {
  const int64_t t_all = (rows + KR - 1) / KR;
  int64_t s = 1024 / ((int64_t)KH * tiles);         // two rounds of 2 workgroups on each of 256 CUs
  if (s < 1) s = 1;
  const int64_t maxs = (t_all + 31) / 32;           // >= 32 K tiles per split
  if (s > maxs) s = maxs;
  if (s < 1) s = 1;
  return (int)s;
}


// ---- exact-float32 variant (v_mfma_f32_32x32x2_f32) -----------------------------------------
// Same work split (one kernel row per workgroup, one tap per wave, split-K over the rows), but a
// stage holds KRF = 16 rows of float32 Gp / Xp as [k][64 channels] (256-B rows): the k = 2 MFMA
// takes one float per lane and operand, read with ds_read_b32 -- 32 consecutive channels per half
// wave, conflict-free without a swizzle.  128 MFMAs of 64 cycles per stage and wave, 2-stage ring.
constexpr int KRF = 16;

template <bool CPLX>
__global__ __launch_bounds__(256, 2) void conv_wgrad_nhwc_f32_kernel(Args g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NP = CPLX ? 2 : 1;
  const int NT = blockDim.x;
  const int split = blockIdx.x / g.KH, kh = blockIdx.x - split * g.KH;
  const int co0 = blockIdx.y * TC, ci0 = blockIdx.z * TC;
  const int tid = threadIdx.x, lane = tid & 63, kw = tid >> 6;
  const int l31 = lane & 31, lk = lane >> 5;
  const int wave_chunk = kw * 64;

  // stage image: [G_r | G_i | X_r | X_i], 16 chunks of 16 B per row, whole pieces
  const int nG = KRF * 16, nX = g.xrows * 16;
  const int stage_bytes = g.npieces * NT * 16;
  const int64_t t_begin = (int64_t)split * g.tiles_per_split;
  const int64_t t_all = (g.rows + KRF - 1) / KRF;
  int64_t t_end = t_begin + g.tiles_per_split;
  t_end = t_end < t_all ? t_end : t_all;
  const int nt = (int)(t_end - t_begin);
  const int64_t xshift = (int64_t)kh * g.dil_h * g.Wp;

  f32x16 acc_r[2][2], acc_i[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      acc_r[i][j] = f32x16{0};
      acc_i[i][j] = f32x16{0};
    }

  // per-thread source pointers for tile t_begin (zero tail rows: no row is ever clamped)
  const uint64_t g_base = (uint64_t)g.g_r, x_base = (uint64_t)g.x_r;
  const uint64_t g_delta = CPLX ? (uint64_t)g.g_i - g_base : 0, x_delta = CPLX ? (uint64_t)g.x_i - x_base : 0;
  uint64_t ptr0[MAXP];
  uint32_t xbits = 0;
#pragma unroll
  for (int q = 0; q < MAXP; ++q) {
    int c = q * NT + tid;
    c = c < NP * (nG + nX) ? c : 0;
    const bool isx = c >= NP * nG;
    c -= isx ? NP * nG : 0;
    const int npl = isx ? nX : nG;
    const bool plane = c >= npl;
    c -= plane ? npl : 0;
    const int k = c >> 4, ch = c & 15;
    const int64_t row = t_begin * KRF + k + (isx ? xshift : 0);
    const int C_ = isx ? g.Ci : g.Co;
    int col = (isx ? ci0 : co0) + ch * 4;
    col = col + 4 <= C_ ? col : C_ - 4;
    const uint64_t mx = 0 - (uint64_t)isx, mp = 0 - (uint64_t)plane;
    const uint64_t base = ((x_base & mx) | (g_base & ~mx)) + (((x_delta & mx) | (g_delta & ~mx)) & mp);
    ptr0[q] = base + 4 * (uint64_t)(row * C_ + col);
    xbits |= (uint32_t)isx << q;
  }
  const uint32_t step_g = KRF * 4 * g.Co, step_x = KRF * 4 * g.Ci;
  // branch-free staging as in the bf16 kernel: all MAXP pieces always, tiles past the end re-load the
  // last one, pieces past npieces go to a dump slot behind the two stages
  const uint32_t smem_off = lds_offset_of(smem);
  const uint32_t wave_lds = (uint32_t)__builtin_amdgcn_readfirstlane(kw) * 1024u;
  const uint32_t dump_off = (uint32_t)(2 * stage_bytes);
  auto stage_q = [&](int buf, int trel, int q) {
    trel = trel < nt ? trel : nt - 1;
    const uint32_t step = ((xbits >> q) & 1) ? step_x : step_g;
    const uint32_t dst = q < g.npieces ? (uint32_t)(buf * stage_bytes + q * NT * 16) : dump_off;
    lds_dma16_at(reinterpret_cast<const void*>(ptr0[q] + (uint64_t)trel * step), smem_off + dst + wave_lds);
  };
  auto stage_all = [&](int buf, int trel) {
#pragma unroll
    for (int q = 0; q < MAXP; ++q) stage_q(buf, trel, q);
  };

  auto compute = [&](int buf, int nbuf, int tnext) {
    const float* sGr = reinterpret_cast<const float*>(smem + buf * stage_bytes);
    const float* sGi = sGr + nG * 4;
    const float* sXr = sGr + NP * nG * 4;
    const float* sXi = sXr + nX * 4;
    const int xk = kw * g.dil_w;
    int q = 0;
#pragma unroll
    for (int s = 0; s < KRF / 2; ++s) {
      const int k = 2 * s + lk;
      float ar[2], ai[2], br[2], bi[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        ar[i] = sGr[k * TC + i * 32 + l31];
        br[i] = sXr[(xk + k) * TC + i * 32 + l31];
        if (CPLX) {
          ai[i] = sGi[k * TC + i * 32 + l31];
          bi[i] = sXi[(xk + k) * TC + i * 32 + l31];
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          // X first: the accumulator holds the tile transposed, row (co) = lane & 31, 4 consecutive ci
          // per register group.  G conj(X): re = gr xr + gi xi, im = gi xr - gr xi
          acc_r[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(br[j], ar[i], acc_r[i][j], 0, 0, 0);
          if (CPLX) {
            acc_i[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(br[j], ai[i], acc_i[i][j], 0, 0, 0);
            acc_r[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bi[j], ai[i], acc_r[i][j], 0, 0, 0);
            acc_i[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bi[j], -ar[i], acc_i[i][j], 0, 0, 0);
          }
        }
      if (q < MAXP) {
        __builtin_amdgcn_sched_barrier(0);
        stage_q(nbuf, tnext, q);
        __builtin_amdgcn_sched_barrier(0);
        ++q;
      }
    }
  };

  if (nt > 0) {
    stage_all(0, 0);
    for (int t = 0; t < nt; ++t) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      compute(t & 1, (t + 1) & 1, t + 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the clamped re-load of the last tile
  }

  const int64_t tile = ((((int64_t)(split * g.KH + kh) * g.KW + kw) * NP) * gridDim.y + blockIdx.y) *
                       gridDim.z + blockIdx.z;
  float* out_r = g.ws + tile * (TC * TC);
  float* out_i = out_r + (int64_t)gridDim.y * gridDim.z * (TC * TC);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int o = (i * 32 + l31) * TC + j * 32 + 8 * q + 4 * lk;
        f4 vr, vi;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          vr.v[e] = acc_r[i][j][4 * q + e];
          vi.v[e] = CPLX ? acc_i[i][j][4 * q + e] : 0.f;
        }
        st4(out_r + o, vr);
        if (CPLX) st4(out_i + o, vi);
      }
}

static int plan_splits_f32(int64_t rows, int KH, int tiles) {
  const int64_t t_all = (rows + KRF - 1) / KRF;
  int64_t s = 1024 / ((int64_t)KH * tiles);
  if (s < 1) s = 1;
  const int64_t maxs = (t_all + 15) / 16;           // >= 16 K tiles per split
  if (s > maxs) s = maxs;
  return (int)(s < 1 ? 1 : s);
}

}  // namespace cw
}  // namespace cplxamd

using namespace cplxamd;

extern "C" {

int64_t cplxamd_conv2d_nhwc_wgrad_ws_bytes(int B, int Hp, int Wp, int Ci, int Co, int KH, int KW,
                                           int cplx) {
  const int tco = (Co + cw::TC - 1) / cw::TC, tci = (Ci + cw::TC - 1) / cw::TC;
  const int s = cw::plan_splits((int64_t)B * Hp * Wp, KH, tco * tci);
  return (int64_t)s * KH * KW * (cplx ? 2 : 1) * tco * tci * cw::TC * cw::TC * (int64_t)sizeof(float);
}

int cplxamd_conv2d_nhwc_wgrad(const void* gp_r, const void* gp_i, const void* xp_r, const void* xp_i,
                              const float* emul, float* dw_r, float* dw_i, int B, int Hp, int Wp,
                              int Ci, int Co, int KH, int KW, int dil_h, int dil_w, void* ws,
                              int64_t ws_bytes, void* stream) {
  if (!gp_r || !xp_r || !dw_r || !ws) return CPLXAMD_EINVAL;
  const bool cplx = gp_i != nullptr;
  if (cplx && (!xp_i || !dw_i)) return CPLXAMD_EINVAL;
  if (B < 0 || Hp <= 0 || Wp <= 0 || Ci <= 0 || Co <= 0 || KH <= 0 || KW <= 0 || dil_h <= 0 ||
      dil_w <= 0)
    return CPLXAMD_EINVAL;
  if (Ci % 8 || Co % 8 || KW > 4 || (KW - 1) * dil_w > 32) return CPLXAMD_ESHAPE;
  const int64_t rows = (int64_t)B * Hp * Wp;
  if ((rows + 64 + (int64_t)KH * dil_h * Wp) * (Ci > Co ? Ci : Co) >= ((int64_t)1 << 31)) return CPLXAMD_ESHAPE;
  auto a16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  if (!a16(gp_r) || !a16(xp_r) || !a16(ws) || (cplx && (!a16(gp_i) || !a16(xp_i))))
    return CPLXAMD_EALIGN;
  if (ws_bytes < cplxamd_conv2d_nhwc_wgrad_ws_bytes(B, Hp, Wp, Ci, Co, KH, KW, cplx)) return CPLXAMD_EWS;
  hipStream_t st = (hipStream_t)stream;
  const int tco = (Co + cw::TC - 1) / cw::TC, tci = (Ci + cw::TC - 1) / cw::TC;
  if (tco > 65535 || tci > 65535) return CPLXAMD_ESHAPE;
  const int NP = cplx ? 2 : 1;
  const int64_t n = (int64_t)Co * Ci * KH * KW;
  if (B == 0) {
    hipError_t e = hipMemsetAsync(dw_r, 0, n * sizeof(float), st);
    if (e == hipSuccess && cplx) e = hipMemsetAsync(dw_i, 0, n * sizeof(float), st);
    return (int)e;
  }
  cw::Args g{(const bf16_t*)gp_r, (const bf16_t*)gp_i, (const bf16_t*)xp_r, (const bf16_t*)xp_i,
             (float*)ws, rows, Wp, Co, Ci, KH, KW, dil_h, dil_w};
  g.splits = cw::plan_splits(rows, KH, tco * tci);
  g.dbg = getenv("CPLXAMD_CONV_DBG") ? atoi(getenv("CPLXAMD_CONV_DBG")) : 0;
  const int64_t t_all = (rows + cw::KR - 1) / cw::KR;
  g.tiles_per_split = (int)((t_all + g.splits - 1) / g.splits);
  g.xrows = cw::KR + (KW - 1) * dil_w;
  const int NT = 64 * KW;
  g.npieces = (NP * (cw::KR * 8 + g.xrows * 8) + NT - 1) / NT;
  if (g.npieces > cw::MAXP) return CPLXAMD_ESHAPE;
  const int smem = (cw::STAGES * g.npieces + 1) * NT * 16;   // ring + one piece of dump
  dim3 grid(g.splits * KH, tco, tci);
  if (cplx)
    cw::conv_wgrad_nhwc_kernel<true><<<grid, NT, smem, st>>>(g);
  else
    cw::conv_wgrad_nhwc_kernel<false><<<grid, NT, smem, st>>>(g);
  CPLXAMD_CHECK_LAUNCH();
  const int rgrid = (int)(((int64_t)KH * KW * tco * tci * cw::TC * cw::TC + 63) / 64);
  cw::wgrad_slab_reduce_kernel<<<rgrid, 64, 0, st>>>((const float*)ws, g.splits, KH, KW, NP, 0, Co, Ci,
                                                     tco, tci, emul, dw_r);
  CPLXAMD_CHECK_LAUNCH();
  if (cplx) {
    cw::wgrad_slab_reduce_kernel<<<rgrid, 64, 0, st>>>((const float*)ws, g.splits, KH, KW, NP, 1, Co,
                                                       Ci, tco, tci, nullptr, dw_i);
    CPLXAMD_CHECK_LAUNCH();
  }
  return 0;
}


int64_t cplxamd_conv2d_nhwc_wgrad_f32_ws_bytes(int B, int Hp, int Wp, int Ci, int Co, int KH, int KW,
                                               int cplx) {
  const int tco = (Co + cw::TC - 1) / cw::TC, tci = (Ci + cw::TC - 1) / cw::TC;
  const int s = cw::plan_splits_f32((int64_t)B * Hp * Wp, KH, tco * tci);
  return (int64_t)s * KH * KW * (cplx ? 2 : 1) * tco * tci * cw::TC * cw::TC * (int64_t)sizeof(float);
}

int cplxamd_conv2d_nhwc_wgrad_f32(const void* gp_r, const void* gp_i, const void* xp_r, const void* xp_i,
                                  const float* emul, float* dw_r, float* dw_i, int B, int Hp, int Wp,
                                  int Ci, int Co, int KH, int KW, int dil_h, int dil_w, void* ws,
                                  int64_t ws_bytes, void* stream) {
  if (!gp_r || !xp_r || !dw_r || !ws) return CPLXAMD_EINVAL;
  const bool cplx = gp_i != nullptr;
  if (cplx && (!xp_i || !dw_i)) return CPLXAMD_EINVAL;
  if (B < 0 || Hp <= 0 || Wp <= 0 || Ci <= 0 || Co <= 0 || KH <= 0 || KW <= 0 || dil_h <= 0 ||
      dil_w <= 0)
    return CPLXAMD_EINVAL;
  if (Ci % 4 || Co % 4 || KW > 4 || (KW - 1) * dil_w > 32) return CPLXAMD_ESHAPE;
  const int64_t rows = (int64_t)B * Hp * Wp;
  if ((rows + 64 + (int64_t)KH * dil_h * Wp) * (Ci > Co ? Ci : Co) >= ((int64_t)1 << 31)) return CPLXAMD_ESHAPE;
  auto a16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  if (!a16(gp_r) || !a16(xp_r) || !a16(ws) || (cplx && (!a16(gp_i) || !a16(xp_i)))) return CPLXAMD_EALIGN;
  if (ws_bytes < cplxamd_conv2d_nhwc_wgrad_f32_ws_bytes(B, Hp, Wp, Ci, Co, KH, KW, cplx)) return CPLXAMD_EWS;
  hipStream_t st = (hipStream_t)stream;
  const int tco = (Co + cw::TC - 1) / cw::TC, tci = (Ci + cw::TC - 1) / cw::TC;
  if (tco > 65535 || tci > 65535) return CPLXAMD_ESHAPE;
  const int NP = cplx ? 2 : 1;
  const int64_t n = (int64_t)Co * Ci * KH * KW;
  if (B == 0) {
    hipError_t e = hipMemsetAsync(dw_r, 0, n * sizeof(float), st);
    if (e == hipSuccess && cplx) e = hipMemsetAsync(dw_i, 0, n * sizeof(float), st);
    return (int)e;
  }
  cw::Args g{(const bf16_t*)gp_r, (const bf16_t*)gp_i, (const bf16_t*)xp_r, (const bf16_t*)xp_i,
             (float*)ws, rows, Wp, Co, Ci, KH, KW, dil_h, dil_w};
  g.splits = cw::plan_splits_f32(rows, KH, tco * tci);
  const int64_t t_all = (rows + cw::KRF - 1) / cw::KRF;
  g.tiles_per_split = (int)((t_all + g.splits - 1) / g.splits);
  g.xrows = cw::KRF + (KW - 1) * dil_w;
  const int NT = 64 * KW;
  g.npieces = (NP * (cw::KRF * 16 + g.xrows * 16) + NT - 1) / NT;
  if (g.npieces > cw::MAXP) return CPLXAMD_ESHAPE;
  g.dbg = 0;
  const int smem = (2 * g.npieces + 1) * NT * 16;   // two stages + one piece of dump
  dim3 grid(g.splits * KH, tco, tci);
  if (cplx)
    cw::conv_wgrad_nhwc_f32_kernel<true><<<grid, NT, smem, st>>>(g);
  else
    cw::conv_wgrad_nhwc_f32_kernel<false><<<grid, NT, smem, st>>>(g);
  CPLXAMD_CHECK_LAUNCH();
  const int rgrid = (int)(((int64_t)KH * KW * tco * tci * cw::TC * cw::TC + 63) / 64);
  cw::wgrad_slab_reduce_kernel<<<rgrid, 64, 0, st>>>((const float*)ws, g.splits, KH, KW, NP, 0, Co, Ci,
                                                    tco, tci, emul, dw_r);
  CPLXAMD_CHECK_LAUNCH();
  if (cplx) {
    cw::wgrad_slab_reduce_kernel<<<rgrid, 64, 0, st>>>((const float*)ws, g.splits, KH, KW, NP, 1, Co, Ci,
                                                      tco, tci, nullptr, dw_i);
    CPLXAMD_CHECK_LAUNCH();
  }
  return 0;
}

}  // extern "C"
