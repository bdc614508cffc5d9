// REAL-valued twin of conv_cl_wgrad.hip: dW[co, ci, kh, kw] = sum_q G[q][co] X[q + shift][ci] on unpadded channels-last
// bf16 planes, float32 out, optionally times an elementwise multiplier in the slab reduce (the d log_sigma2 of the
// local-reparameterization layers: cplxmodule/nn/relevance/real/base.py:116-163).  Same work split (nine taps on eight
// waves at four and a half 32 x 32 blocks each), same borders by out-of-range LDS-DMA, one plane and one MFMA per block.
#include <stdlib.h>

#include "common.h"
#include "launch.h"   // per-call launch policy (CPLXAMD_LAUNCH_SHARED: the chip is shared with collectives)

namespace cplxamd {
namespace clwr {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int KR = 32, NT = 512, TC = 64;
constexpr int G_BYTES = 2 * KR * 128;              // [plane][32 pixels][64 co]
constexpr int XW_ROWS = 40, XW_BYTES = XW_ROWS * 128;   // one (kernel row, plane) window: 34 rows used
constexpr int X_BYTES = 4 * 8192;                  // 6 windows (30 KiB) + 2 KiB the idle lanes of the last piece zero
constexpr int STAGE = G_BYTES + X_BYTES;           // 40 KiB
constexpr int L = 3;                               // LDS-DMA pieces per wave and stage: G (half of it idle), X in 2
constexpr int NBLK = 40;                           // slab blocks per tile: 2 x 18 + the 4 second halves
constexpr uint32_t OOB = 0xFFFFFFF0u;

struct Args {
  const void* g_r;                                 // [P][Co] bf16
  const void* x_r;                                 // [P][Ci] bf16
  float* ws;
  int64_t P;
  uint32_t g_bytes, x_bytes;
  int H, W, Co, Ci, dil_h, dil_w, pad_h, pad_w;      // H x W: the input image = the grid the pixel loop walks
  int Ho, Wo;                                        // the output image (<= H x W, top-left aligned on the grid)
  int strips;                                        // stages per image row: ceil(W / 32)
  int nstages, per_split, splits, tiles_ci;
  int walk;                                          // stage order: 0 = along image rows, 1 = down image columns (conv_cl_wgrad.hip)
};

__device__ __forceinline__ void buf_lds16(i32x4 rsrc, uint32_t voff, uint32_t lds_off_uniform) {
#if defined(__HIP_DEVICE_COMPILE__)
  const uint32_t m0v = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_off_uniform);
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds"
               :
               : "v"(voff), "s"(rsrc), "s"(m0v)
               : "memory");
#endif
}

__device__ __forceinline__ i32x4 make_rsrc(const void* base, uint32_t bytes) {
  const uint64_t a = (uint64_t)(uintptr_t)base;
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32));
  const uint32_t nb = (uint32_t)__builtin_amdgcn_readfirstlane((int)bytes);
  return i32x4{(int)lo, (int)(hi & 0xffffu), (int)nb, 0x00020000};
}

// [k][64 channels] image, 128-B rows; the 64-B half is swapped on every other PAIR of k rows so that the 4 k rows one
// 16-lane group of a transposed read touches fall into distinct banks (same image as conv_nhwc_wgrad.hip)
__device__ __forceinline__ int img_off(int k, int chunk) { return k * 128 + ((chunk ^ (((k >> 1) & 1) << 2)) << 4); }

// byte offset (inside an image) of this lane's transposed read for channels rb..rb+15, first pixel row kb:
// lane m of a 16-lane group addresses T[kb + (m >> 2)][rb + 4 (m & 3)] and receives T[kb .. kb+3][rb + m]
__device__ __forceinline__ uint32_t frag_base(int rb, int kb, int m) {
  const int r = rb + 4 * (m & 3), k = kb + (m >> 2);
  return (uint32_t)(img_off(k, r >> 3) + (r & 7) * 2);
}
// 8 consecutive pixels starting at the base row (+16 ks): two 4 x 16 transposes, rows +0 and +4 (the swizzle only
// looks at bit 1 of the row, so +4 and +16 are plain byte offsets)
__device__ __forceinline__ bf16x8 frag_at(const char* img, uint32_t base, int ks) {
  s16x4 v[2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
    v[h] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)(img + base + (ks * 16 + 4 * h) * 128));
  const s16x8 both = __builtin_shufflevector(v[0], v[1], 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, both);
}

__device__ __forceinline__ bf16x8 neg(bf16x8 v) {
  uint4 u = __builtin_bit_cast(uint4, v);
  u.x ^= 0x80008000u; u.y ^= 0x80008000u; u.z ^= 0x80008000u; u.w ^= 0x80008000u;
  return __builtin_bit_cast(bf16x8, u);
}

// grid: x = split, y = co tile * tiles_ci + ci tile
__global__ __launch_bounds__(NT) void conv_clr_wgrad_kernel(Args g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, lk = lane >> 5, l15 = lane & 15, lg = (lane >> 4) & 1;
  const int split = blockIdx.x;
  const int tco = blockIdx.y / g.tiles_ci, tci = blockIdx.y - tco * g.tiles_ci;
  const int co0 = tco * TC, ci0 = tci * TC;
  const uint32_t wid_u = (uint32_t)__builtin_amdgcn_readfirstlane(wid);

  // ---- this wave's blocks: 32 co rows (half coh) x blocks b = 2 tap + ci half: four full ones and a shared one --
  const int coh = (int)(wid_u >> 2), gq = (int)(wid_u & 3);
  const int fb = gq == 0 ? 0 : (gq == 1 ? 5 : (gq == 2 ? 9 : 14));
  const int hb = gq < 2 ? 4 : 13, hpar = gq & 1;
  uint32_t xa[5];                                   // per block: byte offset of the X fragment base inside the stage
  const uint32_t ga = frag_base(coh * 32 + 16 * lg, 8 * lk, l15);
#pragma unroll
  for (int n = 0; n < 5; ++n) {
    const int b = n < 4 ? fb + n : hb;
    const int tap = b >> 1, cih = b & 1, kh = tap / 3, kw = tap - 3 * kh;
    xa[n] = (uint32_t)(G_BYTES + kh * XW_BYTES) + frag_base(cih * 32 + 16 * lg, kw * g.dil_w + 8 * lk, l15);
  }

  f32x16 acc_r[5];
#pragma unroll
  for (int n = 0; n < 5; ++n) acc_r[n] = f32x16{0};

  // ---- LDS-DMA pieces.  Piece 0: the G tile (waves 0-3; waves 4-7 idle: zeros into the unused half).  Pieces 1, 2: the
  // X windows, unit u = 8 (j - 1) + wave: units 0..14 = (kernel row u / 5, rows 8 (u % 5) .. +8), unit 15 idle.
  const i32x4 rs_g = make_rsrc(g.g_r, g.g_bytes);
  const uint32_t rb_g = (uint32_t)g.Co * 2u, rb_x = (uint32_t)g.Ci * 2u;
  uint32_t vo[L], vflag[L];                          // lane offset inside the stage window; bit0 always out of range,
  {                                                  // bit1 top rows, bit2 bottom rows, bit3 set for every lane
    const int c = (int)(wid_u & 3) * 64 + lane, k = c >> 3, ch = (c & 7) ^ (((k >> 1) & 1) << 2);
    vo[0] = (uint32_t)k * rb_g + (uint32_t)(co0 + ch * 8) * 2u;
    vflag[0] = wid_u < 4 ? (uint32_t)k : 0x7fffffffu;  // (piece 0: the pixel of this lane inside the stage; idle waves: never)
  }
  int kh_of[L];
#pragma unroll
  for (int j = 1; j < L; ++j) {
    const int u = (j - 1) * 8 + (int)wid_u, kh_ = u / 5, sub = u - kh_ * 5;
    const int r = sub * 8 + (lane >> 3), ch = (lane & 7) ^ (((r >> 1) & 1) << 2);
    kh_of[j] = kh_ < 3 ? kh_ : 0;
    vo[j] = (uint32_t)r * rb_x + (uint32_t)(ci0 + ch * 8) * 2u;
    uint32_t f = 8u;
    if (u >= 15 || r >= KR + 2 * g.dil_w) f |= 1u;
    if (r < g.pad_w) f |= 2u;
    vflag[j] = f | ((uint32_t)r << 8);               // (bits 8..: the window row, for the bottom rows of a short stage)
  }
  const i32x4 rs_xr = make_rsrc(g.x_r, g.x_bytes);
  const uint32_t smem_off = lds_offset_of(smem);
  uint32_t dst[L];                                   // LDS destination inside a slot
  dst[0] = (wid_u >> 2) * 4096u + (wid_u & 3) * 1024u;
#pragma unroll
  for (int j = 1; j < L; ++j) {
    const uint32_t u = (uint32_t)(j - 1) * 8u + wid_u, kh_ = u / 5u, sub = u - kh_ * 5u;
    dst[j] = u < 15u ? (uint32_t)G_BYTES + kh_ * (uint32_t)XW_BYTES + sub * 1024u : (uint32_t)G_BYTES + 15u * 1024u;
  }

  // ---- stage pointer of the LDS-DMA (two stages ahead of the MFMAs) ----------------------------------------
  const int t0 = split * g.per_split;
  int nt = g.nstages - t0;
  nt = nt < g.per_split ? nt : g.per_split;
  nt = __builtin_amdgcn_readfirstlane(nt);
  int d_t = 0;                                       // stages issued so far
  // stages walk the image rows; a row takes ceil(W / 32) of them, the last one short when W % 32 != 0 (its missing
  // pixels have no G: out of range, zeros)
  uint32_t d_q0;                                     // first pixel of the stage at the pointer
  int d_w0, d_h, d_b;
  if (g.walk) {                                      // t = (b strips + strip) H + h
    const uint32_t col = (uint32_t)t0 / (uint32_t)g.H;
    d_h = (int)((uint32_t)t0 - col * (uint32_t)g.H);
    d_b = (int)(col / (uint32_t)g.strips);
    d_w0 = (int)(col - (uint32_t)d_b * (uint32_t)g.strips) * KR;
    d_q0 = ((uint32_t)d_b * (uint32_t)g.H + (uint32_t)d_h) * (uint32_t)g.W + (uint32_t)d_w0;
  } else {                                           // t = (b H + h) strips + strip
    const uint32_t row = (uint32_t)t0 / (uint32_t)g.strips;
    d_w0 = (int)((uint32_t)t0 - row * (uint32_t)g.strips) * KR;
    d_b = (int)(row / (uint32_t)g.H);
    d_h = (int)(row - (uint32_t)d_b * (uint32_t)g.H);
    d_q0 = row * (uint32_t)g.W + (uint32_t)d_w0;
  }
  auto issue_piece = [&](int j, uint32_t slot_off) __attribute__((always_inline)) {
    const bool live = d_t < nt;                      // stages past the end: everything out of range (zeros)
    if (j == 0) {
      // G lives on the (Ho, Wo) image: grid pixel (b, h, w0 + k) -> its dense row, nothing beyond row Ho / column Wo
      const uint32_t goff = (((uint32_t)d_b * (uint32_t)g.Ho + (uint32_t)d_h) * (uint32_t)g.Wo + (uint32_t)d_w0) * rb_g;
      const int lim = (live && d_h < g.Ho) ? g.Wo - d_w0 : 0;
      const uint32_t v = (int)vflag[0] < lim ? vo[0] + goff : OOB;
      buf_lds16(rs_g, v, slot_off + dst[0]);
    } else {
      const int kh = kh_of[j];
      const int hh = d_h + kh * g.dil_h - g.pad_h;
      // window rows that would show the next image row's pixels: from (pixels left in this image row) + p_w on
      const uint32_t nval = (uint32_t)(g.W - d_w0 < KR ? g.W - d_w0 : KR);
      const uint32_t bottom = d_w0 + KR >= g.W ? nval + (uint32_t)g.pad_w : 0x7fffffu;
      uint32_t sf = 1u | (d_w0 == 0 ? 2u : 0u) | ((hh < 0 || hh >= g.H || !live) ? 8u : 0u);
      if ((vflag[j] >> 8) >= bottom) sf |= 8u;
      const uint32_t soff = (d_q0 - (uint32_t)g.pad_w + (uint32_t)((kh * g.dil_h - g.pad_h) * g.W)) * rb_x;
      const uint32_t v = (vflag[j] & sf) ? OOB : vo[j] + soff;
      buf_lds16(rs_xr, v, slot_off + dst[j]);
    }
  };
  auto advance = [&]() __attribute__((always_inline)) {
    ++d_t;
    if (g.walk) {
      ++d_h; d_q0 += (uint32_t)g.W;
      if (d_h == g.H) {
        d_h = 0; d_w0 += KR; d_q0 += (uint32_t)KR - (uint32_t)g.H * (uint32_t)g.W;
        if (d_w0 >= g.W) { d_w0 = 0; ++d_b; d_q0 = (uint32_t)d_b * (uint32_t)g.H * (uint32_t)g.W; }
      }
      return;
    }
    d_q0 += KR; d_w0 += KR;
    if (d_w0 >= g.W) { d_q0 -= (uint32_t)(d_w0 - g.W); d_w0 = 0; if (++d_h == g.H) { d_h = 0; ++d_b; } }
  };

  // ---- one stage: 2 sub-steps of 16 pixels x (4 blocks + the shared block on this wave's parity) ---------------
  // One stage = 2 sub-steps of 16 pixels x (4 blocks + the shared block on this wave's parity).  Fragments are read one
  // block ahead of the MFMAs.  The barrier sits in front of the LAST block of the stage, when every read of this slot
  // has been issued and completed: the first fragments of the next stage are then read under that block's MFMAs
  // instead of right behind a barrier at which all eight waves (both of every SIMD) would wait for the LDS together.
  bf16x8 xr[2], gr[2], hr;
  auto mfma_block = [&](int n, bf16x8 ar, bf16x8 pr) __attribute__((always_inline)) {
    // X first: accumulator rows = co, 4 consecutive ci per register group
    acc_r[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar, pr, acc_r[n], 0, 0, 0);
  };
  auto stage = [&](uint32_t cur_off, uint32_t nxt_off, uint32_t next_cur_off) __attribute__((always_inline)) {
    const char* st = smem + cur_off;
    const char* sn = smem + next_cur_off;
    // ---- sub-step 0 (entry: gr[0], xr[0] hold block 0 of this stage)
    {
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const int cur = n & 1;
        if (n < 3) {
          xr[cur ^ 1] = frag_at(st, xa[n + 1], 0);
        } else {
          gr[1] = frag_at(st, ga, 1);
          xr[cur ^ 1] = frag_at(st, xa[0], 1);
          if (hpar == 0) hr = frag_at(st, xa[4], 0);
        }
        mfma_block(n, xr[cur], gr[0]);
        const int piece = n == 0 ? 0 : (n == 2 ? 1 : -1);
        if (piece >= 0) {
          __builtin_amdgcn_sched_barrier(0);
          issue_piece(piece, nxt_off);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (hpar == 0) mfma_block(4, hr, gr[0]);
    }
    // ---- sub-step 1 (xr[0] holds its block 0)
    {
#pragma unroll
      for (int n = 0; n < 3; ++n) {
        const int cur = n & 1;
        xr[cur ^ 1] = frag_at(st, xa[n + 1], 1);
        if (n == 2 && hpar == 1) hr = frag_at(st, xa[4], 1);
        mfma_block(n, xr[cur], gr[1]);
        if (n == 0) {
          __builtin_amdgcn_sched_barrier(0);
          issue_piece(2, nxt_off);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // every read of this slot has completed
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(L) : "memory");       // the next stage has landed (this wave's pieces)
      __builtin_amdgcn_s_barrier();
      gr[0] = frag_at(sn, ga, 0);
      xr[0] = frag_at(sn, xa[0], 0);
      mfma_block(3, xr[1], gr[1]);
      if (hpar == 1) mfma_block(4, hr, gr[1]);
    }
  };

  if (nt > 0) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int j = 0; j < L; ++j) issue_piece(j, smem_off + (uint32_t)(s * STAGE));
      advance();
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(L) : "memory");         // stage 0 landed; stage 1 may be in flight
    __builtin_amdgcn_s_barrier();
    gr[0] = frag_at(smem, ga, 0);
    xr[0] = frag_at(smem, xa[0], 0);
    uint32_t cur = 0, nx1 = STAGE, nx2 = 2u * STAGE;
    for (int t = 0; t < nt; ++t) {
      stage(cur, smem_off + nx2, nx1);
      advance();
      const uint32_t c = cur; cur = nx1; nx1 = nx2; nx2 = c;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // the zero re-loads land before the LDS is released
  }

  // ---- slabs [split][tile][NBLK][plane][32 co][32 ci] --------------------------------------------------------
  const int64_t tile = (int64_t)split * gridDim.y + blockIdx.y;
  float* base = g.ws + tile * (int64_t)(NBLK * 1024);
#pragma unroll
  for (int n = 0; n < 5; ++n) {
    const int b = n < 4 ? fb + n : hb;
    const int id = (n == 4 && hpar) ? 36 + coh * 2 + (hb == 13 ? 1 : 0) : coh * 18 + b;
    float* o_r = base + (int64_t)id * 1024;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f4 vr;
#pragma unroll
      for (int e = 0; e < 4; ++e) vr.v[e] = acc_r[n][4 * q + e];
      st4(o_r + l31 * 32 + 8 * q + 4 * lk, vr);
    }
  }
}

// dW[co][ci][kh][kw] = sum over splits of the block that holds it (+ the second half of a shared block), times emul if
// given (exp(log_sigma2) for d log_sigma2).  64 consecutive SLAB elements x 4 split lanes per block, fixed order.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* ws, int splits, int tiles, int tiles_ci, int Co,
                                                           int Ci, const float* emul, int emul_exp, float* dw) {
  __shared__ float red[4][64];
  const int tile = blockIdx.y;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + tx;                      // [36 blocks][32 co][32 ci]
  const int id = e >> 10, in_blk = e & 1023, col = in_blk >> 5, cil = in_blk & 31;
  const int coh = id / 18, b = id - coh * 18, tap = b >> 1, cih = b & 1;
  const int64_t per_split = (int64_t)tiles * NBLK * 1024;
  const float* p = ws + (int64_t)tile * NBLK * 1024 + e;
  const int64_t o2 = (b == 4 || b == 13) ? (int64_t)(36 + coh * 2 + (b == 13 ? 1 : 0) - id) * 1024 : 0;
  float a4[4] = {0.f, 0.f, 0.f, 0.f};
  int s = ty;
  for (; s + 12 < splits; s += 16) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float v = p[(int64_t)(s + 4 * u) * per_split];
      if (o2) v += p[(int64_t)(s + 4 * u) * per_split + o2];
      a4[u] += v;
    }
  }
  for (; s < splits; s += 4) {
    float v = p[(int64_t)s * per_split];
    if (o2) v += p[(int64_t)s * per_split + o2];
    a4[0] += v;
  }
  red[ty][tx] = (a4[0] + a4[1]) + (a4[2] + a4[3]);
  __syncthreads();
  if (ty != 0) return;
  const float acc = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
  const int co = (tile / tiles_ci) * TC + coh * 32 + col, ci = (tile % tiles_ci) * TC + cih * 32 + cil;
  const int64_t i = ((int64_t)co * Ci + ci) * 9 + tap;
  dw[i] = emul ? acc * (emul_exp ? __expf(emul[i]) : emul[i]) : acc;
}

// shared: the chip is shared with RCCL collectives (CPLXAMD_LAUNCH_SHARED): twice as many, half as long splits,
// so that the workgroups that find their CU taken do not make the launch take two rounds (the workspace is always sized
// for this plan)
static int plan(int64_t nstages, int tiles, int& per_split, bool shared) {
  const int ncu = device_cus();
  int64_t s = ncu / tiles;                            // one workgroup per CU (120 KiB of LDS each), one round
  if (s < 1) s = 1;
  if (shared) s *= 2;
  const int64_t maxs = (nstages + 15) / 16;           // >= 16 stages per split
  if (s > maxs) s = maxs;
  per_split = (int)((nstages + s - 1) / s);
  return (int)((nstages + per_split - 1) / per_split);
}

}  // namespace clwr
}  // namespace cplxamd

using namespace cplxamd;

extern "C" {

static int clwr_shape_ok(int64_t B, int H, int W, int Ci, int Co, int KH, int KW, int dil_h, int dil_w, int pad_h,
                         int pad_w) {
  if (KH != 3 || KW != 3 || Ci % 64 || Co % 64 || pad_h < 0 || pad_w < 0 || pad_h > dil_h ||
      pad_w > dil_w || dil_w > 4 || H + 2 * pad_h - 2 * dil_h <= 0 || W + 2 * pad_w - 2 * dil_w <= 0)
    return 0;
  const int64_t P = B * H * W;
  const int64_t cmax = Ci > Co ? Ci : Co;
  if (P >= ((int64_t)1 << 31) || (P + (int64_t)pad_h * W + 64) * cmax * 2 >= ((int64_t)1 << 32) - 64) return 0;
  return 1;
}

int64_t cplxamd_conv2d_clr_wgrad_ws_bytes(int64_t B, int H, int W, int Ci, int Co) {
  if (B <= 0 || H <= 0 || W <= 0 || Ci <= 0 || Co <= 0) return 0;
  const int tiles = ((Co + 63) / 64) * ((Ci + 63) / 64);
  int per_split = 0;
  const int splits = clwr::plan(B * H * ((W + clwr::KR - 1) / clwr::KR), tiles, per_split, true);
  return (int64_t)splits * tiles * clwr::NBLK * 1024 * 4;
}

// dw: float32 [Co][Ci][3][3] = (sum_q g x) * emul (emul_exp: * exp(emul)); x: [B][H][W][Ci], g: [B][Ho][Wo][Co] bf16.
int cplxamd_conv2d_clr_wgrad(const void* g_, const void* x, const float* emul, int emul_exp, float* dw, int64_t B, int H,
                             int W, int Ci, int Co, int KH, int KW, int dil_h, int dil_w, int pad_h, int pad_w, void* ws,
                             int64_t ws_bytes, void* stream) {
  return cplxamd_conv2d_clr_wgrad_fl(g_, x, emul, emul_exp, dw, B, H, W, Ci, Co, KH, KW, dil_h, dil_w, pad_h, pad_w, ws, ws_bytes,
                                     CPLXAMD_LAUNCH_DEFAULT, stream);
}

int cplxamd_conv2d_clr_wgrad_fl(const void* g_, const void* x, const float* emul, int emul_exp, float* dw, int64_t B, int H,
                                int W, int Ci, int Co, int KH, int KW, int dil_h, int dil_w, int pad_h, int pad_w, void* ws,
                                int64_t ws_bytes, int flags, void* stream) {
  if (!launch_flags_ok(flags)) return CPLXAMD_EINVAL;
  if (!g_ || !x || !dw || B < 0 || H <= 0 || W <= 0 || Ci <= 0 || Co <= 0) return CPLXAMD_EINVAL;
  if (!clwr_shape_ok(B, H, W, Ci, Co, KH, KW, dil_h, dil_w, pad_h, pad_w)) return CPLXAMD_ESHAPE;
  hipStream_t st = (hipStream_t)stream;
  const int64_t P = B * H * W;
  if (P == 0) {
    hipMemsetAsync(dw, 0, (size_t)Co * Ci * 9 * 4, st);
    return 0;
  }
  auto a16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  if (!a16(g_) || !a16(x) || !a16(ws)) return CPLXAMD_EALIGN;
  if (!ws || ws_bytes < cplxamd_conv2d_clr_wgrad_ws_bytes(B, H, W, Ci, Co)) return CPLXAMD_EINVAL;
  clwr::Args g{};
  g.g_r = g_; g.x_r = x; g.ws = (float*)ws; g.P = P;
  g.Ho = H + 2 * pad_h - 2 * dil_h; g.Wo = W + 2 * pad_w - 2 * dil_w;
  g.g_bytes = (uint32_t)(B * g.Ho * g.Wo * Co * 2); g.x_bytes = (uint32_t)(P * Ci * 2);
  g.H = H; g.W = W; g.Co = Co; g.Ci = Ci; g.dil_h = dil_h; g.dil_w = dil_w; g.pad_h = pad_h; g.pad_w = pad_w;
  g.strips = (W + clwr::KR - 1) / clwr::KR;
  g.nstages = (int)(B * H * g.strips);
  g.tiles_ci = Ci / 64;
  const int tiles = (Co / 64) * g.tiles_ci;
  g.splits = clwr::plan(g.nstages, tiles, g.per_split, !launch_owns_chip(flags));
  {
    static const int w = [] { const char* e = getenv("CPLXAMD_CLW_WALK"); return e ? (atoi(e) != 0) : 1; }();
    g.walk = w;
  }
  constexpr int smem = 3 * clwr::STAGE;
  static PerDeviceOnce attr_set;
  if (const int e = set_max_dyn_lds(attr_set, clwr::conv_clr_wgrad_kernel, smem)) return e;
  clwr::conv_clr_wgrad_kernel<<<dim3((unsigned)g.splits, (unsigned)tiles), clwr::NT, smem, st>>>(g);
  CPLXAMD_CHECK_LAUNCH();
  clwr::wgrad_reduce_kernel<<<dim3(36 * 1024 / 64, (unsigned)tiles), 256, 0, st>>>(g.ws, g.splits, tiles, g.tiles_ci, Co, Ci,
                                                                                   emul, emul_exp, dw);
  CPLXAMD_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
