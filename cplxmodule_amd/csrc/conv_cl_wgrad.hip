// Weight gradient of the complex 3 x 3 "same" convolution on UNPADDED channels-last activations (bf16 in, fp32 out):
//
//   dW[co, ci, kh, kw] = sum_q  G[q][co] * conj(X[q + (kh d_h - p_h) W + (kw d_w - p_w)][ci])   over the pixels q whose
//                        tap stays inside the image
//
// (autograd of cplx.convnd, cplxmodule/cplx.py:717-838; dW = G^H-correlation per SURVEY A.1).  Per tap a (T, T) GEMM
// with M = Co, N = Ci, K = pixels; both operands are K-major as stored, so fragments come from ds_read_b64_tr_b16.
//
// Work split.  The r01 kernel gave a workgroup one kernel row and each of its three waves one tap (a 64 x 64 complex
// tile = 128 accumulators): 6 waves per CU at best, two SIMDs with a single wave.  Nine taps do not divide by eight
// waves -- but 36 blocks of 32 x 32 do, at four and a HALF blocks per wave: here ONE workgroup of eight waves owns all
// nine taps of a (64 co x 64 ci) tile over its share of the pixels.  Wave (co half, g = 0..3) holds 32 co rows x
// {4 full (tap, ci half) blocks + one block it shares with its neighbour}; the shared block is split along K (the two
// waves take alternate 16-pixel sub-steps) and lands in the slab twice.  All eight waves read the same G tile and the
// same three X windows (one per kernel row, 32 + halo pixels each): 40 KiB per 32-pixel stage and 288 MFMAs.
// LDS reads per MFMA are 20 % above the 64 x 64 tile's (one G fragment pair feeds 4.5 blocks).
//
// Borders without masks in the K loop.  Stages walk the image rows, ceil(W / 32) per row (the last one short when
// W % 32 != 0: its missing pixels have no G -- out of range, zeros -- and contribute nothing), so a stage lies inside
// one image row: (a) a kernel row that leaves the image does so for the whole stage and (b) the only window rows a
// column tap must not see are the first p_w rows of the first stage of an image row and the rows behind the last
// pixel + p_w of its last stage -- and those LDS rows are read by no other tap of a pixel that has a G.  The LDS-DMA is `buffer_load ... lds`: a lane whose row must read as zero is simply given an
// out-of-range offset and the hardware writes zeros (scripts/r02/bufload_oob.hip).  Rows before / after the tensor
// fall out of range by themselves.
//
// 3-slot ring, one s_barrier per stage, counted vmcnt (5 LDS-DMA pieces per wave and stage, nothing else in the loop);
// fp32 slabs [split][tile][40 blocks][plane][32][32] and a deterministic reduce into dW[Co][Ci][3][3].
#include <stdlib.h>

#include "common.h"
#include "launch.h"   // per-call launch policy (CPLXAMD_LAUNCH_SHARED: the chip is shared with collectives)

namespace cplxamd {
namespace clw {

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
constexpr int L = 5;                               // LDS-DMA pieces per wave and stage
constexpr int NBLK = 40;                           // slab blocks per tile: 2 x 18 + the 4 second halves
constexpr uint32_t OOB = 0xFFFFFFF0u;

struct Args {
  const void* g_r; const void* g_i;                // [P][Co] bf16
  const void* x_r; const void* x_i;                // [P][Ci] bf16
  float* ws;
  int64_t P;
  uint32_t g_bytes, x_bytes;
  int H, W, Co, Ci, dil_h, dil_w, pad_h, pad_w;      // H x W: the input image = the grid the pixel loop walks
  int Ho, Wo;                                        // the output image (<= H x W, top-left aligned on the grid)
  int strips;                                        // stages per image row: ceil(W / 32)
  int nstages, per_split, splits, tiles_ci;
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
__global__ __launch_bounds__(NT) void conv_cl_wgrad_kernel(Args g) {
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
    xa[n] = (uint32_t)(G_BYTES + kh * 2 * XW_BYTES) + frag_base(cih * 32 + 16 * lg, kw * g.dil_w + 8 * lk, l15);
  }

  f32x16 acc_r[5], acc_i[5];
#pragma unroll
  for (int n = 0; n < 5; ++n) { acc_r[n] = f32x16{0}; acc_i[n] = f32x16{0}; }

  // ---- LDS-DMA pieces.  Piece 0: the G tile (waves 0-3 real plane, 4-7 imaginary).  Pieces 1, 2: real plane of X,
  // pieces 3, 4: imaginary plane; unit u = 8 ((j-1) & 1) + wave: units 0..14 = (kernel row u / 5, rows 8 (u % 5) .. +8),
  // unit 15 idle (zeros).  (One buffer descriptor per piece index, so that it stays in scalar registers.)
  const i32x4 rs_g = make_rsrc(wid_u < 4 ? g.g_r : g.g_i, g.g_bytes);
  const uint32_t rb_g = (uint32_t)g.Co * 2u, rb_x = (uint32_t)g.Ci * 2u;
  uint32_t vo[L], vflag[L];                          // lane offset inside the stage window; bit0 always out of range,
  {                                                  // bit1 top rows, bit2 bottom rows, bit3 set for every lane
    const int c = (int)(wid_u & 3) * 64 + lane, k = c >> 3, ch = (c & 7) ^ (((k >> 1) & 1) << 2);
    vo[0] = (uint32_t)k * rb_g + (uint32_t)(co0 + ch * 8) * 2u;
    vflag[0] = (uint32_t)k;                          // (piece 0: the pixel of this lane inside the stage)
  }
  int kh_of[L];
#pragma unroll
  for (int j = 1; j < L; ++j) {
    const int u = ((j - 1) & 1) * 8 + (int)wid_u, kh_ = u / 5, sub = u - kh_ * 5;
    const int r = sub * 8 + (lane >> 3), ch = (lane & 7) ^ (((r >> 1) & 1) << 2);
    kh_of[j] = kh_ < 3 ? kh_ : 0;
    vo[j] = (uint32_t)r * rb_x + (uint32_t)(ci0 + ch * 8) * 2u;
    uint32_t f = 8u;
    if (u >= 15 || r >= KR + 2 * g.dil_w) f |= 1u;
    if (r < g.pad_w) f |= 2u;
    vflag[j] = f | ((uint32_t)r << 8);               // (bits 8..: the window row, for the bottom rows of a short stage)
  }
  const i32x4 rs_xr = make_rsrc(g.x_r, g.x_bytes), rs_xi = make_rsrc(g.x_i, g.x_bytes);
  const uint32_t smem_off = lds_offset_of(smem);
  uint32_t dst[L];                                   // LDS destination inside a slot
  dst[0] = (wid_u >> 2) * 4096u + (wid_u & 3) * 1024u;
#pragma unroll
  for (int j = 1; j < L; ++j) {
    const uint32_t pl = (uint32_t)(j - 1) >> 1, u = ((uint32_t)(j - 1) & 1u) * 8u + wid_u, kh_ = u / 5u, sub = u - kh_ * 5u;
    dst[j] = u < 15u ? (uint32_t)G_BYTES + (kh_ * 2u + pl) * (uint32_t)XW_BYTES + sub * 1024u
                     : (uint32_t)G_BYTES + 30u * 1024u + pl * 1024u;
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
  {
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
      buf_lds16(j < 3 ? rs_xr : rs_xi, v, slot_off + dst[j]);
    }
  };
  auto advance = [&]() __attribute__((always_inline)) {
    ++d_t; d_q0 += KR; d_w0 += KR;
    if (d_w0 >= g.W) { d_q0 -= (uint32_t)(d_w0 - g.W); d_w0 = 0; if (++d_h == g.H) { d_h = 0; ++d_b; } }
  };

  // ---- one stage: 2 sub-steps of 16 pixels x (4 blocks + the shared block on this wave's parity) ---------------
  // One stage = 2 sub-steps of 16 pixels x (4 blocks + the shared block on this wave's parity).  Fragments are read one
  // block ahead of the MFMAs.  The barrier sits in front of the LAST block of the stage, when every read of this slot
  // has been issued and completed: the first fragments of the next stage are then read under that block's MFMAs
  // instead of right behind a barrier at which all eight waves (both of every SIMD) would wait for the LDS together.
  bf16x8 xr[2], xi[2], gr[2], gi[2], hr, hi;
  auto mfma_block = [&](int n, bf16x8 ar, bf16x8 ai, bf16x8 pr, bf16x8 pi, bf16x8 npr) __attribute__((always_inline)) {
    // G conj(X): re = gr xr + gi xi, im = gi xr - gr xi; X first: accumulator rows = co, 4 consecutive ci per group
    acc_r[n] = CPLXAMD_MFMA16(ar, pr, acc_r[n]);
    acc_i[n] = CPLXAMD_MFMA16(ar, pi, acc_i[n]);
    acc_r[n] = CPLXAMD_MFMA16(ai, pi, acc_r[n]);
    acc_i[n] = CPLXAMD_MFMA16(ai, npr, acc_i[n]);
  };
  auto stage = [&](uint32_t cur_off, uint32_t nxt_off, uint32_t next_cur_off) __attribute__((always_inline)) {
    const char* st = smem + cur_off;
    const char* sn = smem + next_cur_off;
    // ---- sub-step 0 (entry: gr[0], gi[0], xr[0], xi[0] hold block 0 of this stage)
    {
      const bf16x8 ngr = neg(gr[0]);
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const int cur = n & 1;
        if (n < 3) {
          xr[cur ^ 1] = frag_at(st, xa[n + 1], 0); xi[cur ^ 1] = frag_at(st + XW_BYTES, xa[n + 1], 0);
        } else {
          gr[1] = frag_at(st, ga, 1); gi[1] = frag_at(st + KR * 128, ga, 1);
          xr[cur ^ 1] = frag_at(st, xa[0], 1); xi[cur ^ 1] = frag_at(st + XW_BYTES, xa[0], 1);
          if (hpar == 0) { hr = frag_at(st, xa[4], 0); hi = frag_at(st + XW_BYTES, xa[4], 0); }
        }
        mfma_block(n, xr[cur], xi[cur], gr[0], gi[0], ngr);
        const int piece = n == 0 ? 0 : (n == 2 ? 1 : (n == 3 ? 2 : -1));
        if (piece >= 0) {
          __builtin_amdgcn_sched_barrier(0);
          issue_piece(piece, nxt_off);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (hpar == 0) mfma_block(4, hr, hi, gr[0], gi[0], ngr);
    }
    // ---- sub-step 1 (xr[0], xi[0] hold its block 0)
    {
      const bf16x8 ngr = neg(gr[1]);
#pragma unroll
      for (int n = 0; n < 3; ++n) {
        const int cur = n & 1;
        xr[cur ^ 1] = frag_at(st, xa[n + 1], 1); xi[cur ^ 1] = frag_at(st + XW_BYTES, xa[n + 1], 1);
        if (n == 2 && hpar == 1) { hr = frag_at(st, xa[4], 1); hi = frag_at(st + XW_BYTES, xa[4], 1); }
        mfma_block(n, xr[cur], xi[cur], gr[1], gi[1], ngr);
        const int piece = n == 0 ? 3 : (n == 2 ? 4 : -1);
        if (piece >= 0) {
          __builtin_amdgcn_sched_barrier(0);
          issue_piece(piece, nxt_off);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // every read of this slot has completed
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(L) : "memory");       // the next stage has landed (this wave's pieces)
      __builtin_amdgcn_s_barrier();
      gr[0] = frag_at(sn, ga, 0); gi[0] = frag_at(sn + KR * 128, ga, 0);
      xr[0] = frag_at(sn, xa[0], 0); xi[0] = frag_at(sn + XW_BYTES, xa[0], 0);
      mfma_block(3, xr[1], xi[1], gr[1], gi[1], ngr);
      if (hpar == 1) mfma_block(4, hr, hi, gr[1], gi[1], ngr);
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
    gr[0] = frag_at(smem, ga, 0); gi[0] = frag_at(smem + KR * 128, ga, 0);
    xr[0] = frag_at(smem, xa[0], 0); xi[0] = frag_at(smem + XW_BYTES, xa[0], 0);
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
  float* base = g.ws + tile * (int64_t)(NBLK * 2 * 1024);
#pragma unroll
  for (int n = 0; n < 5; ++n) {
    const int b = n < 4 ? fb + n : hb;
    const int id = (n == 4 && hpar) ? 36 + coh * 2 + (hb == 13 ? 1 : 0) : coh * 18 + b;
    float* o_r = base + (int64_t)id * 2048;
    float* o_i = o_r + 1024;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f4 vr, vi;
#pragma unroll
      for (int e = 0; e < 4; ++e) { vr.v[e] = acc_r[n][4 * q + e]; vi.v[e] = acc_i[n][4 * q + e]; }
      st4(o_r + l31 * 32 + 8 * q + 4 * lk, vr);
      st4(o_i + l31 * 32 + 8 * q + 4 * lk, vi);
    }
  }
}

// dW[co][ci][kh][kw] (plane) = sum over splits of the block that holds it (+ the second half of a shared block), times
// emul if given.  64 consecutive SLAB elements x 4 split lanes per block (consecutive threads read consecutive
// addresses of a split; lane s sums splits s, s + 4, ... in a fixed order, then the four are added in order): the
// 4-byte writes into dW are scattered but few.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* ws, int splits, int tiles, int tiles_ci, int Co,
                                                           int Ci, const float* emul, float* dw_r, float* dw_i) {
  __shared__ float red[4][64];
  const int tile = blockIdx.y;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + tx;                      // [36 blocks][plane][32 co][32 ci]
  const int id = e >> 11, in_blk = e & 2047, pl = in_blk >> 10, col = (in_blk >> 5) & 31, cil = in_blk & 31;
  const int coh = id / 18, b = id - coh * 18, tap = b >> 1, cih = b & 1;
  const int64_t per_split = (int64_t)tiles * NBLK * 2048;
  const float* p = ws + (int64_t)tile * NBLK * 2048 + e;
  const int64_t o2 = (b == 4 || b == 13) ? (int64_t)(36 + coh * 2 + (b == 13 ? 1 : 0) - id) * 2048 : 0;
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
  float* dw = pl ? dw_i : dw_r;
  dw[i] = emul ? acc * emul[i] : acc;
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

}  // namespace clw
}  // namespace cplxamd

using namespace cplxamd;

extern "C" {

static int clw_shape_ok(int64_t B, int H, int W, int Ci, int Co, int KH, int KW, int dil_h, int dil_w, int pad_h,
                        int pad_w) {
  if (KH != 3 || KW != 3 || Ci % 64 || Co % 64 || pad_h < 0 || pad_w < 0 || pad_h > dil_h ||
      pad_w > dil_w || dil_w > 4 || H + 2 * pad_h - 2 * dil_h <= 0 || W + 2 * pad_w - 2 * dil_w <= 0)
    return 0;
  const int64_t P = B * H * W;
  const int64_t cmax = Ci > Co ? Ci : Co;
  if (P >= ((int64_t)1 << 31) || (P + (int64_t)pad_h * W + 64) * cmax * 2 >= ((int64_t)1 << 32) - 64) return 0;
  return 1;
}

int64_t cplxamd_conv2d_cl_wgrad_ws_bytes(int64_t B, int H, int W, int Ci, int Co) {
  if (B <= 0 || H <= 0 || W <= 0 || Ci <= 0 || Co <= 0) return 0;
  const int tiles = ((Co + 63) / 64) * ((Ci + 63) / 64);
  int per_split = 0;
  const int splits = clw::plan(B * H * ((W + clw::KR - 1) / clw::KR), tiles, per_split, true);
  return (int64_t)splits * tiles * clw::NBLK * 2048 * 4;
}

// dw_r / dw_i: float32 [Co][Ci][3][3]; x: [B][H][W][Ci], g: [B][Ho][Wo][Co] channels-last bf16 planes, Ho = H + 2 pad_h -
// 2 dil_h <= H (any zero padding up to `same`).
int cplxamd_conv2d_cl_wgrad(const void* g_r, const void* g_i, const void* x_r, const void* x_i, const float* emul,
                            float* dw_r, float* dw_i, int64_t B, int H, int W, int Ci, int Co, int KH, int KW, int dil_h,
                            int dil_w, int pad_h, int pad_w, void* ws, int64_t ws_bytes, void* stream) {
  return cplxamd_conv2d_cl_wgrad_fl(g_r, g_i, x_r, x_i, emul, dw_r, dw_i, B, H, W, Ci, Co, KH, KW, dil_h, dil_w, pad_h, pad_w, ws,
                                    ws_bytes, CPLXAMD_LAUNCH_DEFAULT, stream);
}

int cplxamd_conv2d_cl_wgrad_fl(const void* g_r, const void* g_i, const void* x_r, const void* x_i, const float* emul,
                               float* dw_r, float* dw_i, int64_t B, int H, int W, int Ci, int Co, int KH, int KW, int dil_h,
                               int dil_w, int pad_h, int pad_w, void* ws, int64_t ws_bytes, int flags, void* stream) {
  if (!launch_flags_ok(flags)) return CPLXAMD_EINVAL;
  if (!g_r || !g_i || !x_r || !x_i || !dw_r || !dw_i || B < 0 || H <= 0 || W <= 0 || Ci <= 0 || Co <= 0) return CPLXAMD_EINVAL;
  if (!clw_shape_ok(B, H, W, Ci, Co, KH, KW, dil_h, dil_w, pad_h, pad_w)) return CPLXAMD_ESHAPE;
  hipStream_t st = (hipStream_t)stream;
  const int64_t P = B * H * W;
  if (P == 0) {
    hipMemsetAsync(dw_r, 0, (size_t)Co * Ci * 9 * 4, st);
    hipMemsetAsync(dw_i, 0, (size_t)Co * Ci * 9 * 4, st);
    return 0;
  }
  auto a16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  if (!a16(g_r) || !a16(g_i) || !a16(x_r) || !a16(x_i) || !a16(ws)) return CPLXAMD_EALIGN;
  if (!ws || ws_bytes < cplxamd_conv2d_cl_wgrad_ws_bytes(B, H, W, Ci, Co)) return CPLXAMD_EINVAL;
  clw::Args g{};
  g.g_r = g_r; g.g_i = g_i; g.x_r = x_r; g.x_i = x_i; g.ws = (float*)ws; g.P = P;
  g.Ho = H + 2 * pad_h - 2 * dil_h; g.Wo = W + 2 * pad_w - 2 * dil_w;
  g.g_bytes = (uint32_t)(B * g.Ho * g.Wo * Co * 2); g.x_bytes = (uint32_t)(P * Ci * 2);
  g.H = H; g.W = W; g.Co = Co; g.Ci = Ci; g.dil_h = dil_h; g.dil_w = dil_w; g.pad_h = pad_h; g.pad_w = pad_w;
  g.strips = (W + clw::KR - 1) / clw::KR;
  g.nstages = (int)(B * H * g.strips);
  g.tiles_ci = Ci / 64;
  const int tiles = (Co / 64) * g.tiles_ci;
  g.splits = clw::plan(g.nstages, tiles, g.per_split, !launch_owns_chip(flags));
  constexpr int smem = 3 * clw::STAGE;
  static PerDeviceOnce attr_set;
  if (const int e = set_max_dyn_lds(attr_set, clw::conv_cl_wgrad_kernel, smem)) return e;
  clw::conv_cl_wgrad_kernel<<<dim3((unsigned)g.splits, (unsigned)tiles), clw::NT, smem, st>>>(g);
  CPLXAMD_CHECK_LAUNCH();
  clw::wgrad_reduce_kernel<<<dim3(36 * 2048 / 64, (unsigned)tiles), 256, 0, st>>>(g.ws, g.splits, tiles, g.tiles_ci, Co, Ci,
                                                                                   emul, dw_r, dw_i);
  CPLXAMD_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
