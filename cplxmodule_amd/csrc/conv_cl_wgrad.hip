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

#ifndef CLW_NT
#define CLW_NT 3       // FOLD: nontemporal G stores (bit 0) / raw tile loads (bit 1)
#endif
namespace cplxamd {
namespace clw {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int KR = 32, NT = 512, TC = 64;
constexpr int G_BYTES = 2 * KR * 128;              // [plane][32 pixels][64 co]
constexpr int XW_ROWS = 40, XW_BYTES = XW_ROWS * 128;   // one (kernel row, plane) window: 34 rows used
constexpr int X_BYTES = 4 * 8192;                  // 6 windows (30 KiB) + 2 KiB the idle lanes of the last piece zero
constexpr int STAGE = G_BYTES + X_BYTES;           // 40 KiB
constexpr int L = 5;                               // LDS-DMA pieces per wave and stage
constexpr int NBLK = 40;                           // slab blocks per tile: 2 x 18 + the 4 second halves
constexpr uint32_t OOB = 0xFFFFFFF0u;
// FOLD (the batch-norm backward apply folded into the G staging, see the kernel): two raw tiles
// [g_r | g_i | z_r | z_i][32 pixels][128 B] behind the ring
constexpr int RAW = 3 * STAGE, RAW_BYTES = 4 * KR * 128, SMEM_FOLD = RAW + 2 * RAW_BYTES;
static_assert(SMEM_FOLD <= 160 * 1024, "LDS");

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
  int walk;                                          // stage order: 0 = along image rows, 1 = down image columns (strip by strip)
  int sk_co, sk_ci;                                  // tiles (co tile < sk_co, ci tile < sk_ci) are not computed (grid y is compact)
  // FOLD: G = A g + B (z - mu) - k per output channel (bn.hip's backward apply) is formed on the way into the LDS
  const void* z_r; const void* z_i;                // [P][Co] bf16: the batch-norm layer's input = the convolution's output
  const float* coef;                               // [Co][kBnBwdCoef] (bn_bwd_finalize)
  void* dy_r; void* dy_i;                          // [P][Co] bf16 out: G as the MFMAs see it (the data gradient reads it next)
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

// the same for data that is read once (FOLD: the raw g / z tiles): nontemporal, so that it does not push the x rows -- each
// fetched by three kernel rows -- out of the L2
__device__ __forceinline__ void buf_lds16_nt(i32x4 rsrc, uint32_t voff, uint32_t lds_off_uniform) {
#if defined(__HIP_DEVICE_COMPILE__)
  const uint32_t m0v = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_off_uniform);
#if CLW_NT & 2
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen nt lds"
#else
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds"
#endif
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

// compact tile index -> (co tile, ci tile) when the rectangle (co tile < sk_co, ci tile < sk_ci) is left out
__device__ __forceinline__ void tile_of(int v, int tiles_ci, int sk_co, int sk_ci, int& tco, int& tci) {
  const int head = sk_co * (tiles_ci - sk_ci);
  if (v < head) { tco = v / (tiles_ci - sk_ci); tci = sk_ci + v - tco * (tiles_ci - sk_ci); }
  else { v -= head; tco = v / tiles_ci; tci = v - tco * tiles_ci; tco += sk_co; }
}

// grid: x = split, y = the tiles, co tile major (tile_of)
//
// FOLD: G is not in memory yet -- it is the input gradient of the batch-norm layer that consumed this convolution's output
// z, G = E g + C (z - mu) - k per output channel (the 2 x 2 real matrices and constants bn_bwd_finalize leaves in `coef`;
// bn_apply_rows<BWD>'s arithmetic with the means multiplied out: G = E g + C z + c, c = -k - C mu, 9 coefficients per
// channel in registers).  The G tile of a stage is then not LDS-DMA'd: the raw g and z tiles (4 planes x 32 pixels x 64
// channels, two LDS-DMA pieces per wave) land in one of two raw buffers a full stage ahead; during stage t every lane
// turns (2 pixels) x (2 channels) of stage t + 1's raw tiles into G -- 8 LDS dword reads, ~30 packed VALU operations that
// run in the shadow of the other wave's MFMAs, bf16 rounding -- and writes them into the G image of stage t + 1's slot,
// where the MFMAs find it as before; the per-lane column sums of G (the convolution's bias gradient) are kept on the way.
// G also goes to memory (dy_r / dy_i: the data gradient reads it after this launch): during stage t every lane reads 16
// bytes of the finished G image of stage t back -- the lane mapping of the LDS-DMA piece it replaces, inverted -- and
// stores them: one store instruction per wave and stage (stores straight from the forming lanes, 4 bytes each, cost 0.5 ms
// of the launch at cfg3).  The batch-norm layer's apply pass -- 4 planes read, 2 written, 2.4-2.6 ms at cfg3 -- is not
// launched.  VMEM order per wave and stage: raw g, raw z | store | pieces 1, 2, 3, 4; the stage-end wait leaves the 5
// youngest in flight (the raw tiles were issued first and must have landed: every wave reads every wave's lanes of them
// behind the barrier).
template <bool FOLD>
__global__ __launch_bounds__(NT) void conv_cl_wgrad_kernel(Args g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, lk = lane >> 5, l15 = lane & 15, lg = (lane >> 4) & 1;
  const int split = blockIdx.x;
  int tco, tci;
  tile_of((int)blockIdx.y, g.tiles_ci, g.sk_co, g.sk_ci, tco, tci);
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
    // (FOLD: the raw g / z tiles are fetched with this very mapping -- they sit in the LDS swizzled like the G image the
    //  lanes form from them, one offset serves both -- and the store of the finished tile reads G back with it)
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
      buf_lds16(j < 3 ? rs_xr : rs_xi, v, slot_off + dst[j]);
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

  // ---- FOLD: lane (channel pair cp, pixel slot ps) forms G for pixels ps and ps + 16 of the stage at the pointer e_* ----
  const int cp = lane & 31, ps = (lane >> 5) + 2 * (int)wid_u;
  const bool writer = tci == 0;                      // (one ci tile stores G and its sums)
  f32x2 kc[9];                                       // e00 e01 e10 e11 | cuu cuv cvv | c_r c_i, each for this lane's two channels
  uint32_t f_img = 0;                                // this lane's dword inside a raw plane and inside a G image plane
  uint32_t vo_b = 0;                                 // x pieces: lane offset of window row lane >> 3 (FOLD: one for all)
  i32x4 rs_z = rs_g, rs_dy = rs_g;
  // e: the stage whose G is being formed (one ahead of the MFMAs); p: the one before it, whose finished G is being stored
  int e_t = 0, e_w0 = d_w0, e_h = d_h, e_lim = 0, p_lim = 0;
  uint32_t e_row = 0, p_goff = 0;                    // G byte offset of the first pixel of e's image row
  auto e_set = [&]() __attribute__((always_inline)) {
    e_lim = (e_t < nt && e_h < g.Ho) ? g.Wo - e_w0 : 0;
  };
  if constexpr (FOLD) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const float* kp = g.coef + (int64_t)(co0 + 2 * cp + c) * kBnBwdCoef;
#pragma unroll
      for (int j = 0; j < 7; ++j) kc[j][c] = kp[2 + j];
      kc[7][c] = -kp[9] - kp[6] * kp[0] - kp[7] * kp[1];
      kc[8][c] = -kp[10] - kp[7] * kp[0] - kp[8] * kp[1];
    }
    f_img = (uint32_t)(img_off(ps, cp >> 2) + (cp & 3) * 4);     // (pixel ps + 16 swizzles alike: + 16 rows)
    {
      const int rl = lane >> 3, ch = (lane & 7) ^ (((rl >> 1) & 1) << 2);     // (rows r = 8 sub + rl swizzle like rl)
      vo_b = (uint32_t)rl * rb_x + (uint32_t)(ci0 + ch * 8) * 2u;
    }
    rs_z = make_rsrc(wid_u < 4 ? g.z_r : g.z_i, g.g_bytes);
    rs_dy = make_rsrc(wid_u < 4 ? g.dy_r : g.dy_i, g.g_bytes);
    e_row = (((uint32_t)d_b * (uint32_t)g.Ho + (uint32_t)d_h) * (uint32_t)g.Wo) * rb_g;
    e_set();
  }
  auto issue_raw = [&](uint32_t raw_lds) __attribute__((always_inline)) {
    const bool live = d_t < nt;
    const uint32_t goff = (((uint32_t)d_b * (uint32_t)g.Ho + (uint32_t)d_h) * (uint32_t)g.Wo + (uint32_t)d_w0) * rb_g;
    const int lim = (live && d_h < g.Ho) ? g.Wo - d_w0 : 0;
    const uint32_t v = (int)vflag[0] < lim ? vo[0] + goff : OOB;
    buf_lds16_nt(rs_g, v, raw_lds + dst[0]);
    buf_lds16_nt(rs_z, v, raw_lds + (uint32_t)(2 * KR * 128) + dst[0]);
  };
  // the x pieces as issue_piece does them, on ONE lane offset (vo_b) and the lane's window row recomputed per use:
  // the row and flag words of the four pieces are registers this variant does not have
  int sub8_of[L];
#pragma unroll
  for (int j = 1; j < L; ++j) {
    const int u = ((j - 1) & 1) * 8 + (int)wid_u, kh_ = u / 5;
    sub8_of[j] = u >= 15 ? -1 : (u - kh_ * 5) * 8;
  }
  auto issue_piece_f = [&](int j, uint32_t slot_off) __attribute__((always_inline)) {
    const bool live = d_t < nt;
    const int kh = kh_of[j], sub8 = sub8_of[j];
    const int hh = d_h + kh * g.dil_h - g.pad_h;
    const int nval = g.W - d_w0 < KR ? g.W - d_w0 : KR;
    int hi = KR + 2 * g.dil_w;
    if (d_w0 + KR >= g.W && nval + g.pad_w < hi) hi = nval + g.pad_w;
    const int lo = d_w0 == 0 ? g.pad_w : 0;
    const bool none = hh < 0 || hh >= g.H || !live || sub8 < 0;
    const uint32_t soff = (d_q0 - (uint32_t)g.pad_w + (uint32_t)((kh * g.dil_h - g.pad_h) * g.W) + (uint32_t)sub8) * rb_x;
    uint32_t l = (uint32_t)lane;
    asm volatile("" : "+v"(l));                      // (keeps lane >> 3 out of a loop-invariant register)
    const uint32_t span = none ? 0u : (uint32_t)(hi - lo);
    const uint32_t v = ((l >> 3) + (uint32_t)(sub8 - lo)) < span ? vo_b + soff : OOB;
    buf_lds16(j < 3 ? rs_xr : rs_xi, v, slot_off + dst[j]);
  };
  auto e_advance = [&]() __attribute__((always_inline)) {
    p_goff = e_row + (uint32_t)e_w0 * rb_g; p_lim = e_lim;
    ++e_t;
    if (g.walk) {
      ++e_h; e_row += (uint32_t)g.Wo * rb_g;
      if (e_h == g.H) {
        e_h = 0; e_row -= (uint32_t)g.H * (uint32_t)g.Wo * rb_g; e_w0 += KR;
        if (e_w0 >= g.W) { e_w0 = 0; e_row += (uint32_t)g.Ho * (uint32_t)g.Wo * rb_g; }
      }
      e_set();
      return;
    }
    e_w0 += KR;
    if (e_w0 >= g.W) {
      e_w0 = 0;
      e_row += (uint32_t)g.Wo * rb_g;
      if (++e_h == g.H) { e_h = 0; e_row -= (uint32_t)(g.H - g.Ho) * (uint32_t)g.Wo * rb_g; }
    }
    e_set();
  };
  auto unpk = [](uint32_t d) __attribute__((always_inline)) { return f32x2{__uint_as_float(d << 16), __uint_as_float(d & 0xffff0000u)}; };
  // One (pixel, channel pair) item in pieces the stage spreads between MFMAs (a wave that does all of it in one go leaves
  // the matrix pipe idle for the length of an LDS round trip plus ~30 dependent operations, and so does its twin on the
  // SIMD, which reaches the same place at the same time): x_read issues the four LDS reads a block ahead, x_chunk(0..3)
  // are <= 8 packed operations each, one behind every MFMA of a block, x_write puts the two dwords into the G image.
  uint32_t ra = 0, rb = 0, rc = 0, rd = 0, x_dr = 0, x_di = 0;
  f32x2 x_ou = {0.f, 0.f}, x_ov = {0.f, 0.f};
  i32x4 gtile = {0, 0, 0, 0};
  auto x_read = [&](int it, const char* raw) __attribute__((always_inline)) {
    const char* rp = raw + f_img + it * 2048;
    ra = *reinterpret_cast<const uint32_t*>(rp); rb = *reinterpret_cast<const uint32_t*>(rp + KR * 128);
    rc = *reinterpret_cast<const uint32_t*>(rp + 2 * KR * 128); rd = *reinterpret_cast<const uint32_t*>(rp + 3 * KR * 128);
  };
  auto x_chunk = [&](int i, int it) __attribute__((always_inline)) {
    if (i == 0) {
      const f32x2 P = unpk(ra), Q = unpk(rb);
      x_ou = kc[0] * P + kc[7]; x_ov = kc[2] * P + kc[8];
      x_ou = kc[1] * Q + x_ou; x_ov = kc[3] * Q + x_ov;
    } else if (i == 1) {
      const f32x2 U = unpk(rc), V = unpk(rd);
      x_ou = kc[4] * U + x_ou; x_ov = kc[5] * U + x_ov;
      x_ou = kc[5] * V + x_ou; x_ov = kc[6] * V + x_ov;
    } else if (i == 2) {
      const float okf = ps + 16 * it < e_lim ? 1.f : 0.f;    // (pixels without a G: the raw rows read as zeros, G must too)
      x_ou *= okf; x_ov *= okf;
      x_dr = pack_bf16(x_ou[0], x_ou[1]); x_di = pack_bf16(x_ov[0], x_ov[1]);
    }
  };
  auto x_write = [&](int it, char* img) __attribute__((always_inline)) {
    *reinterpret_cast<uint32_t*>(img + f_img + it * 2048) = x_dr;
    *reinterpret_cast<uint32_t*>(img + KR * 128 + f_img + it * 2048) = x_di;
  };
  auto xform = [&](int it, const char* raw, char* img) __attribute__((always_inline)) {    // (prologue: all of it at once)
    x_read(it, raw);
#pragma unroll
    for (int i = 0; i < 4; ++i) x_chunk(i, it);
    x_write(it, img);
  };
  // the finished G image of the stage the MFMAs are in -> dy (the lane mapping of LDS-DMA piece 0, backwards): read ahead
  // of a block, stored behind it
  auto tile_read = [&](const char* slot) __attribute__((always_inline)) {
    gtile = *reinterpret_cast<const i32x4*>(slot + dst[0] + lane * 16);
  };
  auto tile_store = [&]() __attribute__((always_inline)) {
    const uint32_t vs = ((int)vflag[0] < p_lim && writer) ? vo[0] + p_goff : OOB;
#if defined(__HIP_DEVICE_COMPILE__)
#if CLW_NT & 1
    asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen nt" : : "v"(gtile), "v"(vs), "s"(rs_dy) : "memory");
#else
    asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen" : : "v"(gtile), "v"(vs), "s"(rs_dy) : "memory");
#endif
#endif
  };

  // ---- one stage: 2 sub-steps of 16 pixels x (4 blocks + the shared block on this wave's parity) ---------------
  // One stage = 2 sub-steps of 16 pixels x (4 blocks + the shared block on this wave's parity).  Fragments are read one
  // block ahead of the MFMAs.  The barrier sits in front of the LAST block of the stage, when every read of this slot
  // has been issued and completed: the first fragments of the next stage are then read under that block's MFMAs
  // instead of right behind a barrier at which all eight waves (both of every SIMD) would wait for the LDS together.
  bf16x8 xr[2], xi[2], gr[2], gi[2], hr, hi;
  // fill >= 0 (FOLD): x_chunk(0..3) of item `fill` behind the four MFMAs
  auto mfma_block = [&](int n, bf16x8 ar, bf16x8 ai, bf16x8 pr, bf16x8 pi, bf16x8 npr, int fill = -1) __attribute__((always_inline)) {
    // G conj(X): re = gr xr + gi xi, im = gi xr - gr xi; X first: accumulator rows = co, 4 consecutive ci per group
    auto f = [&](int i) __attribute__((always_inline)) {
      if (FOLD && fill >= 0) {
        __builtin_amdgcn_sched_barrier(0);
        x_chunk(i, fill);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    acc_r[n] = CPLXAMD_MFMA16(ar, pr, acc_r[n]); f(0);
    acc_i[n] = CPLXAMD_MFMA16(ar, pi, acc_i[n]); f(1);
    acc_r[n] = CPLXAMD_MFMA16(ai, pi, acc_r[n]); f(2);
    acc_i[n] = CPLXAMD_MFMA16(ai, npr, acc_i[n]); f(3);
  };
  auto stage = [&](uint32_t cur_off, uint32_t nxt_off, uint32_t next_cur_off, uint32_t rawsel) __attribute__((always_inline)) {
    const char* st = smem + cur_off;
    const char* sn = smem + next_cur_off;
    // FOLD: raw tiles of stage t + 2 -> buffer rawsel, G of stage t + 1 from buffer rawsel ^ 1 -> its slot
    const uint32_t raw_dma = smem_off + (uint32_t)RAW + rawsel * (uint32_t)RAW_BYTES;
    const char* raw_src = smem + RAW + (rawsel ^ 1u) * RAW_BYTES;
    char* g_img = smem + next_cur_off;
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
        if (FOLD && n != 2) {                           // LDS reads a block ahead of their use
          __builtin_amdgcn_sched_barrier(0);
          if (n == 0) tile_read(st);
          else x_read(n == 1 ? 0 : 1, raw_src);
          __builtin_amdgcn_sched_barrier(0);
        }
        mfma_block(n, xr[cur], xi[cur], gr[0], gi[0], ngr, n == 2 ? 0 : -1);
        const int piece = n == 0 ? 0 : (n == 2 ? 1 : (n == 3 ? 2 : -1));
        if (FOLD && n == 0) {
          __builtin_amdgcn_sched_barrier(0);
          issue_raw(raw_dma);
          tile_store();
          __builtin_amdgcn_sched_barrier(0);
        } else if (piece > 0) {
          __builtin_amdgcn_sched_barrier(0);
          if (FOLD && n == 2) x_write(0, g_img);
          if (FOLD) issue_piece_f(piece, nxt_off);
          else issue_piece(piece, nxt_off);
          __builtin_amdgcn_sched_barrier(0);
        } else if (piece == 0) {
          __builtin_amdgcn_sched_barrier(0);
          issue_piece(0, nxt_off);
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
        mfma_block(n, xr[cur], xi[cur], gr[1], gi[1], ngr, n == 0 ? 1 : -1);
        const int piece = n == 0 ? 3 : (n == 2 ? 4 : -1);
        if (piece >= 0) {
          __builtin_amdgcn_sched_barrier(0);
          if (FOLD && n == 0) x_write(1, g_img);
          if (FOLD) issue_piece_f(piece, nxt_off);
          else issue_piece(piece, nxt_off);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // every read of this slot has completed (FOLD: and G written)
      // the next stage has landed (this wave's pieces); FOLD: and the raw tiles issued at the head of this one
      if (FOLD) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(L) : "memory");
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
      if (FOLD) issue_raw(smem_off + (uint32_t)(RAW + s * RAW_BYTES));
#pragma unroll
      for (int j = FOLD ? 1 : 0; j < L; ++j) {
        if (FOLD) issue_piece_f(j, smem_off + (uint32_t)(s * STAGE));
        else issue_piece(j, smem_off + (uint32_t)(s * STAGE));
      }
      advance();
    }
    if (FOLD) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // both raw tiles (and both stages) landed
      __builtin_amdgcn_s_barrier();
      xform(0, smem + RAW, smem);                                    // G of stage 0 -> slot 0
      xform(1, smem + RAW, smem);
      e_advance();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(L) : "memory");       // stage 0 landed; stage 1 may be in flight
    }
    __builtin_amdgcn_s_barrier();
    gr[0] = frag_at(smem, ga, 0); gi[0] = frag_at(smem + KR * 128, ga, 0);
    xr[0] = frag_at(smem, xa[0], 0); xi[0] = frag_at(smem + XW_BYTES, xa[0], 0);
    uint32_t cur = 0, nx1 = STAGE, nx2 = 2u * STAGE;
    for (int t = 0; t < nt; ++t) {
      stage(cur, smem_off + nx2, nx1, (uint32_t)(t & 1));
      advance();
      if (FOLD) e_advance();
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
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* ws, int splits, int tiles, int tiles_ci, int sk_co, int sk_ci, int Co,
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
  int tco, tci;
  tile_of(tile, tiles_ci, sk_co, sk_ci, tco, tci);
  const int co = tco * TC + coh * 32 + col, ci = tci * TC + cih * 32 + cil;
  const int64_t i = ((int64_t)co * Ci + ci) * 9 + tap;
  float* dw = pl ? dw_i : dw_r;
  dw[i] = emul ? acc * emul[i] : acc;
}

// Stage order of a split: down the image columns, 32-pixel strip by strip (1, default), or along the image rows (0:
// CPLXAMD_CLW_WALK=0, the order of rounds 2-5).  A stage stages three x rows (one per kernel row); walking DOWN, two of the
// three were fetched by the stage before and are still in the XCD's L2 -- along a row they come back eight stages later,
// behind everything the other 31 workgroups of the XCD fetched meanwhile.  cfg3: 4.33 -> 4.09 ms (FOLD: 5.12 -> 5.01).
static int stage_walk() {
  static const int w = [] { const char* e = getenv("CPLXAMD_CLW_WALK"); return e ? (atoi(e) != 0) : 1; }();
  return w;
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

static int launch_clw(const void* g_r, const void* g_i, const void* x_r, const void* x_i, const float* emul, float* dw_r,
                      float* dw_i, int64_t B, int H, int W, int Ci, int Co, int KH, int KW, int dil_h, int dil_w, int pad_h,
                      int pad_w, int skip_co, int skip_ci, void* ws, int64_t ws_bytes, int flags, void* stream);

int cplxamd_conv2d_cl_wgrad_fl(const void* g_r, const void* g_i, const void* x_r, const void* x_i, const float* emul,
                               float* dw_r, float* dw_i, int64_t B, int H, int W, int Ci, int Co, int KH, int KW, int dil_h,
                               int dil_w, int pad_h, int pad_w, void* ws, int64_t ws_bytes, int flags, void* stream) {
  return launch_clw(g_r, g_i, x_r, x_i, emul, dw_r, dw_i, B, H, W, Ci, Co, KH, KW, dil_h, dil_w, pad_h, pad_w, 0, 0, ws, ws_bytes,
                    flags, stream);
}

#ifdef CPLXAMD_CONV_F16
// The same without the block dW[0:skip_co, 0:skip_ci] (multiples of 64; those entries of dw are left untouched): with
// G = [g1|g0] and X = [x1|x0] and skip = (Co, Ci) the product g1 x1 -- 2^-22 of the result, not part of the three piece
// products -- is not computed: 3 of 4 tiles.
int cplxamd_conv2d_clh_wgrad_skip_fl(const void* g_r, const void* g_i, const void* x_r, const void* x_i, float* dw_r, float* dw_i,
                                     int64_t B, int H, int W, int Ci, int Co, int KH, int KW, int dil_h, int dil_w, int pad_h,
                                     int pad_w, int skip_co, int skip_ci, void* ws, int64_t ws_bytes, int flags, void* stream) {
  if (skip_co < 0 || skip_ci < 0 || skip_co % 64 || skip_ci % 64 || skip_co > Co || skip_ci > Ci ||
      (skip_co == Co && skip_ci == Ci) || (skip_co == 0) != (skip_ci == 0))
    return CPLXAMD_EINVAL;
  return launch_clw(g_r, g_i, x_r, x_i, nullptr, dw_r, dw_i, B, H, W, Ci, Co, KH, KW, dil_h, dil_w, pad_h, pad_w, skip_co,
                    skip_ci, ws, ws_bytes, flags, stream);
}
#endif

static int launch_clw(const void* g_r, const void* g_i, const void* x_r, const void* x_i, const float* emul, float* dw_r,
                      float* dw_i, int64_t B, int H, int W, int Ci, int Co, int KH, int KW, int dil_h, int dil_w, int pad_h,
                      int pad_w, int skip_co, int skip_ci, void* ws, int64_t ws_bytes, int flags, void* stream) {
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
  g.sk_co = skip_co / 64; g.sk_ci = skip_ci / 64;
  g.walk = clw::stage_walk();
  const int tiles = (Co / 64) * g.tiles_ci - g.sk_co * g.sk_ci;
  // (fewer tiles, more splits each: never more slabs than the workspace of the full tile count holds -- plan() rounds the
  //  split count down to whole workgroups per tile)
  g.splits = clw::plan(g.nstages, tiles, g.per_split, !launch_owns_chip(flags));
  if ((int64_t)g.splits * tiles * clw::NBLK * 2048 * 4 > ws_bytes) return CPLXAMD_EWS;
  constexpr int smem = 3 * clw::STAGE;
  static PerDeviceOnce attr_set;
  if (const int e = set_max_dyn_lds(attr_set, clw::conv_cl_wgrad_kernel<false>, smem)) return e;
  clw::conv_cl_wgrad_kernel<false><<<dim3((unsigned)g.splits, (unsigned)tiles), clw::NT, smem, st>>>(g);
  CPLXAMD_CHECK_LAUNCH();
  clw::wgrad_reduce_kernel<<<dim3(36 * 2048 / 64, (unsigned)tiles), 256, 0, st>>>(g.ws, g.splits, tiles, g.tiles_ci, g.sk_co,
                                                                                   g.sk_ci, Co, Ci, emul, dw_r, dw_i);
  CPLXAMD_CHECK_LAUNCH();
  return 0;
}

#ifndef CPLXAMD_CONV_F16
// ---- the weight gradient that also IS the batch-norm layer's backward apply (kernel comment: FOLD) ---------------------
// g: the gradient that reached the batch-norm layer, z: that layer's input (this convolution's output), both [B][Ho][Wo][Co]
// bf16 channels-last planes; coef: [Co][12] float32 from cplxamd_bn_bwd_coef.  Writes dy = the layer's input gradient (bf16
// planes like g: what cplxamd_bn_bwd would have written, to the last bit or one bf16 rounding step) and dW = the weight
// gradient of dy against x.  ws as cplxamd_conv2d_cl_wgrad (cplxamd_conv2d_cl_wgrad_ws_bytes).

int cplxamd_conv2d_cl_wgrad_bn_fl(const void* g_r, const void* g_i, const void* z_r, const void* z_i, const float* coef,
                                  const void* x_r, const void* x_i, void* dy_r, void* dy_i, float* dw_r,
                                  float* dw_i, int64_t B, int H, int W, int Ci, int Co, int KH, int KW, int dil_h, int dil_w,
                                  int pad_h, int pad_w, void* ws, int64_t ws_bytes, int flags, void* stream) {
  if (!launch_flags_ok(flags)) return CPLXAMD_EINVAL;
  if (!g_r || !g_i || !z_r || !z_i || !coef || !x_r || !x_i || !dy_r || !dy_i || !dw_r || !dw_i || B <= 0 || H <= 0 ||
      W <= 0 || Ci <= 0 || Co <= 0)
    return CPLXAMD_EINVAL;
  if (!clw_shape_ok(B, H, W, Ci, Co, KH, KW, dil_h, dil_w, pad_h, pad_w)) return CPLXAMD_ESHAPE;
  hipStream_t st = (hipStream_t)stream;
  auto a16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  if (!a16(g_r) || !a16(g_i) || !a16(z_r) || !a16(z_i) || !a16(x_r) || !a16(x_i) || !a16(dy_r) || !a16(dy_i) || !a16(ws))
    return CPLXAMD_EALIGN;
  if (!ws || ws_bytes < cplxamd_conv2d_cl_wgrad_ws_bytes(B, H, W, Ci, Co)) return CPLXAMD_EINVAL;
  clw::Args g{};
  g.g_r = g_r; g.g_i = g_i; g.x_r = x_r; g.x_i = x_i; g.ws = (float*)ws; g.P = B * H * W;
  g.z_r = z_r; g.z_i = z_i; g.coef = coef; g.dy_r = dy_r; g.dy_i = dy_i;
  g.Ho = H + 2 * pad_h - 2 * dil_h; g.Wo = W + 2 * pad_w - 2 * dil_w;
  g.g_bytes = (uint32_t)(B * g.Ho * g.Wo * Co * 2); g.x_bytes = (uint32_t)(g.P * Ci * 2);
  g.H = H; g.W = W; g.Co = Co; g.Ci = Ci; g.dil_h = dil_h; g.dil_w = dil_w; g.pad_h = pad_h; g.pad_w = pad_w;
  g.strips = (W + clw::KR - 1) / clw::KR;
  g.nstages = (int)(B * H * g.strips);
  g.tiles_ci = Ci / 64;
  const int tiles = (Co / 64) * g.tiles_ci;
  g.splits = clw::plan(g.nstages, tiles, g.per_split, !launch_owns_chip(flags));
  g.walk = clw::stage_walk();
  constexpr int smem = clw::SMEM_FOLD;
  static PerDeviceOnce attr_set;
  if (const int e = set_max_dyn_lds(attr_set, clw::conv_cl_wgrad_kernel<true>, smem)) return e;
  clw::conv_cl_wgrad_kernel<true><<<dim3((unsigned)g.splits, (unsigned)tiles), clw::NT, smem, st>>>(g);
  CPLXAMD_CHECK_LAUNCH();
  clw::wgrad_reduce_kernel<<<dim3(36 * 2048 / 64, (unsigned)tiles), 256, 0, st>>>(g.ws, g.splits, tiles, g.tiles_ci, 0, 0, Co, Ci,
                                                                                   nullptr, dw_r, dw_i);
  CPLXAMD_CHECK_LAUNCH();
  return 0;
}
#endif

}  // extern "C"
