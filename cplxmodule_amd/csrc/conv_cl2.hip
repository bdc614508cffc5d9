// Complex 3 x 3 convolution (stride 1, dilation 1, zero padding up to `same`) on unpadded channels-last bf16
// activations, forward and data gradient: the 2-D-PATCH form of conv_cl.hip (cplx.conv2d, cplxmodule/cplx.py:770-838).
//
// conv_cl.hip stages, per (kernel row, 16-channel slice), the 512 input rows one kernel row needs -- 12 stages of
// 32 KiB per tile for 64 channels, every input row fetched three times (once per kernel row), 4 LDS-DMA pieces of
// 32-byte row slices per stage: its ablation (profiles/r02_conv_cl_ablation.txt) puts 10 % of the kernel on exactly
// that.  Here a workgroup's tile is 16 x 32 output pixels and what is staged per 16-channel slice is the 18 x 34 input
// PATCH under it: all nine taps read the same patch at a constant LDS offset (kh * 34 + kw rows), so
//   * activations are staged once per channel slice, not once per (kernel row, slice): 5 LDS-DMA pieces per 144 MFMAs
//     instead of 12, 2.5 x fewer bytes;
//   * borders need no masks in the K loop at all: a patch pixel outside the image is an out-of-range lane of the
//     `buffer_load ... lds` (zeros); the per-lane source offsets are rebuilt once per tile;
//   * fragment addresses need no arithmetic: patch row pitch x (tile row + kh) is a scalar / immediate, the column part
//     (with its bank swizzle, keyed on the patch COLUMN so that a tap shift along the row leaves it lane-constant) one
//     register per kw.
// The weights keep their own 3-slot ring of (kernel row, slice) sub-stages (12 KiB each, the packed LDS images of
// conv_cl.hip), the patches a 2-slot ring; sub-stage u = 3 slice + kernel row, so the weight slot IS the kernel row and
// everything is compile-time after unrolling 2 slices x 3 kernel rows (C / 16 even).  Persistent workgroups, ring
// through the tile boundaries, epilogue through LDS with streaming stores, bias by LDS-DMA: as conv_cl.hip.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "launch.h"   // per-call launch policy (CPLXAMD_LAUNCH_SHARED: the chip is shared with collectives)

namespace cplxamd {
namespace cl2 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int NT = 512, TH = 16, TW = 32, BN = 64;
constexpr int PH = TH + 2, PW = TW + 2, PPIX = PH * PW;       // 18 x 34 = 612 patch pixels, 32 B each per plane
constexpr int ROWB = PW * 32;                                 // bytes of one patch row
constexpr int A_PLANE = 1280 * 16;                            // 1224 chunks padded to 20 waves of 64 lanes
constexpr int A_SLOT = 2 * A_PLANE, PA = A_SLOT / 8192;       // 40 KiB = 5 LDS-DMA pieces
constexpr int W_PLANE = 3 * 64 * 32, W_BYTES = 2 * W_PLANE, PWN = 2;
constexpr int A_RING = 2 * A_SLOT, W_RING = 3 * W_BYTES;
constexpr int EPI = A_RING + W_RING, EPI_WAVE = 16 * 144 + 512;
constexpr int DUMP = EPI + 8 * EPI_WAVE, SMEM = DUMP + 4096;
// MOM (batch-norm moments of the output in the epilogue): both planes of a round staged at once -- the second plane's 16
// rows per wave behind everything else
constexpr int MOM_X = SMEM, SMEM_MOM = SMEM + 8 * 2048;
static_assert(SMEM_MOM <= 160 * 1024, "LDS");
constexpr int NST = 16;                                       // global stores per wave in the epilogue
constexpr uint32_t OOB = 0xF8000000u;

struct Args {
  const void* x_r; const void* x_i;          // [B][Hi][Wi][C] bf16
  const void* w;                             // packed weights (cplxamd_conv2d_cl_pack)
  const float* bias_r; const float* bias_i;
  void* y_r; void* y_i;                      // [B][Ho][Wo][Cout] bf16
  void* dump;
  const void* fx_r; const void* fx_i; const void* fga;   // FUSE: y += 2 fx (*) fga, all laid out like y
  double* mom;                                           // MOM: [workgroup][Cout][5] partial moments of y (bn.hip's layout)
  uint32_t x_bytes, w_bytes;
  int B, Hi, Wi, Ho, Wo, C, Cout, pad_h, pad_w;
  int C16, NS, tiles_x, tiles_y, tiles_n;
#ifdef CPLXAMD_CONV_F16     // the half-operand build (conv_cl2_f16.hip): float32 output, see epilogue_f32
  int pitch;                                  // channels between two pixels of x (>= C: a channel window of wider planes)
  int cs0, p16;                               // 16-channel slice s of the contraction is slice (cs0 + s) mod p16 of a pixel
  int accumulate;                             // y += result
  const float* scale_a; const float* scale_b; // device {s, 1 / s} of the two operands (nullptr: no scaling)
#endif
};

template <int N> __device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ bf16x8 neg_frag(bf16x8 v) {
  uint4 u = __builtin_bit_cast(uint4, v);
  u.x ^= 0x80008000u; u.y ^= 0x80008000u; u.z ^= 0x80008000u; u.w ^= 0x80008000u;
  return __builtin_bit_cast(bf16x8, u);
}

__device__ __forceinline__ void buf_lds16(i32x4 rsrc, uint32_t voff, uint32_t soff_uniform, uint32_t lds_off_uniform) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds"
               :
               : "v"(voff), "s"(rsrc), "s"(lds_off_uniform), "s"(soff_uniform)
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

enum { FL_NORMAL = 0, FL_FIRST0 = 1, FL_LAST = 2, FL_FIRST1 = 3 };

// FUSE: the input gradient of the local-reparameterization layers in one pass, y = conv(x; w) + 2 fx (*) fga per plane
// (the mean-path data gradient plus the variance path's d|x|^2 term; util.hip dx_accum_kernel's arithmetic on the
// bf16-rounded convolution result, so the fused and the two-launch forms agree to the last bit).
// MOM: the forward statistics of a batch-norm layer that consumes y (sum re, sum im, sum re^2, sum im^2, sum re im per
// output channel over the pixels inside the image, of the bf16 values as stored) leave the kernel as per-workgroup
// partials in the layout bn.hip's finalize sums (cplxamd_bn_fwd_partials): the moment pass over y -- one full read of both
// planes -- is not launched.  Persistent launch whose workgroups each stay on ONE column tile (the per-lane sums belong to
// one set of 64 channels): Cout == 64, or a grid whose per-XCD share is a multiple of the column-tile count.
template <bool FUSE, bool MOM = false>
__global__ __launch_bounds__(NT) void conv_cl2_kernel(Args g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int ntiles = g.B * g.tiles_y * g.tiles_x * g.tiles_n;
  const int nwg = gridDim.x;
  struct Tile { int b, y0, x0, nt; };
  // virtual block id -> tile: XCD x owns a contiguous range of (image, tile row, tile column, column tile) order
  auto origin = [&](int v) __attribute__((always_inline)) -> Tile {
    const int q = ntiles >> 3, r = ntiles & 7, xcd = v & 7, idx = v >> 3;
    int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    Tile t;
    int u = lin / g.tiles_n; t.nt = lin - u * g.tiles_n;
    int u2 = u / g.tiles_x; t.x0 = (u - u2 * g.tiles_x) * TW;
    int u3 = u2 / g.tiles_y; t.y0 = (u2 - u3 * g.tiles_y) * TH;
    t.b = u3;
    t.b = __builtin_amdgcn_readfirstlane(t.b); t.y0 = __builtin_amdgcn_readfirstlane(t.y0);
    t.x0 = __builtin_amdgcn_readfirstlane(t.x0); t.nt = __builtin_amdgcn_readfirstlane(t.nt);
    return t;
  };

  const int tid0 = threadIdx.x;
  const int lane = tid0 & 63;
  const int l31 = lane & 31, lk = lane >> 5;
  const uint32_t smem_off = lds_offset_of(smem);
  const uint32_t wid_u = (uint32_t)__builtin_amdgcn_readfirstlane(tid0 >> 6);
  const uint32_t wave_lds = wid_u * 1024u;

  const i32x4 rs_xr = make_rsrc(g.x_r, g.x_bytes), rs_xi = make_rsrc(g.x_i, g.x_bytes);
  const i32x4 rs_w = make_rsrc(g.w, g.w_bytes);
#ifdef CPLXAMD_CONV_F16
  const uint32_t rowbytes = (uint32_t)g.pitch * 2u;
#else
  const uint32_t rowbytes = (uint32_t)g.C * 2u;
#endif

  // ---- LDS-DMA lane constants.  Patch: chunk id = q * 512 + tid of the slot image [plane][1280 chunks]; chunk pc of a
  // plane sits at patch pixel pc >> 1 = (row pr, column px), position pc & 1, and holds channel half (pc & 1) ^ ((px >> 3)
  // & 1).  prc[q] = pr | px << 8 | half << 16 | valid << 17.
  uint32_t prc[PA], vow[PWN];
#pragma unroll
  for (int q = 0; q < PA; ++q) {
    int pc = q * NT + tid0;
    pc -= pc >= 1280 ? 1280 : 0;
    const int p = pc >> 1, pr = p / PW, px = p - pr * PW;
    prc[q] = (uint32_t)pr | ((uint32_t)px << 8) | ((uint32_t)((pc & 1) ^ ((px >> 3) & 1)) << 16) | ((p < PPIX ? 1u : 0u) << 17);
  }
#pragma unroll
  for (int j = 0; j < PWN; ++j) {
    int p = j * NT + tid0;
    if (p >= W_BYTES / 16) p -= NT;                      // surplus lanes re-read the previous piece into the dump
    vow[j] = (uint32_t)p * 16u;
  }
  // per tile: source offsets of the 5 patch pieces (image pixel of every patch pixel, or out of range)
  auto lane_offsets = [&](const Tile& t, uint32_t (&out)[PA]) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < PA; ++q) {
      const int h = t.y0 - g.pad_h + (int)(prc[q] & 255u), w = t.x0 - g.pad_w + (int)((prc[q] >> 8) & 255u);
      const bool ok = ((prc[q] >> 17) & 1u) && h >= 0 && h < g.Hi && w >= 0 && w < g.Wi;
      out[q] = ok ? (((uint32_t)t.b * (uint32_t)g.Hi + (uint32_t)h) * (uint32_t)g.Wi + (uint32_t)w) * rowbytes + ((prc[q] >> 16) & 1u) * 16u
                  : OOB;
    }
  };

  // ---- fragment addresses.  x: patch pixel (2 wave + i + kh, l31 + kw): column part per kw in a register (+ this
  // wave's first patch row), the rest immediates; w: output channel j * 32 + l31 of the packed image.
  uint32_t a_kw[3], w_rel[2];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) {
    const int px = l31 + kw;
    a_kw[kw] = (uint32_t)(px * 32 + ((lk ^ ((px >> 3) & 1)) << 4)) + wid_u * (uint32_t)(2 * ROWB);
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int co = j * 32 + l31;
    w_rel[j] = (uint32_t)(A_RING + co * 32 + ((lk ^ ((co >> 3) & 1)) << 4));
  }

  f32x16 acc_r[2][2], acc_i[2][2];
  bf16x8 ar[2][2], ai[2][2], br[2][2], bi[2][2];           // [set][block]

  // ---- tiles
  int v = blockIdx.x;
  Tile tc = origin(v);
  bool has_next = v + nwg < ntiles;
  Tile tn = origin(has_next ? v + nwg : v);
  uint32_t voa_c[PA], voa_n[PA];
  lane_offsets(tc, voa_c);
  lane_offsets(tn, voa_n);

  int c_cs = 0;                                            // channel slice the MFMAs are in
  // LDS-DMA: piece q of the patch of slice (c_cs + delta) [of the next tile when that runs past C / 16] -> patch slot
  auto dma_a = [&](int q, int delta, int aslot) __attribute__((always_inline)) {
    int cc = c_cs + delta;
    const bool nx = cc >= g.C16;
    cc -= nx ? g.C16 : 0;
#ifdef CPLXAMD_CONV_F16     // (a window that wraps around the pixel: [h0 | h1 | h0] of rows stored [h1 | h0])
    cc += g.cs0;
    cc -= cc >= g.p16 ? g.p16 : 0;
#endif
    const uint32_t voff = (nx ? voa_n[q] : voa_c[q]) + (uint32_t)cc * 32u;
    const uint32_t dst = smem_off + (uint32_t)(aslot * A_SLOT + q * 8192) + wave_lds;
    if ((uint32_t)(q * 8) + wid_u < 20u) buf_lds16(rs_xr, voff, 0u, dst);
    else buf_lds16(rs_xi, voff, 0u, dst);
  };
  // weights of (slice cs [of the next tile when past C / 16], kernel row kh) -> weight slot; the packed buffer is
  // ordered [column tile][kernel row][slice] (cplxamd_conv2d_cl_pack)
  auto dma_w = [&](int j, int cs, int kh, int wslot) __attribute__((always_inline)) {
    const bool nx = cs >= g.C16;
    const uint32_t woff = (uint32_t)((nx ? tn.nt : tc.nt) * g.NS + kh * g.C16 + (nx ? cs - g.C16 : cs)) * (uint32_t)W_BYTES;
    const bool real = (uint32_t)(j * 8) + wid_u < (uint32_t)(W_BYTES / 1024);
    const uint32_t dst = real ? smem_off + (uint32_t)(A_RING + wslot * W_BYTES + j * 8192) + wave_lds
                              : smem_off + (uint32_t)DUMP + (wave_lds & 4095u);
    buf_lds16(rs_w, vow[j], woff, dst);
  };

  // fragment idx of tap kw of sub-stage (patch slot as, kernel row kh) -> register set st
  auto read_one = [&](int st, int as, int kh, int kw, int idx) __attribute__((always_inline)) {
    const char* wp = smem + kh * W_BYTES + kw * 2048;
    const char* ap = smem + as * A_SLOT + kh * ROWB;
    if (idx == 0) br[st][0] = *reinterpret_cast<const bf16x8*>(wp + w_rel[0]);
    else if (idx == 1) bi[st][0] = *reinterpret_cast<const bf16x8*>(wp + W_PLANE + w_rel[0]);
    else if (idx == 2) ar[st][0] = *reinterpret_cast<const bf16x8*>(ap + a_kw[kw]);
    else if (idx == 3) ai[st][0] = *reinterpret_cast<const bf16x8*>(ap + A_PLANE + a_kw[kw]);
    else if (idx == 4) br[st][1] = *reinterpret_cast<const bf16x8*>(wp + w_rel[1]);
    else if (idx == 5) bi[st][1] = *reinterpret_cast<const bf16x8*>(wp + W_PLANE + w_rel[1]);
    else if (idx == 6) ar[st][1] = *reinterpret_cast<const bf16x8*>(ap + ROWB + a_kw[kw]);
    else ai[st][1] = *reinterpret_cast<const bf16x8*>(ap + ROWB + A_PLANE + a_kw[kw]);
  };

  // 16 MFMAs on register set st; behind group n two fragment reads of (patch slot ras, kernel row rkh, tap rkw) into
  // the other set; `issue` puts this tap's LDS-DMA pieces behind the groups
  auto mfma_sub = [&](int st, int ras, int rkh, int rkw, bool do_read, auto issue) __attribute__((always_inline)) {
    bf16x8 nai[2];
#ifndef CPLXAMD_CL2_ORD1   // first products of all four blocks, then the second ones (8 groups, one fragment read each): -0.7 % against four blocks x four products
#pragma unroll
    for (int ph = 0; ph < 2; ++ph)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (ph == 1 && j == 0) nai[i] = neg_frag(ai[st][i]);
          if (ph == 0) {
            acc_r[i][j] = CPLXAMD_MFMA16(br[st][j], ar[st][i], acc_r[i][j]);
            acc_i[i][j] = CPLXAMD_MFMA16(br[st][j], ai[st][i], acc_i[i][j]);
          } else {
            acc_r[i][j] = CPLXAMD_MFMA16(bi[st][j], nai[i], acc_r[i][j]);
            acc_i[i][j] = CPLXAMD_MFMA16(bi[st][j], ar[st][i], acc_i[i][j]);
          }
          const int g8 = ph * 4 + i * 2 + j;
          __builtin_amdgcn_sched_barrier(0);
          if (do_read) read_one(st ^ 1, ras, rkh, rkw, g8);
          __builtin_amdgcn_sched_barrier(0);
          if (g8 & 1) issue(g8 >> 1);
          __builtin_amdgcn_sched_barrier(0);
        }
    return;
#endif
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (j == 0) nai[i] = neg_frag(ai[st][i]);
        acc_r[i][j] = CPLXAMD_MFMA16(br[st][j], ar[st][i], acc_r[i][j]);
        acc_i[i][j] = CPLXAMD_MFMA16(br[st][j], ai[st][i], acc_i[i][j]);
        acc_r[i][j] = CPLXAMD_MFMA16(bi[st][j], nai[i], acc_r[i][j]);
        acc_i[i][j] = CPLXAMD_MFMA16(bi[st][j], ar[st][i], acc_i[i][j]);
        const int grp = i * 2 + j;
        __builtin_amdgcn_sched_barrier(0);
        if (do_read) {
          read_one(st ^ 1, ras, rkh, rkw, 2 * grp);
          read_one(st ^ 1, ras, rkh, rkw, 2 * grp + 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        issue(grp);
        __builtin_amdgcn_sched_barrier(0);
      }
  };
  auto none = [&](int) __attribute__((always_inline)) {};

  // One sub-stage (slice c_cs, kernel row KH = weight slot, patch slot AS): tap 0 | tap 1 | [all reads of this weight
  // slot done, next sub-stage landed: barrier] | tap 2.  LDS-DMA schedule (patch of slice c: piece 0 behind tap 2 of
  // (c - 2, kernel row 2), pieces 1, 2 behind tap 0 of (c - 1, 0), pieces 3, 4 behind tap 0 of (c - 1, 1); weights of
  // sub-stage u + 2 behind tap 1 of u) -- four or five pieces per sub-stage, everything a sub-stage needs issued at
  // least one sub-stage before its barrier.
  auto body = [&](auto KH, auto ASLOT, auto PARITY, auto FLAVOUR) __attribute__((always_inline)) {
    constexpr int kh = decltype(KH)::value, as = decltype(ASLOT)::value, par = decltype(PARITY)::value;
    constexpr int fl = decltype(FLAVOUR)::value;
    constexpr bool first = fl == FL_FIRST0 || fl == FL_FIRST1;      // (their pieces went out before the stores)
    if ((kh == 0 || kh == 1) && !first)
      mfma_sub(par, as, kh, 1, true, [&](int grp) __attribute__((always_inline)) {
        if (grp == 1) dma_a(kh == 0 ? 1 : 3, 1, as ^ 1);
        if (grp == 3) dma_a(kh == 0 ? 2 : 4, 1, as ^ 1);
      });
    else
      mfma_sub(par, as, kh, 1, true, none);
    if (fl != FL_FIRST0)
      mfma_sub(par ^ 1, as, kh, 2, true, [&](int grp) __attribute__((always_inline)) {
        if (grp == 1) dma_w(0, c_cs + (kh + 2) / 3, (kh + 2) % 3, (kh + 2) % 3);      // sub-stage u + 2
        if (grp == 3) dma_w(1, c_cs + (kh + 2) / 3, (kh + 2) % 3, (kh + 2) % 3);
      });
    else
      mfma_sub(par ^ 1, as, kh, 2, true, none);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // pieces issued since the last piece of sub-stage u + 1's weights: kernel row 0: 1 + 2 + 2, row 1: 2 + 2, row 2: 2
    if (fl == FL_FIRST1) wait_vmcnt<2 + NST>();
    else if (fl != FL_FIRST0) wait_vmcnt<kh == 0 ? 5 : (kh == 1 ? 4 : 2)>();
    __builtin_amdgcn_s_barrier();
    constexpr int nkh = (kh + 1) % 3, nas = kh == 2 ? (as ^ 1) : as;
    if (fl == FL_LAST) {
      // everything the next tile needs before the first store-inclusive wait goes out BEFORE the stores: the whole
      // patch of its slice 1 (-> this patch slot) and the weights of its sub-stage 2 (-> this weight slot)
      mfma_sub(par, nas, nkh, 0, false, [&](int grp) __attribute__((always_inline)) {
        if (grp == 0) { dma_a(0, 2, as); dma_a(1, 2, as); }
        if (grp == 1) { dma_a(2, 2, as); dma_a(3, 2, as); }
        if (grp == 2) { dma_a(4, 2, as); dma_w(0, c_cs + 1, 2, 2); }                  // next tile: slice 0, kernel row 2
        if (grp == 3) dma_w(1, c_cs + 1, 2, 2);
      });
    } else if (kh == 2) {
      mfma_sub(par, nas, nkh, 0, true, [&](int grp) __attribute__((always_inline)) {
        if (grp == 1) dma_a(0, 2, as);
      });
    } else {
      mfma_sub(par, nas, nkh, 0, true, none);
    }
    if (kh == 2) ++c_cs;
  };
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;

  auto opaque_tid = [&]() __attribute__((always_inline)) -> int {
    int t = (int)threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
  };

  // ---- epilogue: wave w owns tile rows 2w, 2w + 1 (blocks i = 0, 1) x 32 pixels x 64 channels x {re, im}; 16 pixels x
  // 128 bytes per round through this wave's LDS rows; NST unpredicated streaming stores (pixels beyond the image edge
  // go to the dump buffer)
  auto epilogue = [&]() __attribute__((always_inline)) {
    const int t = opaque_tid();
    const int ln = t & 63, w_ = t >> 6, q31 = ln & 31, qk = ln >> 5;
    char* reg = smem + EPI + w_ * EPI_WAVE;
    // Staging rows are 128 B with no padding; conflict-free both ways (MI355X_MICROARCH.md, LDS lane groups): the 8-byte
    // slot a lane writes is XORed with its row (16 lanes of a ds_write_b64 group = 16 rows, same slot -> 16 distinct
    // bank pairs), so the 16-byte slot c of row r sits at c ^ (r >> 1) with its halves swapped for odd r; the
    // ds_read_b128 groups {0-3, 12-15, 20-27} ... then cover 16 distinct slots.  (Pitch 144 without the XOR: 2-way on
    // every write, 2-way on 3 of 16 slots of a read -- 11 % of the kernel's LDS cycles were conflict cycles.)
    constexpr int PITCH = 128;
    const int r16 = q31 >> 4, rr = q31 & 15;
    const int64_t ldc = g.Cout;
    if constexpr (FUSE) {
      // per (tile row i, 16-pixel half): the operands of its four stores -- 2 x (fga, fx_r, fx_i) at the lanes' own
      // store positions -- are requested together before the staging (vmcnt retires loads and stores in issue order:
      // one store-acknowledgement wait per batch, as in gemm_bf16_persist.h)
      const uint32_t loff = (uint32_t)(((ln >> 3) * (int)ldc + (ln & 7) * 8) * 2);
#pragma unroll
      for (int hb = 0; hb < 4; ++hb) {
        const int i = hb >> 1, half = hb & 1;
        const int y = tc.y0 + 2 * w_ + i;
        uint4 lga[2], lxr[2], lxi[2];
        bool okl[2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
          const int xb = tc.x0 + half * 16 + sub * 8, x = xb + (ln >> 3);
          okl[sub] = y < g.Ho && x < g.Wo;
          // lanes beyond the image edge read the batch's first pixel (their results go to the dump buffer)
          const int64_t ub = (y < g.Ho && xb < g.Wo) ? ((((int64_t)tc.b * g.Ho + y) * g.Wo + xb) * ldc + tc.nt * BN) * 2 : 0;
          const uint32_t ulo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ub);
          const uint32_t uhi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)ub >> 32));
          const int64_t u = (int64_t)(((uint64_t)uhi << 32) | ulo);
          const uint32_t lo = okl[sub] ? loff : 0u;
          lga[sub] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(g.fga) + u + lo);
          lxr[sub] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(g.fx_r) + u + lo);
          lxi[sub] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(g.fx_i) + u + lo);
        }
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
          bf16_t* out = reinterpret_cast<bf16_t*>(pl ? g.y_i : g.y_r);
          if (r16 == half) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                f4 x;
#pragma unroll
                for (int e = 0; e < 4; ++e) x.v[e] = pl ? acc_i[i][j][4 * q + e] : acc_r[i][j][4 * q + e];
                st4(reinterpret_cast<bf16_t*>(reg + rr * PITCH + (((8 * j + 2 * q + qk) ^ rr) << 3)), x);
              }
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
          for (int sub = 0; sub < 2; ++sub) {
            const int sr = sub * 8 + (ln >> 3);
            const uint4 raw = *reinterpret_cast<const uint4*>(reg + sr * PITCH + (((ln & 7) ^ (sr >> 1)) << 4));
            const bool odd = (ln >> 3) & 1;
            const uint4 val = odd ? uint4{raw.z, raw.w, raw.x, raw.y} : raw;
            const uint4 xv = pl ? lxi[sub] : lxr[sub];
            const uint32_t vw[4] = {val.x, val.y, val.z, val.w}, xw[4] = {xv.x, xv.y, xv.z, xv.w};
            const uint32_t gw[4] = {lga[sub].x, lga[sub].y, lga[sub].z, lga[sub].w};
            typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
            u32x4_t ow;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float d0 = __uint_as_float(vw[e] << 16), d1 = __uint_as_float(vw[e] & 0xffff0000u);
              const float x0 = __uint_as_float(xw[e] << 16), x1 = __uint_as_float(xw[e] & 0xffff0000u);
              const float g0 = __uint_as_float(gw[e] << 16), g1 = __uint_as_float(gw[e] & 0xffff0000u);
              ow[e] = pack_bf16(fmaf(2.0f * x0, g0, d0), fmaf(2.0f * x1, g1, d1));
            }
            const int x = tc.x0 + half * 16 + sub * 8 + (ln >> 3);
            const int col = tc.nt * BN + (ln & 7) * 8;
            bf16_t* dst = okl[sub] ? out + (((int64_t)tc.b * g.Ho + y) * g.Wo + x) * ldc + col
                                   : reinterpret_cast<bf16_t*>(g.dump) + (int64_t)(w_ * 64 + i * 32 + half * 16 + sub * 8 + (ln >> 3)) * ldc + col;
            __builtin_nontemporal_store(ow, reinterpret_cast<u32x4_t*>(dst));
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
      }
      return;
    }
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
      bf16_t* out = reinterpret_cast<bf16_t*>(pl ? g.y_i : g.y_r);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          if (r16 == half) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                f4 x;
#pragma unroll
                for (int e = 0; e < 4; ++e) x.v[e] = pl ? acc_i[i][j][4 * q + e] : acc_r[i][j][4 * q + e];
                st4(reinterpret_cast<bf16_t*>(reg + rr * PITCH + (((8 * j + 2 * q + qk) ^ rr) << 3)), x);
              }
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
          for (int sub = 0; sub < 2; ++sub) {
            const int sr = sub * 8 + (ln >> 3);
            const uint4 raw = *reinterpret_cast<const uint4*>(reg + sr * PITCH + (((ln & 7) ^ (sr >> 1)) << 4));
            const bool odd = (ln >> 3) & 1;
            const uint4 val = odd ? uint4{raw.z, raw.w, raw.x, raw.y} : raw;
            const int y = tc.y0 + 2 * w_ + i, x = tc.x0 + half * 16 + sub * 8 + (ln >> 3);
            const int col = tc.nt * BN + (ln & 7) * 8;
            const bool ok = y < g.Ho && x < g.Wo;
            bf16_t* dst = ok ? out + (((int64_t)tc.b * g.Ho + y) * g.Wo + x) * ldc + col
                             : reinterpret_cast<bf16_t*>(g.dump) + (int64_t)(w_ * 64 + i * 32 + half * 16 + sub * 8 + (ln >> 3)) * ldc + col;
            typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
            __builtin_nontemporal_store(u32x4_t{val.x, val.y, val.z, val.w}, reinterpret_cast<u32x4_t*>(dst));
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
  };

#ifdef CPLXAMD_CONV_F16
  // ---- float32 epilogue of the half-operand build: straight from the MFMA C layout -- lane (pixel q31 of the tile row,
  // column group qk) holds, per (tile row i, 32-channel block j, group q), four consecutive channels: one 16-byte store
  // each (a store instruction covers 32 pixels x 32 bytes; the four q groups of a block complete the 128-byte lines in L2),
  // times 1 / (sa sb), plus the previous value when accumulating; pixels beyond the image edge are skipped.
  auto epilogue_f32 = [&]() __attribute__((always_inline)) {
    const int t = opaque_tid();
    const int ln = t & 63, w_ = t >> 6, q31 = ln & 31, qk = ln >> 5;
    const float alpha = g.scale_a ? g.scale_a[1] * g.scale_b[1] : 1.0f;
    const int64_t ldc = g.Cout;
    const int x = tc.x0 + q31;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int y = tc.y0 + 2 * w_ + i;
      if (y < g.Ho && x < g.Wo) {
        const int64_t base = (((int64_t)tc.b * g.Ho + y) * g.Wo + x) * ldc + tc.nt * BN + 4 * qk;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
          float* out = reinterpret_cast<float*>(pl ? g.y_i : g.y_r) + base;
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              f4 v;
#pragma unroll
              for (int e = 0; e < 4; ++e) v.v[e] = (pl ? acc_i[i][j][4 * q + e] : acc_r[i][j][4 * q + e]) * alpha;
              float* dst = out + j * 32 + 8 * q;
              if (g.accumulate) {
                const f4 o = ld4(dst);
#pragma unroll
                for (int e = 0; e < 4; ++e) v.v[e] += o.v[e];
              }
              st4(dst, v);
            }
        }
      }
    }
  };
#endif

  // ---- MOM epilogue: the same stores, both planes of a round staged at once (the second one in this wave's MOM_X rows),
  // and a second, column-wise read of the staged bf16 rows: lane = (channel pair cp = lane & 31, row group rg = lane >> 5),
  // 8 rows x 2 planes x 4 bytes -- the two row groups take rows of opposite parity in every read, i.e. opposite halves of
  // the 64 banks (the row pitch is half of them; within a row the XOR swizzle keeps the 32 channel pairs on 32 banks).
  // Ten per-lane sums live through the persistent loop; pixels outside the image contribute zeros.
  typedef float f2v __attribute__((ext_vector_type(2)));
  f2v m_r = {0.f, 0.f}, m_i = {0.f, 0.f}, m_rr = {0.f, 0.f}, m_ii = {0.f, 0.f}, m_ri = {0.f, 0.f};
  auto epilogue_mom = [&]() __attribute__((always_inline)) {
    const int t = opaque_tid();
    const int ln = t & 63, w_ = t >> 6, q31 = ln & 31, qk = ln >> 5;
    char* reg0 = smem + EPI + w_ * EPI_WAVE;
    char* reg1 = smem + MOM_X + w_ * 2048;
    constexpr int PITCH = 128;
    const int r16 = q31 >> 4, rr = q31 & 15;
    const int64_t ldc = g.Cout;
    const int cp = ln & 31, rg = ln >> 5;
    // row r = 8 rg + (k ^ rg) of read k: every field of its byte offset -- r << 7, ((cp >> 1) ^ r) << 3, (cp & 1) << 2 --
    // is an XOR of a lane part and a k part, and the fields do not overlap: offset = lane part ^ (k << 7 | k << 3)
    const uint32_t mbase = (uint32_t)((rg << 10) ^ (rg << 7) ^ ((((cp >> 1) ^ (rg << 3) ^ rg) & 15) << 3) ^ ((cp & 1) << 2));
    const bool interior = __builtin_amdgcn_readfirstlane((int)(tc.x0 + TW <= g.Wo && tc.y0 + TH <= g.Ho)) != 0;
    auto moments_of = [&](uint32_t dr, uint32_t di) __attribute__((always_inline)) {
      const f2v R = {__uint_as_float(dr << 16), __uint_as_float(dr & 0xffff0000u)};
      const f2v I = {__uint_as_float(di << 16), __uint_as_float(di & 0xffff0000u)};
      m_r += R; m_i += I;
      m_rr += R * R; m_ii += I * I; m_ri += R * I;
    };
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        if (r16 == half) {
#pragma unroll
          for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                f4 x;
#pragma unroll
                for (int e = 0; e < 4; ++e) x.v[e] = pl ? acc_i[i][j][4 * q + e] : acc_r[i][j][4 * q + e];
                st4(reinterpret_cast<bf16_t*>((pl ? reg1 : reg0) + rr * PITCH + (((8 * j + 2 * q + qk) ^ rr) << 3)), x);
              }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int y = tc.y0 + 2 * w_ + i;
        // every LDS read of the round first -- the four 16-byte rows of the stores, then the moment lanes' sixteen dwords --
        // so that the moment reads land while the stores' addresses are formed and the stores issue
        uint4 raw[2][2];
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
          for (int sub = 0; sub < 2; ++sub) {
            const int sr = sub * 8 + (ln >> 3);
            raw[pl][sub] = *reinterpret_cast<const uint4*>((pl ? reg1 : reg0) + sr * PITCH + (((ln & 7) ^ (sr >> 1)) << 4));
          }
        const bool rowin = interior || __builtin_amdgcn_readfirstlane((int)(y < g.Ho)) != 0;
        uint32_t dr[8], di[8];
        if (rowin) {
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const uint32_t off = mbase ^ (uint32_t)((k << 7) | (k << 3));
            dr[k] = *reinterpret_cast<const uint32_t*>(reg0 + off);
            di[k] = *reinterpret_cast<const uint32_t*>(reg1 + off);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
          bf16_t* out = reinterpret_cast<bf16_t*>(pl ? g.y_i : g.y_r);
#pragma unroll
          for (int sub = 0; sub < 2; ++sub) {
            const uint4 rw = raw[pl][sub];
            const bool odd = (ln >> 3) & 1;
            const uint4 val = odd ? uint4{rw.z, rw.w, rw.x, rw.y} : rw;
            const int x = tc.x0 + half * 16 + sub * 8 + (ln >> 3);
            const int col = tc.nt * BN + (ln & 7) * 8;
            const bool ok = y < g.Ho && x < g.Wo;
            bf16_t* dst = ok ? out + (((int64_t)tc.b * g.Ho + y) * g.Wo + x) * ldc + col
                             : reinterpret_cast<bf16_t*>(g.dump) + (int64_t)(w_ * 64 + i * 32 + half * 16 + sub * 8 + (ln >> 3)) * ldc + col;
            typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
            __builtin_nontemporal_store(u32x4_t{val.x, val.y, val.z, val.w}, reinterpret_cast<u32x4_t*>(dst));
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (rowin) {
          if (interior) {                                        // (scalar branch: most tiles)
#pragma unroll
            for (int k = 0; k < 8; ++k) moments_of(dr[k], di[k]);
          } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              const bool ok = tc.x0 + half * 16 + 8 * rg + (k ^ rg) < g.Wo;
              moments_of(ok ? dr[k] : 0u, ok ? di[k] : 0u);
            }
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
  };

  const uint32_t bias_lds = smem_off + (uint32_t)EPI + wid_u * (uint32_t)EPI_WAVE + 2304u;
  auto bias_dma = [&](int nt_) __attribute__((always_inline)) {
    const int t = opaque_tid();
    const int ln = t & 63;
    if (ln < 32) {
      const float* src = g.bias_r ? ((ln < 16 ? g.bias_r : g.bias_i) + nt_ * BN + 4 * (ln & 15))
                                  : reinterpret_cast<const float*>(g.w) + 4 * ln;
      lds_dma16_at(src, bias_lds);
    }
  };
  auto init_acc = [&]() __attribute__((always_inline)) {
    const int t = opaque_tid();
    const int w_ = t >> 6, qk = (t & 63) >> 5;
    const char* breg = smem + EPI + w_ * EPI_WAVE + 2304;
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f4 b = {{0.f, 0.f, 0.f, 0.f}};
          if (g.bias_r) b = ld4(reinterpret_cast<const float*>(breg + pl * 256 + (j * 32 + 8 * q + 4 * qk) * 4));
#ifdef CPLXAMD_CONV_F16     // (the accumulators are multiplied by 1 / (sa sb) in the epilogue: the bias starts as bias sa sb)
          if (g.bias_r && g.scale_a) {
            const float inv = g.scale_a[0] * g.scale_b[0];
#pragma unroll
            for (int e = 0; e < 4; ++e) b.v[e] *= inv;
          }
#endif
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              if (pl) acc_i[i][j][4 * q + e] = b.v[e];
              else acc_r[i][j][4 * q + e] = b.v[e];
            }
        }
  };
  auto first_frags = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int idx = 0; idx < 8; ++idx) read_one(0, 0, 0, 0, idx);
  };

  // ---- prologue of the first tile: patches of slices 0, 1 and the weights of sub-stages 0, 1, 2, whole (the state
  // every later tile starts from)
  bias_dma(tc.nt);
#pragma unroll
  for (int q = 0; q < PA; ++q) { dma_a(q, 0, 0); dma_a(q, 1, 1); }
#pragma unroll
  for (int s = 0; s < 3; ++s) { dma_w(0, 0, s, s); dma_w(1, 0, s, s); }
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  init_acc();
  first_frags();

  const int npairs = (g.C16 - 2) / 2;                      // pairs of channel slices between the first and the last
  for (;;) {
    c_cs = 0;
    body(I0{}, I0{}, I0{}, I1{});                          // FIRST0   (slice 0: patch slot 0)
    body(I1{}, I0{}, I1{}, I3{});                          // FIRST1
    body(I2{}, I0{}, I0{}, I0{});
    if (g.C16 > 2) {
      body(I0{}, I1{}, I1{}, I0{}); body(I1{}, I1{}, I0{}, I0{}); body(I2{}, I1{}, I1{}, I0{});
      for (int pr = 1; pr < npairs; ++pr) {
        body(I0{}, I0{}, I0{}, I0{}); body(I1{}, I0{}, I1{}, I0{}); body(I2{}, I0{}, I0{}, I0{});
        body(I0{}, I1{}, I1{}, I0{}); body(I1{}, I1{}, I0{}, I0{}); body(I2{}, I1{}, I1{}, I0{});
      }
      body(I0{}, I0{}, I0{}, I0{}); body(I1{}, I0{}, I1{}, I0{}); body(I2{}, I0{}, I0{}, I0{});
    }
    body(I0{}, I1{}, I1{}, I0{});
    body(I1{}, I1{}, I0{}, I0{});
    body(I2{}, I1{}, I1{}, I2{});                          // LAST     (slice C / 16 - 1: patch slot 1)
    bias_dma(has_next ? tn.nt : tc.nt);                    // (older than the stores below)
#ifdef CPLXAMD_CONV_F16
    epilogue_f32();
    if (!has_next) break;
    wait_vmcnt<0>();                                       // (a lane-dependent number of stores: no counted wait)
#else
    if constexpr (MOM) epilogue_mom(); else epilogue();
    if (!has_next) break;
    wait_vmcnt<NST>();                                     // everything issued BEFORE the stores has landed
#endif
    v += nwg;
    tc = tn;
    has_next = v + nwg < ntiles;
    tn = origin(has_next ? v + nwg : v);
#pragma unroll
    for (int q = 0; q < PA; ++q) voa_c[q] = voa_n[q];
    lane_offsets(tn, voa_n);
    init_acc();
    first_frags();
  }
  wait_vmcnt<0>();
  if constexpr (MOM) {
    // per workgroup: row groups (lane ^ 32), then the eight waves through LDS (the rings are idle: nothing in flight)
    const int t = opaque_tid();
    const int ln = t & 63, w_ = t >> 6, cp = ln & 31;
    float vals[10] = {m_r.x, m_r.y, m_i.x, m_i.y, m_rr.x, m_rr.y, m_ii.x, m_ii.y, m_ri.x, m_ri.y};
#pragma unroll
    for (int k = 0; k < 10; ++k) vals[k] += __shfl_xor(vals[k], 32);
    __builtin_amdgcn_s_barrier();
    float* red = reinterpret_cast<float*>(smem);
    if (ln < 32) {
#pragma unroll
      for (int k = 0; k < 10; ++k) red[(w_ * 32 + cp) * 10 + k] = vals[k];
    }
    __syncthreads();
    if (t < 5 * BN) {
      const int ch = t & (BN - 1), k = t / BN;
      double sum = 0.0;
#pragma unroll
      for (int w = 0; w < NT / 64; ++w) sum += (double)red[(w * 32 + (ch >> 1)) * 10 + 2 * k + (ch & 1)];
      // (the row was zeroed by the launcher: the other column tiles' channels of this workgroup's row stay 0)
      g.mom[((int64_t)blockIdx.x * g.Cout + tc.nt * BN + ch) * 5 + k] = sum;
    }
  }
}

}  // namespace cl2
}  // namespace cplxamd

using namespace cplxamd;

extern "C" {

int64_t cplxamd_conv2d_cl_pack_bytes(int N, int C, int KH, int KW);
int64_t cplxamd_conv2d_cl_ws_bytes(int Cout);

static int cl2_cus() { return device_cus() & ~7; }

// Does every workgroup of a `grid`-wide launch over `ntiles` tiles see ONE column tile?  Virtual tile v of workgroup w is
// w, w + grid, ...; its linear index is base(v & 7) + (v >> 3) with the column tile fastest (conv_cl2_kernel: origin), so
// a step of `grid` moves it by grid / 8: the column tile repeats when that is a multiple of the column-tile count.
static bool cl2_mom_tiling_ok(int64_t ntiles, int grid, int tiles_n) {
  if (tiles_n == 1 || ntiles <= grid) return true;
  return (grid % 8) == 0 && ((grid / 8) % tiles_n) == 0;
}

#ifdef CPLXAMD_CONV_F16
// The half-operand build: cplxamd_conv2d_cl2 on IEEE-half planes with float32 output (include/cplxamd.h).
static int launch_cl2h(const void* x_r, const void* x_i, int pitch, int c_start, const void* w_packed, const float* bias_r,
                       const float* bias_i, float* y_r, float* y_i, int accumulate, const float* scale_a,
                       const float* scale_b, int64_t B, int H, int W, int C, int N, int pad_h, int pad_w, int mode, void* ws,
                       int64_t ws_bytes, int flags, void* stream);

int cplxamd_conv2d_cl2h_fl(const void* x_r, const void* x_i, int pitch, const void* w_packed, const float* bias_r,
                           const float* bias_i, float* y_r, float* y_i, int accumulate, const float* scale_a,
                           const float* scale_b, int64_t B, int H, int W, int C, int N, int pad_h, int pad_w, int mode, void* ws,
                           int64_t ws_bytes, int flags, void* stream) {
  return launch_cl2h(x_r, x_i, pitch, -1, w_packed, bias_r, bias_i, y_r, y_i, accumulate, scale_a, scale_b, B, H, W, C, N, pad_h,
                     pad_w, mode, ws, ws_bytes, flags, stream);
}

// The same with a contraction window that WRAPS around the pixel: channel j of the contraction is channel
// (c_start + j) mod pitch of the pixel, C <= 2 pitch - c_start.  With rows stored [h1 | h0] (pitch = 2 c), c_start = c and
// C = 3 c read [h0 | h1 | h0]: against weights packed [w1 | w0 | w0] that is all three piece products of a float32
// convolution in ONE launch (one float32 epilogue, no accumulate pass).
int cplxamd_conv2d_cl2h_wrap_fl(const void* x_r, const void* x_i, int pitch, int c_start, const void* w_packed,
                                const float* bias_r, const float* bias_i, float* y_r, float* y_i, int accumulate,
                                const float* scale_a, const float* scale_b, int64_t B, int H, int W, int C, int N, int pad_h,
                                int pad_w, int mode, void* ws, int64_t ws_bytes, int flags, void* stream) {
  if (c_start < 0 || c_start >= pitch || c_start % 16 || pitch % 16 || C + c_start > 2 * pitch) return CPLXAMD_ESHAPE;
  return launch_cl2h(x_r, x_i, pitch, c_start, w_packed, bias_r, bias_i, y_r, y_i, accumulate, scale_a, scale_b, B, H, W, C, N,
                     pad_h, pad_w, mode, ws, ws_bytes, flags, stream);
}

static int launch_cl2h(const void* x_r, const void* x_i, int pitch, int c_start, const void* w_packed, const float* bias_r,
                       const float* bias_i, float* y_r, float* y_i, int accumulate, const float* scale_a,
                       const float* scale_b, int64_t B, int H, int W, int C, int N, int pad_h, int pad_w, int mode, void* ws,
                       int64_t ws_bytes, int flags, void* stream) {
  if (!launch_flags_ok(flags)) return CPLXAMD_EINVAL;
  if (!x_r || !x_i || !w_packed || !y_r || !y_i || B < 0 || H <= 0 || W <= 0 || C <= 0 || N <= 0 || pad_h < 0 || pad_w < 0 ||
      (bias_r == nullptr) != (bias_i == nullptr) || (mode != 0 && mode != 1) || (scale_a == nullptr) != (scale_b == nullptr))
    return CPLXAMD_EINVAL;
  const int Hs = H + 2 * pad_h - 2, Ws = W + 2 * pad_w - 2;           // the smaller image
  if (C % 32 || N % 64 || (c_start < 0 && pitch < C) || pitch % 8 || Hs <= 0 || Ws <= 0 || Hs > H || Ws > W) return CPLXAMD_ESHAPE;
  if (B == 0) return 0;
  cl2::Args g{};
  g.Hi = mode ? Hs : H; g.Wi = mode ? Ws : W; g.Ho = mode ? H : Hs; g.Wo = mode ? W : Ws;
  if (B * g.Hi * g.Wi * pitch * 2 >= (int64_t)0xF0000000 || B >= 65536) return CPLXAMD_ESHAPE;
  auto a16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  if (!a16(x_r) || !a16(x_i) || !a16(w_packed) || !a16(y_r) || !a16(y_i) || !a16(ws) || (bias_r && (!a16(bias_r) || !a16(bias_i))))
    return CPLXAMD_EALIGN;
  if (!ws || ws_bytes < cplxamd_conv2d_cl_ws_bytes(N)) return CPLXAMD_EINVAL;
  g.x_r = x_r; g.x_i = x_i; g.w = w_packed; g.bias_r = bias_r; g.bias_i = bias_i; g.y_r = y_r; g.y_i = y_i; g.dump = ws;
  g.B = (int)B;
  g.pad_h = mode ? 2 - pad_h : pad_h; g.pad_w = mode ? 2 - pad_w : pad_w;
  g.x_bytes = c_start >= 0 ? (uint32_t)(B * g.Hi * g.Wi * pitch * 2)          // (a wrapping window stays inside its pixel)
                           : (uint32_t)(B * g.Hi * g.Wi * pitch * 2 - (pitch - C) * 2);     // (the window ends C channels into the last pixel)
  g.cs0 = c_start >= 0 ? c_start / 16 : 0; g.p16 = c_start >= 0 ? pitch / 16 : 0x7fffffff;
  g.w_bytes = (uint32_t)cplxamd_conv2d_cl_pack_bytes(N, C, 3, 3);
  g.C = C; g.Cout = N; g.C16 = C / 16; g.NS = 3 * g.C16;
  g.pitch = pitch; g.accumulate = accumulate ? 1 : 0; g.scale_a = scale_a; g.scale_b = scale_b;
  g.tiles_x = (g.Wo + cl2::TW - 1) / cl2::TW; g.tiles_y = (g.Ho + cl2::TH - 1) / cl2::TH; g.tiles_n = N / 64;
  const int64_t ntiles = B * g.tiles_x * g.tiles_y * g.tiles_n;
  if (ntiles > 0x7fffffff) return CPLXAMD_ESHAPE;
  const int ncu = cl2_cus();
  const int grid = (ntiles < ncu || !launch_owns_chip(flags)) ? (int)ntiles : ncu;
  static PerDeviceOnce attr_set;
  if (const int e = set_max_dyn_lds(attr_set, cl2::conv_cl2_kernel<false>, cl2::SMEM)) return e;
  cl2::conv_cl2_kernel<false><<<dim3((unsigned)grid), cl2::NT, cl2::SMEM, (hipStream_t)stream>>>(g);
  CPLXAMD_CHECK_LAUNCH();
  return 0;
}
#else
static int launch_cl2(const void* x_r, const void* x_i, const void* w_packed, const float* bias_r, const float* bias_i,
                      const void* fx_r, const void* fx_i, const void* fga, void* y_r, void* y_i, int64_t B, int H, int W, int C,
                      int N, int KH, int KW, int dil_h, int dil_w, int pad_h, int pad_w, int mode, void* ws, int64_t ws_bytes,
                      int flags, void* stream, double* mom = nullptr, int64_t mom_bytes = 0) {
  if (!launch_flags_ok(flags)) return CPLXAMD_EINVAL;
  if (!x_r || !x_i || !w_packed || !y_r || !y_i || B < 0 || H <= 0 || W <= 0 || C <= 0 || N <= 0 || pad_h < 0 || pad_w < 0 ||
      (bias_r == nullptr) != (bias_i == nullptr) || (mode != 0 && mode != 1))
    return CPLXAMD_EINVAL;
  const int Hs = H + 2 * pad_h - 2, Ws = W + 2 * pad_w - 2;           // the smaller image
  if (KH != 3 || KW != 3 || dil_h != 1 || dil_w != 1 || C % 32 || N % 64 || Hs <= 0 || Ws <= 0 || Hs > H || Ws > W)
    return CPLXAMD_ESHAPE;
  if (B == 0) return 0;
  if (B * H * W * C * 2 >= (int64_t)0xF0000000 || B >= 65536) return CPLXAMD_ESHAPE;
  auto a16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  if (!a16(x_r) || !a16(x_i) || !a16(w_packed) || !a16(y_r) || !a16(y_i) || !a16(ws) || (bias_r && (!a16(bias_r) || !a16(bias_i))) ||
      !a16(fx_r) || !a16(fx_i) || !a16(fga))
    return CPLXAMD_EALIGN;
  if (!ws || ws_bytes < cplxamd_conv2d_cl_ws_bytes(N)) return CPLXAMD_EINVAL;
  cl2::Args g{};
  g.x_r = x_r; g.x_i = x_i; g.w = w_packed; g.bias_r = bias_r; g.bias_i = bias_i; g.y_r = y_r; g.y_i = y_i; g.dump = ws;
  g.fx_r = fx_r; g.fx_i = fx_i; g.fga = fga;
  g.B = (int)B;
  g.Hi = mode ? Hs : H; g.Wi = mode ? Ws : W; g.Ho = mode ? H : Hs; g.Wo = mode ? W : Ws;
  g.pad_h = mode ? 2 - pad_h : pad_h; g.pad_w = mode ? 2 - pad_w : pad_w;
  g.x_bytes = (uint32_t)(B * g.Hi * g.Wi * C * 2);
  g.w_bytes = (uint32_t)cplxamd_conv2d_cl_pack_bytes(N, C, 3, 3);
  g.C = C; g.Cout = N; g.C16 = C / 16; g.NS = 3 * g.C16;
  g.tiles_x = (g.Wo + cl2::TW - 1) / cl2::TW; g.tiles_y = (g.Ho + cl2::TH - 1) / cl2::TH; g.tiles_n = N / 64;
  const int64_t ntiles = B * g.tiles_x * g.tiles_y * g.tiles_n;
  if (ntiles > 0x7fffffff) return CPLXAMD_ESHAPE;
  const int ncu = cl2_cus();
  // chip shared with RCCL collectives (CPLXAMD_LAUNCH_SHARED, launch.h): one workgroup per tile -- a launch that
  // expects every CU for its whole duration would wait for the held ones with its last workgroups
  const int grid = (ntiles < ncu || !launch_owns_chip(flags)) ? (int)ntiles : ncu;
  static PerDeviceOnce attr_set, attr_set_f;
  if (const int e = set_max_dyn_lds(attr_set, cl2::conv_cl2_kernel<false>, cl2::SMEM)) return e;
  if (const int e = set_max_dyn_lds(attr_set_f, cl2::conv_cl2_kernel<true>, cl2::SMEM)) return e;
  if (mom) {
    // one partial row per workgroup: a persistent launch only (at most one workgroup per CU), one column tile
    if (fga || mode != 0 || grid > ncu || !cl2_mom_tiling_ok(ntiles, grid, g.tiles_n)) return CPLXAMD_ESHAPE;
    if (mom_bytes < (int64_t)grid * N * 5 * (int64_t)sizeof(double)) return CPLXAMD_EWS;
    if (g.tiles_n > 1 &&
        hipMemsetAsync(mom, 0, (size_t)grid * N * 5 * sizeof(double), (hipStream_t)stream) != hipSuccess)
      return CPLXAMD_EINVAL;
    static PerDeviceOnce mom_attr;
    if (const int e = set_max_dyn_lds(mom_attr, cl2::conv_cl2_kernel<false, true>, cl2::SMEM_MOM)) return e;
    g.mom = mom;
    cl2::conv_cl2_kernel<false, true><<<dim3((unsigned)grid), cl2::NT, cl2::SMEM_MOM, (hipStream_t)stream>>>(g);
    CPLXAMD_CHECK_LAUNCH();
    return 0;
  }
  if (fga) cl2::conv_cl2_kernel<true><<<dim3((unsigned)grid), cl2::NT, cl2::SMEM, (hipStream_t)stream>>>(g);
  else cl2::conv_cl2_kernel<false><<<dim3((unsigned)grid), cl2::NT, cl2::SMEM, (hipStream_t)stream>>>(g);
  CPLXAMD_CHECK_LAUNCH();
  return 0;
}

// Number of per-workgroup partial rows cplxamd_conv2d_cl2_mom writes for this problem ([rows][N][5] float64), or 0 when the
// moments variant does not take it (kernel / dilation / channel counts cplxamd_conv2d_cl2 declines, a grid whose workgroups
// would change column tile, or the chip
// is shared with collectives -- CPLXAMD_LAUNCH_SHARED -- so that the launch is one workgroup per tile).
int64_t cplxamd_conv2d_cl2_mom_chunks(int64_t B, int H, int W, int C, int N, int KH, int KW, int dil_h, int dil_w, int pad_h,
                                      int pad_w) {
  return cplxamd_conv2d_cl2_mom_chunks_fl(B, H, W, C, N, KH, KW, dil_h, dil_w, pad_h, pad_w, CPLXAMD_LAUNCH_DEFAULT);
}

int64_t cplxamd_conv2d_cl2_mom_chunks_fl(int64_t B, int H, int W, int C, int N, int KH, int KW, int dil_h, int dil_w,
                                         int pad_h, int pad_w, int flags) {
  if (!launch_flags_ok(flags)) return 0;
  const int Hs = H + 2 * pad_h - 2, Ws = W + 2 * pad_w - 2;
  if (B <= 0 || H <= 0 || W <= 0 || pad_h < 0 || pad_w < 0 || KH != 3 || KW != 3 || dil_h != 1 || dil_w != 1 || C <= 0 ||
      C % 32 || N % cl2::BN || Hs <= 0 || Ws <= 0 || Hs > H || Ws > W)
    return 0;
  if (B * H * W * C * 2 >= (int64_t)0xF0000000 || B >= 65536) return 0;
  const int64_t ntiles = B * ((Ws + cl2::TW - 1) / cl2::TW) * ((Hs + cl2::TH - 1) / cl2::TH) * (N / cl2::BN);
  if (ntiles > 0x7fffffff) return 0;
  const int ncu = cl2_cus();
  if (ntiles >= ncu && !launch_owns_chip(flags)) return 0;
  const int grid = ntiles < ncu ? (int)ntiles : ncu;
  return cl2_mom_tiling_ok(ntiles, grid, N / cl2::BN) ? grid : 0;
}

// cplxamd_conv2d_cl2 (forward, mode 0) that ALSO leaves the batch-norm forward moments of its output -- per output
// channel: sum re, sum im, sum re^2, sum im^2, sum re im over every pixel, of the bf16 values as stored -- as
// cplxamd_conv2d_cl2_mom_chunks(...) partial rows [row][N][5] float64 in `partials`, the layout
// cplxamd_bn_fwd_partials sums: the batch-norm layer behind this convolution then needs no pass over y for its
// statistics (cplxmodule/nn/modules/batchnorm.py:62-123 after cplx.py:729-742).  CPLXAMD_ESHAPE when the variant does not
// take the problem (see cplxamd_conv2d_cl2_mom_chunks).
int cplxamd_conv2d_cl2_mom(const void* x_r, const void* x_i, const void* w_packed, const float* bias_r, const float* bias_i,
                           void* y_r, void* y_i, int64_t B, int H, int W, int C, int N, int KH, int KW, int dil_h, int dil_w,
                           int pad_h, int pad_w, double* partials, int64_t partials_bytes, void* ws, int64_t ws_bytes,
                           void* stream) {
  return cplxamd_conv2d_cl2_mom_fl(x_r, x_i, w_packed, bias_r, bias_i, y_r, y_i, B, H, W, C, N, KH, KW, dil_h, dil_w, pad_h, pad_w,
                                   partials, partials_bytes, ws, ws_bytes, CPLXAMD_LAUNCH_DEFAULT, stream);
}

int cplxamd_conv2d_cl2_mom_fl(const void* x_r, const void* x_i, const void* w_packed, const float* bias_r,
                              const float* bias_i, void* y_r, void* y_i, int64_t B, int H, int W, int C, int N, int KH, int KW,
                              int dil_h, int dil_w, int pad_h, int pad_w, double* partials, int64_t partials_bytes, void* ws,
                              int64_t ws_bytes, int flags, void* stream) {
  if (!partials || !launch_flags_ok(flags)) return CPLXAMD_EINVAL;
  if (cplxamd_conv2d_cl2_mom_chunks_fl(B, H, W, C, N, KH, KW, dil_h, dil_w, pad_h, pad_w, flags) <= 0) return CPLXAMD_ESHAPE;
  return launch_cl2(x_r, x_i, w_packed, bias_r, bias_i, nullptr, nullptr, nullptr, y_r, y_i, B, H, W, C, N, KH, KW, dil_h,
                    dil_w, pad_h, pad_w, 0, ws, ws_bytes, flags, stream, partials, partials_bytes);
}

// Same arguments and semantics as cplxamd_conv2d_cl (weights packed by cplxamd_conv2d_cl_pack); built for KH = KW = 3,
// dilation 1, C % 32 == 0, N % 64 == 0 -- CPLXAMD_ESHAPE otherwise (the caller then takes cplxamd_conv2d_cl).
int cplxamd_conv2d_cl2(const void* x_r, const void* x_i, const void* w_packed, const float* bias_r, const float* bias_i,
                       void* y_r, void* y_i, int64_t B, int H, int W, int C, int N, int KH, int KW, int dil_h, int dil_w,
                       int pad_h, int pad_w, int mode, void* ws, int64_t ws_bytes, void* stream) {
  return launch_cl2(x_r, x_i, w_packed, bias_r, bias_i, nullptr, nullptr, nullptr, y_r, y_i, B, H, W, C, N, KH, KW, dil_h,
                    dil_w, pad_h, pad_w, mode, ws, ws_bytes, CPLXAMD_LAUNCH_DEFAULT, stream);
}

int cplxamd_conv2d_cl2_fl(const void* x_r, const void* x_i, const void* w_packed, const float* bias_r, const float* bias_i,
                          void* y_r, void* y_i, int64_t B, int H, int W, int C, int N, int KH, int KW, int dil_h, int dil_w,
                          int pad_h, int pad_w, int mode, void* ws, int64_t ws_bytes, int flags, void* stream) {
  return launch_cl2(x_r, x_i, w_packed, bias_r, bias_i, nullptr, nullptr, nullptr, y_r, y_i, B, H, W, C, N, KH, KW, dil_h,
                    dil_w, pad_h, pad_w, mode, ws, ws_bytes, flags, stream);
}

// Input gradient of the local-reparameterization convolutions (CplxConv2dVD / ARD, cplxmodule/nn/relevance/complex.py
// :157-190 differentiated) in one launch: dx = dgrad(g; w) + 2 x (*) ga per plane, where dgrad is cplxamd_conv2d_cl2
// with mode 1 (w packed with swap = 1: flipped, conjugated, channel-swapped) and ga [B][H][W][N] is the variance path's
// data gradient (cplxamd_conv2d_clr, mode 1).  g_r / g_i: [B][H + 2 pad - 2][W + 2 pad - 2][C] output gradients;
// x_r / x_i / ga / dx_r / dx_i: [B][H][W][N].  Bit-identical to cplxamd_conv2d_cl2 followed by cplxamd_lrt_dx_accum.
// Same shape constraints as cplxamd_conv2d_cl2 (CPLXAMD_ESHAPE otherwise: the caller takes the two launches).
int cplxamd_conv2d_cl2_lrt_dx(const void* g_r, const void* g_i, const void* w_packed, const void* x_r, const void* x_i,
                              const void* ga, void* dx_r, void* dx_i, int64_t B, int H, int W, int C, int N, int pad_h,
                              int pad_w, void* ws, int64_t ws_bytes, void* stream) {
  return cplxamd_conv2d_cl2_lrt_dx_fl(g_r, g_i, w_packed, x_r, x_i, ga, dx_r, dx_i, B, H, W, C, N, pad_h, pad_w, ws, ws_bytes,
                                      CPLXAMD_LAUNCH_DEFAULT, stream);
}

int cplxamd_conv2d_cl2_lrt_dx_fl(const void* g_r, const void* g_i, const void* w_packed, const void* x_r, const void* x_i,
                                 const void* ga, void* dx_r, void* dx_i, int64_t B, int H, int W, int C, int N, int pad_h,
                                 int pad_w, void* ws, int64_t ws_bytes, int flags, void* stream) {
  if (!x_r || !x_i || !ga) return CPLXAMD_EINVAL;
  return launch_cl2(g_r, g_i, w_packed, nullptr, nullptr, x_r, x_i, ga, dx_r, dx_i, B, H, W, C, N, 3, 3, 1, 1, pad_h, pad_w, 1,
                    ws, ws_bytes, flags, stream);
}

#endif   // CPLXAMD_CONV_F16

}  // extern "C"
