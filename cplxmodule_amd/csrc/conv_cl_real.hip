// REAL-valued twin of conv_cl.hip: 3-wide convolutions (stride 1, padding up to `same`) on unpadded channels-last
// bf16 activations -- the variance path of the local-reparameterization convolution layers (conv of |x|^2 with
// exp(log_sigma2), cplxmodule/nn/relevance/complex/base.py:120-135, real/base.py:116-163) and the real Conv2dVD / ARD
// layers -- forward and data gradient.  Same structure (persistent workgroups, 3-slot LDS-DMA ring of (kernel row,
// 16 channels) stages shared by the three taps, borders by masked fragment addresses and the buffer range check,
// epilogue through LDS); what differs from the complex kernel: one plane, one MFMA per block pair instead of four, and
// therefore a wave tile of 128 pixels x 64 channels (4 x 2 blocks = 128 accumulators) and a workgroup tile of 1024
// grid pixels, so that a stage still carries 32 KiB of activations + 6 KiB of weights for its 192 MFMAs.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "launch.h"   // per-call launch policy (CPLXAMD_LAUNCH_SHARED: the chip is shared with collectives)

namespace cplxamd {
namespace clr {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int NT = 512, TM = 1024, BN = 64, IB = 4;
constexpr int A_PLANE = TM * 32 + 64;        // 1024 rows x 16 channels, then the zero row (row 1024)
constexpr int ZROW = TM * 32;                // plane-relative byte address of the zero row
// ablation builds (-DCPLXAMD_CL_DBG=n, timing only): 1 no global stores, 2 every store goes to the dump rows
// (L2-resident), 4 no epilogue at all, 8 no LDS-DMA after the prologue, 16 no start stagger, 32 contiguous A source
#ifndef CPLXAMD_CL_DBG
#define CPLXAMD_CL_DBG 0
#endif
constexpr int kClDbg = CPLXAMD_CL_DBG;
constexpr int NST = (kClDbg & 5) ? 0 : 16;   // global stores per wave in the epilogue (4 blocks x 4 rounds)

template <int KW> struct Geo {
  static constexpr int W_PLANE = KW * 64 * 32;                 // [kw][64 co][16 ch] bf16
  static constexpr int W_BYTES = W_PLANE;
  static constexpr int STAGE = A_PLANE + W_BYTES;
  static constexpr int PWN = (W_BYTES + 8191) / 8192;          // LDS-DMA pieces (8 KiB per workgroup) of the weights
  static constexpr int L = 4 + PWN;                            // pieces per stage and wave
  static constexpr int EPI = 3 * STAGE;                        // per wave: 16 epilogue rows of 144 B, then 512 B of bias
  static constexpr int EPI_WAVE = 16 * 144 + 512;
  static constexpr int DUMP = EPI + 8 * EPI_WAVE;              // where the surplus half of a weight piece goes
  static constexpr int SMEM = DUMP + 4096;
};

struct FastDiv { uint32_t m; int s; };       // n / d for any 32-bit n (round-up method); d == 1: s < 0

struct Args {
  const void* x_r;                           // [P][C] bf16
  const void* w;                             // packed weights (pack_kernel)
  const float* bias_r;                       // [Cout] or null
  void* y_r;                                 // [P][Cout] bf16
  void* dump;                                // 512 x Cout bf16: where the rows a tile does not own are stored
  int64_t P;
  uint32_t x_bytes, w_bytes;                 // bytes of one activation plane / of the packed weights
  int H, W, C, Cout, KH, dil_h, dil_w, pad_h, pad_w;   // H x W: the grid the tiles walk (the larger of the two images)
  int Hi, Wi, Ho, Wo;                        // extent of the input / output image, both top-left aligned on that grid
  int C16, NS, tm_out, tiles_m, tiles_n;
  FastDiv div_w, div_h;
  int stagger, stagger_from;                 // start delay: (blockIdx - stagger_from) * stagger clocks (0 below stagger_from)
};

__device__ __forceinline__ uint32_t fast_div(uint32_t n, FastDiv d) {
  if (d.s < 0) return n;
  const uint32_t t = __umulhi(d.m, n);
  return (t + ((n - t) >> 1)) >> d.s;
}

template <int N> __device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ bf16x8 neg_frag(bf16x8 v) {
  uint4 u = __builtin_bit_cast(uint4, v);
  u.x ^= 0x80008000u; u.y ^= 0x80008000u; u.z ^= 0x80008000u; u.w ^= 0x80008000u;
  return __builtin_bit_cast(bf16x8, u);
}

// LDS-DMA through a buffer descriptor: lane data = 16 bytes at  base + voff + soff, zeros when that is outside
// [0, num_records) (voff wraps in 32 bits, so a "negative" row is out of range too); destination M0 + lane * 16.
__device__ __forceinline__ void buf_lds16(i32x4 rsrc, uint32_t voff, uint32_t soff_uniform, uint32_t lds_off_uniform) {
#if defined(__HIP_DEVICE_COMPILE__)
#ifndef CPLXAMD_CL_DMA_MOD
#define CPLXAMD_CL_DMA_MOD ""
#endif
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen " CPLXAMD_CL_DMA_MOD " lds"
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

template <int KW>
__global__ __launch_bounds__(NT) void conv_clr_kernel(Args g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using G = Geo<KW>;
  constexpr int L = G::L, STAGE = G::STAGE;
  static_assert(L + NST <= 63, "vmcnt is a 6-bit counter");

  const int ntiles = g.tiles_m * g.tiles_n;
  const int nwg = gridDim.x;
  // virtual block id -> tile (XCD x owns a contiguous range of the tile order; consecutive tiles of the order
  // share their halo rows and, across column tiles, their whole input window)
  auto origin = [&](int v, int& r0, int& nt_) __attribute__((always_inline)) {
    const int q = ntiles >> 3, r = ntiles & 7, xcd = v & 7, idx = v >> 3;
    const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int mt = lin / g.tiles_n;
    nt_ = __builtin_amdgcn_readfirstlane(lin - mt * g.tiles_n);
    r0 = __builtin_amdgcn_readfirstlane(mt * g.tm_out);
  };

  const int tid0 = threadIdx.x;
  const int lane = tid0 & 63, wid = tid0 >> 6;
  const int l31 = lane & 31, lk = lane >> 5;
  const int wm = wid * 32 * IB;

  // zero rows (never overwritten: the A pieces cover rows 0..511 exactly)
  if (tid0 < 48) {
    const int slot = tid0 >> 4, r = tid0 & 15;
    *reinterpret_cast<uint32_t*>(smem + slot * STAGE + ZROW + r * 4) = 0u;
  }

  const i32x4 rs_xr = make_rsrc(g.x_r, g.x_bytes);
  const i32x4 rs_w = make_rsrc(g.w, g.w_bytes);
  // (ablation bit 32: rows of the source read as if they were 32 bytes apart -- every LDS-DMA piece one contiguous KiB)
  const uint32_t rowbytes = (kClDbg & 32) ? 32u : (uint32_t)g.C * 2u;

  const uint32_t smem_off = lds_offset_of(smem);
  const uint32_t wid_u = (uint32_t)__builtin_amdgcn_readfirstlane(tid0 >> 6);
  const uint32_t wave_lds = wid_u * 1024u;

  // per-lane source offsets of the LDS-DMA pieces.  A plane: chunk p = j*512 + tid sits at LDS row p >> 1, position
  // p & 1, and holds channel half (p & 1) ^ ((row >> 3) & 1) of input row `row` of the window (so that the 16 lanes
  // of a ds_read_b128 group hit 16 distinct 16-byte bank groups).  Weights: the packed stage IS the LDS image.
  uint32_t voa[4], vow[G::PWN];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int p = j * NT + tid0, row = p >> 1, c = (p & 1) ^ ((row >> 3) & 1);
    voa[j] = (uint32_t)row * rowbytes + (uint32_t)c * 16u;
  }
#pragma unroll
  for (int j = 0; j < G::PWN; ++j) {
    int p = j * NT + tid0;
    if (p >= G::W_BYTES / 16) p -= G::W_BYTES / 16;        // surplus lanes re-read valid chunks into the dump
    vow[j] = (uint32_t)p * 16u;
  }

  // An input image smaller than the grid (the data gradient of a convolution with less than `same` padding reads the
  // (Ho, Wo) output gradient on the (H, W) grid of the input): a window row is a grid pixel, its source the pixel of the
  // same coordinates in the dense input -- no longer window start + row, so the lane offsets are rebuilt per tile
  // (two divisions per piece).  Pixels outside the input alias others or fall out of range; the fragment masks skip them.
  const bool in_dense = g.Hi == g.H && g.Wi == g.W;
  uint32_t voa_c[4], voa_n[4];                     // offsets of the tile the LDS-DMA pointer is in / of the one after it
  auto lane_offsets = [&](int rr, uint32_t (&out)[4]) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (in_dense) { out[j] = voa[j]; continue; }
      const int pr = j * NT + tid0, row = pr >> 1, c = (pr & 1) ^ ((row >> 3) & 1);
      // (the kernel-row part of the shift stays in the scalar offset: it may turn a row that starts above the image,
      //  i.e. at a wrapped "negative" offset, into a valid one two kernel rows later)
      const int64_t pix = (int64_t)rr - g.pad_w + row;
      uint32_t o = 0xF8000000u;
      if (pix >= 0 && pix < g.P) {
        const uint32_t q = (uint32_t)pix, qh = fast_div(q, g.div_w), w = q - qh * (uint32_t)g.W;
        const uint32_t b = fast_div(qh, g.div_h), h = qh - b * (uint32_t)g.H;
        o = ((b * (uint32_t)g.Hi + h) * (uint32_t)g.Wi + w) * rowbytes + (uint32_t)c * 16u;
      }
      out[j] = o;
    }
  };

  // fragment addresses relative to the slot.  x: row = wm + i*32 + l31 + kw*dil_w; w: co = j*32 + l31.
  uint32_t a_rel[IB][KW], w_rel[2];
#pragma unroll
  for (int i = 0; i < IB; ++i)
#pragma unroll
    for (int kw = 0; kw < KW; ++kw) {
      const int row = wm + i * 32 + l31 + kw * g.dil_w;
      a_rel[i][kw] = (uint32_t)(row * 32 + ((lk ^ ((row >> 3) & 1)) << 4));
    }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int co = j * 32 + l31;
    w_rel[j] = (uint32_t)(A_PLANE + co * 32 + ((lk ^ ((co >> 3) & 1)) << 4));
  }

  f32x16 acc_r[IB][2];
  bf16x8 ar[2][IB], br[2][2];                              // [set][block]

  // ---- tiles -------------------------------------------------------------------------------------------------
  int v = blockIdx.x;
  int r0, nt0, r0n, ntn;
  origin(v, r0, nt0);
  bool has_next = v + nwg < ntiles;
  origin(has_next ? v + nwg : v, r0n, ntn);

  // ---- the LDS-DMA stage pointer: runs three stages ahead of the MFMAs, through the tile boundaries ----------
  const uint32_t kh_step = (uint32_t)(g.dil_h * g.Wi) * rowbytes - (uint32_t)(g.C16 - 1) * 32u;
  auto a_origin = [&](int rr) __attribute__((always_inline)) -> uint32_t {
    // window of kernel row 0 starts pad_h image rows and pad_w pixels before the tile (wraps below zero: out of range)
    return in_dense ? (uint32_t)(rr - g.pad_w - g.pad_h * g.W) * rowbytes : (uint32_t)(-g.pad_h * g.Wi) * rowbytes;
  };
  lane_offsets(r0, voa_c);
  lane_offsets(r0n, voa_n);
  uint32_t d_aoff = a_origin(r0), d_woff = (uint32_t)(nt0 * g.NS) * (uint32_t)G::W_BYTES;
  int d_cs = 0, d_u = 0;
  auto dma_advance = [&]() __attribute__((always_inline)) {
    ++d_u; ++d_cs;
    d_woff += (uint32_t)G::W_BYTES;
    if (d_cs == g.C16) { d_cs = 0; d_aoff += kh_step; } else { d_aoff += 32u; }
    if (d_u == g.NS) {                                     // on to the next tile of this workgroup (or this one again)
      d_u = 0; d_cs = 0;
      d_aoff = a_origin(r0n);
      d_woff = (uint32_t)(ntn * g.NS) * (uint32_t)G::W_BYTES;
      voa_c[0] = voa_n[0]; voa_c[1] = voa_n[1]; voa_c[2] = voa_n[2]; voa_c[3] = voa_n[3];
    }
  };
  const uint32_t soff[3] = {smem_off, smem_off + (uint32_t)STAGE, smem_off + 2u * (uint32_t)STAGE};
  // piece q of the stage at the pointer -> slot
  bool dma_on = true;
  auto dma_piece = [&](int q, uint32_t slot_off) __attribute__((always_inline)) {
    if ((kClDbg & 8) && !dma_on) return;
    if (q < 4) buf_lds16(rs_xr, voa_c[q] + d_aoff, 0u, slot_off + (uint32_t)(q * 8192) + wave_lds);
    else {
      const int j = q - 4;
      constexpr int FULLW = G::W_BYTES / 1024;             // waves of weight data in total
      const bool real = (uint32_t)(j * 8) + wid_u < (uint32_t)FULLW;
      const uint32_t dst = real ? slot_off + (uint32_t)(A_PLANE + j * 8192) + wave_lds
                                : smem_off + (uint32_t)G::DUMP + (wave_lds & 4095u);
      buf_lds16(rs_w, vow[j], d_woff, dst);
    }
  };

  // ---- masks: bit kw = column tap kw stays inside the image row, bit 8 + kh = row tap kh stays inside the image
  uint32_t vmask[IB];
  auto tile_masks = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < IB; ++i) {
      const uint32_t q = (uint32_t)(r0 + wm + i * 32 + l31);
      const uint32_t qh = fast_div(q, g.div_w), w = q - qh * (uint32_t)g.W;
      const uint32_t h = qh - fast_div(qh, g.div_h) * (uint32_t)g.H;
      uint32_t m = 0;
#pragma unroll
      for (int kw = 0; kw < KW; ++kw) {
        const int ww = (int)w + kw * g.dil_w - g.pad_w;
        m |= (ww >= 0 && ww < g.Wi) ? (1u << kw) : 0u;
      }
      for (int kh = 0; kh < g.KH; ++kh) {
        const int hh = (int)h + kh * g.dil_h - g.pad_h;
        m |= (hh >= 0 && hh < g.Hi) ? (0x100u << kh) : 0u;
      }
      vmask[i] = m;
    }
  };
  int c_kh = 0, c_cs = 0;                                  // (kernel row, channel slice) of the stage the MFMAs are in
  auto masked = [&](int i, int kw, int kh) __attribute__((always_inline)) -> uint32_t {
    const uint32_t hb = (vmask[i] >> (8 + kh)) & 1u;
    const uint32_t sel = (0u - hb) & vmask[i];
    const uint32_t m = (uint32_t)((int32_t)(sel << (31 - kw)) >> 31);
    return (a_rel[i][kw] & m) | ((uint32_t)ZROW & ~m);
  };

  // fragment idx of tap kw of the stage in slot `sl` -> register set `st` (idx order: what the first MFMA group needs first)
  auto read_one = [&](int st, const char* sl, int kw, int kh, int idx) __attribute__((always_inline)) {
    const char* wr_ = sl + kw * 2048;
    if (idx == 0) br[st][0] = *reinterpret_cast<const bf16x8*>(wr_ + w_rel[0]);
    else if (idx == 1) ar[st][0] = *reinterpret_cast<const bf16x8*>(sl + masked(0, kw, kh));
    else if (idx == 2) br[st][1] = *reinterpret_cast<const bf16x8*>(wr_ + w_rel[1]);
    else if (idx < 6) ar[st][idx - 2] = *reinterpret_cast<const bf16x8*>(sl + masked(idx - 2, kw, kh));
  };

  // 8 MFMAs on register set `st`; behind group n a fragment read of (slot rsl, tap rkw, kernel row rkh) into the other
  // set (6 in all) and the LDS-DMA pieces [q0, q1) of the stage at the pointer -> slot ds, spread over the groups
  auto mfma_sub = [&](int st, const char* rsl, int rkw, int rkh, bool do_read, uint32_t ds, int q0, int q1)
      __attribute__((always_inline)) {
    int q = q0;
#pragma unroll
    for (int i = 0; i < IB; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        acc_r[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(br[st][j], ar[st][i], acc_r[i][j], 0, 0, 0);
        const int grp = i * 2 + j;
        __builtin_amdgcn_sched_barrier(0);
        if (do_read && grp < 6) read_one(st ^ 1, rsl, rkw, rkh, grp);
        __builtin_amdgcn_sched_barrier(0);
        const int left = 8 - grp;
        const int n_now = (q1 - q + left - 1) / left;
#pragma unroll
        for (int r = 0; r < 6; ++r)
          if (r < n_now && q < q1) {
            __builtin_amdgcn_sched_barrier(0);
            dma_piece(q, ds);
            __builtin_amdgcn_sched_barrier(0);
            ++q;
          }
      }
  };

  const char* const sbase[3] = {smem, smem + STAGE, smem + 2 * STAGE};

  // one stage: tap 0 | tap 1 | [all reads of this slot done, next stage landed: barrier] | tap 2.
  // LDS-DMA: pieces 2.. of the stage two ahead (-> the slot the previous stage left) behind taps 0 and 1, pieces 0, 1
  // of the stage three ahead (-> this stage's slot) behind tap 2.  PAR = register set holding (this stage, tap 0).
  auto body = [&](auto SLOT, auto PARITY, auto FLAVOUR) __attribute__((always_inline)) {
    constexpr int kc = decltype(SLOT)::value, par = decltype(PARITY)::value, fl = decltype(FLAVOUR)::value;
    static_assert(KW == 3, "the tap schedule below is written for three taps");
    const uint32_t o0 = soff[kc], o2 = soff[(kc + 2) % 3];
    const char* p0 = sbase[kc];
    const char* p1 = sbase[(kc + 1) % 3];
    const int kh = c_kh;
    if (fl == FL_FIRST0) {                                 // the stage two ahead was issued whole before the stores
      mfma_sub(par, p0, 1, kh, true, o2, 0, 0);
      mfma_sub(par ^ 1, p0, 2, kh, true, o2, 0, 0);
    } else {
      mfma_sub(par, p0, 1, kh, true, o2, 2, 4);
      mfma_sub(par ^ 1, p0, 2, kh, true, o2, 4, L);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // (FIRST0 / FIRST1: the stages they need landed before the stores' wait; a plain vmcnt(L) would wait for the
    //  epilogue's stores, which are older than the pieces issued since)
    if (fl == FL_FIRST1) wait_vmcnt<L + NST>();
    else if (fl != FL_FIRST0) wait_vmcnt<L>();
    __builtin_amdgcn_s_barrier();
    // the MFMAs move on to the next stage
    if (++c_cs == g.C16) { c_cs = 0; ++c_kh; }
    dma_advance();
    if (fl == FL_LAST) mfma_sub(par, p1, 0, 0, false, o0, 0, L);   // whole next-tile stage 2; its first fragments are
    else mfma_sub(par, p1, 0, c_kh, true, o0, 0, 2);               // read after the epilogue (32 registers less there)
  };
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;

  // thread id the compiler cannot see through: addresses of per-tile code are rebuilt from it instead of being
  // hoisted across the K loop (where they would be spilled, and a scratch reload behind the stores waits for them)
  auto opaque_tid = [&]() __attribute__((always_inline)) -> int {
    int t = (int)threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
  };

  // ---- epilogue: 8 rows x 128 bytes per round through this wave's 2 KiB; NST unpredicated stores per wave (rows a
  // tile does not own -- the overlap with the next tile, rows past the tensor -- go to the dump buffer)
  auto epilogue = [&](int nt_) __attribute__((always_inline)) {
    if (kClDbg & 4) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
      for (int i = 0; i < IB; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(acc_r[i][j]));
#endif
      return;
    }
    const int t = opaque_tid();
    const int ln = t & 63, w_ = t >> 6, q31 = ln & 31, qk = ln >> 5;
    const int wm_ = w_ * 32 * IB;
    char* reg = smem + G::EPI + w_ * G::EPI_WAVE;
    constexpr int PITCH = 144;
    const int r16 = q31 >> 4, rr = q31 & 15;
    const int64_t ldc = g.Cout;
    const bool out_dense = g.Ho == g.H && g.Wo == g.W, wide = g.W >= 32 * IB;   // (one carry at most for the 128 rows of a wave)
    uint32_t w_first = 0, h_first = 0, b_first = 0;       // grid coordinates of this lane's first row (r0 + wm + lane / 8)
    if (!out_dense) {
      const uint32_t q = (uint32_t)(r0 + wm_ + (ln >> 3)), qh = fast_div(q, g.div_w);
      w_first = q - qh * (uint32_t)g.W; b_first = fast_div(qh, g.div_h); h_first = qh - b_first * (uint32_t)g.H;
    }
    int64_t own = g.P - r0;
    if (own > g.tm_out) own = g.tm_out;
    {
      bf16_t* out = reinterpret_cast<bf16_t*>(g.y_r);
#pragma unroll
      for (int i = 0; i < IB; ++i)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          if (r16 == half) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                f4 x;
#pragma unroll
                for (int e = 0; e < 4; ++e) x.v[e] = acc_r[i][j][4 * q + e];
                st4(reinterpret_cast<bf16_t*>(reg + rr * PITCH + (j * 32 + 8 * q + 4 * qk) * 2), x);
              }
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
          for (int sub = 0; sub < 2; ++sub) {
            const uint4 val = *reinterpret_cast<const uint4*>(reg + (sub * 8 + (ln >> 3)) * PITCH + (ln & 7) * 16);
            if (kClDbg & 1) continue;
            const int m = wm_ + i * 32 + half * 16 + sub * 8 + (ln >> 3);
            const int col = nt_ * BN + (ln & 7) * 8;
            int64_t orow = (int64_t)r0 + m;
            bool ok = m < own && !(kClDbg & 2);
            if (!out_dense) {                       // output image smaller than the grid: its own dense row index
              uint32_t w = w_first + (uint32_t)(i * 32 + half * 16 + sub * 8), h = h_first, b = b_first;
              if (wide) {
                if (w >= (uint32_t)g.W) { w -= (uint32_t)g.W; if (++h == (uint32_t)g.H) { h = 0; ++b; } }
              } else {
                const uint32_t q = (uint32_t)orow, qh = fast_div(q, g.div_w);
                w = q - qh * (uint32_t)g.W; b = fast_div(qh, g.div_h); h = qh - b * (uint32_t)g.H;
              }
              ok = ok && h < (uint32_t)g.Ho && w < (uint32_t)g.Wo;
              orow = ((int64_t)b * g.Ho + h) * g.Wo + w;
            }
            bf16_t* dst = ok ? out + orow * ldc + col : reinterpret_cast<bf16_t*>(g.dump) + (int64_t)m * ldc + col;
#ifndef CPLXAMD_CL_NO_NT          // streaming stores: the output is read by a later kernel, not by this one (-1.4 %)
            typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
            __builtin_nontemporal_store(u32x4_t{val.x, val.y, val.z, val.w}, reinterpret_cast<u32x4_t*>(dst));
#else
            *reinterpret_cast<uint4*>(dst) = val;
#endif
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
  };

  // bias of column tile nt_ -> this wave's 512 bytes behind its epilogue rows (lanes 0-15 real, 16-31 imaginary)
  const uint32_t bias_lds = smem_off + (uint32_t)G::EPI + wid_u * (uint32_t)G::EPI_WAVE + 2304u;
  auto bias_dma = [&](int nt_) __attribute__((always_inline)) {
    const int t = opaque_tid();
    const int ln = t & 63;
    if (ln < 16) {
      const float* src = g.bias_r ? g.bias_r + nt_ * BN + 4 * ln : reinterpret_cast<const float*>(g.w) + 4 * ln;
      lds_dma16_at(src, bias_lds);
    }
  };
  auto init_acc = [&]() __attribute__((always_inline)) {
    const int t = opaque_tid();
    const int w_ = t >> 6, qk = (t & 63) >> 5;
    const char* breg = smem + G::EPI + w_ * G::EPI_WAVE + 2304;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f4 b = {{0.f, 0.f, 0.f, 0.f}};
        if (g.bias_r) b = ld4(reinterpret_cast<const float*>(breg + (j * 32 + 8 * q + 4 * qk) * 4));
#pragma unroll
        for (int i = 0; i < IB; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc_r[i][j][4 * q + e] = b.v[e];
      }
  };
  auto first_frags = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int idx = 0; idx < 6; ++idx) read_one(0, sbase[0], 0, 0, idx);
  };

  // start stagger: tiles take the same time on every CU, so without it all 256 CUs reach their epilogues together
  // and the 128 KiB per CU of stores arrive at the L2s / HBM as one burst per tile
  if (!(kClDbg & 16)) {
    const int n = __builtin_amdgcn_readfirstlane(((int)blockIdx.x - g.stagger_from) * g.stagger);
    for (int i = 0; i < n; i += 32 * 64) __builtin_amdgcn_s_sleep(32);
  }
  // ---- prologue of the first tile: stages 0, 1, 2 whole (the state every later tile starts from) -------------
  __syncthreads();                                         // zero rows written
  bias_dma(nt0);
#pragma unroll
  for (int s = 0; s < 3; ++s) {
#pragma unroll
    for (int q = 0; q < L; ++q) dma_piece(q, soff[s]);
    if (s < 2) dma_advance();
  }
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  dma_on = false;
  init_acc();
  tile_masks();
  first_frags();

  const int npairs = (g.NS - 6) / 6;
  for (;;) {
    c_kh = 0; c_cs = 0;
    body(I0{}, I0{}, I1{});                                // FIRST0
    body(I1{}, I1{}, I3{});                                // FIRST1
    body(I2{}, I0{}, I0{});
    for (int pr = 0; pr < npairs; ++pr) {
      body(I0{}, I1{}, I0{}); body(I1{}, I0{}, I0{}); body(I2{}, I1{}, I0{});
      body(I0{}, I0{}, I0{}); body(I1{}, I1{}, I0{}); body(I2{}, I0{}, I0{});
    }
    body(I0{}, I1{}, I0{});
    body(I1{}, I0{}, I0{});
    body(I2{}, I1{}, I2{});                                // LAST
    bias_dma(has_next ? ntn : nt0);                        // (older than the stores below)
    epilogue(nt0);
    if (!has_next) break;
    wait_vmcnt<NST>();                                     // everything issued BEFORE the stores has landed
    v += nwg;
    r0 = r0n; nt0 = ntn;
    has_next = v + nwg < ntiles;
    origin(has_next ? v + nwg : v, r0n, ntn);
    lane_offsets(r0n, voa_n);
    init_acc();
    tile_masks();
    first_frags();
  }
  wait_vmcnt<0>();                                         // surplus pieces of the last tail land before the LDS is released
}

// ---- weight packing: [Co][Ci][KH][KW] (bf16) -> per (column tile, kernel row, 16-channel slice) the LDS image
// [kw][64 n][2 chunk positions][8] with chunk position cp holding channel half cp ^ ((n >> 3) & 1).
// dgrad: n runs over Ci, the contraction over Co, both kernel axes flipped.
__global__ void pack_kernel(const bf16_t* w, bf16_t* out, int Co, int Ci, int KH, int KW, int dgrad, int64_t total) {
  const int N = dgrad ? Ci : Co, C = dgrad ? Co : Ci;
  const int C16 = C / 16;
  for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x) {
    int64_t t = o;
    const int e = (int)(t & 7); t >>= 3;
    const int cp = (int)(t & 1); t >>= 1;
    const int nl = (int)(t & 63); t >>= 6;
    const int kw = (int)(t % KW); t /= KW;
    const int cs = (int)(t % C16); t /= C16;
    const int kh = (int)(t % KH); t /= KH;
    const int nt = (int)t;
    const int n = nt * 64 + nl;
    const int c = cs * 16 + ((cp ^ ((nl >> 3) & 1)) << 3) + e;
    bf16_t val = 0;
    if (n < N) {
      if (!dgrad) val = w[(((int64_t)n * Ci + c) * KH + kh) * KW + kw];
      else val = w[(((int64_t)c * Ci + n) * KH + (KH - 1 - kh)) * KW + (KW - 1 - kw)];
    }
    out[o] = val;
  }
}

static FastDiv make_div(uint32_t d) {
  if (d <= 1) return FastDiv{0u, -1};
  int s = 0;
  while ((1ull << s) < d) ++s;                            // s = ceil(log2 d) >= 1
  const uint64_t m = (((1ull << s) - d) << 32) / d + 1;   // ceil(2^(32+s) / d) - 2^32
  return FastDiv{(uint32_t)m, s - 1};
}

}  // namespace clr
}  // namespace cplxamd

using namespace cplxamd;

extern "C" {

int64_t cplxamd_conv2d_clr_pack_bytes(int N, int C, int KH, int KW) {
  if (N <= 0 || C <= 0 || KH <= 0 || KW <= 0 || C % 16) return 0;
  return (int64_t)((N + 63) / 64) * KH * (C / 16) * KW * 64 * 16 * 2;
}

int64_t cplxamd_conv2d_clr_ws_bytes(int Cout) { return Cout > 0 ? (int64_t)clr::TM * Cout * 2 : 0; }

int cplxamd_conv2d_clr_pack(const void* w, void* out, int Co, int Ci, int KH, int KW, int dgrad, void* stream) {
  if (!w || !out || Co <= 0 || Ci <= 0 || KH <= 0 || KW <= 0) return CPLXAMD_EINVAL;
  if ((dgrad ? Co : Ci) % 16) return CPLXAMD_ESHAPE;
  const int64_t total = cplxamd_conv2d_clr_pack_bytes(dgrad ? Ci : Co, dgrad ? Co : Ci, KH, KW) / 2;
  clr::pack_kernel<<<stream_grid(total, 256), 256, 0, (hipStream_t)stream>>>((const bf16_t*)w, (bf16_t*)out, Co, Ci, KH, KW,
                                                                              dgrad, total);
  CPLXAMD_CHECK_LAUNCH();
  return 0;
}

// The real-valued cplxamd_conv2d_cl: same arguments (one plane each), same conditions.
int cplxamd_conv2d_clr(const void* x, const void* w_packed, const float* bias, void* y, int64_t B, int H, int W, int C,
                       int N, int KH, int KW, int dil_h, int dil_w, int pad_h, int pad_w, int mode, void* ws,
                       int64_t ws_bytes, void* stream) {
  return cplxamd_conv2d_clr_fl(x, w_packed, bias, y, B, H, W, C, N, KH, KW, dil_h, dil_w, pad_h, pad_w, mode, ws, ws_bytes,
                               CPLXAMD_LAUNCH_DEFAULT, stream);
}

int cplxamd_conv2d_clr_fl(const void* x, const void* w_packed, const float* bias, void* y, int64_t B, int H, int W, int C,
                          int N, int KH, int KW, int dil_h, int dil_w, int pad_h, int pad_w, int mode, void* ws,
                          int64_t ws_bytes, int flags, void* stream) {
  if (!launch_flags_ok(flags)) return CPLXAMD_EINVAL;
  if (!x || !w_packed || !y || B < 0 || H <= 0 || W <= 0 || C <= 0 || N <= 0 || KH <= 0 || KW <= 0 || dil_h <= 0 ||
      dil_w <= 0 || pad_h < 0 || pad_w < 0 || (mode != 0 && mode != 1))
    return CPLXAMD_EINVAL;
  const int Hs = H + 2 * pad_h - dil_h * (KH - 1), Ws = W + 2 * pad_w - dil_w * (KW - 1);   // the smaller image
  if (KW != 3 || C % 16 || N % 64 || (KH * (C / 16)) % 6 || Hs <= 0 || Ws <= 0 || Hs > H || Ws > W ||
      (KW - 1) * dil_w > 64 || KH > 8)
    return CPLXAMD_ESHAPE;
  const int64_t P = B * H * W;
  if (P == 0) return 0;
  const int64_t halo = ((int64_t)dil_h * (KH - 1) * W + dil_w * (KW - 1) + clr::TM) * C * 2;
  if (P >= ((int64_t)1 << 31) - clr::TM || P * C * 2 + 2 * halo >= (int64_t)0xF0000000) return CPLXAMD_ESHAPE;
  auto a16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  if (!a16(x) || !a16(w_packed) || !a16(y) || !a16(ws) || (bias && !a16(bias))) return CPLXAMD_EALIGN;
  if (!ws || ws_bytes < cplxamd_conv2d_clr_ws_bytes(N)) return CPLXAMD_EINVAL;
  clr::Args g{};
  g.x_r = x; g.w = w_packed; g.bias_r = bias; g.y_r = y; g.dump = ws;
  g.P = P;
  g.Hi = mode ? Hs : H; g.Wi = mode ? Ws : W; g.Ho = mode ? H : Hs; g.Wo = mode ? W : Ws;
  g.x_bytes = (uint32_t)(B * g.Hi * g.Wi * C * 2);
  if (mode) { pad_h = dil_h * (KH - 1) - pad_h; pad_w = dil_w * (KW - 1) - pad_w; }
  g.w_bytes = (uint32_t)cplxamd_conv2d_clr_pack_bytes(N, C, KH, KW);
  g.H = H; g.W = W; g.C = C; g.Cout = N; g.KH = KH; g.dil_h = dil_h; g.dil_w = dil_w; g.pad_h = pad_h; g.pad_w = pad_w;
  g.C16 = C / 16; g.NS = KH * g.C16;
  g.tm_out = clr::TM - (KW - 1) * dil_w;
  g.tiles_m = (int)((P + g.tm_out - 1) / g.tm_out);
  g.tiles_n = N / 64;
  g.div_w = clr::make_div((uint32_t)W); g.div_h = clr::make_div((uint32_t)H);
  static const int stagger_pct = [] { const char* e = getenv("CPLXAMD_CL_STAGGER"); return e ? atoi(e) : 100; }();
  const int ncu = device_cus() & ~7;
  const int64_t ntiles = (int64_t)g.tiles_m * g.tiles_n;
  if (ntiles > 0x7fffffff) return CPLXAMD_ESHAPE;
  // chip shared with RCCL collectives (CPLXAMD_LAUNCH_SHARED, launch.h): one workgroup per tile
  int grid = (ntiles < ncu || !launch_owns_chip(flags)) ? (int)ntiles : ncu;
  g.stagger = 0; g.stagger_from = 0;
  if (ntiles > 2 * grid && ntiles % grid) {
    const int64_t tile_clk = (int64_t)g.NS * 2 * 24 * 32 * 2;
    g.stagger_from = (int)(ntiles % grid);
    g.stagger = (int)(tile_clk * stagger_pct / 100 / (grid - g.stagger_from));
  }
  using G3 = clr::Geo<3>;
  static PerDeviceOnce attr_set;
  if (const int e = set_max_dyn_lds(attr_set, clr::conv_clr_kernel<3>, G3::SMEM)) return e;
  clr::conv_clr_kernel<3><<<dim3((unsigned)grid), clr::NT, G3::SMEM, (hipStream_t)stream>>>(g);
  CPLXAMD_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
