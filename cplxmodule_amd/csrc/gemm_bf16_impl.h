// bf16 complex / real GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x16_bf16):
//   C[m,n] = sum_k A[m,k] * op(B[n,k]) (+ bias[n]),  planar re / im, fp32 accumulation,
//   bf16 or fp32 output.  Each operand may be stored K-contiguous ([rows][K], "N") or
//   K-major ([K][rows], "T"): forward = (N,N), dgrad = (N,T) with the weight as stored,
//   wgrad = (T,T) with the activations / gradients as stored -- no transposed copies.
//
// Complex 4M in ONE K-loop: each staged (Ar, Ai, Br, Bi) tile set feeds four MFMA chains
//   Cr += Ar Br ; Cr += (-Ai) Bi ; Ci += Ar Bi ; Ci += Ai Br
// so every LDS byte is used twice as often as in a real GEMM of the same tile (the reference
// issues 4 separate GEMMs + 2 elementwise passes, cplx.py:641-646).  The sign flip for the
// Ai Bi product (and for conj(B), used by dgrad / wgrad) is an XOR on the packed bf16 fragment.
//
// Structure (profiles/r01_gemm_variants.md has the variant study):
//  * 256 x 128 output tile, 8 waves, each wave a 64 x 64 sub-tile = 2x2 MFMA tiles x {re, im}
//    = 128 accumulator registers; BK = 32.
//  * operand tiles go global -> LDS with global_load_lds_dwordx4 (no VGPR round trip) into a ring
//    of 3 stages (144 KiB); tiles t+1, t+2 stay in flight across the raw s_barrier, the wait for
//    tile t is a COUNTED s_waitcnt vmcnt (cdna_hip_programming.md T3/T4); the 6 LDS-DMA pieces of
//    tile t+2 are spread between the MFMA groups of tile t instead of issued as one burst.
//  * LDS images (the LDS-DMA writes lane-linearly, so every swizzle is applied to the per-lane
//    global SOURCE address and again on the read -- rule 21 of the guide):
//      "N" operand: [rows][32 k], 64-B rows, 16-B chunk index XOR (row >> 2) & 3; fragments by
//                   ds_read_b128 (16 distinct bank slots per lane group);
//      "T" operand: [32 k][rows], 64-B segment index XOR (k & 3); fragments by two
//                   ds_read_b64_tr_b16 (hardware 4 x 16 transpose: lane m of a 16-lane group
//                   addresses T[kb + (m >> 2)][rb + 4 (m & 3)] and receives T[kb..kb+3][rb + m]).
//  * tile order: XCD-contiguous ranges of a grouped order (GROUP_M row panels x all column
//    panels) so the tiles resident on one XCD share A / B panels in that XCD's private L2.
//  * accumulator tiles are kept transposed (B fragment as the first MFMA operand) so each lane
//    owns 4 consecutive output columns: 8-byte (bf16) / 16-byte (fp32) stores.
//  * rolling half-tile pipeline: the s_barrier sits in the middle of a tile, when the wave still holds the second K sub-step's
//    fragments in registers, so the MFMA pipe runs across the barrier and across the LDS latency
//    of the next tile's first fragments (+2-4 % on N(0,1) data, +6 % on zero-filled operands).
//  * few output tiles + long K (wgrad at batch 2^20): split-K into fp32 slabs + a reduce kernel.
// The experiment variants of rounds 2-3 (compile-time ablation bits, the classic per-tile pipeline, alternative MFMA
// orders) were working COPIES of this header; they are in the history only (git show 1a9fb01:scripts/gemm_experiments/),
// their results in profiles/r02_gemm_ablation.md / r03_gemm_pair_issue.txt.  Round 4's family (gemm_bf16_w4.hip) keeps
// its ablation switches in the production file behind W4_BURST / W4_LDP / ... (scripts/r04/w4_build.sh; the W4_DBG / W4_PGRID / W4_PDYN ablations and the CPLXAMD_W4P_CPLX
// selector left the file in round 6: scripts/r06/ablation_switches.patch puts them back).
#include <stdlib.h>

#include <type_traits>

#include "gemm.h"

namespace cplxamd {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32, STAGES = 3;
// complex: 256 x 128 tile, 4 x 2 waves of 64 x 64 (2 x 2 MFMA tiles x {re, im} = 128 accumulators);
// real: 256 x 256 tile, 2 x 4 waves of 128 x 64 (4 x 2 MFMA tiles = 128 accumulators) -- with one
// MFMA chain per staged byte instead of four, the real kernel needs the larger tile to keep the
// LDS-DMA pieces and ds_reads per MFMA where the complex kernel has them.
template <bool CPLX>
struct Cfg {
  static constexpr int NT = 512;
  static constexpr int IB = CPLX ? 2 : 4;                       // 32-row MFMA blocks per wave
  static constexpr int JB = 2;                                  // 32-column MFMA blocks per wave
  static constexpr int WM = CPLX ? 4 : 2, WN = CPLX ? 2 : 4;    // waves along M / N
  static constexpr int BM = 32 * IB * WM, BN = 32 * JB * WN;
  static constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
  static constexpr int STAGE_BYTES = (CPLX ? 2 : 1) * (A_BYTES + B_BYTES);
  // (a fourth 32 KiB slot for the real tile fits in 160 KiB; a ring generalised to NS slots measured
  //  no gain with 4 slots and cost the real kernel 2-5 % with 3 -- profiles/r01_gemm_variants.md)
  static constexpr int SMEM = STAGES * STAGE_BYTES;
  static constexpr int PA = BM * 4 / NT, PB = BN * 4 / NT;     // LDS-DMA pieces per plane
  static constexpr int LOADS = (CPLX ? 2 : 1) * (PA + PB);      // ... per thread per K tile
};


// One LDS-DMA instruction moves piece j (of ROWS*4/NT) of a plane tile.
//  !T: chunk p = j*NT + tid holds (row = p >> 2, kc = (p & 3) ^ ((row >> 2) & 3)) of [rows][K]
//   T: chunk p holds (k = p / (ROWS/8), c = (p % (ROWS/8)) ^ ((k & 3) << 2)) of [K][rows]
// The address is split into a wave-uniform part (plane + tile origin + K position: piece_base, scalar)
// and this lane's byte offset inside the tile (piece_voff: computed ONCE per kernel, 32 bits).
template <int ROWS, bool T, int NT>
__device__ __forceinline__ uint32_t piece_voff(int64_t ld, int row0, int rows, int j) {
  const int p = j * NT + (int)threadIdx.x;
  if (!T) {
    const int row = p >> 2;
    const int kc = (p & 3) ^ ((row >> 2) & 3);
    int grow = row0 + row;
    grow = grow < rows ? grow : rows - 1;      // clamp: out-of-range rows are never stored
    return (uint32_t)(((int64_t)(grow - row0) * ld + kc * 8) * 2);
  }
  constexpr int CPR = ROWS / 8;                // 16-B chunks per k row
  const int k = p / CPR;
  const int c = (p % CPR) ^ ((k & 3) << 2);
  int col = row0 + c * 8;
  col = col + 8 <= rows ? col : rows - 8;      // clamp (rows % 8 == 0 is required)
  return (uint32_t)(((int64_t)k * ld + (col - row0)) * 2);
}
template <bool T>
__device__ __forceinline__ const bf16_t* piece_base(const bf16_t* plane, int64_t ld, int row0, int k0) {
  return T ? plane + (int64_t)k0 * ld + row0 : plane + (int64_t)row0 * ld + (int64_t)k0;
}

// 8 consecutive k of matrix row `row` (k chunk kc of 4) from an "N" image
__device__ __forceinline__ bf16x8 frag_n(const char* lds_plane, int row, int kc) {
  const int off = row * 64 + ((kc ^ ((row >> 2) & 3)) << 4);
  return *reinterpret_cast<const bf16x8*>(lds_plane + off);
}

// same fragment from a "T" image [32 k][ROWS]: two hardware-transposed 4 x 16 reads.
// rb: first row of this lane's 16-row block, kb: first k of the 8, m = lane & 15.
template <int ROWS>
__device__ __forceinline__ bf16x8 frag_t(const char* lds_plane, int rb, int kb, int m) {
  const int r = rb + 4 * (m & 3);
  s16x4 v[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int k = kb + 4 * h + (m >> 2);
    const int off = k * (ROWS * 2) + (((r >> 3) ^ ((k & 3) << 2)) << 4) + (r & 7) * 2;
    v[h] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)(lds_plane + off));
  }
  const s16x8 both = __builtin_shufflevector(v[0], v[1], 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, both);
}

__device__ __forceinline__ bf16x8 neg_frag(bf16x8 v) {
  uint4 u = __builtin_bit_cast(uint4, v);
  u.x ^= 0x80008000u; u.y ^= 0x80008000u; u.z ^= 0x80008000u; u.w ^= 0x80008000u;
  return __builtin_bit_cast(bf16x8, u);
}

// 16-byte epilogue stores (plain: nontemporal stores measured no gain for outputs a later kernel reads)
__device__ __forceinline__ void nt_store16(void* p, uint4 v) { *reinterpret_cast<uint4*>(p) = v; }
__device__ __forceinline__ void nt_store16(float* p, const f4& a) { st4(p, a); }

template <int N> __device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <typename TOUT, bool CPLX, bool CONJ, bool TA, bool TB>
__global__ __launch_bounds__((Cfg<CPLX>::NT), 2) void gemm_bf16_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using C = Cfg<CPLX>;
  constexpr int NT = C::NT;

  // ---- tile coordinates: split-K slice, XCD-contiguous grouped order ------------------------
  constexpr int BM = C::BM, BN = C::BN, IB = C::IB, JB = C::JB;
  const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN;
  const int ntiles = tiles_m * tiles_n;
  int lin = blockIdx.x, split = 0;
  if (g.splits > 1) { split = lin / ntiles; lin -= split * ntiles; }
  int bm, bn;
  if (g.order == 0) {            // natural: consecutive blocks walk N
    bm = lin / tiles_n; bn = lin - bm * tiles_n;
  } else {
    const int q = ntiles >> 3, r = ntiles & 7, xcd = lin & 7, idx = lin >> 3;
    lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;  // bijective
    const int GM = g.group_m;
    const int per_group = GM * tiles_n;
    const int grp = lin / per_group, in_grp = lin - grp * per_group;
    const int first_m = grp * GM;
    const int gm = (tiles_m - first_m) < GM ? (tiles_m - first_m) : GM;
    bm = first_m + in_grp % gm; bn = in_grp / gm;
  }
  const int m0 = __builtin_amdgcn_readfirstlane(bm * BM), n0 = __builtin_amdgcn_readfirstlane(bn * BN);

  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int wm = (wid / C::WN) * (32 * IB), wn = (wid % C::WN) * (32 * JB);
  const int l31 = lane & 31, lk = lane >> 5;
  const int l15 = lane & 15, lg = (lane >> 4) & 1;   // "T" reads: 16-lane group geometry

  const bf16_t* Ar = (const bf16_t*)g.a_r; const bf16_t* Ai = (const bf16_t*)g.a_i;
  const bf16_t* Br = (const bf16_t*)g.b_r; const bf16_t* Bi = (const bf16_t*)g.b_i;
  const int64_t lda = TA ? g.a_cs : g.a_rs, ldb = TB ? g.b_cs : g.b_rs;

  f32x16 acc_r[IB][JB], acc_i[CPLX ? IB : 1][JB];
#pragma unroll
  for (int i = 0; i < IB; ++i)
#pragma unroll
    for (int j = 0; j < JB; ++j) {
      acc_r[i][j] = f32x16{0};
      if (CPLX) acc_i[i][j] = f32x16{0};
    }

  // The bias rides in the accumulators (acc starts at bias[n] instead of 0): its 16-B loads are issued here,
  // ahead of the prologue's LDS-DMA, and land while the first K tiles are fetched.  In the epilogue the same
  // values cost 128 dependent 4-byte loads per wave with nothing to overlap (18 us of the 52 us the bf16
  // forward epilogue took, profiles/r02_gemm_ablation.md).
  const bool bias_in_acc = g.bias_r != nullptr && g.g1 == nullptr && g.splits <= 1;
  f4 bias_v[CPLX ? 2 : 1][JB][4];
  if (bias_in_acc) {
    const bool vec = (g.N & 3) == 0 && (reinterpret_cast<uintptr_t>(g.bias_r) & 15) == 0 &&
                     (!CPLX || (reinterpret_cast<uintptr_t>(g.bias_i) & 15) == 0);
#pragma unroll
    for (int pl = 0; pl < (CPLX ? 2 : 1); ++pl) {
      const float* bias = pl ? g.bias_i : g.bias_r;
#pragma unroll
      for (int j = 0; j < JB; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int col = n0 + wn + j * 32 + 8 * q + 4 * lk;
          if (vec && col + 3 < g.N) {
            bias_v[pl][j][q] = ld4(bias + col);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) bias_v[pl][j][q].v[e] = col + e < g.N ? bias[col + e] : 0.f;
          }
        }
    }
#ifdef CPLXAMD_GEMM_F16    // (the half-operand build only: the bf16 kernels' code is untouched)
    if (g.scale_a) {          // scaled split products (gemm.h): the accumulators are multiplied by 1 / (sa sb) behind the K loop
      const float inv = gemm_alpha_inv(g);
#pragma unroll
      for (int pl = 0; pl < (CPLX ? 2 : 1); ++pl)
#pragma unroll
        for (int j = 0; j < JB; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) bias_v[pl][j][q].v[e] *= inv;
    }
#endif
  }
  auto apply_bias = [&]() __attribute__((always_inline)) {
    if (!bias_in_acc) return;
#pragma unroll
    for (int i = 0; i < IB; ++i)
#pragma unroll
      for (int j = 0; j < JB; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            acc_r[i][j][4 * q + e] = bias_v[0][j][q].v[e];
            if (CPLX) acc_i[i][j][4 * q + e] = bias_v[CPLX ? 1 : 0][j][q].v[e];
          }
  };
  const int kbase = __builtin_amdgcn_readfirstlane(split * g.kchunk);
  // piece q (0 .. LOADS-1) of the K tile starting at k0 into ring slot buf
  const uint32_t smem_off = lds_offset_of(smem);
  const uint32_t wave_lds = (uint32_t)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * 1024u;  // 64 lanes x 16 B
  // this lane's byte offset of every piece (the two planes of an operand share it)
  uint32_t voa[C::PA], vob[C::PB];
#pragma unroll
  for (int j = 0; j < C::PA; ++j) voa[j] = piece_voff<BM, TA, NT>(lda, m0, g.M, j);
#pragma unroll
  for (int j = 0; j < C::PB; ++j) vob[j] = piece_voff<BN, TB, NT>(ldb, n0, g.N, j);
  // piece q (0 .. LOADS-1) of the K tile starting at k0 into ring slot buf
  auto stage_q = [&](int buf, int k0, int q) __attribute__((always_inline)) {
    k0 += kbase;
    const uint32_t s = smem_off + (uint32_t)(buf * C::STAGE_BYTES) + wave_lds;
    if (q < C::PA)
      lds_dma16_sv(piece_base<TA>(Ar, lda, m0, k0), voa[q], s + q * NT * 16);
    else if (q < C::PA + C::PB)
      lds_dma16_sv(piece_base<TB>(Br, ldb, n0, k0), vob[q - C::PA], s + C::A_BYTES + (q - C::PA) * NT * 16);
    else if (q < 2 * C::PA + C::PB)
      lds_dma16_sv(piece_base<TA>(Ai, lda, m0, k0), voa[q - C::PA - C::PB],
                   s + C::A_BYTES + C::B_BYTES + (q - C::PA - C::PB) * NT * 16);
    else
      lds_dma16_sv(piece_base<TB>(Bi, ldb, n0, k0), vob[q - 2 * C::PA - C::PB],
                   s + 2 * C::A_BYTES + C::B_BYTES + (q - 2 * C::PA - C::PB) * NT * 16);
  };
  auto stage_all = [&](int buf, int k0) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < C::LOADS; ++q) stage_q(buf, k0, q);
  };

  auto a_frag = [&](const char* plane, int i, int ks) __attribute__((always_inline)) -> bf16x8 {
    if (TA) return frag_t<BM>(plane, wm + i * 32 + 16 * lg, ks * 16 + 8 * lk, l15);
    return frag_n(plane, wm + i * 32 + l31, ks * 2 + lk);
  };
  auto b_frag = [&](const char* plane, int j, int ks) __attribute__((always_inline)) -> bf16x8 {
    if (TB) return frag_t<BN>(plane, wn + j * 32 + 16 * lg, ks * 16 + 8 * lk, l15);
    return frag_n(plane, wn + j * 32 + l31, ks * 2 + lk);
  };

  const int klen = g.splits > 1 ? ((g.K - kbase) < g.kchunk ? (g.K - kbase) : g.kchunk) : g.K;
  // pinned to a scalar register: in the real kernel the compiler otherwise carries the trip count (and
  // with it the clamped K position of every LDS-DMA piece) in VGPRs
  const int nt = __builtin_amdgcn_readfirstlane(klen / BK);
  {
    // Rolling half-tile pipeline: the barrier sits in the MIDDLE of a tile, when the wave still
    // holds the fragments of the tile's second K sub-step in registers, so the MFMA pipe keeps
    // running across the barrier and across the LDS latency of the next tile's first fragments.
    //   S1 read F[1] <- (tile t, ks 1)           S2 16 MFMAs on F[0] + second half of tile t+2's pieces
    //   S3 lgkmcnt(0), vmcnt (tile t+1 landed), s_barrier   (slot of tile t is free: all in registers)
    //   S4 read F[0] <- (tile t+1, ks 0)         S5 16 MFMAs on F[1] + first half of tile t+3's pieces
    // (S1 / S4 are not bursts: their ds_reads are dealt out behind the MFMA groups of S2 / S5)
    bf16x8 ar[2][IB], br[2][JB], ai[2][IB], bi[2][JB];       // [ks][block]
    auto read_half = [&](int buf, int ks) __attribute__((always_inline)) {
      const char* sA = smem + buf * C::STAGE_BYTES;
      const char* sB = sA + C::A_BYTES;
      const char* sAi = sB + C::B_BYTES;
      const char* sBi = sAi + C::A_BYTES;
#pragma unroll
      for (int i = 0; i < IB; ++i) {
        ar[ks][i] = a_frag(sA, i, ks);
        if (CPLX) ai[ks][i] = a_frag(sAi, i, ks);
      }
#pragma unroll
      for (int j = 0; j < JB; ++j) {
        br[ks][j] = b_frag(sB, j, ks);
        if (CPLX) bi[ks][j] = b_frag(sBi, j, ks);
      }
    };
    // one fragment of (slot buf, sub-step ks), in the order the MFMA groups need them
    constexpr int NFRAG = CPLX ? 2 * IB + 4 : IB + JB;
    auto read_one = [&](int buf, int ks, int idx) __attribute__((always_inline)) {
      const char* sA = smem + buf * C::STAGE_BYTES;
      const char* sB = sA + C::A_BYTES;
      const char* sAi = sB + C::B_BYTES;
      const char* sBi = sAi + C::A_BYTES;
      if (CPLX) {
        // B0 Bi0 A0 Ai0 B1 Bi1 A1 Ai1 A2 Ai2 ...
        if (idx == 0) br[ks][0] = b_frag(sB, 0, ks);
        else if (idx == 1) bi[ks][0] = b_frag(sBi, 0, ks);
        else if (idx == 2) ar[ks][0] = a_frag(sA, 0, ks);
        else if (idx == 3) ai[ks][0] = a_frag(sAi, 0, ks);
        else if (idx == 4) br[ks][1] = b_frag(sB, 1, ks);
        else if (idx == 5) bi[ks][1] = b_frag(sBi, 1, ks);
        else if ((idx & 1) == 0) ar[ks][(idx - 4) / 2] = a_frag(sA, (idx - 4) / 2, ks);
        else ai[ks][(idx - 5) / 2] = a_frag(sAi, (idx - 5) / 2, ks);
      } else {
        // B0 A0 B1 A1, then the remaining B blocks, then the remaining A blocks
        if (idx == 0) br[ks][0] = b_frag(sB, 0, ks);
        else if (idx == 1) ar[ks][0] = a_frag(sA, 0, ks);
        else if (idx == 2) br[ks][1] = b_frag(sB, 1, ks);
        else if (idx == 3) ar[ks][1] = a_frag(sA, 1, ks);
        else if (idx < 2 + JB) br[ks][idx - 2] = b_frag(sB, idx - 2, ks);
        else ar[ks][idx - JB] = a_frag(sA, idx - JB, ks);
      }
    };
    constexpr int H = (C::LOADS + 1) / 2;                     // pieces issued in S5; the rest in S2
    auto mfma_half = [&](int ks, int slot, int tile, int q0, int q1, int rbuf, int rks) __attribute__((always_inline)) {
      bf16x8 nai[IB];
      if (CPLX) {
#pragma unroll
        for (int i = 0; i < IB; ++i) nai[i] = neg_frag(CONJ ? ar[ks][i] : ai[ks][i]);
      }
      int q = q0;
      // tiles past the end re-load the last one into a free slot: no branch in the K loop
      const int k0s = (tile < nt ? tile : nt - 1) * BK;
      // complex: first products of all blocks, then the second ones (see gemm_bf16_persist.h)
      if constexpr (CPLX) {
        constexpr int NG = 2 * IB * JB;
#pragma unroll
        for (int ph = 0; ph < 2; ++ph)
#pragma unroll
          for (int i = 0; i < IB; ++i)
#pragma unroll
            for (int j = 0; j < JB; ++j) {
              if (ph == 0) {
                acc_r[i][j] = CPLXAMD_MFMA16(br[ks][j], ar[ks][i], acc_r[i][j]);
                acc_i[i][j] = CPLXAMD_MFMA16(br[ks][j], ai[ks][i], acc_i[i][j]);
              } else if (CONJ) {
                acc_r[i][j] = CPLXAMD_MFMA16(bi[ks][j], ai[ks][i], acc_r[i][j]);
                acc_i[i][j] = CPLXAMD_MFMA16(bi[ks][j], nai[i], acc_i[i][j]);
              } else {
                acc_r[i][j] = CPLXAMD_MFMA16(bi[ks][j], nai[i], acc_r[i][j]);
                acc_i[i][j] = CPLXAMD_MFMA16(bi[ks][j], ar[ks][i], acc_i[i][j]);
              }
              constexpr int PER = (NFRAG + NG - 1) / NG;
              const int g0 = (ph * IB * JB + i * JB + j) * PER;
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int r = 0; r < PER; ++r)
                if (g0 + r < NFRAG) read_one(rbuf, rks, g0 + r);
              __builtin_amdgcn_sched_barrier(0);
              if (q < q1) {
                __builtin_amdgcn_sched_barrier(0);
                stage_q(slot, k0s, q);
                __builtin_amdgcn_sched_barrier(0);
                ++q;
              }
            }
        return;
      }
#pragma unroll
      for (int i = 0; i < IB; ++i)
#pragma unroll
        for (int j = 0; j < JB; ++j) {
          acc_r[i][j] = CPLXAMD_MFMA16(br[ks][j], ar[ks][i], acc_r[i][j]);
          if (CPLX) {
            acc_i[i][j] = CPLXAMD_MFMA16(br[ks][j], ai[ks][i], acc_i[i][j]);
            if (CONJ) {
              acc_r[i][j] = CPLXAMD_MFMA16(bi[ks][j], ai[ks][i], acc_r[i][j]);
              acc_i[i][j] = CPLXAMD_MFMA16(bi[ks][j], nai[i], acc_i[i][j]);
            } else {
              acc_r[i][j] = CPLXAMD_MFMA16(bi[ks][j], nai[i], acc_r[i][j]);
              acc_i[i][j] = CPLXAMD_MFMA16(bi[ks][j], ar[ks][i], acc_i[i][j]);
            }
          }
          {  // the other half's fragments: a few ds_reads behind every MFMA group instead of one burst of
             // 8-16 (keeps the LDS command FIFO from filling: +1.5 % on the T-operand shapes, 0 elsewhere)
            constexpr int PER = (NFRAG + IB * JB - 1) / (IB * JB);
            const int g0 = (i * JB + j) * PER;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < PER; ++r)
              if (g0 + r < NFRAG) read_one(rbuf, rks, g0 + r);
            __builtin_amdgcn_sched_barrier(0);
          }
          if (q < q1) {
            __builtin_amdgcn_sched_barrier(0);
            stage_q(slot, k0s, q);
            __builtin_amdgcn_sched_barrier(0);
            ++q;
          }
        }
    };
    stage_all(0, 0);
    if (nt > 1) stage_all(1, BK);
    apply_bias();      // (the compiler waits for the bias loads here; they are older than the LDS-DMA above)
    if (nt > 1) wait_vmcnt<C::LOADS>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    read_half(0, 0);
#pragma unroll
    for (int q = 0; q < H; ++q) stage_q(2, (nt > 2 ? 2 : nt - 1) * BK, q);
    // one K tile in ring slot `cur`
    auto tile_body = [&](int cur, int nx1, int nx2, int t) __attribute__((always_inline)) {
      mfma_half(0, nx2, t + 2, H, C::LOADS, cur, 1);          // S1 + S2
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // S3: this wave's F[1] is in registers
      wait_vmcnt<C::LOADS>();                                 // always LOADS younger pieces in flight
      __builtin_amdgcn_s_barrier();
      mfma_half(1, cur, t + 3, 0, H, nx1, 0);                 // S4 + S5 (slot of tile t is free now; past the end the reads hit a stale slot, unused)
    };
    // (-2.0 % on the three bench launches, same-process A/B, profiles/r02_gemm_ablation.md)
    // ring position as a compile-time constant: the K loop is unrolled by the ring depth so that every
    // LDS address is (per-lane base of the slot) + immediate and every M0 value one scalar add
    int t = 0;
    for (; t + 3 <= nt; t += 3) {
      tile_body(0, 1, 2, t);
      tile_body(1, 2, 0, t + 1);
      tile_body(2, 0, 1, t + 2);
    }
    if (t < nt) tile_body(0, 1, 2, t);
    if (t + 1 < nt) tile_body(1, 2, 0, t + 1);
  }

#ifdef CPLXAMD_GEMM_F16
  if (g.scale_a) {              // scaled split products (gemm.h): exact, the scales are powers of two
    const float alpha = gemm_alpha(g);
#pragma unroll
    for (int i = 0; i < IB; ++i)
#pragma unroll
      for (int j = 0; j < JB; ++j) {
        acc_r[i][j] *= alpha;
        if (CPLX) acc_i[i][j] *= alpha;
      }
  }
#endif

  // ---- epilogue.  Transposed 32x32 C/D layout: output row = lane & 31, output columns
  // 8 q + 4 (lane >> 5) + {0..3} for register group q: one 8-B (bf16) / 16-B (fp32) store each.
  TOUT* cr = reinterpret_cast<TOUT*>(g.c_r);
  TOUT* ci = reinterpret_cast<TOUT*>(g.c_i);
  const float beta = gemm_beta(g);
  if (g.splits > 1) {  // fp32 partial slabs [split][plane][M][ldc]; bias / emul applied by the reducer
    const int64_t slab = (int64_t)g.M * g.ldc;
    cr = reinterpret_cast<TOUT*>(g.ws) + (int64_t)split * (CPLX ? 2 : 1) * slab;
    ci = cr + slab;
  }
  // bf16 output without a fused elementwise operand: the wave's tile goes through LDS so that every
  // global store instruction writes whole 128-B lines (8 rows x 64 columns); from the MFMA C layout a
  // store instruction scatters 64 eight-byte pieces over 32 rows (measured: 6.5 % of the complex and
  // 10 % of the real kernel, `profiles/r01_gemm_variants.md`).  One plane and 64 rows per round, each
  // wave in its own 9 KiB of the (now idle) ring.
  if constexpr (sizeof(TOUT) == 2) {
    const bool lds_epi = JB == 2 && g.lds_epilogue && !g.g1 && !g.emul && !g.accumulate && g.splits <= 1 && (g.ldc & 7) == 0 &&
                         (reinterpret_cast<uintptr_t>(cr) & 15) == 0 &&
                         (!CPLX || (reinterpret_cast<uintptr_t>(ci) & 15) == 0);
    if (lds_epi) {
      constexpr int PITCH = 144;                       // bytes per staged row (64 bf16 + 16 B pad)
      wait_vmcnt<0>();                                  // (the clamped LDS-DMA pieces of the loop tail)
      __syncthreads();                                  // every wave is done with the ring
      char* reg = smem + wid * (64 * PITCH);
#pragma unroll
      for (int pl = 0; pl < (CPLX ? 2 : 1); ++pl) {
        TOUT* out = pl ? ci : cr;
        const float* bias = pl ? g.bias_i : g.bias_r;
#pragma unroll
        for (int ih = 0; ih < IB / 2; ++ih) {
#pragma unroll
          for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int j = 0; j < JB; ++j)
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const int cl = j * 32 + 8 * q + 4 * lk;
                const int col = n0 + wn + cl;
                f4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  const float b = (bias && !bias_in_acc && col + e < g.N) ? bias[col + e] : 0.f;
                  v.v[e] = (pl ? acc_i[CPLX ? ih * 2 + ii : 0][j][4 * q + e] : acc_r[ih * 2 + ii][j][4 * q + e]) + b;
                }
                st4(reinterpret_cast<bf16_t*>(reg + (ii * 32 + l31) * PITCH + cl * 2), v);
              }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's tile is in LDS (in-order LDS, own region)
#pragma unroll
          for (int pass = 0; pass < 8; ++pass) {
            const int rl = pass * 8 + (lane >> 3), c8 = (lane & 7) * 8;
            const uint4 v = *reinterpret_cast<const uint4*>(reg + rl * PITCH + c8 * 2);
            const int row = m0 + wm + ih * 64 + rl, col = n0 + wn + c8;
            if (row < g.M && col < g.N) {
              bf16_t* o = reinterpret_cast<bf16_t*>(out) + (int64_t)row * g.ldc + col;
              if (col + 7 < g.N) {
                if (g.fga) {
                  // the LRT input gradient's elementwise term (gemm.h: fga) on the one-tile kernel -- the form the
                  // data-parallel exchange and partial-tile shapes run; arithmetic of gemm_bf16_persist.h's FUSE
                  // epilogue / util.hip dx_accum_kernel: bit-identical
                  const int64_t fo = (int64_t)row * g.fld + col;
                  const uint4 gv = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(g.fga) + fo);
                  const uint4 xv = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(pl ? g.fx_i : g.fx_r) + fo);
                  const uint32_t vw[4] = {v.x, v.y, v.z, v.w}, xw[4] = {xv.x, xv.y, xv.z, xv.w}, gw[4] = {gv.x, gv.y, gv.z, gv.w};
                  uint32_t ow[4];
#pragma unroll
                  for (int e = 0; e < 4; ++e) {
                    const float d0 = __uint_as_float(vw[e] << 16), d1 = __uint_as_float(vw[e] & 0xffff0000u);
                    const float x0 = __uint_as_float(xw[e] << 16), x1 = __uint_as_float(xw[e] & 0xffff0000u);
                    const float g0 = __uint_as_float(gw[e] << 16), g1 = __uint_as_float(gw[e] & 0xffff0000u);
                    ow[e] = pack_bf16(fmaf(2.0f * x0, g0, d0), fmaf(2.0f * x1, g1, d1));
                  }
                  nt_store16(o, uint4{ow[0], ow[1], ow[2], ow[3]});
                } else {
                  nt_store16(o, v);
                }
              } else {
                const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                  if (col + 2 * w < g.N) o[2 * w] = (bf16_t)(w4[w] & 0xffffu);
                  if (col + 2 * w + 1 < g.N) o[2 * w + 1] = (bf16_t)(w4[w] >> 16);
                }
              }
            }
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // reads done before the next round overwrites
        }
      }
      return;
    }
  }
  // float32 output: the same through LDS, 32 rows (one MFMA block row) per round -- a store instruction
  // then writes 4 rows x 256 B instead of 32 rows x 32 B; the elementwise multiplier (LRT log_sigma2
  // gradient) and the accumulate operand are read row-major at the same point.
  if constexpr (sizeof(TOUT) == 4) {
    const bool lds_epi = JB == 2 && g.lds_epilogue && !g.g1 && g.splits <= 1 && (g.ldc & 3) == 0 &&
                         (reinterpret_cast<uintptr_t>(cr) & 15) == 0 &&
                         (!CPLX || (reinterpret_cast<uintptr_t>(ci) & 15) == 0) &&
                         (!g.emul || (reinterpret_cast<uintptr_t>(g.emul) & 15) == 0);
    if (lds_epi) {
      constexpr int PITCH = 272;                       // bytes per staged row (64 floats + 16 B pad)
      wait_vmcnt<0>();
      __syncthreads();
      char* reg = smem + wid * (32 * PITCH);
#pragma unroll
      for (int pl = 0; pl < (CPLX ? 2 : 1); ++pl) {
        float* out = reinterpret_cast<float*>(pl ? ci : cr);
        const float* bias = pl ? g.bias_i : g.bias_r;
#pragma unroll
        for (int i = 0; i < IB; ++i) {
#pragma unroll
          for (int j = 0; j < JB; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int cl = j * 32 + 8 * q + 4 * lk;
              const int col = n0 + wn + cl;
              f4 v;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float b = (bias && !bias_in_acc && col + e < g.N) ? bias[col + e] : 0.f;
                v.v[e] = (pl ? acc_i[CPLX ? i : 0][j][4 * q + e] : acc_r[i][j][4 * q + e]) + b;
              }
              st4(reinterpret_cast<float*>(reg + l31 * PITCH + cl * 4), v);
            }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
          for (int pass = 0; pass < 8; ++pass) {
            const int rl = pass * 4 + (lane >> 4), c4 = (lane & 15) * 4;
            f4 v = ld4(reinterpret_cast<const float*>(reg + rl * PITCH + c4 * 4));
            const int row = m0 + wm + i * 32 + rl, col = n0 + wn + c4;
            if (row < g.M && col < g.N) {
              const int64_t o = (int64_t)row * g.ldc + col;
              if (col + 3 < g.N) {
                if (g.emul && (!pl || g.emul_both)) {
                  const f4 m = ld4(g.emul + o);
#pragma unroll
                  for (int e = 0; e < 4; ++e) v.v[e] *= gemm_emul(g, m.v[e]);
                }
                if (g.accumulate) {
                  const f4 p = ld4(out + o);
#pragma unroll
                  for (int e = 0; e < 4; ++e) v.v[e] += beta * p.v[e];
                }
                nt_store16(out + o, v);
              } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                  if (col + e < g.N) {
                    float x = v.v[e];
                    if (g.emul && (!pl || g.emul_both)) x *= gemm_emul(g, g.emul[o + e]);
                    if (g.accumulate) x += beta * out[o + e];
                    out[o + e] = x;
                  }
              }
            }
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
      }
      return;
    }
  }
  const bool two_planes = CPLX || g.g1;
  const bool vec_ok = (g.ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(cr) & 15) == 0 &&
                      (!two_planes || (reinterpret_cast<uintptr_t>(ci) & 15) == 0) &&
                      (!g.g1 || (g.N & 3) == 0) &&
                      (!g.emul || (reinterpret_cast<uintptr_t>(g.emul) & 15) == 0);
  // one 32-row block of the wave tile; `i` is a compile-time constant (an `#pragma unroll`ed loop
  // over i was left rolled by the compiler for some of the 128-row real variants, which sent the
  // whole accumulator array to scratch)
  auto store_block = [&](auto I) __attribute__((always_inline)) {
    constexpr int i = decltype(I)::value;
    const int row = m0 + wm + i * 32 + l31;
    if (row >= g.M) return;
#pragma unroll
    for (int j = 0; j < JB; ++j) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = n0 + wn + j * 32 + 8 * q + 4 * lk;
        if (col >= g.N) continue;
        const int64_t o = (int64_t)row * g.ldc + col;
        f4 vr, vi;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          vr.v[e] = acc_r[i][j][4 * q + e];
          vi.v[e] = CPLX ? acc_i[CPLX ? i : 0][j][4 * q + e] : 0.f;
        }
        if (!CPLX && g.g1) {                       // Gauss 3M combine (see gemm.h)
          const int64_t od = (int64_t)row * g.N + col;
          if (vec_ok && col + 3 < g.N) {
            const f4 p = ld4(g.g1 + od), t2 = ld4(g.g2 + od);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float q = g.gsign * t2.v[e];
              vi.v[e] = vr.v[e] - p.v[e] - q;
              vr.v[e] = p.v[e] - q;
            }
            if (g.bias_r) {
              const f4 b = ld4(g.bias_r + col), c = ld4(g.bias_i + col);
#pragma unroll
              for (int e = 0; e < 4; ++e) { vr.v[e] += b.v[e]; vi.v[e] += c.v[e]; }
            }
            st4(cr + o, vr);
            st4(ci + o, vi);
          } else {
            for (int e = 0; e < 4 && col + e < g.N; ++e) {
              const float p = g.g1[od + e], q = g.gsign * g.g2[od + e];
              io<TOUT>::st(cr + o + e, p - q + (g.bias_r ? g.bias_r[col + e] : 0.f));
              io<TOUT>::st(ci + o + e, vr.v[e] - p - q + (g.bias_i ? g.bias_i[col + e] : 0.f));
            }
          }
          continue;
        }
        if (vec_ok && col + 3 < g.N) {
          if (g.bias_r && !bias_in_acc) {
            const f4 b = ld4(g.bias_r + col);
#pragma unroll
            for (int e = 0; e < 4; ++e) vr.v[e] += b.v[e];
            if (CPLX) {
              const f4 c = ld4(g.bias_i + col);
#pragma unroll
              for (int e = 0; e < 4; ++e) vi.v[e] += c.v[e];
            }
          }
          if (g.emul) {
            const f4 m = ld4(g.emul + o);
#pragma unroll
            for (int e = 0; e < 4; ++e) vr.v[e] *= gemm_emul(g, m.v[e]);
            if (CPLX && g.emul_both) {
#pragma unroll
              for (int e = 0; e < 4; ++e) vi.v[e] *= gemm_emul(g, m.v[e]);
            }
          }
          if (g.accumulate) {
            const f4 p = ld4(cr + o);
#pragma unroll
            for (int e = 0; e < 4; ++e) vr.v[e] += beta * p.v[e];
            if (CPLX) {
              const f4 p2 = ld4(ci + o);
#pragma unroll
              for (int e = 0; e < 4; ++e) vi.v[e] += beta * p2.v[e];
            }
          }
          st4(cr + o, vr);
          if (CPLX) st4(ci + o, vi);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (col + e >= g.N) break;
            float xr = vr.v[e] + ((g.bias_r && !bias_in_acc) ? g.bias_r[col + e] : 0.f);
            if (g.emul) xr *= gemm_emul(g, g.emul[o + e]);
            if (g.accumulate) xr += beta * io<TOUT>::ld(cr + o + e);
            io<TOUT>::st(cr + o + e, xr);
            if (CPLX) {
              float xi = vi.v[e] + ((g.bias_i && !bias_in_acc) ? g.bias_i[col + e] : 0.f);
              if (g.emul && g.emul_both) xi *= gemm_emul(g, g.emul[o + e]);
              if (g.accumulate) xi += beta * io<TOUT>::ld(ci + o + e);
              io<TOUT>::st(ci + o + e, xi);
            }
          }
        }
      }
    }
  };
  store_block(std::integral_constant<int, 0>{});
  store_block(std::integral_constant<int, 1>{});
  if constexpr (IB == 4) {
    store_block(std::integral_constant<int, 2>{});
    store_block(std::integral_constant<int, 3>{});
  }
  wait_vmcnt<0>();   // the branch-free K loop leaves (unused) LDS-DMA pieces in flight: land them before the LDS is released
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

static int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

template <typename TOUT, bool CPLX, bool CONJ, bool TA, bool TB>
static int launch_kernel_r(const GemmArgs& g0, hipStream_t st) {
  using C = Cfg<CPLX>;
  // read-only tuning knobs, set once from the environment (A/B experiments only)
  static const int order = env_int("CPLXAMD_GEMM_ORDER", 1), gm = env_int("CPLXAMD_GEMM_GROUP_M", 4);
  GemmArgs g = g0;
  static const int ldsepi = env_int("CPLXAMD_GEMM_LDSEPI", 1);   // A/B switch of the LDS-staged epilogue
  g.lds_epilogue = ldsepi;
  g.order = order; g.group_m = gm > 0 ? gm : 1;
  const int64_t tiles = (int64_t)((g.M + C::BM - 1) / C::BM) * ((g.N + C::BN - 1) / C::BN);
  if (tiles * g.splits > 0x7fffffff) return CPLXAMD_ESHAPE;
  if (g.plan) { *g.plan = 1; return 0; }
  static PerDeviceOnce attr_set;
  if (const int e = set_max_dyn_lds(attr_set, gemm_bf16_kernel<TOUT, CPLX, CONJ, TA, TB>, C::SMEM)) return e;
  gemm_bf16_kernel<TOUT, CPLX, CONJ, TA, TB><<<dim3((unsigned)(tiles * g.splits)), C::NT, C::SMEM, st>>>(g);
  CPLXAMD_CHECK_LAUNCH();
  return 0;
}

}  // namespace cplxamd
#include "gemm_bf16_persist.h"
namespace cplxamd {

// persistent form (gemm_bf16_persist.h): more than one round of full tiles, plain (bias-only) epilogue
template <typename TOUT, bool CPLX, bool CONJ, bool TA, bool TB>
static int launch_persist(const GemmArgs& g0, hipStream_t st, bool& taken) {
  using C = Cfg<CPLX>;
  taken = false;
  static const int enabled = env_int("CPLXAMD_GEMM_PERSIST", 1);     // (A/B at run time)
  const GemmArgs& g = g0;
  if (!enabled || !launch_owns_chip(g.flags)) return 0;
  const int ncu = (g.ncu > 0 ? g.ncu : device_cus()) & ~7;
  if (g.splits > 1 || g.g1 || g.emul || g.accumulate || g.scale_a || (g.M % C::BM) || (g.N % C::BN) || g.K / BK < 12) return 0;
  if (g.fga && (!((CPLX ? CONJ : true) && TB && !TA && sizeof(TOUT) == 2) || (g.fld & 7) || !aligned16(g.fga) ||
                !aligned16(g.fx_r) || (CPLX && !aligned16(g.fx_i)) || g.bias_r)) return 0;   // (the caller runs the two-kernel path)
  // instantiated for the layouts the layers launch with a plain epilogue: forward (N,N), input gradient (N,T)
  // -- the weight gradients carry the fused KL accumulate and stay on the one-tile kernel
  constexpr bool kBf16Out = sizeof(TOUT) == 2;
  // (real: (N,N) with either output type -- the bf16 layers' mean GEMM and, since they keep the variance in bf16, their
  //  variance GEMM; the float32-out (N,N) form serves float32 s2 callers -- and (N,T) bf16)
  constexpr bool kInstantiated = !TA && (CPLX ? (CONJ == TB && kBf16Out) : (kBf16Out || !TB));
  if (!kInstantiated) return 0;
  const int64_t tiles = (int64_t)(g.M / C::BM) * (g.N / C::BN);
  if (tiles <= ncu || tiles > 0x7fffffff) return 0;
  const int align = sizeof(TOUT) == 2 ? 7 : 3;
  if ((g.ldc & align) || !aligned16(g.c_r) || (CPLX && !aligned16(g.c_i))) return 0;
  if (g.bias_r && (!aligned16(g.bias_r) || (CPLX && !aligned16(g.bias_i)))) return 0;
  constexpr int smem = 3 * C::STAGE_BYTES + 16384;
  GemmArgs a = g;
  static const int gm = env_int("CPLXAMD_GEMM_GROUP_M", 4);
  a.group_m = gm > 0 ? gm : 1;
  if constexpr (kInstantiated) {
    constexpr bool kFusable = (CPLX ? CONJ : true) && TB && kBf16Out;   // the LRT input gradient, complex and real (gemm.h: fga)
    auto go = [&](auto RR) -> int {
      constexpr int R = decltype(RR)::value;
      if (g.plan) { *g.plan = 2; return 0; }
      if constexpr (kFusable) {
        if (g.fga) {
          static PerDeviceOnce attr_set_f;
          if (const int e = set_max_dyn_lds(attr_set_f, gemm_bf16_persist_kernel<TOUT, CPLX, CONJ, TA, TB, R, true>, smem)) return e;
          gemm_bf16_persist_kernel<TOUT, CPLX, CONJ, TA, TB, R, true><<<dim3((unsigned)ncu), C::NT, smem, st>>>(a);
          CPLXAMD_CHECK_LAUNCH();
          return 0;
        }
      }
      static PerDeviceOnce attr_set;
      if (const int e = set_max_dyn_lds(attr_set, gemm_bf16_persist_kernel<TOUT, CPLX, CONJ, TA, TB, R>, smem)) return e;
      gemm_bf16_persist_kernel<TOUT, CPLX, CONJ, TA, TB, R><<<dim3((unsigned)ncu), C::NT, smem, st>>>(a);
      CPLXAMD_CHECK_LAUNCH();
      return 0;
    };
    const int r = (g.K / BK) % 3;
    const int rc = r == 0 ? go(std::integral_constant<int, 0>{}) : r == 1 ? go(std::integral_constant<int, 1>{})
                                                                          : go(std::integral_constant<int, 2>{});
    if (rc) return rc;
    taken = true;
  }
  return 0;
}

template <typename TOUT, bool CPLX, bool CONJ, bool TA, bool TB>
static int launch_kernel(const GemmArgs& g, hipStream_t st) {
  {
    // round 4: the one-wave-per-SIMD family first (gemm_bf16_w4.hip; same bits)
    bool taken = false;
    const int rc = launch_gemm_bf16_w4(g, CPLX, sizeof(TOUT) == 2 ? CPLXAMD_BF16 : CPLXAMD_F32, TA, TB, st, taken);
    if (rc || taken) return rc;
  }
  {
    bool taken = false;
    const int rc = launch_persist<TOUT, CPLX, CONJ, TA, TB>(g, st, taken);
    if (rc || taken) return rc;
  }
  if (g.fga) {
    // the one-tile kernel carries the fused term in its LDS-staged bf16 epilogue only: the launch must be one that takes
    // that path with whole 16-byte column groups (same predicate as in the kernel), else decline -- never drop it silently
    static const int ldsepi = env_int("CPLXAMD_GEMM_LDSEPI", 1);
    const bool ok = sizeof(TOUT) == 2 && Cfg<CPLX>::JB == 2 && ldsepi && !g.g1 && !g.emul && !g.accumulate && g.splits <= 1 &&
                    !g.bias_r && (g.ldc & 7) == 0 && (g.fld & 7) == 0 && (g.N & 7) == 0 && aligned16(g.c_r) &&
                    (!CPLX || aligned16(g.c_i)) && aligned16(g.fga) && aligned16(g.fx_r) && (!CPLX || aligned16(g.fx_i));
    if (!ok) return CPLXAMD_ESHAPE;
  }
  return launch_kernel_r<TOUT, CPLX, CONJ, TA, TB>(g, st);
}

template <typename TOUT, bool CPLX, bool CONJ>
static int launch_layout(const GemmArgs& g, bool ta, bool tb, hipStream_t st) {
  if (ta) return tb ? launch_kernel<TOUT, CPLX, CONJ, true, true>(g, st)
                    : launch_kernel<TOUT, CPLX, CONJ, true, false>(g, st);
  return tb ? launch_kernel<TOUT, CPLX, CONJ, false, true>(g, st)
            : launch_kernel<TOUT, CPLX, CONJ, false, false>(g, st);
}

template <bool CPLX>
static int launch_dtype(const GemmArgs& g, int out_dtype, bool ta, bool tb, hipStream_t st) {
  if (out_dtype != CPLXAMD_BF16 && out_dtype != CPLXAMD_F32) return CPLXAMD_EINVAL;
  const bool f32 = out_dtype == CPLXAMD_F32;
#ifdef CPLXAMD_GEMM_F16            // the half-operand build: float32 output only
  if (!f32) return CPLXAMD_ESHAPE;
  if constexpr (CPLX) {
    if (g.conj_b) return launch_layout<float, true, true>(g, ta, tb, st);
  }
  return launch_layout<float, CPLX, false>(g, ta, tb, st);
#else
  if constexpr (CPLX) {
    if (g.conj_b)
      return f32 ? launch_layout<float, true, true>(g, ta, tb, st)
                 : launch_layout<bf16_t, true, true>(g, ta, tb, st);
  }
  return f32 ? launch_layout<float, CPLX, false>(g, ta, tb, st)
             : launch_layout<bf16_t, CPLX, false>(g, ta, tb, st);
#endif
}

// split-K plan: use it when the tile count leaves CUs idle and K is long
static int plan_splits(int M, int N, int K, bool cplx) {
  const int bm = cplx ? Cfg<true>::BM : Cfg<false>::BM, bn = cplx ? Cfg<true>::BN : Cfg<false>::BN;
  const int64_t tiles = (int64_t)((M + bm - 1) / bm) * ((N + bn - 1) / bn);
  if (tiles >= 192 || K < 64 * BK) return 1;
  int s = (int)(256 / tiles);
  const int maxs = K / (32 * BK);          // >= 32 K tiles per split
  if (s > maxs) s = maxs;
  if (s > 16) s = 16;
  return s < 2 ? 1 : s;
}

#if GEMM_BF16_TU == 1 && !defined(CPLXAMD_GEMM_F16)
int64_t gemm_bf16_ws_bytes(int M, int N, int K, bool cplx) {
  const int s = plan_splits(M, N, K, cplx);
  return s > 1 ? (int64_t)s * (cplx ? 2 : 1) * M * N * (int64_t)sizeof(float) : 0;
}
#endif

// out = sum_s slab[s] (+ bias[n]) (* emul) (+ out)   for one plane
static __global__ __launch_bounds__(256) void gemm_slab_reduce_kernel(const float* slabs, int splits,
                                                               int64_t slab_stride, int M, int N,
                                                               int64_t ldc, const float* bias,
                                                               const float* emul, int emul_exp, int accumulate,
                                                               const float* beta_p, float* out) {
  const float beta = (accumulate && beta_p) ? *beta_p : 1.0f;
  const int64_t n4 = ((int64_t)M * ldc) >> 2;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    f4 acc = ld4(slabs + 4 * i);
    for (int s = 1; s < splits; ++s) {
      const f4 v = ld4(slabs + (int64_t)s * slab_stride + 4 * i);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc.v[e] += v.v[e];
    }
    const int col = (int)((4 * i) % ldc);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (bias && col + e < N) acc.v[e] += bias[col + e];
    }
    if (emul) {
      const f4 m = ld4(emul + 4 * i);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc.v[e] *= emul_exp ? expf(m.v[e]) : m.v[e];
    }
    if (accumulate) {
      const f4 o = ld4(out + 4 * i);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc.v[e] += beta * o.v[e];
    }
    st4(out + 4 * i, acc);
  }
}

template <bool CPLX>
static int launch_splitk(const GemmArgs& g0, int splits, bool ta, bool tb, hipStream_t st) {
  GemmArgs g = g0;
  g.splits = splits;
  g.kchunk = ((g.K / BK + splits - 1) / splits) * BK;
  // slabs are dense (ldc = N) float partials, no bias / emul / accumulate
  GemmArgs k = g;
  k.ldc = g.N; k.bias_r = k.bias_i = nullptr; k.emul = nullptr; k.accumulate = 0;
  const int rc = launch_dtype<CPLX>(k, CPLXAMD_F32, ta, tb, st);
  if (rc) return rc;
  if (g.plan) { *g.plan = *g.plan == 3 ? 5 : 4; return 0; }
  const int64_t slab = (int64_t)g.M * g.N, stride = (CPLX ? 2 : 1) * slab;
  const int grid = stream_grid(slab >> 2, 256);
  gemm_slab_reduce_kernel<<<grid, 256, 0, st>>>((const float*)g.ws, splits, stride, g.M, g.N, g.ldc,
                                                g.bias_r, g.emul, g.emul_exp, g.accumulate, g.beta, (float*)g.c_r);
  CPLXAMD_CHECK_LAUNCH();
  if (CPLX) {
    gemm_slab_reduce_kernel<<<grid, 256, 0, st>>>((const float*)g.ws + slab, splits, stride, g.M, g.N,
                                                  g.ldc, g.bias_i, g.emul_both ? g.emul : nullptr, g.emul_exp, g.accumulate, g.beta,
                                                  (float*)g.c_i);
    CPLXAMD_CHECK_LAUNCH();
  }
  return 0;
}

template <bool CPLX>
int launch_gemm_bf16(const GemmArgs& g, int out_dtype, hipStream_t st) {
  // operand layouts: K-contiguous rows ("N", *_cs == 1) or K-major ("T", *_rs == 1)
  bool ta, tb;
  if (g.a_cs == 1) ta = false; else if (g.a_rs == 1) ta = true; else return CPLXAMD_ESHAPE;
  if (g.b_cs == 1) tb = false; else if (g.b_rs == 1) tb = true; else return CPLXAMD_ESHAPE;
  if (g.K < BK || (g.K % BK) != 0) return CPLXAMD_ESHAPE;
  const int64_t lda = ta ? g.a_cs : g.a_rs, ldb = tb ? g.b_cs : g.b_rs;
  if ((lda % 8) != 0 || (ldb % 8) != 0) return CPLXAMD_ESHAPE;
  if (lda >= (1 << 22) || ldb >= (1 << 22)) return CPLXAMD_ESHAPE;   // per-lane tile offsets are 32-bit
  if ((ta && ((g.M % 8) != 0 || g.M < 8)) || (tb && ((g.N % 8) != 0 || g.N < 8))) return CPLXAMD_ESHAPE;
  if (!aligned16(g.a_r) || !aligned16(g.b_r)) return CPLXAMD_ESHAPE;
  if (CPLX && (!aligned16(g.a_i) || !aligned16(g.b_i))) return CPLXAMD_ESHAPE;
  if (g.M <= 0 || g.N <= 0) return 0;
  if (out_dtype == CPLXAMD_F32 && g.ws && g.ldc == g.N && (g.N & 3) == 0) {
    const int splits = plan_splits(g.M, g.N, g.K, CPLX);
    if (splits > 1 && g.ws_bytes >= gemm_bf16_ws_bytes(g.M, g.N, g.K, CPLX))
      return launch_splitk<CPLX>(g, splits, ta, tb, st);
  }
  return launch_dtype<CPLX>(g, out_dtype, ta, tb, st);
}

#if GEMM_BF16_TU == 1 && !defined(CPLXAMD_GEMM_F16)
// ---- Gauss 3M: t1 = Ar Br, T2 = Ai Bi, t3 = (Ar + Ai)(Br + s Bi) as three real MFMA GEMMs; the
// combine rides in the third one's epilogue.  25 % fewer MFMAs than 4M, but each real GEMM has half
// the LDS reuse of the fused 4M loop and the operand sums are rounded to bf16.
__global__ __launch_bounds__(256) void gauss_sum_kernel(const bf16_t* r, const bf16_t* i, float sign,
                                                        bf16_t* out, int64_t n8) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < n8; c += stride) {
    const uint4 a = *reinterpret_cast<const uint4*>(r + 8 * c);
    const uint4 b = *reinterpret_cast<const uint4*>(i + 8 * c);
    const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
    uint32_t ow[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float lo = __uint_as_float(aw[e] << 16) + sign * __uint_as_float(bw[e] << 16);
      const float hi = __uint_as_float(aw[e] & 0xffff0000u) + sign * __uint_as_float(bw[e] & 0xffff0000u);
      ow[e] = pack_bf16(lo, hi);
    }
    *reinterpret_cast<uint4*>(out + 8 * c) = uint4{ow[0], ow[1], ow[2], ow[3]};
  }
}

static int64_t align256(int64_t x) { return (x + 255) & ~(int64_t)255; }

int64_t gemm_bf16_gauss_ws_bytes(int M, int N, int K) {
  return align256((int64_t)M * K * 2) + align256((int64_t)N * K * 2) + 2 * (int64_t)M * N * 4;
}

int launch_gemm_bf16_gauss(const GemmArgs& g, int out_dtype, hipStream_t st) {
  bool ta, tb;
  if (g.a_cs == 1) ta = false; else if (g.a_rs == 1) ta = true; else return CPLXAMD_ESHAPE;
  if (g.b_cs == 1) tb = false; else if (g.b_rs == 1) tb = true; else return CPLXAMD_ESHAPE;
  // dense operands only: the sums are formed over the raw storage
  if ((ta ? g.a_cs : g.a_rs) != (ta ? g.M : g.K) || (tb ? g.b_cs : g.b_rs) != (tb ? g.N : g.K))
    return CPLXAMD_ESHAPE;
  if (g.accumulate || (g.N & 3) || ((int64_t)g.M * g.K & 7) || ((int64_t)g.N * g.K & 7))
    return CPLXAMD_ESHAPE;
  if (!g.ws || g.ws_bytes < gemm_bf16_gauss_ws_bytes(g.M, g.N, g.K)) return CPLXAMD_EWS;
  if ((reinterpret_cast<uintptr_t>(g.ws) & 255) != 0) return CPLXAMD_EALIGN;
  if (g.M <= 0 || g.N <= 0) return 0;
  char* w = (char*)g.ws;
  bf16_t* As = (bf16_t*)w; w += align256((int64_t)g.M * g.K * 2);
  bf16_t* Bs = (bf16_t*)w; w += align256((int64_t)g.N * g.K * 2);
  float* t1 = (float*)w; float* t2 = t1 + (int64_t)g.M * g.N;
  const float s = g.conj_b ? -1.0f : 1.0f;
  const int64_t na = (int64_t)g.M * g.K / 8, nb = (int64_t)g.N * g.K / 8;
  gauss_sum_kernel<<<stream_grid(na, 256), 256, 0, st>>>((const bf16_t*)g.a_r, (const bf16_t*)g.a_i,
                                                        1.0f, As, na);
  CPLXAMD_CHECK_LAUNCH();
  gauss_sum_kernel<<<stream_grid(nb, 256), 256, 0, st>>>((const bf16_t*)g.b_r, (const bf16_t*)g.b_i,
                                                        s, Bs, nb);
  CPLXAMD_CHECK_LAUNCH();
  GemmArgs r = g;
  r.a_i = r.b_i = nullptr; r.bias_r = r.bias_i = nullptr; r.conj_b = 0;
  r.ws = nullptr; r.ws_bytes = 0; r.ldc = g.N; r.c_i = nullptr;
  r.c_r = t1;
  int rc = launch_gemm_bf16<false>(r, CPLXAMD_F32, st);
  if (rc) return rc;
  r.a_r = g.a_i; r.b_r = g.b_i; r.c_r = t2;
  rc = launch_gemm_bf16<false>(r, CPLXAMD_F32, st);
  if (rc) return rc;
  GemmArgs f = g;
  f.a_r = As; f.b_r = Bs; f.a_i = f.b_i = nullptr; f.conj_b = 0; f.ws = nullptr; f.ws_bytes = 0;
  f.g1 = t1; f.g2 = t2; f.gsign = s;
  return launch_gemm_bf16<false>(f, out_dtype, st);
}
#endif

#if GEMM_BF16_TU == 1
template int launch_gemm_bf16<false>(const GemmArgs&, int, hipStream_t);
#else
template int launch_gemm_bf16<true>(const GemmArgs&, int, hipStream_t);
#endif

}  // namespace cplxamd
