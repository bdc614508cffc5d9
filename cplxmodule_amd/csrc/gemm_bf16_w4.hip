// One-wave-per-SIMD form of the bf16 complex / real MFMA GEMM (round 4): translation unit 3 of the bf16 GEMM family.
//
//   C[m,n] = sum_k A[m,k] * op(B[n,k]) (+ bias[n]),  planar re / im, fp32 accumulation, bf16 or fp32 output;
//   same operand conventions as gemm_bf16_impl.h ("N" = K-contiguous rows, "T" = K-major as stored), same MFMA order per
//   accumulator block and per K step, hence BIT-IDENTICAL results to the 8-wave kernels (tests/test_gpu_r04.py).
//   Reference semantics: cplx.py:634-648 (linear_naive), nn/relevance/complex/base.py:43-56 (the LRT chain's GEMMs).
//
// Why another kernel (profiles/r03_gemm_pair_issue.txt, r02_gemm_ablation.md).  The 8-wave kernels are
// bound by the REQUEST stream of their LDS-DMA staging: a 32-deep K tile asks for 64 bytes per operand row = half a cache
// line, the other half a K tile (48 KiB of other lines) later: 2426 cycles per K tile against 2048 of MFMA issue.  Halves
// requested back to back merge in flight (2026 cycles) -- but that needs the landing space of TWO K tiles free at once, and
// a fourth 48-KiB LDS slot does not exist.  The register file does: at one wave per SIMD a wave owns 512 registers, the
// 128 x 64 complex wave tile takes 256 of them as accumulators (the AGPR half), and 96 more hold a PAIR of K tiles in
// flight.  So here
//  * 4 waves (256 threads), one per SIMD, 2 x 2 over a 256 x 128 complex (256 x 256 real) tile: wave tile 128 x 64 complex
//    = 16 blocks of 32 x 32 = 256 accumulators (real: 128 x 128);
//  * operands go global -> VGPR (buffer_load_dwordx4) -> LDS (ds_write_b128): the loads of K tiles (t, t+1) are issued back
//    to back, register pair by register pair, so that both halves of every 128-byte line are requested together;
//  * LDS: ring of three 32-deep K-tile slots (144 KiB complex / 96 KiB real), images identical to the 8-wave kernels'
//    (XOR-swizzled "N" rows read by ds_read_b128, "T" rows read by ds_read_b64_tr_b16), ONE s_barrier per K tile placed in
//    the middle of the tile (rolling half-tile pipeline: fragments of the second K sub-step are already in registers);
//  * the K loop is written slot by slot: after every MFMA a fixed list of "fillers" (one fragment read, one LDS write, one
//    load pair, the sign XORs) pinned with sched_barrier -- at one wave per SIMD nothing else hides an instruction, the
//    matrix pipe has room for <= 5 other issues per 32-cycle MFMA (MI355X_MICROARCH.md, constants table);
//  * fragments double-buffered per K sub-step (104 VGPRs complex), negation for the (-Bi) products on the B fragment
//    (8 XORs per sub-step instead of 16 on the A side; same products, same bits).
// Schedule of K tile u (ring slot u % 3, parity u & 1; "sub" = K sub-step of 16):
//   sub 0: MFMAs on F0 | read F1 <- (tile u, sub 1) | write half of the registers of tile u+2 into slot (u+2) % 3
//   s_waitcnt (F1), s_barrier                       (every wave is done reading tile u-1 ... and has written tile u+1)
//   sub 1: MFMAs on F1 | read F0 <- (tile u+1, sub 0) | write the other half | odd u: re-issue the register pairs with
//          tiles (u+3, u+4) as they drain
// Preconditions (the launcher declines otherwise and the 8-wave kernels run): full tiles, K % 64 == 0, K >= 128, no
// split-K, no Gauss combine, 16-byte aligned operands / outputs.
#include <stdlib.h>

#include <type_traits>

#include "gemm.h"

namespace cplxamd {
namespace w4 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// where the two loads of a register pair go, in MFMA slots behind the pair's LDS write (experiments: scripts/r04/w4_build.sh)
#ifndef W4_LDP
#define W4_LDP 0
#endif
#ifndef W4_WAIT1
#define W4_WAIT1 0   // 1: ONE s_waitcnt lgkmcnt(1) at the head of every K sub-step covers all fragment reads of the previous one
#endif
#ifndef W4_LDQ
#define W4_LDQ 0
#endif

constexpr int BK = 32;

template <bool CPLX>
struct Cfg {
  static constexpr int NT = 256;
  static constexpr int IB = 4, JB = CPLX ? 2 : 4;               // 32 x 32 blocks per wave
  static constexpr int WM = 2, WN = 2;
  static constexpr int BM = 32 * IB * WM, BN = 32 * JB * WN;    // 256 x 128 (complex) / 256 x 256 (real)
  static constexpr int NPL = CPLX ? 2 : 1;
  static constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
  static constexpr int SLOT = NPL * (A_BYTES + B_BYTES);        // 48 / 32 KiB
  static constexpr int SMEM = 3 * SLOT;
  static constexpr int PA = BM * 4 / NT, PB = BN * 4 / NT;       // 16-byte pieces per lane, plane and K tile
  static constexpr int NL = NPL * (PA + PB);                     // ... per K tile (12 / 8)
  static constexpr int NFRAG = NPL * (IB + JB);                  // fragments per K sub-step (12 / 8)
  static constexpr int NM = NPL * NPL * IB * JB;                 // MFMAs per K sub-step (32 / 16)
};

// ---- address maps (identical LDS images to gemm_bf16_impl.h: piece_voff / frag_n / frag_t) -------------------
// piece j of a plane tile, chunk p = j * NT + tid:
//  !T: (row = p >> 2, source chunk c = p & 3) of [rows][32 k]      -> LDS row * 64 + ((c ^ ((row >> 2) & 3)) << 4)
//   T: (k = p / (ROWS/8), source chunk c = p % (ROWS/8)) of [32 k][ROWS] -> LDS k * ROWS*2 + ((c ^ ((k & 3) << 2)) << 4)
template <int ROWS, bool T>
__device__ __forceinline__ uint32_t src_voff(int64_t ld, int tid) {          // per-lane byte offset of piece 0
  if (!T) return (uint32_t)((((int64_t)(tid >> 2)) * ld + (tid & 3) * 8) * 2);
  constexpr int CPR = ROWS / 8;
  return (uint32_t)((((int64_t)(tid / CPR)) * ld + (tid % CPR) * 8) * 2);
}
template <int ROWS, bool T>
__device__ __forceinline__ uint32_t piece_stride(int64_t ld) {               // source bytes between pieces j and j + 1
  constexpr int NT = 256;
  if (!T) return (uint32_t)((NT / 4) * ld * 2);
  return (uint32_t)((NT / (ROWS / 8)) * ld * 2);
}
template <int ROWS, bool T>
__device__ __forceinline__ uint32_t dst_off(int tid) {                        // per-lane LDS byte offset of piece 0
  if (!T) {
    const int row = tid >> 2, c = tid & 3;
    return (uint32_t)(row * 64 + ((c ^ ((row >> 2) & 3)) << 4));
  }
  constexpr int CPR = ROWS / 8;
  const int k = tid / CPR, c = tid % CPR;
  return (uint32_t)(k * (ROWS * 2) + ((c ^ ((k & 3) << 2)) << 4));
}
// (piece j adds j * 4096 bytes in every layout: 64 rows x 64 B, or (256 / CPR) k rows x ROWS * 2 B)

// Fragment reads take this lane's LDS byte address (lane part + ring slot: ONE register per (slot, K sub-step) for an "N"
// operand, per (slot, block) for a "T" operand, made opaque so that the compiler neither re-derives nor hoists variants of
// it) plus a compile-time byte offset that fits the 16-bit offset field (plane, block row, K sub-step: < 48 KiB).
typedef __attribute__((address_space(3))) const bf16x8* lds_frag_p;
typedef __attribute__((address_space(3))) s16x4* lds_tr_p;
typedef __attribute__((address_space(3))) u32x4* lds_st_p;
__device__ __forceinline__ bf16x8 lds_frag_n(uint32_t addr, int imm) { return *(lds_frag_p)(uintptr_t)(addr + (uint32_t)imm); }
// "T" image [32 k][ROWS]: two hardware-transposed 4 x 16 reads, k and k + 4
template <int ROWS>
__device__ __forceinline__ bf16x8 lds_frag_t(uint32_t addr, int imm) {
  const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_p)(uintptr_t)(addr + (uint32_t)imm));
  const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_p)(uintptr_t)(addr + (uint32_t)(imm + 4 * ROWS * 2)));
  const s16x8 both = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, both);
}
// lane parts.  "N": row * 64 + ((kc ^ ((row >> 2) & 3)) << 4) with row = (wave origin + lane & 31), kc = 2 ks + (lane >> 5);
// "T": k * ROWS*2 + (((r >> 3) ^ ((k & 3) << 2)) << 4) + (r & 7) * 2 with k = 8 (lane >> 5) + (m >> 2), r = rb + 4 (m & 3),
// rb = wave origin + 32 blk + 16 ((lane >> 4) & 1), m = lane & 15   (gemm_bf16_impl.h: frag_n / frag_t)
__device__ __forceinline__ uint32_t lane_n(int worg, int lane, int ks) {
  const int row = worg + (lane & 31), kc = ks * 2 + (lane >> 5);
  return (uint32_t)(row * 64 + ((kc ^ ((row >> 2) & 3)) << 4));
}
template <int ROWS>
__device__ __forceinline__ uint32_t lane_t(int worg, int lane, int blk) {
  const int m = lane & 15, k = 8 * (lane >> 5) + (m >> 2);
  const int r = worg + blk * 32 + 16 * ((lane >> 4) & 1) + 4 * (m & 3);
  return (uint32_t)(k * (ROWS * 2) + (((r >> 3) ^ ((k & 3) << 2)) << 4) + (r & 7) * 2);
}
__device__ __forceinline__ uint32_t opaque(uint32_t x) { asm volatile("" : "+v"(x)); return x; }
__device__ __forceinline__ bf16x8 neg_frag(bf16x8 v) {
  uint4 u = __builtin_bit_cast(uint4, v);
  u.x ^= 0x80008000u; u.y ^= 0x80008000u; u.z ^= 0x80008000u; u.w ^= 0x80008000u;
  return __builtin_bit_cast(bf16x8, u);
}
__device__ __forceinline__ void store16(void* p, uint4 v) {
  *reinterpret_cast<uint4*>(p) = v;
}
__device__ __forceinline__ void store16(float* p, const f4& a) {
  st4(p, a);
}

#define W4_SB() __builtin_amdgcn_sched_barrier(0)
// s_waitcnt vmcnt(N) alone (gfx9 encoding: vmcnt = bits 3:0 and 15:14, expcnt 6:4 and lgkmcnt 11:8 left at their maxima)
#define W4_VMCNT(N) __builtin_amdgcn_s_waitcnt(0x0F70 | ((N) & 0xF) | ((((N) >> 4) & 3) << 14))

// ---- where the LDS writes sit among the MFMA slots of a K sub-step ---------------------------------------------------
// spread (W4_BURST 0): every sub-step writes half of ITS tile parity's registers (NL / 2 writes): tile u+2 is written over
//   the whole of tile u; the pair (P_j, Q_j) is re-requested behind Q_j's write, so P has 2 sub-steps (~2048 cycles) from
//   request to use, Q has 4.
// burst (W4_BURST 1): both registers of a pair are written in the one window in which both of their ring slots are free --
//   second sub-step of the even tile, first of the odd one -- and re-requested at once: 4 sub-steps (~4096 cycles) for every
//   load, at the price of NL writes + NL loads in each of those two sub-steps and none in the other two.
#ifndef W4_BURST
#define W4_BURST 0
#endif
constexpr int write_slot(bool cplx, bool burst, int k) {
  if (!burst) return cplx ? 6 + 4 * k : (k == 0 ? 6 : k == 1 ? 9 : k == 2 ? 11 : 14);
  if (cplx) { constexpr int t[12] = {6, 7, 10, 11, 12, 13, 14, 15, 18, 19, 20, 21}; return t[k]; }
  constexpr int t[8] = {5, 6, 7, 9, 10, 11, 13, 14};
  return t[k];
}
constexpr int slot_of_write(bool cplx, bool burst, int nw, int m, int off) {
  for (int k = 0; k < nw; ++k)
    if (write_slot(cplx, burst, k) + off == m) return k;
  return -1;
}
// LDS operations a sub-step issues behind its last fragment read (the counted wait in front of the barrier)
constexpr int ops_after_last_read(bool cplx, bool burst, bool writes) { return !writes ? 0 : !burst ? 1 : cplx ? 0 : 2; }

#ifndef W4_EPI_PIPE
#define W4_EPI_PIPE 1   // epilogue operands of round r + 1 requested ahead of the stores of round r (0: behind them, A/B)
#endif
#ifndef W4P_SPLIT
#define W4P_SPLIT 0   // PERSIST: 1 = a select-free copy of the six-tile body for the steady part of the K range (see the K loop)
#endif
#ifndef W4P_RELOAD
#define W4P_RELOAD 1   // PERSIST: re-request the next tile's K tiles 2, 3 behind the epilogue (see the tile loop)
#endif
// PERSIST (round 5 experiment, family bit 7; bf16 output, plain epilogue, no bias, (N,N) / (N,T)): ONE workgroup per CU walks
// the tiles lin0, lin0 + grid, ... and the K-tile ring runs THROUGH the output-tile boundaries -- the loads the one-tile
// form issues past the end of its K range (surplus, unused) fetch the NEXT tile's K tiles 0 .. 3 instead, so that tile's K
// tiles 0, 1 are in the ring and 2, 3 in the registers when the current tile's last MFMA retires; nothing of a prologue is
// left but the accumulator reset.  The ring position advances by nt % 3 per output tile: instead of rotating slot offsets
// in scalar registers (round 4's attempt: adds in the K loop, or spilled staging registers) the per-slot LDS BASE
// REGISTERS swap roles at the boundary (18 v_mov per output tile, none in the K loop), so the K loop is the one-tile
// kernel's instruction for instruction.  The epilogue stages through the ring slot that died with the last K tile.
template <typename TOUT, bool CPLX, bool CONJ, bool TA, bool TB, bool PERSIST = false>
__device__ __forceinline__ void w4_tile(const GemmArgs& g, const int lin0, char* smem) {
  using C = Cfg<CPLX>;
  constexpr int NT = C::NT, BM = C::BM, BN = C::BN, IB = C::IB, JB = C::JB, NPL = C::NPL;
  constexpr int PA = C::PA, PB = C::PB, NL = C::NL, NFRAG = C::NFRAG, NM = C::NM, SLOT = C::SLOT;

  // ---- tile coordinates: XCD-contiguous grouped order (as gemm_bf16_kernel) ---------------------------------
  const int tiles_m = g.M / BM, tiles_n = g.N / BN;
  const int ntiles = tiles_m * tiles_n;
  int split = 0;
  auto origin = [&](int lin, int& om, int& on) __attribute__((always_inline)) {
    if (g.splits > 1) { split = lin / ntiles; lin -= split * ntiles; }     // split-K: block (split, tile), float32 slabs
    const int q = ntiles >> 3, r = ntiles & 7, xcd = lin & 7, idx = lin >> 3;
    lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;  // bijective
    const int GM = g.group_m;
    const int per_group = GM * tiles_n;
    const int grp = lin / per_group, in_grp = lin - grp * per_group;
    const int first_m = grp * GM;
    const int gm = (tiles_m - first_m) < GM ? (tiles_m - first_m) : GM;
    om = __builtin_amdgcn_readfirstlane((first_m + in_grp % gm) * BM);
    on = __builtin_amdgcn_readfirstlane((in_grp / gm) * BN);
  };
  int m0, n0;
  origin(lin0, m0, n0);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int wm = (wid / C::WN) * (32 * IB), wn = (wid % C::WN) * (32 * JB);
  const int l31 = lane & 31, lk = lane >> 5;
  const int l15 = lane & 15, lg = (lane >> 4) & 1;

  const int64_t lda = TA ? g.a_cs : g.a_rs, ldb = TB ? g.b_cs : g.b_rs;
  // plane base pointers of this tile (wave-uniform); a K tile pair advances them by `kstep` bytes
  const int64_t ka = TA ? lda * 2 : 2, kb = TB ? ldb * 2 : 2;       // bytes per k step
  const char* pa[NPL]; const char* pb[NPL];
  auto plane_bases = [&](int tm, int tn, const char* (&qa)[NPL], const char* (&qb)[NPL]) __attribute__((always_inline)) {
    qa[0] = (const char*)g.a_r + (TA ? (int64_t)tm : (int64_t)tm * lda) * 2;
    qb[0] = (const char*)g.b_r + (TB ? (int64_t)tn : (int64_t)tn * ldb) * 2;
    if (CPLX) {
      qa[NPL - 1] = (const char*)g.a_i + (TA ? (int64_t)tm : (int64_t)tm * lda) * 2;
      qb[NPL - 1] = (const char*)g.b_i + (TB ? (int64_t)tn : (int64_t)tn * ldb) * 2;
    }
  };
  plane_bases(m0, n0, pa, pb);
  // PERSIST: the tile after this one (its K tiles 0 .. 3 are requested during this tile's last K tiles), and the source
  // of the register pairs being requested right now (set once per request group: set_request)
  const char* pan[NPL]; const char* pbn[NPL]; const char* rqa[NPL]; const char* rqb[NPL];
  const int ntp = __builtin_amdgcn_readfirstlane(g.K / BK);
  auto set_request = [&](int kt, auto REDIR) __attribute__((always_inline)) {
    // REDIR false: kt is inside this tile's K range (the steady part of the K loop: no select in it)
    const bool nx = decltype(REDIR)::value && kt >= ntp;
    const int kk = (nx ? kt - ntp : kt) & ~1;
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) {
      rqa[pl] = (nx ? pan[pl] : pa[pl]) + (int64_t)kk * BK * ka;
      rqb[pl] = (nx ? pbn[pl] : pb[pl]) + (int64_t)kk * BK * kb;
    }
  };
  const int kbase = __builtin_amdgcn_readfirstlane(split * g.kchunk);
  if (g.splits > 1) {
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) { pa[pl] += (int64_t)kbase * ka; pb[pl] += (int64_t)kbase * kb; }
  }
  const uint32_t voa = src_voff<BM, TA>(lda, tid), vob = src_voff<BN, TB>(ldb, tid);
  const uint32_t psa = (uint32_t)__builtin_amdgcn_readfirstlane((int)piece_stride<BM, TA>(lda));
  const uint32_t psb = (uint32_t)__builtin_amdgcn_readfirstlane((int)piece_stride<BN, TB>(ldb));
  // second K tile of a pair: + 32 k.  "N": 64 bytes further in the same line (immediate); "T": 32 rows further (scalar)
  const uint32_t qa = TA ? (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(32 * ka)) : 0u;
  const uint32_t qb = TB ? (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(32 * kb)) : 0u;
  const uint32_t wra = dst_off<BM, TA>(tid), wrb = dst_off<BN, TB>(tid);

  f32x16 acc_r[IB][JB], acc_i[CPLX ? IB : 1][JB];

  // bias rides in the accumulators (see gemm_bf16_kernel)
  const bool has_bias = g.bias_r != nullptr;
  f4 bias_v[NPL][JB][4];
  if (has_bias) {
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) {
      const float* bias = pl ? g.bias_i : g.bias_r;
#pragma unroll
      for (int j = 0; j < JB; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) bias_v[pl][j][q] = ld4(bias + n0 + wn + j * 32 + 8 * q + 4 * lk);
    }
#ifdef CPLXAMD_GEMM_F16    // (the half-operand build only: the bf16 kernels' code is untouched)
    if (g.scale_a) {          // scaled split products: the accumulators are multiplied by 1 / (sa sb) behind the K loop
      const float inv = gemm_alpha_inv(g);
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
        for (int j = 0; j < JB; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) bias_v[pl][j][q].v[e] *= inv;
    }
#endif
  }

  // ---- staging registers: [parity of the K tile][piece] ------------------------------------------------------
  u32x4 st[2][NL];
  // piece q: [0, PA) A re, [PA, PA+PB) B re, then A im, B im
  auto load_piece = [&](auto PAR, auto Q, int kt) __attribute__((always_inline)) {
    constexpr int par = decltype(PAR)::value, q = decltype(Q)::value;
    constexpr int pl = q / (PA + PB), r = q % (PA + PB);
    constexpr bool isa = r < PA;
    constexpr int j = isa ? r : r - PA;
    // pair base = K tile (kt & ~1); the odd tile of the pair is +32 k
    const char* base;
    if constexpr (PERSIST) {
      base = isa ? rqa[pl] : rqb[pl];          // (set_request(kt) ran for this request group)
    } else {
      const int64_t kby = (int64_t)(kt & ~1) * BK * (isa ? ka : kb);
      base = (isa ? pa[pl] : pb[pl]) + kby;
    }
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7ffffffe, 0x00020000);
    const uint32_t so = (uint32_t)j * (isa ? psa : psb) + (par ? (isa ? qa : qb) : 0u);
    const uint32_t vo = (isa ? voa : vob) + ((par && !(isa ? TA : TB)) ? 64u : 0u);
    st[par][q] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, so, 0);
  };
  // LDS addresses: per ring slot one register per operand for the writes ...
  const uint32_t smem_off = (uint32_t)(uintptr_t)smem;
  uint32_t wa[3], wb[3];
  uint32_t sl[3];                            // PERSIST: byte offset of the ring slot in each ROLE (scalar)
#pragma unroll
  for (int s3 = 0; s3 < 3; ++s3) {
    wa[s3] = opaque(smem_off + s3 * SLOT + wra);
    wb[s3] = opaque(smem_off + s3 * SLOT + C::A_BYTES + wrb);
    sl[s3] = (uint32_t)(s3 * SLOT);
  }
  auto write_piece = [&](auto PAR, auto Q, auto WS) __attribute__((always_inline)) {
    constexpr int par = decltype(PAR)::value, q = decltype(Q)::value, ws = decltype(WS)::value;
    constexpr int pl = q / (PA + PB), r = q % (PA + PB);
    constexpr bool isa = r < PA;
    constexpr int j = isa ? r : r - PA;
    const uint32_t addr = (isa ? wa[ws] : wb[ws]) + (uint32_t)(pl * (C::A_BYTES + C::B_BYTES) + j * 4096);
    *(lds_st_p)(uintptr_t)(addr) = st[par][q];
  };
  // ... and for the fragment reads one per K sub-step ("N") or per block ("T")
  constexpr int NFA = TA ? IB : 2, NFB = TB ? JB : 2;
  uint32_t fa[3][NFA], fb[3][NFB];
#pragma unroll
  for (int s3 = 0; s3 < 3; ++s3) {
#pragma unroll
    for (int x = 0; x < NFA; ++x)
      fa[s3][x] = opaque(smem_off + s3 * SLOT + (TA ? lane_t<BM>(wm, lane, x) : lane_n(wm, lane, x)));
#pragma unroll
    for (int x = 0; x < NFB; ++x)
      fb[s3][x] = opaque(smem_off + s3 * SLOT + C::A_BYTES + (TB ? lane_t<BN>(wn, lane, x) : lane_n(wn, lane, x)));
  }
  // fragment (plane pl, block, K sub-step ks) of ring slot S
  auto a_frag = [&](auto S, auto PL, auto BLK, auto KS) __attribute__((always_inline)) -> bf16x8 {
    constexpr int s3 = decltype(S)::value, pl = decltype(PL)::value, blk = decltype(BLK)::value, ks = decltype(KS)::value;
    constexpr int po = pl * (C::A_BYTES + C::B_BYTES);
    if constexpr (TA) return lds_frag_t<BM>(fa[s3][blk], po + ks * 16 * (BM * 2));
    else return lds_frag_n(fa[s3][ks], po + blk * 2048);
  };
  auto b_frag = [&](auto S, auto PL, auto BLK, auto KS) __attribute__((always_inline)) -> bf16x8 {
    constexpr int s3 = decltype(S)::value, pl = decltype(PL)::value, blk = decltype(BLK)::value, ks = decltype(KS)::value;
    constexpr int po = pl * (C::A_BYTES + C::B_BYTES);
    if constexpr (TB) return lds_frag_t<BN>(fb[s3][blk], po + ks * 16 * (BN * 2));
    else return lds_frag_n(fb[s3][ks], po + blk * 2048);
  };

  // Fragment registers.  MFMA order inside a K sub-step is block row i outermost, so the A fragments of row i die after
  // its 2 * NPL * JB MFMAs and the next sub-step's fragments of that row are read into the SAME registers (rows 0..2);
  // row 3 (last used, first to be needed again too late) and all B fragments alternate between two sets.
  bf16x8 ar[IB + 1], ai[CPLX ? IB + 1 : 1], br[2][JB], bi[2][CPLX ? JB : 1], nbi[CPLX ? JB : 1];
  using PL0 = std::integral_constant<int, 0>; using PL1 = std::integral_constant<int, NPL - 1>;
  auto rd_a = [&](auto F, auto I, auto S, auto KS) __attribute__((always_inline)) {   // row I of (slot S, sub-step KS) -> its register for set F
    constexpr int f = decltype(F)::value, i = decltype(I)::value;
    constexpr int d = i < IB - 1 ? i : IB - 1 + f;
    ar[d] = a_frag(S, PL0{}, I, KS);
  };
  auto rd_ai = [&](auto F, auto I, auto S, auto KS) __attribute__((always_inline)) {
    constexpr int f = decltype(F)::value, i = decltype(I)::value;
    constexpr int d = i < IB - 1 ? i : IB - 1 + f;
    if constexpr (CPLX) ai[d] = a_frag(S, PL1{}, I, KS);
  };
  auto rd_b = [&](auto F, auto J, auto S, auto KS) __attribute__((always_inline)) {
    constexpr int f = decltype(F)::value;
    br[f][decltype(J)::value] = b_frag(S, PL0{}, J, KS);
  };
  auto rd_bi = [&](auto F, auto J, auto S, auto KS) __attribute__((always_inline)) {
    constexpr int f = decltype(F)::value;
    if constexpr (CPLX) bi[f][decltype(J)::value] = b_frag(S, PL1{}, J, KS);
  };

  // MFMA m of a K sub-step on fragment set f.  Per accumulator the order is (br, then bi) per K sub-step as in
  // gemm_bf16_kernel: bit-identical sums.  complex: m = i * 8 + ph * 4 + j * 2 + c;  real: m = i * JB + j
  auto mfma = [&](auto F, auto MM) __attribute__((always_inline)) {
    constexpr int f = decltype(F)::value, m = decltype(MM)::value;
    if constexpr (CPLX) {
      constexpr int i = m / 8, ph = (m % 8) / 4, j = (m % 4) / 2, c = m % 2;
      constexpr int d = i < IB - 1 ? i : IB - 1 + f;
      if constexpr (ph == 0) {
        if constexpr (c == 0) acc_r[i][j] = CPLXAMD_MFMA16(br[f][j], ar[d], acc_r[i][j]);
        else acc_i[i][j] = CPLXAMD_MFMA16(br[f][j], ai[d], acc_i[i][j]);
      } else if constexpr (CONJ) {
        if constexpr (c == 0) acc_r[i][j] = CPLXAMD_MFMA16(bi[f][j], ai[d], acc_r[i][j]);
        else acc_i[i][j] = CPLXAMD_MFMA16(nbi[j], ar[d], acc_i[i][j]);
      } else {
        if constexpr (c == 0) acc_r[i][j] = CPLXAMD_MFMA16(nbi[j], ai[d], acc_r[i][j]);
        else acc_i[i][j] = CPLXAMD_MFMA16(bi[f][j], ar[d], acc_i[i][j]);
      }
    } else {
      constexpr int i = m / JB, j = m % JB;
      constexpr int d = i < IB - 1 ? i : IB - 1 + f;
      acc_r[i][j] = CPLXAMD_MFMA16(br[f][j], ar[d], acc_r[i][j]);
    }
  };

  const int klen = g.splits > 1 ? ((g.K - kbase) < g.kchunk ? (g.K - kbase) : g.kchunk) : g.K;
  const int nt = __builtin_amdgcn_readfirstlane(klen / BK);

  // One K sub-step: NM MFMA slots; behind slot m its fillers.  F = fragment set the MFMAs use (the reads fill the other
  // one / the dead rows), PAR = parity of the K tile being computed (= parity of the registers being written), H = which
  // half of the tile.  Slot table (G = MFMAs per block row = NM / IB):
  //   0 .. 2 JB-1        next B fragments (other set);  0, 1 also: the sign XORs of THIS sub-step's Bi
  //   2 JB, 2 JB + 1     next A row 3 (other set)          [real: JB]
  //   (i+1) G, (i+1) G+1 next A row i = 0, 1, 2 (in place, behind the row's last MFMA)
  //   the NL / 2 LDS writes (odd tiles: each followed by the register pair's two loads) on the free slots in between
  auto sub = [&](auto F, auto PAR, auto H, auto RS, auto RKS, auto WS, auto WS_P, auto WS_Q, int kt_next) __attribute__((always_inline)) {
    constexpr int f = decltype(F)::value, par = decltype(PAR)::value, h = decltype(H)::value;
    using I_F = std::integral_constant<int, f>;
    using I_G = std::integral_constant<int, 1 - f>;
    using I_P = std::integral_constant<int, par>;
    using I0_ = std::integral_constant<int, 0>; using I1_ = std::integral_constant<int, 1>;
    constexpr int G = NM / IB;                                // 8 / 4
    constexpr int NW = NL / 2;                                // writes per sub-step (6 / 4)
    auto slot_fill = [&](auto MM) __attribute__((always_inline)) {
      constexpr int m = decltype(MM)::value;
      // --- fragment reads
      if constexpr (m < NPL * JB) {
        if constexpr (CPLX) {
          if constexpr ((m & 1) == 0) rd_b(I_G{}, std::integral_constant<int, m / 2>{}, RS, RKS);
          else rd_bi(I_G{}, std::integral_constant<int, m / 2>{}, RS, RKS);
        } else {
          rd_b(I_G{}, std::integral_constant<int, m>{}, RS, RKS);
        }
      }
      if constexpr (CPLX && m < JB) nbi[m] = neg_frag(bi[f][m]);
      if constexpr (m == NPL * JB) rd_a(I_G{}, std::integral_constant<int, IB - 1>{}, RS, RKS);
      if constexpr (CPLX && m == NPL * JB + 1) rd_ai(I_G{}, std::integral_constant<int, IB - 1>{}, RS, RKS);
      if constexpr (m >= G && m % G == 0 && m / G <= IB - 1) rd_a(I_G{}, std::integral_constant<int, m / G - 1>{}, RS, RKS);
      if constexpr (CPLX && m >= G && m % G == 1 && m / G <= IB - 1) rd_ai(I_G{}, std::integral_constant<int, m / G - 1>{}, RS, RKS);
      // --- LDS writes + the loads that refill the registers (see write_slot above)
      if constexpr (!W4_BURST) {
        constexpr int wi = slot_of_write(CPLX, false, NW, m, 0), lp = slot_of_write(CPLX, false, NW, m, W4_LDP),
                      lq = slot_of_write(CPLX, false, NW, m, W4_LDQ);
        if constexpr (wi >= 0) write_piece(I_P{}, std::integral_constant<int, h * NW + (wi >= 0 ? wi : 0)>{}, WS);
        if constexpr (par == 1) {
          if constexpr (lp >= 0) { W4_SB(); load_piece(I0_{}, std::integral_constant<int, h * NW + (lp >= 0 ? lp : 0)>{}, kt_next); }
          if constexpr (lq >= 0) { W4_SB(); load_piece(I1_{}, std::integral_constant<int, h * NW + (lq >= 0 ? lq : 0)>{}, kt_next); }
        }
      } else if constexpr (par != h) {
        // window sub-step: (even tile, second half) writes pairs 0 .. NW-1, (odd tile, first half) pairs NW .. NL-1;
        // k-th write: pair k / 2, register k % 2 (P -> slot of tile u+2, Q -> slot of tile u+3 = the even tile's own slot)
        constexpr int k = slot_of_write(CPLX, true, NL, m, 0);
        if constexpr (k >= 0) {
          constexpr int q = (par == 0 ? 0 : NW) + (k >= 0 ? k : 0) / 2, reg = (k >= 0 ? k : 0) % 2;
          if constexpr (reg == 0) write_piece(I0_{}, std::integral_constant<int, q>{}, WS_P);
          else write_piece(I1_{}, std::integral_constant<int, q>{}, WS_Q);
          if constexpr (reg == 1) {
            W4_SB();
            load_piece(I0_{}, std::integral_constant<int, q>{}, kt_next);
            load_piece(I1_{}, std::integral_constant<int, q>{}, kt_next);
          }
        }
      }
    };
    // every fragment read of the previous sub-step is older than its last LDS write: one counted wait instead of one in
    // front of each fragment's first use (the builtin is visible to the compiler's own wait insertion)
    if constexpr (W4_WAIT1 != 0) __builtin_amdgcn_s_waitcnt(0xC07F | (ops_after_last_read(CPLX, W4_BURST, !W4_BURST || par == h) << 8));
    auto run = [&](auto self, auto MM) __attribute__((always_inline)) {
      constexpr int m = decltype(MM)::value;
      if constexpr (m < NM) {
        mfma(I_F{}, MM);
        W4_SB();
        slot_fill(MM);
        W4_SB();
        self(self, std::integral_constant<int, m + 1>{});
      }
    };
    run(run, std::integral_constant<int, 0>{});
  };

  // ---- prologue: K tiles 0, 1 through the registers into slots 0, 1; pair (2, 3) requested ------------------------
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
  auto for_pieces = [&](auto fn) __attribute__((always_inline)) {
    auto go = [&](auto self, auto Q) __attribute__((always_inline)) {
      constexpr int q = decltype(Q)::value;
      if constexpr (q < NL) { fn(Q); self(self, std::integral_constant<int, q + 1>{}); }
    };
    go(go, I0{});
  };
  // PERSIST: the tile walk of this workgroup (lin0, lin0 + grid, ...) and the tile behind the first one
  int lin = lin0, mn = m0, nn = n0;
  bool has_next = false;
  auto look_ahead = [&]() __attribute__((always_inline)) {
    has_next = lin + (int)gridDim.x < ntiles;
    mn = m0; nn = n0;
    if (has_next) origin(lin + (int)gridDim.x, mn, nn);
    plane_bases(mn, nn, pan, pbn);          // (no next tile: this one again -- the look-ahead loads land nowhere that is read)
  };
  if constexpr (PERSIST) { look_ahead(); set_request(0, std::false_type{}); }
  for_pieces([&](auto Q) __attribute__((always_inline)) { load_piece(I0{}, Q, 0); load_piece(I1{}, Q, 0); });
  // acc = bias[n] (or 0)
#pragma unroll
  for (int i = 0; i < IB; ++i)
#pragma unroll
    for (int j = 0; j < JB; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc_r[i][j][4 * q + e] = has_bias ? bias_v[0][j][q].v[e] : 0.f;
          if (CPLX) acc_i[i][j][4 * q + e] = has_bias ? bias_v[NPL - 1][j][q].v[e] : 0.f;
        }
  for_pieces([&](auto Q) __attribute__((always_inline)) {
    write_piece(I0{}, Q, I0{});
    write_piece(I1{}, Q, I1{});
  });
  {
    const int kt = nt > 2 ? 2 : nt - 2;
    if constexpr (PERSIST) set_request(kt, std::true_type{});
    for_pieces([&](auto Q) __attribute__((always_inline)) { load_piece(I0{}, Q, kt); load_piece(I1{}, Q, kt); });
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  auto first_frags = [&]() __attribute__((always_inline)) {
    // fragments of (tile 0, sub 0) -> set 0
    auto goa = [&](auto self, auto I) __attribute__((always_inline)) {
      constexpr int i = decltype(I)::value;
      if constexpr (i < IB) { rd_a(I0{}, I, I0{}, I0{}); rd_ai(I0{}, I, I0{}, I0{}); self(self, std::integral_constant<int, i + 1>{}); }
    };
    auto gob = [&](auto self, auto J) __attribute__((always_inline)) {
      constexpr int j = decltype(J)::value;
      if constexpr (j < JB) { rd_b(I0{}, J, I0{}, I0{}); rd_bi(I0{}, J, I0{}, I0{}); self(self, std::integral_constant<int, j + 1>{}); }
    };
    gob(gob, I0{});
    goa(goa, I0{});
  };
  first_frags();

  // ---- K loop ---------------------------------------------------------------------------------------------------
  // K tile t in ring slot S (compile time), parity P (compile time)
  auto tile = [&](auto SS, auto PP, int t, auto REDIR) __attribute__((always_inline)) {
    constexpr int s = decltype(SS)::value, par = decltype(PP)::value;
    using CUR = std::integral_constant<int, s>; using NXT = std::integral_constant<int, (s + 1) % 3>;
    using WR = std::integral_constant<int, (s + 2) % 3>;
    // the register pairs are re-requested with the K tiles two pairs ahead; past the end: the last pair again (unused)
    constexpr int ahead = (W4_BURST && par == 0) ? 4 : 3;
    int ktn = (t + ahead < nt) ? t + ahead : nt - 2;
    if constexpr (PERSIST) {
      ktn = t + ahead;                       // past the end of this tile's K range: the next tile's K tiles 0 .. 3
      if constexpr (par == 1) set_request(ktn, REDIR);
    }
    // burst: P (tile u+2) -> even tile's WR slot = odd tile's NXT slot; Q (tile u+3) -> even tile's CUR = odd tile's WR
    using WSP = std::conditional_t<par == 0, WR, NXT>;
    using WSQ = std::conditional_t<par == 0, CUR, WR>;
    sub(I0{}, PP, I0{}, CUR{}, I1{}, WR{}, WSP{}, WSQ{}, ktn);
    // every wave: its F1 reads are complete (in-order LDS: all but the operations issued behind the last read)
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(ops_after_last_read(CPLX, W4_BURST, !W4_BURST || par == 1)) : "memory");
    __builtin_amdgcn_s_barrier();
    W4_SB();
    sub(I1{}, PP, I1{}, NXT{}, I0{}, WR{}, WSP{}, WSQ{}, ktn);
  };
  // role swap of the per-slot LDS base registers: new[S] = old[(S + R) % 3]
  auto rotate = [&](auto R) __attribute__((always_inline)) {
    constexpr int r = decltype(R)::value;
    auto rot3 = [&](uint32_t& x0, uint32_t& x1, uint32_t& x2) __attribute__((always_inline)) {
      const uint32_t a = x0, b = x1, c = x2;
      if constexpr (r == 1) { x0 = b; x1 = c; x2 = a; } else { x0 = c; x1 = a; x2 = b; }
    };
    rot3(wa[0], wa[1], wa[2]); rot3(wb[0], wb[1], wb[2]); rot3(sl[0], sl[1], sl[2]);
#pragma unroll
    for (int x = 0; x < NFA; ++x) rot3(fa[0][x], fa[1][x], fa[2][x]);
#pragma unroll
    for (int x = 0; x < NFB; ++x) rot3(fb[0][x], fb[1][x], fb[2][x]);
  };
  do {       // (one pass unless PERSIST)
  {
    int t = 0;
    using RD = std::true_type;            // (only PERSIST looks at it)
    if constexpr (PERSIST && W4P_SPLIT) {
      // steady part: every request of the six tiles (K tiles t + 3 .. t + 8) lies inside this tile's K range.  (Measured
      // as a compile: with two copies of the six-tile body the register allocator reconciles their accumulator
      // assignments with 192 v_accvgpr moves INSIDE the first one -- off by default.)
      for (; t + 9 <= nt; t += 6) {
        using NR = std::false_type;
        tile(I0{}, I0{}, t, NR{});
        tile(I1{}, I1{}, t + 1, NR{});
        tile(I2{}, I0{}, t + 2, NR{});
        tile(I0{}, I1{}, t + 3, NR{});
        tile(I1{}, I0{}, t + 4, NR{});
        tile(I2{}, I1{}, t + 5, NR{});
      }
    }
    for (; t + 6 <= nt; t += 6) {
      tile(I0{}, I0{}, t, RD{});
      tile(I1{}, I1{}, t + 1, RD{});
      tile(I2{}, I0{}, t + 2, RD{});
      tile(I0{}, I1{}, t + 3, RD{});
      tile(I1{}, I0{}, t + 4, RD{});
      tile(I2{}, I1{}, t + 5, RD{});
    }
    if (t < nt) {
      tile(I0{}, I0{}, t, RD{});
      tile(I1{}, I1{}, t + 1, RD{});
      t += 2;
      if (t < nt) {
        tile(I2{}, I0{}, t, RD{});
        tile(I0{}, I1{}, t + 1, RD{});
        if constexpr (PERSIST) rotate(I1{});     // last K tile in slot 0: the next tile's K tile 0 sits in slot 1
      } else {
        if constexpr (PERSIST) rotate(I2{});     // last K tile in slot 1: ... in slot 2
      }
    }
  }

#ifdef CPLXAMD_GEMM_F16
  if constexpr (!PERSIST) {
    if (g.scale_a) {            // scaled split products (gemm.h): exact, the scales are powers of two
      const float alpha = gemm_alpha(g);
#pragma unroll
      for (int i = 0; i < IB; ++i)
#pragma unroll
        for (int j = 0; j < JB; ++j) {
          acc_r[i][j] *= alpha;
          if (CPLX) acc_i[i][j] *= alpha;
        }
    }
  }
#endif

  // ---- epilogue (the one-tile kernel's, for IB = 4 and JB / 2 column halves of 64) ------------------------------
  // output planes: C, or -- split-K -- this split's float32 slab pair [split][plane][M][ldc] of the workspace (dense, no
  // bias / multiplier / accumulate: gemm_slab_reduce_kernel applies those)
  void* out_r = g.c_r;
  void* out_i = g.c_i;
  if (g.splits > 1) {
    const int64_t slab = (int64_t)g.M * g.ldc;
    float* base = reinterpret_cast<float*>(g.ws) + (int64_t)split * NPL * slab;
    out_r = base; out_i = base + slab;
  }
  const float beta = gemm_beta(g);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the surplus loads of the loop's tail target registers nothing reads: the compiler waits for them only where it reuses one)
  // every wave is done with the ring.  PERSIST: the staging slot below is the one whose K tile every wave finished reading
  // before the last mid-tile barrier (role 2 after the swap); the waves still in the last K sub-step touch the other two
  if constexpr (!PERSIST) __builtin_amdgcn_s_barrier();
  // The epilogue's optional operands (fused LRT term; multiplier / accumulate operand) are tested ONCE, outside: a per-load
  // "if (g.fga)" inside the unrolled passes makes the compiler branch around every load and wait for each one separately
  // (cdna_hip_programming.md, "three .s-level traps" (c)) -- and at one wave per SIMD nothing else hides that latency.
  // Global addresses: ONE per-lane 32-bit byte offset per operand kind (lane part of a staged round) + a wave-uniform
  // offset per (round, pass) in a scalar register, through a buffer descriptor of the output tile -- sixteen 64-bit
  // per-lane addresses per round cost more registers than the epilogue has beside the accumulators.
  auto tile_rsrc = [&](const void* plane, int64_t ld, int esize) __attribute__((always_inline)) {
    const char* base = (const char*)plane + ((int64_t)m0 * ld + n0) * esize;
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7ffffffe, 0x00020000);
  };
  auto ldb128 = [](__amdgpu_buffer_rsrc_t r, uint32_t vo, uint32_t so) __attribute__((always_inline)) -> u32x4 {
    return __builtin_amdgcn_raw_buffer_load_b128(r, vo, so, 0);
  };
  auto stb128 = [](u32x4 v, __amdgpu_buffer_rsrc_t r, uint32_t vo, uint32_t so) __attribute__((always_inline)) {
    __builtin_amdgcn_raw_buffer_store_b128(v, r, vo, so, 0);
  };
  if constexpr (sizeof(TOUT) == 2) {
    // bf16: the wave's tile goes through LDS so that a store instruction writes whole 128-byte lines
    auto epi16 = [&](auto FUSE) __attribute__((always_inline)) {
      constexpr bool fuse = decltype(FUSE)::value;
      constexpr int PITCH = 144;                        // bytes per staged row (64 bf16 + 16 B pad)
      // PERSIST: the dead ring slot (complex: 48 KiB >= 4 x 9 KiB); the real kernel's slots are 32 KiB -- its staging rows
      // live behind the ring (96 + 36 KiB of LDS: launch_w4p)
      // lane part: row (lane >> 3) of an 8-row pass, 16-byte column group (lane & 7); wave origin (wm, wn)
      // PERSIST: everything per-lane below is derived from an OPAQUE copy of the thread index, made here: otherwise the
      // loop-invariant epilogue addresses are hoisted out of the tile loop, live across the K loop -- which has not one
      // register to spare -- and a staging register is spilled (with a vmcnt(0)) inside it
      const int tid_e = PERSIST ? (int)opaque((uint32_t)tid) : tid;
      const int lane = tid_e & 63, wid = tid_e >> 6;
      const int wm = (wid / C::WN) * (32 * IB), wn = (wid % C::WN) * (32 * JB);
      const int l31 = lane & 31, lk = lane >> 5;
      const uint32_t vo_c = (uint32_t)((((int64_t)(wm + (lane >> 3)) * g.ldc + wn + (lane & 7) * 8)) * 2);
      char* reg = smem + (PERSIST ? (CPLX ? sl[2] : (uint32_t)C::SMEM) : 0u) + wid * (64 * PITCH);
      const uint32_t vo_f = fuse ? (uint32_t)((((int64_t)(wm + (lane >> 3)) * g.fld + wn + (lane & 7) * 8)) * 2) : 0u;
      const uint32_t p8c = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(8 * g.ldc * 2));
      const uint32_t p8f = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(8 * g.fld * 2));
      const __amdgpu_buffer_rsrc_t rga = tile_rsrc(fuse ? g.fga : g.c_r, fuse ? g.fld : g.ldc, 2);
      const __amdgpu_buffer_rsrc_t rout0 = tile_rsrc(out_r, g.ldc, 2), rout1 = tile_rsrc(CPLX ? out_i : out_r, g.ldc, 2);
      const __amdgpu_buffer_rsrc_t rx0 = tile_rsrc(fuse ? g.fx_r : g.c_r, fuse ? g.fld : g.ldc, 2);
      const __amdgpu_buffer_rsrc_t rx1 = tile_rsrc(fuse ? (CPLX ? g.fx_i : g.fx_r) : g.c_r, fuse ? g.fld : g.ldc, 2);
      // Rounds r = (plane, column half jh, row half ih).  The fused term's operands of round r + 1 (8 passes x (ga, x): 16
      // loads per lane) are requested BEFORE the stores of round r -- vmcnt retires in issue order: requested behind
      // them they could not be used until every one of those stores was acknowledged (see the float32 epilogue below).
      constexpr int NR = NPL * (JB / 2) * (IB / 2);
      u32x4 gv[2][fuse ? 8 : 1], xv[2][fuse ? 8 : 1];
      auto load_ops = [&](int r) __attribute__((always_inline)) {
        const int pl = r / ((JB / 2) * (IB / 2)), jh = (r / (IB / 2)) % (JB / 2), ih = r % (IB / 2);
#pragma unroll
        for (int pass = 0; pass < 8; ++pass) {
          const uint32_t so = (uint32_t)(ih * 8 + pass) * p8f + (uint32_t)(jh * 128);
          gv[r & 1][pass] = ldb128(rga, vo_f, so);
          xv[r & 1][pass] = ldb128(pl ? rx1 : rx0, vo_f, so);
        }
      };
      if constexpr (fuse) load_ops(0);
      // PERSIST: stores count in vmcnt on this target, and the K loop's counted waits are ONE static instruction each for
      // the first trip after a tile boundary and for every steady trip.  With this epilogue's 64 stores on top of the 24
      // look-ahead loads the compiler's bookkeeping overflows the 6-bit counter and it settles on vmcnt(0) INSIDE the K loop
      // (a full drain of the staging stream every six K tiles: measured +2.5 %).  So: the look-ahead loads are waited for
      // here, once (they were requested one to two K tiles ago), and the stores in flight are kept below 32
      if constexpr (PERSIST) W4_VMCNT(0);
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const int pl = r / ((JB / 2) * (IB / 2)), jh = (r / (IB / 2)) % (JB / 2), ih = r % (IB / 2);
        const __amdgpu_buffer_rsrc_t rout = pl ? rout1 : rout0;
        if constexpr (fuse && !W4_EPI_PIPE) { if (r > 0) load_ops(r); }
        // (PERSIST, complex: the compiler copies all 256 accumulators out of the AGPRs at the K loop's exit and spills ~56
        //  registers of the next tile's ring state around this epilogue.  Pinning the tuples in the AGPRs per round with an
        //  empty asm ("+a" on the 16-register tuple, or an asm v_accvgpr_read per element) makes the allocator shuffle
        //  accumulator tuples INSIDE the K loop instead -- 144 to 192 v_accvgpr moves per six K tiles; both tried, both
        //  worse: profiles/r05_gemm_w4_persistent.txt.  The real kernel has the registers to spare and spills nothing.)
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
          for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int cl = jj * 32 + 8 * q + 4 * lk;
              f4 v;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float a = pl ? acc_i[CPLX ? ih * 2 + ii : 0][jh * 2 + jj][4 * q + e] : acc_r[ih * 2 + ii][jh * 2 + jj][4 * q + e];
                // PERSIST: the staging and fragment registers of the NEXT tile are live across this epilogue -- the scheduling
                // barrier behind every staged group keeps the accumulator reads where they are used (otherwise all 256 are
                // copied out of the AGPRs up front and ~100 registers of the ring state are spilled around every tile
                // boundary; an asm read with an "a" operand instead splits the accumulator tuples: moves in the K loop)
                v.v[e] = a;
              }
              st4(reinterpret_cast<bf16_t*>(reg + (ii * 32 + l31) * PITCH + cl * 2), v);
              if constexpr (PERSIST) W4_SB();
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's tile is in LDS (in-order LDS, own region)
        if constexpr (fuse && W4_EPI_PIPE) {
          if (r + 1 < NR) { W4_SB(); load_ops(r + 1); W4_SB(); }
        }
#pragma unroll
        for (int pass = 0; pass < 8; ++pass) {
          const int rl = pass * 8 + (lane >> 3), c8 = (lane & 7) * 8;
          const u32x4 v = *reinterpret_cast<const u32x4*>(reg + rl * PITCH + c8 * 2);
          const uint32_t so = (uint32_t)(ih * 8 + pass) * p8c + (uint32_t)(jh * 128);
          if constexpr (fuse) {
            // LRT input gradient's elementwise term (gemm.h: fga); arithmetic of util.hip dx_accum_kernel: bit-identical
            u32x4 ow;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float d0 = __uint_as_float(v[e] << 16), d1 = __uint_as_float(v[e] & 0xffff0000u);
              const float x0 = __uint_as_float(xv[r & 1][pass][e] << 16), x1 = __uint_as_float(xv[r & 1][pass][e] & 0xffff0000u);
              const float g0 = __uint_as_float(gv[r & 1][pass][e] << 16), g1 = __uint_as_float(gv[r & 1][pass][e] & 0xffff0000u);
              ow[e] = pack_bf16(fmaf(2.0f * x0, g0, d0), fmaf(2.0f * x1, g1, d1));
            }
            stb128(ow, rout, vo_c, so);
          } else {
            stb128(v, rout, vo_c, so);
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // reads done before the next round overwrites
        if constexpr (PERSIST) W4_VMCNT(16);                 // (at most two rounds of stores in flight: see above)
        W4_SB();
      }
    };
    if constexpr (PERSIST) epi16(std::false_type{});
    else { if (g.fga) epi16(std::true_type{}); else epi16(std::false_type{}); }
  } else {
    // float32: 32 rows x 64 columns per round; the elementwise multiplier (LRT log_sigma2 gradient, mask: both planes of a
    // complex result -- the launcher declines a real-plane-only multiplier) and the accumulate operand are read row-major
    auto epi32 = [&](auto HM, auto HACC) __attribute__((always_inline)) {
      constexpr bool hm = decltype(HM)::value, hacc = decltype(HACC)::value;
      constexpr int PITCH = 272;                        // bytes per staged row (64 floats + 16 B pad)
      char* reg = smem + wid * (32 * PITCH);
      const uint32_t vo = (uint32_t)((((int64_t)(wm + (lane >> 4)) * g.ldc + wn + (lane & 15) * 4)) * 4);
      const uint32_t p4 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(4 * g.ldc * 4));
      const __amdgpu_buffer_rsrc_t rm = tile_rsrc(hm ? (const void*)g.emul : g.c_r, g.ldc, 4);
      const __amdgpu_buffer_rsrc_t rout0 = tile_rsrc(out_r, g.ldc, 4), rout1 = tile_rsrc(CPLX ? out_i : out_r, g.ldc, 4);
      // Rounds r = (plane, column half jh, block row i).  The multiplier / accumulate operands of round r + 1 are
      // requested BEFORE the stores of round r go out: vmcnt retires in issue order, so operands requested behind a
      // round's stores could not be used before every one of those stores was acknowledged -- one store-acknowledgement
      // latency per round, eight per tile, in an epilogue nothing overlaps (W4_EPI_PIPE 0: that order, for the A/B).
      constexpr int NR = NPL * (JB / 2) * IB;
      u32x4 mv[2][hm ? 8 : 1], pv[2][hacc ? 8 : 1];
      auto load_ops = [&](int r) __attribute__((always_inline)) {
        const int pl = r / ((JB / 2) * IB), jh = (r / IB) % (JB / 2), i = r % IB;
#pragma unroll
        for (int pass = 0; pass < 8; ++pass) {
          const uint32_t so = (uint32_t)(i * 8 + pass) * p4 + (uint32_t)(jh * 256);
          if constexpr (hm) mv[r & 1][pass] = ldb128(rm, vo, so);
          if constexpr (hacc) pv[r & 1][pass] = ldb128(pl ? rout1 : rout0, vo, so);
        }
      };
      if constexpr (hm || hacc) load_ops(0);
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const int pl = r / ((JB / 2) * IB), jh = (r / IB) % (JB / 2), i = r % IB;
        const __amdgpu_buffer_rsrc_t rout = pl ? rout1 : rout0;
        if constexpr ((hm || hacc) && !W4_EPI_PIPE) { if (r > 0) load_ops(r); }
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int cl = jj * 32 + 8 * q + 4 * lk;
            f4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e)
              v.v[e] = pl ? acc_i[CPLX ? i : 0][jh * 2 + jj][4 * q + e] : acc_r[i][jh * 2 + jj][4 * q + e];
            st4(reinterpret_cast<float*>(reg + l31 * PITCH + cl * 4), v);
          }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr ((hm || hacc) && W4_EPI_PIPE) {
          if (r + 1 < NR) { W4_SB(); load_ops(r + 1); W4_SB(); }
        }
#pragma unroll
        for (int pass = 0; pass < 8; ++pass) {
          const int rl = pass * 4 + (lane >> 4), c4 = (lane & 15) * 4;
          f4 v = ld4(reinterpret_cast<const float*>(reg + rl * PITCH + c4 * 4));
          const uint32_t so = (uint32_t)(i * 8 + pass) * p4 + (uint32_t)(jh * 256);
          if constexpr (hm) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              v.v[e] *= gemm_emul(g, __uint_as_float(mv[r & 1][pass][e]));
              // (rounded product: the 8-wave kernels apply the multiplier in a block of its own, so nothing there can
              //  contract it with the accumulate below; same bits here)
              asm volatile("" : "+v"(v.v[e]));
            }
          }
          if constexpr (hacc) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v.v[e] += beta * __uint_as_float(pv[r & 1][pass][e]);
          }
          const u32x4 w = {__float_as_uint(v.v[0]), __float_as_uint(v.v[1]), __float_as_uint(v.v[2]), __float_as_uint(v.v[3])};
          stb128(w, rout, vo, so);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        W4_SB();
      }
    };
    if (g.emul) { if (g.accumulate) epi32(std::true_type{}, std::true_type{}); else epi32(std::true_type{}, std::false_type{}); }
    else { if (g.accumulate) epi32(std::false_type{}, std::true_type{}); else epi32(std::false_type{}, std::false_type{}); }
  }
  if constexpr (!PERSIST) {
    break;
  } else {
    if (!has_next) break;
    // the next tile: its K tiles 0, 1 are in the ring (roles 0, 1), 2, 3 in the staging registers, the fragments of its
    // first K sub-step in set 0.  Its first LDS writes go to role 2 -- the slot every wave has just staged its rows in
    __builtin_amdgcn_s_barrier();
    lin += (int)gridDim.x; m0 = mn; n0 = nn;
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) { pa[pl] = pan[pl]; pb[pl] = pbn[pl]; }
    look_ahead();
#pragma unroll
    for (int i = 0; i < IB; ++i)
#pragma unroll
      for (int j = 0; j < JB; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          acc_r[i][j][e] = 0.f;
          if (CPLX) acc_i[i][j][e] = 0.f;
        }
    // the fragments of the new tile's first K sub-step once more (the last K sub-step read them already: re-reading
    // here makes those 48 registers free during the epilogue -- an LDS round trip per tile against spilled ring state)
    first_frags();
    if constexpr (W4P_RELOAD && CPLX) {      // (the real kernel has the registers: nothing is spilled there)
      // the new tile's K tiles 2, 3 ONCE MORE: the look-ahead requested them before the epilogue, but registers that stay
      // live across it are what the allocator spills around it (complex: 96 staging registers beside 256 accumulators on
      // their way out).  Re-requested here, the earlier request is a prefetch into the L2 and the registers are free.
      set_request(2, std::false_type{});
      for_pieces([&](auto Q) __attribute__((always_inline)) { load_piece(I0{}, Q, 2); load_piece(I1{}, Q, 2); });
    }
    // nothing of the boundary (the stores, reloads of whatever the allocator spilled around the epilogue) may be in flight
    // when the K loop's counted waits start counting: they are the same static instructions in every trip
    if constexpr (!(W4P_RELOAD && CPLX)) W4_VMCNT(0);
  }
  } while (true);
}

template <typename TOUT, bool CPLX, bool CONJ, bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_bf16_w4_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  w4_tile<TOUT, CPLX, CONJ, TA, TB>(g, (int)blockIdx.x, smem);
}

// PERSIST form (w4_tile<..., true>): one workgroup per CU, bf16 output, plain epilogue
template <bool CPLX, bool CONJ, bool TB>
__global__ __launch_bounds__(256) void gemm_bf16_w4p_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  w4_tile<bf16_t, CPLX, CONJ, false, TB, true>(g, (int)blockIdx.x, smem);
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

template <typename TOUT, bool CPLX, bool CONJ, bool TA, bool TB>
static int launch(const GemmArgs& g0, hipStream_t st) {
  using C = Cfg<CPLX>;
  GemmArgs g = g0;
  static const int gm = env_int("CPLXAMD_GEMM_GROUP_M", 4);
  g.group_m = gm > 0 ? gm : 1;
  const int64_t tiles = (int64_t)(g.M / C::BM) * (g.N / C::BN);
  if constexpr (sizeof(TOUT) == 2 && !TA && !CPLX) {      // (complex: not instantiated, see below)
    // round 5 (family bit 7): the ring through the tile boundaries; needs the chip (one workgroup per CU), more than one
    // round of tiles, the plain epilogue and no bias, K tile count even (K % 64 == 0 holds).  Taken where measured faster
    // (profiles/r05_gemm_w4_persistent.txt): the REAL launches -- (N,N) at every K depth this family runs at, (N,T) from
    // K = 4096 -- whose boundary code compiles without a spill.  The complex form is bit-identical too but 1.6-2.9 %
    // SLOWER than one tile per workgroup (the allocator spills ring state around its epilogue) and is not launched
    // (the A/B switch: scripts/r06/ablation_switches.patch).
    const int ncu = (g.ncu > 0 ? g.ncu : device_cus()) & ~7;
    const bool measured_faster = !CPLX && (!TB || g.K >= 4096);
    if (((launch_family(g.flags) >> 7) & 1) && measured_faster && launch_owns_chip(g.flags) && tiles > ncu && !g.fga &&
        !g.bias_r && g.splits <= 1) {
      if (g.plan) { *g.plan = 6; return 0; }
      static PerDeviceOnce attr_p;
      constexpr int smem_p = C::SMEM + (CPLX ? 0 : 4 * 64 * 144);      // (real: epilogue staging behind the ring)
      if (const int e = set_max_dyn_lds(attr_p, gemm_bf16_w4p_kernel<CPLX, CONJ, TB>, smem_p)) return e;
      gemm_bf16_w4p_kernel<CPLX, CONJ, TB><<<dim3((unsigned)ncu), C::NT, smem_p, st>>>(g);
      CPLXAMD_CHECK_LAUNCH();
      return 0;
    }
  }
  if (g.plan) { *g.plan = 3; return 0; }
  static PerDeviceOnce attr_set;
  if (const int e = set_max_dyn_lds(attr_set, gemm_bf16_w4_kernel<TOUT, CPLX, CONJ, TA, TB>, C::SMEM)) return e;
  const int64_t grid = tiles * g.splits;
  gemm_bf16_w4_kernel<TOUT, CPLX, CONJ, TA, TB><<<dim3((unsigned)grid), C::NT, C::SMEM, st>>>(g);
  CPLXAMD_CHECK_LAUNCH();
  return 0;
}

template <typename TOUT, bool CPLX, bool CONJ>
static int launch_layout(const GemmArgs& g, bool ta, bool tb, hipStream_t st) {
  if (ta) return tb ? launch<TOUT, CPLX, CONJ, true, true>(g, st) : CPLXAMD_ESHAPE;     // (T,N) is launched by nothing
  return tb ? launch<TOUT, CPLX, CONJ, false, true>(g, st) : launch<TOUT, CPLX, CONJ, false, false>(g, st);
}

}  // namespace w4

// 0 and taken = true: launched.  taken = false: the shape / epilogue is not one this kernel takes.
int launch_gemm_bf16_w4(const GemmArgs& g, bool cplx, int out_dtype, bool ta, bool tb, hipStream_t st, bool& taken) {
  taken = false;
  // which launches the family takes (launch_family(g.flags): CPLXAMD_LAUNCH_FAMILY per call, else the process default): bit 0 complex bf16-out, 1 complex bf16-out with the fused LRT
  // term, 2 complex float32-out, 3 real bf16-out (4: with the fused term), 5 real float32-out; bit 6: regardless of K.
  // Without bit 6 the K depth decides too (profiles/r04_gemm_w4_ab.txt): this family pays a full prologue / epilogue per
  // output tile where the 8-wave (N,N) / (N,T) kernels run persistent, so it wins from K = 4096 on, at 1024 <= K < 4096 only
  // on the (N,N) launches, below that nowhere.
  {
    const bool f32o = out_dtype == CPLXAMD_F32;
    const int bit = cplx ? (f32o ? 2 : g.fga ? 1 : 0) : (f32o ? 5 : g.fga ? 4 : 3);
    const int family = launch_family(g.flags);
    if (!((family >> bit) & 1)) return 0;
    if (!((family >> 6) & 1)) {
      const int keff = g.splits > 1 ? g.kchunk : g.K;      // K loop per output tile
      if (keff < 1024) return 0;
      if (keff < 4096 && (ta || tb || g.fga)) return 0;
    }
  }
  const int bm = 256, bn = cplx ? 128 : 256;
  if (g.g1 || g.batch != 1) return 0;
  if (g.splits > 1) {
    // split-K slabs (float32, dense): every split's K range a multiple of 64 and at least 128 deep
    const int last = g.K - (g.splits - 1) * g.kchunk;
    if (out_dtype != CPLXAMD_F32 || !g.ws || (g.kchunk % 64) || last < 128 || (last % 64) || g.kchunk < 128) return 0;
    if (g.bias_r || g.emul || g.accumulate || g.fga) return 0;
    if (!cplx) return 0;       // (measured: complex slabs -2.4 %, real +2.8 % against the 8-wave kernel: profiles/r04_gemm_w4_ab.txt)
  }
  if (ta && !tb) return 0;
  if ((g.M % bm) || (g.N % bn) || (g.K % 64) || g.K < 128) return 0;
  if ((int64_t)(g.M / bm) * (g.N / bn) * g.splits > 0x7fffffff) return 0;
  const int64_t lda = ta ? g.a_cs : g.a_rs, ldb = tb ? g.b_cs : g.b_rs;
  if ((lda % 8) || (ldb % 8) || lda >= (1 << 22) || ldb >= (1 << 22)) return 0;      // 32-bit per-lane tile offsets
  if (g.ldc >= (1 << 20) || g.fld >= (1 << 20)) return 0;                             // ... of the epilogue (256 rows x 4 bytes)
  if (!w4::aligned16(g.a_r) || !w4::aligned16(g.b_r) || !w4::aligned16(g.c_r)) return 0;
  if (cplx && (!w4::aligned16(g.a_i) || !w4::aligned16(g.b_i) || !w4::aligned16(g.c_i))) return 0;
  if (g.bias_r && (!w4::aligned16(g.bias_r) || (cplx && !w4::aligned16(g.bias_i)))) return 0;
  if (out_dtype == CPLXAMD_BF16) {
    if ((g.ldc & 7) || g.emul || g.accumulate) return 0;
    if (g.fga && ((g.fld & 7) || !w4::aligned16(g.fga) || !w4::aligned16(g.fx_r) || (cplx && !w4::aligned16(g.fx_i)) || g.bias_r))
      return 0;
  } else if (out_dtype == CPLXAMD_F32) {
    if ((g.ldc & 3) || g.fga || (g.emul && !w4::aligned16(g.emul))) return 0;
    if (cplx && g.emul && !g.emul_both) return 0;      // (a multiplier on the real plane only: nothing launches that)
  } else {
    return 0;
  }
  int rc;
  const bool f32 = out_dtype == CPLXAMD_F32;
#ifdef CPLXAMD_GEMM_F16          // the half-operand build: float32 output only
  if (!f32) return 0;
  if (cplx) rc = g.conj_b ? w4::launch_layout<float, true, true>(g, ta, tb, st) : w4::launch_layout<float, true, false>(g, ta, tb, st);
  else rc = w4::launch_layout<float, false, false>(g, ta, tb, st);
#else
  if (cplx) {
    if (g.conj_b) rc = f32 ? w4::launch_layout<float, true, true>(g, ta, tb, st) : w4::launch_layout<bf16_t, true, true>(g, ta, tb, st);
    else rc = f32 ? w4::launch_layout<float, true, false>(g, ta, tb, st) : w4::launch_layout<bf16_t, true, false>(g, ta, tb, st);
  } else {
    rc = f32 ? w4::launch_layout<float, false, false>(g, ta, tb, st) : w4::launch_layout<bf16_t, false, false>(g, ta, tb, st);
  }
#endif
  if (rc == CPLXAMD_ESHAPE) return 0;
  if (rc) return rc;
  taken = true;
  return 0;
}

}  // namespace cplxamd
