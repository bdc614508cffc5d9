// bf16 complex / real GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x16_bf16), "NT" form:
//   C[m,n] = sum_k A[m,k] * op(B[n,k]) (+ bias[n]),  A and B K-contiguous (planar re / im),
//   fp32 accumulation, bf16 or fp32 output.
//
// Complex 4M in ONE K-loop: each staged (Ar, Ai, Br, Bi) tile set feeds four MFMA chains
//   Cr += Ar Br ; Cr += (-Ai) Bi ; Ci += Ar Bi ; Ci += Ai Br
// so every LDS byte is used twice as often as in a real GEMM of the same tile (the reference
// issues 4 separate GEMMs + 2 elementwise passes, cplx.py:641-646).  The sign flip for the
// Ai Bi product (and for conj(B), used by dgrad / wgrad) is an XOR on the packed bf16 fragment.
//
// Structure: (64*WM) x (64*WN) output tile per workgroup of WM*WN waves, each wave owning a 64x64
// sub-tile = 2x2 MFMA tiles x {re, im} (128 accumulator registers); BK = 32.  Operand tiles go
// global -> LDS directly with global_load_lds_dwordx4 (no VGPR round trip) into a ring of
// STAGES buffers:
//   STAGES = 2: issue tile t+1, compute tile t, __syncthreads (carries vmcnt(0))      ["2-phase"]
//   STAGES = 3: tiles t+1 and t+2 stay in flight across the (raw) barrier; the wait for tile t
//               is a COUNTED s_waitcnt vmcnt(loads per tile) (cdna_hip_programming.md T3/T4).
// LDS image of a [rows][32 k] bf16 plane: 64-B rows, the four 16-B chunks of a row XOR-swizzled
// with (row >> 2) & 3 so that each ds_read_b128 lane group touches 16 distinct 16-B bank slots;
// the LDS-DMA writes lane-linearly, so the swizzle is applied to the per-lane global SOURCE
// address and again on the read (rule 21 of the guide).
// Tile order: XCD-aware (block b runs on XCD b % 8 -> each XCD gets a contiguous range of the
// grouped tile order, GROUP_M row-panels x all column-panels per group) so that the 32-64 tiles
// resident on one XCD share A / B panels in that XCD's private L2.
#include <stdlib.h>

#include "gemm.h"

namespace cplxamd {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;
constexpr int GROUP_M = 4;

__device__ __forceinline__ void glds16(const void* gsrc, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds(
      (const __attribute__((address_space(1))) void*)gsrc,
      (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// Stage one [ROWS x 32] bf16 plane tile: ROWS*4 16-B chunks spread over NT threads.
// LDS chunk p holds global chunk (row = p >> 2, kc = (p & 3) ^ ((row >> 2) & 3)).
template <int ROWS, int NT>
__device__ __forceinline__ void stage_plane(const bf16_t* base, int64_t ld, int row0, int rows,
                                            int k0, char* lds_plane) {
  const int tid = threadIdx.x;
  const int wave_chunk = (tid >> 6) * 64;
  constexpr int PER = ROWS * 4 / NT;
  static_assert(PER >= 1 && ROWS * 4 % NT == 0, "tile / thread-count mismatch");
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int p = j * NT + tid;
    const int row = p >> 2;
    const int kc = (p & 3) ^ ((row >> 2) & 3);
    int grow = row0 + row;
    grow = grow < rows ? grow : rows - 1;  // clamp: out-of-range rows are never stored
    const bf16_t* src = base + (int64_t)grow * ld + k0 + kc * 8;
    glds16(src, lds_plane + (j * NT + wave_chunk) * 16);
  }
}

// One LDS-DMA instruction (piece j of a plane tile), so that the K loop can spread the pieces of
// tile t+2 between the MFMAs of tile t instead of issuing them in one burst.
template <int NT>
__device__ __forceinline__ void stage_piece(const bf16_t* base, int64_t ld, int row0, int rows,
                                            int k0, char* lds_plane, int j) {
  const int tid = threadIdx.x;
  const int wave_chunk = (tid >> 6) * 64;
  const int p = j * NT + tid;
  const int row = p >> 2;
  const int kc = (p & 3) ^ ((row >> 2) & 3);
  int grow = row0 + row;
  grow = grow < rows ? grow : rows - 1;
  glds16(base + (int64_t)grow * ld + k0 + kc * 8, lds_plane + (j * NT + wave_chunk) * 16);
}

__device__ __forceinline__ bf16x8 lds_frag(const char* lds_plane, int row, int kc) {
  const int off = row * 64 + ((kc ^ ((row >> 2) & 3)) << 4);
  return *reinterpret_cast<const bf16x8*>(lds_plane + off);
}

__device__ __forceinline__ bf16x8 neg_frag(bf16x8 v) {
  uint4 u = __builtin_bit_cast(uint4, v);
  u.x ^= 0x80008000u; u.y ^= 0x80008000u; u.z ^= 0x80008000u; u.w ^= 0x80008000u;
  return __builtin_bit_cast(bf16x8, u);
}

template <int N> __device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <bool CPLX, int WM, int WN, int STAGES>
struct Cfg {
  static constexpr int BM = 64 * WM, BN = 64 * WN, NT = 64 * WM * WN;
  static constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
  static constexpr int STAGE_BYTES = (CPLX ? 2 : 1) * (A_BYTES + B_BYTES);
  static constexpr int SMEM = STAGES * STAGE_BYTES;
  // LDS-DMA instructions each thread issues per K tile
  static constexpr int LOADS = (CPLX ? 2 : 1) * (BM * 4 / NT + BN * 4 / NT);
};

template <typename TOUT, bool CPLX, bool CONJ, int WM, int WN, int STAGES, int SCHED = 0>
__global__ __launch_bounds__(64 * WM * WN) void gemm_bf16_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using C = Cfg<CPLX, WM, WN, STAGES>;
  constexpr int BM = C::BM, BN = C::BN, NT = C::NT;

  // ---- XCD-aware grouped tile order ------------------------------------------------------
  const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN;
  const int ntiles = tiles_m * tiles_n;
  int lin = blockIdx.x, split = 0;
  if (g.splits > 1) { split = lin / ntiles; lin -= split * ntiles; }
  int bm, bn;
  if (g.order == 0) {            // natural: consecutive blocks walk N
    bm = lin / tiles_n; bn = lin - bm * tiles_n;
  } else {
    if (g.order == 1) {          // each XCD gets a contiguous range of the grouped order
      const int q = ntiles >> 3, r = ntiles & 7, xcd = lin & 7, idx = lin >> 3;
      lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;  // bijective
    }
    const int GM = g.group_m;
    const int per_group = GM * tiles_n;
    const int grp = lin / per_group, in_grp = lin - grp * per_group;
    const int first_m = grp * GM;
    const int gm = (tiles_m - first_m) < GM ? (tiles_m - first_m) : GM;
    bm = first_m + in_grp % gm; bn = in_grp / gm;
  }
  const int m0 = bm * BM, n0 = bn * BN;

  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int wm = (wid / WN) * 64, wn = (wid % WN) * 64;
  const int l31 = lane & 31, lk = lane >> 5;

  const bf16_t* Ar = (const bf16_t*)g.a_r; const bf16_t* Ai = (const bf16_t*)g.a_i;
  const bf16_t* Br = (const bf16_t*)g.b_r; const bf16_t* Bi = (const bf16_t*)g.b_i;

  f32x16 acc_r[2][2], acc_i[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      acc_r[i][j] = f32x16{0};
      acc_i[i][j] = f32x16{0};
    }

  const int kbase = split * g.kchunk;
  auto stage = [&](int buf, int k0) {
    k0 += kbase;
    char* s = smem + buf * C::STAGE_BYTES;
    stage_plane<BM, NT>(Ar, g.a_rs, m0, g.M, k0, s);
    stage_plane<BN, NT>(Br, g.b_rs, n0, g.N, k0, s + C::A_BYTES);
    if (CPLX) {
      stage_plane<BM, NT>(Ai, g.a_rs, m0, g.M, k0, s + C::A_BYTES + C::B_BYTES);
      stage_plane<BN, NT>(Bi, g.b_rs, n0, g.N, k0, s + 2 * C::A_BYTES + C::B_BYTES);
    }
  };

  // fragments of BOTH K sub-steps are requested up front (16 ds_read_b128), so the second
  // half's LDS latency hides behind the first half's 16 MFMAs
  auto compute = [&](int buf) {
    const char* sA = smem + buf * C::STAGE_BYTES;
    const char* sB = sA + C::A_BYTES;
    const char* sAi = sB + C::B_BYTES;
    const char* sBi = sAi + C::A_BYTES;
    bf16x8 ar[2][2], br[2][2], ai[2][2], bi[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int kc = ks * 2 + lk;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        ar[ks][i] = lds_frag(sA, wm + i * 32 + l31, kc);
        br[ks][i] = lds_frag(sB, wn + i * 32 + l31, kc);
        if (CPLX) {
          ai[ks][i] = lds_frag(sAi, wm + i * 32 + l31, kc);
          bi[ks][i] = lds_frag(sBi, wn + i * 32 + l31, kc);
        }
      }
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 nai[2];
      if (CPLX) {
        // no conj: re -= Ai Bi, im += Ar Bi ; conj(B): re += Ai Bi, im -= Ar Bi
#pragma unroll
        for (int i = 0; i < 2; ++i) nai[i] = neg_frag(CONJ ? ar[ks][i] : ai[ks][i]);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          // operands swapped (B fragment first): the accumulator holds the TRANSPOSED 32x32
          // tile, i.e. lane <-> output row, registers <-> 4-column groups -> 8/16-byte stores
          acc_r[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(br[ks][j], ar[ks][i], acc_r[i][j], 0, 0, 0);
          if (CPLX) {
            acc_i[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(br[ks][j], ai[ks][i], acc_i[i][j], 0, 0, 0);
            if (CONJ) {
              acc_r[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bi[ks][j], ai[ks][i], acc_r[i][j], 0, 0, 0);
              acc_i[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bi[ks][j], nai[i], acc_i[i][j], 0, 0, 0);
            } else {
              acc_r[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bi[ks][j], nai[i], acc_r[i][j], 0, 0, 0);
              acc_i[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bi[ks][j], ar[ks][i], acc_i[i][j], 0, 0, 0);
            }
          }
        }
    }
  };

  // piece q (0 .. LOADS-1) of the tile at k0 into ring slot buf
  auto stage_q = [&](int buf, int k0, int q) {
    k0 += kbase;
    char* s = smem + buf * C::STAGE_BYTES;
    constexpr int PA = BM * 4 / NT, PB = BN * 4 / NT;      // pieces per A / B plane
    if (q < PA) stage_piece<NT>(Ar, g.a_rs, m0, g.M, k0, s, q);
    else if (q < PA + PB) stage_piece<NT>(Br, g.b_rs, n0, g.N, k0, s + C::A_BYTES, q - PA);
    else if (q < 2 * PA + PB) stage_piece<NT>(Ai, g.a_rs, m0, g.M, k0, s + C::A_BYTES + C::B_BYTES, q - PA - PB);
    else stage_piece<NT>(Bi, g.b_rs, n0, g.N, k0, s + 2 * C::A_BYTES + C::B_BYTES, q - 2 * PA - PB);
  };

  // compute(cur) with the LDS-DMA pieces of the tile at knext spread between the MFMAs
  auto compute_interleaved = [&](int buf, int nbuf, int knext, bool do_stage) {
    const char* sA = smem + buf * C::STAGE_BYTES;
    const char* sB = sA + C::A_BYTES;
    const char* sAi = sB + C::B_BYTES;
    const char* sBi = sAi + C::A_BYTES;
    bf16x8 ar[2][2], br[2][2], ai[2][2], bi[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int kc = ks * 2 + lk;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        ar[ks][i] = lds_frag(sA, wm + i * 32 + l31, kc);
        br[ks][i] = lds_frag(sB, wn + i * 32 + l31, kc);
        if (CPLX) {
          ai[ks][i] = lds_frag(sAi, wm + i * 32 + l31, kc);
          bi[ks][i] = lds_frag(sBi, wn + i * 32 + l31, kc);
        }
      }
    }
    int q = 0;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 nai[2];
      if (CPLX) {
#pragma unroll
        for (int i = 0; i < 2; ++i) nai[i] = neg_frag(CONJ ? ar[ks][i] : ai[ks][i]);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc_r[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(br[ks][j], ar[ks][i], acc_r[i][j], 0, 0, 0);
          if (CPLX) {
            acc_i[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(br[ks][j], ai[ks][i], acc_i[i][j], 0, 0, 0);
            if (CONJ) {
              acc_r[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bi[ks][j], ai[ks][i], acc_r[i][j], 0, 0, 0);
              acc_i[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bi[ks][j], nai[i], acc_i[i][j], 0, 0, 0);
            } else {
              acc_r[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bi[ks][j], nai[i], acc_r[i][j], 0, 0, 0);
              acc_i[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bi[ks][j], ar[ks][i], acc_i[i][j], 0, 0, 0);
            }
          }
          // one LDS-DMA piece after each (i, j) group of MFMAs until the tile is fully requested
          if (q < C::LOADS) {
            __builtin_amdgcn_sched_barrier(0);
            if (do_stage) stage_q(nbuf, knext, q);
            __builtin_amdgcn_sched_barrier(0);
            ++q;
          }
        }
    }
  };

  const int klen = g.splits > 1 ? ((g.K - kbase) < g.kchunk ? (g.K - kbase) : g.kchunk) : g.K;
  const int nt = klen / BK;
  if (STAGES == 2) {
    stage(0, 0);
    __syncthreads();  // the workgroup barrier carries vmcnt(0) for the in-flight LDS-DMA
    for (int t = 0; t < nt; ++t) {
      if (t + 1 < nt) stage((t + 1) & 1, (t + 1) * BK);
      compute(t & 1);
      __syncthreads();
    }
  } else {
    // 3-deep ring: tiles t+1, t+2 in flight while tile t is consumed; one raw barrier per tile.
    stage(0, 0);
    if (nt > 1) stage(1, BK);
    int cur = 0;
    for (int t = 0; t < nt; ++t) {
      if (g.dbg & 1) wait_vmcnt<0>(); else if (t + 1 < nt) wait_vmcnt<C::LOADS>(); else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();           // tile t landed for every wave; buffer (t-1)%3 free
      int nxt = cur + 2; nxt = nxt >= 3 ? nxt - 3 : nxt;
      if (SCHED == 0) {            // burst issue of the next tile's LDS-DMA, then compute
        if (t + 2 < nt && !(g.dbg & 1)) stage(nxt, (t + 2) * BK);
        if (!(g.dbg & 2)) compute(cur);
      } else {                     // LDS-DMA pieces spread between the MFMA groups
        compute_interleaved(cur, nxt, (t + 2) * BK, t + 2 < nt && !(g.dbg & 1));
      }
      cur = cur + 1 == 3 ? 0 : cur + 1;
    }
  }

  // epilogue.  Transposed 32x32 C/D layout: output row = lane & 31, output columns
  // 8 q + 4 (lane >> 5) + {0..3} for register group q = r >> 2: one 8-B (bf16) / 16-B (fp32)
  // store per group instead of four scalar ones.
  TOUT* cr = reinterpret_cast<TOUT*>(g.c_r);
  TOUT* ci = reinterpret_cast<TOUT*>(g.c_i);
  if (g.splits > 1) {  // fp32 partial slabs [split][plane][M][ldc]; bias / emul applied by the reducer
    const int64_t slab = (int64_t)g.M * g.ldc;
    cr = reinterpret_cast<TOUT*>(g.ws) + (int64_t)split * (CPLX ? 2 : 1) * slab;
    ci = cr + slab;
  }
  const bool vec_ok = (g.ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(g.c_r) & 15) == 0 &&
                      (!CPLX || (reinterpret_cast<uintptr_t>(g.c_i) & 15) == 0) &&
                      (!g.emul || (reinterpret_cast<uintptr_t>(g.emul) & 15) == 0);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = m0 + wm + i * 32 + l31;
    if (row >= g.M) continue;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = n0 + wn + j * 32 + 8 * q + 4 * lk;
        if (col >= g.N) continue;
        const int64_t o = (int64_t)row * g.ldc + col;
        f4 vr, vi;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          vr.v[e] = acc_r[i][j][4 * q + e];
          vi.v[e] = CPLX ? acc_i[i][j][4 * q + e] : 0.f;
        }
        if (vec_ok && col + 3 < g.N) {
          if (g.bias_r) {
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
            for (int e = 0; e < 4; ++e) vr.v[e] *= m.v[e];
          }
          if (g.accumulate) {
            const f4 p = ld4(cr + o);
#pragma unroll
            for (int e = 0; e < 4; ++e) vr.v[e] += p.v[e];
            if (CPLX) {
              const f4 p2 = ld4(ci + o);
#pragma unroll
              for (int e = 0; e < 4; ++e) vi.v[e] += p2.v[e];
            }
          }
          st4(cr + o, vr);
          if (CPLX) st4(ci + o, vi);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (col + e >= g.N) break;
            float xr = vr.v[e] + (g.bias_r ? g.bias_r[col + e] : 0.f);
            if (g.emul) xr *= g.emul[o + e];
            if (g.accumulate) xr += io<TOUT>::ld(cr + o + e);
            io<TOUT>::st(cr + o + e, xr);
            if (CPLX) {
              float xi = vi.v[e] + (g.bias_i ? g.bias_i[col + e] : 0.f);
              if (g.accumulate) xi += io<TOUT>::ld(ci + o + e);
              io<TOUT>::st(ci + o + e, xi);
            }
          }
        }
      }
    }
  }
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// read-only tuning choice (set once from the environment; DESIGN.md "GEMM variants")
static int gemm_variant() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("CPLXAMD_GEMM_VARIANT");
    v = e ? atoi(e) : 7;
  }
  return v;
}

template <typename TOUT, bool CPLX, bool CONJ, int WM, int WN, int STAGES, int SCHED = 0>
static int launch_cfg(const GemmArgs& g, hipStream_t st) {
  using C = Cfg<CPLX, WM, WN, STAGES>;
  const int64_t tiles = (int64_t)((g.M + C::BM - 1) / C::BM) * ((g.N + C::BN - 1) / C::BN);
  if (tiles > 0x7fffffff) return CPLXAMD_ESHAPE;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_kernel<TOUT, CPLX, CONJ, WM, WN, STAGES, SCHED>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  gemm_bf16_kernel<TOUT, CPLX, CONJ, WM, WN, STAGES, SCHED><<<dim3((unsigned)(tiles * g.splits)), C::NT, C::SMEM, st>>>(g);
  CPLXAMD_CHECK_LAUNCH();
  return 0;
}

static int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

template <typename TOUT, bool CPLX, bool CONJ>
static int launch_variant(const GemmArgs& g0, hipStream_t st) {
  static const int order = env_int("CPLXAMD_GEMM_ORDER", 1), gm = env_int("CPLXAMD_GEMM_GROUP_M", 4),
                   sp = env_int("CPLXAMD_GEMM_SETPRIO", 0);
  GemmArgs g = g0;
  g.order = order; g.group_m = gm > 0 ? gm : 1; g.setprio = sp;
  static const int dbg = env_int("CPLXAMD_GEMM_DBG", 0);
  g.dbg = dbg;
  if (!CPLX) {
    static const int rv = env_int("CPLXAMD_RGEMM_VARIANT", 0);
    if (rv == 1) return launch_cfg<TOUT, CPLX, CONJ, 4, 4, 3>(g, st);   // real: 256x256, 16 waves
    if (rv == 2) return launch_cfg<TOUT, CPLX, CONJ, 4, 4, 2>(g, st);
  }
  switch (gemm_variant()) {
    case 1: return launch_cfg<TOUT, CPLX, CONJ, 2, 2, 3>(g, st);   // 128x128, 3-stage
    case 2: return launch_cfg<TOUT, CPLX, CONJ, 4, 2, 2>(g, st);   // 256x128, 2-stage
    case 3: return launch_cfg<TOUT, CPLX, CONJ, 4, 2, 3>(g, st);   // 256x128, 3-stage
    case 4: return launch_cfg<TOUT, CPLX, CONJ, 2, 4, 3>(g, st);   // 128x256, 3-stage
    case 7: return launch_cfg<TOUT, CPLX, CONJ, 4, 2, 3, 1>(g, st);   // 256x128, 3-stage, interleaved LDS-DMA
    case 0: return launch_cfg<TOUT, CPLX, CONJ, 2, 2, 2>(g, st);   // 128x128, 2-stage
    default: return launch_cfg<TOUT, CPLX, CONJ, 4, 2, 3, 1>(g, st);  // = 7 (default)
  }
}

// split-K plan for the default 256x128 tile: use it when the tile count leaves CUs idle
static int plan_splits(int M, int N, int K) {
  const int64_t tiles = (int64_t)((M + 255) / 256) * ((N + 127) / 128);
  if (tiles >= 192 || K < 64 * BK) return 1;
  int s = (int)(256 / tiles);
  const int maxs = K / (32 * BK);          // >= 32 K tiles per split
  if (s > maxs) s = maxs;
  if (s > 16) s = 16;
  return s < 2 ? 1 : s;
}

int64_t gemm_bf16_ws_bytes(int M, int N, int K, bool cplx) {
  const int s = plan_splits(M, N, K);
  return s > 1 ? (int64_t)s * (cplx ? 2 : 1) * M * N * (int64_t)sizeof(float) : 0;
}

// out = sum_s slab[s] (+ bias[n]) (* emul) (+ out)   for one plane
__global__ __launch_bounds__(256) void gemm_slab_reduce_kernel(const float* slabs, int splits,
                                                               int64_t slab_stride, int M, int N,
                                                               int64_t ldc, const float* bias,
                                                               const float* emul, int accumulate,
                                                               float* out) {
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
      for (int e = 0; e < 4; ++e) acc.v[e] *= m.v[e];
    }
    if (accumulate) {
      const f4 o = ld4(out + 4 * i);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc.v[e] += o.v[e];
    }
    st4(out + 4 * i, acc);
  }
}

template <bool CPLX>
static int launch_gemm_bf16_splitk(const GemmArgs& g0, int splits, hipStream_t st) {
  GemmArgs g = g0;
  g.splits = splits;
  g.kchunk = ((g.K / BK + splits - 1) / splits) * BK;
  const bool conj = CPLX && g.conj_b;
  // slabs use ldc = N (dense); the kernel writes float partials, no bias / emul / accumulate
  GemmArgs k = g;
  k.ldc = g.N; k.bias_r = k.bias_i = nullptr; k.emul = nullptr; k.accumulate = 0;
  const int rc = conj ? launch_variant<float, CPLX, true>(k, st) : launch_variant<float, CPLX, false>(k, st);
  if (rc) return rc;
  const int64_t slab = (int64_t)g.M * g.N, stride = (CPLX ? 2 : 1) * slab;
  const int grid = stream_grid(slab >> 2, 256);
  if (g.ldc != g.N) return CPLXAMD_ESHAPE;
  gemm_slab_reduce_kernel<<<grid, 256, 0, st>>>((const float*)g.ws, splits, stride, g.M, g.N, g.ldc,
                                                g.bias_r, g.emul, g.accumulate, (float*)g.c_r);
  CPLXAMD_CHECK_LAUNCH();
  if (CPLX) {
    gemm_slab_reduce_kernel<<<grid, 256, 0, st>>>((const float*)g.ws + slab, splits, stride, g.M, g.N,
                                                  g.ldc, g.bias_i, nullptr, g.accumulate,
                                                  (float*)g.c_i);
    CPLXAMD_CHECK_LAUNCH();
  }
  return 0;
}

template <bool CPLX>
int launch_gemm_bf16(const GemmArgs& g, int out_dtype, hipStream_t st) {
  if (g.a_cs != 1 || g.b_cs != 1 || g.K < BK || (g.K % BK) != 0) return CPLXAMD_ESHAPE;
  if ((g.a_rs % 8) != 0 || (g.b_rs % 8) != 0) return CPLXAMD_ESHAPE;
  if (!aligned16(g.a_r) || !aligned16(g.b_r)) return CPLXAMD_ESHAPE;
  if (CPLX && (!aligned16(g.a_i) || !aligned16(g.b_i))) return CPLXAMD_ESHAPE;
  if (g.M <= 0 || g.N <= 0) return 0;
  if (out_dtype == CPLXAMD_F32 && g.ws && g.ldc == g.N && (g.N & 3) == 0 && (gemm_variant() == 3 || gemm_variant() == 7)) {
    const int splits = plan_splits(g.M, g.N, g.K);
    if (splits > 1 && g.ws_bytes >= gemm_bf16_ws_bytes(g.M, g.N, g.K, CPLX))
      return launch_gemm_bf16_splitk<CPLX>(g, splits, st);
  }
  const bool conj = CPLX && g.conj_b;
  if (out_dtype == CPLXAMD_BF16)
    return conj ? launch_variant<bf16_t, CPLX, true>(g, st) : launch_variant<bf16_t, CPLX, false>(g, st);
  if (out_dtype == CPLXAMD_F32)
    return conj ? launch_variant<float, CPLX, true>(g, st) : launch_variant<float, CPLX, false>(g, st);
  return CPLXAMD_EINVAL;
}

template int launch_gemm_bf16<true>(const GemmArgs&, int, hipStream_t);
template int launch_gemm_bf16<false>(const GemmArgs&, int, hipStream_t);

}  // namespace cplxamd
