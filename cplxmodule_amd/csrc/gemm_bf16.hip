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
// Structure (cdna_hip_programming.md sec. 5, "minimum 2-phase"): 128x128 (complex) output tile,
// BK = 32, 256 threads = 4 waves as 2x2, each wave 64x64 outputs = 2x2 MFMA tiles x {re, im}
// (128 accumulator registers).  Operand tiles go global -> LDS directly with
// global_load_lds_dwordx4 (no VGPR round trip), double buffered, one barrier per K tile.
// LDS image of a [128 rows][32 k] bf16 plane: 64 B rows, the four 16-B chunks of a row
// XOR-swizzled with (row >> 2) & 3 so that each ds_read_b128 lane group touches 16 distinct
// 16-B bank slots (conflict-free); because the LDS-DMA writes lane-linearly, the swizzle is
// applied to the per-lane global SOURCE address and again on the read (rule 21 of the guide).
#include "gemm.h"

namespace cplxamd {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int PLANE_BYTES = BM * BK * 2;  // 8 KiB per operand plane per stage

template <bool CPLX>
struct Smem {
  static constexpr int NPLANES = CPLX ? 4 : 2;            // Ar, (Ai), Br, (Bi)
  static constexpr int STAGE_BYTES = NPLANES * PLANE_BYTES;
  static constexpr int TOTAL = 2 * STAGE_BYTES;           // 64 KiB complex, 32 KiB real
};

__device__ __forceinline__ void glds16(const void* gsrc, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds(
      (const __attribute__((address_space(1))) void*)gsrc,
      (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// Stage one [128 x 32] bf16 plane tile: 512 16-B chunks, 2 per thread.
// LDS chunk p = j*256 + tid holds global chunk (row = p >> 2, kc = (p & 3) ^ ((row >> 2) & 3)).
__device__ __forceinline__ void stage_plane(const bf16_t* base, int64_t ld, int row0, int rows,
                                            int k0, char* lds_plane) {
  const int tid = threadIdx.x;
  const int wave_chunk = (tid >> 6) * 64;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int p = j * 256 + tid;
    const int row = p >> 2;
    const int kc = (p & 3) ^ ((row >> 2) & 3);
    int grow = row0 + row;
    grow = grow < rows ? grow : rows - 1;  // clamp: out-of-range rows are never stored
    const bf16_t* src = base + (int64_t)grow * ld + k0 + kc * 8;
    glds16(src, lds_plane + (j * 256 + wave_chunk) * 16);
  }
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

template <typename TOUT, bool CPLX, bool CONJ>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using S = Smem<CPLX>;

  // tile coordinates: consecutive blocks walk N first (they share the A row panel)
  const int tiles_n = (g.N + BN - 1) / BN;
  const int bm = blockIdx.x / tiles_n, bn = blockIdx.x % tiles_n;
  const int m0 = bm * BM, n0 = bn * BN;

  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int wm = (wid >> 1) * 64, wn = (wid & 1) * 64;
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

  auto stage = [&](int buf, int k0) {
    char* s = smem + buf * S::STAGE_BYTES;
    stage_plane(Ar, g.a_rs, m0, g.M, k0, s);
    stage_plane(Br, g.b_rs, n0, g.N, k0, s + PLANE_BYTES);
    if (CPLX) {
      stage_plane(Ai, g.a_rs, m0, g.M, k0, s + 2 * PLANE_BYTES);
      stage_plane(Bi, g.b_rs, n0, g.N, k0, s + 3 * PLANE_BYTES);
    }
  };

  const int nt = g.K / BK;
  stage(0, 0);
  __syncthreads();  // the workgroup barrier carries vmcnt(0) for the in-flight LDS-DMA

  for (int t = 0; t < nt; ++t) {
    const int cur = t & 1;
    if (t + 1 < nt) stage(cur ^ 1, (t + 1) * BK);
    const char* s = smem + cur * S::STAGE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int kc = ks * 2 + lk;
      bf16x8 ar[2], br[2], ai[2], bi[2], nai[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        ar[i] = lds_frag(s, wm + i * 32 + l31, kc);
        br[i] = lds_frag(s + PLANE_BYTES, wn + i * 32 + l31, kc);
        if (CPLX) {
          ai[i] = lds_frag(s + 2 * PLANE_BYTES, wm + i * 32 + l31, kc);
          bi[i] = lds_frag(s + 3 * PLANE_BYTES, wn + i * 32 + l31, kc);
        }
      }
      if (CPLX) {
        // no conj: re -= Ai Bi, im += Ar Bi ; conj(B): re += Ai Bi, im -= Ar Bi
#pragma unroll
        for (int i = 0; i < 2; ++i) nai[i] = neg_frag(CONJ ? ar[i] : ai[i]);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc_r[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[i], br[j], acc_r[i][j], 0, 0, 0);
          if (CPLX) {
            acc_i[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ai[i], br[j], acc_i[i][j], 0, 0, 0);
            if (CONJ) {
              acc_r[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ai[i], bi[j], acc_r[i][j], 0, 0, 0);
              acc_i[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(nai[i], bi[j], acc_i[i][j], 0, 0, 0);
            } else {
              acc_r[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(nai[i], bi[j], acc_r[i][j], 0, 0, 0);
              acc_i[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[i], bi[j], acc_i[i][j], 0, 0, 0);
            }
          }
        }
    }
    __syncthreads();
  }

  // epilogue.  32x32 C/D layout: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  TOUT* cr = reinterpret_cast<TOUT*>(g.c_r);
  TOUT* ci = reinterpret_cast<TOUT*>(g.c_i);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + wn + j * 32 + l31;
    if (col >= g.N) continue;
    const float b_r = g.bias_r ? g.bias_r[col] : 0.0f;
    const float b_i = (CPLX && g.bias_i) ? g.bias_i[col] : 0.0f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (row >= g.M) continue;
        const int64_t o = (int64_t)row * g.ldc + col;
        float vr = acc_r[i][j][r] + b_r;
        if (g.emul) vr *= g.emul[o];
        if (g.accumulate) vr += io<TOUT>::ld(cr + o);
        io<TOUT>::st(cr + o, vr);
        if (CPLX) {
          float vi = acc_i[i][j][r] + b_i;
          if (g.accumulate) vi += io<TOUT>::ld(ci + o);
          io<TOUT>::st(ci + o, vi);
        }
      }
    }
  }
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <bool CPLX>
int launch_gemm_bf16(const GemmArgs& g, int out_dtype, hipStream_t st) {
  if (g.a_cs != 1 || g.b_cs != 1 || g.K < BK || (g.K % BK) != 0) return CPLXAMD_ESHAPE;
  if ((g.a_rs % 8) != 0 || (g.b_rs % 8) != 0) return CPLXAMD_ESHAPE;
  if (!aligned16(g.a_r) || !aligned16(g.b_r)) return CPLXAMD_ESHAPE;
  if (CPLX && (!aligned16(g.a_i) || !aligned16(g.b_i))) return CPLXAMD_ESHAPE;
  if (g.M <= 0 || g.N <= 0) return 0;
  const int64_t tiles = (int64_t)((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
  if (tiles > 0x7fffffff) return CPLXAMD_ESHAPE;
  const int smem = Smem<CPLX>::TOTAL;
  dim3 grid((unsigned)tiles);
#define LAUNCH(TOUT, CONJ)                                                                   \
  do {                                                                                       \
    static bool attr_set = false;                                                            \
    if (!attr_set) {                                                                         \
      hipFuncSetAttribute((const void*)gemm_bf16_kernel<TOUT, CPLX, CONJ>,                   \
                          hipFuncAttributeMaxDynamicSharedMemorySize, smem);                 \
      attr_set = true;                                                                       \
    }                                                                                        \
    gemm_bf16_kernel<TOUT, CPLX, CONJ><<<grid, 256, smem, st>>>(g);                          \
  } while (0)
  const bool conj = CPLX && g.conj_b;
  if (out_dtype == CPLXAMD_BF16) {
    if (conj) LAUNCH(bf16_t, true); else LAUNCH(bf16_t, false);
  } else if (out_dtype == CPLXAMD_F32) {
    if (conj) LAUNCH(float, true); else LAUNCH(float, false);
  } else {
    return CPLXAMD_EINVAL;
  }
#undef LAUNCH
  CPLXAMD_CHECK_LAUNCH();
  return 0;
}

template int launch_gemm_bf16<true>(const GemmArgs&, int, hipStream_t);
template int launch_gemm_bf16<false>(const GemmArgs&, int, hipStream_t);

}  // namespace cplxamd
