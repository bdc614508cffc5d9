// Generic complex / real GEMM on the exact-f32 matrix cores (v_mfma_f32_32x32x2_f32):
//   C[m,n] = sum_k A[m,k] * op(B[n,k]) (+ bias[n]),   any element strides, any M, N, K,
//   float32 or bf16 inputs (bf16 is widened exactly), float32 accumulation.
// This is the parity path (fp32 results == an fmaf chain in k order) and the fallback for
// shapes / layouts the bf16 fast path (gemm_bf16_impl.h) does not take.
//
// Reference arithmetic: cplx.linear_naive cplx.py:634-648, Cplx.__matmul__ cplx.py:167-174,
// F.linear in the LRT variance term nn/relevance/complex/base.py:50-54.
//
// Tiling: 128x128 outputs per 256-thread block (4 waves as 2x2, each 64x64 = 2x2 MFMA tiles x
// {re, im}), BK = 16, operands staged k-major in LDS (register-prefetched, double-buffered) so
// that every ds_read_b32 is conflict-free; split-K for few-tile / long-K shapes.
#include "gemm.h"

namespace cplxamd {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int GBK = 16;
// Tile = 64 NB x 64 NB outputs per 256-thread block (waves 2 x 2, each NB x NB MFMA tiles x {re, im}): NB = 2 for problems
// of many tiles; NB = 1 for small and skinny ones (a 256 x 10 head on 128-wide tiles ran 92 % of its MFMAs on padding and
// had 2 tiles to split K over).
template <int NB> struct GT {
  static constexpr int BM = 64 * NB, LD = BM + 1, PLANE = GBK * LD, PER = BM * GBK / 256;   // PER: elements per thread and plane
};

template <typename T>
__device__ __forceinline__ float ldg(const void* p, int64_t off) {
  return io<T>::ld(reinterpret_cast<const T*>(p) + off);
}

// One thread's 8 elements of a [128 rows x GBK] operand tile: global -> registers (fetch), then
// registers -> LDS as dst[k][row] (commit).  Splitting the two lets the loads of tile t+1 fly
// while the MFMAs of tile t run.
template <int NB> struct TileRegs { float v[GT<NB>::PER]; };

template <typename TIN, int NB>
__device__ __forceinline__ TileRegs<NB> fetch_tile(const void* src, int64_t rs, int64_t cs, int row0,
                                               int k0, int rows, int K) {
  constexpr int BM = GT<NB>::BM, PER = GT<NB>::PER, KSTEP = 256 / BM;
  const int t = threadIdx.x;
  TileRegs<NB> o;
  if (cs == 1 || rs != 1) {  // k fastest across threads
    const int k = t & 15, rb = t >> 4;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int gr = row0 + rb + 16 * j, gk = k0 + k;
      o.v[j] = (gr < rows && gk < K) ? ldg<TIN>(src, (int64_t)gr * rs + (int64_t)gk * cs) : 0.0f;
    }
  } else {  // rows fastest across threads
    const int r = t % BM, kb = t / BM;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int gr = row0 + r, gk = k0 + kb + KSTEP * j;
      o.v[j] = (gr < rows && gk < K) ? ldg<TIN>(src, (int64_t)gr * rs + (int64_t)gk * cs) : 0.0f;
    }
  }
  return o;
}

template <int NB>
__device__ __forceinline__ void commit_tile(float* dst, const TileRegs<NB>& o, int64_t rs, int64_t cs) {
  constexpr int BM = GT<NB>::BM, PER = GT<NB>::PER, KSTEP = 256 / BM, LD = GT<NB>::LD;
  const int t = threadIdx.x;
  if (cs == 1 || rs != 1) {
    const int k = t & 15, rb = t >> 4;
#pragma unroll
    for (int j = 0; j < PER; ++j) dst[k * LD + rb + 16 * j] = o.v[j];
  } else {
    const int r = t % BM, kb = t / BM;
#pragma unroll
    for (int j = 0; j < PER; ++j) dst[(kb + KSTEP * j) * LD + r] = o.v[j];
  }
}

// dynamic LDS: [2 buffers][A_r, B_r (, A_i, B_i)][GBK][GLD] floats (66 KiB for the complex kernel)
template <typename TIN, typename TOUT, bool CPLX, int NB>
__global__ __launch_bounds__(256, 2) void gemm_generic_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) float gsm[];
  constexpr int NPL = CPLX ? 4 : 2;
  constexpr int GBM = GT<NB>::BM, GBN = GT<NB>::BM, GLD = GT<NB>::LD, GPLANE = GT<NB>::PLANE;
  auto plane = [&](int buf, int which) { return gsm + (buf * NPL + which) * GPLANE; };   // 0 Ar 1 Br 2 Ai 3 Bi

  const int m0 = blockIdx.y * GBM, n0 = blockIdx.x * GBN;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int wm = (wid >> 1) * 32 * NB, wn = (wid & 1) * 32 * NB;    // 2 x 2 waves, NB x NB MFMA tiles each
  const int l31 = lane & 31, lk = lane >> 5;
  const float sgn = g.conj_b ? -1.0f : 1.0f;

  f32x16 acc_r[NB][NB], acc_i[NB][NB];
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      acc_r[i][j] = f32x16{0};
      acc_i[i][j] = f32x16{0};
    }

  // split-K: this block covers K range [kb, ke)
  int kb = 0, ke = g.K;
  if (g.splits > 1) {
    kb = blockIdx.z * g.kchunk;
    ke = kb + g.kchunk < g.K ? kb + g.kchunk : g.K;
  }
  // batched launch: blockIdx.z is the batch entry (split-K is off then)
  const int64_t bz = g.batch > 1 ? blockIdx.z : 0;
  const TIN* pa_r = reinterpret_cast<const TIN*>(g.a_r) + bz * g.a_bs;
  const TIN* pb_r = reinterpret_cast<const TIN*>(g.b_r) + bz * g.b_bs;
  const TIN* pa_i = CPLX ? reinterpret_cast<const TIN*>(g.a_i) + bz * g.a_bs : nullptr;
  const TIN* pb_i = CPLX ? reinterpret_cast<const TIN*>(g.b_i) + bz * g.b_bs : nullptr;
  TileRegs<NB> ra, rb, rai, rbi;
  auto fetch = [&](int k0) {
    ra = fetch_tile<TIN, NB>(pa_r, g.a_rs, g.a_cs, m0, k0, g.M, ke);
    rb = fetch_tile<TIN, NB>(pb_r, g.b_rs, g.b_cs, n0, k0, g.N, ke);
    if (CPLX) {
      rai = fetch_tile<TIN, NB>(pa_i, g.a_rs, g.a_cs, m0, k0, g.M, ke);
      rbi = fetch_tile<TIN, NB>(pb_i, g.b_rs, g.b_cs, n0, k0, g.N, ke);
    }
  };
  if (kb < ke) fetch(kb);
  int buf = 0;
  for (int k0 = kb; k0 < ke; k0 += GBK, buf ^= 1) {
    commit_tile<NB>(plane(buf, 0), ra, g.a_rs, g.a_cs);
    commit_tile<NB>(plane(buf, 1), rb, g.b_rs, g.b_cs);
    if (CPLX) {
      commit_tile<NB>(plane(buf, 2), rai, g.a_rs, g.a_cs);
      commit_tile<NB>(plane(buf, 3), rbi, g.b_rs, g.b_cs);
    }
    __syncthreads();                       // tile visible; the other buffer is free again
    if (k0 + GBK < ke) fetch(k0 + GBK);     // in flight during the MFMAs below
    const float* Ar = plane(buf, 0);
    const float* Br = plane(buf, 1);
    const float* Ai = plane(buf, CPLX ? 2 : 0);
    const float* Bi = plane(buf, CPLX ? 3 : 1);
#pragma unroll
    for (int kk = 0; kk < GBK; kk += 2) {
      float ar[NB], br[NB], ai[NB], bi[NB];
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        ar[i] = Ar[(kk + lk) * GLD + wm + i * 32 + l31];
        br[i] = Br[(kk + lk) * GLD + wn + i * 32 + l31];
        if (CPLX) {
          ai[i] = Ai[(kk + lk) * GLD + wm + i * 32 + l31];
          bi[i] = sgn * Bi[(kk + lk) * GLD + wn + i * 32 + l31];
        }
      }
#pragma unroll
      for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          acc_r[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[i], br[j], acc_r[i][j], 0, 0, 0);
          if (CPLX) {
            acc_r[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(-ai[i], bi[j], acc_r[i][j], 0, 0, 0);
            acc_i[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[i], bi[j], acc_i[i][j], 0, 0, 0);
            acc_i[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ai[i], br[j], acc_i[i][j], 0, 0, 0);
          }
        }
    }
  }

  // C/D layout of a 32x32 tile: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  TOUT* cr = reinterpret_cast<TOUT*>(g.c_r) + bz * g.c_bs;
  TOUT* ci = reinterpret_cast<TOUT*>(g.c_i) + (CPLX ? bz * g.c_bs : 0);
  const float beta = gemm_beta(g);
  float* slab = g.splits > 1 ? reinterpret_cast<float*>(g.ws) + (int64_t)blockIdx.z * (CPLX ? 2 : 1) * g.M * g.N
                             : nullptr;   // fp32 partial slabs [split][plane][M][N]; bias / emul / accumulate: the reducer
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int col = n0 + wn + j * 32 + l31;
    if (col >= g.N) continue;
    const float b_r = g.bias_r ? g.bias_r[col] : 0.0f;
    const float b_i = (CPLX && g.bias_i) ? g.bias_i[col] : 0.0f;
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (row >= g.M) continue;
        if (slab) {
          slab[(int64_t)row * g.N + col] = acc_r[i][j][r];
          if (CPLX) slab[(int64_t)g.M * g.N + (int64_t)row * g.N + col] = acc_i[i][j][r];
          continue;
        }
        const int64_t o = (int64_t)row * g.ldc + col;
        float vr = acc_r[i][j][r] + b_r;
        if (g.emul) vr *= gemm_emul(g, g.emul[o]);
        if (g.accumulate) vr += beta * io<TOUT>::ld(cr + o);
        io<TOUT>::st(cr + o, vr);
        if (CPLX) {
          float vi = acc_i[i][j][r] + b_i;
          if (g.emul && g.emul_both) vi *= gemm_emul(g, g.emul[o]);
          if (g.accumulate) vi += beta * io<TOUT>::ld(ci + o);
          io<TOUT>::st(ci + o, vi);
        }
      }
  }
}

// tile edge: 64 for small / skinny problems (at most 32 tiles of 128 x 128), else 128
static int gemm_generic_tile(int M, int N) {
  return (int64_t)((M + 127) / 128) * ((N + 127) / 128) <= 32 ? 64 : 128;
}

// split-K plan of the generic kernel: few output tiles and a long K (small layers at large
// batch, heads with few outputs) would leave most CUs idle and expose the load latency serially
int gemm_generic_splits(int M, int N, int K) {
  const int bm = gemm_generic_tile(M, N);
  const int64_t tiles = (int64_t)((M + bm - 1) / bm) * ((N + bm - 1) / bm);
  // a layer that is ONE or a few tiles (cfg1: 64 x 128 x 128) runs its whole K loop of 64-cycle f32 MFMAs on one
  // CU -- 24 us (real) / 52 us (complex) per launch, which IS the launch-bound config's step time: such shapes
  // split already from K = 64 with >= 2 K tiles per split
  const bool tiny = tiles <= 8;
  if (tiles >= 128 || K < (tiny ? 4 : 16) * GBK) return 1;
  int64_t s = 512 / tiles;
  const int64_t maxs = K / ((tiny ? 2 : 4) * GBK);     // >= 4 (tiny: 2) K tiles per split
  if (s > maxs) s = maxs;
  if (s > 64) s = 64;
  return s < 2 ? 1 : (int)s;
}

int64_t gemm_generic_ws_bytes(int M, int N, int K, bool cplx) {
  const int s = gemm_generic_splits(M, N, K);
  return s > 1 ? (int64_t)s * (cplx ? 2 : 1) * M * N * (int64_t)sizeof(float) : 0;
}

// out (plane) = sum_s slab[s] (+ bias[n]) (* emul) (+ out), any output type / leading dimension
template <typename TOUT>
__global__ __launch_bounds__(256) void generic_slab_reduce_kernel(const float* slabs, int splits,
                                                                  int64_t slab_stride, int M, int N,
                                                                  int64_t ldc, const float* bias_r, const float* bias_i,
                                                                  const float* emul, int emul_both, int emul_exp,
                                                                  int accumulate, const float* beta_p, TOUT* out_r,
                                                                  TOUT* out_i) {
  // blockIdx.y = plane (complex: both planes in one launch)
  const int pl = blockIdx.y;
  const float* bias = pl ? bias_i : bias_r;
  TOUT* out = pl ? out_i : out_r;
  if (pl && !emul_both) emul = nullptr;
  slabs += (int64_t)pl * M * N;
  const float beta = (accumulate && beta_p) ? *beta_p : 1.0f;
  const int64_t n = (int64_t)M * N, stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    float a4[4] = {0.f, 0.f, 0.f, 0.f};        // four loads in flight (a head's 49 splits were 49 dependent round trips)
    int s = 0;
    for (; s + 3 < splits; s += 4) {
#pragma unroll
      for (int u = 0; u < 4; ++u) a4[u] += slabs[(int64_t)(s + u) * slab_stride + i];
    }
    for (; s < splits; ++s) a4[0] += slabs[(int64_t)s * slab_stride + i];
    float acc = (a4[0] + a4[1]) + (a4[2] + a4[3]);
    const int row = (int)(i / N), col = (int)(i - (int64_t)row * N);
    const int64_t o = (int64_t)row * ldc + col;
    if (bias) acc += bias[col];
    if (emul) acc *= emul_exp ? expf(emul[o]) : emul[o];
    if (accumulate) acc += beta * io<TOUT>::ld(out + o);
    io<TOUT>::st(out + o, acc);
  }
}

template <bool CPLX>
int launch_gemm_generic(const GemmArgs& g0, int in_dtype, int out_dtype, hipStream_t st) {
  if (g0.M <= 0 || g0.N <= 0) return 0;
  GemmArgs g = g0;
  g.splits = 1;
  const int want = g.batch > 1 ? 1 : gemm_generic_splits(g.M, g.N, g.K);
  if (want > 1 && g.ws && g.ws_bytes >= gemm_generic_ws_bytes(g.M, g.N, g.K, CPLX)) {
    g.splits = want;
    g.kchunk = (((g.K + GBK - 1) / GBK + want - 1) / want) * GBK;   // whole K tiles, covers the tail
  }
  if (g.batch > 65535) return CPLXAMD_ESHAPE;
  const int bm = gemm_generic_tile(g.M, g.N);
  dim3 grid((g.N + bm - 1) / bm, (g.M + bm - 1) / bm, g.batch > 1 ? g.batch : g.splits);
  const int smem = 2 * (CPLX ? 4 : 2) * (bm == 64 ? GT<1>::PLANE : GT<2>::PLANE) * (int)sizeof(float);
  auto go = [&](auto kern) -> int {
    if (smem > 64 * 1024) {
      hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
      if (e != hipSuccess) return (int)e;
    }
    kern<<<grid, 256, smem, st>>>(g);
    CPLXAMD_CHECK_LAUNCH();
    return 0;
  };
  int rc;
#define GEN_GO(TI, TO) (bm == 64 ? go(gemm_generic_kernel<TI, TO, CPLX, 1>) : go(gemm_generic_kernel<TI, TO, CPLX, 2>))
  if (in_dtype == CPLXAMD_F32 && out_dtype == CPLXAMD_F32) rc = GEN_GO(float, float);
  else if (in_dtype == CPLXAMD_BF16 && out_dtype == CPLXAMD_F32) rc = GEN_GO(bf16_t, float);
  else if (in_dtype == CPLXAMD_BF16 && out_dtype == CPLXAMD_BF16) rc = GEN_GO(bf16_t, bf16_t);
  else if (in_dtype == CPLXAMD_F32 && out_dtype == CPLXAMD_BF16) rc = GEN_GO(float, bf16_t);
  else return CPLXAMD_EINVAL;
#undef GEN_GO
  if (rc) return rc;
  if (g.splits > 1) {
    const int64_t slab = (int64_t)g.M * g.N, stride = (CPLX ? 2 : 1) * slab;
    const dim3 rgrid(stream_grid(slab, 256), CPLX ? 2 : 1);
    const float* src = (const float*)g.ws;
    if (out_dtype == CPLXAMD_F32)
      generic_slab_reduce_kernel<float><<<rgrid, 256, 0, st>>>(src, g.splits, stride, g.M, g.N, g.ldc, g.bias_r, g.bias_i,
                                                              g.emul, g.emul_both, g.emul_exp, g.accumulate, g.beta,
                                                              (float*)g.c_r, (float*)g.c_i);
    else
      generic_slab_reduce_kernel<bf16_t><<<rgrid, 256, 0, st>>>(src, g.splits, stride, g.M, g.N, g.ldc, g.bias_r, g.bias_i,
                                                               g.emul, g.emul_both, g.emul_exp, g.accumulate, g.beta,
                                                               (bf16_t*)g.c_r, (bf16_t*)g.c_i);
    CPLXAMD_CHECK_LAUNCH();
  }
  return 0;
}

template int launch_gemm_generic<true>(const GemmArgs&, int, int, hipStream_t);
template int launch_gemm_generic<false>(const GemmArgs&, int, int, hipStream_t);

}  // namespace cplxamd
