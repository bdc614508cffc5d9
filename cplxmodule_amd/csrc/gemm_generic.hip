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

constexpr int GBM = 128, GBN = 128, GBK = 16, GLD = GBM + 1;
constexpr int GPLANE = GBK * GLD;                      // floats per staged operand plane
constexpr int GPER = GBM * GBK / 256;                  // elements per thread and operand plane (8)

template <typename T>
__device__ __forceinline__ float ldg(const void* p, int64_t off) {
  return io<T>::ld(reinterpret_cast<const T*>(p) + off);
}

// One thread's 8 elements of a [128 rows x GBK] operand tile: global -> registers (fetch), then
// registers -> LDS as dst[k][row] (commit).  Splitting the two lets the loads of tile t+1 fly
// while the MFMAs of tile t run.
struct TileRegs { float v[GPER]; };

template <typename TIN>
__device__ __forceinline__ TileRegs fetch_tile(const void* src, int64_t rs, int64_t cs, int row0,
                                               int k0, int rows, int K) {
  const int t = threadIdx.x;
  TileRegs o;
  if (cs == 1 || rs != 1) {  // k fastest across threads
    const int k = t & 15, rb = t >> 4;
#pragma unroll
    for (int j = 0; j < GPER; ++j) {
      const int gr = row0 + rb + 16 * j, gk = k0 + k;
      o.v[j] = (gr < rows && gk < K) ? ldg<TIN>(src, (int64_t)gr * rs + (int64_t)gk * cs) : 0.0f;
    }
  } else {  // rows fastest across threads
    const int r = t & 127, kb = t >> 7;
#pragma unroll
    for (int j = 0; j < GPER; ++j) {
      const int gr = row0 + r, gk = k0 + kb + 2 * j;
      o.v[j] = (gr < rows && gk < K) ? ldg<TIN>(src, (int64_t)gr * rs + (int64_t)gk * cs) : 0.0f;
    }
  }
  return o;
}

__device__ __forceinline__ void commit_tile(float* dst, const TileRegs& o, int64_t rs, int64_t cs) {
  const int t = threadIdx.x;
  if (cs == 1 || rs != 1) {
    const int k = t & 15, rb = t >> 4;
#pragma unroll
    for (int j = 0; j < GPER; ++j) dst[k * GLD + rb + 16 * j] = o.v[j];
  } else {
    const int r = t & 127, kb = t >> 7;
#pragma unroll
    for (int j = 0; j < GPER; ++j) dst[(kb + 2 * j) * GLD + r] = o.v[j];
  }
}

// dynamic LDS: [2 buffers][A_r, B_r (, A_i, B_i)][GBK][GLD] floats (66 KiB for the complex kernel)
template <typename TIN, typename TOUT, bool CPLX>
__global__ __launch_bounds__(256, 2) void gemm_generic_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) float gsm[];
  constexpr int NPL = CPLX ? 4 : 2;
  auto plane = [&](int buf, int which) { return gsm + (buf * NPL + which) * GPLANE; };   // 0 Ar 1 Br 2 Ai 3 Bi

  const int m0 = blockIdx.y * GBM, n0 = blockIdx.x * GBN;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int wm = (wid >> 1) * 64, wn = (wid & 1) * 64;    // 2 x 2 waves, 64 x 64 each = 2 x 2 MFMA tiles
  const int l31 = lane & 31, lk = lane >> 5;
  const float sgn = g.conj_b ? -1.0f : 1.0f;

  f32x16 acc_r[2][2], acc_i[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
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
  TileRegs ra, rb, rai, rbi;
  auto fetch = [&](int k0) {
    ra = fetch_tile<TIN>(pa_r, g.a_rs, g.a_cs, m0, k0, g.M, ke);
    rb = fetch_tile<TIN>(pb_r, g.b_rs, g.b_cs, n0, k0, g.N, ke);
    if (CPLX) {
      rai = fetch_tile<TIN>(pa_i, g.a_rs, g.a_cs, m0, k0, g.M, ke);
      rbi = fetch_tile<TIN>(pb_i, g.b_rs, g.b_cs, n0, k0, g.N, ke);
    }
  };
  if (kb < ke) fetch(kb);
  int buf = 0;
  for (int k0 = kb; k0 < ke; k0 += GBK, buf ^= 1) {
    commit_tile(plane(buf, 0), ra, g.a_rs, g.a_cs);
    commit_tile(plane(buf, 1), rb, g.b_rs, g.b_cs);
    if (CPLX) {
      commit_tile(plane(buf, 2), rai, g.a_rs, g.a_cs);
      commit_tile(plane(buf, 3), rbi, g.b_rs, g.b_cs);
    }
    __syncthreads();                       // tile visible; the other buffer is free again
    if (k0 + GBK < ke) fetch(k0 + GBK);     // in flight during the MFMAs below
    const float* Ar = plane(buf, 0);
    const float* Br = plane(buf, 1);
    const float* Ai = plane(buf, CPLX ? 2 : 0);
    const float* Bi = plane(buf, CPLX ? 3 : 1);
#pragma unroll
    for (int kk = 0; kk < GBK; kk += 2) {
      float ar[2], br[2], ai[2], bi[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        ar[i] = Ar[(kk + lk) * GLD + wm + i * 32 + l31];
        br[i] = Br[(kk + lk) * GLD + wn + i * 32 + l31];
        if (CPLX) {
          ai[i] = Ai[(kk + lk) * GLD + wm + i * 32 + l31];
          bi[i] = sgn * Bi[(kk + lk) * GLD + wn + i * 32 + l31];
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
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
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + wn + j * 32 + l31;
    if (col >= g.N) continue;
    const float b_r = g.bias_r ? g.bias_r[col] : 0.0f;
    const float b_i = (CPLX && g.bias_i) ? g.bias_i[col] : 0.0f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
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

// split-K plan of the generic kernel: few output tiles and a long K (small layers at large
// batch, heads with few outputs) would leave most CUs idle and expose the load latency serially
int gemm_generic_splits(int M, int N, int K) {
  const int64_t tiles = (int64_t)((M + GBM - 1) / GBM) * ((N + GBN - 1) / GBN);
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
                                                                  int64_t ldc, const float* bias,
                                                                  const float* emul, int emul_exp, int accumulate,
                                                                  const float* beta_p, TOUT* out) {
  const float beta = (accumulate && beta_p) ? *beta_p : 1.0f;
  const int64_t n = (int64_t)M * N, stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    float acc = 0.f;
    for (int s = 0; s < splits; ++s) acc += slabs[(int64_t)s * slab_stride + i];
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
  dim3 grid((g.N + GBN - 1) / GBN, (g.M + GBM - 1) / GBM, g.batch > 1 ? g.batch : g.splits);
  constexpr int smem = 2 * (CPLX ? 4 : 2) * GPLANE * (int)sizeof(float);
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
  if (in_dtype == CPLXAMD_F32 && out_dtype == CPLXAMD_F32) rc = go(gemm_generic_kernel<float, float, CPLX>);
  else if (in_dtype == CPLXAMD_BF16 && out_dtype == CPLXAMD_F32) rc = go(gemm_generic_kernel<bf16_t, float, CPLX>);
  else if (in_dtype == CPLXAMD_BF16 && out_dtype == CPLXAMD_BF16) rc = go(gemm_generic_kernel<bf16_t, bf16_t, CPLX>);
  else if (in_dtype == CPLXAMD_F32 && out_dtype == CPLXAMD_BF16) rc = go(gemm_generic_kernel<float, bf16_t, CPLX>);
  else return CPLXAMD_EINVAL;
  if (rc) return rc;
  if (g.splits > 1) {
    const int64_t slab = (int64_t)g.M * g.N, stride = (CPLX ? 2 : 1) * slab;
    const int rgrid = stream_grid(slab, 256);
    for (int pl = 0; pl < (CPLX ? 2 : 1); ++pl) {
      const float* src = (const float*)g.ws + pl * slab;
      const float* bias = pl ? g.bias_i : g.bias_r;
      void* out = pl ? g.c_i : g.c_r;
      if (out_dtype == CPLXAMD_F32)
        generic_slab_reduce_kernel<float><<<rgrid, 256, 0, st>>>(src, g.splits, stride, g.M, g.N, g.ldc, bias,
                                                                (pl && !g.emul_both) ? nullptr : g.emul, g.emul_exp, g.accumulate, g.beta, (float*)out);
      else
        generic_slab_reduce_kernel<bf16_t><<<rgrid, 256, 0, st>>>(src, g.splits, stride, g.M, g.N, g.ldc, bias,
                                                                 (pl && !g.emul_both) ? nullptr : g.emul, g.emul_exp, g.accumulate, g.beta, (bf16_t*)out);
      CPLXAMD_CHECK_LAUNCH();
    }
  }
  return 0;
}

template int launch_gemm_generic<true>(const GemmArgs&, int, int, hipStream_t);
template int launch_gemm_generic<false>(const GemmArgs&, int, int, hipStream_t);

}  // namespace cplxamd
