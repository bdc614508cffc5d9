// Generic complex / real GEMM on the exact-f32 matrix cores (v_mfma_f32_32x32x2_f32):
//   C[m,n] = sum_k A[m,k] * op(B[n,k]) (+ bias[n]),   any element strides, any M, N, K,
//   float32 or bf16 inputs (bf16 is widened exactly), float32 accumulation.
// This is the parity path (fp32 results == an fmaf chain in k order) and the fallback for
// shapes / layouts the bf16 fast path (gemm_bf16.hip) does not take.
//
// Reference arithmetic: cplx.linear_naive cplx.py:634-648, Cplx.__matmul__ cplx.py:167-174,
// F.linear in the LRT variance term nn/relevance/complex/base.py:50-54.
//
// Tiling: 64x64 outputs per 256-thread block (4 waves as 2x2, one 32x32 MFMA tile each),
// BK = 16, operands staged k-major in LDS so that every ds_read_b32 is conflict-free.
#include "gemm.h"

namespace cplxamd {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int GBM = 64, GBN = 64, GBK = 16, GLD = GBM + 1;

template <typename T>
__device__ __forceinline__ float ldg(const void* p, int64_t off) {
  return io<T>::ld(reinterpret_cast<const T*>(p) + off);
}

// One thread's 4 elements of a [64 rows x GBK] operand tile: global -> registers (fetch), then
// registers -> LDS as dst[k][row] (commit).  Splitting the two lets the loads of tile t+1 fly
// while the MFMAs of tile t run.
struct TileRegs { float v[4]; };

template <typename TIN>
__device__ __forceinline__ TileRegs fetch_tile(const void* src, int64_t rs, int64_t cs, int row0,
                                               int k0, int rows, int K) {
  const int t = threadIdx.x;
  TileRegs o;
  if (cs == 1 || rs != 1) {  // k fastest across threads
    const int k = t & 15, rb = t >> 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int gr = row0 + rb + 16 * j, gk = k0 + k;
      o.v[j] = (gr < rows && gk < K) ? ldg<TIN>(src, (int64_t)gr * rs + (int64_t)gk * cs) : 0.0f;
    }
  } else {  // rows fastest across threads
    const int r = t & 63, kb = t >> 6;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int gr = row0 + r, gk = k0 + kb + 4 * j;
      o.v[j] = (gr < rows && gk < K) ? ldg<TIN>(src, (int64_t)gr * rs + (int64_t)gk * cs) : 0.0f;
    }
  }
  return o;
}

__device__ __forceinline__ void commit_tile(float (*dst)[GLD], const TileRegs& o, int64_t rs,
                                            int64_t cs) {
  const int t = threadIdx.x;
  if (cs == 1 || rs != 1) {
    const int k = t & 15, rb = t >> 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) dst[k][rb + 16 * j] = o.v[j];
  } else {
    const int r = t & 63, kb = t >> 6;
#pragma unroll
    for (int j = 0; j < 4; ++j) dst[kb + 4 * j][r] = o.v[j];
  }
}

template <typename TIN, typename TOUT, bool CPLX>
__global__ __launch_bounds__(256) void gemm_generic_kernel(GemmArgs g) {
  // two LDS buffers: tile t+1 is committed while other waves may still read tile t
  __shared__ float As_r[2][GBK][GLD], Bs_r[2][GBK][GLD];
  __shared__ float As_i[CPLX ? 2 : 1][CPLX ? GBK : 1][GLD], Bs_i[CPLX ? 2 : 1][CPLX ? GBK : 1][GLD];

  const int m0 = blockIdx.y * GBM, n0 = blockIdx.x * GBN;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int wm = (wid >> 1) * 32, wn = (wid & 1) * 32;
  const int l31 = lane & 31, lk = lane >> 5;
  const float sgn = g.conj_b ? -1.0f : 1.0f;

  f32x16 acc_r = {0}, acc_i = {0};

  // split-K: this block covers K range [kb, ke)
  int kb = 0, ke = g.K;
  if (g.splits > 1) {
    kb = blockIdx.z * g.kchunk;
    ke = kb + g.kchunk < g.K ? kb + g.kchunk : g.K;
  }
  TileRegs ra, rb, rai, rbi;
  auto fetch = [&](int k0) {
    ra = fetch_tile<TIN>(g.a_r, g.a_rs, g.a_cs, m0, k0, g.M, ke);
    rb = fetch_tile<TIN>(g.b_r, g.b_rs, g.b_cs, n0, k0, g.N, ke);
    if (CPLX) {
      rai = fetch_tile<TIN>(g.a_i, g.a_rs, g.a_cs, m0, k0, g.M, ke);
      rbi = fetch_tile<TIN>(g.b_i, g.b_rs, g.b_cs, n0, k0, g.N, ke);
    }
  };
  if (kb < ke) fetch(kb);
  int buf = 0;
  for (int k0 = kb; k0 < ke; k0 += GBK, buf ^= 1) {
    commit_tile(As_r[buf], ra, g.a_rs, g.a_cs);
    commit_tile(Bs_r[buf], rb, g.b_rs, g.b_cs);
    if (CPLX) {
      commit_tile(As_i[buf], rai, g.a_rs, g.a_cs);
      commit_tile(Bs_i[buf], rbi, g.b_rs, g.b_cs);
    }
    __syncthreads();                       // tile visible; the other buffer is free again
    if (k0 + GBK < ke) fetch(k0 + GBK);     // in flight during the MFMAs below
#pragma unroll
    for (int kk = 0; kk < GBK; kk += 2) {
      const float ar = As_r[buf][kk + lk][wm + l31];
      const float br = Bs_r[buf][kk + lk][wn + l31];
      acc_r = __builtin_amdgcn_mfma_f32_32x32x2f32(ar, br, acc_r, 0, 0, 0);
      if (CPLX) {
        const float ai = As_i[buf][kk + lk][wm + l31];
        const float bi = sgn * Bs_i[buf][kk + lk][wn + l31];
        acc_r = __builtin_amdgcn_mfma_f32_32x32x2f32(-ai, bi, acc_r, 0, 0, 0);
        acc_i = __builtin_amdgcn_mfma_f32_32x32x2f32(ar, bi, acc_i, 0, 0, 0);
        acc_i = __builtin_amdgcn_mfma_f32_32x32x2f32(ai, br, acc_i, 0, 0, 0);
      }
    }
  }

  // C/D layout of the 32x32 tile: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  const int col = n0 + wn + l31;
  if (col >= g.N) return;
  const float b_r = g.bias_r ? g.bias_r[col] : 0.0f;
  const float b_i = (CPLX && g.bias_i) ? g.bias_i[col] : 0.0f;
  TOUT* cr = reinterpret_cast<TOUT*>(g.c_r);
  TOUT* ci = reinterpret_cast<TOUT*>(g.c_i);
  if (g.splits > 1) {   // fp32 partial slabs [split][plane][M][N]; bias / emul / accumulate: the reducer
    float* slab = reinterpret_cast<float*>(g.ws) + (int64_t)blockIdx.z * (CPLX ? 2 : 1) * g.M * g.N;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * lk;
      if (row >= g.M) continue;
      slab[(int64_t)row * g.N + col] = acc_r[r];
      if (CPLX) slab[(int64_t)g.M * g.N + (int64_t)row * g.N + col] = acc_i[r];
    }
    return;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * lk;
    if (row >= g.M) continue;
    const int64_t o = (int64_t)row * g.ldc + col;
    float vr = acc_r[r] + b_r;
    if (g.emul) vr *= g.emul[o];
    if (g.accumulate) vr += io<TOUT>::ld(cr + o);
    io<TOUT>::st(cr + o, vr);
    if (CPLX) {
      float vi = acc_i[r] + b_i;
      if (g.accumulate) vi += io<TOUT>::ld(ci + o);
      io<TOUT>::st(ci + o, vi);
    }
  }
}

// split-K plan of the generic kernel: few output tiles and a long K (small layers at large
// batch, heads with few outputs) would leave most CUs idle and expose the load latency serially
int gemm_generic_splits(int M, int N, int K) {
  const int64_t tiles = (int64_t)((M + GBM - 1) / GBM) * ((N + GBN - 1) / GBN);
  if (tiles >= 128 || K < 16 * GBK) return 1;
  int64_t s = 512 / tiles;
  const int64_t maxs = K / (4 * GBK);                 // >= 4 K tiles per split
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
                                                                  const float* emul, int accumulate,
                                                                  TOUT* out) {
  const int64_t n = (int64_t)M * N, stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    float acc = 0.f;
    for (int s = 0; s < splits; ++s) acc += slabs[(int64_t)s * slab_stride + i];
    const int row = (int)(i / N), col = (int)(i - (int64_t)row * N);
    const int64_t o = (int64_t)row * ldc + col;
    if (bias) acc += bias[col];
    if (emul) acc *= emul[o];
    if (accumulate) acc += io<TOUT>::ld(out + o);
    io<TOUT>::st(out + o, acc);
  }
}

template <bool CPLX>
int launch_gemm_generic(const GemmArgs& g0, int in_dtype, int out_dtype, hipStream_t st) {
  if (g0.M <= 0 || g0.N <= 0) return 0;
  GemmArgs g = g0;
  g.splits = 1;
  const int want = gemm_generic_splits(g.M, g.N, g.K);
  if (want > 1 && g.ws && g.ws_bytes >= gemm_generic_ws_bytes(g.M, g.N, g.K, CPLX)) {
    g.splits = want;
    g.kchunk = (((g.K + GBK - 1) / GBK + want - 1) / want) * GBK;   // whole K tiles, covers the tail
  }
  dim3 grid((g.N + GBN - 1) / GBN, (g.M + GBM - 1) / GBM, g.splits);
  if (in_dtype == CPLXAMD_F32 && out_dtype == CPLXAMD_F32)
    gemm_generic_kernel<float, float, CPLX><<<grid, 256, 0, st>>>(g);
  else if (in_dtype == CPLXAMD_BF16 && out_dtype == CPLXAMD_F32)
    gemm_generic_kernel<bf16_t, float, CPLX><<<grid, 256, 0, st>>>(g);
  else if (in_dtype == CPLXAMD_BF16 && out_dtype == CPLXAMD_BF16)
    gemm_generic_kernel<bf16_t, bf16_t, CPLX><<<grid, 256, 0, st>>>(g);
  else if (in_dtype == CPLXAMD_F32 && out_dtype == CPLXAMD_BF16)
    gemm_generic_kernel<float, bf16_t, CPLX><<<grid, 256, 0, st>>>(g);
  else
    return CPLXAMD_EINVAL;
  CPLXAMD_CHECK_LAUNCH();
  if (g.splits > 1) {
    const int64_t slab = (int64_t)g.M * g.N, stride = (CPLX ? 2 : 1) * slab;
    const int rgrid = stream_grid(slab, 256);
    for (int pl = 0; pl < (CPLX ? 2 : 1); ++pl) {
      const float* src = (const float*)g.ws + pl * slab;
      const float* bias = pl ? g.bias_i : g.bias_r;
      void* out = pl ? g.c_i : g.c_r;
      if (out_dtype == CPLXAMD_F32)
        generic_slab_reduce_kernel<float><<<rgrid, 256, 0, st>>>(src, g.splits, stride, g.M, g.N, g.ldc, bias,
                                                                pl ? nullptr : g.emul, g.accumulate, (float*)out);
      else
        generic_slab_reduce_kernel<bf16_t><<<rgrid, 256, 0, st>>>(src, g.splits, stride, g.M, g.N, g.ldc, bias,
                                                                 pl ? nullptr : g.emul, g.accumulate, (bf16_t*)out);
      CPLXAMD_CHECK_LAUNCH();
    }
  }
  return 0;
}

template int launch_gemm_generic<true>(const GemmArgs&, int, int, hipStream_t);
template int launch_gemm_generic<false>(const GemmArgs&, int, int, hipStream_t);

}  // namespace cplxamd
