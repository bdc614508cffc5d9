// K5: local-reparameterization noise injection  y = mu + eps * sqrt(max(s2, 1e-8))  and its
// backward, with the noise either supplied (parity mode) or generated in-kernel from a
// counter-based Philox4x32-7 (common.h: kPhiloxRounds) stream (so the backward regenerates it instead of storing it).
//
// Reference arithmetic (file:line under /root/reference/cplxmodule):
//   nn/relevance/complex/base.py:56, nn/relevance/real/base.py:49, cplx.py:544-562 (randn).
//
// HBM traffic (fp32, complex, Philox): read mu_r, mu_i, s2, write y_r, y_i = 20 B/output;
// backward: read g_r, g_i, s2, write g_s2 = 16 B/output.
//
// Noise stream ("DESIGN.md: noise stream"): group g = 4 normals from one Philox call with
// counter (g_lo, g_hi, offset_lo, offset_hi) and key (seed_lo, seed_hi); u_k = ((x_k >> 8) + 0.5)
// * 2^-24; (z0, z1) = BoxMuller(u0, u1), (z2, z3) = BoxMuller(u2, u3) with
// BoxMuller(a, b) = sqrt(-2 ln a) * (cos 2 pi b, sin 2 pi b).
//   real layers   : elements 4g .. 4g+3            = z0, z1, z2, z3
//   complex layers: element 2g   (eps_r, eps_i)    = (z0, z1) / sqrt 2
//                   element 2g+1 (eps_r, eps_i)    = (z2, z3) / sqrt 2
#include "common.h"

// The reference issues separate torch ops (one rounding each); keep the compiler from
// fusing a*b+c into fma so that parity-mode results are bit-identical.  Explicit fmaf()
// calls below are deliberate.
#pragma clang fp contract(off)

namespace cplxamd {

constexpr int kRpThreads = 256;

// A value every lane read from the same address, pinned to scalar registers: the compiler cannot prove that a loaded
// stream position is wave-uniform and would otherwise run Philox's key schedule (2 adds per round) on the vector ALU --
// 14 of ~70 instructions per call in kernels that PMC shows 93 % VALU-busy (profiles/r03_reparam_pmc.txt).
__device__ __forceinline__ uint64_t uniform64(uint64_t v) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
  return ((uint64_t)hi << 32) | lo;
}

struct u32x4 { uint32_t v[4]; };

__device__ __forceinline__ u32x4 philox4x32(uint64_t ctr_lo, uint64_t ctr_hi, uint64_t key) {
  uint32_t c0 = (uint32_t)ctr_lo, c1 = (uint32_t)(ctr_lo >> 32);
  uint32_t c2 = (uint32_t)ctr_hi, c3 = (uint32_t)(ctr_hi >> 32);
  uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
#pragma unroll
  for (int r = 0; r < kPhiloxRounds; ++r) {
    // one 32x32->64 product each (v_mad_u64_u32) instead of a mul_lo + mul_hi pair
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t lo0 = (uint32_t)p0, hi0 = (uint32_t)(p0 >> 32);
    const uint32_t lo1 = (uint32_t)p1, hi1 = (uint32_t)(p1 >> 32);
    // three-input XOR in one instruction (v_bitop3_b32, truth table 0x96)
    c0 = __builtin_amdgcn_bitop3_b32(hi1, c1, k0, 0x96);
    c1 = lo1;
    c2 = __builtin_amdgcn_bitop3_b32(hi0, c3, k1, 0x96);
    c3 = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return u32x4{{c0, c1, c2, c3}};
}

__device__ __forceinline__ float u01(uint32_t x) {
  // ((x >> 8) + 0.5) * 2^-24 as ONE fused multiply-add: bit-identical to the add-then-multiply of the stream definition
  // (below 2^23 both are exact; from 2^23 on the sum is a tie that either form rounds to the same even neighbour)
  return fmaf((float)(x >> 8), 5.9604644775390625e-08f, 2.98023223876953125e-08f);
}

// scale = 1 for real noise, 1/sqrt(2) for complex
__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float scale, float& z0,
                                           float& z1) {
  // scale * sqrt(-2 ln u) = sqrt((-2 ln 2 scale^2) log2 u): v_log_f32 is a base-2 logarithm, so the natural-log
  // conversion, the factor -2 and the scale are ONE multiplication under the square root
  const float r = __builtin_amdgcn_sqrtf((-1.3862943611198906f * scale * scale) * __builtin_amdgcn_logf(u01(a)));
  const float ang = u01(b);                  // in revolutions: v_sin/v_cos take x / 2 pi
  z0 = r * __builtin_amdgcn_cosf(ang);
  z1 = r * __builtin_amdgcn_sinf(ang);
}

__device__ __forceinline__ f4 philox_normals(uint64_t group, uint64_t seed, uint64_t offset,
                                             float scale) {
  const u32x4 x = philox4x32(group, offset, seed);
  f4 z;
  box_muller(x.v[0], x.v[1], scale, z.v[0], z.v[1]);
  box_muller(x.v[2], x.v[3], scale, z.v[2], z.v[3]);
  return z;
}

constexpr float kRsqrt2 = 0.70710678118654752f;

// noise for the 4 consecutive elements 4*i4 .. 4*i4+3
template <bool CPLX>
__device__ __forceinline__ void noise_vec(int64_t i4, uint64_t seed, uint64_t offset, f4& er,
                                          f4& ei) {
  if (CPLX) {
    const f4 a = philox_normals(2 * (uint64_t)i4, seed, offset, kRsqrt2);
    const f4 b = philox_normals(2 * (uint64_t)i4 + 1, seed, offset, kRsqrt2);
    er = f4{{a.v[0], a.v[2], b.v[0], b.v[2]}};
    ei = f4{{a.v[1], a.v[3], b.v[1], b.v[3]}};
  } else {
    er = philox_normals((uint64_t)i4, seed, offset, 1.0f);
  }
}

template <bool CPLX>
__device__ __forceinline__ void noise_one(int64_t e, uint64_t seed, uint64_t offset, float& er,
                                          float& ei) {
  if (CPLX) {
    const f4 a = philox_normals((uint64_t)e >> 1, seed, offset, kRsqrt2);
    er = (e & 1) ? a.v[2] : a.v[0];
    ei = (e & 1) ? a.v[3] : a.v[1];
  } else {
    const f4 a = philox_normals((uint64_t)e >> 2, seed, offset, 1.0f);
    er = a.v[e & 3];
    ei = 0.0f;
  }
}

__device__ __forceinline__ float lrt_std(float s2) { return rn_sqrt(fmaxf(s2, 1e-8f)); }

// bf16 activations: the result is rounded to 8 bits of mantissa anyway, so the 1-ulp hardware
// sqrt / rsq replace the correctly rounded sqrt and division (20 VALU ops per output less; the
// float32 path keeps the reference's exact arithmetic)
template <typename T> __device__ __forceinline__ float lrt_std_t(float s2) {
  if (sizeof(T) == 2) return __builtin_amdgcn_sqrtf(fmaxf(s2, 1e-8f));
  return lrt_std(s2);
}

template <typename T, typename TS, bool CPLX, bool PHILOX>
__global__ __launch_bounds__(kRpThreads) void reparam_fwd_kernel(
    const T* mu_r, const T* mu_i, const TS* s2, const T* eps_r, const T* eps_i, uint64_t seed,
    uint64_t offset, const uint64_t* state, T* y_r, T* y_i, int64_t n) {
  if (PHILOX && state) { seed = uniform64(state[0]); offset = uniform64(state[1]); }   // device-resident stream position
  // 8 outputs per thread and iteration: all loads first (one 16-B access per bf16 plane, two per
  // float32 plane), then 2 x 4 outputs; the tail (< 8 elements) is done one element per thread
  const int64_t n8 = n >> 3;
  const int64_t stride = (int64_t)gridDim.x * kRpThreads;
  for (int64_t i = (int64_t)blockIdx.x * kRpThreads + threadIdx.x; i < n8; i += stride) {
    const f8 s = ld8(s2 + 8 * i);
    const f8 mr = ld8(mu_r + 8 * i);
    f8 mi, er, ei, yr, yi;
    if (CPLX) mi = ld8(mu_i + 8 * i);
    if (!PHILOX) {
      er = ld8(eps_r + 8 * i);
      if (CPLX) ei = ld8(eps_i + 8 * i);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (PHILOX) noise_vec<CPLX>(2 * i + h, seed, offset, er.h[h], ei.h[h]);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float sd = lrt_std_t<T>(s.h[h].v[j]);
        yr.h[h].v[j] = mr.h[h].v[j] + er.h[h].v[j] * sd;
        if (CPLX) yi.h[h].v[j] = mi.h[h].v[j] + ei.h[h].v[j] * sd;
      }
    }
    st8(y_r + 8 * i, yr);
    if (CPLX) st8(y_i + 8 * i, yi);
  }
  if (blockIdx.x == 0) {
    const int64_t e = (n8 << 3) + threadIdx.x;
    if (e < n) {
      float er, ei = 0.0f;
      if (PHILOX) {
        noise_one<CPLX>(e, seed, offset, er, ei);
      } else {
        er = io<T>::ld(eps_r + e);
        if (CPLX) ei = io<T>::ld(eps_i + e);
      }
      const float sd = lrt_std_t<T>(io<TS>::ld(s2 + e));
      io<T>::st(y_r + e, io<T>::ld(mu_r + e) + er * sd);
      if (CPLX) io<T>::st(y_i + e, io<T>::ld(mu_i + e) + ei * sd);
    }
  }
}

__device__ __forceinline__ float lrt_gs2(float gs, float s2) {
  // torch.clamp passes the gradient AT the boundary
  return s2 >= 1e-8f ? (gs * 0.5f) / lrt_std(s2) : 0.0f;
}
template <typename T> __device__ __forceinline__ float lrt_gs2_t(float gs, float s2) {
  if (sizeof(T) == 2) return s2 >= 1e-8f ? (gs * 0.5f) * __builtin_amdgcn_rsqf(s2) : 0.0f;
  return lrt_gs2(gs, s2);
}

template <typename T, typename TG, typename TS, bool CPLX, bool PHILOX>
__global__ __launch_bounds__(kRpThreads) void reparam_bwd_kernel(
    const T* g_r, const T* g_i, const TS* s2, const T* eps_r, const T* eps_i, uint64_t seed,
    uint64_t offset, const uint64_t* state, TG* g_s2, int64_t n) {
  if (PHILOX && state) { seed = uniform64(state[0]); offset = uniform64(state[1]); }
  const int64_t n8 = n >> 3;
  const int64_t stride = (int64_t)gridDim.x * kRpThreads;
  for (int64_t i = (int64_t)blockIdx.x * kRpThreads + threadIdx.x; i < n8; i += stride) {
    const f8 s = ld8(s2 + 8 * i);
    const f8 gr = ld8(g_r + 8 * i);
    f8 gi, er, ei, o;
    if (CPLX) gi = ld8(g_i + 8 * i);
    if (!PHILOX) {
      er = ld8(eps_r + 8 * i);
      if (CPLX) ei = ld8(eps_i + 8 * i);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (PHILOX) noise_vec<CPLX>(2 * i + h, seed, offset, er.h[h], ei.h[h]);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float gs = gr.h[h].v[j] * er.h[h].v[j];
        if (CPLX) gs = gs + gi.h[h].v[j] * ei.h[h].v[j];
        o.h[h].v[j] = lrt_gs2_t<T>(gs, s.h[h].v[j]);
      }
    }
    st8(g_s2 + 8 * i, o);
  }
  if (blockIdx.x == 0) {
    const int64_t e = (n8 << 3) + threadIdx.x;
    if (e < n) {
      float er, ei = 0.0f;
      if (PHILOX) {
        noise_one<CPLX>(e, seed, offset, er, ei);
      } else {
        er = io<T>::ld(eps_r + e);
        if (CPLX) ei = io<T>::ld(eps_i + e);
      }
      float gs = io<T>::ld(g_r + e) * er;
      if (CPLX) gs = gs + io<T>::ld(g_i + e) * ei;
      io<TG>::st(g_s2 + e, lrt_gs2_t<T>(gs, io<TS>::ld(s2 + e)));
    }
  }
}

// The same backward on a [rows][cols] matrix (the storage of a [B, O] gradient, or of channels-last [B H W][C]
// planes) that ALSO sums g_r / g_i per column -- the bias gradient of the layer (dbr = sum_b G_r), which
// otherwise costs one more pass over the two gradient planes.  A thread owns 8 consecutive columns; TX = min(cols / 8,
// 256) threads span a row (strip of 2048 columns), TY = 256 / TX rows per step; the grid is (strips, row chunks).  The
// element -> Philox counter mapping is that of the flat kernel (linear index), so both draw identical noise.
// partial: [chunks][planes][cols] float32, summed in fixed order by reparam_cols_final (deterministic).
template <typename T, typename TG, typename TS, bool CPLX, bool PHILOX>
__global__ __launch_bounds__(kRpThreads) void reparam_bwd_cols_kernel(
    const T* g_r, const T* g_i, const TS* s2, const T* eps_r, const T* eps_i, uint64_t seed,
    uint64_t offset, const uint64_t* state, TG* g_s2, int64_t rows, int cols, int64_t rows_per_chunk,
    float* partial) {
  __shared__ float red[kRpThreads * 16];
  if (PHILOX && state) { seed = uniform64(state[0]); offset = uniform64(state[1]); }
  const int CG = cols >> 3;
  const int TX = CG < kRpThreads ? CG : kRpThreads, TY = kRpThreads / TX;
  const int tx = (int)threadIdx.x % TX, ty = (int)threadIdx.x / TX;
  const int col = (int)blockIdx.x * (kRpThreads * 8) + tx * 8;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_chunk;
  const int64_t r1 = r0 + rows_per_chunk < rows ? r0 + rows_per_chunk : rows;
  constexpr int NP = CPLX ? 2 : 1;
  float acc[NP][8];
#pragma unroll
  for (int p = 0; p < NP; ++p)
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[p][c] = 0.0f;
  for (int64_t r = r0 + ty; r < r1; r += TY) {
    const int64_t i = (r * cols + col) >> 3;
    const f8 s = ld8(s2 + 8 * i);
    const f8 gr = ld8(g_r + 8 * i);
    f8 gi, er, ei, o;
    if (CPLX) gi = ld8(g_i + 8 * i);
    if (!PHILOX) {
      er = ld8(eps_r + 8 * i);
      if (CPLX) ei = ld8(eps_i + 8 * i);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (PHILOX) noise_vec<CPLX>(2 * i + h, seed, offset, er.h[h], ei.h[h]);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float gs = gr.h[h].v[j] * er.h[h].v[j];
        if (CPLX) gs = gs + gi.h[h].v[j] * ei.h[h].v[j];
        o.h[h].v[j] = lrt_gs2_t<T>(gs, s.h[h].v[j]);
        acc[0][4 * h + j] += gr.h[h].v[j];
        if (CPLX) acc[NP - 1][4 * h + j] += gi.h[h].v[j];
      }
    }
    st8(g_s2 + 8 * i, o);
  }
  float* dst = partial + (int64_t)blockIdx.y * NP * cols;
  if (TY == 1) {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      st4(dst + (int64_t)p * cols + col, f4{{acc[p][0], acc[p][1], acc[p][2], acc[p][3]}});
      st4(dst + (int64_t)p * cols + col + 4, f4{{acc[p][4], acc[p][5], acc[p][6], acc[p][7]}});
    }
    return;
  }
#pragma unroll
  for (int p = 0; p < NP; ++p)
#pragma unroll
    for (int c = 0; c < 8; ++c) red[((ty * NP + p) * TX + tx) * 8 + c] = acc[p][c];
  __syncthreads();
  // (TY > 1 means one strip: cols = TX * 8)
  for (int f = (int)threadIdx.x; f < NP * cols; f += kRpThreads) {
    const int p = f / cols, c = f % cols;
    float t = 0.0f;
    for (int l = 0; l < TY; ++l) t += red[((l * NP + p) * TX) * 8 + c];
    dst[(int64_t)p * cols + c] = t;
  }
}

// out_r[c] / out_i[c] = sum over chunks of partial[chunk][plane][c]: 16 columns per block, 64 chunk lanes, fixed order
__global__ __launch_bounds__(1024) void reparam_cols_final(const float* partial, int chunks, int cols, int planes,
                                                           float* out_r, float* out_i) {
  __shared__ float red[64][17];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + tx, n = planes * cols;
  float acc = 0.f;
  if (c < n)
    for (int j = ty; j < chunks; j += 64) acc += partial[(int64_t)j * n + c];
  red[ty][tx] = acc;
  __syncthreads();
  if (ty == 0 && c < n) {
    float t = 0.f;
#pragma unroll 8
    for (int l = 0; l < 64; ++l) t += red[l][tx];
    if (c < cols) out_r[c] = t;
    else out_i[c - cols] = t;
  }
}

struct ColsPlan { int strips, chunks; int64_t rows_per_chunk; };
static bool cols_plan(int64_t rows, int cols, ColsPlan& p) {
  if (rows <= 0 || cols < 8 || cols % 8) return false;
  const int cg = cols / 8;
  if (cg < kRpThreads ? (kRpThreads % cg != 0) : (cols % (kRpThreads * 8) != 0)) return false;
  if (rows * (int64_t)cols >= ((int64_t)1 << 40)) return false;
  p.strips = cg < kRpThreads ? 1 : cols / (kRpThreads * 8);
  const int ty = cg < kRpThreads ? kRpThreads / cg : 1;
  // >= 16 row steps per thread, at most 2048 workgroups
  int64_t chunks = rows / (16 * (int64_t)ty);
  const int64_t cap = 2048 / p.strips > 0 ? 2048 / p.strips : 1;
  if (chunks > cap) chunks = cap;
  if (chunks < 1) chunks = 1;
  int64_t rpc = (rows + chunks - 1) / chunks;
  rpc = (rpc + ty - 1) / ty * ty;
  p.rows_per_chunk = rpc;
  p.chunks = (int)((rows + rpc - 1) / rpc);
  return p.chunks <= 65535;
}

template <typename T, typename TG, typename TS>
static int launch_bwd_cols(const void* g_r, const void* g_i, const void* s2, const void* eps_r,
                           const void* eps_i, uint64_t seed, uint64_t offset, const uint64_t* state,
                           void* g_s2, int64_t rows, int cols, const ColsPlan& p, float* partial, hipStream_t st) {
  const dim3 grid((unsigned)p.strips, (unsigned)p.chunks);
  const bool cplx = g_i != nullptr, philox = eps_r == nullptr;
#define RP_BWDC(C, P)                                                                          \
  reparam_bwd_cols_kernel<T, TG, TS, C, P><<<grid, kRpThreads, 0, st>>>(                       \
      (const T*)g_r, (const T*)g_i, (const TS*)s2, (const T*)eps_r, (const T*)eps_i, seed, offset, \
      state, (TG*)g_s2, rows, cols, p.rows_per_chunk, partial)
  if (cplx && philox) RP_BWDC(true, true);
  else if (cplx) RP_BWDC(true, false);
  else if (philox) RP_BWDC(false, true);
  else RP_BWDC(false, false);
#undef RP_BWDC
  CPLXAMD_CHECK_LAUNCH();
  return 0;
}

template <bool CPLX>
__global__ __launch_bounds__(kRpThreads) void philox_normal_kernel(float* er, float* ei,
                                                                   uint64_t seed, uint64_t offset,
                                                                   int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * kRpThreads;
  for (int64_t e = (int64_t)blockIdx.x * kRpThreads + threadIdx.x; e < n; e += stride) {
    float a, b;
    noise_one<CPLX>(e, seed, offset, a, b);
    er[e] = a;
    if (CPLX) ei[e] = b;
  }
}

template <typename T, typename TS>
static int launch_fwd(const void* mu_r, const void* mu_i, const void* s2, const void* eps_r,
                      const void* eps_i, uint64_t seed, uint64_t offset, const uint64_t* state,
                      void* y_r, void* y_i, int64_t n, hipStream_t st) {
  const int grid = stream_grid(n >> 3, kRpThreads);
  const bool cplx = mu_i != nullptr, philox = eps_r == nullptr;
#define RP_FWD(C, P)                                                                       \
  reparam_fwd_kernel<T, TS, C, P><<<grid, kRpThreads, 0, st>>>(                            \
      (const T*)mu_r, (const T*)mu_i, (const TS*)s2, (const T*)eps_r, (const T*)eps_i, seed, offset,  \
      state, (T*)y_r, (T*)y_i, n)
  if (cplx && philox) RP_FWD(true, true);
  else if (cplx) RP_FWD(true, false);
  else if (philox) RP_FWD(false, true);
  else RP_FWD(false, false);
#undef RP_FWD
  CPLXAMD_CHECK_LAUNCH();
  return 0;
}

template <typename T, typename TG, typename TS>
static int launch_bwd(const void* g_r, const void* g_i, const void* s2, const void* eps_r,
                      const void* eps_i, uint64_t seed, uint64_t offset, const uint64_t* state,
                      void* g_s2, int64_t n, hipStream_t st) {
  const int grid = stream_grid(n >> 3, kRpThreads);
  const bool cplx = g_i != nullptr, philox = eps_r == nullptr;
#define RP_BWD(C, P)                                                                      \
  reparam_bwd_kernel<T, TG, TS, C, P><<<grid, kRpThreads, 0, st>>>(                       \
      (const T*)g_r, (const T*)g_i, (const TS*)s2, (const T*)eps_r, (const T*)eps_i, seed, offset,   \
      state, (TG*)g_s2, n)
  if (cplx && philox) RP_BWD(true, true);
  else if (cplx) RP_BWD(true, false);
  else if (philox) RP_BWD(false, true);
  else RP_BWD(false, false);
#undef RP_BWD
  CPLXAMD_CHECK_LAUNCH();
  return 0;
}

// used[0..1] = state[0..1]; state[1] += 1   (one stochastic forward consumes one offset)
__global__ void philox_advance_kernel(uint64_t* state, uint64_t* used) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    used[0] = state[0];
    used[1] = state[1];
    state[1] = state[1] + 1;
  }
}

}  // namespace cplxamd

using namespace cplxamd;

static bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

extern "C" {

int cplxamd_lrt_reparam_fwd_ex(const void* mu_r, const void* mu_i, const void* s2,
                               const void* eps_r, const void* eps_i, uint64_t seed,
                               uint64_t offset, const uint64_t* state, void* y_r, void* y_i,
                               int64_t n, int dtype, int s2_dtype, void* stream) {
  if (!mu_r || !s2 || !y_r || n < 0) return CPLXAMD_EINVAL;
  if (!al16(mu_r) || !al16(mu_i) || !al16(s2) || !al16(eps_r) || !al16(eps_i) || !al16(y_r) || !al16(y_i))
    return CPLXAMD_EALIGN;
  if ((mu_i == nullptr) != (y_i == nullptr)) return CPLXAMD_EINVAL;
  if (eps_r && mu_i && !eps_i) return CPLXAMD_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CPLXAMD_F32 && s2_dtype == CPLXAMD_F32)
    return launch_fwd<float, float>(mu_r, mu_i, s2, eps_r, eps_i, seed, offset, state, y_r, y_i, n, st);
  if (dtype == CPLXAMD_BF16 && s2_dtype == CPLXAMD_F32)
    return launch_fwd<bf16_t, float>(mu_r, mu_i, s2, eps_r, eps_i, seed, offset, state, y_r, y_i, n, st);
  if (dtype == CPLXAMD_BF16 && s2_dtype == CPLXAMD_BF16)
    return launch_fwd<bf16_t, bf16_t>(mu_r, mu_i, s2, eps_r, eps_i, seed, offset, state, y_r, y_i, n, st);
  return CPLXAMD_EINVAL;
}

int cplxamd_lrt_reparam_fwd(const void* mu_r, const void* mu_i, const float* s2,
                            const void* eps_r, const void* eps_i, uint64_t seed,
                            uint64_t offset, const uint64_t* state, void* y_r, void* y_i,
                            int64_t n, int dtype, void* stream) {
  return cplxamd_lrt_reparam_fwd_ex(mu_r, mu_i, s2, eps_r, eps_i, seed, offset, state, y_r, y_i, n, dtype,
                                    CPLXAMD_F32, stream);
}

int cplxamd_lrt_reparam_bwd_ex(const void* g_r, const void* g_i, const void* s2,
                               const void* eps_r, const void* eps_i, uint64_t seed,
                               uint64_t offset, const uint64_t* state, void* g_s2, int64_t n,
                               int dtype, int gs2_dtype, int s2_dtype, void* stream) {
  if (!g_r || !s2 || !g_s2 || n < 0) return CPLXAMD_EINVAL;
  if (eps_r && g_i && !eps_i) return CPLXAMD_EINVAL;
  if (!al16(g_r) || !al16(g_i) || !al16(s2) || !al16(eps_r) || !al16(eps_i) || !al16(g_s2))
    return CPLXAMD_EALIGN;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CPLXAMD_F32 && gs2_dtype == CPLXAMD_F32 && s2_dtype == CPLXAMD_F32)
    return launch_bwd<float, float, float>(g_r, g_i, s2, eps_r, eps_i, seed, offset, state, g_s2, n, st);
  if (dtype == CPLXAMD_BF16 && gs2_dtype == CPLXAMD_F32 && s2_dtype == CPLXAMD_F32)
    return launch_bwd<bf16_t, float, float>(g_r, g_i, s2, eps_r, eps_i, seed, offset, state, g_s2, n, st);
  if (dtype == CPLXAMD_BF16 && gs2_dtype == CPLXAMD_BF16 && s2_dtype == CPLXAMD_F32)
    return launch_bwd<bf16_t, bf16_t, float>(g_r, g_i, s2, eps_r, eps_i, seed, offset, state, g_s2, n, st);
  if (dtype == CPLXAMD_BF16 && gs2_dtype == CPLXAMD_BF16 && s2_dtype == CPLXAMD_BF16)
    return launch_bwd<bf16_t, bf16_t, bf16_t>(g_r, g_i, s2, eps_r, eps_i, seed, offset, state, g_s2, n, st);
  if (dtype == CPLXAMD_BF16 && gs2_dtype == CPLXAMD_F32 && s2_dtype == CPLXAMD_BF16)
    return launch_bwd<bf16_t, float, bf16_t>(g_r, g_i, s2, eps_r, eps_i, seed, offset, state, g_s2, n, st);
  return CPLXAMD_EINVAL;
}

int cplxamd_lrt_reparam_bwd(const void* g_r, const void* g_i, const float* s2,
                            const void* eps_r, const void* eps_i, uint64_t seed,
                            uint64_t offset, const uint64_t* state, void* g_s2, int64_t n,
                            int dtype, int gs2_dtype, void* stream) {
  return cplxamd_lrt_reparam_bwd_ex(g_r, g_i, s2, eps_r, eps_i, seed, offset, state, g_s2, n, dtype, gs2_dtype,
                                    CPLXAMD_F32, stream);
}

int64_t cplxamd_lrt_reparam_bwd_cols_ws_bytes(int64_t rows, int cols) {
  ColsPlan p;
  if (!cols_plan(rows, cols, p)) return 0;
  return (int64_t)p.chunks * 2 * cols * (int64_t)sizeof(float);
}

int cplxamd_lrt_reparam_bwd_cols(const void* g_r, const void* g_i, const void* s2,
                                 const void* eps_r, const void* eps_i, uint64_t seed,
                                 uint64_t offset, const uint64_t* state, void* g_s2, int64_t rows,
                                 int cols, int dtype, int gs2_dtype, int s2_dtype, float* sum_r,
                                 float* sum_i, void* ws, int64_t ws_bytes, void* stream) {
  if (!g_r || !s2 || !g_s2 || !sum_r || !ws || rows < 0) return CPLXAMD_EINVAL;
  if ((g_i == nullptr) != (sum_i == nullptr)) return CPLXAMD_EINVAL;
  if (eps_r && g_i && !eps_i) return CPLXAMD_EINVAL;
  ColsPlan p;
  if (!cols_plan(rows, cols, p)) return CPLXAMD_ESHAPE;
  if (!al16(g_r) || !al16(g_i) || !al16(s2) || !al16(eps_r) || !al16(eps_i) || !al16(g_s2) || !al16(ws))
    return CPLXAMD_EALIGN;
  if (ws_bytes < cplxamd_lrt_reparam_bwd_cols_ws_bytes(rows, cols)) return CPLXAMD_EWS;
  hipStream_t st = (hipStream_t)stream;
  float* partial = (float*)ws;
  int rc = CPLXAMD_EINVAL;
#define RP_GO(T, TG, TS) \
  rc = launch_bwd_cols<T, TG, TS>(g_r, g_i, s2, eps_r, eps_i, seed, offset, state, g_s2, rows, cols, p, partial, st)
  if (dtype == CPLXAMD_F32 && gs2_dtype == CPLXAMD_F32 && s2_dtype == CPLXAMD_F32) RP_GO(float, float, float);
  else if (dtype == CPLXAMD_BF16 && gs2_dtype == CPLXAMD_F32 && s2_dtype == CPLXAMD_F32) RP_GO(bf16_t, float, float);
  else if (dtype == CPLXAMD_BF16 && gs2_dtype == CPLXAMD_BF16 && s2_dtype == CPLXAMD_F32) RP_GO(bf16_t, bf16_t, float);
  else if (dtype == CPLXAMD_BF16 && gs2_dtype == CPLXAMD_BF16 && s2_dtype == CPLXAMD_BF16) RP_GO(bf16_t, bf16_t, bf16_t);
  else if (dtype == CPLXAMD_BF16 && gs2_dtype == CPLXAMD_F32 && s2_dtype == CPLXAMD_BF16) RP_GO(bf16_t, float, bf16_t);
#undef RP_GO
  if (rc) return rc;
  const int planes = g_i ? 2 : 1;
  reparam_cols_final<<<(planes * cols + 15) / 16, 1024, 0, st>>>(partial, p.chunks, cols, planes, sum_r, sum_i);
  CPLXAMD_CHECK_LAUNCH();
  return 0;
}

int cplxamd_philox_advance(uint64_t* state, uint64_t* used, void* stream) {
  if (!state || !used) return CPLXAMD_EINVAL;
  philox_advance_kernel<<<1, 64, 0, (hipStream_t)stream>>>(state, used);
  CPLXAMD_CHECK_LAUNCH();
  return 0;
}

int cplxamd_philox_normal(float* eps_r, float* eps_i, uint64_t seed, uint64_t offset,
                          int64_t n, void* stream) {
  if (!eps_r || n < 0) return CPLXAMD_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int grid = stream_grid(n, kRpThreads);
  if (eps_i)
    philox_normal_kernel<true><<<grid, kRpThreads, 0, st>>>(eps_r, eps_i, seed, offset, n);
  else
    philox_normal_kernel<false><<<grid, kRpThreads, 0, st>>>(eps_r, eps_i, seed, offset, n);
  CPLXAMD_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
