// Operand preparation of the float32-accurate products on the bf16 matrix pipe ("x3" mode).
//
// A float32 value is the exact sum of three bf16 values: x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1)
// (both differences are exact in float32; 3 x 8 significand bits cover float32's 24).  A product x w then needs the six
// terms x0 w0, x0 w1, x1 w0, x0 w2, x2 w0, x1 w1 -- the three dropped ones are below 2^-24 |x w| -- each of which the bf16
// MFMA forms exactly and accumulates in float32: float32-level results at 1/6 of the bf16 rate (417 TFLOP/s bound) instead
// of the 157 TFLOP/s of the float32 MFMA.  This file writes the pieces in the layouts that let the EXISTING bf16 GEMM
// kernels (gemm_bf16_w4.hip, gemm_bf16_impl.h) run the six terms as three K-concatenated launches:
//     A side  [x2 | x1 | x0]            (suffixes of it are the A operands)
//     B side  [w2 | w1 w1 | w0 w0 w0]   (the matching B operands, replicated along K)
//     launch 1: x0 . w2      launch 2: [x1|x0] . [w1|w1]      launch 3: [x2|x1|x0] . [w0|w0|w0]     (smallest terms first)
// Replaces the arithmetic of cplx.linear / the LRT variance products in float32: cplxmodule/cplx.py:641-646,
// nn/relevance/complex/base.py:43-56, real/base.py:43-49.
#include "common.h"

// |x|^2 = xr * xr + xi * xi with one rounding per operation, as the reference's separate torch ops (and cplxamd_abs2)
#pragma clang fp contract(off)

namespace cplxamd {

template <int OP>
__device__ __forceinline__ float split_value(float a, float b) {
  if constexpr (OP == 1) return a * a + b * b;          // |x|^2 of two planes (b = 0: x^2)
  else if constexpr (OP == 2) return expf(a);           // exp(log_sigma2)
  else return a;
}

// 8 consecutive elements of one row per thread: two 16-B loads (per plane), one 16-B store per piece.  The piece
// pattern is a template argument (a run-time table would put the three piece vectors into scratch memory); IDX = the
// type of the flat item index (32-bit wherever rows * cols / 8 fits: the row / column split is an integer division).
template <int OP, int PATTERN, typename IDX>
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ src, const float* __restrict__ src2,
                                                     int64_t ld_src, bf16_t* __restrict__ dst, int64_t ld_dst,
                                                     int64_t piece_stride, int64_t rows, int cols8) {
  const IDX items = (IDX)(rows * cols8);
  const IDX stride = (IDX)gridDim.x * 256;
  for (IDX it = (IDX)blockIdx.x * 256 + threadIdx.x; it < items; it += stride) {
    const IDX r = it / (IDX)cols8;
    const int c = (int)(it - r * (IDX)cols8) * 8;
    const f8 v = ld8(src + (int64_t)r * ld_src + c);
    f8 u;
    if constexpr (OP == 1) {
      if (src2) u = ld8(src2 + (int64_t)r * ld_src + c);
      else u = f8{{f4{{0.f, 0.f, 0.f, 0.f}}, f4{{0.f, 0.f, 0.f, 0.f}}}};
    }
    f8 t0, t1, t2;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float b = 0.f;
        if constexpr (OP == 1) b = u.h[h].v[e];
        const float x = split_value<OP>(v.h[h].v[e], b);
        const float x0 = bf16_to_f32(f32_to_bf16(x));
        // inf / nan: the leading piece carries it alone (inf - inf would turn an infinity into a nan)
        const bool fin = fabsf(x) <= 3.3895313892515355e38f;     // largest finite bf16
        const float r1 = fin ? x - x0 : 0.f;                     // exact
        const float x1 = bf16_to_f32(f32_to_bf16(r1));
        t0.h[h].v[e] = x0; t1.h[h].v[e] = x1; t2.h[h].v[e] = r1 - x1;   // (exact; rounded to bf16 by the store)
      }
    }
    bf16_t* o = dst + (int64_t)r * ld_dst + c;
    if constexpr (PATTERN == CPLXAMD_SPLIT_A) {         // (x2, x1, x0)
      st8(o, t2); st8(o + piece_stride, t1); st8(o + 2 * piece_stride, t0);
    } else {                                            // (w2, w1, w1, w0, w0, w0)
      st8(o, t2); st8(o + piece_stride, t1); st8(o + 2 * piece_stride, t1);
      st8(o + 3 * piece_stride, t0); st8(o + 4 * piece_stride, t0); st8(o + 5 * piece_stride, t0);
    }
  }
}

template <int OP, int PATTERN>
static int launch_split3(const float* src, const float* src2, int64_t ld_src, bf16_t* dst, int64_t ld_dst, int64_t piece_stride,
                  int64_t rows, int cols8, hipStream_t st) {
  const int64_t items = rows * cols8;
  const int grid = stream_grid(items, 256);
  if (items + (int64_t)grid * 256 < (int64_t)0x7fffffff)
    split3_kernel<OP, PATTERN, uint32_t><<<grid, 256, 0, st>>>(src, src2, ld_src, dst, ld_dst, piece_stride, rows, cols8);
  else
    split3_kernel<OP, PATTERN, int64_t><<<grid, 256, 0, st>>>(src, src2, ld_src, dst, ld_dst, piece_stride, rows, cols8);
  CPLXAMD_CHECK_LAUNCH();
  return 0;
}

// ---- half-precision two-term pieces ('x2' mode) -------------------------------------------------------------------------
// x s = h0 + h1 (+ a remainder below 2^-22 |x s|) with h0 = half(x s), h1 = half(x s - h0); s a power of two that puts the
// tensor's largest magnitude into [2^14, 2^15) so that the 5-bit exponent of IEEE half loses nothing that matters norm-wise:
// an element keeps all 22 bits while |x| >= 2^-17 max|x|, below that its second piece goes subnormal (absolute error
// <= 2^-25 in scaled units = 2^-39 max|x|).  Three piece products (h0 w0, h0 w1, h1 w0) instead of bf16's six.
// Pieces: pattern A (h1, h0); pattern B (w1, w0, w0):   launch 1: h0 . w1    launch 2: [h1|h0] . [w0|w0].

// stage 1 of max |op(x)|: per-block maxima (NaN / inf propagate as inf: the final stage then falls back to scale 1)
template <int OP>
__global__ __launch_bounds__(256) void absmax_partial_kernel(const float* __restrict__ src, const float* __restrict__ src2,
                                                             int64_t ld_src, int64_t rows, int cols4, float* partial) {
  __shared__ float red[4];
  const int64_t items = rows * cols4;
  const int64_t stride = (int64_t)gridDim.x * 256;
  float m = 0.f;
  for (int64_t it = (int64_t)blockIdx.x * 256 + threadIdx.x; it < items; it += stride) {
    const int64_t r = it / cols4;
    const int c = (int)(it - r * cols4) * 4;
    const f4 v = ld4(src + r * ld_src + c);
    f4 u = {{0.f, 0.f, 0.f, 0.f}};
    if constexpr (OP == 1 || OP == 3) { if (src2) u = ld4(src2 + r * ld_src + c); }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      // OP 3: the larger magnitude of two planes (the planes of a complex operand share one scale)
      const float a = OP == 3 ? fmaxf(fabsf(v.v[e]), fabsf(u.v[e])) + (v.v[e] != v.v[e] || u.v[e] != u.v[e] ? NAN : 0.f)
                              : fabsf(split_value<(OP == 3 ? 0 : OP)>(v.v[e], u.v[e]));
      m = (a > m || a != a) ? (a != a ? INFINITY : a) : m;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// stage 2: scale[0] = s = 2^(15 - e) with max < 2^e, scale[1] = 1 / s; s = 1 for an all-zero or non-finite tensor
__global__ __launch_bounds__(256) void absmax_final_kernel(const float* partial, int n, float* scale) {
  __shared__ float red[4];
  float m = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) m = fmaxf(m, partial[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float s = 1.0f, inv = 1.0f;
    if (m > 0.f && m <= 3.4028234e38f) {
      int e;
      (void)frexpf(m, &e);                       // m = f 2^e, f in [0.5, 1)
      int k = 15 - e;
      k = k > 120 ? 120 : (k < -120 ? -120 : k);
      s = ldexpf(1.0f, k);
      inv = ldexpf(1.0f, -k);
    }
    scale[0] = s;
    scale[1] = inv;
  }
}

template <int OP, int PATTERN, typename IDX>
__global__ __launch_bounds__(256) void split2h_kernel(const float* __restrict__ src, const float* __restrict__ src2,
                                                      int64_t ld_src, uint16_t* __restrict__ dst, int64_t ld_dst,
                                                      int64_t piece_stride, int64_t rows, int cols8, const float* scale) {
  const float sc = scale[0];
  const IDX items = (IDX)(rows * cols8);
  const IDX stride = (IDX)gridDim.x * 256;
  for (IDX it = (IDX)blockIdx.x * 256 + threadIdx.x; it < items; it += stride) {
    const IDX r = it / (IDX)cols8;
    const int c = (int)(it - r * (IDX)cols8) * 8;
    const f8 v = ld8(src + (int64_t)r * ld_src + c);
    f8 u;
    if constexpr (OP == 1) {
      if (src2) u = ld8(src2 + (int64_t)r * ld_src + c);
      else u = f8{{f4{{0.f, 0.f, 0.f, 0.f}}, f4{{0.f, 0.f, 0.f, 0.f}}}};
    }
    uint32_t h0[4], h1[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      uint16_t a0[2], a1[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int h = (2 * k + q) >> 2, e = (2 * k + q) & 3;
        float b = 0.f;
        if constexpr (OP == 1) b = u.h[h].v[e];
        const float x = split_value<OP>(v.h[h].v[e], b) * sc;
        const _Float16 p0 = (_Float16)x;                         // round to nearest even
        const bool fin = fabsf(x) <= 65504.0f;
        const float r1 = fin ? x - (float)p0 : 0.f;              // exact
        const _Float16 p1 = (_Float16)r1;
        a0[q] = __builtin_bit_cast(uint16_t, p0);
        a1[q] = __builtin_bit_cast(uint16_t, p1);
      }
      h0[k] = (uint32_t)a0[0] | ((uint32_t)a0[1] << 16);
      h1[k] = (uint32_t)a1[0] | ((uint32_t)a1[1] << 16);
    }
    const uint4 v0 = uint4{h0[0], h0[1], h0[2], h0[3]}, v1 = uint4{h1[0], h1[1], h1[2], h1[3]};
    uint16_t* o = dst + (int64_t)r * ld_dst + c;
    *reinterpret_cast<uint4*>(o) = v1;
    *reinterpret_cast<uint4*>(o + piece_stride) = v0;
    if constexpr (PATTERN == CPLXAMD_SPLIT_B) *reinterpret_cast<uint4*>(o + 2 * piece_stride) = v0;
  }
}

template <int OP, int PATTERN>
static int launch_split2h(const float* src, const float* src2, int64_t ld_src, uint16_t* dst, int64_t ld_dst,
                          int64_t piece_stride, int64_t rows, int cols8, const float* scale, hipStream_t st) {
  const int64_t items = rows * cols8;
  const int grid = stream_grid(items, 256);
  if (items + (int64_t)grid * 256 < (int64_t)0x7fffffff)
    split2h_kernel<OP, PATTERN, uint32_t><<<grid, 256, 0, st>>>(src, src2, ld_src, dst, ld_dst, piece_stride, rows, cols8, scale);
  else
    split2h_kernel<OP, PATTERN, int64_t><<<grid, 256, 0, st>>>(src, src2, ld_src, dst, ld_dst, piece_stride, rows, cols8, scale);
  CPLXAMD_CHECK_LAUNCH();
  return 0;
}

}  // namespace cplxamd
using namespace cplxamd;

extern "C" int cplxamd_split3(const float* src, const float* src2, int64_t ld_src, void* dst, int64_t ld_dst,
                              int64_t piece_stride, int64_t rows, int cols, int op, int pattern, void* stream) {
  if (!src || !dst || rows < 0 || cols < 0 || ld_src < cols || ld_dst < cols) return CPLXAMD_EINVAL;
  if (op < 0 || op > 2 || (src2 && op != 1)) return CPLXAMD_EINVAL;
  if (pattern != CPLXAMD_SPLIT_A && pattern != CPLXAMD_SPLIT_B) return CPLXAMD_EINVAL;
  if ((cols & 7) || (ld_src & 3) || (ld_dst & 7) || (piece_stride & 7)) return CPLXAMD_ESHAPE;
  if ((reinterpret_cast<uintptr_t>(src) & 15) || (reinterpret_cast<uintptr_t>(dst) & 15) ||
      (src2 && (reinterpret_cast<uintptr_t>(src2) & 15)))
    return CPLXAMD_EALIGN;
  if (rows == 0 || cols == 0) return 0;
  const int cols8 = cols / 8;
  hipStream_t st = (hipStream_t)stream;
  bf16_t* d = (bf16_t*)dst;
#define SPLIT3(OP)                                                                                                     \
  (pattern == CPLXAMD_SPLIT_A ? launch_split3<OP, CPLXAMD_SPLIT_A>(src, src2, ld_src, d, ld_dst, piece_stride, rows, cols8, st) \
                              : launch_split3<OP, CPLXAMD_SPLIT_B>(src, src2, ld_src, d, ld_dst, piece_stride, rows, cols8, st))
  return op == 0 ? SPLIT3(0) : op == 1 ? SPLIT3(1) : SPLIT3(2);
#undef SPLIT3
}

extern "C" int64_t cplxamd_absmax_ws_bytes(void) { return 2048 * (int64_t)sizeof(float); }

extern "C" int cplxamd_absmax_scale(const float* src, const float* src2, int64_t ld_src, int64_t rows, int cols, int op,
                                    float* scale, void* ws, void* stream) {
  if (!src || !scale || !ws || rows < 0 || cols < 0 || ld_src < cols) return CPLXAMD_EINVAL;
  if (op < 0 || op > 3 || (src2 && op != 1 && op != 3) || (op == 3 && !src2)) return CPLXAMD_EINVAL;
  if ((cols & 3) || (ld_src & 3)) return CPLXAMD_ESHAPE;
  if ((reinterpret_cast<uintptr_t>(src) & 15) || (src2 && (reinterpret_cast<uintptr_t>(src2) & 15))) return CPLXAMD_EALIGN;
  hipStream_t st = (hipStream_t)stream;
  const int cols4 = cols / 4;
  const int grid = stream_grid(rows * cols4, 256);
  float* partial = (float*)ws;
  if (op == 0) absmax_partial_kernel<0><<<grid, 256, 0, st>>>(src, nullptr, ld_src, rows, cols4, partial);
  else if (op == 1) absmax_partial_kernel<1><<<grid, 256, 0, st>>>(src, src2, ld_src, rows, cols4, partial);
  else if (op == 2) absmax_partial_kernel<2><<<grid, 256, 0, st>>>(src, nullptr, ld_src, rows, cols4, partial);
  else absmax_partial_kernel<3><<<grid, 256, 0, st>>>(src, src2, ld_src, rows, cols4, partial);
  CPLXAMD_CHECK_LAUNCH();
  absmax_final_kernel<<<1, 256, 0, st>>>(partial, (rows == 0 || cols == 0) ? 0 : grid, scale);
  CPLXAMD_CHECK_LAUNCH();
  return 0;
}

// The second stage alone: n per-block maxima formed by a producer (cplxamd_bn_bwd_sums_amax) -> scale = {s, 1 / s}.
extern "C" int cplxamd_absmax_scale_partials(const float* partial, int n, float* scale, void* stream) {
  if (!partial || !scale || n < 0) return CPLXAMD_EINVAL;
  absmax_final_kernel<<<1, 256, 0, (hipStream_t)stream>>>(partial, n, scale);
  CPLXAMD_CHECK_LAUNCH();
  return 0;
}

extern "C" int cplxamd_split2h(const float* src, const float* src2, int64_t ld_src, void* dst, int64_t ld_dst,
                               int64_t piece_stride, int64_t rows, int cols, int op, int pattern, const float* scale,
                               void* stream) {
  if (!src || !dst || !scale || rows < 0 || cols < 0 || ld_src < cols || ld_dst < cols) return CPLXAMD_EINVAL;
  if (op < 0 || op > 2 || (src2 && op != 1)) return CPLXAMD_EINVAL;
  if (pattern != CPLXAMD_SPLIT_A && pattern != CPLXAMD_SPLIT_B) return CPLXAMD_EINVAL;
  if ((cols & 7) || (ld_src & 3) || (ld_dst & 7) || (piece_stride & 7)) return CPLXAMD_ESHAPE;
  if ((reinterpret_cast<uintptr_t>(src) & 15) || (reinterpret_cast<uintptr_t>(dst) & 15) ||
      (src2 && (reinterpret_cast<uintptr_t>(src2) & 15)))
    return CPLXAMD_EALIGN;
  if (rows == 0 || cols == 0) return 0;
  const int cols8 = cols / 8;
  hipStream_t st = (hipStream_t)stream;
  uint16_t* d = (uint16_t*)dst;
#define SPLIT2(OP)                                                                                                      \
  (pattern == CPLXAMD_SPLIT_A ? launch_split2h<OP, CPLXAMD_SPLIT_A>(src, src2, ld_src, d, ld_dst, piece_stride, rows, cols8, scale, st) \
                              : launch_split2h<OP, CPLXAMD_SPLIT_B>(src, src2, ld_src, d, ld_dst, piece_stride, rows, cols8, scale, st))
  return op == 0 ? SPLIT2(0) : op == 1 ? SPLIT2(1) : SPLIT2(2);
#undef SPLIT2
}
