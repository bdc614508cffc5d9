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
