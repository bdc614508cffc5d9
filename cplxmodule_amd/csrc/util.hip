// HBM-bound elementwise / layout helpers of the layers (include/cplxamd.h, last section).
#include "common.h"

// The reference issues separate torch ops (one rounding each); keep the compiler from
// fusing a*b+c into fma so that parity-mode results are bit-identical.  Explicit fmaf()
// calls below are deliberate.
#pragma clang fp contract(off)

namespace cplxamd {

constexpr int kUT = 256;

template <typename TI, typename TO, int OP>  // OP: 0 cast, 1 abs2 (cplx), 2 sqr (real), 3 exp, 4 modulus
__global__ __launch_bounds__(kUT) void ew_kernel(const TI* a, const TI* b, TO* out, int64_t n) {
  const int64_t n4 = n >> 2, stride = (int64_t)gridDim.x * kUT;
  for (int64_t i = (int64_t)blockIdx.x * kUT + threadIdx.x; i < n4; i += stride) {
    const f4 x = ld4(a + 4 * i);
    f4 y, o;
    if (OP == 1 || OP == 4) y = ld4(b + 4 * i);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (OP == 0) o.v[j] = x.v[j];
      if (OP == 1) o.v[j] = __fadd_rn(__fmul_rn(x.v[j], x.v[j]), __fmul_rn(y.v[j], y.v[j]));
      if (OP == 2) o.v[j] = x.v[j] * x.v[j];
      if (OP == 3) o.v[j] = expf(x.v[j]);
      if (OP == 4) o.v[j] = __fsqrt_rn(__fmaf_rn(y.v[j], y.v[j], __fmul_rn(x.v[j], x.v[j])));
    }
    st4(out + 4 * i, o);
  }
  if (blockIdx.x == 0) {
    const int64_t e = (n4 << 2) + threadIdx.x;
    if (e < n) {
      const float x = io<TI>::ld(a + e);
      float o = x;
      if (OP == 1) {
        const float y = io<TI>::ld(b + e);
        o = __fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y));
      }
      if (OP == 2) o = x * x;
      if (OP == 3) o = expf(x);
      if (OP == 4) {
        const float y = io<TI>::ld(b + e);
        o = __fsqrt_rn(__fmaf_rn(y, y, __fmul_rn(x, x)));
      }
      io<TO>::st(out + e, o);
    }
  }
}

template <int OP>
static int launch_ew(const void* a, const void* b, void* out, int64_t n, int in_dtype,
                     int out_dtype, hipStream_t st) {
  const int grid = stream_grid(n >> 2, kUT);
#define EW(TI, TO) ew_kernel<TI, TO, OP><<<grid, kUT, 0, st>>>((const TI*)a, (const TI*)b, (TO*)out, n)
  if (in_dtype == CPLXAMD_F32 && out_dtype == CPLXAMD_F32) EW(float, float);
  else if (in_dtype == CPLXAMD_F32 && out_dtype == CPLXAMD_BF16) EW(float, bf16_t);
  else if (in_dtype == CPLXAMD_BF16 && out_dtype == CPLXAMD_F32) EW(bf16_t, float);
  else if (in_dtype == CPLXAMD_BF16 && out_dtype == CPLXAMD_BF16) EW(bf16_t, bf16_t);
  else return CPLXAMD_EINVAL;
#undef EW
  CPLXAMD_CHECK_LAUNCH();
  return 0;
}

// 64x64 tile transpose through LDS (+1 pad), coalesced on both sides
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const T* in, int64_t ld_in, T* out,
                                                        int64_t ld_out, int rows, int cols) {
  __shared__ T tile[64][65];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int r = r0 + ty + 4 * j, c = c0 + tx;
    if (r < rows && c < cols) tile[ty + 4 * j][tx] = in[(int64_t)r * ld_in + c];
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int c = c0 + ty + 4 * j, r = r0 + tx;
    if (r < rows && c < cols) out[(int64_t)c * ld_out + r] = tile[tx][ty + 4 * j];
  }
}

// out[c] = sum_r in[r, c]: each block owns 64 columns x a row slab; fp32 atomics-free two-level
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* in, int64_t ld, float* out, int rows,
                                                     int cols) {
  __shared__ float part[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int ty = threadIdx.x >> 6;
  float acc = 0.0f;
  if (c < cols)
    for (int r = ty; r < rows; r += 4) acc += io<T>::ld(in + (int64_t)r * ld + c);
  part[ty][threadIdx.x & 63] = acc;
  __syncthreads();
  if (ty == 0 && c < cols)
    out[c] = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
}

template <typename T, typename TG, bool CPLX>
__global__ __launch_bounds__(kUT) void dx_accum_kernel(T* dxr, T* dxi, const T* xr, const T* xi,
                                                       const TG* ga, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * kUT;
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * kUT + threadIdx.x; i < n4; i += stride) {
    const f4 g = ld4(ga + 4 * i);
    f4 d = ld4(dxr + 4 * i);
    const f4 x = ld4(xr + 4 * i);
#pragma unroll
    for (int j = 0; j < 4; ++j) d.v[j] = fmaf(2.0f * x.v[j], g.v[j], d.v[j]);
    st4(dxr + 4 * i, d);
    if (CPLX) {
      f4 e = ld4(dxi + 4 * i);
      const f4 y = ld4(xi + 4 * i);
#pragma unroll
      for (int j = 0; j < 4; ++j) e.v[j] = fmaf(2.0f * y.v[j], g.v[j], e.v[j]);
      st4(dxi + 4 * i, e);
    }
  }
  if (blockIdx.x == 0) {
    const int64_t e = (n4 << 2) + threadIdx.x;
    if (e < n) {
      const float g = io<TG>::ld(ga + e);
      io<T>::st(dxr + e, fmaf(2.0f * io<T>::ld(xr + e), g, io<T>::ld(dxr + e)));
      if (CPLX) io<T>::st(dxi + e, fmaf(2.0f * io<T>::ld(xi + e), g, io<T>::ld(dxi + e)));
    }
  }
}

}  // namespace cplxamd

using namespace cplxamd;

extern "C" {

int cplxamd_abs2(const void* xr, const void* xi, void* out, int64_t n, int in_dtype,
                 int out_dtype, void* stream) {
  if (!xr || !out || n < 0) return CPLXAMD_EINVAL;
  if (xi) return launch_ew<1>(xr, xi, out, n, in_dtype, out_dtype, (hipStream_t)stream);
  return launch_ew<2>(xr, nullptr, out, n, in_dtype, out_dtype, (hipStream_t)stream);
}

int cplxamd_modulus(const float* xr, const float* xi, float* out, int64_t n, void* stream) {
  if (!xr || !xi || !out || n < 0) return CPLXAMD_EINVAL;
  return launch_ew<4>(xr, xi, out, n, CPLXAMD_F32, CPLXAMD_F32, (hipStream_t)stream);
}

int cplxamd_exp(const float* x, void* out, int64_t n, int out_dtype, void* stream) {
  if (!x || !out || n < 0) return CPLXAMD_EINVAL;
  return launch_ew<3>(x, nullptr, out, n, CPLXAMD_F32, out_dtype, (hipStream_t)stream);
}

int cplxamd_cast(const void* in, void* out, int64_t n, int in_dtype, int out_dtype,
                 void* stream) {
  if (!in || !out || n < 0) return CPLXAMD_EINVAL;
  return launch_ew<0>(in, nullptr, out, n, in_dtype, out_dtype, (hipStream_t)stream);
}

int cplxamd_transpose(const void* in, int64_t ld_in, void* out, int64_t ld_out, int rows,
                      int cols, int dtype, void* stream) {
  if (!in || !out || rows < 0 || cols < 0 || ld_in < cols || ld_out < rows) return CPLXAMD_EINVAL;
  if (rows == 0 || cols == 0) return 0;
  dim3 grid((cols + 63) / 64, (rows + 63) / 64);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CPLXAMD_F32)
    transpose_kernel<float><<<grid, 256, 0, st>>>((const float*)in, ld_in, (float*)out, ld_out, rows, cols);
  else if (dtype == CPLXAMD_BF16)
    transpose_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)in, ld_in, (bf16_t*)out, ld_out, rows, cols);
  else
    return CPLXAMD_EINVAL;
  CPLXAMD_CHECK_LAUNCH();
  return 0;
}

int cplxamd_colsum(const void* in, int64_t ld, float* out, int rows, int cols, int dtype,
                   void* stream) {
  if (!in || !out || rows < 0 || cols < 0) return CPLXAMD_EINVAL;
  if (cols == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int grid = (cols + 63) / 64;
  if (dtype == CPLXAMD_F32)
    colsum_kernel<float><<<grid, 256, 0, st>>>((const float*)in, ld, out, rows, cols);
  else if (dtype == CPLXAMD_BF16)
    colsum_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)in, ld, out, rows, cols);
  else
    return CPLXAMD_EINVAL;
  CPLXAMD_CHECK_LAUNCH();
  return 0;
}

int cplxamd_lrt_dx_accum(void* dxr, void* dxi, const void* xr, const void* xi, const void* ga,
                         int64_t n, int dtype, int ga_dtype, void* stream) {
  if (!dxr || !xr || !ga || n < 0 || ((dxi == nullptr) != (xi == nullptr))) return CPLXAMD_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int grid = stream_grid(n >> 2, kUT);
#define DX(T, TG)                                                                              \
  do {                                                                                         \
    if (dxi) dx_accum_kernel<T, TG, true><<<grid, kUT, 0, st>>>((T*)dxr, (T*)dxi, (const T*)xr, (const T*)xi, (const TG*)ga, n); \
    else dx_accum_kernel<T, TG, false><<<grid, kUT, 0, st>>>((T*)dxr, (T*)dxi, (const T*)xr, (const T*)xi, (const TG*)ga, n);    \
  } while (0)
  if (dtype == CPLXAMD_F32 && ga_dtype == CPLXAMD_F32) DX(float, float);
  else if (dtype == CPLXAMD_BF16 && ga_dtype == CPLXAMD_BF16) DX(bf16_t, bf16_t);
  else if (dtype == CPLXAMD_BF16 && ga_dtype == CPLXAMD_F32) DX(bf16_t, float);
  else if (dtype == CPLXAMD_F32 && ga_dtype == CPLXAMD_BF16) DX(float, bf16_t);
  else return CPLXAMD_EINVAL;
#undef DX
  CPLXAMD_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
