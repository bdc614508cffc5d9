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
      if (OP == 1) o.v[j] = x.v[j] * x.v[j] + y.v[j] * y.v[j];
      if (OP == 2) o.v[j] = x.v[j] * x.v[j];
      if (OP == 3) o.v[j] = expf(x.v[j]);
      if (OP == 4) o.v[j] = rn_sqrt(fmaf(y.v[j], y.v[j], x.v[j] * x.v[j]));
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
        o = x * x + y * y;
      }
      if (OP == 2) o = x * x;
      if (OP == 3) o = expf(x);
      if (OP == 4) {
        const float y = io<TI>::ld(b + e);
        o = rn_sqrt(fmaf(y, y, x * x));
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

// out[c] = sum_r in[r, c].  Stage 1: block = 64 column-quads x 4 row lanes over one row chunk,
// 8-16 B loads per lane, partial[chunk][c]; stage 2 sums the chunks.  (cols % 4 != 0: scalar path.)
constexpr int kColChunks = 1024;

template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* in, int64_t ld, float* partial,
                                                             int rows, int cols) {
  __shared__ float red[4][64][4];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = (blockIdx.x * 64 + tx) * 4;
  const int rows_per = (rows + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * rows_per;
  int r1 = r0 + rows_per;
  if (r1 > rows) r1 = rows;
  f4 acc = {{0.f, 0.f, 0.f, 0.f}};
  if (c < cols) {
    int r = r0 + ty;
    for (; r + 12 < r1; r += 16) {              // 4 independent loads in flight per thread
      f4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = ld4(in + (int64_t)(r + 4 * u) * ld + c);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc.v[j] += (v[0].v[j] + v[1].v[j]) + (v[2].v[j] + v[3].v[j]);
    }
    for (; r < r1; r += 4) {
      const f4 v = ld4(in + (int64_t)r * ld + c);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc.v[j] += v.v[j];
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) red[ty][tx][j] = acc.v[j];
  __syncthreads();
  if (ty == 0 && c < cols) {
    f4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o.v[j] = red[0][tx][j] + red[1][tx][j] + red[2][tx][j] + red[3][tx][j];
    st4(partial + (int64_t)blockIdx.y * cols + c, o);
  }
}

// 16 columns per block, 64 chunk lanes (fixed summation order: deterministic)
__global__ __launch_bounds__(1024) void colsum_final_kernel(const float* partial, int chunks, int cols,
                                                            float* out) {
  __shared__ float red[64][17];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + tx;
  float acc = 0.f;
  if (c < cols)
    for (int j = ty; j < chunks; j += 64) acc += partial[(int64_t)j * cols + c];
  red[ty][tx] = acc;
  __syncthreads();
  if (ty == 0 && c < cols) {
    float t = 0.f;
#pragma unroll 8
    for (int l = 0; l < 64; ++l) t += red[l][tx];
    out[c] = t;
  }
}

// tall and narrow ([B H W][C] activations: the bias gradient of a channels-last convolution): a thread owns 8
// consecutive columns (one 16-byte load for bf16), 256 / (cols / 8) row lanes per block, 4 rows in flight per thread
template <typename T>
__global__ __launch_bounds__(256) void colsum_rows_kernel(const T* in, float* partial, int64_t rows, int cols) {
  __shared__ float red[256 * 8];
  const int CG = cols >> 3, RL = 256 / CG;
  const int cg = threadIdx.x % CG, rl = threadIdx.x / CG;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (rl < RL) {
    const int64_t step = (int64_t)gridDim.x * RL;
    int64_t r = (int64_t)blockIdx.x * RL + rl;
    for (; r + 3 * step < rows; r += 4 * step) {
      f8 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = ld8(in + (r + u * step) * cols + cg * 8);
#pragma unroll
      for (int c = 0; c < 8; ++c)
        acc[c] += (v[0].h[c >> 2].v[c & 3] + v[1].h[c >> 2].v[c & 3]) + (v[2].h[c >> 2].v[c & 3] + v[3].h[c >> 2].v[c & 3]);
    }
    for (; r < rows; r += step) {
      const f8 v = ld8(in + r * cols + cg * 8);
#pragma unroll
      for (int c = 0; c < 8; ++c) acc[c] += v.h[c >> 2].v[c & 3];
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) red[rl * cols + cg * 8 + c] = acc[c];
  }
  __syncthreads();
  for (int f = threadIdx.x; f < cols; f += 256) {
    float t = 0.f;
    for (int l = 0; l < RL; ++l) t += red[l * cols + f];
    partial[(int64_t)blockIdx.x * cols + f] = t;
  }
}

// few rows (a head's [batch, outputs] gradient): 64 columns x 16 row lanes per block, 4 loads in flight per thread;
// blockIdx.y = plane (the two planes of a complex bias gradient in one launch)
template <typename T>
__global__ __launch_bounds__(1024) void colsum_kernel(const T* in0, const T* in1, int64_t ld, float* out0, float* out1,
                                                      int rows, int cols) {
  __shared__ float part[16][64];
  const T* in = blockIdx.y ? in1 : in0;
  float* out = blockIdx.y ? out1 : out0;
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int ty = threadIdx.x >> 6;
  float a[4] = {0.f, 0.f, 0.f, 0.f};
  if (c < cols) {
    int r = ty;
    for (; r + 48 < rows; r += 64) {
#pragma unroll
      for (int u = 0; u < 4; ++u) a[u] += io<T>::ld(in + (int64_t)(r + 16 * u) * ld + c);
    }
    for (; r < rows; r += 16) a[0] += io<T>::ld(in + (int64_t)r * ld + c);
  }
  part[ty][threadIdx.x & 63] = (a[0] + a[1]) + (a[2] + a[3]);
  __syncthreads();
  if (ty == 0 && c < cols) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += part[w][threadIdx.x];
    out[c] = t;
  }
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

// d|z| = g * z / |z|, 0 at z == 0 (the subgradient torch.norm uses, cplx.py:183-192; sqrt(re^2 + im^2)
// differentiated by autograd gives inf * 0 = NaN there)
template <typename T>
__global__ __launch_bounds__(kUT) void abs_bwd_kernel(const T* g, const T* xr, const T* xi, T* dxr, T* dxi,
                                                      int64_t n) {
  const int64_t n4 = n >> 2, stride = (int64_t)gridDim.x * kUT;
  for (int64_t i = (int64_t)blockIdx.x * kUT + threadIdx.x; i < n4; i += stride) {
    const f4 gg = ld4(g + 4 * i), a = ld4(xr + 4 * i), b = ld4(xi + 4 * i);
    f4 da, db;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float r = rn_sqrt(fmaf(b.v[j], b.v[j], a.v[j] * a.v[j]));
      const float s = r > 0.0f ? gg.v[j] / r : 0.0f;
      da.v[j] = s * a.v[j];
      db.v[j] = s * b.v[j];
    }
    st4(dxr + 4 * i, da);
    st4(dxi + 4 * i, db);
  }
  if (blockIdx.x == 0) {
    const int64_t e = (n4 << 2) + threadIdx.x;
    if (e < n) {
      const float a = io<T>::ld(xr + e), b = io<T>::ld(xi + e);
      const float r = rn_sqrt(fmaf(b, b, a * a));
      const float s = r > 0.0f ? io<T>::ld(g + e) / r : 0.0f;
      io<T>::st(dxr + e, s * a);
      io<T>::st(dxi + e, s * b);
    }
  }
}

// out = in * mask for one or two planes (masked layers: the sparsified weight, and -- with the planes
// being gradients -- the weight gradient; nn/masked/{real,complex}.py), float32 mask, any in/out dtype
template <typename TI, typename TO>
__global__ __launch_bounds__(kUT) void mask_mul_kernel(const TI* ar, const TI* ai, const float* mask, TO* outr,
                                                       TO* outi, int64_t n) {
  const int64_t n4 = n >> 2, stride = (int64_t)gridDim.x * kUT;
  for (int64_t i = (int64_t)blockIdx.x * kUT + threadIdx.x; i < n4; i += stride) {
    const f4 m = ld4(mask + 4 * i);
    f4 a = ld4(ar + 4 * i);
#pragma unroll
    for (int j = 0; j < 4; ++j) a.v[j] *= m.v[j];
    st4(outr + 4 * i, a);
    if (ai) {
      f4 b = ld4(ai + 4 * i);
#pragma unroll
      for (int j = 0; j < 4; ++j) b.v[j] *= m.v[j];
      st4(outi + 4 * i, b);
    }
  }
  if (blockIdx.x == 0) {
    const int64_t e = (n4 << 2) + threadIdx.x;
    if (e < n) {
      io<TO>::st(outr + e, io<TI>::ld(ar + e) * mask[e]);
      if (ai) io<TO>::st(outi + e, io<TI>::ld(ai + e) * mask[e]);
    }
  }
}

}  // namespace cplxamd

using namespace cplxamd;

extern "C" {

int cplxamd_cplx_abs_fwd(const void* xr, const void* xi, void* out, int64_t n, int dtype, void* stream) {
  if (!xr || !xi || !out || n < 0) return CPLXAMD_EINVAL;
  return launch_ew<4>(xr, xi, out, n, dtype, dtype, (hipStream_t)stream);
}

int cplxamd_cplx_abs_bwd(const void* g, const void* xr, const void* xi, void* dxr, void* dxi, int64_t n,
                         int dtype, void* stream) {
  if (!g || !xr || !xi || !dxr || !dxi || n < 0) return CPLXAMD_EINVAL;
  const int grid = stream_grid(n >> 2, kUT);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CPLXAMD_F32)
    abs_bwd_kernel<float><<<grid, kUT, 0, st>>>((const float*)g, (const float*)xr, (const float*)xi,
                                                (float*)dxr, (float*)dxi, n);
  else if (dtype == CPLXAMD_BF16)
    abs_bwd_kernel<bf16_t><<<grid, kUT, 0, st>>>((const bf16_t*)g, (const bf16_t*)xr, (const bf16_t*)xi,
                                                 (bf16_t*)dxr, (bf16_t*)dxi, n);
  else
    return CPLXAMD_EINVAL;
  CPLXAMD_CHECK_LAUNCH();
  return 0;
}

int cplxamd_mask_mul(const void* in_r, const void* in_i, const float* mask, void* out_r, void* out_i,
                     int64_t n, int in_dtype, int out_dtype, void* stream) {
  if (!in_r || !mask || !out_r || n < 0 || ((in_i == nullptr) != (out_i == nullptr))) return CPLXAMD_EINVAL;
  const int grid = stream_grid(n >> 2, kUT);
  hipStream_t st = (hipStream_t)stream;
#define MM(TI, TO) mask_mul_kernel<TI, TO><<<grid, kUT, 0, st>>>((const TI*)in_r, (const TI*)in_i, mask, (TO*)out_r, (TO*)out_i, n)
  if (in_dtype == CPLXAMD_F32 && out_dtype == CPLXAMD_F32) MM(float, float);
  else if (in_dtype == CPLXAMD_F32 && out_dtype == CPLXAMD_BF16) MM(float, bf16_t);
  else if (in_dtype == CPLXAMD_BF16 && out_dtype == CPLXAMD_F32) MM(bf16_t, float);
  else if (in_dtype == CPLXAMD_BF16 && out_dtype == CPLXAMD_BF16) MM(bf16_t, bf16_t);
  else return CPLXAMD_EINVAL;
#undef MM
  CPLXAMD_CHECK_LAUNCH();
  return 0;
}

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

int64_t cplxamd_colsum_ws_bytes(int cols) { return (int64_t)kColChunks * cols * sizeof(float); }

int cplxamd_colsum(const void* in, int64_t ld, float* out, int rows, int cols, int dtype,
                   void* ws, void* stream) {
  if (!in || !out || rows < 0 || cols < 0) return CPLXAMD_EINVAL;
  if (cols == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const bool vec = ws && (cols % 4 == 0) && (ld % 4 == 0) && rows >= 64 &&
                   ((reinterpret_cast<uintptr_t>(in) & 15) == 0);
  if (vec && cols % 8 == 0 && cols <= 512 && ld == cols && rows >= 8192) {
    const int RL = 256 / (cols / 8);
    int64_t chunks = ((int64_t)rows + 4 * RL - 1) / (4 * RL);
    if (chunks > kColChunks) chunks = kColChunks;
    if (dtype == CPLXAMD_F32)
      colsum_rows_kernel<float><<<(unsigned)chunks, 256, 0, st>>>((const float*)in, (float*)ws, rows, cols);
    else if (dtype == CPLXAMD_BF16)
      colsum_rows_kernel<bf16_t><<<(unsigned)chunks, 256, 0, st>>>((const bf16_t*)in, (float*)ws, rows, cols);
    else
      return CPLXAMD_EINVAL;
    CPLXAMD_CHECK_LAUNCH();
    colsum_final_kernel<<<(cols + 15) / 16, 1024, 0, st>>>((const float*)ws, (int)chunks, cols, out);
    CPLXAMD_CHECK_LAUNCH();
    return 0;
  }
  if (vec) {
    const int gx = (cols / 4 + 63) / 64;
    int chunks = (2048 + gx - 1) / gx;            // ~8 blocks per CU
    if (chunks > rows / 256) chunks = rows / 256; // >= 256 rows per chunk (the final pass is serial in chunks)
    if (chunks > kColChunks) chunks = kColChunks;
    if (chunks < 1) chunks = 1;
    dim3 grid(gx, chunks);
    if (dtype == CPLXAMD_F32)
      colsum_partial_kernel<float><<<grid, 256, 0, st>>>((const float*)in, ld, (float*)ws, rows, cols);
    else if (dtype == CPLXAMD_BF16)
      colsum_partial_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)in, ld, (float*)ws, rows, cols);
    else
      return CPLXAMD_EINVAL;
    CPLXAMD_CHECK_LAUNCH();
    colsum_final_kernel<<<(cols + 15) / 16, 1024, 0, st>>>((const float*)ws, chunks, cols, out);
    CPLXAMD_CHECK_LAUNCH();
    return 0;
  }
  const int grid = (cols + 63) / 64;
  if (dtype == CPLXAMD_F32)
    colsum_kernel<float><<<grid, 1024, 0, st>>>((const float*)in, nullptr, ld, out, nullptr, rows, cols);
  else if (dtype == CPLXAMD_BF16)
    colsum_kernel<bf16_t><<<grid, 1024, 0, st>>>((const bf16_t*)in, nullptr, ld, out, nullptr, rows, cols);
  else
    return CPLXAMD_EINVAL;
  CPLXAMD_CHECK_LAUNCH();
  return 0;
}

/* Column sums of the two planes of a complex [rows, cols] tensor (the complex bias gradient of a linear layer): one
 * launch where cplxamd_colsum takes its few-rows kernel, else two cplxamd_colsum passes (ws: as for cplxamd_colsum). */
int cplxamd_colsum2(const void* in_r, const void* in_i, int64_t ld, float* out_r, float* out_i, int rows, int cols,
                    int dtype, void* ws, void* stream) {
  if (!in_r || !in_i || !out_r || !out_i || rows < 0 || cols < 0) return CPLXAMD_EINVAL;
  if (cols == 0) return 0;
  const bool vec = ws && (cols % 4 == 0) && (ld % 4 == 0) && rows >= 64 && ((reinterpret_cast<uintptr_t>(in_r) & 15) == 0) &&
                   ((reinterpret_cast<uintptr_t>(in_i) & 15) == 0);
  if (vec) {
    const int rc = cplxamd_colsum(in_r, ld, out_r, rows, cols, dtype, ws, stream);
    return rc ? rc : cplxamd_colsum(in_i, ld, out_i, rows, cols, dtype, ws, stream);
  }
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((cols + 63) / 64, 2);
  if (dtype == CPLXAMD_F32)
    colsum_kernel<float><<<grid, 1024, 0, st>>>((const float*)in_r, (const float*)in_i, ld, out_r, out_i, rows, cols);
  else if (dtype == CPLXAMD_BF16)
    colsum_kernel<bf16_t><<<grid, 1024, 0, st>>>((const bf16_t*)in_r, (const bf16_t*)in_i, ld, out_r, out_i, rows, cols);
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
