// K3: complex batch normalisation (Trabelsi et al.), forward and backward, as
//   pass 1  per-feature moment reduction (wave shuffles + fp64 accumulation)  -> partials
//   tiny    per-feature finalize: 2x2 inverse square root, affine folded in, running stats
//   pass 2  elementwise apply
// instead of the reference's ~25 full-tensor passes (cplxmodule/nn/modules/batchnorm.py:62-123
// whiten2x2, :254-278 affine).  Layout: planar (re, im), each [B, F, S] contiguous (S = product
// of the spatial dims, 1 for [B, F] inputs).  HBM traffic fwd: 2 reads + 1 write of both planes.
//
// Backward is derived by hand through the closed-form inverse square root (gradients flow
// through s and t, batchnorm.py:105-113); formulas in oracle/cplx_oracle.py:cplx_batch_norm_bwd.
#include "common.h"

namespace cplxamd {

constexpr int kBnT = 256;
constexpr int kBnAmaxSlots = 2048;   // per-block maxima of the row-kernel backward (cplxamd_bn_bwd_sums_amax): its grid never exceeds this
constexpr int kBnMaxChunks = 512;
constexpr int kSaved = 8;    // per-feature saved stats: mu, mv, p, q, w, vuu, vuv, vvv
constexpr int kFwdCoef = 8;  // mu, mv, a00, a01, a10, a11, b0, b1
constexpr int kBwdCoef = kBnBwdCoef; // mu, mv, e00, e01, e10, e11, cuu, cuv, cvv, ku, kv, pad

// The channels-last row kernels stream planes of gigabytes, every byte touched once per pass: nontemporal accesses for
// the bf16 planes (bit 0: stores, bit 1: loads), as in kl.hip: forward -4 %, backward -4 % same box.
#ifndef BN_NT
#define BN_NT 3
#endif
typedef uint32_t u32x4_nt __attribute__((ext_vector_type(4)));
typedef float f32x4_nt __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f8 ld8s(const bf16_t* p) {
#if BN_NT & 2
  const u32x4_nt t = __builtin_nontemporal_load(reinterpret_cast<const u32x4_nt*>(p));
  f8 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    r.h[e >> 1].v[(e & 1) * 2] = __uint_as_float(t[e] << 16);
    r.h[e >> 1].v[(e & 1) * 2 + 1] = __uint_as_float(t[e] & 0xffff0000u);
  }
  return r;
#else
  return ld8(p);
#endif
}
// (float32 planes: plain accesses -- nontemporal ones cost the backward 25 % on the same box, scripts/r06/bn_nt_ab.py)
__device__ __forceinline__ f8 ld8s(const float* p) { return ld8(p); }
__device__ __forceinline__ void st8s(bf16_t* p, const f8& a) {
#if BN_NT & 1
  u32x4_nt w;
#pragma unroll
  for (int e = 0; e < 4; ++e) w[e] = pack_bf16(a.h[e >> 1].v[(e & 1) * 2], a.h[e >> 1].v[(e & 1) * 2 + 1]);
  __builtin_nontemporal_store(w, reinterpret_cast<u32x4_nt*>(p));
#else
  st8(p, a);
#endif
}
__device__ __forceinline__ void st8s(float* p, const f8& a) { st8(p, a); }

struct BnGeom {
  int64_t B, S;
  int F;
  int seg;        // segments per plane
  int64_t seglen; // elements per segment (multiple of 4)
  int chunks;     // gridDim.y of the reduction kernels
  bool small;     // S > 1 with small planes: bn_reduce_small
};

static BnGeom bn_geom(int64_t B, int F, int64_t S) {
  BnGeom g{B, S, F, 1, S, 1, false};
  if (S > 1 && S < 1024 && B * S < 0x7fffffff) {
    // small planes (the 28 x 28 ... 7 x 7 maps of cfg5): blocks of >= 4096 elements of ONE feature, at most 64 per feature
    g.small = true;
    const int64_t want = (B * S + 4095) / 4096;
    g.chunks = (int)(want < 64 ? want : 64);
  } else if (S > 1) {
    // aim at ~4096 blocks of >= 4096 elements
    int64_t want = 4096 / (F > 0 ? F : 1);
    if (want < 1) want = 1;
    int64_t seg = 1;
    while (B * seg < want && S / (seg * 2) >= 4096) seg *= 2;
    g.seg = (int)seg;
    g.seglen = ((S + seg - 1) / seg + 3) / 4 * 4;
    int64_t units = B * seg;
    g.chunks = (int)(units < want ? units : want);
  } else {
    int64_t want = 2048 / ((F + 63) / 64);
    if (want < 1) want = 1;
    int64_t rows4 = (B + 3) / 4;
    g.chunks = (int)(rows4 < want ? rows4 : want);
  }
  if (g.chunks > kBnMaxChunks) g.chunks = kBnMaxChunks;
  if (g.chunks < 1) g.chunks = 1;
  return g;
}

template <int NS> struct Acc { double v[NS]; };

template <int NS, bool BWD>
__device__ __forceinline__ void accum(Acc<NS>& a, float x0, float x1, float g0, float g1,
                                      float mu, float mv) {
  if (!BWD) {
    const double u = x0, v = x1;
    a.v[0] += u; a.v[1] += v; a.v[2] += u * u; a.v[3] += v * v; a.v[4] += u * v;
  } else {
    const float cu = x0 - mu, cv = x1 - mv;
    a.v[0] += (double)g0; a.v[1] += (double)g1;
    a.v[2] += (double)(g0 * cu); a.v[3] += (double)(g0 * cv);
    a.v[4] += (double)(g1 * cu); a.v[5] += (double)(g1 * cv);
  }
}

// S > 1: one block per (feature, chunk); partial[(chunk*F + f)*NS + j]
template <typename T, int NS, bool BWD>
__global__ __launch_bounds__(kBnT) void bn_reduce_planes(const T* xr, const T* xi, const T* gr,
                                                         const T* gi, const float* saved,
                                                         BnGeom g, double* partial) {
  __shared__ double red[kBnT / 64];
  const int f = blockIdx.x;
  float mu = 0.f, mv = 0.f;
  if (BWD) { mu = saved[f]; mv = saved[g.F + f]; }
  Acc<NS> a;
#pragma unroll
  for (int j = 0; j < NS; ++j) a.v[j] = 0.0;
  const int64_t units = g.B * g.seg;
  const bool vec = (g.S & 3) == 0;
  for (int64_t unit = blockIdx.y; unit < units; unit += gridDim.y) {
    const int64_t b = unit / g.seg;
    const int sg = (int)(unit - b * g.seg);
    const int64_t s0 = (int64_t)sg * g.seglen;
    int64_t s1 = s0 + g.seglen;
    if (s1 > g.S) s1 = g.S;
    const int64_t base = (b * g.F + f) * g.S;
    if (vec) {
      for (int64_t s = s0 + 4 * threadIdx.x; s < s1; s += 4 * kBnT) {
        const f4 u = ld4(xr + base + s), v = ld4(xi + base + s);
        f4 p = u, q = v;
        if (BWD) { p = ld4(gr + base + s); q = ld4(gi + base + s); }
#pragma unroll
        for (int j = 0; j < 4; ++j) accum<NS, BWD>(a, u.v[j], v.v[j], p.v[j], q.v[j], mu, mv);
      }
    } else {
      for (int64_t s = s0 + threadIdx.x; s < s1; s += kBnT) {
        const float u = io<T>::ld(xr + base + s), v = io<T>::ld(xi + base + s);
        float p = 0.f, q = 0.f;
        if (BWD) { p = io<T>::ld(gr + base + s); q = io<T>::ld(gi + base + s); }
        accum<NS, BWD>(a, u, v, p, q, mu, mv);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NS; ++j) {
    const double s = block_sum<double, kBnT>(a.v[j], red);
    if (threadIdx.x == 0) partial[((int64_t)blockIdx.y * g.F + f) * NS + j] = s;
  }
}

// S > 1, small planes: one block per (feature, chunk) walks the feature's B * S elements as ONE index range, four
// independent loads per thread in flight.  (bn_reduce_planes gives every (b, f) plane to a block: a 7 x 7 plane keeps 49
// of 256 threads busy, and each of the B x F blocks pays the six block reductions -- 17 us for cfg5's first layer, whose
// 12.8 MB are 2 us of HBM time.)
template <typename T, int NS, bool BWD>
__global__ __launch_bounds__(kBnT) void bn_reduce_small(const T* xr, const T* xi, const T* gr, const T* gi,
                                                        const float* saved, int B, int F, int S, double* partial) {
  __shared__ double red[kBnT / 64][NS];
  const int f = blockIdx.x;
  float mu = 0.f, mv = 0.f;
  if (BWD) { mu = saved[f]; mv = saved[F + f]; }
  Acc<NS> a;
#pragma unroll
  for (int j = 0; j < NS; ++j) a.v[j] = 0.0;
  const unsigned E = (unsigned)B * (unsigned)S, stride = gridDim.y * kBnT;
  for (unsigned i0 = blockIdx.y * kBnT + threadIdx.x; i0 < E; i0 += 4u * stride) {
    float u[4], v[4], p[4], q[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const unsigned i = i0 + r * stride;
      const unsigned ii = i < E ? i : 0u, b = ii / (unsigned)S;
      const int64_t off = ((int64_t)b * F + f) * S + (ii - b * (unsigned)S);
      u[r] = io<T>::ld(xr + off); v[r] = io<T>::ld(xi + off);
      p[r] = BWD ? io<T>::ld(gr + off) : 0.f; q[r] = BWD ? io<T>::ld(gi + off) : 0.f;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (i0 + r * stride < E) accum<NS, BWD>(a, u[r], v[r], p[r], q[r], mu, mv);
  }
  // all NS moments through ONE barrier (six block_sum calls were twelve): waves reduce by shuffles, the first NS lanes of
  // wave 0 add the kBnT / 64 wave sums in a fixed order
#pragma unroll
  for (int j = 0; j < NS; ++j) {
    const double t = wave_sum(a.v[j]);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][j] = t;
  }
  __syncthreads();
  if (threadIdx.x < NS) {
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < kBnT / 64; ++w) t += red[w][threadIdx.x];
    partial[((int64_t)blockIdx.y * F + f) * NS + threadIdx.x] = t;
  }
}

// S == 1 ([B, F] input): threads run along F (coalesced), 4 row lanes per block
template <typename T, int NS, bool BWD>
__global__ __launch_bounds__(kBnT) void bn_reduce_cols(const T* xr, const T* xi, const T* gr,
                                                       const T* gi, const float* saved, BnGeom g,
                                                       double* partial) {
  __shared__ double red[4][64];
  const int fx = threadIdx.x & 63, by = threadIdx.x >> 6;
  const int f = blockIdx.x * 64 + fx;
  Acc<NS> a;
#pragma unroll
  for (int j = 0; j < NS; ++j) a.v[j] = 0.0;
  if (f < g.F) {
    float mu = 0.f, mv = 0.f;
    if (BWD) { mu = saved[f]; mv = saved[g.F + f]; }
    for (int64_t b = (int64_t)blockIdx.y * 4 + by; b < g.B; b += 4 * (int64_t)gridDim.y) {
      const int64_t o = b * g.F + f;
      const float u = io<T>::ld(xr + o), v = io<T>::ld(xi + o);
      float p = 0.f, q = 0.f;
      if (BWD) { p = io<T>::ld(gr + o); q = io<T>::ld(gi + o); }
      accum<NS, BWD>(a, u, v, p, q, mu, mv);
    }
  }
#pragma unroll
  for (int j = 0; j < NS; ++j) {
    red[by][fx] = a.v[j];
    __syncthreads();
    if (by == 0 && f < g.F)
      partial[((int64_t)blockIdx.y * g.F + f) * NS + j] = red[0][fx] + red[1][fx] + red[2][fx] + red[3][fx];
    __syncthreads();
  }
}

// S == 1 with many rows and F % 8 == 0 (channels-last activations [B H W][F]): a thread owns 8 consecutive channels
// (16-byte accesses for bf16), kBnT / (F / 8) row lanes per block, two rows in flight per thread.
constexpr int kBnRowChunks = 1024;
template <typename T, int NS, bool BWD>
__global__ __launch_bounds__(kBnT) void bn_reduce_rows(const T* xr, const T* xi, const T* gr, const T* gi,
                                                       const float* saved, int64_t R, int F, double* partial) {
  __shared__ double red[kBnT * 8];
  const int CG = F >> 3, RL = kBnT / CG;
  const int cg = threadIdx.x % CG, rl = threadIdx.x / CG;
  Acc<NS> a[8];
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int j = 0; j < NS; ++j) a[c].v[j] = 0.0;
  float mu[8], mv[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    mu[c] = (BWD && rl < RL) ? saved[cg * 8 + c] : 0.f;
    mv[c] = (BWD && rl < RL) ? saved[F + cg * 8 + c] : 0.f;
  }
  if (rl < RL) {
    const int64_t step = (int64_t)gridDim.x * RL;
    for (int64_t r = (int64_t)blockIdx.x * RL + rl; r < R; r += 2 * step) {
      const int64_t o0 = r * F + cg * 8;
      const bool two = r + step < R;
      const int64_t o1 = two ? o0 + step * F : o0;
      const f8 u0 = ld8s(xr + o0), v0 = ld8s(xi + o0), u1 = ld8s(xr + o1), v1 = ld8s(xi + o1);
      f8 p0 = u0, q0 = v0, p1 = u1, q1 = v1;
      if (BWD) { p0 = ld8s(gr + o0); q0 = ld8s(gi + o0); p1 = ld8s(gr + o1); q1 = ld8s(gi + o1); }
#pragma unroll
      for (int c = 0; c < 8; ++c)
        accum<NS, BWD>(a[c], u0.h[c >> 2].v[c & 3], v0.h[c >> 2].v[c & 3], p0.h[c >> 2].v[c & 3], q0.h[c >> 2].v[c & 3], mu[c], mv[c]);
      if (two) {
#pragma unroll
        for (int c = 0; c < 8; ++c)
          accum<NS, BWD>(a[c], u1.h[c >> 2].v[c & 3], v1.h[c >> 2].v[c & 3], p1.h[c >> 2].v[c & 3], q1.h[c >> 2].v[c & 3], mu[c], mv[c]);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NS; ++j) {
    if (rl < RL) {
#pragma unroll
      for (int c = 0; c < 8; ++c) red[rl * F + cg * 8 + c] = a[c].v[j];
    }
    __syncthreads();
    for (int f = threadIdx.x; f < F; f += kBnT) {
      double t = 0.0;
      for (int l = 0; l < RL; ++l) t += red[l * F + f];
      partial[((int64_t)blockIdx.x * F + f) * NS + j] = t;
    }
    __syncthreads();
  }
}

// SUMS: also the per-channel sums of the two OUTPUT planes (as stored, i.e. after rounding to T) -> sums_partial[block]
// [2][F]: the bias gradient of the layer that produced this layer's input is exactly that column sum of dX, and this
// pass already holds every dX value in registers.
template <typename T, bool BWD, bool SUMS>
__global__ __launch_bounds__(kBnT) void bn_apply_rows(const T* xr, const T* xi, const T* gr, const T* gi, T* yr, T* yi,
                                                      const float* coef, int64_t R, int F, float* sums_partial,
                                                      float* amax_partial = nullptr) {
  __shared__ float sred[SUMS ? kBnT * 16 : 1];
  const int CG = F >> 3, RL = kBnT / CG;
  const int cg = threadIdx.x % CG, rl = threadIdx.x / CG;
  float su[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, sv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float amax = 0.f;                    // SUMS + amax_partial: max |.| over both output planes as stored (this block's rows)
  if (rl < RL) {
  constexpr int NC = BWD ? kBwdCoef : kFwdCoef;
  float k[8][BWD ? 11 : 8];
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int j = 0; j < (BWD ? 11 : 8); ++j) k[c][j] = coef[(int64_t)(cg * 8 + c) * NC + j];
  const int64_t step = (int64_t)gridDim.x * RL;
  for (int64_t r = (int64_t)blockIdx.x * RL + rl; r < R; r += step) {
    const int64_t o = r * F + cg * 8;
    const f8 u = ld8s(xr + o), v = ld8s(xi + o);
    f8 p = u, q = v, ou, ov;
    if (BWD) { p = ld8s(gr + o); q = ld8s(gi + o); }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float cu = u.h[c >> 2].v[c & 3] - k[c][0], cv = v.h[c >> 2].v[c & 3] - k[c][1];
      if (!BWD) {
        ou.h[c >> 2].v[c & 3] = fmaf(k[c][2], cu, fmaf(k[c][3], cv, k[c][6]));
        ov.h[c >> 2].v[c & 3] = fmaf(k[c][4], cu, fmaf(k[c][5], cv, k[c][7]));
      } else {
        const float pp = p.h[c >> 2].v[c & 3], qq = q.h[c >> 2].v[c & 3];
        ou.h[c >> 2].v[c & 3] = k[c][2] * pp + k[c][3] * qq + k[c][6] * cu + k[c][7] * cv - k[c][9];
        ov.h[c >> 2].v[c & 3] = k[c][4] * pp + k[c][5] * qq + k[c][8] * cv + k[c][7] * cu - k[c][10];
      }
    }
    st8s(yr + o, ou);
    st8s(yi + o, ov);
    if (SUMS) {
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        float a = ou.h[c >> 2].v[c & 3], b = ov.h[c >> 2].v[c & 3];
        if (sizeof(T) == 2) { a = bf16_to_f32(f32_to_bf16(a)); b = bf16_to_f32(f32_to_bf16(b)); }
        su[c] += a; sv[c] += b;
        amax = fmaxf(amax, fmaxf(fabsf(a), fabsf(b)));
      }
    }
  }
  }
  if (SUMS) {
    if (rl < RL) {
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        sred[rl * 2 * F + cg * 8 + c] = su[c];
        sred[rl * 2 * F + F + cg * 8 + c] = sv[c];
      }
    }
    __syncthreads();
    for (int f = threadIdx.x; f < 2 * F; f += kBnT) {
      float t = 0.f;
      for (int l = 0; l < RL; ++l) t += sred[l * 2 * F + f];
      sums_partial[(int64_t)blockIdx.x * 2 * F + f] = t;
    }
    if (amax_partial) {                // (a producer-side absmax: the consumer that cuts these planes into half pieces
      __syncthreads();                 //  -- x3.py 'x2' -- needs max |.| of both and would otherwise read them once more)
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
      if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = amax;
      __syncthreads();
      if (threadIdx.x == 0) {
        float m = sred[0];
        for (int w = 1; w < kBnT / 64; ++w) m = fmaxf(m, sred[w]);
        amax_partial[blockIdx.x] = m;
      }
    }
  }
}

// out[n] = sum over chunks of partial[chunk][n] (fixed order): 16 columns per block x 64 chunk lanes, so that a couple
// of thousand partial rows are a few dozen loads per thread, not a serial walk
__global__ __launch_bounds__(1024) void bn_sums_final(const float* partial, int chunks, int n, float* out) {
  __shared__ float red[64][17];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + tx;
  float acc = 0.f;
  if (c < n)
    for (int j = ty; j < chunks; j += 64) acc += partial[(int64_t)j * n + c];
  red[ty][tx] = acc;
  __syncthreads();
  if (ty == 0 && c < n) {
    float t = 0.f;
#pragma unroll 8
    for (int l = 0; l < 64; ++l) t += red[l][tx];
    out[c] = t;
  }
}

struct Whiten { double p, q, w; };

__device__ __forceinline__ Whiten inv_sqrt_2x2(double a, double b, double d) {
  const double s = sqrt(a * d - b * b);
  const double t = s * sqrt(a + d + 2.0 * s);
  return Whiten{(d + s) / t, -b / t, (a + s) / t};
}

// Cross-rank statistics (the optional SyncBN row): the moment totals of one rank leave the library
// as [F][NS] doubles (`moments_out`), the caller sums them over the ranks (one all-reduce of F * NS + 1 doubles,
// the last one the position count) and the finalize + apply passes run from the summed totals (`moments_in`,
// `count_dev` on the device: no host round trip).  Backward: the affine parameters' gradients stay LOCAL sums
// (`local_in`; the data-parallel gradient exchange averages them like every other parameter gradient), the input
// gradient uses the totals of all ranks.
struct BnSync {
  long long* tracked_inc = nullptr;      // forward: *tracked_inc += 1 in the finalize launch (num_batches_tracked)
  double* moments_out = nullptr;
  const double* moments_in = nullptr;
  int moments_chunks = 1;                // rows of moments_in ([row][F][NS]): 1 = totals
  const double* local_in = nullptr;
  const double* count_dev = nullptr;
  float* coef_out = nullptr;             // backward: leave the apply coefficients here and do NOT apply (cplxamd_bn_bwd_coef)
  float* dx_sums_out = nullptr;          // ... and the per-feature sums of dX, [2][F], from the sums of the pass
  float* amax_partial = nullptr;         // backward rows path with dx_sums: per-block max |dX| (kBnAmaxSlots floats, zeroed by the caller)
};

template <int NS>
__global__ __launch_bounds__(64) void bn_collapse(const double* partial, int chunks, int F, double* out) {
  const int f = blockIdx.x;
  double s[NS];
  for (int j = 0; j < NS; ++j) s[j] = 0.0;
  for (int c = threadIdx.x; c < chunks; c += 64)
    for (int j = 0; j < NS; ++j) s[j] += partial[((int64_t)c * F + f) * NS + j];
  for (int j = 0; j < NS; ++j) s[j] = wave_sum(s[j]);
  if (threadIdx.x == 0)
    for (int j = 0; j < NS; ++j) out[(int64_t)f * NS + j] = s[j];
}

// forward finalize: one wave per feature (lanes stride over the chunk partials, wave reduction,
// lane 0 does the per-feature algebra)
__global__ __launch_bounds__(64) void bn_fwd_finalize(const double* partial, int chunks, int F, double count,
                                const float* weight, const float* bias, float* running_mean,
                                float* running_var, int training, float momentum, float eps,
                                float* saved, float* coef, const double* count_dev, long long* tracked_inc) {
  const int f = blockIdx.x;
  if (tracked_inc && f == 0 && threadIdx.x == 0) *tracked_inc += 1;
  if (count_dev) count = *count_dev;
  double mu, mv, vuu, vuv, vvv;
  double s[5] = {0, 0, 0, 0, 0};
  if (training) {
    for (int c = threadIdx.x; c < chunks; c += 64)
      for (int j = 0; j < 5; ++j) s[j] += partial[((int64_t)c * F + f) * 5 + j];
    for (int j = 0; j < 5; ++j) s[j] = wave_sum(s[j]);
  }
  if (threadIdx.x != 0) return;
  if (training) {
    mu = s[0] / count; mv = s[1] / count;
    vuu = s[2] / count - mu * mu + (double)eps;
    vvv = s[3] / count - mv * mv + (double)eps;
    vuv = s[4] / count - mu * mv;
    if (running_mean) {
      // batchnorm.py:73,99  x += momentum * (new - x), in float32 like the reference
      const float fm[2] = {(float)mu, (float)mv};
      const float fc[4] = {(float)vuu, (float)vuv, (float)vuv, (float)vvv};
      for (int j = 0; j < 2; ++j) running_mean[j * F + f] += momentum * (fm[j] - running_mean[j * F + f]);
      for (int j = 0; j < 4; ++j) running_var[j * F + f] += momentum * (fc[j] - running_var[j * F + f]);
    }
  } else {
    mu = running_mean[f]; mv = running_mean[F + f];
    vuu = running_var[f]; vuv = running_var[F + f]; vvv = running_var[3 * F + f];
  }
  const Whiten r = inv_sqrt_2x2(vuu, vuv, vvv);
  saved[0 * F + f] = (float)mu; saved[1 * F + f] = (float)mv;
  saved[2 * F + f] = (float)r.p; saved[3 * F + f] = (float)r.q; saved[4 * F + f] = (float)r.w;
  saved[5 * F + f] = (float)vuu; saved[6 * F + f] = (float)vuv; saved[7 * F + f] = (float)vvv;
  double w00 = 1, w01 = 0, w10 = 0, w11 = 1, b0 = 0, b1 = 0;
  if (weight) {
    w00 = weight[f]; w01 = weight[F + f]; w10 = weight[2 * F + f]; w11 = weight[3 * F + f];
    b0 = bias[f]; b1 = bias[F + f];
  }
  // zu = p cu + q cv ; zv = q cu + w cv ; out = W z + b
  float* c = coef + (int64_t)f * kFwdCoef;
  c[0] = (float)mu; c[1] = (float)mv;
  c[2] = (float)(w00 * r.p + w01 * r.q); c[3] = (float)(w00 * r.q + w01 * r.w);
  c[4] = (float)(w10 * r.p + w11 * r.q); c[5] = (float)(w10 * r.q + w11 * r.w);
  c[6] = (float)b0; c[7] = (float)b1;
}

__global__ __launch_bounds__(64) void bn_bwd_finalize(const double* partial, int chunks, int F, double count,
                                const float* weight, const float* saved, int training,
                                float* dweight, float* dbias, float* coef, const double* local,
                                const double* count_dev, float* dx_sums = nullptr) {
  const int f = blockIdx.x;
  if (count_dev) count = *count_dev;
  double s[6] = {0, 0, 0, 0, 0, 0};
  for (int c = threadIdx.x; c < chunks; c += 64)
    for (int j = 0; j < 6; ++j) s[j] += partial[((int64_t)c * F + f) * 6 + j];
  for (int j = 0; j < 6; ++j) s[j] = wave_sum(s[j]);
  if (threadIdx.x != 0) return;
  const double Sgu = s[0], Sgv = s[1], Suu = s[2], Suv = s[3], Svu = s[4], Svv = s[5];
  const double p = saved[2 * F + f], q = saved[3 * F + f], w = saved[4 * F + f];
  double w00 = 1, w01 = 0, w10 = 0, w11 = 1;
  if (weight) {
    w00 = weight[f]; w01 = weight[F + f]; w10 = weight[2 * F + f]; w11 = weight[3 * F + f];
  }
  {
    // parameter gradients: this rank's sums when the statistics are shared between ranks (BnSync::local_in)
    const double* l = local ? local + (int64_t)f * 6 : s;
    if (dweight) {
      dweight[0 * F + f] = (float)(p * l[2] + q * l[3]);   // sum gou zu
      dweight[1 * F + f] = (float)(q * l[2] + w * l[3]);   // sum gou zv
      dweight[2 * F + f] = (float)(p * l[4] + q * l[5]);   // sum gov zu
      dweight[3 * F + f] = (float)(q * l[4] + w * l[5]);   // sum gov zv
    }
    if (dbias) { dbias[f] = (float)l[0]; dbias[F + f] = (float)l[1]; }
  }
  float* c = coef + (int64_t)f * kBwdCoef;
  c[0] = saved[f]; c[1] = saved[F + f];
  // gzu = w00 gou + w10 gov ; gzv = w01 gou + w11 gov ; gu = p gzu + q gzv + ... ; gv = q gzu + w gzv + ...
  c[2] = (float)(p * w00 + q * w01); c[3] = (float)(p * w10 + q * w11);
  c[4] = (float)(q * w00 + w * w01); c[5] = (float)(q * w10 + w * w11);
  double cuu = 0, cuv = 0, cvv = 0, ku = 0, kv = 0;
  if (training) {
    const double a = saved[5 * F + f], b = saved[6 * F + f], d = saved[7 * F + f];
    const double sgzu = w00 * Sgu + w10 * Sgv, sgzv = w01 * Sgu + w11 * Sgv;
    const double gp = w00 * Suu + w10 * Svu, gr_ = w00 * Suv + w10 * Svv;
    const double gq = w01 * Suu + w11 * Svu, gw = w01 * Suv + w11 * Svv;
    const double gqr = gq + gr_;
    const double sd = sqrt(a * d - b * b), tau = a + d + 2.0 * sd, rt = sqrt(tau), t = sd * rt;
    const double dsv[3] = {d / (2.0 * sd), a / (2.0 * sd), -b / sd};   // d s / d(a, d, b)
    double gcov[3];
    for (int X = 0; X < 3; ++X) {
      const double dtau = (X == 2 ? 0.0 : 1.0) + 2.0 * dsv[X];
      const double dt = dsv[X] * rt + sd * dtau / (2.0 * rt);
      const double dp = (((X == 1 ? 1.0 : 0.0) + dsv[X]) * t - (d + sd) * dt) / (t * t);
      const double dw = (((X == 0 ? 1.0 : 0.0) + dsv[X]) * t - (a + sd) * dt) / (t * t);
      const double dq = (-(X == 2 ? 1.0 : 0.0) * t + b * dt) / (t * t);
      gcov[X] = gp * dp + gw * dw + gqr * dq;
    }
    cuu = 2.0 * gcov[0] / count; cvv = 2.0 * gcov[1] / count; cuv = gcov[2] / count;
    ku = (p * sgzu + q * sgzv) / count; kv = (q * sgzu + w * sgzv) / count;
  }
  c[6] = (float)cuu; c[7] = (float)cuv; c[8] = (float)cvv; c[9] = (float)ku; c[10] = (float)kv;
  c[11] = 0.f;
  if (dx_sums) {
    // sum over rows of dX = E sum(g) + C sum(x - mu) - count k; sum(x - mu) = 0 against the batch mean, C = 0 otherwise
    dx_sums[f] = (float)((p * w00 + q * w01) * Sgu + (p * w10 + q * w11) * Sgv - count * ku);
    dx_sums[F + f] = (float)((q * w00 + w * w01) * Sgu + (q * w10 + w * w11) * Sgv - count * kv);
  }
}

// apply, S > 1: grid (ceil(S/(4*256)) , B*F)
template <typename T, bool BWD>
__global__ __launch_bounds__(kBnT) void bn_apply_planes(const T* xr, const T* xi, const T* gr,
                                                        const T* gi, T* yr, T* yi,
                                                        const float* coef, int F, int64_t S) {
  const int f = blockIdx.y % F;
  const float* c = coef + (int64_t)f * (BWD ? kBwdCoef : kFwdCoef);
  const float mu = c[0], mv = c[1], a00 = c[2], a01 = c[3], a10 = c[4], a11 = c[5];
  const float k6 = c[6], k7 = c[7];
  float cvv = 0.f, ku = 0.f, kv = 0.f;
  if (BWD) { cvv = c[8]; ku = c[9]; kv = c[10]; }
  const int64_t base = (int64_t)blockIdx.y * S;
  if ((S & 3) == 0) {
    const int64_t s = ((int64_t)blockIdx.x * kBnT + threadIdx.x) * 4;
    if (s >= S) return;
    const f4 u = ld4(xr + base + s), v = ld4(xi + base + s);
    f4 ou, ov;
    if (!BWD) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float cu = u.v[j] - mu, cv = v.v[j] - mv;
        ou.v[j] = fmaf(a00, cu, fmaf(a01, cv, k6));
        ov.v[j] = fmaf(a10, cu, fmaf(a11, cv, k7));
      }
    } else {
      const f4 p = ld4(gr + base + s), q = ld4(gi + base + s);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float cu = u.v[j] - mu, cv = v.v[j] - mv;
        ou.v[j] = a00 * p.v[j] + a01 * q.v[j] + k6 * cu + k7 * cv - ku;
        ov.v[j] = a10 * p.v[j] + a11 * q.v[j] + cvv * cv + k7 * cu - kv;
      }
    }
    st4(yr + base + s, ou);
    st4(yi + base + s, ov);
  } else {
    for (int j = 0; j < 4; ++j) {
      const int64_t s = ((int64_t)blockIdx.x * 4 + j) * kBnT + threadIdx.x;
      if (s >= S) return;
      const float cu = io<T>::ld(xr + base + s) - mu, cv = io<T>::ld(xi + base + s) - mv;
      float ou, ov;
      if (!BWD) {
        ou = fmaf(a00, cu, fmaf(a01, cv, k6));
        ov = fmaf(a10, cu, fmaf(a11, cv, k7));
      } else {
        const float p = io<T>::ld(gr + base + s), q = io<T>::ld(gi + base + s);
        ou = a00 * p + a01 * q + k6 * cu + k7 * cv - ku;
        ov = a10 * p + a11 * q + cvv * cv + k7 * cu - kv;
      }
      io<T>::st(yr + base + s, ou);
      io<T>::st(yi + base + s, ov);
    }
  }
}

// apply, S == 1
template <typename T, bool BWD>
__global__ __launch_bounds__(kBnT) void bn_apply_cols(const T* xr, const T* xi, const T* gr,
                                                      const T* gi, T* yr, T* yi, const float* coef,
                                                      int F, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * kBnT;
  for (int64_t e = (int64_t)blockIdx.x * kBnT + threadIdx.x; e < n; e += stride) {
    const int f = (int)(e % F);
    const float* c = coef + (int64_t)f * (BWD ? kBwdCoef : kFwdCoef);
    const float cu = io<T>::ld(xr + e) - c[0], cv = io<T>::ld(xi + e) - c[1];
    float ou, ov;
    if (!BWD) {
      ou = fmaf(c[2], cu, fmaf(c[3], cv, c[6]));
      ov = fmaf(c[4], cu, fmaf(c[5], cv, c[7]));
    } else {
      const float p = io<T>::ld(gr + e), q = io<T>::ld(gi + e);
      ou = c[2] * p + c[3] * q + c[6] * cu + c[7] * cv - c[9];
      ov = c[4] * p + c[5] * q + c[8] * cv + c[7] * cu - c[10];
    }
    io<T>::st(yr + e, ou);
    io<T>::st(yi + e, ov);
  }
}

static bool bn_rows_ok(int64_t B, int F, int64_t S) { return S == 1 && F % 8 == 0 && F <= 1024 && B >= 4096; }
static int bn_max_chunks(int F) { return (F % 8 == 0 && F <= 1024) ? kBnRowChunks : kBnMaxChunks; }
static int64_t bn_coef_off(int F) { return (int64_t)bn_max_chunks(F) * F * 6 * sizeof(double); }
static int64_t bn_sums_off(int F) { return (bn_coef_off(F) + (int64_t)F * kBwdCoef * sizeof(float) + 255) / 256 * 256; }
static int64_t bn_ws_bytes(int F) {
  // [moment partials][coefficients][2048 x 2F partial sums of the row-kernel backward's dX (only F % 8 == 0, F <= 1024)]
  return bn_sums_off(F) + ((F % 8 == 0 && F <= 1024) ? (int64_t)2048 * 2 * F * sizeof(float) : 0);
}

template <typename T, bool BWD>
static int bn_run(const void* xr, const void* xi, const void* gr, const void* gi, void* yr,
                  void* yi, int64_t B, int F, int64_t S, const float* weight, const float* bias,
                  float* running_mean, float* running_var, float* saved, float* dweight,
                  float* dbias, int training, float momentum, float eps, void* ws,
                  hipStream_t st, float* dx_sums = nullptr, BnSync sync = BnSync()) {
  const BnGeom g = bn_geom(B, F, S);
  double* partial = (double*)ws;
  float* coef = sync.coef_out ? sync.coef_out : (float*)((char*)ws + bn_coef_off(F));
  const bool rows = bn_rows_ok(B, F, S);
  int chunks = g.chunks;
  if (rows) {
    const int RL = kBnT / (F / 8);
    const int64_t want = (B + 2 * RL - 1) / (2 * RL);
    chunks = (int)(want < kBnRowChunks ? want : kBnRowChunks);
  }
  constexpr int NS = BWD ? 6 : 5;
  const bool need_reduce = (BWD ? true : (training != 0)) && !sync.moments_in;
  if (need_reduce) {
    if (rows) {
      bn_reduce_rows<T, NS, BWD><<<chunks, kBnT, 0, st>>>((const T*)xr, (const T*)xi, (const T*)gr, (const T*)gi,
                                                         saved, B, F, partial);
    } else if (g.small) {
      dim3 grid(F, g.chunks);
      bn_reduce_small<T, NS, BWD><<<grid, kBnT, 0, st>>>((const T*)xr, (const T*)xi, (const T*)gr, (const T*)gi, saved,
                                                         (int)B, F, (int)S, partial);
    } else if (S > 1) {
      dim3 grid(F, g.chunks);
      bn_reduce_planes<T, NS, BWD><<<grid, kBnT, 0, st>>>((const T*)xr, (const T*)xi, (const T*)gr,
                                                          (const T*)gi, saved, g, partial);
    } else {
      dim3 grid((F + 63) / 64, g.chunks);
      bn_reduce_cols<T, NS, BWD><<<grid, kBnT, 0, st>>>((const T*)xr, (const T*)xi, (const T*)gr,
                                                        (const T*)gi, saved, g, partial);
    }
    CPLXAMD_CHECK_LAUNCH();
  }
  if (sync.moments_out) {                              // first half of a cross-rank pass: this rank's totals only
    bn_collapse<NS><<<F, 64, 0, st>>>(partial, chunks, F, sync.moments_out);
    CPLXAMD_CHECK_LAUNCH();
    return 0;
  }
  const double count = (double)B * (double)S;
  const double* totals = sync.moments_in ? sync.moments_in : partial;
  const int tchunks = sync.moments_in ? sync.moments_chunks : chunks;
  if (!BWD)
    bn_fwd_finalize<<<F, 64, 0, st>>>(totals, tchunks, F, count, weight, bias, running_mean,
                                        running_var, training, momentum, eps, saved, coef, sync.count_dev, sync.tracked_inc);
  else
    bn_bwd_finalize<<<F, 64, 0, st>>>(totals, tchunks, F, count, weight, saved, training,
                                        dweight, dbias, coef, sync.local_in, sync.count_dev, sync.dx_sums_out);
  CPLXAMD_CHECK_LAUNCH();
  if (BWD && sync.coef_out) return 0;                  // the consumer of the coefficients applies them (conv_cl_wgrad.hip FOLD)
  if (rows) {
    const int RL = kBnT / (F / 8);
    const int grid = stream_grid((B + RL - 1) / RL * kBnT, kBnT);
    if (BWD && dx_sums) {
      float* sp = (float*)((char*)ws + bn_sums_off(F));
      bn_apply_rows<T, BWD, true><<<grid, kBnT, 0, st>>>((const T*)xr, (const T*)xi, (const T*)gr, (const T*)gi, (T*)yr,
                                                        (T*)yi, coef, B, F, sp, grid <= kBnAmaxSlots ? sync.amax_partial : nullptr);
      CPLXAMD_CHECK_LAUNCH();
      bn_sums_final<<<(2 * F + 15) / 16, 1024, 0, st>>>(sp, grid, 2 * F, dx_sums);
    } else {
      bn_apply_rows<T, BWD, false><<<grid, kBnT, 0, st>>>((const T*)xr, (const T*)xi, (const T*)gr, (const T*)gi, (T*)yr,
                                                         (T*)yi, coef, B, F, nullptr);
    }
    CPLXAMD_CHECK_LAUNCH();
  } else if (S > 1) {
    const int64_t planes = B * F;
    if (planes > 0x7fffffff / 1) return CPLXAMD_ESHAPE;
    // gridDim.y is limited to 65535: fold planes over several launches of a whole number of batch entries
    // (F > 65535 features with spatial dims would make that step 0: not a shape this kernel serves)
    if (F > 65535) return CPLXAMD_ESHAPE;
    const unsigned gx = (unsigned)((S + 4 * kBnT - 1) / (4 * kBnT));
    for (int64_t p0 = 0; p0 < planes; p0 += 65535 - (65535 % F)) {
      int64_t np = planes - p0;
      const int64_t cap = 65535 - (65535 % F);
      if (np > cap) np = cap;
      dim3 grid(gx, (unsigned)np);
      const int64_t off = p0 * S;
      bn_apply_planes<T, BWD><<<grid, kBnT, 0, st>>>(
          (const T*)xr + off, (const T*)xi + off, gr ? (const T*)gr + off : nullptr,
          gi ? (const T*)gi + off : nullptr, (T*)yr + off, (T*)yi + off, coef, F, S);
      CPLXAMD_CHECK_LAUNCH();
    }
  } else {
    const int64_t n = B * F;
    bn_apply_cols<T, BWD><<<stream_grid(n, kBnT), kBnT, 0, st>>>(
        (const T*)xr, (const T*)xi, (const T*)gr, (const T*)gi, (T*)yr, (T*)yi, coef, F, n);
    CPLXAMD_CHECK_LAUNCH();
  }
  return 0;
}

}  // namespace cplxamd

using namespace cplxamd;

extern "C" {

int64_t cplxamd_bn_ws_bytes(int F) { return bn_ws_bytes(F); }

int cplxamd_bn_fwd_ex(const void* xr, const void* xi, void* yr, void* yi, int64_t B, int F,
                      int64_t S, const float* weight, const float* bias, float* running_mean,
                      float* running_var, float* saved, int training, int dtype, float momentum,
                      float eps, int64_t* tracked_inc, void* ws, int64_t ws_bytes, void* stream) {
  if (!xr || !xi || !yr || !yi || !saved || !ws || B <= 0 || F <= 0 || S <= 0) return CPLXAMD_EINVAL;
  if ((weight == nullptr) != (bias == nullptr)) return CPLXAMD_EINVAL;
  if ((running_mean == nullptr) != (running_var == nullptr)) return CPLXAMD_EINVAL;
  if (!training && !running_mean) return CPLXAMD_EINVAL;
  if (ws_bytes < bn_ws_bytes(F)) return CPLXAMD_EWS;
  hipStream_t st = (hipStream_t)stream;
  BnSync extra;
  extra.tracked_inc = reinterpret_cast<long long*>(tracked_inc);
  if (dtype == CPLXAMD_F32)
    return bn_run<float, false>(xr, xi, nullptr, nullptr, yr, yi, B, F, S, weight, bias,
                                running_mean, running_var, saved, nullptr, nullptr, training,
                                momentum, eps, ws, st, nullptr, extra);
  if (dtype == CPLXAMD_BF16)
    return bn_run<bf16_t, false>(xr, xi, nullptr, nullptr, yr, yi, B, F, S, weight, bias,
                                 running_mean, running_var, saved, nullptr, nullptr, training,
                                 momentum, eps, ws, st, nullptr, extra);
  return CPLXAMD_EINVAL;
}

int cplxamd_bn_fwd(const void* xr, const void* xi, void* yr, void* yi, int64_t B, int F,
                   int64_t S, const float* weight, const float* bias, float* running_mean,
                   float* running_var, float* saved, int training, int dtype, float momentum,
                   float eps, void* ws, int64_t ws_bytes, void* stream) {
  return cplxamd_bn_fwd_ex(xr, xi, yr, yi, B, F, S, weight, bias, running_mean, running_var, saved, training, dtype, momentum,
                           eps, nullptr, ws, ws_bytes, stream);
}

int cplxamd_bn_rows_path(int64_t B, int F, int64_t S) { return bn_rows_ok(B, F, S) ? 1 : 0; }

int cplxamd_bn_bwd_sums(const void* gr, const void* gi, const void* xr, const void* xi, void* dxr,
                        void* dxi, int64_t B, int F, int64_t S, const float* weight,
                        const float* saved, float* dweight, float* dbias, int training, int dtype,
                        float* dx_sums, void* ws, int64_t ws_bytes, void* stream);

int cplxamd_bn_bwd(const void* gr, const void* gi, const void* xr, const void* xi, void* dxr,
                   void* dxi, int64_t B, int F, int64_t S, const float* weight,
                   const float* saved, float* dweight, float* dbias, int training, int dtype,
                   void* ws, int64_t ws_bytes, void* stream) {
  return cplxamd_bn_bwd_sums(gr, gi, xr, xi, dxr, dxi, B, F, S, weight, saved, dweight, dbias, training, dtype, nullptr,
                             ws, ws_bytes, stream);
}

int cplxamd_bn_bwd_sums(const void* gr, const void* gi, const void* xr, const void* xi, void* dxr,
                        void* dxi, int64_t B, int F, int64_t S, const float* weight,
                        const float* saved, float* dweight, float* dbias, int training, int dtype,
                        float* dx_sums, void* ws, int64_t ws_bytes, void* stream) {
  if (dx_sums && !bn_rows_ok(B, F, S)) return CPLXAMD_ESHAPE;
  if (!gr || !gi || !xr || !xi || !dxr || !dxi || !saved || !ws || B <= 0 || F <= 0 || S <= 0)
    return CPLXAMD_EINVAL;
  if (ws_bytes < bn_ws_bytes(F)) return CPLXAMD_EWS;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CPLXAMD_F32)
    return bn_run<float, true>(xr, xi, gr, gi, dxr, dxi, B, F, S, weight, nullptr, nullptr,
                               nullptr, const_cast<float*>(saved), dweight, dbias, training, 0.f,
                               0.f, ws, st, dx_sums);
  if (dtype == CPLXAMD_BF16)
    return bn_run<bf16_t, true>(xr, xi, gr, gi, dxr, dxi, B, F, S, weight, nullptr, nullptr,
                                nullptr, const_cast<float*>(saved), dweight, dbias, training, 0.f,
                                0.f, ws, st, dx_sums);
  return CPLXAMD_EINVAL;
}

// cplxamd_bn_bwd_sums that also leaves the per-block maxima of |dX| (both planes, as stored) in amax_partial[2048] (float32,
// ZEROED by the caller): cplxamd_absmax_scale_partials turns them into the power-of-two scale a consumer cutting dX into
// half pieces needs -- a pass over both planes less.  Row-kernel path only (as dx_sums).
int cplxamd_bn_bwd_sums_amax(const void* gr, const void* gi, const void* xr, const void* xi, void* dxr, void* dxi, int64_t B,
                             int F, int64_t S, const float* weight, const float* saved, float* dweight, float* dbias,
                             int training, int dtype, float* dx_sums, float* amax_partial, void* ws, int64_t ws_bytes,
                             void* stream) {
  if (!dx_sums || !amax_partial || !bn_rows_ok(B, F, S)) return CPLXAMD_ESHAPE;
  if (!gr || !gi || !xr || !xi || !dxr || !dxi || !saved || !ws || B <= 0 || F <= 0 || S <= 0) return CPLXAMD_EINVAL;
  if (ws_bytes < bn_ws_bytes(F)) return CPLXAMD_EWS;
  hipStream_t st = (hipStream_t)stream;
  BnSync sync;
  sync.amax_partial = amax_partial;
  if (dtype == CPLXAMD_F32)
    return bn_run<float, true>(xr, xi, gr, gi, dxr, dxi, B, F, S, weight, nullptr, nullptr, nullptr,
                               const_cast<float*>(saved), dweight, dbias, training, 0.f, 0.f, ws, st, dx_sums, sync);
  if (dtype == CPLXAMD_BF16)
    return bn_run<bf16_t, true>(xr, xi, gr, gi, dxr, dxi, B, F, S, weight, nullptr, nullptr, nullptr,
                                const_cast<float*>(saved), dweight, dbias, training, 0.f, 0.f, ws, st, dx_sums, sync);
  return CPLXAMD_EINVAL;
}

// The first half of cplxamd_bn_bwd: sums + finalize (dweight, dbias as there) and the per-channel coefficients of the
// input gradient, dX = E g + C (x - mu) - k, left in coef[F][12] (float32: mu mv | e00 e01 e10 e11 | cuu cuv cvv | ku kv |
// pad) for a consumer that applies them itself -- cplxamd_conv2d_cl_wgrad_bn_fl forms dX while it stages it.
int cplxamd_bn_bwd_coef(const void* gr, const void* gi, const void* xr, const void* xi, int64_t B, int F, int64_t S,
                        const float* weight, const float* saved, float* dweight, float* dbias, int training, int dtype,
                        float* coef, float* dx_sums, void* ws, int64_t ws_bytes, void* stream) {
  if (!gr || !gi || !xr || !xi || !saved || !coef || !ws || B <= 0 || F <= 0 || S <= 0) return CPLXAMD_EINVAL;
  if (ws_bytes < bn_ws_bytes(F)) return CPLXAMD_EWS;
  hipStream_t st = (hipStream_t)stream;
  BnSync sync;
  sync.coef_out = coef;
  sync.dx_sums_out = dx_sums;
  if (dtype == CPLXAMD_F32)
    return bn_run<float, true>(xr, xi, gr, gi, nullptr, nullptr, B, F, S, weight, nullptr, nullptr, nullptr,
                               const_cast<float*>(saved), dweight, dbias, training, 0.f, 0.f, ws, st, nullptr, sync);
  if (dtype == CPLXAMD_BF16)
    return bn_run<bf16_t, true>(xr, xi, gr, gi, nullptr, nullptr, B, F, S, weight, nullptr, nullptr, nullptr,
                                const_cast<float*>(saved), dweight, dbias, training, 0.f, 0.f, ws, st, nullptr, sync);
  return CPLXAMD_EINVAL;
}

/* ---- statistics shared between ranks (see BnSync): moments -> [caller: all-reduce] -> *_sync ------------------- */
int cplxamd_bn_moments(const void* xr, const void* xi, const void* gr, const void* gi, const float* saved,
                       int64_t B, int F, int64_t S, int dtype, double* moments, void* ws, int64_t ws_bytes,
                       void* stream) {
  if (!xr || !xi || !moments || !ws || B <= 0 || F <= 0 || S <= 0) return CPLXAMD_EINVAL;
  if ((gr == nullptr) != (gi == nullptr) || (gr && !saved)) return CPLXAMD_EINVAL;
  if (ws_bytes < bn_ws_bytes(F)) return CPLXAMD_EWS;
  hipStream_t st = (hipStream_t)stream;
  BnSync sync;
  sync.moments_out = moments;
  float* sv = const_cast<float*>(saved);
  if (dtype == CPLXAMD_F32)
    return gr ? bn_run<float, true>(xr, xi, gr, gi, nullptr, nullptr, B, F, S, nullptr, nullptr, nullptr, nullptr, sv,
                                    nullptr, nullptr, 1, 0.f, 0.f, ws, st, nullptr, sync)
              : bn_run<float, false>(xr, xi, nullptr, nullptr, nullptr, nullptr, B, F, S, nullptr, nullptr, nullptr,
                                     nullptr, sv, nullptr, nullptr, 1, 0.f, 0.f, ws, st, nullptr, sync);
  if (dtype == CPLXAMD_BF16)
    return gr ? bn_run<bf16_t, true>(xr, xi, gr, gi, nullptr, nullptr, B, F, S, nullptr, nullptr, nullptr, nullptr, sv,
                                     nullptr, nullptr, 1, 0.f, 0.f, ws, st, nullptr, sync)
              : bn_run<bf16_t, false>(xr, xi, nullptr, nullptr, nullptr, nullptr, B, F, S, nullptr, nullptr, nullptr,
                                      nullptr, sv, nullptr, nullptr, 1, 0.f, 0.f, ws, st, nullptr, sync);
  return CPLXAMD_EINVAL;
}

int cplxamd_bn_fwd_sync(const void* xr, const void* xi, void* yr, void* yi, int64_t B, int F, int64_t S,
                        const float* weight, const float* bias, float* running_mean, float* running_var,
                        float* saved, int dtype, float momentum, float eps, const double* moments,
                        const double* count, void* ws, int64_t ws_bytes, void* stream) {
  if (!xr || !xi || !yr || !yi || !saved || !ws || !moments || !count || B <= 0 || F <= 0 || S <= 0)
    return CPLXAMD_EINVAL;
  if ((weight == nullptr) != (bias == nullptr)) return CPLXAMD_EINVAL;
  if ((running_mean == nullptr) != (running_var == nullptr)) return CPLXAMD_EINVAL;
  if (ws_bytes < bn_ws_bytes(F)) return CPLXAMD_EWS;
  hipStream_t st = (hipStream_t)stream;
  BnSync sync;
  sync.moments_in = moments;
  sync.count_dev = count;
  if (dtype == CPLXAMD_F32)
    return bn_run<float, false>(xr, xi, nullptr, nullptr, yr, yi, B, F, S, weight, bias, running_mean, running_var,
                                saved, nullptr, nullptr, 1, momentum, eps, ws, st, nullptr, sync);
  if (dtype == CPLXAMD_BF16)
    return bn_run<bf16_t, false>(xr, xi, nullptr, nullptr, yr, yi, B, F, S, weight, bias, running_mean, running_var,
                                 saved, nullptr, nullptr, 1, momentum, eps, ws, st, nullptr, sync);
  return CPLXAMD_EINVAL;
}

// Training-mode forward whose statistics pass has been done by the PRODUCER of x: `partials` holds `chunks` rows
// [row][F][5] float64 of (sum re, sum im, sum re^2, sum im^2, sum re im) over disjoint parts of the batch (the layout of
// this file's own moment kernels; cplxamd_conv2d_cl2_mom writes one row per workgroup).  Finalize + apply only: x is read
// once.  Arguments as cplxamd_bn_fwd_ex with training = 1.
int cplxamd_bn_fwd_partials(const void* xr, const void* xi, void* yr, void* yi, int64_t B, int F, int64_t S,
                            const float* weight, const float* bias, float* running_mean, float* running_var,
                            float* saved, int dtype, float momentum, float eps, int64_t* tracked_inc,
                            const double* partials, int chunks, void* ws, int64_t ws_bytes, void* stream) {
  if (!xr || !xi || !yr || !yi || !saved || !ws || !partials || chunks <= 0 || B <= 0 || F <= 0 || S <= 0)
    return CPLXAMD_EINVAL;
  if ((weight == nullptr) != (bias == nullptr)) return CPLXAMD_EINVAL;
  if ((running_mean == nullptr) != (running_var == nullptr)) return CPLXAMD_EINVAL;
  if (ws_bytes < bn_ws_bytes(F)) return CPLXAMD_EWS;
  hipStream_t st = (hipStream_t)stream;
  BnSync sync;
  sync.moments_in = partials;
  sync.moments_chunks = chunks;
  sync.tracked_inc = reinterpret_cast<long long*>(tracked_inc);
  if (dtype == CPLXAMD_F32)
    return bn_run<float, false>(xr, xi, nullptr, nullptr, yr, yi, B, F, S, weight, bias, running_mean, running_var,
                                saved, nullptr, nullptr, 1, momentum, eps, ws, st, nullptr, sync);
  if (dtype == CPLXAMD_BF16)
    return bn_run<bf16_t, false>(xr, xi, nullptr, nullptr, yr, yi, B, F, S, weight, bias, running_mean, running_var,
                                 saved, nullptr, nullptr, 1, momentum, eps, ws, st, nullptr, sync);
  return CPLXAMD_EINVAL;
}

int cplxamd_bn_bwd_sync(const void* gr, const void* gi, const void* xr, const void* xi, void* dxr, void* dxi,
                        int64_t B, int F, int64_t S, const float* weight, const float* saved, float* dweight,
                        float* dbias, int dtype, float* dx_sums, const double* moments, const double* local_moments,
                        const double* count, void* ws, int64_t ws_bytes, void* stream) {
  if (dx_sums && !bn_rows_ok(B, F, S)) return CPLXAMD_ESHAPE;
  if (!gr || !gi || !xr || !xi || !dxr || !dxi || !saved || !ws || !moments || !local_moments || !count || B <= 0 ||
      F <= 0 || S <= 0)
    return CPLXAMD_EINVAL;
  if (ws_bytes < bn_ws_bytes(F)) return CPLXAMD_EWS;
  hipStream_t st = (hipStream_t)stream;
  BnSync sync;
  sync.moments_in = moments;
  sync.local_in = local_moments;
  sync.count_dev = count;
  float* sv = const_cast<float*>(saved);
  if (dtype == CPLXAMD_F32)
    return bn_run<float, true>(xr, xi, gr, gi, dxr, dxi, B, F, S, weight, nullptr, nullptr, nullptr, sv, dweight, dbias,
                               1, 0.f, 0.f, ws, st, dx_sums, sync);
  if (dtype == CPLXAMD_BF16)
    return bn_run<bf16_t, true>(xr, xi, gr, gi, dxr, dxi, B, F, S, weight, nullptr, nullptr, nullptr, sv, dweight, dbias,
                                1, 0.f, 0.f, ws, st, dx_sums, sync);
  return CPLXAMD_EINVAL;
}

}  // extern "C"
