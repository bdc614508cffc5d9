// Shared device helpers for the cplxamd kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/cplxamd.h"

#define CPLXAMD_CHECK_LAUNCH()                          \
  do {                                                  \
    hipError_t e__ = hipGetLastError();                 \
    if (e__ != hipSuccess) return (int)e__;             \
  } while (0)

// The 16-bit matrix instruction of a translation unit.  The MFMA GEMM and channels-last convolution sources are compiled
// twice: as they are (bf16 operands) and, from gemm_f16*.hip / conv_cl*_f16.hip with CPLXAMD_GEMM_F16 / CPLXAMD_CONV_F16
// defined, for IEEE-half operands -- same staging, same LDS images (a 16-bit pattern is a 16-bit pattern; the conjugate's
// sign flip is bit 15 in both), only the matrix instruction differs.  The half variants write float32 (the fp16 split
// products of cplxmodule_amd/x3.py).
#if defined(CPLXAMD_GEMM_F16) || defined(CPLXAMD_CONV_F16)
typedef _Float16 cplxamd_f16x8 __attribute__((ext_vector_type(8)));
#define CPLXAMD_MFMA16(a, b, c) \
  __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(cplxamd_f16x8, a), __builtin_bit_cast(cplxamd_f16x8, b), c, 0, 0, 0)
#else
#define CPLXAMD_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#endif

namespace cplxamd {

constexpr int kWave = 64;

// Rounds of the Philox4x32 counter-based generator behind the in-kernel noise (reparam.hip, layout.hip;
// numpy statement + Random123 known answers: oracle/philox.py).  7 is the smallest round count for which
// Philox4x32 passes BigCrush (Salmon et al., SC'11, "Parallel random numbers: as easy as 1, 2, 3", table 2);
// 10 is Random123's default safety margin.  The bf16 noise-injection kernels are bound by the generator's
// quarter-rate 32x32->64 multiplies, not by HBM, so the margin costs throughput directly
// (profiles/r02_reparam_rounds.txt).
#ifndef CPLXAMD_PHILOX_ROUNDS
#define CPLXAMD_PHILOX_ROUNDS 7
#endif
constexpr int kPhiloxRounds = CPLXAMD_PHILOX_ROUNDS;

typedef uint16_t bf16_t;  // raw bf16 bits

// floats per channel of the batch-norm backward coefficients (bn.hip bn_bwd_finalize; read by conv_cl_wgrad.hip's FOLD kernel):
// mu, mv, e00, e01, e10, e11, cuu, cuv, cvv, ku, kv, pad
constexpr int kBnBwdCoef = 12;

__device__ __forceinline__ float bf16_to_f32(bf16_t v) {
  return __uint_as_float(((uint32_t)v) << 16);
}

// round-to-nearest-even, NaN stays NaN (same rule as torch's float -> bfloat16): the gfx950 hardware
// conversion v_cvt_pk_bf16_f32 (one instruction per PAIR; the integer emulation it replaces was ~5 VALU ops
// per value, which showed in the VALU-bound bf16 noise-injection kernels)
typedef __bf16 bf16x2_hw __attribute__((ext_vector_type(2)));
typedef float f32x2_hw __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
  const f32x2_hw v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_hw));
}
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  return __builtin_bit_cast(bf16_t, (__bf16)f);
}

template <typename T> struct io;
template <> struct io<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct io<bf16_t> {
  static __device__ __forceinline__ float ld(const bf16_t* p) { return bf16_to_f32(*p); }
  static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

// One LDS-DMA instruction (global_load_lds_dwordx4): every lane moves the 16 bytes at its global
// address straight into LDS at  lds_wave_base + lane * 16  (no VGPR round trip).
// Issued as inline asm ON PURPOSE: the compiler treats the builtin as an LDS store that may alias
// every later ds_read and inserts `s_waitcnt vmcnt(0)` in front of those reads, which silently
// turns a multi-stage ring into a fully synchronous copy (found in the ISA of the round-1 kernels:
// a vmcnt(0) right behind every counted wait).  With the asm form the compiler does not track the
// transfer at all; the kernels order LDS-DMA writes against reads themselves with counted
// `s_waitcnt vmcnt(N)` + `s_barrier`.  M0 carries the wave-uniform LDS byte address (the low 32
// bits of a flat LDS pointer are the LDS offset); the compiler itself never uses M0 on gfx9+.
__device__ __forceinline__ void lds_dma16(const void* gsrc, const void* lds_wave_base) {
  const uint32_t m0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds_wave_base);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"
               :
               : "s"(m0), "v"(gsrc)
               : "memory");
}

// Same, with the LDS destination given as a wave-uniform BYTE OFFSET that the caller built from
// scalar values (lds_offset_of() once per kernel + integer arithmetic on uniform values): the M0
// set-up then stays on the scalar unit -- a generic LDS pointer costs two VALU ops and a
// v_readfirstlane per transfer.
__device__ __forceinline__ uint32_t lds_offset_of(const void* lds_ptr) {
  return __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds_ptr);
}
__device__ __forceinline__ void lds_dma16_at(const void* gsrc, uint32_t lds_off_uniform) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"
               :
               : "s"(lds_off_uniform), "v"(gsrc)
               : "memory");
}

// Scalar-base form: global address = sbase (wave-uniform, SGPR pair) + voff (32-bit per-lane byte
// offset).  The per-lane part of a tile piece does not depend on the K position, so it is computed
// once per kernel; advancing along K is scalar arithmetic on sbase -- no VALU work per transfer.
__device__ __forceinline__ void lds_dma16_sv(const void* sbase_uniform, uint32_t voff,
                                             uint32_t lds_off_uniform) {
  // readfirstlane pins the base to scalar registers even where the compiler's divergence analysis
  // gives up.  (The builtin returns a SIGNED int: widen through uint32_t, or the low half
  // sign-extends into the high one -- that was a GPU memory fault in the first version.)
  const uint64_t a = (uint64_t)(uintptr_t)sbase_uniform;
  const uint64_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a);
  const uint64_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32));
  const uint64_t sb = (hi << 32) | lo;
  // (readfirstlane: a no-op for a value that already lives in an SGPR; a loop-carried ring position may not)
  const uint32_t m0v = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_off_uniform);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
               :
               : "s"(m0v), "v"(voff), "s"(sb)
               : "memory");
}

// 4-wide vector access (16 B for float, 8 B for bf16)
struct f4 { float v[4]; };
// two f4 halves = 8 elements: one 16-B access for bf16, two for float (needs 16-B alignment of p)
struct f8 { f4 h[2]; };
__device__ __forceinline__ f4 ld4(const float* p) {
  float4 t = *reinterpret_cast<const float4*>(p);
  return f4{{t.x, t.y, t.z, t.w}};
}
__device__ __forceinline__ void st4(float* p, const f4& a) {
  *reinterpret_cast<float4*>(p) = make_float4(a.v[0], a.v[1], a.v[2], a.v[3]);
}
__device__ __forceinline__ f4 ld4(const bf16_t* p) {
  uint2 t = *reinterpret_cast<const uint2*>(p);
  return f4{{__uint_as_float(t.x << 16), __uint_as_float(t.x & 0xffff0000u),
             __uint_as_float(t.y << 16), __uint_as_float(t.y & 0xffff0000u)}};
}
__device__ __forceinline__ void st4(bf16_t* p, const f4& a) {
  uint2 t;
  t.x = pack_bf16(a.v[0], a.v[1]);
  t.y = pack_bf16(a.v[2], a.v[3]);
  *reinterpret_cast<uint2*>(p) = t;
}
__device__ __forceinline__ f8 ld8(const float* p) { return f8{{ld4(p), ld4(p + 4)}}; }
__device__ __forceinline__ void st8(float* p, const f8& a) { st4(p, a.h[0]); st4(p + 4, a.h[1]); }
__device__ __forceinline__ f8 ld8(const bf16_t* p) {
  const uint4 t = *reinterpret_cast<const uint4*>(p);
  const uint32_t w[4] = {t.x, t.y, t.z, t.w};
  f8 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    r.h[e >> 1].v[(e & 1) * 2] = __uint_as_float(w[e] << 16);
    r.h[e >> 1].v[(e & 1) * 2 + 1] = __uint_as_float(w[e] & 0xffff0000u);
  }
  return r;
}
__device__ __forceinline__ void st8(bf16_t* p, const f8& a) {
  uint32_t w[4];
#pragma unroll
  for (int e = 0; e < 4; ++e)
    w[e] = pack_bf16(a.h[e >> 1].v[(e & 1) * 2], a.h[e >> 1].v[(e & 1) * 2 + 1]);
  *reinterpret_cast<uint4*>(p) = uint4{w[0], w[1], w[2], w[3]};
}

// ---- correctly rounded float32 primitives ---------------------------------------------------
// HIP's __fsqrt_rn is the NATIVE (1 ulp) square root and __fmul_rn / __fadd_rn are plain
// operators (clang __clang_hip_math.h); sqrtf() IS correctly rounded under hipcc's default
// -fhip-fp32-correctly-rounded-divide-sqrt.  Files that need one rounding per operation also
// carry `#pragma clang fp contract(off)`.
__device__ __forceinline__ float rn_sqrt(float x) { return sqrtf(x); }

// log evaluated in fp64 and rounded once (= correctly rounded float32 log).  The empty asm
// hides the float origin of `d`, otherwise LLVM shrinks (float)log((double)x) to logf(x).
__device__ __forceinline__ float exact_logf(float x) {
  double d = (double)x;
  asm volatile("" : "+v"(d));
  return (float)log(d);
}

// ---- wave / block reductions (wave = 64 lanes) ---------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Sum over a block of NT threads; result valid in thread 0.  `smem` holds NT/64 T's.
template <typename T, int NT>
__device__ __forceinline__ T block_sum(T v, T* smem) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) smem[wid] = v;
  __syncthreads();
  T r = T(0);
  if (threadIdx.x < NT / 64) r = smem[threadIdx.x];
  if (wid == 0) r = wave_sum(r);
  __syncthreads();
  return r;
}

// grid for HBM-bound streaming kernels: cap at 256 CUs x 8 blocks, grid-stride the rest
inline int stream_grid(int64_t work_items, int block) {
  int64_t g = (work_items + block - 1) / block;
  if (g < 1) g = 1;
  if (g > 2048) g = 2048;
  return (int)g;
}

}  // namespace cplxamd
