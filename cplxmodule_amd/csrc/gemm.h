// Internal interface between the GEMM translation units.
#pragma once
#include "common.h"
#include "launch.h"

namespace cplxamd {

struct GemmArgs {
  const void* a_r; const void* a_i; int64_t a_rs, a_cs;   // A[m*a_rs + k*a_cs]
  const void* b_r; const void* b_i; int64_t b_rs, b_cs;   // B[n*b_rs + k*b_cs]
  const float* bias_r; const float* bias_i; const float* emul;
  void* c_r; void* c_i; int64_t ldc;
  int M, N, K;
  int conj_b, accumulate;
  int order = 1, group_m = 4, lds_epilogue = 1;   // bf16 kernel knobs (gemm_bf16_impl.h; env CPLXAMD_GEMM_*)
  // split-K (bf16 kernel, fp32 output): block (split, tile) covers K range [split*kchunk, ...)
  // and writes slab `split` of the workspace; a second kernel reduces the slabs.
  int splits = 1; int kchunk = 0; void* ws = nullptr; int64_t ws_bytes = 0;
  // Gauss 3M combine (real bf16 kernel only): this launch computes t3 = (Ar+Ai)(Br+Bi'); with the
  // dense fp32 slabs t1 = Ar Br and T2 = Ai Bi the epilogue stores
  //   c_r = t1 - gsign T2 + bias_r,   c_i = t3 - t1 - gsign T2 + bias_i      (gsign = -1: conj(B))
  const float* g1 = nullptr; const float* g2 = nullptr; float gsign = 1.0f;
  // batched (generic kernel only): blockIdx.z = batch entry, operands advance by these element strides
  int batch = 1; int64_t a_bs = 0, b_bs = 0, c_bs = 0;
  // epilogue extensions (fused KL + LRT backward): with `accumulate`, C = result + (*beta) * C where beta
  // is a float on the DEVICE (nullptr: 1); emul_exp: the multiplier is exp(emul[m,n]) (emul = log_sigma2)
  const float* beta = nullptr; int emul_exp = 0;
  int emul_both = 0;   // complex GEMM: the (real) multiplier applies to both planes (masked layers' dW * mask)
  // LRT input gradient fused into the epilogue (bf16-out (N,T) kernels, complex and real, persistent and one-tile): C = A op(B) + 2 X (*) ga with the
  // layer input X = (fx_r, fx_i) [M, N] and ga = d|x|^2 [M, N] (all bf16, row pitch fld): the separate
  // cplxamd_lrt_dx_accum pass (7 plane passes over [B, I]) disappears.  Same arithmetic as the two-kernel path:
  // round(acc) to bf16 first, then fmaf(2 x, ga, that) rounded to bf16.
  const void* fx_r = nullptr; const void* fx_i = nullptr; const void* fga = nullptr; int64_t fld = 0;
  // operand scales of the half-precision split products (x3.py 'x2' mode, csrc/split.hip): the operands were multiplied by
  // powers of two sa, sb before they were cut into fp16 pieces; scale_a / scale_b point at DEVICE float[2] = {s, 1 / s}.
  // The accumulators are multiplied by 1 / (sa sb) right behind the K loop (exact: powers of two), a bias that rides in
  // the accumulators starts as bias * (sa sb).  nullptr (both): no scaling.
  const float* scale_a = nullptr; const float* scale_b = nullptr;
  // per-call launch policy (host side only; include/cplxamd.h CPLXAMD_LAUNCH_*, launch.h)
  int flags = 0;
  // dry run (cplxamd_gemm_plan): the launchers write the code of the kernel they WOULD launch here and return; ncu > 0
  // replaces the device's CU count (so that the dispatch can be read without a device)
  int* plan = nullptr; int ncu = 0;
};

__device__ __forceinline__ float gemm_beta(const GemmArgs& g) { return (g.accumulate && g.beta) ? *g.beta : 1.0f; }
// 1 / (sa sb) and sa sb of the scaled split products (1, 1 without scales)
__device__ __forceinline__ float gemm_alpha(const GemmArgs& g) { return g.scale_a ? g.scale_a[1] * g.scale_b[1] : 1.0f; }
__device__ __forceinline__ float gemm_alpha_inv(const GemmArgs& g) { return g.scale_a ? g.scale_a[0] * g.scale_b[0] : 1.0f; }

// (CPLXAMD_MFMA16, the 16-bit matrix instruction of this translation unit: common.h)
__device__ __forceinline__ float gemm_emul(const GemmArgs& g, float m) { return g.emul_exp ? expf(m) : m; }

// any strides / shapes, exact-f32 MFMA (gemm_generic.hip)
template <bool CPLX>
int launch_gemm_generic(const GemmArgs& g, int in_dtype, int out_dtype, hipStream_t st);

int64_t gemm_generic_ws_bytes(int M, int N, int K, bool cplx);   // split-K scratch (0: none)

// bf16 MFMA fast path (gemm_bf16_impl.h); returns CPLXAMD_ESHAPE when the arguments do not
// qualify so that the caller can fall back to the generic kernel.
template <bool CPLX>
int launch_gemm_bf16(const GemmArgs& g, int out_dtype, hipStream_t st);

// one-wave-per-SIMD kernels (gemm_bf16_w4.hip): rc != 0 is an error; taken = false means "not a shape / epilogue this
// family takes" and the caller goes on to the 8-wave kernels
int launch_gemm_bf16_w4(const GemmArgs& g, bool cplx, int out_dtype, bool ta, bool tb, hipStream_t st, bool& taken);

// Gauss 3M (3 real MFMA GEMMs + fused combine) for dense bf16 operands; ESHAPE otherwise
int launch_gemm_bf16_gauss(const GemmArgs& g, int out_dtype, hipStream_t st);
int64_t gemm_bf16_gauss_ws_bytes(int M, int N, int K);

// Launch form, per call (g.flags; launch.h).  A persistent launch owns every CU for its whole duration and gives workgroup
// j the tiles j, j + #CU, ...: if other kernels hold some CUs (an RCCL all-reduce overlapping the backward pass), the
// workgroups that find no CU start only when the first ones END, and the launch takes twice as long.  One workgroup per
// tile degrades by the share of CUs taken instead: the data-parallel hook passes CPLXAMD_LAUNCH_SHARED while its
// collectives are in flight.  The kernel family (one-wave-per-SIMD where it applies) is launch_family(g.flags).

// workspace the bf16 path wants for split-K at this shape (0: no split-K)
int64_t gemm_bf16_ws_bytes(int M, int N, int K, bool cplx);

// the same two families on IEEE-half operands, float32 output only (gemm_f16.hip, gemm_f16_cplx.hip, gemm_f16_w4.hip)
template <bool CPLX>
int launch_gemm_f16(const GemmArgs& g, int out_dtype, hipStream_t st);
int launch_gemm_f16_w4(const GemmArgs& g, bool cplx, int out_dtype, bool ta, bool tb, hipStream_t st, bool& taken);

}  // namespace cplxamd
