/* cplxamd.h -- C ABI of libcplxamd.so: the MI355X (gfx950) kernels behind the
 * cplxmodule hot path (complex linear / conv / batch-norm forward+backward and the
 * variational-dropout local-reparameterization + KL path).
 *
 * Conventions
 *  - plain device pointers, sizes and element strides; no torch / C++ types.
 *  - complex tensors are two planar arrays (re, im): the reference's `Cplx` layout,
 *    /root/reference/cplxmodule/cplx.py:10-52.
 *  - every entry point is asynchronous on `stream` (a hipStream_t passed as void*),
 *    never allocates, never synchronises, and returns 0 on success, a hipError_t
 *    (> 0) for a failed launch, or a negative CPLXAMD_E* code for a bad argument.
 *  - scratch memory is supplied by the caller; sizes come from the *_ws_bytes() helpers.
 *
 * Each entry point cites the reference function (file:line under /root/reference) whose
 * arithmetic it replaces.
 */
#ifndef CPLXAMD_H
#define CPLXAMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CPLXAMD_ABI_VERSION 24

/* element types of activations / outputs */
enum { CPLXAMD_F32 = 0, CPLXAMD_BF16 = 1,
       CPLXAMD_F16 = 2 /* IEEE half: operand type of cplxamd_cgemm_sc_fl / cplxamd_rgemm_sc_fl only */ };

/* complex product algorithm of cplxamd_cgemm */
enum {
  CPLXAMD_ALGO_4M = 0,   /* four real products in one K loop     (cplx.py:634-648 linear_naive) */
  CPLXAMD_ALGO_3M = 1    /* Gauss: three real MFMA GEMMs + combine (cplx.py:651-672 linear_3m)   */
};

/* KL penalty kinds */
enum {
  CPLXAMD_KL_REAL_VD = 0,  /* cplxmodule/nn/relevance/real/vd.py:54-76     */
  CPLXAMD_KL_REAL_ARD = 1, /* cplxmodule/nn/relevance/real/ard.py:10-39    */
  CPLXAMD_KL_CPLX_VD = 2,  /* cplxmodule/nn/relevance/complex/vd.py:95-99  */
  CPLXAMD_KL_CPLX_ARD = 3, /* cplxmodule/nn/relevance/complex/ard.py:9-39  */
  /* SURVEY 8(f) row 4, cplxmodule/nn/relevance/extensions/complex.py: */
  CPLXAMD_KL_CPLX_VD_APPROX = 4,    /* :113-117 softplus-sigmoid approximation          */
  CPLXAMD_KL_CPLX_VD_SCALEFREE = 5, /* :43-46   log|w| - log_sigma2 - Ei(-1/alpha) / 2  */
  CPLXAMD_KL_CPLX_VD_BOGUS = 6      /* :142-160 value -log_alpha (Ei dropped), exact slope  */
};

/* error codes (negative; positive values are hipError_t) */
enum {
  CPLXAMD_OK = 0,
  CPLXAMD_EINVAL = -1,   /* bad enum / null pointer / negative size */
  CPLXAMD_EALIGN = -2,   /* pointer or leading dimension not aligned as required */
  CPLXAMD_ESHAPE = -3,   /* shape not supported by this entry point */
  CPLXAMD_EWS = -4       /* workspace too small */
};

/* Per-call launch policy (ABI 19): the `flags` argument of the `*_fl` entry points.  The library keeps NO mutable state
 * that a launch depends on (SURVEY 8(b) "Threading": re-entrant, no globals except read-only tuning tables): two threads /
 * streams / models in one process choose their launch forms independently, call by call.  GEMM results do not depend on
 * the flags (every form of a GEMM launch produces the same bits; tests/test_gpu_r04.py, test_gpu_gemm_persist.py,
 * test_gpu_r05.py), nor do the convolutions' forward / data-gradient results; the convolution WEIGHT gradients and the
 * conv -> batch-norm moments are sums of per-workgroup float32 partials whose number follows the flags (SHARED: twice as
 * many splits), so they agree across flags to float32 summation order, not bit for bit.
 *   CPLXAMD_LAUNCH_SHARED     other kernels hold compute units while this launch runs (an RCCL all-reduce overlapping the
 *                             backward pass): one workgroup per tile instead of the persistent forms, twice as many weight-
 *                             gradient splits -- a launch that expects every CU would wait for the held ones with its last
 *                             workgroups.
 *   CPLXAMD_LAUNCH_EXCLUSIVE  the chip is this launch's: persistent forms wherever they exist.
 *   neither                   the process default (cplxamd_gemm_set_persistent, deprecated; 1 = exclusive at start).
 *   CPLXAMD_LAUNCH_FAMILY(m)  bf16 GEMM kernel family mask of THIS launch (bit layout of cplxamd_gemm_set_family);
 *                             without it the process default applies (0xbf, env CPLXAMD_GEMM_W4).
 * Both SHARED and EXCLUSIVE, or unknown bits: CPLXAMD_EINVAL.  The flag-less entry points are the `*_fl` ones with
 * flags = 0. */
enum {
  CPLXAMD_LAUNCH_DEFAULT = 0,
  CPLXAMD_LAUNCH_SHARED = 1,
  CPLXAMD_LAUNCH_EXCLUSIVE = 2,
  CPLXAMD_LAUNCH_FAMILY_SET = 0x100,
  CPLXAMD_LAUNCH_FAMILY_SHIFT = 16
};
#define CPLXAMD_LAUNCH_FAMILY(mask) (CPLXAMD_LAUNCH_FAMILY_SET | (((mask) & 0xff) << CPLXAMD_LAUNCH_FAMILY_SHIFT))

int cplxamd_abi_version(void);

/* ------------------------------------------------------------------------------------
 * K6/K7  log-alpha, KL penalty (+ reduction, + gradients), relevance masks.
 * Replaces: GaussianMixin.log_alpha      nn/relevance/{complex/base.py:27-31, real/base.py:23-26}
 *           Cplx.__abs__                  cplx.py:183-192  (stack + norm)
 *           *.penalty                     (the four files listed at the KL kinds above)
 *           ExpiFunction fwd/bwd          nn/relevance/complex/vd.py:15-44 (scipy host round trip)
 *           named_penalties' .sum()       nn/relevance/base.py:135-139
 *           RelevanceMixin.relevance      nn/relevance/real/vd.py:16-19, complex/vd.py:50-53
 * `wi` is NULL for the real kinds.  All tensors float32, contiguous, n elements.
 * ---------------------------------------------------------------------------------- */

/* bytes of scratch the KL reductions need (independent of n) */
int64_t cplxamd_vd_kl_ws_bytes(void);

/* out_elem[n] (nullable) = penalty; out_sum[1] (nullable) = sum(penalty) (fp64 accumulate). */
int cplxamd_vd_kl_fwd(const float* wr, const float* wi, const float* log_sigma2, int kind,
                      float* out_elem, float* out_sum, void* ws, int64_t n, void* stream);

/* Gradient of  sum_j g_j * penalty_j  wrt (log_sigma2, wr, wi).  The upstream gradient is
 * either a tensor g_elem[n] or, when g_elem is NULL, the scalar *g_scalar read on the device
 * (no host sync).  Gradient outputs are nullable; they are overwritten, not accumulated. */
int cplxamd_vd_kl_bwd(const float* wr, const float* wi, const float* log_sigma2, int kind,
                      const float* g_elem, const float* g_scalar, float* g_log_sigma2,
                      float* g_wr, float* g_wi, int64_t n, void* stream);

/* One pass: out_sum = sum(penalty) AND the gradients of gscale * sum(penalty). */
int cplxamd_vd_kl_fwd_bwd(const float* wr, const float* wi, const float* log_sigma2, int kind,
                          float gscale, float* out_sum, float* g_log_sigma2, float* g_wr,
                          float* g_wi, void* ws, int64_t n, void* stream);

/* Per-step operand preparation of a bf16 VD / ARD layer fused with its KL term, ONE pass over
 * (wr, wi, log_sigma2): bf16 copies of the weight planes and of exp(log_sigma2) (the operands of the mean
 * and variance GEMMs; replaces two casts + torch.exp, complex/base.py:50-52) and, with_kl != 0,
 * out_sum = sum(penalty) plus its UNSCALED gradients (as cplxamd_vd_kl_fwd_bwd with gscale = 1).
 * Every output is nullable; wi NULL = real layer; n % 4 == 0 (else CPLXAMD_ESHAPE).
 * HBM: 12 B read + 6 B (bf16 operands) + 12 B (gradients) written per element. */
int cplxamd_vd_prep_kl(const float* wr, const float* wi, const float* log_sigma2, int kind, int with_kl,
                       void* wr_bf16, void* wi_bf16, void* s_bf16, float* out_sum, float* g_log_sigma2,
                       float* g_wr, float* g_wi, void* ws, int64_t n, void* stream);

/* out[n] = log_sigma2 - 2 log(|w| + 1e-12), evaluated so that it reproduces the reference's
 * float32 CPU result bit-for-bit wherever the libm log is correctly rounded. */
int cplxamd_vd_log_alpha(const float* wr, const float* wi, const float* log_sigma2, float* out,
                         int64_t n, void* stream);

/* g_w = g * d log_alpha / d w = -2 g w / (|w| (|w| + 1e-12)) (real: -2 g sign(w) / (|w| + 1e-12)), 0 at
 * w == 0; the gradient wrt log_sigma2 is g itself.  Backward of the differentiable `.log_alpha` property. */
int cplxamd_vd_log_alpha_bwd(const float* g, const float* wr, const float* wi, float* g_wr, float* g_wi,
                             int64_t n, void* stream);

/* mask[n] = (log_alpha <= threshold) ? 1.f : 0.f;  count[1] (nullable, int64) = #ones. */
int cplxamd_vd_mask(const float* wr, const float* wi, const float* log_sigma2, float threshold,
                    float* mask, int64_t* count, void* ws, int64_t n, void* stream);

/* torch_expi seam (nn/relevance/complex/vd.py:44): y = Ei(x), gx = g * exp(x) / x. */
int cplxamd_expi_fwd(const float* x, float* y, int64_t n, void* stream);
int cplxamd_expi_bwd(const float* g, const float* x, float* gx, int64_t n, void* stream);

/* ------------------------------------------------------------------------------------
 * K5  local-reparameterization noise injection.
 * Replaces: CplxLinearGaussian.forward line 56   nn/relevance/complex/base.py:56
 *           LinearGaussian.forward line 49       nn/relevance/real/base.py:49
 *           cplx.randn / randn_like              cplx.py:544-562
 *   y = mu + eps * sqrt(max(s2, 1e-8))
 * mu_i / y_i / eps_i NULL  => real layer (eps ~ N(0,1)); else eps_r, eps_i ~ N(0,1/2).
 * eps_r NULL => noise from the counter-based Philox4x32-7 stream (seed, offset) defined in
 * DESIGN.md ("noise stream"); the backward regenerates it from the same (seed, offset).
 * `state` (nullable): device uint64[2] = {seed, offset}; when given it overrides the two host
 * scalars, so a hipGraph that captured the launch draws fresh noise on every replay
 * (cplxamd_philox_advance moves the stream position on the device).
 * s2 is float32; mu / y / eps / g have element type `dtype`; every pointer 16-byte aligned
 * (CPLXAMD_EALIGN otherwise).
 * ---------------------------------------------------------------------------------- */
int cplxamd_lrt_reparam_fwd(const void* mu_r, const void* mu_i, const float* s2,
                            const void* eps_r, const void* eps_i, uint64_t seed,
                            uint64_t offset, const uint64_t* state, void* y_r, void* y_i,
                            int64_t n, int dtype, void* stream);

/* g_s2 = (g_r*eps_r + g_i*eps_i) * 0.5 / sqrt(max(s2,1e-8)) * [s2 >= 1e-8]
 * g_s2 has element type gs2_dtype (float32, or bf16 when it feeds the bf16 GEMMs). */
int cplxamd_lrt_reparam_bwd(const void* g_r, const void* g_i, const float* s2,
                            const void* eps_r, const void* eps_i, uint64_t seed,
                            uint64_t offset, const uint64_t* state, void* g_s2, int64_t n,
                            int dtype, int gs2_dtype, void* stream);

/* The same two with the element type of s2 as an argument (`s2_dtype`: float32, or bf16 together with bf16
 * mu / g): the variance GEMM / convolution of a bf16 layer writes s2 in bf16, 2 bytes per output less in each
 * direction and no float32 [B, O] tensor kept for the backward. */
int cplxamd_lrt_reparam_fwd_ex(const void* mu_r, const void* mu_i, const void* s2,
                               const void* eps_r, const void* eps_i, uint64_t seed,
                               uint64_t offset, const uint64_t* state, void* y_r, void* y_i,
                               int64_t n, int dtype, int s2_dtype, void* stream);
int cplxamd_lrt_reparam_bwd_ex(const void* g_r, const void* g_i, const void* s2,
                               const void* eps_r, const void* eps_i, uint64_t seed,
                               uint64_t offset, const uint64_t* state, void* g_s2, int64_t n,
                               int dtype, int gs2_dtype, int s2_dtype, void* stream);

/* cplxamd_lrt_reparam_bwd_ex on a [rows][cols] matrix (a [B, O] gradient, or channels-last [B H W][C] planes) that
 * also returns the column sums of g_r / g_i -- the layer's bias gradient (dbr = sum_b G_r, cplx.py:646 under autograd)
 * -- without another pass over the gradient: sum_r / sum_i float32 [cols] (sum_i NULL for a real layer, g_i NULL).
 * Same noise as the flat entry point (the counter is the linear element index).  cols % 8 == 0 and cols / 8 a divisor
 * of 256 or cols a multiple of 2048, else CPLXAMD_ESHAPE (call the flat entry point + cplxamd_colsum).
 * ws: cplxamd_lrt_reparam_bwd_cols_ws_bytes(rows, cols) bytes, 16-byte aligned. */
int64_t cplxamd_lrt_reparam_bwd_cols_ws_bytes(int64_t rows, int cols);
int cplxamd_lrt_reparam_bwd_cols(const void* g_r, const void* g_i, const void* s2,
                                 const void* eps_r, const void* eps_i, uint64_t seed,
                                 uint64_t offset, const uint64_t* state, void* g_s2, int64_t rows,
                                 int cols, int dtype, int gs2_dtype, int s2_dtype, float* sum_r,
                                 float* sum_i, void* ws, int64_t ws_bytes, void* stream);

/* used[0..1] = state[0..1]; state[1] += 1  (device-resident noise stream position) */
int cplxamd_philox_advance(uint64_t* state, uint64_t* used, void* stream);

/* Writes the Philox noise itself (float32), for tests: real (eps_i NULL) or complex. */
int cplxamd_philox_normal(float* eps_r, float* eps_i, uint64_t seed, uint64_t offset,
                          int64_t n, void* stream);

/* ------------------------------------------------------------------------------------
 * K1/K4  complex and real GEMM,  C[m,n] = sum_k A[m,k] * op(B[n,k]) (+ bias[n]).
 * Replaces: cplx.linear_naive / linear_3m / linear_cat   cplx.py:634-694
 *           Cplx.__matmul__                               cplx.py:167-174
 *           the LRT variance GEMM                         nn/relevance/complex/base.py:50-54,
 *                                                         nn/relevance/real/base.py:48
 * and their autograd backward (dX = G conj(W), dW = G^T conj(X)).
 * A is addressed as A[m*a_rs + k*a_cs], B as B[n*b_rs + k*b_cs] (element strides), so one
 * entry point covers NT / NN / TN.  conj_b: use conj(B).  in_dtype: element type of A and B;
 * out_dtype: element type of C (row-major, leading dimension ldc).  bias is float32 [N].
 * accumulate != 0: C += result (C must then be float32).
 * The MFMA fast path (bf16 inputs, a_cs == b_cs == 1, K % 32 == 0, 16-byte aligned rows) is
 * chosen automatically; everything else runs the generic float32-MFMA kernel.
 * algo: CPLXAMD_ALGO_4M, or CPLXAMD_ALGO_3M = Gauss's three products t1 = Ar Br, t2 = Ai Bi,
 * t3 = (Ar + Ai)(Br + Bi): dense bf16 operands only (ESHAPE otherwise, never a silent 4M), needs
 * `ws` of cplxamd_cgemm3m_ws_bytes(M, N, K) bytes (256-B aligned), no accumulate.  The operand
 * sums are rounded to bf16, so the result is NOT bit-identical to 4M.
 * ---------------------------------------------------------------------------------- */
int cplxamd_cgemm(const void* a_r, const void* a_i, int64_t a_rs, int64_t a_cs,
                  const void* b_r, const void* b_i, int64_t b_rs, int64_t b_cs,
                  const float* bias_r, const float* bias_i, void* c_r, void* c_i, int64_t ldc,
                  int M, int N, int K, int conj_b, int in_dtype, int out_dtype, int accumulate,
                  int algo, void* ws, int64_t ws_bytes, void* stream);

/* Same, with (i) emul (nullable, float32 [M,N], leading dimension ldc, float32 C only): both planes of the
 * result are multiplied by it -- the weight gradient of a masked layer, dW * mask, nn/masked/complex.py:33-82;
 * (ii) a DEVICE-side scale on the accumulate operand: accumulate != 0 -> C = result + (*beta) * C
 * (beta NULL: 1).  Lets the weight-gradient GEMM of a VD / ARD layer finish `dW = dW_data + g_kl * dW_kl`
 * in its epilogue, g_kl being the upstream gradient of the KL term that autograd hands over as a device
 * scalar (replaces autograd's separate accumulation pass over every parameter gradient). */
int cplxamd_cgemm_ex(const void* a_r, const void* a_i, int64_t a_rs, int64_t a_cs,
                     const void* b_r, const void* b_i, int64_t b_rs, int64_t b_cs,
                     const float* bias_r, const float* bias_i, const float* emul, void* c_r, void* c_i,
                     int64_t ldc, int M, int N, int K, int conj_b, int in_dtype, int out_dtype, int accumulate,
                     const float* beta, int algo, void* ws, int64_t ws_bytes, void* stream);

/* cplxamd_cgemm_ex with a per-call launch policy (CPLXAMD_LAUNCH_*, top of this file). */
int cplxamd_cgemm_fl(const void* a_r, const void* a_i, int64_t a_rs, int64_t a_cs,
                     const void* b_r, const void* b_i, int64_t b_rs, int64_t b_cs,
                     const float* bias_r, const float* bias_i, const float* emul, void* c_r, void* c_i,
                     int64_t ldc, int M, int N, int K, int conj_b, int in_dtype, int out_dtype, int accumulate,
                     const float* beta, int algo, void* ws, int64_t ws_bytes, int flags, void* stream);

int64_t cplxamd_cgemm3m_ws_bytes(int M, int N, int K);

/* DEPRECATED (ABI 19: pass CPLXAMD_LAUNCH_SHARED / _EXCLUSIVE to the `*_fl` entry points instead; this setter only moves
 * the default that calls with neither flag read).  Process-wide switch of the persistent form of the bf16 GEMM kernels (returns the previous setting; 1 at start).
 * Persistent launches assume the whole chip: turn them off (0) while other kernels are expected to hold CUs -- e.g. an
 * RCCL all-reduce overlapping the backward pass -- and the GEMMs run one workgroup per tile, which shares CUs gracefully.
 * Results are bit-identical either way (tests/test_gpu_gemm_persist.py). */
int cplxamd_gemm_set_persistent(int on);

/* Process-wide choice of the bf16 GEMM kernel family per launch kind (returns the previous mask).  `mask` bits: which
 * launches go to the one-wave-per-SIMD kernels of round 4 (gemm_bf16_w4.hip: 4 waves, 128 x 64 complex wave tile in the
 * accumulator half of the 512-register file, operands staged through registers with both halves of every cache line
 * requested together) wherever those take the shape (full 256 x 128 complex / 256 x 256 real tiles, K % 64 == 0):
 *   bit 0 complex, bf16 out          bit 1 complex, bf16 out with the fused LRT input-gradient term     bit 2 complex, float32 out
 *   bit 3 real, bf16 out             bit 4 real, bf16 out with the fused term                           bit 5 real, float32 out
 *   bit 6 regardless of the K depth (without it: K >= 4096, or 1024 <= K < 4096 on (N,N) launches -- the family pays a
 *         prologue and an epilogue per output tile where the 8-wave kernels run persistent; profiles/r04_gemm_w4_ab.txt)
 *   bit 7 (round 5) the PERSISTENT form of the family -- one workgroup per CU, the K-tile ring running through the output-
 *         tile boundaries -- where it is built and measured faster: real bf16-out launches with the plain epilogue and no
 *         bias ((N,N); (N,T) from K = 4096), more tiles than CUs, and only for a launch that owns the chip
 *         (not CPLXAMD_LAUNCH_SHARED).  profiles/r05_gemm_w4_persistent.txt
 * 0 = the 8-wave LDS-DMA kernels of rounds 1-3 everywhere, -1 = every bit.  Start value 0xbf (env CPLXAMD_GEMM_W4=<mask>
 * overrides).  Both families produce the same bits (same
 * MFMA sequence per accumulator; tests/test_gpu_r04.py). */
int cplxamd_gemm_set_family(int mask);   /* DEPRECATED as a run-time switch: CPLXAMD_LAUNCH_FAMILY(mask) per call */

/* Optional scratch for split-K (few output tiles, long K -- e.g. the weight gradient at batch
 * 2^20, or a 10-output head): pass >= this many bytes as `ws` to cgemm / rgemm; ws may be NULL
 * (no split-K).  With split-K the float32 result is a sum of per-split partial sums. */
int64_t cplxamd_gemm_ws_bytes(int M, int N, int K, int cplx, int in_dtype, int out_dtype);

/* Input gradient of the complex LRT linear layer in ONE launch (nn/relevance/complex/base.py:43-56
 * differentiated):  dX = G conj(W) + 2 X (*) ga,  G [M, K] the output gradient, W [K, N] the bf16 weight as stored
 * ([O, I] row-major: w_rs = 1 over n ... pass the strides of W read as B[n, k], i.e. (1, I)), X [M, N] the layer
 * input and ga = d s2 . exp(log_sigma2) [M, N] (cplxamd_rgemm), all bf16, X / ga with row pitch ldx.  The elementwise
 * term rides in the epilogue of the persistent complex kernel (same arithmetic as cplxamd_cgemm followed by
 * cplxamd_lrt_dx_accum: bit-identical results), which saves the 7 plane passes of that second kernel.
 * Launches the persistent kernel does not take (partial tiles, fewer tiles than CUs, CPLXAMD_LAUNCH_SHARED = the
 * data-parallel form) carry the term in the one-tile kernel's staged epilogue, same bits.  CPLXAMD_ESHAPE when neither
 * epilogue applies (row pitches / N not multiples of 8 elements, unaligned operands): run the two calls instead --
 * nothing is dropped silently. */
int cplxamd_cgemm_lrt_dx(const void* g_r, const void* g_i, int64_t g_rs, int64_t g_cs,
                         const void* w_r, const void* w_i, int64_t w_rs, int64_t w_cs,
                         const void* x_r, const void* x_i, const void* ga, int64_t ldx,
                         void* dx_r, void* dx_i, int64_t ldc, int M, int N, int K, int dtype, void* stream);
/* The same for the REAL local-reparameterization layers (LinearVD / LinearARD, nn/relevance/real/base.py:43-49
 * differentiated): dX = G W + 2 X (*) ga in the epilogue of the persistent real kernel; bit-identical to cplxamd_rgemm
 * followed by cplxamd_lrt_dx_accum.  CPLXAMD_ESHAPE = run those two. */
int cplxamd_rgemm_lrt_dx(const void* g, int64_t g_rs, int64_t g_cs, const void* w, int64_t w_rs, int64_t w_cs,
                         const void* x, const void* ga, int64_t ldx, void* dx, int64_t ldc, int M, int N, int K, int dtype,
                         void* stream);

/* The two fused input gradients and cplxamd_rgemm_ex with a per-call launch policy (CPLXAMD_LAUNCH_*). */
int cplxamd_cgemm_lrt_dx_fl(const void* g_r, const void* g_i, int64_t g_rs, int64_t g_cs,
                            const void* w_r, const void* w_i, int64_t w_rs, int64_t w_cs,
                            const void* x_r, const void* x_i, const void* ga, int64_t ldx,
                            void* dx_r, void* dx_i, int64_t ldc, int M, int N, int K, int dtype, int flags, void* stream);
int cplxamd_rgemm_lrt_dx_fl(const void* g, int64_t g_rs, int64_t g_cs, const void* w, int64_t w_rs, int64_t w_cs,
                            const void* x, const void* ga, int64_t ldx, void* dx, int64_t ldc, int M, int N, int K,
                            int dtype, int flags, void* stream);
int cplxamd_rgemm_fl(const void* a, int64_t a_rs, int64_t a_cs, const void* b, int64_t b_rs,
                     int64_t b_cs, const float* bias, const float* emul, int emul_exp, void* c, int64_t ldc,
                     int M, int N, int K, int in_dtype, int out_dtype, int accumulate, const float* beta,
                     void* ws, int64_t ws_bytes, int flags, void* stream);
/* Which kernel a bf16 cplxamd_cgemm_fl / cplxamd_rgemm_fl call with these shapes, layouts ("N" = K-contiguous rows: ta /
 * tb = 0) and flags launches -- a pure function of its arguments and of `ncu`, the device's CU count (0: ask the current
 * device): 0 generic float32-exact kernel (the bf16 path declines), 1 8-wave one-tile, 2 8-wave persistent, 3
 * one-wave-per-SIMD (w4), 4 split-K slabs on the 8-wave kernels, 5 split-K slabs on w4, 6 w4 persistent.  `epi`: 0 plain / bias, 1 fused
 * LRT input-gradient term, 2 float32 accumulate / multiplier epilogue.  Dense operands, aligned pointers and a
 * workspace of cplxamd_gemm_ws_bytes are assumed.  (Dispatch made inspectable: the test of the per-call flags reads it.) */
int cplxamd_gemm_plan(int cplx, int M, int N, int K, int ta, int tb, int out_dtype, int epi, int flags, int ncu);

/* Batched complex GEMM (Cplx.__matmul__ on [..., M, K] @ [..., K, N], cplx.py:167-181): `batch`
 * independent products in ONE launch of the exact-f32 MFMA kernel (any strides, any dtype pair);
 * entry z reads A + z*a_bs, B + z*b_bs and writes C + z*c_bs (strides in elements).  batch <= 65535. */
int cplxamd_cgemm_batched(const void* a_r, const void* a_i, int64_t a_rs, int64_t a_cs, int64_t a_bs,
                          const void* b_r, const void* b_i, int64_t b_rs, int64_t b_cs, int64_t b_bs,
                          void* c_r, void* c_i, int64_t ldc, int64_t c_bs, int batch, int M, int N, int K,
                          int conj_b, int in_dtype, int out_dtype, void* stream);

/* real GEMM; emul (nullable, float32 [M,N] with leading dimension ldc): C = (A B^T) * emul. */
int cplxamd_rgemm(const void* a, int64_t a_rs, int64_t a_cs, const void* b, int64_t b_rs,
                  int64_t b_cs, const float* bias, const float* emul, void* c, int64_t ldc,
                  int M, int N, int K, int in_dtype, int out_dtype, int accumulate, void* ws,
                  int64_t ws_bytes, void* stream);

/* Same; emul_exp != 0: the multiplier is exp(emul[m,n]) (emul = log_sigma2: the LRT gradient
 * d log_sigma2 = (g_s2^T |x|^2) * exp(log_sigma2), nn/relevance/complex/base.py:52, without a
 * materialised exp); accumulate with the device-side scale *beta as in cplxamd_cgemm_ex. */
int cplxamd_rgemm_ex(const void* a, int64_t a_rs, int64_t a_cs, const void* b, int64_t b_rs,
                     int64_t b_cs, const float* bias, const float* emul, int emul_exp, void* c, int64_t ldc,
                     int M, int N, int K, int in_dtype, int out_dtype, int accumulate, const float* beta,
                     void* ws, int64_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * elementwise / layout helpers used by the layers (all HBM-bound streaming kernels)
 * ---------------------------------------------------------------------------------- */
/* out = xr^2 + xi^2 (xi NULL: xr^2)            nn/relevance/complex/base.py:51 */
int cplxamd_abs2(const void* xr, const void* xi, void* out, int64_t n, int in_dtype,
                 int out_dtype, void* stream);
/* out = |x| = sqrt(xr^2 + xi^2), float32, rounded exactly like torch's CPU 2-norm
 * (Cplx.__abs__, cplx.py:183-192) */
int cplxamd_modulus(const float* xr, const float* xi, float* out, int64_t n, void* stream);
/* abs(Cplx) for float32 / bf16 planes and its backward d x = g x / |x| with 0 at x == 0 (the
 * subgradient of the reference's stack + norm, cplx.py:183-192) */
int cplxamd_cplx_abs_fwd(const void* xr, const void* xi, void* out, int64_t n, int dtype, void* stream);
int cplxamd_cplx_abs_bwd(const void* g, const void* xr, const void* xi, void* dxr, void* dxi, int64_t n,
                         int dtype, void* stream);
/* out = in * mask (float32 mask) for one (in_i = out_i = NULL) or two planes, with the dtype
 * conversion of the bf16 path folded in: the sparsified weight of the masked layers
 * (nn/masked/real.py:25-71, complex.py:33-82) and, applied to gradients, its backward. */
int cplxamd_mask_mul(const void* in_r, const void* in_i, const float* mask, void* out_r, void* out_i,
                     int64_t n, int in_dtype, int out_dtype, void* stream);
/* out = exp(x), x float32                      nn/relevance/complex/base.py:52 */
int cplxamd_exp(const float* x, void* out, int64_t n, int out_dtype, void* stream);
/* dtype conversion */
int cplxamd_cast(const void* in, void* out, int64_t n, int in_dtype, int out_dtype, void* stream);

/* ---- float32-accurate products on the bf16 matrix pipe ("x3" operands; csrc/split.hip) -------------------------------
 * Replaces the float32 arithmetic of cplx.linear (cplxmodule/cplx.py:641-646) and of the local-reparameterization
 * variance products (nn/relevance/complex/base.py:43-56, real/base.py:43-49) at 2^-24-level accuracy without the 157-TFLOP/s
 * float32 MFMA: a float32 value is the exact sum of three bf16 values x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1),
 * and a product needs the six terms x0 w0, x0 w1, x1 w0, x0 w2, x2 w0, x1 w1 (the other three are below 2^-24 |x w|), each
 * exact in the bf16 MFMA with float32 accumulation.
 * cplxamd_split3 writes the pieces of  op(src)  [rows][cols] (row pitch ld_src; float32) as bf16 matrices
 *     dst + p * piece_stride + r * ld_dst + c,   p = 0 .. 2 (pattern A) or 0 .. 5 (pattern B)
 *   CPLXAMD_SPLIT_A   (x2, x1, x0)               -- the activation side; SUFFIXES of it are GEMM operands
 *   CPLXAMD_SPLIT_B   (w2, w1, w1, w0, w0, w0)   -- the weight side, replicated so that the six terms are three launches of
 *                     cplxamd_cgemm_fl / cplxamd_rgemm_fl (bf16 in, float32 out, accumulate from the second on) with
 *                     the pieces concatenated along K:   x0 . w2,   [x1|x0] . [w1|w1],   [x2|x1|x0] . [w0|w0|w0]
 *   (piece_stride = cols, ld_dst = 3 cols / 6 cols: pieces side by side in a row = K-concatenation of K-contiguous operands;
 *    piece_stride = rows * ld_dst: pieces stacked = K-concatenation of a K-major operand.)
 * op: CPLXAMD_SPLIT_ID  src;  CPLXAMD_SPLIT_ABS2  src^2 + src2^2 (src2 NULL: src^2; complex/base.py:51);
 *     CPLXAMD_SPLIT_EXP  exp(src) (log_sigma2 -> sigma^2, complex/base.py:52).  src2 only with ABS2.
 * A non-finite value is carried by the leading piece alone (the others 0).
 * cols % 8 == 0, ld_src % 4 == 0, ld_dst % 8 == 0, piece_stride % 8 == 0 (CPLXAMD_ESHAPE), 16-byte aligned pointers (CPLXAMD_EALIGN).
 * HBM: 4 (8 with src2) bytes read, 6 (pattern A) / 12 (pattern B) bytes written per element. */
enum { CPLXAMD_SPLIT_A = 0, CPLXAMD_SPLIT_B = 1 };
enum { CPLXAMD_SPLIT_ID = 0, CPLXAMD_SPLIT_ABS2 = 1, CPLXAMD_SPLIT_EXP = 2, CPLXAMD_SPLIT_MAX2 = 3 /* absmax only */ };
int cplxamd_split3(const float* src, const float* src2, int64_t ld_src, void* dst, int64_t ld_dst, int64_t piece_stride,
                   int64_t rows, int cols, int op, int pattern, void* stream);

/* ---- the same on IEEE-half pieces: three piece products instead of six ("x2" operands) -------------------------------
 * half has 11 significand bits: x s = h0 + h1 to 2^-22 with h0 = half(x s), h1 = half(x s - h0), and a product needs
 * h0 w0, h0 w1, h1 w0 (h1 w1 is below 2^-22 |x w|) -- half the matrix work of the bf16 split at 2^-22 instead of 2^-24.
 * The 5-bit exponent needs care: every operand is multiplied by a power of two s chosen from its largest magnitude
 * (cplxamd_absmax_scale: max |op(x)| s in [2^14, 2^15); s = 1 for an all-zero or non-finite tensor) before it is cut, and
 * the GEMM multiplies its accumulators by 1 / (sa sb) behind the K loop (exact).  An element keeps all 22 bits while
 * |x| >= 2^-17 max |x|; below that its second piece is subnormal and the element's error is <= 2^-39 max |x| absolute --
 * norm-wise the product is accurate to 2^-22 whatever the dynamic range.
 *   cplxamd_absmax_scale   scale[0] = s, scale[1] = 1 / s (device floats) for op(src) [rows][cols]; ws >= cplxamd_absmax_ws_bytes();
 *                          op as cplxamd_split3, or CPLXAMD_SPLIT_MAX2: max(|src|, |src2|) -- the two planes of a complex
 *                          operand must share one scale (the four real products of the complex GEMM mix them)
 *   cplxamd_split2h        pieces of op(src) * scale[0] as IEEE half, layouts as cplxamd_split3:
 *                          CPLXAMD_SPLIT_A (h1, h0)      CPLXAMD_SPLIT_B (w1, w0, w0)
 *                          launches:  h0 . w1,   [h1|h0] . [w0|w0]
 *   cplxamd_cgemm_sc_fl / cplxamd_rgemm_sc_fl   cplxamd_cgemm_fl / cplxamd_rgemm_fl for half operands (in_dtype = CPLXAMD_F16),
 *                          float32 C, with the two operands' scale buffers (both NULL: no scaling).  The MFMA kernel families
 *                          of the bf16 path compiled for v_mfma_f32_32x32x16_f16; shapes they decline: CPLXAMD_ESHAPE (there
 *                          is no generic fallback -- run the bf16 pieces). */
int64_t cplxamd_absmax_ws_bytes(void);
int cplxamd_absmax_scale(const float* src, const float* src2, int64_t ld_src, int64_t rows, int cols, int op, float* scale,
                         void* ws, void* stream);
int cplxamd_split2h(const float* src, const float* src2, int64_t ld_src, void* dst, int64_t ld_dst, int64_t piece_stride,
                    int64_t rows, int cols, int op, int pattern, const float* scale, void* stream);
int cplxamd_cgemm_sc_fl(const void* a_r, const void* a_i, int64_t a_rs, int64_t a_cs,
                        const void* b_r, const void* b_i, int64_t b_rs, int64_t b_cs,
                        const float* bias_r, const float* bias_i, const float* emul, void* c_r, void* c_i, int64_t ldc,
                        int M, int N, int K, int conj_b, int in_dtype, int accumulate, const float* beta,
                        const float* scale_a, const float* scale_b, void* ws, int64_t ws_bytes, int flags, void* stream);
int cplxamd_rgemm_sc_fl(const void* a, int64_t a_rs, int64_t a_cs, const void* b, int64_t b_rs, int64_t b_cs,
                        const float* bias, const float* emul, int emul_exp, void* c, int64_t ldc, int M, int N, int K,
                        int in_dtype, int accumulate, const float* beta, const float* scale_a, const float* scale_b,
                        void* ws, int64_t ws_bytes, int flags, void* stream);

/* The channels-last 3 x 3 convolutions on IEEE-half pieces (ABI 22; csrc/conv_cl2_f16.hip, conv_cl_wgrad_f16.hip: the bf16
 * kernels' sources compiled for the half matrix instruction): cplx.conv2d (cplxmodule/cplx.py:717-838) and its autograd in
 * float32-level accuracy for float32 layers -- the pieces are the [h1|h0] rows of cplxamd_split2h over the channels-last
 * planes read as [B H W][C] matrices, the weights' pieces are packed by cplxamd_conv2d_cl_pack (16-bit agnostic).
 *   cplxamd_conv2d_cl2h_fl   cplxamd_conv2d_cl2_fl with x: half planes [B][H][W][pitch] of which the C channels behind the
 *                            given pointers are convolved (pitch >= C: a channel window, e.g. the h0 half of [h1|h0]),
 *                            y: FLOAT32 [B][Ho][Wo][N], accumulate != 0: y += result, the result times 1 / (sa sb) from the
 *                            operands' scale buffers (both NULL: none), bias added once (pass it to the first launch).
 *                            forward y = h0 * w1 (+ b), then += [h1|h0] * [w0|w0]; the data gradient likewise (mode 1).
 *                            B H W pitch 2 < 0xF0000000 bytes per plane (CPLXAMD_ESHAPE: chunk the batch).
 *   cplxamd_conv2d_clh_wgrad(_fl / _ws_bytes)   cplxamd_conv2d_cl_wgrad on half planes: with G = [g1|g0] (2 Co channels) and
 *                            X = [x1|x0] (2 Ci) one launch gives the four piece blocks [2 Co][2 Ci][3][3]; the weight gradient is
 *                            (g0 x1) + (g1 x0) + (g0 x0) times 1 / (sg sx). */
int cplxamd_conv2d_cl2h_fl(const void* x_r, const void* x_i, int pitch, const void* w_packed, const float* bias_r,
                           const float* bias_i, float* y_r, float* y_i, int accumulate, const float* scale_a,
                           const float* scale_b, int64_t B, int H, int W, int C, int N, int pad_h, int pad_w, int mode, void* ws,
                           int64_t ws_bytes, int flags, void* stream);
/* ABI 24: the same with a contraction window that WRAPS around the pixel -- channel j of the contraction is channel
 * (c_start + j) mod pitch of the pixel (c_start, pitch multiples of 16; C + c_start <= 2 pitch).  Rows stored [h1 | h0]
 * (pitch = 2 c) read with c_start = c, C = 3 c give [h0 | h1 | h0]; against weights packed [w1 | w0 | w0] that is all three
 * piece products of a float32 convolution in ONE launch (one float32 epilogue instead of two, no accumulate pass). */
int cplxamd_conv2d_cl2h_wrap_fl(const void* x_r, const void* x_i, int pitch, int c_start, const void* w_packed,
                                const float* bias_r, const float* bias_i, float* y_r, float* y_i, int accumulate,
                                const float* scale_a, const float* scale_b, int64_t B, int H, int W, int C, int N, int pad_h,
                                int pad_w, int mode, void* ws, int64_t ws_bytes, int flags, void* stream);
int64_t cplxamd_conv2d_clh_wgrad_ws_bytes(int64_t B, int H, int W, int Ci, int Co);
int cplxamd_conv2d_clh_wgrad(const void* g_r, const void* g_i, const void* x_r, const void* x_i, const float* emul,
                             float* dw_r, float* dw_i, int64_t B, int H, int W, int Ci, int Co, int KH, int KW, int dil_h,
                             int dil_w, int pad_h, int pad_w, void* ws, int64_t ws_bytes, void* stream);
int cplxamd_conv2d_clh_wgrad_fl(const void* g_r, const void* g_i, const void* x_r, const void* x_i, const float* emul,
                                float* dw_r, float* dw_i, int64_t B, int H, int W, int Ci, int Co, int KH, int KW, int dil_h,
                                int dil_w, int pad_h, int pad_w, void* ws, int64_t ws_bytes, int flags, void* stream);
/* ABI 24: the same without the block dW[0:skip_co, 0:skip_ci] (multiples of 64, both zero or both positive; those entries of
 * dw are left untouched): with G = [g1|g0], X = [x1|x0] and skip = (Co, Ci) the product g1 x1 -- 2^-22 of the result, not
 * one of the three piece products -- is not computed (3 of 4 tiles). */
int cplxamd_conv2d_clh_wgrad_skip_fl(const void* g_r, const void* g_i, const void* x_r, const void* x_i, float* dw_r, float* dw_i,
                                     int64_t B, int H, int W, int Ci, int Co, int KH, int KW, int dil_h, int dil_w, int pad_h,
                                     int pad_w, int skip_co, int skip_ci, void* ws, int64_t ws_bytes, int flags, void* stream);
/* out[c, r] = in[r, c]  (rows x cols row-major in, ld = leading dims) */
int cplxamd_transpose(const void* in, int64_t ld_in, void* out, int64_t ld_out, int rows,
                      int cols, int dtype, void* stream);
/* out[n] = sum_m in[m, n]   (bias gradient; float32 out); ws (nullable -> slow path) holds
 * cplxamd_colsum_ws_bytes(cols) bytes of partial sums */
int64_t cplxamd_colsum_ws_bytes(int cols);
int cplxamd_colsum(const void* in, int64_t ld, float* out, int rows, int cols, int dtype,
                   void* ws, void* stream);
/* Column sums of both planes of a complex [rows, cols] tensor (a linear layer's complex bias gradient): one launch on
 * cplxamd_colsum's few-rows path, else two cplxamd_colsum passes.  ws as for cplxamd_colsum. */
int cplxamd_colsum2(const void* in_r, const void* in_i, int64_t ld, float* out_r, float* out_i, int rows, int cols,
                    int dtype, void* ws, void* stream);
/* dxr += 2 xr ga ; dxi += 2 xi ga   (LRT backward; xi/dxi NULL for real) */
int cplxamd_lrt_dx_accum(void* dxr, void* dxi, const void* xr, const void* xi, const void* ga,
                         int64_t n, int dtype, int ga_dtype, void* stream);

/* ------------------------------------------------------------------------------------
 * K2  complex / real 2-d convolution (cross-correlation, NCHW, zero padding) as implicit GEMM.
 * Replaces: cplx.convnd / convnd_quick / convnd_naive / conv2d   cplx.py:717-838
 *           the LRT variance conv   nn/relevance/complex/base.py:125-133, real/base.py:152-162
 * and their autograd backward.  Planes of element type `dtype`; xi / wi / yi NULL => real conv.
 * geom = int[14] {B, Ci, Co, H, W, KH, KW, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w, groups};
 * weight [Co, Ci/groups, KH, KW]; bias float32 [Co] (nullable pair).
 *   fwd   : y = x (*) w (+ bias)            (no conjugation, cplx.py:719-726)
 *   dgrad : dx = g (*)^T conj(w)
 *   wgrad : dw = sum g conj(x)  (float32 out; emul nullable: dw_r *= emul, the LRT exp(ls2) factor)
 *           split-K partial slabs live in `ws` (>= cplxamd_conv2d_wgrad_ws_bytes).
 * ---------------------------------------------------------------------------------- */
int cplxamd_conv2d_out_shape(const int* geom, int* ho, int* wo);
int cplxamd_conv2d_fwd(const void* xr, const void* xi, const void* wr, const void* wi,
                       const float* bias_r, const float* bias_i, void* yr, void* yi,
                       const int* geom, int dtype, void* stream);
int cplxamd_conv2d_dgrad(const void* gr, const void* gi, const void* wr, const void* wi,
                         void* dxr, void* dxi, const int* geom, int dtype, void* stream);
int cplxamd_conv2d_wgrad_splits(const int* geom);
int64_t cplxamd_conv2d_wgrad_ws_bytes(const int* geom, int cplx);
int cplxamd_conv2d_wgrad(const void* gr, const void* gi, const void* xr, const void* xi,
                         const float* emul, float* dwr, float* dwi, const int* geom, int dtype,
                         void* ws, int64_t ws_bytes, void* stream);
/* cplxamd_conv2d_wgrad that also returns the bias gradient dbr / dbi [Co] (float32; complex: one [2][Co] array, dbi ==
 * dbr + Co; NULL = not wanted): the sum of G over batch and pixels rides as one more column of the weight-gradient GEMM
 * (its B entries are (1, 0)) and through the same slab sum -- no launches of its own -- unless that column would start a
 * tile of its own, in which case the library runs cplxamd_chansum2 / cplxamd_chansum.  Same workspace. */
int cplxamd_conv2d_wgrad_bias(const void* gr, const void* gi, const void* xr, const void* xi,
                              const float* emul, float* dwr, float* dwi, float* dbr, float* dbi, const int* geom,
                              int dtype, void* ws, int64_t ws_bytes, void* stream);
/* bf16 fast path of the same three operations (bf16 MFMA, register-gathered operand tiles).
 * ktab: device copy of the int[3*T] table written by cplxamd_conv2d_ktab_fill (a host-side
 * helper; mode 0 = fwd / wgrad table over (ci, kh, kw), 1 = dgrad table over (co, kh, kw)).
 * They return CPLXAMD_ESHAPE when the shape does not qualify (K % 32 != 0, strided dgrad):
 * the caller then uses the generic entry points above.  dgrad takes the weight repacked to
 * [groups][Ci/g][Co/g * KH * KW]. */
int cplxamd_conv2d_ktab_size(const int* geom, int mode);
int cplxamd_conv2d_ktab_fill(const int* geom, int mode, int* host_out);
int cplxamd_conv2d_bf16_fwd(const void* xr, const void* xi, const void* wr, const void* wi,
                            const float* bias_r, const float* bias_i, void* yr, void* yi,
                            const int* geom, const int* ktab, void* stream);
int cplxamd_conv2d_bf16_dgrad(const void* gr, const void* gi, const void* wtr, const void* wti,
                              void* dxr, void* dxi, const int* geom, const int* ktab,
                              void* stream);
int64_t cplxamd_conv2d_bf16_wgrad_ws_bytes(const int* geom, int cplx);
int cplxamd_conv2d_bf16_wgrad(const void* gr, const void* gi, const void* xr, const void* xi,
                              const float* emul, float* dwr, float* dwi, const int* geom,
                              const int* ktab, void* ws, int64_t ws_bytes, void* stream);

/* bf16 convolution, stride 1, groups 1, as ONE shifted-row MFMA GEMM over a zero-padded
 * channels-last copy of the input (conv_nhwc.hip): no gathers, operands move by LDS-DMA.
 *   cplxamd_nhwc_pad : planar NCHW bf16 x[B,C,H,W] -> out[B, Hp, Wp, C] (C % 8 == 0) with the image at
 *       rows pad_h.., columns pad_w.. and zeros elsewhere (Hp >= H + pad_h, Wp >= W + pad_w)
 *   cplxamd_conv2d_nhwc : with r(b, i, j) the flattened row index of grid position (i, j) of image b,
 *         y[b, co, ho, wo] = sum_{kh,kw,c} xp[r(b, ho + oh, wo + ow) + row_bias + kh*dil_h*Wp + kw*dil_w][c]
 *                                          * op(w[co, kh, kw, c]) (+ bias[co])      ho < Hout, wo < Wout
 *       xp_* channels-last bf16 grid [B, Hp, Wp, C], w_* bf16 [KH][KW][C/16][Cout][16] (C % 32 == 0,
 *       (KW-1)*dil_w <= 32), y_* planar NCHW [B, Cout, Hout, Wout] of out_dtype; xp_i == NULL: real
 *       convolution; conj_w: use conj(w).  row_bias <= 0; the buffer must hold -row_bias zero rows
 *       before the grid and 320 + (KH-1)*dil_h*Wp + (KW-1)*dil_w READABLE rows behind it (the bf16
 *       kernel does not clamp row indices; those rows only feed outputs that are not stored).
 * Forward (input padded by (ph, pw)): row_bias = oh = ow = 0, Hout = Hp-(KH-1)*dil_h,
 * w[kh][kw][c/16][co][c%16] = weight[co][c][kh][kw].  Data gradient: xp = the output gradient laid
 * top-left on the INPUT's padded grid (cplxamd_nhwc_pad(g, pad 0, Hp, Wp)), the roles of the two
 * channel dimensions swapped, the kernel flipped in both spatial dimensions, conj_w = 1,
 * row_bias = -((KH-1)*dil_h*Wp + (KW-1)*dil_w), (oh, ow) = (ph, pw), (Hout, Wout) = (H, W): the same
 * gradient grid then also feeds cplxamd_conv2d_nhwc_wgrad.
 * Replaces the same reference code as cplxamd_conv2d_fwd / _dgrad (cplx.py:717-838). */
int cplxamd_nhwc_pad(const void* x, void* out, int B, int C, int H, int W, int pad_h, int pad_w,
                     int Hp, int Wp, void* stream);
int cplxamd_conv2d_nhwc(const void* xp_r, const void* xp_i, const void* w_r, const void* w_i,
                        const float* bias_r, const float* bias_i, void* y_r, void* y_i, int B,
                        int Hp, int Wp, int C, int Cout, int KH, int KW, int dil_h, int dil_w,
                        int conj_w, int64_t row_bias, int oh, int ow, int Hout, int Wout,
                        int out_dtype, void* stream);

/* Exact-float32 versions of the two entry points above (conv_nhwc_f32.hip, v_mfma_f32_32x32x2_f32):
 * float32 grids / outputs, C % 16 == 0, weights packed [KH][KW][C/4][Cout][4]. */
int cplxamd_nhwc_pad_f32(const void* x, void* out, int B, int C, int H, int W, int pad_h, int pad_w,
                         int Hp, int Wp, void* stream);
int cplxamd_conv2d_nhwc_f32(const void* xp_r, const void* xp_i, const void* w_r, const void* w_i,
                            const float* bias_r, const float* bias_i, void* y_r, void* y_i, int B,
                            int Hp, int Wp, int C, int Cout, int KH, int KW, int dil_h, int dil_w,
                            int conj_w, int64_t row_bias, int oh, int ow, int Hout, int Wout,
                            void* stream);

/* Weight gradient on the same channels-last copies (conv_nhwc_wgrad.hip):
 *   dw[co, ci, kh, kw] = sum_r gp[r][co] * conj(xp[r + kh*dil_h*Wp + kw*dil_w][ci]) (* emul, real only)
 * over the rows r = (b, hp, wp) of the padded input grid.  gp_*: the output gradient laid on that
 * grid by cplxamd_nhwc_pad(g, .., pad_h = 0, pad_w = 0, Hp, Wp).  Both buffers carry ZERO tail rows
 * after the last image so that the kernel never clamps a row: gp up to a multiple of 32 rows, xp
 * 32 + (KH-1)*dil_h*Wp + (KW-1)*dil_w rows.  Ci % 8 == Co % 8 == 0, KW <= 4.  dw_* float32
 * [Co, Ci, KH, KW]; ws >= cplxamd_conv2d_nhwc_wgrad_ws_bytes. */
int64_t cplxamd_conv2d_nhwc_wgrad_ws_bytes(int B, int Hp, int Wp, int Ci, int Co, int KH, int KW,
                                           int cplx);
int cplxamd_conv2d_nhwc_wgrad(const void* gp_r, const void* gp_i, const void* xp_r, const void* xp_i,
                              const float* emul, float* dw_r, float* dw_i, int B, int Hp, int Wp,
                              int Ci, int Co, int KH, int KW, int dil_h, int dil_w, void* ws,
                              int64_t ws_bytes, void* stream);
/* exact-float32 version (float32 grids; Ci % 4 == Co % 4 == 0; gp tail up to a multiple of 16 rows) */
int64_t cplxamd_conv2d_nhwc_wgrad_f32_ws_bytes(int B, int Hp, int Wp, int Ci, int Co, int KH, int KW,
                                               int cplx);
int cplxamd_conv2d_nhwc_wgrad_f32(const void* gp_r, const void* gp_i, const void* xp_r, const void* xp_i,
                                  const float* emul, float* dw_r, float* dw_i, int B, int Hp, int Wp,
                                  int Ci, int Co, int KH, int KW, int dil_h, int dil_w, void* ws,
                                  int64_t ws_bytes, void* stream);
/* ---- complex convolution on UNPADDED channels-last planes (csrc/conv_cl.hip, conv_cl_wgrad.hip), bf16 in, fp32
 * accumulation: stride 1, groups 1, kernel width 3, zero padding 0 <= 2 pad <= dil (K - 1) per dimension (up to `same`).
 * Activations x: [B][H][W][C], y: [B][Ho][Wo][N] (Ho = H + 2 pad_h - dil_h (KH - 1), likewise Wo) -- the storage of
 * torch.channels_last tensors; no padded copies, no layout passes between layers that stay channels-last.
 * Replaces cplx.conv2d -> convnd and its autograd (cplxmodule/cplx.py:717-838) for these shapes; everything else:
 * CPLXAMD_ESHAPE, and the caller takes cplxamd_conv2d_nhwc / _bf16 / _fwd.
 *   cplxamd_conv2d_cl_pack: bf16 weight planes [Co][Ci][KH][KW] -> the kernel's per-stage LDS images (bytes:
 *       cplxamd_conv2d_cl_pack_bytes(N, C, KH, KW) with (N, C) = (Co, Ci) forward, (Ci, Co) data gradient); dgrad = 1
 *       folds the flip, the channel swap and the conjugate of the data gradient into the packing.
 *   cplxamd_conv2d_cl: mode 0 forward (C = Ci, N = Co, bias optional), mode 1 the data gradient OF that forward
 *       (x = output gradient [B][Ho][Wo][C = Co], y = input gradient [B][H][W][N = Ci]; H, W, pad_* still the forward's).
 *       Needs C % 16 == 0, N % 64 == 0, (KH * C / 16) % 6 == 0; ws >= cplxamd_conv2d_cl_ws_bytes(N).
 *   cplxamd_conv2d_cl_wgrad: dW[Co][Ci][3][3] (float32 planes, optionally times emul) from x [B][H][W][Ci] and
 *       g [B][Ho][Wo][Co]; needs KH = KW = 3, Ci % 64 == Co % 64 == 0 (any image width);
 *       ws >= cplxamd_conv2d_cl_wgrad_ws_bytes(B, H, W, Ci, Co).  Deterministic (fixed-order slab reduction).
 *   cplxamd_cl_to_nchw: channels-last [B][S][C] -> planar [B][C][S] (bf16; C % 8 == S % 8 == 0): the way back for a
 *       caller whose tensors are plain contiguous (the other direction is cplxamd_nhwc_pad with zero padding). */
int cplxamd_cl_to_nchw(const void* x_cl, void* out, int64_t B, int C, int64_t S, void* stream);
int64_t cplxamd_conv2d_cl_pack_bytes(int N, int C, int KH, int KW);
int64_t cplxamd_conv2d_cl_ws_bytes(int N);
int cplxamd_conv2d_cl_pack(const void* w_r, const void* w_i, void* out, int Co, int Ci, int KH, int KW, int dgrad,
                           void* stream);
int cplxamd_conv2d_cl(const void* x_r, const void* x_i, const void* w_packed, const float* bias_r, const float* bias_i,
                      void* y_r, void* y_i, int64_t B, int H, int W, int C, int N, int KH, int KW, int dil_h, int dil_w,
                      int pad_h, int pad_w, int mode, void* ws, int64_t ws_bytes, void* stream);
/* cplxamd_conv2d_cl for dilation 1 in its 2-d-patch form (csrc/conv_cl2.hip: a 16 x 32 pixel tile stages the 18 x 34 input
 * patch once per 16-channel slice for all nine taps): same arguments, same packed weights, same results; needs KH = KW = 3,
 * dilation 1, C % 32 == 0 -- CPLXAMD_ESHAPE otherwise. */
int cplxamd_conv2d_cl2(const void* x_r, const void* x_i, const void* w_packed, const float* bias_r, const float* bias_i,
                       void* y_r, void* y_i, int64_t B, int H, int W, int C, int N, int KH, int KW, int dil_h, int dil_w,
                       int pad_h, int pad_w, int mode, void* ws, int64_t ws_bytes, void* stream);
/* Input gradient of the local-reparameterization convolutions (CplxConv2dVD / CplxConv2dARD: the layer of
 * cplxmodule/nn/relevance/complex.py:157-190, differentiated) in ONE launch:
 *     dx = dgrad(g; w) + 2 x (*) ga      per plane,
 * dgrad = cplxamd_conv2d_cl2 with mode 1 (w_packed from cplxamd_conv2d_cl_pack with dgrad = 1) and ga the variance
 * path's data gradient (cplxamd_conv2d_clr, mode 1).  g_r / g_i: [B][H + 2 pad - 2][W + 2 pad - 2][C] output gradients;
 * x_r / x_i / ga / dx_r / dx_i: [B][H][W][N], all bf16 channels-last.  Bit-identical to cplxamd_conv2d_cl2 followed by
 * cplxamd_lrt_dx_accum (the sum is formed on the bf16-rounded convolution result), minus one read-modify-write pass
 * over dx.  Shapes as cplxamd_conv2d_cl2 (CPLXAMD_ESHAPE otherwise: take the two launches). */
int cplxamd_conv2d_cl2_lrt_dx(const void* g_r, const void* g_i, const void* w_packed, const void* x_r, const void* x_i,
                              const void* ga, void* dx_r, void* dx_i, int64_t B, int H, int W, int C, int N, int pad_h,
                              int pad_w, void* ws, int64_t ws_bytes, void* stream);
/* The convolution in front of a batch-norm layer (the conv -> BN pair of the reference's networks:
 * cplxmodule/cplx.py:729-742 followed by nn/modules/batchnorm.py:62-123) with the layer's FORWARD STATISTICS formed in the
 * convolution's epilogue: cplxamd_conv2d_cl2 (forward, mode 0) that also writes, per workgroup, one row
 * [N][5] float64 of (sum re, sum im, sum re^2, sum im^2, sum re im) over the output pixels it produced -- of the bf16
 * values as stored -- into `partials`.  cplxamd_conv2d_cl2_mom_chunks: the number of rows (0: the variant does not take
 * the problem -- a shape cplxamd_conv2d_cl2 declines, a grid whose workgroups would change column tile, or
 * CPLXAMD_LAUNCH_SHARED -- use
 * cplxamd_conv2d_cl2 and the layer's own moment pass).  cplxamd_bn_fwd_partials then runs finalize + apply only: one
 * read of y less (csrc/conv_cl2.hip: conv_cl2_kernel<false, true>). */
int64_t cplxamd_conv2d_cl2_mom_chunks(int64_t B, int H, int W, int C, int N, int KH, int KW, int dil_h, int dil_w, int pad_h,
                                      int pad_w);
int cplxamd_conv2d_cl2_mom(const void* x_r, const void* x_i, const void* w_packed, const float* bias_r, const float* bias_i,
                           void* y_r, void* y_i, int64_t B, int H, int W, int C, int N, int KH, int KW, int dil_h, int dil_w,
                           int pad_h, int pad_w, double* partials, int64_t partials_bytes, void* ws, int64_t ws_bytes,
                           void* stream);
int64_t cplxamd_conv2d_cl_wgrad_ws_bytes(int64_t B, int H, int W, int Ci, int Co);
int cplxamd_conv2d_cl_wgrad(const void* g_r, const void* g_i, const void* x_r, const void* x_i, const float* emul,
                            float* dw_r, float* dw_i, int64_t B, int H, int W, int Ci, int Co, int KH, int KW, int dil_h,
                            int dil_w, int pad_h, int pad_w, void* ws, int64_t ws_bytes, void* stream);
/* The six launches above with a per-call launch policy (CPLXAMD_LAUNCH_SHARED: one workgroup per tile / twice the
 * weight-gradient splits; cplxamd_conv2d_cl2_mom declines under it -- CPLXAMD_ESHAPE, cplxamd_conv2d_cl2_mom_chunks_fl 0). */
int cplxamd_conv2d_cl_fl(const void* x_r, const void* x_i, const void* w_packed, const float* bias_r, const float* bias_i,
                         void* y_r, void* y_i, int64_t B, int H, int W, int C, int N, int KH, int KW, int dil_h, int dil_w,
                         int pad_h, int pad_w, int mode, void* ws, int64_t ws_bytes, int flags, void* stream);
int cplxamd_conv2d_cl2_fl(const void* x_r, const void* x_i, const void* w_packed, const float* bias_r, const float* bias_i,
                          void* y_r, void* y_i, int64_t B, int H, int W, int C, int N, int KH, int KW, int dil_h, int dil_w,
                          int pad_h, int pad_w, int mode, void* ws, int64_t ws_bytes, int flags, void* stream);
int cplxamd_conv2d_cl2_lrt_dx_fl(const void* g_r, const void* g_i, const void* w_packed, const void* x_r, const void* x_i,
                                 const void* ga, void* dx_r, void* dx_i, int64_t B, int H, int W, int C, int N, int pad_h,
                                 int pad_w, void* ws, int64_t ws_bytes, int flags, void* stream);
int64_t cplxamd_conv2d_cl2_mom_chunks_fl(int64_t B, int H, int W, int C, int N, int KH, int KW, int dil_h, int dil_w,
                                         int pad_h, int pad_w, int flags);
int cplxamd_conv2d_cl2_mom_fl(const void* x_r, const void* x_i, const void* w_packed, const float* bias_r,
                              const float* bias_i, void* y_r, void* y_i, int64_t B, int H, int W, int C, int N, int KH, int KW,
                              int dil_h, int dil_w, int pad_h, int pad_w, double* partials, int64_t partials_bytes, void* ws,
                              int64_t ws_bytes, int flags, void* stream);
int cplxamd_conv2d_cl_wgrad_fl(const void* g_r, const void* g_i, const void* x_r, const void* x_i, const float* emul,
                               float* dw_r, float* dw_i, int64_t B, int H, int W, int Ci, int Co, int KH, int KW, int dil_h,
                               int dil_w, int pad_h, int pad_w, void* ws, int64_t ws_bytes, int flags, void* stream);
/* REAL-valued twins of the three entry points above (csrc/conv_cl_real.hip, conv_cl_wgrad_real.hip): one plane each,
 * same shapes and conditions.  They carry the variance path of the local-reparameterization convolution layers
 * (conv of |x|^2 with exp(log_sigma2): nn/relevance/complex/base.py:120-135, real/base.py:116-163) and the real
 * Conv2dVD / ARD layers; cplxamd_conv2d_clr_wgrad can multiply the result by emul or exp(emul) ([Co][Ci][3][3] float32:
 * d log_sigma2 = (sum g |x|^2) * exp(log_sigma2)). */
int64_t cplxamd_conv2d_clr_pack_bytes(int N, int C, int KH, int KW);
int64_t cplxamd_conv2d_clr_ws_bytes(int N);
int cplxamd_conv2d_clr_pack(const void* w, void* out, int Co, int Ci, int KH, int KW, int dgrad, void* stream);
int cplxamd_conv2d_clr(const void* x, const void* w_packed, const float* bias, void* y, int64_t B, int H, int W, int C,
                       int N, int KH, int KW, int dil_h, int dil_w, int pad_h, int pad_w, int mode, void* ws,
                       int64_t ws_bytes, void* stream);
int64_t cplxamd_conv2d_clr_wgrad_ws_bytes(int64_t B, int H, int W, int Ci, int Co);
int cplxamd_conv2d_clr_wgrad(const void* g, const void* x, const float* emul, int emul_exp, float* dw, int64_t B, int H,
                             int W, int Ci, int Co, int KH, int KW, int dil_h, int dil_w, int pad_h, int pad_w, void* ws,
                             int64_t ws_bytes, void* stream);
int cplxamd_conv2d_clr_fl(const void* x, const void* w_packed, const float* bias, void* y, int64_t B, int H, int W, int C,
                          int N, int KH, int KW, int dil_h, int dil_w, int pad_h, int pad_w, int mode, void* ws,
                          int64_t ws_bytes, int flags, void* stream);
int cplxamd_conv2d_clr_wgrad_fl(const void* g, const void* x, const float* emul, int emul_exp, float* dw, int64_t B, int H,
                                int W, int Ci, int Co, int KH, int KW, int dil_h, int dil_w, int pad_h, int pad_w, void* ws,
                                int64_t ws_bytes, int flags, void* stream);
/* out[c] = sum over (batch, spatial) of an NCHW tensor (conv bias gradient); ws >= 64*C*8 bytes */
int cplxamd_chansum(const void* x, float* out, int64_t B, int C, int64_t S, int dtype, void* ws,
                    void* stream);
/* The complex bias gradient: out_r[c], out_i[c] (one [2][C] array: out_i == out_r + C) = the same sums of the two
 * planes, both planes per launch (2 launches instead of 4: small models are bound by the NUMBER of dependent
 * launches).  ws >= 2*64*C*8 bytes. */
int cplxamd_chansum2(const void* x_r, const void* x_i, float* out_r, float* out_i, int64_t B, int C, int64_t S, int dtype,
                     void* ws, void* stream);

/* ------------------------------------------------------------------------------------
 * K3  complex batch normalisation (2x2 whitening + 2x2 affine), forward and backward.
 * Replaces: whiten2x2        nn/modules/batchnorm.py:62-123
 *           cplx_batch_norm  nn/modules/batchnorm.py:189-278 (and its autograd backward)
 * Tensors: planar x / y / g [B, F, S] contiguous (S = product of spatial dims, 1 for [B,F]);
 * weight [2,2,F], bias [2,F], running_mean [2,F], running_var [2,2,F] float32 (nullable pairs);
 * saved [8,F] float32 = per-feature (mean_u, mean_v, p, q, w, Vuu, Vuv, Vvv) handed from the
 * forward to the backward.  training != 0: batch statistics (biased covariance, eps on the
 * diagonal) and in-place running-stat update x += momentum (new - x); else running stats.
 * ---------------------------------------------------------------------------------- */
int64_t cplxamd_bn_ws_bytes(int F);
int cplxamd_bn_fwd(const void* xr, const void* xi, void* yr, void* yi, int64_t B, int F,
                   int64_t S, const float* weight, const float* bias, float* running_mean,
                   float* running_var, float* saved, int training, int dtype, float momentum,
                   float eps, void* ws, int64_t ws_bytes, void* stream);
/* cplxamd_bn_fwd that also adds 1 to *tracked_inc (the module's int64 num_batches_tracked; may be NULL) in its finalize
 * launch: one launch less per layer and step than an elementwise add of its own. */
int cplxamd_bn_fwd_ex(const void* xr, const void* xi, void* yr, void* yi, int64_t B, int F,
                      int64_t S, const float* weight, const float* bias, float* running_mean,
                      float* running_var, float* saved, int training, int dtype, float momentum,
                      float eps, int64_t* tracked_inc, void* ws, int64_t ws_bytes, void* stream);
int cplxamd_bn_bwd(const void* gr, const void* gi, const void* xr, const void* xi, void* dxr,
                   void* dxi, int64_t B, int F, int64_t S, const float* weight,
                   const float* saved, float* dweight, float* dbias, int training, int dtype,
                   void* ws, int64_t ws_bytes, void* stream);
/* The same, plus (dx_sums != NULL) the per-feature sums over all rows of the two dX planes as stored, float32 [2][F] --
 * the bias gradient of a convolution / linear layer whose output this batch-norm normalised, for free: the apply pass
 * already holds every dX value.  Only on the row-kernel path (cplxamd_bn_rows_path(B, F, S) == 1: S == 1, F % 8 == 0,
 * F <= 1024, B >= 4096), else CPLXAMD_ESHAPE. */
int cplxamd_bn_rows_path(int64_t B, int F, int64_t S);
int cplxamd_bn_bwd_sums(const void* gr, const void* gi, const void* xr, const void* xi, void* dxr,
                        void* dxi, int64_t B, int F, int64_t S, const float* weight,
                        const float* saved, float* dweight, float* dbias, int training, int dtype,
                        float* dx_sums, void* ws, int64_t ws_bytes, void* stream);

/* ABI 24: cplxamd_bn_bwd_sums that also leaves per-block maxima of |dX| (both planes, as stored) in amax_partial[2048]
 * (float32, ZEROED by the caller; row-kernel path only): cplxamd_absmax_scale_partials(amax_partial, 2048, scale) then gives
 * the {s, 1 / s} that cplxamd_absmax_scale would have computed from dX -- the consumer that cuts dX into half pieces (a
 * float32 convolution's backward on 'x2' pieces) reads the planes once less. */
int cplxamd_bn_bwd_sums_amax(const void* gr, const void* gi, const void* xr, const void* xi, void* dxr, void* dxi, int64_t B,
                             int F, int64_t S, const float* weight, const float* saved, float* dweight, float* dbias,
                             int training, int dtype, float* dx_sums, float* amax_partial, void* ws, int64_t ws_bytes,
                             void* stream);
int cplxamd_absmax_scale_partials(const float* partial, int n, float* scale, void* stream);

/* ABI 24: the backward of a batch-norm layer that directly follows a channels-last 3 x 3 convolution, WITHOUT its apply
 * pass (nn/modules/batchnorm.py:189-278 under autograd followed by the autograd of cplx.py:717-838).  The layer's input
 * gradient is dX = E g + C (x - mu) - k with per-channel 2 x 2 real matrices E, C and constants k:
 *   cplxamd_bn_bwd_coef            sums + finalize of cplxamd_bn_bwd (dweight, dbias as there) and coef[F][12] float32 =
 *                                  (mu mv | e00 e01 e10 e11 | cuu cuv cvv | ku kv | pad); no dX is written.  dx_sums (may be
 *                                  NULL): float32 [2][F], the per-feature sums of dX over all rows -- the bias gradient of the
 *                                  layer that produced x -- from the sums this pass has anyway: E sum(g) - count k (the C term
 *                                  sums to zero against the batch mean and C = 0 in evaluation mode; in training mode the
 *                                  whole expression is zero up to rounding, which is what the reference's float32 sum of dX
 *                                  gives too);
 *   cplxamd_conv2d_cl_wgrad_bn_fl  the convolution's weight gradient, which forms dX tile by tile on the way into its
 *                                  LDS stages (bn_apply's arithmetic with the means multiplied out: the same bf16 values
 *                                  up to one rounding step in a few elements), multiplies it against x, and also stores it
 *                                  (dy_r / dy_i: the data gradient reads it next).  g, z (= the batch-norm layer's input =
 *                                  the convolution's output), dy: [B][Ho][Wo][Co] bf16 channels-last planes; x:
 *                                  [B][H][W][Ci]; shapes and ws as cplxamd_conv2d_cl_wgrad.
 * Against cplxamd_bn_bwd_sums + cplxamd_conv2d_cl_wgrad_fl: one pass over 6 planes less (4 read, 2 written). */
int cplxamd_bn_bwd_coef(const void* gr, const void* gi, const void* xr, const void* xi, int64_t B, int F, int64_t S,
                        const float* weight, const float* saved, float* dweight, float* dbias, int training, int dtype,
                        float* coef, float* dx_sums, void* ws, int64_t ws_bytes, void* stream);
int cplxamd_conv2d_cl_wgrad_bn_fl(const void* g_r, const void* g_i, const void* z_r, const void* z_i, const float* coef,
                                  const void* x_r, const void* x_i, void* dy_r, void* dy_i, float* dw_r, float* dw_i,
                                  int64_t B, int H, int W, int Ci, int Co, int KH, int KW, int dil_h, int dil_w, int pad_h,
                                  int pad_w, void* ws, int64_t ws_bytes, int flags, void* stream);

/* Batch statistics shared between data-parallel ranks ("optional SyncBN"; the reference normalises
 * with the statistics of the local batch only, nn/modules/batchnorm.py:70-99).  A pass is split around ONE
 * all-reduce issued by the caller:
 *   cplxamd_bn_moments  -> this rank's totals, double [F][5] = sum u, v, u^2, v^2, u v (gr == gi == NULL) or
 *                          double [F][6] = the backward sums of g and g x (x - mean) (gr, gi, saved given);
 *   [caller: all-reduce(SUM) of the totals and of the position count B * S, both staying on the device]
 *   cplxamd_bn_fwd_sync / cplxamd_bn_bwd_sync -> finalize + apply from the summed totals (`moments`) and `count`
 *                          (device pointer to ONE double).  Backward: dweight / dbias are this rank's sums (from
 *                          `local_moments`, the array cplxamd_bn_moments wrote before the all-reduce) -- the gradient
 *                          exchange averages them like every other parameter gradient -- while dX uses the totals.
 * Always training-mode statistics; everything else as cplxamd_bn_fwd / cplxamd_bn_bwd_sums. */
int cplxamd_bn_moments(const void* xr, const void* xi, const void* gr, const void* gi, const float* saved,
                       int64_t B, int F, int64_t S, int dtype, double* moments, void* ws, int64_t ws_bytes,
                       void* stream);
int cplxamd_bn_fwd_sync(const void* xr, const void* xi, void* yr, void* yi, int64_t B, int F, int64_t S,
                        const float* weight, const float* bias, float* running_mean, float* running_var,
                        float* saved, int dtype, float momentum, float eps, const double* moments,
                        const double* count, void* ws, int64_t ws_bytes, void* stream);
/* Training-mode forward with the statistics pass done by the producer of x: `partials` = `chunks` rows [row][F][5]
 * float64 of (sum re, sum im, sum re^2, sum im^2, sum re im) over disjoint parts of the batch (what
 * cplxamd_conv2d_cl2_mom writes).  Finalize + apply; otherwise as cplxamd_bn_fwd_ex with training = 1. */
int cplxamd_bn_fwd_partials(const void* xr, const void* xi, void* yr, void* yi, int64_t B, int F, int64_t S,
                            const float* weight, const float* bias, float* running_mean, float* running_var,
                            float* saved, int dtype, float momentum, float eps, int64_t* tracked_inc,
                            const double* partials, int chunks, void* ws, int64_t ws_bytes, void* stream);
int cplxamd_bn_bwd_sync(const void* gr, const void* gi, const void* xr, const void* xi, void* dxr, void* dxi,
                        int64_t B, int F, int64_t S, const float* weight, const float* saved, float* dweight,
                        float* dbias, int dtype, float* dx_sums, const double* moments, const double* local_moments,
                        const double* count, void* ws, int64_t ws_bytes, void* stream);

/* Elementwise complex product (div == 0) or quotient (div != 0) of two planar complex tensors in one launch
 * (Cplx.__mul__ / __truediv__, cplxmodule/cplx.py:135-165: 6 / 12 elementwise torch kernels), with the reference's
 * operation order -- no fused multiply-add -- so float32 values are bit-identical.  conj_b: b is conjugated first;
 * neg: the result is negated (together they give the gradients: d(ab)/da = g conj(b), d(a/b)/da = g / conj(b),
 * d(a/b)/db = -(g conj(a/b)) / conj(b)).  16-byte aligned planes of n elements. */
int cplxamd_cplx_mul(const void* a_r, const void* a_i, const void* b_r, const void* b_i, void* o_r, void* o_i, int64_t n,
                     int div, int conj_b, int neg, int dtype, void* stream);
/* ReLU applied to the real and the imaginary plane (CplxToCplx[torch.nn.ReLU], nn/modules/base.py:167-199) in one launch.
 * bwd == 0: o = relu(a) (NaN passes, as torch).  bwd != 0: a = the saved OUTPUTS, o = (a <= 0 ? 0 : g)
 * (= aten::threshold_backward on the result).  16-byte aligned planes of n elements. */
int cplxamd_split_relu(const void* a_r, const void* a_i, const void* g_r, const void* g_i, void* o_r, void* o_i, int64_t n,
                       int bwd, int dtype, void* stream);
/* ------------------------------------------------------------------------------------
 * layout converters and the non-GEMM layers either side of the path.
 *   cplxamd_deinterleave / _interleave : x[2n] <-> (re[n], im[n])   cplx.py:451-470
 *       (from_interleaved_real(copy=True) / to_interleaved_real along the last dim)
 *   cplxamd_modrelu_fwd / _bwd : y = z * relu(1 - tau / max(|z|, 1e-5))   cplx.py:565-616
 *       tau: tau_numel == 0 -> the value `tau_value`; 1 -> read from the device pointer (a
 *       learnable scalar, no host sync); n -> one threshold per element (pre-broadcast).
 *       bwd writes dz and, if dtau != NULL, the elementwise d/dtau (float32 [n]; the caller
 *       sums it down to the parameter's shape).
 *   cplxamd_cplx_dropout : y = x * keep / (1 - p), ONE Bernoulli(1 - p) draw per complex element
 *       (nn/modules/extra.py:7-25); keep comes from the Philox4x32-7 stream (seed, offset) or
 *       the device pair `state`; applying it to the gradient is the backward.
 * All planes contiguous, 16-byte aligned, n = number of complex elements.
 * ---------------------------------------------------------------------------------- */
int cplxamd_deinterleave(const void* x, void* re, void* im, int64_t n, int dtype, void* stream);
int cplxamd_interleave(const void* re, const void* im, void* out, int64_t n, int dtype, void* stream);
int cplxamd_modrelu_fwd(const void* zr, const void* zi, const float* tau, float tau_value, int tau_numel,
                        void* yr, void* yi, int64_t n, int dtype, void* stream);
int cplxamd_modrelu_bwd(const void* zr, const void* zi, const float* tau, float tau_value, int tau_numel,
                        const void* gr, const void* gi, void* dzr, void* dzi, float* dtau, int64_t n,
                        int dtype, void* stream);
int cplxamd_cplx_dropout(const void* xr, const void* xi, void* yr, void* yi, double p, uint64_t seed,
                         uint64_t offset, const uint64_t* state, int64_t n, int dtype, void* stream);

/* Complex abs-max pooling (cplx.max_poolnd, cplx.py:1114-1175): in every window the element of
 * largest modulus keeps both its parts (first maximum in row-major window order, as torch).
 * pool = int[14]: B, C, H, W, Ho, Wo, kh, kw, sh, sw, ph, pw, dh, dw.  idx: int32 [B, C, Ho, Wo],
 * the selected position h * W + w, written by fwd and read by bwd (a deterministic gather). */
int cplxamd_cplx_maxpool2d_fwd(const void* zr, const void* zi, void* yr, void* yi, int32_t* idx,
                               const int* pool, int dtype, void* stream);
int cplxamd_cplx_maxpool2d_bwd(const void* gr, const void* gi, const int32_t* idx, void* dzr, void* dzi,
                               const int* pool, int dtype, void* stream);
/* the same pair on channels-last planes ([B][H][W][C] in, [B][Ho][Wo][C] out and idx): a pooling layer inside a
 * channels-last convolution stack needs no layout copy */
int cplxamd_cplx_maxpool2d_fwd_cl(const void* zr, const void* zi, void* yr, void* yi, int32_t* idx,
                                  const int* pool, int dtype, void* stream);
int cplxamd_cplx_maxpool2d_bwd_cl(const void* gr, const void* gi, const int32_t* idx, void* dzr, void* dzi,
                                  const int* pool, int dtype, void* stream);

/* Bilinear layers: cplx.bilinear (cplx.py:1062-1087), the variance term of
 * CplxBilinearGaussian / BilinearGaussian (nn/relevance/complex/base.py:59-84, real/base.py:52-77).
 *   y[b,o] = sum_i u[b,i] T[b,o,i] + bias[o],   u = conj(x1) if conj_u else x1,
 * where T[b,(o,i)] = sum_j W[o,i,j] x2[b,j] comes from cplxamd_cgemm / cplxamd_rgemm with the
 * weight read as stored ([O*I1, I2]).  u: [B, I1], T: [B, O, I1], y, g: [B, O], all contiguous and
 * of one dtype; bias float32 [O] or NULL.  Real tensors: every imaginary plane NULL.
 * bwd: dT = g conj(u) (real: g u), du = sum_o g conj(T) mapped back to x1 (conjugated again when
 * conj_u); either output pair may be NULL (T may then be NULL too when du is not wanted). */
int cplxamd_bilinear_reduce_fwd(const void* ur, const void* ui, const void* tr, const void* ti,
                                const float* bias_r, const float* bias_i, void* yr, void* yi, int64_t B,
                                int O, int I1, int conj_u, int dtype, void* stream);
int cplxamd_bilinear_reduce_bwd(const void* ur, const void* ui, const void* tr, const void* ti,
                                const void* gr, const void* gi, void* dx1r, void* dx1i, void* dtr,
                                void* dti, int64_t B, int O, int I1, int conj_u, int dtype, void* stream);

/* ------------------------------------------------------------------------------------
 * float64 (ABI 23; csrc/f64.hip): the contractions of the reference's `.double()` models and the exponential integral,
 * as a PARITY mode -- plain v_fma_f64 kernels, nothing tuned.  The float64 layers of the host package
 * (cplxmodule_amd/f64.py) spell their elementwise algebra with torch ops under autograd, as the reference does, and call
 * these for what torch would hand to a vendor library (or to scipy on the host).
 *   cplxamd_gemm_f64     C[z][m][n] = sum_k A[z][m][k] op(B[z][n][k]) (+ bias[n]); element strides (row, column, batch) per
 *                        operand, planar complex (a_i / b_i / c_i all non-NULL) or real (all NULL); conj_b: conj(B).
 *                        cplx.py:634-648, :167-174 and their autograd (dX = G conj(W), dW = G^T conj(X)).
 *   cplxamd_conv2d_f64   mode 0: y = x (*) w + bias (p = x, q = w); mode 1: dx = data gradient (p = g, q = w); mode 2:
 *                        dw = weight gradient (p = g, q = x); NCHW planes, complex or real, any stride / padding / dilation /
 *                        groups; geom = {B, Ci, Co, H, W, KH, KW, sh, sw, ph, pw, dh, dw, groups}.  cplx.py:717-838.
 *   cplxamd_expi_f64     y = Ei(x), both signs, ~1e-15 relative to scipy.special.expi.  nn/relevance/complex/vd.py:15-44.
 * ---------------------------------------------------------------------------------- */
int cplxamd_gemm_f64(const double* a_r, const double* a_i, int64_t a_rs, int64_t a_cs, int64_t a_bs,
                     const double* b_r, const double* b_i, int64_t b_rs, int64_t b_cs, int64_t b_bs,
                     const double* bias_r, const double* bias_i, double* c_r, double* c_i, int64_t ldc, int64_t c_bs,
                     int batch, int M, int N, int K, int conj_b, void* stream);
int cplxamd_conv2d_f64(const double* p_r, const double* p_i, const double* q_r, const double* q_i, const double* bias_r,
                       const double* bias_i, double* out_r, double* out_i, const int* geom, int mode, void* stream);
int cplxamd_expi_f64(const double* x, double* y, int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CPLXAMD_H */
