// C-ABI dispatch for the complex / real GEMM entry points (include/cplxamd.h).
#include "gemm.h"

#include <stdlib.h>

namespace cplxamd {
// DEPRECATED process-wide defaults (launch.h): read only by launches whose flags leave the choice open
std::atomic<int> g_default_persistent{1};
// default: every launch kind, subject to the K-depth rule of gemm_bf16_w4.hip (profiles/r04_gemm_w4_ab.txt)
static int env_w4() { const char* e = getenv("CPLXAMD_GEMM_W4"); return e ? (int)strtol(e, nullptr, 0) & 0xff : 0xbf; }
std::atomic<int> g_default_family{env_w4()};
}
using namespace cplxamd;

extern "C" {

int cplxamd_gemm_set_persistent(int on) {
  return g_default_persistent.exchange(on ? 1 : 0);
}

int cplxamd_gemm_set_family(int w4) {
  return g_default_family.exchange(w4 < 0 ? 0xff : (w4 & 0xff));
}

/* scratch the bf16 path wants for split-K at this shape (0 = none) */
int64_t cplxamd_gemm_ws_bytes(int M, int N, int K, int cplx, int in_dtype, int out_dtype) {
  // the generic kernel is the fallback of every bf16 shape as well: ask for the larger of the two
  const int64_t gen = gemm_generic_ws_bytes(M, N, K, cplx != 0);
  if (in_dtype != CPLXAMD_BF16 || out_dtype != CPLXAMD_F32) return gen;
  const int64_t fast = gemm_bf16_ws_bytes(M, N, K, cplx != 0);
  return fast > gen ? fast : gen;
}

int64_t cplxamd_cgemm3m_ws_bytes(int M, int N, int K) { return gemm_bf16_gauss_ws_bytes(M, N, K); }

int cplxamd_cgemm(const void* a_r, const void* a_i, int64_t a_rs, int64_t a_cs,
                  const void* b_r, const void* b_i, int64_t b_rs, int64_t b_cs,
                  const float* bias_r, const float* bias_i, void* c_r, void* c_i, int64_t ldc,
                  int M, int N, int K, int conj_b, int in_dtype, int out_dtype, int accumulate,
                  int algo, void* ws, int64_t ws_bytes, void* stream) {
  return cplxamd_cgemm_ex(a_r, a_i, a_rs, a_cs, b_r, b_i, b_rs, b_cs, bias_r, bias_i, nullptr, c_r, c_i, ldc, M, N, K,
                          conj_b, in_dtype, out_dtype, accumulate, nullptr, algo, ws, ws_bytes, stream);
}

int cplxamd_cgemm_ex(const void* a_r, const void* a_i, int64_t a_rs, int64_t a_cs,
                     const void* b_r, const void* b_i, int64_t b_rs, int64_t b_cs,
                     const float* bias_r, const float* bias_i, const float* emul, void* c_r, void* c_i, int64_t ldc,
                     int M, int N, int K, int conj_b, int in_dtype, int out_dtype, int accumulate,
                     const float* beta, int algo, void* ws, int64_t ws_bytes, void* stream) {
  return cplxamd_cgemm_fl(a_r, a_i, a_rs, a_cs, b_r, b_i, b_rs, b_cs, bias_r, bias_i, emul, c_r, c_i, ldc, M, N, K, conj_b,
                          in_dtype, out_dtype, accumulate, beta, algo, ws, ws_bytes, CPLXAMD_LAUNCH_DEFAULT, stream);
}

int cplxamd_cgemm_fl(const void* a_r, const void* a_i, int64_t a_rs, int64_t a_cs,
                     const void* b_r, const void* b_i, int64_t b_rs, int64_t b_cs,
                     const float* bias_r, const float* bias_i, const float* emul, void* c_r, void* c_i, int64_t ldc,
                     int M, int N, int K, int conj_b, int in_dtype, int out_dtype, int accumulate,
                     const float* beta, int algo, void* ws, int64_t ws_bytes, int flags, void* stream) {
  if (!launch_flags_ok(flags)) return CPLXAMD_EINVAL;
  if (!a_r || !a_i || !b_r || !b_i || !c_r || !c_i) return CPLXAMD_EINVAL;
  if (M < 0 || N < 0 || K < 0 || ldc < N) return CPLXAMD_EINVAL;
  if ((bias_r == nullptr) != (bias_i == nullptr)) return CPLXAMD_EINVAL;
  if (accumulate && out_dtype != CPLXAMD_F32) return CPLXAMD_EINVAL;
  if (algo != CPLXAMD_ALGO_4M && algo != CPLXAMD_ALGO_3M) return CPLXAMD_EINVAL;
  if (emul && (out_dtype != CPLXAMD_F32 || algo != CPLXAMD_ALGO_4M)) return CPLXAMD_EINVAL;
  GemmArgs g{a_r, a_i, a_rs, a_cs, b_r, b_i, b_rs, b_cs, bias_r, bias_i, emul,
             c_r, c_i, ldc, M, N, K, conj_b ? 1 : 0, accumulate ? 1 : 0};
  g.ws = ws; g.ws_bytes = ws_bytes; g.beta = accumulate ? beta : nullptr; g.emul_both = emul ? 1 : 0;
  g.flags = flags;
  hipStream_t st = (hipStream_t)stream;
  if (algo == CPLXAMD_ALGO_3M)     // Gauss: dense bf16 operands only, never a silent 4M fallback
    return in_dtype == CPLXAMD_BF16 ? launch_gemm_bf16_gauss(g, out_dtype, st) : CPLXAMD_ESHAPE;
  if (in_dtype == CPLXAMD_BF16) {
    const int rc = launch_gemm_bf16<true>(g, out_dtype, st);
    if (rc != CPLXAMD_ESHAPE) return rc;
  }
  return launch_gemm_generic<true>(g, in_dtype, out_dtype, st);
}

int cplxamd_cgemm_lrt_dx(const void* g_r, const void* g_i, int64_t g_rs, int64_t g_cs,
                         const void* w_r, const void* w_i, int64_t w_rs, int64_t w_cs,
                         const void* x_r, const void* x_i, const void* ga, int64_t ldx,
                         void* dx_r, void* dx_i, int64_t ldc, int M, int N, int K, int dtype, void* stream) {
  return cplxamd_cgemm_lrt_dx_fl(g_r, g_i, g_rs, g_cs, w_r, w_i, w_rs, w_cs, x_r, x_i, ga, ldx, dx_r, dx_i, ldc, M, N, K, dtype,
                                 CPLXAMD_LAUNCH_DEFAULT, stream);
}

int cplxamd_cgemm_lrt_dx_fl(const void* g_r, const void* g_i, int64_t g_rs, int64_t g_cs,
                            const void* w_r, const void* w_i, int64_t w_rs, int64_t w_cs,
                            const void* x_r, const void* x_i, const void* ga, int64_t ldx,
                            void* dx_r, void* dx_i, int64_t ldc, int M, int N, int K, int dtype, int flags, void* stream) {
  if (!launch_flags_ok(flags)) return CPLXAMD_EINVAL;
  if (!g_r || !g_i || !w_r || !w_i || !x_r || !x_i || !ga || !dx_r || !dx_i) return CPLXAMD_EINVAL;
  if (M < 0 || N < 0 || K < 0 || ldc < N || ldx < N) return CPLXAMD_EINVAL;
  if (dtype != CPLXAMD_BF16) return CPLXAMD_ESHAPE;
  GemmArgs g{g_r, g_i, g_rs, g_cs, w_r, w_i, w_rs, w_cs, nullptr, nullptr, nullptr,
             dx_r, dx_i, ldc, M, N, K, 1, 0};
  g.fx_r = x_r; g.fx_i = x_i; g.fga = ga; g.fld = ldx; g.flags = flags;
  return launch_gemm_bf16<true>(g, CPLXAMD_BF16, (hipStream_t)stream);
}

int cplxamd_rgemm_lrt_dx(const void* gg, int64_t g_rs, int64_t g_cs, const void* w, int64_t w_rs, int64_t w_cs,
                         const void* x, const void* ga, int64_t ldx, void* dx, int64_t ldc, int M, int N, int K, int dtype,
                         void* stream) {
  return cplxamd_rgemm_lrt_dx_fl(gg, g_rs, g_cs, w, w_rs, w_cs, x, ga, ldx, dx, ldc, M, N, K, dtype, CPLXAMD_LAUNCH_DEFAULT,
                                 stream);
}

int cplxamd_rgemm_lrt_dx_fl(const void* gg, int64_t g_rs, int64_t g_cs, const void* w, int64_t w_rs, int64_t w_cs,
                            const void* x, const void* ga, int64_t ldx, void* dx, int64_t ldc, int M, int N, int K, int dtype,
                            int flags, void* stream) {
  if (!launch_flags_ok(flags)) return CPLXAMD_EINVAL;
  if (!gg || !w || !x || !ga || !dx) return CPLXAMD_EINVAL;
  if (M < 0 || N < 0 || K < 0 || ldc < N || ldx < N) return CPLXAMD_EINVAL;
  if (dtype != CPLXAMD_BF16) return CPLXAMD_ESHAPE;
  GemmArgs g{gg, nullptr, g_rs, g_cs, w, nullptr, w_rs, w_cs, nullptr, nullptr, nullptr,
             dx, nullptr, ldc, M, N, K, 0, 0};
  g.fx_r = x; g.fx_i = nullptr; g.fga = ga; g.fld = ldx; g.flags = flags;
  return launch_gemm_bf16<false>(g, CPLXAMD_BF16, (hipStream_t)stream);
}

int cplxamd_cgemm_batched(const void* a_r, const void* a_i, int64_t a_rs, int64_t a_cs, int64_t a_bs,
                          const void* b_r, const void* b_i, int64_t b_rs, int64_t b_cs, int64_t b_bs,
                          void* c_r, void* c_i, int64_t ldc, int64_t c_bs, int batch, int M, int N, int K,
                          int conj_b, int in_dtype, int out_dtype, void* stream) {
  if (!a_r || !a_i || !b_r || !b_i || !c_r || !c_i) return CPLXAMD_EINVAL;
  if (batch < 0 || M < 0 || N < 0 || K < 0 || ldc < N) return CPLXAMD_EINVAL;
  if (batch == 0) return 0;
  GemmArgs g{a_r, a_i, a_rs, a_cs, b_r, b_i, b_rs, b_cs, nullptr, nullptr, nullptr,
             c_r, c_i, ldc, M, N, K, conj_b ? 1 : 0, 0};
  g.batch = batch; g.a_bs = a_bs; g.b_bs = b_bs; g.c_bs = c_bs;
  return launch_gemm_generic<true>(g, in_dtype, out_dtype, (hipStream_t)stream);
}

int cplxamd_rgemm(const void* a, int64_t a_rs, int64_t a_cs, const void* b, int64_t b_rs,
                  int64_t b_cs, const float* bias, const float* emul, void* c, int64_t ldc,
                  int M, int N, int K, int in_dtype, int out_dtype, int accumulate,
                  void* ws, int64_t ws_bytes, void* stream) {
  return cplxamd_rgemm_ex(a, a_rs, a_cs, b, b_rs, b_cs, bias, emul, 0, c, ldc, M, N, K, in_dtype, out_dtype,
                          accumulate, nullptr, ws, ws_bytes, stream);
}

int cplxamd_rgemm_ex(const void* a, int64_t a_rs, int64_t a_cs, const void* b, int64_t b_rs,
                     int64_t b_cs, const float* bias, const float* emul, int emul_exp, void* c, int64_t ldc,
                     int M, int N, int K, int in_dtype, int out_dtype, int accumulate, const float* beta,
                     void* ws, int64_t ws_bytes, void* stream) {
  return cplxamd_rgemm_fl(a, a_rs, a_cs, b, b_rs, b_cs, bias, emul, emul_exp, c, ldc, M, N, K, in_dtype, out_dtype, accumulate,
                          beta, ws, ws_bytes, CPLXAMD_LAUNCH_DEFAULT, stream);
}

int cplxamd_rgemm_fl(const void* a, int64_t a_rs, int64_t a_cs, const void* b, int64_t b_rs,
                     int64_t b_cs, const float* bias, const float* emul, int emul_exp, void* c, int64_t ldc,
                     int M, int N, int K, int in_dtype, int out_dtype, int accumulate, const float* beta,
                     void* ws, int64_t ws_bytes, int flags, void* stream) {
  if (!launch_flags_ok(flags)) return CPLXAMD_EINVAL;
  if (!a || !b || !c) return CPLXAMD_EINVAL;
  if (M < 0 || N < 0 || K < 0 || ldc < N) return CPLXAMD_EINVAL;
  if (accumulate && out_dtype != CPLXAMD_F32) return CPLXAMD_EINVAL;
  GemmArgs g{a, nullptr, a_rs, a_cs, b, nullptr, b_rs, b_cs, bias, nullptr, emul,
             c, nullptr, ldc, M, N, K, 0, accumulate ? 1 : 0};
  g.ws = ws; g.ws_bytes = ws_bytes; g.beta = accumulate ? beta : nullptr; g.emul_exp = (emul && emul_exp) ? 1 : 0;
  g.flags = flags;
  hipStream_t st = (hipStream_t)stream;
  if (in_dtype == CPLXAMD_BF16) {
    const int rc = launch_gemm_bf16<false>(g, out_dtype, st);
    if (rc != CPLXAMD_ESHAPE) return rc;
  }
  return launch_gemm_generic<false>(g, in_dtype, out_dtype, st);
}

/* the GEMMs of the half-precision split products (include/cplxamd.h): IEEE-half or bf16 operands, float32 output, the
 * operands' power-of-two scales undone behind the K loop.  No generic fallback: CPLXAMD_ESHAPE = take the bf16 pieces. */
int cplxamd_cgemm_sc_fl(const void* a_r, const void* a_i, int64_t a_rs, int64_t a_cs,
                        const void* b_r, const void* b_i, int64_t b_rs, int64_t b_cs,
                        const float* bias_r, const float* bias_i, const float* emul, void* c_r, void* c_i, int64_t ldc,
                        int M, int N, int K, int conj_b, int in_dtype, int accumulate, const float* beta,
                        const float* scale_a, const float* scale_b, void* ws, int64_t ws_bytes, int flags, void* stream) {
  if (!launch_flags_ok(flags)) return CPLXAMD_EINVAL;
  if (!a_r || !a_i || !b_r || !b_i || !c_r || !c_i) return CPLXAMD_EINVAL;
  if (M < 0 || N < 0 || K < 0 || ldc < N) return CPLXAMD_EINVAL;
  if ((bias_r == nullptr) != (bias_i == nullptr) || (scale_a == nullptr) != (scale_b == nullptr)) return CPLXAMD_EINVAL;
  if (in_dtype != CPLXAMD_F16) return CPLXAMD_EINVAL;
  GemmArgs g{a_r, a_i, a_rs, a_cs, b_r, b_i, b_rs, b_cs, bias_r, bias_i, emul,
             c_r, c_i, ldc, M, N, K, conj_b ? 1 : 0, accumulate ? 1 : 0};
  g.ws = ws; g.ws_bytes = ws_bytes; g.beta = accumulate ? beta : nullptr; g.emul_both = emul ? 1 : 0;
  g.scale_a = scale_a; g.scale_b = scale_b; g.flags = flags;
  return launch_gemm_f16<true>(g, CPLXAMD_F32, (hipStream_t)stream);
}

int cplxamd_rgemm_sc_fl(const void* a, int64_t a_rs, int64_t a_cs, const void* b, int64_t b_rs, int64_t b_cs,
                        const float* bias, const float* emul, int emul_exp, void* c, int64_t ldc, int M, int N, int K,
                        int in_dtype, int accumulate, const float* beta, const float* scale_a, const float* scale_b,
                        void* ws, int64_t ws_bytes, int flags, void* stream) {
  if (!launch_flags_ok(flags)) return CPLXAMD_EINVAL;
  if (!a || !b || !c) return CPLXAMD_EINVAL;
  if (M < 0 || N < 0 || K < 0 || ldc < N || (scale_a == nullptr) != (scale_b == nullptr)) return CPLXAMD_EINVAL;
  if (in_dtype != CPLXAMD_F16) return CPLXAMD_EINVAL;
  GemmArgs g{a, nullptr, a_rs, a_cs, b, nullptr, b_rs, b_cs, bias, nullptr, emul,
             c, nullptr, ldc, M, N, K, 0, accumulate ? 1 : 0};
  g.ws = ws; g.ws_bytes = ws_bytes; g.beta = accumulate ? beta : nullptr; g.emul_exp = (emul && emul_exp) ? 1 : 0;
  g.scale_a = scale_a; g.scale_b = scale_b; g.flags = flags;
  return launch_gemm_f16<false>(g, CPLXAMD_F32, (hipStream_t)stream);
}

/* dispatch of a bf16 GEMM call as a pure function (include/cplxamd.h): the launchers run "dry" (GemmArgs::plan) */
int cplxamd_gemm_plan(int cplx, int M, int N, int K, int ta, int tb, int out_dtype, int epi, int flags, int ncu) {
  if (!launch_flags_ok(flags) || M <= 0 || N <= 0 || K <= 0 || epi < 0 || epi > 2) return CPLXAMD_EINVAL;
  if (out_dtype != CPLXAMD_BF16 && out_dtype != CPLXAMD_F32) return CPLXAMD_EINVAL;
  if ((epi == 1 && out_dtype != CPLXAMD_BF16) || (epi == 2 && out_dtype != CPLXAMD_F32)) return CPLXAMD_EINVAL;
  // stand-in operands: dense, 256-byte aligned addresses that nothing dereferences
  void* const p = reinterpret_cast<void*>(uintptr_t{1} << 20);
  const float* const pf = reinterpret_cast<const float*>(p);
  GemmArgs g{p, cplx ? p : nullptr, ta ? 1 : K, ta ? M : 1, p, cplx ? p : nullptr, tb ? 1 : K, tb ? N : 1,
             nullptr, nullptr, (epi == 2 && !cplx) ? pf : nullptr, p, cplx ? p : nullptr, N, M, N, K, (cplx && (epi != 0 || tb)) ? 1 : 0, epi == 2};   // (the layers' transposed-weight launches are the conjugated ones)
  if (epi == 1) { g.fx_r = p; g.fx_i = cplx ? p : nullptr; g.fga = p; g.fld = N; }
  if (epi == 2) { g.beta = pf; g.emul_exp = cplx ? 0 : 1; }   // the layers' weight gradients: + beta * dKL (real: times exp(log_sigma2))
  g.ws = p; g.ws_bytes = INT64_MAX; g.flags = flags;
  int code = 0;
  g.plan = &code; g.ncu = ncu;
  const int rc = cplx ? launch_gemm_bf16<true>(g, out_dtype, nullptr) : launch_gemm_bf16<false>(g, out_dtype, nullptr);
  if (rc == CPLXAMD_ESHAPE) return 0;
  return rc ? rc : code;
}

}  // extern "C"
