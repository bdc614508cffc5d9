// C-ABI dispatch for the complex / real GEMM entry points (include/cplxamd.h).
#include "gemm.h"

#include <stdlib.h>

namespace cplxamd {
int g_gemm_persistent = 1;
// default: every launch kind, subject to the K-depth rule of gemm_bf16_w4.hip (profiles/r04_gemm_w4_ab.txt)
static int env_w4() { const char* e = getenv("CPLXAMD_GEMM_W4"); return e ? (int)strtol(e, nullptr, 0) : 0x3f; }
int g_gemm_w4 = env_w4();
}
using namespace cplxamd;

extern "C" {

int cplxamd_gemm_set_persistent(int on) {
  const int prev = g_gemm_persistent;
  g_gemm_persistent = on ? 1 : 0;
  return prev;
}

int cplxamd_gemm_set_family(int w4) {
  const int prev = g_gemm_w4;
  g_gemm_w4 = w4 < 0 ? 0x7f : (w4 & 0x7f);
  return prev;
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
  if (!a_r || !a_i || !b_r || !b_i || !c_r || !c_i) return CPLXAMD_EINVAL;
  if (M < 0 || N < 0 || K < 0 || ldc < N) return CPLXAMD_EINVAL;
  if ((bias_r == nullptr) != (bias_i == nullptr)) return CPLXAMD_EINVAL;
  if (accumulate && out_dtype != CPLXAMD_F32) return CPLXAMD_EINVAL;
  if (algo != CPLXAMD_ALGO_4M && algo != CPLXAMD_ALGO_3M) return CPLXAMD_EINVAL;
  if (emul && (out_dtype != CPLXAMD_F32 || algo != CPLXAMD_ALGO_4M)) return CPLXAMD_EINVAL;
  GemmArgs g{a_r, a_i, a_rs, a_cs, b_r, b_i, b_rs, b_cs, bias_r, bias_i, emul,
             c_r, c_i, ldc, M, N, K, conj_b ? 1 : 0, accumulate ? 1 : 0};
  g.ws = ws; g.ws_bytes = ws_bytes; g.beta = accumulate ? beta : nullptr; g.emul_both = emul ? 1 : 0;
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
  if (!g_r || !g_i || !w_r || !w_i || !x_r || !x_i || !ga || !dx_r || !dx_i) return CPLXAMD_EINVAL;
  if (M < 0 || N < 0 || K < 0 || ldc < N || ldx < N) return CPLXAMD_EINVAL;
  if (dtype != CPLXAMD_BF16) return CPLXAMD_ESHAPE;
  GemmArgs g{g_r, g_i, g_rs, g_cs, w_r, w_i, w_rs, w_cs, nullptr, nullptr, nullptr,
             dx_r, dx_i, ldc, M, N, K, 1, 0};
  g.fx_r = x_r; g.fx_i = x_i; g.fga = ga; g.fld = ldx;
  return launch_gemm_bf16<true>(g, CPLXAMD_BF16, (hipStream_t)stream);
}

int cplxamd_rgemm_lrt_dx(const void* gg, int64_t g_rs, int64_t g_cs, const void* w, int64_t w_rs, int64_t w_cs,
                         const void* x, const void* ga, int64_t ldx, void* dx, int64_t ldc, int M, int N, int K, int dtype,
                         void* stream) {
  if (!gg || !w || !x || !ga || !dx) return CPLXAMD_EINVAL;
  if (M < 0 || N < 0 || K < 0 || ldc < N || ldx < N) return CPLXAMD_EINVAL;
  if (dtype != CPLXAMD_BF16) return CPLXAMD_ESHAPE;
  GemmArgs g{gg, nullptr, g_rs, g_cs, w, nullptr, w_rs, w_cs, nullptr, nullptr, nullptr,
             dx, nullptr, ldc, M, N, K, 0, 0};
  g.fx_r = x; g.fx_i = nullptr; g.fga = ga; g.fld = ldx;
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
  if (!a || !b || !c) return CPLXAMD_EINVAL;
  if (M < 0 || N < 0 || K < 0 || ldc < N) return CPLXAMD_EINVAL;
  if (accumulate && out_dtype != CPLXAMD_F32) return CPLXAMD_EINVAL;
  GemmArgs g{a, nullptr, a_rs, a_cs, b, nullptr, b_rs, b_cs, bias, nullptr, emul,
             c, nullptr, ldc, M, N, K, 0, accumulate ? 1 : 0};
  g.ws = ws; g.ws_bytes = ws_bytes; g.beta = accumulate ? beta : nullptr; g.emul_exp = (emul && emul_exp) ? 1 : 0;
  hipStream_t st = (hipStream_t)stream;
  if (in_dtype == CPLXAMD_BF16) {
    const int rc = launch_gemm_bf16<false>(g, out_dtype, st);
    if (rc != CPLXAMD_ESHAPE) return rc;
  }
  return launch_gemm_generic<false>(g, in_dtype, out_dtype, st);
}

}  // extern "C"
