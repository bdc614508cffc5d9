// The drop-in boundary without Python or torch: a C++ program that links libcplxamd.so through include/cplxamd.h only,
// runs the complex linear map of cplxmodule/cplx.py:634-648 (y = x W^T + b, W [out, in]) on device buffers it allocated
// itself -- float32 operands (the generic kernel) and bf16 operands (the MFMA kernels) -- and checks both against plain
// loops on the host; then (ABI 19) the same entry point with per-call launch flags from two threads on two streams.
// Build + run (tests/test_gpu_r04.py does exactly this on the GPU box):
//   hipcc --offload-arch=gfx950 -O2 -I include examples/cabi_linear.cpp -L cplxmodule_amd -lcplxamd \
//         -Wl,-rpath,$PWD/cplxmodule_amd -o /tmp/cabi_linear && /tmp/cabi_linear
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <cmath>
#include <thread>
#include <vector>

#include "cplxamd.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

static uint16_t to_bf16(float f) {                 // round to nearest even
  uint32_t u; memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static float from_bf16(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

template <typename T> static int upload(void** dst, const std::vector<T>& src) {
  HIP_OK(hipMalloc(dst, src.size() * sizeof(T)));
  HIP_OK(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
  return 0;
}

int main() {
  if (cplxamd_abi_version() != CPLXAMD_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 2; }
  const int B = 256, I = 256, O = 256;              // one 256 x 128 MFMA tile column per plane pair, K % 64 == 0
  std::vector<float> xr(B * I), xi(B * I), wr(O * I), wi(O * I), br(O), bi(O);
  uint32_t s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((int)(s >> 9) - (1 << 22)) / (float)(1 << 22); };
  for (auto* v : {&xr, &xi}) for (auto& e : *v) e = from_bf16(to_bf16(rnd()));           // bf16-representable values:
  for (auto* v : {&wr, &wi}) for (auto& e : *v) e = from_bf16(to_bf16(0.1f * rnd()));    // both runs see the same numbers
  for (auto* v : {&br, &bi}) for (auto& e : *v) e = rnd();
  std::vector<double> ref_r(B * O), ref_i(B * O);
  for (int b = 0; b < B; ++b)
    for (int o = 0; o < O; ++o) {
      double sr = br[o], si = bi[o];
      for (int k = 0; k < I; ++k) {
        sr += (double)xr[b * I + k] * wr[o * I + k] - (double)xi[b * I + k] * wi[o * I + k];
        si += (double)xr[b * I + k] * wi[o * I + k] + (double)xi[b * I + k] * wr[o * I + k];
      }
      ref_r[b * O + o] = sr; ref_i[b * O + o] = si;
    }
  double scale = 0;
  for (double v : ref_r) scale = std::fmax(scale, std::fabs(v));

  void *dxr, *dxi, *dwr, *dwi, *dbr, *dbi, *dyr, *dyi;
  if (upload(&dxr, xr) || upload(&dxi, xi) || upload(&dwr, wr) || upload(&dwi, wi) || upload(&dbr, br) || upload(&dbi, bi)) return 2;
  HIP_OK(hipMalloc(&dyr, B * O * 4)); HIP_OK(hipMalloc(&dyi, B * O * 4));
  std::vector<float> yr(B * O), yi(B * O);

  // float32 in, float32 out
  int rc = cplxamd_cgemm(dxr, dxi, I, 1, dwr, dwi, I, 1, (const float*)dbr, (const float*)dbi, dyr, dyi, O, B, O, I, 0,
                         CPLXAMD_F32, CPLXAMD_F32, 0, CPLXAMD_ALGO_4M, nullptr, 0, nullptr);
  if (rc) { fprintf(stderr, "cplxamd_cgemm(f32): error %d\n", rc); return 1; }
  HIP_OK(hipDeviceSynchronize());
  HIP_OK(hipMemcpy(yr.data(), dyr, B * O * 4, hipMemcpyDeviceToHost)); HIP_OK(hipMemcpy(yi.data(), dyi, B * O * 4, hipMemcpyDeviceToHost));
  double e32 = 0;
  for (int j = 0; j < B * O; ++j) e32 = std::fmax(e32, std::fmax(std::fabs(yr[j] - ref_r[j]), std::fabs(yi[j] - ref_i[j])));
  printf("float32 operands: max |err| / max |ref| = %.3g\n", e32 / scale);

  // bf16 in (the same values), float32 out: the MFMA kernels
  std::vector<uint16_t> hxr(B * I), hxi(B * I), hwr(O * I), hwi(O * I);
  for (int j = 0; j < B * I; ++j) { hxr[j] = to_bf16(xr[j]); hxi[j] = to_bf16(xi[j]); }
  for (int j = 0; j < O * I; ++j) { hwr[j] = to_bf16(wr[j]); hwi[j] = to_bf16(wi[j]); }
  void *bxr, *bxi, *bwr, *bwi;
  if (upload(&bxr, hxr) || upload(&bxi, hxi) || upload(&bwr, hwr) || upload(&bwi, hwi)) return 2;
  rc = cplxamd_cgemm(bxr, bxi, I, 1, bwr, bwi, I, 1, (const float*)dbr, (const float*)dbi, dyr, dyi, O, B, O, I, 0,
                     CPLXAMD_BF16, CPLXAMD_F32, 0, CPLXAMD_ALGO_4M, nullptr, 0, nullptr);
  if (rc) { fprintf(stderr, "cplxamd_cgemm(bf16): error %d\n", rc); return 1; }
  HIP_OK(hipDeviceSynchronize());
  HIP_OK(hipMemcpy(yr.data(), dyr, B * O * 4, hipMemcpyDeviceToHost)); HIP_OK(hipMemcpy(yi.data(), dyi, B * O * 4, hipMemcpyDeviceToHost));
  double e16 = 0;
  for (int j = 0; j < B * O; ++j) e16 = std::fmax(e16, std::fmax(std::fabs(yr[j] - ref_r[j]), std::fabs(yi[j] - ref_i[j])));
  printf("bf16 operands   : max |err| / max |ref| = %.3g\n", e16 / scale);
  bool ok = e32 <= 1e-5 * scale && e16 <= 1e-5 * scale;       // exact products of bf16 values, float32 accumulation

  // ABI 19: the launch form is an ARGUMENT.  Two host threads, two HIP streams, opposite policies (one launching as if a
  // collective held compute units, one as the owner of the chip), at the same time, on one larger problem -- the library
  // keeps no launch state, so neither sees the other's choice, and the results are the same bits.
  {
    const int Bb = 4096, Kb = 512, Nb = 4096;       // 16 x 32 complex tiles: more than one per CU
    std::vector<uint16_t> ha((size_t)Bb * Kb), hb((size_t)Nb * Kb);
    for (auto& e : ha) e = to_bf16(rnd());
    for (auto& e : hb) e = to_bf16(0.1f * rnd());
    void *a_r, *a_i, *b_r, *b_i;
    if (upload(&a_r, ha) || upload(&a_i, ha) || upload(&b_r, hb) || upload(&b_i, hb)) return 2;
    void* out[3][2];
    for (auto& o : out) for (auto& pl : o) HIP_OK(hipMalloc(&pl, (size_t)Bb * Nb * 2));
    const int fam0 = CPLXAMD_LAUNCH_FAMILY(0);      // the 8-wave family: it has both forms for this launch
    const int kinds[2] = {cplxamd_gemm_plan(1, Bb, Nb, Kb, 0, 0, CPLXAMD_BF16, 0, fam0 | CPLXAMD_LAUNCH_SHARED, 0),
                          cplxamd_gemm_plan(1, Bb, Nb, Kb, 0, 0, CPLXAMD_BF16, 0, fam0 | CPLXAMD_LAUNCH_EXCLUSIVE, 0)};
    printf("cplxamd_gemm_plan: SHARED -> kernel %d, EXCLUSIVE -> kernel %d (1 one tile per workgroup, 2 persistent)\n", kinds[0], kinds[1]);
    auto launch = [&](int flags, void** o, hipStream_t st) {
      return cplxamd_cgemm_fl(a_r, a_i, Kb, 1, b_r, b_i, Kb, 1, nullptr, nullptr, nullptr, o[0], o[1], Nb, Bb, Nb, Kb, 0,
                              CPLXAMD_BF16, CPLXAMD_BF16, 0, nullptr, CPLXAMD_ALGO_4M, nullptr, 0, flags, st);
    };
    if (launch(CPLXAMD_LAUNCH_DEFAULT, out[2], nullptr)) return 1;                  // serial reference, library defaults
    HIP_OK(hipDeviceSynchronize());
    hipStream_t st[2];
    HIP_OK(hipStreamCreate(&st[0])); HIP_OK(hipStreamCreate(&st[1]));
    int rcs[2] = {0, 0};
    std::thread t0([&] { for (int r = 0; r < 8 && !rcs[0]; ++r) rcs[0] = launch(fam0 | CPLXAMD_LAUNCH_SHARED, out[0], st[0]); });
    std::thread t1([&] { for (int r = 0; r < 8 && !rcs[1]; ++r) rcs[1] = launch(fam0 | CPLXAMD_LAUNCH_EXCLUSIVE, out[1], st[1]); });
    t0.join(); t1.join();
    HIP_OK(hipDeviceSynchronize());
    if (rcs[0] || rcs[1]) { fprintf(stderr, "cplxamd_cgemm_fl: error %d / %d\n", rcs[0], rcs[1]); return 1; }
    std::vector<uint16_t> h[3];
    for (int v = 0; v < 3; ++v) { h[v].resize((size_t)Bb * Nb); HIP_OK(hipMemcpy(h[v].data(), out[v][0], h[v].size() * 2, hipMemcpyDeviceToHost)); }
    const bool same = h[0] == h[2] && h[1] == h[2] && kinds[0] == 1 && kinds[1] == 2;
    printf("two threads / two streams, SHARED and EXCLUSIVE concurrently: %s\n", same ? "identical bits, different kernels" : "MISMATCH");
    if (cplxamd_cgemm_fl(a_r, a_i, Kb, 1, b_r, b_i, Kb, 1, nullptr, nullptr, nullptr, out[0][0], out[0][1], Nb, Bb, Nb, Kb, 0,
                         CPLXAMD_BF16, CPLXAMD_BF16, 0, nullptr, CPLXAMD_ALGO_4M, nullptr, 0,
                         CPLXAMD_LAUNCH_SHARED | CPLXAMD_LAUNCH_EXCLUSIVE, nullptr) != CPLXAMD_EINVAL) ok = false;   // contradictory flags
    ok = ok && same;
  }
  printf(ok ? "cabi_linear OK\n" : "cabi_linear FAILED\n");
  return ok ? 0 : 1;
}
