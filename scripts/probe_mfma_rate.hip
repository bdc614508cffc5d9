// Sustained MFMA rate under the chip's power management: v_mfma_f32_32x32x16_bf16 vs
// v_mfma_f32_16x16x32_bf16 on N(0,1) register data, no memory traffic in the loop.
//   hipcc --offload-arch=gfx950 -O3 scripts/probe_mfma_rate.hip -o /tmp/probe_mfma && /tmp/probe_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(256) void mfma_loop(const bf16x8* src, float* out, int iters) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  bf16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = src[(t * 8 + i) & 0xffff]; b[i] = src[(t * 8 + 4 + i) & 0xffff]; }
  float s = 0.f;
  if (KIND == 0) {
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x16{0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i & 3], b[(i >> 1) & 3], acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][7];
  } else {
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
  }
  out[t] = s;
}

int main() {
  const int n = 1 << 16;
  std::vector<unsigned short> h(n * 8);
  srand(1);
  for (auto& v : h) {
    float u1 = (rand() + 1.f) / (RAND_MAX + 2.f), u2 = (rand() + 1.f) / (RAND_MAX + 2.f);
    float z = sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
    unsigned int bits; memcpy(&bits, &z, 4);
    v = (unsigned short)(bits >> 16);
  }
  bf16x8* src; float* out;
  hipMalloc(&src, n * 16); hipMalloc(&out, 256 * 4096 * 4);
  hipMemcpy(src, h.data(), n * 16, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wps = 1; wps <= 2; ++wps) {                 // waves per SIMD
    const int blocks = 256 * wps;                      // 4 waves per block, one block per CU and wave slot
    for (int kind = 0; kind < 2; ++kind) {
      const int iters = 20000;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        if (kind == 0) mfma_loop<0><<<blocks, 256>>>(src, out, iters); else mfma_loop<1><<<blocks, 256>>>(src, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flop = (double)blocks * 4 * iters * (kind == 0 ? 8 * 32768.0 : 16 * 16384.0);
        printf("%s waves/SIMD=%d: %.3f ms  %.0f TF/s\n", kind == 0 ? "32x32x16" : "16x16x32", wps, ms, flop / ms / 1e9);
      }
    }
  }
  return 0;
}
