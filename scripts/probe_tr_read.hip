// Probe of ds_read_b64_tr_b16 (gfx950): which 16-bit element does lane i receive in slot j when
// lane m supplies the address of 4 consecutive elements E[m][0..3] (8 bytes)?
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/probe_tr scripts/probe_tr_read.hip && /tmp/probe_tr
// Result on MI355X (profiles/r01_tr_read_probe.txt): within each group of 16 lanes, lane i slot j
// receives E[4 j + (i >> 2)][i & 3] -- a 4 x 16 transpose.  gemm_bf16.hip / conv_nhwc_wgrad.hip rely
// on it: with lane m addressing T[kb + (m >> 2)][rb + 4 (m & 3)], lane i gets T[kb .. kb+3][rb + i].
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef short s16x4 __attribute__((ext_vector_type(4)));

__global__ void probe(uint16_t* out) {
  __shared__ uint16_t lds[64 * 4];
  const int lane = threadIdx.x;
  for (int c = 0; c < 4; ++c) lds[lane * 4 + c] = (uint16_t)(lane * 16 + c);   // E[m][c] = 16 m + c
  __syncthreads();
  const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4*)(lds + lane * 4));
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (uint16_t)v[j];
}

int main() {
  uint16_t* d;
  uint16_t h[256];
  hipMalloc(&d, sizeof(h));
  probe<<<1, 64>>>(d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int ok = 1;
  for (int i = 0; i < 64; ++i) {
    printf("lane %2d:", i);
    for (int j = 0; j < 4; ++j) {
      const int m = h[i * 4 + j] / 16, c = h[i * 4 + j] % 16;
      printf("  E[%2d][%d]", m, c);
      const int g = i & ~15, il = i & 15;
      ok &= (m == g + 4 * j + (il >> 2)) && (c == (il & 3));
    }
    printf("\n");
  }
  printf("%s\n", ok ? "MAPPING OK: lane i slot j <- E[16 (i / 16) + 4 j + ((i % 16) >> 2)][i & 3]"
                    : "MAPPING DIFFERS from the one the kernels assume");
  return ok ? 0 : 1;
}
