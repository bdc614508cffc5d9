// Probe: out-of-range behaviour of `buffer_load_dwordx4 ... offen lds` on gfx950 (what lands in LDS for lanes whose
// offset is past num_records, and whether the SGPR offset takes part in the range check / wraps).
//   hipcc --offload-arch=gfx950 -O2 scripts/r02/bufload_oob.hip -o /tmp/bufload_oob && /tmp/bufload_oob
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const char* x, uint32_t* out, unsigned bytes, unsigned soff, unsigned vbase) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  ((uint32_t*)smem)[threadIdx.x * 4 + 0] = 0xAAAAAAAAu;
  ((uint32_t*)smem)[threadIdx.x * 4 + 1] = 0xAAAAAAAAu;
  ((uint32_t*)smem)[threadIdx.x * 4 + 2] = 0xAAAAAAAAu;
  ((uint32_t*)smem)[threadIdx.x * 4 + 3] = 0xAAAAAAAAu;
  __syncthreads();
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)x);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((uintptr_t)x >> 32));
  i32x4 rs = {(int)lo, (int)(hi & 0xffff), (int)bytes, 0x00020000};
  unsigned voff = vbase + threadIdx.x * 16;
  unsigned ldsb = (unsigned)(uintptr_t)smem;
  unsigned so = __builtin_amdgcn_readfirstlane(soff);
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds" :: "v"(voff), "s"(rs), "s"(ldsb), "s"(so) : "memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  out[threadIdx.x] = ((uint32_t*)smem)[threadIdx.x * 4];
}
int main() {
  const unsigned N = 4096;            // bytes the descriptor covers; the allocation is larger on both sides
  char* base; uint32_t* out;
  hipMalloc(&base, 3 * N); hipMalloc(&out, 64 * 4);
  std::vector<uint32_t> h(3 * N / 4);
  for (size_t i = 0; i < h.size(); ++i) h[i] = 0x1000 + (uint32_t)i * 4;     // value = byte offset in the allocation + 0x1000
  hipMemcpy(base, h.data(), 3 * N, hipMemcpyHostToDevice);
  struct { const char* name; unsigned soff, vbase; } cases[] = {
      {"in range                  ", 0, 0},
      {"voff crosses the end      ", 0, N - 512},
      {"soff crosses the end      ", N - 512, 0},
      {"soff = -256 (wrapped)     ", 0xFFFFFF00u, 0},
      {"voff = -256 (wrapped)     ", 0, 0xFFFFFF00u},
      {"soff=-256, voff=+512      ", 0xFFFFFF00u, 512},
  };
  for (auto& c : cases) {
    k<<<1, 64, 1024>>>(base + N, out, N, c.soff, c.vbase);
    uint32_t r[64];
    hipMemcpy(r, out, sizeof(r), hipMemcpyDeviceToHost);
    printf("%s:", c.name);
    for (int l = 0; l < 64; l += 4) {
      const uint32_t v = r[l];
      if (v == 0xAAAAAAAAu) printf(" keep");
      else if (v == 0) printf(" zero");
      else printf(" %+d", (int)(v - 0x1000) - (int)N);     // byte offset relative to the descriptor base
    }
    printf("\n");
  }
  return 0;
}
