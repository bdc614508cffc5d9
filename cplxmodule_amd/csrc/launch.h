// Host-side launch policy shared by the GEMM and channels-last convolution launchers (SURVEY 8(b) "Threading": the entry
// points are re-entrant -- what a launch does depends on its ARGUMENTS only).
//
//  * The launch form is chosen PER CALL by the `flags` argument of the `*_fl` entry points (include/cplxamd.h,
//    CPLXAMD_LAUNCH_*).  The two process-wide setters of ABI <= 18 (cplxamd_gemm_set_persistent / _set_family) survive as
//    DEPRECATED defaults: atomics that only a call whose flags leave the choice open (flags == 0, or the flag-less entry
//    points) reads.  Nothing in the Python host writes them any more (cplxmodule_amd/dp.py passes flags).
//  * One-time per-kernel set-up (hipFuncSetAttribute of the dynamic LDS size) and the CU count are cached PER DEVICE: a
//    process that drives several GPUs pays them once on each (`static bool attr_set` was per process).
#pragma once
#include <atomic>

#include "common.h"

namespace cplxamd {

extern std::atomic<int> g_default_persistent;   // gemm.hip; 1 at start
extern std::atomic<int> g_default_family;       // gemm.hip; 0xbf at start (env CPLXAMD_GEMM_W4)

// may this launch assume the whole chip for its whole duration (persistent forms: one workgroup per CU walking a static
// tile list)?  CPLXAMD_LAUNCH_SHARED: no -- other kernels (an RCCL all-reduce) hold CUs; one workgroup per tile.
inline bool launch_owns_chip(int flags) {
  if (flags & CPLXAMD_LAUNCH_SHARED) return false;
  if (flags & CPLXAMD_LAUNCH_EXCLUSIVE) return true;
  return g_default_persistent.load(std::memory_order_relaxed) != 0;
}
// bf16 GEMM kernel family mask of this launch (cplxamd_gemm_set_family's bit layout)
inline int launch_family(int flags) {
  if (flags & CPLXAMD_LAUNCH_FAMILY_SET) return (flags >> CPLXAMD_LAUNCH_FAMILY_SHIFT) & 0xff;
  return g_default_family.load(std::memory_order_relaxed);
}
inline bool launch_flags_ok(int flags) {
  const int known = CPLXAMD_LAUNCH_SHARED | CPLXAMD_LAUNCH_EXCLUSIVE | CPLXAMD_LAUNCH_FAMILY_SET |
                    (0xff << CPLXAMD_LAUNCH_FAMILY_SHIFT);
  return (flags & ~known) == 0 && (flags & (CPLXAMD_LAUNCH_SHARED | CPLXAMD_LAUNCH_EXCLUSIVE)) !=
                                      (CPLXAMD_LAUNCH_SHARED | CPLXAMD_LAUNCH_EXCLUSIVE);
}

inline int current_device() {
  int d = 0;
  return hipGetDevice(&d) == hipSuccess ? (d & 63) : 0;
}

// "done once per device" bit set (two threads racing on the first launch both do the -- idempotent -- set-up)
struct PerDeviceOnce {
  std::atomic<uint64_t> mask{0};
};
template <typename K>
inline int set_max_dyn_lds(PerDeviceOnce& once, K kernel, int bytes) {
  const uint64_t bit = 1ull << current_device();
  if (once.mask.load(std::memory_order_acquire) & bit) return 0;
  const hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return (int)e;
  once.mask.fetch_or(bit, std::memory_order_release);
  return 0;
}

// compute units of the current device (>= 8), cached per device
inline int device_cus() {
  static std::atomic<int> cus[64];
  const int dev = current_device();
  int n = cus[dev].load(std::memory_order_relaxed);
  if (n == 0) {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess ||
        n < 8)
      n = 8;
    cus[dev].store(n, std::memory_order_relaxed);
  }
  return n;
}

}  // namespace cplxamd
