// devstate.h -- per-device host state of the library.  A process may drive several GPUs (one rank per GPU is the
// normal deployment, but nothing here may depend on it): the CU count and the dynamic-LDS attribute of a kernel
// (hipFuncSetAttribute) belong to the CURRENT device, so both are cached per device ordinal.  Host threads may launch
// concurrently: the caches are atomics, and a racing first use only repeats an idempotent query / attribute call.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>

namespace gldm_dev {

constexpr int kMaxDevices = 64;

// Ordinal of the current device, or -1 when the runtime cannot say or the ordinal is outside the cache (callers then
// take the uncached path: query / set the attribute every time, which is correct, only slower).
inline int current_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return -1;
  return dev;
}

inline int cu_count() {
  static std::atomic<int> cus[kMaxDevices];
  const int dev = current_device();
  if (dev >= 0) {
    const int c = cus[dev].load(std::memory_order_relaxed);
    if (c) return c;
  }
  int n = 0, d = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&d) == hipSuccess && hipGetDeviceProperties(&prop, d) == hipSuccess) n = prop.multiProcessorCount;
  n = n > 0 ? n : 256;
  if (dev >= 0) cus[dev].store(n, std::memory_order_relaxed);
  return n;
}

// Raise a kernel's dynamic-LDS limit to at least `bytes` on the current device.  `Tag` makes one array per call site
// (= per kernel); the array holds the largest size already set, so a later call that asks for more re-applies the
// attribute, and two threads racing on the first launch both set it (the launch of either then finds it set).
template <class Tag>
inline void allow_dynamic_lds(const void *kernel, int bytes) {
  static std::atomic<int> set_bytes[kMaxDevices];
  const int dev = current_device();
  if (dev >= 0 && set_bytes[dev].load(std::memory_order_acquire) >= bytes) return;
  (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (dev >= 0) {
    int cur = set_bytes[dev].load(std::memory_order_relaxed);
    while (cur < bytes && !set_bytes[dev].compare_exchange_weak(cur, bytes, std::memory_order_release)) {
    }
  }
}

}  // namespace gldm_dev
