// devstate.h -- per-device host state of the library.  A process may drive several GPUs (one rank per GPU is the
// normal deployment, but nothing here may depend on it): the CU count and the dynamic-LDS attribute of a kernel
// (hipFuncSetAttribute) belong to the CURRENT device, so both are cached per device ordinal.
#pragma once
#include <hip/hip_runtime.h>

namespace gldm_dev {

constexpr int kMaxDevices = 64;

inline int current_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) dev = 0;
  return dev;
}

inline int cu_count() {
  static int cus[kMaxDevices] = {0};
  const int dev = current_device();
  if (!cus[dev]) {
    hipDeviceProp_t prop;
    int n = 0;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    cus[dev] = n > 0 ? n : 256;
  }
  return cus[dev];
}

// Raise a kernel's dynamic-LDS limit once per (kernel, device).  `Tag` makes one flag array per call site.
template <class Tag>
inline void allow_dynamic_lds(const void *kernel, int bytes) {
  static bool done[kMaxDevices] = {false};
  const int dev = current_device();
  if (!done[dev]) {
    (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    done[dev] = true;
  }
}

}  // namespace gldm_dev
