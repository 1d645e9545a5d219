// wstream.h -- packed weight fragments read through a buffer descriptor (shared by the GEMM cores).
#ifndef GLDM_WSTREAM_H_
#define GLDM_WSTREAM_H_
#include <hip/hip_runtime.h>

namespace {
using wstream_f32x4 = __attribute__((ext_vector_type(4))) float;
using wstream_u32x4 = __attribute__((ext_vector_type(4))) unsigned;

// Stream of packed weight fragments: fragment f = 64 lanes x 16 bytes at byte offset 1024 f of `wp`;
// the accessors take i = 64 f (the index of the fragment's first 16-byte element).
// Read with buffer_load_dwordx4 through a wave-uniform descriptor: base in SGPRs, the fragment's byte
// offset as the SCALAR offset, the lane's 16 bytes as a constant 32-bit vector offset.  A global_load of
// the same bytes carries a 64-bit address per lane; issuing it between MFMAs costs the SIMD ~60 cycles of
// matrix issue per instruction (measured: 256x256 k3 conv 84 % -> 95 % of the MFMA-bound time,
// tools/micro/gemm_pm_rate).  `i` must be wave uniform.
struct WStream {
  __amdgpu_buffer_rsrc_t r;
  int v;
  __device__ __forceinline__ WStream(const float *wp, int lane) {
    const unsigned long long a = (unsigned long long)wp;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    r = __builtin_amdgcn_make_buffer_rsrc((void *)(((unsigned long long)hi << 32) | lo), 0, 0x7FFFFFFF, 0x00020000);
    v = lane * 16;
  }
  // the same fragment as raw bits (split-f16 planes: 8 bf16 per lane)
  __device__ __forceinline__ wstream_u32x4 raw(size_t i) const {
    return __builtin_bit_cast(wstream_u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, v, (int)(i * 16), 0));
  }
  // fragment at byte offset `sbytes` (wave uniform, in an SGPR) + `imm` (compile-time constant < 4096: the instruction's
  // offset field): three planes of one (m-tile, block) are one scalar offset and immediates 0 / 1024 / 2048, instead of a
  // scalar offset computed (and, in the big kernels, spilled and reloaded) per load
  __device__ __forceinline__ wstream_u32x4 raw_at(int sbytes, int imm) const {
    return __builtin_bit_cast(wstream_u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, v + imm, sbytes, 0));
  }
  __device__ __forceinline__ wstream_f32x4 operator[](size_t i) const {
    const auto q = __builtin_amdgcn_raw_buffer_load_b128(r, v, (int)(i * 16), 0);
    return wstream_f32x4{__uint_as_float(q[0]), __uint_as_float(q[1]), __uint_as_float(q[2]), __uint_as_float(q[3])};
  }
};
}  // namespace
#endif
