// v_mfma_f32_16x16x32_bf16 as the split-bf16 GEMM cores issue it: NACC accumulator tiles, six products per tile and trip
// (a_hi b_lo, a_lo b_hi, a_mid b_mid, a_hi b_mid, a_mid b_hi, a_hi b_hi), operands in registers, no loads.
//   ORDER 0: chain-major  (the six dependent MFMAs of a tile back to back, then the next tile)
//   ORDER 1: product-major (one product of every tile, then the next product: consecutive MFMAs are independent)
// 4 or 8 waves per CU (1 or 2 per SIMD); per-wave cycle counts tell how two waves share a SIMD's matrix pipe.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/mfma_bf_chain.hip -o tools/micro/build/mfma_bf_chain
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
__device__ __forceinline__ f32x4 mf(const u32x4 &a, const u32x4 &b, const f32x4 &c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <int NACC, int ORDER>
__global__ __launch_bounds__(512) void k(float *out, long long *cyc, int iters) {
  f32x4 acc[NACC];
  u32x4 a[3], b[NACC][3];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
  const unsigned s = 0x3f803f80u + threadIdx.x;
  for (int p = 0; p < 3; ++p) {
    a[p] = u32x4{s, s + p, s, s};
    for (int i = 0; i < NACC; ++i) b[i][p] = u32x4{s + i, s, s + p, s};
  }
  const int ia[6] = {0, 2, 1, 0, 1, 0}, ib[6] = {2, 0, 1, 1, 0, 0};
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (ORDER == 0) {
#pragma unroll
      for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int q = 0; q < 6; ++q) acc[i] = mf(a[ia[q]], b[i][ib[q]], acc[i]);
    } else {
#pragma unroll
      for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = mf(a[ia[q]], b[i][ib[q]], acc[i]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  const long long t1 = __builtin_readcyclecounter();
  float r = 0;
  for (int i = 0; i < NACC; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}
template <int NACC, int ORDER>
void run(int waves, int iters) {
  float *out; long long *cyc, h[8];
  (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 64);
  k<NACC, ORDER><<<256, waves * 64>>>(out, cyc, iters);
  k<NACC, ORDER><<<256, waves * 64>>>(out, cyc, iters);
  (void)hipDeviceSynchronize();
  (void)hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
  const double n = (double)iters * NACC * 6;
  printf("waves/CU=%d tiles=%2d %s: cycles per MFMA of the wave:", waves, NACC, ORDER ? "product-major" : "chain-major  ");
  for (int w = 0; w < waves; ++w) printf(" %.1f", h[w] / n);
  printf("\n");
  (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
  for (int w : {4, 8}) {
    run<1, 0>(w, 2000); run<4, 0>(w, 1000); run<4, 1>(w, 1000); run<10, 0>(w, 400); run<10, 1>(w, 400);
  }
  return 0;
}
