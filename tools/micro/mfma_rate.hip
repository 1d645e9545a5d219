// Microbenchmark: v_mfma_f32_16x16x4_f32 issue rate with W waves per workgroup (one WG per CU)
// and A independent accumulators per wave.  hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x4 = __attribute__((ext_vector_type(4))) float;
template <int A>
__global__ void k(float *out, long long *cyc, int iters) {
  f32x4 acc[A];
  for (int i = 0; i < A; ++i) acc[i] = f32x4{0, 0, 0, 0};
  float a = threadIdx.x * 0.001f, b = threadIdx.x * 0.002f;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < A; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < A; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int A>
void run(int waves, int iters) {
  float *out; long long *cyc, h;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<A><<<256, waves * 64>>>(out, cyc, iters);
  hipEventRecord(e0); k<A><<<256, waves * 64>>>(out, cyc, iters); hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  double mf = (double)iters * A;   // per wave
  double wps = waves / 4.0;        // waves per SIMD
  printf("waves/CU=%d acc=%d: %.1f cycles per MFMA per SIMD (wave0: %.1f cyc/MFMA), %.1f TFLOP/s\n", waves, A,
         h / (mf * (wps < 1 ? 1 : wps)), h / mf, 256.0 * waves * mf * 2048 / (ms * 1e-3) / 1e12);
  hipFree(out); hipFree(cyc);
}
int main() {
  for (int w : {4, 8, 16}) { run<1>(w, 4000); run<2>(w, 4000); run<4>(w, 4000); run<8>(w, 2000); }
  return 0;
}
