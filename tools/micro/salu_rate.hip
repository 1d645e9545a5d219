// How many scalar instructions per clock does a CU issue?  W waves per CU (W / 4 per SIMD) each run a chain-free
// stream of s_add_u32 / s_xor_b32 on private SGPRs; the time per instruction tells whether the scalar unit is per SIMD or
// shared.  Also: v_readfirstlane + dependent s_add (the tape-decode pattern) and plain VALU for comparison.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/salu_rate.hip -o tools/micro/build/salu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(1024) void k(int *out, long long *cyc, int iters, int seed) {
  int a = seed, b = seed + 1, c = seed + 2, d = seed + 3, e = seed + 4, f = seed + 5, g = seed + 6, h = seed + 7;
  int v = threadIdx.x;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {   // 32 independent-ish scalar ops per trip (8 chains of 4)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        asm volatile("s_add_u32 %0, %0, 1\n s_xor_b32 %1, %1, 3\n s_add_u32 %2, %2, 5\n s_xor_b32 %3, %3, 7\n"
                     "s_add_u32 %4, %4, 9\n s_xor_b32 %5, %5, 11\n s_add_u32 %6, %6, 13\n s_xor_b32 %7, %7, 15\n"
                     : "+s"(a), "+s"(b), "+s"(c), "+s"(d), "+s"(e), "+s"(f), "+s"(g), "+s"(h) : : "scc");
    } else if (MODE == 1) {   // decode pattern: v_readfirstlane -> dependent scalar add, 8 per trip
      asm volatile("v_readfirstlane_b32 %0, %8\n s_nop 0\n s_add_u32 %0, %0, 1\n v_readfirstlane_b32 %1, %8\n s_nop 0\n s_add_u32 %1, %1, 1\n"
                   "v_readfirstlane_b32 %2, %8\n s_nop 0\n s_add_u32 %2, %2, 1\n v_readfirstlane_b32 %3, %8\n s_nop 0\n s_add_u32 %3, %3, 1\n"
                   "v_readfirstlane_b32 %4, %8\n s_nop 0\n s_add_u32 %4, %4, 1\n v_readfirstlane_b32 %5, %8\n s_nop 0\n s_add_u32 %5, %5, 1\n"
                   "v_readfirstlane_b32 %6, %8\n s_nop 0\n s_add_u32 %6, %6, 1\n v_readfirstlane_b32 %7, %8\n s_nop 0\n s_add_u32 %7, %7, 1\n"
                   : "=s"(a), "=s"(b), "=s"(c), "=s"(d), "=s"(e), "=s"(f), "=s"(g), "=s"(h) : "v"(v) : "scc");
    } else {   // 32 VALU ops per trip
#pragma unroll
      for (int r = 0; r < 32; ++r) asm volatile("v_add_u32 %0, %0, 1" : "+v"(v));
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[threadIdx.x >> 6] = t1 - t0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ e ^ f ^ g ^ h ^ v;
}
template <int MODE>
void run(int waves, const char *what, int per_trip) {
  int *out; long long *cyc, h[16];
  (void)hipMalloc(&out, 256 * 1024 * 4); (void)hipMalloc(&cyc, 128);
  const int iters = 4000;
  k<MODE><<<256, waves * 64>>>(out, cyc, iters, 1);
  k<MODE><<<256, waves * 64>>>(out, cyc, iters, 1);
  (void)hipDeviceSynchronize();
  (void)hipMemcpy(h, cyc, 128, hipMemcpyDeviceToHost);
  double mx = 0; for (int w = 0; w < waves; ++w) mx = h[w] > mx ? (double)h[w] : mx;
  printf("%-28s waves/CU=%2d: %.2f cycles per instruction and wave, %.2f instructions per clock and CU\n", what, waves,
         mx / (iters * (double)per_trip), waves * iters * (double)per_trip / mx);
  (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
  for (int w : {1, 4, 8, 16}) run<0>(w, "s_add / s_xor stream", 32);
  for (int w : {1, 4, 8, 16}) run<1>(w, "readfirstlane + s_add", 24);
  for (int w : {1, 4, 8, 16}) run<2>(w, "v_add_u32 stream", 32);
  return 0;
}
