// Per-shape rate of the engine's conv_gemm (the real code, included below) in isolation:
// cycles per call vs the MFMA-bound ideal, with 1 or 2 workgroups per CU.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tools/micro/gemm_rate.hip -o tools/micro/build/gemm_rate
#include "../../graspldm_amd/csrc/resnet1d.hip"
#include <vector>

namespace {
template <int NC, int L>
__global__ __launch_bounds__(Geo<NC>::kThreads, 2) void gemm_probe(const float *w, int cin, int cout, int taps, int iters,
                                                                  long long *cycles) {
  using GG = Geo<NC>;
  extern __shared__ float lds[];
#ifdef GLDM_PROBE_COPIES
  w += ((blockIdx.x / 256) & 1) * (1 << 19);
#endif
  Ctx c{w, lds, (int)threadIdx.x, (int)threadIdx.x >> 6, (int)threadIdx.x & 63, 0, GG::kNT};
  for (int i = c.tid; i < GG::kLdsFloats; i += GG::kThreads) lds[i] = 0.f;
  __syncthreads();
#ifdef GLDM_PROBE_SKEW
  {
    const long long w0 = wall_clock64();
    const int ticks = ((blockIdx.x / 8) % 32) * GLDM_PROBE_SKEW;  // 10 ns units
    while (wall_clock64() - w0 < ticks) __builtin_amdgcn_s_sleep(8);
  }
#endif
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    Ctx cc = c;
    asm volatile("" : "+v"(cc.tid), "+v"(cc.lane));  // as run_tape does: no hoisting of lane-derived offsets
    int ci = cin, co = cout, tp = taps;
    asm volatile("" : "+s"(ci), "+s"(co), "+s"(tp));  // per-op values in the real kernel (read from the tape)
#ifdef GLDM_PROBE_COLD
    cc.w = w + (size_t)(i % 48) * (1 << 19);  // 48 x 2 MiB apart: every call streams weights that left L2 (96 MiB cycle)
#endif
    conv_gemm<NC, L>(cc, 0, 1 << 18, lds + GG::kBufX, ci, tp, lds + GG::kBufH, co, false);
  }
  const long long t1 = __builtin_readcyclecounter();
  if (c.tid == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int NC>
void run(const float *w, long long *dcyc, int wgs, int cin, int cout, int taps) {
  const int iters = 200;
  const size_t lds = (size_t)Geo<NC>::kLdsFloats * 4;
  (void)hipFuncSetAttribute((const void *)gemm_probe<NC, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((gemm_probe<NC, 4>), dim3(wgs), dim3(Geo<NC>::kThreads), lds, 0, w, cin, cout, taps, iters, dcyc);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL((gemm_probe<NC, 4>), dim3(wgs), dim3(Geo<NC>::kThreads), lds, 0, w, cin, cout, taps, iters, dcyc);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(wgs);
  (void)hipMemcpy(h.data(), dcyc, wgs * sizeof(long long), hipMemcpyDeviceToHost);
  double avg = 0;
  for (auto v : h) avg += (double)v;
  avg /= wgs * (double)iters;
  // MFMAs of the tile, spread over the 4 SIMDs at 32 cycles each; a CU holding 2 workgroups shares them
  const double mfma = ((cout + 15) / 16) * (double)((taps * cin + 3) / 4) * (NC / 16);
  const int per_cu = (wgs + 255) / 256;
  const double ideal = mfma * 32.0 / 4.0 * per_cu;
  const double us = ms * 1e3 / iters;
  printf("NC=%d wgs=%4d cin=%3d cout=%3d taps=%d : %7.0f clk %7.3f us | ideal %6.0f clk %6.3f us | eff %5.1f%%\n",
         NC, wgs, cin, cout, taps, avg, us, ideal, ideal / 2400.0, 100.0 * (ideal / 2400.0) / us);
}
}  // namespace

int main() {
  float *w; long long *dcyc;
  (void)hipMalloc(&w, (size_t)50 * (1 << 21));
  (void)hipMemset(w, 0, (size_t)50 * (1 << 21));
  (void)hipMalloc(&dcyc, 1024 * sizeof(long long));
#ifdef GLDM_PROBE_SHORT
  const int shapes[][3] = {{256, 256, 3}, {128, 128, 3}, {128, 192, 1}, {64, 64, 3}};
#else
  const int shapes[][3] = {{256, 256, 3}, {128, 256, 3}, {128, 128, 3}, {128, 192, 1}, {128, 128, 1}, {64, 128, 3},
                           {64, 64, 3},   {64, 192, 1},  {128, 64, 1},  {32, 64, 3},   {32, 32, 3},   {32, 192, 1},
                           {128, 32, 1},  {4, 32, 3},    {4, 4, 3},     {4, 192, 1},   {128, 4, 1}};
#endif
  for (auto &s : shapes) {
    run<32>(w, dcyc, 256, s[0], s[1], s[2]);
    run<32>(w, dcyc, 512, s[0], s[1], s[2]);
    if (s[2] == 1) run<64>(w, dcyc, 256, s[0], s[1], s[2]);
  }
  return 0;
}
