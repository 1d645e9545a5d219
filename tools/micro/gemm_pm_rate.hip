// Position-major k = 3 conv core (gemm_pm3: 64-column tiles, 8 waves, one workgroup per CU) against the
// sample-major core (conv_gemm: 32-column tiles, 4 waves, two workgroups per CU): time per call for the same
// 64 columns per CU.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tools/micro/gemm_pm_rate.hip -o tools/micro/build/gemm_pm_rate
#include "../../graspldm_amd/csrc/resnet1d.hip"
#include <vector>

namespace {
// plain store of the accumulators (what conv_gemm does without the GroupNorm epilogue)
template <int MT, int P0, int NP>
__device__ __forceinline__ void pm3_call(const Ctx &c, const float *wp, int cin, int mt0, const float *src, float *dst) {
  f32x4 acc[MT][NP];
#pragma unroll
  for (int mi = 0; mi < MT; ++mi)
#pragma unroll
    for (int p = 0; p < NP; ++p) acc[mi][p] = f32x4{0.f, 0.f, 0.f, 0.f};
  gemm_pm3<MT, P0, NP>(c, wp, cin >> 4, mt0, src, acc);
  const int col = c.lane & 15, kq = c.lane >> 4;
  lds_f *d3 = (lds_f *)dst;
#pragma unroll
  for (int mi = 0; mi < MT; ++mi)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int p = 0; p < NP; ++p) d3[swz<64>(16 * (mt0 + mi) + 4 * kq + r, 16 * (P0 + p) + col)] = acc[mi][p][r];
}

__global__ __launch_bounds__(512, 2) void pm_probe(const float *w, int cin, int cout, int iters, long long *cycles) {
  using GG = Geo<64>;
  extern __shared__ float lds[];
  Ctx c{w, lds, (int)threadIdx.x, __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), (int)threadIdx.x & 63, 0, GG::kNT};
  for (int i = c.tid; i < GG::kLdsFloats; i += GG::kThreads) lds[i] = 0.37f * (float)((i * 2654435761u >> 20) & 1023) / 1024.f - 0.18f;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    Ctx cc = c;
    asm volatile("" : "+v"(cc.tid), "+v"(cc.lane));
    int ci = cin, co = cout;
    asm volatile("" : "+s"(ci), "+s"(co));
    const int mtiles = co >> 4, wv = cc.wave;
    const float *src = lds + GG::kBufX;
    float *dst = lds + GG::kBufH;
    if (mtiles == 16) pm3_call<2, 0, 4>(cc, w, ci, 2 * wv, src, dst);
    else if (mtiles == 8) pm3_call<1, 0, 4>(cc, w, ci, wv, src, dst);
    else if (mtiles == 4) { if (wv < 4) pm3_call<1, 0, 2>(cc, w, ci, wv & 3, src, dst); else pm3_call<1, 2, 2>(cc, w, ci, wv & 3, src, dst); }
    else if (mtiles == 2) {
      const int p = wv >> 1;
      if (p == 0) pm3_call<1, 0, 1>(cc, w, ci, wv & 1, src, dst);
      else if (p == 1) pm3_call<1, 1, 1>(cc, w, ci, wv & 1, src, dst);
      else if (p == 2) pm3_call<1, 2, 1>(cc, w, ci, wv & 1, src, dst);
      else pm3_call<1, 3, 1>(cc, w, ci, wv & 1, src, dst);
    }
    __syncthreads();
  }
  const long long t1 = __builtin_readcyclecounter();
  if (c.tid == 0) cycles[blockIdx.x] = t1 - t0;
}

__global__ __launch_bounds__(512, 2) void pm_1x1_probe(const float *w, int cin, int cout, int iters, long long *cycles) {
  using GG = Geo<64>;
  extern __shared__ float lds[];
  Ctx c{w, lds, (int)threadIdx.x, __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), (int)threadIdx.x & 63, 0, GG::kNT};
  for (int i = c.tid; i < GG::kLdsFloats; i += GG::kThreads) lds[i] = 0.37f * (float)((i * 2654435761u >> 20) & 1023) / 1024.f - 0.18f;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    Ctx cc = c;
    asm volatile("" : "+v"(cc.tid), "+v"(cc.lane));
    int ci = cin, co = cout;
    asm volatile("" : "+s"(ci), "+s"(co));
    conv_gemm<64, 4>(cc, 0, -1, lds + GG::kBufX, ci, 1, lds + GG::kBufH, co, false);
  }
  const long long t1 = __builtin_readcyclecounter();
  if (c.tid == 0) cycles[blockIdx.x] = t1 - t0;
}

__global__ __launch_bounds__(512, 2) void pm_gn_probe(const float *w, int cin, int cout, int mode, int iters, long long *cycles) {
  using GG = Geo<64>;
  extern __shared__ float lds[];
  Ctx c{w, lds, (int)threadIdx.x, __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), (int)threadIdx.x & 63, 0, GG::kNT};
  for (int i = c.tid; i < GG::kLdsFloats; i += GG::kThreads) lds[i] = 0.37f * (float)((i * 2654435761u >> 20) & 1023) / 1024.f - 0.18f;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    Ctx cc = c;
    asm volatile("" : "+v"(cc.tid), "+v"(cc.lane));
    int ci = cin, co = cout, md = mode;
    asm volatile("" : "+s"(ci), "+s"(co), "+s"(md));
    const GnEpilogue g{md, 1 << 19, (1 << 19) + 1024, md == 1 ? (1 << 19) + 4096 : -1, (1 << 19) + 2048, 16, co, co / 4, lds + GG::kBufH};
    conv_gemm<64, 4>(cc, 0, 1 << 18, lds + GG::kBufX, ci, 3, lds + GG::kBufH, co, false, 0, g);
  }
  const long long t1 = __builtin_readcyclecounter();
  if (c.tid == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int NC, int L>
__global__ __launch_bounds__(Geo<NC>::kThreads, 2) void sm_probe(const float *w, int cin, int cout, int iters, long long *cycles) {
  using GG = Geo<NC>;
  extern __shared__ float lds[];
  Ctx c{w, lds, (int)threadIdx.x, (int)threadIdx.x >> 6, (int)threadIdx.x & 63, 0, GG::kNT};
  for (int i = c.tid; i < GG::kLdsFloats; i += GG::kThreads) lds[i] = 0.37f * (float)((i * 2654435761u >> 20) & 1023) / 1024.f - 0.18f;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    Ctx cc = c;
    asm volatile("" : "+v"(cc.tid), "+v"(cc.lane));
    int ci = cin, co = cout, tp = 3;
    asm volatile("" : "+s"(ci), "+s"(co), "+s"(tp));
    conv_gemm<NC, L>(cc, 0, 1 << 18, lds + GG::kBufX, ci, tp, lds + GG::kBufH, co, false);
  }
  const long long t1 = __builtin_readcyclecounter();
  if (c.tid == 0) cycles[blockIdx.x] = t1 - t0;
}

double avg_cycles(long long *dcyc, int wgs, int iters) {
  std::vector<long long> h(wgs);
  (void)hipMemcpy(h.data(), dcyc, wgs * sizeof(long long), hipMemcpyDeviceToHost);
  double a = 0;
  for (auto v : h) a += (double)v;
  return a / wgs / iters;
}

float time_launch(void (*launch)(), int reps = 3) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  launch();
  (void)hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) launch();
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}
}  // namespace

int main() {
  float *w; long long *dcyc;
  (void)hipMalloc(&w, (size_t)8 << 20);
  {
    std::vector<float> hw((size_t)2 << 20);
    unsigned x = 12345u;
    for (auto &v : hw) { x = x * 1664525u + 1013904223u; v = ((float)(x >> 8) / 16777216.f - 0.5f) * 0.1f; }
    (void)hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
  }
  (void)hipMalloc(&dcyc, 1024 * sizeof(long long));
  const int iters = 200;
  const size_t lds64 = (size_t)Geo<64>::kLdsFloats * 4, lds32 = (size_t)Geo<32>::kLdsFloats * 4;
  (void)hipFuncSetAttribute((const void *)pm_probe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds64);
  (void)hipFuncSetAttribute((const void *)pm_gn_probe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds64);
  (void)hipFuncSetAttribute((const void *)sm_probe<32, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds32);
  const int shapes[][2] = {{256, 256}, {128, 256}, {128, 128}, {64, 128}, {64, 64}, {32, 64}, {32, 32}};
  (void)hipFuncSetAttribute((const void *)pm_1x1_probe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds64);
  {
    const int s11[][2] = {{128, 192}, {64, 192}, {32, 192}, {128, 128}, {128, 64}, {128, 32}};
    for (auto &sh : s11) {
      static int q_cin, q_cout; static const float *q_w; static long long *q_c;
      q_cin = sh[0]; q_cout = sh[1]; q_w = w; q_c = dcyc;
      (void)time_launch([] { hipLaunchKernelGGL(pm_1x1_probe, dim3(256), dim3(512), (size_t)Geo<64>::kLdsFloats * 4, 0, q_w, q_cin, q_cout, 200, q_c); }, 1);
      const double cyc = avg_cycles(dcyc, 256, iters);
      const double ideal = (sh[1] / 16) * (sh[0] / 4.0) * 4 * 32.0 / 4;
      printf("1x1 cin=%3d cout=%3d on 64 columns: %6.0f cycles per call (MFMA-bound %5.0f = %4.1f%%)\n", sh[0], sh[1], cyc, ideal, 100 * ideal / cyc);
    }
  }
  for (auto &sh : shapes) {
    const int cin = sh[0], cout = sh[1];
    static int s_cin, s_cout; static const float *s_w; static long long *s_c;
    s_cin = cin; s_cout = cout; s_w = w; s_c = dcyc;
    const float ms_sm = time_launch([] { hipLaunchKernelGGL((sm_probe<32, 4>), dim3(512), dim3(256), (size_t)Geo<32>::kLdsFloats * 4, 0, s_w, s_cin, s_cout, 200, s_c); });
    const double cyc_sm = avg_cycles(dcyc, 512, iters);
    const float ms_pm = time_launch([] { hipLaunchKernelGGL(pm_probe, dim3(256), dim3(512), (size_t)Geo<64>::kLdsFloats * 4, 0, s_w, s_cin, s_cout, 200, s_c); });
    const double cyc_pm = avg_cycles(dcyc, 256, iters);
    double cyc_gn[3];
    for (int md = 0; md < 3; ++md) {
      static int s_md; s_md = md;
      (void)time_launch([] { hipLaunchKernelGGL(pm_gn_probe, dim3(256), dim3(512), (size_t)Geo<64>::kLdsFloats * 4, 0, s_w, s_cin, s_cout, s_md, 200, s_c); }, 1);
      cyc_gn[md] = avg_cycles(dcyc, 256, iters);
    }
    const double ideal10 = (cout / 16) * (3.0 * cin / 4) * 4 * 32.0 / 4 * 10 / 12;
    // MFMA-bound ideal for 64 columns per CU at the sample-major count (12 tile-MFMAs) and at the position-major count (10)
    const double mf12 = (cout / 16) * (3.0 * cin / 4) * 4 * 32.0 / 4 / 2400.0;  // us
    const double us_sm = ms_sm * 1e3 / iters, us_pm = ms_pm * 1e3 / iters;
    printf("cin=%3d cout=%3d k3 | sample-major 2x32 cols: %7.3f us (%5.1f%% of MFMA peak) | position-major 1x64 cols: %7.3f us (%5.1f%% of peak on the algorithmic FLOP) | speedup %.3f | cycles/call sm %.0f pm %.0f (pm ideal at 10 tile-MFMAs %.0f = %.1f%%) | conv_gemm<64,4> plain %.0f GN %.0f GN+res %.0f\n",
           cin, cout, us_sm, 100 * mf12 / us_sm, us_pm, 100 * mf12 / us_pm, us_sm / us_pm, cyc_sm, cyc_pm, ideal10, 100 * ideal10 / cyc_pm, cyc_gn[0], cyc_gn[1], cyc_gn[2]);
  }
  return 0;
}
