// How long does a short batch of loads take on a CU whose other workgroup streams conv weights?
// Even workgroups (blockIdx / 256 even) run the engine's 256x256 k=3 conv in a loop; odd ones time a batch of
// `nload` 16-byte loads (L2-hot, like the parameters of a short op).  Build as tools/micro/gemm_rate.hip.
#include "../../graspldm_amd/csrc/resnet1d.hip"
#include <vector>

namespace {
template <int MODE>  // 0: partner idle, 1: partner streams weights
__global__ __launch_bounds__(256, 2) void probe(const float *w, const float *small, int iters, int nload, long long *cyc,
                                                float *sink) {
  using GG = Geo<32>;
  extern __shared__ float lds[];
  Ctx c{w, lds, (int)threadIdx.x, __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), (int)threadIdx.x & 63, 0, GG::kNT};
  for (int i = c.tid; i < GG::kLdsFloats; i += GG::kThreads) lds[i] = 0.f;
  __syncthreads();
  const bool heavy = ((blockIdx.x / 256) & 1) == 0;
  if (heavy) {
    if (MODE == 1) {
      for (int i = 0; i < iters / 4; ++i) {
        Ctx cc = c;
        asm volatile("" : "+v"(cc.tid), "+v"(cc.lane));
        int ci = 256, co = 256, tp = 3;
        asm volatile("" : "+s"(ci), "+s"(co), "+s"(tp));
        conv_gemm<32, 4>(cc, 0, 1 << 18, lds + GG::kBufX, ci, tp, lds + GG::kBufH, co, false);
      }
    }
    return;
  }
  // probe workgroup: wave 0 only, like a short op
  if (c.wave != 0) return;
  long long total = 0;
  float acc = 0.f;
  const f32x4 *p = reinterpret_cast<const f32x4 *>(small) + c.lane;
  for (int it = 0; it < iters; ++it) {
    const long long t0 = __builtin_readcyclecounter();
    if (nload < 0) {  // scalar path: |nload| s_load_dwordx4 through the scalar cache instead of the vector-memory path
      typedef int i32x4 __attribute__((ext_vector_type(4)));
      for (int k = 0; k < -nload; ++k) {
        const float *q = small + (size_t)((k * 7 + it) & 255) * 256;
        i32x4 sv;
        asm volatile("s_load_dwordx4 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(sv) : "s"(q) : "memory");
        acc += __int_as_float(sv.x) + __int_as_float(sv.w);
      }
      asm volatile("" : "+v"(acc));
      total += __builtin_readcyclecounter() - t0;
      __builtin_amdgcn_s_sleep(64);
      __builtin_amdgcn_s_sleep(64);
      continue;
    }
    f32x4 v[48];
#pragma unroll
    for (int k = 0; k < 48; ++k)
      if (k < nload) v[k] = p[(size_t)((k * 7 + it) & 255) * 64];
#pragma unroll
    for (int k = 0; k < 48; ++k)
      if (k < nload) acc += v[k].x + v[k].w;
    asm volatile("" : "+v"(acc));
    total += __builtin_readcyclecounter() - t0;
    __builtin_amdgcn_s_sleep(64);  // ~4 k cycles between batches, like the compute between two short ops
    __builtin_amdgcn_s_sleep(64);
  }
  if (c.lane == 0) cyc[blockIdx.x] = total;
  sink[blockIdx.x * 64 + c.lane] = acc;
}
}  // namespace

int main() {
  float *w, *small, *sink; long long *dcyc;
  (void)hipMalloc(&w, (size_t)8 << 20); (void)hipMemset(w, 0, (size_t)8 << 20);
  (void)hipMalloc(&small, 1 << 20); (void)hipMemset(small, 0, 1 << 20);
  (void)hipMalloc(&sink, 512 * 64 * 4);
  (void)hipMalloc(&dcyc, 512 * sizeof(long long));
  const size_t lds = (size_t)Geo<32>::kLdsFloats * 4;
  (void)hipFuncSetAttribute((const void *)probe<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  (void)hipFuncSetAttribute((const void *)probe<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int iters = 400;
  for (int nload : {1, 8, 48, -1, -8}) {
    for (int mode = 0; mode < 2; ++mode) {
      (void)hipMemset(dcyc, 0, 512 * sizeof(long long));
      if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(512), dim3(256), lds, 0, w, small, iters, nload, dcyc, sink);
      else hipLaunchKernelGGL(probe<1>, dim3(512), dim3(256), lds, 0, w, small, iters, nload, dcyc, sink);
      (void)hipDeviceSynchronize();
      std::vector<long long> h(512);
      (void)hipMemcpy(h.data(), dcyc, 512 * sizeof(long long), hipMemcpyDeviceToHost);
      double avg = 0; int n = 0;
      for (int b = 0; b < 512; ++b) if (((b / 256) & 1) == 1) { avg += (double)h[b]; ++n; }
      printf("%3d loads of 16 B (negative: scalar loads, one after the other), partner %s: %7.0f cycles per batch\n", nload, mode ? "streaming a 256x256 k3 conv" : "idle", avg / n / iters);
    }
  }
  return 0;
}
