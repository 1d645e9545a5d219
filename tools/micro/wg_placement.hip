// Where do the workgroups of a 2-per-CU launch land?  Prints, per blockIdx, the (XCC, SE, CU) id the
// dispatcher chose, and which blockIdx pairs share a CU.  Build: hipcc --offload-arch=gfx950 -O2 wg_placement.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <map>
__global__ void probe(unsigned *smid, int spin_us) {
  extern __shared__ float lds[];
  if (threadIdx.x == 0) smid[blockIdx.x] = __smid();
  lds[threadIdx.x] = threadIdx.x;
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin_us * 100) __builtin_amdgcn_s_sleep(32);
  if (lds[threadIdx.x] < 0) smid[0] = 0;
}
int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 512, threads = argc > 2 ? atoi(argv[2]) : 256;
  const size_t lds = argc > 3 ? atoi(argv[3]) : 76 * 1024;
  unsigned *d;
  (void)hipMalloc(&d, n * sizeof(unsigned));
  (void)hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(probe, dim3(n), dim3(threads), lds, 0, d, 200);
  std::vector<unsigned> h(n);
  (void)hipMemcpy(h.data(), d, n * sizeof(unsigned), hipMemcpyDeviceToHost);
  std::map<unsigned, std::vector<int>> by;
  for (int i = 0; i < n; ++i) by[h[i]].push_back(i);
  printf("%d workgroups on %zu distinct CUs\n", n, by.size());
  for (int i = 0; i < 24 && i < n; ++i) printf("block %3d -> xcc %u se %u cu %u\n", i, h[i] >> 6, (h[i] >> 4) & 3, h[i] & 15);
  std::map<int, int> delta;
  for (auto &kv : by) if (kv.second.size() == 2) delta[kv.second[1] - kv.second[0]]++;
  for (auto &kv : delta) printf("pairs with blockIdx distance %d: %d\n", kv.first, kv.second);
  std::map<size_t, int> occ;
  for (auto &kv : by) occ[kv.second.size()]++;
  for (auto &kv : occ) printf("CUs holding %zu workgroups: %d\n", kv.first, kv.second);
  return 0;
}
