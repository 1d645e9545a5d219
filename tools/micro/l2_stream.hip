// How many bytes per clock does ONE CU draw from L2 / Infinity Cache when all it does is read fragment-ordered weights?
// Every workgroup reads the same `bytes`-sized set over and over (like the engines' weight stream: all CUs, same data),
// 1 KiB per wave-instruction, `DEPTH` instructions in flight per wave.  Forms:
//   0  buffer_load_dwordx4, wave-uniform descriptor + scalar offset (csrc/wstream.h)
//   1  global_load_dwordx4 (64-bit address per lane)
//   2  buffer_load_dwordx4 ... lds   (gfx950: 16 B per lane straight into LDS, no VGPR)
//   3  form 0 with the nt bit (streaming hint)
//   4  buffer_load_dwordx2 (512 B per instruction)
// Prints bytes / shader clock / CU (s_memtime) and the shader clock itself (against the 100 MHz s_memrealtime).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/l2_stream.hip -o tools/micro/build/l2_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

namespace {
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;

template <int FORM, int DEPTH>
__global__ __launch_bounds__(1024) void stream(const unsigned *w, int frags, int passes, int stagger, long long *cyc,
                                               long long *real, unsigned *sink) {
  extern __shared__ unsigned lds[];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int waves = blockDim.x >> 6;
  const unsigned long long a = (unsigned long long)w;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  __amdgpu_buffer_rsrc_t r =
      __builtin_amdgcn_make_buffer_rsrc((void *)(((unsigned long long)hi << 32) | lo), 0, 0x7FFFFFFF, 0x00020000);
  const int per_wave = frags / waves;          // fragments a wave reads per pass
  const int first = (stagger * (int)blockIdx.x) % per_wave;
  u32x4 v[DEPTH];
  unsigned acc = 0;
  auto load = [&](int k, int i) {   // i-th fragment of this wave (wraps)
    const int f = wave + waves * ((first + i) % per_wave);
    if (FORM == 0) v[k] = __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16, f * 1024, 0);
    if (FORM == 3) v[k] = __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16, f * 1024, 2);
    if (FORM == 1) v[k] = *reinterpret_cast<const u32x4 *>(w + (size_t)f * 256 + lane * 4);
    if (FORM == 4) {
      const u32x2 q = __builtin_amdgcn_raw_buffer_load_b64(r, lane * 8, f * 512, 0);
      v[k] = u32x4{q[0], q[1], 0u, 0u};
    }
    if (FORM == 2)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)(lds + (wave * DEPTH + k) * 256), 16,
                                               lane * 16, f * 1024, 0, 0);
  };
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
  const int total = per_wave * passes;
#pragma unroll
  for (int k = 0; k < DEPTH; ++k) load(k, k);
  for (int i = DEPTH; i < total; i += DEPTH) {
#pragma unroll
    for (int k = 0; k < DEPTH; ++k) {
      if (FORM == 2) {
        // one slot's load must have landed before it is re-targeted: in-order vmcnt, DEPTH - 1 younger ones may fly
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory");
      } else {
        acc += v[k][0] ^ v[k][3];
      }
      load(k, i + k);
    }
  }
  if (FORM == 2) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    acc = lds[(wave * DEPTH) * 256 + lane];
  } else {
#pragma unroll
    for (int k = 0; k < DEPTH; ++k) acc += v[k][0] ^ v[k][3];
  }
  asm volatile("" : "+v"(acc));
  __syncthreads();
  const long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
  if (tid == 0) {
    cyc[blockIdx.x] = t1 - t0;
    real[blockIdx.x] = r1 - r0;
  }
  sink[blockIdx.x * 64 + lane] = acc;
}

template <int FORM, int DEPTH>
void run(const unsigned *w, size_t bytes, int grid, int waves, int stagger, long long *dcyc, long long *dreal, unsigned *sink) {
  const int frags = (int)(bytes / (FORM == 4 ? 512 : 1024));
  const int passes = (int)((size_t)(64 << 20) / bytes) + 1;   // ~64 MiB per workgroup
  const size_t lds = FORM == 2 ? (size_t)waves * DEPTH * 1024 : 0;
  if (lds > 160 * 1024) return;
  if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)stream<FORM, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  for (int rep = 0; rep < 2; ++rep)
    hipLaunchKernelGGL((stream<FORM, DEPTH>), dim3(grid), dim3(64 * waves), lds, 0, w, frags, passes, stagger, dcyc, dreal, sink);
  if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed form %d depth %d\n", FORM, DEPTH); return; }
  std::vector<long long> c(grid), rl(grid);
  (void)hipMemcpy(c.data(), dcyc, grid * sizeof(long long), hipMemcpyDeviceToHost);
  (void)hipMemcpy(rl.data(), dreal, grid * sizeof(long long), hipMemcpyDeviceToHost);
  double sc = 0, sr = 0;
  for (int i = 0; i < grid; ++i) { sc += (double)c[i]; sr += (double)rl[i]; }
  sc /= grid; sr /= grid;
  const double moved = (double)(frags / waves) * waves * passes * (FORM == 4 ? 512.0 : 1024.0);
  printf("form %d depth %2d set %5.1f MiB grid %3d waves %2d stagger %3d : %6.1f B/clk/CU  (%.0f cycles, shader clock %.2f GHz, %.2f TB/s chip)\n",
         FORM, DEPTH, bytes / 1048576.0, grid, waves, stagger, moved / sc, sc, sc / (sr * 10.0) , moved * grid / (sr * 10e-9) / 1e12);
}
}  // namespace

int main() {
  unsigned *w, *sink; long long *dcyc, *dreal;
  const size_t cap = (size_t)64 << 20;
  (void)hipMalloc(&w, cap); (void)hipMemset(w, 1, cap);
  (void)hipMalloc(&sink, 1024 * 64 * 4);
  (void)hipMalloc(&dcyc, 1024 * sizeof(long long));
  (void)hipMalloc(&dreal, 1024 * sizeof(long long));
  for (size_t bytes : {(size_t)1 << 20, (size_t)6 << 20, (size_t)48 << 20})
    for (int grid : {1, 256})
      for (int waves : {4, 8, 16}) {
        run<0, 4>(w, bytes, grid, waves, 0, dcyc, dreal, sink);
        run<0, 8>(w, bytes, grid, waves, 0, dcyc, dreal, sink);
        run<0, 16>(w, bytes, grid, waves, 0, dcyc, dreal, sink);
        run<1, 8>(w, bytes, grid, waves, 0, dcyc, dreal, sink);
        run<2, 4>(w, bytes, grid, waves, 0, dcyc, dreal, sink);
        run<2, 8>(w, bytes, grid, waves, 0, dcyc, dreal, sink);
        run<3, 8>(w, bytes, grid, waves, 0, dcyc, dreal, sink);
        run<4, 8>(w, bytes, grid, waves, 0, dcyc, dreal, sink);
        if (grid > 1) run<0, 8>(w, bytes, grid, waves, 37, dcyc, dreal, sink);
      }
  return 0;
}
