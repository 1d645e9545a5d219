// split_mix.hip -- is f16(x - f32(f16(x))) by v_fma_mixlo/mixhi_f16 the same bits as by v_cvt_f32_f16 + v_sub + v_cvt_pk_f16_f32?
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/build/split_mix tools/micro/split_mix.hip && tools/micro/build/split_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ void k(const float *x, unsigned *out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * i + 1 >= n) return;
  const float a = x[2 * i], b = x[2 * i + 1];
  const f16x2 h = __builtin_convertvector(f32x2{a, b}, f16x2);
  float ra = a - (float)h[0], rb = b - (float)h[1];
  asm volatile("" : "+v"(ra), "+v"(rb));
  const unsigned lo_ref = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{ra, rb}, f16x2));
  unsigned hi, lo;
  asm("v_cvt_pk_f16_f32 %0, %2, %3\n\t"
      "v_fma_mixlo_f16 %1, %0, -1.0, %2 op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %1, %0, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
      : "=&v"(hi), "=&v"(lo) : "v"(a), "v"(b));
  out[4 * i] = __builtin_bit_cast(unsigned, h); out[4 * i + 1] = hi; out[4 * i + 2] = lo_ref; out[4 * i + 3] = lo;
}
int main() {
  const int n = 1 << 20;
  float *hx = new float[n];
  unsigned s = 12345u;
  for (int i = 0; i < n; ++i) {   // magnitudes 2^-30 .. 2^14, both signs
    s = s * 1664525u + 1013904223u;
    const float m = 1.0f + (float)(s >> 9) * (1.0f / 8388608.0f);
    s = s * 1664525u + 1013904223u;
    const int e = (int)(s >> 24) % 45 - 30;
    hx[i] = ldexpf(m, e) * ((s & 1) ? -1.f : 1.f);
  }
  float *dx; unsigned *dout;
  hipMalloc(&dx, n * 4); hipMalloc(&dout, n * 8);
  hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 2 / 256), dim3(256), 0, 0, dx, dout, n);
  unsigned *ho = new unsigned[2 * n];
  hipMemcpy(ho, dout, n * 8, hipMemcpyDeviceToHost);
  long bad_hi = 0, bad_lo = 0; int shown = 0;
  for (int i = 0; i < n / 2; ++i) {
    if (ho[4 * i] != ho[4 * i + 1]) ++bad_hi;
    if (ho[4 * i + 2] != ho[4 * i + 3]) {
      ++bad_lo;
      if (shown++ < 6) printf("x = %.9g %.9g: lo ref %08x mix %08x\n", hx[2 * i], hx[2 * i + 1], ho[4 * i + 2], ho[4 * i + 3]);
    }
  }
  printf("pairs %d: hi mismatches %ld, lo mismatches %ld\n", n / 2, bad_hi, bad_lo);
  return 0;
}
