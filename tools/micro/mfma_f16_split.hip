// Does the matrix pipe keep f16 subnormal operands, and how accurate is an f32 product computed from f16 pieces?
//   x = hi + lo  (two f16 numbers: 11 + 11 significant bits),  a b ~ a_hi b_hi + a_hi b_lo + a_lo b_hi   (three MFMAs)
// against the split-bf16 form (three bf16 pieces, six MFMAs) and an f64 reference, on a 16 x 16 x K product whose operands
// span many binades (so that most lo parts are f16 subnormals).  Also: cycles per v_mfma_f32_16x16x32_f16.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/mfma_f16_split.hip -o tools/micro/build/mfma_f16_split
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;

// A [16][K] row major, B [K][16]: lane l holds A[l & 15][8 (l >> 4) + j + 32 kb], B[8 (l >> 4) + j + 32 kb][l & 15]
template <int MODE>   // 0: f16 x 2 (3 products), 1: bf16 x 3 (6 products), 2: f16 x 2 with lo scaled by 2^11 (second accumulator)
__global__ void prod(const float *A, const float *B, float *C, int K) {
  const int l = threadIdx.x, r = l & 15, g = l >> 4;
  f32x4 acc = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
  for (int kb = 0; kb < K / 32; ++kb) {
    float a[8], b[8];
    for (int j = 0; j < 8; ++j) {
      a[j] = A[r * K + 32 * kb + 8 * g + j];
      b[j] = B[(32 * kb + 8 * g + j) * 16 + r];
    }
    if (MODE == 0 || MODE == 2) {
      f16x8 ah, al, bh, bl;
      const float s = MODE == 2 ? 2048.f : 1.f;
      for (int j = 0; j < 8; ++j) {
        ah[j] = (_Float16)a[j]; al[j] = (_Float16)((a[j] - (float)ah[j]) * s);
        bh[j] = (_Float16)b[j]; bl[j] = (_Float16)((b[j] - (float)bh[j]) * s);
      }
      if (MODE == 0) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0);
      } else {
        acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc2, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc2, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0);
      }
    } else {
      bf16x8 ap[3], bp[3];
      for (int j = 0; j < 8; ++j) {
        float x = a[j];
        for (int p = 0; p < 3; ++p) { ap[p][j] = (__bf16)x; x -= (float)ap[p][j]; }
        x = b[j];
        for (int p = 0; p < 3; ++p) { bp[p][j] = (__bf16)x; x -= (float)bp[p][j]; }
      }
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[0], bp[2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[2], bp[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[1], bp[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[0], bp[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[1], bp[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[0], bp[0], acc, 0, 0, 0);
    }
  }
  for (int q = 0; q < 4; ++q) C[(4 * g + q) * 16 + r] = acc[q] + acc2[q] * (1.0f / 2048.f);
}

// subnormal probe: A = 2^-20 everywhere (an f16 subnormal), B = 2^10: sum over 32 k = 32 * 2^-10 if kept, 0 if flushed
__global__ void probe(float *out) {
  f16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)9.5367431640625e-07f; b[j] = (_Float16)1024.f; }
  f32x4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
  out[threadIdx.x] = acc[0];
}

template <bool F16>
__global__ __launch_bounds__(512) void rate(float *out, long long *cyc, int iters) {
  f32x4 acc[4];
  f16x8 a[2], b[4][2];
  bf16x8 ab[3], bb[4][3];
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0, 0, 0, 0};
  for (int p = 0; p < 3; ++p)
    for (int j = 0; j < 8; ++j) {
      if (p < 2) a[p][j] = (_Float16)(1.0f + threadIdx.x * 0.001f + p);
      ab[p][j] = (__bf16)(1.0f + threadIdx.x * 0.001f + p);
      for (int i = 0; i < 4; ++i) {
        if (p < 2) b[i][p][j] = (_Float16)(0.5f + i + p);
        bb[i][p][j] = (__bf16)(0.5f + i + p);
      }
    }
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (F16) {
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[i][1], acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1], b[i][0], acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[i][0], acc[i], 0, 0, 0);
      } else {
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab[0], bb[i][2], acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab[2], bb[i][0], acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab[1], bb[i][1], acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab[0], bb[i][1], acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab[1], bb[i][0], acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab[0], bb[i][0], acc[i], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  const long long t1 = __builtin_readcyclecounter();
  float r = 0;
  for (int i = 0; i < 4; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

int main() {
  float *d; (void)hipMalloc(&d, 64 * 4);
  probe<<<1, 64>>>(d);
  float h[64]; (void)hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
  printf("subnormal probe: sum of 32 x (2^-20 * 2^10) = %.6g (kept: %.6g, flushed: 0)\n", h[0], 32.0 / 1024.0);
  const int K = 768;
  std::mt19937 rng(1);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::uniform_real_distribution<float> ex(-12.f, 3.f);
  for (int wide = 0; wide < 2; ++wide) {
    std::vector<float> A(16 * K), B(K * 16), C(256);
    for (auto &v : A) v = nd(rng) * (wide ? std::exp2(ex(rng)) : 1.f);
    for (auto &v : B) v = nd(rng) * (wide ? std::exp2(ex(rng)) : 1.f);
    float *dA, *dB, *dC;
    (void)hipMalloc(&dA, A.size() * 4); (void)hipMalloc(&dB, B.size() * 4); (void)hipMalloc(&dC, 1024);
    (void)hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    std::vector<double> ref(256), mag(256);
    std::vector<float> f32(256);
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) {
        double s = 0, m = 0; float f = 0.f;
        for (int k = 0; k < K; ++k) {
          s += (double)A[i * K + k] * B[k * 16 + j]; m += std::fabs((double)A[i * K + k] * B[k * 16 + j]);
          f = std::fmaf(A[i * K + k], B[k * 16 + j], f);
        }
        ref[i * 16 + j] = s; mag[i * 16 + j] = m; f32[i * 16 + j] = f;
      }
    double e32 = 0;
    for (int i = 0; i < 256; ++i) e32 = std::fmax(e32, std::fabs(f32[i] - ref[i]) / mag[i]);
    printf("%s operands, K = %d: max |err| / sum |a b|:  f32 fma chain %.2e", wide ? "wide-range" : "unit-scale", K, e32);
    for (int mode = 0; mode < 3; ++mode) {
      if (mode == 0) prod<0><<<1, 64>>>(dA, dB, dC, K);
      else if (mode == 1) prod<1><<<1, 64>>>(dA, dB, dC, K);
      else prod<2><<<1, 64>>>(dA, dB, dC, K);
      (void)hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost);
      double e = 0;
      for (int i = 0; i < 256; ++i) e = std::fmax(e, std::fabs(C[i] - ref[i]) / mag[i]);
      printf("  %s %.2e", mode == 0 ? "f16x2 (3 MFMA)" : (mode == 1 ? "bf16x3 (6 MFMA)" : "f16x2 scaled lo"), e);
    }
    printf("\n");
  }
  float *out; long long *cyc, hc[8];
  (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 64);
  for (int w : {4, 8}) {
    rate<true><<<256, w * 64>>>(out, cyc, 1000); rate<true><<<256, w * 64>>>(out, cyc, 1000);
    (void)hipDeviceSynchronize(); (void)hipMemcpy(hc, cyc, 64, hipMemcpyDeviceToHost);
    printf("waves/CU=%d f16x2 : cycles per 16x16x32 product (3 MFMAs):", w);
    for (int i = 0; i < w; ++i) printf(" %.1f", hc[i] / 4000.0);
    rate<false><<<256, w * 64>>>(out, cyc, 1000); rate<false><<<256, w * 64>>>(out, cyc, 1000);
    (void)hipDeviceSynchronize(); (void)hipMemcpy(hc, cyc, 64, hipMemcpyDeviceToHost);
    printf("\nwaves/CU=%d bf16x3: cycles per 16x16x32 product (6 MFMAs):", w);
    for (int i = 0; i < w; ++i) printf(" %.1f", hc[i] / 4000.0);
    printf("\n");
  }
  return 0;
}
