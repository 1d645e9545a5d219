// point_ops.hip -- gfx950 kernels for the point-cloud operators of the
// GraspLDM encoder path, behind the C ABI declared in include/gldm.h.
//
// Design notes (CDNA4, wave64):
//  * every launch spreads one cloud over many workgroups where the algorithm
//    allows it (the reference launches ONE block per cloud for all of these);
//  * ball query is a wave-per-centre ordered compaction: 64 candidates per
//    step, __ballot + prefix popcount keep the reference's "first U in index
//    order" semantics exactly;
//  * FPS keeps coordinates and running distances in registers/LDS for all M
//    rounds, one barrier per round, 64-bit (distance, tie-key) wave arg-max;
//  * the voxel scatter-mean is deterministic: points are bitonic-sorted by
//    (voxel, index) in LDS and each voxel is summed in ascending point index;
//  * distance / interpolation arithmetic is compiled with -ffp-contract=off so
//    strict-inequality decisions and 8-corner sums match the scalar oracle bit
//    for bit.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "gldm.h"
#include "devstate.h"

#define GLDM_API extern "C" __attribute__((visibility("default")))

namespace {

constexpr int kWave = 64;

inline int launch_status() { return hipGetLastError() == hipSuccess ? GLDM_OK : GLDM_ERR_LAUNCH; }
inline hipStream_t as_stream(gldm_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// ------------------------------------------------------------ ball query --
// One wave per centre, kCentresPerBlock centres per 256-thread block.  Point
// coordinates are staged once per block in LDS when they fit (n <= 5120),
// otherwise they are read through L2.
constexpr int kBqBlock = 256;
constexpr int kBqMaxLdsPoints = 5120;  // 60 KiB

template <bool kUseLds>
__device__ __forceinline__ int ball_query_wave(const float *px, const float *py, const float *pz, int n, float cx,
                                               float cy, float cz, float r2, int u, int lane, int32_t *o) {
  int cnt = 0, first = 0;
  for (int base = 0; base < n && cnt < u; base += kWave) {
    const int k = base + lane;
    bool hit = false;
    if (k < n) {
      const float dx = cx - px[k];
      const float dy = cy - py[k];
      const float dz = cz - pz[k];
      const float d2 = dx * dx + dy * dy + dz * dz;
      hit = d2 < r2;
    }
    const unsigned long long mask = __ballot(hit);
    if (mask != 0ull) {
      if (cnt == 0) first = base + __ffsll((long long)mask) - 1;
      const int slot = cnt + __popcll(mask & ((1ull << lane) - 1ull));
      if (hit && slot < u) o[slot] = k;
      cnt += __popcll(mask);
    }
  }
  cnt = cnt < u ? cnt : u;
  const int fill = first;  // 0 when the ball is empty
  for (int v = cnt + lane; v < u; v += kWave) o[v] = fill;
  return cnt;
}

template <bool kUseLds>
__global__ __launch_bounds__(kBqBlock) void ball_query_kernel(const float *__restrict__ centers,
                                                              const float *__restrict__ points, int n, int m,
                                                              float r2, int u, int centres_per_block,
                                                              int32_t *__restrict__ out) {
  extern __shared__ float s_pts[];
  const int b = blockIdx.y;
  points += (size_t)b * 3 * n;
  centers += (size_t)b * 3 * m;
  out += (size_t)b * m * u;
  const float *px = points, *py = points + n, *pz = points + 2 * n;
  if (kUseLds) {
    for (int i = threadIdx.x; i < 3 * n; i += kBqBlock) s_pts[i] = points[i];
    __syncthreads();
    px = s_pts;
    py = s_pts + n;
    pz = s_pts + 2 * n;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int j0 = blockIdx.x * centres_per_block;
  const int j1 = min(j0 + centres_per_block, m);
  for (int j = j0 + wave; j < j1; j += kBqBlock / kWave) {
    ball_query_wave<kUseLds>(px, py, pz, n, centers[j], centers[j + m], centers[j + 2 * m], r2, u, lane,
                             out + (size_t)j * u);
  }
}

// -------------------------------------------------------------- grouping --
// Thread = 4 consecutive (centre, neighbour) slots; the 4 indices are loaded
// once and reused for a chunk of channels.  Stores are 16 B per lane.
constexpr int kGrpBlock = 256;
constexpr int kGrpChannelsPerBlock = 16;

__global__ __launch_bounds__(kGrpBlock) void grouping_kernel(const float *__restrict__ feat,
                                                             const int32_t *__restrict__ idx, int c, int n,
                                                             int mu, float *__restrict__ out) {
  const int b = blockIdx.z;
  feat += (size_t)b * c * n;
  idx += (size_t)b * mu;
  out += (size_t)b * c * mu;
  const int c0 = blockIdx.y * kGrpChannelsPerBlock;
  const int c1 = min(c0 + kGrpChannelsPerBlock, c);
  const int p = (blockIdx.x * kGrpBlock + threadIdx.x) * 4;
  if (p >= mu) return;
  if (p + 3 < mu && (mu & 3) == 0) {
    const int4 id = *reinterpret_cast<const int4 *>(idx + p);
    for (int l = c0; l < c1; ++l) {
      const float *f = feat + (size_t)l * n;
      float4 v = make_float4(f[id.x], f[id.y], f[id.z], f[id.w]);
      *reinterpret_cast<float4 *>(out + (size_t)l * mu + p) = v;
    }
  } else {
    for (int q = p; q < min(p + 4, mu); ++q) {
      const int id = idx[q];
      for (int l = c0; l < c1; ++l) out[(size_t)l * mu + q] = feat[(size_t)l * n + id];
    }
  }
}

__global__ __launch_bounds__(256) void gather_kernel(const float *__restrict__ feat,
                                                     const int32_t *__restrict__ idx, int c, int n, int m,
                                                     float *__restrict__ out) {
  const int b = blockIdx.z, l = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= m) return;
  out[((size_t)b * c + l) * m + j] = feat[((size_t)b * c + l) * n + idx[(size_t)b * m + j]];
}

// ------------------------------------------------------------------- FPS --
// One workgroup per cloud.  Thread t owns points t, t+T, t+2T, ... (kept in
// registers together with their running min-distance).  Round winner = arg-max
// of a 64-bit key: high word = distance bits (distances are >= +0 so the bit
// pattern is monotone), low word = ~((k mod 512) << 22 | k), which reproduces
// the reference's tie resolution (strict '>' inside a 512-stride scan, strict
// '<' in the shared-memory tree -> lowest slot, then lowest k).
constexpr int kFpsMaxPerThread = 8;

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const unsigned long long o = __shfl_xor(v, off, kWave);
    v = o > v ? o : v;
  }
  return v;
}

// kRule 0: the reference's CUDA kernel (sampling.cu:86-167): coords [3][n], squared distances, start 1e38.
// kRule 1: the reference's host-side greedy sampler PointCloudHelpers.farthest_points
//          (grasp_ldm/utils/pointcloud_helpers.py:160-217, used by regularize_pc_point_count :124-158):
//          points [n][3], EUCLIDEAN distance sqrt((dx^2 + dy^2) + dz^2) in f32 (numpy's summation order),
//          start 1e7, np.argmax tie rule = lowest index.
template <int kThreads, int kPerThread, int kRule>
__global__ __launch_bounds__(kThreads) void fps_kernel(const float *__restrict__ coords, int n, int m,
                                                       int32_t *__restrict__ out) {
  extern __shared__ float s_xyz[];  // [3][n]
  __shared__ unsigned long long s_part[2][kThreads / kWave];
  const int b = blockIdx.x;
  coords += (size_t)b * 3 * n;
  out += (size_t)b * m;
  const int tid = threadIdx.x;
  if (kRule == 0) {
    for (int i = tid; i < 3 * n; i += kThreads) s_xyz[i] = coords[i];
  } else {
    for (int i = tid; i < 3 * n; i += kThreads) s_xyz[(i % 3) * n + i / 3] = coords[i];
  }
  float x[kPerThread], y[kPerThread], z[kPerThread], dist[kPerThread];
#pragma unroll
  for (int q = 0; q < kPerThread; ++q) {
    const int k = tid + q * kThreads;
    const bool ok = k < n;
    x[q] = ok ? coords[kRule == 0 ? k : 3 * k] : 0.f;
    y[q] = ok ? coords[kRule == 0 ? k + n : 3 * k + 1] : 0.f;
    z[q] = ok ? coords[kRule == 0 ? k + 2 * n : 3 * k + 2] : 0.f;
    dist[q] = kRule == 0 ? 1e38f : 1e7f;
  }
  if (tid == 0) out[0] = 0;
  __syncthreads();
  int old = 0;
  const int wave = tid >> 6, lane = tid & 63;
  constexpr int kWaves = kThreads / kWave;
  for (int j = 1; j < m; ++j) {
    const float x1 = s_xyz[old], y1 = s_xyz[old + n], z1 = s_xyz[old + 2 * n];
    unsigned long long best = 0ull;  // below every real key (low word of a real key is never 0)
#pragma unroll
    for (int q = 0; q < kPerThread; ++q) {
      const int k = tid + q * kThreads;
      if (k < n) {
        float d = (x[q] - x1) * (x[q] - x1) + (y[q] - y1) * (y[q] - y1) + (z[q] - z1) * (z[q] - z1);
        if (kRule == 1) d = __fsqrt_rn(d);
        const float d2 = d < dist[q] ? d : dist[q];
        dist[q] = d2;
        const unsigned int tie = kRule == 0 ? ~(((unsigned int)(k & 511) << 22) | (unsigned int)k) : ~(unsigned int)k;
        const unsigned long long key = ((unsigned long long)__float_as_uint(d2) << 32) | tie;
        best = key > best ? key : best;
      }
    }
    best = wave_max_u64(best);
    if (kWaves > 1) {
      if (lane == 0) s_part[j & 1][wave] = best;
      __syncthreads();
      unsigned long long v = lane < kWaves ? s_part[j & 1][lane] : 0ull;
#pragma unroll
      for (int off = kWaves / 2; off >= 1; off >>= 1) {
        const unsigned long long o = __shfl_xor(v, off, kWave);
        v = o > v ? o : v;
      }
      best = __shfl(v, 0, kWave);
    }
    old = (int)((~(unsigned int)best) & 0x3FFFFFu);
    if (tid == 0) out[j] = old;
  }
}

// One WAVE per cloud (n <= 1024): lane l owns points l, l + 64, ... (kPer of them, in registers with their running
// min-distances), so a round is 10 VALU instructions per point, an in-lane arg-max and ONE wave reduction -- no LDS
// exchange, no barrier (the 16-wave form above spends most of a round's 1.2 us in its two reductions and the barrier
// between them: 0.62 ms per 1024 -> 512 sampling whatever the batch).  The reduction runs on the two 32-bit halves of the
// key one after the other (distance bits, then the tie word among the lanes that hold the maximal distance): four DPP row
// steps + four v_readlane each.  The previous winner's coordinates come from LDS (a broadcast read).  Same keys, same
// arithmetic (-ffp-contract=off) as fps_kernel: bit-identical picks.
template <int CTRL>
__device__ __forceinline__ unsigned dpp_shr_u32(unsigned v) {   // lanes without a source read 0 (the identity of max)
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
  v = max(v, dpp_shr_u32<0x111>(v));   // row_shr:1
  v = max(v, dpp_shr_u32<0x112>(v));   // row_shr:2
  v = max(v, dpp_shr_u32<0x114>(v));   // row_shr:4
  v = max(v, dpp_shr_u32<0x118>(v));   // row_shr:8: lane 15 of every row holds the row's maximum
  const unsigned a = __builtin_amdgcn_readlane(v, 15), b = __builtin_amdgcn_readlane(v, 31);
  const unsigned c = __builtin_amdgcn_readlane(v, 47), d = __builtin_amdgcn_readlane(v, 63);
  return max(max(a, b), max(c, d));
}
template <int kPer, int kRule>
__global__ __launch_bounds__(64) void fps_wave_kernel(const float *__restrict__ coords, int n, int m,
                                                      int32_t *__restrict__ out) {
  extern __shared__ float s_xyz[];  // [3][n]
  const int b = blockIdx.x, lane = threadIdx.x;
  coords += (size_t)b * 3 * n;
  out += (size_t)b * m;
  // Slots beyond n: distance 0 for ever (min(d, 0) = 0) and tie word 0, i.e. key 0, below every real key (whose low word
  // is never 0): the round needs no per-point test (16 exec-masked branches per round otherwise).
  float x[kPer], y[kPer], z[kPer], dist[kPer];
  unsigned tiew[kPer];
#pragma unroll
  for (int q = 0; q < kPer; ++q) {
    const int k = lane + 64 * q;
    const bool ok = k < n;
    x[q] = ok ? coords[kRule == 0 ? k : 3 * k] : 0.f;
    y[q] = ok ? coords[kRule == 0 ? k + n : 3 * k + 1] : 0.f;
    z[q] = ok ? coords[kRule == 0 ? k + 2 * n : 3 * k + 2] : 0.f;
    dist[q] = ok ? (kRule == 0 ? 1e38f : 1e7f) : 0.f;
    const unsigned int tie = kRule == 0 ? ~(((unsigned int)(k & 511) << 22) | (unsigned int)k) : ~(unsigned int)k;
    tiew[q] = ok ? tie : 0u;
    if (ok) {
      s_xyz[k] = x[q];
      s_xyz[k + n] = y[q];
      s_xyz[k + 2 * n] = z[q];
    }
  }
  if (lane == 0) out[0] = 0;
  __syncthreads();
  int old = 0;
  for (int j = 1; j < m; ++j) {
    const float x1 = s_xyz[old], y1 = s_xyz[old + n], z1 = s_xyz[old + 2 * n];
    unsigned long long best = 0ull;
#pragma unroll
    for (int q = 0; q < kPer; ++q) {
      float d = (x[q] - x1) * (x[q] - x1) + (y[q] - y1) * (y[q] - y1) + (z[q] - z1) * (z[q] - z1);
      if (kRule == 1) d = __fsqrt_rn(d);
      const float d2 = d < dist[q] ? d : dist[q];
      dist[q] = d2;
      const unsigned long long key = ((unsigned long long)__float_as_uint(d2) << 32) | tiew[q];
      best = key > best ? key : best;
    }
    const unsigned hi = (unsigned)(best >> 32), lo = (unsigned)best;
    const unsigned hmax = wave_max_u32(hi);
    const unsigned lmax = wave_max_u32(hi == hmax ? lo : 0u);
    old = (int)((~lmax) & 0x3FFFFFu);
    if (lane == 0) out[j] = old;
  }
}

// -------------------------------------------------- 3-NN + interpolation --
__global__ __launch_bounds__(256) void three_nn_kernel(const float *__restrict__ points,
                                                       const float *__restrict__ centers, int n, int m,
                                                       int32_t *__restrict__ idx, float *__restrict__ wgt) {
  extern __shared__ float s_c[];  // [3][m]
  const int b = blockIdx.y;
  points += (size_t)b * 3 * n;
  centers += (size_t)b * 3 * m;
  idx += (size_t)b * 3 * n;
  wgt += (size_t)b * 3 * n;
  for (int i = threadIdx.x; i < 3 * m; i += 256) s_c[i] = centers[i];
  __syncthreads();
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const float ux = points[j], uy = points[j + n], uz = points[j + 2 * n];
  double best0 = 1e40, best1 = 1e40, best2 = 1e40;
  int i0 = 0, i1 = 0, i2 = 0;
  for (int k = 0; k < m; ++k) {
    const float cx = s_c[k], cy = s_c[k + m], cz = s_c[k + 2 * m];
    const float d = (ux - cx) * (ux - cx) + (uy - cy) * (uy - cy) + (uz - cz) * (uz - cz);
    if (d < best2) {
      best2 = d; i2 = k;
      if (d < best1) {
        best2 = best1; i2 = i1;
        best1 = d; i1 = k;
        if (d < best0) {
          best1 = best0; i1 = i0;
          best0 = d; i0 = k;
        }
      }
    }
  }
  best0 = fmax(fmin((double)1e10f, best0), (double)1e-10f);
  best1 = fmax(fmin((double)1e10f, best1), (double)1e-10f);
  best2 = fmax(fmin((double)1e10f, best2), (double)1e-10f);
  const float d0d1 = (float)(best0 * best1);
  const float d0d2 = (float)(best0 * best2);
  const float d1d2 = (float)(best1 * best2);
  const float inv = 1.0f / (d0d1 + d0d2 + d1d2);
  wgt[j] = d1d2 * inv;
  idx[j] = i0;
  wgt[j + n] = d0d2 * inv;
  idx[j + n] = i1;
  wgt[j + 2 * n] = d0d1 * inv;
  idx[j + 2 * n] = i2;
}

__global__ __launch_bounds__(256) void three_interp_kernel(const float *__restrict__ cfeat,
                                                           const int32_t *__restrict__ idx,
                                                           const float *__restrict__ wgt, int c, int m, int n,
                                                           float *__restrict__ out) {
  const int b = blockIdx.z;
  cfeat += (size_t)b * c * m;
  idx += (size_t)b * 3 * n;
  wgt += (size_t)b * 3 * n;
  out += (size_t)b * c * n;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const float w1 = wgt[j], w2 = wgt[j + n], w3 = wgt[j + 2 * n];
  const int a1 = idx[j], a2 = idx[j + n], a3 = idx[j + 2 * n];
  const int c0 = blockIdx.y * 16, c1 = min(c0 + 16, c);
  for (int l = c0; l < c1; ++l) {
    const float *f = cfeat + (size_t)l * m;
    out[(size_t)l * n + j] = f[a1] * w1 + f[a2] * w2 + f[a3] * w3;
  }
}

// -------------------------------------------------------- voxel coords ----
// One block per cloud: f64 tree mean per axis, optional max-norm scaling,
// clamp, round-half-even (rintf == torch.round).
constexpr int kVcBlock = 256;

__device__ __forceinline__ double block_sum_f64(double v, double *s_red) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, kWave);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) s_red[wave] = v;
  __syncthreads();
  double t = 0.0;
  for (int w = 0; w < kVcBlock / kWave; ++w) t += s_red[w];
  return t;
}

__global__ __launch_bounds__(kVcBlock) void voxel_coords_kernel(const float *__restrict__ coords, int n, int r,
                                                                int normalize, float eps,
                                                                float *__restrict__ norm_coords,
                                                                int32_t *__restrict__ vox) {
  __shared__ double s_red[kVcBlock / kWave];
  __shared__ float s_max[kVcBlock / kWave];
  const int b = blockIdx.x;
  coords += (size_t)b * 3 * n;
  norm_coords += (size_t)b * 3 * n;
  vox += (size_t)b * 3 * n;
  float mean[3];
  for (int a = 0; a < 3; ++a) {
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += kVcBlock) s += (double)coords[a * n + i];
    mean[a] = (float)(block_sum_f64(s, s_red) / (double)n);
  }
  float denom = 1.f;
  if (normalize) {
    float mx = 0.f;
    for (int i = threadIdx.x; i < n; i += kVcBlock) {
      const float x = coords[i] - mean[0], y = coords[n + i] - mean[1], z = coords[2 * n + i] - mean[2];
      mx = fmaxf(mx, sqrtf(x * x + y * y + z * z));
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, kWave));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = 0.f;
    for (int w = 0; w < kVcBlock / kWave; ++w) mx = fmaxf(mx, s_max[w]);
    denom = mx * 2.0f + eps;
  }
  const float rf = (float)r, hi = (float)(r - 1);
  for (int i = threadIdx.x; i < 3 * n; i += kVcBlock) {
    const int a = i / n;
    float v = coords[i] - mean[a];
    v = normalize ? (v / denom + 0.5f) : ((v + 1.f) / 2.0f);
    v = fminf(fmaxf(v * rf, 0.f), hi);
    norm_coords[i] = v;
    vox[i] = (int32_t)rintf(v);
  }
}

// --------------------------------------------------------- avg voxelize ---
// grid = (channel chunks, clouds).  Each block sorts (voxel, point) keys with a
// bitonic network in LDS, then every segment leader sums its voxel's points in
// ascending point index for the block's channels.
constexpr int kVoxBlock = 1024;
constexpr int kVoxChannelsPerBlock = 8;

template <int kPerThread>
__global__ __launch_bounds__(kVoxBlock) void avg_voxelize_kernel(const float *__restrict__ feat,
                                                                 const int32_t *__restrict__ vc, int c, int n,
                                                                 int np2, int log_np2, int r,
                                                                 float *__restrict__ out,
                                                                 int32_t *__restrict__ ind,
                                                                 int32_t *__restrict__ cnt) {
  extern __shared__ unsigned int s_key[];  // [np2]
  const int b = blockIdx.y;
  const int r2 = r * r, r3 = r2 * r;
  feat += (size_t)b * c * n;
  vc += (size_t)b * 3 * n;
  out += (size_t)b * c * r3;
  ind += (size_t)b * n;
  cnt += (size_t)b * r3;
  const int tid = threadIdx.x;
  for (int i = tid; i < np2; i += kVoxBlock) {
    unsigned int key = 0xFFFFFFFFu;
    if (i < n) {
      const int v = vc[i] * r2 + vc[i + n] * r + vc[i + 2 * n];
      key = ((unsigned int)v << log_np2) | (unsigned int)i;
      if (blockIdx.x == 0) ind[i] = v;
    }
    s_key[i] = key;
  }
  __syncthreads();
  for (int size = 2; size <= np2; size <<= 1) {
    for (int stride = size >> 1; stride >= 1; stride >>= 1) {
      for (int t = tid; t < np2 / 2; t += kVoxBlock) {
        const int lo = 2 * t - (t & (stride - 1));
        const int hi = lo + stride;
        const bool up = (lo & size) == 0;
        const unsigned int a = s_key[lo], bb = s_key[hi];
        if ((a > bb) == up) {
          s_key[lo] = bb;
          s_key[hi] = a;
        }
      }
      __syncthreads();
    }
  }
  const unsigned int imask = (1u << log_np2) - 1u;
  const int c0 = blockIdx.x * kVoxChannelsPerBlock, c1 = min(c0 + kVoxChannelsPerBlock, c);
  for (int p = tid; p < n; p += kVoxBlock) {
    const unsigned int key = s_key[p];
    const unsigned int v = key >> log_np2;
    if (p > 0 && (s_key[p - 1] >> log_np2) == v) continue;  // not a segment leader
    int len = 1;
    while (p + len < n && (s_key[p + len] >> log_np2) == v) ++len;
    if (blockIdx.x == 0) cnt[v] = len;
    const float div = (float)(1.0 / (double)(float)len);
    for (int l = c0; l < c1; ++l) {
      const float *f = feat + (size_t)l * n;
      float acc = 0.f;
      for (int q = 0; q < len; ++q) acc += f[s_key[p + q] & imask] * div;
      out[(size_t)l * r3 + v] = acc;
    }
  }
}

// The same scatter-mean with the grid assembled in LDS and written out DENSE: a block sorts its cloud's keys once,
// then walks its channels in rounds of G: zero a [G][r^3] grid in LDS, every segment leader puts its voxel's means
// there (the same sums in the same order as above), and the block writes the rows out with coalesced 16-byte stores.
// The output needs no memset and no scattered 4-byte stores: 0.157 -> see profiles (the 24^3 x 3 and 12^3 x 48 grids of the
// shipped encoder).  Block (0, b) also writes the dense count grid.  r^3 % 4 == 0, keys + one grid row within LDS.
__global__ __launch_bounds__(kVoxBlock) void avg_voxelize_dense_kernel(const float *__restrict__ feat,
                                                                       const int32_t *__restrict__ vc, int c, int n,
                                                                       int np2, int log_np2, int r, int chunk, int g_rows,
                                                                       int stage_feat, float *__restrict__ out,
                                                                       int32_t *__restrict__ ind,
                                                                       int32_t *__restrict__ cnt) {
  extern __shared__ unsigned int s_key[];  // [np2] keys, [np2] occupied-voxel list, a counter, then [g_rows][r^3] floats (+ features)
  const int b = blockIdx.y;
  const int r2 = r * r, r3 = r2 * r;
  feat += (size_t)b * c * n;
  vc += (size_t)b * 3 * n;
  out += (size_t)b * c * r3;
  ind += (size_t)b * n;
  cnt += (size_t)b * r3;
  const int tid = threadIdx.x;
  for (int i = tid; i < np2; i += kVoxBlock) {
    unsigned int key = 0xFFFFFFFFu;
    if (i < n) {
      const int v = vc[i] * r2 + vc[i + n] * r + vc[i + 2 * n];
      key = ((unsigned int)v << log_np2) | (unsigned int)i;
      if (blockIdx.x == 0) ind[i] = v;
    }
    s_key[i] = key;
  }
  __syncthreads();
  if (np2 <= kVoxBlock) {
    // one key per thread, in a register: strides below the wave width are exchanged with shuffles (no barrier), the
    // others through LDS -- 10 barrier-separated stages instead of 55 for 1024 points
    unsigned int key = tid < np2 ? s_key[tid] : 0xFFFFFFFFu;
    for (int size = 2; size <= np2; size <<= 1) {
      const bool up = (tid & size) == 0;
      for (int stride = size >> 1; stride >= 1; stride >>= 1) {
        unsigned int other;
        if (stride >= kWave) {
          __syncthreads();   // the previous exchange's readers are done
          if (tid < np2) s_key[tid] = key;
          __syncthreads();
          other = tid < np2 ? s_key[tid ^ stride] : 0xFFFFFFFFu;
        } else {
          other = (unsigned int)__shfl_xor((int)key, stride, kWave);
        }
        const bool lower = (tid & stride) == 0;
        const unsigned int mn = key < other ? key : other, mx = key < other ? other : key;
        key = (lower == up) ? mn : mx;
      }
    }
    __syncthreads();
    if (tid < np2) s_key[tid] = key;
    __syncthreads();
  } else {
    for (int size = 2; size <= np2; size <<= 1) {
      for (int stride = size >> 1; stride >= 1; stride >>= 1) {
        for (int t = tid; t < np2 / 2; t += kVoxBlock) {
          const int lo = 2 * t - (t & (stride - 1));
          const int hi = lo + stride;
          const bool up = (lo & size) == 0;
          const unsigned int a = s_key[lo], bb = s_key[hi];
          if ((a > bb) == up) {
            s_key[lo] = bb;
            s_key[hi] = a;
          }
        }
        __syncthreads();
      }
    }
  }
  const unsigned int imask = (1u << log_np2) - 1u;
  const int c0 = blockIdx.x * chunk, c1 = min(c0 + chunk, c);
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  // ---- the occupied voxels as a list (leader position | points << 16), once per block.  A voxel's sum is a serial
  // chain (ascending point index: the order the result is defined in), but the chains of different channels and voxels
  // are independent: the rounds below deal (voxel, channel) pairs to the threads, so that a crowded voxel (70 points in
  // one 12^3 voxel of an unnormalised cloud) costs 70 steps on a handful of threads, not 70 x channels on one.
  unsigned int *s_lead = s_key + np2;            // [n]
  int *s_count = reinterpret_cast<int *>(s_lead + np2);
  float *grid = reinterpret_cast<float *>(s_count + 4);
  if (tid == 0) *s_count = 0;
  __syncthreads();
  for (int p = tid; p < n; p += kVoxBlock) {
    const unsigned int v = s_key[p] >> log_np2;
    if (p > 0 && (s_key[p - 1] >> log_np2) == v) continue;  // not a segment leader
    int len = 1;
    while (p + len < n && (s_key[p + len] >> log_np2) == v) ++len;
    s_lead[atomicAdd(s_count, 1)] = (unsigned int)p | ((unsigned int)len << 16);
  }
  __syncthreads();
  const int n_lead = *s_count;
  // the count grid: one round of its own, by the cloud's first block
  if (blockIdx.x == 0) {
    for (int i = tid; i < r3 / 4; i += kVoxBlock) reinterpret_cast<float4 *>(grid)[i] = z4;
    __syncthreads();
    for (int i = tid; i < n_lead; i += kVoxBlock) {
      const unsigned int e = s_lead[i];
      reinterpret_cast<int *>(grid)[s_key[e & 0xFFFFu] >> log_np2] = (int)(e >> 16);
    }
    __syncthreads();
    for (int i = tid; i < r3 / 4; i += kVoxBlock)
      reinterpret_cast<float4 *>(cnt)[i] = reinterpret_cast<const float4 *>(grid)[i];
    __syncthreads();
  }
  // feature rows of the round: staged in LDS behind the grid when they fit (coalesced loads once, instead of global
  // round trips inside the chains).  stage_feat == 2: fetched as 16-byte loads into registers a round AHEAD (requested
  // before the previous round's chains, stored after them): with one block per CU nothing else hides those round trips
  float *sf = grid + (size_t)g_rows * r3;
  // four named registers (an indexed float4[4] behind the lambda and the data-dependent round loop was kept in scratch:
  // 96 B per lane of private memory in a latency-bound kernel)
  float4 pre0 = z4, pre1 = z4, pre2 = z4, pre3 = z4;
  auto fetch = [&](int l0, int rows) {
    const float4 *src = reinterpret_cast<const float4 *>(feat + (size_t)l0 * n);
    const int cnt4 = (rows * n) >> 2;
    // unconditional loads on clamped addresses; lanes beyond the rows keep a value nobody stores
    const int last = cnt4 > 0 ? cnt4 - 1 : 0;
    pre0 = src[min(tid, last)];
    pre1 = src[min(tid + kVoxBlock, last)];
    pre2 = src[min(tid + 2 * kVoxBlock, last)];
    pre3 = src[min(tid + 3 * kVoxBlock, last)];
  };
  if (stage_feat == 2) fetch(c0, min(g_rows, c1 - c0));
  for (int l0 = c0; l0 < c1; l0 += g_rows) {
    const int rows = min(g_rows, c1 - l0);
    for (int i = tid; i < rows * (r3 / 4); i += kVoxBlock) reinterpret_cast<float4 *>(grid)[i] = z4;
    if (stage_feat == 2) {
      const int cnt4 = (rows * n) >> 2;
      float4 *sf4 = reinterpret_cast<float4 *>(sf);
      if (tid < cnt4) sf4[tid] = pre0;
      if (tid + kVoxBlock < cnt4) sf4[tid + kVoxBlock] = pre1;
      if (tid + 2 * kVoxBlock < cnt4) sf4[tid + 2 * kVoxBlock] = pre2;
      if (tid + 3 * kVoxBlock < cnt4) sf4[tid + 3 * kVoxBlock] = pre3;
    } else if (stage_feat) {
      for (int i = tid; i < rows * n; i += kVoxBlock) sf[i] = feat[(size_t)l0 * n + i];
    }
    __syncthreads();
    if (stage_feat == 2 && l0 + g_rows < c1) fetch(l0 + g_rows, min(g_rows, c1 - l0 - g_rows));
    // (voxel, channel) pairs, voxel-minor: neighbouring lanes read different points of one row
    for (int w = tid; w < n_lead * rows; w += kVoxBlock) {
      const int l = w / n_lead, id = w - l * n_lead;
      const unsigned int e = s_lead[id];
      const int p = (int)(e & 0xFFFFu), len = (int)(e >> 16);
      const float div = (float)(1.0 / (double)(float)len);
      const float *f = stage_feat ? sf + (size_t)l * n : feat + (size_t)(l0 + l) * n;
      float acc = 0.f;
      for (int q = 0; q < len; ++q) acc += f[s_key[p + q] & imask] * div;
      grid[(size_t)l * r3 + (s_key[p] >> log_np2)] = acc;
    }
    __syncthreads();
    float4 *o4 = reinterpret_cast<float4 *>(out + (size_t)l0 * r3);
    for (int i = tid; i < rows * (r3 / 4); i += kVoxBlock) o4[i] = reinterpret_cast<const float4 *>(grid)[i];
    __syncthreads();
  }
}

// ------------------------------------------------------ devoxelize ---------
constexpr int kDevBlock = 256;
constexpr int kDevChannelsPerBlock = 16;

__global__ __launch_bounds__(kDevBlock) void trilinear_devoxelize_kernel(const float *__restrict__ coords,
                                                                         const float *__restrict__ feat, int c,
                                                                         int n, int r, int is_training,
                                                                         float *__restrict__ outs,
                                                                         int32_t *__restrict__ inds,
                                                                         float *__restrict__ wgts) {
  const int b = blockIdx.z;
  const int r2 = r * r, r3 = r2 * r;
  coords += (size_t)b * 3 * n;
  feat += (size_t)b * c * r3;
  outs += (size_t)b * c * n;
  const int i = blockIdx.x * kDevBlock + threadIdx.x;
  if (i >= n) return;
  const float x = coords[i], y = coords[i + n], z = coords[i + 2 * n];
  const float xl = floorf(x), yl = floorf(y), zl = floorf(z);
  const float xd1 = x - xl, yd1 = y - yl, zd1 = z - zl;
  const float xd0 = 1.0f - xd1, yd0 = 1.0f - yd1, zd0 = 1.0f - zd1;
  const float w000 = xd0 * yd0 * zd0, w001 = xd0 * yd0 * zd1;
  const float w010 = xd0 * yd1 * zd0, w011 = xd0 * yd1 * zd1;
  const float w100 = xd1 * yd0 * zd0, w101 = xd1 * yd0 * zd1;
  const float w110 = xd1 * yd1 * zd0, w111 = xd1 * yd1 * zd1;
  const int xlo = (int)xl, ylo = (int)yl, zlo = (int)zl;
  const int xhi = (xd1 > 0) ? -1 : 0, yhi = (yd1 > 0) ? -1 : 0, zhi = (zd1 > 0) ? 1 : 0;
  const int i000 = xlo * r2 + ylo * r + zlo;
  const int i001 = i000 + zhi;
  const int i010 = i000 + (yhi & r);
  const int i011 = i010 + zhi;
  const int i100 = i000 + (xhi & r2);
  const int i101 = i100 + zhi;
  const int i110 = i100 + (yhi & r);
  const int i111 = i110 + zhi;
  if (is_training && blockIdx.y == 0) {
    float *w = wgts + (size_t)b * 8 * n;
    int32_t *id = inds + (size_t)b * 8 * n;
    w[i] = w000; w[i + n] = w001; w[i + 2 * n] = w010; w[i + 3 * n] = w011;
    w[i + 4 * n] = w100; w[i + 5 * n] = w101; w[i + 6 * n] = w110; w[i + 7 * n] = w111;
    id[i] = i000; id[i + n] = i001; id[i + 2 * n] = i010; id[i + 3 * n] = i011;
    id[i + 4 * n] = i100; id[i + 5 * n] = i101; id[i + 6 * n] = i110; id[i + 7 * n] = i111;
  }
  const int c0 = blockIdx.y * kDevChannelsPerBlock, c1 = min(c0 + kDevChannelsPerBlock, c);
  for (int l = c0; l < c1; ++l) {
    const float *f = feat + (size_t)l * r3;
    outs[(size_t)l * n + i] = w000 * f[i000] + w001 * f[i001] + w010 * f[i010] + w011 * f[i011] +
                              w100 * f[i100] + w101 * f[i101] + w110 * f[i110] + w111 * f[i111];
  }
}

// ------------------------------------------------- fused SA gather ---------
// BallQuery.forward as one kernel.  Block = kSaCentres centres of one cloud:
//   phase 1: wave-per-centre ordered ball query (points staged in LDS), the
//            neighbour indices stay in LDS;
//   phase 2: every channel row of the grouped tensor for these centres is a
//            contiguous run of kSaCentres*u floats in HBM -> 16-byte coalesced
//            stores; reads are index gathers served from L2 (the per-cloud
//            feature slab is <= 512 KiB and shared by all blocks of the cloud).
constexpr int kSaBlock = 256;
typedef float f32x4_t __attribute__((ext_vector_type(4)));

template <bool kUseLds>
__global__ __launch_bounds__(kSaBlock) void sa_group_kernel(const float *__restrict__ points,
                                                            const float *__restrict__ centers,
                                                            const float *__restrict__ feat, int c, int n, int m,
                                                            float r2, int u, int centres_per_block,
                                                            float *__restrict__ out,
                                                            int32_t *__restrict__ idx_out) {
  extern __shared__ float s_mem[];
  const int b = blockIdx.y;
  points += (size_t)b * 3 * n;
  centers += (size_t)b * 3 * m;
  if (feat) feat += (size_t)b * c * n;
  out += (size_t)b * (3 + c) * m * u;
  const int j0 = blockIdx.x * centres_per_block;
  const int nj = min(centres_per_block, m - j0);
  int32_t *s_idx = reinterpret_cast<int32_t *>(s_mem);       // [centres_per_block*u]
  float *s_ctr = s_mem + centres_per_block * u;              // [3][centres_per_block]
  float *s_pts = s_ctr + 3 * centres_per_block;              // [3][n] (kUseLds)
  const float *px = points, *py = points + n, *pz = points + 2 * n;
  if (kUseLds) {
    for (int i = threadIdx.x; i < 3 * n; i += kSaBlock) s_pts[i] = points[i];
    px = s_pts;
    py = s_pts + n;
    pz = s_pts + 2 * n;
  }
  for (int i = threadIdx.x; i < 3 * nj; i += kSaBlock) {
    const int a = i / nj, jj = i - a * nj;
    s_ctr[a * centres_per_block + jj] = centers[a * m + j0 + jj];
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int jj = wave; jj < nj; jj += kSaBlock / kWave) {
    ball_query_wave<kUseLds>(px, py, pz, n, s_ctr[jj], s_ctr[centres_per_block + jj],
                             s_ctr[2 * centres_per_block + jj], r2, u, lane, s_idx + jj * u);
  }
  __syncthreads();
  const int run = nj * u;  // contiguous floats per channel row for this block
  if (idx_out) {
    int32_t *io = idx_out + ((size_t)b * m + j0) * u;
    for (int p = threadIdx.x; p < run; p += kSaBlock) io[p] = s_idx[p];
  }
  const size_t mu = (size_t)m * u;
  float *obase = out + (size_t)j0 * u;
  // coordinate rows: neighbour minus centre
  for (int p = threadIdx.x; p < 3 * run; p += kSaBlock) {
    const int a = p / run, q = p - a * run;
    const int id = s_idx[q];
    const int jj = q / u;
    const float v = (a == 0 ? px[id] : (a == 1 ? py[id] : pz[id])) - s_ctr[a * centres_per_block + jj];
    obase[(size_t)a * mu + q] = v;
  }
  // feature rows
  if (feat) {
    if ((run & 3) == 0 && (mu & 3) == 0) {
      const int run4 = run >> 2;
      for (int p = threadIdx.x; p < run4; p += kSaBlock) {
        const int4 id = *reinterpret_cast<const int4 *>(s_idx + 4 * p);
        float *o = obase + 3 * mu + 4 * p;
#pragma unroll 4
        for (int l = 0; l < c; ++l) {
          const float *f = feat + (size_t)l * n;
          f32x4_t v = {f[id.x], f[id.y], f[id.z], f[id.w]};
          __builtin_nontemporal_store(v, reinterpret_cast<f32x4_t *>(o + (size_t)l * mu));  // streamed once
        }
      }
    } else {
      for (int p = threadIdx.x; p < run; p += kSaBlock) {
        const int id = s_idx[p];
        for (int l = 0; l < c; ++l) obase[(3 + (size_t)l) * mu + p] = feat[(size_t)l * n + id];
      }
    }
  }
}

int ilog2_ceil(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}

}  // namespace

// =========================================================== C ABI =========

GLDM_API int gldm_abi_version(void) { return 10; }

// ---------------------------------------------------------------- row max --
// out[row] = max over n of x[row][0..n): the global pooling of PointNetAModule (pointnet.py:40-44, `.max(dim=-1)`) over
// [b * c] rows.  One wave per row, 16-byte loads where the row allows, DPP-free shuffle reduction (a row is a few hundred
// floats: the launch is one pass over x).  NaN propagates as in torch.max (a NaN in the row -> NaN).
namespace {
__global__ __launch_bounds__(256) void row_max_kernel(const float *__restrict__ x, long long rows, int n, float *__restrict__ out) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const float *p = x + row * n;
  float m = -__builtin_inff();
  bool nan = false;
  if ((n & 3) == 0 && (((size_t)p & 15) == 0)) {
    for (int i = lane; i < (n >> 2); i += 64) {
      const float4 v = reinterpret_cast<const float4 *>(p)[i];
      nan |= (v.x != v.x) | (v.y != v.y) | (v.z != v.z) | (v.w != v.w);
      m = fmaxf(fmaxf(m, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
    }
  } else {
    for (int i = lane; i < n; i += 64) {
      const float v = p[i];
      nan |= v != v;
      m = fmaxf(m, v);
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, kWave));
  const bool any_nan = __ballot(nan) != 0ull;
  if (lane == 0) out[row] = any_nan ? __builtin_nanf("") : m;
}
}  // namespace

GLDM_API int gldm_row_max(const float *x, long long rows, int n, float *out, gldm_stream_t stream) {
  if (!x || !out || rows <= 0 || n <= 0) return GLDM_ERR_INVALID_ARG;
  hipLaunchKernelGGL(row_max_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, rows, n, out);
  return hipGetLastError() == hipSuccess ? GLDM_OK : GLDM_ERR_LAUNCH;
}

GLDM_API const char *gldm_status_string(int status) {
  switch (status) {
    case GLDM_OK: return "ok";
    case GLDM_ERR_INVALID_ARG: return "invalid argument (null pointer or non-positive size)";
    case GLDM_ERR_LAUNCH: return "HIP kernel launch failed";
    case GLDM_ERR_UNSUPPORTED: return "shape not supported by the gfx950 kernels";
    case GLDM_ERR_WORKSPACE: return "workspace too small";
    default: return "unknown gldm status";
  }
}

GLDM_API int gldm_ball_query(const float *centers, const float *points, int b, int n, int m, float radius, int u,
                             int32_t *out, gldm_stream_t stream) {
  if (!centers || !points || !out || b <= 0 || n <= 0 || m <= 0 || u <= 0) return GLDM_ERR_INVALID_ARG;
  const float r2 = radius * radius;
  const int cpb = 16;
  dim3 grid(ceil_div(m, cpb), b);
  if (n <= kBqMaxLdsPoints) {
    hipLaunchKernelGGL(ball_query_kernel<true>, grid, dim3(kBqBlock), (size_t)3 * n * sizeof(float),
                       as_stream(stream), centers, points, n, m, r2, u, cpb, out);
  } else {
    hipLaunchKernelGGL(ball_query_kernel<false>, grid, dim3(kBqBlock), 0, as_stream(stream), centers, points, n,
                       m, r2, u, cpb, out);
  }
  return launch_status();
}

GLDM_API int gldm_grouping_forward(const float *features, const int32_t *idx, int b, int c, int n, int m, int u,
                                   float *out, gldm_stream_t stream) {
  if (!features || !idx || !out || b <= 0 || c <= 0 || n <= 0 || m <= 0 || u <= 0) return GLDM_ERR_INVALID_ARG;
  const int mu = m * u;
  dim3 grid(ceil_div(ceil_div(mu, 4), kGrpBlock), ceil_div(c, kGrpChannelsPerBlock), b);
  hipLaunchKernelGGL(grouping_kernel, grid, dim3(kGrpBlock), 0, as_stream(stream), features, idx, c, n, mu, out);
  return launch_status();
}

GLDM_API int gldm_gather_features_forward(const float *features, const int32_t *idx, int b, int c, int n, int m,
                                          float *out, gldm_stream_t stream) {
  if (!features || !idx || !out || b <= 0 || c <= 0 || n <= 0 || m <= 0) return GLDM_ERR_INVALID_ARG;
  dim3 grid(ceil_div(m, 256), c, b);
  hipLaunchKernelGGL(gather_kernel, grid, dim3(256), 0, as_stream(stream), features, idx, c, n, m, out);
  return launch_status();
}

namespace {
template <int kThreads, int kPerThread, int kRule>
int launch_fps(const float *coords, int b, int n, int m, int32_t *out, hipStream_t s) {
  hipLaunchKernelGGL((fps_kernel<kThreads, kPerThread, kRule>), dim3(b), dim3(kThreads), (size_t)3 * n * sizeof(float),
                     s, coords, n, m, out);
  return launch_status();
}
template <int kPer, int kRule>
int launch_fps_wave(const float *coords, int b, int n, int m, int32_t *out, hipStream_t s) {
  hipLaunchKernelGGL((fps_wave_kernel<kPer, kRule>), dim3(b), dim3(64), (size_t)3 * n * sizeof(float), s, coords, n, m, out);
  return launch_status();
}
template <int kRule>
int dispatch_fps(const float *coords, int b, int n, int m, int32_t *out_idx, hipStream_t s) {
  // up to 1024 points: one wave per cloud, no barrier in the round (fps_wave_kernel)
  if (n <= 64) return launch_fps_wave<1, kRule>(coords, b, n, m, out_idx, s);
  if (n <= 128) return launch_fps_wave<2, kRule>(coords, b, n, m, out_idx, s);
  if (n <= 256) return launch_fps_wave<4, kRule>(coords, b, n, m, out_idx, s);
  if (n <= 512) return launch_fps_wave<8, kRule>(coords, b, n, m, out_idx, s);
  if (n <= 1024) return launch_fps_wave<16, kRule>(coords, b, n, m, out_idx, s);
  if (n <= 2048) return launch_fps<1024, 2, kRule>(coords, b, n, m, out_idx, s);
  if (n <= 4096) return launch_fps<1024, 4, kRule>(coords, b, n, m, out_idx, s);
  return launch_fps<1024, 8, kRule>(coords, b, n, m, out_idx, s);
}

// Raw-cloud front end: centre + scale a sensor cloud [n][3] (tools/inference.py:570-591 normalize_input) and
// gather rows of it by index (point-count regularisation).  The mean is accumulated in f64 in a fixed tree.
constexpr int kNcBlock = 256;
struct Vec3 { float v[3]; };
__global__ __launch_bounds__(kNcBlock) void normalize_cloud_kernel(const float *__restrict__ pc, int n, Vec3 shift,
                                                                   Vec3 scale, float *__restrict__ out,
                                                                   float *__restrict__ mean_out) {
  __shared__ double s_sum[3][kNcBlock / kWave];
  __shared__ float s_mean[3];
  const int b = blockIdx.x, tid = threadIdx.x;
  pc += (size_t)b * 3 * n;
  out += (size_t)b * 3 * n;
  double acc[3] = {0.0, 0.0, 0.0};
  for (int k = tid; k < n; k += kNcBlock)
#pragma unroll
    for (int a = 0; a < 3; ++a) acc[a] += (double)pc[3 * k + a];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    double v = acc[a];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, kWave);
    if ((tid & 63) == 0) s_sum[a][tid >> 6] = v;
  }
  __syncthreads();
  if (tid < 3) {
    double v = 0.0;
    for (int w = 0; w < kNcBlock / kWave; ++w) v += s_sum[tid][w];
    const float m = (float)(v / (double)n);
    s_mean[tid] = m;
    mean_out[(size_t)b * 3 + tid] = m;
  }
  __syncthreads();
  for (int i = tid; i < 3 * n; i += kNcBlock) {
    const int a = i % 3;
    const float sh = a == 0 ? shift.v[0] : (a == 1 ? shift.v[1] : shift.v[2]);
    const float sc = a == 0 ? scale.v[0] : (a == 1 ? scale.v[1] : scale.v[2]);
    out[i] = ((pc[i] - s_mean[a]) - sh) / sc;
  }
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const float *__restrict__ pc, const int32_t *__restrict__ idx,
                                                          int n, int m, float *__restrict__ out) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 3 * m) return;
  const int j = i / 3, a = i - 3 * j;
  out[(size_t)b * 3 * m + i] = pc[((size_t)b * n + idx[(size_t)b * m + j]) * 3 + a];
}
}  // namespace

GLDM_API int gldm_furthest_point_sampling(const float *coords, int b, int n, int m, int32_t *out_idx,
                                          gldm_stream_t stream) {
  if (!coords || !out_idx || b <= 0 || n <= 0 || m < 0) return GLDM_ERR_INVALID_ARG;
  if (m == 0) return GLDM_OK;
  if (n > 1024 * kFpsMaxPerThread) return GLDM_ERR_UNSUPPORTED;
  return dispatch_fps<0>(coords, b, n, m, out_idx, as_stream(stream));
}

GLDM_API int gldm_farthest_points_euclid(const float *points, int b, int n, int m, int32_t *out_idx,
                                         gldm_stream_t stream) {
  if (!points || !out_idx || b <= 0 || n <= 0 || m < 0 || m > n) return GLDM_ERR_INVALID_ARG;
  if (m == 0) return GLDM_OK;
  if (n > 1024 * kFpsMaxPerThread) return GLDM_ERR_UNSUPPORTED;
  return dispatch_fps<1>(points, b, n, m, out_idx, as_stream(stream));
}

GLDM_API int gldm_normalize_cloud(const float *pc, int b, int n, float shift_x, float shift_y, float shift_z,
                                  float scale_x, float scale_y, float scale_z, float *pc_out, float *mean_out,
                                  gldm_stream_t stream) {
  if (!pc || !pc_out || !mean_out || b <= 0 || n <= 0 || scale_x == 0.f || scale_y == 0.f || scale_z == 0.f)
    return GLDM_ERR_INVALID_ARG;
  hipLaunchKernelGGL(normalize_cloud_kernel, dim3(b), dim3(kNcBlock), 0, as_stream(stream), pc, n,
                     Vec3{{shift_x, shift_y, shift_z}}, Vec3{{scale_x, scale_y, scale_z}}, pc_out, mean_out);
  return launch_status();
}

GLDM_API int gldm_gather_points(const float *pc, const int32_t *idx, int b, int n, int m, float *out,
                                gldm_stream_t stream) {
  if (!pc || !idx || !out || b <= 0 || n <= 0 || m <= 0) return GLDM_ERR_INVALID_ARG;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(ceil_div(3 * m, 256), b), dim3(256), 0, as_stream(stream), pc, idx, n,
                     m, out);
  return launch_status();
}

GLDM_API int gldm_three_nn_interpolate_forward(const float *points, const float *centers,
                                               const float *center_features, int b, int c, int m, int n,
                                               float *out, int32_t *idx, float *wgt, gldm_stream_t stream) {
  if (!points || !centers || !center_features || !out || !idx || !wgt || b <= 0 || c <= 0 || m <= 0 || n <= 0)
    return GLDM_ERR_INVALID_ARG;
  if ((size_t)3 * m * sizeof(float) > 64 * 1024) return GLDM_ERR_UNSUPPORTED;
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(three_nn_kernel, dim3(ceil_div(n, 256), b), dim3(256), (size_t)3 * m * sizeof(float), s,
                     points, centers, n, m, idx, wgt);
  int st = launch_status();
  if (st != GLDM_OK) return st;
  hipLaunchKernelGGL(three_interp_kernel, dim3(ceil_div(n, 256), ceil_div(c, 16), b), dim3(256), 0, s,
                     center_features, idx, wgt, c, m, n, out);
  return launch_status();
}

GLDM_API int gldm_voxel_coords(const float *coords, int b, int n, int r, int normalize, float eps,
                               float *norm_coords, int32_t *vox_coords, gldm_stream_t stream) {
  if (!coords || !norm_coords || !vox_coords || b <= 0 || n <= 0 || r <= 0) return GLDM_ERR_INVALID_ARG;
  hipLaunchKernelGGL(voxel_coords_kernel, dim3(b), dim3(kVcBlock), 0, as_stream(stream), coords, n, r, normalize,
                     eps, norm_coords, vox_coords);
  return launch_status();
}

GLDM_API int gldm_avg_voxelize_forward(const float *features, const int32_t *vox_coords, int b, int c, int n,
                                       int r, float *out, int32_t *ind, int32_t *cnt, gldm_stream_t stream) {
  if (!features || !vox_coords || !out || !ind || !cnt || b <= 0 || c <= 0 || n <= 0 || r <= 0)
    return GLDM_ERR_INVALID_ARG;
  const int log_np2 = ilog2_ceil(n);
  const int np2 = 1 << log_np2;
  if (n > 8192 || r > 64 || (long long)r * r * r * np2 > (1ll << 32)) return GLDM_ERR_UNSUPPORTED;
  hipStream_t s = as_stream(stream);
  const size_t r3 = (size_t)r * r * r;
  // grids whose rows fit LDS beside the keys: assembled on chip and written dense (no memsets, no scattered stores)
  {
    const size_t key_bytes = (size_t)2 * np2 * sizeof(unsigned int) + 16, row_bytes = r3 * sizeof(float);
    const size_t room = (size_t)150 * 1024 - key_bytes;
    if (r3 % 4 == 0 && row_bytes <= room) {
      // rows per round: grid row + (when it fits) the row's features, both in LDS
      const size_t both = row_bytes + (size_t)n * sizeof(float);
      int stage_feat = both <= room ? 1 : 0;
      int g_rows = (int)(room / (stage_feat ? both : row_bytes));
      if (g_rows > 16) g_rows = 16;
      if (g_rows > c) g_rows = c;
      if (stage_feat && n % 4 == 0 && (size_t)g_rows * n <= (size_t)4 * 4 * kVoxBlock) stage_feat = 2;   // register prefetch form
      // A block sorts its cloud's keys once and then walks `chunk` channels in rounds of g_rows: a cloud is split over
      // several blocks only while the launch would otherwise leave compute units idle
      auto lds_of = [&](int rows) { return key_bytes + (size_t)rows * (stage_feat ? both : row_bytes); };
      const int rounds = ceil_div(c, g_rows);
      int parts = gldm_dev::cu_count() / b;   // (measured: more, smaller blocks per cloud re-sort more than they overlap)
      parts = parts < 1 ? 1 : (parts > rounds ? rounds : parts);
      const int chunk = ceil_div(rounds, parts) * g_rows;
      const size_t lds_bytes = lds_of(g_rows);
      struct VoxDenseTag { int site; };
      gldm_dev::allow_dynamic_lds<VoxDenseTag>(reinterpret_cast<const void *>(&avg_voxelize_dense_kernel), 160 * 1024);
      hipLaunchKernelGGL(avg_voxelize_dense_kernel, dim3(ceil_div(c, chunk), b), dim3(kVoxBlock), lds_bytes, s, features,
                         vox_coords, c, n, np2, log_np2, r, chunk, g_rows, stage_feat, out, ind, cnt);
      return launch_status();
    }
  }
  if (hipMemsetAsync(out, 0, (size_t)b * c * r3 * sizeof(float), s) != hipSuccess) return GLDM_ERR_LAUNCH;
  if (hipMemsetAsync(cnt, 0, (size_t)b * r3 * sizeof(int32_t), s) != hipSuccess) return GLDM_ERR_LAUNCH;
  dim3 grid(ceil_div(c, kVoxChannelsPerBlock), b);
  hipLaunchKernelGGL(avg_voxelize_kernel<1>, grid, dim3(kVoxBlock), (size_t)np2 * sizeof(unsigned int), s,
                     features, vox_coords, c, n, np2, log_np2, r, out, ind, cnt);
  return launch_status();
}

GLDM_API int gldm_trilinear_devoxelize_forward(const float *coords, const float *features, int b, int c, int n,
                                               int r, int is_training, float *out, int32_t *inds, float *wgts,
                                               gldm_stream_t stream) {
  if (!coords || !features || !out || b <= 0 || c <= 0 || n <= 0 || r <= 0) return GLDM_ERR_INVALID_ARG;
  if (is_training && (!inds || !wgts)) return GLDM_ERR_INVALID_ARG;
  dim3 grid(ceil_div(n, kDevBlock), ceil_div(c, kDevChannelsPerBlock), b);
  hipLaunchKernelGGL(trilinear_devoxelize_kernel, grid, dim3(kDevBlock), 0, as_stream(stream), coords, features,
                     c, n, r, is_training, out, inds, wgts);
  return launch_status();
}

GLDM_API int gldm_sa_group(const float *points, const float *centers, const float *features, int b, int c, int n,
                           int m, float radius, int u, float *out, int32_t *idx_out, gldm_stream_t stream) {
  if (!points || !centers || !out || b <= 0 || c < 0 || n <= 0 || m <= 0 || u <= 0) return GLDM_ERR_INVALID_ARG;
  if (c > 0 && !features) return GLDM_ERR_INVALID_ARG;
  const float r2 = radius * radius;
  // 16 centres per block: every thread owns one 16-byte slot of each channel row's 4 KiB run
  // (u = 64); measured best of {4, 8, 16, 32} on MI355X (tools/bench_point_ops.py).
  int cpb = 16;
#ifdef GLDM_DEBUG_KNOBS
  {
    const char *e = getenv("GLDM_SA_CPB");  // tuning knob (diagnostic builds only)
    if (e) cpb = atoi(e);
  }
#endif
  while (cpb > 1 && (size_t)cpb * u * sizeof(int32_t) > 32 * 1024) cpb >>= 1;
  const size_t fixed = (size_t)cpb * u * sizeof(int32_t) + (size_t)3 * cpb * sizeof(float);
  dim3 grid(ceil_div(m, cpb), b);
  const float *f = c > 0 ? features : nullptr;
  if (n <= kBqMaxLdsPoints) {
    hipLaunchKernelGGL(sa_group_kernel<true>, grid, dim3(kSaBlock), fixed + (size_t)3 * n * sizeof(float),
                       as_stream(stream), points, centers, f, c, n, m, r2, u, cpb, out, idx_out);
  } else {
    hipLaunchKernelGGL(sa_group_kernel<false>, grid, dim3(kSaBlock), fixed, as_stream(stream), points, centers, f,
                       c, n, m, r2, u, cpb, out, idx_out);
  }
  return launch_status();
}
