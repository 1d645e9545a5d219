// resnet1d.hip -- the fused 1-D ResNet engine of GraspLDM on gfx950:
//   * gldm_denoise : the whole reverse-diffusion loop (T steps of
//                    TimeConditionedResNet1D.forward + DDIM/DDPM update) in ONE
//                    launch; the latent never leaves the chip between steps;
//   * gldm_decode  : ConditionalGraspPoseDecoder.forward (in_layer, ResNet1D,
//                    tmrp / class_logits heads);
//   * gldm_r1d_cond_embed, gldm_pose_epilogue : the small ops either side.
//
// Mapping to CDNA4
//   A workgroup (8 waves) owns a tile of kCols = 64 activation columns =
//   64/L samples x L positions (L = 4: 16 latents, L = 16: 4 decoder rows) and
//   walks every layer with activations resident in LDS as [channel][column]
//   (XOR-swizzled so MFMA B-fragment reads are conflict free).  Every conv /
//   1x1 is a GEMM  W[Cout x taps*Cin] * im2col(X)[taps*Cin x 64]  on
//   v_mfma_f32_16x16x4_f32 (exact f32: the parity budget is 1e-4 on poses after
//   100 steps, and the reference's eps is dtype dependent, so no bf16 here).
//   A operands (weights, standardised and laid out in fragment order on the
//   host) stream from L2 straight into VGPRs as 16-byte coalesced loads, one
//   k-block ahead; B operands are LDS reads with the k=3 halo handled by a lane
//   mask instead of a materialised im2col.  GroupNorm / LayerNorm / softmax run
//   with lane = column so rows are read conflict-free and statistics reduce by
//   wave shuffles plus one small cross-wave exchange.
//   LinearAttention at n = L is reassociated:  out = V (K^T Q)  (an L x L
//   matrix per sample and head) instead of (V K^T) Q: 8x fewer FLOPs, same math.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "gldm.h"

#define GLDM_API extern "C" __attribute__((visibility("default")))

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef __attribute__((address_space(3))) float lds_f;   // explicit LDS pointers: 32-bit ds_* addressing
typedef __attribute__((address_space(3))) f32x4 lds_f4;

constexpr int kThreads = 512;
constexpr int kWaves = kThreads / 64;
constexpr int kCols = 64;               // activation columns per workgroup
constexpr int kHeads = 4, kDimHead = 32, kHidden = kHeads * kDimHead;  // LinearAttention defaults
constexpr int kMaxC = 256;

// LDS map (floats).  X: block input / residual stream, H: scratch.
constexpr int kBufX = 0;
constexpr int kBufH = kMaxC * kCols;            // 16384
constexpr int kBufY = 128 * kCols;              // 8192  (attention: LayerNorm output, later to_out output)
constexpr int kBufO = kBufH;                    // 16384 (attention output, 128 rows)
constexpr int kBufQKV = kBufO + kHidden * kCols;  // 24576 (two heads of q,k,v: 192 rows)
constexpr int kArena = kBufQKV + 192 * kCols;   // 36864
constexpr int kMiscLat = kArena;                // [64] current latent row
constexpr int kMiscEps = kMiscLat + kCols;      // [64]
constexpr int kMiscG = kMiscEps + kCols;        // [S][E] <= 256
constexpr int kMiscRed1 = kMiscG + 320;         // [8][64]
constexpr int kMiscRed2 = kMiscRed1 + kWaves * kCols;
constexpr int kLdsFloats = kMiscRed2 + kWaves * kCols;  // 38336 floats = 149.75 KiB
static_assert(kLdsFloats * 4 <= 160 * 1024, "LDS budget");

__device__ __forceinline__ int swz(int row, int col) { return row * kCols + (col ^ ((row & 1) << 4)); }

__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
__device__ __forceinline__ float silu(float x) { return x * __builtin_amdgcn_rcpf(1.0f + fast_exp(-x)); }

struct Ctx {
  const float *w;   // packed weights
  float *lds;
  int tid, wave, lane;
  int skip;
  long long *dbg;   // diagnostic stamps (GLDM_R1D_STAMP), null in production
  int *call;
  int nta;          // active 16-column tiles of this workgroup (4 = full, 1 = tail tile)
};

// ---------------------------------------------------------------- GEMM ----
// Every conv / 1x1 is  acc[mi][ni] += W[16(mt0+mi).., :] * im2col(src)[:, 16(nt0+ni)..]
// on v_mfma_f32_16x16x4_f32.  Packed weights: k = tap * Cin + ci, 16-deep k-blocks.
//
// Fast path (Cin % 16 == 0): per 16-channel block the B fragments are read from LDS ONCE
// (unconditional, batched reads) and the k=3 halo comes from DPP lane shifts inside each
// 16-lane row: the left/right taps of column n live in lanes n-1 / n+1 of the same row.
// A 16-column tile never straddles a sample (L divides 16), so row-boundary lanes are
// exactly the lanes whose tap falls outside the sample -> zero fill (bound_ctrl) for
// L = 16, an extra (col % L) mask for L = 4.
template <int L>
__device__ __forceinline__ float tap_left(float v, bool keep) {  // value of column n-1
  const float f = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111 /*row_shr:1*/, 0xf, 0xf, true));
  return (L >= 16 || keep) ? f : 0.f;
}
template <int L>
__device__ __forceinline__ float tap_right(float v, bool keep) {  // value of column n+1
  const float f = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x101 /*row_shl:1*/, 0xf, 0xf, true));
  return (L >= 16 || keep) ? f : 0.f;
}

__device__ __forceinline__ float f4_get(const f32x4 &v, int j) { return v[j]; }

template <int L, int TAPS, int MT, int NT>
__device__ __forceinline__ void gemm_fast(const Ctx &c, const float *__restrict__ wp, int cblocks, int mt0, int nt0,
                                          const float *src, f32x4 (&acc)[MT][NT]) {
  // k = 3 convs keep one accumulator set per tap: Y_t = W_t X on UNSHIFTED columns, and the
  // halo shift is applied once to the 16x16 result tiles (out[n] = Y0[n-1] + Y1[n] + Y2[n+1],
  // lane shifts inside the 16-lane rows of the C/D layout) -- the k-loop is loads + MFMA only.
  constexpr int PF = (MT * TAPS >= 6) ? 2 : 4;  // weight blocks in flight (register budget)
  const int col = c.lane & 15, kq = c.lane >> 4;
  const int kblocks = TAPS * cblocks;
  typedef const __attribute__((address_space(1))) f32x4 *gf4p;  // plain global_load (vmcnt only)
  gf4p wv = (gf4p)(reinterpret_cast<const f32x4 *>(wp) + c.lane);
  int boff[4][NT];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) boff[j][ni] = swz(4 * j + kq, 16 * (nt0 + ni) + col);
  f32x4 a[PF][TAPS][MT];
  float b[2][4][NT];
  f32x4 side[TAPS > 1 ? 2 : 1][MT][NT];  // tap 0 and tap 2 partial results (tap 1 goes to acc)
  if (TAPS == 3) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) side[t][mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  auto load_a = [&](int buf, int cb) {
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) a[buf][t][mi] = wv[((size_t)(mt0 + mi) * kblocks + t * cblocks + cb) * 64];
  };
  const lds_f *src3 = (const lds_f *)src;
  auto load_b = [&](int buf, int cb) {
    const lds_f *s = src3 + cb * 16 * kCols;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int ni = 0; ni < NT; ++ni) b[buf][j][ni] = s[boff[j][ni]];
  };
  auto compute = [&](int abuf, int bbuf) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (TAPS == 3) {
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
          for (int ni = 0; ni < NT; ++ni)
            side[0][mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[abuf][0][mi][j], b[bbuf][j][ni], side[0][mi][ni], 0, 0, 0);
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
          for (int ni = 0; ni < NT; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[abuf][TAPS > 1 ? 1 : 0][mi][j], b[bbuf][j][ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
          for (int ni = 0; ni < NT; ++ni)
            side[TAPS > 1 ? 1 : 0][mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[abuf][TAPS > 2 ? 2 : 0][mi][j], b[bbuf][j][ni], side[TAPS > 1 ? 1 : 0][mi][ni], 0, 0, 0);
      } else {
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
          for (int ni = 0; ni < NT; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[abuf][0][mi][j], b[bbuf][j][ni], acc[mi][ni], 0, 0, 0);
      }
    }
  };
  // loads are UNCONDITIONAL (block index clamped, a redundant re-load at the tail is harmless):
  // a branch around a load forces s_waitcnt 0 at the join and serialises the pipeline
  const int last = cblocks - 1;
#pragma unroll
  for (int u = 0; u < PF - 1; ++u) load_a(u, u < last ? u : last);
  load_b(0, 0);
  for (int cb0 = 0; cb0 < cblocks; cb0 += PF) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int cb = cb0 + u;
      if (cb < cblocks) {
        load_a((u + PF - 1) % PF, cb + PF - 1 < last ? cb + PF - 1 : last);
        load_b((u + 1) & 1, cb + 1 < last ? cb + 1 : last);
        __builtin_amdgcn_sched_barrier(0);
        compute(u, u & 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  if (TAPS == 3) {
    const bool keepL = (col & (L - 1)) != 0, keepR = (col & (L - 1)) != (L - 1);
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
      for (int ni = 0; ni < NT; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          acc[mi][ni][r] += tap_left<L>(side[0][mi][ni][r], keepL) + tap_right<L>(side[TAPS > 1 ? 1 : 0][mi][ni][r], keepR);
  }
}

// Generic path (Cin % 16 != 0: the 4-channel level of the latent denoiser): masked reads.
template <int L, int MT, int NT>
__device__ __forceinline__ void gemm_small(const Ctx &c, const float *__restrict__ wp, int kblocks, int mt0, int nt0,
                                           const float *src, int cin, int ktaps, f32x4 (&acc)[MT][NT]) {
  const int col = c.lane & 15, kq = c.lane >> 4;
  const f32x4 *wv = reinterpret_cast<const f32x4 *>(wp);
  int dk = 0, cib = 0;
  const int pad = ktaps == 3 ? 1 : 0;
  for (int kb = 0; kb < kblocks; ++kb) {
    f32x4 a[MT];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) a[mi] = wv[((size_t)(mt0 + mi) * kblocks + kb) * 64 + c.lane];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ci = cib + kq;
      const int shift = dk - pad;
      float b[NT];
#pragma unroll
      for (int ni = 0; ni < NT; ++ni) {
        const int n = 16 * (nt0 + ni) + col;
        const int p = (n & (L - 1)) + shift;
        const bool ok = (dk < ktaps) && (ci < cin) && (p >= 0) && (p < L);
        float v = src[swz(ok ? ci : 0, ok ? n + shift : 0)];
        asm volatile("" : "+v"(v));  // keep the LDS read unconditional (no branch + wait per element)
        b[ni] = ok ? v : 0.f;
      }
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4_get(a[mi], j), b[ni], acc[mi][ni], 0, 0, 0);
      cib += 4;
      if (cib >= cin) {
        cib = 0;
        ++dk;
      }
    }
  }
}

template <int MT, int NT>
__device__ __forceinline__ void store_tiles(const Ctx &c, const f32x4 (&acc)[MT][NT], int mt0, int nt0, float *dst,
                                            int cout, const float *__restrict__ bias, int act) {
  const int col = c.lane & 15, kq = c.lane >> 4;
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * (mt0 + mi) + 4 * kq + r;
      if (row < cout) {
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) {
          const float v = acc[mi][ni][r];
          ((lds_f *)dst)[swz(row, 16 * (nt0 + ni) + col)] = act ? fmaxf(v, 0.f) : v;
        }
      }
    }
  }
}

template <int L, int MT, int NT>
__device__ __forceinline__ void gemm_fixed(const Ctx &c, const float *wp, int mt0, int nt0, bool active,
                                           const float *src, int cin, int ktaps, float *dst, int cout,
                                           const float *bias, bool alias, int act) {
  // the bias is folded into the accumulator start value (its L2 load overlaps the GEMM)
  f32x4 acc[MT][NT];
  {
    const int kq = c.lane >> 4;
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
      f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
      if (bias) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * (mt0 + mi) + 4 * kq + r;
          bv[r] = bias[row < cout ? row : cout - 1];
        }
      }
#pragma unroll
      for (int ni = 0; ni < NT; ++ni) acc[mi][ni] = bv;
    }
  }
  if (active) {
    if ((cin & 15) == 0) {
      if (ktaps == 3) gemm_fast<L, 3, MT, NT>(c, wp, cin >> 4, mt0, nt0, src, acc);
      else gemm_fast<L, 1, MT, NT>(c, wp, cin >> 4, mt0, nt0, src, acc);
    } else {
      gemm_small<L, MT, NT>(c, wp, (ktaps * cin + 15) >> 4, mt0, nt0, src, cin, ktaps, acc);
    }
  }
  if (c.dbg && c.lane == 0) c.dbg[(*c.call * 8 + c.wave) * 4 + 1] = (long long)__builtin_readcyclecounter();
  if (alias) __syncthreads();
  if (active) store_tiles<MT, NT>(c, acc, mt0, nt0, dst, cout, bias, act);
}

// dst[cout][64] = W * im2col(src[cin][64]) + bias.  Ends with a barrier.
// alias: dst overlaps src -> all reads complete (barrier) before any store.
// Output widths are 16 x {1, 2, 4, 8, 12, 16} rows (validate() enforces it).
template <int L>
__device__ __noinline__ void conv_gemm(const Ctx &c, int w_off, int b_off, const float *src, int cin, int ktaps,
                                       float *dst, int cout, bool alias, int act = 0) {
  if (c.skip & 8) return;
  // arguments of a non-inlined device function arrive in VGPRs: make the wave-uniform ones
  // scalar again so loop bounds and address bases stay on the SALU
  w_off = __builtin_amdgcn_readfirstlane(w_off);
  b_off = __builtin_amdgcn_readfirstlane(b_off);
  cin = __builtin_amdgcn_readfirstlane(cin);
  ktaps = __builtin_amdgcn_readfirstlane(ktaps);
  cout = __builtin_amdgcn_readfirstlane(cout);
  act = __builtin_amdgcn_readfirstlane(act);
  {
    const int so = __builtin_amdgcn_readfirstlane((int)(src - c.lds)), dd = __builtin_amdgcn_readfirstlane((int)(dst - c.lds));
    src = c.lds + so;
    dst = c.lds + dd;
  }
  const float *wp = c.w + w_off;
  const float *bias = b_off >= 0 ? c.w + b_off : nullptr;
  const int mtiles = (cout + 15) >> 4;
  const int w = c.wave;
  if (c.dbg && c.lane == 0) {
    c.dbg[(*c.call * 8 + c.wave) * 4 + 0] = (long long)__builtin_readcyclecounter();
    c.dbg[(*c.call * 8 + c.wave) * 4 + 3] = ((long long)cout << 32) | ((long long)cin << 8) | ktaps;
  }
  if (c.nta == 1) {
    // tail workgroup: only columns 0..15 are live -> one n-tile, m-tiles spread over the waves
    if (mtiles == 16) gemm_fixed<L, 2, 1>(c, wp, 2 * w, 0, true, src, cin, ktaps, dst, cout, bias, alias, act);
    else if (mtiles == 12) gemm_fixed<L, 2, 1>(c, wp, 2 * (w < 6 ? w : 0), 0, w < 6, src, cin, ktaps, dst, cout, bias, alias, act);
    else gemm_fixed<L, 1, 1>(c, wp, w < mtiles ? w : 0, 0, w < mtiles, src, cin, ktaps, dst, cout, bias, alias, act);
  } else if (mtiles == 16) {
    gemm_fixed<L, 2, 4>(c, wp, 2 * w, 0, true, src, cin, ktaps, dst, cout, bias, alias, act);
  } else if (mtiles == 12) {
    gemm_fixed<L, 3, 2>(c, wp, 3 * (w & 3), 2 * (w >> 2), true, src, cin, ktaps, dst, cout, bias, alias, act);
  } else if (mtiles == 8) {
    gemm_fixed<L, 1, 4>(c, wp, w, 0, true, src, cin, ktaps, dst, cout, bias, alias, act);
  } else if (mtiles == 4) {
    gemm_fixed<L, 1, 2>(c, wp, w & 3, 2 * (w >> 2), true, src, cin, ktaps, dst, cout, bias, alias, act);
  } else if (mtiles == 2) {
    gemm_fixed<L, 1, 1>(c, wp, w & 1, w >> 1, true, src, cin, ktaps, dst, cout, bias, alias, act);
  } else {
    gemm_fixed<L, 1, 1>(c, wp, 0, w & 3, w < 4, src, cin, ktaps, dst, cout, bias, alias, act);
  }
  __syncthreads();
  if (c.dbg && c.lane == 0) c.dbg[(*c.call * 8 + c.wave) * 4 + 2] = (long long)__builtin_readcyclecounter();
  if (c.dbg) ++*c.call;
}

// ----------------------------------------------------------- GroupNorm ----
// In place on `buf` (or accumulated into `res` when given):
//   y = silu( GN(buf) * gamma + beta  [ * scale_sum + shift_sum ] )  [ + res ]
// lane = column; a group's channels are split over the waves that own it; all loads of a
// wave are issued up front (rows are register resident through both statistics passes).
template <int L, int RPW>
__device__ __forceinline__ void group_norm_rows(const Ctx &c, float *buf, float *res, int C, int cpg, int wpg, int awpg,
                                                int gamma_off, int beta_off, const float *ss, int S) {
  float *red1 = c.lds + kMiscRed1, *red2 = c.lds + kMiscRed2;
  const int n = c.lane, s = n / L;
  const int g = c.wave / wpg, sub = c.wave % wpg;
  const bool active = sub < awpg;
  const int row0 = g * cpg + (active ? sub : 0) * RPW;
  float v[RPW], sc[RPW], sh[RPW];
#pragma unroll
  for (int i = 0; i < RPW; ++i) v[i] = buf[swz(row0 + i, n)];
  if (ss) {
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      sc[i] = ss[(row0 + i) * S + s];
      sh[i] = ss[(C + row0 + i) * S + s];
    }
  }
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < RPW; ++i) sum += v[i];
#pragma unroll
  for (int off = 1; off < L; off <<= 1) sum += __shfl_xor(sum, off, 64);
  red1[c.wave * kCols + n] = active ? sum : 0.f;
  __syncthreads();
  const float cnt = (float)(cpg * L);
  float tot = 0.f;
  for (int q = 0; q < awpg; ++q) tot += red1[(g * wpg + q) * kCols + n];
  const float mean = tot / cnt;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < RPW; ++i) {
    const float d = v[i] - mean;
    sq += d * d;
  }
#pragma unroll
  for (int off = 1; off < L; off <<= 1) sq += __shfl_xor(sq, off, 64);
  red2[c.wave * kCols + n] = active ? sq : 0.f;
  __syncthreads();
  float vt = 0.f;
  for (int q = 0; q < awpg; ++q) vt += red2[(g * wpg + q) * kCols + n];
  const float rstd = 1.0f / sqrtf(vt / cnt + 1e-5f);
  const float *gamma = c.w + gamma_off, *beta = c.w + beta_off;
  if (active) {
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int row = row0 + i;
      float y = (v[i] - mean) * rstd * gamma[row] + beta[row];
      if (ss) y = y * sc[i] + sh[i];
      y = silu(y);
      const int a = swz(row, n);
      if (res) res[a] = res[a] + y;
      else buf[a] = y;
    }
  }
  __syncthreads();
}

template <int L>
__device__ __noinline__ void group_norm_pass(const Ctx &c, float *buf, float *res, int C, int groups, int gamma_off,
                                             int beta_off, const float *ss, int S) {
  if (c.skip & 1) return;
  const int cpg = C / groups;            // channels per group
  const int wpg = kWaves / groups;       // waves that share one group
  const int awpg = cpg < wpg ? cpg : wpg;
  const int rpw = cpg / awpg;            // rows per active wave (power of two <= 32)
  switch (rpw) {
    case 1: group_norm_rows<L, 1>(c, buf, res, C, cpg, wpg, awpg, gamma_off, beta_off, ss, S); break;
    case 2: group_norm_rows<L, 2>(c, buf, res, C, cpg, wpg, awpg, gamma_off, beta_off, ss, S); break;
    case 4: group_norm_rows<L, 4>(c, buf, res, C, cpg, wpg, awpg, gamma_off, beta_off, ss, S); break;
    case 8: group_norm_rows<L, 8>(c, buf, res, C, cpg, wpg, awpg, gamma_off, beta_off, ss, S); break;
    case 16: group_norm_rows<L, 16>(c, buf, res, C, cpg, wpg, awpg, gamma_off, beta_off, ss, S); break;
    default: group_norm_rows<L, 32>(c, buf, res, C, cpg, wpg, awpg, gamma_off, beta_off, ss, S); break;
  }
}

// ----------------------------------------------------------- LayerNorm ----
// dst = LN_channels(src) * g  (or res += LN(src) * g when res != null)
template <int RPW>
__device__ __forceinline__ void layer_norm_rows(const Ctx &c, const float *src, float *dst, float *res, int C,
                                                int g_off) {
  float *red1 = c.lds + kMiscRed1, *red2 = c.lds + kMiscRed2;
  const int n = c.lane;
  float v[RPW];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < RPW; ++i) {
    const int row = c.wave + i * kWaves;
    const float x = src[swz(row < C ? row : 0, n)];
    v[i] = row < C ? x : 0.f;
    sum += v[i];
  }
  red1[c.wave * kCols + n] = sum;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int q = 0; q < kWaves; ++q) tot += red1[q * kCols + n];
  const float mean = tot / (float)C;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < RPW; ++i) {
    const int row = c.wave + i * kWaves;
    const float d = row < C ? v[i] - mean : 0.f;
    sq += d * d;
  }
  red2[c.wave * kCols + n] = sq;
  __syncthreads();
  float vt = 0.f;
#pragma unroll
  for (int q = 0; q < kWaves; ++q) vt += red2[q * kCols + n];
  const float rstd = 1.0f / sqrtf(vt / (float)C + 1e-5f);
  const float *g = c.w + g_off;
#pragma unroll
  for (int i = 0; i < RPW; ++i) {
    const int row = c.wave + i * kWaves;
    if (row < C) {
      const float y = (v[i] - mean) * rstd * g[row];
      const int a = swz(row, n);
      if (res) res[a] = res[a] + y;
      else dst[a] = y;
    }
  }
  __syncthreads();
}

__device__ __noinline__ void layer_norm_pass(const Ctx &c, const float *src, float *dst, float *res, int C, int g_off) {
  if (c.skip & 2) return;
  const int rpw = (C + kWaves - 1) / kWaves;
  if (rpw <= 1) layer_norm_rows<1>(c, src, dst, res, C, g_off);
  else if (rpw <= 2) layer_norm_rows<2>(c, src, dst, res, C, g_off);
  else if (rpw <= 4) layer_norm_rows<4>(c, src, dst, res, C, g_off);
  else if (rpw <= 8) layer_norm_rows<8>(c, src, dst, res, C, g_off);
  else layer_norm_rows<16>(c, src, dst, res, C, g_off);
}

// -------------------------------------------------- linear attention -------
// qkv: [192][64] = q(2 heads x 32) | k | v for one head pair; writes 64 rows of o.
// Wave w: head hl = w >> 2 of the pair, output rows e in [8 (w & 3), +8).
template <int L>
__device__ __noinline__ void attention_pair(const Ctx &c, float *qkv, float *o_rows) {
  if (c.skip & 4) return;
  const int n = c.lane, sbase = n & ~(L - 1);
  const int hl = c.wave >> 2, part = c.wave & 3, e0 = 8 * part;
  const int qrow0 = hl * kDimHead, krow0 = 64 + hl * kDimHead, vrow0 = 128 + hl * kDimHead;
  if constexpr (L == 4) {
    // The 32 head channels are split over the 4 waves of a head (8 each); softmax statistics
    // and the 4x4 matrix A = softmax_n(k)^T softmax_d(q) are combined through LDS.
    float *red1 = c.lds + kMiscRed1, *red2 = c.lds + kMiscRed2;
    float *part_a = qkv;  // q rows are dead after phase 1: [8 waves][4][64] partial A
    float q[8];
    float qmax = -3.0e38f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      q[i] = qkv[swz(qrow0 + e0 + i, n)];
      qmax = fmaxf(qmax, q[i]);
    }
    float4 kv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) kv[i] = *reinterpret_cast<const float4 *>(&qkv[swz(krow0 + e0 + i, sbase)]);
    red1[c.wave * kCols + n] = qmax;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < 4; ++w) qmax = fmaxf(qmax, red1[(hl * 4 + w) * kCols + n]);
    float qsum = 0.f, a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float e = fast_exp(q[i] - qmax);
      qsum += e;
      const float km = fmaxf(fmaxf(kv[i].x, kv[i].y), fmaxf(kv[i].z, kv[i].w));
      const float k0 = fast_exp(kv[i].x - km), k1 = fast_exp(kv[i].y - km), k2 = fast_exp(kv[i].z - km),
                  k3 = fast_exp(kv[i].w - km);
      const float f = e * __builtin_amdgcn_rcpf(k0 + k1 + k2 + k3);
      a0 += k0 * f; a1 += k1 * f; a2 += k2 * f; a3 += k3 * f;
    }
    red2[c.wave * kCols + n] = qsum;
    part_a[(c.wave * 4 + 0) * kCols + n] = a0;
    part_a[(c.wave * 4 + 1) * kCols + n] = a1;
    part_a[(c.wave * 4 + 2) * kCols + n] = a2;
    part_a[(c.wave * 4 + 3) * kCols + n] = a3;
    float4 vv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) vv[i] = *reinterpret_cast<const float4 *>(&qkv[swz(vrow0 + e0 + i, sbase)]);
    __syncthreads();
    qsum = 0.f; a0 = a1 = a2 = a3 = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int ww = hl * 4 + w;
      qsum += red2[ww * kCols + n];
      a0 += part_a[(ww * 4 + 0) * kCols + n];
      a1 += part_a[(ww * 4 + 1) * kCols + n];
      a2 += part_a[(ww * 4 + 2) * kCols + n];
      a3 += part_a[(ww * 4 + 3) * kCols + n];
    }
    const float sc = 0.17677669529663687f * __builtin_amdgcn_rcpf(qsum);  // dim_head ** -0.5 / sum
    a0 *= sc; a1 *= sc; a2 *= sc; a3 *= sc;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      o_rows[swz(hl * kDimHead + e0 + i, n)] = vv[i].x * a0 + vv[i].y * a1 + vv[i].z * a2 + vv[i].w * a3;
    __syncthreads();
  } else {
    // L = 16 (pose decoder, once per grasp): every wave recomputes A for its head
    float qmax = -3.0e38f;
    for (int d = 0; d < kDimHead; ++d) qmax = fmaxf(qmax, qkv[swz(qrow0 + d, n)]);
    float qsum = 0.f;
    for (int d = 0; d < kDimHead; ++d) qsum += fast_exp(qkv[swz(qrow0 + d, n)] - qmax);
    const float qscale = 0.17677669529663687f / qsum;
    float A[L];
#pragma unroll
    for (int i = 0; i < L; ++i) A[i] = 0.f;
    for (int d = 0; d < kDimHead; ++d) {
      const float qd = fast_exp(qkv[swz(qrow0 + d, n)] - qmax) * qscale;
      float kv[L];
      float kmax = -3.0e38f;
#pragma unroll
      for (int i = 0; i < L; ++i) {
        kv[i] = qkv[swz(krow0 + d, sbase + i)];
        kmax = fmaxf(kmax, kv[i]);
      }
      float ksum = 0.f;
#pragma unroll
      for (int i = 0; i < L; ++i) {
        kv[i] = fast_exp(kv[i] - kmax);
        ksum += kv[i];
      }
      const float f = qd / ksum;
#pragma unroll
      for (int i = 0; i < L; ++i) A[i] += kv[i] * f;
    }
    for (int e = e0; e < e0 + 8; ++e) {
      float acc = 0.f;
#pragma unroll
      for (int i = 0; i < L; ++i) acc += qkv[swz(vrow0 + e, sbase + i)] * A[i];
      o_rows[swz(hl * kDimHead + e, n)] = acc;
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------- the network ----
struct RunArgs {
  gldm_r1d_desc d;
  const float *weights;
  const float *temb;      // [T][E] or null
  const float *cemb;      // [n_cond][R][E]
  int samples_per_cond;
  const float *x_in;      // denoise: [n][L]; decode: z_h [n][D]
  int n_samples;
  const int32_t *timesteps;
  const int32_t *sample_t;
  int n_steps;
  int sched_kind, clip_sample;
  const float *sched_coef;
  const float *step_noise;
  float *out0;            // denoise: x_out [n][L]; decode: tmrp [n][6]
  float *out1;            // decode: logit [n]
  float *ws;              // [tiles][ss_rows][S]
  int skip;               // diagnostic phase-skip mask (GLDM_R1D_SKIP env; 0 in production)
  long long *dbg;         // diagnostic per-GEMM stamps (GLDM_R1D_STAMP env)
  int full_tiles, tail_tiles, tail_samples;
};

// scale/shift rows of one ResnetBlock: ss[row][s] = Wmlp[row,:] . G[s,:] + (R b + R on scale rows),
// rows 0..C-1 = sum_r (scale_r + 1), rows C..2C-1 = sum_r shift_r.  One 16x16 MFMA tile per 16
// rows (N = the tile's samples), written to this workgroup's L2-resident scratch [2C][S].
template <int L>
__device__ __noinline__ void scale_shift_table(const Ctx &c, const gldm_r1d_resblock &rb, int C, int E, float *ss,
                                               int S) {
  if (c.skip & 16) return;
  const float *G = c.lds + kMiscG;
  const int col = c.lane & 15, kq = c.lane >> 4;
  const int sidx = col < S ? col : S - 1;
  const int mtiles = (2 * C + 15) >> 4, kblocks = (E + 15) >> 4;
  const f32x4 *wv = reinterpret_cast<const f32x4 *>(c.w + rb.ss_w);
  const float *bias = c.w + rb.ss_b;
  for (int mt = c.wave; mt < mtiles; mt += kWaves) {
    f32x4 acc;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * mt + 4 * kq + r;
      acc[r] = bias[row < 2 * C ? row : 2 * C - 1];
    }
    for (int kb = 0; kb < kblocks; ++kb) {
      const f32x4 a = wv[((size_t)mt * kblocks + kb) * 64 + c.lane];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int e = 16 * kb + 4 * j + kq;
        const float b = e < E ? G[sidx * E + e] : 0.f;
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b, acc, 0, 0, 0);
      }
    }
    if (col < S) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * mt + 4 * kq + r;
        if (row < 2 * C) ss[row * S + col] = acc[r];
      }
    }
  }
}

template <int L>
__device__ void resnet_block(const Ctx &c, const gldm_r1d_desc &d, const gldm_r1d_resblock &rb, int C,
                             float *ss_tile, int S) {
  float *X = c.lds + kBufX, *H = c.lds + kBufH;
  scale_shift_table<L>(c, rb, C, d.emb_dim, ss_tile, S);  // published by the barrier that ends conv_gemm
  conv_gemm<L>(c, rb.c1_w, rb.c1_b, X, C, 3, H, C, false);
  group_norm_pass<L>(c, H, nullptr, C, d.groups, rb.n1_w, rb.n1_b, (c.skip & 16) ? nullptr : ss_tile, S);
  conv_gemm<L>(c, rb.c2_w, rb.c2_b, H, C, 3, H, C, true);
  group_norm_pass<L>(c, H, X, C, d.groups, rb.n2_w, rb.n2_b, nullptr, S);
}

template <int L>
__device__ void attention_block(const Ctx &c, const gldm_r1d_level &lv, int C) {
  float *X = c.lds + kBufX, *Y = c.lds + kBufY, *O = c.lds + kBufO, *QKV = c.lds + kBufQKV;
  layer_norm_pass(c, X, Y, nullptr, C, lv.ln_g);
  for (int p = 0; p < 2; ++p) {
    conv_gemm<L>(c, lv.qkv_w[p], -1, Y, C, 1, QKV, 192, false);
    attention_pair<L>(c, QKV, O + p * 64 * kCols);
  }
  conv_gemm<L>(c, lv.out_w, lv.out_b, O, kHidden, 1, Y, C, false);
  layer_norm_pass(c, Y, nullptr, X, C, lv.ln2_g);
}

#pragma clang fp contract(off)
__device__ __forceinline__ float scheduler_update(int kind, int clip, const float *cf, float x, float eps, float noise) {
  float x0 = (x - cf[0] * eps) / cf[1];
  if (clip) x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
  if (kind == GLDM_SCHED_DDIM) {
    const float a = cf[2] * x0;
    const float b = cf[3] * eps;
    return a + b;
  }
  const float a = cf[4] * x0;
  const float b = cf[5] * x;
  float prev = a + b;
  if (cf[7] != 0.f) {
    const float nz = cf[6] * noise;
    prev = prev + nz;
  }
  return prev;
}
#pragma clang fp contract(fast)

template <int L>
__global__ __launch_bounds__(kThreads, 2) void r1d_kernel(const RunArgs a) {
  extern __shared__ float lds[];
  const gldm_r1d_desc &d = a.d;
  constexpr int S = kCols / L;
  int call_idx = 0;
  Ctx c{a.weights, lds, (int)threadIdx.x, (int)threadIdx.x >> 6, (int)threadIdx.x & 63, a.skip,
        blockIdx.x == 0 ? a.dbg : nullptr, &call_idx, 4};
  // Tiles: the first `full_tiles` workgroups own S samples each; the remainder of the batch is
  // spread over `tail_tiles` workgroups of `tail_samples` (one 16-column n-tile when it fits)
  // so that the last wave of workgroups is not 64-column tiles on a quarter of the CUs.
  const int tile = blockIdx.x;
  const bool is_tail = tile >= a.full_tiles;
  const int samp0 = is_tail ? a.full_tiles * S + (tile - a.full_tiles) * a.tail_samples : tile * S;
  c.nta = (is_tail && a.tail_samples * L <= 16) ? 1 : 4;
  const int nsamp = is_tail ? a.tail_samples : S;  // samples this workgroup owns
  const int E = d.emb_dim, R = d.cond_rows;
  float *lat = lds + kMiscLat, *epsr = lds + kMiscEps, *G = lds + kMiscG;
  float *X = lds + kBufX;
  float *ss_tile = a.ws + (size_t)tile * d.ss_rows * S;  // [2 Cmax][S], reused by every ResnetBlock
  const bool has_in = d.latent_dim > 0, has_head = d.n_head > 0;

  for (int i = c.tid; i < kLdsFloats; i += kThreads) lds[i] = 0.f;  // dead columns must stay finite
  __syncthreads();
  // ---- latent row for this tile
  if (c.tid < kCols) {
    const int s = c.tid / L, l = c.tid % L;
    const int gi = min(samp0 + min(s, nsamp - 1), a.n_samples - 1);
    float v;
    if (has_in) {
      const float *wi = a.weights + d.in_w + l * d.latent_dim;
      v = a.weights[d.in_b + l];
      for (int q = 0; q < d.latent_dim; ++q) v += wi[q] * a.x_in[(size_t)gi * d.latent_dim + q];
    } else {
      v = a.x_in[(size_t)gi * L + l];
    }
    lat[c.tid] = v;
  }
  __syncthreads();

  for (int step = 0; step < a.n_steps; ++step) {
    call_idx = 0;
    // ---- G[s][e] = sum_r silu(temb[t][e] + cemb[cond][r][e])
    for (int i = c.tid; i < S * E; i += kThreads) {
      const int s = i / E, e = i - s * E;
      const int gi = min(samp0 + min(s, nsamp - 1), a.n_samples - 1);
      const float *ce = a.cemb + ((size_t)(gi / a.samples_per_cond) * R) * E + e;
      float te = 0.f;
      if (a.temb) {
        const int t = a.sample_t ? a.sample_t[gi] : a.timesteps[step];
        te = a.temb[(size_t)t * E + e];
      }
      float g = 0.f;
      for (int r = 0; r < R; ++r) g += silu(te + ce[r * E]);
      G[s * E + e] = g;
    }
    __syncthreads();
    // ---- init conv (k = 7, one input channel)
    const int C0 = d.dims[0];
    for (int i = c.tid; i < C0 * kCols; i += kThreads) {
      const int ch = i / kCols, n = i - ch * kCols;
      const int l = n & (L - 1), base = n - l;
      float acc = a.weights[d.init_b + ch];
      const float *wk = a.weights + d.init_w + ch * 7;
#pragma unroll
      for (int q = 0; q < 7; ++q) {
        const int p = l + q - 3;
        if (p >= 0 && p < L) acc += wk[q] * lat[base + p];
      }
      X[swz(ch, n)] = acc;
    }
    __syncthreads();

    int rbi = 0;
    for (int lv = 0; lv < d.n_levels; ++lv) {
      const int C = d.dims[lv], Cn = d.dims[lv + 1];
      resnet_block<L>(c, d, d.rb[rbi++], C, ss_tile, S);
      resnet_block<L>(c, d, d.rb[rbi++], C, ss_tile, S);
      attention_block<L>(c, d.lv[lv], C);
      conv_gemm<L>(c, d.lv[lv].down_w, d.lv[lv].down_b, X, C, 3, X, Cn, true);
    }
    const int CF = d.dims[d.n_levels];
    resnet_block<L>(c, d, d.rb[rbi], CF, ss_tile, S);

    // ---- final 1x1 conv to one channel: eps[n] = b + sum_c w[c] X[c][n]
    {
      float *red1 = lds + kMiscRed1;
      float part = 0.f;
      for (int row = c.wave; row < CF; row += kWaves) part += a.weights[d.final_w + row] * X[swz(row, c.lane)];
      red1[c.wave * kCols + c.lane] = part;
      __syncthreads();
      if (c.tid < kCols) {
        float e = a.weights[d.final_b];
#pragma unroll
        for (int q = 0; q < kWaves; ++q) e += red1[q * kCols + c.tid];
        epsr[c.tid] = e;
        if (a.sched_kind != GLDM_SCHED_NONE) {
          const int s = c.tid / L, l = c.tid % L;
          const int gi = min(samp0 + min(s, nsamp - 1), a.n_samples - 1);
          const float *cf = a.sched_coef + (size_t)step * GLDM_SCHED_COEF_STRIDE;
          float nz = 0.f;
          if (a.sched_kind == GLDM_SCHED_DDPM && cf[7] != 0.f && a.step_noise)
            nz = a.step_noise[((size_t)step * a.n_samples + gi) * L + l];
          lat[c.tid] = scheduler_update(a.sched_kind, a.clip_sample, cf, lat[c.tid], e, nz);
        }
      }
      __syncthreads();
    }
  }

  // ---- outputs
  if (!has_head) {
    if (c.tid < kCols) {
      const int s = c.tid / L, l = c.tid % L;
      const int gi = samp0 + s;
      if (s < nsamp && gi < a.n_samples) a.out0[(size_t)gi * L + l] = a.sched_kind == GLDM_SCHED_NONE ? epsr[c.tid] : lat[c.tid];
    }
  } else {
    // heads: rows 0..5 tmrp, row 6 class logit; input = the L-vector of each sample
    const int nh = d.n_head;
    if (c.tid < S * nh) {
      const int s = c.tid / nh, r = c.tid - s * nh;
      const int gi = samp0 + s;
      if (s < nsamp && gi < a.n_samples) {
        const float *wr = a.weights + d.head_w + r * L;
        float acc = a.weights[d.head_b + r];
        for (int l = 0; l < L; ++l) acc += wr[l] * epsr[s * L + l];
        if (r < 6) a.out0[(size_t)gi * 6 + r] = acc;
        else a.out1[gi] = acc;
      }
    }
  }
}

__global__ void cond_embed_kernel(const float *__restrict__ z, const float *__restrict__ w,
                                  const float *__restrict__ b, int rows_total, int dc, int e, float *__restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows_total * e) return;
  const int row = i / e, col = i - row * e;
  const float *zr = z + (size_t)row * dc, *wr = w + (size_t)col * dc;
  float acc = b[col];
  for (int q = 0; q < dc; ++q) acc += wr[q] * zr[q];
  out[i] = acc / (1.0f + expf(-acc));
}

#pragma clang fp contract(off)
__global__ void pose_epilogue_kernel(const float *__restrict__ tmrp, const float *__restrict__ logit,
                                     const float *__restrict__ mean, const float *__restrict__ stdv, int n, int gpc,
                                     float *__restrict__ H, float *__restrict__ un, float *__restrict__ conf) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int cl = i / gpc;
  float v[6];
  for (int q = 0; q < 6; ++q) {
    v[q] = tmrp[(size_t)i * 6 + q] * stdv[cl * 6 + q] + mean[cl * 6 + q];
    un[(size_t)i * 6 + q] = v[q];
  }
  // rotations.py:218-252 (mrp -> quat) and :171-215 (quat -> rotmat, SciPy convention)
  const float magsq = v[3] * v[3] + v[4] * v[4] + v[5] * v[5];
  const float den = 1.0f + magsq;
  const float x = (2.0f * v[3]) / den, y = (2.0f * v[4]) / den, z = (2.0f * v[5]) / den;
  const float w = (1.0f - magsq) / den;
  const float x2 = x * x, y2 = y * y, z2 = z * z, w2 = w * w;
  const float xy = x * y, zw = z * w, xz = x * z, yw = y * w, yz = y * z, xw = x * w;
  float *h = H + (size_t)i * 16;
  h[0] = x2 - y2 - z2 + w2;  h[1] = 2.0f * (xy - zw);     h[2] = 2.0f * (xz + yw);      h[3] = v[0];
  h[4] = 2.0f * (xy + zw);   h[5] = -x2 + y2 - z2 + w2;   h[6] = 2.0f * (yz - xw);      h[7] = v[1];
  h[8] = 2.0f * (xz - yw);   h[9] = 2.0f * (yz + xw);     h[10] = -x2 - y2 + z2 + w2;   h[11] = v[2];
  h[12] = 0.f; h[13] = 0.f; h[14] = 0.f; h[15] = 1.0f;
  if (conf) conf[i] = 1.0f / (1.0f + expf(-logit[i]));
}
#pragma clang fp contract(fast)


// ====================================================================== fused SA ==
// PointNetSAModule core (ext/pvcnn/modules/pointnet.py:100-111 without the FPS):
//   grouped = cat(p[idx] - centre, f[idx])          (BallQuery.forward)
//   out[b, :, j] = max_k  SharedMLP2d(grouped)[b, :, j, k]
// One workgroup = one 64-column tile = 64/U centres x U neighbours.  The grouped tensor
// ([B, 3+C, M, U], 4.3 MB per cloud at SSG-SA2) never exists in HBM: the neighbour tile is
// gathered straight into LDS, the MLP layers (BatchNorm folded, ReLU) run on the same
// f32-MFMA GEMM core as the ResNet1D engine with weights streamed from L2, and the max
// over the U neighbours is taken on chip.  HBM traffic: 12N + 4CN + 12M + 4MU (idx)
// in, 4 Cout M out per cloud.
struct SaArgs {
  const float *points, *centers, *feat;
  const int32_t *idx;
  const float *weights;
  float *out;
  int c, n, m, u, n_layers;
  int cin_pad[4], cout[4], w_off[4], b_off[4];
};

__global__ __launch_bounds__(kThreads, 2) void sa_mlp_kernel(const SaArgs a) {
  extern __shared__ float lds[];
  Ctx c{a.weights, lds, (int)threadIdx.x, (int)threadIdx.x >> 6, (int)threadIdx.x & 63, 0, nullptr, nullptr, 4};
  const int b = blockIdx.y, tile = blockIdx.x;
  const int cpt = kCols / a.u;                 // centres per tile
  const int j0 = tile * cpt;
  const float *pts = a.points + (size_t)b * 3 * a.n;
  const float *ctr = a.centers + (size_t)b * 3 * a.m;
  const float *feat = a.feat ? a.feat + (size_t)b * a.c * a.n : nullptr;
  const int32_t *idx = a.idx + ((size_t)b * a.m + j0) * a.u;
  float *X = lds + kBufX, *H = lds + kBufH;
  // ---- gather the neighbour tile: rows 0..2 relative coords, 3..3+C features, zero pad
  {
    const int col = c.lane, jj = col / a.u;
    const bool live = j0 + jj < a.m;
    const int id = live ? idx[col] : 0;
    const int rows = a.cin_pad[0];
    for (int r = c.wave; r < rows; r += kWaves) {
      float v = 0.f;
      if (live) {
        if (r < 3) v = pts[r * a.n + id] - ctr[r * a.m + j0 + jj];
        else if (r < 3 + a.c) v = feat[(size_t)(r - 3) * a.n + id];
      }
      X[swz(r, col)] = v;
    }
  }
  __syncthreads();
  // ---- grouped MLP (1x1 convs + folded BN + ReLU), ping-pong X <-> H
  float *src = X, *dst = H;
  for (int l = 0; l < a.n_layers; ++l) {
    conv_gemm<16>(c, a.w_off[l], a.b_off[l], src, a.cin_pad[l], 1, dst, a.cout[l], false, 1);
    float *t = src; src = dst; dst = t;
  }
  // ---- max over the U neighbours of each centre
  const int cout = a.cout[a.n_layers - 1];
  float *out = a.out + (size_t)b * cout * a.m;
  for (int i = c.tid; i < cout * cpt; i += kThreads) {
    const int row = i / cpt, jj = i - row * cpt;
    if (j0 + jj >= a.m) continue;
    float mx = -3.0e38f;
    for (int k = 0; k < a.u; ++k) mx = fmaxf(mx, src[swz(row, jj * a.u + k)]);
    out[(size_t)row * a.m + j0 + jj] = mx;
  }
}

bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

int validate(const gldm_r1d_desc *d) {
  if (!d) return GLDM_ERR_INVALID_ARG;
  if (d->seq_len != 4 && d->seq_len != 16) return GLDM_ERR_UNSUPPORTED;
  if (d->n_levels < 1 || d->n_levels > GLDM_R1D_MAX_LEVELS) return GLDM_ERR_UNSUPPORTED;
  const int S = kCols / d->seq_len;
  if (d->emb_dim <= 0 || S * d->emb_dim > 320) return GLDM_ERR_UNSUPPORTED;
  if (d->groups <= 0 || kWaves % d->groups != 0) return GLDM_ERR_UNSUPPORTED;
  for (int i = 0; i <= d->n_levels; ++i) {
    const int C = d->dims[i];
    if (C < d->groups || C > kMaxC || C < 4 || !pow2(C) || C % d->groups != 0) return GLDM_ERR_UNSUPPORTED;
    const int cpg = C / d->groups, wpg = kWaves / d->groups;
    const int awpg = cpg < wpg ? cpg : wpg;
    if (cpg % awpg != 0 || cpg / awpg > 32) return GLDM_ERR_UNSUPPORTED;
    if (i < d->n_levels && C > 128) return GLDM_ERR_UNSUPPORTED;  // attention levels keep 4 regions in LDS
  }
  return GLDM_OK;
}

struct Tiling { int full_tiles, tail_tiles, tail_samples; };

Tiling make_tiling(int n_samples, int L) {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  const int S = kCols / L, unit = 16 / L;  // samples per 64-column tile / per 16-column n-tile
  const int tiles = (n_samples + S - 1) / S;
  Tiling t{tiles, 0, 0};
  if (tiles <= cus || tiles % cus == 0) return t;
  const int full = (tiles / cus) * cus;
  const int left = n_samples - full * S;          // < cus * S samples for the last wave of workgroups
  const int units = (left + unit - 1) / unit;
  const int upt = (units + cus - 1) / cus;        // n-tiles per tail workgroup (1..4)
  if (upt >= 4) return t;
  t.full_tiles = full;
  t.tail_samples = upt * unit;
  t.tail_tiles = (left + t.tail_samples - 1) / t.tail_samples;
  return t;
}

int launch_r1d(const RunArgs &a_in, hipStream_t s) {
  const int L = a_in.d.seq_len;
  const Tiling tl = make_tiling(a_in.n_samples, L);
  const int tiles = tl.full_tiles + tl.tail_tiles;
  const size_t lds_bytes = (size_t)kLdsFloats * sizeof(float);
  RunArgs a = a_in;
  a.full_tiles = tl.full_tiles; a.tail_tiles = tl.tail_tiles; a.tail_samples = tl.tail_samples;
  {
    const char *e = getenv("GLDM_R1D_SKIP");
    a.skip = e ? atoi(e) : 0;
  }
  const bool stamp = getenv("GLDM_R1D_STAMP") != nullptr;  // diagnostic: blocks, copies and prints
  static long long *dbg = nullptr;
  if (stamp && !dbg) (void)hipMalloc(&dbg, 64 * 8 * 4 * sizeof(long long));
  a.dbg = stamp ? dbg : nullptr;
  if (stamp) (void)hipMemset(dbg, 0, 64 * 8 * 4 * sizeof(long long));
  if (L == 4) {
    static bool attr4 = false;
    if (!attr4) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&r1d_kernel<4>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
      attr4 = true;
    }
    hipLaunchKernelGGL(r1d_kernel<4>, dim3(tiles), dim3(kThreads), lds_bytes, s, a);
  } else {
    static bool attr16 = false;
    if (!attr16) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&r1d_kernel<16>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
      attr16 = true;
    }
    hipLaunchKernelGGL(r1d_kernel<16>, dim3(tiles), dim3(kThreads), lds_bytes, s, a);
  }
  if (stamp) {
    static long long host[64 * 8 * 4];
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(host, dbg, sizeof(host), hipMemcpyDeviceToHost);
    for (int cidx = 0; cidx < 64 && host[cidx * 32 + 3]; ++cidx) {
      const long long meta = host[cidx * 32 + 3];
      long long t0 = host[cidx * 32], lo = 1ll << 62, hi = 0, end = 0;
      for (int w = 0; w < 8; ++w) {
        t0 = host[(cidx * 8 + w) * 4] < t0 ? host[(cidx * 8 + w) * 4] : t0;
        const long long d = host[(cidx * 8 + w) * 4 + 1];
        if (d) { lo = d < lo ? d : lo; hi = d > hi ? d : hi; }
        end = host[(cidx * 8 + w) * 4 + 2] > end ? host[(cidx * 8 + w) * 4 + 2] : end;
      }
      printf("gemm %2d cout=%3d cin=%3d taps=%d  first wave done %6lld  last wave done %6lld  end %6lld\n", cidx,
             (int)(meta >> 32), (int)((meta >> 8) & 0xffffff), (int)(meta & 0xff), lo - t0, hi - t0, end - t0);
    }
  }
  return hipGetLastError() == hipSuccess ? GLDM_OK : GLDM_ERR_LAUNCH;
}

}  // namespace

GLDM_API int gldm_r1d_cond_embed(const float *z_cond, const float *w, const float *b, int n_cond, int rows, int dc,
                                 int e, float *cemb, gldm_stream_t stream) {
  if (!z_cond || !w || !b || !cemb || n_cond <= 0 || rows <= 0 || dc <= 0 || e <= 0) return GLDM_ERR_INVALID_ARG;
  const int total = n_cond * rows * e;
  hipLaunchKernelGGL(cond_embed_kernel, dim3((total + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     z_cond, w, b, n_cond * rows, dc, e, cemb);
  return hipGetLastError() == hipSuccess ? GLDM_OK : GLDM_ERR_LAUNCH;
}

GLDM_API long long gldm_r1d_workspace_bytes(const gldm_r1d_desc *desc, int n_samples) {
  if (validate(desc) != GLDM_OK || n_samples <= 0) return -1;
  const int S = kCols / desc->seq_len;
  const Tiling tl = make_tiling(n_samples, desc->seq_len);
  return (long long)(tl.full_tiles + tl.tail_tiles) * desc->ss_rows * S * (long long)sizeof(float);
}

GLDM_API int gldm_denoise(const gldm_r1d_desc *desc, const float *weights, const float *temb, const float *cemb,
                          int samples_per_cond, const float *x_in, int n_samples, const int32_t *timesteps,
                          const int32_t *sample_t, int n_steps, int sched_kind, int clip_sample,
                          const float *sched_coef, const float *step_noise, float *x_out, void *workspace,
                          gldm_stream_t stream) {
  int st = validate(desc);
  if (st != GLDM_OK) return st;
  if (!weights || !cemb || !x_in || !x_out || !workspace || n_samples <= 0 || n_steps <= 0 || samples_per_cond <= 0)
    return GLDM_ERR_INVALID_ARG;
  if (desc->latent_dim != 0) return GLDM_ERR_INVALID_ARG;
  if (temb && !timesteps && !sample_t) return GLDM_ERR_INVALID_ARG;
  if (sched_kind != GLDM_SCHED_NONE && !sched_coef) return GLDM_ERR_INVALID_ARG;
  if (sched_kind == GLDM_SCHED_NONE && n_steps != 1) return GLDM_ERR_INVALID_ARG;
  RunArgs a{};
  a.d = *desc;
  a.weights = weights; a.temb = temb; a.cemb = cemb; a.samples_per_cond = samples_per_cond;
  a.x_in = x_in; a.n_samples = n_samples; a.timesteps = timesteps; a.sample_t = sample_t; a.n_steps = n_steps;
  a.sched_kind = sched_kind; a.clip_sample = clip_sample; a.sched_coef = sched_coef; a.step_noise = step_noise;
  a.out0 = x_out; a.out1 = nullptr; a.ws = reinterpret_cast<float *>(workspace);
  return launch_r1d(a, reinterpret_cast<hipStream_t>(stream));
}

GLDM_API int gldm_decode(const gldm_r1d_desc *desc, const float *weights, const float *cemb, int samples_per_cond,
                         const float *z_h, int n_samples, float *tmrp, float *logit, void *workspace,
                         gldm_stream_t stream) {
  int st = validate(desc);
  if (st != GLDM_OK) return st;
  if (!weights || !cemb || !z_h || !tmrp || !logit || !workspace || n_samples <= 0 || samples_per_cond <= 0)
    return GLDM_ERR_INVALID_ARG;
  if (desc->latent_dim <= 0 || desc->n_head != 7) return GLDM_ERR_INVALID_ARG;
  RunArgs a{};
  a.d = *desc;
  a.weights = weights; a.temb = nullptr; a.cemb = cemb; a.samples_per_cond = samples_per_cond;
  a.x_in = z_h; a.n_samples = n_samples; a.n_steps = 1; a.sched_kind = GLDM_SCHED_NONE;
  a.out0 = tmrp; a.out1 = logit; a.ws = reinterpret_cast<float *>(workspace);
  return launch_r1d(a, reinterpret_cast<hipStream_t>(stream));
}

GLDM_API int gldm_pose_epilogue(const float *tmrp, const float *logit, const float *grasp_mean,
                                const float *grasp_std, int n, int grasps_per_cloud, float *H, float *tmrp_unnorm,
                                float *confidence, gldm_stream_t stream) {
  if (!tmrp || !grasp_mean || !grasp_std || !H || !tmrp_unnorm || n <= 0 || grasps_per_cloud <= 0)
    return GLDM_ERR_INVALID_ARG;
  if (confidence && !logit) return GLDM_ERR_INVALID_ARG;
  hipLaunchKernelGGL(pose_epilogue_kernel, dim3((n + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     tmrp, logit, grasp_mean, grasp_std, n, grasps_per_cloud, H, tmrp_unnorm, confidence);
  return hipGetLastError() == hipSuccess ? GLDM_OK : GLDM_ERR_LAUNCH;
}

GLDM_API int gldm_sa_mlp_forward(const float *points, const float *centers, const float *features,
                                 const int32_t *idx, const float *weights, int b, int c, int n, int m, int u,
                                 int n_layers, const int32_t *cin_pad, const int32_t *cout, const int32_t *w_off,
                                 const int32_t *b_off, float *out, gldm_stream_t stream) {
  if (!points || !centers || !idx || !weights || !out || !cin_pad || !cout || !w_off || !b_off || b <= 0 || c < 0 ||
      n <= 0 || m <= 0 || u <= 0)
    return GLDM_ERR_INVALID_ARG;
  if (c > 0 && !features) return GLDM_ERR_INVALID_ARG;
  if (n_layers < 1 || n_layers > 4) return GLDM_ERR_UNSUPPORTED;
  if (u > kCols || (kCols % u) != 0) return GLDM_ERR_UNSUPPORTED;
  SaArgs a{};
  a.points = points; a.centers = centers; a.feat = c > 0 ? features : nullptr; a.idx = idx; a.weights = weights;
  a.out = out; a.c = c; a.n = n; a.m = m; a.u = u; a.n_layers = n_layers;
  for (int l = 0; l < n_layers; ++l) {
    const int mt = (cout[l] + 15) >> 4;
    if (cin_pad[l] <= 0 || (cin_pad[l] & 15) || cin_pad[l] > kMaxC || cout[l] > kMaxC ||
        !(mt == 1 || mt == 2 || mt == 4 || mt == 8 || mt == 12 || mt == 16) || (cout[l] & 15))
      return GLDM_ERR_UNSUPPORTED;
    if (l > 0 && cin_pad[l] != cout[l - 1]) return GLDM_ERR_INVALID_ARG;
    a.cin_pad[l] = cin_pad[l]; a.cout[l] = cout[l]; a.w_off[l] = w_off[l]; a.b_off[l] = b_off[l];
  }
  if (cin_pad[0] < 3 + c) return GLDM_ERR_INVALID_ARG;
  const size_t lds_bytes = (size_t)(kBufH + kMaxC * kCols) * sizeof(float);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&sa_mlp_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds_bytes);
    attr = true;
  }
  const int cpt = kCols / u;
  hipLaunchKernelGGL(sa_mlp_kernel, dim3((m + cpt - 1) / cpt, b), dim3(kThreads), lds_bytes,
                     reinterpret_cast<hipStream_t>(stream), a);
  return hipGetLastError() == hipSuccess ? GLDM_OK : GLDM_ERR_LAUNCH;
}
