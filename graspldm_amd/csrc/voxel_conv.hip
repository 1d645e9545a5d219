// voxel_conv.hip -- the voxel branch of PVConv (ext/pvcnn/modules/pvconv.py:47-84) on gfx950:
//   Conv3d(k=3, p=1)  as an implicit GEMM on v_mfma_f32_16x16x4_f32   (gldm_conv3d_k3)
//   GroupNorm(8) + Swish (+ per-channel sums for the SE gate)          (gldm_groupnorm_swish)
//   SE gate (se.py:12-25)                                              (gldm_se_gate)
//   trilinear devoxelize x gate + point-branch features                (gldm_devoxelize_fused)
//
// conv3d mapping: a workgroup (4 waves) owns a 4 x 4 x r brick of output voxels (16 r outputs =
// r n-tiles of 16) for ALL output channels.  Per 16-input-channel block the input brick with
// its one-voxel halo (6 x 6 x (r+2), zero padded at the grid border) is staged in LDS once and
// serves all 27 taps: a tap is a constant LDS offset, so the k-loop is LDS reads + MFMA only.
// Weights (fragment order, k = tap * Cin_pad + ci) stream from L2 one tap ahead.  The LDS row
// stride is = 16 (mod 32) dwords so the 4-row x 16-voxel B-fragment reads are conflict free.
// Two workgroups fit per CU (62 KiB LDS each): one stages while the other computes.
// The conv epilogue also emits per-brick per-channel (sum, sum of squares) so GroupNorm needs
// no extra pass over the tensor for its statistics (combined in f64, fixed order).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "gldm.h"
#include "wstream.h"
#include "devstate.h"

#define GLDM_API extern "C" __attribute__((visibility("default")))

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef __attribute__((address_space(3))) float lds_f;

#ifdef GLDM_DEBUG_KNOBS
__device__ long long g_c3_stamp[32];
#define GLDM_C3_STAMP(i) \
  do { if (blockIdx.x == 5 && blockIdx.y == (gridDim.y >> 1) && threadIdx.x == 0) g_c3_stamp[i] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define GLDM_C3_STAMP(i) do {} while (0)
#endif

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float row16_sum(float x) {  // every lane of a 16-lane DPP row ends with the row's sum
  x += dpp_mov<0xB1>(x);   // quad_perm [1,0,3,2]
  x += dpp_mov<0x4E>(x);   // quad_perm [2,3,0,1]
  x += dpp_mov<0x141>(x);  // row_half_mirror
  x += dpp_mov<0x140>(x);  // row_mirror
  return x;
}

constexpr int kConvThreads = 256;
constexpr int kBrick = 4;  // brick is kBrick x kBrick x r output voxels

// z extent of a brick row in LDS: r + 2, or, where the brick is staged four z at a time, r + 8: float4 slots aligned with
// the grid's own z (slot k = z 4 k - 4 .. 4 k - 1), so that every load is a 16-byte aligned dwordx4 inside the grid and
// the two end slots of a row are all zero padding, written once (resolutions up to 24: the wider bricks would lose
// their second workgroup per CU to the padding)
__host__ __device__ constexpr bool brick_vec4(int r) { return r <= 24; }
__host__ __device__ constexpr int brick_zp(int r) { return brick_vec4(r) ? r + 8 : r + 2; }  // vec4: z = -4 .. r + 3
__host__ __device__ constexpr int brick_row_stride(int r) {  // dwords per channel row in LDS, = 16 mod 32
  return ((6 * 6 * brick_zp(r) + 31) / 32 * 32 + 16) - 32 >= 6 * 6 * brick_zp(r) ? (6 * 6 * brick_zp(r) + 31) / 32 * 32 - 16
                                                                                 : (6 * 6 * brick_zp(r) + 31) / 32 * 32 + 16;
}

// JN: k-steps of 4 input channels per tap and 16-channel block that hold real channels (Cin <= 4: 1, else 4)
template <int MT, int NTW, int JN>
__global__ __launch_bounds__(kConvThreads, 2) void conv3d_k3_kernel(const float *__restrict__ x,
                                                                    const float *__restrict__ wp,
                                                                    const float *__restrict__ bias, int cin, int cout,
                                                                    int r_arg, float *__restrict__ y,
                                                                    float *__restrict__ partial, int cout_total, int co0,
                                                                    int out_cl) {
  // out_cl: y is written channel-last, [b][r^3][cout_total] (see conv3d_k3_pl_kernel); cout % 4 == 0 then
  // cout output channels starting at channel co0 of a cout_total-channel conv (wp / bias already point at the slice):
  // wide convs (256 ch @ 8^3, 128 ch @ 16^3: PVCNN2's feature propagation) run as two launches of half the m-tiles
  constexpr int r = 4 * NTW;  // the instantiation fixes the resolution: every division below is by a constant (with a
  (void)r_arg;                // run-time r the index maps of a workgroup cost 8 k cycles before its first load)
  extern __shared__ float lds[];
  // Everything outside the tap loops (index maps, staging, barriers, epilogue) is short latency-bound work that shares
  // its SIMD with the other workgroups' MFMA streams: it gets issue priority (the streams need one slot per 32 cycles
  // and lose nothing); measured on the 48 -> 96 conv at 12^3, its epilogue took 70 k cycles without.
  __builtin_amdgcn_s_setprio(3);
  GLDM_C3_STAMP(0);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int col = lane & 15, kq = lane >> 4;
  const int bpr = r / kBrick;                       // bricks per axis
  const int bx0 = (blockIdx.x / bpr) * kBrick, by0 = (blockIdx.x % bpr) * kBrick;
  const int b = blockIdx.y;
  constexpr bool kVec = brick_vec4(r);
  constexpr int r3 = r * r * r, zp = brick_zp(r), bv = 36 * zp, bvp = brick_row_stride(r);
  const int cblocks = (cin + 15) >> 4, kblocks = 27 * cblocks;
  x += (size_t)b * cin * r3;
  y += out_cl ? (size_t)b * cout_total * r3 + co0 : ((size_t)b * cout_total + co0) * r3;
  const WStream wv(wp, lane);  // weight fragments: buffer loads, scalar offsets (see wstream.h)
  const lds_f *l3 = (const lds_f *)lds;

  // output voxel of (n-tile, lane column) in brick coordinates; LDS base of its (0,0,0) tap
  int obase[NTW], gvox[NTW];
#pragma unroll
  for (int ni = 0; ni < NTW; ++ni) {
    const int o = 16 * (wave * NTW + ni) + col;       // 0 .. 16 r - 1
    const int iz = o % r, ixy = o / r, ix = ixy >> 2, iy = ixy & 3;
    obase[ni] = (ix * 6 + iy) * zp + iz + (kVec ? 3 : 0) + kq * bvp;  // + row kq of each 4-row k-step (vec4: z = -1 sits at 3)
    gvox[ni] = ((bx0 + ix) * r + by0 + iy) * r + iz;
  }
  f32x4 acc[MT][NTW];
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) {
    f32x4 bvv;
#pragma unroll
    for (int q = 0; q < 4; ++q) bvv[q] = bias[min(16 * mi + 4 * kq + q, cout - 1)];
#pragma unroll
    for (int ni = 0; ni < NTW; ++ni) acc[mi][ni] = bvv;
  }

  // Staging map, computed once.  Scalar form: thread t copies brick elements t, t + 256, ... (SQ of them) of every
  // channel.  Vector form (kVec): a brick row is r / 4 + 2 float4 slots aligned with the grid's z; thread t owns the
  // interior slot t (36 rows x r / 4 of them, at most one per thread): ONE aligned dwordx4 load per channel and ONE
  // ds_write_b128 -- 16 + 16 instructions per thread and block at r = 24 instead of 64 + 64 -- and nothing else: the
  // end slots of every row and the rows outside the grid are zero for the whole life of the workgroup and are written
  // once, before the first block.  That matters because every instruction of these phases waits for a slot between
  // the co-resident workgroup's MFMAs (45-90 cycles each): the staging is priced by its instruction count.
  constexpr int kSlotsRow = r / 4, kSlots = 36 * kSlotsRow;
  static_assert(!kVec || kSlots <= kConvThreads, "one interior slot per thread");
  constexpr int SQ = kVec ? 1 : (36 * (4 * NTW + 2) + kConvThreads - 1) / kConvThreads;  // scalar: 4 at r = 24, 2 at r = 12
  constexpr int kStageCh = JN == 1 ? 4 : 16;  // a <= 4-channel input only ever reads LDS rows 0..3 (one k-step per tap)
  int s_lds[SQ], s_glb[SQ];
#pragma unroll
  for (int q = 0; q < SQ; ++q) {
    const int rem = tid + q * kConvThreads;
    s_lds[q] = -1;
    s_glb[q] = -1;
    if (kVec) {
      if (rem < kSlots) {
        const int ixy = rem / kSlotsRow, k = rem - ixy * kSlotsRow;
        const int gx = bx0 + ixy / 6 - 1, gy = by0 + ixy % 6 - 1;
        s_lds[q] = ixy * zp + 4 * (k + 1);
        if ((unsigned)gx < (unsigned)r && (unsigned)gy < (unsigned)r) s_glb[q] = (gx * r + gy) * r + 4 * k;
      }
    } else if (rem < bv) {
      s_lds[q] = rem;
      const int ixy = rem / zp, izp = rem - ixy * zp;
      const int gx = bx0 + ixy / 6 - 1, gy = by0 + ixy % 6 - 1, gz = izp - 1;
      if ((unsigned)gx < (unsigned)r && (unsigned)gy < (unsigned)r && (unsigned)gz < (unsigned)r)
        s_glb[q] = (gx * r + gy) * r + gz;
    }
  }
  if constexpr (kVec) {  // the zero padding of the brick, once: both end slots of every row, interior slots of outside rows
    const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
    if (tid < 72) {
      float *lc = lds + (tid >> 1) * zp + ((tid & 1) ? 4 * (kSlotsRow + 1) : 0);
#pragma unroll
      for (int ci = 0; ci < kStageCh; ++ci) *reinterpret_cast<f32x4 *>(lc + ci * bvp) = z4;
    }
    if (s_lds[0] >= 0 && s_glb[0] < 0) {
      float *lc = lds + s_lds[0];
#pragma unroll
      for (int ci = 0; ci < kStageCh; ++ci) *reinterpret_cast<f32x4 *>(lc + ci * bvp) = z4;
    }
  }
  // The staging is a latency problem (60 KB per block and workgroup, every load a memory round trip), so the NEXT
  // block's 16 channels are requested before the current block's MFMA loop and only written to LDS after it: the
  // round trips hide behind 1944 MFMAs per wave.
  float stg[kVec ? 1 : SQ][kVec ? 1 : kStageCh];
  f32x4 stv[kVec ? kStageCh : 1];
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x), 0, cin * r3 * 4, 0x00020000);
  auto stage_load = [&](int cb) {
    if constexpr (kVec) {
      // unconditional (threads without a slot read slot 0 and drop it); a channel past cin lies beyond the cloud's
      // slice and reads 0 by the buffer rule
      const int g0 = s_glb[0] >= 0 ? s_glb[0] : 0;
#pragma unroll
      for (int ci = 0; ci < kStageCh; ++ci)
        stv[ci] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, ((cb * 16 + ci) * r3 + g0) * 4, 0, 0));
    } else {
      const float *xc = x + (size_t)(cb * 16) * r3;
#pragma unroll
      for (int q = 0; q < SQ; ++q)
#pragma unroll
        for (int ci = 0; ci < kStageCh; ++ci) {
          stg[q][ci] = 0.f;
          if (s_glb[q] >= 0 && cb * 16 + ci < cin) stg[q][ci] = xc[(size_t)ci * r3 + s_glb[q]];
        }
    }
  };
  auto stage_store = [&]() {
    if constexpr (kVec) {
      if (s_glb[0] >= 0) {
        float *lc = lds + s_lds[0];
#pragma unroll
        for (int ci = 0; ci < kStageCh; ++ci) *reinterpret_cast<f32x4 *>(lc + ci * bvp) = stv[ci];
      }
    } else {
#pragma unroll
      for (int q = 0; q < SQ; ++q)
        if (s_lds[q] >= 0) {
          float *lc = lds + s_lds[q];
#pragma unroll
          for (int ci = 0; ci < kStageCh; ++ci) lc[ci * bvp] = stg[q][ci];
        }
    }
  };
  constexpr int kStageLoads = SQ * kStageCh;  // load instructions per thread and block
  constexpr bool kLate = r <= 12 && kStageLoads <= 32;  // measured per resolution (see the tap loop)
  constexpr bool kPipe = SQ <= 4;  // 64 staged registers beside the accumulators; wider bricks (r = 32) stage in place
  GLDM_C3_STAMP(1);
  if (kPipe) stage_load(0);
  GLDM_C3_STAMP(2);
  for (int cb = 0; cb < cblocks; ++cb) {
    __syncthreads();  // previous block's reads are done
    if (cb < 3) GLDM_C3_STAMP(3 + 4 * cb);
    if (!kPipe) stage_load(cb);
    stage_store();
    if (cb < 3) GLDM_C3_STAMP(4 + 4 * cb);
    if (kPipe && !kLate && cb + 1 < cblocks) stage_load(cb + 1);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    if (cb < 3) GLDM_C3_STAMP(5 + 4 * cb);
    // ---- 27 taps x JN k-steps of MFMA; weights one tap ahead, B fragments (LDS) one k-step ahead
    f32x4 a_cur[MT], a_nxt[MT];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) a_cur[mi] = wv[((size_t)mi * kblocks + cb) * 64];
    float bf[NTW], bn[NTW];
#pragma unroll
    for (int ni = 0; ni < NTW; ++ni) bf[ni] = l3[obase[ni]];  // tap 0, k-step 0
    __builtin_amdgcn_s_setprio(0);
    for (int tap = 0; tap < 27; ++tap) {
      // vmcnt is in order: staged loads requested before the weight loads make the first fragment wait for them.
      // Small bricks (r = 12: 32 loads per thread) are requested behind the LAST weight load of the block instead, with
      // only the last taps' MFMAs in front (8-9 % faster there; for r = 24, 64 loads inside the tap loop, 14 % slower).
#ifdef GLDM_DEBUG_KNOBS
      if (cb < 2 && (tap == 1 || tap == 2 || tap == 14)) GLDM_C3_STAMP(14 + 3 * cb + (tap == 1 ? 0 : (tap == 2 ? 1 : 2)));
#endif
      if (kLate && tap == 25 && cb + 1 < cblocks) stage_load(cb + 1);
      const int tn = tap + 1 < 27 ? tap + 1 : tap;
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) a_nxt[mi] = wv[((size_t)mi * kblocks + tn * cblocks + cb) * 64];
      __builtin_amdgcn_sched_barrier(0);   // pinned in front of this tap's MFMAs: the scheduler sinks the request to its use
      const int toff = ((tap / 9) * 6 + (tap / 3) % 3) * zp + tap % 3;
      const int tnoff = ((tn / 9) * 6 + (tn / 3) % 3) * zp + tn % 3;
#pragma unroll
      for (int j = 0; j < JN; ++j) {
        const int noff = j + 1 < JN ? toff + 4 * (j + 1) * bvp : tnoff;  // the read after the last one is redundant
#pragma unroll
        for (int ni = 0; ni < NTW; ++ni) bn[ni] = l3[obase[ni] + noff];
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
          for (int ni = 0; ni < NTW; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[mi][j], bf[ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
        for (int ni = 0; ni < NTW; ++ni) bf[ni] = bn[ni];
      }
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) a_cur[mi] = a_nxt[mi];
    }
    __builtin_amdgcn_s_setprio(3);
    if (cb < 3) GLDM_C3_STAMP(6 + 4 * cb);
  }
  // ---- epilogue: per-channel partial statistics of this brick, then the stores.  The stores come LAST: a barrier
  // behind them waits for every one of them to be written (vmcnt(0): 70 k cycles per workgroup at r = 12).
  __syncthreads();
  GLDM_C3_STAMP(20);
  float *s_part = lds;  // [4 waves][MT*16][2]
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int co = 16 * mi + 4 * kq + q;
      float s = 0.f, s2 = 0.f;
#pragma unroll
      for (int ni = 0; ni < NTW; ++ni) {
        const float v = acc[mi][ni][q];
        s += v;
        s2 += v * v;
      }
      s = row16_sum(s);   // over the 16 columns of the tile: DPP (a __shfl_xor is an LDS round trip, 8 per row here)
      s2 = row16_sum(s2);
      if (col == 0 && co < cout) {
        s_part[(wave * MT * 16 + co) * 2] = s;
        s_part[(wave * MT * 16 + co) * 2 + 1] = s2;
      }
    }
  }
  GLDM_C3_STAMP(24);
  __syncthreads();
  GLDM_C3_STAMP(25);
  if (tid < cout) {
    float s = 0.f, s2 = 0.f;
    for (int w = 0; w < 4; ++w) {
      s += s_part[(w * MT * 16 + tid) * 2];
      s2 += s_part[(w * MT * 16 + tid) * 2 + 1];
    }
    float *p = partial + (((size_t)b * gridDim.x + blockIdx.x) * cout_total + co0 + tid) * 2;
    p[0] = s;
    p[1] = s2;
  }
  GLDM_C3_STAMP(26);
  if (out_cl) {
#pragma unroll
    for (int ni = 0; ni < NTW; ++ni)
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
        if (16 * mi + 4 * kq < cout) *reinterpret_cast<f32x4 *>(y + (size_t)gvox[ni] * cout_total + 16 * mi + 4 * kq) = acc[mi][ni];
  } else {
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int co = 16 * mi + 4 * kq + q;
        if (co < cout) {
#pragma unroll
          for (int ni = 0; ni < NTW; ++ni) y[(size_t)co * r3 + gvox[ni]] = acc[mi][ni][q];
        }
      }
  }
  GLDM_C3_STAMP(21);
}

// ---- the same conv on split-f16 operands (the shipped encoder's two shapes: 48 ch @ 24^3 and 96 ch @ 12^3) ------------
// v_mfma_f32_16x16x32_bf16 with every f32 operand written as hi + mid + lo (three bf16 numbers, exact) and the six
// partial products of weight >= 2^-16 accumulated in f32: the error of an f32 rounding per product at 6/16 of the
// f32-MFMA time (see csrc/resnet1d.hip, "split-f16 GEMM core").  K is walked as (16-channel block, PAIR of taps):
// lane group g of a fragment = (tap 2 p + (g >> 1), channels 8 (g & 1) .. + 7); 27 taps = 13 pairs + one half-empty
// (zero weights).  Work is walked in groups of 3 m-tiles x 3 n-tiles so that the A sets (double buffered), the B planes
// and the accumulators fit 256 registers.  Weights: graspldm_amd/voxel.py: pack_conv3d_f16x2.
// (First form, measured and replaced: the brick kept f32 in LDS, every lane splitting its 8 channels of a voxel again
// for each (tap pair, n-tile): 2.4 / 0.9 ms per launch at 256 clouds against 2.1 / 0.8 with the pre-split planes below,
// 3.55 / 1.25 on the f32 pipe.)
typedef __attribute__((ext_vector_type(8))) _Float16 c3_f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 c3_f16x2;
typedef __attribute__((ext_vector_type(2))) float c3_f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned c3_u32x4;
constexpr int kC3Split = 2;   // planes per operand: hi | lo (f16)

// x[0..7] -> the hi and lo planes of a fragment: x = hi + lo up to 2^-22 |x| (the matrix pipe keeps f16 subnormals)
__device__ __forceinline__ void c3_split(const float (&x)[8], c3_u32x4 (&pl)[kC3Split]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float a = x[2 * q], b = x[2 * q + 1];
    const c3_f16x2 h = __builtin_convertvector(c3_f32x2{a, b}, c3_f16x2);
    const float ra = __builtin_fmaf((float)h[0], -1.0f, a), rb = __builtin_fmaf((float)h[1], -1.0f, b);
    pl[0][q] = __builtin_bit_cast(unsigned, h);
    pl[1][q] = __builtin_bit_cast(unsigned, __builtin_convertvector(c3_f32x2{ra, rb}, c3_f16x2));
  }
}
__device__ __forceinline__ f32x4 c3_mfma(const c3_u32x4 &a, const c3_u32x4 &b, const f32x4 &c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(c3_f16x8, a), __builtin_bit_cast(c3_f16x8, b), c, 0, 0, 0);
}
// acc += A B, both operands split: the three partial products, small terms first
__device__ __forceinline__ f32x4 c3_mfma3(const c3_u32x4 (&a)[kC3Split], const c3_u32x4 (&b)[kC3Split], f32x4 acc) {
  acc = c3_mfma(a[0], b[1], acc);
  acc = c3_mfma(a[1], b[0], acc);
  return c3_mfma(a[0], b[0], acc);
}

constexpr int kPairs = 14;  // tap pairs per 16-channel block (the last one holds tap 26 and zeros)

// Range scale of a staged brick (see csrc/resnet1d.hip, "range scale of split operands"): f16 has 5 exponent bits, the
// grid's magnitude is the data's (voxel averages of raw features).  s = 1 while 2^-8 <= m < 2^14: every bit as without it.
__device__ __forceinline__ float c3_range_pow2(float m) {
  int e = (int)((__float_as_uint(m) >> 23) & 0xffu) - 127;
  if ((e >= -8 && e < 14) || e < -100 || e > 100) return 1.0f;
  e = e < -40 ? -40 : e;
  return __uint_as_float((unsigned)(e - 13 + 127) << 23);   // m / s in [2^13, 2^14)
}
__device__ __forceinline__ float c3_pow2_inv(float s) { return __uint_as_float((254u << 23) - __float_as_uint(s)); }
__device__ __forceinline__ float c3_wave_max(float x) {   // x >= 0: every lane ends with the wave's maximum
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) x = fmaxf(x, __shfl_xor(x, off, 64));
  return x;
}

// ---- split-f16 conv with the brick PRE-SPLIT in LDS --------------------------------------------------------------------
// Splitting a lane's 8 channels of a voxel for every (tap pair, n-tile) that touches it means ~27 splits per element
// and wave, 36 VALU instructions per 18-36 MFMAs, with the two waves of a SIMD doing it in lock step.  Here the
// staging threads split each brick element ONCE and keep the three bf16 planes in LDS, channel-minor:
//   [plane hi|mid|lo][channel half][voxel of the 6 x 6 x (r + 2) haloed brick][8 channels]   (16 B per voxel, half and plane)
// so a B fragment plane is ONE ds_read_b128 per lane (voxel of its column + the tap of its lane group, channel half)
// and the tap loop is LDS reads + buffer loads + MFMA.  96 B per voxel instead of 64: the 24^3 brick takes 90 KiB, one
// workgroup of 8 waves per CU (3 n-tiles each); the 12^3 brick 48 KiB, 4 waves, three workgroups per CU.
// Staging: an item = (voxel, 8-channel half): 8 dword loads one channel apart (coalesced along z), 36 VALU, three
// ds_write_b128; the next block's raw values are requested before the tap loop and split / stored after it.
// R: grid resolution; ZB: z extent of the brick (R, or a divisor of it: the 24^3 grid runs as two 4 x 4 x 12 half bricks per
// (x, y) -- 48 KiB of planes instead of 90, three workgroups of 4 waves per CU instead of one of 8, so that one brick's
// staging, barriers and epilogue run under another's MFMAs, like the 12^3 conv always did; the two halves of a brick add
// their GroupNorm partials into one zeroed slot with atomics: two addends, so the sum does not depend on their order).
// ACT: the input is the RAW output of the previous conv and GroupNorm + Swish are applied while the brick is staged:
// x' = swish(a[b][ch] x + s[b][ch]) per in-grid element, (a, s) = in_coef (gldm_groupnorm_coef: the previous conv's partial
// statistics folded with the norm's affine).  The zero padding outside the grid is the activated tensor's padding and stays
// zero.  Replaces a read + write pass of groupnorm_swish_kernel over the tensor (0.26 ms at 48 ch @ 24^3 per 256 clouds) by
// ~9 VALU instructions per staged element under the co-resident waves' MFMAs.
template <int MT, int R, int ZB, int WAVES, bool ACT = false>
__global__ __launch_bounds__(64 * WAVES, (ZB * 16 / WAVES / 16 >= 6) ? 1 : 2) void conv3d_k3_pl_kernel(const float *__restrict__ x,
                                                                                      const float *__restrict__ wp3,
                                                                                      const float *__restrict__ bias, int cin,
                                                                                      int cout, float *__restrict__ y,
                                                                                      float *__restrict__ partial,
                                                                                      const float *__restrict__ in_coef,
                                                                                      int out_cl) {
  // out_cl: y is written channel-LAST, [b][r^3][cout] (one 16-byte store per accumulator instead of four 4-byte ones):
  // the layout of a PVConv's last conv, whose readers (SE squeeze, devoxelize) then take a voxel's channels as one run
  constexpr int r = R, NTW = ZB / WAVES, kThreads = 64 * WAVES, kZParts = R / ZB;
  // tiles walked in GM x GN groups: 3 x 3 for the shipped encoder's 48 / 96 channels on 24^3 / 12^3 bricks; the power-of-two
  // widths and grids of PVCNN2 (32 / 64 / 128 channels at 32^3 .. 4^3) in groups of 4, 2 or 1
  constexpr int GM = MT % 3 == 0 ? 3 : (MT % 4 == 0 ? 4 : (MT % 2 == 0 ? 2 : 1));
  constexpr int GN = NTW % 3 == 0 ? 3 : (NTW % 2 == 0 ? 2 : 1);
  static_assert(R % ZB == 0 && ZB % WAVES == 0, "whole n-tiles per wave");
  constexpr int MG = MT / GM, NG = NTW / GN;
  // (Measured and kept out: 4 waves x 6 n-tiles on the 24^3 brick -- half the weight-fragment deliveries, but more than
  // 256 registers, hence one wave per SIMD -- 0.74 ms per launch against 0.67 for 8 waves x 3 n-tiles.)
  constexpr int zp = ZB + 2, nvox = 36 * zp, r3 = r * r * r;
  constexpr int hs = (nvox + 15) & ~15;   // units between the two channel halves of a plane: a multiple of the 16-unit bank row
  constexpr int kItems = 2 * nvox, kRounds = (kItems + kThreads - 1) / kThreads;
  // Two plane sets where the brick leaves room for them and the workgroup has the CU to itself anyway (24^3: 2 x 60 KiB): a
  // wave that is through the tap loop of block cb writes the planes of block cb + 1 into the OTHER set at once instead of
  // waiting at a barrier for the slowest wave's last reads (phase stamps: 2-5 k cycles of such waits per block, then the
  // stores, then a second barrier); one barrier per block is left.
  constexpr bool kDB = R == 24 && ZB == 24;
  // Range scale (raw input only: an activated one is bounded by the norm's affine).  Every 16-channel block is split as
  // x / s_run, s_run the largest power-of-two scale any block so far has asked for (c3_range_pow2 of the block's largest
  // staged magnitude: the waves publish theirs in front of the block's first barrier); when it grows the accumulators --
  // bias included -- are rescaled, and the epilogue multiplies them back.  Ordinary data: s_run = 1 throughout.
  constexpr bool kRanged = !ACT;
  extern __shared__ float lds[];
  GLDM_C3_STAMP(0);
  __builtin_amdgcn_s_setprio(3);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int col = lane & 15, kq = lane >> 4;
  const int bpr = r / kBrick;
  const int bxy = blockIdx.x / kZParts, bz0 = (blockIdx.x % kZParts) * ZB;
  const int bx0 = (bxy / bpr) * kBrick, by0 = (bxy % bpr) * kBrick;
  const int b = blockIdx.y;
  const int cblocks = (cin + 15) >> 4, kblocks = kPairs * cblocks;
  x += (size_t)b * cin * r3;
  y += (size_t)b * cout * r3;
  const WStream wv(wp3, lane);
  typedef __attribute__((address_space(3))) c3_u32x4 lds_c4;
  constexpr int kSet = kC3Split * 2 * hs;   // 16-byte units of one plane set
  lds_c4 *pl0 = (lds_c4 *)lds;   // 16-byte units: plane * 2 hs + half * hs + voxel.  A ds_read_b128 is served in 16-lane groups
                                // that mix columns 0-3, 12-15 of one lane row with columns 4-11 of the next (the other channel
                                // half): with hs a multiple of 16 units the two sets fall on disjoint banks

  // fragment read base of (n-tile, lane column): voxel (0,0,0)-tap of the column, this lane's channel half
  int vb[NTW];
#pragma unroll
  for (int ni = 0; ni < NTW; ++ni) {
    const int o = 16 * (wave * NTW + ni) + col;
    const int iz = o % ZB, ixy = o / ZB, ix = ixy >> 2, iy = ixy & 3;
    vb[ni] = (ix * 6 + iy) * zp + iz + (kq & 1) * hs;
  }
  f32x4 acc[MT][NTW];
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) {
    f32x4 bvv;
#pragma unroll
    for (int q = 0; q < 4; ++q) bvv[q] = bias[16 * mi + 4 * kq + q];   // cout == 16 MT (the launcher checks)
#pragma unroll
    for (int ni = 0; ni < NTW; ++ni) acc[mi][ni] = bvv;
  }
  // ---- staging map: item i = (voxel i >> 1 ... ) ordered so that consecutive threads take consecutive z of one half
  int s_lds[kRounds], s_glb[kRounds];
#pragma unroll
  for (int q = 0; q < kRounds; ++q) {
    const int it = tid + q * kThreads;
    s_lds[q] = -1;
    s_glb[q] = -1;
    if (it < kItems) {
      const int h = it / nvox, v = it - h * nvox;       // half-major: a wave's loads run along z
      const int ixy = v / zp, izp = v - ixy * zp;
      const int gx = bx0 + ixy / 6 - 1, gy = by0 + ixy % 6 - 1, gz = bz0 + izp - 1;
      s_lds[q] = h * hs + v;
      if ((unsigned)gx < (unsigned)r && (unsigned)gy < (unsigned)r && (unsigned)gz < (unsigned)r)
        s_glb[q] = (gx * r + gy) * r + gz + 8 * h * r3;
    }
  }
  // halo voxels outside the grid stay zero for the whole kernel: written once, all three planes
  {
    const c3_u32x4 z4 = c3_u32x4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int q = 0; q < kRounds; ++q)
      if (s_lds[q] >= 0 && s_glb[q] < 0) {
#pragma unroll
        for (int p3 = 0; p3 < kC3Split; ++p3) {
          pl0[p3 * 2 * hs + s_lds[q]] = z4;
          if constexpr (kDB) pl0[kSet + p3 * 2 * hs + s_lds[q]] = z4;
        }
      }
  }
  float stg[kRounds][8];
  auto stage_load = [&](int cb) {
    const float *xc = x + (size_t)(cb * 16) * r3;
#pragma unroll
    for (int q = 0; q < kRounds; ++q) {
      const int g0 = s_glb[q] >= 0 ? s_glb[q] : 0;   // unconditional loads on a clamped address
#pragma unroll
      for (int j = 0; j < 8; ++j) stg[q][j] = xc[(size_t)j * r3 + g0];
    }
  };
  // ACT: (a, s) of every input channel of this cloud, behind the planes (the launcher adds the room): read back as
  // wave-wide broadcasts of four 16-byte pairs-of-pairs per staged item
  float *s_coef = lds + (kDB ? 2 : 1) * kSet * 4;   // [cin][2]
  if constexpr (ACT) {
    for (int i = tid; i < 2 * cin; i += kThreads) s_coef[i] = in_coef[(size_t)b * cin * 2 + i];
  }
  unsigned s_run_u = 0x3f800000u;   // s_run's bits, kept on the scalar side
  float *rng = s_coef;   // [WAVES] (raw input: no coefficients there; the launcher adds the room)
  auto stage_store = [&](int cb) {
    lds_c4 *pl = pl0 + (kDB ? (cb & 1) * kSet : 0);
#pragma unroll
    for (int q = 0; q < kRounds; ++q)
      if (s_glb[q] >= 0) {
        if constexpr (kRanged) {
          const float inv_run = __uint_as_float((254u << 23) - s_run_u);
#pragma unroll
          for (int j = 0; j < 8; ++j) stg[q][j] *= inv_run;
        }
        if constexpr (ACT) {
          const int h = (tid + q * kThreads) / nvox;
          const f32x4 *cf = reinterpret_cast<const f32x4 *>(s_coef + 2 * (16 * cb + 8 * h));   // (a, s) x 8 channels
#pragma unroll
          for (int j2 = 0; j2 < 4; ++j2) {
            const f32x4 c4 = cf[j2];
            const float t0 = fmaf(stg[q][2 * j2], c4[0], c4[1]), t1 = fmaf(stg[q][2 * j2 + 1], c4[2], c4[3]);
            stg[q][2 * j2] = t0 * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * t0));
            stg[q][2 * j2 + 1] = t1 * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * t1));
          }
        }
        c3_u32x4 p3[kC3Split];
        c3_split(stg[q], p3);
#pragma unroll
        for (int k = 0; k < kC3Split; ++k) pl[k * 2 * hs + s_lds[q]] = p3[k];
      }
  };
  GLDM_C3_STAMP(1);
  stage_load(0);
  GLDM_C3_STAMP(2);
  for (int cb = 0; cb < cblocks; ++cb) {
    if constexpr (kRanged) {
      float mx = 0.f;
#pragma unroll
      for (int q = 0; q < kRounds; ++q)
        if (s_glb[q] >= 0) {
#pragma unroll
          for (int j = 0; j < 8; ++j) mx = fmaxf(mx, fabsf(stg[q][j]));
        }
      mx = c3_wave_max(mx);
      if (lane == 0) rng[wave] = mx;
    }
    if (!kDB || kRanged || cb == 0) __syncthreads();   // the previous block's readers are done (two sets: its planes are not touched)
    if constexpr (kRanged) {
      float mx = 0.f;
#pragma unroll
      for (int w8 = 0; w8 < WAVES; ++w8) mx = fmaxf(mx, rng[w8]);
      const unsigned need = __builtin_amdgcn_readfirstlane(__float_as_uint(c3_range_pow2(mx)));
      const unsigned s_new = (cb == 0 || need > s_run_u) ? need : s_run_u;   // positive floats order like their bits
      if (s_new != s_run_u) {   // wave uniform, never taken on ordinary data
        const float f = __uint_as_float(s_run_u) * c3_pow2_inv(__uint_as_float(s_new));
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
          for (int ni = 0; ni < NTW; ++ni) acc[mi][ni] *= f;
        s_run_u = s_new;
      }
    }
    if (cb < 3) GLDM_C3_STAMP(3 + 4 * cb);
    stage_store(cb);
    if (cb + 1 < cblocks) stage_load(cb + 1);
    __builtin_amdgcn_sched_barrier(0);
    if (cb < 3) GLDM_C3_STAMP(4 + 4 * cb);
    __syncthreads();   // two sets: also "everyone is through block cb - 1's taps", so block cb + 1 may overwrite its set
    if (cb < 3) GLDM_C3_STAMP(5 + 4 * cb);
    const lds_c4 *pl = pl0 + (kDB ? (cb & 1) * kSet : 0);
    c3_u32x4 a[2][GM][kC3Split];
    auto load_a = [&](int buf, int step) {   // step = pair * MG + mg
      const int p = step / MG, mg = step - p * MG;
#pragma unroll
      for (int mi = 0; mi < GM; ++mi)
#pragma unroll
        for (int k = 0; k < kC3Split; ++k)
          a[buf][mi][k] = wv.raw_at((((GM * mg + mi) * kblocks + cb * kPairs + p) * kC3Split) * 1024, k * 1024);
    };
    load_a(0, 0);
    __builtin_amdgcn_s_setprio(0);
    constexpr int kSteps = kPairs * MG;
    constexpr int kUnits = kPairs * NG;   // (pair, n-group) units: the B planes of unit u + 1 are requested during unit u
    constexpr bool kBPre = NG > 1;   // a second B set costs 36 registers: only where a wave has two n-groups
    c3_u32x4 bs[kBPre ? 2 : 1][GN][kC3Split];
    auto load_b = [&](int buf, int unit) {
      const int p = unit / NG, ng = unit - p * NG;
      const int ta = 2 * p, tb = 2 * p + 1 < 27 ? 2 * p + 1 : 26;
      const int offa = ((ta / 9) * 6 + (ta / 3) % 3) * zp + ta % 3, offb = ((tb / 9) * 6 + (tb / 3) % 3) * zp + tb % 3;
      const int toff = (kq >> 1) ? offb : offa;
#pragma unroll
      for (int q = 0; q < GN; ++q)
#pragma unroll
        for (int k = 0; k < kC3Split; ++k) bs[buf][q][k] = pl[k * 2 * hs + vb[GN * ng + q] + toff];
    };
    if (kBPre) load_b(0, 0);
    for (int p0 = 0; p0 < kPairs; p0 += 2) {
#pragma unroll
      for (int pu = 0; pu < 2; ++pu) {
        const int p = p0 + pu;
#pragma unroll
        for (int ng = 0; ng < NG; ++ng) {
          const int unit = p * NG + ng;
          const int bcur = kBPre ? ((pu * NG + ng) & 1) : 0;   // p0 * NG is even
          if (kBPre) load_b(bcur ^ 1, unit + 1 < kUnits ? unit + 1 : unit);
          else load_b(0, unit);
#pragma unroll
          for (int mg = 0; mg < MG; ++mg) {
            const int step = p * MG + mg;
            const int cur = (pu * MG + mg) & 1;
            // the A set of the next (pair, m-group); with two n-groups a set serves both (requested in the first)
            if (ng == 0) load_a(cur ^ 1, step + 1 < kSteps ? step + 1 : step);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mi = 0; mi < GM; ++mi)
#pragma unroll
              for (int q = 0; q < GN; ++q)
                acc[GM * mg + mi][GN * ng + q] = c3_mfma3(a[cur][mi], bs[bcur][q], acc[GM * mg + mi][GN * ng + q]);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    }
    if (cb < 3) GLDM_C3_STAMP(6 + 4 * cb);
    __builtin_amdgcn_s_setprio(3);
  }
  // ---- epilogue: per-channel partial statistics of this brick, then the stores
  __syncthreads();
  GLDM_C3_STAMP(20);
  if constexpr (kRanged) {
    if (s_run_u != 0x3f800000u) {
      const float s_run = __uint_as_float(s_run_u);
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NTW; ++ni) acc[mi][ni] *= s_run;
    }
  }
  float *s_part = lds;  // [WAVES][MT * 16][2]
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int co = 16 * mi + 4 * kq + q;
      float s = 0.f, s2 = 0.f;
#pragma unroll
      for (int ni = 0; ni < NTW; ++ni) {
        const float v = acc[mi][ni][q];
        s += v;
        s2 += v * v;
      }
      s = row16_sum(s);
      s2 = row16_sum(s2);
      if (col == 0) {
        s_part[(wave * MT * 16 + co) * 2] = s;
        s_part[(wave * MT * 16 + co) * 2 + 1] = s2;
      }
    }
  }
  __syncthreads();
  if (tid < cout) {
    float s = 0.f, s2 = 0.f;
    for (int w = 0; w < WAVES; ++w) {
      s += s_part[(w * MT * 16 + tid) * 2];
      s2 += s_part[(w * MT * 16 + tid) * 2 + 1];
    }
    float *p = partial + (((size_t)b * (gridDim.x / kZParts) + bxy) * cout + tid) * 2;
    if constexpr (kZParts == 1) {
      p[0] = s;
      p[1] = s2;
    } else {   // the slot was zeroed by the launcher; two addends: order independent
      atomicAdd(p, s);
      atomicAdd(p + 1, s2);
    }
  }
  // the output voxels of (n-tile, lane column), derived here from an opaque copy of the lane id: computed in front of
  // the block loop they were NTW more registers alive across it (one instantiation spilled)
  int gvox[NTW];
  {
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));
#pragma unroll
    for (int ni = 0; ni < NTW; ++ni) {
      const int o = 16 * (wave * NTW + ni) + (lane_o & 15);
      const int iz = o % ZB, ixy = o / ZB, ix = ixy >> 2, iy = ixy & 3;
      gvox[ni] = ((bx0 + ix) * r + by0 + iy) * r + bz0 + iz;
    }
  }
  if (out_cl) {
#pragma unroll
    for (int ni = 0; ni < NTW; ++ni)
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
        *reinterpret_cast<f32x4 *>(y + (size_t)gvox[ni] * cout + 16 * mi + 4 * kq) = acc[mi][ni];
  } else {
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int co = 16 * mi + 4 * kq + q;
#pragma unroll
        for (int ni = 0; ni < NTW; ++ni) y[(size_t)co * r3 + gvox[ni]] = acc[mi][ni][q];
      }
  }
  GLDM_C3_STAMP(21);
}

// ---- split-f16 conv of a FEW input channels (the encoder's first voxel conv: 3 -> 48 at 24^3) ------------------------
// With k = tap * 16 + ci (conv3d_k3_kernel, JN = 1) a 3-channel input pays 27 k-steps of 4 on the f32 pipe for 81 real
// products per output: 0.49 ms per 256 clouds at 0.55 matrix-pipe occupancy, for a tensor whose store takes 0.1 ms.  Here
// K is packed tap-major, channel-minor without padding between taps: k = tap * CIN + ci < 27 CIN, rounded up ONCE to a
// multiple of 32 (CIN = 3: 81 -> 96 = three k-blocks of v_mfma_f32_16x16x32_bf16), six bf16 partial products per f32
// product as everywhere else.  Lane (g, col) of a B fragment gathers its 8 consecutive k = (tap, ci) pairs of its voxel
// from the f32 brick in LDS (8 ds_read_b32 at per-lane offsets, one n-tile ahead of their use) and splits them; the A
// planes of a k-block (MT m-tiles x 3 planes) are read once per workgroup.  A workgroup of 8 waves owns a 4 x 4 x R
// brick, R / 8 n-tiles per wave, 117 registers: two workgroups per CU; the haloed input brick is CIN x 6 x 6 x (R + 2)
// floats (11 KiB at R = 24).  0.32 ms.  (Measured and dropped: persistent workgroups with the weight planes in LDS and the
// next brick's input requested under the current k-loop -- 216 registers, one workgroup per CU: 0.41 ms; the same with
// two 4-wave workgroups per CU spills: 0.59 ms.)  Weights: graspldm_amd/voxel.py: pack_conv3d_fewch_f16x2.
template <int MT, int R, int CIN, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 4) void conv3d_k3_fewch_sp_kernel(const float *__restrict__ x,
                                                                           const float *__restrict__ wp3,
                                                                           const float *__restrict__ bias,
                                                                           float *__restrict__ y,
                                                                           float *__restrict__ partial) {
  constexpr int r = R, NTW = R / WAVES, kThreads = 64 * WAVES, zp = R + 2, nvox = 36 * zp, r3 = r * r * r;
  static_assert(R % WAVES == 0, "whole n-tiles per wave");
  constexpr int K = 27 * CIN, KB = (K + 31) / 32, cout = 16 * MT;
  constexpr int cs = nvox + ((nvox & 31) == 0 ? 8 : 0);   // channel stride in LDS
  extern __shared__ float lds[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int col = lane & 15, g = lane >> 4;
  const int bpr = r / kBrick;
  const int bx0 = (blockIdx.x / bpr) * kBrick, by0 = (blockIdx.x % bpr) * kBrick;
  const int b = blockIdx.y;
  x += (size_t)b * CIN * r3;
  y += (size_t)b * cout * r3;
  // ---- the haloed brick, f32, zero outside the grid; its largest magnitude for the range scale (c3_range_pow2: the
  // grid holds voxel averages of raw coordinates / features)
  float mx = 0.f;
  for (int i = tid; i < CIN * nvox; i += kThreads) {
    const int ci = i / nvox, v = i - ci * nvox;
    const int ixy = v / zp, izp = v - ixy * zp;
    const int gx = bx0 + ixy / 6 - 1, gy = by0 + ixy % 6 - 1, gz = izp - 1;
    float val = 0.f;
    if ((unsigned)gx < (unsigned)r && (unsigned)gy < (unsigned)r && (unsigned)gz < (unsigned)r)
      val = x[(size_t)ci * r3 + (gx * r + gy) * r + gz];
    lds[ci * cs + v] = val;
    mx = fmaxf(mx, fabsf(val));
  }
  float *rng = lds + CIN * cs;   // [WAVES], behind the brick (the launcher adds the room)
  mx = c3_wave_max(mx);
  if (lane == 0) rng[wave] = mx;
  int vb[NTW], gvox[NTW];
#pragma unroll
  for (int ni = 0; ni < NTW; ++ni) {
    const int o = 16 * (wave * NTW + ni) + col;
    const int iz = o % r, ixy = o / r, ix = ixy >> 2, iy = ixy & 3;
    vb[ni] = (ix * 6 + iy) * zp + iz;
    gvox[ni] = ((bx0 + ix) * r + by0 + iy) * r + iz;
  }
  const WStream wv(wp3, lane);
  const lds_f *l3 = (const lds_f *)lds;
  __syncthreads();
  float s_in;
  {
    float m8 = 0.f;
#pragma unroll
    for (int w8 = 0; w8 < WAVES; ++w8) m8 = fmaxf(m8, rng[w8]);
    s_in = c3_range_pow2(__uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(m8))));
  }
  const float inv_in = c3_pow2_inv(s_in);
  f32x4 acc[MT][NTW];
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) {
    const f32x4 bvv = *reinterpret_cast<const f32x4 *>(bias + 16 * mi + 4 * g) * inv_in;
#pragma unroll
    for (int ni = 0; ni < NTW; ++ni) acc[mi][ni] = bvv;
  }
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    c3_u32x4 a[MT][kC3Split];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
      for (int pl = 0; pl < kC3Split; ++pl) a[mi][pl] = wv.raw_at(((mi * KB + kb) * kC3Split) * 1024, pl * 1024);
    int off[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = 32 * kb + 8 * g + j;
      const int tap = k / CIN, ci = k - tap * CIN;
      // k >= K: zero weights; any finite brick element serves
      off[j] = k < K ? ci * cs + ((tap / 9) * 6 + (tap / 3) % 3) * zp + tap % 3 : 0;
    }
    // the gathers run one n-tile ahead of the split + MFMAs that use them
    float v[2][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[0][j] = l3[vb[0] + off[j]];
#pragma unroll
    for (int ni = 0; ni < NTW; ++ni) {
      if (ni + 1 < NTW) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[(ni + 1) & 1][j] = l3[vb[ni + 1] + off[j]];
      }
      c3_u32x4 b3[kC3Split];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[ni & 1][j] *= inv_in;
      c3_split(v[ni & 1], b3);
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) acc[mi][ni] = c3_mfma3(a[mi], b3, acc[mi][ni]);
    }
  }
  if (s_in != 1.0f) {   // wave uniform
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
      for (int ni = 0; ni < NTW; ++ni) acc[mi][ni] *= s_in;
  }
  // ---- epilogue: per-channel partial statistics of this brick, then the stores (as in conv3d_k3_pl_kernel)
  __syncthreads();
  float *s_part = lds;  // [WAVES][MT * 16][2]
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int co = 16 * mi + 4 * g + q;
      float s = 0.f, s2 = 0.f;
#pragma unroll
      for (int ni = 0; ni < NTW; ++ni) {
        const float v = acc[mi][ni][q];
        s += v;
        s2 += v * v;
      }
      s = row16_sum(s);
      s2 = row16_sum(s2);
      if (col == 0) {
        s_part[(wave * MT * 16 + co) * 2] = s;
        s_part[(wave * MT * 16 + co) * 2 + 1] = s2;
      }
    }
  }
  __syncthreads();
  if (tid < cout) {
    float s = 0.f, s2 = 0.f;
    for (int w = 0; w < WAVES; ++w) {
      s += s_part[(w * MT * 16 + tid) * 2];
      s2 += s_part[(w * MT * 16 + tid) * 2 + 1];
    }
    float *p = partial + (((size_t)b * gridDim.x + blockIdx.x) * cout + tid) * 2;
    p[0] = s;
    p[1] = s2;
  }
#pragma unroll
  for (int mi = 0; mi < MT; ++mi)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int co = 16 * mi + 4 * g + q;
#pragma unroll
      for (int ni = 0; ni < NTW; ++ni) y[(size_t)co * r3 + gvox[ni]] = acc[mi][ni][q];
    }
}

// GroupNorm(groups) + Swish over [B, C, r^3]; statistics from the conv's per-brick partials.
// grid = (groups, B); optional per-channel sum of the OUTPUT (for the SE squeeze).
__global__ __launch_bounds__(512) void groupnorm_swish_kernel(float *__restrict__ y, const float *__restrict__ partial,
                                                              const float *__restrict__ gamma,
                                                              const float *__restrict__ beta, int c, int r3,
                                                              int nbricks, int groups, float eps,
                                                              float *__restrict__ chan_sum) {
  __shared__ double s_stat[2];
  __shared__ float s_red[8];
  const int g = blockIdx.x, b = blockIdx.y, cpg = c / groups;
  const int tid = threadIdx.x;
  if (tid < 64) {
    double s = 0.0, s2 = 0.0;
    for (int i = tid; i < nbricks * cpg; i += 64) {
      const int br = i / cpg, ch = g * cpg + i % cpg;
      const float *p = partial + (((size_t)b * nbricks + br) * c + ch) * 2;
      s += (double)p[0];
      s2 += (double)p[1];
    }
    for (int off = 32; off >= 1; off >>= 1) {
      s += __shfl_xor(s, off, 64);
      s2 += __shfl_xor(s2, off, 64);
    }
    if (tid == 0) {
      const double n = (double)cpg * r3, mean = s / n;
      s_stat[0] = mean;
      s_stat[1] = 1.0 / sqrt(fmax(s2 / n - mean * mean, 0.0) + (double)eps);
    }
  }
  __syncthreads();
  const float mean = (float)s_stat[0], rstd = (float)s_stat[1];
  for (int cc = 0; cc < cpg; ++cc) {
    const int ch = g * cpg + cc;
    float *row = y + ((size_t)b * c + ch) * r3;
    const float ga = gamma[ch] * rstd, be = beta[ch] - mean * rstd * gamma[ch];
    float acc = 0.f;
    const int n4 = (r3 & 3) ? 0 : r3 >> 2;   // rows are 16-byte aligned only when r^3 is a multiple of 4
    if (r3 & 3) {
      for (int i = tid; i < r3; i += 512) {
        const float t = row[i] * ga + be;
        const float o = t / (1.0f + __expf(-t));
        acc += o;
        row[i] = o;
      }
    }
    for (int i = tid; i < n4; i += 512) {
      float4 v = reinterpret_cast<float4 *>(row)[i];
      float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float t = o[q] * ga + be;
        o[q] = t / (1.0f + __expf(-t));
        acc += o[q];
      }
      reinterpret_cast<float4 *>(row)[i] = make_float4(o[0], o[1], o[2], o[3]);
    }
    if (chan_sum) {
      for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
      __syncthreads();
      if ((tid & 63) == 0) s_red[tid >> 6] = acc;
      __syncthreads();
      if (tid == 0) {
        float t = 0.f;
        for (int w = 0; w < 8; ++w) t += s_red[w];
        chan_sum[(size_t)b * c + ch] = t;
      }
    }
  }
}

// SE gate: gate = sigmoid(W2 act(W1 mean)), W1 [c/red, c], W2 [c, c/red]; one block per cloud.
__global__ void se_gate_kernel(const float *__restrict__ chan_sum, const float *__restrict__ w1,
                               const float *__restrict__ w2, int c, int hid, int r3, int use_relu,
                               float *__restrict__ gate, int parts) {
  extern __shared__ float s[];  // mean[c], h[hid]
  const int b = blockIdx.x, tid = threadIdx.x;
  float *mean = s, *h = s + c;
  // chan_sum [b][parts][c]: partial sums of the squeeze, added in index order
  for (int i = tid; i < c; i += blockDim.x) {
    float t = chan_sum[(size_t)b * parts * c + i];
    for (int p = 1; p < parts; ++p) t += chan_sum[((size_t)b * parts + p) * c + i];
    mean[i] = t / (float)r3;
  }
  __syncthreads();
  for (int i = tid; i < hid; i += blockDim.x) {
    float a = 0.f;
    for (int q = 0; q < c; ++q) a += w1[i * c + q] * mean[q];
    h[i] = use_relu ? fmaxf(a, 0.f) : a / (1.0f + expf(-a));
  }
  __syncthreads();
  for (int i = tid; i < c; i += blockDim.x) {
    float a = 0.f;
    for (int q = 0; q < hid; ++q) a += w2[i * hid + q] * h[q];
    gate[(size_t)b * c + i] = 1.0f / (1.0f + expf(-a));
  }
}

// GroupNorm as per-(cloud, channel) coefficients: y = a x + s with a = gamma rstd, s = beta - mean a; statistics from the
// conv's per-brick partials, combined in f64 in a fixed order exactly as groupnorm_swish_kernel does.  grid = (groups, B).
__global__ __launch_bounds__(64) void groupnorm_coef_kernel(const float *__restrict__ partial, const float *__restrict__ gamma,
                                                            const float *__restrict__ beta, int c, int r3, int nbricks,
                                                            int groups, float eps, float *__restrict__ coef) {
  const int g = blockIdx.x, b = blockIdx.y, cpg = c / groups, tid = threadIdx.x;
  double s = 0.0, s2 = 0.0;
  for (int i = tid; i < nbricks * cpg; i += 64) {
    const int br = i / cpg, ch = g * cpg + i % cpg;
    const float *p = partial + (((size_t)b * nbricks + br) * c + ch) * 2;
    s += (double)p[0];
    s2 += (double)p[1];
  }
  for (int off = 32; off >= 1; off >>= 1) {
    s += __shfl_xor(s, off, 64);
    s2 += __shfl_xor(s2, off, 64);
  }
  const double n = (double)cpg * r3, mean_d = s / n;
  const float mean = (float)mean_d, rstd = (float)(1.0 / sqrt(fmax(s2 / n - mean_d * mean_d, 0.0) + (double)eps));
  if (tid < cpg) {
    const int ch = g * cpg + tid;
    const float a = gamma[ch] * rstd;
    coef[((size_t)b * c + ch) * 2] = a;
    coef[((size_t)b * c + ch) * 2 + 1] = beta[ch] - mean * rstd * gamma[ch];
  }
}

__device__ __forceinline__ float swish_fast(float t) {
  return t * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * t));
}

// chan_sum[b][ch] = sum over the voxels of swish(a x + s): the SE squeeze of a GroupNorm + Swish output that is never
// written (read-only pass; the consumers apply the same map on the fly).  grid = (C, B), rows of r^3 floats.
__global__ __launch_bounds__(256) void gn_swish_sum_kernel(const float *__restrict__ y, const float *__restrict__ coef, int c,
                                                           int r3, float *__restrict__ chan_sum) {
  __shared__ float s_red[4];
  const int ch = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const float *row = y + ((size_t)b * c + ch) * r3;
  const float a = coef[((size_t)b * c + ch) * 2], s = coef[((size_t)b * c + ch) * 2 + 1];
  float acc = 0.f;
  if ((r3 & 3) == 0) {
    const float4 *row4 = reinterpret_cast<const float4 *>(row);
    for (int i = tid; i < (r3 >> 2); i += 256) {
      const float4 v = row4[i];
      acc += swish_fast(fmaf(v.x, a, s)) + swish_fast(fmaf(v.y, a, s)) + swish_fast(fmaf(v.z, a, s)) + swish_fast(fmaf(v.w, a, s));
    }
  } else {
    for (int i = tid; i < r3; i += 256) acc += swish_fast(fmaf(row[i], a, s));
  }
  for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if ((tid & 63) == 0) s_red[tid >> 6] = acc;
  __syncthreads();
  if (tid == 0) chan_sum[(size_t)b * c + ch] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

// The same squeeze over a channel-LAST tensor [b][r^3][c]: block (part, b) sums its share of the voxels for every channel
// (thread = (voxel stripe, channel quad), 16-byte loads), parts[b][part][c] leaves; se_gate_kernel adds the parts in order.
constexpr int kSumParts = 8;
__global__ __launch_bounds__(256) void gn_swish_sum_cl_kernel(const float *__restrict__ y, const float *__restrict__ coef, int c,
                                                              int r3, float *__restrict__ parts) {
  __shared__ f32x4 s_acc[256];
  const int part = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int quads = c >> 2, stripes = 256 / quads;          // threads beyond stripes * quads idle
  const int qd = tid % quads, stripe = tid / quads;
  const int v0 = (int)((long long)r3 * part / kSumParts), v1 = (int)((long long)r3 * (part + 1) / kSumParts);
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  if (stripe < stripes) {
    const f32x4 *cf = reinterpret_cast<const f32x4 *>(coef + ((size_t)b * c + 4 * qd) * 2);
    const f32x4 c01 = cf[0], c23 = cf[1];   // (a0, s0, a1, s1), (a2, s2, a3, s3)
    const f32x4 *row = reinterpret_cast<const f32x4 *>(y + (size_t)b * r3 * c) + qd;
    for (int v = v0 + stripe; v < v1; v += stripes) {
      const f32x4 x = row[(size_t)v * quads];
      acc[0] += swish_fast(fmaf(x[0], c01[0], c01[1]));
      acc[1] += swish_fast(fmaf(x[1], c01[2], c01[3]));
      acc[2] += swish_fast(fmaf(x[2], c23[0], c23[1]));
      acc[3] += swish_fast(fmaf(x[3], c23[2], c23[3]));
    }
  }
  s_acc[tid] = acc;
  __syncthreads();
  if (tid < quads) {
    f32x4 t = s_acc[tid];
    for (int st = 1; st < stripes; ++st) {
      const f32x4 o = s_acc[st * quads + tid];
      t[0] += o[0]; t[1] += o[1]; t[2] += o[2]; t[3] += o[3];
    }
    *reinterpret_cast<f32x4 *>(parts + ((size_t)b * kSumParts + part) * c + 4 * tid) = t;
  }
}

// devoxelize_fused_kernel over a channel-LAST raw conv output [b][r^3][c] (coef required): a point's corner is ONE run of
// c floats, read as 16-byte loads by c / 4 neighbouring lanes, instead of c dword gathers from c cache lines (the
// channel-major form is bound by the address path: 64 lines per wave instruction, 0.24 ms per 48 x 24^3 x 256 clouds).
// Block = 64 points; item = (point, channel quad); the results cross LDS so that the stores (and the reads of `add`) run
// along the points.  c % 4 == 0, c <= 256.
__global__ __launch_bounds__(256) void devoxelize_cl_kernel(const float *__restrict__ coords, const float *__restrict__ feat,
                                                            const float *__restrict__ coef, const float *__restrict__ gate,
                                                            const float *__restrict__ add, int c, int n, int r,
                                                            float *__restrict__ outs) {
  extern __shared__ float s_tile[];   // [c][65]
  const int b = blockIdx.y, p0 = blockIdx.x * 64, tid = threadIdx.x;
  const int r2 = r * r, r3 = r2 * r, quads = c >> 2;
  coords += (size_t)b * 3 * n;
  const f32x4 *f4 = reinterpret_cast<const f32x4 *>(feat + (size_t)b * r3 * c);
  for (int it = tid; it < 64 * quads; it += 256) {
    const int pt = it / quads, qd = it - pt * quads;
    const int i = min(p0 + pt, n - 1);
    const float x = coords[i], y = coords[i + n], z = coords[i + 2 * n];
    const float xl = floorf(x), yl = floorf(y), zl = floorf(z);
    const float xd1 = x - xl, yd1 = y - yl, zd1 = z - zl;
    const float xd0 = 1.0f - xd1, yd0 = 1.0f - yd1, zd0 = 1.0f - zd1;
    const float w[8] = {xd0 * yd0 * zd0, xd0 * yd0 * zd1, xd0 * yd1 * zd0, xd0 * yd1 * zd1,
                        xd1 * yd0 * zd0, xd1 * yd0 * zd1, xd1 * yd1 * zd0, xd1 * yd1 * zd1};
    const int i000 = (int)xl * r2 + (int)yl * r + (int)zl;
    const int zh = zd1 > 0 ? 1 : 0, yh = yd1 > 0 ? r : 0, xh = xd1 > 0 ? r2 : 0;
    const int idx[8] = {i000, i000 + zh, i000 + yh, i000 + yh + zh, i000 + xh, i000 + xh + zh, i000 + xh + yh, i000 + xh + yh + zh};
    f32x4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = f4[(size_t)idx[k] * quads + qd];
    const f32x4 *cf = reinterpret_cast<const f32x4 *>(coef + ((size_t)b * c + 4 * qd) * 2);
    const f32x4 c01 = cf[0], c23 = cf[1];
    const float ca[4] = {c01[0], c01[2], c23[0], c23[2]}, cs[4] = {c01[1], c01[3], c23[1], c23[3]};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float o = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) o += w[k] * swish_fast(fmaf(v[k][q], ca[q], cs[q]));
      s_tile[(4 * qd + q) * 65 + pt] = o;
    }
  }
  __syncthreads();
  for (int e = tid; e < c * 64; e += 256) {
    const int ch = e >> 6, pt = e & 63;
    if (p0 + pt < n) {
      const float gt = gate ? gate[(size_t)b * c + ch] : 1.0f;
      const size_t o = ((size_t)b * c + ch) * n + p0 + pt;
      outs[o] = gt * s_tile[ch * 65 + pt] + (add ? add[o] : 0.f);
    }
  }
}

// out[b,c,i] = gate[b,c] * trilinear(V[b,c], coords[b,:,i]) + add[b,c,i]
__global__ __launch_bounds__(256) void devoxelize_fused_kernel(const float *__restrict__ coords,
                                                               const float *__restrict__ feat,
                                                               const float *__restrict__ gate,
                                                               const float *__restrict__ add, int c, int n, int r,
                                                               float *__restrict__ outs,
                                                               const float *__restrict__ coef) {
  // coef != NULL: `feat` is a raw conv output and GroupNorm + Swish are applied to the 8 corners on the fly
  // (swish(a f + s), (a, s) per cloud and channel: groupnorm_coef_kernel)
  const int b = blockIdx.z;
  const int r2 = r * r, r3 = r2 * r;
  coords += (size_t)b * 3 * n;
  feat += (size_t)b * c * r3;
  outs += (size_t)b * c * n;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float x = coords[i], y = coords[i + n], z = coords[i + 2 * n];
  const float xl = floorf(x), yl = floorf(y), zl = floorf(z);
  const float xd1 = x - xl, yd1 = y - yl, zd1 = z - zl;
  const float xd0 = 1.0f - xd1, yd0 = 1.0f - yd1, zd0 = 1.0f - zd1;
  const float w000 = xd0 * yd0 * zd0, w001 = xd0 * yd0 * zd1, w010 = xd0 * yd1 * zd0, w011 = xd0 * yd1 * zd1;
  const float w100 = xd1 * yd0 * zd0, w101 = xd1 * yd0 * zd1, w110 = xd1 * yd1 * zd0, w111 = xd1 * yd1 * zd1;
  const int i000 = (int)xl * r2 + (int)yl * r + (int)zl;
  const int zh = zd1 > 0 ? 1 : 0, yh = yd1 > 0 ? r : 0, xh = xd1 > 0 ? r2 : 0;
  const int i001 = i000 + zh, i010 = i000 + yh, i011 = i010 + zh;
  const int i100 = i000 + xh, i101 = i100 + zh, i110 = i100 + yh, i111 = i110 + zh;
  const int c0 = blockIdx.y * 16, c1 = min(c0 + 16, c);
  for (int l = c0; l < c1; ++l) {
    const float *f = feat + (size_t)l * r3;
    float f0 = f[i000], f1 = f[i001], f2 = f[i010], f3 = f[i011], f4 = f[i100], f5 = f[i101], f6 = f[i110], f7 = f[i111];
    if (coef) {
      const float a = coef[((size_t)b * c + l) * 2], s = coef[((size_t)b * c + l) * 2 + 1];
      f0 = swish_fast(fmaf(f0, a, s)); f1 = swish_fast(fmaf(f1, a, s)); f2 = swish_fast(fmaf(f2, a, s));
      f3 = swish_fast(fmaf(f3, a, s)); f4 = swish_fast(fmaf(f4, a, s)); f5 = swish_fast(fmaf(f5, a, s));
      f6 = swish_fast(fmaf(f6, a, s)); f7 = swish_fast(fmaf(f7, a, s));
    }
    const float v = w000 * f0 + w001 * f1 + w010 * f2 + w011 * f3 + w100 * f4 + w101 * f5 + w110 * f6 + w111 * f7;
    const float gt = gate ? gate[(size_t)b * c + l] : 1.0f;
    const float ad = add ? add[((size_t)b * c + l) * n + i] : 0.f;
    outs[(size_t)l * n + i] = gt * v + ad;
  }
}

template <int MT, int NTW, int JN>
int launch_conv_jn(const float *x, const float *wp, const float *bias, int b, int cin, int cout, int r, float *y,
                float *partial, hipStream_t s, int cout_total, int co0, int out_cl) {
  const size_t lds_bytes = (size_t)16 * brick_row_stride(r) * sizeof(float);
  struct Tag {};  // one flag array per instantiation
  gldm_dev::allow_dynamic_lds<Tag>(reinterpret_cast<const void *>(&conv3d_k3_kernel<MT, NTW, JN>), (int)lds_bytes);
  const int bpr = r / kBrick;
  hipLaunchKernelGGL((conv3d_k3_kernel<MT, NTW, JN>), dim3(bpr * bpr, b), dim3(kConvThreads), lds_bytes, s, x, wp, bias,
                     cin, cout, r, y, partial, cout_total, co0, out_cl);
  return hipGetLastError() == hipSuccess ? GLDM_OK : GLDM_ERR_LAUNCH;
}

// y[b, c, :] = act(y[b, c, :] + bias[c]) in place: the epilogue of the k = 1 convs that run as plain
// library GEMMs (one pass instead of a bias pass and an activation pass).
__global__ __launch_bounds__(256) void bias_act_kernel(float *__restrict__ y, const float *__restrict__ bias, int c,
                                                       long long n, int relu) {
  const long long row = blockIdx.y;  // b * c + channel
  const float bv = bias[row % c];
  float *p = y + row * n;
  const long long n4 = n >> 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    float4 v = reinterpret_cast<float4 *>(p)[i];
    v.x += bv; v.y += bv; v.z += bv; v.w += bv;
    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    reinterpret_cast<float4 *>(p)[i] = v;
  }
}

// ---- narrow k = 1 convs and the Linear over the point axis (the pieces of the shipped encoder that used to go to
// MIOpen / rocBLAS: SharedMLP 3 -> 48 and 48 -> 96 of the PVConv point branches, shared_mlp.py:6-35, and
// out_layer[1] = Linear(n_points -> latent) over the POINT axis, pc_encoders.py:60-82,104-111).  Too small for the
// matrix pipe (<= 4.6 k MAC per point): lane = point, the point's cin inputs in registers, weights wave-uniform on the
// scalar path, fma chain in k order from the bias.
template <int CIN>
__global__ __launch_bounds__(256) void pointwise_small_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                              const float *__restrict__ bias, int cout, long long n,
                                                              int relu, float *__restrict__ y) {
  const int b = blockIdx.y;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float *xb = x + (size_t)b * CIN * n + i;
  float v[CIN];
#pragma unroll
  for (int ci = 0; ci < CIN; ++ci) v[ci] = xb[(size_t)ci * n];
  float *yb = y + (size_t)b * cout * n + i;
  // eight output channels at a time: eight independent fma chains per lane (one chain per pass left the vector pipe waiting
  // on its own result: 0.106 ms for 48 -> 96 over 256 x 1024 points, three times its instruction count)
  constexpr int kCh = 8;
  int co = 0;
  for (; co + kCh <= cout; co += kCh) {
    const float *wr = w + (size_t)co * CIN;
    float acc[kCh];
#pragma unroll
    for (int k = 0; k < kCh; ++k) acc[k] = bias ? bias[co + k] : 0.f;
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
      for (int k = 0; k < kCh; ++k) acc[k] = fmaf(wr[k * CIN + ci], v[ci], acc[k]);
#pragma unroll
    for (int k = 0; k < kCh; ++k) yb[(size_t)(co + k) * n] = relu ? fmaxf(acc[k], 0.f) : acc[k];
  }
  for (; co < cout; ++co) {
    const float *wr = w + (size_t)co * CIN;
    float acc = bias ? bias[co] : 0.f;
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci) acc = fmaf(wr[ci], v[ci], acc);
    yb[(size_t)co * n] = relu ? fmaxf(acc, 0.f) : acc;
  }
}

// ---- any-shape k = 1 conv (SharedMLP / feature-propagation layers outside the fused launches' shape sets) --------------
// y[b, co, i] = act(bias[co] + sum_ci W[co][ci] x[b, ci, i]) for ANY (cin, cout, n), weights as stored by nn.Conv1d
// ([cout][cin], BatchNorm folded by the caller): exact f32 products on v_mfma_f32_16x16x4_f32.  A 256-thread workgroup
// owns a 64-row x 64-point output tile (wave = m-tile, four n-tiles); K is staged 16 channels at a time through LDS with
// bounds masks (W rows padded to 17 words, x rows to 80: both fragment reads conflict free).  Replaces the library GEMM
// (rocBLAS / MIOpen through F.conv1d) + bias / activation pass these layers used to take: PointNet++ / PVCNN2 widths such
// as 384 -> 256 over 128 centres.  Not a speed-of-light kernel (one 16-deep stage per barrier pair); the shipped encoder
// never comes here.
__global__ __launch_bounds__(256) void pointwise_any_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                            const float *__restrict__ bias, int cin, int cout, long long n,
                                                            int relu, float *__restrict__ y) {
  __shared__ float Ws[64 * 17];
  __shared__ float Xs[16 * 80];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, col = lane & 15, kq = lane >> 4;
  const long long c0 = (long long)blockIdx.x * 64;
  const int r0 = blockIdx.y * 64, b = blockIdx.z;
  x += (size_t)b * cin * n;
  y += (size_t)b * cout * n;
  f32x4 acc[4];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) acc[ni] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int wr = tid >> 2, wk = (tid & 3) * 4;          // W stage: row, first k of the thread's four
  const int xk = tid >> 4, xc = (tid & 15) * 4;         // x stage: channel, first point of the thread's four
  for (int k0 = 0; k0 < cin; k0 += 16) {
    float wv[4], xv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bool okw = r0 + wr < cout && k0 + wk + q < cin;
      wv[q] = okw ? w[(size_t)(r0 + wr) * cin + k0 + wk + q] : 0.f;
      const bool okx = k0 + xk < cin && c0 + xc + q < n;
      xv[q] = okx ? x[(size_t)(k0 + xk) * n + c0 + xc + q] : 0.f;
    }
    __syncthreads();   // the previous stage's readers are done
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      Ws[wr * 17 + wk + q] = wv[q];
      Xs[xk * 80 + xc + q] = xv[q];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float a = Ws[(16 * wave + col) * 17 + 4 * j + kq];
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
        acc[ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, Xs[(4 * j + kq) * 80 + 16 * ni + col], acc[ni], 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = r0 + 16 * wave + 4 * kq + r;
    if (row < cout) {
      const float bv = bias ? bias[row] : 0.f;
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const long long i = c0 + 16 * ni + col;
        if (i < n) {
          const float v = acc[ni][r] + bv;
          y[(size_t)row * n + i] = relu ? fmaxf(v, 0.f) : v;
        }
      }
    }
  }
}

// y[row, o] = bias[o] + sum_n W[o, n] x[row, n]: one workgroup per row (rows = batch x channels: a few hundred),
// the row staged in LDS, thread = output feature, four interleaved k-ordered fma chains.
__global__ __launch_bounds__(256) void linear_rows_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                          const float *__restrict__ bias, int n, int nout,
                                                          float *__restrict__ y) {
  extern __shared__ float xs[];
  const int row = blockIdx.x;
  for (int i = threadIdx.x; i < n; i += 256) xs[i] = x[(size_t)row * n + i];
  __syncthreads();
  for (int o = threadIdx.x; o < nout; o += 256) {
    const float *wr = w + (size_t)o * n;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;   // four interleaved chains: shorter dependency and error chains
    for (int i = 0; i < n; i += 4) {
      const float4 wv = *reinterpret_cast<const float4 *>(wr + i);
      a0 = fmaf(wv.x, xs[i], a0);
      a1 = fmaf(wv.y, xs[i + 1], a1);
      a2 = fmaf(wv.z, xs[i + 2], a2);
      a3 = fmaf(wv.w, xs[i + 3], a3);
    }
    y[(size_t)row * nout + o] = ((a0 + a1) + (a2 + a3)) + (bias ? bias[o] : 0.f);
  }
}

// ---- any-shape Conv3d(k = 3, p = 1) for voxel shapes without an MFMA instantiation (PVCNN2's 256 ch @ 8^3 and
// 128 ch @ 16^3 feature-propagation convs: 16 m-tiles of accumulators do not fit a wave).  Direct form on the VALU: a
// workgroup owns a 4 x 4 x r brick like the MFMA kernels (so the GroupNorm partials have the same layout), a thread a
// voxel, weights [cout][cin][27] as stored by nn.Conv3d, wave-uniform on the scalar path.  Correctness path, not a fast
// one: these shapes are outside the shipped encoder.
__global__ __launch_bounds__(256) void conv3d_k3_generic_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                                const float *__restrict__ bias, int cin, int cout, int r,
                                                                float *__restrict__ y, float *__restrict__ partial) {
  __shared__ float s_red[2][4];
  const int bpr = (r + kBrick - 1) / kBrick, r3 = r * r * r, nvox = 16 * r;   // any r: the last brick row / column may be partial
  const int bx0 = (blockIdx.x / bpr) * kBrick, by0 = (blockIdx.x % bpr) * kBrick, b = blockIdx.y;
  x += (size_t)b * cin * r3;
  y += (size_t)b * cout * r3;
  const int tid = threadIdx.x;
  for (int co = 0; co < cout; ++co) {
    float s = 0.f, s2 = 0.f;
    for (int v = tid; v < nvox; v += 256) {
      const int iz = v % r, ixy = v / r, gx = bx0 + (ixy >> 2), gy = by0 + (ixy & 3);
      if (gx >= r || gy >= r) continue;
      float acc = bias[co];
      for (int ci = 0; ci < cin; ++ci) {
        const float *xc = x + (size_t)ci * r3;
        const float *wc = w + ((size_t)co * cin + ci) * 27;
#pragma unroll
        for (int tap = 0; tap < 27; ++tap) {
          const int nx = gx + tap / 9 - 1, ny = gy + (tap / 3) % 3 - 1, nz = iz + tap % 3 - 1;
          if ((unsigned)nx < (unsigned)r && (unsigned)ny < (unsigned)r && (unsigned)nz < (unsigned)r)
            acc = fmaf(wc[tap], xc[(nx * r + ny) * r + nz], acc);
        }
      }
      y[(size_t)co * r3 + (gx * r + gy) * r + iz] = acc;
      s += acc;
      s2 += acc * acc;
    }
    for (int off = 32; off >= 1; off >>= 1) {
      s += __shfl_xor(s, off, 64);
      s2 += __shfl_xor(s2, off, 64);
    }
    __syncthreads();
    if ((tid & 63) == 0) { s_red[0][tid >> 6] = s; s_red[1][tid >> 6] = s2; }
    __syncthreads();
    if (tid == 0) {
      float *p = partial + (((size_t)b * gridDim.x + blockIdx.x) * cout + co) * 2;
      p[0] = s_red[0][0] + s_red[0][1] + s_red[0][2] + s_red[0][3];
      p[1] = s_red[1][0] + s_red[1][1] + s_red[1][2] + s_red[1][3];
    }
  }
}

}  // namespace

GLDM_API long long gldm_conv3d_partial_floats(int b, int cout, int r) {
  if (b <= 0 || cout <= 0 || r <= 0) return -1;
  const long long bpr = (r + kBrick - 1) / kBrick;   // a resolution that is not a multiple of the brick has partial edge bricks
  return (long long)b * bpr * bpr * cout * 2;
}

template <int MT, int NTW>
int launch_conv(const float *x, const float *wp, const float *bias, int b, int cin, int cout, int r, float *y,
                float *partial, hipStream_t s, int out_cl, int cout_total = -1, int co0 = 0) {
  if (cout_total < 0) cout_total = cout;
  // a 3-channel input (the first voxel conv) has one real k-step per tap: skip the three of padding
  return cin <= 4 ? launch_conv_jn<MT, NTW, 1>(x, wp, bias, b, cin, cout, r, y, partial, s, cout_total, co0, out_cl)
                  : launch_conv_jn<MT, NTW, 4>(x, wp, bias, b, cin, cout, r, y, partial, s, cout_total, co0, out_cl);
}

// a conv of 2 MH m-tiles as two launches of MH (the accumulators of 16 m-tiles do not fit a wave)
template <int MH, int NTW>
int launch_conv_halves(const float *x, const float *wp, const float *bias, int b, int cin, int cout, int r, float *y,
                       float *partial, hipStream_t s, int out_cl) {
  const size_t kblocks = 27 * (size_t)((cin + 15) >> 4);
  const int half = 16 * MH;
  const int rc = launch_conv<MH, NTW>(x, wp, bias, b, cin, half, r, y, partial, s, out_cl, cout, 0);
  if (rc != GLDM_OK) return rc;
  return launch_conv<MH, NTW>(x, wp + (size_t)MH * kblocks * 256, bias + half, b, cin, cout - half, r, y, partial, s, out_cl, cout, half);
}

static int conv3d_k3_impl(const float *x, const float *w_packed, const float *bias, int b, int cin, int cout, int r,
                          float *y, float *partial, int out_cl, gldm_stream_t stream) {
  if (!x || !w_packed || !bias || !y || !partial || b <= 0 || cin <= 0 || cout <= 0 || r <= 0)
    return GLDM_ERR_INVALID_ARG;
  if (out_cl && cout % 4) return GLDM_ERR_UNSUPPORTED;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int mt = (cout + 15) / 16, ntw = r / 4;
  if (r % 4 || (size_t)16 * brick_row_stride(r) * 4 > 160 * 1024) return GLDM_ERR_UNSUPPORTED;

#ifdef GLDM_DEBUG_KNOBS
  struct StampDump {  // diagnostic builds: GLDM_C3_STAMP=1 prints the phase clocks of one mid-grid workgroup per call
    hipStream_t s; int cin, cout, r;
    ~StampDump() {
      if (!getenv("GLDM_C3_STAMP")) return;
      long long h[32];
      (void)hipStreamSynchronize(s);
      (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_c3_stamp), sizeof(h));
      printf("conv3d %d->%d @%d: setup %lld, stage0 issue %lld", cin, cout, r, h[1] - h[0], h[2] - h[1]);
      for (int cb = 0; cb < (cin + 15) / 16 && cb < 3; ++cb)
        printf(" | cb%d: wait %lld store %lld barrier %lld taps %lld", cb, h[3 + 4 * cb] - (cb ? h[2 + 4 * cb] : h[2]),
               h[4 + 4 * cb] - h[3 + 4 * cb], h[5 + 4 * cb] - h[4 + 4 * cb], h[6 + 4 * cb] - h[5 + 4 * cb]);
      printf(" | epilogue %lld (stats %lld, barrier %lld, partials %lld, stores %lld), total %lld", h[21] - h[20], h[24] - h[20],
             h[25] - h[24], h[26] - h[25], h[21] - h[26], h[21] - h[0]);
      for (int cb = 0; cb < 2; ++cb)
        printf(" | cb%d taps: first %lld, second %lld, 2..13 %lld", cb, h[14 + 3 * cb] - h[5 + 4 * cb],
               h[15 + 3 * cb] - h[14 + 3 * cb], h[16 + 3 * cb] - h[15 + 3 * cb]);
      printf("\n");
    }
  } dump{s, cin, cout, r};
#endif
#define GLDM_CONV_CASE(M, N) \
  if (mt == M && ntw == N) return launch_conv<M, N>(x, w_packed, bias, b, cin, cout, r, y, partial, s, out_cl)
  GLDM_CONV_CASE(3, 6);   // 48 ch @ 24^3  (shipped fpc/ppc PVCNN encoder)
  GLDM_CONV_CASE(6, 3);   // 96 ch @ 12^3
  GLDM_CONV_CASE(2, 8);   // 32 ch @ 32^3  (PVCNN2)
  GLDM_CONV_CASE(4, 4);   // 64 ch @ 16^3
  GLDM_CONV_CASE(8, 2);   // 128 ch @ 8^3
  // half-width / half-resolution variants (the reference's encoder benchmark runs PVCNN / PVCNN2 at 0.5 / 0.5)
  GLDM_CONV_CASE(2, 4);   // 32 ch @ 16^3
  GLDM_CONV_CASE(4, 2);   // 64 ch @ 8^3
  GLDM_CONV_CASE(1, 4);   // 16 ch @ 16^3
  GLDM_CONV_CASE(2, 2);   // 32 ch @ 8^3
  GLDM_CONV_CASE(4, 1);   // 64 ch @ 4^3
  GLDM_CONV_CASE(8, 1);   // 128 ch @ 4^3
#undef GLDM_CONV_CASE
  // 64 ch @ 32^3 (PVCNN2 at full width under numerics.f32_only(): the split kernel serves it otherwise): 4 m-tiles x 8
  // n-tiles of accumulators spilled 58-70 registers as one launch -- two launches of the 32-channel instantiation instead
  if (mt == 4 && ntw == 8) return launch_conv_halves<2, 8>(x, w_packed, bias, b, cin, cout, r, y, partial, s, out_cl);
  if (mt == 16 && ntw == 2) return launch_conv_halves<8, 2>(x, w_packed, bias, b, cin, cout, r, y, partial, s, out_cl);  // 256 ch @ 8^3
  if (mt == 8 && ntw == 4) return launch_conv_halves<4, 4>(x, w_packed, bias, b, cin, cout, r, y, partial, s, out_cl);   // 128 ch @ 16^3
  return GLDM_ERR_UNSUPPORTED;
}

GLDM_API int gldm_conv3d_k3(const float *x, const float *w_packed, const float *bias, int b, int cin, int cout, int r,
                            float *y, float *partial, gldm_stream_t stream) {
  return conv3d_k3_impl(x, w_packed, bias, b, cin, cout, r, y, partial, 0, stream);
}

GLDM_API int gldm_conv3d_k3_cl(const float *x, const float *w_packed, const float *bias, int b, int cin, int cout, int r,
                               float *y_cl, float *partial, gldm_stream_t stream) {
  return conv3d_k3_impl(x, w_packed, bias, b, cin, cout, r, y_cl, partial, 1, stream);
}

template <int MT, int R, int ZB, int WAVES, bool ACT>
int launch_conv_pl_act(const float *x, const float *wp3, const float *bias, int b, int cin, int cout, float *y, float *partial,
                       const float *in_coef, int out_cl, hipStream_t s) {
  const size_t lds_bytes = (size_t)(R == 24 && ZB == 24 ? 2 : 1) * kC3Split * 2 * ((36 * (ZB + 2) + 15) & ~15) * 16 +
                           (ACT ? (size_t)2 * cin * sizeof(float) : 64 /* the waves' range words */);
  struct Tag {};
  gldm_dev::allow_dynamic_lds<Tag>(reinterpret_cast<const void *>(&conv3d_k3_pl_kernel<MT, R, ZB, WAVES, ACT>), (int)lds_bytes);
  const int bpr = R / kBrick;
  if (R != ZB) {   // the halves of a brick add their partials into one slot
    if (hipMemsetAsync(partial, 0, (size_t)b * bpr * bpr * cout * 2 * sizeof(float), s) != hipSuccess) return GLDM_ERR_LAUNCH;
  }
  hipLaunchKernelGGL((conv3d_k3_pl_kernel<MT, R, ZB, WAVES, ACT>), dim3(bpr * bpr * (R / ZB), b), dim3(64 * WAVES), lds_bytes,
                     s, x, wp3, bias, cin, cout, y, partial, in_coef, out_cl);
  return hipGetLastError() == hipSuccess ? GLDM_OK : GLDM_ERR_LAUNCH;
}
template <int MT, int R, int ZB, int WAVES>
int launch_conv_pl(const float *x, const float *wp3, const float *bias, int b, int cin, int cout, float *y, float *partial,
                   const float *in_coef, int out_cl, hipStream_t s) {
  return in_coef ? launch_conv_pl_act<MT, R, ZB, WAVES, true>(x, wp3, bias, b, cin, cout, y, partial, in_coef, out_cl, s)
                 : launch_conv_pl_act<MT, R, ZB, WAVES, false>(x, wp3, bias, b, cin, cout, y, partial, nullptr, out_cl, s);
}

static int conv3d_k3_f16x2_impl(const float *x, const float *in_coef, const float *w_split, const float *bias, int b, int cin,
                                 int cout, int r, float *y, float *partial, int out_cl, gldm_stream_t stream) {
  if (!x || !w_split || !bias || !y || !partial || b <= 0 || cin <= 0 || cout <= 0 || r <= 0) return GLDM_ERR_INVALID_ARG;
  if ((in_coef || out_cl) && cin % 16) return GLDM_ERR_UNSUPPORTED;   // both live in the plane-staging kernels
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (cin == 3 && cout == 48 && r == 24) {   // the first voxel conv: K = 81 packed into three k-blocks
    constexpr int kZp = 26, kNvox = 36 * kZp;
    const size_t lds_bytes = (size_t)3 * (kNvox + ((kNvox & 31) == 0 ? 8 : 0)) * sizeof(float) + 64;   // + the waves' range words
    hipLaunchKernelGGL((conv3d_k3_fewch_sp_kernel<3, 24, 3, 8>), dim3(36, b), dim3(512), lds_bytes, s, x, w_split, bias, y, partial);
    return hipGetLastError() == hipSuccess ? GLDM_OK : GLDM_ERR_LAUNCH;
  }
  if (cin % 16) return GLDM_ERR_UNSUPPORTED;   // the kernels below: cout == 16 MT exactly, no row guards
#ifdef GLDM_DEBUG_KNOBS
  struct StampDump {  // diagnostic builds: GLDM_C3_STAMP=1 prints the phase clocks of one mid-grid workgroup per call
    hipStream_t s; int cin, cout, r, b;
    ~StampDump() {
      if (!getenv("GLDM_C3_STAMP")) return;
      long long h[32];
      (void)hipStreamSynchronize(s);
      (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_c3_stamp), sizeof(h));
      printf("conv3d split %d->%d @%d b=%d: setup %lld, stage0 issue %lld", cin, cout, r, b, h[1] - h[0], h[2] - h[1]);
      for (int cb = 0; cb < cin / 16 && cb < 3; ++cb)
        printf(" | cb%d: wait %lld store %lld barrier %lld taps %lld", cb, h[3 + 4 * cb] - (cb ? h[2 + 4 * cb] : h[2]),
               h[4 + 4 * cb] - h[3 + 4 * cb], h[5 + 4 * cb] - h[4 + 4 * cb], h[6 + 4 * cb] - h[5 + 4 * cb]);
      printf(" | epilogue %lld, total %lld\n", h[21] - h[20], h[21] - h[0]);
    }
  } dump{s, cin, cout, r, b};
#endif
  // (4 x 4 x 12 half bricks at 24^3 -- <3, 24, 12, 4>: 48 KiB of planes, two co-resident workgroups of 4 waves -- measured
  // 2.12 ms against 2.05 for the full-z brick: the kernel's 220 registers allow two waves per SIMD either way, and the
  // halves pay a z halo and the weight stream twice.  Kept as an instantiable option, not used.  Round 6, with three f16
  // products instead of six bf16 ones: A/B of the whole encoder on one box 4.69 / 4.72 ms (full brick) against 4.63 / 4.84
  // (halves) -- inside the run-to-run spread, still not used.)
  // Round 6, two more forms of the 24^3 conv measured on one box against this one (whole shipped encoder, 256 clouds, A/B by an
  // environment switch of the diagnostic build; phase stamps of a mid-grid workgroup: taps 31 k of a brick's 70 k cycles,
  // store phases 2.4-5.2 k and barrier waits 2.9-3.8 k per 16-channel block, epilogue 5.4 k):
  //  * block cb + 1 split and stored INSIDE block cb's tap loop, one staging round behind every second tap pair (the other
  //    plane set is free there): the store phases shrank to 0.3-1.1 k, the taps grew 10.4 -> 12.2 k and the barrier waits
  //    3.5 -> 6-7 k -- 4.49-4.50 ms against 4.48: nothing.  The staging VALU work does not hide under the other wave's MFMAs;
  //  * <3, 24, 24, 4>: four waves x six n-tiles (half the weight-fragment deliveries per MFMA), two workgroups per CU, no
  //    staging prefetch across the tap loop (64 registers): 18-24 spilled registers, 4.58 against 4.63 ms: 1 %, not kept.
  if (cout == 48 && r == 24) return launch_conv_pl<3, 24, 24, 8>(x, w_split, bias, b, cin, cout, y, partial, in_coef, out_cl, s);
  if (cout == 96 && r == 12) return launch_conv_pl<6, 12, 12, 4>(x, w_split, bias, b, cin, cout, y, partial, in_coef, out_cl, s);
  // PVCNN2's power-of-two shapes (round 5; f32-MFMA kernels before): <MT, R, ZB, WAVES>
  if (cout == 32 && r == 32) return launch_conv_pl<2, 32, 32, 8>(x, w_split, bias, b, cin, cout, y, partial, in_coef, out_cl, s);
  if (cout == 64 && r == 32) return launch_conv_pl<4, 32, 32, 8>(x, w_split, bias, b, cin, cout, y, partial, in_coef, out_cl, s);
  if (cout == 32 && r == 16) return launch_conv_pl<2, 16, 16, 4>(x, w_split, bias, b, cin, cout, y, partial, in_coef, out_cl, s);
  if (cout == 64 && r == 16) return launch_conv_pl<4, 16, 16, 4>(x, w_split, bias, b, cin, cout, y, partial, in_coef, out_cl, s);
  if (cout == 128 && r == 16) return launch_conv_pl<8, 16, 16, 8>(x, w_split, bias, b, cin, cout, y, partial, in_coef, out_cl, s);
  if (cout == 64 && r == 8) return launch_conv_pl<4, 8, 8, 4>(x, w_split, bias, b, cin, cout, y, partial, in_coef, out_cl, s);
  if (cout == 128 && r == 8) return launch_conv_pl<8, 8, 8, 4>(x, w_split, bias, b, cin, cout, y, partial, in_coef, out_cl, s);
  if (cout == 256 && r == 8) return launch_conv_pl<16, 8, 8, 8>(x, w_split, bias, b, cin, cout, y, partial, in_coef, out_cl, s);
  if (cout == 128 && r == 4) return launch_conv_pl<8, 4, 4, 4>(x, w_split, bias, b, cin, cout, y, partial, in_coef, out_cl, s);
  return GLDM_ERR_UNSUPPORTED;
}

GLDM_API int gldm_conv3d_k3_f16x2(const float *x, const float *w_split, const float *bias, int b, int cin, int cout, int r,
                                   float *y, float *partial, gldm_stream_t stream) {
  return conv3d_k3_f16x2_impl(x, nullptr, w_split, bias, b, cin, cout, r, y, partial, 0, stream);
}

GLDM_API int gldm_conv3d_k3_f16x2_gn(const float *x, const float *in_coef, const float *w_split, const float *bias, int b,
                                      int cin, int cout, int r, float *y, float *partial, int out_channel_last,
                                      gldm_stream_t stream) {
  return conv3d_k3_f16x2_impl(x, in_coef, w_split, bias, b, cin, cout, r, y, partial, out_channel_last ? 1 : 0, stream);
}

GLDM_API int gldm_groupnorm_coef(const float *partial, const float *gamma, const float *beta, int b, int c, int r, int groups,
                                 float eps, float *coef, gldm_stream_t stream) {
  if (!partial || !gamma || !beta || !coef || b <= 0 || c <= 0 || r <= 0 || groups <= 0 || c % groups || c / groups > 64)
    return GLDM_ERR_INVALID_ARG;
  const int nbricks = ((r + kBrick - 1) / kBrick) * ((r + kBrick - 1) / kBrick);
  hipLaunchKernelGGL(groupnorm_coef_kernel, dim3(groups, b), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), partial, gamma,
                     beta, c, r * r * r, nbricks, groups, eps, coef);
  return hipGetLastError() == hipSuccess ? GLDM_OK : GLDM_ERR_LAUNCH;
}

GLDM_API int gldm_gn_swish_chan_sum(const float *y, const float *coef, int b, int c, int r, float *chan_sum,
                                    gldm_stream_t stream) {
  if (!y || !coef || !chan_sum || b <= 0 || c <= 0 || r <= 0) return GLDM_ERR_INVALID_ARG;
  hipLaunchKernelGGL(gn_swish_sum_kernel, dim3(c, b), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), y, coef, c,
                     r * r * r, chan_sum);
  return hipGetLastError() == hipSuccess ? GLDM_OK : GLDM_ERR_LAUNCH;
}

GLDM_API int gldm_groupnorm_swish(float *y, const float *partial, const float *gamma, const float *beta, int b, int c,
                                  int r, int groups, float eps, float *chan_sum, gldm_stream_t stream) {
  if (!y || !partial || !gamma || !beta || b <= 0 || c <= 0 || r <= 0 || groups <= 0 || c % groups)
    return GLDM_ERR_INVALID_ARG;
  const int nbricks = ((r + kBrick - 1) / kBrick) * ((r + kBrick - 1) / kBrick);
  hipLaunchKernelGGL(groupnorm_swish_kernel, dim3(groups, b), dim3(512), 0, reinterpret_cast<hipStream_t>(stream), y,
                     partial, gamma, beta, c, r * r * r, nbricks, groups, eps, chan_sum);
  return hipGetLastError() == hipSuccess ? GLDM_OK : GLDM_ERR_LAUNCH;
}

GLDM_API int gldm_se_gate(const float *chan_sum, const float *w1, const float *w2, int b, int c, int hidden, int r,
                          int use_relu, float *gate, gldm_stream_t stream) {
  if (!chan_sum || !w1 || !w2 || !gate || b <= 0 || c <= 0 || hidden <= 0 || r <= 0) return GLDM_ERR_INVALID_ARG;
  hipLaunchKernelGGL(se_gate_kernel, dim3(b), dim3(128), (size_t)(c + hidden) * sizeof(float),
                     reinterpret_cast<hipStream_t>(stream), chan_sum, w1, w2, c, hidden, r * r * r, use_relu, gate, 1);
  return hipGetLastError() == hipSuccess ? GLDM_OK : GLDM_ERR_LAUNCH;
}

GLDM_API int gldm_se_gate_parts(const float *chan_parts, int parts, const float *w1, const float *w2, int b, int c, int hidden,
                                int r, int use_relu, float *gate, gldm_stream_t stream) {
  if (!chan_parts || !w1 || !w2 || !gate || b <= 0 || c <= 0 || hidden <= 0 || r <= 0 || parts <= 0) return GLDM_ERR_INVALID_ARG;
  hipLaunchKernelGGL(se_gate_kernel, dim3(b), dim3(128), (size_t)(c + hidden) * sizeof(float),
                     reinterpret_cast<hipStream_t>(stream), chan_parts, w1, w2, c, hidden, r * r * r, use_relu, gate, parts);
  return hipGetLastError() == hipSuccess ? GLDM_OK : GLDM_ERR_LAUNCH;
}

GLDM_API int gldm_squeeze_parts(void) { return kSumParts; }

GLDM_API int gldm_gn_swish_chan_sum_cl(const float *y, const float *coef, int b, int c, int r, float *chan_parts,
                                       gldm_stream_t stream) {
  if (!y || !coef || !chan_parts || b <= 0 || c <= 0 || r <= 0) return GLDM_ERR_INVALID_ARG;
  if (c % 4 || c > 1024) return GLDM_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(gn_swish_sum_cl_kernel, dim3(kSumParts, b), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), y, coef, c,
                     r * r * r, chan_parts);
  return hipGetLastError() == hipSuccess ? GLDM_OK : GLDM_ERR_LAUNCH;
}

GLDM_API int gldm_devoxelize_gn_cl_fused(const float *coords, const float *features_cl, const float *coef, const float *gate,
                                         const float *add, int b, int c, int n, int r, float *out, gldm_stream_t stream) {
  if (!coords || !features_cl || !coef || !out || b <= 0 || c <= 0 || n <= 0 || r <= 0) return GLDM_ERR_INVALID_ARG;
  if (c % 4 || c > 256) return GLDM_ERR_UNSUPPORTED;
  struct DevoxClTag { int site; };
  gldm_dev::allow_dynamic_lds<DevoxClTag>(reinterpret_cast<const void *>(&devoxelize_cl_kernel), 256 * 65 * (int)sizeof(float));
  hipLaunchKernelGGL(devoxelize_cl_kernel, dim3((n + 63) / 64, b), dim3(256), (size_t)c * 65 * sizeof(float),
                     reinterpret_cast<hipStream_t>(stream), coords, features_cl, coef, gate, add, c, n, r, out);
  return hipGetLastError() == hipSuccess ? GLDM_OK : GLDM_ERR_LAUNCH;
}

GLDM_API int gldm_devoxelize_fused(const float *coords, const float *features, const float *gate, const float *add,
                                   int b, int c, int n, int r, float *out, gldm_stream_t stream) {
  if (!coords || !features || !out || b <= 0 || c <= 0 || n <= 0 || r <= 0) return GLDM_ERR_INVALID_ARG;
  hipLaunchKernelGGL(devoxelize_fused_kernel, dim3((n + 255) / 256, (c + 15) / 16, b), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), coords, features, gate, add, c, n, r, out, nullptr);
  return hipGetLastError() == hipSuccess ? GLDM_OK : GLDM_ERR_LAUNCH;
}

GLDM_API int gldm_devoxelize_gn_fused(const float *coords, const float *features, const float *coef, const float *gate,
                                      const float *add, int b, int c, int n, int r, float *out, gldm_stream_t stream) {
  if (!coords || !features || !coef || !out || b <= 0 || c <= 0 || n <= 0 || r <= 0) return GLDM_ERR_INVALID_ARG;
  hipLaunchKernelGGL(devoxelize_fused_kernel, dim3((n + 255) / 256, (c + 15) / 16, b), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), coords, features, gate, add, c, n, r, out, coef);
  return hipGetLastError() == hipSuccess ? GLDM_OK : GLDM_ERR_LAUNCH;
}

GLDM_API int gldm_conv3d_k3_generic(const float *x, const float *w, const float *bias, int b, int cin, int cout, int r,
                                    float *y, float *partial, gldm_stream_t stream) {
  if (!x || !w || !bias || !y || !partial || b <= 0 || cin <= 0 || cout <= 0 || r <= 0) return GLDM_ERR_INVALID_ARG;
  const int bpr = (r + kBrick - 1) / kBrick;
  hipLaunchKernelGGL(conv3d_k3_generic_kernel, dim3(bpr * bpr, b), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, w,
                     bias, cin, cout, r, y, partial);
  return hipGetLastError() == hipSuccess ? GLDM_OK : GLDM_ERR_LAUNCH;
}

GLDM_API int gldm_pointwise_small(const float *x, const float *w, const float *bias, int b, int cin, int cout, long long n,
                                  int relu, float *y, gldm_stream_t stream) {
  if (!x || !w || !y || b <= 0 || cin <= 0 || cout <= 0 || n <= 0) return GLDM_ERR_INVALID_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)((n + 255) / 256), b);
#define GLDM_PS_CASE(C) \
  if (cin == C) { hipLaunchKernelGGL(pointwise_small_kernel<C>, grid, dim3(256), 0, s, x, w, bias, cout, n, relu, y); \
                  return hipGetLastError() == hipSuccess ? GLDM_OK : GLDM_ERR_LAUNCH; }
  GLDM_PS_CASE(3) GLDM_PS_CASE(6) GLDM_PS_CASE(16) GLDM_PS_CASE(24) GLDM_PS_CASE(32) GLDM_PS_CASE(48) GLDM_PS_CASE(64)
#undef GLDM_PS_CASE
  return GLDM_ERR_UNSUPPORTED;
}

GLDM_API int gldm_pointwise_any(const float *x, const float *w, const float *bias, int b, int cin, int cout, long long n,
                                int relu, float *y, gldm_stream_t stream) {
  if (!x || !w || !y || b <= 0 || cin <= 0 || cout <= 0 || n <= 0) return GLDM_ERR_INVALID_ARG;
  const long long tiles = (n + 63) / 64;
  if (tiles > 0x7fffffffLL || b > 65535 || (cout + 63) / 64 > 65535) return GLDM_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(pointwise_any_kernel, dim3((unsigned)tiles, (cout + 63) / 64, b), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), x, w, bias, cin, cout, n, relu, y);
  return hipGetLastError() == hipSuccess ? GLDM_OK : GLDM_ERR_LAUNCH;
}

GLDM_API int gldm_linear_rows(const float *x, const float *w, const float *bias, int rows, int n, int nout, float *y,
                              gldm_stream_t stream) {
  if (!x || !w || !y || rows <= 0 || n <= 0 || nout <= 0) return GLDM_ERR_INVALID_ARG;
  if ((n & 3) || (size_t)n * 4 > 64 * 1024) return GLDM_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(linear_rows_kernel, dim3(rows), dim3(256), (size_t)n * sizeof(float),
                     reinterpret_cast<hipStream_t>(stream), x, w, bias, n, nout, y);
  return hipGetLastError() == hipSuccess ? GLDM_OK : GLDM_ERR_LAUNCH;
}

GLDM_API int gldm_bias_act(float *y, const float *bias, int b, int c, long long n, int relu, gldm_stream_t stream) {
  if (!y || !bias || b <= 0 || c <= 0 || n <= 0) return GLDM_ERR_INVALID_ARG;
  if (n & 3) return GLDM_ERR_UNSUPPORTED;  // rows must stay 16-byte aligned
  const long long n4 = n >> 2;
  const int bx = (int)((n4 + 255) / 256 < 1 ? 1 : ((n4 + 255) / 256 > 64 ? 64 : (n4 + 255) / 256));
  hipLaunchKernelGGL(bias_act_kernel, dim3(bx, b * c), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), y, bias, c, n,
                     relu);
  return hipGetLastError() == hipSuccess ? GLDM_OK : GLDM_ERR_LAUNCH;
}
