// resnet1d.hip -- the fused 1-D ResNet engine of GraspLDM on gfx950:
//   * gldm_denoise : the whole reverse-diffusion loop (T steps of
//                    TimeConditionedResNet1D.forward + DDIM/DDPM update) in ONE
//                    launch; the latent never leaves the chip between steps;
//   * gldm_decode  : ConditionalGraspPoseDecoder.forward (in_layer, ResNet1D,
//                    tmrp / class_logits heads);
//   * gldm_sa_mlp_forward : the fused set-abstraction core (gather + grouped MLP + max)
//                    on the same GEMM core;
//   * gldm_r1d_cond_embed, gldm_pose_epilogue : the small ops either side.
//
// Mapping to CDNA4
//   A workgroup owns a tile of NC activation columns = NC/L samples x L positions and walks
//   every layer with activations resident in LDS as [channel][column] (XOR-swizzled so MFMA
//   B-fragment reads are conflict free).  Two geometries are built from one template:
//     NC = 32: 4 waves, 80 KiB LDS  -> TWO workgroups per CU.  The network is ~50 short
//              barrier-separated phases per step; a second, independent workgroup fills the
//              MFMA pipe while the first sits in a norm pass, a barrier or an L2 round trip.
//              This is the production geometry of the denoiser / decoder.
//     NC = 64: 8 waves, 150 KiB LDS -> one workgroup per CU (fused set abstraction: one centre
//              with U = 64 neighbours per tile).
//   Every conv / 1x1 is a GEMM  W[Cout x taps*Cin] * X[taps*Cin x NC]  on
//   v_mfma_f32_16x16x4_f32 (exact f32: the parity budget is 1e-4 on poses after 100 steps and
//   the reference's eps is dtype dependent, so no bf16 here).  Weights (standardised and laid
//   out in fragment order on the host) stream L2 -> VGPR as 16-byte coalesced global loads, a
//   few 16-channel blocks ahead; B operands are unconditional, batched LDS reads.  k = 3 convs
//   keep one accumulator set per tap on UNSHIFTED columns and apply the halo shift once to the
//   result tiles (lane shifts inside the 16-lane rows of the C/D layout), so the k-loop is
//   loads + MFMA only (tap-major: a tap's fragment registers are refilled right after its sweep).
//   GroupNorm, the time/condition scale-shift, SiLU and the residual add live in the conv epilogue
//   (statistics over the wave's own accumulators); LayerNorm / softmax run with lane = (row slot,
//   column) and reduce with DPP / permlane swaps.  The step is a tape of ~48 ops in LDS, interpreted
//   by a switch with every phase inlined (see run_tape).
//   LinearAttention at n = L is reassociated:  out = V (K^T Q)  (an L x L matrix per sample
//   and head) instead of (V K^T) Q: 8x fewer FLOPs, same math.
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "gldm.h"
#include "wstream.h"
#include "devstate.h"

#define GLDM_API extern "C" __attribute__((visibility("default")))

// Diagnostic knobs (phase skipping, per-op cycle stamps, workgroup stagger) exist only in builds made
// with -DGLDM_DEBUG_KNOBS (make EXTRA=-DGLDM_DEBUG_KNOBS); the shipped kernels contain none of them.
#ifdef GLDM_DEBUG_KNOBS
#define GLDM_SKIP(c, bit) ((c).skip & (bit))
#define GLDM_STAMPS(p) (p)
#else
#define GLDM_SKIP(c, bit) false
#define GLDM_STAMPS(p) ((long long *)nullptr)
#endif

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef __attribute__((address_space(3))) float lds_f;   // explicit LDS pointers: 32-bit ds_* addressing
typedef __attribute__((address_space(3))) f32x4 lds_f4;

constexpr int kHeads = 4, kDimHead = 32, kHidden = kHeads * kDimHead;  // LinearAttention defaults
constexpr int kMaxC = 256;
constexpr int kMaxSegs = 64;  // tiles (+ one spliced step segment) a persistent workgroup can be given

// Geometry and LDS map (floats) of one workgroup.  X: block input / residual stream, H: scratch.
template <int NC>
struct Geo {
  static constexpr int kWaves = NC / 8;           // 8 waves at 64 columns, 4 at 32
  static constexpr int kThreads = kWaves * 64;
  static constexpr int kNT = NC / 16;             // 16-column n-tiles
  static constexpr int kRP = 64 / NC;             // rows a wave touches per pass: lane = (sub, column)
  static constexpr int kSlots = kWaves * kRP;     // row slots of the norm passes
  static constexpr int kBufX = 0;
  static constexpr int kBufH = kMaxC * NC;
  static constexpr int kBufY = 128 * NC;          // attention: LayerNorm output, later to_out output
  static constexpr int kBufO = kBufH;             // attention output, 128 rows
  static constexpr int kBufQKV = kBufO + kHidden * NC;  // two heads of q,k,v: 192 rows
  static constexpr int kArena = kBufQKV + 192 * NC;
  static constexpr int kMiscLat = kArena;         // [NC] current latent row
  static constexpr int kMiscEps = kMiscLat + NC;
  static constexpr int kMiscG = kMiscEps + NC;    // [S][E] <= 320
  static constexpr int kMiscRed1 = kMiscG + 320;  // [kWaves][NC] cross-wave exchange slots (per wave and column)
  static constexpr int kMiscRed2 = kMiscRed1 + kWaves * NC;
  static constexpr int kMiscTape = kMiscRed2 + kWaves * NC;  // [kMaxOps][12] ints: the step program (+ its length)
  static constexpr int kMiscSegs = kMiscTape + 1024;         // [kMaxSegs][4] ints: this workgroup's (tile, s0, s1) list
  static constexpr int kMiscOld = kMiscSegs + 256;           // [NC] previous step's denoised row (DPM++ 2M)
  // 64-column engines: per-sample range of a ResnetBlock's H (conv_pm3_wave): [8 waves][16] published bounds, [16] scales
  static constexpr int kMiscHb = kMiscOld + NC;
  static constexpr int kMiscHs = kMiscHb + (NC == 64 ? 8 * 16 : 0);
  static constexpr int kMiscQ = kMiscHs + (NC == 64 ? 16 : 0);    // quad engine hand-shake words (quad_narrow.h): 8 + 4 x 64 ints
  static constexpr int kMiscQTab = kMiscQ + (NC == 64 ? 16 + 4 * 64 : 0);   // [2 x 380] byte offsets of the quad engines' weight streams (292 / 336 / 380 fragments)
  static constexpr int kLdsFloats = kMiscQTab + (NC == 64 ? 2 * 380 : 0);
};
static_assert(Geo<64>::kLdsFloats * 4 <= 160 * 1024, "LDS budget (1 WG/CU)");
static_assert(Geo<32>::kLdsFloats * 4 * 2 <= 160 * 1024, "LDS budget (2 WG/CU)");

// (Tried: XOR with row bit 0 ^ row bit 2, which also frees the accumulator stores (rows 4 kq + r) of their 2-way bank
// conflict.  The B-fragment reads of a k-block then need two base registers instead of one with immediate offsets,
// and every GEMM phase got 5-7 % slower.)
template <int NC>
__device__ __forceinline__ int swz(int row, int col) { return row * NC + (col ^ ((row & 1) << 4)); }

// Position-major engine (64 columns, split-f16 GEMMs): a B fragment of v_mfma_f32_16x16x32_f16 is rows 8 g + j
// (g = lane >> 4, j = 0..7) of one column per lane, so the four lane groups of a read sit 8 rows apart in the same
// columns.  XOR-ing the column's position tile with bits 3-4 of the row sends them to four different 16-bank groups:
// every B read is conflict free, with ONE lane base per tile (the XOR does not depend on j or on the 32-row block).
__device__ __forceinline__ int pswz(int row, int col) { return row * 64 + (col ^ (((row >> 3) & 3) << 4)); }

// Split operands (see "split-f16 GEMM core" below): every f32 value travels as kSplit = 2 f16 numbers, hi + lo.
constexpr int kSplit = 2;
constexpr int kFragBytes = kSplit * 1024;        // one weight fragment: [plane][lane 64][8 f16]

// Pre-split activation planes of the 64-column engines.  A tensor that is only ever read as a GEMM B operand is kept
// in LDS already split into its two f16 planes, in B-fragment order:
//   [32-channel block kb][plane hi|lo][g = 0..3][column 0..63][8 f16 = channels 32 kb + 8 g + 0..7]
// (8 KiB per 32 channels).  The producer's epilogue splits each element ONCE (its accumulators hold 4 consecutive
// channels of a column: one ds_write_b64 per plane); the eight consumer waves read a whole fragment plane with one
// ds_read_b128 per lane and their k-loops are loads + MFMA only.
constexpr int kPlaneH = 128 * 64;                 // H planes: floats [8192, 16384): 4 blocks of 32 channels
constexpr int kPlaneX = kPlaneH + 4 * kSplit * 1024;   // X planes: floats [16384, 24576)
constexpr int kPlaneMaxC = 128;
// The 256-channel level (only ever the last one: a down conv into it and one ResnetBlock) keeps ONE tensor in LDS as
// planes (8 blocks x 8 KiB = 64 KiB, floats [16384, 32768): behind the level's f32 residual stream X, rows 0..255 =
// floats [0, 16384)): the down conv writes the planes of the new residual stream X there (and X itself as f32 rows);
// conv1 reads the planes and, behind its GroupNorm exchange barrier (every wave is past its k-loop), overwrites them with
// the planes of H; conv2 reads those and adds act(GN(conv)) to the f32 rows for the final 1x1.
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
typedef __attribute__((address_space(3))) u32x4 lds_u4;
typedef __attribute__((address_space(3))) u32x2_t lds_u2;
// Geometry of the plane rows.  LL = 4 (position-major tiles of the 4-position denoiser): 64 columns per (block, plane, g)
// row.  LL = 16 (the 16-position nets: pose decoder, ppc denoiser; column = 4 * position + sample, 4 samples per tile):
// 72 entries per row, the 64 columns at entries 4 .. 67 between four ZERO entries on each side, so that the taps of a
// k = 3 conv are the same reads shifted by one position = 4 entries (entry 4 t + column for tap t), with no masks.
template <int LL>
struct PG {
  static constexpr int kCols = LL == 16 ? 72 : 64;   // 16-byte entries per row
  static constexpr int kOff = LL == 16 ? 4 : 0;      // entry of column 0
  static constexpr int kPlaneU4 = 4 * kCols;         // entries per plane of a 32-channel block
  static constexpr int kBlockU4 = kSplit * kPlaneU4; // per block
  static constexpr int kBlockFloats = 4 * kBlockU4;
  static constexpr int kH = 128 * 64;                // H planes (floats), 4 blocks
  static constexpr int kX = kH + 4 * kBlockFloats;   // X planes
  // the 256-channel level's one set, 8 blocks.  LL = 4: behind the 256 f32 rows of X; LL = 16: over both regions (its f32
  // rows 128 .. 255 lie over the first blocks: the residual stream is parked in global scratch, Ctx::park)
  static constexpr int kW = LL == 16 ? kH : 256 * 64;
  static constexpr int kEnd = kW + 8 * kBlockFloats;
};
static_assert(PG<4>::kH == kPlaneH && PG<4>::kX == kPlaneX, "plane regions");
static_assert(PG<4>::kEnd <= 512 * 64, "position-major planes end in front of the attention exchange slots");
static_assert(PG<16>::kEnd <= Geo<64>::kArena, "padded planes fit the arena");
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
// (a, b) -> packed f16 hi parts and packed f16 lo parts: x = hi + lo up to 2^-22 |x| (f16 subnormals are kept by the
// matrix pipe -- tools/micro/mfma_f16_split -- so small lo parts lose nothing but bits below 2^-25).
// v_cvt_pk_f16_f32 (round to nearest even), the remainders from the packed halves by v_fma_mix_f32, v_cvt_pk_f16_f32.
__device__ __forceinline__ void split_f16x2(float a, float b, unsigned &hi, unsigned &lo) {
  const f16x2 h = __builtin_convertvector(f32x2{a, b}, f16x2);
  const float ra = __builtin_fmaf((float)h[0], -1.0f, a), rb = __builtin_fmaf((float)h[1], -1.0f, b);
  hi = __builtin_bit_cast(unsigned, h);
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{ra, rb}, f16x2));
}
// rows c0 .. c0 + 3 (c0 % 4 == 0) of column n -> the two planes
template <int LL = 4>
__device__ __forceinline__ void store_planes4(float *planes, int c0, int n, float v0, float v1, float v2, float v3) {
  unsigned h0, h1, l0, l1;
  split_f16x2(v0, v1, h0, l0);
  split_f16x2(v2, v3, h1, l1);
  // dword address: (((kb * kSplit + plane) * 4 + g) * kCols + kOff + n) * 4 + 2 * (half of the 8-group)
  using G = PG<LL>;
  const int a = ((((c0 >> 5) * kSplit) * 4 + ((c0 >> 3) & 3)) * G::kCols + G::kOff + n) * 4 + ((c0 >> 2) & 1) * 2;
  lds_u2 *d = (lds_u2 *)(planes + a);
  d[0] = u32x2_t{h0, h1};
  d[2 * G::kPlaneU4] = u32x2_t{l0, l1};    // next plane: kPlaneU4 entries of 16 bytes = 2 kPlaneU4 u2
}

// ---- range scale of split operands ------------------------------------------------------------------------------------
// f16 carries 5 exponent bits: a value of 65520 or more has hi = inf (and lo = x - inf = NaN), one below 2^-14 a subnormal
// hi.  Where the DATA sets an operand's magnitude (a gathered neighbourhood, a cloud's features, the ReLU outputs behind
// them: BatchNorm is folded, so everything scales with the input) the tile is split as x / s with s a power of two chosen
// from the tile's largest magnitude (measured where the staged values sit in registers, a bound  R m + B  -- R the layer's
// largest row sum of |W|, B its largest |bias| -- for the hidden layers behind them) and s is folded back where the
// accumulators leave the matrix pipe: exact, wave uniform, three or four VALU instructions per tile and layer.  s = 1 for
// anything ordinary (2^-8 <= m < 2^14): every bit is then what it was without the scale.
__device__ __forceinline__ float range_pow2(float m) {   // m >= 0 (a maximum of magnitudes or a bound on one), wave uniform
  int e = (int)((__float_as_uint(m) >> 23) & 0xffu) - 127;   // floor(log2 m) of a normal m
  if ((e >= -8 && e < 14) || e < -100 || e > 100) return 1.0f;   // ordinary; nothing there; beyond rescue (inf / nan included)
  e = e < -40 ? -40 : e;                                     // biases divided by s stay finite
  return __uint_as_float((unsigned)(e - 13 + 127) << 23);    // m / s in [2^13, 2^14)
}
__device__ __forceinline__ float pow2_inv(float s) { return __uint_as_float((254u << 23) - __float_as_uint(s)); }   // s = 2^k, |k| <= 126

__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
__device__ __forceinline__ float silu(float x) { return x * __builtin_amdgcn_rcpf(1.0f + fast_exp(-x)); }


// Cross-lane reductions on the VALU: DPP operands for the lanes of a sample (quad / row mirrors: each
// step adds the partial sum of the complementary lane group, so every lane ends with the total) and
// v_permlane32_swap for the two halves of a wave.  A ds_bpermute shuffle costs an LDS round trip each.
// (mov_dpp leaves the destination's previous value undefined for lanes without a source -- every control used with it
// covers all lanes; update_dpp(0, ...) made the compiler clear the destination with a v_mov_b32 in front of every one.)
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
  static_assert(CTRL <= 0xFF || (CTRL >= 0x121 && CTRL <= 0x12F) || CTRL == 0x140 || CTRL == 0x141, "a control that covers all lanes");
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), CTRL, 0xf, 0xf, false));
}
// max(a, b) as ONE instruction: fmaxf on values that come out of a bit cast (DPP / permlane results) gets a canonicalising
// v_max_f32 x, x per operand in front of it
__device__ __forceinline__ float vmax(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// max(x, x seen through the DPP control): one v_max_f32_dpp (the s_nop covers the VALU-write -> DPP-read hazard, which
// nobody checks inside an asm statement)
template <int CTRL>
__device__ __forceinline__ float dpp_max(float x) {
  float r;
  if constexpr (CTRL == 0x124) asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 row_ror:4 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(x));
  else if constexpr (CTRL == 0x128) asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(x));
  else if constexpr (CTRL == 0xB1) asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(x));
  else if constexpr (CTRL == 0x4E) asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(x));
  else static_assert(CTRL == 0x124, "add the control's assembler spelling");
  return r;
}
template <int L>
__device__ __forceinline__ float group_sum(float x) {  // sum over the L lanes (columns) of a sample
  x += dpp_mov<0xB1>(x);                   // quad_perm [1,0,3,2]
  x += dpp_mov<0x4E>(x);                   // quad_perm [2,3,0,1]
  if constexpr (L >= 8) x += dpp_mov<0x141>(x);   // row_half_mirror
  if constexpr (L >= 16) x += dpp_mov<0x140>(x);  // row_mirror
  return x;
}
__device__ __forceinline__ float half_sum(float x) {  // lanes i and i ^ 32
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float row_pair_sum(float x) {  // lanes i and i ^ 16
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float row_pair_max(float x) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return vmax(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float half_max(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return vmax(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

__device__ __forceinline__ float row16_max(float x) {  // max over the 16 lanes of a DPP row
  x = dpp_max<0xB1>(x);   // quad_perm [1,0,3,2]
  x = dpp_max<0x4E>(x);   // quad_perm [2,3,0,1]
  x = fmaxf(x, dpp_mov<0x141>(x));  // row_half_mirror
  x = fmaxf(x, dpp_mov<0x140>(x));  // row_mirror
  return x;
}

#if defined(GLDM_DEBUG_KNOBS) && defined(GLDM_WAVE_STAMPS)   // per-wave conv stamps cost ~1.5 k cycles per op: their own switch
__device__ long long g_wv_stamp[8][32][8];   // per wave, ring of the last 32 position-major convs: in, k-loop done, out, shape,
                                             // statistics published (in front of the exchange barrier), partners merged
__device__ int g_wv_cnt[8];
#define GLDM_WV_STAMP(c, k, v) \
  do { if (blockIdx.x == 0 && (c).lane == 0) g_wv_stamp[(c).wave][g_wv_cnt[(c).wave] & 31][k] = (v); } while (0)
#define GLDM_WV_NEXT(c) do { if (blockIdx.x == 0 && (c).lane == 0) g_wv_cnt[(c).wave]++; } while (0)
#else
#define GLDM_WV_STAMP(c, k, v) do {} while (0)
#define GLDM_WV_NEXT(c) do {} while (0)
#endif
struct Ctx {
  const float *w;   // packed weights
  float *lds;
  int tid, wave, lane;
  int skip;         // diagnostic phase-skip mask (GLDM_R1D_SKIP), 0 in production
  int nta;          // live 16-column n-tiles of this workgroup (kNT = full tile, 1 = tail tile)
  // scale / shift rows precomputed per conditioning cloud (pose decoder: ss_table_kernel), for the tile's samples 0 and
  // 1 (16-position engine: a 16-column n-tile is one sample), or null: computed in the epilogue
  const float *ss_row[2] = {nullptr, nullptr};
  const float *ss_lane = nullptr;   // 16-position 64-column engine: the same rows for THIS LANE's sample (lane & 3), or null
  // position-major engine, 256-channel level: this workgroup's 64 KiB of global scratch where the residual stream is
  // parked (f32) between the level's down conv and the end of its ResnetBlock, while LDS holds the split planes
  float *park = nullptr;
};

// ---------------------------------------------------------------- GEMM ----
// Every conv / 1x1 is  acc[mi][ni] += W[16(mt0+mi).., :] * im2col(src)[:, 16(nt0+ni)..]
// on v_mfma_f32_16x16x4_f32.  Packed weights: k = tap * Cin + ci, 16-deep k-blocks.
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

// Fast path (Cin % 16 == 0).  A 16-column tile never straddles a sample (L divides 16), so the
// row-boundary lanes of the halo shift are exactly the lanes whose tap falls outside the sample:
// zero fill (bound_ctrl) for L = 16, an extra (col % L) mask for L = 4.
struct NoPre { __device__ __forceinline__ void operator()() const {} };
// pre(): work that does not depend on the GEMM, run right after the first weight fragments have been requested (it
// then costs nothing while their round trip is outstanding).
template <int NC, int L, int TAPS, int MT, int NT, int PF, class PRE = NoPre>
__device__ __forceinline__ void gemm_fast_pf(const Ctx &c, const float *__restrict__ wp, int cblocks, int mt0, int nt0,
                                             const float *src, f32x4 (&acc)[MT][NT], const PRE &pre = PRE()) {
  // PF = weight blocks in flight; cblocks % PF == 0.  The unrolled body is UNCONDITIONAL: a load
  // whose only consumer sits behind a branch is sunk into that branch by the compiler (and then
  // waited for at once), and a branch around a load forces s_waitcnt 0 at the join.  Block indices
  // are clamped instead; the redundant re-loads at the tail are harmless.
  const int col = c.lane & 15, kq = c.lane >> 4;
  const int kblocks = TAPS * cblocks;
  const WStream wv(wp, c.lane);
  // B-fragment addresses.  Row 4 j + kq has the parity of kq for every j, so the swizzle is the same for all four
  // k-steps, and with nt0 even it only swaps the n-tiles of a pair: at 8 n-tiles (nt0 = 0) two lane-dependent bases
  // (even / odd n-tile) plus compile-time offsets, which the reads carry as immediates -- not 32 registers.  (At 2 and
  // 4 n-tiles the engine's phases measured 1 % slower this way, spill-free as they became.)
  int boff[4][NT];
  if constexpr (NT >= 8) {
    int bb[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) bb[e] = swz<NC>(kq, 16 * (nt0 + e) + col);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int ni = 0; ni < NT; ++ni) boff[j][ni] = bb[ni & 1] + 4 * j * NC + 16 * (ni & ~1);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int ni = 0; ni < NT; ++ni) boff[j][ni] = swz<NC>(4 * j + kq, 16 * (nt0 + ni) + col);
  }
  f32x4 a[PF][TAPS][MT];
  // B values: double buffered over k-blocks, except at 8 n-tiles, where a k-step's 8+ MFMAs are cover enough: the row
  // of step j is refilled from the next block the moment step j's MFMAs have issued (three steps to arrive), in ONE
  // set of registers (32 fewer at NT = 8).
  constexpr bool kBS = NT >= 8 && PF > 1;
  float b[kBS ? 1 : 2][4][NT];
  f32x4 side[TAPS > 1 ? 2 : 1][MT][NT];  // tap 0 and tap 2 partial results (tap 1 goes to acc)
  if (TAPS == 3) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) side[t][mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // part = -1: the whole block; part 0..2: the third of the block's fragments issued beside
  // k-step `part` (a burst of every wave's loads at the block boundary stalls all of them in the
  // vector-memory issue queue while the MFMA pipe idles: spread, the two pipes overlap)
  auto load_a = [&](int buf, int cb, int part) {
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
        if (part < 0 || (t * MT + mi) % 3 == part)
          a[buf][t][mi] = wv[((size_t)(mt0 + mi) * kblocks + t * cblocks + cb) * 64];
  };
  const lds_f *src3 = (const lds_f *)src;
  auto load_b = [&](int buf, int cb) {
    const lds_f *s = src3 + cb * 16 * NC;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int ni = 0; ni < NT; ++ni) b[buf][j][ni] = s[boff[j][ni]];
  };
  auto mfma_step = [&](int abuf, int bbuf, int j) {
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
  };
  const int last = cblocks - 1;
  if constexpr (PF == 1) {
    for (int cb = 0; cb < cblocks; ++cb) {
      load_a(0, cb, -1);
      load_b(0, cb);
      if (cb == 0) pre();
#pragma unroll
      for (int j = 0; j < 4; ++j) mfma_step(0, 0, j);
    }
  } else {
#pragma unroll
    for (int u = 0; u < PF - 1; ++u) load_a(u, u < last ? u : last, -1);
    load_b(0, 0);
    __builtin_amdgcn_sched_barrier(0);
    pre();
    __builtin_amdgcn_sched_barrier(0);
    for (int cb0 = 0; cb0 < cblocks; cb0 += PF) {
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int cb = cb0 + u;
        const int acb = cb + PF - 1 < last ? cb + PF - 1 : last, bcb = cb + 1 < last ? cb + 1 : last;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (j < 3) load_a((u + PF - 1) % PF, acb, j);
          if (!kBS && j == 0) load_b((u + 1) & 1, bcb);
          __builtin_amdgcn_sched_barrier(0);
          mfma_step(u, kBS ? 0 : (u & 1), j);
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (kBS) {
            const lds_f *sn = src3 + bcb * 16 * NC;
#pragma unroll
            for (int ni = 0; ni < NT; ++ni) b[0][j][ni] = sn[boff[j][ni]];
          }
        }
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

// k = 3 convs, tap-major: inside a 16-channel block the three taps are swept one after the other
// (4 k-steps each), and the moment a tap's sweep has issued its MFMAs its fragment registers are
// refilled with the NEXT block's fragments of that tap.  Every weight load then has exactly one
// block of MFMAs (48 at 2 x 2 tiles) to arrive, with ONE set of fragment registers and never more
// than a block's worth of loads in flight per wave: the double-buffered form kept up to two, and
// a long weight stream queued in the CU's vector-memory path is what delays the co-resident
// workgroup's short phases.
template <int NC, int L, int MT, int NT>
__device__ __forceinline__ void gemm_fast_tap3(const Ctx &c, const float *__restrict__ wp, int cblocks, int mt0, int nt0,
                                               const float *src, f32x4 (&acc)[MT][NT]) {
  const int col = c.lane & 15, kq = c.lane >> 4;
  const int kblocks = 3 * cblocks;
  const WStream wv(wp, c.lane);
  int boff[4][NT];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) boff[j][ni] = swz<NC>(4 * j + kq, 16 * (nt0 + ni) + col);
  f32x4 a[3][MT];
  float b[2][4][NT];
  f32x4 side[2][MT][NT];  // tap 0 and tap 2 partial results (tap 1 goes to acc)
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
      for (int ni = 0; ni < NT; ++ni) side[t][mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
  const lds_f *src3 = (const lds_f *)src;
  auto load_b = [&](int buf, int cb) {
    const lds_f *s = src3 + cb * 16 * NC;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int ni = 0; ni < NT; ++ni) b[buf][j][ni] = s[boff[j][ni]];
  };
  const int last = cblocks - 1;
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) a[t][mi] = wv[((size_t)(mt0 + mi) * kblocks + t * cblocks) * 64];
  load_b(0, 0);
  for (int cb0 = 0; cb0 < cblocks; cb0 += 2) {  // two blocks per trip: the B double buffer alternates statically
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int cb = cb0 + u;
      const int nb = cb + 1 < last ? cb + 1 : last;  // clamped: the loads stay unconditional
      load_b(1 - u, nb);
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int mi = 0; mi < MT; ++mi)
#pragma unroll
            for (int ni = 0; ni < NT; ++ni) {
              f32x4 &d = t == 1 ? acc[mi][ni] : side[t >> 1][mi][ni];
              d = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][mi][j], b[u][j][ni], d, 0, 0, 0);
            }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) a[t][mi] = wv[((size_t)(mt0 + mi) * kblocks + t * cblocks + nb) * 64];
      }
    }
  }
  const bool keepL = (col & (L - 1)) != 0, keepR = (col & (L - 1)) != (L - 1);
#pragma unroll
  for (int mi = 0; mi < MT; ++mi)
#pragma unroll
    for (int ni = 0; ni < NT; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        acc[mi][ni][r] += tap_left<L>(side[0][mi][ni][r], keepL) + tap_right<L>(side[1][mi][ni][r], keepR);
}


// ---- split-f16 GEMM core of the 64-column engines ------------------------------------------------------------
// f32 matrix products on the f16 matrix pipe.  v_mfma_f32_16x16x32_f16 delivers 16x the FLOP/cycle of
// v_mfma_f32_16x16x4_f32, so an f32 product computed EXACTLY ENOUGH from f16 pieces still wins: every f32 operand is
// written as hi + lo, two f16 numbers (11 + 11 significant bits; the matrix pipe keeps f16 subnormals, so a small lo
// part loses only bits below 2^-25), and a product a b is the sum of three partial products,
//   a b ~ a_hi b_lo + a_lo b_hi + a_hi b_hi      (three MFMAs, f32 accumulation),
// the dropped one (lo lo) being <= 2^-22 |a b|.  3/16 of the f32-MFMA time -- and half of what the three-piece bf16
// split of rounds 3-4 took (six products) at the same measured accuracy: on a 16 x 16 x 768 product the error relative
// to sum |a b| is 1.3e-7 (f32 fma chain: 1.3e-7; bf16 x 3: 1.5e-7), on operands spread over 15 binades 3.5e-7 (5.6e-7;
// 3.4e-7) -- tools/micro/mfma_f16_split, profiles/r05_mfma_f16_split.txt.  Weights are split once on the host
// (r1d_pack.py: mfma_a_fragments_f16x2, layout in gldm.h); activations are split by the producing epilogue
// (store_planes4) or as they are read from LDS (split_planes8).  Range: |x| < 65504 (f16); the packers refuse weights
// beyond it, activations of these nets are O(10) behind their norms.
// Measured against the reference's vectors: single forwards 1.7e-6 from the f32 graph, 100 DDIM steps 1.8e-6
// (tools/study/f16x2_error.py), well inside the 2e-5 / 1e-4 parity bars.
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

// x[0..7] (consecutive k of one column) -> the planes of a B fragment
__device__ __forceinline__ void split_planes8(const float (&x)[8], u32x4 (&pl)[kSplit]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    unsigned h, l;
    split_f16x2(x[2 * q], x[2 * q + 1], h, l);
    pl[0][q] = h;
    pl[1][q] = l;
  }
}
__device__ __forceinline__ f32x4 mfma_h(const u32x4 &a, const u32x4 &b, const f32x4 &c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// acc += A B with both operands split: small terms first
__device__ __forceinline__ f32x4 mfma_split(const u32x4 (&a)[kSplit], const u32x4 (&b)[kSplit], f32x4 acc) {
  acc = mfma_h(a[0], b[1], acc);
  acc = mfma_h(a[1], b[0], acc);
  return mfma_h(a[0], b[0], acc);
}

// Position-major k = 3 conv (see gemm_pm3 below for the tile algebra) on split-f16 operands.  wp3: split fragments
// of [Cout x 3 Cin], k = tap * Cin + ci, 32-deep k-blocks (Cin % 32 == 0).  One set of A registers per tap: the
// moment a tap's MFMAs have issued, its registers are refilled with the next channel block's fragments of that tap,
// which then have the two other taps' MFMAs (and the partner wave's) to arrive.  The raw f32 B values of the next
// block are read from LDS while the current block's MFMAs run and split at the top of the next trip.
#if defined(GLDM_DEBUG_KNOBS) && defined(GLDM_EXP_NO_A)   // timing experiments only (wrong results): operands of block 0 reused
constexpr bool kExpNoA = true;
#else
constexpr bool kExpNoA = false;
#endif
#if defined(GLDM_DEBUG_KNOBS) && defined(GLDM_EXP_NO_B)
constexpr bool kExpNoB = true;
#else
constexpr bool kExpNoB = false;
#endif
// First weight fragments of a position-major k = 3 conv, requested by the CALLER ahead of the conv (the fused
// ResnetBlock op asks for its second conv's while the first conv's epilogue runs): block 0's three tap sets of a
// one-m-tile wave, tap 0's set of a two-m-tile wave -- what the k-loop would otherwise request cold and wait ~1 k cycles for.
struct NoPreA { static constexpr bool on = false; };
template <int MT>
struct PreA {
  static constexpr bool on = true;
  static constexpr int kSets = MT == 1 ? 3 : 1;
  u32x4 a[kSets][MT][kSplit];
  __device__ __forceinline__ void request(const WStream &wv, int mt0, int cin) {
    const int kb32 = cin >> 5, kblocks = 3 * kb32;
#pragma unroll
    for (int t = 0; t < kSets; ++t)
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int pl = 0; pl < kSplit; ++pl) a[t][mi][pl] = wv.raw_at(((mt0 + mi) * kblocks + t * kb32) * kFragBytes, pl * 1024);
  }
};

// The conv with the B operand read from pre-split planes: ds_read_b128 per (tile, plane), no VALU.
// B planes of the next channel block are requested while the current block's MFMAs run (second register set).
template <int MT, int P0, int NP, class PRE = NoPreA>
__device__ __forceinline__ void gemm_pm3_pl(const Ctx &c, const float *__restrict__ wp3, int cin, int mt0,
                                            const float *planes, f32x4 (&acc)[MT][NP], const PRE &pre = PRE()) {
  constexpr int PB0 = P0 > 0 ? P0 - 1 : 0, PB1 = P0 + NP < 4 ? P0 + NP : 3, NB = PB1 - PB0 + 1;
  const int col = c.lane & 15, g = c.lane >> 4;
  const int kb32 = cin >> 5, kblocks = 3 * kb32;
  const WStream wv(wp3, c.lane);
  const lds_u4 *pl3 = (const lds_u4 *)planes + g * 64 + 16 * PB0 + col;   // + (kb * kSplit + plane) * 256 + 16 q
  // A registers.  One m-tile per wave: a set per tap, refilled with the next block's fragments right after the tap's
  // MFMAs (a whole block to arrive).  Two m-tiles: 72 registers that way, so two sets alternate over the tap steps
  // instead (the next step's fragments are requested in front of the current step's MFMAs: 36-48 of them, and the
  // partner wave's, to arrive); the trip covers two blocks so that the alternation is static.
  constexpr int NA = MT == 1 ? 3 : 2;
  constexpr int NBUF = (MT == 1 && NB <= 3) ? 2 : 1;   // 4 tiles x 2 sets = 96 registers: spills
  u32x4 a[NA][MT][kSplit];
  u32x4 bs[NBUF][NB][kSplit];
  auto load_a = [&](int buf, int t, int kb) {
    if (kExpNoA && kb > 0) return;
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
      for (int pl = 0; pl < kSplit; ++pl)
        a[buf][mi][pl] = wv.raw_at(((mt0 + mi) * kblocks + t * kb32 + kb) * kFragBytes, pl * 1024);
  };
  auto load_b = [&](int buf, int kb) {
    if (kExpNoB && kb > 0) return;
#pragma unroll
    for (int q = 0; q < NB; ++q)
#pragma unroll
      for (int pl = 0; pl < kSplit; ++pl) bs[buf][q][pl] = pl3[(kb * kSplit + pl) * 256 + 16 * q];
  };
  auto tap_mfmas = [&](int abuf, int bbuf, int t) {
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int sp = P0 + p + t - 1;
        if (sp >= 0 && sp <= 3) {
          const int qi = sp - PB0 < 0 ? 0 : (sp - PB0 >= NB ? NB - 1 : sp - PB0);
          acc[mi][p] = mfma_split(a[abuf][mi], bs[bbuf][qi], acc[mi][p]);
        }
      }
  };
  const int last = kb32 - 1;
  load_b(0, 0);
  auto first_a = [&]() {   // block 0's fragments (three tap sets, or tap 0's with two m-tiles): the caller's, or requested here
#pragma unroll
    for (int t = 0; t < (MT == 1 ? 3 : 1); ++t) {
      if constexpr (PRE::on) {
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
          for (int pl = 0; pl < kSplit; ++pl) a[t][mi][pl] = pre.a[t][mi][pl];
      } else {
        load_a(t, t, 0);
      }
    }
  };
  if constexpr (MT == 1) {
    first_a();
    auto block = [&](int bbuf, int nb) {
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        __builtin_amdgcn_sched_barrier(0);
        tap_mfmas(t, bbuf, t);
        __builtin_amdgcn_sched_barrier(0);
        load_a(t, t, nb);
      }
    };
    if (kb32 == 1) {
      block(0, 0);
      return;
    }
    if constexpr (NBUF == 2) {
      for (int kb0 = 0; kb0 < kb32; kb0 += 2) {   // kb32 is 2 or 4 here: the two sets of B planes alternate statically
        load_b(1, kb0 + 1);
        block(0, kb0 + 1);
        load_b(0, kb0 + 2 < last ? kb0 + 2 : last);
        block(NBUF - 1, kb0 + 2 < last ? kb0 + 2 : last);
      }
    } else {
      for (int kb = 0; kb < kb32; ++kb) {
        if (kb) load_b(0, kb);
        block(0, kb < last ? kb + 1 : last);
      }
    }
  } else {
    first_a();
    auto step = [&](int st, int kb0) {
        const int t = st % 3, kb = kb0 + st / 3;
        const int nt = (st + 1) % 3, nkb = kb0 + (st + 1) / 3;
        if (t == 0 && kb > 0) load_b(0, kb);
        load_a((st + 1) & 1, nt, nkb < last ? nkb : last);
        __builtin_amdgcn_sched_barrier(0);
        tap_mfmas(st & 1, 0, t);
        __builtin_amdgcn_sched_barrier(0);
    };
    if (kb32 == 1) {
      step(0, 0); step(1, 0); step(2, 0);
    } else {
      for (int kb0 = 0; kb0 < kb32; kb0 += 2) {
#pragma unroll
        for (int st = 0; st < 6; ++st) step(st, kb0);
      }
    }
  }
}

// k = 3 conv of the 16-position engine (LL = 16: column = 4 * position + sample).  A tap is the same B read shifted by
// one position = 4 entries of the zero-padded plane rows (PG<16>): no masks, and no (tap, position) product is padding
// except at a sample's two end positions (46 of 48 are real).  Every (tap, tile) has its own B fragments, read from LDS
// in front of the MFMAs they feed: one m-tile per wave -> the next tap step's set while the current one multiplies (two
// register sets; the trip covers two channel blocks so that the alternation is static); two m-tiles (256 channels) -> tile
// by tile, the next tile's planes under the current tile's 12 MFMAs.  A fragments exactly as in gemm_pm3_pl.
// T0, NT: the wave's n-tiles (tile = 4 consecutive positions x 4 samples).  Cin = 16 runs as one 32-channel block whose
// upper half has zero weights (r1d_pack.pad_cin32).
template <int MT, int T0, int NT, class PRE = NoPreA>
__device__ __forceinline__ void gemm_sm3_pl(const Ctx &c, const float *__restrict__ wp3, int cin, int mt0,
                                            const float *planes, f32x4 (&acc)[MT][NT], const PRE &pre = PRE()) {
  using G = PG<16>;
  const int col = c.lane & 15, g = c.lane >> 4;
  const int kb32 = (cin + 31) >> 5, kblocks = 3 * kb32;
  const WStream wv(wp3, c.lane);
  const lds_u4 *pl3 = (const lds_u4 *)planes + g * G::kCols + 16 * T0 + col;   // + (kb * kSplit + plane) * kPlaneU4 + 16 q + 4 t
  constexpr int NA = MT == 1 ? 3 : 2;
  u32x4 a[NA][MT][kSplit];
  auto load_a = [&](int buf, int t, int kb) {
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
      for (int pl = 0; pl < kSplit; ++pl)
        a[buf][mi][pl] = wv.raw_at(((mt0 + mi) * kblocks + t * kb32 + kb) * kFragBytes, pl * 1024);
  };
  auto first_a = [&]() {   // block 0's fragments (three tap sets, or tap 0's with two m-tiles): the caller's, or requested here
#pragma unroll
    for (int t = 0; t < (MT == 1 ? 3 : 1); ++t) {
      if constexpr (PRE::on) {
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
          for (int pl = 0; pl < kSplit; ++pl) a[t][mi][pl] = pre.a[t][mi][pl];
      } else {
        load_a(t, t, 0);
      }
    }
  };
  const int last = kb32 - 1;
  if constexpr (MT == 1) {
    // ONE rolling set of B fragments: the moment a tile's six MFMAs of a tap step have issued, its registers are refilled
    // with the same tile's planes of the NEXT tap step, which then have the other tiles' MFMAs to arrive (two full sets
    // were 96 registers at four tiles: spills inside the loop)
    u32x4 bs[NT][kSplit];
    auto load_b1 = [&](int q, int kb, int t) {
#pragma unroll
      for (int pl = 0; pl < kSplit; ++pl) bs[q][pl] = pl3[(kb * kSplit + pl) * G::kPlaneU4 + 16 * q + 4 * t];
    };
#pragma unroll
    for (int q = 0; q < NT; ++q) load_b1(q, 0, 0);
    first_a();
    // st: tap step inside a trip of two blocks (0..5); A set = tap
    auto step = [&](int st, int kb0) {
      const int t = st % 3, kb = kb0 + st / 3;
      const int nt = (st + 1) % 3, nkb0 = kb0 + (st + 1) / 3, nkb = nkb0 < last ? nkb0 : last;
#pragma unroll
      for (int q = 0; q < NT; ++q) {
        __builtin_amdgcn_sched_barrier(0);
        acc[0][q] = mfma_split(a[t][0], bs[q], acc[0][q]);
        __builtin_amdgcn_sched_barrier(0);
        load_b1(q, nkb, nt);
      }
      load_a(t, t, kb < last ? kb + 1 : last);
    };
    if (kb32 == 1) {
      step(0, 0); step(1, 0); step(2, 0);
    } else {
      for (int kb0 = 0; kb0 < kb32; kb0 += 2) {
#pragma unroll
        for (int st = 0; st < 6; ++st) step(st, kb0);
      }
    }
  } else {
    static_assert(MT == 1 || NT == 4, "two m-tiles per wave: all four tiles");
    u32x4 bs[2][kSplit];
    auto load_b1 = [&](int buf, int kb, int t, int q) {
#pragma unroll
      for (int pl = 0; pl < kSplit; ++pl) bs[buf][pl] = pl3[(kb * kSplit + pl) * G::kPlaneU4 + 16 * q + 4 * t];
    };
    first_a();
    load_b1(0, 0, 0, 0);
    auto step = [&](int st, int kb0) {
      const int t = st % 3, kb = kb0 + st / 3;
      const int nt = (st + 1) % 3, nkb0 = kb0 + (st + 1) / 3, nkb = nkb0 < last ? nkb0 : last;
      load_a((st + 1) & 1, nt, nkb);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (q < 3) load_b1((q + 1) & 1, kb, t, q + 1);
        else load_b1(0, nkb, nt, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) acc[mi][q] = mfma_split(a[st & 1][mi], bs[q & 1], acc[mi][q]);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    for (int kb0 = 0; kb0 < kb32; kb0 += 2) {   // 4 or 8 blocks here
#pragma unroll
      for (int st = 0; st < 6; ++st) step(st, kb0);
    }
  }
}

// 1x1 conv with the B operand from pre-split planes (the folded-LayerNorm qkv conv reads the X planes).
// MS: stride between the wave's m-tiles (the fused qkv + attention phase takes a head's q, k and v rows: 8 m-tiles apart).
// FIRST: NoFirst, or Frag3 = block 0's fragments of the first m-tile, requested by the caller ahead of the call (by value in
// registers: a pointer to them would put the array on the stack).
struct NoFirst { static constexpr bool on = false; };
struct Frag3 { static constexpr bool on = true; u32x4 p[kSplit]; };
template <int KB32, int MT, int NT, class PRE = NoPre, int MS = 1, int LL = 4, class FIRST = NoFirst>
__device__ __forceinline__ void gemm1_pl(const Ctx &c, const float *__restrict__ wp3, int mt0, int nt0, const float *planes,
                                         f32x4 (&acc)[MT][NT], const PRE &pre = PRE(), const FIRST &first = FIRST()) {
  const int col = c.lane & 15, g = c.lane >> 4;
  const WStream wv(wp3, c.lane);
  using PGx = PG<LL>;
  const lds_u4 *pl3 = (const lds_u4 *)planes + g * PGx::kCols + PGx::kOff + 16 * nt0 + col;
  u32x4 a[2][MT][kSplit];
  u32x4 bs[2][kSplit];
  auto load_a = [&](int buf, int kb) {
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
      const int sb = ((mt0 + mi * MS) * KB32 + kb) * kFragBytes;   // one scalar offset per (m-tile, block), the planes by immediates
#pragma unroll
      for (int pl = 0; pl < kSplit; ++pl) a[buf][mi][pl] = wv.raw_at(sb, pl * 1024);
    }
  };
  auto load_b = [&](int buf, int kb, int ni) {
#pragma unroll
    for (int pl = 0; pl < kSplit; ++pl) bs[buf][pl] = pl3[(kb * kSplit + pl) * PGx::kPlaneU4 + 16 * ni];
  };
  if constexpr (FIRST::on) {
#pragma unroll
    for (int pl = 0; pl < kSplit; ++pl) a[0][0][pl] = first.p[pl];
    if constexpr (MT > 1) {
#pragma unroll
      for (int mi = 1; mi < MT; ++mi) {
        const int sb = ((mt0 + mi * MS) * KB32) * kFragBytes;
#pragma unroll
        for (int pl = 0; pl < kSplit; ++pl) a[0][mi][pl] = wv.raw_at(sb, pl * 1024);
      }
    }
  } else {
    load_a(0, 0);
  }
  load_b(0, 0, 0);
  __builtin_amdgcn_sched_barrier(0);
  pre();
  __builtin_amdgcn_sched_barrier(0);
  // The requests are pinned in front of the MFMAs they are to run under: left to the scheduler, the next block's
  // fragment loads sank to their first use (load, s_waitcnt vmcnt(0), MFMA -- six to nine L2 round trips per block; the
  // 128-channel qkv conv took 18.4 k cycles for 9.2 k of MFMAs).
#pragma unroll
  for (int kb = 0; kb < KB32; ++kb) {
    if (kb + 1 < KB32) load_a((kb + 1) & 1, kb + 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) {
      const int step = kb * NT + ni, nxt = step + 1;
      if (nxt < KB32 * NT) load_b(nxt & 1, nxt / NT, nxt % NT);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) acc[mi][ni] = mfma_split(a[kb & 1][mi], bs[step & 1], acc[mi][ni]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

template <int NC, int L, int TAPS, int MT, int NT>
__device__ __forceinline__ void gemm_fast(const Ctx &c, const float *__restrict__ wp, int cblocks, int mt0, int nt0,
                                          const float *src, f32x4 (&acc)[MT][NT]) {
  constexpr int PFMAX = (MT * TAPS > 6) ? 2 : (MT * TAPS > 4 ? 2 : 4);  // register budget
  if constexpr (TAPS == 3) {
    if ((cblocks & 1) == 0) {
      gemm_fast_tap3<NC, L, MT, NT>(c, wp, cblocks, mt0, nt0, src, acc);
      return;
    }
  }
  if (PFMAX == 4 && (cblocks & 3) == 0) gemm_fast_pf<NC, L, TAPS, MT, NT, PFMAX>(c, wp, cblocks, mt0, nt0, src, acc);
  else if ((cblocks & 1) == 0) gemm_fast_pf<NC, L, TAPS, MT, NT, 2>(c, wp, cblocks, mt0, nt0, src, acc);
  else gemm_fast_pf<NC, L, TAPS, MT, NT, 1>(c, wp, cblocks, mt0, nt0, src, acc);
}

// Generic path (Cin % 16 != 0: the 4-channel level of the latent denoiser): masked reads.
template <int NC, int L, int MT, int NT>
__device__ __forceinline__ void gemm_small(const Ctx &c, const float *__restrict__ wp, int kblocks, int mt0, int nt0,
                                           const float *src, int cin, int ktaps, f32x4 (&acc)[MT][NT]) {
  const int col = c.lane & 15, kq = c.lane >> 4;
  const WStream wv(wp, c.lane);
  const lds_f *src3 = (const lds_f *)src;
  int dk = 0, cib = 0;
  const int pad = ktaps == 3 ? 1 : 0;
  for (int kb = 0; kb < kblocks; ++kb) {
    f32x4 a[MT];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) a[mi] = wv[((size_t)(mt0 + mi) * kblocks + kb) * 64];
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
        float v = src3[swz<NC>(ok ? ci : 0, ok ? n + shift : 0)];
        asm volatile("" : "+v"(v));  // keep the LDS read unconditional (no branch + wait per element)
        b[ni] = ok ? v : 0.f;
      }
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mi][j], b[ni], acc[mi][ni], 0, 0, 0);
      cib += 4;
      if (cib >= cin) {
        cib = 0;
        ++dk;
      }
    }
  }
}

template <int NC, int MT, int NT>
__device__ __forceinline__ void store_tiles(const Ctx &c, const f32x4 (&acc)[MT][NT], int mt0, int nt0, float *dst,
                                            int cout, int act) {
  const int col = c.lane & 15, kq = c.lane >> 4;
  lds_f *d3 = (lds_f *)dst;
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * (mt0 + mi) + 4 * kq + r;
      if (row < cout) {
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) {
          const float v = acc[mi][ni][r];
          d3[swz<NC>(row, 16 * (nt0 + ni) + col)] = act ? fmaxf(v, 0.f) : v;
        }
      }
    }
  }
}

// GroupNorm fused into the conv epilogue (Block: proj -> GroupNorm -> [scale/shift] -> SiLU,
// resnets.py:104-122; ResnetBlock: time/cond MLP -> (scale, shift), residual add, :125-151).
// The rows of a group always sit inside one wave's accumulators (4 groups; a wave owns a quarter of
// the rows, or part of its single m-tile on narrow levels), so the statistics are reductions over
// registers: in-lane over tiles and the 4 rows of a lane, DPP over the columns of the sample,
// permlane swaps over the row quarters.  The scale/shift rows are not a table either: each wave
// computes the ones for its own output rows with a few MFMAs against G (the per-sample embedding sum
// in LDS) right here, in the accumulator layout of the conv tile.  No LDS round trip, no barrier.
struct GnEpilogue {
  int mode;            // 0: plain conv, 1: dst = act(GN(conv)), 2: res += act(GN(conv))
  int gamma_off, beta_off;
  int ss_w, ss_b, E;   // packed [2C x E] scale/shift Linear (A fragments) + combined bias, or ss_w < 0
  int C, cpg;          // channels, channels per group (1, 4, 8, 16 or the rows of a wave)
  float *res;          // residual stream (mode 2)
  int tab_off = 0;     // this ResnetBlock's rows in the per-cloud scale/shift table (Ctx::ss_row)
};

// One wave's share of a GEMM: PASSES x MT m-tiles by NT n-tiles, one k-sweep per pass.  Passes keep
// the register footprint of a sweep small (MT * TAPS <= 6 fragments per block) so that every variant
// fits beside the other phases of the kernel without spilling; the extra cost of a pass is one
// pipeline fill.  The bias is folded into the accumulator start value.
template <int NC, int L, int TAPS, int MT, int NT, int PASSES>
__device__ __forceinline__ void gemm_passes(const Ctx &c, const float *wp, int mt0, int nt0, bool active,
                                            const float *src, int cin, float *dst, int cout, const float *bias,
                                            bool alias, int act, const GnEpilogue &g) {
  using GG = Geo<NC>;
  f32x4 acc[PASSES][MT][NT];
  const int kq = c.lane >> 4, col = c.lane & 15;
  // ---- parameters of the GroupNorm epilogue.  Every load goes out in one batch: before the k-sweep when the
  // variant is a single pass (registers to spare: the round trip hides behind the GEMM), else at the start of
  // the epilogue, in the shadow of the statistics.
  constexpr bool kEarlyParams = TAPS == 3 && NC == 32 && PASSES == 1;
  const bool has_ss = g.ss_w >= 0;
  const bool wide = g.C >= 16;  // else C = 4: one m-tile holds scale rows 0..3 (row quarter 0) and shift rows 4..7
  const int ekb = g.E >> 4;
  const WStream wss(c.w + (has_ss ? g.ss_w : 0), c.lane);
  f32x4 ga[PASSES][MT], be[PASSES][MT], sc0[PASSES][MT], sh0[PASSES][MT], a_sc[PASSES][MT], a_sh[PASSES][MT];
#define GLDM_LOAD_GN_PARAMS()                                                                                       \
  _Pragma("unroll") for (int p = 0; p < PASSES; ++p) _Pragma("unroll") for (int mi = 0; mi < MT; ++mi) {            \
    const int mt_ = mt0 + p * MT + mi;                                                                              \
    const int row0_ = 16 * mt_ + 4 * kq;                                                                            \
    const int prow_ = row0_ + 3 < cout ? row0_ : 0; /* rows past cout (narrow levels) are not stored */             \
    ga[p][mi] = *reinterpret_cast<const f32x4 *>(c.w + g.gamma_off + prow_);                                        \
    be[p][mi] = *reinterpret_cast<const f32x4 *>(c.w + g.beta_off + prow_);                                         \
    sc0[p][mi] = f32x4{1.f, 1.f, 1.f, 1.f};                                                                         \
    sh0[p][mi] = f32x4{0.f, 0.f, 0.f, 0.f};                                                                         \
    if (has_ss) {                                                                                                   \
      const float *sb_ = c.w + g.ss_b;                                                                              \
      if (wide) { /* scale rows: m-tile mt, shift rows: m-tile C/16 + mt of the [2C x E] Linear */                  \
        sc0[p][mi] = *reinterpret_cast<const f32x4 *>(sb_ + row0_);                                                 \
        sh0[p][mi] = *reinterpret_cast<const f32x4 *>(sb_ + g.C + row0_);                                           \
        a_sc[p][mi] = wss[(size_t)mt_ * ekb * 64];                                                                  \
        a_sh[p][mi] = wss[(size_t)((g.C >> 4) + mt_) * ekb * 64];                                                   \
      } else {                                                                                                      \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) sc0[p][mi][r] = sb_[4 * kq + r < 2 * g.C ? 4 * kq + r : 0];   \
        a_sc[p][mi] = wss[0];                                                                                       \
      }                                                                                                             \
    }                                                                                                               \
  }
  if (kEarlyParams && g.mode && active) { GLDM_LOAD_GN_PARAMS() }
#pragma unroll
  for (int p = 0; p < PASSES; ++p) {
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
      f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
      if (bias) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * (mt0 + p * MT + mi) + 4 * kq + r;
          bv[r] = bias[row < cout ? row : cout - 1];
        }
      }
#pragma unroll
      for (int ni = 0; ni < NT; ++ni) acc[p][mi][ni] = bv;
    }
    if (active) {
      if ((cin & 15) == 0) gemm_fast<NC, L, TAPS, MT, NT>(c, wp, cin >> 4, mt0 + p * MT, nt0, src, acc[p]);
      else gemm_small<NC, L, MT, NT>(c, wp, (TAPS * cin + 15) >> 4, mt0 + p * MT, nt0, src, cin, TAPS, acc[p]);
    }
  }
  if constexpr (TAPS == 3 && NC == 32) {
    if (g.mode) {
      if (!active) return;
      const float inv_cnt = 1.0f / (float)(g.cpg * L);  // a power of two: exact
      lds_f *d3 = (lds_f *)(g.mode == 2 ? g.res : dst);
      if (!kEarlyParams) { GLDM_LOAD_GN_PARAMS() }
#pragma unroll
      for (int ni = 0; ni < NT; ++ni) {
        const int n = 16 * (nt0 + ni) + col;
        const lds_f *Gs = (const lds_f *)(c.lds + GG::kMiscG) + (n / L) * g.E;  // this column's sample
        // ---- statistics
        float mean[PASSES][MT][4], rstd[PASSES][MT][4];
        if (g.cpg >= 32) {  // the group is everything this wave accumulates
          float s1 = 0.f;
#pragma unroll
          for (int p = 0; p < PASSES; ++p)
#pragma unroll
            for (int mi = 0; mi < MT; ++mi)
#pragma unroll
              for (int r = 0; r < 4; ++r) s1 += acc[p][mi][ni][r];
          s1 = half_sum(row_pair_sum(group_sum<L>(s1)));
          const float m = s1 * inv_cnt;
          float s2 = 0.f;
#pragma unroll
          for (int p = 0; p < PASSES; ++p)
#pragma unroll
            for (int mi = 0; mi < MT; ++mi)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const float dx = acc[p][mi][ni][r] - m;
                s2 += dx * dx;
              }
          s2 = half_sum(row_pair_sum(group_sum<L>(s2)));
          const float rs = __builtin_amdgcn_rsqf(s2 * inv_cnt + 1e-5f);
#pragma unroll
          for (int p = 0; p < PASSES; ++p)
#pragma unroll
            for (int mi = 0; mi < MT; ++mi)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                mean[p][mi][r] = m;
                rstd[p][mi][r] = rs;
              }
        } else {
#pragma unroll
          for (int p = 0; p < PASSES; ++p)
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) {
              if (g.cpg == 1) {  // every accumulator row is its own group
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  const float x = acc[p][mi][ni][r];
                  const float m = group_sum<L>(x) * inv_cnt;
                  const float dx = x - m;
                  mean[p][mi][r] = m;
                  rstd[p][mi][r] = __builtin_amdgcn_rsqf(group_sum<L>(dx * dx) * inv_cnt + 1e-5f);
                }
              } else {  // 4, 8 or 16 rows of this m-tile: the lane's 4 rows, then row quarters
                float s1 = acc[p][mi][ni][0] + acc[p][mi][ni][1] + acc[p][mi][ni][2] + acc[p][mi][ni][3];
                s1 = group_sum<L>(s1);
                if (g.cpg >= 8) s1 = row_pair_sum(s1);
                if (g.cpg >= 16) s1 = half_sum(s1);
                const float m = s1 * inv_cnt;
                float s2 = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  const float dx = acc[p][mi][ni][r] - m;
                  s2 += dx * dx;
                }
                s2 = group_sum<L>(s2);
                if (g.cpg >= 8) s2 = row_pair_sum(s2);
                if (g.cpg >= 16) s2 = half_sum(s2);
                const float rs = __builtin_amdgcn_rsqf(s2 * inv_cnt + 1e-5f);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  mean[p][mi][r] = m;
                  rstd[p][mi][r] = rs;
                }
              }
            }
        }
        // ---- scale/shift rows of this column's sample, normalise, SiLU, store / accumulate
        float gb[4];
        if (has_ss) {
#pragma unroll
          for (int j = 0; j < 4; ++j) gb[j] = Gs[4 * j + kq];
        }
#pragma unroll
        for (int p = 0; p < PASSES; ++p)
#pragma unroll
          for (int mi = 0; mi < MT; ++mi) {
            const int mt = mt0 + p * MT + mi;
            const int row0 = 16 * mt + 4 * kq;
            f32x4 sc = sc0[p][mi], sh = sh0[p][mi];
            const float *tab = L == 16 ? (((nt0 + ni) & 1) ? c.ss_row[1] : c.ss_row[0]) : nullptr;  // wave uniform (a select: a run-time index keeps Ctx in scratch)
            if (has_ss && wide && tab) {
              // The pose decoder's embedding does not depend on the grasp: the rows were computed once per cloud
              // (ss_table_kernel).  In here they cost 32 MFMAs per m-tile and SAMPLE (E = 64), 17 % on top of a
              // 256-wide conv's own, 15 of every n-tile's 16 columns repeating the first.
              sc = *reinterpret_cast<const f32x4 *>(tab + g.tab_off + row0);
              sh = *reinterpret_cast<const f32x4 *>(tab + g.tab_off + g.C + row0);
            } else if (has_ss) {
              if (wide) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                  sc = __builtin_amdgcn_mfma_f32_16x16x4f32(a_sc[p][mi][j], gb[j], sc, 0, 0, 0);
                  sh = __builtin_amdgcn_mfma_f32_16x16x4f32(a_sh[p][mi][j], gb[j], sh, 0, 0, 0);
                }
                // (E = 64, the pose decoder: 32 of these per m-tile and sample, 17 % on top of a 256-wide conv's own
                // MFMAs -- a 16-column n-tile is ONE sample there, so 15 of its 16 columns repeat the first.  Requesting
                // the kb >= 1 fragments together instead of one round trip each changed nothing: it is MFMA time.)
                for (int kb = 1; kb < ekb; ++kb) {  // wide embeddings
                  const f32x4 a2 = wss[((size_t)mt * ekb + kb) * 64], a3 = wss[((size_t)((g.C >> 4) + mt) * ekb + kb) * 64];
#pragma unroll
                  for (int j = 0; j < 4; ++j) {
                    const float bj = Gs[16 * kb + 4 * j + kq];
                    sc = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[j], bj, sc, 0, 0, 0);
                    sh = __builtin_amdgcn_mfma_f32_16x16x4f32(a3[j], bj, sh, 0, 0, 0);
                  }
                }
              } else {
                f32x4 t = sc;
#pragma unroll
                for (int j = 0; j < 4; ++j) t = __builtin_amdgcn_mfma_f32_16x16x4f32(a_sc[p][mi][j], gb[j], t, 0, 0, 0);
                for (int kb = 1; kb < ekb; ++kb) {
                  const f32x4 a2 = wss[(size_t)kb * 64];
#pragma unroll
                  for (int j = 0; j < 4; ++j) t = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[j], Gs[16 * kb + 4 * j + kq], t, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(t[r]), __float_as_uint(t[r]), false, false);
                  sc[r] = t[r];                    // valid in row quarter 0, the only one that is stored
                  sh[r] = __uint_as_float(sw[1]);  // quarter 1's value seen from quarter 0
                }
              }
            }
            // the four values of the lane, without wave-uniform branches between them (mode, scale/shift and the row
            // bound are tested once per m-tile: tested per value they cut the exp / rcp chains into basic blocks)
            auto finish4 = [&](auto mode_c, auto ss_c, auto full_c) {
              constexpr int kMode = decltype(mode_c)::value;
              constexpr bool kSS = decltype(ss_c)::value, kFull = decltype(full_c)::value;
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                float y = (acc[p][mi][ni][r] - mean[p][mi][r]) * rstd[p][mi][r] * ga[p][mi][r] + be[p][mi][r];
                if (kSS) y = y * sc[r] + sh[r];
                y = silu(y);
                if (kFull || row0 + r < cout) {
                  const int a = swz<NC>(row0 + r, n);
                  d3[a] = kMode == 2 ? d3[a] + y : y;
                }
              }
            };
            using std::integral_constant;
            typedef integral_constant<bool, true> T;
            typedef integral_constant<bool, false> F;
            const bool full = row0 + 3 < cout;
            if (g.mode == 2) {
              if (has_ss) { if (full) finish4(integral_constant<int, 2>{}, T{}, T{}); else finish4(integral_constant<int, 2>{}, T{}, F{}); }
              else { if (full) finish4(integral_constant<int, 2>{}, F{}, T{}); else finish4(integral_constant<int, 2>{}, F{}, F{}); }
            } else {
              if (has_ss) { if (full) finish4(integral_constant<int, 1>{}, T{}, T{}); else finish4(integral_constant<int, 1>{}, T{}, F{}); }
              else { if (full) finish4(integral_constant<int, 1>{}, F{}, T{}); else finish4(integral_constant<int, 1>{}, F{}, F{}); }
            }
          }
      }
      return;
    }
  }
  if (alias) __syncthreads();
  if (active) {
#pragma unroll
    for (int p = 0; p < PASSES; ++p) store_tiles<NC, MT, NT>(c, acc[p], mt0 + p * MT, nt0, dst, cout, act);
  }
}

#undef GLDM_LOAD_GN_PARAMS

constexpr int kOpInts = 12, kMaxOps = 84;  // op tape: 84 * 12 = 1008 ints; the op count lives in int 1023
constexpr int kPmMaxOps = 8 * GLDM_R1D_MAX_LEVELS + 2;   // 64-column engines: at most 8 entries per level + the last ResnetBlock's 2

// =========================================================================================================
// Position-major engine pieces (L = 4, 64-column tiles = 16 samples x 4 positions, column = 16 * pos + sample,
// 8 waves, one workgroup per CU).  1x1 convs, LayerNorm and the final 1x1 are layout agnostic and shared with
// the sample-major engine; what follows are the layout-aware phases.
// =========================================================================================================

// One wave's share of a k = 3 conv: MT m-tiles x out positions P0..P0+NP-1, with the GroupNorm / scale-shift /
// SiLU / residual epilogue.  A group's rows no longer sit in one wave (8 waves share 16 m-tiles and the taps
// tie the 4 position tiles together), so the statistics are combined across waves: every wave reduces its own
// share to (sum, M2 about its own mean) per sample -- in-lane over its accumulators, permlane swaps over the
// row quarters -- publishes the pair in LDS, and after ONE barrier merges its partners' pairs with the
// parallel-variance formula (exact for equal counts, no E[x^2] - m^2 cancellation).
//   GK 0: partner = the adjacent wave (C = 256: 2 m-tiles per wave, C = 128: 1; all 4 positions)
//   GK 1: partner = wave ^ 4 (C = 64: one m-tile = one group, positions split in two halves)
//   GK 2: four waves (one per position) x two groups per m-tile (C = 32: 8 channels per group)
struct NoHook { __device__ __forceinline__ void operator()() const {} };
constexpr float sqrt_up(int n) {   // >= sqrt(n), n a power of two
  float r = 1.f;
  while (n >= 4) { r *= 2.f; n /= 4; }
  return n >= 2 ? r * 1.41422f : r;
}
// FIN: which epilogues this instance carries (code size: the kernel's straight-line phases must stay inside the instruction
// cache): 0 = none (the level's down conv, mode 0), 1 = block1 (H = act(GN(conv)), scale/shift at run time), 2 = block2
// (X += act(GN(conv)), no scale/shift).  Modes 1 and 2 only ever run inside the fused ResnetBlock op.
// LL: 4 = position-major tiles of the 4-position denoiser (P0 / NP: positions); 16 = the 16-position engine (P0 / NP: the
// wave's n-tiles of 4 positions x 4 samples; sample = lane & 3; GK 3: C = 16, one m-tile whose four row quarters are the
// four groups, waves 0-3 one tile each -- waves 4-7 repeat their work with `live` false so that every wave meets the
// barriers).  The statistics of a sample then also sum over the four positions inside a tile (DPP row rotations).
template <int MT, int P0, int NP, int GK, int FIN = 0, class PRE = NoPreA, class HOOK = NoHook, int LL = 4>
__device__ __forceinline__ void conv_pm3_wave(const Ctx &c, const float *wp, const float *bias, int mt0,
                                              const float *src, int cin, float *dst, int cout, bool alias,
                                              const GnEpilogue &g, const PRE &pre = PRE(), const HOOK &hook = HOOK(),
                                              bool live = true) {
  using GG = Geo<64>;
  using PGx = PG<LL>;
  const int kq = c.lane >> 4, cl = c.lane & 15;
  const int sm = LL == 16 ? (c.lane & 3) : cl;   // the lane's sample
  f32x4 acc[MT][NP];
  const bool has_ss = FIN == 1 && g.ss_w >= 0;
  const bool ss_tab = LL == 16 && has_ss && c.ss_lane != nullptr;   // rows precomputed per cloud (pose decoder)
  const int ekb = g.E >> 4;
  const WStream wss(c.w + (has_ss ? g.ss_w : 0), c.lane);
  f32x4 ga[MT], be[MT], sc[MT], sh[MT], a_sc[MT], a_sh[MT];
  // Epilogue parameters are requested right after the k-sweep (measured: requesting them before it, live through
  // the sweep, is 0-10 % slower on the narrow convs and no faster on the wide ones: tools/micro/gemm_pm_rate)
  auto load_params = [&]() {
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
      const int mt = mt0 + mi, row0 = 16 * mt + 4 * kq;
      ga[mi] = *reinterpret_cast<const f32x4 *>(c.w + g.gamma_off + row0);
      be[mi] = *reinterpret_cast<const f32x4 *>(c.w + g.beta_off + row0);
      sc[mi] = f32x4{1.f, 1.f, 1.f, 1.f};
      sh[mi] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (ss_tab) {
        sc[mi] = *reinterpret_cast<const f32x4 *>(c.ss_lane + g.tab_off + row0);
        sh[mi] = *reinterpret_cast<const f32x4 *>(c.ss_lane + g.tab_off + g.C + row0);
      } else if (has_ss) {
        const float *sb = c.w + g.ss_b;
        sc[mi] = *reinterpret_cast<const f32x4 *>(sb + row0);
        sh[mi] = *reinterpret_cast<const f32x4 *>(sb + g.C + row0);
        a_sc[mi] = wss[(size_t)mt * ekb * 64];
        a_sh[mi] = wss[(size_t)((g.C >> 4) + mt) * ekb * 64];
      }
    }
  };
  // Range of H.  A ResnetBlock's H = act((scale + 1) GN(conv1) + shift) is the one operand whose size is set by the DATA
  // (the conditioning embedding through scale / shift: 5e5 with wild conditioning rows, tests/test_cli.py), and f16
  // planes end at 65504.  block1 therefore writes H / hs, hs a power of two per SAMPLE chosen from a bound it already has
  // in registers (1 for anything ordinary: every operation below is then bit for bit what it was); block2 computes
  // conv(H) / hs (bias / hs in the accumulator) and folds hs back into its GroupNorm coefficients (per-sample
  // statistics: mean and M2 scale with hs and hs^2).  Everything else these engines feed the f16 pipe is bounded by the
  // weights alone (norm outputs, residual sums of them, softmax-weighted values).
  lds_f *hsc = (lds_f *)(c.lds + GG::kMiscHs);
  float hs = 1.0f;
  if constexpr (FIN == 2) hs = hsc[sm];
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) {
    f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
    if (bias) bv = *reinterpret_cast<const f32x4 *>(bias + 16 * (mt0 + mi) + 4 * kq);
    if constexpr (FIN == 2) {
      const float hinv = __builtin_amdgcn_rcpf(hs);   // a power of two: exact
#pragma unroll
      for (int r = 0; r < 4; ++r) bv[r] *= hinv;
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) acc[mi][p] = bv;
  }
  // B operand: pre-split planes.  Up to 128 input channels: the X planes or the H planes, by which buffer `src` is; 256
  // (the last level's ResnetBlock): the level's one plane set (kW)
  GLDM_WV_STAMP(c, 0, (long long)__builtin_readcyclecounter());
  GLDM_WV_STAMP(c, 3, (long long)(cin * 1000 + cout));
  const bool src_is_x = src == c.lds + GG::kBufX;
  const float *bplanes = c.lds + (cin > kPlaneMaxC ? PGx::kW : (src_is_x ? PGx::kX : PGx::kH));
  if constexpr (LL == 16) gemm_sm3_pl<MT, P0, NP, PRE>(c, wp, cin, mt0, bplanes, acc, pre);
  else gemm_pm3_pl<MT, P0, NP, PRE>(c, wp, cin, mt0, bplanes, acc, pre);
  GLDM_WV_STAMP(c, 1, (long long)__builtin_readcyclecounter());
  if (FIN != 0) load_params();
  // 256-channel level (two m-tiles per wave).  16-position engine: this lane's slice of the parked residual stream, one
  // f32x4 per (m-tile, position) (its 256 f32 rows do not fit beside the padded planes); the 4-position engine keeps the
  // f32 rows in LDS (PG<4>::kW lies behind them)
  constexpr bool kWide = MT == 2;
  constexpr bool kPark = kWide && LL == 16;
  f32x4 *pk = reinterpret_cast<f32x4 *>(c.park) + (size_t)(c.wave * MT * NP) * 64 + c.lane;
  f32x4 parked[FIN == 2 && kPark ? MT : 1][FIN == 2 && kPark ? NP : 1];
  if constexpr (FIN == 2 && kPark) {   // requested here, in flight under the statistics and the exchange barrier
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
      for (int p = 0; p < NP; ++p) parked[mi][p] = pk[(mi * NP + p) * 64];
  }
  hook();   // the fused ResnetBlock's request for its second conv's first fragments: in flight under this epilogue
  if constexpr (FIN == 0) {  // the level's down conv: the new residual stream X as planes, and as f32 rows (up to 128
                             // channels: in LDS) or parked in global scratch (256: see kPlaneW)
    if (alias) __syncthreads();
    if (live) {
      lds_f *d3 = (lds_f *)dst;
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          if constexpr (kPark) {
            pk[(mi * NP + p) * 64] = acc[mi][p];
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) d3[pswz(16 * (mt0 + mi) + 4 * kq + r, 16 * (P0 + p) + cl)] = acc[mi][p][r];
          }
          store_planes4<LL>(c.lds + (kWide ? PGx::kW : PGx::kX), 16 * (mt0 + mi) + 4 * kq, 16 * (P0 + p) + cl, acc[mi][p][0],
                            acc[mi][p][1], acc[mi][p][2], acc[mi][p][3]);
        }
    }
    GLDM_WV_STAMP(c, 2, (long long)__builtin_readcyclecounter());
    GLDM_WV_NEXT(c);
    return;
  }
  if constexpr (FIN != 0) {
  // ---- this wave's share of the statistics, per sample
  constexpr int kRows = GK == 3 ? 4 : (GK == 2 ? 8 : 16 * MT);   // rows behind one published pair
  constexpr int kNloc = kRows * NP * (LL == 16 ? 4 : 1);         // values behind it
  constexpr int kParts = (GK == 2 || GK == 3) ? 4 : 2;
  auto over_sample = [&](float v) {   // the lanes that hold this sample's other positions / row quarters of the group
    if constexpr (LL == 16) {
      v += dpp_mov<0x124>(v);   // row_ror:4
      v += dpp_mov<0x128>(v);   // row_ror:8: the four positions inside the tile
    }
    if constexpr (GK != 3) v = row_pair_sum(v);
    if constexpr (GK == 0 || GK == 1) v = half_sum(v);
    return v;
  };
  float s1 = 0.f;
#pragma unroll
  for (int mi = 0; mi < MT; ++mi)
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int r = 0; r < 4; ++r) s1 += acc[mi][p][r];
  s1 = over_sample(s1);
  const float mloc = s1 * (1.0f / (float)kNloc);
  float s2 = 0.f;
#pragma unroll
  for (int mi = 0; mi < MT; ++mi)
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float dx = acc[mi][p][r] - mloc;
        s2 += dx * dx;
      }
  s2 = over_sample(s2);
  lds_f *red1 = (lds_f *)(c.lds + GG::kMiscRed1), *red2 = (lds_f *)(c.lds + GG::kMiscRed2);
  const int slot = GK == 3 ? kq : (GK == 2 ? (kq >> 1) : 0);
  const bool pub = (GK == 3 || (kq & (GK == 2 ? 1 : 3)) == 0) && (LL != 16 || (cl >> 2) == 0);
  if (pub && live) {
    red1[(c.wave * 4 + slot) * 16 + sm] = s1;
    red2[(c.wave * 4 + slot) * 16 + sm] = s2;
  }
  // ---- scale / shift rows of this lane's sample (the same for all positions: once per m-tile).  They do not depend
  // on the statistics, so they are issued in front of the exchange barrier: the MFMA chain runs while the wave waits
#if defined(GLDM_DEBUG_KNOBS) && defined(GLDM_EXP_NO_SS)   // timing experiment only (wrong results): no scale/shift chain
  if (false) {
#else
  if (has_ss && !ss_tab) {
#endif
    const lds_f *Gs = (const lds_f *)(c.lds + GG::kMiscG) + sm * g.E;
    float gb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) gb[j] = Gs[4 * j + kq];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        sc[mi] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_sc[mi][j], gb[j], sc[mi], 0, 0, 0);
        sh[mi] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_sh[mi][j], gb[j], sh[mi], 0, 0, 0);
      }
      for (int kb = 1; kb < ekb; ++kb) {
        const int mt = mt0 + mi;
        const f32x4 a2 = wss[((size_t)mt * ekb + kb) * 64], a3 = wss[((size_t)((g.C >> 4) + mt) * ekb + kb) * 64];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float bj = Gs[16 * kb + 4 * j + kq];
          sc[mi] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[j], bj, sc[mi], 0, 0, 0);
          sh[mi] = __builtin_amdgcn_mfma_f32_16x16x4f32(a3[j], bj, sh[mi], 0, 0, 0);
        }
      }
    }
  }
  if constexpr (FIN == 1) {
    // |H| <= |gamma (scale + 1)| R + |beta (scale + 1) + shift| over this lane's rows, R = sqrt(group size) >= any
    // normalised value of the group; max over the lanes of the sample, then over the eight waves behind the barrier
    constexpr float kR = sqrt_up(kNloc * kParts);
    float hb = 0.f;
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        hb = fmaxf(hb, fmaf(__builtin_fabsf(ga[mi][r] * sc[mi][r]), kR, __builtin_fabsf(fmaf(be[mi][r], sc[mi][r], sh[mi][r]))));
    if constexpr (LL == 16) {
      hb = dpp_max<0x124>(hb);
      hb = dpp_max<0x128>(hb);
    }
    hb = half_max(row_pair_max(hb));
    if (kq == 0 && (LL != 16 || (cl >> 2) == 0)) ((lds_f *)(c.lds + GG::kMiscHb))[c.wave * 16 + sm] = hb;   // shadow waves too
  }
  GLDM_WV_STAMP(c, 4, (long long)__builtin_readcyclecounter());
  __syncthreads();
  GLDM_WV_STAMP(c, 5, (long long)__builtin_readcyclecounter());
  float hinv = 1.0f;   // block1: 1 / hs of this lane's sample
  if constexpr (FIN == 1) {
    const lds_f *hbp = (const lds_f *)(c.lds + GG::kMiscHb) + sm;
    float hb = hbp[0];
#pragma unroll
    for (int q = 1; q < 8; ++q) hb = fmaxf(hb, hbp[16 * q]);
    // hb in [2^k, 2^(k+1)): hs = 2^max(0, k - 14) puts H / hs below 2^15
    int e = (int)((__float_as_uint(hb) >> 23) & 0xffu) - 127 - 14;
    e = e < 0 ? 0 : e;
    hinv = __uint_as_float((unsigned)(127 - e) << 23);
    if (c.wave == 0 && kq == 0 && (LL != 16 || (cl >> 2) == 0)) hsc[sm] = __uint_as_float((unsigned)(127 + e) << 23);
  }
  float tot = 0.f, ps1[kParts], ps2[kParts];
#pragma unroll
  for (int q = 0; q < kParts; ++q) {
    // partners: GK 0 the adjacent wave; GK 1 wave ^ 4 (the other half of the positions / tiles); GK 2 the four waves of the
    // m-tile (one per position / tile); GK 3 waves 0-3 (one tile each)
    const int pw = GK == 0 ? ((c.wave & ~1) + q) : (GK == 1 ? ((c.wave & 3) + 4 * q) : (GK == 2 ? ((c.wave & 1) + 2 * q) : q));
    ps1[q] = red1[(pw * 4 + slot) * 16 + sm];
    ps2[q] = red2[(pw * 4 + slot) * 16 + sm];
    tot += ps1[q];
  }
  const float mean = tot * (1.0f / (float)(kNloc * kParts));
  float m2 = 0.f;
#pragma unroll
  for (int q = 0; q < kParts; ++q) {
    const float dm = ps1[q] * (1.0f / (float)kNloc) - mean;
    m2 += ps2[q] + (float)kNloc * dm * dm;
  }
  // block2: the accumulators hold conv / hs: the true variance is hs^2 times theirs, and (acc - mean') hs rstd is the
  // normalised value (hs = 1: the same bits as without it)
  const float rstd = __builtin_amdgcn_rsqf((m2 * hs) * hs * (1.0f / (float)(kNloc * kParts)) + 1e-5f) * hs;
  // mode 1 (block1): H = y, as planes only (H is only ever a conv input);
  // mode 2 (block2): X += y in f32 (the residual stream), plus the planes of the new X up to 128 channels; at 256
  // channels X = parked + y as f32 rows only (what follows is the final 1x1 conv).
  // One straight-line instance per (mode, scale/shift): with the two tested per VALUE (wave-uniform branches inside
  // the unrolled loops) every value was its own chain of basic blocks -- 61 branches and no overlap between the 16-32
  // exp / rcp chains of a lane: 3.8-4.9 k cycles for the 16 values of a one-m-tile conv, alone on the SIMD or not.
  lds_f *d3 = (lds_f *)(FIN == 2 ? g.res : dst);
  auto finish = [&](auto mode_c, auto ss_c) {
    constexpr int kMode = decltype(mode_c)::value;
    constexpr bool kSS = decltype(ss_c)::value;
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
      // GroupNorm, its affine and the scale/shift are ONE fma per value, y = acc A + B, and SiLU's exponent a second one
      // from the same accumulator (u = -log2(e) y): the coefficients depend on the row only and serve the wave's NP
      // positions (with a single position per wave the plain chain is shorter).  VALU instructions per value: 6 instead of 9.
      constexpr bool kFold = NP >= 2;
      float fa[4], fb[4], na[4], nb[4];
      if constexpr (kFold) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float A = rstd * ga[mi][r];
          float B = be[mi][r] - mean * A;
          if (kSS) {
            B = B * sc[mi][r] + sh[mi][r];
            A = A * sc[mi][r];
          }
          fa[r] = A; fb[r] = B;
          na[r] = -1.44269504088896340736f * A; nb[r] = -1.44269504088896340736f * B;
        }
      }
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        float y[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float t;
          if constexpr (kFold) {
            const float v = acc[mi][p][r];
            t = fmaf(v, fa[r], fb[r]);
            t = t * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(fmaf(v, na[r], nb[r])));
          } else {
            t = (acc[mi][p][r] - mean) * (rstd * ga[mi][r]) + be[mi][r];
            if (kSS) t = t * sc[mi][r] + sh[mi][r];
            t = silu(t);
          }
          const int a = pswz(16 * (mt0 + mi) + 4 * kq + r, 16 * (P0 + p) + cl);
          if constexpr (kMode == 2) {
            if constexpr (kPark) t = parked[mi][p][r] + t;
            else t = d3[a] + t;
            d3[a] = t;
          }
          y[r] = kMode == 1 ? t * hinv : t;
        }
        if constexpr (!(kMode == 2 && kWide))
          store_planes4<LL>(c.lds + (kMode == 2 ? PGx::kX : (kWide ? PGx::kW : PGx::kH)), 16 * (mt0 + mi) + 4 * kq,
                            16 * (P0 + p) + cl, y[0], y[1], y[2], y[3]);
      }
    }
  };
  using std::integral_constant;
  if (live) {
    if constexpr (FIN == 2) {
      finish(integral_constant<int, 2>{}, integral_constant<bool, false>{});
    } else {
      if (has_ss) finish(integral_constant<int, 1>{}, integral_constant<bool, true>{});
      else finish(integral_constant<int, 1>{}, integral_constant<bool, false>{});
    }
  }
  }
  GLDM_WV_STAMP(c, 2, (long long)__builtin_readcyclecounter());
  GLDM_WV_NEXT(c);
}

// Conv1d(4 -> cout, k = 3) of the first level (Cin = 4 is below the MFMA k-block): VALU, lane = column, wave w
// computes output channels 4w .. 4w+3 (cout = 32).  Weights are wave uniform (scalar loads); fma chain in the
// MFMA's k order (tap major, channel minor) from the bias.  src may alias dst.
template <int ROUNDS>   // 32 output channels per round (4 per wave); cout == 32 ROUNDS exactly: no per-value guards
__device__ __forceinline__ void conv_pm3_cin4_rounds(const Ctx &c, const float *wp, const float *bias, const float *src,
                                                     float *dst, bool alias) {
  const int n = c.lane, p = n >> 4;
  const lds_f *s3 = (const lds_f *)src;
  float x[3][4];
#pragma unroll
  for (int ci = 0; ci < 4; ++ci) {
    const float l = s3[pswz(ci, p > 0 ? n - 16 : n)], m = s3[pswz(ci, n)], r = s3[pswz(ci, p < 3 ? n + 16 : n)];
    x[0][ci] = p > 0 ? l : 0.f;
    x[1][ci] = m;
    x[2][ci] = p < 3 ? r : 0.f;
  }
  // all rounds are computed before the (possibly aliased) destination is written
  float out[ROUNDS][4];
#pragma unroll
  for (int rd = 0; rd < ROUNDS; ++rd) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int cc = __builtin_amdgcn_readfirstlane(rd * 32 + c.wave * 4 + k);
      float acc = bias ? bias[cc] : 0.f;
      const float *wr = wp + (size_t)(cc >> 4) * 256 + (cc & 15) * 4;  // [(ci * 16 + co % 16) * 4 + tap]
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int ci = 0; ci < 4; ++ci) acc = fmaf(wr[ci * 64 + t], x[t][ci], acc);
      out[rd][k] = acc;
    }
  }
  if (alias) __syncthreads();
  lds_f *d3 = (lds_f *)dst;
#pragma unroll
  for (int rd = 0; rd < ROUNDS; ++rd) {
#pragma unroll
    for (int k = 0; k < 4; ++k) d3[pswz(rd * 32 + c.wave * 4 + k, n)] = out[rd][k];
    // the new residual stream's planes (a wave's 4 channels are one half of an 8-group)
    if (ROUNDS * 32 <= kPlaneMaxC)
      store_planes4(c.lds + kPlaneX, rd * 32 + c.wave * 4, n, out[rd][0], out[rd][1], out[rd][2], out[rd][3]);
  }
}
__device__ __forceinline__ void conv_pm3_cin4_any(const Ctx &c, const float *wp, const float *bias, const float *src,
                                              float *dst, int cout, bool alias) {
  const int n = c.lane, p = n >> 4;
  const lds_f *s3 = (const lds_f *)src;
  float x[3][4];
#pragma unroll
  for (int ci = 0; ci < 4; ++ci) {
    const float l = s3[pswz(ci, p > 0 ? n - 16 : n)], m = s3[pswz(ci, n)], r = s3[pswz(ci, p < 3 ? n + 16 : n)];
    x[0][ci] = p > 0 ? l : 0.f;
    x[1][ci] = m;
    x[2][ci] = p < 3 ? r : 0.f;
  }
  // 32 output channels per round (4 per wave); wider first levels take more rounds.  All rounds are computed before
  // the (possibly aliased) destination is written.
  constexpr int kMaxRounds = kMaxC / 32;
  float out[kMaxRounds][4];
#pragma unroll
  for (int rd = 0; rd < kMaxRounds; ++rd) {
    if (rd * 32 < cout) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int co = __builtin_amdgcn_readfirstlane(rd * 32 + c.wave * 4 + k);
        const int cc = co < cout ? co : cout - 1;
        float acc = bias ? bias[cc] : 0.f;
        const float *wr = wp + (size_t)(cc >> 4) * 256 + (cc & 15) * 4;  // [(ci * 16 + co % 16) * 4 + tap]
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int ci = 0; ci < 4; ++ci) acc = fmaf(wr[ci * 64 + t], x[t][ci], acc);
        out[rd][k] = acc;
      }
    }
  }
  if (alias) __syncthreads();
  lds_f *d3 = (lds_f *)dst;
#pragma unroll
  for (int rd = 0; rd < kMaxRounds; ++rd) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int co = rd * 32 + c.wave * 4 + k;
      if (co < cout) d3[pswz(co, n)] = out[rd][k];
    }
    // the new residual stream's planes (a wave's 4 channels are one half of an 8-group; cout % 32 == 0)
    if (rd * 32 < cout && cout <= kPlaneMaxC)
      store_planes4(c.lds + kPlaneX, rd * 32 + c.wave * 4, n, out[rd][0], out[rd][1], out[rd][2], out[rd][3]);
  }
}

__device__ __forceinline__ void conv_pm3_cin4(const Ctx &c, const float *wp, const float *bias, const float *src,
                                              float *dst, int cout, bool alias) {
  // the shipped 32-channel level without per-value `co < cout` tests (32 branches in the common body); wider first
  // levels on the common body (an instance per width spilled 361 registers across the kernel)
  if (cout == 32) conv_pm3_cin4_rounds<1>(c, wp, bias, src, dst, alias);
  else conv_pm3_cin4_any(c, wp, bias, src, dst, cout, alias);
}

// ResnetBlock of the 4-channel level on the VALU of one wave: lane = column (16 samples x 4 positions), every
// lane carries all 4 channels of its column.  Taps come from the lanes 16 below / above (the neighbouring
// positions of the same sample), GroupNorm (one channel per group) reduces over lanes n ^ 16, n ^ 32.  Every
// weight is wave uniform (scalar loads: nothing queues in the vector-memory path).
__device__ __forceinline__ void resblock4_pm(const Ctx &c, const int (&o)[kOpInts], int E) {
  using GG = Geo<64>;
  if (c.wave != 0) return;
  const int n = c.lane, sm = n & 15, p = n >> 4;
  const bool has_l = p != 0, has_r = p != 3;
  lds_f *X = (lds_f *)(c.lds + GG::kBufX);
  const lds_f *G = (const lds_f *)(c.lds + GG::kMiscG) + sm * E;
  const float *w = c.w;
  const int c1_w = o[1], c1_b = o[2], n1_w = o[3], n1_b = o[4], c2_w = o[5], c2_b = o[6], n2_w = o[7], n2_b = o[8],
            ss_w = o[9], ss_b = o[10];
  // Every weight of the phase is staged by explicit 16-byte loads issued in two batches in front of their use (wave-uniform
  // addresses; all offsets are multiples of 4 floats in the packed buffer).  Left to the scheduler, one build of this
  // kernel issued them one by one with a full wait each (18.7 k cycles for the phase instead of 5.8 k).
  struct ConvW { f32x4 wt[16], b, g, be; };   // W[co][ci][tap] at f32x4 index ci * 16 + co (co < 4), taps in .xyz
  auto load_conv = [&](ConvW &cw, int w_off, int b_off, int g_off, int be_off) {
    const f32x4 *wv = reinterpret_cast<const f32x4 *>(w + w_off);
#pragma unroll
    for (int ci = 0; ci < 4; ++ci)
#pragma unroll
      for (int co = 0; co < 4; ++co) cw.wt[4 * ci + co] = wv[ci * 16 + co];
    cw.b = *reinterpret_cast<const f32x4 *>(w + b_off);
    cw.g = *reinterpret_cast<const f32x4 *>(w + g_off);
    cw.be = *reinterpret_cast<const f32x4 *>(w + be_off);
  };
  f32x4 wss[4][8], bss[2];   // scale/shift Linear: W[row][e] at ((e & 3) * 16 + row) * 4 + (e >> 2): f32x4 (e & 3) * 16 + row
  {
    const f32x4 *wv = reinterpret_cast<const f32x4 *>(w + ss_w);
#pragma unroll
    for (int kq = 0; kq < 4; ++kq)
#pragma unroll
      for (int row = 0; row < 8; ++row) wss[kq][row] = wv[kq * 16 + row];
    bss[0] = *reinterpret_cast<const f32x4 *>(w + ss_b);
    bss[1] = *reinterpret_cast<const f32x4 *>(w + ss_b + 4);
  }
  ConvW k1, k2;
  load_conv(k1, c1_w, c1_b, n1_w, n1_b);
  float x[4];
#pragma unroll
  for (int ci = 0; ci < 4; ++ci) x[ci] = X[pswz(ci, n)];
  float gq[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) gq[e] = e < E ? G[e] : 0.f;
  __builtin_amdgcn_sched_barrier(0);
  float sc[4], sh[4];
#pragma unroll
  for (int co = 0; co < 4; ++co) {
    float a = bss[0][co], b = bss[1][co];
#pragma unroll
    for (int kq = 0; kq < 4; ++kq)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        a += wss[kq][co][j] * gq[4 * j + kq];
        b += wss[kq][4 + co][j] * gq[4 * j + kq];
      }
    sc[co] = a;
    sh[co] = b;
  }
  __builtin_amdgcn_sched_barrier(0);
  load_conv(k2, c2_w, c2_b, n2_w, n2_b);   // in flight under the first conv
  __builtin_amdgcn_sched_barrier(0);
  auto conv3 = [&](const float (&in)[4], const ConvW &cw, float (&out)[4]) {
    float lft[4], rgt[4];
#pragma unroll
    for (int ci = 0; ci < 4; ++ci) {
      const float a = __shfl(in[ci], (n + 48) & 63, 64), b = __shfl(in[ci], (n + 16) & 63, 64);
      lft[ci] = has_l ? a : 0.f;
      rgt[ci] = has_r ? b : 0.f;
    }
#pragma unroll
    for (int co = 0; co < 4; ++co) {
      float acc = cw.b[co];
#pragma unroll
      for (int ci = 0; ci < 4; ++ci) {
        const f32x4 wr = cw.wt[4 * ci + co];
        acc += wr[0] * lft[ci];
        acc += wr[1] * in[ci];
        acc += wr[2] * rgt[ci];
      }
      out[co] = acc;
    }
  };
  auto gn_act = [&](float (&v)[4], const ConvW &cw, bool ss) {
#pragma unroll
    for (int co = 0; co < 4; ++co) {
      const float m = half_sum(row_pair_sum(v[co])) * 0.25f;
      const float d = v[co] - m;
      const float rs = __builtin_amdgcn_rsqf(half_sum(row_pair_sum(d * d)) * 0.25f + 1e-5f);
      float y = d * rs * cw.g[co] + cw.be[co];
      if (ss) y = y * sc[co] + sh[co];
      v[co] = silu(y);
    }
  };
  float y[4], z[4];
  conv3(x, k1, y);
  gn_act(y, k1, true);
  conv3(y, k2, z);
  gn_act(z, k2, false);
#pragma unroll
  for (int co = 0; co < 4; ++co) X[pswz(co, n)] = x[co] + z[co];
}

// to_out of LinearAttention: Conv1d(128 -> C, k = 1) -> LayerNorm over the channels -> residual add
// (resnets.py:211-235, Residual(PreNorm(...)) :59-65,116-124), as ONE phase of the 64-column engine: the LayerNorm
// statistics of a column are merged across the waves that hold its rows -- per-wave (sum, M2 about the wave's own
// mean) over its 16 rows, one LDS exchange, parallel-variance merge -- and x += LN(y) g is applied to the
// accumulators.  Replaces a conv phase that stored y plus a LayerNorm phase that read it back (3 barriers).
// NPW = partner waves of the LayerNorm merge, FULL = every row of the m-tile is a channel (C >= 16): compile-time, so that
// the per-value code has no wave-uniform branches (each value used to be its own chain of basic blocks)
// FIRST: block 0's fragments of the wave's m-tile, requested by the caller ahead of the call (Frag3), or NoFirst.
template <int NT, int NPW, bool FULL, int LL = 4, class FIRST = NoFirst>
__device__ __forceinline__ void out_ln_wave(const Ctx &c, const float *wp, const float *bias, int mt0, int nt0,
                                            bool active, int p0, const float *src, int cin, float *xres, int C,
                                            const float *gain, const FIRST &first = FIRST()) {
  using GG = Geo<64>;
  constexpr int NC = 64;
  const int kq = c.lane >> 4, col = c.lane & 15;
  f32x4 acc[1][NT];
  lds_f *red1 = (lds_f *)(c.lds + GG::kMiscRed1), *red2 = (lds_f *)(c.lds + GG::kMiscRed2);
  const int row0 = 16 * mt0 + 4 * kq;
  const int nloc = C < 16 ? C : 16;  // rows behind one published pair
  f32x4 gv = f32x4{0.f, 0.f, 0.f, 0.f};
  if (active) {
    const f32x4 bv = *reinterpret_cast<const f32x4 *>(bias + row0);
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) acc[0][ni] = bv;
    gemm1_pl<kHidden / 32, 1, NT, NoPre, 1, LL, FIRST>(c, wp, mt0, nt0, c.lds + PG<LL>::kH, acc, NoPre(), first);  // the attention output's planes (cin = 128)
    gv = *reinterpret_cast<const f32x4 *>(gain + row0);
    const float inv_n = __builtin_amdgcn_rcpf((float)nloc);
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) {
      float s1 = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) s1 += (FULL || row0 + r < C) ? acc[0][ni][r] : 0.f;
      s1 = half_sum(row_pair_sum(s1));
      const float ml = s1 * inv_n;
      float s2 = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float d = (FULL || row0 + r < C) ? acc[0][ni][r] - ml : 0.f;
        s2 += d * d;
      }
      s2 = half_sum(row_pair_sum(s2));
      if (kq == 0) {
        red1[c.wave * NC + 16 * (nt0 + ni) + col] = s1;
        red2[c.wave * NC + 16 * (nt0 + ni) + col] = s2;
      }
    }
  }
  __syncthreads();
  if (active) {
    const float inv_c = __builtin_amdgcn_rcpf((float)C), inv_n = __builtin_amdgcn_rcpf((float)nloc);
    lds_f *x3 = (lds_f *)xres;
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) {
      const int cc = 16 * (nt0 + ni) + col;
      float ps[NPW], pm[NPW], tot = 0.f;
#pragma unroll
      for (int q = 0; q < NPW; ++q) {  // each partner's pair is read once
        ps[q] = red1[(p0 + q) * NC + cc];
        pm[q] = red2[(p0 + q) * NC + cc];
        tot += ps[q];
      }
      const float mean = tot * inv_c;
      float m2 = 0.f;
#pragma unroll
      for (int q = 0; q < NPW; ++q) {
        const float dm = ps[q] * inv_n - mean;
        m2 += pm[q] + (float)nloc * dm * dm;
      }
      const float rstd = __builtin_amdgcn_rsqf(m2 * inv_c + 1e-5f);
      float xn[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (FULL || row0 + r < C) {
          const int a = pswz(row0 + r, cc);
          xn[r] = x3[a] + (acc[0][ni][r] - mean) * rstd * gv[r];
          x3[a] = xn[r];
        }
      if (FULL && C >= 16) store_planes4<LL>(c.lds + PG<LL>::kX, row0, cc, xn[0], xn[1], xn[2], xn[3]);  // the new X's planes
    }
  }
  __syncthreads();
}

// The 128-channel level's to_out with its first weight fragments requested by the caller in front of the attention op
// (run_tape: OP_QKVATT followed by OP_OUTLN run as one case): cold at the top of this op they cost an L2 round trip with all
// eight waves waiting.
__device__ __forceinline__ Frag3 out_ln_request(const Ctx &c, int w_off) {
  const WStream wv(c.w + w_off, c.lane);
  Frag3 f;
#pragma unroll
  for (int pl = 0; pl < kSplit; ++pl) f.p[pl] = wv.raw_at((c.wave * (kHidden / 32)) * kFragBytes, pl * 1024);   // m-tile = wave, block 0
  return f;
}
template <int LL = 4>
__device__ __forceinline__ void out_ln_pm128(const Ctx &c, int w_off, int b_off, const float *src, int cin, float *xres, int g_off,
                                             const Frag3 &first) {
  out_ln_wave<4, 8, true, LL, Frag3>(c, c.w + w_off, c.w + b_off, c.wave, 0, true, 0, src, cin, xres, 128, c.w + g_off, first);
}
template <int LL = 4>
__device__ __forceinline__ void out_ln_pm(const Ctx &c, int w_off, int b_off, const float *src, int cin, float *xres,
                                          int C, int g_off) {
  const float *wp = c.w + w_off, *bias = c.w + b_off, *gain = c.w + g_off;
  const int mtiles = (C + 15) >> 4, w = c.wave;
  if (mtiles == 8) out_ln_wave<4, 8, true, LL>(c, wp, bias, w, 0, true, 0, src, cin, xres, C, gain);
  else if (mtiles == 4) out_ln_wave<2, 4, true, LL>(c, wp, bias, w & 3, 2 * (w >> 2), true, 4 * (w >> 2), src, cin, xres, C, gain);
  else if (mtiles == 2) out_ln_wave<1, 2, true, LL>(c, wp, bias, w & 1, w >> 1, true, 2 * (w >> 1), src, cin, xres, C, gain);
  else if (LL == 16) out_ln_wave<1, 1, true, LL>(c, wp, bias, 0, w & 3, w < 4, w & 3, src, cin, xres, C, gain);   // C = 16
  else out_ln_wave<1, 1, false, LL>(c, wp, bias, 0, w & 3, w < 4, w & 3, src, cin, xres, C, gain);
}

// Rows of the per-column LayerNorm statistics the fused qkv + attention phase needs (see qkv_att_pm): wave w owns columns
// 8 w .. 8 w + 7, lane = (row part, column), two passes over the C / 8 values in its registers (the reference's mean,
// then sum (x - mean)^2), the parts meet through DPP / permlane swaps, (mean, rstd) go to LDS.
template <int RP>  // rows per lane: C / 8
__device__ __forceinline__ void column_stats8(const Ctx &c, const float *src, float inv_c) {
  constexpr int NC = 64;
  using GG = Geo<NC>;
  const lds_f *s3 = (const lds_f *)src;
  const int n = 8 * c.wave + (c.lane & 7), rp = c.lane >> 3;
  float v[RP], sum = 0.f;
#pragma unroll
  for (int i = 0; i < RP; ++i) {
    v[i] = s3[pswz(rp + 8 * i, n)];
    sum += v[i];
  }
  auto all_parts = [](float x) {  // lanes differing in bits 3, 4, 5
    x += dpp_mov<0x128>(x);  // row_ror:8
    return half_sum(row_pair_sum(x));
  };
  const float mean = all_parts(sum) * inv_c;
  float m2 = 0.f;
#pragma unroll
  for (int i = 0; i < RP; ++i) {
    const float d = v[i] - mean;
    m2 = fmaf(d, d, m2);
  }
  const float rstd = __builtin_amdgcn_rsqf(all_parts(m2) * inv_c + 1e-5f);
  if (rp == 0) {
    ((lds_f *)(c.lds + GG::kMiscRed1))[n] = mean;
    ((lds_f *)(c.lds + GG::kMiscRed2))[n] = rstd;
  }
}

// Residual(PreNorm(LinearAttention)) up to its to_out conv (resnets.py:104-124, 211-235) as ONE phase of the position-major
// engine: PreNorm LayerNorm, to_qkv 1x1 conv and the attention core of all four heads, with q, k and v never leaving the
// accumulators (they used to be stored as a [384][64] f32 block -- 96 KiB through ds_write_b32 -- and read back by a
// separate three-barrier attention phase).
//   * Wave = (head h, channel half): its three m-tiles are rows 32 h + 16 half .. + 15 of q, of k and of v (to_qkv's own
//     row order: m-tiles 2 h + half, + 8, + 16).  In the C layout of the MFMA a lane then holds, for ONE sample (lane & 15),
//     four channels (4 (lane >> 4) + r) at all four positions (the n-tiles), of q, k and v alike.
//   * LayerNorm is folded into the conv for the 16 | C levels:  W (g (x - mean) rstd) = rstd (W' x - mean s),  W' = W diag(g),
//     s = W' 1  (both prepared on the host); the column statistics are taken while the first weight fragments are on their
//     way (column_stats8) and applied to the accumulators after the one barrier that publishes them.  The 4-channel level
//     normalises its columns in the lanes and multiplies with K = 4 f32 MFMAs (the packed A fragment's first element is
//     the operand as it stands).
//   * Attention, reassociated at n = L = 4:  out[e][n] = sum_m v[e][m] A[m][n],  A[m][n] = scale / sum_d exp(q[d][n] - max_n)
//     * sum_d softmax_m(k[d])[m] exp(q[d][n] - max_n).  The key softmax over a sample's 4 positions and the products with v
//     are in-lane; the sums over d (32 channels of the head) are in-lane over the lane's 4 channels, permlane swaps over
//     the four row quarters, and ONE exchange through LDS between the two waves of the head: each publishes
//     (max, sum, A) taken over its own 16 channels and merges the partner's like two blocks of an online softmax.
//   * The output leaves as split-f16 planes (it is only ever the B operand of to_out), into the H-plane region.
// Barriers: statistics, exchange, end (the two phases this replaces had five).
constexpr int kAttExch = 512 * 64;   // floats [32768, 35840): 8 waves x 24 rows x 16 samples, behind the X planes
static_assert(kAttExch + 8 * 24 * 16 <= Geo<64>::kArena, "attention exchange slots");
template <int KB32>   // C / 32 (0: the 4-channel level)
__device__ __forceinline__ void qkv_att_pm(const Ctx &c, int w_off, int s_off, const float *src, int C) {
  constexpr int NC = 64;
  using GG = Geo<NC>;
  if (GLDM_SKIP(c, 8)) return;
  const int head = c.wave >> 1, half = c.wave & 1;
  const int mt0 = 2 * head + half;   // q rows; k: m-tile + 8, v: + 16
  const int sm = c.lane & 15, kq = c.lane >> 4;
  f32x4 acc[3][4];
  if constexpr (KB32 == 0) {
    const int n = c.lane;
    const lds_f *s3 = (const lds_f *)src;
    float x[4];
#pragma unroll
    for (int ci = 0; ci < 4; ++ci) x[ci] = s3[pswz(ci, n)];
    const float mean = (x[0] + x[1] + x[2] + x[3]) * 0.25f;
    float vt = 0.f;
#pragma unroll
    for (int ci = 0; ci < 4; ++ci) {
      const float d = x[ci] - mean;
      vt += d * d;
    }
    const float rstd = __builtin_amdgcn_rsqf(vt * 0.25f + 1e-5f);
    // every wave normalises all 64 columns itself and writes the same values to the same scratch rows, then reads its
    // own writes in B-operand order (lane col + 16 k = y[k] of column 16 j + col): no barrier
    lds_f *ysc = (lds_f *)(c.lds + GG::kMiscRed1);   // [4][64]
#pragma unroll
    for (int k = 0; k < 4; ++k) ysc[64 * k + n] = (x[k] - mean) * rstd;
    const WStream wv(c.w + w_off, c.lane);
    f32x4 f[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) f[t] = wv[(size_t)(mt0 + 8 * t) * 64];
    float b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = ysc[64 * kq + 16 * j + sm];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(f[t][0], b[j], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
  } else {
    const float *wp = c.w + w_off, *srow = c.w + s_off;
    f32x4 sv[3];
#pragma unroll
    for (int mi = 0; mi < 3; ++mi) sv[mi] = *reinterpret_cast<const f32x4 *>(srow + 16 * (mt0 + 8 * mi) + 4 * kq);
#pragma unroll
    for (int mi = 0; mi < 3; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float inv_c = __builtin_amdgcn_rcpf((float)C);  // C: a power of two -> exact
    auto stats = [&]() { column_stats8<4 * KB32>(c, src, inv_c); };
    gemm1_pl<KB32, 3, 4, decltype(stats), 8>(c, wp, mt0, 0, c.lds + kPlaneX, acc, stats);
    __syncthreads();  // every column's (mean, rstd) is in LDS
    const lds_f *mean3 = (const lds_f *)(c.lds + GG::kMiscRed1), *rstd3 = (const lds_f *)(c.lds + GG::kMiscRed2);
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const float rstd = rstd3[16 * ni + sm];
      const float mr = mean3[16 * ni + sm] * rstd;
#pragma unroll
      for (int mi = 0; mi < 3; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[mi][ni][r] = acc[mi][ni][r] * rstd - mr * sv[mi][r];
    }
  }
  // ---- keys: softmax over the sample's 4 positions, per channel (in lane)
  float kn[4][4];   // [position][channel r]
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float km = fmaxf(fmaxf(acc[1][0][r], acc[1][1][r]), fmaxf(acc[1][2][r], acc[1][3][r]));
    float ks = 0.f;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      kn[p][r] = fast_exp(acc[1][p][r] - km);
      ks += kn[p][r];
    }
    const float inv = __builtin_amdgcn_rcpf(ks);
#pragma unroll
    for (int p = 0; p < 4; ++p) kn[p][r] *= inv;
  }
  // ---- queries: exp(q - max) and their sum over this wave's 16 channels of the head, per position
  float qe[4][4], qm[4], qs[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    float m = fmaxf(fmaxf(acc[0][p][0], acc[0][p][1]), fmaxf(acc[0][p][2], acc[0][p][3]));
    m = half_max(row_pair_max(m));
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      qe[p][r] = fast_exp(acc[0][p][r] - m);
      sum += qe[p][r];
    }
    qm[p] = m;
    qs[p] = half_sum(row_pair_sum(sum));
  }
  // ---- A[m][n] over this wave's channels
  float A[4][4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      float a = kn[m][0] * qe[n][0];
#pragma unroll
      for (int r = 1; r < 4; ++r) a = fmaf(kn[m][r], qe[n][r], a);
      A[m][n] = half_sum(row_pair_sum(a));
    }
  // ---- exchange with the other half of the head (wave ^ 1): rows [max 4 | sum 4 | A 16] x 16 samples
  lds_f *ex = (lds_f *)(c.lds + kAttExch);
  if (kq == 0) {
    lds_f *mine = ex + c.wave * 384 + sm;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      mine[16 * n] = qm[n];
      mine[16 * (4 + n)] = qs[n];
#pragma unroll
      for (int m = 0; m < 4; ++m) mine[16 * (8 + 4 * m + n)] = A[m][n];
    }
  }
  __syncthreads();
  const lds_f *oth = ex + (c.wave ^ 1) * 384 + sm;
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    const float mo = oth[16 * n], so = oth[16 * (4 + n)];
    const float m = fmaxf(qm[n], mo);
    const float fs = fast_exp(qm[n] - m), fo = fast_exp(mo - m);
    const float sc = 0.17677669529663687f * __builtin_amdgcn_rcpf(qs[n] * fs + so * fo);   // dim_head ** -0.5 / sum
#pragma unroll
    for (int mm = 0; mm < 4; ++mm) A[mm][n] = (A[mm][n] * fs + oth[16 * (8 + 4 * mm + n)] * fo) * sc;
  }
  // ---- out[e][n] = sum_m v[e][m] A[m][n] for the lane's 4 channels e, as planes (32-channel block = head)
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    float o[4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
      o[r] = acc[2][0][r] * A[0][n] + acc[2][1][r] * A[1][n] + acc[2][2][r] * A[2][n] + acc[2][3][r] * A[3][n];
    store_planes4(c.lds + kPlaneH, kDimHead * head + 16 * half + 4 * kq, 16 * n + sm, o[0], o[1], o[2], o[3]);
  }
  __syncthreads();
}


// ---- attention block of the 16-position 64-column engine: qkv_att16_pm (behind quad16_narrow.h, whose core it shares).
// (Rounds 4-5: qkv_ln16_pm + attention16_pm, q | k | v through LDS as a [384][64] f32 block over both plane regions.)
// The four zero entries either side of every plane row (8 blocks x 3 planes x 4 g rows x 8 entries) of the 16-position
// engine.  Two things overwrite them: the f32 rows of the 256-channel level's output (rows 128 .. 255 of X lie over the
// first H-plane blocks) and the weight ring of the wave-local narrow levels (quad16_narrow.h): restored behind each.
__device__ __forceinline__ void zero_plane_pads16(float *lds, int t) {
  using G = PG<16>;
  lds_u4 *pl = (lds_u4 *)(lds + G::kH);
  const u32x4 z4 = u32x4{0u, 0u, 0u, 0u};
  for (int i = t; i < 8 * 12 * 8; i += 512) {
    const int rowi = i >> 3, e8 = i & 7;
    pl[rowi * G::kCols + (e8 < 4 ? e8 : 64 + e8)] = z4;
  }
}

// A whole ResnetBlock of the position-major engine as ONE op: conv1 (GroupNorm, scale/shift, SiLU -> H), barrier, conv2
// (GroupNorm, SiLU, X += ...).  Straight-line code, so the first weight fragments of conv2 can be requested while conv1's
// epilogue runs and be there when its k-loop starts: requested cold at the top of the loop they cost ~1.8 k cycles per
// conv (a build that loads no fragments at all runs the step 8 % faster; across two tape ops the compiler waits for
// such loads at the op switch).  One tape decode and dispatch less, too.
template <int MT, int P0, int NP, int GK, int LL = 4>
__device__ __forceinline__ void resblock_pm_wave(const Ctx &c, const float *wp1, const float *b1, const float *wp2,
                                                 const float *b2, int mt0, float *X, float *H, int C,
                                                 const GnEpilogue &g1, const GnEpilogue &g2, long long *stamp2,
                                                 bool live = true) {
  PreA<MT> pa;
  const WStream wv2(wp2, c.lane);
  const int cin = C < 32 ? 32 : C;   // a 16-channel level is one zero-padded 32-channel block of planes
  auto ask = [&]() { pa.request(wv2, mt0, cin); };
  conv_pm3_wave<MT, P0, NP, GK, 1, NoPreA, decltype(ask), LL>(c, wp1, b1, mt0, X, C, H, C, false, g1, NoPreA(), ask, live);
  __syncthreads();
  if (stamp2 && c.tid == 0) *stamp2 = (long long)__builtin_readcyclecounter();
  conv_pm3_wave<MT, P0, NP, GK, 2, PreA<MT>, NoHook, LL>(c, wp2, b2, mt0, H, C, X, C, false, g2, pa, NoHook(), live);
}
template <int LL = 4>
__device__ __forceinline__ void resblock_pm(const Ctx &c, int w1, int b1, int w2, int b2, float *X, float *H, int C,
                                            const GnEpilogue &g1, const GnEpilogue &g2, long long *stamp2) {
  const float *wp1 = c.w + w1, *bp1 = c.w + b1, *wp2 = c.w + w2, *bp2 = c.w + b2;
  const int w = c.wave;
  if (C == 256) resblock_pm_wave<2, 0, 4, 0, LL>(c, wp1, bp1, wp2, bp2, 2 * w, X, H, C, g1, g2, stamp2);
  else if (C == 128) resblock_pm_wave<1, 0, 4, 0, LL>(c, wp1, bp1, wp2, bp2, w, X, H, C, g1, g2, stamp2);
  else if (C == 64) {
    if (w < 4) resblock_pm_wave<1, 0, 2, 1, LL>(c, wp1, bp1, wp2, bp2, w & 3, X, H, C, g1, g2, stamp2);
    else resblock_pm_wave<1, 2, 2, 1, LL>(c, wp1, bp1, wp2, bp2, w & 3, X, H, C, g1, g2, stamp2);
  } else if (C == 32 || LL == 4) {
    const int pw = w >> 1;
    if (pw == 0) resblock_pm_wave<1, 0, 1, 2, LL>(c, wp1, bp1, wp2, bp2, w & 1, X, H, C, g1, g2, stamp2);
    else if (pw == 1) resblock_pm_wave<1, 1, 1, 2, LL>(c, wp1, bp1, wp2, bp2, w & 1, X, H, C, g1, g2, stamp2);
    else if (pw == 2) resblock_pm_wave<1, 2, 1, 2, LL>(c, wp1, bp1, wp2, bp2, w & 1, X, H, C, g1, g2, stamp2);
    else resblock_pm_wave<1, 3, 1, 2, LL>(c, wp1, bp1, wp2, bp2, w & 1, X, H, C, g1, g2, stamp2);
  } else if constexpr (LL == 16) {   // C = 16: one m-tile, a tile per wave of 0-3; waves 4-7 shadow them
    const int t = w & 3;
    const bool live = w < 4;
    if (t == 0) resblock_pm_wave<1, 0, 1, 3, 16>(c, wp1, bp1, wp2, bp2, 0, X, H, C, g1, g2, stamp2, live);
    else if (t == 1) resblock_pm_wave<1, 1, 1, 3, 16>(c, wp1, bp1, wp2, bp2, 0, X, H, C, g1, g2, stamp2, live);
    else if (t == 2) resblock_pm_wave<1, 2, 1, 3, 16>(c, wp1, bp1, wp2, bp2, 0, X, H, C, g1, g2, stamp2, live);
    else resblock_pm_wave<1, 3, 1, 3, 16>(c, wp1, bp1, wp2, bp2, 0, X, H, C, g1, g2, stamp2, live);
  }
  __syncthreads();
}

// dst[cout][NC] = W * im2col(src[cin][NC]) + bias: the waves split the output rows (all n-tiles each).
// Ends with a barrier.  alias: dst overlaps src -> all reads complete (barrier) before any store.
// Output widths are 16 x {1, 2, 4, 8, 12, 16} rows (validate() enforces it).
// (Tried and dropped: running the <= 64-channel levels column-parallel, one wave per n-tile with no
// barriers inside the level: those phases are bound by per-wave issue, not by the barriers, and with
// half the waves active every op took 1.7-2x longer.  And the opposite, 8 waves per 32-column tile
// with the statistics of a wide group exchanged between wave pairs: correct, but at 128 VGPRs per
// wave the k-loops spill and the launch was 5-7 % slower than with 4 waves.)
template <int NC, int L>
__device__ __forceinline__ void conv_gemm(const Ctx &c, int w_off, int b_off, const float *src, int cin, int ktaps,
                                          float *dst, int cout, bool alias, int act = 0,
                                          const GnEpilogue &g = GnEpilogue{0, 0, 0, -1, 0, 0, 0, 0, nullptr}) {
  if (GLDM_SKIP(c, 8)) return;
  const float *wp = c.w + w_off;
  const float *bias = b_off >= 0 ? c.w + b_off : nullptr;
  const int mtiles = (cout + 15) >> 4;
  const int w = c.wave;
  // G3: k = 3 taps (<= 2 m-tiles per sweep), G1: 1x1
#define GLDM_G3(MT, NT, P, mt0, nt0, on) gemm_passes<NC, L, 3, MT, NT, P>(c, wp, mt0, nt0, on, src, cin, dst, cout, bias, alias, act, g)
#define GLDM_G1(MT, NT, P, mt0, nt0, on) gemm_passes<NC, L, 1, MT, NT, P>(c, wp, mt0, nt0, on, src, cin, dst, cout, bias, alias, act, g)
  if constexpr (NC == 64) {
    if (ktaps == 3) {
      // 64-column engines: 8 waves share the m-tiles (L = 4: the three taps tie the 4 position tiles together; L = 16:
      // tiles of 4 positions x 4 samples, taps by shifted plane reads).  Only the levels' down convs come this way.
      constexpr int LL = L == 16 ? 16 : 4;
      if (LL == 4 && (cin & 15)) conv_pm3_cin4(c, wp, bias, src, dst, cout, alias);
      else if (mtiles == 16) conv_pm3_wave<2, 0, 4, 0, 0, NoPreA, NoHook, LL>(c, wp, bias, 2 * w, src, cin, dst, cout, alias, g);
      else if (mtiles == 8) conv_pm3_wave<1, 0, 4, 0, 0, NoPreA, NoHook, LL>(c, wp, bias, w, src, cin, dst, cout, alias, g);
      else if (mtiles == 4) {
        if (w < 4) conv_pm3_wave<1, 0, 2, 1, 0, NoPreA, NoHook, LL>(c, wp, bias, w & 3, src, cin, dst, cout, alias, g);
        else conv_pm3_wave<1, 2, 2, 1, 0, NoPreA, NoHook, LL>(c, wp, bias, w & 3, src, cin, dst, cout, alias, g);
      } else {  // 2 m-tiles: wave = (m-tile, position / tile)
        const int pw = w >> 1;
        if (pw == 0) conv_pm3_wave<1, 0, 1, 2, 0, NoPreA, NoHook, LL>(c, wp, bias, w & 1, src, cin, dst, cout, alias, g);
        else if (pw == 1) conv_pm3_wave<1, 1, 1, 2, 0, NoPreA, NoHook, LL>(c, wp, bias, w & 1, src, cin, dst, cout, alias, g);
        else if (pw == 2) conv_pm3_wave<1, 2, 1, 2, 0, NoPreA, NoHook, LL>(c, wp, bias, w & 1, src, cin, dst, cout, alias, g);
        else conv_pm3_wave<1, 3, 1, 2, 0, NoPreA, NoHook, LL>(c, wp, bias, w & 1, src, cin, dst, cout, alias, g);
      }
    }
    // 1x1 layers (layout agnostic): 8 waves x 4 n-tiles; also the fused set abstraction
    else if (mtiles == 16) GLDM_G1(2, 4, 1, 2 * w, 0, true);
    else if (mtiles == 12) GLDM_G1(3, 2, 1, 3 * (w & 3), 2 * (w >> 2), true);
    else if (mtiles == 8) GLDM_G1(1, 4, 1, w, 0, true);
    else if (mtiles == 4) GLDM_G1(1, 2, 1, w & 3, 2 * (w >> 2), true);
    else if (mtiles == 2) GLDM_G1(1, 1, 1, w & 1, w >> 1, true);
    else GLDM_G1(1, 1, 1, 0, w & 3, w < 4);
  } else if (ktaps == 3) {
    if (c.nta == 1) {  // tail workgroup: only columns 0..15 are live
      if (mtiles == 16) GLDM_G3(2, 1, 2, 4 * w, 0, true);
      else if (mtiles == 12) GLDM_G3(1, 1, 3, 3 * w, 0, true);
      else if (mtiles == 8) GLDM_G3(2, 1, 1, 2 * w, 0, true);
      else GLDM_G3(1, 1, 1, w < mtiles ? w : 0, 0, w < mtiles);
    } else if (mtiles == 16) GLDM_G3(2, 2, 2, 4 * w, 0, true);
    else if (mtiles == 12) GLDM_G3(1, 2, 3, 3 * w, 0, true);
    else if (mtiles == 8) GLDM_G3(2, 2, 1, 2 * w, 0, true);
    else if (mtiles == 4) GLDM_G3(1, 2, 1, w, 0, true);
    else if (mtiles == 2) GLDM_G3(1, 1, 1, w & 1, w >> 1, true);
    else GLDM_G3(1, 1, 1, 0, w & 1, w < 2);
  } else {
    if (c.nta == 1) {
      if (mtiles == 16) GLDM_G1(4, 1, 1, 4 * w, 0, true);
      else if (mtiles == 12) GLDM_G1(3, 1, 1, 3 * w, 0, true);
      else if (mtiles == 8) GLDM_G1(2, 1, 1, 2 * w, 0, true);
      else GLDM_G1(1, 1, 1, w < mtiles ? w : 0, 0, w < mtiles);
    } else if (mtiles == 16) GLDM_G1(4, 2, 1, 4 * w, 0, true);
    else if (mtiles == 12) GLDM_G1(3, 2, 1, 3 * w, 0, true);
    else if (mtiles == 8) GLDM_G1(2, 2, 1, 2 * w, 0, true);
    else if (mtiles == 4) GLDM_G1(1, 2, 1, w, 0, true);
    else if (mtiles == 2) GLDM_G1(1, 1, 1, w & 1, w >> 1, true);
    else GLDM_G1(1, 1, 1, 0, w & 1, w < 2);
  }
#undef GLDM_G3
#undef GLDM_G1
  __syncthreads();
}

// ----------------------------------------------------------- LayerNorm ----
// dst = LN_channels(src) * g  (or res += LN(src) * g when res != null).  Row slot = wave*kRP + sub
// (kSlots = kWaves * kRP of them), rows slot + kSlots i.
template <int NC, int ITERS>
__device__ __forceinline__ void layer_norm_rows(const Ctx &c, const float *src, float *dst, float *res, int C,
                                                int g_off) {
  using GG = Geo<NC>;
  float *red1 = c.lds + GG::kMiscRed1, *red2 = c.lds + GG::kMiscRed2;
  const int n = c.lane & (NC - 1), slot = c.wave * GG::kRP + c.lane / NC;
  const lds_f *s3 = (const lds_f *)src;
  const float *g = c.w + g_off;
  float v[ITERS], gv[ITERS];  // the gains are requested up front: their L2 round trip hides behind the statistics
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < ITERS; ++i) {
    const int row = slot + GG::kSlots * i;
    const float x = s3[swz<NC>(row < C ? row : 0, n)];
    gv[i] = g[row < C ? row : 0];
    v[i] = row < C ? x : 0.f;
    sum += v[i];
  }
  if (GG::kRP == 2) sum = half_sum(sum);
  red1[c.wave * NC + n] = sum;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int q = 0; q < GG::kWaves; ++q) tot += red1[q * NC + n];
  const float mean = tot / (float)C;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < ITERS; ++i) {
    const int row = slot + GG::kSlots * i;
    const float d = row < C ? v[i] - mean : 0.f;
    sq += d * d;
  }
  if (GG::kRP == 2) sq = half_sum(sq);
  red2[c.wave * NC + n] = sq;
  __syncthreads();
  float vt = 0.f;
#pragma unroll
  for (int q = 0; q < GG::kWaves; ++q) vt += red2[q * NC + n];
  const float rstd = __builtin_amdgcn_rsqf(vt / (float)C + 1e-5f);
  lds_f *d3 = (lds_f *)dst, *r3 = (lds_f *)res;
#pragma unroll
  for (int i = 0; i < ITERS; ++i) {
    const int row = slot + GG::kSlots * i;
    if (row < C) {
      const float y = (v[i] - mean) * rstd * gv[i];
      const int a = swz<NC>(row, n);
      if (res) r3[a] = r3[a] + y;
      else d3[a] = y;
    }
  }
  __syncthreads();
}

// One-exchange form for the 64-column engine (lane = column, wave = row slot): every wave reduces its rows to
// (sum, M2 about its own mean), ONE barrier, then the parallel-variance merge of the 8 pairs: two barriers per
// LayerNorm instead of three.
template <int ITERS>
__device__ __forceinline__ void layer_norm_rows_1x(const Ctx &c, const float *src, float *dst, float *res, int C,
                                                   int g_off) {
  constexpr int NC = 64;
  using GG = Geo<NC>;
  lds_f *red1 = (lds_f *)(c.lds + GG::kMiscRed1), *red2 = (lds_f *)(c.lds + GG::kMiscRed2);
  const int n = c.lane, slot = c.wave;
  const lds_f *s3 = (const lds_f *)src;
  const float *g = c.w + g_off;
  float v[ITERS], gv[ITERS];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < ITERS; ++i) {
    const int row = slot + GG::kSlots * i;
    const float x = s3[swz<NC>(row < C ? row : 0, n)];
    gv[i] = g[row < C ? row : 0];
    v[i] = row < C ? x : 0.f;
    sum += v[i];
  }
  const int cnt = slot < C ? (C - slot + GG::kSlots - 1) / GG::kSlots : 0;  // rows of this slot (wave uniform)
  const float mloc = sum * __builtin_amdgcn_rcpf((float)(cnt > 0 ? cnt : 1));  // C, cnt: powers of two -> exact
  float m2 = 0.f;
#pragma unroll
  for (int i = 0; i < ITERS; ++i) {
    const int row = slot + GG::kSlots * i;
    const float d = row < C ? v[i] - mloc : 0.f;
    m2 += d * d;
  }
  red1[slot * NC + n] = sum;
  red2[slot * NC + n] = m2;
  __syncthreads();
  float tot = 0.f, ps[GG::kWaves], pm[GG::kWaves];
#pragma unroll
  for (int q = 0; q < GG::kWaves; ++q) {
    ps[q] = red1[q * NC + n];
    pm[q] = red2[q * NC + n];
    tot += ps[q];
  }
  const float inv_c = __builtin_amdgcn_rcpf((float)C);
  const float mean = tot * inv_c;
  float vt = 0.f;
#pragma unroll
  for (int q = 0; q < GG::kWaves; ++q) {
    const int cq = q < C ? (C - q + GG::kSlots - 1) / GG::kSlots : 0;  // wave uniform
    const float fq = (float)cq, iq = __builtin_amdgcn_rcpf((float)(cq > 0 ? cq : 1));
    const float dm = ps[q] * iq - mean;
    vt += cq > 0 ? pm[q] + fq * dm * dm : 0.f;
  }
  const float rstd = __builtin_amdgcn_rsqf(vt * inv_c + 1e-5f);
  lds_f *d3 = (lds_f *)dst, *r3 = (lds_f *)res;
#pragma unroll
  for (int i = 0; i < ITERS; ++i) {
    const int row = slot + GG::kSlots * i;
    if (row < C) {
      const float y = (v[i] - mean) * rstd * gv[i];
      const int a = swz<NC>(row, n);
      if (res) r3[a] = r3[a] + y;
      else d3[a] = y;
    }
  }
  __syncthreads();
}

template <int NC>
__device__ __forceinline__ void layer_norm_pass(const Ctx &c, const float *src, float *dst, float *res, int C, int g_off) {
  if (GLDM_SKIP(c, 2)) return;
  C = __builtin_amdgcn_readfirstlane(C);
  if constexpr (NC == 64) {
    const int it1 = (C + Geo<NC>::kSlots - 1) / Geo<NC>::kSlots;
    if (it1 <= 1) layer_norm_rows_1x<1>(c, src, dst, res, C, g_off);
    else if (it1 <= 4) layer_norm_rows_1x<4>(c, src, dst, res, C, g_off);
    else if (it1 <= 8) layer_norm_rows_1x<8>(c, src, dst, res, C, g_off);
    else layer_norm_rows_1x<16>(c, src, dst, res, C, g_off);
    return;
  }
  const int it = (C + Geo<NC>::kSlots - 1) / Geo<NC>::kSlots;
  if (it <= 1) layer_norm_rows<NC, 1>(c, src, dst, res, C, g_off);
  else if (it <= 2) layer_norm_rows<NC, 2>(c, src, dst, res, C, g_off);
  else if (it <= 4) layer_norm_rows<NC, 4>(c, src, dst, res, C, g_off);
  else if (it <= 8) layer_norm_rows<NC, 8>(c, src, dst, res, C, g_off);
  else layer_norm_rows<NC, 16>(c, src, dst, res, C, g_off);
}

// -------------------------------------------------- linear attention -------
// qkv: [192][NC] = q(2 heads x 32) | k | v for one head pair; writes 64 rows of o.
// (Sample-major 32-column engine; the 64-column engine has attention_pair_pm.)
template <int NC, int L>
__device__ __forceinline__ void attention_pair(const Ctx &c, float *qkv, float *o_rows) {
  if (GLDM_SKIP(c, 4)) return;
  using GG = Geo<NC>;
  static_assert(NC == 32 && GG::kWaves == 4 && (L == 4 || L == 16), "sample-major engine geometry");
  const lds_f *q3 = (const lds_f *)qkv;
  lds_f *o3 = (lds_f *)o_rows;
  if constexpr (L == 4 && NC == 32 && GG::kWaves == 4) {
    // wave = (head of the pair, 16-column half); lane = (column, part): 8 of the head's 32 channels.  The softmax
    // statistics and the 4 x 4 matrix A = softmax_n(k)^T softmax_d(q) combine over the parts with permlane swaps
    // inside the wave: no LDS exchange and no barrier but the final one.
    const int h2 = c.wave >> 1, nn = 16 * (c.wave & 1) + (c.lane & 15), pt = c.lane >> 4, d0 = 8 * pt, sb = nn & ~3;
    const int qr = h2 * kDimHead + d0, kr = 64 + h2 * kDimHead + d0, vr = 128 + h2 * kDimHead + d0;
    float q[8];
    float qmax = -3.0e38f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      q[i] = q3[swz<NC>(qr + i, nn)];
      qmax = fmaxf(qmax, q[i]);
    }
    f32x4 kv[8], vv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) kv[i] = *(const lds_f4 *)(q3 + swz<NC>(kr + i, sb));
#pragma unroll
    for (int i = 0; i < 8; ++i) vv[i] = *(const lds_f4 *)(q3 + swz<NC>(vr + i, sb));
    qmax = half_max(row_pair_max(qmax));
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
    qsum = half_sum(row_pair_sum(qsum));
    a0 = half_sum(row_pair_sum(a0)); a1 = half_sum(row_pair_sum(a1));
    a2 = half_sum(row_pair_sum(a2)); a3 = half_sum(row_pair_sum(a3));
    const float sc = 0.17677669529663687f * __builtin_amdgcn_rcpf(qsum);  // dim_head ** -0.5 / sum
    a0 *= sc; a1 *= sc; a2 *= sc; a3 *= sc;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      o3[swz<NC>(h2 * kDimHead + d0 + i, nn)] = vv[i].x * a0 + vv[i].y * a1 + vv[i].z * a2 + vv[i].w * a3;
    __syncthreads();
  } else {
    // L = 16 (pose decoder): 2 samples x 16 positions per tile, 4 waves.  Both softmaxes are normalised in place
    // first (keys over a sample's 16 positions: two lanes per (head, channel, sample); queries over the 32 channels,
    // scaled by dim_head^-0.5: four lanes per (head, column)), then a wave per (head, sample) takes
    // A = Kn^T Qn (16 x 16, K = 32) and out = V A (32 x 16, K = 16) as 16 MFMAs: A's accumulator registers are the
    // B operand of the second product as they stand (k-step r = key positions {4 kq + r}).  (The first form had every
    // lane recompute A with 512 exponentials: 30-80 k cycles per op, a third of the whole decode.)
    static_assert(NC == 32, "decoder tile");
    const int t = c.tid;
    lds_f *w3 = (lds_f *)qkv;
    {
      const int item = t >> 1, half = t & 1, h = item >> 6, d = (item >> 1) & 31, sm = item & 1;
      const int row = 64 + h * kDimHead + d, col0 = sm * 16 + 8 * half;
      float k[8];
      float km = -3.0e38f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        k[j] = w3[swz<NC>(row, col0 + j)];
        km = fmaxf(km, k[j]);
      }
      km = dpp_max<0xB1>(km);  // the other half of the sample's positions: lane ^ 1
      float ks = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        k[j] = fast_exp(k[j] - km);
        ks += k[j];
      }
      ks += dpp_mov<0xB1>(ks);
      const float inv = __builtin_amdgcn_rcpf(ks);
#pragma unroll
      for (int j = 0; j < 8; ++j) w3[swz<NC>(row, col0 + j)] = k[j] * inv;
    }
    {
      const int item = t >> 2, qt = t & 3, h = item >> 5, col = item & 31;
      const int row0 = h * kDimHead + 8 * qt;
      float q[8];
      float qm = -3.0e38f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        q[j] = w3[swz<NC>(row0 + j, col)];
        qm = fmaxf(qm, q[j]);
      }
      qm = dpp_max<0xB1>(qm);
      qm = dpp_max<0x4E>(qm);  // the four channel quarters of a column: one quad
      float qs = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        q[j] = fast_exp(q[j] - qm);
        qs += q[j];
      }
      qs += dpp_mov<0xB1>(qs);
      qs += dpp_mov<0x4E>(qs);
      const float sc = 0.17677669529663687f * __builtin_amdgcn_rcpf(qs);  // dim_head ** -0.5 / sum
#pragma unroll
      for (int j = 0; j < 8; ++j) w3[swz<NC>(row0 + j, col)] = q[j] * sc;
    }
    __syncthreads();
    const int h = c.wave >> 1, sm = c.wave & 1, m = c.lane & 15, kq = c.lane >> 4;
    f32x4 am = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 8; ++j) {  // channel d = 4 j + kq
      const float ka = q3[swz<NC>(64 + h * kDimHead + 4 * j + kq, sm * 16 + m)];  // Kn^T[key position m][d]
      const float qb = q3[swz<NC>(h * kDimHead + 4 * j + kq, sm * 16 + m)];       // Qn[d][query position m]
      am = __builtin_amdgcn_mfma_f32_16x16x4f32(ka, qb, am, 0, 0, 0);
    }
    // am: lane (query position m, kq), register r = A[key position 4 kq + r][m]
    f32x4 o0 = f32x4{0.f, 0.f, 0.f, 0.f}, o1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) {  // k-step r: key positions 4 kq + r
      const float v0 = q3[swz<NC>(128 + h * kDimHead + m, sm * 16 + 4 * kq + r)];
      const float v1 = q3[swz<NC>(128 + h * kDimHead + 16 + m, sm * 16 + 4 * kq + r)];
      o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(v0, am[r], o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(v1, am[r], o1, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      o3[swz<NC>(h * kDimHead + 4 * kq + r, sm * 16 + m)] = o0[r];
      o3[swz<NC>(h * kDimHead + 16 + 4 * kq + r, sm * 16 + m)] = o1[r];
    }
    __syncthreads();
  }
}


#include "quad_narrow.h"
#include "quad16_narrow.h"

// ---------------------------------------------------------- the network ----
// Step-segment hand-off between workgroups (a tile whose steps are split over a chain of slots).
// The whole state of a tile between two steps is its latent row: NC floats.  Each float travels as ONE
// naturally aligned 8-byte granule {value, tag} written by one write-through (agent-scope) store and
// polled with agent-scope loads: a granule is never torn, so no fence, flag or ordering between
// granules is needed (MI355X_MICROARCH.md, "R2's granule").  tag = (launch epoch, step count) is
// unique per launch and step; the zero-initialised workspace holds tag 0, which no hand-off uses.
struct ChainHdr { unsigned ticket, done, epoch, error; };
constexpr int kChainHdrBytes = 256;
constexpr int kParkBytes = 256 * 64 * 4;   // one [256][64] f32 tile per workgroup (Ctx::park)
__device__ __forceinline__ unsigned chain_tag(unsigned epoch, int step) { return (epoch << 12) | (unsigned)step; }
__device__ __forceinline__ void chain_give(unsigned long long *g, float v, unsigned tag) {
  __hip_atomic_store(g, ((unsigned long long)tag << 32) | __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float chain_take(unsigned long long *g, unsigned tag, unsigned *error) {
  unsigned long long x = 0;
  for (int spin = 0; spin < (1 << 22); ++spin) {  // bounded: ~2 s; a healthy wait is a few ms at most
    x = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((unsigned)(x >> 32) == tag) return __uint_as_float((unsigned)x);
    __builtin_amdgcn_s_sleep(16);
  }
  // The wait expired: the tile's latent is lost.  The error word is what the host reads (R1dEngine checks it after
  // every chained launch and raises); the NaN makes every output of this tile NaN whether or not anyone looks.
  __hip_atomic_store(error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return __builtin_nanf("");
}

struct RunArgs {
  gldm_r1d_desc d;
  const float *weights;
  const float *temb;      // [T][E] or null
  const float *cemb;      // [n_cond][R][E]
  const float *semb;      // [n][E] per-sample embedding added to the time embedding (class conditioning), or null
  int samples_per_cond;
  const float *x_in;      // denoise: [n][L]; decode: z_h [n][D]
  int n_samples;
  const int32_t *timesteps;
  const int32_t *sample_t;
  int n_steps;
  int sched_kind, clip_sample;
  const float *sched_coef;
  const float *step_noise;
  // DDPM noise drawn IN the kernel when step_noise is null and noise_on is set: unit normals from Philox4x32-10 keyed on
  // noise_seed, counter = (global latent index noise_base + gi, position block, step): independent of tiling, batch
  // split and world size by construction
  unsigned long long noise_seed;
  long long noise_base;
  int noise_on;
  float *out0;            // denoise: x_out [n][L]; decode: tmrp [n][6]
  float *out1;            // decode: logit [n]
  float *ws;              // workspace: chain header + hand-off granules (+ the decoder's scale/shift table)
  float *park;            // position-major engine with a 256-channel last level: 64 KiB of scratch per workgroup (Ctx::park)
  const float *ss_tab;    // [n_cond][ss_stride] scale/shift rows of every ResnetBlock per conditioning cloud, or null
  int ss_stride;
  int skip;               // diagnostic phase-skip mask (GLDM_R1D_SKIP env; 0 in production)
  // Work distribution (make_plan): `slots` persistent workgroups; slot p owns tiles p, p + slots, ... (`rounds` of
  // them) for all steps; each of the `left_tiles` tiles beyond the whole rounds is cut into `chain` step segments of
  // `seglen` steps that consecutive slots run one after the other, handing the latent on through the workspace.
  int slots, rounds, left_tiles, chain, seglen;
  int stagger_ticks, n_cus;  // start offset (100 MHz ticks) of the second workgroup of a CU
  long long *stamps;         // diagnostic (GLDM_R1D_STAMP): cycle counter at every op of the last step, block 0
  // Position-major engine: the step program, built on the host (launch_r1d), travels in the kernel arguments: an op's
  // 12 ints are three scalar loads from the kernarg segment, straight into SGPRs.  (From the LDS copy the interpreter
  // paid three ds_read_b128, their latency and 12 v_readfirstlane per op -- ~0.7 k cycles on every wave, 28 ops a step.)
  int pm_nops;
  int pm_tape[kPmMaxOps * kOpInts];
};

// One denoiser / decoder step is a fixed program of barrier-separated ops (45 for the shipped
// denoiser: per level 2 x [conv+GN, conv+GN+residual], LN, 2 x [qkv conv, attention], out conv, LN,
// down conv).  It is written once per workgroup into LDS (12 ints per op) and interpreted by a switch
// inside the step loop, so every phase body exists once, inlined, with registers allocated across the
// whole kernel: no calls, no callee-save traffic and no spilled kernel state between phases.
enum { OP_CONV = 1, OP_RES4 = 2, OP_LN = 3, OP_ATT = 4, OP_QKVLN = 5, OP_OUTLN = 6, OP_QKVATT = 7, OP_QUAD = 8 };
// The levels of 4, 32 and 64 channels in front of a 128-channel one run wave-local (quad_narrow.h): the shipped denoiser
__host__ __device__ __forceinline__ bool quad_levels(const gldm_r1d_desc &d) {
  return d.seq_len == 4 && d.emb_dim == 16 && d.n_levels >= 3 && d.dims[0] == 4 && d.dims[1] == 32 && d.dims[2] == 64 &&
         d.dims[3] == 128 && d.lv[0].out_wq > 0 && d.lv[1].out_wq > 0 && d.lv[1].qkvn_wq > 0 && d.lv[1].down_wq > 0 &&
         d.lv[2].out_wq > 0 && d.lv[2].qkvn_wq > 0 && d.lv[2].down_wq > 0 && d.rb[2].c1_wq > 0 && d.rb[2].c2_wq > 0 &&
         d.rb[3].c1_wq > 0 && d.rb[3].c2_wq > 0 && d.rb[4].c1_wq > 0 && d.rb[4].c2_wq > 0 && d.rb[5].c1_wq > 0 && d.rb[5].c2_wq > 0;
}
// The same for the 16-position nets (quad16_narrow.h): levels of 16, 32 and 64 channels in front of a 128-channel one.  The
// scale / shift rows come from the per-cloud table (pose decoder: ss_table_rows) or the 64-wide embedding (the launcher
// checks that one of the two holds).
__host__ __device__ __forceinline__ bool quad16_levels(const gldm_r1d_desc &d) {
  return d.seq_len == 16 && d.groups == 4 && d.n_levels >= 3 && d.dims[0] == 16 && d.dims[1] == 32 && d.dims[2] == 64 && d.dims[3] == 128 &&
         d.lv[0].out_wq > 0 && d.lv[0].qkvn_wq > 0 && d.lv[0].down_wq > 0 && d.lv[1].out_wq > 0 && d.lv[1].qkvn_wq > 0 &&
         d.lv[1].down_wq > 0 && d.lv[2].out_wq > 0 && d.lv[2].qkvn_wq > 0 && d.lv[2].down_wq > 0 && d.rb[0].c1_wq > 0 &&
         d.rb[0].c2_wq > 0 && d.rb[1].c1_wq > 0 && d.rb[1].c2_wq > 0 && d.rb[2].c1_wq > 0 && d.rb[2].c2_wq > 0 && d.rb[3].c1_wq > 0 &&
         d.rb[3].c2_wq > 0 && d.rb[4].c1_wq > 0 && d.rb[4].c2_wq > 0 && d.rb[5].c1_wq > 0 && d.rb[5].c2_wq > 0;
}
// conv flags (int 7): taps | alias << 8 | GroupNorm epilogue mode << 9 | (scale/shift table offset / 4) << 12
constexpr int kFlagAlias = 1 << 8;
constexpr int kFlagFused = 1 << 11;   // position-major engine: this conv and the next tape entry are one ResnetBlock op
constexpr int kFlagTabShift = 12;

// ResnetBlock of a 4-channel level with 4-position samples (the first level of the latent denoiser) on the
// VALU of ONE wave, in registers: as two MFMA ops it is two padded 16x16 tiles and two long epilogues for
// 48 MACs per output.  lane = (channel pair h, column n): every lane keeps all 4 input channels of its column
// and produces output channels 2h, 2h+1.  Taps come from the neighbouring lanes of the quad (= the sample),
// GroupNorm (one channel per group) is a quad reduction, the two channel pairs meet through a permlane swap.
// Weights are read in their packed MFMA-fragment order: conv W[co][ci][tap] at ((ci * 16 + co) * 4 + tap),
// scale/shift Linear W[row][e] at (((e & 3) * 16 + row) * 4 + (e >> 2)).
template <int NC>
__device__ __forceinline__ void resblock4_valu(const Ctx &c, const int (&o)[kOpInts], int E) {
  using GG = Geo<NC>;
  if (c.wave != 0) return;
  const int n = c.lane & 31, h = c.lane >> 5, l = n & 3;
  const bool has_l = l != 0, has_r = l != 3;
  lds_f *X = (lds_f *)(c.lds + GG::kBufX);
  const lds_f *G = (const lds_f *)(c.lds + GG::kMiscG) + (n >> 2) * E;
  const float *w = c.w;
  const int c1_w = o[1], c1_b = o[2], n1_w = o[3], n1_b = o[4], c2_w = o[5], c2_b = o[6], n2_w = o[7], n2_b = o[8],
            ss_w = o[9], ss_b = o[10];
  // ---- every parameter of the block is requested up front (one L2 round trip for all of them)
  f32x4 wss[4][2][2], wc1[4][2], wc2[4][2];
  float bss[2][2], bc1[2], bc2[2], g1[2], b1[2], g2[2], b2[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int co = 2 * h + k;
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) {
      wss[kq][k][0] = *reinterpret_cast<const f32x4 *>(w + ss_w + (kq * 16 + co) * 4);      // scale row co
      wss[kq][k][1] = *reinterpret_cast<const f32x4 *>(w + ss_w + (kq * 16 + 4 + co) * 4);  // shift row 4 + co
      wc1[kq][k] = *reinterpret_cast<const f32x4 *>(w + c1_w + (kq * 16 + co) * 4);         // kq = input channel
      wc2[kq][k] = *reinterpret_cast<const f32x4 *>(w + c2_w + (kq * 16 + co) * 4);
    }
    bss[k][0] = w[ss_b + co]; bss[k][1] = w[ss_b + 4 + co];
    bc1[k] = w[c1_b + co]; bc2[k] = w[c2_b + co];
    g1[k] = w[n1_w + co]; b1[k] = w[n1_b + co]; g2[k] = w[n2_w + co]; b2[k] = w[n2_b + co];
  }
  float x[4];
#pragma unroll
  for (int ci = 0; ci < 4; ++ci) x[ci] = X[swz<NC>(ci, n)];
  // ---- scale / shift of my two channels: rows 2h + k (scale, the +1 is in the packed bias) and 4 + 2h + k (shift)
  float sc[2], sh[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    sc[k] = bss[k][0];
    sh[k] = bss[k][1];
  }
#pragma unroll
  for (int kq = 0; kq < 4; ++kq) {
    f32x4 gq = f32x4{0.f, 0.f, 0.f, 0.f};  // G[4 j + kq], j = 0..3 (E = 16: the whole embedding)
#pragma unroll
    for (int j = 0; j < 4; ++j) gq[j] = 4 * j + kq < E ? G[4 * j + kq] : 0.f;
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        sc[k] += wss[kq][k][0][j] * gq[j];
        sh[k] += wss[kq][k][1][j] * gq[j];
      }
  }
  auto conv3 = [&](const float (&in)[4], const f32x4 (&wc)[4][2], const float (&bc)[2], float (&out)[2]) {
    float lft[4], rgt[4];
#pragma unroll
    for (int ci = 0; ci < 4; ++ci) {
      const float a = dpp_mov<0x90>(in[ci]), b = dpp_mov<0xF9>(in[ci]);  // quad_perm [0,0,1,2] / [1,2,3,3]
      lft[ci] = has_l ? a : 0.f;
      rgt[ci] = has_r ? b : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      float acc = bc[k];
#pragma unroll
      for (int ci = 0; ci < 4; ++ci) {
        acc += wc[ci][k].x * lft[ci];
        acc += wc[ci][k].y * in[ci];
        acc += wc[ci][k].z * rgt[ci];
      }
      out[k] = acc;
    }
  };
  auto gn_act = [&](float (&v)[2], const float (&gw)[2], const float (&gb)[2], bool ss) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const float m = group_sum<4>(v[k]) * 0.25f;
      const float d = v[k] - m;
      const float rs = __builtin_amdgcn_rsqf(group_sum<4>(d * d) * 0.25f + 1e-5f);
      float y = d * rs * gw[k] + gb[k];
      if (ss) y = y * sc[k] + sh[k];
      v[k] = silu(y);
    }
  };
  auto gather4 = [&](const float (&mine)[2], float (&all)[4]) {  // channels 0,1 live in half 0, channels 2,3 in half 1
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(mine[k]), __float_as_uint(mine[k]), false, false);
      all[k] = __uint_as_float(r[0]);      // [lo, lo]: half 0's value in both halves
      all[2 + k] = __uint_as_float(r[1]);  // [hi, hi]
    }
  };
  float y[2], ya[4], z[2];
  conv3(x, wc1, bc1, y);
  gn_act(y, g1, b1, !GLDM_SKIP(c, 16));
  gather4(y, ya);
  conv3(ya, wc2, bc2, z);
  gn_act(z, g2, b2, false);
#pragma unroll
  for (int k = 0; k < 2; ++k) X[swz<NC>(2 * h + k, n)] = x[2 * h + k] + z[k];
}


// GroupNorm rides in the conv epilogue: the rows of a group must sit inside one wave's accumulators
__host__ __device__ __forceinline__ bool gn_fusable(int C, int groups) {
  if (groups != 4 || C % 4) return false;
  const int cpg = C / 4;
  return cpg == 1 || cpg == 4 || cpg == 8 || (cpg % 16 == 0 && cpg <= 64);
}

template <int NC>
__host__ __device__ __forceinline__ int build_tape(const gldm_r1d_desc &d, int *tape, bool use_quad = true) {
  using GG = Geo<NC>;
  int n = 0;
  auto emit = [&](int type, int a1 = 0, int a2 = 0, int a3 = 0, int a4 = 0, int a5 = 0, int a6 = 0, int a7 = 0,
                  int a8 = 0, int a9 = 0, int a10 = 0, int a11 = 0) {
    int *o = tape + kOpInts * n++;
    o[0] = type; o[1] = a1; o[2] = a2; o[3] = a3; o[4] = a4; o[5] = a5; o[6] = a6; o[7] = a7;
    o[8] = a8; o[9] = a9; o[10] = a10; o[11] = a11;
  };
  constexpr int X = GG::kBufX, H = GG::kBufH, Y = GG::kBufY, O = GG::kBufO, QKV = GG::kBufQKV;
  int tab_off = 0;  // rows [2C] of every ResnetBlock, in block order (ss_table_kernel writes the same layout)
  auto resblock = [&](const gldm_r1d_resblock &rb, int C, bool last_of_pair) {
    const int toff = tab_off;
    tab_off += 2 * C;
    if (C == 4 && d.seq_len == 4) {  // one-wave VALU form; the barrier comes after the second block
      emit(OP_RES4, rb.c1_w, rb.c1_b, rb.n1_w, rb.n1_b, rb.c2_w, rb.c2_b, rb.n2_w, rb.n2_b, rb.ss_w, rb.ss_b,
           last_of_pair ? 1 : 0);
      return;
    }
    // the position-major engine reads the split-f16 copies of the conv weights (gemm_pm3_bf)
    emit(OP_CONV, NC == 64 ? rb.c1_w3 : rb.c1_w, rb.c1_b, X, H, C, C,
         3 | (1 << 9) | (NC == 64 ? kFlagFused : 0) | ((toff >> 2) << kFlagTabShift), rb.n1_w, rb.n1_b, rb.ss_w, rb.ss_b);
    emit(OP_CONV, NC == 64 ? rb.c2_w3 : rb.c2_w, rb.c2_b, H, X, C, C, 3 | (2 << 9), rb.n2_w, rb.n2_b, -1, 0);  // X += act(GN(conv(H)))
  };
  const bool quad = NC == 64 && use_quad && (quad_levels(d) || quad16_levels(d));
  if (quad) emit(OP_QUAD);
  // constant indices only: a dynamically indexed kernel argument is copied to scratch memory
#pragma unroll
  for (int lv = 0; lv < GLDM_R1D_MAX_LEVELS; ++lv) {
    if (quad && lv < 3) { tab_off += 4 * d.dims[lv]; continue; }   // two ResnetBlocks' table rows each (16-position engine only)
    if (lv < d.n_levels) {
      const int C = d.dims[lv], Cn = d.dims[lv + 1];
      resblock(d.rb[2 * lv], C, false);
      resblock(d.rb[2 * lv + 1], C, true);
      const gldm_r1d_level &v = d.lv[lv];
      // position-major engine (C <= 128 at attention levels): X rows [0, 128) | q,k,v of the four heads: 384 rows, o in
      // place of q.  The PreNorm LayerNorm is folded into the qkv conv (C = 4: computed in the lanes of the qkv phase).
      int Oa = O;
      if (NC == 64) {   // PreNorm + to_qkv + attention core of the four heads: one op, the output as planes (kPlaneH)
        emit(OP_QKVATT, C == 4 ? v.qkvn_w : v.qkvn_w3, v.qkvn_s, X, C);
        Oa = kPlaneH;
      } else {
        emit(OP_LN, X, Y, -1, C, v.ln_g);
        emit(OP_CONV, v.qkv_w[0], -1, Y, QKV, C, 192, 1);
        emit(OP_ATT, QKV, O);
        emit(OP_CONV, v.qkv_w[1], -1, Y, QKV, C, 192, 1);
        emit(OP_ATT, QKV, O + 64 * NC);
      }
      if (NC == 64 && (C == 4 || C == 16 || C == 32 || C == 64 || C == 128)) {  // to_out conv + LayerNorm + residual: one phase
        emit(OP_OUTLN, v.out_w3, v.out_b, Oa, X, kHidden, C, v.ln2_g);
      } else {
        emit(OP_CONV, v.out_w, v.out_b, Oa, Y, kHidden, C, 1);
        emit(OP_LN, Y, -1, X, C, v.ln2_g);
      }
      emit(OP_CONV, (NC == 64 && C >= 16) ? v.down_w3 : v.down_w, v.down_b, X, X, C, Cn, 3 | kFlagAlias);
    }
  }
#pragma unroll
  for (int lv = 1; lv <= GLDM_R1D_MAX_LEVELS; ++lv)
    if (lv == d.n_levels) resblock(d.rb[2 * lv], d.dims[lv], true);
  return n;
}

// One tape entry -> 12 wave-uniform ints.  LDS tape (sample-major engines): three 16-byte reads + v_readfirstlane;
// kernarg tape (position-major engine): scalar loads from the constant address space, no vector instruction at all.
typedef __attribute__((address_space(4))) const int kernarg_int;
typedef __attribute__((ext_vector_type(4))) int i32x4;
template <bool KARG>
__device__ __forceinline__ void read_op(const int *tape, kernarg_int *ktape, int op, int (&o)[kOpInts]) {
  if constexpr (KARG) {
    typedef __attribute__((address_space(4))) const i32x4 kernarg_i4;
    kernarg_i4 *t4 = reinterpret_cast<kernarg_i4 *>(ktape + kOpInts * op);
    const i32x4 q0 = t4[0], q1 = t4[1], q2 = t4[2];
    o[0] = q0.x; o[1] = q0.y; o[2] = q0.z; o[3] = q0.w; o[4] = q1.x; o[5] = q1.y; o[6] = q1.z; o[7] = q1.w;
    o[8] = q2.x; o[9] = q2.y; o[10] = q2.z; o[11] = q2.w;
  } else {
    const int4 *t4 = reinterpret_cast<const int4 *>(tape + kOpInts * op);
    const int4 q0 = t4[0], q1 = t4[1], q2 = t4[2];
    o[0] = q0.x; o[1] = q0.y; o[2] = q0.z; o[3] = q0.w; o[4] = q1.x; o[5] = q1.y; o[6] = q1.z; o[7] = q1.w;
    o[8] = q2.x; o[9] = q2.y; o[10] = q2.z; o[11] = q2.w;
#pragma unroll
    for (int i = 0; i < kOpInts; ++i) o[i] = __builtin_amdgcn_readfirstlane(o[i]);
  }
}

// ---- PreNorm + to_qkv + attention core of the 16-position 64-column engine as ONE op (round 6) ---------------------------
// Until round 6 this was two ops: qkv_ln16_pm wrote q | k | v as a [384][64] f32 block over both plane regions and
// attention16_pm worked from there (three barriers, 96 KiB through ds_write_b32 and back, the plane rows' zero entries
// restored behind it): 25 k of a 236 k-cycle pass at 128 channels.  The wave-local chain of the narrow levels
// (quad16_narrow.h) keeps a (head, sample) pair's q, k and v on the accumulators; the same holds here once the GEMM is dealt
// by PAIRS instead of by rows: wave w = head w >> 1, samples 2 (w & 1) and 2 (w & 1) + 1; an n-tile is one SAMPLE's 16 positions
// (columns 4 p + s of the engine's layout, p = lane & 15: the B fragment reads stride over the plane row), its six m-tiles the
// head's q, k and v rows (to_qkv's own order: m-tiles 2 h + half, + 8, + 16), v with the MFMA operands swapped so that it
// arrives transposed (quad16_attention_head).  LayerNorm folded into the conv as before (W' = W diag(g), s = W' 1: rstd
// (W' x - mean s), the column statistics taken by column_stats8 under the first fragment loads and applied behind ONE barrier);
// q and k leave multiplied by log2(e) for the core's 2^x.  The output goes to the H planes (head h = 32-channel block h) for
// out_ln_pm, as before.  Nothing overwrites a plane region any more.
// Same MFMA count as the row split (144 per wave at 128 channels); a wave now draws six m-tiles' fragments for two n-tiles
// instead of three for four (the head's pair of waves share them in L1).
template <int KB32>   // ceil(C / 32): a 16-channel level is one zero-padded block
__device__ __forceinline__ void qkv_att16_pm(const Ctx &c, int w_off, int s_off, const float *src, int C) {
  using GG = Geo<64>;
  using PGx = PG<16>;
  const float *wp = c.w + w_off, *srow = c.w + s_off;
  const int h = c.wave >> 1, s0 = 2 * (c.wave & 1);
  const int p = c.lane & 15, g = c.lane >> 4;
  const WStream wv(wp, c.lane);
  // m-tile of accumulator i = 2 part + half: rows 128 part + 32 h + 16 half ..
  auto mtile = [&](int i) { return 8 * (i >> 1) + 2 * h + (i & 1); };
  const lds_u4 *pl3 = (const lds_u4 *)(c.lds + PGx::kX) + g * PGx::kCols + PGx::kOff + 4 * p + s0;
  // One set of fragment registers: m-tile i's are refilled with the next block's behind the matrix instructions of m-tile
  // i + 1 (a load into registers the pipe is still reading waits for it; double buffered, 96 registers of fragments beside 48
  // accumulators, the op spilled 46).
  u32x4 a[6][kSplit], bs[2][kSplit];
  f32x4 acc[6][2];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int e = 0; e < 2; ++e) acc[i][e] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto load_a = [&](int i, int kb) {
    const int sb = (mtile(i) * KB32 + kb) * kFragBytes;
#pragma unroll
    for (int pl = 0; pl < kSplit; ++pl) a[i][pl] = wv.raw_at(sb, pl * 1024);
  };
#pragma unroll
  for (int i = 0; i < 6; ++i) load_a(i, 0);
  __builtin_amdgcn_sched_barrier(0);
  {
    const float inv_c = __builtin_amdgcn_rcpf((float)C);  // C: a power of two -> exact
    if (C == 16) column_stats8<2>(c, src, inv_c);
    else column_stats8<4 * KB32>(c, src, inv_c);
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int kb = 0; kb < KB32; ++kb) {
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int pl = 0; pl < kSplit; ++pl) bs[e][pl] = pl3[(kb * kSplit + pl) * PGx::kPlaneU4 + e];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        if (i < 4) acc[i][e] = mfma_split(a[i], bs[e], acc[i][e]);
        else acc[i][e] = mfma_split(bs[e], a[i], acc[i][e]);   // v^T: rows = positions, columns = v rows
      }
      __builtin_amdgcn_sched_barrier(0);
      if (kb + 1 < KB32 && i >= 1) load_a(i - 1, kb + 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (kb + 1 < KB32) load_a(5, kb + 1);
    __builtin_amdgcn_sched_barrier(0);
  }
  f32x4 sv[4];   // s of the q and k rows 4 g + r of the wave's m-tiles; of the v rows: one per lane (row lane & 15)
#pragma unroll
  for (int i = 0; i < 4; ++i) sv[i] = *reinterpret_cast<const f32x4 *>(srow + 16 * mtile(i) + 4 * g);
  const float svv[2] = {srow[16 * mtile(4) + p], srow[16 * mtile(5) + p]};
  __syncthreads();  // every column's (mean, rstd) is in LDS
  const lds_f *mean3 = (const lds_f *)(c.lds + GG::kMiscRed1), *rstd3 = (const lds_f *)(c.lds + GG::kMiscRed2);
  constexpr float kL2e = 1.44269504088896340736f;
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int sm = s0 + e;
    {   // q, k: the lane's column is position p of the sample
      const float rstd = rstd3[4 * p + sm];
      const float mr = mean3[4 * p + sm] * rstd;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][e][r] = (acc[i][e][r] * rstd - mr * sv[i][r]) * kL2e;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {   // v^T: register r is position 4 g + r
      const float rstd = rstd3[4 * (4 * g + r) + sm];
      const float mr = mean3[4 * (4 * g + r) + sm] * rstd;
      acc[4][e][r] = acc[4][e][r] * rstd - mr * svv[0];
      acc[5][e][r] = acc[5][e][r] * rstd - mr * svv[1];
    }
    const f32x4 qa[2] = {acc[0][e], acc[1][e]}, ka[2] = {acc[2][e], acc[3][e]}, vt[2] = {acc[4][e], acc[5][e]};
    f32x4 o[2];
    quad16_attention_head(qa, ka, vt, o);
#pragma unroll
    for (int u = 0; u < 2; ++u)   // o[u][r]: channel 16 u + 4 g + r of head h at position p of sample sm
      store_planes4<16>(c.lds + PGx::kH, kDimHead * h + 16 * u + 4 * g, 4 * p + sm, o[u][0], o[u][1], o[u][2], o[u][3]);
  }
  __syncthreads();
}


template <int NC, int L>
__device__ __forceinline__ void run_tape(const Ctx &c0, const gldm_r1d_desc &d, const int *tape, kernarg_int *ktape, int n_ops, int E, long long *stamps) {
#if defined(GLDM_DEBUG_KNOBS) && defined(GLDM_EXP_LDS_TAPE)   // A/B experiment: the tape from LDS as before
  constexpr bool KARG = false;
#else
  constexpr bool KARG = NC == 64;
#endif
  for (int op = 0; op < n_ops; ++op) {
    if (stamps && c0.tid == 0) stamps[op] = (long long)__builtin_readcyclecounter();
    // the lane ids are laundered per op: otherwise every variant's lane-derived LDS offsets are
    // hoisted out of the step loop as invariants and live (spilled) across the whole kernel
    Ctx c = c0;
    asm volatile("" : "+v"(c.tid), "+v"(c.lane));
    int o[kOpInts];
    read_op<KARG>(tape, ktape, op, o);
    // the short latency-bound phases get issue priority over the co-resident workgroup's long MFMA
    // streams (which need one issue slot per 32 cycles and lose nothing)
    if (o[0] == OP_CONV && o[5] * o[6] >= 128 * 128) __builtin_amdgcn_s_setprio(0);
    else __builtin_amdgcn_s_setprio(3);
    switch (o[0]) {
      case OP_CONV: {
        if constexpr (NC == 64) {
          if (o[7] & kFlagFused) {   // ResnetBlock: this entry is conv1, the next one conv2 (same width, H -> X)
            int q[kOpInts];
            read_op<KARG>(tape, ktape, op + 1, q);
            const GnEpilogue g1{1, o[8], o[9], o[10], o[11], E, o[6], o[6] / 4, c.lds + o[4], (o[7] >> kFlagTabShift) << 2};
            const GnEpilogue g2{2, q[8], q[9], -1, 0, E, q[6], q[6] / 4, c.lds + q[4], 0};
            resblock_pm<L == 16 ? 16 : 4>(c, o[1], o[2], q[1], q[2], c.lds + o[3], c.lds + o[4], o[6], g1, g2, stamps ? stamps + op + 1 : nullptr);
            ++op;
            break;
          }
        }
        const int mode = GLDM_SKIP(c, 1) ? 0 : (o[7] >> 9) & 3;
        const GnEpilogue g{mode, o[8], o[9], GLDM_SKIP(c, 16) ? -1 : o[10], o[11], E, o[6], o[6] / 4, c.lds + o[4],
                           (o[7] >> kFlagTabShift) << 2};
        conv_gemm<NC, L>(c, o[1], o[2], c.lds + o[3], o[5], o[7] & 255, c.lds + o[4], o[6], (o[7] & kFlagAlias) != 0, 0, g);
        break;
      }
      case OP_QUAD:
        if constexpr (NC == 64 && L == 4) {
          if (c.wave < 4)
            quad_narrow_levels(c, (kernarg_desc *)((__attribute__((address_space(4))) const char *)__builtin_amdgcn_kernarg_segment_ptr() +
                                                    offsetof(RunArgs, d)));
#if defined(GLDM_DEBUG_KNOBS) && defined(GLDM_QEXP_DUP)
          // timing experiment (wrong results): GLDM_QEXP_DUP=2 -> waves 4-7 run the chain of quads 0-3 a second time beside
          // them (two compute waves per SIMD); =1 -> they do nothing (one compute wave per SIMD, no loader)
          else if (GLDM_QEXP_DUP == 2)
            quad_narrow_levels(c, (kernarg_desc *)((__attribute__((address_space(4))) const char *)__builtin_amdgcn_kernarg_segment_ptr() +
                                                    offsetof(RunArgs, d)));
#else
          else quad_loader(c);
#endif
          __syncthreads();
        } else if constexpr (NC == 64 && L == 16) {
          kernarg_desc *dkp = (kernarg_desc *)((__attribute__((address_space(4))) const char *)__builtin_amdgcn_kernarg_segment_ptr() +
                                               offsetof(RunArgs, d));
          const float *sstab;   // the wave's sample's scale / shift rows of the six narrow ResnetBlocks
          if (c.ss_lane) {   // pose decoder: the per-cloud table (wave uniform: the launch has one or not); Ctx::ss_lane of a
                             // lane whose sample (lane & 3) is the wave's
            const unsigned long long pv = (unsigned long long)c.ss_lane;
            const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)pv, c.wave & 3);
            const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(pv >> 32), c.wave & 3);
            sstab = (const float *)(((unsigned long long)hi << 32) | lo);
          } else {   // time-conditioned net: the step's rows by all eight waves first, into LDS behind the ring
            quad16_ss_rows(c, dkp);
            __syncthreads();
            sstab = c.lds + kQ16SsLds + (c.wave & 3) * kQ16SsRows;
          }
          if (c.wave < 4) quad16_narrow_levels<1>(c, dkp, sstab);
          else quad_loader<QStream16<1>>(c);
          __syncthreads();
          zero_plane_pads16(c.lds, c.tid);   // the ring's slots lie over the H plane rows (conv2 of the next ResnetBlock reads them behind a barrier)
        }
        break;
      case OP_RES4:
        if (!GLDM_SKIP(c, 8)) {
          if constexpr (NC == 64 && L == 4) resblock4_pm(c, o, E);
          else if constexpr (NC != 64) resblock4_valu<NC>(c, o, E);
        }
        if (o[11]) __syncthreads();
        break;
      case OP_QKVATT:
        if constexpr (NC == 64) {
          if (o[4] == 128) {   // the 128-channel level: this op and the to_out op behind it as one case (see out_ln_request)
            int q[kOpInts];
            read_op<KARG>(tape, ktape, op + 1, q);
            if (q[0] == OP_OUTLN && q[6] == 128) {
              const Frag3 first = out_ln_request(c, q[1]);
              if constexpr (L == 4) qkv_att_pm<4>(c, o[1], o[2], c.lds + o[3], o[4]);
              else qkv_att16_pm<4>(c, o[1], o[2], c.lds + o[3], o[4]);
              ++op;
              if (stamps && c0.tid == 0) stamps[op] = (long long)__builtin_readcyclecounter();
              out_ln_pm128<L == 16 ? 16 : 4>(c, q[1], q[2], c.lds + q[3], q[5], c.lds + q[4], q[7], first);
              break;
            }
          }
        }
        if constexpr (NC == 64 && L == 4) {
          if (o[4] == 128) qkv_att_pm<4>(c, o[1], o[2], c.lds + o[3], o[4]);
          else if (o[4] == 64) qkv_att_pm<2>(c, o[1], o[2], c.lds + o[3], o[4]);
          else if (o[4] == 32) qkv_att_pm<1>(c, o[1], o[2], c.lds + o[3], o[4]);
          else qkv_att_pm<0>(c, o[1], o[2], c.lds + o[3], o[4]);
        } else if constexpr (NC == 64 && L == 16) {
          if (o[4] == 128) qkv_att16_pm<4>(c, o[1], o[2], c.lds + o[3], o[4]);
          else if (o[4] == 64) qkv_att16_pm<2>(c, o[1], o[2], c.lds + o[3], o[4]);
          else qkv_att16_pm<1>(c, o[1], o[2], c.lds + o[3], o[4]);
        }
        break;
      case OP_OUTLN:
        if constexpr (NC == 64) out_ln_pm<L == 16 ? 16 : 4>(c, o[1], o[2], c.lds + o[3], o[5], c.lds + o[4], o[6], o[7]);
        break;
      case OP_LN:
        layer_norm_pass<NC>(c, c.lds + o[1], o[2] >= 0 ? c.lds + o[2] : nullptr, o[3] >= 0 ? c.lds + o[3] : nullptr,
                            o[4], o[5]);
        break;
      default:
        if constexpr (NC != 64) attention_pair<NC, L>(c, c.lds + o[1], c.lds + o[2]);
        break;
    }
  }
}

// Philox4x32-10 (Salmon et al., SC'11: the counter-based generator of curand / torch's device RNG) and Box-Muller: four
// unit normals per (key, counter).  z[0..3] of counter (latent, position block, step) are positions 4 block + 0..3.
__device__ __forceinline__ void philox_normal4(unsigned long long seed, unsigned c0, unsigned c1, unsigned c2, unsigned c3, float (&z)[4]) {
  unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
    c1 = (unsigned)p1; c3 = (unsigned)p0; c0 = n0; c2 = n2;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  const unsigned u[4] = {c0, c1, c2, c3};
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const float a = ((float)u[2 * h] + 1.0f) * 2.3283064365386963e-10f;   // (0, 1]
    const float ang = (float)u[2 * h + 1] * (6.283185307179586f * 2.3283064365386963e-10f);
    const float rad = sqrtf(-2.0f * __logf(a));
    z[2 * h] = rad * __cosf(ang);
    z[2 * h + 1] = rad * __sinf(ang);
  }
}

#pragma clang fp contract(off)
// DPM-Solver++(2M) step of ElucidatedDiffusion.sample_using_dpmpp (elucidated_diffusion.py:259-313) on one element:
// denoised = c_skip x + c_out net (:134, optional clamp :137); denoised_d = (1 - gamma) denoised + gamma old (:303);
// x' = (sigma_next / sigma) x - expm1(-h) denoised_d (:305).  cf = [c_in, c_skip, c_out, 1 - gamma, gamma,
// sigma_fn(t_next) / sigma_fn(t), expm1(-h), use_old].  Every operation rounds once, in the reference's order.
__device__ __forceinline__ float dpmpp_update(int clamp, const float *cf, float x, float net, float *old) {
  const float a = cf[1] * x;
  const float b = cf[2] * net;
  float den = a + b;
  if (clamp) den = fminf(fmaxf(den, -1.0f), 1.0f);
  float d = den;
  if (cf[7] != 0.f) {
    const float p = cf[3] * den;
    const float q = cf[4] * *old;
    d = p + q;
  }
  *old = den;
  const float u = cf[5] * x;
  const float v = cf[6] * d;
  return u - v;
}

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

template <int NC, int L>
__global__ __launch_bounds__(Geo<NC>::kThreads, 2) void r1d_kernel(const RunArgs a) {
  using GG = Geo<NC>;
  extern __shared__ float lds[];
  const gldm_r1d_desc &d = a.d;
  constexpr int S = NC / L;
  // column <-> (sample, position): sample-major tiles (column = sample * L + position) or, for the 64-column
  // engine of the L = 4 denoiser, position-major ones (column = 16 * position + sample)
  // (4 positions: column = 16 * position + sample, 16 samples) or of the 16-position nets (column = 4 * position + sample,
  // 4 samples)
  constexpr bool PM = NC == 64;
  static_assert(!PM || L == 4 || L == 16, "64-column engines: 4- and 16-position nets");
  auto samp_of = [](int n) { return PM ? (L == 4 ? (n & 15) : (n & 3)) : n / L; };
  auto pos_of = [](int n) { return PM ? (L == 4 ? (n >> 4) : (n >> 2)) : n % L; };
  auto col_of = [](int sm, int l) { return PM ? (L == 4 ? 16 * l + sm : 4 * l + sm) : sm * L + l; };
  Ctx c{a.weights, lds, (int)threadIdx.x, __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), (int)threadIdx.x & 63,
        a.skip, GG::kNT};
  if constexpr (PM && L == 16) c.park = a.park + (size_t)blockIdx.x * (kParkBytes / 4);
  const int E = d.emb_dim, R = d.cond_rows;
  float *lat = lds + GG::kMiscLat, *epsr = lds + GG::kMiscEps, *G = lds + GG::kMiscG;
  float *X = lds + GG::kBufX;
  const bool has_in = d.latent_dim > 0, has_head = d.n_head > 0;
  int CF = d.dims[1];  // width of the last level (constant indices: see build_tape)
#pragma unroll
  for (int lv = 2; lv <= GLDM_R1D_MAX_LEVELS; ++lv) CF = lv == d.n_levels ? d.dims[lv] : CF;

  for (int i = c.tid; i < GG::kLdsFloats; i += GG::kThreads) lds[i] = 0.f;  // dead columns must stay finite
  __syncthreads();
  // the wave-local narrow levels are in use when the step program (built by the launcher) starts with their op
  const bool quad16 = PM && L == 16 && a.pm_nops > 0 && a.pm_tape[0] == OP_QUAD;
  if constexpr (PM && L == 4) {
    static_assert(2 * kQNEnd <= GG::kLdsFloats - GG::kMiscQTab, "kMiscQTab size");
    if (quad_levels(d)) quad_build_table(d, reinterpret_cast<int *>(lds + GG::kMiscQTab), c.tid, GG::kThreads);
  }
  if constexpr (PM && L == 16) {
    if (quad16) quad16_build_table<1>(d, reinterpret_cast<int *>(lds + GG::kMiscQTab), c.tid, GG::kThreads);
  }
  int *tape = reinterpret_cast<int *>(lds + GG::kMiscTape);
  ChainHdr *hdr = reinterpret_cast<ChainHdr *>(a.ws);
  unsigned long long *state = reinterpret_cast<unsigned long long *>(a.ws) + kChainHdrBytes / 8;
  const bool chained = a.left_tiles > 0 && a.chain > 1;
  if (c.tid == 0) {
#if defined(GLDM_DEBUG_KNOBS) && defined(GLDM_EXP_LDS_TAPE)
    tape[1023] = build_tape<NC>(d, tape);
#else
    tape[1023] = PM ? a.pm_nops : build_tape<NC>(d, tape);
#endif
    // Slot = order of arrival, not blockIdx: a slot only ever waits for the slot before it, which has
    // then already started, so the hand-offs cannot deadlock whatever the dispatch order or residency.
    tape[1022] = chained ? (int)__hip_atomic_fetch_add(&hdr->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                         : (int)blockIdx.x;
    tape[1021] = chained ? (int)hdr->epoch : 0;  // written by the previous launch on this workspace
  }
  __syncthreads();
  const int n_ops = __builtin_amdgcn_readfirstlane(tape[1023]);
  // the kernel's one argument (RunArgs, by value) sits at offset 0 of the kernarg segment
  kernarg_int *ktape = (kernarg_int *)((__attribute__((address_space(4))) const char *)__builtin_amdgcn_kernarg_segment_ptr() +
                                       offsetof(RunArgs, pm_tape));
  const int slot = __builtin_amdgcn_readfirstlane(tape[1022]);
  const unsigned epoch = (unsigned)__builtin_amdgcn_readfirstlane(tape[1021]);
#ifdef GLDM_DEBUG_KNOBS
  if (a.stagger_ticks > 0 && ((blockIdx.x / a.n_cus) & 1)) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < a.stagger_ticks) __builtin_amdgcn_s_sleep(32);
  }
#endif
  // ---- this slot's program: its own tiles for all steps, with (at most) one step segment of a left-over tile
  // spliced in at the step at which the previous slot of the chain delivers it.  Built once, by one thread,
  // into LDS (like the op tape): the step loop keeps no scheduling state in registers.
  int *segs = reinterpret_cast<int *>(lds + GG::kMiscSegs);
  if (c.tid == 0) {
    const int N = a.n_steps;
    const int cj = slot / a.chain, ci = slot - cj * a.chain;
    const bool has_left = a.left_tiles > 0 && cj < a.left_tiles && ci * a.seglen < N;
    const int cut = has_left ? ci * a.seglen : 0;
    int n = 0;
    auto put = [&](int tile, int s0, int s1) {
      segs[4 * n] = tile; segs[4 * n + 1] = s0; segs[4 * n + 2] = s1;
      ++n;
    };
    if (cut > 0) put(slot, 0, cut);  // own tile 0 runs in two parts around the spliced segment
    if (has_left) put(a.rounds * a.slots + cj, cut, min(N, cut + a.seglen));
    put(slot, cut, N);
    for (int r = 1; r < a.rounds; ++r) put(r * a.slots + slot, 0, N);
    segs[4 * kMaxSegs - 1] = n;
  }
  __syncthreads();
  const int nseg = __builtin_amdgcn_readfirstlane(segs[4 * kMaxSegs - 1]);
  for (int sg = 0; sg < nseg; ++sg) {
  const int tile = __builtin_amdgcn_readfirstlane(segs[4 * sg]);
  const int s0 = __builtin_amdgcn_readfirstlane(segs[4 * sg + 1]), s1 = __builtin_amdgcn_readfirstlane(segs[4 * sg + 2]);
  const int N = a.n_steps;
  const int samp0 = tile * S;
  const int nsamp = min(S, a.n_samples - samp0);  // samples this tile holds (the batch's last tile may be short)
  c.nta = (nsamp * L <= 16) ? 1 : GG::kNT;
  if (L == 16 && a.ss_tab) {
    if constexpr (PM) {   // this lane's sample (lane & 3) -> its cloud's rows
      const int gi = min(samp0 + min(c.lane & 3, nsamp - 1), a.n_samples - 1);
      c.ss_lane = a.ss_tab + (size_t)(gi / a.samples_per_cond) * a.ss_stride;
    } else {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int gi = min(samp0 + min(q, nsamp - 1), a.n_samples - 1);
        c.ss_row[q] = a.ss_tab + (size_t)(gi / a.samples_per_cond) * a.ss_stride;
      }
    }
  }
  // ---- latent row for this tile: from the input (first step) or from the slot that ran the steps before s0
  int tid_s = c.tid;   // opaque (see tid_o below): the segment prologue's pointers are rebuilt per segment, not kept from kernel start
  asm volatile("" : "+v"(tid_s));
  if (tid_s < NC) {
    const int s = samp_of(tid_s), l = pos_of(tid_s);
    const int gi = min(samp0 + min(s, nsamp - 1), a.n_samples - 1);
    float v;
    if (s0 > 0) {
      v = chain_take(state + (size_t)tile * NC + tid_s, chain_tag(epoch, s0), &hdr->error);
    } else if (has_in) {
      const float *wi = a.weights + d.in_w + l * d.latent_dim;
      v = a.weights[d.in_b + l];
      for (int q = 0; q < d.latent_dim; ++q) v += wi[q] * a.x_in[(size_t)gi * d.latent_dim + q];
    } else {
      v = a.x_in[(size_t)gi * L + l];
    }
    lat[tid_s] = v;
  }
  __syncthreads();

  // The conditioning part of the embedding sum is the same at every step: thread i < S E keeps its (sample, e) entry's
  // cond rows in registers for the whole segment (validate(): S E <= 320 <= threads of the 64-column engine; the
  // 32-column engines take the loop below) and requests the NEXT step's time-embedding value a step ahead, so that the
  // step prologue is three SiLUs and an LDS store instead of a chain of L2 round trips.
  constexpr int kMaxR = 4;
  // 64-column engines (the step program travels in the kernel arguments: the LDS tape region holds three scalars at its
  // end): the cond rows wait there, [r][thread], instead of in registers -- four values alive across the whole op tape were
  // spilled to scratch and reloaded every step.  (More rows than fit there: the general loop below.)
#if defined(GLDM_DEBUG_KNOBS) && defined(GLDM_EXP_LDS_TAPE)
  constexpr bool kCeLds = false;   // (that experiment reads the step program from the region)
#else
  constexpr bool kCeLds = PM;
#endif
  const bool g_fast = S * E <= GG::kThreads && R <= kMaxR && !a.sample_t && (!kCeLds || R * S * E <= 1020);
  lds_f *ce_l = (lds_f *)(lds + GG::kMiscTape);
  float ce_reg[kCeLds ? 1 : kMaxR] = {}, se_reg = 0.f, te_next = 0.f;
  if (g_fast && tid_s < S * E) {
    const int s = tid_s / E, e = tid_s - s * E;
    const int gi = min(samp0 + min(s, nsamp - 1), a.n_samples - 1);
    const float *ce = a.cemb + ((size_t)(gi / a.samples_per_cond) * R) * E + e;
    if constexpr (kCeLds) {
#pragma unroll
      for (int r = 0; r < kMaxR; ++r)
        if (r < R) ce_l[r * S * E + tid_s] = ce[r * E];
    } else {
#pragma unroll
      for (int r = 0; r < kMaxR; ++r) ce_reg[r] = ce[(r < R ? r : 0) * E];
    }
    if (a.semb) se_reg = a.semb[(size_t)gi * E + e];  // latent_emb += cls_emb (class_conditioned_resnet.py:99-101)
    if (a.temb) te_next = a.temb[(size_t)a.timesteps[s0] * E + e];
  }
  for (int step = s0; step < s1; ++step) {
    if (GLDM_STAMPS(a.stamps) && blockIdx.x == 0 && c.tid == 0) a.stamps[kMaxOps + 1] = (long long)__builtin_readcyclecounter();
    // The thread index as this step's short phases see it: opaque, so that the index maps and LDS addresses derived from
    // it are recomputed here (a handful of integer instructions) instead of being hoisted out of the step loop, kept
    // across the whole op tape and spilled (14 scratch reloads per step, each a round trip in front of its use).
    int tid_o = c.tid;
    asm volatile("" : "+v"(tid_o));
    // ---- G[s][e] = sum_r silu(temb[t][e] + cemb[cond][r][e])
    if (g_fast) {
      if (tid_o < S * E) {
        const float te = te_next + se_reg;
        if (a.temb) te_next = a.temb[(size_t)a.timesteps[step + 1 < s1 ? step + 1 : step] * E + (tid_o % E)];
        float g = 0.f;
        if constexpr (kCeLds) {   // (a thread reads back what it wrote itself: no barrier)
#pragma unroll
          for (int r = 0; r < kMaxR; ++r) g += r < R ? silu(te + ce_l[(r < R ? r : 0) * S * E + tid_o]) : 0.f;
        } else {
#pragma unroll
          for (int r = 0; r < kMaxR; ++r) g += r < R ? silu(te + ce_reg[r]) : 0.f;
        }
        G[tid_o] = g;
      }
    } else if (!GLDM_SKIP(c, 32))
    for (int i = tid_o; i < S * E; i += GG::kThreads) {
      const int s = i / E, e = i - s * E;
      const int gi = min(samp0 + min(s, nsamp - 1), a.n_samples - 1);
      const float *ce = a.cemb + ((size_t)(gi / a.samples_per_cond) * R) * E + e;
      float te = 0.f;
      if (a.temb) {
        const int t = a.sample_t ? a.sample_t[gi] : a.timesteps[step];
        te = a.temb[(size_t)t * E + e];
      }
      if (a.semb) te += a.semb[(size_t)gi * E + e];  // latent_emb += cls_emb (class_conditioned_resnet.py:99-101)
      float g = 0.f;
      for (int r = 0; r < R; ++r) g += silu(te + ce[r * E]);
      G[s * E + e] = g;
    }
    if constexpr (PM) {   // quad engine hand-shake words (quad_narrow.h): every step starts from zero
      if ((L == 4 || quad16) && tid_o < 16 + 4 * 64) reinterpret_cast<int *>(lds + GG::kMiscQ)[tid_o] = 0;
    }
    // ---- init conv (k = 7, one input channel); the barrier below also publishes G
    const int C0 = d.dims[0];
    // DPM++ feeds the network c_in(sigma) * x (elucidated_diffusion.py:127-128)
    const float in_scale = a.sched_kind == GLDM_SCHED_DPMPP ? a.sched_coef[(size_t)step * GLDM_SCHED_COEF_STRIDE] : 1.0f;
    const bool scale_in = a.sched_kind == GLDM_SCHED_DPMPP;
    if (!GLDM_SKIP(c, 64))
    for (int i = tid_o; i < C0 * NC; i += GG::kThreads) {
      const int ch = i / NC, n = i - ch * NC;
      const int l = pos_of(n), sm = samp_of(n);
      float acc = a.weights[d.init_b + ch];
      const float *wk = a.weights + d.init_w + ch * 7;
      // the seven taps are requested together, in range or not: behind the range test each was a round trip of its own
      float wq[7];
#pragma unroll
      for (int q = 0; q < 7; ++q) wq[q] = wk[q];
#pragma unroll
      for (int q = 0; q < 7; ++q) {
        const int p = l + q - 3;
        const bool in = p >= 0 && p < L;
        float xv = lat[col_of(sm, in ? p : l)];
        if (scale_in) xv = in_scale * xv;
        acc = in ? acc + wq[q] * xv : acc;
      }
      X[PM ? pswz(ch, n) : swz<NC>(ch, n)] = acc;
    }
    __syncthreads();
    if constexpr (PM && L == 16) if (!quad16) {   // the first ResnetBlock reads the planes of the init conv's 16 rows (the
                                                  // wave-local chain reads the f32 rows and restores the zero entries itself)
      zero_plane_pads16(lds, c.tid);   // the previous step's (or tile's) 256-channel output rows lie over some of them
      {   // ... and over channels 16 .. 31 of H-plane block 0, which the 16-channel level multiplies with zero weights
        lds_u4 *hb = (lds_u4 *)(lds + PG<16>::kH);
        const u32x4 z4 = u32x4{0u, 0u, 0u, 0u};
        if (c.tid < 3 * 2 * 64) {
          const int plane = c.tid / 128, gg = 2 + ((c.tid >> 6) & 1), col = c.tid & 63;
          hb[(plane * 4 + gg) * PG<16>::kCols + PG<16>::kOff + col] = z4;
        }
      }
      if (c.tid < 256) {
        const int rg = c.tid >> 6, n = c.tid & 63;
        const lds_f *x3 = (const lds_f *)X;
        store_planes4<16>(lds + PG<16>::kX, 4 * rg, n, x3[pswz(4 * rg, n)], x3[pswz(4 * rg + 1, n)], x3[pswz(4 * rg + 2, n)],
                          x3[pswz(4 * rg + 3, n)]);
      }
      __syncthreads();
    }

    run_tape<NC, L>(c, d, tape, ktape, n_ops, E, blockIdx.x == 0 ? GLDM_STAMPS(a.stamps) : nullptr);
    if (GLDM_STAMPS(a.stamps) && blockIdx.x == 0 && c.tid == 0) a.stamps[n_ops] = (long long)__builtin_readcyclecounter();

    // ---- final 1x1 conv to one channel: eps[n] = b + sum_c w[c] X[c][n]
    if (!GLDM_SKIP(c, 128)) {
      // this step's scheduler coefficients, requested now (two 16-byte loads, in flight under the reduction): read one
      // by one where the update uses them they were six dependent L2 round trips per step
      f32x4 cf_lo = f32x4{0.f, 1.f, 0.f, 0.f}, cf_hi = f32x4{0.f, 0.f, 0.f, 0.f};
      if (a.sched_kind != GLDM_SCHED_NONE) {
        const f32x4 *cfp = reinterpret_cast<const f32x4 *>(a.sched_coef + (size_t)step * GLDM_SCHED_COEF_STRIDE);
        cf_lo = cfp[0];
        cf_hi = cfp[1];
      }
      const float final_b = a.weights[d.final_b];
      float *red1 = lds + GG::kMiscRed1;
      int tid_f = c.tid;
      asm volatile("" : "+v"(tid_f));
      const int lane_f = tid_f & 63;
      const int n = lane_f & (NC - 1), slot = c.wave * GG::kRP + lane_f / NC;
      float part = 0.f;
      for (int row = slot; row < CF; row += GG::kSlots)
        part += a.weights[d.final_w + row] * X[PM ? pswz(row, n) : swz<NC>(row, n)];
      if (GG::kRP == 2) part = half_sum(part);
      red1[c.wave * NC + n] = part;
      __syncthreads();
      if (tid_f < NC) {
        float e = final_b;
#pragma unroll
        for (int q = 0; q < GG::kWaves; ++q) e += red1[q * NC + tid_f];
        epsr[tid_f] = e;
        if (a.sched_kind != GLDM_SCHED_NONE) {
          const int s = samp_of(tid_f), l = pos_of(tid_f);
          const int gi = min(samp0 + min(s, nsamp - 1), a.n_samples - 1);
          const float cf[8] = {cf_lo[0], cf_lo[1], cf_lo[2], cf_lo[3], cf_hi[0], cf_hi[1], cf_hi[2], cf_hi[3]};
          float nz = 0.f;
          if (a.sched_kind == GLDM_SCHED_DDPM && cf[7] != 0.f) {
            if (a.step_noise) {
              nz = a.step_noise[((size_t)step * a.n_samples + gi) * L + l];
            } else if (a.noise_on) {
              const unsigned long long g = (unsigned long long)(a.noise_base + gi);
              float z4[4];
              philox_normal4(a.noise_seed, (unsigned)g, (unsigned)(g >> 32), (unsigned)(l >> 2), (unsigned)step, z4);
              nz = (l & 3) == 0 ? z4[0] : ((l & 3) == 1 ? z4[1] : ((l & 3) == 2 ? z4[2] : z4[3]));
            }
          }
          if (a.sched_kind == GLDM_SCHED_DPMPP) lat[tid_f] = dpmpp_update(a.clip_sample, cf, lat[tid_f], e, lds + GG::kMiscOld + tid_f);
          else lat[tid_f] = scheduler_update(a.sched_kind, a.clip_sample, cf, lat[tid_f], e, nz);
        }
      }
      __syncthreads();
    }
  }

  // ---- outputs of the segment: the result after the last step, else the latent for the next slot of the chain
  int tid_e = c.tid;   // opaque like tid_o: nothing derived from it lives across the steps
  asm volatile("" : "+v"(tid_e));
  if (s1 < N) {
    if (tid_e < NC) chain_give(state + (size_t)tile * NC + tid_e, lat[tid_e], chain_tag(epoch, s1));
  } else if (!has_head) {
    if (tid_e < NC) {
      const int s = samp_of(tid_e), l = pos_of(tid_e);
      const int gi = samp0 + s;
      if (s < nsamp && gi < a.n_samples) a.out0[(size_t)gi * L + l] = a.sched_kind == GLDM_SCHED_NONE ? epsr[tid_e] : lat[tid_e];
    }
  } else {
    // heads: rows 0..5 tmrp, row 6 class logit; input = the L-vector of each sample
    int nh = d.n_head;
    asm volatile("" : "+s"(nh));   // opaque here: the reciprocal of the division below was computed at kernel entry and spilled
    if (tid_e < S * nh) {
      const int s = tid_e / nh, r = tid_e - s * nh;
      const int gi = samp0 + s;
      if (s < nsamp && gi < a.n_samples) {
        const float *wr = a.weights + d.head_w + r * L;
        float acc = a.weights[d.head_b + r];
        for (int l = 0; l < L; ++l) acc += wr[l] * epsr[col_of(s, l)];
        if (r < 6) a.out0[(size_t)gi * 6 + r] = acc;
        else a.out1[gi] = acc;
      }
    }
  }
  __syncthreads();  // lat / epsr are rewritten by the next segment
  }  // segments
  if (chained && c.tid == 0) {  // the last workgroup to finish re-arms the header for the next launch
    const unsigned done = __hip_atomic_fetch_add(&hdr->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (done == gridDim.x - 1) {
      __hip_atomic_store(&hdr->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&hdr->done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&hdr->epoch, (epoch + 1u) & 0xFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
// One workgroup = one 64-column tile = 64/U centres x U neighbours (NC = 64 geometry).  The
// grouped tensor ([B, 3+C, M, U], 4.3 MB per cloud at SSG-SA2) never exists in HBM: the
// neighbour tile is gathered straight into LDS, the MLP layers (BatchNorm folded, ReLU) run on
// the same f32-MFMA GEMM core as the ResNet1D engine with weights streamed from L2, and the max
// over the U neighbours is taken on chip.  HBM traffic: 12N + 4CN + 12M + 4MU (idx) in,
// 4 Cout M out per cloud.
struct SaArgs {
  const float *points, *centers, *feat;
  const int32_t *idx;
  const float *weights;
  float *out;
  int c, n, m, u, n_layers;
  int cin_pad[4], cout[4], w_off[4], b_off[4];
  // split-f16 kernel: range scales on (range_pow2).  gain_r / gain_b: per layer, the largest row sum of |W| and the largest
  // |bias| (BatchNorm folded), from the packer: |layer output| <= gain_r * max |input| + gain_b
  int ranged;
  float gain_r[4], gain_b[4];
  // split-f16 kernel, first layer hoisted (gldm_sa_mlp_forward_f16x2_pre): pre [b][n][c1] = W1b f + b1 per POINT (one
  // pointwise GEMM per cloud instead of one per (centre, neighbour) pair: every point sits in ~16 balls), wa_off: float
  // index in `weights` of W1a [c1][4] (the coordinate columns x, y, z, 0).  The MFMA layers are then layers 2.. of the module.
  const float *pre;
  int c1, wa_off;
  int pre_bcast;   // pre is ONE row [c1] for every point (a module without features: the row is the folded bias b1)
};

__global__ __launch_bounds__(Geo<64>::kThreads, 2) void sa_mlp_kernel(const SaArgs a) {
  using GG = Geo<64>;
  constexpr int NC = 64;
  extern __shared__ float lds[];
  Ctx c{a.weights, lds, (int)threadIdx.x, __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), (int)threadIdx.x & 63,
        0, GG::kNT};
  const int b = blockIdx.y, tile = blockIdx.x;
  const int cpt = NC / a.u;                 // centres per tile
  const int j0 = tile * cpt;
  const float *pts = a.points + (size_t)b * 3 * a.n;
  const float *ctr = a.centers + (size_t)b * 3 * a.m;
  const float *feat = a.feat ? a.feat + (size_t)b * a.c * a.n : nullptr;
  const int32_t *idx = a.idx + ((size_t)b * a.m + j0) * a.u;
  float *X = lds + GG::kBufX, *H = lds + GG::kBufH;
  // ---- gather the neighbour tile: rows 0..2 relative coords, 3..3+C features, zero pad
  {
    const int col = c.lane, jj = col / a.u;
    const bool live = j0 + jj < a.m;
    const int id = live ? idx[col] : 0;
    const int rows = a.cin_pad[0];
    // six rows per wave in flight at a time: every element is a scattered memory round trip, and issued one by one
    // (load, wait, store) the gather took as long as the tile's MFMAs.  (Requesting the NEXT tile's rows before the
    // MLP from a persistent workgroup was slower: the in-order vmcnt makes the first weight fragment wait for them.)
    for (int r0 = c.wave; r0 < rows; r0 += 6 * GG::kWaves) {
      float v[6];
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        const int r = r0 + q * GG::kWaves;
        v[q] = 0.f;
        if (live && r < rows) {
          if (r < 3) v[q] = pts[r * a.n + id] - ctr[r * a.m + j0 + jj];
          else if (r < 3 + a.c) v[q] = feat[(size_t)(r - 3) * a.n + id];
        }
      }
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        const int r = r0 + q * GG::kWaves;
        if (r < rows) X[swz<NC>(r, col)] = v[q];
      }
    }
  }
  __syncthreads();
  // ---- grouped MLP (1x1 convs + folded BN + ReLU), ping-pong X <-> H
  float *src = X, *dst = H;
  for (int l = 0; l < a.n_layers; ++l) {
    conv_gemm<NC, 4>(c, a.w_off[l], a.b_off[l], src, a.cin_pad[l], 1, dst, a.cout[l], false, 1);   // 1x1 layers only (L is the k = 3 convs' layout)
    float *t = src; src = dst; dst = t;
  }
  // ---- max over the U neighbours of each centre
  const int cout = a.cout[a.n_layers - 1];
  float *out = a.out + (size_t)b * cout * a.m;
  for (int i = c.tid; i < cout * cpt; i += GG::kThreads) {
    const int row = i / cpt, jj = i - row * cpt;
    if (j0 + jj >= a.m) continue;
    float mx = -3.0e38f;
    for (int k = 0; k < a.u; ++k) mx = fmaxf(mx, src[swz<NC>(row, jj * a.u + k)]);
    out[(size_t)row * a.m + j0 + jj] = mx;
  }
}

bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

constexpr int kEngineNC = 32;  // denoiser / decoder geometry: 32 columns, two workgroups per CU
int engine_nc() { return kEngineNC; }

int validate(const gldm_r1d_desc *d) {
  if (!d) return GLDM_ERR_INVALID_ARG;
  if (d->seq_len != 4 && d->seq_len != 16) return GLDM_ERR_UNSUPPORTED;
  if (d->n_levels < 1 || d->n_levels > GLDM_R1D_MAX_LEVELS || 12 * d->n_levels + 2 > kMaxOps) return GLDM_ERR_UNSUPPORTED;
  const int nc = engine_nc(), waves = 4;
  const int S = nc / d->seq_len;
  if (d->emb_dim <= 0 || (d->emb_dim & 15) || S * d->emb_dim > 320) return GLDM_ERR_UNSUPPORTED;
  if (d->groups != 4 || waves != 4) return GLDM_ERR_UNSUPPORTED;  // GroupNorm lives in the conv epilogue
  for (int i = 0; i <= d->n_levels; ++i) {
    const int C = d->dims[i];
    if (C > kMaxC || C < 4 || !pow2(C) || !gn_fusable(C, d->groups)) return GLDM_ERR_UNSUPPORTED;
    if (i < d->n_levels && C > 128) return GLDM_ERR_UNSUPPORTED;  // attention levels keep 4 regions in LDS
  }
  return GLDM_OK;
}

// Rows of the per-cloud scale/shift table (ss_table_kernel) when the engine uses one: the 16-position pose decoder
// (no time embedding, wide embedding: E >= 32).  0 = no table.
int ss_table_rows(const gldm_r1d_desc *d) {
  if (d->seq_len != 16 || d->latent_dim <= 0 || d->emb_dim < 32 || d->emb_dim > 256 || d->dims[0] < 16) return 0;
  int rows = 0;
  for (int i = 0; i < 2 * d->n_levels + 1; ++i) rows += 2 * d->dims[i < 2 * d->n_levels ? i / 2 : d->n_levels];
  return rows;
}

// The position-major 64-column engine (r1d_kernel<64, 4>) serves the latent denoiser of the shipped
// configurations: 4 positions, no input layer / heads, emb_dim 16, a 4-channel first level and 32..256-channel
// ones after it.  Anything else runs on the sample-major 32-column engine.
bool pm_supported(const gldm_r1d_desc *d) {
  if (d->seq_len != 4 || d->latent_dim != 0 || d->n_head != 0 || d->emb_dim != 16 || d->groups != 4) return false;
  if (d->dims[0] != 4) return false;
  for (int i = 0; i < d->n_levels; ++i)  // to_qkv with the PreNorm gain folded in (ABI 4 packers provide it)
    if (d->lv[i].qkvn_w <= 0 || d->lv[i].qkvn_s <= 0) return false;
  for (int i = 1; i < d->n_levels; ++i)  // split-f16 conv weights (ABI 5 packers provide them)
    if (d->lv[i].down_w3 <= 0 || d->lv[i].qkvn_w3 <= 0 || d->lv[i].out_w3 <= 0 || d->rb[2 * i].c1_w3 <= 0 || d->rb[2 * i].c2_w3 <= 0 || d->rb[2 * i + 1].c1_w3 <= 0 ||
        d->rb[2 * i + 1].c2_w3 <= 0)
      return false;
  if (d->rb[2 * d->n_levels].c1_w3 <= 0 || d->rb[2 * d->n_levels].c2_w3 <= 0 || d->lv[0].out_w3 <= 0) return false;
  for (int i = 1; i <= d->n_levels; ++i) {
    const int C = d->dims[i];
    if (!(C == 32 || C == 64 || C == 128 || C == 256)) return false;
    if (i < d->n_levels && C > 128) return false;
  }
  return true;
}

// The 16-position 64-column engine (r1d_kernel<64, 16>: tiles of 4 samples x 16 positions, column = 4 * position +
// sample, split-f16 GEMMs on pre-split planes) serves the nets both shipped experiments run at 16 positions: the pose
// decoder (latent_dim > 0, heads) and the ppc experiment's latent denoiser.  emb_dim 64, a 16-channel first level
// and 32..256-channel ones after it, every split-f16 weight copy present (ABI 5 packers; 16-channel levels zero-padded
// to one 32-channel block: r1d_pack.pad_cin32).
bool pm16_supported(const gldm_r1d_desc *d) {
  if (d->seq_len != 16 || d->emb_dim != 64 || d->groups != 4 || d->dims[0] != 16 || d->cond_rows > 4) return false;
  for (int i = 0; i < d->n_levels; ++i)
    if (d->lv[i].qkvn_w3 <= 0 || d->lv[i].qkvn_s <= 0 || d->lv[i].out_w3 <= 0 || d->lv[i].down_w3 <= 0) return false;
  for (int i = 0; i <= 2 * d->n_levels; ++i)
    if (d->rb[i].c1_w3 <= 0 || d->rb[i].c2_w3 <= 0) return false;
  for (int i = 1; i <= d->n_levels; ++i) {
    const int C = d->dims[i];
    if (!(C == 32 || C == 64 || C == 128 || C == 256)) return false;
    if (i < d->n_levels && C > 128) return false;
  }
  return true;
}
bool wide_engine(const gldm_r1d_desc *d) { return pm_supported(d) || pm16_supported(d); }   // 64-column tiles

using gldm_dev::cu_count;

// Scratch behind the hand-off granules where the 16-position 64-column engine parks the residual stream of a 256-channel
// last level (PG<16>::kW): 64 KiB per workgroup of the launch (at most one per CU), 256-byte aligned.  *base is rounded up to the
// alignment; returns the bytes to add behind it (0: this descriptor never parks).
struct WsLayout { long long tiles, ss_off, park_off, total; int nc; };
long long park_bytes(const gldm_r1d_desc *d, long long tiles, long long *base);
// header | hand-off granules | (256-byte aligned) the decoder's scale/shift table | (256-byte aligned) park scratch
WsLayout ws_layout(const gldm_r1d_desc *d, int n_samples) {
  WsLayout w{};
  w.nc = wide_engine(d) ? 64 : engine_nc();
  const int S = w.nc / d->seq_len;
  w.tiles = (n_samples + S - 1) / S;
  long long bytes = kChainHdrBytes + w.tiles * w.nc * 8;
  w.ss_off = -1;
  if (ss_table_rows(d) > 0) {
    bytes = (bytes + 255) & ~255LL;
    w.ss_off = bytes;
    bytes += (long long)n_samples * ss_table_rows(d) * 4;
  }
  const long long pb = park_bytes(d, w.tiles, &bytes);
  w.park_off = pb > 0 ? bytes : -1;
  w.total = bytes + pb;
  return w;
}
long long park_bytes(const gldm_r1d_desc *d, long long tiles, long long *base) {
  if (pm_supported(d) || !pm16_supported(d) || d->dims[d->n_levels] != 256) return 0;   // the 4-position engine keeps it in LDS
  *base = (*base + 255) & ~255LL;
  const long long wgs = tiles < cu_count() ? tiles : cu_count();
  return wgs * kParkBytes;
}

// Work plan of one launch.  `slots` = workgroups resident at once (two 32-column tiles per CU).  A batch of
// up to `slots` tiles is one workgroup per tile.  A larger one runs `slots` persistent workgroups: each owns
// `rounds` whole tiles, and the `left` tiles beyond the whole rounds are NOT run as a last, partly empty
// round (5120 latents = 640 tiles on 512 slots would take two rounds for 1.25 rounds of work): each is cut
// along the step axis into `chain` segments that a group of `chain` consecutive slots runs in turn, so every
// slot ends up with (nearly) the same number of tile-steps.
struct Plan { int grid, slots, rounds, left, chain, seglen; };

Plan make_plan(int n_samples, int n_steps, int L, int nc, bool allow_chain = true) {
  int slots = cu_count() * (nc == 32 ? 2 : 1);
#ifdef GLDM_DEBUG_KNOBS
  // co-residence experiment: fewer persistent workgroups than CUs, so that another stream's kernels find free
  // CUs while the fused launch runs
  if (const char *e = getenv("GLDM_R1D_SLOTS")) slots = atoi(e) > 0 ? atoi(e) : slots;
#endif
  const int S = nc / L;
  const int tiles = (n_samples + S - 1) / S;
  Plan p{tiles, slots, 1, 0, 1, n_steps};
  if (tiles <= slots) return p;
  if (tiles / slots + 2 > kMaxSegs) { p.grid = -1; return p; }  // more tiles per workgroup than its list holds
  p.grid = slots;
  p.rounds = tiles / slots;
  p.left = tiles - p.rounds * slots;
  if (p.left > 0) {
    int g = slots / p.left;
    if (g > 8) g = 8;             // a hand-off is ~3 us; finer cuts buy nothing
    if (g > n_steps) g = n_steps;
    if (n_steps > 4095) g = 1;  // chain_tag() carries the step in 12 bits: longer loops run their left-over tiles whole
    if (!allow_chain) g = 1;  // DPM++ carries two rows of state per tile (x and the previous denoised): whole tiles only
    p.chain = g < 1 ? 1 : g;
    p.seglen = (n_steps + p.chain - 1) / p.chain;
  }
  return p;
}

template <int NC, int L>
int launch_one(const RunArgs &a, int tiles, hipStream_t s) {
  size_t lds_bytes = (size_t)Geo<NC>::kLdsFloats * sizeof(float);
#ifdef GLDM_DEBUG_KNOBS
  if (getenv("GLDM_R1D_ONE_WG") && lds_bytes < 100 * 1024) lds_bytes = 100 * 1024;  // one workgroup per CU: phases on their own
#endif
  struct Tag {};
  gldm_dev::allow_dynamic_lds<Tag>(reinterpret_cast<const void *>(&r1d_kernel<NC, L>), (int)lds_bytes);
  hipLaunchKernelGGL((r1d_kernel<NC, L>), dim3(tiles), dim3(Geo<NC>::kThreads), lds_bytes, s, a);
  return hipGetLastError() == hipSuccess ? GLDM_OK : GLDM_ERR_LAUNCH;
}

int launch_r1d(const RunArgs &a_in, hipStream_t s) {
  const bool pm4 = pm_supported(&a_in.d), pm16 = !pm4 && pm16_supported(&a_in.d), pm = pm4 || pm16;
  const int L = a_in.d.seq_len, nc = pm ? 64 : engine_nc();
  const Plan pl = make_plan(a_in.n_samples, a_in.n_steps, L, nc, a_in.sched_kind != GLDM_SCHED_DPMPP);
  const int tiles = pl.grid;
  if (tiles <= 0) return GLDM_ERR_UNSUPPORTED;  // > ~250 k samples in one launch: split the batch
  RunArgs a = a_in;
  a.slots = pl.slots; a.rounds = pl.rounds; a.left_tiles = pl.left; a.chain = pl.chain; a.seglen = pl.seglen;
  a.park = nullptr;
  a.pm_nops = 0;
  if (pm) {
    static_assert(8 * GLDM_R1D_MAX_LEVELS + 2 <= kPmMaxOps && sizeof(RunArgs) <= 4096, "64-column tape in the kernel arguments");
    bool use_quad = true;
#ifdef GLDM_DEBUG_KNOBS
    if (getenv("GLDM_R1D_NOQUAD")) use_quad = false;   // A/B: the narrow levels as position-major phases
#endif
    // 16-position nets: the wave-local levels take their scale / shift rows from the per-cloud table or a 64-wide embedding
    if (pm16 && !(a_in.ss_tab != nullptr || a_in.d.emb_dim == 64)) use_quad = false;
    a.pm_nops = build_tape<64>(a_in.d, a.pm_tape, use_quad);
    const WsLayout wl = ws_layout(&a_in.d, a_in.n_samples);
    if (wl.park_off >= 0) a.park = reinterpret_cast<float *>(reinterpret_cast<char *>(a_in.ws) + wl.park_off);
  }
  a.n_cus = cu_count();
#ifdef GLDM_DEBUG_KNOBS
  // diagnostic builds only (make EXTRA=-DGLDM_DEBUG_KNOBS): phase skipping, a start offset for the
  // second workgroup of a CU, and per-op cycle stamps.  The shipped library reads no environment.
  {
    const char *e = getenv("GLDM_R1D_SKIP");
    a.skip = e ? atoi(e) : 0;
    const char *g = getenv("GLDM_R1D_STAGGER_US");
    a.stagger_ticks = g ? atoi(g) * 100 : 0;
  }
  const bool stamp = getenv("GLDM_R1D_STAMP") != nullptr;  // blocks, copies and prints
#else
  const bool stamp = false;
#endif
  static long long *dstamps = nullptr;
  if (stamp && !dstamps) (void)hipMalloc(&dstamps, (kMaxOps + 2) * sizeof(long long));
  a.stamps = stamp ? dstamps : nullptr;
  const int rc = pm4 ? launch_one<64, 4>(a, tiles, s)
                     : (pm16 ? launch_one<64, 16>(a, tiles, s)
                             : (L == 4 ? launch_one<kEngineNC, 4>(a, tiles, s) : launch_one<kEngineNC, 16>(a, tiles, s)));
  if (stamp) {
    static long long host[kMaxOps + 2];
    static const char *names[] = {"", "CONV", "RES4", "LN", "ATT", "QKVLN", "OUTLN", "QKVAT", "QUAD"};
    const bool quad = (pm4 || pm16) && a.pm_nops > 0 && a.pm_tape[0] == OP_QUAD;
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(host, dstamps, sizeof(host), hipMemcpyDeviceToHost);
    // the tape is rebuilt on the host only to label the stamps
    int dims[GLDM_R1D_MAX_LEVELS + 1];
    for (int i = 0; i <= a.d.n_levels; ++i) dims[i] = a.d.dims[i];
    int op = 0;
    auto line = [&](int type, int C, int cout, int taps) {
      printf("op %3d %-5s C=%3d cout=%3d k=%d : %7lld clk\n", op, names[type], C, cout, taps, host[op + 1] - host[op]);
      ++op;
    };
    auto resblock = [&](int C) {
      if (C == 4 && a.d.seq_len == 4) line(2, C, C, 3);
      else { line(1, C, C, 3); line(1, C, C, 3); }
    };
    if (quad) line(8, dims[0], 128, 3);
    for (int lv = quad ? 3 : 0; lv < a.d.n_levels; ++lv) {
      const int C = dims[lv];
      resblock(C); resblock(C);
      if (pm) line(7, C, 384, 1);
      else { line(3, C, C, 0); line(1, C, 192, 1); line(4, C, 64, 0); line(1, C, 192, 1); line(4, C, 64, 0); }
      if (pm && (C == 4 || C == 16 || C == 32 || C == 64 || C == 128)) line(6, 128, C, 1);
      else { line(1, 128, C, 1); line(3, C, C, 0); }
      line(1, C, dims[lv + 1], 3);
    }
    resblock(dims[a.d.n_levels]);
    printf("step total (ops): %lld clk; step prologue (embedding sums, init conv): %lld clk\n", host[op] - host[0],
           host[0] - host[kMaxOps + 1]);
#ifdef GLDM_DEBUG_KNOBS
    if (quad) {
      static long long qs[4][16];
      (void)hipMemcpyFromSymbol(qs, HIP_SYMBOL(g_q_stamp), sizeof(qs));
      static const char *qn4[] = {"pads", "rb4", "rb4", "att4", "down4", "rb32", "rb32", "att32", "down32", "rb64", "rb64", "att64", "down64"};
      static const char *qn16[] = {"pads", "rb16", "rb16", "att16", "down16", "rb32", "rb32", "att32", "down32", "rb64", "rb64", "att64", "down64"};
      const char **qn = pm16 ? qn16 : qn4;
      for (int q = 0; q < 4; ++q) {
        printf("quad %d:", q);
        for (int i = 1; i <= 12; ++i) printf(" %s %lld", qn[i], qs[q][i] - qs[q][i - 1]);
        printf(" | total %lld, polls that waited %lld\n", qs[q][12] - qs[q][0], qs[q][13]);
      }
    }
#ifdef GLDM_WAVE_STAMPS
    if (pm) {   // per-wave view of the last 32 position-major convs of workgroup 0: k-loop / epilogue, relative to wave 0's entry
      static long long wv[8][32][8];
      int cnt[8];
      (void)hipMemcpyFromSymbol(wv, HIP_SYMBOL(g_wv_stamp), sizeof(wv));
      (void)hipMemcpyFromSymbol(cnt, HIP_SYMBOL(g_wv_cnt), sizeof(cnt));
      for (int i = 0; i < 32; ++i) {
        const int e = (cnt[0] + i) & 31;
        printf("conv %3lld->%3lld:", wv[0][e][3] / 1000, wv[0][e][3] % 1000);
        for (int w = 0; w < 8; ++w)
          printf("  w%d in %+5lld loop %6lld epi %6lld (stats %5lld wait %5lld finish %5lld) |", w, wv[w][e][0] - wv[0][e][0],
                 wv[w][e][1] - wv[w][e][0], wv[w][e][2] - wv[w][e][1], wv[w][e][4] - wv[w][e][1], wv[w][e][5] - wv[w][e][4],
                 wv[w][e][2] - wv[w][e][5]);
        printf("\n");
      }
    }
#endif
#endif
  }
  return rc;
}


// ============================================================== fused pointwise MLP layer ==
// y[b, :, cols] = act(W x[b, :, cols] + bias) for a k = 1 Conv1d + folded BatchNorm + ReLU of SharedMLP
// (ext/pvcnn/modules/shared_mlp.py:6-35) in the native [B, C, N] layout, and optionally, on the accumulators
// before they are stored, the head  z[b, :, cols] = Wh y + bh  (PVCNNEncoder: conv_downscale + out_layer[0]
// folded, pc_encoders.py:104-111).  With the head fused `y` may be NULL: the encoder's [B, 1536, N] tensor
// (1.6 GB per 256 clouds) then never reaches HBM.
// A persistent workgroup of 8 waves takes 32 points of one cloud at a time: the [cin][32] input tile is staged
// once in LDS (swizzled like the engine's activations) and every wave sweeps its share of the output rows over it
// with the engine's GEMM core, weights streamed as buffer-load fragments.  The head product uses each 16-row
// block of y straight from the accumulators as the B operand (lane (kq, col) register r = row 4 kq + r = k-step
// r of a 16x16x4 MFMA), against head weights packed in that k order; the waves' partial z tiles meet in LDS.
struct PwArgs {
  const float *x, *w, *bias, *head_w, *head_b;
  float *y, *z;
  int cin, cout, n, relu, hout, tiles_per_cloud, total_tiles;
  // optional layer in front (x [b, cin0, n] -> relu(W0 x + b0) = the [cin][32] tile of the main layer, never in HBM)
  const float *w0, *bias0;
  int cin0;
  int dyn_first;   // split-f16 kernel: units (pairs of m-tiles) >= dyn_first are handed out at run time
  int ticket_off;  // ... from a ticket at this float index of the LDS plan
  int x0_in_planes; // 48-column tiles: the front layer's f32 tile lies under the planes (see pointwise_mlp_sp_kernel)
  // split-f16 kernel, ADD instantiation: an addend in front of the activation, add[cloud * add_bs + row * add_rs + col * add_cs]
  // (a per-cloud bias: bs = cout, rs = 1, cs = 0; a [b, cout, n] tensor: bs = cout * n, rs = n, cs = 1)
  const float *add;
  long long add_bs, add_rs, add_cs;
  // split-f16 kernel: range scales (range_pow2).  The staged input tile's is measured; the front layer's output planes take
  // theirs from the bound gain0_r * max |x| + gain0_b (largest row sum of |W0|, largest |bias0|).  rng_off: float index of the
  // eight per-wave range words in the LDS plan.  ranged == 0: operands are split as they are.
  int ranged, rng_off;
  float gain0_r, gain0_b;
  int y_point_major;   // split-f16 kernel: y is [b, n, cout] (a lane's four consecutive rows of a column: one 16-byte store)
  int cin_rows;        // split-f16 kernel without a front layer: rows x really has (cin = that padded to whole 128-deep trips
                       // of the weight ring: the planes of the rows beyond are zero, like the weights' columns there)
};

__global__ __launch_bounds__(512, 2) void pointwise_mlp_kernel(const PwArgs a) {
  constexpr int NC = 32;
  extern __shared__ float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  Ctx c{a.w, lds, tid, wave, lane, 0, 2};
  const int col = lane & 15, kq = lane >> 4;
  const int cblocks = a.cin >> 4, mtiles = a.cout >> 4, mt_per_wave = mtiles >> 3;
  float *zpart = lds + a.cin * NC;  // [8 waves][16 rows][32 cols]
  const WStream hw(a.head_w ? a.head_w : a.w, lane);
  for (int tile = blockIdx.x; tile < a.total_tiles; tile += gridDim.x) {
    const int b = tile / a.tiles_per_cloud, c0 = (tile - b * a.tiles_per_cloud) * NC;
    __syncthreads();  // the previous tile's readers are done
    if (a.w0) {
      // layer in front: stage its [cin0][32] input tile behind the z partials, sweep its output rows (= the main
      // layer's input rows) with the same GEMM core and leave them in LDS as the main layer's tile
      float *x0 = zpart + 8 * 16 * NC;
      const float *xb0 = a.x + (size_t)b * a.cin0 * a.n + c0;
      for (int i = tid; i < a.cin0 * 8; i += 512) {
        const int row = i >> 3, q = i & 7;
        *reinterpret_cast<f32x4 *>(x0 + swz<NC>(row, 4 * q)) = *reinterpret_cast<const f32x4 *>(xb0 + (size_t)row * a.n + 4 * q);
      }
      __syncthreads();
      const int cb0 = a.cin0 >> 4, mt_per_wave0 = a.cin >> 7;  // cin output rows = cin / 16 m-tiles over 8 waves
      for (int ps = 0; ps < mt_per_wave0; ps += 2) {
        const int mt0 = wave * mt_per_wave0 + ps;
        f32x4 acc[2][2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          const f32x4 bv = *reinterpret_cast<const f32x4 *>(a.bias0 + 16 * (mt0 + mi) + 4 * kq);
          acc[mi][0] = bv;
          acc[mi][1] = bv;
        }
        gemm_fast_pf<NC, 4, 1, 2, 2, 2>(c, a.w0, cb0, mt0, 0, x0, acc);
        lds_f *d3 = (lds_f *)lds;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
              d3[swz<NC>(16 * (mt0 + mi) + 4 * kq + r, 16 * ni + col)] = fmaxf(acc[mi][ni][r], 0.f);
      }
    } else {
      const float *xb = a.x + (size_t)b * a.cin * a.n + c0;
      for (int i = tid; i < a.cin * 8; i += 512) {
        const int row = i >> 3, q = i & 7;
        const f32x4 v = *reinterpret_cast<const f32x4 *>(xb + (size_t)row * a.n + 4 * q);
        *reinterpret_cast<f32x4 *>(lds + swz<NC>(row, 4 * q)) = v;
      }
    }
    __syncthreads();
    f32x4 zacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    for (int ps = 0; ps < mt_per_wave; ps += 2) {
      const int mt0 = wave * mt_per_wave + ps;
      f32x4 acc[2][2];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const f32x4 bv = *reinterpret_cast<const f32x4 *>(a.bias + 16 * (mt0 + mi) + 4 * kq);
        acc[mi][0] = bv;
        acc[mi][1] = bv;
      }
      if ((cblocks & 3) == 0) gemm_fast_pf<NC, 4, 1, 2, 2, 4>(c, a.w, cblocks, mt0, 0, lds, acc);
      else gemm_fast_pf<NC, 4, 1, 2, 2, 2>(c, a.w, cblocks, mt0, 0, lds, acc);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        if (a.relu) {
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[mi][ni][r] = fmaxf(acc[mi][ni][r], 0.f);
        }
        if (a.y) {
          float *yb = a.y + ((size_t)b * a.cout + 16 * (mt0 + mi) + 4 * kq) * a.n + c0 + col;
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) __builtin_nontemporal_store(acc[mi][ni][r], yb + (size_t)r * a.n + 16 * ni);
        }
        if (a.head_w) {
          const f32x4 ah = hw[(size_t)(mt0 + mi) * 64];
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
              zacc[ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(ah[r], acc[mi][ni][r], zacc[ni], 0, 0, 0);
        }
      }
    }
    if (a.head_w) {
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) zpart[(wave * 16 + 4 * kq + r) * NC + 16 * ni + col] = zacc[ni][r];
      __syncthreads();
      for (int i = tid; i < a.hout * NC; i += 512) {
        const int row = i / NC, cc = i - row * NC;
        float v = a.head_b ? a.head_b[row] : 0.f;
#pragma unroll
        for (int w8 = 0; w8 < 8; ++w8) v += zpart[(w8 * 16 + row) * NC + cc];
        a.z[((size_t)b * a.hout + row) * a.n + c0 + cc] = v;
      }
    }
  }
}

// ---- the same layer(s) on split-f16 operands -------------------------------------------------------------------------
// pointwise_mlp_kernel with the main GEMM on v_mfma_f32_16x16x32_bf16 (6 partial products per f32 product, see the
// split-f16 core of the position-major engine): weights pre-split on the host (mfma_a_fragments_f16x2), the [cin][32]
// input tile split ONCE while it is staged and kept in LDS as B-fragment planes
//   [32-channel block][plane hi|mid|lo][g][32 columns][8 bf16]      (6 KiB per block; cin = 768: 144 KiB)
// so the eight waves' k-loops are ds_read_b128 + buffer loads + MFMA.  A 32-column tile re-uses a weight fragment
// for two n-tiles only: 7 MB of fragments per tile, 62 B/clk if the MFMAs were never to wait -- above the 50 B/clk a CU
// draws from L2 (tools/micro/l2_stream), so fragments are requested four blocks ahead (ring of four register sets: 24 KiB
// in flight per wave) and the ring is kept full across the units of output rows.
// (Measured and dropped: 64-column tiles with K walked in 256-channel chunks -- planes of a chunk in LDS, accumulators of
// half the output rows kept across the chunks, two passes, front layer recomputed per pass on the bf16 pipe: half the
// weight bytes per column, yet the same 0.97-1.04 ms per 329 clouds as this kernel's 1.04: the stream is not what it
// waits for in the end.)
// The optional layer in front (96 -> 768: an eighth of the FLOPs) runs on the same pipe (pw_front_split: 20-25 k cycles
// per tile on the f32 pipe before) and writes its ReLU output straight into those planes; the head product is taken on the
// accumulators exactly as in the f32 kernel (the C layout of the two MFMA shapes is the same).
// f32 [row][NC] tile of the front layer: swz<32> on the first 32 columns (the two n-tiles trade places on odd rows), any
// further n-tile in place
template <int NC>
__device__ __forceinline__ int pw_swz(int row, int col) { return row * NC + (col < 32 ? (col ^ ((row & 1) << 4)) : col); }
template <int NC>   // columns of the tile: 32 or 48
__device__ __forceinline__ void store_planes4_pw(float *planes, int c0, int n, float v0, float v1, float v2, float v3) {
  unsigned h0, h1, l0, l1;
  split_f16x2(v0, v1, h0, l0);
  split_f16x2(v2, v3, h1, l1);
  const int a = ((((c0 >> 5) * kSplit) * 4 + ((c0 >> 3) & 3)) * NC + n) * 4 + ((c0 >> 2) & 1) * 2;   // dwords
  lds_u2 *d = (lds_u2 *)(planes + a);
  d[0] = u32x2_t{h0, h1};
  d[8 * NC] = u32x2_t{l0, l1};   // next plane: 4 * NC * 4 dwords
}

// The layer in front of the split-f16 main layer, on the same pipe: x0 = the f32 [cin0][32] tile (swizzled), w0s =
// split fragments of W0 [cin x cin0], KB0 = cin0 / 32.  A wave splits the whole tile ONCE into registers (its B planes
// serve all of the wave's m-tiles) and walks its m-tiles in pairs; the A registers of a (m-tile, block) are refilled
// with the next pair's fragments as soon as its MFMAs have issued.  Output: ReLU, split, into the main layer's planes.
// bsc = 1 / (range scale of the input tile), osc = that scale / the scale of the output planes (range_pow2; 1 and 1 for
// ordinary data)
template <int KB0, int NT>
__device__ __forceinline__ void pw_front_split(const WStream &w0s, const float *bias0, const float *x0, float *planes,
                                               int wave, int lane, int mt_per_wave0, bool x0_in_planes, float bsc, float osc) {
  constexpr int NC = 16 * NT;
  const int col = lane & 15, kq = lane >> 4;
  u32x4 bp[KB0][NT][kSplit];
#pragma unroll
  for (int kb = 0; kb < KB0; ++kb)
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = x0[pw_swz<NC>(32 * kb + 8 * kq + j, 16 * ni + col)] * bsc;
      split_planes8(v, bp[kb][ni]);
    }
  if (x0_in_planes) __syncthreads();   // 48-column tiles: the f32 tile lies under the planes this layer is about to write
  u32x4 af[2][KB0][kSplit];
  const int mt_first = wave * mt_per_wave0, mt_last = mt_first + mt_per_wave0 - 2;
  auto load_a = [&](int mi, int kb, int mt0) {
#pragma unroll
    for (int pl = 0; pl < kSplit; ++pl) af[mi][kb][pl] = w0s.raw_at(((mt0 + mi) * KB0 + kb) * kFragBytes, pl * 1024);
  };
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int kb = 0; kb < KB0; ++kb) load_a(mi, kb, mt_first);
  for (int mt0 = mt_first; mt0 <= mt_last; mt0 += 2) {
    const int mtn = mt0 + 2 <= mt_last ? mt0 + 2 : mt_last;
    f32x4 acc[2][NT];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const f32x4 bv = *reinterpret_cast<const f32x4 *>(bias0 + 16 * (mt0 + mi) + 4 * kq) * bsc;
#pragma unroll
      for (int ni = 0; ni < NT; ++ni) acc[mi][ni] = bv;
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int kb = 0; kb < KB0; ++kb) {
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) acc[mi][ni] = mfma_split(af[mi][kb], bp[kb][ni], acc[mi][ni]);
        __builtin_amdgcn_sched_barrier(0);
        load_a(mi, kb, mtn);   // pinned here: the scheduler sinks such requests to their first use otherwise
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < NT; ++ni)
        store_planes4_pw<NC>(planes, 16 * (mt0 + mi) + 4 * kq, 16 * ni + col, fmaxf(acc[mi][ni][0], 0.f) * osc,
                             fmaxf(acc[mi][ni][1], 0.f) * osc, fmaxf(acc[mi][ni][2], 0.f) * osc, fmaxf(acc[mi][ni][3], 0.f) * osc);
  }
}

#ifdef GLDM_DEBUG_KNOBS
__device__ long long g_pw_stamp[64];
#define GLDM_PW_STAMP(i) \
  do { if (blockIdx.x == 5 && tile == 5 + 2 * (int)gridDim.x && (threadIdx.x & 63) == 0 && (wave == 0 || wave == 7)) \
         g_pw_stamp[(wave ? 32 : 0) + (i)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define GLDM_PW_STAMP(i) do {} while (0)
#endif
// NT: n-tiles per tile.  2 = 32 points (96 KiB of planes at cin = 768).  3 = 48 points (144 KiB): a weight fragment then
// serves three n-tiles -- with three f16 products per block the kernel is bound by the CU's L2 rate (5 MB of fragments per
// tile at 52 B/clk = 96 k cycles against 59 k of MFMAs at 32 points), so bytes per POINT are what counts.  n % 16 == 0: a
// cloud's last tile holds 1-3 whole n-tiles (`ntv`); the others are computed on zeros and never stored.
// MU: m-tiles per unit of output rows (2; 1 for layers of fewer than 256 rows, whose 4-7 two-tile units left waves idle:
// the 128-row feature-propagation layers and the set-abstraction first layer per point)
// (Tried for MU = 1: a 128-register bound, two workgroups per CU -- these launches are short tiles whose staging -> barrier
// -> k-loop -> store chain is latency -- 96 spilled registers: the ring of four A sets and two B sets does not fit.)
template <bool ADD, int NT, int MU = 2>
__global__ __launch_bounds__(512, 2) void pointwise_mlp_sp_kernel(const PwArgs a) {
  constexpr int NC = 16 * NT;
  extern __shared__ float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 15, kq = lane >> 4;
  const int kb32 = a.cin >> 5, mtiles = a.cout >> 4;
  float *planes = lds;                       // [kb32][kSplit][4][NC][4 dwords]
  float *zpart = lds;                        // [8 waves][16 rows][NC cols], over the planes once they are dead
  float *zdyn = lds + a.cin * (NC / 2) * kSplit;   // behind the planes: head products of the drawn units [unit - dyn_first][hout][NC]
  // front layer's f32 input tile [cin0][NC]: behind the planes, under zdyn (dead by then); 48-column tiles have no room
  // there -- it lies UNDER the planes and the front layer takes it into registers, then a barrier, before it writes them
  float *x0 = a.x0_in_planes ? lds : zdyn;
  int *ticket = (int *)(lds + a.ticket_off); // next unit of output rows to hand out (main layer)
  const WStream hw(a.head_w ? a.head_w : a.w, lane);
  const WStream wv(a.w, lane);
  const lds_u4 *pl3 = (const lds_u4 *)planes + kq * NC + col;   // + ((kb * kSplit + plane) * 4) * NC + 16 ni
  for (int tile = blockIdx.x; tile < a.total_tiles; tile += gridDim.x) {
    const int b = tile / a.tiles_per_cloud, c0 = (tile - b * a.tiles_per_cloud) * NC;
    const int ntv = min(NT, (a.n - c0) >> 4);   // whole n-tiles of this tile that exist
    __syncthreads();  // the previous tile's readers are done
    GLDM_PW_STAMP(0);
    if (tid == 0) *ticket = a.dyn_first;
    // range scale of the main layer's planes (range_pow2): the accumulators run in its units, `v = acc * s_main + bias` below
    float s_main = 1.0f;
    float *rng = lds + a.rng_off;   // [8]: the waves' largest staged magnitudes
    auto range_publish = [&](float mx) {
      mx = half_max(row_pair_max(row16_max(mx)));
      if (lane == 0) rng[wave] = mx;
    };
    auto range_read = [&]() {
      const f32x4 r0 = *reinterpret_cast<const f32x4 *>(rng), r1 = *reinterpret_cast<const f32x4 *>(rng + 4);
      const float mx = fmaxf(fmaxf(fmaxf(r0[0], r0[1]), fmaxf(r0[2], r0[3])), fmaxf(fmaxf(r1[0], r1[1]), fmaxf(r1[2], r1[3])));
      return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(mx)));
    };
    if (a.w0) {
      const float *xb0 = a.x + (size_t)b * a.cin0 * a.n + c0;
      float mx = 0.f;
      for (int i = tid; i < a.cin0 * (NC / 4); i += 512) {
        const int row = i / (NC / 4), q = i - row * (NC / 4);
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (4 * q < 16 * ntv) v = *reinterpret_cast<const f32x4 *>(xb0 + (size_t)row * a.n + 4 * q);
        *reinterpret_cast<f32x4 *>(x0 + pw_swz<NC>(row, 4 * q)) = v;
        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
      }
      if (a.ranged) range_publish(mx);
      __syncthreads();
      GLDM_PW_STAMP(1);
      float bsc = 1.0f, osc = 1.0f;
      if (a.ranged) {
        const float m0 = range_read(), s0 = range_pow2(m0);
        s_main = range_pow2(a.gain0_r * m0 + a.gain0_b);
        bsc = pow2_inv(s0);
        osc = s0 * pow2_inv(s_main);
      }
      const int mt_per_wave0 = a.cin >> 7;   // cin / 16 m-tiles over 8 waves
      const WStream w0s(a.w0, lane);
      switch (a.cin0 >> 5) {
        case 1: pw_front_split<1, NT>(w0s, a.bias0, x0, planes, wave, lane, mt_per_wave0, a.x0_in_planes != 0, bsc, osc); break;
        case 2: pw_front_split<2, NT>(w0s, a.bias0, x0, planes, wave, lane, mt_per_wave0, a.x0_in_planes != 0, bsc, osc); break;
        default: pw_front_split<3, NT>(w0s, a.bias0, x0, planes, wave, lane, mt_per_wave0, a.x0_in_planes != 0, bsc, osc); break;
      }
    } else {
      // stage + split: item = (8-channel group, column)
      const float *xb = a.x + (size_t)b * a.cin_rows * a.n + c0;
      // The tile's range scale needs its largest magnitude before anything is split.  Up to kHold items per thread (cin <=
      // 256 at 32 points: the feature-propagation and per-point layers of the set-abstraction backbones) the staged values
      // wait in registers across the exchange barrier: ONE pass over the input.  Wider tiles take a first pass for the maximum
      // and read the tile again (from L2): measured on the feature-propagation layers of PointNet2SSG, the two-pass form alone
      // cost 60-90 % of a launch.
      constexpr int kHold = 2;
      const int items = (a.cin >> 3) * NC;
      auto stage_store = [&](int i, float (&v)[8], float inv) {
        const int kg = i / NC, scol = i - kg * NC, row = 8 * kg;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= inv;
        u32x4 pl[kSplit];
        split_planes8(v, pl);
        lds_u4 *d = (lds_u4 *)planes + (((row >> 5) * kSplit) * 4 + ((row >> 3) & 3)) * NC + scol;
        d[0] = pl[0];
        d[4 * NC] = pl[1];
      };
      auto stage_load = [&](int i, float (&v)[8]) {
        const int kg = i / NC, scol = i - kg * NC, row = 8 * kg;
        const bool in = scol < 16 * ntv && row < a.cin_rows;   // (cin_rows % 8 == 0: whole groups)
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = in ? xb[(size_t)(row + j) * a.n + scol] : 0.f;
      };
      if (NT == 2 && a.ranged && items <= kHold * 512) {   // (48-point tiles exist for inputs of 640 rows and more only)
        float hv[kHold][8];
        float mx = 0.f;
#pragma unroll
        for (int q = 0; q < kHold; ++q) {
          const int i = tid + 512 * q;
          if (i < items) {
            stage_load(i, hv[q]);
#pragma unroll
            for (int j = 0; j < 8; ++j) mx = fmaxf(mx, fabsf(hv[q][j]));
          }
        }
        range_publish(mx);
        __syncthreads();
        s_main = range_pow2(range_read());
        const float inv = pow2_inv(s_main);
#pragma unroll
        for (int q = 0; q < kHold; ++q) {
          const int i = tid + 512 * q;
          if (i < items) stage_store(i, hv[q], inv);
        }
      } else {
        float inv = 1.0f;
        if (a.ranged) {
          float mx = 0.f;
          for (int i = tid; i < items; i += 512) {
            float v[8];
            stage_load(i, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) mx = fmaxf(mx, fabsf(v[j]));
          }
          range_publish(mx);
          __syncthreads();
          s_main = range_pow2(range_read());
          inv = pow2_inv(s_main);
        }
        for (int i = tid; i < items; i += 512) {
          float v[8];
          stage_load(i, v);
          stage_store(i, v, inv);
        }
      }
    }
    GLDM_PW_STAMP(2);
    __syncthreads();
    GLDM_PW_STAMP(3);
    f32x4 zacc[NT];
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) zacc[ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    // ---- the output rows in units of two m-tiles, handed out at run time.  With a fixed share per wave the older wave
    // of a SIMD gets the matrix pipe whenever it wants it, finishes its share at 95 % of the pair's rate and then idles
    // at the tile's last barrier while its partner, alone, cannot hide its own LDS / weight latencies (stamps: wave 0
    // done after 131 k cycles, wave 7 after 166 k, a lone wave at 58 % of the pipe).  A wave that is done takes the next
    // unit off an LDS ticket instead; the unit after the current one is drawn before the current k-loop so that its
    // first weight fragments are requested from inside that loop (the ring of four A sets never drains), and the B
    // planes of block k + 1 are read in front of the MFMAs of block k.
    const int units = mtiles / MU;
    u32x4 af[4][MU][kSplit];
    auto load_a = [&](int buf, int mt0, int kb) {
#pragma unroll
      for (int mi = 0; mi < MU; ++mi)
#pragma unroll
        for (int pl = 0; pl < kSplit; ++pl) af[buf][mi][pl] = wv.raw_at(((mt0 + mi) * kb32 + kb) * kFragBytes, pl * 1024);
    };
    u32x4 bs[2][NT][kSplit];
    auto load_b = [&](int buf, int kb) {
#pragma unroll
      for (int ni = 0; ni < NT; ++ni)
#pragma unroll
        for (int pl = 0; pl < kSplit; ++pl) bs[buf][ni][pl] = pl3[(kb * kSplit + pl) * 4 * NC + 16 * ni];
    };
    auto draw = [&]() {
      int t = 0;
      if (lane == 0) t = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      return __builtin_amdgcn_readfirstlane(t);
    };
    // Units below dyn_first are dealt round robin (wave w: w, w + 8, ...), the rest drawn.  The head sum must not depend
    // on who drew what: a drawn unit's head product goes to its own LDS slot (zdyn, over the dead front-layer tile), only
    // the dealt ones accumulate in the wave's zacc, and the final sum walks waves, then slots, in index order.
    const int dyn_first = a.dyn_first;
    int unit = wave;
#pragma unroll
    for (int u = 0; u < 4; ++u) load_a(u, MU * (unit < units ? unit : 0), u);   // a wave without a unit requests unit 0's (unused)
    load_b(0, 0);
    while (unit < units) {
      const int mt0 = MU * unit;
      const int nxt = unit + 8 < dyn_first ? unit + 8 : draw();
      const int mtn = MU * (nxt < units ? nxt : unit);   // past the end: harmless re-reads of this unit's fragments
      // bias and head fragments of this unit: requested now, used behind the k-loop (the bias is added last)
      f32x4 acc[MU][NT], bv[MU], ah[MU];
#pragma unroll
      for (int mi = 0; mi < MU; ++mi) {
        bv[mi] = *reinterpret_cast<const f32x4 *>(a.bias + 16 * (mt0 + mi) + 4 * kq);
        ah[mi] = hw[(size_t)(a.head_w ? mt0 + mi : 0) * 64];
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      for (int kb0 = 0; kb0 < kb32; kb0 += 4) {
        const bool tail = kb0 + 4 >= kb32;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int kb = kb0 + u;
          load_b((u + 1) & 1, kb + 1 < kb32 ? kb + 1 : 0);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int mi = 0; mi < MU; ++mi)
#pragma unroll
            for (int ni = 0; ni < NT; ++ni) acc[mi][ni] = mfma_split(af[u][mi], bs[u & 1][ni], acc[mi][ni]);
          __builtin_amdgcn_sched_barrier(0);
          load_a(u, tail ? mtn : mt0, tail ? u : kb + 4);
        }
      }
#pragma unroll
      for (int mi = 0; mi < MU; ++mi) {
#pragma unroll
        for (int ni = 0; ni < NT; ++ni)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v = __builtin_fmaf(acc[mi][ni][r], s_main, bv[mi][r]);   // s_main = 1: the plain sum, bit for bit
            if constexpr (ADD)
              v += a.add[(long long)b * a.add_bs + (long long)(16 * (mt0 + mi) + 4 * kq + r) * a.add_rs +
                         (long long)(c0 + (ni < ntv ? 16 * ni + col : col)) * a.add_cs];
            acc[mi][ni][r] = a.relu ? fmaxf(v, 0.f) : v;
          }
        if (a.y && a.y_point_major) {
          float *yb = a.y + ((size_t)b * a.n + c0 + col) * a.cout + 16 * (mt0 + mi) + 4 * kq;
#pragma unroll
          for (int ni = 0; ni < NT; ++ni)
            if (ni < ntv) *reinterpret_cast<f32x4 *>(yb + (size_t)16 * ni * a.cout) = acc[mi][ni];
        } else if (a.y) {
          float *yb = a.y + ((size_t)b * a.cout + 16 * (mt0 + mi) + 4 * kq) * a.n + c0 + col;
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int ni = 0; ni < NT; ++ni)
              if (ni < ntv) __builtin_nontemporal_store(acc[mi][ni][r], yb + (size_t)r * a.n + 16 * ni);
        }
      }
      if (a.head_w) {
        const bool drawn = unit >= dyn_first;
        f32x4 zu[NT];
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) zu[ni] = drawn ? f32x4{0.f, 0.f, 0.f, 0.f} : zacc[ni];
#pragma unroll
        for (int mi = 0; mi < MU; ++mi) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int ni = 0; ni < NT; ++ni)
              zu[ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(ah[mi][r], acc[mi][ni][r], zu[ni], 0, 0, 0);
        }
        if (drawn) {
          float *slot = zdyn + (unit - dyn_first) * a.hout * NC;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (4 * kq + r < a.hout) {
#pragma unroll
              for (int ni = 0; ni < NT; ++ni) slot[(4 * kq + r) * NC + 16 * ni + col] = zu[ni][r];
            }
        } else {
#pragma unroll
          for (int ni = 0; ni < NT; ++ni) zacc[ni] = zu[ni];
        }
      }
      unit = nxt;
    }
    GLDM_PW_STAMP(16);
    if (a.head_w) {
      __syncthreads();  // every wave is done with the planes: the z partials go over them
      GLDM_PW_STAMP(17);
#pragma unroll
      for (int ni = 0; ni < NT; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) zpart[(wave * 16 + 4 * kq + r) * NC + 16 * ni + col] = zacc[ni][r];
      __syncthreads();
      // fixed summation tree: four lanes per output (waves 2p, 2p + 1 and every fourth slot from p), then the quad
      const int nd = units - dyn_first;
      for (int i0 = 0; i0 < a.hout * NC * 4; i0 += 512) {
        const int i = i0 + tid, o = i >> 2, p = i & 3;
        const bool live = o < a.hout * NC;
        const int row = live ? o / NC : 0, cc = live ? o - row * NC : 0;
        float v = zpart[((2 * p) * 16 + row) * NC + cc] + zpart[((2 * p + 1) * 16 + row) * NC + cc];
        for (int d = p; d < nd; d += 4) v += zdyn[(d * a.hout + row) * NC + cc];
        v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
        v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
        if (live && p == 0 && cc < 16 * ntv) a.z[((size_t)b * a.hout + row) * a.n + c0 + cc] = v + (a.head_b ? a.head_b[row] : 0.f);
      }
    }
    GLDM_PW_STAMP(18);
  }
}


// ======================================================== fused set abstraction, 128-column tiles ==
// sa_mlp_kernel at twice the tile: 128 columns = 128 / U centres x U neighbours per workgroup (8 waves, one workgroup
// per CU).  Every weight fragment then serves 8 n-tiles, a layer's fill / epilogue / barrier is paid once per 128
// columns, and the last layer's output is never stored: max over a centre's neighbours is taken on the accumulators
// (in-lane over the centre's n-tiles, DPP over the 16 columns of a tile; ReLU after the max, it is monotone) and only
// [cout][centres] leaves the CU.  LDS: region A [max(cin_pad0, cout1)][128] (the gathered tile, later layer 2's
// output) + region B [cout0][128] (+ [cout2] for 4 layers).  Shapes outside this plan run on sa_mlp_kernel.

template <int MT, int NT>
__device__ __forceinline__ void sa2_tiles(const Ctx &c, const SaArgs &a, int l, int mt0, int nt0, const float *src,
                                          float *dst, bool last, int j0, float *outb) {
  constexpr int NC = 128;
  const int col = c.lane & 15, kq = c.lane >> 4;
  const float *wp = a.weights + a.w_off[l], *bias = a.weights + a.b_off[l];
  f32x4 acc[MT][NT];
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) {
    const f32x4 bv = *reinterpret_cast<const f32x4 *>(bias + 16 * (mt0 + mi) + 4 * kq);
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) acc[mi][ni] = bv;
  }
  const int cblocks = a.cin_pad[l] >> 4;
  if ((cblocks & 3) == 0) gemm_fast_pf<NC, 16, 1, MT, NT, 4>(c, wp, cblocks, mt0, nt0, src, acc);
  else gemm_fast_pf<NC, 16, 1, MT, NT, 2>(c, wp, cblocks, mt0, nt0, src, acc);  // the launcher checked: even
  if (!last) {
    lds_f *d3 = (lds_f *)dst;
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni)
          d3[swz<NC>(16 * (mt0 + mi) + 4 * kq + r, 16 * (nt0 + ni) + col)] = fmaxf(acc[mi][ni][r], 0.f);
    return;
  }
  // max over each centre's U columns: tpc = U / 16 n-tiles per centre (1, 2 or 4; NT is a multiple of it)
  const int tpc = a.u >> 4;
#pragma unroll
  for (int mi = 0; mi < MT; ++mi)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float m[NT];
#pragma unroll
      for (int ni = 0; ni < NT; ++ni) m[ni] = acc[mi][ni][r];
      if (tpc >= 2) {
#pragma unroll
        for (int ni = 0; ni < NT; ni += 2) m[ni] = fmaxf(m[ni], m[ni + 1 < NT ? ni + 1 : ni]);
      }
      if (tpc >= 4) {
#pragma unroll
        for (int ni = 0; ni < NT; ni += 4) m[ni] = fmaxf(m[ni], m[ni + 2 < NT ? ni + 2 : ni]);
      }
#pragma unroll
      for (int ni = 0; ni < NT; ++ni) {
        if (ni % tpc) continue;  // wave uniform
        const float v = fmaxf(row16_max(m[ni]), 0.f);
        const int jj = (nt0 + ni) / tpc;  // centre of the tile
        if (col == 0 && j0 + jj < a.m) outb[(size_t)(16 * (mt0 + mi) + 4 * kq + r) * a.m + j0 + jj] = v;
      }
    }
}

// The gather of tile t + 1 is requested in front of tile t's last layer and stored after it: its scattered round
// trips (12 k cycles exposed per tile before) run under the longest GEMM of the tile.  Everything about it is
// UNCONDITIONAL -- clamped tile index, clamped addresses, the store of a last redundant tile -- because a load behind a
// branch, or one whose only consumer is behind a branch, is waited for on the spot.  (The in-order vmcnt makes the
// layer's first weight fragment wait for the gather's loads: a few hundred cycles once per tile, measured.)
constexpr int kSaFly = 32;  // feature rows per thread in flight: 4 row quarters x 32 = 128 feature channels

struct SaTile {
  const float *pts, *ctr, *feat;
  int id, jj, j0;
  bool live;
};

__device__ __forceinline__ SaTile sa2_tile(const SaArgs &a, int t, int tiles_per_cloud, int cpt, int col) {
  const int b = t / tiles_per_cloud, tile = t - b * tiles_per_cloud;
  SaTile s;
  s.j0 = tile * cpt;
  s.jj = col / a.u;
  s.live = s.j0 + s.jj < a.m;
  s.pts = a.points + (size_t)b * 3 * a.n;
  s.ctr = a.centers + (size_t)b * 3 * a.m;
  s.feat = a.feat ? a.feat + (size_t)b * a.c * a.n : a.points;
  const int32_t *idx = a.idx + ((size_t)b * a.m + s.j0) * a.u;
  s.id = idx[s.live ? col : 0];
  s.id = s.live ? s.id : 0;
  return s;
}

__device__ __forceinline__ void sa2_gather_load(const SaArgs &a, const SaTile &s, int rq, float &xyz, float (&v)[kSaFly]) {
  const int r3 = rq < 3 ? rq : 0;
  xyz = s.pts[r3 * a.n + s.id] - s.ctr[r3 * a.m + (s.live ? s.j0 + s.jj : 0)];
  const int cmax = a.c > 0 ? a.c - 1 : 0;
#pragma unroll
  for (int q = 0; q < kSaFly; ++q) {
    const int f = rq + 4 * q;
    v[q] = s.feat[(size_t)(f < cmax ? f : cmax) * a.n + s.id];
  }
}

__device__ __forceinline__ void sa2_gather_store(const SaArgs &a, const SaTile &s, int rq, int col, float xyz,
                                                 const float (&v)[kSaFly], float *A) {
  constexpr int NC = 128;
  lds_f *A3 = (lds_f *)A;
  const int rows = a.cin_pad[0];
  if (rq < 3) A3[swz<NC>(rq, col)] = s.live ? xyz : 0.f;
#pragma unroll
  for (int q = 0; q < kSaFly; ++q) {
    const int f = rq + 4 * q;
    if (3 + f < rows) A3[swz<NC>(3 + f, col)] = (s.live && f < a.c) ? v[q] : 0.f;
  }
  for (int r = 3 + 4 * kSaFly + rq; r < rows; r += 4) A3[swz<NC>(r, col)] = 0.f;  // zero pad beyond 131 rows
}

__global__ __launch_bounds__(512, 1) void sa_mlp2_kernel(const SaArgs a, int rows_a, int tiles_per_cloud, int total_tiles) {
  constexpr int NC = 128;
  extern __shared__ float lds[];
  Ctx c{a.weights, lds, (int)threadIdx.x, __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), (int)threadIdx.x & 63,
        0, 8};
  const int cpt = NC / a.u;  // centres per tile
  float *A = lds, *B = lds + (size_t)rows_a * NC;
  const int w = c.wave, col = c.tid & (NC - 1), rq = c.tid >> 7;
  auto layer = [&](int l, const float *src, float *dst, int j0, float *outb) {
    const bool last = l == a.n_layers - 1;
    const int mtiles = a.cout[l] >> 4;
    // the lane ids are laundered per call: otherwise every variant's lane-derived LDS offsets are hoisted out of the
    // tile loop as invariants and live (spilled) across the whole kernel
    Ctx cl = c;
    asm volatile("" : "+v"(cl.tid), "+v"(cl.lane));
    if (mtiles >= 8) {  // one m-tile x all 8 n-tiles per pass: every weight fragment serves 128 columns
      for (int p = 0; p < (mtiles >> 3); ++p) sa2_tiles<1, 8>(cl, a, l, w + 8 * p, 0, src, dst, last, j0, outb);
    } else if (mtiles == 4) sa2_tiles<1, 4>(cl, a, l, w & 3, 4 * (w >> 2), src, dst, last, j0, outb);
    else sa2_tiles<1, 2>(cl, a, l, w & 1, 2 * (w >> 1), src, dst, last, j0, outb);
  };
  // persistent workgroups (a tile is 45 us of work and a workgroup launch several)
  int t = blockIdx.x;
  {
    const SaTile s = sa2_tile(a, t, tiles_per_cloud, cpt, col);
    float xyz, v[kSaFly];
    sa2_gather_load(a, s, rq, xyz, v);
    sa2_gather_store(a, s, rq, col, xyz, v, A);
  }
  __syncthreads();
  for (; t < total_tiles; t += gridDim.x) {
    const int b = t / tiles_per_cloud, j0 = (t - b * tiles_per_cloud) * cpt;
    float *outb = a.out + (size_t)b * a.cout[a.n_layers - 1] * a.m;
    float *src = A, *dst = B;
    for (int l = 0; l + 1 < a.n_layers; ++l) {
      layer(l, src, dst, j0, outb);
      __syncthreads();
      float *tsw = src; src = dst; dst = tsw;
    }
    const int tn = t + (int)gridDim.x < total_tiles ? t + (int)gridDim.x : t;
    const SaTile s = sa2_tile(a, tn, tiles_per_cloud, cpt, col);
    float xyz, v[kSaFly];
    sa2_gather_load(a, s, rq, xyz, v);
    layer(a.n_layers - 1, src, dst, j0, outb);
    __syncthreads();  // the last layer may have been reading region A
    sa2_gather_store(a, s, rq, col, xyz, v, A);
    __syncthreads();
  }
}


// ======================================================== fused set abstraction on split-f16 planes ==
// The same module core (gather + grouped MLP + max over the neighbours, ext/pvcnn/modules/pointnet.py:100-111) with the
// GEMMs on the bf16 matrix pipe: every f32 product as six bf16 partial products (hi / mid / lo splits of both operands,
// f32 accumulation: see the split-f16 core above), 6/16 of the f32-MFMA time.  A tile is 64 columns = 64 / U centres x U
// neighbours; the gathered tile and every hidden layer's output live in LDS as pre-split planes in B-fragment order
// (the position-major engine's geometry: 12 KiB per 32 channels), written once by their producer (the gather threads hold
// four consecutive channels of a column; a layer's epilogue its accumulators' four consecutive rows), so the k-loops are
// ds_read_b128 + buffer loads + MFMA (gemm1_pl).  Region A: the gathered tile, later the odd hidden layers' outputs;
// region B: the even ones'.  The last layer is never stored: max over a centre's neighbours on the accumulators.
// Persistent workgroups; the next tile's gather is requested in front of the last layer and stored behind it.
// Shapes: cin_pad a multiple of 32 (zero weights beyond the real rows), hidden widths multiples of 32 up to 256, U in
// {16, 32, 64}; anything else runs on the f32 kernels above.
constexpr int kSaBlockFloats = PG<4>::kBlockFloats;   // one 32-channel block of a 64-column tile's planes
constexpr int kSa3Quads = 9;   // row quads per gather thread: 8 threads per column x 9 x 4 rows >= 259 + padding
template <int NT, class FIRST>
__device__ __forceinline__ void sa3_gemm(const Ctx &c, const float *wp, int kb, int mt, int nt0, const float *planes,
                                         f32x4 (&acc)[1][NT], const FIRST &first) {
  switch (kb) {
    case 1: gemm1_pl<1, 1, NT, NoPre, 1, 4, FIRST>(c, wp, mt, nt0, planes, acc, NoPre(), first); break;
    case 2: gemm1_pl<2, 1, NT, NoPre, 1, 4, FIRST>(c, wp, mt, nt0, planes, acc, NoPre(), first); break;
    case 3: gemm1_pl<3, 1, NT, NoPre, 1, 4, FIRST>(c, wp, mt, nt0, planes, acc, NoPre(), first); break;
    case 4: gemm1_pl<4, 1, NT, NoPre, 1, 4, FIRST>(c, wp, mt, nt0, planes, acc, NoPre(), first); break;
    case 5: gemm1_pl<5, 1, NT, NoPre, 1, 4, FIRST>(c, wp, mt, nt0, planes, acc, NoPre(), first); break;
    case 6: gemm1_pl<6, 1, NT, NoPre, 1, 4, FIRST>(c, wp, mt, nt0, planes, acc, NoPre(), first); break;
    case 9: gemm1_pl<9, 1, NT, NoPre, 1, 4, FIRST>(c, wp, mt, nt0, planes, acc, NoPre(), first); break;   // 259 + 3 -> 288 rows (PVCNN2 SA4)
    default: gemm1_pl<8, 1, NT, NoPre, 1, 4, FIRST>(c, wp, mt, nt0, planes, acc, NoPre(), first); break;
  }
}
// The wave's first m-tile of layer l (the mapping of sa_mlp3_kernel) and the request for its block-0 fragments: issued
// right behind the previous layer's k-loop, in flight under its epilogue and the barrier (each layer of a tile used to
// start with a cold L2 round trip: ~1.9 k cycles against 2-5 k of MFMAs).
__device__ __forceinline__ int sa3_first_mt(const SaArgs &a, int l, int w) {
  const int mtiles = a.cout[l] >> 4;
  return (l + 1 < a.n_layers && mtiles < 8) ? (mtiles == 4 ? (w & 3) : (w & 1)) : w;
}
__device__ __forceinline__ Frag3 sa3_request(const Ctx &c, const SaArgs &a, int l) {
  const WStream wv(a.weights + a.w_off[l], c.lane);
  const int mt = sa3_first_mt(a, l, c.wave), kb = a.cin_pad[l] >> 5;
  const int mtc = mt < (a.cout[l] >> 4) ? mt : 0;   // waves beyond a narrow last layer: any valid fragment
  Frag3 f;
#pragma unroll
  for (int pl = 0; pl < kSplit; ++pl) f.p[pl] = wv.raw_at(mtc * kb * kFragBytes, pl * 1024);
  return f;
}
// REQ: request the next layer's first fragments right behind this k-loop (in flight under the epilogue and the barrier)
// bsc = 1 / (scale of the input planes): the accumulators run in the input's units; osc = that scale / the scale of the
// output planes (both 1 for ordinary data: range_pow2)
template <int NT, class FIRST, bool REQ>
__device__ __forceinline__ Frag3 sa3_hidden(const Ctx &c, const SaArgs &a, int l, int mt, int nt0, const float *src, float *dst,
                                            const FIRST &first, float bsc, float osc) {
  const int col = c.lane & 15, kq = c.lane >> 4;
  f32x4 acc[1][NT];
  const f32x4 bv = *reinterpret_cast<const f32x4 *>(a.weights + a.b_off[l] + 16 * mt + 4 * kq) * bsc;
#pragma unroll
  for (int ni = 0; ni < NT; ++ni) acc[0][ni] = bv;
  sa3_gemm<NT, FIRST>(c, a.weights + a.w_off[l], a.cin_pad[l] >> 5, mt, nt0, src, acc, first);
  Frag3 nxt{};
  if constexpr (REQ) nxt = sa3_request(c, a, l + 1);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int ni = 0; ni < NT; ++ni)
    store_planes4(dst, 16 * mt + 4 * kq, 16 * (nt0 + ni) + col, fmaxf(acc[0][ni][0], 0.f) * osc, fmaxf(acc[0][ni][1], 0.f) * osc,
                  fmaxf(acc[0][ni][2], 0.f) * osc, fmaxf(acc[0][ni][3], 0.f) * osc);
  return nxt;
}
// last layer: m-tile mt over all four n-tiles, max over each centre's U / 16 tiles and 16 columns, ReLU, one value per row
// STAGED (the single-tile kernels): the pooled values are not stored here.  A workgroup walks RUNS of consecutive tiles,
// i.e. 8 consecutive centres of a cloud; the pooled value of (row, centre s of the run) goes to stage[row][s] in LDS and,
// at the end of the run, all threads write the rows' 32-byte runs (sa3_flush) -- where the unstaged form wrote every (row,
// centre) as a 4-byte store of its own into its own 32-byte sector (268 MB of HBM writes per launch at SSG-SA2 for a
// 33.5 MB tensor).  LDS, not registers, carries the run: values kept in registers across tiles were spilled to scratch, and
// a scratch reload queues behind the next tile's gather in the in-order vmcnt (measured: 1.02 -> 1.26 ms).
constexpr int kSaRun = 8;   // centres per staged run
template <class FIRST, bool STAGED = false>
__device__ __forceinline__ void sa3_last(const Ctx &c, const SaArgs &a, int l, int mt, const float *src, int j0, float *outb,
                                         const FIRST &first, float bsc, float osc, float *stage = nullptr, int slot0 = 0) {
  const int col = c.lane & 15, kq = c.lane >> 4;
  f32x4 acc[1][4];
  const f32x4 bv = *reinterpret_cast<const f32x4 *>(a.weights + a.b_off[l] + 16 * mt + 4 * kq) * bsc;
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) acc[0][ni] = bv;
  sa3_gemm<4, FIRST>(c, a.weights + a.w_off[l], a.cin_pad[l] >> 5, mt, 0, src, acc, first);
  const int tpc = a.u >> 4;   // n-tiles per centre: 1, 2 or 4
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float m[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) m[ni] = acc[0][ni][r];
    if (tpc >= 2) { m[0] = fmaxf(m[0], m[1]); m[2] = fmaxf(m[2], m[3]); }
    if (tpc >= 4) m[0] = fmaxf(m[0], m[2]);
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      if (ni % tpc) continue;  // wave uniform
      const float v = fmaxf(row16_max(m[ni]), 0.f) * osc;
      const int jj = ni / tpc;
      if constexpr (STAGED) {
        if (col == 0 && j0 + jj < a.m) ((lds_f *)stage)[(16 * mt + 4 * kq + r) * kSaRun + slot0 + jj] = v;
      } else {
        if (col == 0 && j0 + jj < a.m) outb[(size_t)(16 * mt + 4 * kq + r) * a.m + j0 + jj] = v;
      }
    }
  }
}
// the staged run -> out[b][row][jbase .. jbase + count) for every row of the last layer: thread = (row, half of the run)
__device__ __forceinline__ void sa3_flush(const SaArgs &a, const float *stage, int tid, int b, int jbase, int count) {
  const int rows = a.cout[a.n_layers - 1];
  for (int i = tid; i < 2 * rows; i += 512) {
    const int row = i >> 1, h4 = 4 * (i & 1), left = count - h4;
    if (left <= 0) continue;
    const f32x4 v = *reinterpret_cast<const lds_f4 *>((const lds_f *)stage + row * kSaRun + h4);
    float *o = a.out + ((size_t)b * rows + row) * a.m + jbase + h4;
    if (left >= 4 && (((size_t)o & 15) == 0)) *reinterpret_cast<f32x4 *>(o) = v;
    else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (e < left) o[e] = v[e];
    }
  }
}

// One layer of the multi-tile form over all `sub` tiles of a workgroup pass: the m-tile's weight fragments (KB blocks x 3
// planes) are requested ONCE and stay in registers while the tiles' planes stream through -- per tile only LDS reads,
// MFMAs and the epilogue remain (through gemm1_pl every tile paid the fragments' round trip again: latency bound at 24-96
// MFMAs per call).  LAST: max over the neighbours instead of the plane stores.
template <int KB, int NT, bool LAST>
__device__ __forceinline__ void sa3_layer_multi(const Ctx &c, const SaArgs &a, int l, int mt, int nt0, int sub, int per,
                                                int src_off, int dst_off, int T, int tiles_per_cloud, int total_tiles,
                                                float bsc, float osc) {
  const int col = c.lane & 15, kq = c.lane >> 4, g = kq;
  const WStream wv(a.weights + a.w_off[l], c.lane);
  u32x4 af[KB][kSplit];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb)
#pragma unroll
    for (int pl = 0; pl < kSplit; ++pl) af[kb][pl] = wv.raw_at((mt * KB + kb) * kFragBytes, pl * 1024);
  const f32x4 bv = *reinterpret_cast<const f32x4 *>(a.weights + a.b_off[l] + 16 * mt + 4 * kq) * bsc;
  const int cpt = 64 / a.u, tpc = a.u >> 4;
  for (int st = 0; st < sub; ++st) {
    const lds_u4 *pl3 = (const lds_u4 *)(c.lds + st * per + src_off) + g * 64 + 16 * nt0 + col;
    f32x4 acc[NT];
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) acc[ni] = bv;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      u32x4 bs[NT][kSplit];
#pragma unroll
      for (int ni = 0; ni < NT; ++ni)
#pragma unroll
        for (int pl = 0; pl < kSplit; ++pl) bs[ni][pl] = pl3[(kb * kSplit + pl) * 256 + 16 * ni];
#pragma unroll
      for (int ni = 0; ni < NT; ++ni) acc[ni] = mfma_split(af[kb], bs[ni], acc[ni]);
    }
    if constexpr (!LAST) {
#pragma unroll
      for (int ni = 0; ni < NT; ++ni)
        store_planes4(c.lds + st * per + dst_off, 16 * mt + 4 * kq, 16 * (nt0 + ni) + col, fmaxf(acc[ni][0], 0.f) * osc,
                      fmaxf(acc[ni][1], 0.f) * osc, fmaxf(acc[ni][2], 0.f) * osc, fmaxf(acc[ni][3], 0.f) * osc);
    } else {
      static_assert(!LAST || NT == 4, "the max runs over all four n-tiles of a tile");
      const int t = T * sub + st;
      if (t < total_tiles) {   // wave uniform
        const int b = t / tiles_per_cloud, j0 = (t - b * tiles_per_cloud) * cpt;
        float *outb = a.out + (size_t)b * a.cout[l] * a.m;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float m[4];
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) m[ni] = acc[ni < NT ? ni : 0][r];
          if (tpc >= 2) { m[0] = fmaxf(m[0], m[1]); m[2] = fmaxf(m[2], m[3]); }
          if (tpc >= 4) m[0] = fmaxf(m[0], m[2]);
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) {
            if (ni % tpc) continue;  // wave uniform
            const float v = fmaxf(row16_max(m[ni]), 0.f) * osc;
            const int jj = ni / tpc;
            if (col == 0 && j0 + jj < a.m) outb[(size_t)(16 * mt + 4 * kq + r) * a.m + j0 + jj] = v;
          }
        }
      }
    }
  }
}
template <int NT, bool LAST>
__device__ __forceinline__ void sa3_layer_multi_kb(const Ctx &c, const SaArgs &a, int l, int mt, int nt0, int sub, int per,
                                                   int src_off, int dst_off, int T, int tiles_per_cloud, int total_tiles,
                                                   float bsc, float osc) {
  switch (a.cin_pad[l] >> 5) {
    case 1: sa3_layer_multi<1, NT, LAST>(c, a, l, mt, nt0, sub, per, src_off, dst_off, T, tiles_per_cloud, total_tiles, bsc, osc); break;
    case 2: sa3_layer_multi<2, NT, LAST>(c, a, l, mt, nt0, sub, per, src_off, dst_off, T, tiles_per_cloud, total_tiles, bsc, osc); break;
    default: sa3_layer_multi<4, NT, LAST>(c, a, l, mt, nt0, sub, per, src_off, dst_off, T, tiles_per_cloud, total_tiles, bsc, osc); break;
  }
}

// SUBMAX > 1: narrow nets (SSG SA1: 3 -> 64 -> 64 -> 128 over 512 centres per cloud) have 1.3 k cycles of MFMAs per
// 64-column tile against ~13 k of per-tile cost (four barriers, three cold layer starts, the gather's round trip): a
// workgroup then takes `sub` consecutive tiles at once -- their planes side by side in LDS, every layer swept over all of
// them between two barriers, the weight fragments of the later ones coming from L1.  QUADS: row quads a gather thread
// holds per tile (9 covers 288 input rows; the multi-tile form takes 32-row inputs: one quad).
// PRE: the module's first layer is not a GEMM here.  W1 [x - c; f] = W1a (x - c) + W1b f, and W1b f + b1 depends on the
// POINT only: the caller computes it once per cloud (a.pre), the gather threads fetch a neighbour's 4 rows of it instead of
// 4 feature rows, add the three coordinate products and apply the ReLU -- the tile that goes into LDS is the first
// layer's OUTPUT.  One k-loop, one plane-writing epilogue and one barrier per tile less, 29 % fewer MFMAs at SSG-SA2.
// (The second launch bound is waves per SIMD: 4 = two co-resident workgroups, i.e. 128 registers.  The hoisted form with
// up to four row quads per gather thread fits them; the general one holds nine quads and runs one workgroup per CU.)
template <int SUBMAX, int QUADS, bool PRE = false>
__global__ __launch_bounds__(512, (PRE && QUADS <= 4) ? 4 : 2) void sa_mlp3_kernel(const SaArgs a, int blocks_a, int blocks_b, int sub,
                                                         int tiles_per_cloud, int total_tiles) {
  extern __shared__ float lds[];
  Ctx c{a.weights, lds, (int)threadIdx.x, __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), (int)threadIdx.x & 63,
        0, 4};
  const int cpt = 64 / a.u;  // centres per tile
  const int per = (blocks_a + blocks_b) * kSaBlockFloats;   // floats of one tile's two plane regions
  const int w = c.wave, col = c.lane, qg = c.wave;   // gather: thread = (column, row-quad group)
  const int nquads = a.cin_pad[0] >> 2;
  const int supers = (total_tiles + sub - 1) / sub;
  float gv[SUBMAX][QUADS][4];
  float dxyz[SUBMAX][PRE ? 4 : 1];   // PRE: the column's x - c (3) and its live mask
  // The neighbour index of a column is requested a tile ahead of the gather that dereferences it (idx_request at the end
  // of the previous gather_load): read where it is used, every tile began with an exposed round trip for the index in
  // front of the gather's own.
  int id_pf[SUBMAX];
  auto idx_request = [&](int T) {
#pragma unroll
    for (int st = 0; st < SUBMAX; ++st) {
      const int t0 = T * sub + (st < sub ? st : 0);
      const int t = t0 < total_tiles ? t0 : total_tiles - 1;
      const int b = t / tiles_per_cloud, tile = t - b * tiles_per_cloud, j0 = tile * cpt, jj = col / a.u;
      const bool live = j0 + jj < a.m;
      id_pf[st] = (a.idx + ((size_t)b * a.m + j0) * a.u)[live ? col : 0];
    }
  };
  auto gather_load = [&](int T) {   // idx_request(T) went before
#pragma unroll
    for (int st = 0; st < SUBMAX; ++st) {
      const int t0 = T * sub + (st < sub ? st : 0);
      const int t = t0 < total_tiles ? t0 : total_tiles - 1;   // clamped: every load stays unconditional
      const int b = t / tiles_per_cloud, tile = t - b * tiles_per_cloud, j0 = tile * cpt, jj = col / a.u;
      const bool live = j0 + jj < a.m;
      const float *pts = a.points + (size_t)b * 3 * a.n, *ctr = a.centers + (size_t)b * 3 * a.m;
      const float *feat = a.feat ? a.feat + (size_t)b * a.c * a.n : a.points;
      const int id = live ? id_pf[st] : 0;
      const int jc = live ? j0 + jj : 0, cmax = a.c > 0 ? a.c - 1 : 0;
      if constexpr (PRE) {
        // pre is POINT-major, [b][n][c1]: a neighbour's rows are one run of c1 floats, a thread's row quad one 16-byte load
        // (channel-major, the 4-byte gathers of a tile were 8192 cache-line requests: the texture addresser, not the
        // matrix pipe, bounded the kernel -- a layer less changed nothing)
        const f32x4 *prow = reinterpret_cast<const f32x4 *>(a.pre + (a.pre_bcast ? (size_t)0 : ((size_t)b * a.n + id) * a.c1));
#pragma unroll
        for (int e = 0; e < 3; ++e) dxyz[st][e] = pts[e * a.n + id] - ctr[e * a.m + jc];
        dxyz[st][3] = live ? 1.0f : 0.0f;
        const int qmax = (a.c1 >> 2) - 1;
#pragma unroll
        for (int i = 0; i < QUADS; ++i) {
          const int rq = qg + 8 * i;
          const f32x4 v4 = prow[rq < qmax ? rq : qmax];
#pragma unroll
          for (int e = 0; e < 4; ++e) gv[st][i][e] = v4[e];
        }
        continue;
      }
#pragma unroll
      for (int i = 0; i < QUADS; ++i) {
        const int rq = qg + 8 * i;   // rows 4 rq .. 4 rq + 3: [x y z f0] for quad 0, f[4 rq - 3 ..] after it
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int ch = 4 * rq + e;
          float v;
          if (ch < 3) v = pts[ch * a.n + id] - ctr[ch * a.m + jc];
          else { const int f = ch - 3; v = feat[(size_t)(f < cmax ? f : cmax) * a.n + id]; v = f < a.c ? v : 0.f; }
          gv[st][i][e] = live ? v : 0.f;
        }
      }
    }
  };
  // PRE: gv holds the neighbours' rows of W1b f + b1; add W1a (x - c), ReLU -> the first layer's output (dead columns and
  // rows beyond c1: zero).  W1a's rows are wave uniform (a wave's threads share their row quads).
  auto gather_finish = [&]() {
    if constexpr (PRE) {
      const f32x4 *wa = reinterpret_cast<const f32x4 *>(a.weights + a.wa_off);
#pragma unroll
      for (int i = 0; i < QUADS; ++i) {
        const int rq = qg + 8 * i;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int row = 4 * rq + e;
          const f32x4 w4 = wa[row < a.c1 ? row : 0];
#pragma unroll
          for (int st = 0; st < SUBMAX; ++st) {
            float h = gv[st][i][e];
            h = __builtin_fmaf(w4[0], dxyz[st][0], h);
            h = __builtin_fmaf(w4[1], dxyz[st][1], h);
            h = __builtin_fmaf(w4[2], dxyz[st][2], h);
            gv[st][i][e] = row < a.c1 ? fmaxf(h, 0.f) * dxyz[st][3] : 0.f;
          }
        }
      }
    }
  };
  // Range scale of the gathered tile(s) (range_pow2): every wave publishes the largest magnitude it holds in front of a
  // barrier the tile needs anyway, all read the eight words behind it.  m_in / s_in: of the tile(s) about to be stored.
  float *rng = lds + (size_t)sub * per;   // [8], behind the planes (the launcher adds the room)
  float m_in = 0.f, s_in = 1.f;
  auto range_publish = [&]() {
    float mx = 0.f;
#pragma unroll
    for (int st = 0; st < SUBMAX; ++st)
#pragma unroll
      for (int i = 0; i < QUADS; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) mx = fmaxf(mx, fabsf(gv[st][i][e]));
    mx = half_max(row_pair_max(row16_max(mx)));
    if (c.lane == 0) rng[c.wave] = mx;
  };
  auto range_read = [&]() {
    const f32x4 r0 = *reinterpret_cast<const f32x4 *>(rng), r1 = *reinterpret_cast<const f32x4 *>(rng + 4);
    const float mx = fmaxf(fmaxf(fmaxf(r0[0], r0[1]), fmaxf(r0[2], r0[3])), fmaxf(fmaxf(r1[0], r1[1]), fmaxf(r1[2], r1[3])));
    m_in = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(mx)));
    s_in = range_pow2(m_in);
  };
  auto gather_store = [&]() {
    const float inv = pow2_inv(s_in);
#pragma unroll
    for (int st = 0; st < SUBMAX; ++st) {
      if (st < sub) {
#pragma unroll
        for (int i = 0; i < QUADS; ++i) {
          const int rq = qg + 8 * i;
          if (rq < nquads)
            store_planes4(lds + st * per, 4 * rq, col, gv[st][i][0] * inv, gv[st][i][1] * inv, gv[st][i][2] * inv, gv[st][i][3] * inv);
        }
      }
    }
  };
  Frag3 frag;   // block-0 fragments of the wave's first m-tile of the NEXT layer to run
  // The request for the next layer's first fragments sits behind this layer's (first) k-loop, in front of its stores: in
  // flight under the epilogue and the barrier.  Straight-line code: the fragments travel by value.
  auto hidden = [&](int l, bool a_to_b, float bsc, float osc) {
    const int mtiles = a.cout[l] >> 4;
    Ctx cl = c;
    asm volatile("" : "+v"(cl.tid), "+v"(cl.lane));
    if constexpr (SUBMAX > 1) {   // weights once per layer, the tiles stream through (sa3_layer_multi)
      const int so = a_to_b ? 0 : blocks_a * kSaBlockFloats, dof = a_to_b ? blocks_a * kSaBlockFloats : 0;
      if (mtiles >= 8) {
        for (int p = 0; p < (mtiles >> 3); ++p) sa3_layer_multi_kb<4, false>(cl, a, l, w + 8 * p, 0, sub, per, so, dof, 0, 1, 0, bsc, osc);
      } else if (mtiles == 4) sa3_layer_multi_kb<2, false>(cl, a, l, w & 3, 2 * (w >> 2), sub, per, so, dof, 0, 1, 0, bsc, osc);
      else sa3_layer_multi_kb<1, false>(cl, a, l, w & 1, w >> 1, sub, per, so, dof, 0, 1, 0, bsc, osc);
      return;
    }
    const Frag3 cur = frag;
    for (int st = 0; st < sub; ++st) {
      const float *src = lds + st * per + (a_to_b ? 0 : blocks_a * kSaBlockFloats);
      float *dst = lds + st * per + (a_to_b ? blocks_a * kSaBlockFloats : 0);
      if (st == 0) {
        if (mtiles == 8) frag = sa3_hidden<4, Frag3, true>(cl, a, l, w, 0, src, dst, cur, bsc, osc);
        else if (mtiles == 16) {
          sa3_hidden<4, Frag3, false>(cl, a, l, w, 0, src, dst, cur, bsc, osc);
          frag = sa3_hidden<4, NoFirst, true>(cl, a, l, w + 8, 0, src, dst, NoFirst(), bsc, osc);
        } else if (mtiles == 4) frag = sa3_hidden<2, Frag3, true>(cl, a, l, w & 3, 2 * (w >> 2), src, dst, cur, bsc, osc);
        else frag = sa3_hidden<1, Frag3, true>(cl, a, l, w & 1, w >> 1, src, dst, cur, bsc, osc);
      } else if constexpr (SUBMAX > 1) {
        if (mtiles == 8) sa3_hidden<4, NoFirst, false>(cl, a, l, w, 0, src, dst, NoFirst(), bsc, osc);
        else if (mtiles == 16) {
          sa3_hidden<4, NoFirst, false>(cl, a, l, w, 0, src, dst, NoFirst(), bsc, osc);
          sa3_hidden<4, NoFirst, false>(cl, a, l, w + 8, 0, src, dst, NoFirst(), bsc, osc);
        } else if (mtiles == 4) sa3_hidden<2, NoFirst, false>(cl, a, l, w & 3, 2 * (w >> 2), src, dst, NoFirst(), bsc, osc);
        else sa3_hidden<1, NoFirst, false>(cl, a, l, w & 1, w >> 1, src, dst, NoFirst(), bsc, osc);
      }
    }
  };
  // Tile order.  SUBMAX > 1: workgroup w takes super-tiles w, w + grid, ...  SUBMAX == 1: RUNS of consecutive tiles = 8
  // consecutive centres of a cloud, so that the pooled rows leave as 32-byte runs (SaOutStage); the runs are dealt
  // w, w + grid, ... in an order that gives the workgroups of one XCD (blockIdx % 8: the dispatcher's round robin)
  // neighbouring runs: at any time the chip works on ~32 clouds and an XCD on four of them, whose features stay in its L2.
  // (Measured and dropped: one contiguous range of 64 tiles per workgroup, 16-centre runs -- every workgroup on a cloud of
  // its own, 256 clouds live at once: 1.02 -> 1.31 ms at SSG-SA2, the gathers miss L2.)
  constexpr bool kRuns = SUBMAX == 1;
  const int grid = gridDim.x;
  const int run = !kRuns ? 1 : (supers < 8 * grid ? 1 : (cpt >= kSaRun ? 1 : kSaRun / cpt));   // small launches: a tile per workgroup at a time
  const int w8 = !kRuns ? (int)blockIdx.x
                        : ((grid & 7) == 0 ? ((int)blockIdx.x & 7) * (grid >> 3) + ((int)blockIdx.x >> 3) : (int)blockIdx.x);
  auto tile_of = [&](int q) { return (w8 + (q / run) * grid) * run + q % run; };   // this workgroup's q-th (super-)tile
  int q = 0, T = tile_of(0);
  if (T >= supers) return;   // (whole workgroup: no barrier has been reached yet)
  float *stage = rng + 16;   // [rows of the last layer][kSaRun] (kRuns; the launcher adds the room)
  int st_count = 0, st_b = 0, st_jbase = 0;
  idx_request(T);
  gather_load(T);
  {
    const int T1 = tile_of(1);
    idx_request(T1 < supers ? T1 : T);
  }
  frag = sa3_request(c, a, 0);
  gather_finish();
  if (a.ranged) {
    range_publish();
    __syncthreads();
    range_read();
  }
  gather_store();
  __syncthreads();
  for (; T < supers; T = tile_of(++q)) {
    bool a_to_b = true;
    // scales of this tile's planes, layer by layer: the input's is measured, a hidden layer's follows from the bound
    // |out| <= gain_r * max |in| + gain_b (true units)
    float bnd = m_in, s_cur = s_in;
    for (int l = 0; l + 1 < a.n_layers; ++l) {
      bnd = a.gain_r[l] * bnd + a.gain_b[l];
      const float s_nxt = a.ranged ? range_pow2(bnd) : 1.0f;
      hidden(l, a_to_b, pow2_inv(s_cur), s_cur * pow2_inv(s_nxt));
      s_cur = s_nxt;
      __syncthreads();
      a_to_b = !a_to_b;
    }
    const int Tq = tile_of(q + 1), Tn = Tq < supers ? Tq : T;
    gather_load(Tn);
    {
      const int T2 = tile_of(q + 2);
      idx_request(T2 < supers ? T2 : Tn);   // for the gather of the tile after next
    }
    {
      const int l = a.n_layers - 1, mtiles = a.cout[l] >> 4;
      const float bsc = pow2_inv(s_cur), osc = s_cur;
      Ctx cl = c;
      asm volatile("" : "+v"(cl.tid), "+v"(cl.lane));
      if constexpr (SUBMAX > 1) {
        const int so = a_to_b ? 0 : blocks_a * kSaBlockFloats;
        for (int mt = w; mt < mtiles; mt += 8)
          sa3_layer_multi_kb<4, true>(cl, a, l, mt, 0, sub, per, so, 0, T, tiles_per_cloud, total_tiles, bsc, osc);
      }
      const Frag3 cur = frag;
      for (int st = 0; st < (SUBMAX > 1 ? 0 : sub); ++st) {
        const int t = T * sub + st;
        if (t >= total_tiles) break;   // wave uniform
        const int b = t / tiles_per_cloud, j0 = (t - b * tiles_per_cloud) * cpt;
        float *outb = a.out + (size_t)b * a.cout[l] * a.m;
        const float *src = lds + st * per + (a_to_b ? 0 : blocks_a * kSaBlockFloats);
        if constexpr (kRuns) {
          const int live = a.m - j0 < cpt ? a.m - j0 : cpt;          // centres of this tile that exist
          if (st_count == 0) { st_b = b; st_jbase = j0; }
          if (w < mtiles) sa3_last<Frag3, true>(cl, a, l, w, src, j0, outb, cur, bsc, osc, stage, st_count);
          for (int mt = w + 8; mt < mtiles; mt += 8) sa3_last<NoFirst, true>(cl, a, l, mt, src, j0, outb, NoFirst(), bsc, osc, stage, st_count);
          st_count += live;
        } else {
        if (st == 0) {
          if (w < mtiles) sa3_last<Frag3>(cl, a, l, w, src, j0, outb, cur, bsc, osc);
        } else {
          if (w < mtiles) sa3_last<NoFirst>(cl, a, l, w, src, j0, outb, NoFirst(), bsc, osc);
        }
        for (int mt = w + 8; mt < mtiles; mt += 8) sa3_last<NoFirst>(cl, a, l, mt, src, j0, outb, NoFirst(), bsc, osc);
        }
      }
      frag = sa3_request(c, a, 0);   // the next tile's first layer
    }
    gather_finish();                 // the next tile's gathered values have long landed
    if (a.ranged) range_publish();
    __syncthreads();  // the last layer may have been reading region A
    if constexpr (kRuns) {
      // the run ends here unless the next tile continues it (same cloud, the next centres, room in the stage)
      bool more = Tq < supers;
      if (more) {
        const int bn = Tq / tiles_per_cloud, jn = (Tq - bn * tiles_per_cloud) * cpt;
        const int liven = a.m - jn < cpt ? a.m - jn : cpt;
        more = bn == st_b && jn == st_jbase + st_count && st_count + liven <= kSaRun;
      }
      if (!more) {
        sa3_flush(a, stage, c.tid, st_b, st_jbase, st_count);
        st_count = 0;
      }
    }
    if (a.ranged) range_read();
    gather_store();
    __syncthreads();
  }
}


// Scale / shift rows of every ResnetBlock for one conditioning cloud (ResnetBlock.mlp, resnets.py:125-151, when the
// embedding has no time part: the pose decoder):  ss[rb][row] = comb_b[row] + sum_e W[row][e] G[e],  G = sum over the
// cond rows of SiLU(cemb) -- the value the conv epilogue would compute with MFMAs for every sample and column.
// Table layout per cloud: blocks in tape order, [scale rows (C) | shift rows (C)] each (build_tape's tab_off).
__global__ __launch_bounds__(256) void ss_table_kernel(const gldm_r1d_desc d, const float *__restrict__ w,
                                                       const float *__restrict__ cemb, int stride,
                                                       float *__restrict__ tab) {
  __shared__ float G[256];
  const int cond = blockIdx.x, E = d.emb_dim, R = d.cond_rows, ekb = E >> 4;
  for (int e = threadIdx.x; e < E; e += blockDim.x) {
    float g = 0.f;
    for (int r = 0; r < R; ++r) g += silu(cemb[((size_t)cond * R + r) * E + e]);
    G[e] = g;
  }
  __syncthreads();
  float *out = tab + (size_t)cond * stride;
  int off = 0;
  const int n_rb = 2 * d.n_levels + 1;
  for (int i = 0; i < n_rb; ++i) {
    const int C = d.dims[i < 2 * d.n_levels ? i / 2 : d.n_levels];
    const gldm_r1d_resblock &rb = d.rb[i];
    for (int row = threadIdx.x; row < 2 * C; row += blockDim.x) {
      // packed A fragments of the [2C x E] Linear: W[row][e] at ((mt * ekb + kb) * 64 + 16 kq + i) * 4 + j,
      // row = 16 mt + i, e = 16 kb + 4 j + kq
      const float *wr = w + rb.ss_w + (size_t)(row >> 4) * ekb * 256 + (row & 15) * 4;
      float acc = w[rb.ss_b + row];
      for (int kb = 0; kb < ekb; ++kb)
        for (int j = 0; j < 4; ++j)
          for (int kq = 0; kq < 4; ++kq) acc = fmaf(wr[kb * 256 + kq * 64 + j], G[16 * kb + 4 * j + kq], acc);
      out[off + row] = acc;
    }
    off += 2 * C;
  }
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
  // header + one 8-byte hand-off granule per activation column of every tile (see ChainHdr) + table + park scratch
  return ws_layout(desc, n_samples).total;
}

GLDM_API int gldm_r1d_tile_columns(const gldm_r1d_desc *desc) {
  const int st = validate(desc);
  if (st != GLDM_OK) return st;
  return wide_engine(desc) ? 64 : engine_nc();
}

GLDM_API int gldm_denoise(const gldm_r1d_desc *desc, const float *weights, const float *temb, const float *cemb,
                          int samples_per_cond, const float *x_in, int n_samples, const int32_t *timesteps,
                          const int32_t *sample_t, int n_steps, int sched_kind, int clip_sample,
                          const float *sched_coef, const float *step_noise, const float *sample_emb, float *x_out,
                          void *workspace, gldm_stream_t stream) {
  int st = validate(desc);
  if (st != GLDM_OK) return st;
  if (!weights || !cemb || !x_in || !x_out || !workspace || n_samples <= 0 || n_steps <= 0 || samples_per_cond <= 0)
    return GLDM_ERR_INVALID_ARG;
  if (desc->latent_dim != 0) return GLDM_ERR_INVALID_ARG;
  if (temb && !timesteps && !sample_t) return GLDM_ERR_INVALID_ARG;
  if (sched_kind != GLDM_SCHED_NONE && (!sched_coef || ((unsigned long long)sched_coef & 15))) return GLDM_ERR_INVALID_ARG;  // rows are read as 16-byte loads
  if (sched_kind == GLDM_SCHED_NONE && n_steps != 1) return GLDM_ERR_INVALID_ARG;
  RunArgs a{};
  a.d = *desc;
  a.weights = weights; a.temb = temb; a.cemb = cemb; a.semb = sample_emb; a.samples_per_cond = samples_per_cond;
  a.x_in = x_in; a.n_samples = n_samples; a.timesteps = timesteps; a.sample_t = sample_t; a.n_steps = n_steps;
  a.sched_kind = sched_kind; a.clip_sample = clip_sample; a.sched_coef = sched_coef; a.step_noise = step_noise;
  a.out0 = x_out; a.out1 = nullptr; a.ws = reinterpret_cast<float *>(workspace);
  return launch_r1d(a, reinterpret_cast<hipStream_t>(stream));
}

GLDM_API int gldm_denoise_rng(const gldm_r1d_desc *desc, const float *weights, const float *temb, const float *cemb,
                              int samples_per_cond, const float *x_in, int n_samples, const int32_t *timesteps, int n_steps,
                              int clip_sample, const float *sched_coef, unsigned long long noise_seed,
                              long long noise_base, const float *sample_emb, float *x_out, void *workspace,
                              gldm_stream_t stream) {
  int st = validate(desc);
  if (st != GLDM_OK) return st;
  if (!weights || !cemb || !x_in || !x_out || !workspace || n_samples <= 0 || n_steps <= 0 || samples_per_cond <= 0 ||
      noise_base < 0)
    return GLDM_ERR_INVALID_ARG;
  if (desc->latent_dim != 0 || (temb && !timesteps)) return GLDM_ERR_INVALID_ARG;
  if (!sched_coef || ((unsigned long long)sched_coef & 15)) return GLDM_ERR_INVALID_ARG;
  RunArgs a{};
  a.d = *desc;
  a.weights = weights; a.temb = temb; a.cemb = cemb; a.semb = sample_emb; a.samples_per_cond = samples_per_cond;
  a.x_in = x_in; a.n_samples = n_samples; a.timesteps = timesteps; a.sample_t = nullptr; a.n_steps = n_steps;
  a.sched_kind = GLDM_SCHED_DDPM; a.clip_sample = clip_sample; a.sched_coef = sched_coef; a.step_noise = nullptr;
  a.noise_seed = noise_seed; a.noise_base = noise_base; a.noise_on = 1;
  a.out0 = x_out; a.out1 = nullptr; a.ws = reinterpret_cast<float *>(workspace);
  return launch_r1d(a, reinterpret_cast<hipStream_t>(stream));
}

// The generator on its own (n latents x L positions of one step), for the statistical tests: out [n][L]
__global__ void philox_normal_kernel(unsigned long long seed, long long base, int step, int n, int L, float *out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * L) return;
  const int gi = i / L, l = i - gi * L;
  const unsigned long long g = (unsigned long long)(base + gi);
  float z4[4];
  philox_normal4(seed, (unsigned)g, (unsigned)(g >> 32), (unsigned)(l >> 2), (unsigned)step, z4);
  out[i] = z4[l & 3];
}
GLDM_API int gldm_step_noise_rng(unsigned long long noise_seed, long long noise_base, int step, int n_samples, int seq_len,
                                 float *out, gldm_stream_t stream) {
  if (!out || n_samples <= 0 || seq_len <= 0 || step < 0 || noise_base < 0) return GLDM_ERR_INVALID_ARG;
  const int total = n_samples * seq_len;
  hipLaunchKernelGGL(philox_normal_kernel, dim3((total + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), noise_seed,
                     noise_base, step, n_samples, seq_len, out);
  return hipGetLastError() == hipSuccess ? GLDM_OK : GLDM_ERR_LAUNCH;
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
  const int rows = ss_table_rows(desc);
  if (rows > 0) {  // the table lives behind the hand-off granules of the workspace (gldm_r1d_workspace_bytes)
    const long long off = ws_layout(desc, n_samples).ss_off;
    float *tab = reinterpret_cast<float *>(reinterpret_cast<char *>(workspace) + off);
    const int n_cond = (n_samples + samples_per_cond - 1) / samples_per_cond;
    hipLaunchKernelGGL(ss_table_kernel, dim3(n_cond), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), *desc, weights,
                       cemb, rows, tab);
    if (hipGetLastError() != hipSuccess) return GLDM_ERR_LAUNCH;
    a.ss_tab = tab;
    a.ss_stride = rows;
  }
  return launch_r1d(a, reinterpret_cast<hipStream_t>(stream));
}

GLDM_API int gldm_pose_epilogue(const float *tmrp, const float *logit, const float *grasp_mean,
                                const float *grasp_std, int n, int grasps_per_cloud, int n_clouds, float *H,
                                float *tmrp_unnorm, float *confidence, gldm_stream_t stream) {
  if (!tmrp || !grasp_mean || !grasp_std || !H || !tmrp_unnorm || n <= 0 || grasps_per_cloud <= 0 || n_clouds <= 0)
    return GLDM_ERR_INVALID_ARG;
  // every grasp's cloud row (i / grasps_per_cloud) must exist in mean/std [n_clouds, 6]
  if ((long long)n_clouds * grasps_per_cloud < (long long)n) return GLDM_ERR_INVALID_ARG;
  if (confidence && !logit) return GLDM_ERR_INVALID_ARG;
  hipLaunchKernelGGL(pose_epilogue_kernel, dim3((n + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     tmrp, logit, grasp_mean, grasp_std, n, grasps_per_cloud, H, tmrp_unnorm, confidence);
  return hipGetLastError() == hipSuccess ? GLDM_OK : GLDM_ERR_LAUNCH;
}


namespace {
int launch_pointwise(const float *x, const float *w0, const float *b0, int cin0, const float *w, const float *bias, int b,
                     int cin_arg, int cout, int n, int relu, const float *head_w, const float *head_b, int hout, float *y,
                     float *z, hipStream_t stream, bool split_f16 = false, const float *add = nullptr, long long add_bs = 0,
                     long long add_rs = 0, long long add_cs = 0, const float *front_gain = nullptr, bool y_point_major = false) {
  if (y_point_major && (!split_f16 || !y)) return GLDM_ERR_UNSUPPORTED;
  if (add && !split_f16) return GLDM_ERR_UNSUPPORTED;
  // The split launch without a front layer takes any multiple of 8 input rows: K is padded to whole 128-deep trips of its
  // weight ring (the caller's fragments carry zero columns there: r1d_pack.mfma_a_fragments_f16x2 of the padded matrix)
  const int cin_rows = cin_arg;
  const int cin = (split_f16 && !w0 && cin_arg > 0 && (cin_arg & 7) == 0) ? (cin_arg + 127) & ~127 : cin_arg;
  if (!x || !w || !bias || b <= 0 || cin <= 0 || cout <= 0 || n <= 0) return GLDM_ERR_INVALID_ARG;
  if (!y && !head_w) return GLDM_ERR_INVALID_ARG;
  if (head_w && (!z || hout <= 0 || hout > 16)) return GLDM_ERR_INVALID_ARG;
  // k-blocks in pairs, 32-point tiles; output rows: 2 m-tiles x 8 waves per round on the f32 kernel, units of two m-tiles on
  // the split one (fewer than eight units -- 64 .. 224 output rows -- leave waves without a unit idle)
  // the split launch hands out its output rows in units of two m-tiles, or of one where two would leave waves idle (fewer
  // than 256 rows, no front layer / head): then any multiple of 16 rows
  // (narrow inputs only: 32-point planes within half a CU's LDS, i.e. the 32-point tile form stays)
  const int mu = (split_f16 && !w0 && !head_w && cout < 256 && ((size_t)cin * 32 * 2 * kSplit + 64) * 2 <= (size_t)160 * 1024) ? 1 : 2;
  if ((cin & 31) || (split_f16 ? (n & 15) : (n & 31)) || (split_f16 ? (cout & (16 * mu - 1)) : (cout & 255))) return GLDM_ERR_UNSUPPORTED;
  if ((w0 || head_w) && (cout & 255)) return GLDM_ERR_UNSUPPORTED;   // front layer / head: whole rounds of units only
  if (w0 && (!b0 || cin0 <= 0 || (cin0 & 31) || (cin & 255))) return GLDM_ERR_UNSUPPORTED;
  size_t lds_bytes = ((size_t)cin * 32 + 8 * 16 * 32 + (w0 ? (size_t)cin0 * 32 : 0)) * sizeof(float);
  if (split_f16) {  // `w` and `w0` hold split-f16 fragments: planes of the tile + the front layer's f32 tile
    if (cin & 127) return GLDM_ERR_UNSUPPORTED;  // the A ring walks four 32-deep blocks per trip
    if (w0 && cin0 > 96) return GLDM_ERR_UNSUPPORTED;  // the front layer keeps its whole split tile in registers (72)
  }
  int dyn_first = 0, ticket_off = 0, nt = 2, x0_in_planes = 0;
  if (split_f16) {
    // LDS plan: planes | head products of the drawn units (the front layer's f32 tile lies under them: dead by then) |
    // ticket.  As many units are drawn as have room for their head slot (all but the first round when there is no head).
    // Tile width: 48 points where 32-point planes already take more than half a CU's LDS (one workgroup per CU either way)
    // and the 48-point plan fits; the front tile then goes UNDER the planes (x0_in_planes).
    const size_t cap = (size_t)160 * 1024 - 64;   // ticket + range words
    const int units = cout / (16 * mu);
    auto plan = [&](int ncol, bool x0_under, size_t &planes, size_t &region, int &first) {
      planes = (size_t)cin * ncol * 2 * kSplit;   // bytes: cin x ncol x kSplit f16
      region = (w0 && !x0_under) ? (size_t)cin0 * ncol * sizeof(float) : 0;
      if (planes + region > cap) return false;
      if (x0_under && (size_t)cin0 * ncol * sizeof(float) > planes) return false;
      int drawn = units > 8 ? units - 8 : 0;
      if (head_w) {
        const size_t slot = (size_t)hout * ncol * sizeof(float);
        const int room = (int)((cap - planes) / slot);
        if (drawn > room) drawn = room;
        if (planes < (size_t)8 * 16 * ncol * sizeof(float)) drawn = 0;   // z partials need the planes' space
      }
      first = (units - drawn + 7) & ~7;   // whole rounds are dealt
      if (head_w && (size_t)(units - first) * hout * ncol * sizeof(float) > region)
        region = (size_t)(units - first) * hout * ncol * sizeof(float);
      return planes + region <= cap;
    };
    size_t planes = 0, region = 0;
    if (!plan(32, false, planes, region, dyn_first)) return GLDM_ERR_UNSUPPORTED;
    if ((planes + region + 64) * 2 > (size_t)160 * 1024 && n >= 48) {
      size_t p3 = 0, r3 = 0;
      int f3 = 0;
      if (plan(48, w0 != nullptr, p3, r3, f3)) {
        nt = 3; planes = p3; region = r3; dyn_first = f3; x0_in_planes = w0 ? 1 : 0;
      }
    }
    ticket_off = (int)((planes + region) / sizeof(float));
    lds_bytes = planes + region + 64;   // ticket (16 B) + the eight range words
  }
  if (lds_bytes > 160 * 1024) return GLDM_ERR_UNSUPPORTED;
  struct PwTag { int site; };
  struct PwBfTag { int site; };
  struct PwBfAddTag { int site; };
  struct PwBf3Tag { int site; };
  struct PwBfAdd3Tag { int site; };
  struct PwBf1Tag { int site; };
  struct PwBfAdd1Tag { int site; };
  if (split_f16 && mu == 1 && nt == 2) {
    if (add) gldm_dev::allow_dynamic_lds<PwBfAdd1Tag>(reinterpret_cast<const void *>(&pointwise_mlp_sp_kernel<true, 2, 1>), 160 * 1024);
    else gldm_dev::allow_dynamic_lds<PwBf1Tag>(reinterpret_cast<const void *>(&pointwise_mlp_sp_kernel<false, 2, 1>), 160 * 1024);
  } else if (split_f16 && add && nt == 3) gldm_dev::allow_dynamic_lds<PwBfAdd3Tag>(reinterpret_cast<const void *>(&pointwise_mlp_sp_kernel<true, 3>), 160 * 1024);
  else if (split_f16 && nt == 3) gldm_dev::allow_dynamic_lds<PwBf3Tag>(reinterpret_cast<const void *>(&pointwise_mlp_sp_kernel<false, 3>), 160 * 1024);
  else if (split_f16 && add) gldm_dev::allow_dynamic_lds<PwBfAddTag>(reinterpret_cast<const void *>(&pointwise_mlp_sp_kernel<true, 2>), 160 * 1024);
  else if (split_f16) gldm_dev::allow_dynamic_lds<PwBfTag>(reinterpret_cast<const void *>(&pointwise_mlp_sp_kernel<false, 2>), 160 * 1024);
  else gldm_dev::allow_dynamic_lds<PwTag>(reinterpret_cast<const void *>(&pointwise_mlp_kernel), 160 * 1024);
  PwArgs a{};
  a.x = x; a.w = w; a.bias = bias; a.head_w = head_w; a.head_b = head_b; a.y = y; a.z = z;
  a.cin = cin; a.cout = cout; a.n = n; a.relu = relu; a.hout = hout;
  a.w0 = w0; a.bias0 = b0; a.cin0 = cin0;
  a.dyn_first = dyn_first; a.ticket_off = ticket_off; a.x0_in_planes = x0_in_planes;
  a.add = add; a.add_bs = add_bs; a.add_rs = add_rs; a.add_cs = add_cs;
  // range scales: a lone layer measures its input tile; with a layer in front the caller's gain bounds its output
  a.y_point_major = y_point_major ? 1 : 0;
  a.cin_rows = cin_rows;
  a.rng_off = ticket_off + 4;
  a.ranged = split_f16 && (!w0 || front_gain);
  if (w0 && front_gain) {
    a.gain0_r = front_gain[0]; a.gain0_b = front_gain[1];
    if (!(a.gain0_r >= 0.f) || !(a.gain0_b >= 0.f)) return GLDM_ERR_INVALID_ARG;
  }
  a.tiles_per_cloud = (n + 16 * nt - 1) / (16 * nt);
  a.total_tiles = b * a.tiles_per_cloud;
  const int per_cu = lds_bytes * 2 <= 160 * 1024 ? 2 : 1;
  int grid = cu_count() * per_cu;
  if (grid > a.total_tiles) grid = a.total_tiles;
  if (split_f16 && mu == 1 && nt == 2) {
    if (add) hipLaunchKernelGGL((pointwise_mlp_sp_kernel<true, 2, 1>), dim3(grid), dim3(512), lds_bytes, stream, a);
    else hipLaunchKernelGGL((pointwise_mlp_sp_kernel<false, 2, 1>), dim3(grid), dim3(512), lds_bytes, stream, a);
  } else if (split_f16 && add && nt == 3) hipLaunchKernelGGL((pointwise_mlp_sp_kernel<true, 3>), dim3(grid), dim3(512), lds_bytes, stream, a);
  else if (split_f16 && nt == 3) hipLaunchKernelGGL((pointwise_mlp_sp_kernel<false, 3>), dim3(grid), dim3(512), lds_bytes, stream, a);
  else if (split_f16 && add) hipLaunchKernelGGL((pointwise_mlp_sp_kernel<true, 2>), dim3(grid), dim3(512), lds_bytes, stream, a);
  else if (split_f16) hipLaunchKernelGGL((pointwise_mlp_sp_kernel<false, 2>), dim3(grid), dim3(512), lds_bytes, stream, a);
  else hipLaunchKernelGGL(pointwise_mlp_kernel, dim3(grid), dim3(512), lds_bytes, stream, a);
#ifdef GLDM_DEBUG_KNOBS
  if (split_f16 && getenv("GLDM_PW_STAMP")) {   // diagnostic builds: phase clocks of one steady-state tile (waves 0 and 7)
    long long h[64];
    (void)hipStreamSynchronize(stream);
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_pw_stamp), sizeof(h));
    for (int w = 0; w < 2; ++w) {
      const long long *q = h + 32 * w;
      printf("pointwise split %d(%d)->%d b=%d wave %d: stage %lld front %lld barrier %lld |", cin, cin0, cout, b, w ? 7 : 0,
             q[1] - q[0], q[2] - q[1], q[3] - q[2]);
      printf(" main %lld wait %lld head %lld total %lld\n", q[16] - q[3], q[17] - q[16], q[18] - q[17], q[18] - q[0]);
    }
  }
#endif
  return hipGetLastError() == hipSuccess ? GLDM_OK : GLDM_ERR_LAUNCH;
}
}  // namespace

GLDM_API int gldm_pointwise_mlp(const float *x, const float *w_packed, const float *bias, int b, int cin, int cout,
                                int n, int relu, const float *head_w_packed, const float *head_bias, int hout,
                                float *y, float *z, gldm_stream_t stream) {
  return launch_pointwise(x, nullptr, nullptr, 0, w_packed, bias, b, cin, cout, n, relu, head_w_packed, head_bias, hout,
                          y, z, reinterpret_cast<hipStream_t>(stream));
}

GLDM_API int gldm_pointwise_mlp2(const float *x, const float *w0_packed, const float *bias0, int cin0,
                                 const float *w_packed, const float *bias, int b, int cin, int cout, int n,
                                 const float *head_w_packed, const float *head_bias, int hout, float *y, float *z,
                                 gldm_stream_t stream) {
  if (!w0_packed) return GLDM_ERR_INVALID_ARG;
  return launch_pointwise(x, w0_packed, bias0, cin0, w_packed, bias, b, cin, cout, n, 1, head_w_packed, head_bias,
                          hout, y, z, reinterpret_cast<hipStream_t>(stream));
}

GLDM_API int gldm_pointwise_mlp_f16x2(const float *x, const float *w_split, const float *bias, int b, int cin, int cout,
                                       int n, int relu, const float *head_w_packed, const float *head_bias, int hout,
                                       float *y, float *z, gldm_stream_t stream) {
  return launch_pointwise(x, nullptr, nullptr, 0, w_split, bias, b, cin, cout, n, relu, head_w_packed, head_bias, hout,
                          y, z, reinterpret_cast<hipStream_t>(stream), true);
}

GLDM_API int gldm_pointwise_mlp_f16x2_pm(const float *x, const float *w_split, const float *bias, int b, int cin, int cout,
                                          int n, int relu, float *y_point_major, gldm_stream_t stream) {
  return launch_pointwise(x, nullptr, nullptr, 0, w_split, bias, b, cin, cout, n, relu, nullptr, nullptr, 0, y_point_major,
                          nullptr, reinterpret_cast<hipStream_t>(stream), true, nullptr, 0, 0, 0, nullptr, true);
}

GLDM_API int gldm_pointwise_mlp_f16x2_add(const float *x, const float *w_split, const float *bias, const float *add,
                                           long long add_cloud_stride, long long add_row_stride, long long add_col_stride, int b,
                                           int cin, int cout, int n, int relu, float *y, gldm_stream_t stream) {
  if (!add) return GLDM_ERR_INVALID_ARG;
  return launch_pointwise(x, nullptr, nullptr, 0, w_split, bias, b, cin, cout, n, relu, nullptr, nullptr, 0, y, nullptr,
                          reinterpret_cast<hipStream_t>(stream), true, add, add_cloud_stride, add_row_stride, add_col_stride);
}

GLDM_API int gldm_pointwise_mlp2_f16x2(const float *x, const float *w0_packed, const float *bias0, int cin0,
                                        const float *w_split, const float *bias, int b, int cin, int cout, int n,
                                        const float *head_w_packed, const float *head_bias, int hout,
                                        const float *front_gain, float *y, float *z, gldm_stream_t stream) {
  if (!w0_packed) return GLDM_ERR_INVALID_ARG;
  return launch_pointwise(x, w0_packed, bias0, cin0, w_split, bias, b, cin, cout, n, 1, head_w_packed, head_bias,
                          hout, y, z, reinterpret_cast<hipStream_t>(stream), true, nullptr, 0, 0, 0, front_gain);
}

namespace {
// pre != nullptr: the first layer hoisted (sa_mlp3_kernel<.., PRE>): `features` unused, c = 0, cin_pad[0] = rows of pre
int launch_sa3(const float *points, const float *centers, const float *features, const float *pre, int wa_off, int pre_bcast,
               const int32_t *idx, const float *weights, int b, int c, int n, int m, int u,
               int n_layers, const int32_t *cin_pad, const int32_t *cout, const int32_t *w3_off,
               const int32_t *b_off, const float *range_gain, float *out, gldm_stream_t stream) {
  if (!points || !centers || !idx || !weights || !out || !cin_pad || !cout || !w3_off || !b_off || b <= 0 || c < 0 ||
      n <= 0 || m <= 0 || u <= 0)
    return GLDM_ERR_INVALID_ARG;
  if (c > 0 && !features) return GLDM_ERR_INVALID_ARG;
  if (pre && (c != 0 || wa_off < 0 || (wa_off & 3))) return GLDM_ERR_INVALID_ARG;
  if (n_layers < 1 || n_layers > 4 || !(u == 16 || u == 32 || u == 64)) return GLDM_ERR_UNSUPPORTED;
  SaArgs a{};
  a.points = points; a.centers = centers; a.feat = c > 0 ? features : nullptr; a.idx = idx; a.weights = weights;
  a.out = out; a.c = c; a.n = n; a.m = m; a.u = u; a.n_layers = n_layers;
  a.ranged = range_gain != nullptr;
  a.pre = pre; a.c1 = pre ? cin_pad[0] : 0; a.wa_off = wa_off; a.pre_bcast = pre_bcast ? 1 : 0;
  for (int l = 0; l < n_layers && range_gain; ++l) {
    a.gain_r[l] = range_gain[2 * l];
    a.gain_b[l] = range_gain[2 * l + 1];
    if (!(a.gain_r[l] >= 0.f) || !(a.gain_b[l] >= 0.f)) return GLDM_ERR_INVALID_ARG;
  }
  int blocks_a = 0, blocks_b = 0;
  for (int l = 0; l < n_layers; ++l) {
    const int kb = cin_pad[l] >> 5, mt = cout[l] >> 4;
    if (cin_pad[l] <= 0 || (cin_pad[l] & 31) || !(kb <= 6 || kb == 8 || kb == 9) || cout[l] <= 0 || (cout[l] & 15)) return GLDM_ERR_UNSUPPORTED;
    if (l > 0 && cin_pad[l] != cout[l - 1]) return GLDM_ERR_INVALID_ARG;
    if (l + 1 < n_layers) {   // hidden layer: its output is the next layer's planes
      if ((cout[l] & 31) || !(mt == 2 || mt == 4 || mt == 8 || mt == 16)) return GLDM_ERR_UNSUPPORTED;
      int &blk = (l & 1) ? blocks_a : blocks_b;
      blk = blk > (cout[l] >> 5) ? blk : (cout[l] >> 5);
    }
    a.cin_pad[l] = cin_pad[l]; a.cout[l] = cout[l]; a.w_off[l] = w3_off[l]; a.b_off[l] = b_off[l];
  }
  if (!pre && (cin_pad[0] < 3 + c || cin_pad[0] > 32 * kSa3Quads)) return GLDM_ERR_UNSUPPORTED;
  if (pre && cin_pad[0] > 256) return GLDM_ERR_UNSUPPORTED;
  blocks_a = blocks_a > (cin_pad[0] >> 5) ? blocks_a : (cin_pad[0] >> 5);
  const size_t tile_bytes = (size_t)(blocks_a + blocks_b) * kSaBlockFloats * sizeof(float);
  // behind the planes: the eight range words and (single-tile kernels) the staged output rows of a run
  const size_t kRngBytes = 64 + (size_t)cout[n_layers - 1] * kSaRun * sizeof(float);
  if (tile_bytes + kRngBytes > 160 * 1024) return GLDM_ERR_UNSUPPORTED;
  const int cpt = 64 / u, tpc = (m + cpt - 1) / cpt, total = tpc * b;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  // narrow nets (32-row inputs, every K in {32, 64, 128}): the multi-tile kernel -- the layer's weights once per
  // workgroup pass, `sub` tiles' planes side by side in LDS (SSG SA1: three 48 KiB tiles, 2.52 -> 2.07 ms).  Two
  // co-resident workgroups of one tile each (the kernel fits 128 registers) measured slower: 2.40 ms.
  // one or two row quads per gather thread; the hoisted form (pre): the multi-tile kernel with two tiles per pass
  bool kb_ok = cin_pad[0] == 32 || cin_pad[0] == 64;
  for (int l = 0; l < n_layers; ++l) kb_ok = kb_ok && (cin_pad[l] == 32 || cin_pad[l] == 64 || cin_pad[l] == 128);
  int sub = kb_ok ? (int)(((size_t)160 * 1024 - kRngBytes) / tile_bytes) : 1;
  if (sub > 4) sub = 4;
  // two workgroups of two tiles each rather than one of four where LDS allows (the kernel fits 128 registers): the phases of
  // one overlap the other's (SSG-SA1 at 256 clouds: 1.55 -> 1.47 ms)
  if (sub > 2 && (tile_bytes * 2 + kRngBytes) * 2 <= (size_t)160 * 1024) sub = 2;
  if (pre && sub > 2) sub = 2;
  while (sub > 1 && (total + sub - 1) / sub < 2 * cu_count()) --sub;
  if (sub > 1) {
    const int supers = (total + sub - 1) / sub;
    const int per_cu_m = (tile_bytes * sub + kRngBytes) * 2 <= (size_t)160 * 1024 ? 2 : 1;
    const int grid = supers < cu_count() * per_cu_m ? supers : cu_count() * per_cu_m;
    if (pre) {
      struct Sa3mPreTag { int site; };
      gldm_dev::allow_dynamic_lds<Sa3mPreTag>(reinterpret_cast<const void *>(&sa_mlp3_kernel<2, 2, true>), 160 * 1024);
      hipLaunchKernelGGL((sa_mlp3_kernel<2, 2, true>), dim3(grid), dim3(512), tile_bytes * sub + kRngBytes, s, a, blocks_a, blocks_b, sub, tpc, total);
    } else if (cin_pad[0] == 32) {
      struct Sa3mTag { int site; };
      gldm_dev::allow_dynamic_lds<Sa3mTag>(reinterpret_cast<const void *>(&sa_mlp3_kernel<4, 1>), 160 * 1024);
      hipLaunchKernelGGL((sa_mlp3_kernel<4, 1>), dim3(grid), dim3(512), tile_bytes * sub + kRngBytes, s, a, blocks_a, blocks_b, sub, tpc, total);
    } else {
      struct Sa3m2Tag { int site; };
      gldm_dev::allow_dynamic_lds<Sa3m2Tag>(reinterpret_cast<const void *>(&sa_mlp3_kernel<4, 2>), 160 * 1024);
      hipLaunchKernelGGL((sa_mlp3_kernel<4, 2>), dim3(grid), dim3(512), tile_bytes * sub + kRngBytes, s, a, blocks_a, blocks_b, sub, tpc, total);
    }
  } else {
    const int per_cu = (tile_bytes + kRngBytes) * 2 <= 160 * 1024 ? 2 : 1;
    int grid = cu_count() * per_cu;
    if (grid > total) grid = total;
    if (pre && cin_pad[0] <= 128) {
      struct Sa3Pre4Tag { int site; };
      gldm_dev::allow_dynamic_lds<Sa3Pre4Tag>(reinterpret_cast<const void *>(&sa_mlp3_kernel<1, 4, true>), 160 * 1024);
      hipLaunchKernelGGL((sa_mlp3_kernel<1, 4, true>), dim3(grid), dim3(512), tile_bytes + kRngBytes, s, a, blocks_a, blocks_b, 1, tpc, total);
    } else if (pre) {
      struct Sa3PreTag { int site; };
      gldm_dev::allow_dynamic_lds<Sa3PreTag>(reinterpret_cast<const void *>(&sa_mlp3_kernel<1, 8, true>), 160 * 1024);
      hipLaunchKernelGGL((sa_mlp3_kernel<1, 8, true>), dim3(grid), dim3(512), tile_bytes + kRngBytes, s, a, blocks_a, blocks_b, 1, tpc, total);
    } else {
      struct Sa3Tag { int site; };
      gldm_dev::allow_dynamic_lds<Sa3Tag>(reinterpret_cast<const void *>(&sa_mlp3_kernel<1, kSa3Quads>), 160 * 1024);
      hipLaunchKernelGGL((sa_mlp3_kernel<1, kSa3Quads>), dim3(grid), dim3(512), tile_bytes + kRngBytes, s, a, blocks_a, blocks_b, 1, tpc, total);
    }
  }
  return hipGetLastError() == hipSuccess ? GLDM_OK : GLDM_ERR_LAUNCH;
}
}  // namespace

GLDM_API int gldm_sa_mlp_forward_f16x2(const float *points, const float *centers, const float *features,
                                        const int32_t *idx, const float *weights, int b, int c, int n, int m, int u,
                                        int n_layers, const int32_t *cin_pad, const int32_t *cout, const int32_t *w3_off,
                                        const int32_t *b_off, const float *range_gain, float *out, gldm_stream_t stream) {
  return launch_sa3(points, centers, features, nullptr, 0, 0, idx, weights, b, c, n, m, u, n_layers, cin_pad, cout, w3_off, b_off,
                    range_gain, out, stream);
}

GLDM_API int gldm_sa_mlp_forward_f16x2_pre(const float *points, const float *centers, const float *pre, int pre_broadcast,
                                            const int32_t *idx, const float *weights, int wa_off, int b, int n, int m, int u, int n_layers,
                                            const int32_t *cin_pad, const int32_t *cout, const int32_t *w3_off,
                                            const int32_t *b_off, const float *range_gain, float *out, gldm_stream_t stream) {
  if (!pre) return GLDM_ERR_INVALID_ARG;
  return launch_sa3(points, centers, nullptr, pre, wa_off, pre_broadcast, idx, weights, b, 0, n, m, u, n_layers, cin_pad, cout, w3_off,
                    b_off, range_gain, out, stream);
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
  if (u > 64 || (64 % u) != 0) return GLDM_ERR_UNSUPPORTED;
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
  {  // 128-column tiles when the layer plan fits: widths 32 / 64 / 128 / 256 k, U a multiple of 16, both regions in LDS
    bool ok = (u == 16 || u == 32 || u == 64) && n_layers >= 1 && c <= 4 * kSaFly;
    int rows_a = cin_pad[0], rows_b = 0;
    for (int l = 0; l < n_layers && ok; ++l) {
      const int mt = cout[l] >> 4;
      ok = (mt == 2 || mt == 4 || (mt >= 8 && (mt & 7) == 0)) && (cin_pad[l] & 31) == 0;
      if (l + 1 < n_layers) {  // stored outputs: even layers -> B, odd layers -> A
        if (l & 1) rows_a = rows_a > cout[l] ? rows_a : cout[l];
        else rows_b = rows_b > cout[l] ? rows_b : cout[l];
      } else if (ok) {  // the max runs on one wave's n-tiles: a centre's U / 16 tiles must not straddle two waves
        const int nt = mt >= 8 ? 8 : (mt == 4 ? 4 : 2);
        ok = (u >> 4) <= nt;
      }
    }
    const size_t lds2 = (size_t)(rows_a + rows_b) * 128 * sizeof(float);
#ifdef GLDM_DEBUG_KNOBS
    static const bool tile64 = getenv("GLDM_SA_TILE64") != nullptr;  // diagnostic builds: force the 64-column kernel
#else
    constexpr bool tile64 = false;  // the shipped library reads no environment
#endif
    if (ok && lds2 <= 160 * 1024 && !tile64) {
      struct Sa2Tag { int site; };
      gldm_dev::allow_dynamic_lds<Sa2Tag>(reinterpret_cast<const void *>(&sa_mlp2_kernel), 160 * 1024);
      const int cpt2 = 128 / u, tpc = (m + cpt2 - 1) / cpt2, total = tpc * b;
      const int grid = total < cu_count() ? total : cu_count();
      hipLaunchKernelGGL(sa_mlp2_kernel, dim3(grid), dim3(512), lds2, reinterpret_cast<hipStream_t>(stream), a, rows_a,
                         tpc, total);
      return hipGetLastError() == hipSuccess ? GLDM_OK : GLDM_ERR_LAUNCH;
    }
  }
  const size_t lds_bytes = (size_t)(Geo<64>::kBufH + kMaxC * 64) * sizeof(float);
  struct SaTag { int site; };
  gldm_dev::allow_dynamic_lds<SaTag>(reinterpret_cast<const void *>(&sa_mlp_kernel), (int)lds_bytes);
  const int cpt = 64 / u;
  hipLaunchKernelGGL(sa_mlp_kernel, dim3((m + cpt - 1) / cpt, b), dim3(Geo<64>::kThreads), lds_bytes,
                     reinterpret_cast<hipStream_t>(stream), a);
  return hipGetLastError() == hipSuccess ? GLDM_OK : GLDM_ERR_LAUNCH;
}
