// quad_narrow.h -- the narrow levels (4, 32 and 64 channels) of the 4-position latent denoiser, wave-local and
// register-resident.
//
// Included by resnet1d.hip (uses its Ctx, Geo, WStream, split-f16 helpers and cross-lane reductions).
//
// Why.  On the position-major engine a narrow level is ~7 barrier-separated phases of 8 cooperating waves, each
// with a few hundred cycles of matrix work and 4-9 k cycles of latency (cold parameter loads, a cross-wave GroupNorm
// exchange, two barriers, the interpreter's set-up): ops 0-18 of the shipped denoiser took 123 k of a 260 k-cycle
// step for 6 % of its FLOPs (profiles/r05_denoise_stamps_f16_start.txt).  Nothing in those levels needs more than
// one wave: 4 samples x 4 positions are exactly one 16-column MFMA n-tile.
//
// How.  Waves q and q + 4 (the two waves of one SIMD) own samples 4 q .. 4 q + 3 of the workgroup's 16 ("a quad") and walk
//   init level (4 ch):  ResnetBlock x 2, attention, down conv 4 -> 32
//   32-channel level :  ResnetBlock x 2, attention, down conv 32 -> 64
//   64-channel level :  ResnetBlock x 2, attention, down conv 64 -> 128
// on their own, with NO barrier and NO activation tensor in LDS: every m-tile of every layer over its single n-tile (column =
// 4 * position + sample).
//   * The residual stream lives in registers in the accumulator layout: lane (column, g) holds rows 4 g + r of every m-tile.
//   * A GEMM's B operand is made from that layout in place: the lane's 8 values of m-tiles 2 kb and 2 kb + 1, split into
//     f16 hi / lo pairs, ARE its fragment of 32-channel block kb -- with k-slot 8 g + j standing for channel
//     32 kb + 16 (j >> 2) + 4 g + (j & 3), which is the column order the packer stores these layers' weights in
//     (r1d_pack.quad_perm32, include/gldm.h "Quad column order").  The sum over k does not care.
//   * A k = 3 tap is the same fragment shifted by one position = 4 lanes inside the 16-lane row: DPP row shifts of the
//     packed registers, zero fill = the conv's zero padding.
//   * GroupNorm, LayerNorm and the attention softmaxes reduce inside the wave (in-lane, DPP over the positions,
//     permlane swaps over the row quarters); the H range scale (conv_pm3_wave) needs no exchange either.
//   * The pair splits every layer's m-tiles and every attention block's heads, and swaps what the next layer needs
//     through LDS mailboxes (see QW / qw_xchg below).
//   * Weights stream through ONE ring in LDS along a compile-time list of every fragment the chain consumes, in
//     consumption order (qstream_off), filled by LDS-DMA by the waves themselves: no layer starts cold, and one copy
//     serves the four quads.
// The last down conv writes the 128-channel residual stream where the
// position-major engine expects it (f32 rows + X planes, column = 16 * position + sample) and the tape goes on there.
//
// The 4-channel level runs on the f32 matrix pipe with ONE value per lane: lane (column, kq) holds channel kq of its
// column, which is exactly the B operand of v_mfma_f32_16x16x4_f32 (k = kq), the taps are DPP row shifts by 4 lanes,
// and the weight rows are gathered so that output channel ch lands in row 4 ch: register 0 of lane (column, kq = ch).
#ifndef GLDM_QUAD_NARROW_H_
#define GLDM_QUAD_NARROW_H_

#ifdef GLDM_DEBUG_KNOBS
__device__ long long g_q_stamp[8][16];   // per wave: cycle counter behind every stage of the chain (last step of workgroup 0)
#define GLDM_QSTAMP(c, i) do { if (blockIdx.x == 0 && (c).lane == 0) g_q_stamp[(c).wave][i] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define GLDM_QSTAMP(c, i) do {} while (0)
#endif

template <int CTRL>
__device__ __forceinline__ float dpp_zero(float x) {   // lanes without a source read 0
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ u32x4 dpp_zero4(const u32x4 &v) {
  u32x4 o;
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v[i], CTRL, 0xf, 0xf, true);
  return o;
}
__device__ __forceinline__ float pos_sum(float v) {   // over the 4 positions of a sample: lanes col, col + 4, + 8, + 12 of a row
  v += dpp_mov<0x124>(v);
  return v + dpp_mov<0x128>(v);
}
__device__ __forceinline__ float pos_max(float v) { return dpp_max<0x128>(dpp_max<0x124>(v)); }
__device__ __forceinline__ float kq_sum(float v) { return half_sum(row_pair_sum(v)); }   // over the four row quarters
__device__ __forceinline__ float kq_max(float v) { return half_max(row_pair_max(v)); }

// the lane's rows of m-tiles 2 kb (a) and 2 kb + 1 (b) -> its B fragment of block kb, quad column order
__device__ __forceinline__ void qsplit8(const f32x4 &a, const f32x4 &b, u32x4 (&pl)[kSplit]) {
  unsigned h, l;
  split_f16x2(a[0], a[1], h, l); pl[0][0] = h; pl[1][0] = l;
  split_f16x2(a[2], a[3], h, l); pl[0][1] = h; pl[1][1] = l;
  split_f16x2(b[0], b[1], h, l); pl[0][2] = h; pl[1][2] = l;
  split_f16x2(b[2], b[3], h, l); pl[0][3] = h; pl[1][3] = l;
}

// ---- the weight stream -------------------------------------------------------------------------------------------------
// Matrices of the chain (byte offsets of their fragment arrays in the packed weight buffer):
enum { QM_OUT0 = 0, QM_SS2, QM_C1_2, QM_C2_2, QM_SS3, QM_C1_3, QM_C2_3, QM_QKV1, QM_OUT1, QM_DOWN1,
       QM_SS4, QM_C1_4, QM_C2_4, QM_SS5, QM_C1_5, QM_C2_5, QM_QKV2, QM_OUT2, QM_DOWN2, QM_COUNT };
struct QOff { int mat, a, b; };   // a slot's two 1-KiB halves: byte offsets a and b inside matrix `mat`
// A quad is worked by a PAIR of waves on one SIMD: wave q (half H = 0) and wave q + 4 (H = 1).  One wave per SIMD issues
// an instruction every 4-5 cycles whatever it is; measured, the chain of a single wave was 61 k cycles WITHOUT any MFMA
// and 77 k with them (profiles/r05_quad_stamps.txt) -- pure issue time.  Two waves on a SIMD alternate, so a layer's
// m-tiles (and an attention block's heads) are split between the halves, which hand each other what the next layer needs
// through LDS mailboxes (qw_xchg).  H = 0 owns the first half of every layer's m-tiles and heads 0, 1.
//
// Every fragment of the chain is consumed by exactly one half (of each quad).  The stream lists them in consumption order
// with the halves interleaved: own fragment P of half H is stream fragment n = 2 P + H, so both halves walk the stream
// at the same pace.  Own sequences, in order (MT m-tiles, KB 32-channel blocks, M2 = MT / 2):
//   4-channel level      : to_out of the half's two heads (2)
//   ResnetBlock          : scale/shift Linear of the own m-tiles (M2: halves = scale m-tile | shift m-tile, f32 fragments)
//                          | conv1 (tap step st = 3 kb + t, own m-tile ml) | conv2            (M2 (1 + 6 KB))
//   attention            : to_qkv of own head 0 (6 m-tiles x KB) | of own head 1 | to_out (own m-tile ml, head 0..3)
//                                                                                             (12 KB + 2 MT)
//   down conv to 2 MT    : (tap step, own OUTPUT m-tile of MT)                                (3 KB MT)
constexpr QOff qfrag(int mat, int f) { return QOff{mat, 2048 * f, 2048 * f + 1024}; }
constexpr int qrb_len(int MT, int KB) { return (MT / 2) * (1 + 6 * KB); }
constexpr QOff qrb_own(int mss, int MT, int KB, int H, int i) {
  const int M2 = MT / 2;
  if (i < M2) return QOff{mss, 1024 * (H * M2 + i), 1024 * (MT + H * M2 + i)};
  i -= M2;
  const int conv = i / (3 * KB * M2), j = i % (3 * KB * M2), st = j / M2, ml = j % M2, kb = st / 3, t = st % 3;
  return qfrag(mss + 1 + conv, (H * M2 + ml) * 3 * KB + t * KB + kb);
}
constexpr int qatt_len(int MT, int KB) { return 12 * KB + 2 * MT; }
constexpr QOff qatt_own(int mqkv, int MT, int KB, int H, int i) {
  if (i < 12 * KB) {
    const int h = 2 * H + i / (6 * KB), r = i % (6 * KB), mt6 = r / KB, kb = r % KB;
    return qfrag(mqkv, (2 * h + (mt6 & 1) + 8 * (mt6 >> 1)) * KB + kb);
  }
  const int r = i - 12 * KB, ml = r / 4, h = r % 4;
  return qfrag(mqkv + 1, (H * (MT / 2) + ml) * 4 + h);
}
constexpr int qdown_len(int MT, int KB) { return 3 * KB * MT; }
constexpr QOff qdown_own(int mat, int MT, int KB, int H, int i) {   // MT own output m-tiles: H MT + ml
  const int st = i / MT, ml = i % MT, kb = st / 3, t = st % 3;
  return qfrag(mat, (H * MT + ml) * 3 * KB + t * KB + kb);
}
constexpr int qlevel_len(int MT, int KB) { return 2 * qrb_len(MT, KB) + qatt_len(MT, KB) + qdown_len(MT, KB); }
constexpr QOff qlevel_own(int m0, int MT, int KB, int H, int i) {
  if (i < qrb_len(MT, KB)) return qrb_own(m0, MT, KB, H, i);
  i -= qrb_len(MT, KB);
  if (i < qrb_len(MT, KB)) return qrb_own(m0 + 3, MT, KB, H, i);
  i -= qrb_len(MT, KB);
  if (i < qatt_len(MT, KB)) return qatt_own(m0 + 6, MT, KB, H, i);
  return qdown_own(m0 + 8, MT, KB, H, i - qatt_len(MT, KB));
}
// own positions of the levels
constexpr int kQP0 = 0;                               // 4-channel level: 2
constexpr int kQP1 = 2;                               // 32-channel level
constexpr int kQP2 = kQP1 + qlevel_len(2, 1);         // 64-channel level
constexpr int kQPEnd = kQP2 + qlevel_len(4, 2);
constexpr int kQNEnd = 2 * kQPEnd;                    // stream fragments
constexpr QOff qstream_off(int n) {
  const int H = n & 1, P = n >> 1;
  if (P < kQP1) return qfrag(QM_OUT0, 2 * H + P);
  if (P < kQP2) return qlevel_own(QM_SS2, 2, 1, H, P - kQP1);
  return qlevel_own(QM_SS4, 4, 2, H, P - kQP2);
}

// The fragments reach the waves through LDS: a ring of kQS 2-KiB slots (kQG fragments = a group = 8 KiB) filled by
// LDS-DMA (global_load_lds_dwordx4: a fragment plane is 64 lanes x 16 bytes, exactly one instruction, no registers) from
// the stream table in LDS (qtab: the two byte offsets of every fragment, quad_build_table).  One copy in LDS serves the four
// quads: every narrow weight crosses L2 -> CU once per tile-step.  All eight waves are consumers AND loaders: wave W loads
// the groups g = W (mod 8), one at a time, at "ticks" placed every fourth own fragment (and inside every wait loop, so
// that nobody can wait for a group whose loader is itself waiting).
// Hand-shake, all in LDS (sync words, zeroed in the step prologue; every lane of a wave stores the same value to the
// same word -- no branch):
//   fill[g % 8] = g + 1   once group g has landed (its loader: DMA, later s_waitcnt vmcnt(0), store);
//   done[W]     = G       wave W needs no group below G any more (stored at every fourth own fragment);
//   a loader refills the slots of group g - 8 with group g only when every done[W] >= g - 7.
constexpr int kQG = 4;            // fragments per group
constexpr int kQS = 52;           // LDS slots (104 KiB): 13 groups.  The ring has to be DEEP: a group takes ~2.6 k cycles from its
                                  // DMAs to LDS, the convs swallow one in ~150, and only the stages in between (epilogues,
                                  // attention cores) give the loaders time to get ahead again
constexpr int kQSG = kQS / kQG;   // slot groups
constexpr int kQGroups = kQNEnd / kQG;
static_assert(kQNEnd % kQG == 0 && kQS % kQG == 0, "whole groups");
constexpr int kQR = 4;            // own fragments a wave holds in registers ahead of their use
constexpr int kQDmaAge = 10;      // own fragments between issuing a group's DMAs and waiting for them at a tick
constexpr int kQSpinMax = 1 << 20; // every wait is bounded (a healthy one is a few polls): a lost hand-shake ends in wrong numbers, not a hang
// The slots take the arena from 8 KiB to 112 KiB: in front of them the f32 rows 0 .. 3 the chain starts from, behind
// them the mailboxes.  The chain's last conv writes over them (f32 rows 0 .. 127, X planes): it keeps its results in
// registers until every wave of the workgroup is through (a barrier inside quad_narrow_levels).
constexpr int kQSlot0 = 2048;
__host__ __device__ constexpr int qslot_floats(int s) { return kQSlot0 + s * 512; }
constexpr int kQBox = kQSlot0 + kQS * 512;   // mailboxes: 8 waves x 2 x 2 KiB behind the slots
static_assert(kQSlot0 >= 4 * 64 && kQBox + 16 * 512 <= Geo<64>::kArena, "slots and mailboxes fit the arena");
typedef __attribute__((address_space(3))) int lds_i;
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glob_void;
enum { QS_FILL = 0, QS_DONE = 16, QS_WSEQ = 24, QS_RSEQ = 32 };   // sync words (fill: kQSG of them)

__device__ __forceinline__ int lds_poll(const lds_i *p) {
  int v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ int lds_min8(const lds_i *p) {   // min of 8 consecutive words (16-byte aligned)
  typedef __attribute__((ext_vector_type(4))) int i4;
  i4 a, b;
  asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)" : "=&v"(a), "=&v"(b) : "v"(p) : "memory");
  const int m = min(min(min(a[0], a[1]), min(a[2], a[3])), min(min(b[0], b[1]), min(b[2], b[3])));
  return __builtin_amdgcn_readfirstlane(m);
}

// the stream table (once per kernel): absolute byte offsets of every fragment's two halves
__device__ __forceinline__ void quad_build_table(const gldm_r1d_desc &d, int *qtab, int tid, int nthreads) {
  const int mbf[QM_COUNT] = {d.lv[0].out_wq,
                             d.rb[2].ss_w, d.rb[2].c1_wq, d.rb[2].c2_wq, d.rb[3].ss_w, d.rb[3].c1_wq, d.rb[3].c2_wq,
                             d.lv[1].qkvn_wq, d.lv[1].out_wq, d.lv[1].down_wq,
                             d.rb[4].ss_w, d.rb[4].c1_wq, d.rb[4].c2_wq, d.rb[5].ss_w, d.rb[5].c1_wq, d.rb[5].c2_wq,
                             d.lv[2].qkvn_wq, d.lv[2].out_wq, d.lv[2].down_wq};
  for (int n = tid; n < kQNEnd; n += nthreads) {
    const QOff o = qstream_off(n);
    int base = 0;
#pragma unroll
    for (int m = 0; m < QM_COUNT; ++m) base = o.mat == m ? mbf[m] * 4 : base;   // constant indices only (kernel argument)
    qtab[2 * n] = base + o.a;
    qtab[2 * n + 1] = base + o.b;
  }
}

template <int H>
struct QW {
  u32x4 s[kQR][2];   // register ring: own fragment P in s[P % kQR]
  float *lds;
  lds_i *sync;
  const lds_i *qtab;
  const char *wb;    // weights + this lane's 16 bytes
  int lane, quad;
  int flag;          // fill word of the NEXT group, read a group ahead (a stale "not yet" only costs the slow path)
  int ack;           // the partner's RSEQ as last seen
  int xseq;          // exchanges done
  int dma_group, dma_age, next_load;   // loader state: group in flight (-1: none), own fragments since, next group to load
  int spins;         // diagnostic builds: polls that waited
};

// loader duty (see above).  blocking: wait for the DMAs in flight whatever their age (wait loops; the chain's end)
template <int H>
__device__ __forceinline__ void qw_tick(QW<H> &q, bool blocking) {
  if (q.dma_group >= 0 && (blocking || q.dma_age >= kQDmaAge)) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    q.sync[QS_FILL + q.dma_group % kQSG] = q.dma_group + 1;
    q.dma_group = -1;
  }
  if (q.dma_group < 0 && q.next_load < kQGroups) {
    const int g = q.next_load;
    if (g < kQSG || lds_min8(q.sync + QS_DONE) >= g - kQSG + 1) {   // the slots still hold group g - kQSG: everybody must be done with it
      const int s0 = (g % kQSG) * kQG;
#pragma unroll
      for (int f = 0; f < kQG; ++f) {
        const int n = g * kQG + f;
        const int oa = __builtin_amdgcn_readfirstlane(q.qtab[2 * n]), ob = __builtin_amdgcn_readfirstlane(q.qtab[2 * n + 1]);
        float *slot = q.lds + kQSlot0 + (s0 + f) * 512;
        __builtin_amdgcn_global_load_lds((glob_void *)(q.wb + oa), (lds_void *)slot, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((glob_void *)(q.wb + ob), (lds_void *)(slot + 256), 16, 0, 0);
      }
      q.dma_group = g;
      q.dma_age = 0;
      q.next_load = g + 8;
    }
  }
}
// request own fragment P from its LDS slot into its registers.  At a group's first own fragment: the group must have
// landed -- checked on a word read one group earlier (straight-line code: the reads in flight stay in flight), with a
// polling loop (that keeps ticking) only if that said "not yet".
template <int H, int P>
__device__ __forceinline__ void qw_load(QW<H> &q) {
  if constexpr (P < kQPEnd) {
    constexpr int n = 2 * P + H;
    if constexpr ((P & 1) == 0) {
      constexpr int g = n / kQG;
      if (__builtin_expect(q.flag != g + 1, 0)) {
        for (int spin = 0; spin < kQSpinMax && lds_poll(q.sync + QS_FILL + g % kQSG) != g + 1; ++spin) {
          qw_tick<H>(q, spin > 4);   // its own DMA group is waited for (vmcnt) only when the wait drags on
          __builtin_amdgcn_s_sleep(1);
#ifdef GLDM_DEBUG_KNOBS
          ++q.spins;
#endif
        }
      }
      if constexpr (g + 1 < kQGroups) q.flag = *(volatile lds_i *)(q.sync + QS_FILL + (g + 1) % kQSG);
    }
    const lds_u4 *slot = (const lds_u4 *)(q.lds + qslot_floats(n % kQS)) + q.lane;
    q.s[P % kQR][0] = slot[0];
    q.s[P % kQR][1] = slot[64];
  }
}
// acc += (own fragment P) * B; then the registers of the PREVIOUS own fragment are refilled (this one's are still being read
// by the MFMAs just issued); every fourth own fragment: the done word and a loader tick
template <int H, int P>
__device__ __forceinline__ f32x4 qw_mfma(QW<H> &q, const u32x4 (&b)[kSplit], f32x4 acc) {
  acc = mfma_split(q.s[P % kQR], b, acc);
  ++q.dma_age;
  if constexpr ((P & 3) == 0) {
    q.sync[QS_DONE + q.quad + 4 * H] = (2 * P + H) / kQG;
    qw_tick<H>(q, false);
  }
  if constexpr (P >= 1) qw_load<H, P - 1 + kQR>(q);
  return acc;
}
// the same bookkeeping for a fragment that is not multiplied here (the scale/shift Linear's f32 fragments)
template <int H, int P>
__device__ __forceinline__ void qw_retire(QW<H> &q) {
  ++q.dma_age;
  if constexpr ((P & 3) == 0) {
    q.sync[QS_DONE + q.quad + 4 * H] = (2 * P + H) / kQG;
    qw_tick<H>(q, false);
  }
  if constexpr (P >= 1) qw_load<H, P - 1 + kQR>(q);
}

// The two halves of a quad swap NW dwords per lane through their mailboxes (2 x 2 KiB per wave: [plane][lane][16 bytes],
// alternating between the two so that a fast wave's next message cannot overwrite one its partner has not read yet).
// The hand-over is the WORKGROUP barrier: the eight waves run the same program on symmetric data and reach every swap
// within a few hundred cycles of each other; flags polled through LDS cost each swap ~1.5 k cycles (two waves of a SIMD,
// the older one served first by the matrix pipe, waiting for each other in 64-cycle sleeps).
template <int H, int NW>
__device__ __forceinline__ void qw_xchg(QW<H> &q, const unsigned (&mine)[NW], unsigned (&theirs)[NW]) {
  static_assert(NW == 1 || NW == 2 || NW == 4 || NW == 8, "payload");
  const int me = q.quad + 4 * H, pa = q.quad + 4 * (1 - H);
  const int par = (q.xseq++) & 1;
  float *mybox = q.lds + kQBox + (2 * me + par) * 512, *pabox = q.lds + kQBox + (2 * pa + par) * 512;
  if constexpr (NW == 8) {
    ((lds_u4 *)mybox)[q.lane] = u32x4{mine[0], mine[1], mine[2], mine[3]};
    ((lds_u4 *)mybox)[64 + q.lane] = u32x4{mine[4], mine[5], mine[6], mine[7]};
  } else if constexpr (NW == 4) {
    ((lds_u4 *)mybox)[q.lane] = u32x4{mine[0], mine[1], mine[2], mine[3]};
  } else if constexpr (NW == 2) {
    ((lds_u2 *)mybox)[q.lane] = u32x2_t{mine[0], mine[1]};
  } else {
    ((lds_i *)mybox)[q.lane] = (int)mine[0];
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  if constexpr (NW == 8) {
    const u32x4 a = ((const lds_u4 *)pabox)[q.lane], b = ((const lds_u4 *)pabox)[64 + q.lane];
    theirs[0] = a[0]; theirs[1] = a[1]; theirs[2] = a[2]; theirs[3] = a[3];
    theirs[4] = b[0]; theirs[5] = b[1]; theirs[6] = b[2]; theirs[7] = b[3];
  } else if constexpr (NW == 4) {
    const u32x4 a = ((const lds_u4 *)pabox)[q.lane];
    theirs[0] = a[0]; theirs[1] = a[1]; theirs[2] = a[2]; theirs[3] = a[3];
  } else if constexpr (NW == 2) {
    const u32x2_t a = ((const lds_u2 *)pabox)[q.lane];
    theirs[0] = a[0]; theirs[1] = a[1];
  } else {
    theirs[0] = (unsigned)((const lds_i *)pabox)[q.lane];
  }
}
template <int H>
__device__ __forceinline__ float qw_xchg1(QW<H> &q, float v) {
  const unsigned m[1] = {__float_as_uint(v)};
  unsigned t[1];
  qw_xchg<H, 1>(q, m, t);
  return __uint_as_float(t[0]);
}

// The full fragment planes of a tensor from the halves' own m-tiles: KB = 2 (64 channels): block H is the own one, block
// 1 - H comes from the partner; KB = 1 (32 channels): the own m-tile is elements 2 H, 2 H + 1 of either plane.
template <int H, int KB>
__device__ __forceinline__ void qw_planes(QW<H> &q, const f32x4 (&own)[KB], u32x4 (&pl)[KB][kSplit]) {
  if constexpr (KB == 2) {
    qsplit8(own[0], own[1], pl[H]);
    const unsigned m[8] = {pl[H][0][0], pl[H][0][1], pl[H][0][2], pl[H][0][3], pl[H][1][0], pl[H][1][1], pl[H][1][2], pl[H][1][3]};
    unsigned t[8];
    qw_xchg<H, 8>(q, m, t);
    pl[1 - H][0] = u32x4{t[0], t[1], t[2], t[3]};
    pl[1 - H][1] = u32x4{t[4], t[5], t[6], t[7]};
  } else {
    unsigned m[4], t[4];
    split_f16x2(own[0][0], own[0][1], m[0], m[2]);
    split_f16x2(own[0][2], own[0][3], m[1], m[3]);
    qw_xchg<H, 4>(q, m, t);
    if constexpr (H == 0) {
      pl[0][0] = u32x4{m[0], m[1], t[0], t[1]};
      pl[0][1] = u32x4{m[2], m[3], t[2], t[3]};
    } else {
      pl[0][0] = u32x4{t[0], t[1], m[0], m[1]};
      pl[0][1] = u32x4{t[2], t[3], m[2], m[3]};
    }
  }
}

// A k = 3 conv of the half's M own m-tiles over the quad's one n-tile, own position P0: acc[ml] += W[own m-tile ml, (tap,
// channel)] * taps(xp).  xp[kb]: the input's full fragment planes; the taps are row shifts by 4 lanes.
template <int H, int P0, int M, int KB>
__device__ __forceinline__ void qw_conv3(QW<H> &q, const u32x4 (&xp)[KB][kSplit], f32x4 (&acc)[M]) {
  using std::integral_constant;
  auto body = [&](auto st_c) {
    constexpr int st = decltype(st_c)::value, kb = st / 3, t = st % 3;
    u32x4 bs[kSplit];
#pragma unroll
    for (int pl = 0; pl < kSplit; ++pl)
      bs[pl] = t == 1 ? xp[kb][pl] : (t == 0 ? dpp_zero4<0x114>(xp[kb][pl]) : dpp_zero4<0x104>(xp[kb][pl]));   // position p - 1 / p + 1
    auto per_m = [&](auto ml_c) {
      constexpr int ml = decltype(ml_c)::value;
      acc[ml] = qw_mfma<H, P0 + st * M + ml>(q, bs, acc[ml]);
    };
    per_m(integral_constant<int, 0>{});
    if constexpr (M > 1) per_m(integral_constant<int, 1>{});
    if constexpr (M > 2) { per_m(integral_constant<int, 2>{}); per_m(integral_constant<int, 3>{}); }
    __builtin_amdgcn_sched_barrier(0);
  };
  body(integral_constant<int, 0>{}); body(integral_constant<int, 1>{}); body(integral_constant<int, 2>{});
  if constexpr (KB > 1) { body(integral_constant<int, 3>{}); body(integral_constant<int, 4>{}); body(integral_constant<int, 5>{}); }
}

// GroupNorm statistics of a conv output held as acc[MT] (rows 16 mi + 4 kq + r of the lane's column): groups of CPG
// channels x the 4 positions of the lane's sample; two passes (mean, then the centred squares), everything in the wave.
template <int MT, int CPG>
__device__ __forceinline__ void qgn_stats(const f32x4 (&acc)[MT], float (&mean)[MT], float (&var)[MT]) {
  static_assert(CPG == 8 || CPG == 16, "groups of half an m-tile or a whole one");
  constexpr float inv_n = 1.0f / (float)(CPG * 4);
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) {
    float s = (acc[mi][0] + acc[mi][1]) + (acc[mi][2] + acc[mi][3]);
    s = row_pair_sum(pos_sum(s));
    if (CPG == 16) s = half_sum(s);
    const float m = s * inv_n;
    float v = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float d = acc[mi][r] - m;
      v = fmaf(d, d, v);
    }
    v = row_pair_sum(pos_sum(v));
    if (CPG == 16) v = half_sum(v);
    mean[mi] = m;
    var[mi] = v * inv_n;
  }
}

struct QRb { int c1_b, n1_w, n1_b, c2_b, n2_w, n2_b, ss_b; };

// One ResnetBlock of a 32- or 64-channel level, own position P0 (ss | conv1 | conv2), on the half's M2 = MT / 2 m-tiles:
// x += act(GN(conv2(act((scale + 1) GN(conv1(x)) + shift)))).  xr: the own m-tiles of the residual stream, xp: its full
// fragment planes (kept current on exit).  GroupNorm groups (C / 4 channels) never straddle the halves.
template <int H, int P0, int MT, int KB>
__device__ __forceinline__ void quad_resblock(const Ctx &c, QW<H> &q, const QRb &rb, f32x4 (&xr)[MT / 2],
                                              u32x4 (&xp)[KB][kSplit], int smp) {
  using GG = Geo<64>;
  using std::integral_constant;
  constexpr int C = 16 * MT, CPG = C / 4, M2 = MT / 2;
  static_assert(M2 == KB, "own m-tiles = one block of planes (64 channels) or half of one (32)");
  const int kq = c.lane >> 4;
  const float *w = c.w;
  // ---- scale / shift rows of the lane's sample: [2 C x 16] Linear on the f32 matrix pipe against the embedding sums
  const lds_f *Gs = (const lds_f *)(c.lds + GG::kMiscG) + smp * 16;
  float gb[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) gb[j] = Gs[4 * j + kq];
  f32x4 sc[M2], sh[M2], g1[M2], be1[M2], b1[M2];
#pragma unroll
  for (int ml = 0; ml < M2; ++ml) {
    const int row0 = 16 * (H * M2 + ml) + 4 * kq;
    sc[ml] = *reinterpret_cast<const f32x4 *>(w + rb.ss_b + row0);
    sh[ml] = *reinterpret_cast<const f32x4 *>(w + rb.ss_b + C + row0);
    b1[ml] = *reinterpret_cast<const f32x4 *>(w + rb.c1_b + row0);
    g1[ml] = *reinterpret_cast<const f32x4 *>(w + rb.n1_w + row0);
    be1[ml] = *reinterpret_cast<const f32x4 *>(w + rb.n1_b + row0);
  }
  {
    auto ss_m = [&](auto ml_c) {
      constexpr int ml = decltype(ml_c)::value;
      const u32x4 a_sc = q.s[(P0 + ml) % kQR][0], a_sh = q.s[(P0 + ml) % kQR][1];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        sc[ml] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a_sc[j]), gb[j], sc[ml], 0, 0, 0);
        sh[ml] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a_sh[j]), gb[j], sh[ml], 0, 0, 0);
      }
      qw_retire<H, P0 + ml>(q);
    };
    ss_m(integral_constant<int, 0>{});
    if constexpr (M2 > 1) ss_m(integral_constant<int, 1>{});
  }
  // range of H (see conv_pm3_wave): a power of two per sample from a bound on |(scale + 1) GN + shift| over ALL channels
  constexpr float kR = sqrt_up(CPG * 4);
  float hb = 0.f;
#pragma unroll
  for (int ml = 0; ml < M2; ++ml)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      hb = fmaxf(hb, fmaf(__builtin_fabsf(g1[ml][r] * sc[ml][r]), kR, __builtin_fabsf(fmaf(be1[ml][r], sc[ml][r], sh[ml][r]))));
  hb = kq_max(pos_max(hb));
  hb = vmax(hb, qw_xchg1<H>(q, hb));
  int e = (int)((__float_as_uint(hb) >> 23) & 0xffu) - 127 - 14;
  e = e < 0 ? 0 : e;
  const float hinv = __uint_as_float((unsigned)(127 - e) << 23), hs = __uint_as_float((unsigned)(127 + e) << 23);
  __builtin_amdgcn_sched_barrier(0);
  // ---- conv1
  f32x4 acc[M2];
#pragma unroll
  for (int ml = 0; ml < M2; ++ml) acc[ml] = f32x4{0.f, 0.f, 0.f, 0.f};
  qw_conv3<H, P0 + M2, M2, KB>(q, xp, acc);
  f32x4 b2[M2], g2[M2], be2[M2];   // block2's parameters: in flight under block1's epilogue and conv2
#pragma unroll
  for (int ml = 0; ml < M2; ++ml) {
    const int row0 = 16 * (H * M2 + ml) + 4 * kq;
    b2[ml] = *reinterpret_cast<const f32x4 *>(w + rb.c2_b + row0);
    g2[ml] = *reinterpret_cast<const f32x4 *>(w + rb.n2_w + row0);
    be2[ml] = *reinterpret_cast<const f32x4 *>(w + rb.n2_b + row0);
  }
#pragma unroll
  for (int ml = 0; ml < M2; ++ml)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[ml][r] += b1[ml][r];
  float mean[M2], var[M2];
  qgn_stats<M2, CPG>(acc, mean, var);
#pragma unroll
  for (int ml = 0; ml < M2; ++ml) {
    const float rstd = __builtin_amdgcn_rsqf(var[ml] + 1e-5f);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float A = rstd * g1[ml][r];
      float B = be1[ml][r] - mean[ml] * A;
      B = B * sc[ml][r] + sh[ml][r];
      A = A * sc[ml][r];
      acc[ml][r] = silu(fmaf(acc[ml][r], A, B)) * hinv;
    }
  }
  u32x4 hp[KB][kSplit];
  qw_planes<H, KB>(q, acc, hp);
  // ---- conv2 on H / hs
#pragma unroll
  for (int ml = 0; ml < M2; ++ml) acc[ml] = f32x4{0.f, 0.f, 0.f, 0.f};
  qw_conv3<H, P0 + M2 + 3 * KB * M2, M2, KB>(q, hp, acc);
#pragma unroll
  for (int ml = 0; ml < M2; ++ml)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[ml][r] = fmaf(b2[ml][r], hinv, acc[ml][r]);
  qgn_stats<M2, CPG>(acc, mean, var);
#pragma unroll
  for (int ml = 0; ml < M2; ++ml) {
    const float rstd = __builtin_amdgcn_rsqf((var[ml] * hs) * hs + 1e-5f) * hs;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float A = rstd * g2[ml][r];
      xr[ml][r] += silu(fmaf(acc[ml][r], A, be2[ml][r] - mean[ml] * A));
    }
  }
  qw_planes<H, KB>(q, xr, xp);
}

// LinearAttention core of one head at 4 positions on the quad's accumulators: qa / ka / va[half] = rows 16 half + 4 kq + r
// of the head's q, k, v at the lane's column (position p, sample s).  Returns out[half][r] (resnets.py:223-235):
//   k: softmax over the sample's positions; q: softmax over the head's 32 channels, times dim_head^-0.5;
//   A[m][n] = sum_d k[d][m] q[d][n];  out[e][n] = sum_m v[e][m] A[m][n].
// Positions of the same sample sit 4 lanes apart in the row: position p - j is a row rotation by 4 j.
__device__ __forceinline__ void quad_attention_head(const f32x4 (&qa)[2], const f32x4 (&ka)[2], const f32x4 (&va)[2], f32x4 (&out)[2]) {
  float kn[2][4], qe[2][4];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float m = pos_max(ka[h][r]);
      const float ex = fast_exp(ka[h][r] - m);
      kn[h][r] = ex * __builtin_amdgcn_rcpf(pos_sum(ex));
    }
  float qm = fmaxf(fmaxf(fmaxf(qa[0][0], qa[0][1]), fmaxf(qa[0][2], qa[0][3])), fmaxf(fmaxf(qa[1][0], qa[1][1]), fmaxf(qa[1][2], qa[1][3])));
  qm = kq_max(qm);
  float qs = 0.f;
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      qe[h][r] = fast_exp(qa[h][r] - qm);
      qs += qe[h][r];
    }
  const float qscale = 0.17677669529663687f * __builtin_amdgcn_rcpf(kq_sum(qs));   // dim_head ** -0.5 / sum
  // A_j = sum_d k[d][p - j] q[d][p], j = 0..3 (this lane's channels, then the row quarters)
  float A[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float a = 0.f;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float kr = j == 0 ? kn[h][r] : (j == 1 ? dpp_mov<0x124>(kn[h][r]) : (j == 2 ? dpp_mov<0x128>(kn[h][r]) : dpp_mov<0x12C>(kn[h][r])));
        a = fmaf(kr, qe[h][r], a);
      }
    A[j] = kq_sum(a) * qscale;
  }
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float o = va[h][r] * A[0];
      o = fmaf(dpp_mov<0x124>(va[h][r]), A[1], o);
      o = fmaf(dpp_mov<0x128>(va[h][r]), A[2], o);
      o = fmaf(dpp_mov<0x12C>(va[h][r]), A[3], o);
      out[h][r] = o;
    }
}

// per-column LayerNorm statistics over all C channels from the halves' own rows (two passes, as the reference: the mean,
// then the centred squares), each pass summed with the partner
template <int H, int M2, int C>
__device__ __forceinline__ void quad_col_stats(QW<H> &q, const f32x4 (&v)[M2], float &mean, float &rstd) {
  float s = 0.f;
#pragma unroll
  for (int ml = 0; ml < M2; ++ml) s += (v[ml][0] + v[ml][1]) + (v[ml][2] + v[ml][3]);
  s = kq_sum(s);
  const float so = qw_xchg1<H>(q, s);
  mean = (H == 0 ? s + so : so + s) * (1.0f / (float)C);   // the same order in both halves: the same bits
  float d2 = 0.f;
#pragma unroll
  for (int ml = 0; ml < M2; ++ml)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float d = v[ml][r] - mean;
      d2 = fmaf(d, d, d2);
    }
  d2 = kq_sum(d2);
  const float d2o = qw_xchg1<H>(q, d2);
  rstd = __builtin_amdgcn_rsqf((H == 0 ? d2 + d2o : d2o + d2) * (1.0f / (float)C) + 1e-5f);
}

struct QLv { int qkvn_s, out_b, ln2_g; };

// Residual(PreNorm(LinearAttention)) of a 32- or 64-channel level, own position P0 (to_qkv of the two own heads | to_out
// of the own m-tiles): xr += LN(to_out(attention(to_qkv(LN(xr))))).  The PreNorm is folded into to_qkv as in qkv_att_pm
// (W' = W diag(g), s = W' 1); a head's output is the B fragment of its slice of to_out as it stands -- both halves need
// all four, so the heads' output planes are swapped.
template <int H, int P0, int MT, int KB>
__device__ __forceinline__ void quad_attention(const Ctx &c, QW<H> &q, const QLv &lv, f32x4 (&xr)[MT / 2],
                                               u32x4 (&xp)[KB][kSplit]) {
  using std::integral_constant;
  constexpr int C = 16 * MT, M2 = MT / 2;
  const int kq = c.lane >> 4;
  const float *w = c.w;
  float mean, rstd;
  quad_col_stats<H, M2, C>(q, xr, mean, rstd);
  const float mr = mean * rstd;
  f32x4 oacc[M2];
#pragma unroll
  for (int ml = 0; ml < M2; ++ml) oacc[ml] = *reinterpret_cast<const f32x4 *>(w + lv.out_b + 16 * (H * M2 + ml) + 4 * kq);
  u32x4 op[kHeads][kSplit];
  auto head = [&](auto hh_c) {
    constexpr int hh = decltype(hh_c)::value, h = 2 * H + hh, PH = P0 + hh * 6 * KB;
    f32x4 sv[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) sv[i] = *reinterpret_cast<const f32x4 *>(w + lv.qkvn_s + 16 * (2 * h + (i & 1) + 8 * (i >> 1)) + 4 * kq);
    f32x4 qkv[6];   // [part q|k|v][half]: m-tiles 2 h + half + 8 part of to_qkv
#pragma unroll
    for (int i = 0; i < 6; ++i) qkv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto mt6 = [&](auto i_c) {
      constexpr int i = decltype(i_c)::value;
      qkv[i] = qw_mfma<H, PH + i * KB>(q, xp[0], qkv[i]);
      if constexpr (KB > 1) qkv[i] = qw_mfma<H, PH + i * KB + 1>(q, xp[KB - 1], qkv[i]);
    };
    mt6(integral_constant<int, 0>{}); mt6(integral_constant<int, 1>{}); mt6(integral_constant<int, 2>{});
    mt6(integral_constant<int, 3>{}); mt6(integral_constant<int, 4>{}); mt6(integral_constant<int, 5>{});
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) qkv[i][r] = qkv[i][r] * rstd - mr * sv[i][r];
    f32x4 o[2];
    const f32x4 qa[2] = {qkv[0], qkv[1]}, ka[2] = {qkv[2], qkv[3]}, va[2] = {qkv[4], qkv[5]};
    quad_attention_head(qa, ka, va, o);
    qsplit8(o[0], o[1], op[h]);
    const unsigned m[8] = {op[h][0][0], op[h][0][1], op[h][0][2], op[h][0][3], op[h][1][0], op[h][1][1], op[h][1][2], op[h][1][3]};
    unsigned t[8];
    qw_xchg<H, 8>(q, m, t);
    constexpr int ho = 2 * (1 - H) + hh;   // the partner's head of this round
    op[ho][0] = u32x4{t[0], t[1], t[2], t[3]};
    op[ho][1] = u32x4{t[4], t[5], t[6], t[7]};
  };
  head(integral_constant<int, 0>{}); head(integral_constant<int, 1>{});
  {   // to_out: own m-tiles x the four heads
    auto om = [&](auto i_c) {
      constexpr int i = decltype(i_c)::value, ml = i / 4, h = i % 4;
      oacc[ml] = qw_mfma<H, P0 + 12 * KB + i>(q, op[h], oacc[ml]);
    };
    om(integral_constant<int, 0>{}); om(integral_constant<int, 1>{}); om(integral_constant<int, 2>{}); om(integral_constant<int, 3>{});
    if constexpr (M2 > 1) { om(integral_constant<int, 4>{}); om(integral_constant<int, 5>{}); om(integral_constant<int, 6>{}); om(integral_constant<int, 7>{}); }
  }
  // to_out's LayerNorm over the channels, residual add
  float m2, rs2;
  quad_col_stats<H, M2, C>(q, oacc, m2, rs2);
#pragma unroll
  for (int ml = 0; ml < M2; ++ml) {
    const f32x4 gv = *reinterpret_cast<const f32x4 *>(w + lv.ln2_g + 16 * (H * M2 + ml) + 4 * kq);
#pragma unroll
    for (int r = 0; r < 4; ++r) xr[ml][r] += (oacc[ml][r] - m2) * rs2 * gv[r];
  }
  qw_planes<H, KB>(q, xr, xp);
}

// ---- the 4-channel level: one value per lane, lane (column, kq) = channel kq of the column ---------------------------
struct QRb4 { int c1_w, c1_b, n1_w, n1_b, c2_w, c2_b, n2_w, n2_b, ss_w, ss_b; };

// k = 3 conv 4 -> 4 of one value per lane.  The packed f32 fragment of W is lane (kq = ci, row co) -> [tap 0..3]; output
// channel ch is wanted in row 4 ch, so lane (row i, kq) takes the fragment of row i / 4 when i % 4 == 0 and zeros otherwise.
__device__ __forceinline__ f32x4 q4_weights(const float *w, int off, int lane) {
  const int i = lane & 15, kq = lane >> 4;
  f32x4 f = *reinterpret_cast<const f32x4 *>(w + off + ((kq * 16 + (i >> 2)) * 4));
  if (i & 3) f = f32x4{0.f, 0.f, 0.f, 0.f};
  return f;
}
__device__ __forceinline__ float q4_conv(const f32x4 &wt, float bias, float x) {
  f32x4 acc = f32x4{bias, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wt[0], dpp_zero<0x114>(x), acc, 0, 0, 0);   // row_shr:4: position p - 1
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wt[1], x, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wt[2], dpp_zero<0x104>(x), acc, 0, 0, 0);   // row_shl:4: position p + 1
  return acc[0];
}
__device__ __forceinline__ float q4_gn(float y, float gamma, float beta) {   // one channel per group: over the 4 positions
  const float m = pos_sum(y) * 0.25f;
  const float d = y - m;
  const float rs = __builtin_amdgcn_rsqf(pos_sum(d * d) * 0.25f + 1e-5f);
  return d * rs * gamma + beta;
}
struct QRb4W { f32x4 w1, w2, wss; float b1, g1, be1, b2, g2, be2, ssb0, ssb1; };
__device__ __forceinline__ QRb4W quad_resblock4_load(const Ctx &c, const QRb4 &rb) {
  const int i = c.lane & 15, kq = c.lane >> 4;
  const float *w = c.w;
  QRb4W p;
  p.w1 = q4_weights(w, rb.c1_w, c.lane);
  p.w2 = q4_weights(w, rb.c2_w, c.lane);
  // scale / shift Linear [8 x 16]: scale row ch -> row 4 ch, shift row 4 + ch -> row 4 ch + 1
  const int srow = (i & 3) == 0 ? (i >> 2) : 4 + (i >> 2);
  p.wss = *reinterpret_cast<const f32x4 *>(w + rb.ss_w + ((kq * 16 + srow) * 4));
  if ((i & 3) > 1) p.wss = f32x4{0.f, 0.f, 0.f, 0.f};
  p.b1 = w[rb.c1_b + kq]; p.g1 = w[rb.n1_w + kq]; p.be1 = w[rb.n1_b + kq];
  p.b2 = w[rb.c2_b + kq]; p.g2 = w[rb.n2_w + kq]; p.be2 = w[rb.n2_b + kq];
  p.ssb0 = w[rb.ss_b + kq]; p.ssb1 = w[rb.ss_b + 4 + kq];
  return p;
}
__device__ __forceinline__ float quad_resblock4(const Ctx &c, const QRb4W &p, float x, int smp) {
  using GG = Geo<64>;
  const int kq = c.lane >> 4;
  f32x4 ss = f32x4{p.ssb0, p.ssb1, 0.f, 0.f};
  const lds_f *Gs = (const lds_f *)(c.lds + GG::kMiscG) + smp * 16;
#pragma unroll
  for (int j = 0; j < 4; ++j) ss = __builtin_amdgcn_mfma_f32_16x16x4f32(p.wss[j], Gs[4 * j + kq], ss, 0, 0, 0);
  float y = q4_conv(p.w1, p.b1, x);
  y = silu(q4_gn(y, p.g1, p.be1) * ss[0] + ss[1]);
  float z = q4_conv(p.w2, p.b2, y);
  z = silu(q4_gn(z, p.g2, p.be2));
  return x + z;
}

struct QLv4 { int qkvn_w, out_b, ln2_g; };

// attention of the 4-channel level: LayerNorm over the lane's column (the four row quarters), to_qkv as K = 4 f32 MFMAs
// (the lane's normalised value IS the B operand), the common core, to_out (its rows packed at row 4 ch) from the ring.
// x is held by BOTH halves (they run the level's ResnetBlocks twice over, bit for bit alike); each takes two heads and
// the partial to_out sums are swapped.
template <int H>
__device__ __forceinline__ float quad_attention4(const Ctx &c, QW<H> &q, const QLv4 &lv, float x) {
  using std::integral_constant;
  const int kq = c.lane >> 4;
  const float *w = c.w;
  float fq[2][6];
#pragma unroll
  for (int hh = 0; hh < 2; ++hh)
#pragma unroll
    for (int t = 0; t < 6; ++t) fq[hh][t] = w[lv.qkvn_w + ((2 * (2 * H + hh) + (t & 1) + 8 * (t >> 1)) * 64 + c.lane) * 4];
  const float outb = w[lv.out_b + kq], g2 = w[lv.ln2_g + kq];
  const float mean = kq_sum(x) * 0.25f;
  const float d = x - mean;
  const float xn = d * __builtin_amdgcn_rsqf(kq_sum(d * d) * 0.25f + 1e-5f);
  f32x4 oacc = f32x4{0.f, 0.f, 0.f, 0.f};
  auto head = [&](auto hh_c) {
    constexpr int hh = decltype(hh_c)::value;
    f32x4 qkv[6];
#pragma unroll
    for (int t = 0; t < 6; ++t) qkv[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(fq[hh][t], xn, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    f32x4 o[2];
    const f32x4 qa[2] = {qkv[0], qkv[1]}, ka[2] = {qkv[2], qkv[3]}, va[2] = {qkv[4], qkv[5]};
    quad_attention_head(qa, ka, va, o);
    u32x4 op[kSplit];
    qsplit8(o[0], o[1], op);
    oacc = qw_mfma<H, kQP0 + hh>(q, op, oacc);
  };
  head(integral_constant<int, 0>{}); head(integral_constant<int, 1>{});
  const float yo = qw_xchg1<H>(q, oacc[0]);
  const float y = outb + (H == 0 ? oacc[0] + yo : yo + oacc[0]);   // heads 0, 1 first in both halves: the same bits
  const float m2 = kq_sum(y) * 0.25f;
  const float d2 = y - m2;
  return x + d2 * __builtin_amdgcn_rsqf(kq_sum(d2 * d2) * 0.25f + 1e-5f) * g2;
}

typedef __attribute__((address_space(4))) const gldm_r1d_desc kernarg_desc;
// dk: the descriptor where it lies in the kernel-argument segment.  Every stage re-reads the few offsets it needs with
// scalar loads through a laundered copy of the pointer: read through a reference to the by-value argument they were all
// loaded at kernel entry, kept across the whole kernel and spilled (v_readlane in front of every use).
#define GLDM_QDESC() asm volatile("" : "+s"(dk))
// The chain of half H of quad `quad`.  Entry: X rows 0 .. 3 (f32, position-major columns) hold the init conv's output, G the
// embedding sums, the sync words are zero.  Exit: the 128-channel residual stream as f32 rows 0 .. 127 and X planes in the
// position-major layout.  The caller puts a barrier behind it.
template <int H>
__device__ __forceinline__ void quad_narrow_levels(const Ctx &c, kernarg_desc *dk, int quad) {
  using GG = Geo<64>;
  using std::integral_constant;
  const int col = c.lane & 15, kq = c.lane >> 4;
  const int p = col >> 2, sl = col & 3;
  const int smp = 4 * quad + sl;       // the lane's sample inside the workgroup's tile
  const int pmcol = 16 * p + smp;      // its column in the position-major layout
  const float *w = c.w;
  QW<H> q;
  q.lds = c.lds;
  q.sync = (lds_i *)(c.lds + GG::kMiscQ);
  q.qtab = (const lds_i *)(c.lds + GG::kMiscQTab);
  q.wb = reinterpret_cast<const char *>(c.w) + c.lane * 16;
  q.lane = c.lane;
  q.quad = quad;
  q.flag = 0;   // group 0: the polling path
  q.ack = 0;
  q.xseq = 0;
  q.dma_group = -1;
  q.dma_age = 0;
  q.next_load = quad + 4 * H;
  q.spins = 0;
  GLDM_QSTAMP(c, 0);
  qw_tick<H>(q, false);   // the first eight groups: one per wave (the ring's other groups follow as the ticks come)
  qw_load<H, 0>(q); qw_load<H, 1>(q); qw_load<H, 2>(q);   // own fragments 0 .. kQR - 2; then each use requests one more
  static_assert(kQR == 4, "priming loads");
#define GLDM_QRB4(i) QRb4{dk->rb[i].c1_w, dk->rb[i].c1_b, dk->rb[i].n1_w, dk->rb[i].n1_b, dk->rb[i].c2_w, dk->rb[i].c2_b, dk->rb[i].n2_w, dk->rb[i].n2_b, dk->rb[i].ss_w, dk->rb[i].ss_b}
#define GLDM_QRB(i) QRb{dk->rb[i].c1_b, dk->rb[i].n1_w, dk->rb[i].n1_b, dk->rb[i].c2_b, dk->rb[i].n2_w, dk->rb[i].n2_b, dk->rb[i].ss_b}
#define GLDM_QLV(i) QLv{dk->lv[i].qkvn_s, dk->lv[i].out_b, dk->lv[i].ln2_g}
  // ---- 4-channel level (both halves alike up to the attention)
  GLDM_QDESC();
  const QRb4W p0 = quad_resblock4_load(c, GLDM_QRB4(0)), p1 = quad_resblock4_load(c, GLDM_QRB4(1));
  float x = ((const lds_f *)(c.lds + GG::kBufX))[pswz(kq, pmcol)];
  x = quad_resblock4(c, p0, x, smp);
  GLDM_QSTAMP(c, 1);
  x = quad_resblock4(c, p1, x, smp);
  GLDM_QSTAMP(c, 2);
  GLDM_QDESC();
  x = quad_attention4<H>(c, q, QLv4{dk->lv[0].qkvn_w, dk->lv[0].out_b, dk->lv[0].ln2_g}, x);
  GLDM_QSTAMP(c, 3);
  // down conv 4 -> 32: K = 3 taps x 4 channels as three K = 4 steps of the f32 MFMA (k-step = tap, k = channel = kq);
  // the half's own output m-tile H
  f32x4 x32[1];
  u32x4 xp32[1][kSplit];
  {
    GLDM_QDESC();
    const WStream wd(w + dk->lv[0].down_w, c.lane);
    const float xl = dpp_zero<0x114>(x), xrr = dpp_zero<0x104>(x);
    const f32x4 a = wd[(size_t)H * 64];
    f32x4 acc = *reinterpret_cast<const f32x4 *>(w + dk->lv[0].down_b + 16 * H + 4 * kq);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], xl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], xrr, acc, 0, 0, 0);
    x32[0] = acc;
    qw_planes<H, 1>(q, x32, xp32);
  }
  // ---- 32-channel level
  GLDM_QSTAMP(c, 4);
  constexpr int kRb1 = qrb_len(2, 1), kAtt1 = qatt_len(2, 1);
  GLDM_QDESC();
  quad_resblock<H, kQP1, 2, 1>(c, q, GLDM_QRB(2), x32, xp32, smp);
  GLDM_QSTAMP(c, 5);
  GLDM_QDESC();
  quad_resblock<H, kQP1 + kRb1, 2, 1>(c, q, GLDM_QRB(3), x32, xp32, smp);
  GLDM_QSTAMP(c, 6);
  GLDM_QDESC();
  quad_attention<H, kQP1 + 2 * kRb1, 2, 1>(c, q, GLDM_QLV(1), x32, xp32);
  GLDM_QSTAMP(c, 7);
  f32x4 x64[2];   // own m-tiles 2 H, 2 H + 1 of the 64-channel level
  u32x4 xp64[2][kSplit];
  {
    GLDM_QDESC();
    const int down_b1 = dk->lv[1].down_b;
#pragma unroll
    for (int ml = 0; ml < 2; ++ml) x64[ml] = *reinterpret_cast<const f32x4 *>(w + down_b1 + 16 * (2 * H + ml) + 4 * kq);
    qw_conv3<H, kQP1 + 2 * kRb1 + kAtt1, 2, 1>(q, xp32, x64);
    qw_planes<H, 2>(q, x64, xp64);
  }
  // ---- 64-channel level
  GLDM_QSTAMP(c, 8);
  constexpr int kRb2 = qrb_len(4, 2), kAtt2 = qatt_len(4, 2);
  GLDM_QDESC();
  quad_resblock<H, kQP2, 4, 2>(c, q, GLDM_QRB(4), x64, xp64, smp);
  GLDM_QSTAMP(c, 9);
  GLDM_QDESC();
  quad_resblock<H, kQP2 + kRb2, 4, 2>(c, q, GLDM_QRB(5), x64, xp64, smp);
  GLDM_QSTAMP(c, 10);
  GLDM_QDESC();
  quad_attention<H, kQP2 + 2 * kRb2, 4, 2>(c, q, GLDM_QLV(2), x64, xp64);
  GLDM_QSTAMP(c, 11);
  // down conv 64 -> 128, the half's four output m-tiles: the 128-channel level's residual stream, position-major
  {
    lds_f *X3 = (lds_f *)(c.lds + GG::kBufX);
    GLDM_QDESC();
    const int down_b2 = dk->lv[2].down_b;
    f32x4 acc[4];
#pragma unroll
    for (int ml = 0; ml < 4; ++ml) acc[ml] = *reinterpret_cast<const f32x4 *>(w + down_b2 + 16 * (4 * H + ml) + 4 * kq);
    qw_conv3<H, kQP2 + 2 * kRb2 + kAtt2, 4, 2>(q, xp64, acc);
    qw_tick<H>(q, true);   // nothing of this wave's stays in flight
    __syncthreads();       // every wave is through with the ring and the mailboxes: the results go over them
#pragma unroll
    for (int ml = 0; ml < 4; ++ml) {
      const int row0 = 16 * (4 * H + ml) + 4 * kq;
#pragma unroll
      for (int r = 0; r < 4; ++r) X3[pswz(row0 + r, pmcol)] = acc[ml][r];
      store_planes4<4>(c.lds + PG<4>::kX, row0, pmcol, acc[ml][0], acc[ml][1], acc[ml][2], acc[ml][3]);
    }
  }
  static_assert(kQP2 + 2 * kRb2 + kAtt2 + qdown_len(4, 2) == kQPEnd, "stream length");
  GLDM_QSTAMP(c, 12);
#ifdef GLDM_DEBUG_KNOBS
  if (blockIdx.x == 0 && c.lane == 0) g_q_stamp[c.wave][13] = q.spins;
#endif
#undef GLDM_QDESC
#undef GLDM_QRB4
#undef GLDM_QRB
#undef GLDM_QLV
}

#endif  // GLDM_QUAD_NARROW_H_
