// quad_narrow.h -- the narrow levels (4, 32 and 64 channels) of the 4-position latent denoiser, wave-local and
// register-resident.
//
// Included by resnet1d.hip (uses its Ctx, Geo, WStream, split-f16 helpers and cross-lane reductions).
//
// Why.  On the position-major engine a narrow level is ~7 barrier-separated phases of 8 cooperating waves, each
// with a few hundred cycles of matrix work and 4-9 k cycles of latency (cold parameter loads, a cross-wave GroupNorm
// exchange, two barriers, the interpreter's set-up): ops 0-18 of the shipped denoiser took 123 k of a 260 k-cycle
// step for 16 % of its FLOPs (profiles/r05_denoise_stamps_f16_start.txt); this chain takes 72 k
// (profiles/r05_denoise_stamps_end_of_round.txt).  Nothing in those levels needs more than
// one wave: 4 samples x 4 positions are exactly one 16-column MFMA n-tile.
//
// How.  Wave q < 4 (one per SIMD) owns samples 4 q .. 4 q + 3 of the workgroup's 16 ("a quad") and walks
//   init level (4 ch):  ResnetBlock x 2, attention, down conv 4 -> 32
//   32-channel level :  ResnetBlock x 2, attention, down conv 32 -> 64
//   64-channel level :  ResnetBlock x 2, attention, down conv 64 -> 128
// on its own, with NO barrier and NO activation in LDS: every m-tile of every layer over its single n-tile (column =
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
//   * Weights stream through ONE ring of kQR fragment slots along a compile-time list of every fragment the chain
//     consumes, in consumption order (qstream_off): the moment a slot's MFMAs have issued it is refilled with the
//     fragment kQR places further down the list, whatever layer that belongs to -- no layer starts cold.
// Waves 4-7 wait at the barrier behind the chain.  The last down conv writes the 128-channel residual stream where the
// position-major engine expects it (f32 rows + X planes, column = 16 * position + sample) and the tape goes on there.
// Price: a weight fragment serves one n-tile per wave, so the four quads together draw every narrow weight four times
// from L2 / L1 (2.2 MB per tile-step) -- at these widths still cheaper than the phases it replaces.
//
// The 4-channel level runs on the f32 matrix pipe with ONE value per lane: lane (column, kq) holds channel kq of its
// column, which is exactly the B operand of v_mfma_f32_16x16x4_f32 (k = kq), the taps are DPP row shifts by 4 lanes,
// and the weight rows are gathered so that output channel ch lands in row 4 ch: register 0 of lane (column, kq = ch).
#ifndef GLDM_QUAD_NARROW_H_
#define GLDM_QUAD_NARROW_H_

#ifdef GLDM_DEBUG_KNOBS
__device__ long long g_q_stamp[4][16];   // per quad: cycle counter behind every stage of the chain (last step of workgroup 0)
#define GLDM_QSTAMP(c, i) do { if (blockIdx.x == 0 && (c).lane == 0) g_q_stamp[(c).wave & 3][i] = (long long)__builtin_readcyclecounter(); } while (0)
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
// The same two reductions over EIGHT independent values at once, in place, as 16 v_max_f32_dpp / v_add_f32_dpp in one block:
// one at a time every DPP read sits right behind the write of its operand and pays the hazard's wait states (or, from the
// intrinsics, a v_mov_b32_dpp + the op); eight deep, an instruction's operand was written eight instructions earlier and only
// the block's first needs the s_nop.  (The chain is issue bound: instructions are what it costs.)
#define GLDM_DPP8_STAGE1(op, i, j) "v_" op "_f32_dpp %" #i ", %" #j ", %" #j " row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
#define GLDM_DPP8_STAGE2(op, i) "v_" op "_f32_dpp %" #i ", %" #i ", %" #i " row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
#define GLDM_DPP8(op)                                                                                                  \
  "s_nop 1\n\t"                                                                                                         \
  GLDM_DPP8_STAGE1(op, 0, 8) GLDM_DPP8_STAGE1(op, 1, 9) GLDM_DPP8_STAGE1(op, 2, 10) GLDM_DPP8_STAGE1(op, 3, 11)          \
  GLDM_DPP8_STAGE1(op, 4, 12) GLDM_DPP8_STAGE1(op, 5, 13) GLDM_DPP8_STAGE1(op, 6, 14) GLDM_DPP8_STAGE1(op, 7, 15)        \
  GLDM_DPP8_STAGE2(op, 0) GLDM_DPP8_STAGE2(op, 1) GLDM_DPP8_STAGE2(op, 2) GLDM_DPP8_STAGE2(op, 3)                        \
  GLDM_DPP8_STAGE2(op, 4) GLDM_DPP8_STAGE2(op, 5) GLDM_DPP8_STAGE2(op, 6) GLDM_DPP8_STAGE2(op, 7)
#define GLDM_DPP8_OPERANDS                                                                                             \
  : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7])            \
  : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7])
// x[i] = max over the sample's four positions of x[i], IN PLACE -- on purpose: its callers hand it copies of accumulators.
// Nobody checks hazards inside an asm statement, and a matrix instruction's result may be read by the VALU only 7-11 wait
// states after it issued: an asm block must never take accumulators as operands as they leave the matrix pipe.  (Until round 6
// this was r[i] = max(x[i]) with the copies coalesced away: in the 4-channel level's attention the first v_max_f32_dpp read a
// k accumulator 3 wait states behind its v_mfma_f32_16x16x4_f32 -- a stale maximum now and then, which softmax's shift
// invariance turns into last-bit noise: tools/isa/dpp_hazard_scan.py, second check.)  The copy is the compiler's v_mov, placed
// behind the wait states it inserts itself.
#define GLDM_DPP8_STAGE1I(op, i) "v_" op "_f32_dpp %" #i ", %" #i ", %" #i " row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
__device__ __forceinline__ void pos_max8(float (&x)[8]) {
  asm("s_nop 1\n\t"
      GLDM_DPP8_STAGE1I("max", 0) GLDM_DPP8_STAGE1I("max", 1) GLDM_DPP8_STAGE1I("max", 2) GLDM_DPP8_STAGE1I("max", 3)
      GLDM_DPP8_STAGE1I("max", 4) GLDM_DPP8_STAGE1I("max", 5) GLDM_DPP8_STAGE1I("max", 6) GLDM_DPP8_STAGE1I("max", 7)
      GLDM_DPP8_STAGE2("max", 0) GLDM_DPP8_STAGE2("max", 1) GLDM_DPP8_STAGE2("max", 2) GLDM_DPP8_STAGE2("max", 3)
      GLDM_DPP8_STAGE2("max", 4) GLDM_DPP8_STAGE2("max", 5) GLDM_DPP8_STAGE2("max", 6) GLDM_DPP8_STAGE2("max", 7)
      : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
}
__device__ __forceinline__ void pos_sum8(const float (&x)[8], float (&r)[8]) {   // r[i] = sum over the sample's four positions of x[i]
  asm(GLDM_DPP8("add") GLDM_DPP8_OPERANDS);
}
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
// Stream sections, in consumption order.  Split fragments: halves = the hi and lo planes of fragment f (a = 2048 f,
// b = a + 1024).  Scale/shift Linear (f32 fragments, 1 KiB per m-tile): halves = the scale m-tile mi and the shift m-tile
// MT + mi.
//   conv (MT m-tiles, KB blocks): (tap step st = 3 kb + t, m-tile mi) -> fragment mi * 3 KB + t * KB + kb
//   qkv of head h: (part, half, kb) -> m-tile 2 h + half + 8 part, fragment m-tile * KB + kb
//   to_out of head h: m-tile mi -> fragment mi * 4 + h
constexpr QOff qconv_frag(int mat, int MT, int KB, int i, int mt0 = 0) {
  const int st = i / MT, mi = i % MT, kb = st / 3, t = st % 3;
  const int f = (mt0 + mi) * 3 * KB + t * KB + kb;
  return QOff{mat, 2048 * f, 2048 * f + 1024};
}
constexpr QOff qrb_frag(int mss, int MT, int KB, int i) {   // one ResnetBlock: ss (MT) | conv1 (3 KB MT) | conv2 (3 KB MT)
  if (i < MT) return QOff{mss, 1024 * i, 1024 * (MT + i)};
  i -= MT;
  if (i < 3 * KB * MT) return qconv_frag(mss + 1, MT, KB, i);
  return qconv_frag(mss + 2, MT, KB, i - 3 * KB * MT);
}
constexpr int qrb_len(int MT, int KB) { return MT + 6 * KB * MT; }
constexpr QOff qatt_frag(int mqkv, int MT, int KB, int i) {   // per head: qkv (6 KB) | to_out (MT)
  const int per = 6 * KB + MT, h = i / per, r = i % per;
  if (r < 6 * KB) {
    const int mt6 = r / KB, kb = r % KB, part = mt6 / 2, half = mt6 % 2;
    const int f = (2 * h + half + 8 * part) * KB + kb;
    return QOff{mqkv, 2048 * f, 2048 * f + 1024};
  }
  const int f = (r - 6 * KB) * 4 + h;
  return QOff{mqkv + 1, 2048 * f, 2048 * f + 1024};
}
constexpr int qatt_len(int MT, int KB) { return 4 * (6 * KB + MT); }
// level (MT, KB): rb | rb | attention | down conv to 2 MT m-tiles (in passes of DP m-tiles)
constexpr int qlevel_len(int MT, int KB) { return 2 * qrb_len(MT, KB) + qatt_len(MT, KB) + 2 * MT * 3 * KB; }
constexpr QOff qlevel_frag(int m0, int MT, int KB, int DP, int i) {
  if (i < qrb_len(MT, KB)) return qrb_frag(m0, MT, KB, i);
  i -= qrb_len(MT, KB);
  if (i < qrb_len(MT, KB)) return qrb_frag(m0 + 3, MT, KB, i);
  i -= qrb_len(MT, KB);
  if (i < qatt_len(MT, KB)) return qatt_frag(m0 + 6, MT, KB, i);
  i -= qatt_len(MT, KB);
  const int pass = i / (DP * 3 * KB);
  return qconv_frag(m0 + 8, DP, KB, i % (DP * 3 * KB), DP * pass);
}
constexpr int kQN0 = 0;                               // the 4-channel level's four to_out fragments
constexpr int kQN1 = 4;                               // the 32-channel level
constexpr int kQN2 = kQN1 + qlevel_len(2, 1);         // the 64-channel level
constexpr int kQNEnd = kQN2 + qlevel_len(4, 2);
constexpr QOff qstream_off(int n) {
  if (n < kQN1) return QOff{QM_OUT0, 2048 * n, 2048 * n + 1024};
  if (n < kQN2) return qlevel_frag(QM_SS2, 2, 1, 4, n - kQN1);
  return qlevel_frag(QM_SS4, 4, 2, 4, n - kQN2);
}

// The fragments reach the quads through LDS.  Waves 4-7 (idle otherwise) are LOADERS: loader j copies the stream's groups
// g = j, j + 4, ... (kQG fragments = 8 KiB each) from global memory into a ring of kQS 2-KiB slots with LDS-DMA
// (global_load_lds_dwordx4: a fragment plane is 64 lanes x 16 bytes, exactly one instruction, no registers), a group ahead
// of the one it is waiting for; quads read a fragment's planes from its slot a few fragments ahead of the MFMAs that use
// them (register ring of kQR).  One copy in LDS serves the four quads: every narrow weight crosses L2 -> CU once per
// tile-step, and what bounds a quad is its own arithmetic, not bytes in flight / latency (with a register ring of six
// fragments per quad straight from L2 the chain took 110 k cycles for 13 k of MFMAs: profiles/r05_quad_stamps.txt).
// Hand-shake, all in LDS: qfill[g % 8] = g + 1 once group g has landed (loader: DMA, s_waitcnt vmcnt, store);
// qdone[q] = groups quad q has finished reading (store behind the MFMAs of the group's last fragment).  A loader refills
// slot group g % 8 only when every quad is done with group g - 8.  Both words are zeroed in the step prologue.
constexpr int kQG = 4;            // fragments per group
constexpr int kQS = 32;           // LDS slots (64 KiB): 8 groups
constexpr int kQGroups = kQNEnd / kQG;
static_assert(kQNEnd % kQG == 0 && kQS % kQG == 0, "whole groups");
constexpr int kQR = 6;            // fragments a quad holds in registers ahead of their use
constexpr int kQSpinMax = 1 << 20; // every wait is bounded (a healthy one is a few polls): a lost hand-shake ends in wrong numbers, not a hang
// slots 0..15 in the H-plane region of the wide levels, 16..31 behind their X planes: neither overlaps the f32 rows
// 0 .. 127 or the X planes the chain's last conv writes while other quads are still on their way
__host__ __device__ constexpr int qslot_floats(int s) { return (s < 16 ? PG<4>::kH : PG<4>::kX + 4 * PG<4>::kBlockFloats) + (s & 15) * 512; }
static_assert(PG<4>::kH + 16 * 512 <= PG<4>::kX && PG<4>::kX + 4 * PG<4>::kBlockFloats + 16 * 512 <= 512 * 64, "slot regions");
typedef __attribute__((address_space(3))) int lds_i;
// A weight stream as the ring sees it: its length, the (matrix, byte offsets) of fragment n, the LDS slots.  QStream4: the
// chain of this file; quad16_narrow.h defines the one of the 16-position nets.
struct QStream4 {
  static constexpr int kEnd = kQNEnd, kGroups = kQGroups;
  static constexpr QOff off(int n) { return qstream_off(n); }
  __host__ __device__ static constexpr int slot_floats(int s) { return qslot_floats(s); }
};

struct QRing {
  u32x4 s[kQR][2];
  float *lds;
  lds_i *sync;       // [0..7] qfill, [16 + 64 q] qdone of quad q (every lane of the quad stores its own word: no branch)
  int lane, quad;
  int flag;          // qfill word of the NEXT group, read a group ahead (a stale "not yet" only costs the slow path)
  int spins = 0;     // diagnostic builds: polls that found the group not there yet
};
__device__ __forceinline__ int lds_poll(const lds_i *p) {
  int v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return __builtin_amdgcn_readfirstlane(v);
}
// request stream fragment n from its LDS slot into its registers.  At a group's first fragment: the group must have landed --
// checked on a word read one group earlier (straight-line code: the reads in flight stay in flight), with a polling loop
// only if that said "not yet".
template <int N, class ST = QStream4>
__device__ __forceinline__ void qring_load(QRing &ring) {
  if constexpr (N < ST::kEnd) {
    if constexpr (N % kQG == 0) {
      constexpr int g = N / kQG;
#if defined(GLDM_DEBUG_KNOBS) && defined(GLDM_QEXP_DUP)   // timing experiment (wrong results): nobody loads, nobody waits; stale slots are read
      if (false) {
#else
      if (__builtin_expect(ring.flag != g + 1, 0)) {
#endif
        for (int spin = 0; spin < kQSpinMax && lds_poll(ring.sync + (g & 7)) != g + 1; ++spin) {
          __builtin_amdgcn_s_sleep(1);
#ifdef GLDM_DEBUG_KNOBS
          ++ring.spins;
#endif
        }
      }
      if constexpr (g + 1 < ST::kGroups) ring.flag = *(volatile lds_i *)(ring.sync + ((g + 1) & 7));
    }
#if defined(GLDM_DEBUG_KNOBS) && defined(GLDM_QEXP_NOLOAD)   // timing experiment (wrong results): fragments are never read
    if (N >= kQR) return;
#endif
    const lds_u4 *slot = (const lds_u4 *)(ring.lds + ST::slot_floats(N % kQS)) + ring.lane;
    ring.s[N % kQR][0] = slot[0];
    ring.s[N % kQR][1] = slot[64];
  }
}
// acc += (stream fragment N) * B, then refill the registers; behind a group's last fragment: tell the loaders
template <int N, class ST = QStream4>
__device__ __forceinline__ f32x4 qring_mfma(QRing &ring, const u32x4 (&b)[kSplit], f32x4 acc) {
#if defined(GLDM_DEBUG_KNOBS) && defined(GLDM_QEXP_NOMFMA)   // timing experiment (wrong results): no matrix instruction
  acc[0] += __uint_as_float(ring.s[N % kQR][0][0] ^ b[0][0]);
#else
  acc = mfma_split(ring.s[N % kQR], b, acc);
#endif
  if constexpr (N % kQG == kQG - 1) ring.sync[16 + 64 * ring.quad + ring.lane] = N / kQG + 1;
  // refill the registers of the PREVIOUS fragment: this one's are still being read by the MFMAs just issued (a load into
  // them waits for the matrix pipe to have taken its operands)
  if constexpr (N >= 1) qring_load<N - 1 + kQR, ST>(ring);
  return acc;
}

// ---- loader side (waves 4-7) ---------------------------------------------------------------------------------------------
// A run-time loop over the stream table in LDS (qtab: the two byte offsets of every fragment, written once per kernel by
// quad_build_table): fully unrolled with compile-time offsets the loader was 23 k instructions of straight-line code, and
// its trip through the instruction cache every step slowed the quads beside it.
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glob_void;
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
template <class ST = QStream4>
__device__ __forceinline__ void quad_loader(const Ctx &c) {
  using GG = Geo<64>;
  lds_i *sync = (lds_i *)(c.lds + GG::kMiscQ);
  const lds_i *qtab = (const lds_i *)(c.lds + GG::kMiscQTab);
  const char *wb = reinterpret_cast<const char *>(c.w) + c.lane * 16;
  for (int g = c.wave - 4; g < ST::kGroups; g += 4) {
    if (g >= 8) {   // the slots still hold group g - 8: every quad must be done with it
      for (int spin = 0; spin < kQSpinMax; ++spin) {
        const int d0 = lds_poll(sync + 16), d1 = lds_poll(sync + 80), d2 = lds_poll(sync + 144), d3 = lds_poll(sync + 208);
        if (min(min(d0, d1), min(d2, d3)) >= g - 7) break;
        __builtin_amdgcn_s_sleep(2);
      }
    }
    const int s0 = (g & 7) * kQG;
#pragma unroll
    for (int f = 0; f < kQG; ++f) {
      const int n = g * kQG + f;
      const int oa = __builtin_amdgcn_readfirstlane(qtab[2 * n]), ob = __builtin_amdgcn_readfirstlane(qtab[2 * n + 1]);
      const int sl = s0 + f;
      float *slot = c.lds + ST::slot_floats(sl);
      __builtin_amdgcn_global_load_lds((glob_void *)(wb + oa), (lds_void *)slot, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glob_void *)(wb + ob), (lds_void *)(slot + 256), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (c.lane == 0) sync[g & 7] = g + 1;
  }
}

// A k = 3 conv over the quad's one n-tile at stream position N0: acc[mi] += W[m-tile mi, (tap, channel)] * taps(xp).
// xp[kb]: the input's fragment planes; the taps are row shifts by 4 lanes.
// SH: lanes between neighbouring positions of a sample (4: this file's columns = 4 * position + sample; 1: quad16_narrow.h)
template <int N0, int MT, int KB, int SH = 4, class ST = QStream4>
__device__ __forceinline__ void qconv3(QRing &ring, const u32x4 (&xp)[KB][kSplit], f32x4 (&acc)[MT]) {
  using std::integral_constant;
  auto body = [&](auto st_c) {
    constexpr int st = decltype(st_c)::value, kb = st / 3, t = st % 3;
    u32x4 bs[kSplit];
#pragma unroll
    for (int pl = 0; pl < kSplit; ++pl)
      bs[pl] = t == 1 ? xp[kb][pl] : (t == 0 ? dpp_zero4<0x110 + SH>(xp[kb][pl]) : dpp_zero4<0x100 + SH>(xp[kb][pl]));   // row_shr / row_shl: position p - 1 / p + 1
    auto per_m = [&](auto mi_c) {
      constexpr int mi = decltype(mi_c)::value;
      acc[mi] = qring_mfma<N0 + st * MT + mi, ST>(ring, bs, acc[mi]);
    };
    per_m(integral_constant<int, 0>{});
    if constexpr (MT > 1) per_m(integral_constant<int, 1>{});
    if constexpr (MT > 2) { per_m(integral_constant<int, 2>{}); per_m(integral_constant<int, 3>{}); }
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

// One ResnetBlock of a 32- or 64-channel level at stream position N0 (ss | conv1 | conv2):
// x += act(GN(conv2(act((scale + 1) GN(conv1(x)) + shift)))).  xr: the residual stream, xp: its fragment planes (kept
// current on exit).
template <int N0, int MT, int KB>
__device__ __forceinline__ void quad_resblock(const Ctx &c, QRing &ring, const QRb &rb, f32x4 (&xr)[MT],
                                              u32x4 (&xp)[KB][kSplit], int smp) {
  using GG = Geo<64>;
  using std::integral_constant;
  constexpr int C = 16 * MT, CPG = C / 4;
  const int kq = c.lane >> 4;
  const float *w = c.w;
  // ---- scale / shift rows of the lane's sample: [2 C x 16] Linear on the f32 matrix pipe against the embedding sums
  const lds_f *Gs = (const lds_f *)(c.lds + GG::kMiscG) + smp * 16;
  float gb[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) gb[j] = Gs[4 * j + kq];
  f32x4 sc[MT], sh[MT], g1[MT], be1[MT], b1[MT];
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) {
    const int row0 = 16 * mi + 4 * kq;
    sc[mi] = *reinterpret_cast<const f32x4 *>(w + rb.ss_b + row0);
    sh[mi] = *reinterpret_cast<const f32x4 *>(w + rb.ss_b + C + row0);
    b1[mi] = *reinterpret_cast<const f32x4 *>(w + rb.c1_b + row0);
    g1[mi] = *reinterpret_cast<const f32x4 *>(w + rb.n1_w + row0);
    be1[mi] = *reinterpret_cast<const f32x4 *>(w + rb.n1_b + row0);
  }
  {
    auto ss_m = [&](auto mi_c) {
      constexpr int mi = decltype(mi_c)::value;
      const u32x4 a_sc = ring.s[(N0 + mi) % kQR][0], a_sh = ring.s[(N0 + mi) % kQR][1];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        sc[mi] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a_sc[j]), gb[j], sc[mi], 0, 0, 0);
        sh[mi] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a_sh[j]), gb[j], sh[mi], 0, 0, 0);
      }
      if constexpr (N0 + mi >= 1) qring_load<N0 + mi - 1 + kQR>(ring);
    };
    ss_m(integral_constant<int, 0>{}); ss_m(integral_constant<int, 1>{});
    if constexpr (MT > 2) { ss_m(integral_constant<int, 2>{}); ss_m(integral_constant<int, 3>{}); }
  }
  __builtin_amdgcn_sched_barrier(0);
  // ---- conv1
  f32x4 acc[MT];
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) acc[mi] = f32x4{0.f, 0.f, 0.f, 0.f};
  qconv3<N0 + MT, MT, KB>(ring, xp, acc);
  f32x4 b2[MT], g2[MT], be2[MT];   // block2's parameters: in flight under block1's epilogue and conv2
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) {
    const int row0 = 16 * mi + 4 * kq;
    b2[mi] = *reinterpret_cast<const f32x4 *>(w + rb.c2_b + row0);
    g2[mi] = *reinterpret_cast<const f32x4 *>(w + rb.n2_w + row0);
    be2[mi] = *reinterpret_cast<const f32x4 *>(w + rb.n2_b + row0);
  }
#pragma unroll
  for (int mi = 0; mi < MT; ++mi)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[mi][r] += b1[mi][r];
  float mean[MT], var[MT];
  qgn_stats<MT, CPG>(acc, mean, var);
  // range of H (see conv_pm3_wave): a power of two per sample from a bound on |(scale + 1) GN + shift|
  constexpr float kR = sqrt_up(CPG * 4);
  float hb = 0.f;
#pragma unroll
  for (int mi = 0; mi < MT; ++mi)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      hb = fmaxf(hb, fmaf(__builtin_fabsf(g1[mi][r] * sc[mi][r]), kR, __builtin_fabsf(fmaf(be1[mi][r], sc[mi][r], sh[mi][r]))));
  hb = kq_max(pos_max(hb));
  int e = (int)((__float_as_uint(hb) >> 23) & 0xffu) - 127 - 14;
  e = e < 0 ? 0 : e;
  const float hinv = __uint_as_float((unsigned)(127 - e) << 23), hs = __uint_as_float((unsigned)(127 + e) << 23);
  u32x4 hp[KB][kSplit];
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) {
    const float rstd = __builtin_amdgcn_rsqf(var[mi] + 1e-5f);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float A = rstd * g1[mi][r];
      float B = be1[mi][r] - mean[mi] * A;
      B = B * sc[mi][r] + sh[mi][r];
      A = A * sc[mi][r];
      acc[mi][r] = silu(fmaf(acc[mi][r], A, B)) * hinv;
    }
  }
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) qsplit8(acc[2 * kb], acc[2 * kb + 1], hp[kb]);
  // ---- conv2 on H / hs
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) acc[mi] = f32x4{0.f, 0.f, 0.f, 0.f};
  qconv3<N0 + MT + 3 * KB * MT, MT, KB>(ring, hp, acc);
#pragma unroll
  for (int mi = 0; mi < MT; ++mi)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[mi][r] = fmaf(b2[mi][r], hinv, acc[mi][r]);
  qgn_stats<MT, CPG>(acc, mean, var);
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) {
    const float rstd = __builtin_amdgcn_rsqf((var[mi] * hs) * hs + 1e-5f) * hs;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float A = rstd * g2[mi][r];
      xr[mi][r] += silu(fmaf(acc[mi][r], A, be2[mi][r] - mean[mi] * A));
    }
  }
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) qsplit8(xr[2 * kb], xr[2 * kb + 1], xp[kb]);
}

// LinearAttention core of one head at 4 positions on the quad's accumulators: qa / ka / va[half] = rows 16 half + 4 kq + r
// of the head's q, k, v at the lane's column (position p, sample s).  Returns out[half][r] (resnets.py:223-235):
//   k: softmax over the sample's positions; q: softmax over the head's 32 channels, times dim_head^-0.5;
//   A[m][n] = sum_d k[d][m] q[d][n];  out[e][n] = sum_m v[e][m] A[m][n].
// Positions of the same sample sit 4 lanes apart in the row: position p - j is a row rotation by 4 j.
// acc += (a seen through a row rotation by 4 J lanes) * b as ONE v_fmac_f32_dpp (the compiler emits v_mov_b32_dpp + v_fma).
// The DPP operand must not have been written by the two instructions in front (nobody checks that inside an asm statement):
// the callers put an s_nop behind the code that produces `a`.
template <int J>
__device__ __forceinline__ float fmac_ror(float acc, float a, float b) {
  static_assert(J >= 1 && J <= 3, "rotations by 4, 8, 12 lanes");
  if constexpr (J == 1) asm("v_fmac_f32_dpp %0, %1, %2 row_ror:4 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(a), "v"(b));
  else if constexpr (J == 2) asm("v_fmac_f32_dpp %0, %1, %2 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(a), "v"(b));
  else asm("v_fmac_f32_dpp %0, %1, %2 row_ror:12 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(a), "v"(b));
  return acc;
}
// L2E: q and k arrive multiplied by log2(e) (the callers fold it into the normalisation they apply anyway)
template <bool L2E>
__device__ __forceinline__ void quad_attention_head(const f32x4 (&qa)[2], const f32x4 (&ka)[2], const f32x4 (&va)[2], f32x4 (&out)[2]) {
  constexpr float kL2e = 1.44269504088896340736f;
  float kn[2][4], qe[2][4];
  {
    float m8[8], e8[8], s8[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) m8[i] = ka[i >> 2][i & 3];
    pos_max8(m8);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float d = ka[i >> 2][i & 3] - m8[i];
      e8[i] = L2E ? __builtin_amdgcn_exp2f(d) : __builtin_amdgcn_exp2f(d * kL2e);
    }
    pos_sum8(e8, s8);
#pragma unroll
    for (int i = 0; i < 8; ++i) kn[i >> 2][i & 3] = e8[i] * __builtin_amdgcn_rcpf(s8[i]);
  }
  float qm = fmaxf(fmaxf(fmaxf(qa[0][0], qa[0][1]), fmaxf(qa[0][2], qa[0][3])), fmaxf(fmaxf(qa[1][0], qa[1][1]), fmaxf(qa[1][2], qa[1][3])));
  qm = kq_max(qm);
  float qs = 0.f;
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      qe[h][r] = L2E ? __builtin_amdgcn_exp2f(qa[h][r] - qm) : __builtin_amdgcn_exp2f((qa[h][r] - qm) * kL2e);
      qs += qe[h][r];
    }
  const float qscale = 0.17677669529663687f * __builtin_amdgcn_rcpf(kq_sum(qs));   // dim_head ** -0.5 / sum
  // A_j = sum_d k[d][p - j] q[d][p], j = 0..3 (this lane's channels, then the row quarters)
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_nop 1");   // kn is read through DPP below
  float A[4];
  {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        a0 = fmaf(kn[h][r], qe[h][r], a0);
        a1 = fmac_ror<1>(a1, kn[h][r], qe[h][r]);
        a2 = fmac_ror<2>(a2, kn[h][r], qe[h][r]);
        a3 = fmac_ror<3>(a3, kn[h][r], qe[h][r]);
      }
    A[0] = kq_sum(a0) * qscale; A[1] = kq_sum(a1) * qscale; A[2] = kq_sum(a2) * qscale; A[3] = kq_sum(a3) * qscale;
  }
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float o = va[h][r] * A[0];
      o = fmac_ror<1>(o, va[h][r], A[1]);
      o = fmac_ror<2>(o, va[h][r], A[2]);
      o = fmac_ror<3>(o, va[h][r], A[3]);
      out[h][r] = o;
    }
}

struct QLv { int qkvn_s, out_b, ln2_g; };

// Residual(PreNorm(LinearAttention)) of a 32- or 64-channel level at stream position N0 (per head: qkv | to_out):
// xr += LN(to_out(attention(to_qkv(LN(xr))))).  The PreNorm is folded into to_qkv as in qkv_att_pm (W' = W diag(g),
// s = W' 1); a head's output is the B fragment of its slice of to_out as it stands.
template <int N0, int MT, int KB>
__device__ __forceinline__ void quad_attention(const Ctx &c, QRing &ring, const QLv &lv, f32x4 (&xr)[MT],
                                               u32x4 (&xp)[KB][kSplit]) {
  using std::integral_constant;
  constexpr int C = 16 * MT, kPer = 6 * KB + MT;
  const int kq = c.lane >> 4;
  const float *w = c.w;
  // column statistics of the residual stream (LayerNorm over the channels)
  float s = 0.f;
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) s += (xr[mi][0] + xr[mi][1]) + (xr[mi][2] + xr[mi][3]);
  const float mean = kq_sum(s) * (1.0f / (float)C);
  float v = 0.f;
#pragma unroll
  for (int mi = 0; mi < MT; ++mi)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float d = xr[mi][r] - mean;
      v = fmaf(d, d, v);
    }
  const float rstd = __builtin_amdgcn_rsqf(kq_sum(v) * (1.0f / (float)C) + 1e-5f);
  // B operand of to_qkv: the normalised column itself, split here (MT values per lane, once per level).  The packer's
  // quad copy of W' = W diag(g) carries log2(e) on the q and k rows, so a head's accumulators are what its softmaxes
  // exponentiate: nothing stands between the MFMAs and the attention core (48 VALU instructions and six parameter
  // loads per head before).
  u32x4 xn[KB][kSplit];
  {
    const float mr = mean * rstd;
    f32x4 t[MT];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r) t[mi][r] = fmaf(xr[mi][r], rstd, -mr);
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) qsplit8(t[2 * kb], t[2 * kb + 1], xn[kb]);
  }
  f32x4 oacc[MT];
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) oacc[mi] = *reinterpret_cast<const f32x4 *>(w + lv.out_b + 16 * mi + 4 * kq);
  auto head = [&](auto h_c) {
    constexpr int h = decltype(h_c)::value, NH = N0 + h * kPer;
    f32x4 qkv[6];   // [part q|k|v][half]: m-tiles 2 h + half + 8 part of to_qkv
#pragma unroll
    for (int i = 0; i < 6; ++i) qkv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto mt6 = [&](auto i_c) {
      constexpr int i = decltype(i_c)::value;
      qkv[i] = qring_mfma<NH + i * KB>(ring, xn[0], qkv[i]);
      if constexpr (KB > 1) qkv[i] = qring_mfma<NH + i * KB + 1>(ring, xn[KB - 1], qkv[i]);
    };
    mt6(integral_constant<int, 0>{}); mt6(integral_constant<int, 1>{}); mt6(integral_constant<int, 2>{});
    mt6(integral_constant<int, 3>{}); mt6(integral_constant<int, 4>{}); mt6(integral_constant<int, 5>{});
    f32x4 o[2];
    const f32x4 qa[2] = {qkv[0], qkv[1]}, ka[2] = {qkv[2], qkv[3]}, va[2] = {qkv[4], qkv[5]};
    quad_attention_head<true>(qa, ka, va, o);
    u32x4 op[kSplit];
    qsplit8(o[0], o[1], op);
    auto om = [&](auto mi_c) {
      constexpr int mi = decltype(mi_c)::value;
      oacc[mi] = qring_mfma<NH + 6 * KB + mi>(ring, op, oacc[mi]);
    };
    om(integral_constant<int, 0>{}); om(integral_constant<int, 1>{});
    if constexpr (MT > 2) { om(integral_constant<int, 2>{}); om(integral_constant<int, 3>{}); }
  };
  head(integral_constant<int, 0>{}); head(integral_constant<int, 1>{}); head(integral_constant<int, 2>{}); head(integral_constant<int, 3>{});
  // to_out's LayerNorm over the channels, residual add
  float s1 = 0.f;
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) s1 += (oacc[mi][0] + oacc[mi][1]) + (oacc[mi][2] + oacc[mi][3]);
  const float m2 = kq_sum(s1) * (1.0f / (float)C);
  float v2 = 0.f;
#pragma unroll
  for (int mi = 0; mi < MT; ++mi)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float d = oacc[mi][r] - m2;
      v2 = fmaf(d, d, v2);
    }
  const float rs2 = __builtin_amdgcn_rsqf(kq_sum(v2) * (1.0f / (float)C) + 1e-5f);
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) {
    const f32x4 gv = *reinterpret_cast<const f32x4 *>(w + lv.ln2_g + 16 * mi + 4 * kq);
#pragma unroll
    for (int r = 0; r < 4; ++r) xr[mi][r] += (oacc[mi][r] - m2) * rs2 * gv[r];
  }
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) qsplit8(xr[2 * kb], xr[2 * kb + 1], xp[kb]);
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
__device__ __forceinline__ float quad_attention4(const Ctx &c, QRing &ring, const QLv4 &lv, float x) {
  using std::integral_constant;
  const int kq = c.lane >> 4;
  const float *w = c.w;
  float fq[kHeads][6];
#pragma unroll
  for (int h = 0; h < kHeads; ++h)
#pragma unroll
    for (int t = 0; t < 6; ++t)   // q and k rows times log2(e): their softmaxes exponentiate with 2^x
      fq[h][t] = w[lv.qkvn_w + ((2 * h + (t & 1) + 8 * (t >> 1)) * 64 + c.lane) * 4] * (t < 4 ? 1.44269504088896340736f : 1.0f);
  const float outb = w[lv.out_b + kq], g2 = w[lv.ln2_g + kq];
  const float mean = kq_sum(x) * 0.25f;
  const float d = x - mean;
  const float xn = d * __builtin_amdgcn_rsqf(kq_sum(d * d) * 0.25f + 1e-5f);
  f32x4 oacc = f32x4{outb, 0.f, 0.f, 0.f};
  auto head = [&](auto h_c) {
    constexpr int h = decltype(h_c)::value;
    f32x4 qkv[6];
#pragma unroll
    for (int t = 0; t < 6; ++t) qkv[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(fq[h][t], xn, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    f32x4 o[2];
    const f32x4 qa[2] = {qkv[0], qkv[1]}, ka[2] = {qkv[2], qkv[3]}, va[2] = {qkv[4], qkv[5]};
    quad_attention_head<true>(qa, ka, va, o);
    u32x4 op[kSplit];
    qsplit8(o[0], o[1], op);
    oacc = qring_mfma<kQN0 + h>(ring, op, oacc);
  };
  head(integral_constant<int, 0>{}); head(integral_constant<int, 1>{}); head(integral_constant<int, 2>{}); head(integral_constant<int, 3>{});
  const float y = oacc[0];
  const float m2 = kq_sum(y) * 0.25f;
  const float d2 = y - m2;
  return x + d2 * __builtin_amdgcn_rsqf(kq_sum(d2 * d2) * 0.25f + 1e-5f) * g2;
}

// The chain.  d: the descriptor (constant indices only).  Entry: X rows 0 .. 3 (f32, position-major columns) hold the init
// conv's output, G the embedding sums.  Exit: the 128-channel residual stream as f32 rows 0 .. 127 and X planes in the
// position-major layout.  Runs on waves 0-3; the caller puts a barrier behind it.
typedef __attribute__((address_space(4))) const gldm_r1d_desc kernarg_desc;
// dk: the descriptor where it lies in the kernel-argument segment.  Every stage re-reads the few offsets it needs with
// scalar loads through a laundered copy of the pointer: read through a reference to the by-value argument they were all
// loaded at kernel entry, kept across the whole kernel and spilled (v_readlane in front of every use).
#define GLDM_QDESC() asm volatile("" : "+s"(dk))
__device__ __forceinline__ void quad_narrow_levels(const Ctx &c, kernarg_desc *dk) {
  using GG = Geo<64>;
  using std::integral_constant;
  const int q = c.wave & 3, col = c.lane & 15, kq = c.lane >> 4;   // (wave > 3: GLDM_QEXP_DUP only)
  const int p = col >> 2, sl = col & 3;
  const int smp = 4 * q + sl;          // the lane's sample inside the workgroup's tile
  const int pmcol = 16 * p + smp;      // its column in the position-major layout
  const float *w = c.w;
  QRing ring;
  ring.lds = c.lds;
  ring.sync = (lds_i *)(c.lds + GG::kMiscQ);
  ring.lane = c.lane;
  ring.quad = q;
  ring.flag = 0;   // group 0: the polling path
  GLDM_QSTAMP(c, 0);
  qring_load<0>(ring); qring_load<1>(ring); qring_load<2>(ring); qring_load<3>(ring); qring_load<4>(ring); qring_load<5>(ring);   // fragments 0 .. kQR - 1; then the use of fragment N >= 1 requests fragment N - 1 + kQR
  static_assert(kQR == 6, "priming loads");
#define GLDM_QRB4(i) QRb4{dk->rb[i].c1_w, dk->rb[i].c1_b, dk->rb[i].n1_w, dk->rb[i].n1_b, dk->rb[i].c2_w, dk->rb[i].c2_b, dk->rb[i].n2_w, dk->rb[i].n2_b, dk->rb[i].ss_w, dk->rb[i].ss_b}
#define GLDM_QRB(i) QRb{dk->rb[i].c1_b, dk->rb[i].n1_w, dk->rb[i].n1_b, dk->rb[i].c2_b, dk->rb[i].n2_w, dk->rb[i].n2_b, dk->rb[i].ss_b}
#define GLDM_QLV(i) QLv{dk->lv[i].qkvn_s, dk->lv[i].out_b, dk->lv[i].ln2_g}
  // ---- 4-channel level
  GLDM_QDESC();
  const QRb4W p0 = quad_resblock4_load(c, GLDM_QRB4(0)), p1 = quad_resblock4_load(c, GLDM_QRB4(1));
  float x = ((const lds_f *)(c.lds + GG::kBufX))[pswz(kq, pmcol)];
  x = quad_resblock4(c, p0, x, smp);
  GLDM_QSTAMP(c, 1);
  x = quad_resblock4(c, p1, x, smp);
  GLDM_QSTAMP(c, 2);
  GLDM_QDESC();
  x = quad_attention4(c, ring, QLv4{dk->lv[0].qkvn_w, dk->lv[0].out_b, dk->lv[0].ln2_g}, x);
  GLDM_QSTAMP(c, 3);
  // down conv 4 -> 32: K = 3 taps x 4 channels as three K = 4 steps of the f32 MFMA (k-step = tap, k = channel = kq)
  f32x4 x32[2];
  u32x4 xp32[1][kSplit];
  {
    GLDM_QDESC();
    const WStream wd(w + dk->lv[0].down_w, c.lane);
    const float xl = dpp_zero<0x114>(x), xrr = dpp_zero<0x104>(x);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const f32x4 a = wd[(size_t)mi * 64];
      f32x4 acc = *reinterpret_cast<const f32x4 *>(w + dk->lv[0].down_b + 16 * mi + 4 * kq);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], xl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], xrr, acc, 0, 0, 0);
      x32[mi] = acc;
    }
    qsplit8(x32[0], x32[1], xp32[0]);
  }
  // ---- 32-channel level
  GLDM_QSTAMP(c, 4);
  constexpr int kRb1 = qrb_len(2, 1), kAtt1 = qatt_len(2, 1);
  GLDM_QDESC();
  quad_resblock<kQN1, 2, 1>(c, ring, GLDM_QRB(2), x32, xp32, smp);
  GLDM_QSTAMP(c, 5);
  GLDM_QDESC();
  quad_resblock<kQN1 + kRb1, 2, 1>(c, ring, GLDM_QRB(3), x32, xp32, smp);
  GLDM_QSTAMP(c, 6);
  GLDM_QDESC();
  quad_attention<kQN1 + 2 * kRb1, 2, 1>(c, ring, GLDM_QLV(1), x32, xp32);
  GLDM_QSTAMP(c, 7);
  f32x4 x64[4];
  u32x4 xp64[2][kSplit];
  {
    GLDM_QDESC();
    const int down_b1 = dk->lv[1].down_b;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) x64[mi] = *reinterpret_cast<const f32x4 *>(w + down_b1 + 16 * mi + 4 * kq);
    qconv3<kQN1 + 2 * kRb1 + kAtt1, 4, 1>(ring, xp32, x64);
    qsplit8(x64[0], x64[1], xp64[0]);
    qsplit8(x64[2], x64[3], xp64[1]);
  }
  // ---- 64-channel level
  GLDM_QSTAMP(c, 8);
  constexpr int kRb2 = qrb_len(4, 2), kAtt2 = qatt_len(4, 2);
  GLDM_QDESC();
  quad_resblock<kQN2, 4, 2>(c, ring, GLDM_QRB(4), x64, xp64, smp);
  GLDM_QSTAMP(c, 9);
  GLDM_QDESC();
  quad_resblock<kQN2 + kRb2, 4, 2>(c, ring, GLDM_QRB(5), x64, xp64, smp);
  GLDM_QSTAMP(c, 10);
  GLDM_QDESC();
  quad_attention<kQN2 + 2 * kRb2, 4, 2>(c, ring, GLDM_QLV(2), x64, xp64);
  GLDM_QSTAMP(c, 11);
  // down conv 64 -> 128, four m-tiles at a time: the 128-channel level's residual stream, position-major
  {
    lds_f *X3 = (lds_f *)(c.lds + GG::kBufX);
    GLDM_QDESC();
    const int down_b2 = dk->lv[2].down_b;
    auto pass = [&](auto h_c) {
      constexpr int half = decltype(h_c)::value;
      f32x4 acc[4];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) acc[mi] = *reinterpret_cast<const f32x4 *>(w + down_b2 + 16 * (4 * half + mi) + 4 * kq);
      qconv3<kQN2 + 2 * kRb2 + kAtt2 + 24 * half, 4, 2>(ring, xp64, acc);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        const int row0 = 16 * (4 * half + mi) + 4 * kq;
#pragma unroll
        for (int r = 0; r < 4; ++r) X3[pswz(row0 + r, pmcol)] = acc[mi][r];
        store_planes4<4>(c.lds + PG<4>::kX, row0, pmcol, acc[mi][0], acc[mi][1], acc[mi][2], acc[mi][3]);
      }
    };
    pass(integral_constant<int, 0>{}); pass(integral_constant<int, 1>{});
  }
  GLDM_QSTAMP(c, 12);
#ifdef GLDM_DEBUG_KNOBS
  if (blockIdx.x == 0 && c.lane == 0) g_q_stamp[c.wave & 3][13] = ring.spins;
#endif
  static_assert(kQN2 + 2 * kRb2 + kAtt2 + 48 == kQNEnd, "stream length");
#undef GLDM_QDESC
#undef GLDM_QRB4
#undef GLDM_QRB
#undef GLDM_QLV
}

#endif  // GLDM_QUAD_NARROW_H_
