// quad16_narrow.h -- the narrow levels (16, 32 and 64 channels) of the 16-position nets (pose decoder, `ppc` denoiser),
// wave-local and register-resident: the idea of quad_narrow.h where a SAMPLE is a whole n-tile.
//
// Included by resnet1d.hip behind quad_narrow.h (uses its weight ring, loader, qconv3 and split helpers).
//
// On the 16-position 64-column engine (r1d_kernel<64, 16>: 4 samples x 16 positions per workgroup) the three narrow levels
// were 24 barrier-separated ops of 8 cooperating waves: 146 k of a 296 k-cycle pass for 15 % of its FLOPs, with q | k | v of
// every attention going through LDS as f32 (profiles/r06_decode_stamps.txt).  Here wave q < 4 owns sample q and walks
//   16-channel level:  ResnetBlock x 2, attention, down conv 16 -> 32
//   32-channel level:  ResnetBlock x 2, attention, down conv 32 -> 64
//   64-channel level:  ResnetBlock x 2, attention, down conv 64 -> 128
// on its own: lane (p, g) = position p = lane & 15 of the sample, rows 4 g + r of every m-tile (the accumulator layout).
//   * B operands are made in place from the accumulators (quad column order, r1d_pack.quad_perm32); a 16-channel level is
//     one 32-channel block whose upper half is zero (the packer pads the weights: pad_cin32).
//   * A k = 3 tap is the neighbouring lane of the 16-lane row: row_shr:1 / row_shl:1 with zero fill = the conv's padding.
//   * GroupNorm: a group is CPG channels x the row's 16 lanes; LayerNorm and the query softmax reduce over the four lane
//     rows; the key softmax over the 16 lanes of a row.
//   * Attention core on the f32 matrix pipe, nothing through LDS: A = Kn^T Qn (16 x 16, K = the head's 32 channels) takes
//     the q and k accumulators as its operands as they stand (lane index = position on both sides); v is produced
//     TRANSPOSED -- v^T = xn^T Wv^T, i.e. the to_qkv MFMAs of the v rows with the operands swapped (the normalised column
//     fragment as A, the weight fragment as B) -- so that its accumulators are the A operand of out = V A, A's the B operand.
//   * scale / shift of a ResnetBlock: rows of a table -- the per-cloud one of the pose decoder (Ctx::ss_lane), or, for the
//     time-conditioned nets (the `ppc` denoiser: the rows change with the step), 448 rows per sample that ALL EIGHT waves
//     compute in front of the chain (quad16_ss_rows: the six [2 C x 64] Linears as 28 m-tiles of 16 f32 MFMAs, dealt over the
//     waves, into LDS behind the ring).  Inside the chain the same Linears cost one wave per sample 448 dependent f32 matrix
//     instructions, 11.6 k cycles a step (the SSF = 4 stream below: built first, kept as an option, not instantiated).
// Weights: the same LDS ring and loader waves as quad_narrow.h over this chain's fragment list (QStream16<SSF>: 336 / 380
// fragments).
#ifndef GLDM_QUAD16_NARROW_H_
#define GLDM_QUAD16_NARROW_H_

// matrices of the chain: per level  ss | c1 | c2 (block 0), ss | c1 | c2 (block 1), qkv, to_out, down
constexpr int kQ16Mats = 27;
// SSF: stream slots of a ResnetBlock's scale / shift Linear per m-tile.  4: the [2 C x 64] f32 fragments, slot (mi, kb) =
// k-block kb of the scale m-tile mi and of the shift m-tile MT + mi (1 KiB each).  1: the table serves the rows; the slots
// stay in the stream (never used: whole groups, one code path for the ring).
template <int SSF>
struct Q16Pos {
  static constexpr int rb(int MT, int KB) { return SSF * MT + 6 * KB * MT; }
  static constexpr int level(int MT, int KB) { return 2 * rb(MT, KB) + qatt_len(MT, KB) + 2 * MT * 3 * KB; }
  static constexpr int kN1 = level(1, 1);               // the 32-channel level
  static constexpr int kN2 = kN1 + level(2, 1);         // the 64-channel level
  static constexpr int kUsed = kN2 + level(4, 2);
  static constexpr int kEnd = (kUsed + kQG - 1) / kQG * kQG;   // whole groups: the tail repeats fragment 0 (loaded, never used)
};
template <int SSF>
constexpr QOff q16rb_frag(int mss, int MT, int KB, int i) {   // one ResnetBlock: ss (SSF MT) | conv1 (3 KB MT) | conv2 (3 KB MT)
  if (i < SSF * MT) {
    if (SSF == 1) return QOff{mss, 1024 * i, 1024 * (MT + i)};
    const int mi = i / SSF, kb = i % SSF;
    return QOff{mss, 1024 * (SSF * mi + kb), 1024 * (SSF * (MT + mi) + kb)};
  }
  i -= SSF * MT;
  if (i < 3 * KB * MT) return qconv_frag(mss + 1, MT, KB, i);
  return qconv_frag(mss + 2, MT, KB, i - 3 * KB * MT);
}
template <int SSF>
constexpr QOff q16level_frag(int m0, int MT, int KB, int DP, int i) {   // rb | rb | attention | down conv to 2 MT m-tiles
  constexpr Q16Pos<SSF> P{};
  if (i < P.rb(MT, KB)) return q16rb_frag<SSF>(m0, MT, KB, i);
  i -= P.rb(MT, KB);
  if (i < P.rb(MT, KB)) return q16rb_frag<SSF>(m0 + 3, MT, KB, i);
  i -= P.rb(MT, KB);
  if (i < qatt_len(MT, KB)) return qatt_frag(m0 + 6, MT, KB, i);
  i -= qatt_len(MT, KB);
  const int pass = i / (DP * 3 * KB);
  return qconv_frag(m0 + 8, DP, KB, i % (DP * 3 * KB), DP * pass);
}
template <int SSF>
constexpr QOff q16stream_off(int n) {
  using P = Q16Pos<SSF>;
  if (n < P::kN1) return q16level_frag<SSF>(0, 1, 1, 2, n);
  if (n < P::kN2) return q16level_frag<SSF>(9, 2, 1, 4, n - P::kN1);
  if (n < P::kUsed) return q16level_frag<SSF>(18, 4, 2, 4, n - P::kN2);
  return QOff{0, 0, 1024};
}
template <int SSF>
struct QStream16 {
  static constexpr int kEnd = Q16Pos<SSF>::kEnd, kGroups = kEnd / kQG;
  static constexpr QOff off(int n) { return q16stream_off<SSF>(n); }
  // slots 0..15 in the H-plane region, 16..31 behind the X planes: neither overlaps the f32 rows 0 .. 127 or the X planes
  // the chain's last conv writes.  (They do overwrite the zero entries at the ends of the H plane rows: the op restores them.)
  __host__ __device__ static constexpr int slot_floats(int s) {
    return (s < 16 ? PG<16>::kH : PG<16>::kX + 4 * PG<16>::kBlockFloats) + (s & 15) * 512;
  }
};
static_assert(PG<16>::kH + 16 * 512 <= PG<16>::kX && PG<16>::kX + 4 * PG<16>::kBlockFloats + 16 * 512 <= Geo<64>::kArena, "slot regions");
static_assert(2 * QStream16<4>::kEnd <= Geo<64>::kLdsFloats - Geo<64>::kMiscQTab && QStream16<1>::kEnd <= QStream16<4>::kEnd, "kMiscQTab size");

template <int SSF>
__device__ __forceinline__ void quad16_build_table(const gldm_r1d_desc &d, int *qtab, int tid, int nthreads) {
  const int mbf[kQ16Mats] = {
      d.rb[0].ss_w, d.rb[0].c1_wq, d.rb[0].c2_wq, d.rb[1].ss_w, d.rb[1].c1_wq, d.rb[1].c2_wq, d.lv[0].qkvn_wq, d.lv[0].out_wq, d.lv[0].down_wq,
      d.rb[2].ss_w, d.rb[2].c1_wq, d.rb[2].c2_wq, d.rb[3].ss_w, d.rb[3].c1_wq, d.rb[3].c2_wq, d.lv[1].qkvn_wq, d.lv[1].out_wq, d.lv[1].down_wq,
      d.rb[4].ss_w, d.rb[4].c1_wq, d.rb[4].c2_wq, d.rb[5].ss_w, d.rb[5].c1_wq, d.rb[5].c2_wq, d.lv[2].qkvn_wq, d.lv[2].out_wq, d.lv[2].down_wq};
  for (int n = tid; n < QStream16<SSF>::kEnd; n += nthreads) {
    const QOff o = q16stream_off<SSF>(n);
    int base = 0;
#pragma unroll
    for (int m = 0; m < kQ16Mats; ++m) base = o.mat == m ? mbf[m] * 4 : base;   // constant indices only (kernel argument)
    qtab[2 * n] = base + o.a;
    qtab[2 * n + 1] = base + o.b;
  }
}

// the residual stream's m-tiles -> its B fragment planes (a lone m-tile: the block's upper 16 channels are zero)
template <int MT, int KB>
__device__ __forceinline__ void q16_split(const f32x4 (&x)[MT], u32x4 (&pl)[KB][kSplit]) {
  if constexpr (MT == 1) {
    qsplit8(x[0], f32x4{0.f, 0.f, 0.f, 0.f}, pl[0]);
  } else {
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) qsplit8(x[2 * kb], x[2 * kb + 1], pl[kb]);
  }
}

// GroupNorm statistics of acc[MT] (rows 16 mi + 4 g + r at the lane's position): groups of CPG channels x 16 positions
template <int MT, int CPG>
__device__ __forceinline__ void q16_gn_stats(const f32x4 (&acc)[MT], float (&mean)[MT], float (&var)[MT]) {
  static_assert(CPG == 4 || CPG == 8 || CPG == 16, "a lane's rows, two lane rows or the whole m-tile");
  constexpr float inv_n = 1.0f / (float)(CPG * 16);
  auto over_group = [](float v) {
    v = group_sum<16>(v);
    if constexpr (CPG >= 8) v = row_pair_sum(v);
    if constexpr (CPG == 16) v = half_sum(v);
    return v;
  };
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) {
    const float m = over_group((acc[mi][0] + acc[mi][1]) + (acc[mi][2] + acc[mi][3])) * inv_n;
    float v = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float d = acc[mi][r] - m;
      v = fmaf(d, d, v);
    }
    mean[mi] = m;
    var[mi] = over_group(v) * inv_n;
  }
}

// One ResnetBlock at stream position N0 (ss | conv1 | conv2).  SSF = 1: sst = the sample's scale / shift rows of this
// block in the per-cloud table ([C] scale, [C] shift); SSF = 4: the Linear on the 64 embedding sums of sample `smp`.
template <int N0, int MT, int KB, int SSF>
__device__ __forceinline__ void quad16_resblock(const Ctx &c, QRing &ring, const QRb &rb, const float *sst, f32x4 (&xr)[MT],
                                                u32x4 (&xp)[KB][kSplit], int smp) {
  using GG = Geo<64>;
  using std::integral_constant;
  using ST = QStream16<SSF>;
  constexpr int C = 16 * MT, CPG = C / 4;
  const int kq = c.lane >> 4;
  const float *w = c.w;
  // Parameter loads are placed where their latency is covered and not earlier (this chain's 64-channel level with everything
  // requested up front, as quad_narrow.h does, spilled nine registers): block1's behind the scale / shift part, in flight under
  // conv1; block2's behind block1's epilogue, in flight under conv2.
  f32x4 sc[MT], sh[MT], g1[MT], be1[MT], b1[MT];
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) {
    const int row0 = 16 * mi + 4 * kq;
    const float *sb = SSF == 1 ? sst : w + rb.ss_b;
    sc[mi] = *reinterpret_cast<const f32x4 *>(sb + row0);
    sh[mi] = *reinterpret_cast<const f32x4 *>(sb + C + row0);
  }
  {
    // SSF = 4: B operand of k-step (kb, j) = embedding sum 16 kb + 4 j + g of the wave's sample, the same in every column
    const lds_f *Gs = (const lds_f *)(c.lds + GG::kMiscG) + smp * 64 + kq;
    auto ss_f = [&](auto i_c) {   // stream slot N0 + i
      constexpr int i = decltype(i_c)::value, mi = i / SSF, kb = i % SSF;
      if constexpr (SSF == 4) {
        const u32x4 a_sc = ring.s[(N0 + i) % kQR][0], a_sh = ring.s[(N0 + i) % kQR][1];
        float gb[4];   // read per slot (four ds_read_b32 with immediate offsets): sixteen values kept across the block were spilled
#pragma unroll
        for (int j = 0; j < 4; ++j) gb[j] = Gs[16 * kb + 4 * j];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          sc[mi] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a_sc[j]), gb[j], sc[mi], 0, 0, 0);
          sh[mi] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a_sh[j]), gb[j], sh[mi], 0, 0, 0);
        }
      }
      if constexpr (N0 + i >= 1) qring_load<N0 + i - 1 + kQR, ST>(ring);
    };
    auto ss_m = [&](auto mi_c) {
      constexpr int mi = decltype(mi_c)::value;
      ss_f(integral_constant<int, SSF * mi>{});
      if constexpr (SSF == 4) { ss_f(integral_constant<int, 4 * mi + 1>{}); ss_f(integral_constant<int, 4 * mi + 2>{}); ss_f(integral_constant<int, 4 * mi + 3>{}); }
    };
    ss_m(integral_constant<int, 0>{});
    if constexpr (MT > 1) ss_m(integral_constant<int, 1>{});
    if constexpr (MT > 2) { ss_m(integral_constant<int, 2>{}); ss_m(integral_constant<int, 3>{}); }
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) {
    const int row0 = 16 * mi + 4 * kq;
    b1[mi] = *reinterpret_cast<const f32x4 *>(w + rb.c1_b + row0);
    g1[mi] = *reinterpret_cast<const f32x4 *>(w + rb.n1_w + row0);
    be1[mi] = *reinterpret_cast<const f32x4 *>(w + rb.n1_b + row0);
  }
  // ---- conv1
  f32x4 acc[MT];
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) acc[mi] = f32x4{0.f, 0.f, 0.f, 0.f};
  qconv3<N0 + SSF * MT, MT, KB, 1, ST>(ring, xp, acc);
#pragma unroll
  for (int mi = 0; mi < MT; ++mi)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[mi][r] += b1[mi][r];
  float mean[MT], var[MT];
  q16_gn_stats<MT, CPG>(acc, mean, var);
  // range of H (conv_pm3_wave): a power of two per sample = per wave, from a bound on |(scale + 1) GN + shift|
  constexpr float kR = sqrt_up(CPG * 16);
  float hb = 0.f;
#pragma unroll
  for (int mi = 0; mi < MT; ++mi)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      hb = fmaxf(hb, fmaf(__builtin_fabsf(g1[mi][r] * sc[mi][r]), kR, __builtin_fabsf(fmaf(be1[mi][r], sc[mi][r], sh[mi][r]))));
  hb = kq_max(hb);   // rows only: every lane of a row holds the same parameters
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
  q16_split<MT, KB>(acc, hp);
  __builtin_amdgcn_sched_barrier(0);
  f32x4 b2[MT], g2[MT], be2[MT];
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) {
    const int row0 = 16 * mi + 4 * kq;
    b2[mi] = *reinterpret_cast<const f32x4 *>(w + rb.c2_b + row0);
    g2[mi] = *reinterpret_cast<const f32x4 *>(w + rb.n2_w + row0);
    be2[mi] = *reinterpret_cast<const f32x4 *>(w + rb.n2_b + row0);
  }
  // ---- conv2 on H / hs
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) acc[mi] = f32x4{0.f, 0.f, 0.f, 0.f};
  qconv3<N0 + SSF * MT + 3 * KB * MT, MT, KB, 1, ST>(ring, hp, acc);
#pragma unroll
  for (int mi = 0; mi < MT; ++mi)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[mi][r] = fmaf(b2[mi][r], hinv, acc[mi][r]);
  q16_gn_stats<MT, CPG>(acc, mean, var);
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) {
    const float rstd = __builtin_amdgcn_rsqf((var[mi] * hs) * hs + 1e-5f) * hs;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float A = rstd * g2[mi][r];
      xr[mi][r] += silu(fmaf(acc[mi][r], A, be2[mi][r] - mean[mi] * A));
    }
  }
  q16_split<MT, KB>(xr, xp);
}

// acc += A B with the operands of a weight fragment and an activation fragment SWAPPED: rows = the activation's columns
// (positions), columns = the weight's rows.  The three partial products of mfma_split, small terms first.
__device__ __forceinline__ f32x4 q16_mfma_swapped(const u32x4 (&wf)[kSplit], const u32x4 (&x)[kSplit], f32x4 acc) {
  return mfma_split(x, wf, acc);
}

// max / sum over the 16 lanes of a row of EIGHT independent values at once: four rotation stages (8, 4, 2, 1) of eight
// v_max_f32_dpp / v_add_f32_dpp each -- an instruction's operand was written eight instructions earlier, so only the block's
// first needs the s_nop of the VALU-write -> DPP-read hazard (GLDM_DPP8 in quad_narrow.h: the chain is issue bound).
// NOBODY checks hazards inside an asm statement, and a matrix instruction's result needs 7-11 wait states before a VALU
// instruction may read it: an asm block must never take accumulators as its operands as they come out of the matrix pipe
// (a build whose max block read the k accumulators directly, and whose query maximum was a v_max3_f32 in asm, returned
// stale maxima now and then -- softmax is shift invariant, so the results moved in the last bit only, from run to run:
// tests/test_r1d_gpu.py::test_decoder_more_tiles_than_slots).  row16_max8 therefore works IN PLACE on copies the compiler
// makes (its v_mov reads the accumulators behind the wait states it inserts itself); row16_sum8 reads VALU results (v_exp).
// tools/isa/dpp_hazard_scan.py checks both distances in the final ISA.
#define GLDM_R16_FIRST(op, n)                                                                                          \
  "v_" op "_f32_dpp %0, %8, %8 row_ror:" #n " row_mask:0xf bank_mask:0xf\n\t"                                          \
  "v_" op "_f32_dpp %1, %9, %9 row_ror:" #n " row_mask:0xf bank_mask:0xf\n\t"                                          \
  "v_" op "_f32_dpp %2, %10, %10 row_ror:" #n " row_mask:0xf bank_mask:0xf\n\t"                                        \
  "v_" op "_f32_dpp %3, %11, %11 row_ror:" #n " row_mask:0xf bank_mask:0xf\n\t"                                        \
  "v_" op "_f32_dpp %4, %12, %12 row_ror:" #n " row_mask:0xf bank_mask:0xf\n\t"                                        \
  "v_" op "_f32_dpp %5, %13, %13 row_ror:" #n " row_mask:0xf bank_mask:0xf\n\t"                                        \
  "v_" op "_f32_dpp %6, %14, %14 row_ror:" #n " row_mask:0xf bank_mask:0xf\n\t"                                        \
  "v_" op "_f32_dpp %7, %15, %15 row_ror:" #n " row_mask:0xf bank_mask:0xf\n\t"
#define GLDM_R16_STAGE(op, n)                                                                                          \
  "v_" op "_f32_dpp %0, %0, %0 row_ror:" #n " row_mask:0xf bank_mask:0xf\n\t"                                          \
  "v_" op "_f32_dpp %1, %1, %1 row_ror:" #n " row_mask:0xf bank_mask:0xf\n\t"                                          \
  "v_" op "_f32_dpp %2, %2, %2 row_ror:" #n " row_mask:0xf bank_mask:0xf\n\t"                                          \
  "v_" op "_f32_dpp %3, %3, %3 row_ror:" #n " row_mask:0xf bank_mask:0xf\n\t"                                          \
  "v_" op "_f32_dpp %4, %4, %4 row_ror:" #n " row_mask:0xf bank_mask:0xf\n\t"                                          \
  "v_" op "_f32_dpp %5, %5, %5 row_ror:" #n " row_mask:0xf bank_mask:0xf\n\t"                                          \
  "v_" op "_f32_dpp %6, %6, %6 row_ror:" #n " row_mask:0xf bank_mask:0xf\n\t"                                          \
  "v_" op "_f32_dpp %7, %7, %7 row_ror:" #n " row_mask:0xf bank_mask:0xf\n\t"
__device__ __forceinline__ void row16_max8(float (&x)[8]) {   // in place (see above)
  asm("s_nop 1\n\t" GLDM_R16_STAGE("max", 8) GLDM_R16_STAGE("max", 4) GLDM_R16_STAGE("max", 2) GLDM_R16_STAGE("max", 1)
      : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
}
__device__ __forceinline__ void row16_sum8(const float (&x)[8], float (&r)[8]) {   // x: VALU results
  asm("s_nop 1\n\t" GLDM_R16_FIRST("add", 8) GLDM_R16_STAGE("add", 4) GLDM_R16_STAGE("add", 2) GLDM_R16_STAGE("add", 1)
      : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7])
      : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));
}

// LinearAttention core of one head at 16 positions (resnets.py:223-235).  qa / ka[half][r]: rows 16 half + 4 g + r of the
// head's q and k (times log2 e) at the lane's position; vt[half][r] = v[e = 16 half + (lane & 15)][position 4 g + r].
// Returns out[half][r] = rows 16 half + 4 g + r of the head's output at the lane's position.
// Where a head's ~2.5 k cycles go (timing builds with parts removed, 16-channel level): the 64 DPP instructions of the key
// softmax 0.55 k (8.6 cycles each), the 16 f32 matrix instructions 0.42 k, exp / rcp 0.08 k, to_qkv / to_out / the query softmax /
// splits 1.55 k.  Measured without gain: the core software-pipelined over the heads (G0 S0 | G1 [M0 || S1] T0 | ...: the
// softmax of head h + 1 interleaved by hand with the f32 matrix instructions of head h, the weight stream in that order:
// 81 k against 80 k for the op) -- one wave per SIMD pays its instruction count, not the matrix pipe's latency.
__device__ __forceinline__ void quad16_attention_head(const f32x4 (&qa)[2], const f32x4 (&ka)[2], const f32x4 (&vt)[2], f32x4 (&out)[2]) {
  // keys: softmax over the 16 positions = the lanes of the row
  float kn[2][4];
  {
    float m8[8], e8[8], s8[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) m8[i] = ka[i >> 2][i & 3];
    row16_max8(m8);
#pragma unroll
    for (int i = 0; i < 8; ++i) e8[i] = __builtin_amdgcn_exp2f(ka[i >> 2][i & 3] - m8[i]);
    row16_sum8(e8, s8);
#pragma unroll
    for (int i = 0; i < 8; ++i) kn[i >> 2][i & 3] = e8[i] * __builtin_amdgcn_rcpf(s8[i]);
  }
  // queries: softmax over the head's 32 channels = the lane's 8 rows x the four lane rows, times dim_head ** -0.5
  float qm = fmaxf(fmaxf(fmaxf(qa[0][0], qa[0][1]), fmaxf(qa[0][2], qa[0][3])), fmaxf(fmaxf(qa[1][0], qa[1][1]), fmaxf(qa[1][2], qa[1][3])));
  qm = kq_max(qm);
  float qe[2][4], qs = 0.f;
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      qe[h][r] = __builtin_amdgcn_exp2f(qa[h][r] - qm);
      qs += qe[h][r];
    }
  const float qscale = 0.17677669529663687f * __builtin_amdgcn_rcpf(kq_sum(qs));
  // A[m][n] = sum_d kn[d][m] qe[d][n]: k-step (half, r) holds channels d = 16 half + 4 g + r on both operands
  f32x4 am = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int r = 0; r < 4; ++r) am = __builtin_amdgcn_mfma_f32_16x16x4f32(kn[h][r], qe[h][r], am, 0, 0, 0);
  // am[r] = A[key position 4 g + r][query position = lane & 15]; the query normalisation is per column
#pragma unroll
  for (int r = 0; r < 4; ++r) am[r] *= qscale;
  out[0] = f32x4{0.f, 0.f, 0.f, 0.f};
  out[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int r = 0; r < 4; ++r) {   // k-step r: key positions 4 g + r
    out[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(vt[0][r], am[r], out[0], 0, 0, 0);
    out[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(vt[1][r], am[r], out[1], 0, 0, 0);
  }
}

// Residual(PreNorm(LinearAttention)) at stream position N0 (per head: qkv | to_out), as quad_attention
template <int N0, int MT, int KB, int SSF>
__device__ __forceinline__ void quad16_attention(const Ctx &c, QRing &ring, const QLv &lv, f32x4 (&xr)[MT], u32x4 (&xp)[KB][kSplit]) {
  using std::integral_constant;
  using ST = QStream16<SSF>;
  constexpr int C = 16 * MT, kPer = 6 * KB + MT;
  const int kq = c.lane >> 4;
  const float *w = c.w;
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
  u32x4 xn[KB][kSplit];
  {
    const float mr = mean * rstd;
    f32x4 t[MT];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r) t[mi][r] = fmaf(xr[mi][r], rstd, -mr);
    q16_split<MT, KB>(t, xn);
  }
  f32x4 oacc[MT];
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) oacc[mi] = *reinterpret_cast<const f32x4 *>(w + lv.out_b + 16 * mi + 4 * kq);
  auto head = [&](auto h_c) {
    constexpr int h = decltype(h_c)::value, NH = N0 + h * kPer;
    f32x4 qkv[6];   // [part q|k|v][half]: m-tiles 2 h + half + 8 part of to_qkv; v transposed
#pragma unroll
    for (int i = 0; i < 6; ++i) qkv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto mt6 = [&](auto i_c) {
      constexpr int i = decltype(i_c)::value;
      qkv[i] = qring_mfma<NH + i * KB, ST>(ring, xn[0], qkv[i]);
      if constexpr (KB > 1) qkv[i] = qring_mfma<NH + i * KB + 1, ST>(ring, xn[KB - 1], qkv[i]);
    };
    // v^T: the same fragments with the operands swapped (the ring's refill and hand-shake as in qring_mfma)
    auto vt6 = [&](auto i_c) {
      constexpr int i = decltype(i_c)::value;
      auto step = [&](auto n_c, const u32x4 (&x)[kSplit]) {
        constexpr int N = decltype(n_c)::value;
        qkv[i] = q16_mfma_swapped(ring.s[N % kQR], x, qkv[i]);
        if constexpr (N % kQG == kQG - 1) ring.sync[16 + 64 * ring.quad + ring.lane] = N / kQG + 1;
        if constexpr (N >= 1) qring_load<N - 1 + kQR, ST>(ring);
      };
      step(integral_constant<int, NH + i * KB>{}, xn[0]);
      if constexpr (KB > 1) step(integral_constant<int, NH + i * KB + 1>{}, xn[KB - 1]);
    };
    mt6(integral_constant<int, 0>{}); mt6(integral_constant<int, 1>{}); mt6(integral_constant<int, 2>{}); mt6(integral_constant<int, 3>{});
    vt6(integral_constant<int, 4>{}); vt6(integral_constant<int, 5>{});
    f32x4 o[2];
    const f32x4 qa[2] = {qkv[0], qkv[1]}, ka[2] = {qkv[2], qkv[3]}, vt[2] = {qkv[4], qkv[5]};
    quad16_attention_head(qa, ka, vt, o);
    u32x4 op[kSplit];
    qsplit8(o[0], o[1], op);
    auto om = [&](auto mi_c) {
      constexpr int mi = decltype(mi_c)::value;
      oacc[mi] = qring_mfma<NH + 6 * KB + mi, ST>(ring, op, oacc[mi]);
    };
    om(integral_constant<int, 0>{});
    if constexpr (MT > 1) om(integral_constant<int, 1>{});
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
  q16_split<MT, KB>(xr, xp);
}

// The scale / shift rows of the six narrow ResnetBlocks for the tile's four samples, by all eight waves: LDS rows
// [sample][448] in the table's order (block i at 0, 32, 64, 128, 192, 320: [C] scale then [C] shift), = ss_b + ss_w G[sample]
// (ss_w: f32 fragments [m-tile][k-block 4][lane][4], r1d_pack.mfma_a_fragments; G: the 64 embedding sums, Geo::kMiscG).
// Columns of the MFMA = samples (column & 3); columns 0 .. 3 are stored.  The caller puts a barrier behind it.
constexpr int kQ16SsRows = 448;
constexpr int kQ16SsLds = PG<16>::kX + 4 * PG<16>::kBlockFloats + 16 * 512;   // behind the ring's last slot: 2048 floats to the arena's end
static_assert(kQ16SsLds + 4 * kQ16SsRows <= Geo<64>::kArena, "scale / shift rows behind the ring");
__device__ __forceinline__ void quad16_ss_rows(const Ctx &c, kernarg_desc *dk) {
  using GG = Geo<64>;
  const int col = c.lane & 15, kq = c.lane >> 4;
  const lds_f *Gs = (const lds_f *)(c.lds + GG::kMiscG) + (col & 3) * 64 + kq;
  lds_f *rows = (lds_f *)(c.lds + kQ16SsLds) + (col & 3) * kQ16SsRows;
  const int ssw[6] = {dk->rb[0].ss_w, dk->rb[1].ss_w, dk->rb[2].ss_w, dk->rb[3].ss_w, dk->rb[4].ss_w, dk->rb[5].ss_w};
  const int ssb[6] = {dk->rb[0].ss_b, dk->rb[1].ss_b, dk->rb[2].ss_b, dk->rb[3].ss_b, dk->rb[4].ss_b, dk->rb[5].ss_b};
  float gb[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) gb[i] = Gs[4 * i];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int t = c.wave + 8 * k;   // m-tile 0 .. 27 of the six blocks' 2 + 2 + 4 + 4 + 8 + 8 (wave uniform)
    if (t < 28) {
      // block, its first m-tile and table offset: constant indices only (kernel argument)
      int w_off = ssw[0], b_off = ssb[0], mt = t, tab = 0;
      if (t >= 2) { w_off = ssw[1]; b_off = ssb[1]; mt = t - 2; tab = 32; }
      if (t >= 4) { w_off = ssw[2]; b_off = ssb[2]; mt = t - 4; tab = 64; }
      if (t >= 8) { w_off = ssw[3]; b_off = ssb[3]; mt = t - 8; tab = 128; }
      if (t >= 12) { w_off = ssw[4]; b_off = ssb[4]; mt = t - 12; tab = 192; }
      if (t >= 20) { w_off = ssw[5]; b_off = ssb[5]; mt = t - 20; tab = 320; }
      const WStream wv(c.w + w_off, c.lane);
      f32x4 a[4];
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) a[kb] = wv[(size_t)(mt * 4 + kb) * 64];
      f32x4 acc = *reinterpret_cast<const f32x4 *>(c.w + b_off + 16 * mt + 4 * kq);
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kb][j], gb[4 * kb + j], acc, 0, 0, 0);
      if (col < 4) {
#pragma unroll
        for (int r = 0; r < 4; ++r) rows[tab + 16 * mt + 4 * kq + r] = acc[r];
      }
    }
  }
}

// The chain.  Entry: X rows 0 .. 15 (f32, column = 4 * position + sample) hold the init conv's output, G the embedding
// sums.  Exit: the 128-channel residual stream as f32 rows 0 .. 127 and X planes in that layout.  Runs on waves 0-3 (waves
// 4-7: quad_loader<QStream16>); the caller puts a barrier behind it and restores the H plane rows' zero entries.
// sstab: the wave's sample's row of the per-cloud scale / shift table, or null.
#define GLDM_QDESC() asm volatile("" : "+s"(dk))
template <int SSF>
__device__ __forceinline__ void quad16_narrow_levels(const Ctx &c, kernarg_desc *dk, const float *sstab) {
  using GG = Geo<64>;
  using std::integral_constant;
  using ST = QStream16<SSF>;
  using P = Q16Pos<SSF>;
  const int q = c.wave & 3, p = c.lane & 15, kq = c.lane >> 4;
  const int pmcol = 4 * p + q;     // the lane's column in the engine's layout
  const float *w = c.w;
  QRing ring;
  ring.lds = c.lds;
  ring.sync = (lds_i *)(c.lds + GG::kMiscQ);
  ring.lane = c.lane;
  ring.quad = q;
  ring.flag = 0;
  GLDM_QSTAMP(c, 0);
  qring_load<0, ST>(ring); qring_load<1, ST>(ring); qring_load<2, ST>(ring); qring_load<3, ST>(ring); qring_load<4, ST>(ring); qring_load<5, ST>(ring);
  static_assert(kQR == 6, "priming loads");
#define GLDM_QRB(i) QRb{dk->rb[i].c1_b, dk->rb[i].n1_w, dk->rb[i].n1_b, dk->rb[i].c2_b, dk->rb[i].n2_w, dk->rb[i].n2_b, dk->rb[i].ss_b}
#define GLDM_QLV(i) QLv{dk->lv[i].qkvn_s, dk->lv[i].out_b, dk->lv[i].ln2_g}
  // table rows of ResnetBlock i: 2 C per block, blocks in order (build_tape's tab_off)
  auto sst = [&](int off) { return SSF == 1 ? sstab + off : nullptr; };
  const lds_f *X3r = (const lds_f *)(c.lds + GG::kBufX);
  // ---- 16-channel level
  f32x4 x16[1];
  u32x4 xp16[1][kSplit];
#pragma unroll
  for (int r = 0; r < 4; ++r) x16[0][r] = X3r[pswz(4 * kq + r, pmcol)];
  q16_split<1, 1>(x16, xp16);
  constexpr int kRb0 = P::rb(1, 1), kAtt0 = qatt_len(1, 1);
  GLDM_QDESC();
  quad16_resblock<0, 1, 1, SSF>(c, ring, GLDM_QRB(0), sst(0), x16, xp16, q);
  GLDM_QSTAMP(c, 1);
  GLDM_QDESC();
  quad16_resblock<kRb0, 1, 1, SSF>(c, ring, GLDM_QRB(1), sst(32), x16, xp16, q);
  GLDM_QSTAMP(c, 2);
  GLDM_QDESC();
  quad16_attention<2 * kRb0, 1, 1, SSF>(c, ring, GLDM_QLV(0), x16, xp16);
  GLDM_QSTAMP(c, 3);
  f32x4 x32[2];
  u32x4 xp32[1][kSplit];
  {
    GLDM_QDESC();
    const int down_b0 = dk->lv[0].down_b;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) x32[mi] = *reinterpret_cast<const f32x4 *>(w + down_b0 + 16 * mi + 4 * kq);
    qconv3<2 * kRb0 + kAtt0, 2, 1, 1, ST>(ring, xp16, x32);
    q16_split<2, 1>(x32, xp32);
  }
  // ---- 32-channel level
  GLDM_QSTAMP(c, 4);
  constexpr int kRb1 = P::rb(2, 1), kAtt1 = qatt_len(2, 1);
  GLDM_QDESC();
  quad16_resblock<P::kN1, 2, 1, SSF>(c, ring, GLDM_QRB(2), sst(64), x32, xp32, q);
  GLDM_QSTAMP(c, 5);
  GLDM_QDESC();
  quad16_resblock<P::kN1 + kRb1, 2, 1, SSF>(c, ring, GLDM_QRB(3), sst(128), x32, xp32, q);
  GLDM_QSTAMP(c, 6);
  GLDM_QDESC();
  quad16_attention<P::kN1 + 2 * kRb1, 2, 1, SSF>(c, ring, GLDM_QLV(1), x32, xp32);
  GLDM_QSTAMP(c, 7);
  f32x4 x64[4];
  u32x4 xp64[2][kSplit];
  {
    GLDM_QDESC();
    const int down_b1 = dk->lv[1].down_b;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) x64[mi] = *reinterpret_cast<const f32x4 *>(w + down_b1 + 16 * mi + 4 * kq);
    qconv3<P::kN1 + 2 * kRb1 + kAtt1, 4, 1, 1, ST>(ring, xp32, x64);
    q16_split<4, 2>(x64, xp64);
  }
  // ---- 64-channel level
  GLDM_QSTAMP(c, 8);
  constexpr int kRb2 = P::rb(4, 2), kAtt2 = qatt_len(4, 2);
  GLDM_QDESC();
  quad16_resblock<P::kN2, 4, 2, SSF>(c, ring, GLDM_QRB(4), sst(192), x64, xp64, q);
  GLDM_QSTAMP(c, 9);
  GLDM_QDESC();
  quad16_resblock<P::kN2 + kRb2, 4, 2, SSF>(c, ring, GLDM_QRB(5), sst(320), x64, xp64, q);
  GLDM_QSTAMP(c, 10);
  GLDM_QDESC();
  quad16_attention<P::kN2 + 2 * kRb2, 4, 2, SSF>(c, ring, GLDM_QLV(2), x64, xp64);
  GLDM_QSTAMP(c, 11);
  // down conv 64 -> 128, four m-tiles at a time: the 128-channel level's residual stream
  {
    lds_f *X3 = (lds_f *)(c.lds + GG::kBufX);
    GLDM_QDESC();
    const int down_b2 = dk->lv[2].down_b;
    auto pass = [&](auto h_c) {
      constexpr int half = decltype(h_c)::value;
      f32x4 acc[4];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) acc[mi] = *reinterpret_cast<const f32x4 *>(w + down_b2 + 16 * (4 * half + mi) + 4 * kq);
      qconv3<P::kN2 + 2 * kRb2 + kAtt2 + 24 * half, 4, 2, 1, ST>(ring, xp64, acc);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        const int row0 = 16 * (4 * half + mi) + 4 * kq;
#pragma unroll
        for (int r = 0; r < 4; ++r) X3[pswz(row0 + r, pmcol)] = acc[mi][r];
        store_planes4<16>(c.lds + PG<16>::kX, row0, pmcol, acc[mi][0], acc[mi][1], acc[mi][2], acc[mi][3]);
      }
    };
    pass(integral_constant<int, 0>{}); pass(integral_constant<int, 1>{});
  }
  GLDM_QSTAMP(c, 12);
#ifdef GLDM_DEBUG_KNOBS
  if (blockIdx.x == 0 && c.lane == 0) g_q_stamp[c.wave & 3][13] = ring.spins;
#endif
  static_assert(P::kN2 + 2 * kRb2 + kAtt2 + 48 == P::kUsed, "stream length");
#undef GLDM_QRB
#undef GLDM_QLV
}
#undef GLDM_QDESC

#endif  // GLDM_QUAD16_NARROW_H_
