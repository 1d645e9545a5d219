/*
 * gldm.h -- C ABI of libgldm_hip.so: the MI355X (gfx950) implementation of
 * GraspLDM's grasp-generation hot path.
 *
 * This is the drop-in boundary.  Every entry point takes
 * raw DEVICE pointers, plain sizes and an explicit HIP stream; outputs are
 * caller-allocated; nothing here allocates, frees, synchronises or exits the
 * process.  Return value: 0 = launched, negative = gldm_status (below).  All
 * tensors are dense, row-major, f32 / int32 exactly as in the reference:
 * coords [B,3,N], features [B,C,N], indices int32.
 *
 * "ref:" lines cite the reference interface each function replaces, relative
 * to /root/reference/grasp_ldm/models/modules/ext/pvcnn/modules/functional/src/
 * unless they start with grasp_ldm/ or tools/.
 */
#ifndef GLDM_H_
#define GLDM_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *gldm_stream_t; /* hipStream_t; NULL = the null stream */

enum gldm_status {
  GLDM_OK = 0,
  GLDM_ERR_INVALID_ARG = -1, /* null pointer, non-positive size, unsupported shape */
  GLDM_ERR_LAUNCH = -2,      /* hipGetLastError() after the launch was not hipSuccess */
  GLDM_ERR_UNSUPPORTED = -3, /* shape outside what the kernels were built for */
  GLDM_ERR_WORKSPACE = -4    /* caller workspace too small */
};

/* ABI version (bumped on any signature change) and a static message per status. */
int gldm_abi_version(void);
const char *gldm_status_string(int status);

/* ---------------------------------------------------------------- point ops */

/* ref: ball_query/ball_query.hpp:6-8, ball_query.cu:19-59 (pybind `ball_query`).
 * First `u` points (ascending index) with |p - c|^2 < radius^2 (strict, f32, no
 * FMA contraction); slots beyond the hit count repeat the first hit; all zero
 * when the ball is empty.  Writes every slot of out[b,m,u]. */
int gldm_ball_query(const float *centers /*[b,3,m]*/, const float *points /*[b,3,n]*/,
                    int b, int n, int m, float radius, int u,
                    int32_t *out /*[b,m,u]*/, gldm_stream_t stream);

/* ref: grouping/grouping.hpp:6, grouping.cu:18-44 (pybind `grouping_forward`).
 * out[b,c,j,k] = features[b,c,idx[b,j,k]]. */
int gldm_grouping_forward(const float *features /*[b,c,n]*/, const int32_t *idx /*[b,m,u]*/,
                          int b, int c, int n, int m, int u,
                          float *out /*[b,c,m,u]*/, gldm_stream_t stream);

/* ref: sampling/sampling.hpp:6, sampling.cu:17-39 (pybind `gather_features_forward`).
 * out[b,c,j] = features[b,c,idx[b,j]]. */
int gldm_gather_features_forward(const float *features /*[b,c,n]*/, const int32_t *idx /*[b,m]*/,
                                 int b, int c, int n, int m,
                                 float *out /*[b,c,m]*/, gldm_stream_t stream);

/* ref: sampling/sampling.hpp:10, sampling.cpp:43-58, sampling.cu:86-174
 * (pybind `furthest_point_sampling`).  Iterative FPS from index 0; tie rule of
 * the reference's 512-thread tree (max distance, then min (k mod 512), then
 * min k).  Distances live on chip; no scratch buffer.  n <= 8192. */
int gldm_furthest_point_sampling(const float *coords /*[b,3,n]*/, int b, int n, int m,
                                 int32_t *out_idx /*[b,m]*/, gldm_stream_t stream);

/* ref: interpolate/neighbor_interpolate.hpp:7-10, neighbor_interpolate.cu:20-131
 * (pybind `three_nearest_neighbors_interpolate_forward`). */
int gldm_three_nn_interpolate_forward(const float *points /*[b,3,n]*/, const float *centers /*[b,3,m]*/,
                                      const float *center_features /*[b,c,m]*/,
                                      int b, int c, int m, int n,
                                      float *out /*[b,c,n]*/, int32_t *idx /*[b,3,n]*/, float *wgt /*[b,3,n]*/,
                                      gldm_stream_t stream);

/* ref: voxelization/vox.hpp:7-9, vox.cpp:17-43, vox.cu:18-72,112-119
 * (pybind `avg_voxelize_forward`).  Deterministic: each voxel's mean is summed
 * in ascending point index (the reference uses f32 atomics in arbitrary order).
 * Writes all of out/ind/cnt (no pre-zeroing needed).  n <= 8192, r <= 64. */
int gldm_avg_voxelize_forward(const float *features /*[b,c,n]*/, const int32_t *vox_coords /*[b,3,n]*/,
                              int b, int c, int n, int r,
                              float *out /*[b,c,r^3]*/, int32_t *ind /*[b,n]*/, int32_t *cnt /*[b,r^3]*/,
                              gldm_stream_t stream);

/* ref: interpolate/trilinear_devox.hpp:7-10, trilinear_devox.cpp:18-55,
 * trilinear_devox.cu:21-105 (pybind `trilinear_devoxelize_forward`).
 * inds/wgts [b,8,n] are written only when is_training != 0 (may be NULL otherwise). */
int gldm_trilinear_devoxelize_forward(const float *coords /*[b,3,n]*/, const float *features /*[b,c,r^3]*/,
                                      int b, int c, int n, int r, int is_training,
                                      float *out /*[b,c,n]*/, int32_t *inds, float *wgts,
                                      gldm_stream_t stream);

/* ref: grasp_ldm/models/modules/ext/pvcnn/modules/voxelization.py:16-35
 * (Voxelization.forward before F.avg_voxelize): mean-centre, scale to [0,r-1]
 * (normalize != 0: divide by 2*max|p| + eps and add 0.5; else (p+1)/2), clamp,
 * round-half-even.  The per-axis mean is accumulated in f64 in a fixed tree. */
int gldm_voxel_coords(const float *coords /*[b,3,n]*/, int b, int n, int r, int normalize, float eps,
                      float *norm_coords /*[b,3,n]*/, int32_t *vox_coords /*[b,3,n]*/,
                      gldm_stream_t stream);

/* ------------------------------------------------ raw-cloud front end */

/* ref: grasp_ldm/utils/pointcloud_helpers.py:160-217 (PointCloudHelpers.farthest_points with
 * distance_by_translation_point :219-223, as called by regularize_pc_point_count :124-158): greedy
 * farthest-point selection on a sensor-layout cloud [b,n,3].  First centre = index 0 (np.argmax of
 * the all-1e7 start vector), distance = sqrt((dx^2 + dy^2) + dz^2) in f32, ties -> lowest index.
 * m <= n <= 8192.  Bit-identical indices to the numpy routine on f32 input. */
int gldm_farthest_points_euclid(const float *points /*[b,n,3]*/, int b, int n, int m,
                                int32_t *out_idx /*[b,m]*/, gldm_stream_t stream);

/* ref: tools/inference.py:570-591 (InferenceLDM.normalize_input) and
 * grasp_ldm/inference/inference_base.py:181-212: pc_out = ((pc - mean_points(pc)) - shift) / scale
 * per axis (shift = _INPUT_PC_SHIFT, scale = _INPUT_PC_SCALE of set_normalization_params :103-130),
 * mean_out[b,:] = per-cloud mean (f64 accumulation, rounded once). */
int gldm_normalize_cloud(const float *pc /*[b,n,3]*/, int b, int n, float shift_x, float shift_y, float shift_z,
                         float scale_x, float scale_y, float scale_z,
                         float *pc_out /*[b,n,3]*/, float *mean_out /*[b,3]*/, gldm_stream_t stream);

/* out[b,j,:] = pc[b,idx[b,j],:]  (row gather behind every point-count regularisation:
 * pointcloud_helpers.py:40-71,124-158). */
int gldm_gather_points(const float *pc /*[b,n,3]*/, const int32_t *idx /*[b,m]*/, int b, int n, int m,
                       float *out /*[b,m,3]*/, gldm_stream_t stream);

/* ------------------------------------------------------- set abstraction */

/* ref: grasp_ldm/models/modules/ext/pvcnn/modules/ball_query.py:16-34
 * (BallQuery.forward = ball_query + grouping(coords) - centre + grouping(features)
 * + concat) as ONE kernel: the "set-abstraction gather".  features may be NULL
 * (c = 0).  out[b, 0:3, j, k] = p[idx] - centre_j ; out[b, 3:3+c, j, k] = f[idx].
 * idx_out [b,m,u] is optional (NULL to skip). */
int gldm_sa_group(const float *points /*[b,3,n]*/, const float *centers /*[b,3,m]*/,
                  const float *features /*[b,c,n] or NULL*/,
                  int b, int c, int n, int m, float radius, int u,
                  float *out /*[b,3+c,m,u]*/, int32_t *idx_out, gldm_stream_t stream);


/* ------------------------------------- 1-D ResNet engine (denoiser, decoder) */

/* Architecture descriptor of one ResNet1D / TimeConditionedResNet1D instance
 * (ref: grasp_ldm/models/modules/resnets.py:263-424,427-616) plus offsets, in
 * floats, into ONE packed weight buffer prepared on the host at load time
 * (graspldm_amd/r1d_pack.py): weight-standardised conv weights, MFMA
 * A-fragment order, combined scale/shift biases.  Plain C struct, filled by the
 * host side; the kernels never parse checkpoints. */
#define GLDM_R1D_MAX_LEVELS 6
#define GLDM_R1D_MAX_RESBLOCKS (2 * GLDM_R1D_MAX_LEVELS + 1)

typedef struct gldm_r1d_resblock {
  int32_t c1_w, c1_b;   /* block1.proj  (packed A fragments), bias [C]            */
  int32_t n1_w, n1_b;   /* block1.norm  gamma/beta [C]                            */
  int32_t c2_w, c2_b;   /* block2.proj                                            */
  int32_t n2_w, n2_b;   /* block2.norm                                            */
  int32_t ss_w, ss_b;   /* mlp.1 as a [2C x E] GEMM (packed A fragments) and the
                           combined bias R*b (+R on the C scale rows): [2C]       */
  int32_t c1_w3, c2_w3; /* ABI 5: the two conv weights again as split-f16 fragments (see below);
                           0 = absent                                              */
  int32_t c1_wq, c2_wq; /* ABI 9: the same fragments with the columns of every 32-channel block in
                           "quad" order (see below); 0 = absent                    */
} gldm_r1d_resblock;

typedef struct gldm_r1d_level {
  int32_t ln_g;         /* PreNorm LayerNorm gain [C]                             */
  int32_t qkv_w[2];     /* to_qkv rows for heads {0,1} and {2,3}: 192 x C packed  */
  int32_t out_w, out_b; /* to_out.0 [C x 128] packed, bias [C]                    */
  int32_t ln2_g;        /* to_out.1 LayerNorm gain [C]                            */
  int32_t down_w, down_b; /* Conv1d(C -> C', k=3) packed, bias [C']               */
  int32_t qkvn_w;       /* ABI 4: to_qkv with the PreNorm gain folded in, W' = W diag(g), in to_qkv's own
                           row order (q | k | v, head h at rows 32 h of each third): 384 x C packed;
                           0 / negative = absent (the position-major engine is then not used)       */
  int32_t qkvn_s;       /* ABI 4: row sums of W' [384] (the mean term of the folded LayerNorm)     */
  int32_t qkvn_w3, out_w3, down_w3; /* ABI 5: qkvn_w / out_w / down_w as split-f16 fragments; 0 = absent */
  int32_t qkvn_wq, out_wq, down_wq; /* ABI 9: the same in quad column order; out_wq of a 4-channel level has
                                       channel ch in row 4 ch of its one m-tile; qkvn_wq = W' = W diag(g) with
                                       its q and k rows (the first 256) times log2(e): the engine multiplies it
                                       with the normalised column and exponentiates with 2^x; 0 = absent   */
} gldm_r1d_level;

/* Quad column order (ABI 9; r1d_pack.quad_perm32): the wave-local engine of the narrow levels (csrc/quad_narrow.h) feeds
 * a GEMM's B operand straight from the accumulator layout of the previous one -- lane (column, g) holds rows 4 g + r of
 * m-tiles 2 kb and 2 kb + 1 -- so k-slot 8 g + j of a 32-channel block kb stands for channel 32 kb + 16 (j >> 2) + 4 g +
 * (j & 3), and the weights' columns are stored in that order: W_q[:, 32 kb + 8 g + j] = W[:, 32 kb + 16 (j >> 2) + 4 g + (j & 3)].
 * 16-position nets (round 6, csrc/quad16_narrow.h: the 16 / 32 / 64-channel levels in front of a 128-channel one; one wave =
 * one sample's 16 positions): the same copies, taken from the weights with every tap's channels padded to whole 32-channel
 * blocks first (a 16-channel level is one block whose k-slots j >= 4 are zero): quad_perm32(pad_cin32(W, C, taps)).  Packers
 * that leave them 0 get the barrier-separated phases of the 64-column engine instead: same results, ~20 % slower. */

/* Split-f16 weight fragments (layout since ABI 5, two f16 planes since ABI 9; graspldm_amd/r1d_pack.py:
 * mfma_a_fragments_f16x2).  Every f32 weight is written as hi + lo, two f16 numbers (hi = f16(w), lo = f16(w - hi):
 * 2 x 11 significant bits, |w - hi - lo| <= 2^-22 |w|; |w| < 65504 or the packer raises); the matrix [M, K]
 * (K % 32 == 0) is stored as [M/16][K/32][plane hi|lo][lane 64][8 f16] = 2 KiB per fragment, lane l holding
 * W[16 mt + (l & 15)][32 kb + 8 (l >> 4) + j], j = 0..7: the A operand of v_mfma_f32_16x16x32_f16.  The 64-column
 * engines compute every f32 product as the three partial products hi*hi + hi*lo + lo*hi on the f16 matrix pipe with f32
 * accumulation (the pipe keeps f16 subnormals); the dropped lo*lo term is <= 2^-22 of |a||b| per product, the order of an
 * f32 rounding, at 3/16 of the f32-MFMA time.  (ABI 5-8 stored three bf16 planes, hi|mid|lo, for six products.) */

typedef struct gldm_r1d_desc {
  int32_t seq_len;      /* L: 4 (latent denoiser) or 16 (pose decoder)            */
  int32_t n_levels;
  int32_t dims[GLDM_R1D_MAX_LEVELS + 1]; /* channel widths; dims[0] = init_dim    */
  int32_t emb_dim;      /* E = 4 * dim                                            */
  int32_t cond_rows;    /* R: rows of the conditioning latent (3), 1 if 2-D       */
  int32_t groups;       /* GroupNorm groups (resnet_block_groups)                 */
  int32_t init_w, init_b; /* init_conv [C0][7], [C0]                              */
  int32_t ss_rows;      /* 2 * max(dims) (unused since ABI 2: the scale/shift rows are
                           computed in the conv epilogue; kept for layout)        */
  gldm_r1d_resblock rb[GLDM_R1D_MAX_RESBLOCKS]; /* 2 per level, then final        */
  gldm_r1d_level lv[GLDM_R1D_MAX_LEVELS];
  int32_t final_w, final_b; /* final_conv [dims[n_levels]], [1]                   */
  /* decoder only (ref: grasp_ldm/models/grasp_vae.py:358-436) */
  int32_t latent_dim;   /* D of z_h; 0 for the denoiser                           */
  int32_t in_w, in_b;   /* in_layer [L][D], [L]                                   */
  int32_t head_w, head_b; /* rows tmrp(6) then class_logits(1): [7][L], [7]       */
  int32_t n_head;       /* 7                                                      */
} gldm_r1d_desc;

enum gldm_sched_kind { GLDM_SCHED_NONE = 0, GLDM_SCHED_DDIM = 1, GLDM_SCHED_DDPM = 2, GLDM_SCHED_DPMPP = 3 };
#define GLDM_SCHED_COEF_STRIDE 8
/* per-step coefficient row (f32, the table 16-byte aligned; computed on the host exactly like the
 * scheduler library does on 0-dim f32 tensors):
 *   [0] sqrt(1-abar_t) [1] sqrt(abar_t)
 *   DDIM: [2] sqrt(abar_prev) [3] sqrt(1-abar_prev-sigma^2)
 *   DDPM: [4] coef_x0 [5] coef_xt [6] sqrt(variance) [7] 1 if noise is added (t>0)
 * GLDM_SCHED_DPMPP (ref: grasp_ldm/models/diffusion/elucidated_diffusion.py:259-313, DPM-Solver++(2M) of
 * ElucidatedDiffusion; the network sees c_in x and time = c_noise(sigma), so `temb` holds one row per STEP and
 * timesteps = 0..n_steps-1; clip_sample = the sampler's `clamp`):
 *   [0] c_in [1] c_skip [2] c_out [3] 1-gamma [4] gamma [5] sigma_fn(t_next)/sigma_fn(t) [6] expm1(-h)
 *   [7] 1 if the previous step's denoised row is blended in (not on the first step, not when sigma_next = 0) */

/* ref: resnets.py:484-494,587-594 (input_emb_layers = Linear + SiLU on z_cond).
 * cemb[i,r,:] = silu(W z_cond[i,r,:] + b). */
int gldm_r1d_cond_embed(const float *z_cond /*[n_cond,R,Dc]*/, const float *w /*[E,Dc]*/, const float *b /*[E]*/,
                        int n_cond, int rows, int dc, int e, float *cemb /*[n_cond,R,E]*/, gldm_stream_t stream);

/* Bytes of workspace gldm_denoise / gldm_decode need for n_samples (-1 if the descriptor is not
 * supported: groups must be 4, widths powers of two in {4, 16, 32, 64, 128, 256}, at most 128 on levels
 * with attention, emb_dim % 16 == 0).  A step stays on chip; the workspace carries only the
 * work-distribution header (GLDM_R1D_WS_*: 256 bytes) and one 8-byte {latent value, tag} hand-off
 * granule per activation column, used when a batch does not fill whole rounds of workgroups and the
 * left-over tiles are split along the step axis over several workgroups.  For a pose-decoder descriptor
 * (seq_len 16, latent_dim > 0, emb_dim >= 32) it also holds, behind those, the ResnetBlocks' scale/shift
 * rows per conditioning cloud (4 bytes x n_samples x sum of 2 C over the blocks: sized for one grasp per
 * cloud), written by gldm_decode itself before the fused launch.  For a 16-position latent-denoiser
 * descriptor (the `ppc` experiment) of the 64-column engine whose last level has 256 channels it holds, behind the granules
 * (256-byte aligned), 64 KiB of scratch per workgroup of the launch (min(tiles, compute units)): the level's residual
 * stream is parked there, by the lanes that re-load it, while LDS holds its padded split-f16 planes (the 4-position
 * engine keeps both in LDS since ABI 9 and needs header + granules only).
 * Contract: the caller ZEROES the workspace once, when it allocates it; a workspace is used by one
 * launch at a time (launches on the same stream may share it, concurrent streams may not); the
 * library re-arms it at the end of every launch.  The 32-bit word at byte GLDM_R1D_WS_ERROR is set
 * to 1 if a hand-off wait ever ran into its (seconds long) bound: outputs of that launch are invalid. */
#define GLDM_R1D_WS_ERROR 12
long long gldm_r1d_workspace_bytes(const gldm_r1d_desc *desc, int n_samples);

/* Which engine gldm_denoise / gldm_decode run this descriptor on: 64 = the position-major engine (64-column tiles = 16
 * samples x 4 positions, or 4 samples x 16 positions; GEMMs as split-f16 products on the f16 matrix pipe: nets packed with
 * the split fields), 32 = the sample-major engine (32-column tiles, f32 matrix pipe: every other supported shape), or a negative GLDM_ERR_* status for a descriptor no engine takes.  No reference
 * counterpart: reporting only (bench.py labels its roofline record with it). */
int gldm_r1d_tile_columns(const gldm_r1d_desc *desc);

/* ref: grasp_ldm/models/diffusion/gaussian_diffusion.py:232-277 (sample loop:
 * eps = model(x, t, z_cond); x = scheduler.step(eps, t, x)) and
 * resnets.py:558-616 (TimeConditionedResNet1D.forward), fused: ONE launch runs
 * all n_steps for every latent; x stays on chip between steps.
 *  - sched_kind NONE with n_steps = 1 returns eps (= the module's forward);
 *    sample_t (optional, [n]) then gives a per-sample timestep.
 *  - temb is the host-precomputed time_mlp table [T, E] (resnets.py:517-522).
 *  - sample i is conditioned on cemb[i / samples_per_cond].
 *  - step_noise [n_steps, n, D] is read by DDPM steps with coef[7] != 0.
 *  - sample_emb [n, E] (optional) is added to the time embedding of each sample before the
 *    conditioning embedding: the class embedding of ClassTimeConditionedResNet1D
 *    (grasp_ldm/models/modules/class_conditioned_resnet.py:43-46,99-101). */
int gldm_denoise(const gldm_r1d_desc *desc, const float *weights, const float *temb, const float *cemb,
                 int samples_per_cond, const float *x_in /*[n,1,L]*/, int n_samples,
                 const int32_t *timesteps /*[n_steps]*/, const int32_t *sample_t, int n_steps,
                 int sched_kind, int clip_sample, const float *sched_coef /*[n_steps,8]*/,
                 const float *step_noise, const float *sample_emb, float *x_out /*[n,1,L]*/, void *workspace,
                 gldm_stream_t stream);

/* gldm_denoise for DDPM throughput runs with the per-step noise drawn IN the kernel (ABI 9): the reference draws one
 * [n,1,D] normal tensor per step on the device (gaussian_diffusion.py:258-272); a fused launch fed from memory needs all of
 * them up front ([steps,n,1,D]: 205 MB per 12,800-latent batch of BASELINE configs[4]).  Here every (latent, position,
 * step) gets its unit normal from Philox4x32-10 + Box-Muller keyed on `noise_seed`, counter = (noise_base + latent index,
 * position / 4, step): results depend on the seed and on a latent's GLOBAL index only -- not on tiling, batch splits or the
 * world size (a rank passes the global index of its first latent as noise_base).  Not bit-compatible with torch's stream:
 * parity tests use gldm_denoise with recorded noise.  Arguments as gldm_denoise (DDPM, no per-sample timesteps). */
int gldm_denoise_rng(const gldm_r1d_desc *desc, const float *weights, const float *temb, const float *cemb,
                     int samples_per_cond, const float *x_in /*[n,1,L]*/, int n_samples,
                     const int32_t *timesteps /*[n_steps]*/, int n_steps, int clip_sample,
                     const float *sched_coef /*[n_steps,8]*/, unsigned long long noise_seed, long long noise_base,
                     const float *sample_emb, float *x_out /*[n,1,L]*/, void *workspace, gldm_stream_t stream);
/* The same generator on its own: out[i][l] = the normal gldm_denoise_rng adds to latent noise_base + i, position l, at
 * step `step` (statistical tests; no reference counterpart). */
int gldm_step_noise_rng(unsigned long long noise_seed, long long noise_base, int step, int n_samples, int seq_len,
                        float *out /*[n,L]*/, gldm_stream_t stream);

/* ref: grasp_ldm/models/grasp_vae.py:401-436 (ConditionalGraspPoseDecoder.forward:
 * in_layer -> ResNet1D -> tmrp / class_logits heads). */
int gldm_decode(const gldm_r1d_desc *desc, const float *weights, const float *cemb, int samples_per_cond,
                const float *z_h /*[n,D]*/, int n_samples, float *tmrp /*[n,6]*/, float *logit /*[n,1]*/,
                void *workspace, gldm_stream_t stream);

/* ref: tools/inference.py:64-94,628-647 + grasp_ldm/utils/rotations.py:171-302:
 * un = tmrp*std+mean; H = tmrp_to_H(un); conf = sigmoid(logit).  mean AND std are
 * per cloud, [n_clouds,6] each (a caller holding the reference's broadcastable [1,6] std expands
 * it first); grasp i belongs to cloud i / grasps_per_cloud; n <= n_clouds * grasps_per_cloud is
 * checked (GLDM_ERR_INVALID_ARG). */
int gldm_pose_epilogue(const float *tmrp /*[n,6]*/, const float *logit /*[n]*/, const float *grasp_mean,
                       const float *grasp_std, int n, int grasps_per_cloud, int n_clouds, float *H /*[n,4,4]*/,
                       float *tmrp_unnorm /*[n,6]*/, float *confidence /*[n]*/, gldm_stream_t stream);

/* ref: grasp_ldm/models/modules/ext/pvcnn/modules/pointnet.py:100-111 (PointNetSAModule.forward
 * after FPS + ball query): neighbour gather + grouped SharedMLP2d (Conv2d k1 + eval BatchNorm
 * folded + ReLU, shared_mlp.py:6-35) + max over the U neighbours, fused; the grouped tensor
 * never reaches HBM.  `weights` holds, per layer l, the folded weight [cout x cin_pad] in MFMA
 * A-fragment order at w_off[l] and the folded bias at b_off[l] (graspldm_amd/sa_pack.py);
 * cin_pad[0] = roundup(3 + c, 16), u must divide 64, cout <= 256. */
int gldm_sa_mlp_forward(const float *points /*[b,3,n]*/, const float *centers /*[b,3,m]*/,
                        const float *features /*[b,c,n] or NULL*/, const int32_t *idx /*[b,m,u]*/,
                        const float *weights, int b, int c, int n, int m, int u, int n_layers,
                        const int32_t *cin_pad, const int32_t *cout, const int32_t *w_off, const int32_t *b_off,
                        float *out /*[b,cout_last,m]*/, gldm_stream_t stream);

/* The same module core with the GEMMs on the f16 matrix pipe (every f32 product as three f16 partial products of the
 * hi / lo splits of both operands, f32 accumulation: the arithmetic of the denoiser engines): 64-column tiles whose
 * gathered rows and hidden-layer outputs live in LDS as pre-split planes.  `weights` holds, per layer, the split-f16 A
 * fragments of [cout x cin_pad] at w3_off[l] (graspldm_amd/r1d_pack.py: mfma_a_fragments_f16x2; cin_pad a multiple of 32,
 * zero beyond the real rows) and the folded bias at b_off[l].  Shapes: cin_pad[0] <= 288, hidden widths multiples of 32
 * (32 / 64 / 128 / 256), U in {16, 32, 64}; GLDM_ERR_UNSUPPORTED otherwise (callers then use gldm_sa_mlp_forward).
 * `range_gain` (HOST memory, [n_layers][2] = per layer the largest row sum of |W| and the largest |bias|, BatchNorm folded;
 * ABI 10): f16 has 5 exponent bits, so every 64-column tile is split as x / s with s a power of two taken from the tile's
 * largest gathered magnitude, the hidden layers' planes with one taken from the bound gain_r * max|in| + gain_b, and s is
 * folded back on the accumulators (exact; s = 1, i.e. every bit as without it, while 2^-8 <= magnitude < 2^14).  NULL: no
 * scales -- the caller vouches for |values| < 65504 everywhere. */
int gldm_sa_mlp_forward_f16x2(const float *points /*[b,3,n]*/, const float *centers /*[b,3,m]*/,
                               const float *features /*[b,c,n] or NULL*/, const int32_t *idx /*[b,m,u]*/,
                               const float *weights, int b, int c, int n, int m, int u, int n_layers,
                               const int32_t *cin_pad, const int32_t *cout, const int32_t *w3_off, const int32_t *b_off,
                               const float *range_gain /*host [n_layers][2] or NULL*/,
                               float *out /*[b,cout_last,m]*/, gldm_stream_t stream);

/* The same launch with the module's FIRST layer hoisted out of the (centre, neighbour) pairs (ABI 10).  Its input is the
 * concatenation [x - centre; f] (ball_query.py:21-33), so  W1 [x - c; f] + b1 = W1a (x - c) + (W1b f + b1)  and the second
 * term depends on the point only: the caller computes pre = W1b f + b1 once per cloud, POINT-major [b, n, c1]
 * (gldm_pointwise_mlp_f16x2_pm; every point sits in ~m u / n balls, 16 at SSG-SA2: a neighbour's rows are then one run of
 * c1 floats and a gather thread's four rows one 16-byte load), the gather fetches a neighbour's rows of `pre` instead of its
 * features, adds the three coordinate products (W1a [c1][4] = columns x, y, z, 0 at float index wa_off of `weights`) and
 * applies the ReLU.  The layer tables describe layers 2.. of the module (cin_pad[0] = c1 = rows of pre, a multiple of 32;
 * rows the module does not have carry zero weights and zero pre).  pre_broadcast != 0: a module WITHOUT features -- its first
 * layer is W1a (x - c) + b1, `pre` is the one row b1 [c1] shared by every point.  Same arithmetic for layers 2.., the first layer's
 * products are f32 (VALU) + the caller's GEMM: results differ from gldm_sa_mlp_forward_f16x2 in the last bits. */
int gldm_sa_mlp_forward_f16x2_pre(const float *points /*[b,3,n]*/, const float *centers /*[b,3,m]*/,
                                   const float *pre /*[b,n,c1]; [c1] with pre_broadcast*/, int pre_broadcast,
                                   const int32_t *idx /*[b,m,u]*/, const float *weights,
                                   int wa_off, int b, int n, int m, int u, int n_layers,
                                   const int32_t *cin_pad, const int32_t *cout, const int32_t *w3_off, const int32_t *b_off,
                                   const float *range_gain /*host [n_layers][2] or NULL*/,
                                   float *out /*[b,cout_last,m]*/, gldm_stream_t stream);

/* ref: grasp_ldm/models/modules/ext/pvcnn/modules/shared_mlp.py:6-35 (Conv1d k = 1 + eval BatchNorm folded + ReLU),
 * one layer, in the native [b, c, n] layout: y = act(W x + bias).  `w_packed` = the folded weight [cout, cin] in MFMA
 * A-fragment order (graspldm_amd/r1d_pack.py: mfma_a_fragments).  Optional fused head on the accumulators:
 * z = Wh y + bh with hout <= 16 rows (grasp_ldm/models/modules/pc_encoders.py:104-111: conv_downscale and out_layer[0]
 * folded into one [hout x cout] matrix); `head_w_packed` = [16, cout] in A-fragment order with the k index of every
 * 16-block permuted k' = 4 (k % 4) + k / 4 (graspldm_amd/dense.py: pack_head).  With a head, y may be NULL and the
 * [b, cout, n] tensor never reaches HBM.  cin % 32 == 0, cout % 256 == 0, n % 32 == 0, 4 (32 cin + 4096) <= 160 KiB. */
int gldm_pointwise_mlp(const float *x /*[b,cin,n]*/, const float *w_packed, const float *bias /*[cout]*/,
                       int b, int cin, int cout, int n, int relu,
                       const float *head_w_packed, const float *head_bias /*[hout] or NULL*/, int hout,
                       float *y /*[b,cout,n] or NULL*/, float *z /*[b,hout,n] or NULL*/, gldm_stream_t stream);

/* Two consecutive SharedMLP layers (both with ReLU) and the optional head in ONE launch: the first layer's output
 * (x [b,cin0,n] -> [b,cin,n]) is produced tile by tile in LDS as the second layer's input and never reaches HBM
 * (the shipped encoder's 96 -> 768 -> 1536 -> head: 0.8 GB less written and read per 256 clouds).
 * cin0 % 32 == 0, cin % 256 == 0, other constraints as gldm_pointwise_mlp, 4 (32 (cin + cin0) + 4096) <= 160 KiB. */
int gldm_pointwise_mlp2(const float *x /*[b,cin0,n]*/, const float *w0_packed, const float *bias0 /*[cin]*/, int cin0,
                        const float *w_packed, const float *bias /*[cout]*/, int b, int cin, int cout, int n,
                        const float *head_w_packed, const float *head_bias, int hout,
                        float *y /*[b,cout,n] or NULL*/, float *z /*[b,hout,n] or NULL*/, gldm_stream_t stream);

/* ref: shared_mlp.py:6-35 for NARROW layers (the PVConv point branches 3 -> 48, 48 -> 96): y = act(W x + bias) over
 * [b, cin, n] with W [cout, cin] row major (BatchNorm folded by the caller); cin in {3, 6, 16, 24, 32, 48, 64}.
 * VALU kernel (lane = point), k-ordered fma chain from the bias. */
int gldm_pointwise_small(const float *x /*[b,cin,n]*/, const float *w /*[cout,cin]*/, const float *bias /*[cout] or NULL*/,
                         int b, int cin, int cout, long long n, int relu, float *y /*[b,cout,n]*/, gldm_stream_t stream);

/* ref: the same module for EVERY other shape (ext/pvcnn/modules/shared_mlp.py:6-35, ext/pvcnn/modules/pointnet.py:117-135:
 * the feature-propagation SharedMLPs of PointNet++ / PVCNN2, e.g. 384 -> 256 over 128 centres): any cin, cout, n; weights
 * [cout][cin] as stored (BatchNorm folded by the caller), exact f32 products on the f32 matrix pipe, 64 x 64 output tiles.
 * These layers went to the GEMM library (rocBLAS / MIOpen through F.conv1d) + gldm_bias_act before ABI 7. */
int gldm_pointwise_any(const float *x /*[b,cin,n]*/, const float *w /*[cout,cin]*/, const float *bias /*[cout] or NULL*/,
                       int b, int cin, int cout, long long n, int relu, float *y /*[b,cout,n]*/, gldm_stream_t stream);

/* ref: pc_encoders.py:60-82,104-111: out_layer[1] = nn.Linear(n_points, latent) applied over the POINT axis of
 * [B, C, N]: y[row, :] = W x[row, :] + bias for rows = B * C; n % 4 == 0, n <= 16384. */
int gldm_linear_rows(const float *x /*[rows,n]*/, const float *w /*[nout,n]*/, const float *bias /*[nout] or NULL*/,
                     int rows, int n, int nout, float *y /*[rows,nout]*/, gldm_stream_t stream);

/* The same two entry points with the MAIN layer's weights as split-f16 fragments (graspldm_amd/r1d_pack.py:
 * mfma_a_fragments_f16x2; layout above): the GEMM runs on the f16 matrix pipe with three partial products per f32
 * product and f32 accumulation (error of the order of one f32 rounding per product, 3/16 of the f32-MFMA time).  The
 * input tile is split once while it is staged.  Since ABI 6 the front layer's weights `w0_split` are split-f16 fragments
 * too (cin0 % 32 == 0, cin0 <= 96; its f32 input tile is split once per wave into registers); `head_w_packed` stays f32
 * fragments.  cin % 128 == 0, cout % 32 == 0 (with a front layer or a head: % 256; fewer than 256 output rows leave waves
 * idle), n % 32 == 0, 4 (48 cin + 32 cin0) + 16 <= 160 KiB. */
/* (ABI 10: without a front layer cin may be any multiple of 8 -- `w_split` then holds the fragments of W zero-padded to a
 * multiple of 128 columns, the K the launch walks; cout any multiple of 16 below 256 rows, of 32 from there.) */
int gldm_pointwise_mlp_f16x2(const float *x /*[b,cin,n]*/, const float *w_split, const float *bias /*[cout]*/,
                              int b, int cin, int cout, int n, int relu,
                              const float *head_w_packed, const float *head_bias, int hout,
                              float *y /*[b,cout,n] or NULL*/, float *z /*[b,hout,n] or NULL*/, gldm_stream_t stream);

/* The same launch with an addend in front of the activation: y = act(W x + bias + add), add[cloud * add_cloud_stride +
 * row * add_row_stride + col * add_col_stride] -- a per-cloud bias (strides cout, 1, 0) or a [b, cout, n] tensor
 * (cout * n, n, 1).  It serves layers whose input is a concatenation (pointnet.py:117-135 PointNetFPModule, :11-46
 * PointNetAModule): W [x1; x2] = W1 x1 + W2 x2, the wide part here, the other part (three coordinate rows, or one centre's
 * feature vector broadcast to every point) as the addend, and the concatenated tensor is never built. */
/* gldm_pointwise_mlp_f16x2 writing y POINT-major, [b, n, cout] (a lane's four consecutive output rows of a point: one
 * 16-byte store): the layout gldm_sa_mlp_forward_f16x2_pre gathers from. */
int gldm_pointwise_mlp_f16x2_pm(const float *x /*[b,cin,n]*/, const float *w_split, const float *bias /*[cout]*/, int b, int cin,
                                 int cout, int n, int relu, float *y_point_major /*[b,n,cout]*/, gldm_stream_t stream);
int gldm_pointwise_mlp_f16x2_add(const float *x /*[b,cin,n]*/, const float *w_split, const float *bias /*[cout]*/,
                                  const float *add, long long add_cloud_stride, long long add_row_stride,
                                  long long add_col_stride, int b, int cin, int cout, int n, int relu,
                                  float *y /*[b,cout,n]*/, gldm_stream_t stream);
/* Range of the split operands (ABI 10): the one-layer launches above split every input tile as x / s, s a power of two from
 * the tile's largest magnitude (1 while 2^-8 <= magnitude < 2^14: then every bit is as without it), and fold s back on the
 * accumulators.  The two-layer launch below does the same for its input tile and scales the front layer's output planes by
 * the bound front_gain[0] * max|x| + front_gain[1] (HOST memory: largest row sum of |W0|, largest |bias0|); front_gain ==
 * NULL: no scales in that launch -- the caller vouches for |values| < 65504. */
int gldm_pointwise_mlp2_f16x2(const float *x /*[b,cin0,n]*/, const float *w0_split, const float *bias0, int cin0,
                               const float *w_split, const float *bias /*[cout]*/, int b, int cin, int cout, int n,
                               const float *head_w_packed, const float *head_bias, int hout,
                               const float *front_gain /*host [2] or NULL*/,
                               float *y /*[b,cout,n] or NULL*/, float *z /*[b,hout,n] or NULL*/, gldm_stream_t stream);

/* ref: grasp_ldm/models/modules/ext/pvcnn/modules/pointnet.py:40-44 (PointNetAModule: `features.max(dim=-1)`, the global
 * pooling over a cloud's centres).  out[row] = max over x[row][0..n) for rows = b * c; NaN propagates as in torch.max. */
int gldm_row_max(const float *x /*[rows,n]*/, long long rows, int n, float *out /*[rows]*/, gldm_stream_t stream);

/* ---------------------------------------------------------- voxel branch of PVConv */

/* ref: grasp_ldm/models/modules/ext/pvcnn/modules/pvconv.py:48-66 (nn.Conv3d k=3 p=1 on the
 * [b, c, r, r, r] voxel grid).  Implicit GEMM on f32 MFMA; `w_packed` = weight [cout, cin, 3,3,3]
 * re-laid as [cout, 27 * cin_pad] (k = tap * cin_pad + ci, tap = (dx*3+dy)*3+dz, cin_pad =
 * roundup(cin,16)) in MFMA A-fragment order (graspldm_amd/voxel.py).  Also writes, per 4x4xr
 * brick and output channel, (sum, sum of squares) of the outputs to `partial`
 * [gldm_conv3d_partial_floats()] for the following GroupNorm.  r % 4 == 0. */
long long gldm_conv3d_partial_floats(int b, int cout, int r);
int gldm_conv3d_k3(const float *x /*[b,cin,r^3]*/, const float *w_packed, const float *bias /*[cout]*/,
                   int b, int cin, int cout, int r, float *y /*[b,cout,r^3]*/, float *partial,
                   gldm_stream_t stream);

/* gldm_conv3d_k3 writing its output channel-last, [b, r^3, cout] (the layout gldm_gn_swish_chan_sum_cl and
 * gldm_devoxelize_gn_cl_fused read: a voxel stack's last conv); cout % 4 == 0. */
int gldm_conv3d_k3_cl(const float *x /*[b,cin,r^3]*/, const float *w_packed, const float *bias /*[cout]*/, int b, int cin,
                      int cout, int r, float *y_cl /*[b,r^3,cout]*/, float *partial, gldm_stream_t stream);

/* The same conv for ANY channel counts (r % 4 == 0) with the weight as nn.Conv3d stores it, [cout, cin, 27] f32: a
 * direct VALU kernel for voxel shapes the MFMA kernels are not instantiated for (PVCNN2's 256 ch @ 8^3, 128 ch @ 16^3);
 * same `partial` layout.  A correctness path: no shape of the shipped encoder uses it. */
int gldm_conv3d_k3_generic(const float *x /*[b,cin,r^3]*/, const float *w /*[cout,cin,27]*/, const float *bias,
                           int b, int cin, int cout, int r, float *y /*[b,cout,r^3]*/, float *partial,
                           gldm_stream_t stream);

/* The same conv with split-f16 weights (graspldm_amd/voxel.py: pack_conv3d_f16x2: [cout, cblocks * 14 * 32] with
 * k = ((16-channel block) * 14 + tap pair) * 32 + 16 (tap - 2 pair) + channel, as mfma_a_fragments_f16x2 fragments):
 * three f16 partial products per f32 product on the f16 matrix pipe, f32 accumulation.  Built for the shipped
 * encoder's shapes (cout 48 at r = 24, cout 96 at r = 12 with cin % 16 == 0; and the first conv, cin = 3 -> 48 at r = 24,
 * whose weights are packed tap-major without padding between taps, k = tap * 3 + ci < 81 in three 32-deep blocks:
 * pack_conv3d_fewch_f16x2); GLDM_ERR_UNSUPPORTED otherwise. */
int gldm_conv3d_k3_f16x2(const float *x /*[b,cin,r^3]*/, const float *w_split, const float *bias /*[cout]*/,
                          int b, int cin, int cout, int r, float *y /*[b,cout,r^3]*/, float *partial,
                          gldm_stream_t stream);

/* ref: pvconv.py:57-66 (Conv3d -> GroupNorm(8) -> Swish -> Conv3d): a conv of the voxel stack with its neighbours' work
 * folded in.  in_coef [b, cin, 2] (optional) = (a, s) from gldm_groupnorm_coef: x is the previous conv's RAW output and
 * x' = swish(a x + s) is applied per in-grid element while the bricks are staged (the activated tensor is never written).
 * out_channel_last != 0: y is written [b, r^3, cout] (a voxel's channels as one run: the layout the squeeze and
 * devoxelize passes below read; `partial` is unchanged).  cin % 16 == 0, the cout / r of gldm_conv3d_k3_f16x2. */
int gldm_conv3d_k3_f16x2_gn(const float *x /*[b,cin,r^3] raw*/, const float *in_coef /*[b,cin,2] or NULL*/,
                             const float *w_split, const float *bias /*[cout]*/, int b, int cin, int cout, int r,
                             float *y /*[b,cout,r^3] or [b,r^3,cout]*/, float *partial, int out_channel_last,
                             gldm_stream_t stream);

/* ref: pvconv.py:57-66 (nn.GroupNorm(8, c)) as per-(cloud, channel) coefficients: GN(x) = a x + s, a = gamma rstd,
 * s = beta - mean a, statistics from a conv's `partial` (combined in f64 in a fixed order, as gldm_groupnorm_swish).
 * c / groups <= 64. */
int gldm_groupnorm_coef(const float *partial, const float *gamma, const float *beta, int b, int c, int r, int groups,
                        float eps, float *coef /*[b,c,2]*/, gldm_stream_t stream);

/* chan_sum[b,c] = sum over the r^3 voxels of swish(a y + s): the SE squeeze (se.py:12-25) of a GroupNorm + Swish output
 * that is never written (read-only pass over the raw conv output). */
int gldm_gn_swish_chan_sum(const float *y /*[b,c,r^3] raw*/, const float *coef /*[b,c,2]*/, int b, int c, int r,
                           float *chan_sum /*[b,c]*/, gldm_stream_t stream);

/* The same squeeze over a channel-last tensor, as gldm_squeeze_parts() partial sums per cloud (fixed split of the voxels,
 * added in index order by gldm_se_gate_parts: deterministic, no atomics).  c % 4 == 0. */
int gldm_squeeze_parts(void);
int gldm_gn_swish_chan_sum_cl(const float *y /*[b,r^3,c] raw*/, const float *coef /*[b,c,2]*/, int b, int c, int r,
                              float *chan_parts /*[b,parts,c]*/, gldm_stream_t stream);

/* ref: pvconv.py:57-66 (nn.GroupNorm(8, c) + Swish), in place on y; statistics from `partial`
 * (combined in f64 in a fixed order).  chan_sum [b,c] (optional) receives the per-channel sum of
 * the OUTPUT: the squeeze of the SE block that follows. */
int gldm_groupnorm_swish(float *y /*[b,c,r^3]*/, const float *partial, const float *gamma, const float *beta,
                         int b, int c, int r, int groups, float eps, float *chan_sum, gldm_stream_t stream);

/* ref: grasp_ldm/models/modules/ext/pvcnn/modules/se.py:12-25: gate = sigmoid(W2 act(W1 mean)),
 * act = ReLU (use_relu) or Swish. */
int gldm_se_gate(const float *chan_sum /*[b,c]*/, const float *w1 /*[hidden,c]*/, const float *w2 /*[c,hidden]*/,
                 int b, int c, int hidden, int r, int use_relu, float *gate /*[b,c]*/, gldm_stream_t stream);
int gldm_se_gate_parts(const float *chan_parts /*[b,parts,c]*/, int parts, const float *w1, const float *w2, int b, int c,
                       int hidden, int r, int use_relu, float *gate /*[b,c]*/, gldm_stream_t stream);

/* ref: pvconv.py:79-83: trilinear_devoxelize(SE(v)) + point_features(x) in one pass:
 * out = gate[b,c] * trilinear(V) + add  (gate / add may be NULL). */
int gldm_devoxelize_fused(const float *coords /*[b,3,n]*/, const float *features /*[b,c,r^3]*/,
                          const float *gate /*[b,c]*/, const float *add /*[b,c,n]*/, int b, int c, int n, int r,
                          float *out /*[b,c,n]*/, gldm_stream_t stream);

/* The same pass over a RAW conv output: GroupNorm + Swish (coef [b,c,2] from gldm_groupnorm_coef) applied to the eight
 * corner values of every point before the trilinear blend: out = gate * trilinear(swish(a V + s)) + add. */
int gldm_devoxelize_gn_fused(const float *coords /*[b,3,n]*/, const float *features /*[b,c,r^3] raw*/,
                             const float *coef /*[b,c,2]*/, const float *gate /*[b,c]*/, const float *add /*[b,c,n]*/,
                             int b, int c, int n, int r, float *out /*[b,c,n]*/, gldm_stream_t stream);
/* ... and over a channel-LAST raw conv output: a point's corner is one run of c floats (16-byte loads by neighbouring
 * lanes) instead of c gathers from c cache lines.  c % 4 == 0, c <= 256. */
int gldm_devoxelize_gn_cl_fused(const float *coords /*[b,3,n]*/, const float *features_cl /*[b,r^3,c] raw*/,
                                const float *coef /*[b,c,2]*/, const float *gate /*[b,c]*/, const float *add /*[b,c,n]*/,
                                int b, int c, int n, int r, float *out /*[b,c,n]*/, gldm_stream_t stream);

/* y[b, c, :] = act(y[b, c, :] + bias[c]) in place (relu != 0: ReLU).  Epilogue of the k = 1 Conv1d /
 * Conv2d + BatchNorm(eval, folded) + ReLU of ext/pvcnn/modules/shared_mlp.py:24-36 when the GEMM itself
 * runs as a plain library call.  n % 4 == 0 (rows stay 16-byte aligned). */
int gldm_bias_act(float *y /*[b,c,n]*/, const float *bias /*[c]*/, int b, int c, long long n, int relu,
                  gldm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GLDM_H_ */
