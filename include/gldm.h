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

#ifdef __cplusplus
}
#endif
#endif /* GLDM_H_ */
