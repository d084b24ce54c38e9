/*
 * hold_hip.h -- C ABI of libholdhip.so, the MI355X (gfx950) implementation of the HOLD
 * volumetric hand-object rendering hot path.
 *
 * The reference (zc-alexfan/hold) has no FFI / plugin interface: its boundary for this path is
 * the in-process Python nn.Module call surface.  Each entry point below names
 * the reference function(s) whose arithmetic it replaces (paths relative to the reference root).
 * The host-side mirror of the reference interface lives in the hold_amd python package and binds these with
 * ctypes.
 *
 * Conventions: plain pointers are DEVICE pointers unless noted; all floating point is fp32;
 * every function enqueues work on `stream` and returns 0, or a negative HOLD_E_* code (no
 * exceptions, no global state, no allocation -- scratch is passed in by the caller).
 */
#ifndef HOLD_HIP_H
#define HOLD_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* hold_stream_t; /* hipStream_t */

#define HOLD_OK 0
#define HOLD_E_ARG (-1)    /* bad argument (null pointer, misaligned leading dimension, ...) */
#define HOLD_E_LAUNCH (-2) /* hipGetLastError() != hipSuccess after a launch */

int hold_abi_version(void);

/* ------------------------------------------------------------------------------------------
 * Dense layer GEMM on the fp32 matrix cores (v_mfma_f32_32x32x2_f32), epilogue-fused.
 *   C[p][n] = epi( alpha * sum_k A[p][k] * W[n][k] + bias[n] )
 * Replaces every nn.Linear (+softplus/ReLU/sigmoid) of ImplicitNet.forward
 * (code/src/networks/shape_net.py:118-126) and RenderingNet.forward
 * (code/src/networks/texture_net.py:94-101), and -- with W transposed by the caller -- the
 * input-gradient / backward / double-backward sweeps torch autograd runs for
 * code/src/engine/volsdf_utils.py:71-96.
 * Requirements: lda, ldw multiples of 4 floats; A, W 16-byte aligned; K multiple of 4.
 * ---------------------------------------------------------------------------------------- */
enum hold_epilogue {
  HOLD_EPI_NONE = 0,      /* y                                                    */
  HOLD_EPI_SOFTPLUS = 1,  /* softplus(y, beta=100, threshold=20)                   */
  HOLD_EPI_RELU = 2,      /* max(y,0)                                              */
  HOLD_EPI_SIGMOID = 3,   /* 1/(1+exp(-y))                                         */
  HOLD_EPI_MUL_DSP = 4,   /* y * softplus'(.) recovered from aux1 = softplus output,
                             + aux2 (optional additive term)                       */
  HOLD_EPI_MUL_DRELU = 5, /* y * (aux1 > 0)                                        */
  HOLD_EPI_DBWD = 6,      /* double-backward of the softplus gate: with s = softplus'(aux1):
                             C = y*s ; out2 = 100 * y * aux2 * (1-s)   (aux2 = t = u*s)   */
  HOLD_EPI_MUL_DSIG = 7   /* y * aux1 * (1-aux1)  (aux1 = sigmoid output)           */
};

typedef struct hold_gemm_desc {
  const float* A;   int32_t lda;    /* [P][K] activations, row-major                    */
  const float* W;   int32_t ldw;    /* [N][K] weights, row-major (torch Linear layout)  */
  const float* bias;                /* [N] or NULL                                      */
  float* C;         int32_t ldc;    /* [P][>=min(N,n_split)] output                     */
  int32_t P, N, K;
  float alpha;
  int32_t epilogue;                 /* enum hold_epilogue, applies to columns < n_split */
  int32_t n_split;                  /* columns n >= n_split are stored raw (alpha*acc+bias)
                                       to C2[p][n-n_split]; set n_split = N to disable   */
  float* C2;        int32_t ldc2;
  const float* aux1; int32_t ldaux1;
  const float* aux2; int32_t ldaux2;
  float* out2;      int32_t ldout2;
  int32_t accumulate;               /* 1: C += result (HOLD_EPI_NONE / raw columns only) */
  const float* r1_row; int32_t ldr1;/* optional rank-1 term added to y before the epilogue function:          */
  const float* r1_col;              /*   y += r1_row[p * ldr1] * r1_col[n]   (both NULL to disable)           */
} hold_gemm_desc;

int hold_gemm_nt(const hold_gemm_desc* d, hold_stream_t stream);
/* same contract and epilogues, split-precision arithmetic: A and W fragments are decomposed into three bf16 limbs as
 * they leave LDS (exact 8+8+8-bit truncation split), six limb products on v_mfma_f32_32x32x16_bf16, fp32 accumulation
 * (dropped products <= 2^-23 relative). */
int hold_gemm_nt_x6(const hold_gemm_desc* d, hold_stream_t stream);

/* Weight gradient: dW[n][k] (+)= sum_p R[p][n] * X[p][k];  db[n] (+)= sum_p R[p][n] (db may be NULL).
 * Split over P into `splits` partial tiles in `workspace` (>= hold_wgrad_workspace_floats floats),
 * reduced deterministically by a second kernel.  Replaces the weight/bias gradient GEMMs of
 * torch autograd for every Linear on the path. */
int64_t hold_wgrad_workspace_floats(int32_t N, int32_t K, int32_t splits);
int hold_wgrad(const float* R, int32_t ldr, const float* X, int32_t ldx, int32_t P, int32_t N, int32_t K,
               float* dW, int32_t lddw, float* db, int32_t accumulate, int32_t splits, float* workspace,
               hold_stream_t stream);
/* same contract, split-precision arithmetic: both operands are decomposed into three bf16 limbs as they leave LDS
 * (exact 8+8+8-bit split of the significand), six limb products on v_mfma_f32_32x32x16_bf16, fp32 accumulation. */
int hold_wgrad_x6(const float* R, int32_t ldr, const float* X, int32_t ldx, int32_t P, int32_t N, int32_t K,
                  float* dW, int32_t lddw, float* db, int32_t accumulate, int32_t splits, float* workspace,
                  hold_stream_t stream);
/* same contract; the shapes the whole-dW workgroups take (N in 129..256, K in 256..320, P a multiple of 16 >= 4 096, rows of R
 * and X at least 256 floats wide) in the TWO-LIMB fp16 arithmetic "f16x3" (hold_amd/csrc/wgrad_r6.hip: wgrad_h3_body): both
 * operands, scaled by powers of two, as hi = RN_f16(s x), lo = RN_f16(s x - hi), the products hi hi + hi lo + lo hi on
 * v_mfma_f32_32x32x16_f16 with fp32 accumulation.  Every workgroup derives its two scales from a sample of its own rows
 * (sampled maximum -> [2^6, 2^7), 2^9 of headroom; a value beyond fp16's range gives inf / NaN in dW, never a silently wrong
 * number) and un-scales its partial tile exactly before the fp32 reduction.  Every other shape: hold_wgrad_x6. */
int hold_wgrad_h3(const float* R, int32_t ldr, const float* X, int32_t ldx, int32_t P, int32_t N, int32_t K,
                  float* dW, int32_t lddw, float* db, int32_t accumulate, int32_t splits, float* workspace,
                  hold_stream_t stream);

/* Narrow layer, split-precision arithmetic of hold_gemm_nt_x6:  C[p][n] (+)= sum_k A[p][k] W[n][k],  K = 256, N <= 64
 * (hold_amd/csrc/rnarrow.hip).  For the GEMMs of the path whose output is a few dozen columns wide -- d sdf / d embedding
 * (N = 39: torch autograd's `grad_output @ lin0.weight`, reference code/src/networks/shape_net.py:108-116) and the
 * non-feature columns of the colour net's input gradient (N = 16 / 48, code/src/networks/texture_net.py:89-101): A is
 * streamed once, only ceil(N / 32) output tiles are computed.  A and W 16-byte aligned with leading dimensions that are
 * multiples of 4 and >= 256; C any 4-byte aligned view with ldc >= N. */
int hold_gemm_narrow_x6(const float* A, int32_t lda, const float* W, int32_t ldw, float* C, int32_t ldc, int64_t P,
                        int32_t N, int32_t accumulate, hold_stream_t stream);

/* Several weight gradients over the SAME P points in one launch (the 15 of a node's implicit-net backward -- torch
 * autograd's per-Linear `grad_output.T @ input`, reference code/src/networks/shape_net.py:84-130 -- plus the colour
 * net's): item i contributes  R_i[:, :N]^T X_i[:, :256]  (and the column sums of R_i[:, :N] when db is given) to its
 * dW / db.  Items with the same dW must be adjacent in the list and are summed into it in ONE deterministic reduction;
 * `accumulate` of the first item of such a run decides whether dW / db's previous content is kept.  A compute unit
 * works on one (item, share of the points): the launch writes one partial tile per compute unit for all items together
 * instead of one per compute unit and item.  Split-precision arithmetic of hold_wgrad_x6.  Requirements: n_items <= 24,
 * P a multiple of 16, R and X at least 256 floats wide in memory (256 columns of both are read; what the columns >= N
 * of R hold is never used), 16-byte aligned, leading dimensions multiples of 4; HOLD_E_ARG otherwise.
 * workspace >= hold_wgrad_group_workspace_floats() floats. */
typedef struct hold_wgrad_item {
  const float* R; const float* X;   /* [P, ldr], [P, ldx] */
  float* dW; float* db;             /* [N, lddw] (256 columns written), [N] or NULL */
  int32_t ldr, ldx, lddw, N;
  int32_t accumulate, reserved;
} hold_wgrad_item;
int64_t hold_wgrad_group_workspace_floats(void);
int hold_wgrad_group_x6(const hold_wgrad_item* items, int32_t n_items, int64_t P, float* workspace,
                        hold_stream_t stream);
/* the same launch in the two-limb fp16 arithmetic of hold_wgrad_h3 (scales per workgroup = per (item, share of the points)) */
int hold_wgrad_group_h3(const hold_wgrad_item* items, int32_t n_items, int64_t P, float* workspace,
                        hold_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Exact sample compaction (hold_amd/csrc/compact.hip; reference semantics preserved: code/src/engine/volsdf_utils.py:220-251,
 * code/src/engine/density.py:21-26 -- the reference integrates every sample).  A sample is DEAD when the Laplace density of its
 * sdf and the exponential exp(-|sdf| / beta) of the density's derivatives are both exact fp32 zeros, evaluated with the
 * compositor's own expressions: its compositing weight and every gradient that would flow through it are then exactly zero.
 * hold_alive_count: block_counts[b] = live samples among sdf[b * 1024 .. +1024) (hold_alive_blocks(P) blocks);
 * hold_alive_index: idx[block_offsets[b] + rank] = p for the live samples of block b in ascending order (block_offsets = the
 * exclusive prefix sum of the counts) -- wave ballots + popcount ranks, deterministic.  sdf: [P] with stride ld floats.
 * ---------------------------------------------------------------------------------------- */
int64_t hold_alive_blocks(int64_t P);
int hold_alive_count(const float* sdf, int32_t ld, int64_t P, float beta, int32_t* block_counts, hold_stream_t stream);
int hold_alive_index(const float* sdf, int32_t ld, int64_t P, float beta, const int64_t* block_offsets, int64_t* idx,
                     hold_stream_t stream);
/* mask[p] = 1 (live) / 0 (dead), the same predicate: what the compaction of a BATCH of frames ranks frame by frame (every frame keeps
 * one common compacted row count, hold_amd/field.py:_compaction -- the reference's training batch is 10 frames x 128 rays,
 * code/confs/general.yaml:82) */
int hold_alive_mask(const float* sdf, int32_t ld, int64_t P, float beta, uint8_t* mask, hold_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Per-point kernels (hold_amd/csrc/points.hip)
 * ---------------------------------------------------------------------------------------- */

/* out[(r*S+s)][0..2] = cam_loc[r] + z[r][s] * ray_dirs[r]
 * (code/src/model/renderables/mano_node.py:111, code/src/engine/ray_sampler.py:162) */
int hold_ray_points(const float* cam_loc, const float* ray_dirs, const float* z, int32_t ldz, int32_t S,
                    int64_t n_rays, float* out, int32_t ldo, hold_stream_t stream);

/* Fourier positional encoding [x, sin(2^k x), cos(2^k x)]_{k<L} (+ optional BARF per-column weights,
 * + optional per-frame condition vector appended), written to out (and a second copy to out2 for the
 * skip connection).  code/src/engine/embedders.py:18-50, :92-122; shape_net.py:98-116. */
int hold_embed_fwd(const float* x, int32_t ldx, int32_t d_in, int32_t L, const float* barf_w, int64_t P, float* out,
                   int32_t ldo, float* out2, int32_t ldo2, const float* cond, int32_t cond_dim,
                   int64_t pts_per_frame, hold_stream_t stream);
/* gx[p][:] (+)= (d embed / d x)^T ge[p][:]   (d_in = 3).  Chain rule torch autograd applies for
 * code/src/engine/volsdf_utils.py:89-96 (gradient of sdf w.r.t. canonical points). */
int hold_embed_bwd(const float* x, int32_t ldx, int32_t L, const float* barf_w, int64_t P, const float* ge,
                   int32_t ldge, float* gx, int32_t ldgx, int32_t accumulate, hold_stream_t stream);
/* double backward of the line above: gebar = (d embed/dx) gbar ; xbar += gbar * (d2 embed/dx2 . ge).
 * gebar2 (NULL to disable): a second copy of gebar's 3 + 6 L columns; it MAY alias ge (each entry of ge is read before the
 * same thread overwrites it) -- the host keeps ge in the skip columns of t_3, where the ascending sweep of the double
 * backward (hold_chain_r6, HOLD_CHAIN_DBWD) expects gebar afterwards.  gebar itself must not alias ge. */
int hold_embed_bwd2(const float* x, int32_t ldx, int32_t L, const float* barf_w, int64_t P, const float* ge,
                    int32_t ldge, const float* gbar, int32_t ldgb, float* gebar, int32_t ldgeb, float* xbar,
                    int32_t ldxb, float* gebar2, int32_t ldgeb2, hold_stream_t stream);

/* KNN(K=15) skinning-weight lookup against the frame's posed (or the canonical) MANO vertices, fused
 * with inverse LBS when xc_out != NULL:  w = sum_k softmax-like conf_k * W[idx_k]  (detached),
 * x_c = (sum_j w_j T_j)^-1 [x;1].   Replaces pytorch3d.ops.knn_points + KNNDeformer.forward /
 * query_skinning_weights_multi / skinning (code/src/model/mano/deformer.py:34-68, :84-105, :145-170).
 * verts: [B][n_verts][3] with frame stride verts_frame_stride floats (0 = shared canonical verts), 60 <= n_verts <= 800;
 * skin_w [n_verts][16] (16-byte aligned); tfs [B][16][4][4]; w_out [P][16] (nullable); xc_out [P][ldxc] (nullable).
 * The K smallest distances are selected exactly as a full insertion scan in vertex order would (ties: lower index first);
 * internally a threshold from every 4th vertex and a one-bit-per-vertex filter cut the insertions to ~100 per point. */
int hold_knn_invlbs_fwd(const float* x, int32_t ldx, int64_t P, int64_t pts_per_frame, const float* verts,
                        int64_t verts_frame_stride, int32_t n_verts, const float* skin_w, const float* tfs,
                        float* w_out, float* xc_out, int32_t ldxc, hold_stream_t stream);
/* x_c = (sum_j w_j T_j)^-1 [x;1] with given weights (n_bones = 16) or a single rigid transform per
 * frame (n_bones = 1: ObjectDeformer.forward inverse, code/src/model/obj/deformer.py:10-41). */
int hold_invskin_fwd(const float* x, int32_t ldx, int64_t P, int64_t pts_per_frame, const float* w, const float* tfs,
                     int32_t n_bones, float* xc, int32_t ldxc, hold_stream_t stream);
/* ray generation: get_camera_params / lift, code/src/datasets/utils.py:230-282 (pose-matrix branch),
 * plus the per-ray broadcast of the camera centre (mano_node.py:87-92).  uv [B][rays_per_frame][2] pixel coordinates,
 * pose [B][4][4] camera-to-world, intrinsics [B][ld][ld] (ld = 3 or 4) -> ray_dirs [n_rays][3] unit, cam_loc [n_rays][3]. */
int hold_raygen(const float* uv, const float* pose, const float* intrinsics, int32_t ld_intr, int64_t n_rays,
                int64_t rays_per_frame, float* ray_dirs, float* cam_loc, hold_stream_t stream);
/* forward LBS of query points, cano -> deformed: x' = (sum_j w_j T_j) [x;1] (skinning(inverse=False),
 * code/src/model/mano/deformer.py:145-170; n_bones = 1: ObjectDeformer.forward, obj/deformer.py:10-31). */
int hold_skin_fwd(const float* x, int32_t ldx, int64_t P, int64_t pts_per_frame, const float* w, const float* tfs,
                  int32_t n_bones, float* xd, int32_t ldxd, hold_stream_t stream);
int hold_invskin_bwd(const float* xc, int32_t ldxc, const float* w, const float* tfs, int32_t n_bones, int64_t P,
                     int64_t pts_per_frame, const float* xcbar, int32_t ldxb, float* dtfs /* [B][n_bones][16], += */,
                     hold_stream_t stream);
/* canonical normal n = normalize(g . J^-1, eps 1e-6), J = sum_j w_j T_j[:3,:3]
 * (extract_features, code/src/engine/volsdf_utils.py:68-81, :100-102) and its backward. */
int hold_normal_fwd(const float* g, int32_t ldg, const float* w, const float* tfs, int32_t n_bones, int64_t P,
                    int64_t pts_per_frame, float* n_out, int32_t ldn, hold_stream_t stream);
int hold_normal_bwd(const float* g, int32_t ldg, const float* w, const float* tfs, int32_t n_bones, int64_t P,
                    int64_t pts_per_frame, const float* nbar, int32_t ldnb, float* gbar, int32_t ldgb,
                    float* dtfs /* += */, hold_stream_t stream);
/* helpers: per-frame column sums (+=), per-frame broadcast into columns, strided column copy */
int hold_frame_colsum(const float* X, int32_t ldx, int32_t col0, int32_t ncols, int64_t P, int64_t pts_per_frame,
                      float* out /* [B][ncols], += */, hold_stream_t stream);
int hold_frame_bcast(const float* src, int32_t ncols, int64_t P, int64_t pts_per_frame, float* out, int32_t ldo,
                     int32_t col0, hold_stream_t stream);
int hold_copy_cols(const float* src, int32_t lds, float* dst, int32_t ldd, int32_t ncols, int64_t P,
                   int32_t accumulate, hold_stream_t stream);

/* NeRF++ inverted-sphere re-parameterisation (background.py:102-135): out[p] = (unit xyz, 1/r) */
int hold_bg_points(const float* cam_loc, const float* ray_dirs, const float* depth, int32_t S, int64_t n_rays, float R,
                   float* out, int32_t ldo, hold_stream_t stream);
/* out[p] = A[p][:K] . w + b (+ *b_dev when b_dev != NULL: a bias that is a trained parameter stays on the device)
 * (sdf row of ImplicitNet's last layer) */
int hold_rowdot(const float* A, int32_t lda, const float* w, int32_t K, float b, const float* b_dev, int64_t P, float* out,
                int32_t ldo, hold_stream_t stream);
/* t[p][n] = w[n] * softplus'(h[p][n]): seed of the d sdf/d x reverse sweep (volsdf_utils.py:89-96) */
int hold_seed_dsp(const float* h, int32_t ldh, const float* w, int32_t N, int64_t P, float* t, int32_t ldt,
                  hold_stream_t stream);
/* out[n] += sum_p X[p][n] */
int hold_colsum(const float* X, int32_t ldx, int32_t N, int64_t P, float* out, hold_stream_t stream);
/* weighted column sums, deterministic (two passes, fixed grid): out[n] (+)= sum_p w[p] * X[p][n]  (w NULL: plain sums).
 * N % 4 == 0, N <= 1024, X 16-byte aligned rows; workspace >= hold_wcolsum_workspace_floats(N) floats.  The rank-1
 * companion of hold_wgrad: the gradient of a single weight row whose cotangent is a [P] vector (the sdf row of lin8,
 * code/src/networks/shape_net.py:118-130). */
int64_t hold_wcolsum_workspace_floats(int32_t N);
/* The 3-output colour head of the rendering nets (last Linear + sigmoid, code/src/networks/texture_net.py:95-101;
 * background: code/src/model/renderables/background.py:62-70) as HBM-streaming kernels:
 *   fwd: out[p][c] = act(sum_k A[p][k] W[c][k] + bias[c]), c < 3 (act = sigmoid if `sigmoid`, else identity)
 *   bwd: given dy[p][c] (cotangent of the pre-activation), in ONE pass over R (= the ReLU output that fed the head):
 *        rr[p][k] = (R[p][k] > 0) ? sum_c dy[p][c] W[c][k] : 0,  dW[c][k] (+)= sum_p dy[p][c] R[p][k],
 *        db4[c] (+)= sum_p dy[p][c]  (db4 has room for 4 floats; element 3 is written as 0 / left unchanged)
 * K % 4 == 0, 16-byte aligned rows; workspace >= hold_head3_workspace_floats(K); deterministic reductions. */
int64_t hold_head3_workspace_floats(int32_t K);
int hold_head3_fwd(const float* A, int32_t lda, const float* W, int32_t ldw, const float* bias, int32_t K, int64_t P,
                   float* out, int32_t ldo, int32_t sigmoid, hold_stream_t stream);
int hold_head3_bwd(const float* dy, int32_t ldy, const float* R, int32_t ldr, const float* W, int32_t ldw, int32_t K,
                   int64_t P, float* rr, int32_t ldrr, float* dW, int32_t lddw, float* db4, int32_t accumulate,
                   float* workspace, hold_stream_t stream);
int hold_wcolsum(const float* X, int32_t ldx, int32_t N, int64_t P, const float* w, float* out, int32_t accumulate,
                 float* workspace, hold_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * VolSDF error-bound sampler (hold_amd/csrc/sampler.hip) -- ErrorBoundSampler.get_z_vals,
 * code/src/engine/ray_sampler.py:128-352; one call per numbered step of the reference loop.
 * Windows z/sdf are [n_rays][ld] with the first S entries valid and sorted.
 * ---------------------------------------------------------------------------------------- */
/* :54-80 + :152-156 : far = sphere exit, z = 128 uniform (stratified with t_rand in training),
 * beta = sqrt(sum dists^2 / (4 log(1+eps))).  *err_flag = 1 if a ray misses the sphere (:16-18). */
int hold_sampler_init(const float* cam_loc, const float* ray_dirs, int64_t n_rays, float R, float near, int32_t n0,
                      float eps, const float* t_rand, float* z, int32_t ldz, float* beta, float* far_out,
                      int32_t* err_flag, hold_stream_t stream);
/* :179-220 : (scatter sdf of the previous round's new samples), d* bound, beta bisection;
 * atomically maxes beta into *maxbeta_bits (float bits; caller zeroes it) for the global test at :244 */
int hold_sampler_beta(const float* z, float* sdf, int32_t ld, int32_t S, int64_t n_rays, const float* sdf_new,
                      const int32_t* slot, int32_t n_new, float* beta, float beta0, float eps, int32_t beta_iters,
                      uint32_t* maxbeta_bits, hold_stream_t stream);
/* :223-311 : pdf (error bound if more, opacity weights otherwise) -> CDF -> inverse CDF at u
 * ([n_new] shared when u_stride = 0, else [n_rays][n_new]); if more, stable-merges the samples into
 * the window (slot_out[j] = merged index of sample j; sdf at those slots is filled next round). */
int hold_sampler_sample(float* z, float* sdf, int32_t ld, int32_t S, int64_t n_rays, const float* beta, int32_t more,
                        float add_tiny, const float* u, int64_t u_stride, int32_t n_new, float* samples_out,
                        int32_t* slot_out, hold_stream_t stream);
/* :313-336 : sort([z_samples, near, far, z[:, idx_extra]]) */
int hold_sampler_final(const float* z_samples, int32_t ns, const float* z, int32_t ld, const int32_t* idx_extra,
                       int32_t nx, const float* far, float near, int64_t n_rays, float* out, int32_t ldo,
                       hold_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Compositor (hold_amd/csrc/composite.hip): LaplaceDensity (code/src/engine/density.py:21-30),
 * density2weight (code/src/engine/volsdf_utils.py:220-251), integrate (rendering.py:18-22),
 * merge_factors with the off-by-one trim (code/src/hold/hold_utils.py:76-121) and
 * volumetric_render (:243-271) for every node and for the merged composite.
 * Packed per-ray outputs: [rgb3, sum_w, normal3, depth, bg_weight, 0,0,0].
 * ---------------------------------------------------------------------------------------- */
#define HOLD_MAX_NODES 3
#define HOLD_RENDER_W 12
typedef struct hold_composite_desc {
  int32_t n_nodes, S;
  int64_t n_rays;
  const float* z[HOLD_MAX_NODES];      /* [n_rays][S] sorted                      */
  const float* sdf[HOLD_MAX_NODES];    /* [n_rays][S]                             */
  const float* color[HOLD_MAX_NODES];  /* [n_rays*S][ldc]                         */
  const float* normal[HOLD_MAX_NODES]; /* [n_rays*S][ldn]                         */
  int32_t ldc[HOLD_MAX_NODES], ldn[HOLD_MAX_NODES], class_id[HOLD_MAX_NODES];
  float beta[HOLD_MAX_NODES];          /* |beta_param| + beta_min                 */
  float* out_node[HOLD_MAX_NODES];     /* [n_rays][HOLD_RENDER_W]                 */
  float* out_comp;                     /* [n_rays][HOLD_RENDER_W]                 */
  float* out_sem;                      /* [n_rays][4] composite fg semantics      */
  float* out_w;                        /* [n_rays][n_nodes*S-2*n_nodes+1] or NULL */
  float* out_zmerge;                   /* same shape or NULL                      */
  /* backward only */
  const float* d_node[HOLD_MAX_NODES]; /* [n_rays][HOLD_RENDER_W]                 */
  const float* d_comp;
  const float* d_sem;
  float* d_sdf[HOLD_MAX_NODES];        /* [n_rays][S]                             */
  float* d_color[HOLD_MAX_NODES];      /* [n_rays*S][3]                           */
  float* d_normal[HOLD_MAX_NODES];     /* [n_rays*S][3]                           */
  float* d_beta;                       /* [HOLD_MAX_NODES], +=                    */
  /* forward, optional: per-node compositing weights `<node>.fg_weights` of volumetric_render (hold_utils.py:259-262) */
  float* out_w_node[HOLD_MAX_NODES];   /* [n_rays][S] or NULL                     */
} hold_composite_desc;
int hold_composite_fwd(const hold_composite_desc* d, hold_stream_t stream);
int hold_composite_bwd(const hold_composite_desc* d, hold_stream_t stream);
/* Background.bg_volume_rendering + integrate (code/src/model/renderables/background.py:95-100,137-165) */
int hold_bg_composite_fwd(const float* z_desc, const float* sdf, const float* rgb, int32_t ld_rgb, int32_t S,
                          int64_t n_rays, float* out_rgb, float* w_out, hold_stream_t stream);
int hold_bg_composite_bwd(const float* z_desc, const float* sdf, const float* rgb, int32_t ld_rgb, int32_t S,
                          int64_t n_rays, const float* d_out, float* d_sdf, float* d_rgb, hold_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * MANO forward LBS + server post-processing (hold_amd/csrc/mano.hip), one workgroup per frame.
 * lbs() code/src/utils/external/lbs.py:139-251, MANO.forward body_models.py:601-685 (pose_mean, fingertips),
 * GenericServer.forward code/src/model/mano/server.py:62-99 (scale/transl, tfs . tfs_c_inv).
 * Outputs: verts [B][778][3], jnts [B][21][3] (nullable), tfs [B][16][4][4], v_posed [B][778][3] (nullable).
 * Backward: given d_tfs and/or d_verts (either may be NULL) -> d_pose [B][48], d_betas [B][10], d_transl [B][3].
 * ---------------------------------------------------------------------------------------- */
typedef struct hold_mano_model {
  const float* v_template;  /* [778][3]      */
  const float* shapedirs;   /* [778][3][10]  */
  const float* posedirs;    /* [135][2334]   */
  const float* J_regressor; /* [16][778]     */
  const int32_t* parents;   /* [16], parents[0] = -1 */
  const float* lbs_weights; /* [778][16]     */
  const float* pose_mean;   /* [48]          */
  const float* tfs_c_inv;   /* [16][4][4] or NULL for absolute transforms */
} hold_mano_model;
int hold_mano_lbs_fwd(const hold_mano_model* m, int32_t n_frames, const float* betas, const float* full_pose,
                      const float* scene_scale, const float* transl, float* verts, float* jnts, float* tfs,
                      float* v_posed, hold_stream_t stream);
int hold_mano_lbs_bwd(const hold_mano_model* m, int32_t n_frames, const float* betas, const float* full_pose,
                      const float* scene_scale, const float* transl, const float* d_tfs, const float* d_verts,
                      float* d_pose, float* d_betas, float* d_transl, hold_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Fused SDF-only ImplicitNet (hold_amd/csrc/fused_sdf.hip): embedding + 8 softplus layers + sdf row in ONE
 * launch, activations resident in LDS, for the sampler's no-grad queries
 * (shape_net.py:84-130 via volsdf_utils.py:150-169 inside ray_sampler.py:169-178).
 * wpack: weights in MFMA-fragment order, hold_fused_sdf_pack_floats() floats:
 *   for layer l (K_l = 40 for l = 0, else 256; layer 3 padded to 256 rows, layer 4 pre-scaled by 1/sqrt 2):
 *   [K_l/8 chunks][8 n-tiles][2 halves h][32 rows i][4] = W_l[32*nt + i][8*chunk + 4*h + c]
 * bias [8][256]; w8 [256] + b8 = sdf row of the last layer; barf_w [39] or NULL.
 * ---------------------------------------------------------------------------------------- */
int64_t hold_fused_sdf_pack_floats(void);
int hold_fused_sdf(const float* xc, int32_t ldx, int64_t P, const float* wpack, const float* bias, const float* w8,
                   float b8, const float* barf_w, float* sdf, int32_t ld_sdf, hold_stream_t stream);

/* Split-precision variant (selected by hold_amd.set_precision("f32x6"), the default since its round-2 hardware
 * validation: 1.4e-6 max abs vs the fp32-MFMA kernel, 176 vs 124 TFLOP/s fp32-equivalent): the same
 * contract as hold_fused_sdf with split-precision arithmetic -- every fp32 operand is the exact sum of three bf16 limbs
 * (limb t = bf16 rounding of what limbs < t left over), six of the nine limb products on v_mfma_f32_32x32x16_bf16 with
 * fp32 accumulation (dropped terms <= 2^-24 relative; scripts/split_precision_study.py).
 * wpack_x6: hold_fused_sdf_x6_pack_bytes() bytes of bf16:
 *   for layer l (K_l = 48 for l = 0 (40 zero-padded), else 256; rows/scaling as for hold_fused_sdf):
 *   [K_l/16 steps][3 limbs t][8 n-tiles][2 halves h][32 rows i][8] = limb_t(W_l)[32*nt + i][16*step + 8*h + e] */
int64_t hold_fused_sdf_x6_pack_bytes(void);
int hold_fused_sdf_x6(const float* xc, int32_t ldx, int64_t P, const void* wpack_x6, const float* bias, const float* w8,
                      float b8, const float* barf_w, float* sdf, int32_t ld_sdf, hold_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Register-resident trunk (hold_amd/csrc/rmlp.hip): lin0..lin7 of ImplicitNet.forward (shape_net.py:84-130) with the
 * layer outputs kept in the wave's accumulator registers -- one wave per SIMD owns 32 points for the whole network, the
 * previous layer's accumulators become the next layer's MFMA B operand after softplus + limb split in registers, the weight
 * limbs are the only stream (LDS-DMA ring shared by the four waves of a workgroup).  Split-precision arithmetic of
 * hold_fused_sdf_x6 (three bf16 limbs of both operands, six products, fp32 accumulation).
 * wpack_r6: hold_trunk_r6_pack_bytes() bytes of bf16, [115 k steps][8 n-tiles nt][3 limbs t][2 halves h][32 rows i][8 e]:
 *   k steps 0..2   = layer 0 (K = 48, columns 39.. zero):  limb_t(W_0)[32 nt + i][16 step + 8 h + e]
 *   k steps 3 + 16 (l - 1) + j, j = 0..15 = layer l = 1..7 (rows / scaling as for hold_fused_sdf):
 *                    limb_t(W_l)[32 nt + i][32 (j / 2) + 16 (j % 2) + 8 (e / 4) + 4 h + e % 4]
 *   (the order in which a lane holds the previous layer's outputs after v_mfma_f32_32x32x16_bf16).
 * hold_fused_sdf_r6: the contract of hold_fused_sdf_x6 (the sampler's SDF query), except that b8 -- the bias of lin8's sdf
 *   row, a trained parameter -- is read on the device (pointer to one float): a host copy costs the training step a
 *   stream drain per node.
 * hold_trunk_r6: training forward, h[l] [P][ldh] (l = 0..7) = softplus outputs of lin0..lin7, columns 217..255 of h[3] =
 *   the embedding (skip concat); replaces embed + hold_chain_x6(SOFTPLUS) of the forward trunk.
 * ---------------------------------------------------------------------------------------- */
int64_t hold_trunk_r6_pack_bytes(void);
int hold_fused_sdf_r6(const float* xc, int32_t ldx, int64_t P, const void* wpack_r6, const float* bias, const float* w8,
                      const float* b8, const float* barf_w, float* sdf, int32_t ld_sdf, hold_stream_t stream);
int hold_trunk_r6(const float* xc, int32_t ldx, int64_t P, const void* wpack_r6, const float* bias, const float* barf_w,
                  float* const* h, int32_t ldh, hold_stream_t stream);
/* CONDITIONAL launches of the two (the device-side fallback of the f16x3 entry points below; no host read involved):
 * `guard` = 4 x uint32 of device memory, zeroed once by the caller, used by ONE stream at a time.  guard == NULL: exactly
 * hold_fused_sdf_r6 / hold_trunk_r6.  Otherwise every workgroup exits at once unless guard[0] != 0; if the kernel ran, the
 * last workgroup to finish adds 1 to guard[2] (a monotone count the caller may read whenever it likes) and clears guard[0]
 * (and guard[1], the arrival counter it uses). */
int hold_fused_sdf_r6_if(const float* xc, int32_t ldx, int64_t P, const void* wpack_r6, const float* bias, const float* w8,
                         const float* b8, const float* barf_w, float* sdf, int32_t ld_sdf, uint32_t* guard,
                         hold_stream_t stream);
int hold_trunk_r6_if(const float* xc, int32_t ldx, int64_t P, const void* wpack_r6, const float* bias, const float* barf_w,
                     float* const* h, int32_t ldh, uint32_t* guard, hold_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * The same trunk in the TWO-LIMB fp16 arithmetic "f16x3" (hold_amd/csrc/rmlp_h3.hip; shape_net.py:84-130): every operand
 * x, scaled by an exact power of two s, is hi + lo with hi = RN_f16(s x), lo = RN_f16(s x - hi); the products
 * hi_w hi_x + hi_w lo_x + lo_w hi_x go to v_mfma_f32_32x32x16_f16 with fp32 accumulation -- three matrix instructions per
 * product instead of six, 16 KiB of weight limbs per k step instead of 24; error against fp64 as hold_fused_sdf_r6
 * (tests/test_rmlp_gpu.py holds it to <= 1.5 x that kernel's).  Activations are scaled by hold_trunk_h3_act_scale() = 2^6
 * inside the kernel; weights by a per-matrix s_w[l] = 2^k chosen by the caller with max |W_l| s_w[l] < 2^15 ([2^13, 2^14)
 * recommended).
 * OVERFLOW GUARD: a (scaled) activation >= 65504, i.e. an activation >= 1023.5, has no fp16 representation.  The kernels keep
 *   the exact maximum of every value they split and set guard[0] when one left the range (guard: the 4 words of
 *   hold_*_r6_if above; NULL = unreported).  With wpack_r6 / bias (hold_trunk_r6's operands for the same weights; both or
 *   neither; they require guard) the entry point enqueues the f32x6 kernel right behind as a conditional launch: the result
 *   of an overflowing call is then hold_fused_sdf_r6's / hold_trunk_r6's bit for bit, guard[2] counts such calls and
 *   guard[0] is clear again -- never a silent infinity, and no host synchronisation.
 * wpack_h3: hold_trunk_h3_pack_bytes() bytes of fp16, [116 k steps][8 n-tiles nt][2 limbs t][2 halves h][32 rows i][8 e]:
 *   k steps 0..3 = layer 0 with K = 64 (columns 39.. zero: FOUR k steps, one more than wpack_r6's K = 48 -- the ring slot of a
 *   k step is then the same in every layer), limb_t(s_w[0] W_0)[32 nt + i][16 step + 8 h + e]; k steps 4 + 16 (l - 1) + j =
 *   layer l = 1..7 in the rows and k order of wpack_r6, limb_t(s_w[l] W_l).
 * bias_scaled: [8][256] = bias_l s_w[l] hold_trunk_h3_act_scale();  c3: [8] = 1 / s_w[l] (device memory).
 * hold_fused_sdf_h3 / hold_trunk_h3: otherwise the contracts of hold_fused_sdf_r6 / hold_trunk_r6 (outputs in fp32, unscaled).
 * ---------------------------------------------------------------------------------------- */
int64_t hold_trunk_h3_pack_bytes(void);
float hold_trunk_h3_act_scale(void);
int hold_fused_sdf_h3(const float* xc, int32_t ldx, int64_t P, const void* wpack_h3, const float* bias_scaled,
                      const float* c3, const float* w8, const float* b8, const float* barf_w, float* sdf, int32_t ld_sdf,
                      uint32_t* guard, const void* wpack_r6, const float* bias, hold_stream_t stream);
int hold_trunk_h3(const float* xc, int32_t ldx, int64_t P, const void* wpack_h3, const float* bias_scaled, const float* c3,
                  const float* barf_w, float* const* h, int32_t ldh, uint32_t* guard, const void* wpack_r6,
                  const float* bias, hold_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * LDS-resident layer chains of the ImplicitNet for the TRAINING path (hold_amd/csrc/chain.hip): up to 8 consecutive
 * 256-wide layers of one sweep in one launch; the running activation stays in LDS, per-layer side inputs are read
 * from and per-layer results written to HBM directly from the accumulators.  Replaces n_layers hold_gemm_nt calls of
 *   SOFTPLUS: ImplicitNet.forward lin0..lin7          (shape_net.py:108-125), stores h_l
 *   DSP     : the descending sweeps torch.autograd derives from it (d sdf/d a_l for the canonical normal,
 *             volsdf_utils.py:79-93, and the first-order backward), v_{l-1} = (M_j v_l) * sp'(aux1_j) [+ aux2_j]
 *   DBWD    : the ascending second-order sweep of create_graph=True (volsdf_utils.py:87):
 *             tb = M_j vb ; out_j = tb * sp'(aux1_j) ; out2_j = 100 * tb * aux2_j * (1 - sp'(aux1_j))
 * in [P][8*first_chunks] (first_chunks = 5: K = 40, or 32: K = 256); all other matrices [P][256] with row stride ld.
 * wpack: hold_chain_pack_floats(first_chunks, n_layers) floats, layer j = matrix M_j [256 out][K_j in] (zero padded) in
 *   MFMA-fragment order [K_j/8 chunks][8 n-tiles][2 halves h][32 rows i][4] = M_j[32*nt + i][8*chunk + 4*h + c]
 *   (the layout of hold_fused_sdf's wpack).
 * skip_layer (or -1): chain layer whose output columns 217.. are special: SOFTPLUS / DBWD replace them by side[:, 0..38]
 *   (side [P][40], row stride ld_side; out2 gets 0); DSP stores the raw products (no sp' factor, no aux2) there.
 * ---------------------------------------------------------------------------------------- */
enum hold_chain_mode { HOLD_CHAIN_SOFTPLUS = 0, HOLD_CHAIN_DSP = 1, HOLD_CHAIN_DBWD = 2 };
typedef struct {
  int64_t P;
  int32_t mode, n_layers, first_chunks, skip_layer;
  const float* in;
  int32_t ld_in;
  const float* side;
  int32_t ld_side;
  const float* wpack;
  int32_t ld;
  const float* bias[8];
  const float* aux1[8];
  const float* aux2[8];
  float* out[8];
  float* out2[8];
  int32_t skip_out; /* first special output column of skip_layer: 0 = 217 (the foreground nets); hold_chain_r6 (DSP) also
                       takes 172, the background net's (256 - 84 embedding columns, background.py:44-58); others reject it */
} hold_chain_desc;
int64_t hold_chain_pack_floats(int32_t first_chunks, int32_t n_layers);
int hold_chain(const hold_chain_desc* d, hold_stream_t stream);
/* Split-precision variant: the same descriptor and semantics, the layer products as three-limb bf16 splits on
 * v_mfma_f32_32x32x16_bf16 (six limb products, fp32 accumulation; activations split from fp32 LDS as they are fetched).
 * d->wpack then points at hold_chain_x6_pack_bytes(first_chunks, n_layers) bytes of bf16 limbs: layer j, K_j = 48 for
 * first_chunks = 5 (columns 40..47 zero) else 256,
 *   [K_j/16 steps][3 limbs t][8 n-tiles][2 halves h][32 rows i][8] = limb_t(M_j)[32*nt + i][16*step + 8*h + e]
 * (the layout of hold_fused_sdf_x6's wpack_x6, which IS the pack of the forward-type sweeps). */
int64_t hold_chain_x6_pack_bytes(int32_t first_chunks, int32_t n_layers);
int hold_chain_x6(const hold_chain_desc* d, hold_stream_t stream);

/* Register-resident variants of the backward sweeps (hold_amd/csrc/rchain.hip; structure of hold_trunk_r6, side inputs and
 * results moved as whole 128-byte lines through swizzled LDS tiles): the descriptor and semantics of hold_chain_x6 for
 *   mode DSP  (n_layers 7, first_chunks 32, skip_layer 3; out[] entries may be NULL, aux2 optional).  d->wpack =
 *             hold_chain_r6_pack_bytes() bytes: [7 x 16 k steps][8 nt][3 limbs][2 h][32 i][8 e], every layer
 *             limb_t(M_j)[32 nt + i][32 (s / 2) + 16 (s % 2) + 8 (e / 4) + 4 h + e % 4] (the k order of hold_trunk_r6);
 *   mode DBWD (n_layers 8, first_chunks 5, skip_layer 3; aux1, aux2, out, out2 all given).  d->wpack = the weight stream of
 *             hold_trunk_r6 (hold_trunk_r6_pack_bytes() bytes).  d->side is NOT read: the skip layer's side columns (the
 *             39 columns of the chain input that become columns 217.. of out[3]) are taken from aux2[3][:, 217..255],
 *             where the caller stores them before the launch (out2[3] is 0 in those columns, as with hold_chain_x6). */
int64_t hold_chain_r6_pack_bytes(void);
int hold_chain_r6(const hold_chain_desc* d, hold_stream_t stream);
/* ... as a CONDITIONAL launch (guard: the 4 words of hold_fused_sdf_r6_if; NULL = hold_chain_r6): the fallback of hold_chain_h3 */
int hold_chain_r6_if(const hold_chain_desc* d, uint32_t* guard, hold_stream_t stream);

/* The same sweeps in the TWO-LIMB fp16 arithmetic "f16x3" (hold_amd/csrc/rchain_h3.hip; the arithmetic of hold_trunk_h3):
 * 24 matrix instructions per k step instead of 48.  The weights scale per matrix at pack time (s_w[j] = 2^k with
 * max |M_j| s_w[j] in [2^13, 2^14)); the running cotangent -- the B operand -- carries a power-of-two scale PER POINT that the
 * kernel predicts layer by layer from the exact maximum of the point's values in the layer before (2^9 of headroom; the chain
 * input's maximum is exact).
 *   mode DSP : d->wpack = hold_chain_h3_pack_bytes() bytes of fp16, [7 x 16 k steps][8 nt][2 limbs][2 h][32 i][8 e], the rows and
 *              k order of hold_chain_r6's stream, limb_t(s_w[j] M_j); skip_out 217 (0 = 217), or 172 without aux2 as hold_chain_r6;
 *   mode DBWD: d->wpack = the stream of hold_trunk_h3 (hold_trunk_h3_pack_bytes() bytes; layer 0 = four k steps).
 * c3: [n_layers] = 1 / s_w[j] of the chain layers, device memory.  Otherwise the contract of hold_chain_r6.
 * OVERFLOW GUARD (as hold_fused_sdf_h3): a point whose values grow by more than the headroom within one layer would leave
 * fp16's range; the kernel then sets guard[0] (guard NULL: unreported), and with wpack_r6 (hold_chain_r6's stream for the same
 * matrices; requires guard) the entry point enqueues hold_chain_r6_if behind it: the results of such a launch are
 * hold_chain_r6's, guard[2] counts it, no host synchronisation. */
int64_t hold_chain_h3_pack_bytes(void);
int hold_chain_h3(const hold_chain_desc* d, const float* c3, uint32_t* guard, const void* wpack_r6, hold_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * One 256-wide layer with the register-resident structure (hold_amd/csrc/rgemm.hip): the rendering net's layers
 * (code/src/networks/texture_net.py:95-101), their input gradients and lin8's feature rows (shape_net.py:128-130) --
 * C[P][256] (row stride ldc) = epi(A[P][K] . W^T + bias), epilogue 0 none, 1 ReLU, 2 multiply by (aux > 0) (no bias);
 * arithmetic of hold_gemm_nt_x6.  K a multiple of 16 in 256 .. 320.
 * wpack: hold_gemm_r6_pack_bytes(K) bytes of bf16, [KS k steps j][8 n-tiles nt][3 limbs t][2 halves h][32 rows i][8 e] =
 *   limb_t(W)[32 nt + i][16 j + 8 (e / 4) + 4 h + e % 4], KS = 4 ceil(K / 64), zero for columns >= K and rows >= N.
 * 32-bit offsets: P * max(lda, ldc, ld_aux) * 4 < 2^32 (the caller splits by rows).
 * ---------------------------------------------------------------------------------------- */
int64_t hold_gemm_r6_pack_bytes(int32_t K);
int hold_gemm_r6(const float* A, int32_t lda, int64_t P, const void* wpack, int32_t K, const float* bias, int32_t epilogue,
                 const float* aux, int32_t ld_aux, float* C, int32_t ldc, hold_stream_t stream);
/* ... with the per-row maxima of C as a second output (amax_out [P] = max |C[p][:]|, or NULL) and as a CONDITIONAL launch (guard: the
 * 4 words of hold_fused_sdf_r6_if; NULL = always run): the fallback of hold_gemm_h3 */
int hold_gemm_r6_if(const float* A, int32_t lda, int64_t P, const void* wpack, int32_t K, const float* bias, int32_t epilogue,
                    const float* aux, int32_t ld_aux, float* C, int32_t ldc, float* amax_out, uint32_t* guard,
                    hold_stream_t stream);

/* hold_gemm_r6 in the TWO-LIMB fp16 arithmetic "f16x3" (hold_amd/csrc/rgemm_h3.hip): 24 matrix instructions per k step instead of 48.
 * wpack_h3: hold_gemm_h3_pack_bytes(K) bytes of fp16, [KS k steps j][8 n-tiles nt][2 limbs t][2 halves h][32 rows i][8 e] =
 *   limb_t(s_w W)[32 nt + i][16 j + 8 (e / 4) + 4 h + e % 4] (the rows and k order of hold_gemm_r6's stream), s_w = 2^k with
 *   max |W| s_w in [2^13, 2^14); c3 = 1 / s_w, one float in device memory.
 * Every ROW of A (a point) is scaled by its own power of two: max(amax_in[p], amax_floor) goes to [2^12, 2^13).  amax_in [P] (or
 *   NULL) = an upper bound of |A[p][c]| over the columns its producer wrote -- exact when it is the amax_out of the launch that wrote
 *   A -- and amax_floor bounds the remaining columns (with amax_in == NULL: every column; must then be > 0).  amax_out [P] (or NULL)
 *   receives max |C[p][:]| for the next launch.  Values up to 2^3 x that bound are representable; beyond it the overflow guard of
 *   hold_fused_sdf_h3 applies (guard, wpack_r6 = hold_gemm_r6's stream of the same W: conditional recomputation in f32x6, C and
 *   amax_out then hold_gemm_r6_if's). */
int64_t hold_gemm_h3_pack_bytes(int32_t K);
int hold_gemm_h3(const float* A, int32_t lda, int64_t P, const void* wpack_h3, const float* c3, int32_t K, const float* bias,
                 int32_t epilogue, const float* aux, int32_t ld_aux, float* C, int32_t ldc, const float* amax_in, float amax_floor,
                 float* amax_out, uint32_t* guard, const void* wpack_r6, hold_stream_t stream);
/* ... with the ReLU masks of the rendering net as BITS (round 6).  texture_net.py:95-101 applies ReLU after lin0..lin3, and autograd's
 * backward multiplies the cotangent by (activation > 0): the product used to stream the [P][256] fp32 activation a second time for that
 * (epilogue 2's aux, 1 KiB per point and layer).  relu_bits_out (epilogue 1; [P][8] dwords, 16-byte aligned, or NULL): bit n of row p =
 * (C[p][n] > 0), written from the values in the epilogue registers.  mask_bits_in (epilogue 2; such a matrix, or NULL): the mask is
 * taken from it -- 32 bytes per point, and the launch has one side matrix instead of two; aux may then be NULL unless wpack_r6 is given
 * (the f32x6 fallback still masks by aux).  When an epilogue-1 launch falls back, a third conditional launch rebuilds its bits from the
 * recomputed C (it compares guard[2] with guard[3], the count it last saw).  Otherwise hold_gemm_h3. */
int hold_gemm_h3_bits(const float* A, int32_t lda, int64_t P, const void* wpack_h3, const float* c3, int32_t K, const float* bias,
                      int32_t epilogue, const float* aux, int32_t ld_aux, float* C, int32_t ldc, const float* amax_in,
                      float amax_floor, float* amax_out, uint32_t* relu_bits_out, const uint32_t* mask_bits_in, uint32_t* guard,
                      const void* wpack_r6, hold_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Weight normalisation of all layers of a net in one launch per direction (hold_amd/csrc/wnorm.hip): every Linear of
 * ImplicitNet / RenderingNet is torch.nn.utils.weight_norm'ed (code/src/networks/shape_net.py:79-80, texture_net.py:40-41),
 * w = v * (g / ||v||_row).  fwd: layers[l].w [rows][ldw] out.  bwd: dw in ([rows][ldw]; NULL = this layer received no
 * gradient and is skipped), dv [rows][ldv] / dg [rows] out -- stored, or with accumulate != 0 ADDED (the optimiser's flat
 * gradient bucket is written directly: no per-parameter accumulation launches).  t = <dw, v>, n = ||v||:
 * dg = t / n, dv = dw g / n - v t g / n^3.
 * ---------------------------------------------------------------------------------------- */
#define HOLD_WN_MAX_LAYERS 16
typedef struct {
  const float* v;  /* [rows][ldv] direction parameter (weight_v) */
  const float* g;  /* [rows] magnitude parameter (weight_g) */
  float* w;        /* fwd out [rows][ldw] */
  const float* dw; /* bwd in  [rows][ldw] or NULL */
  float* dv;       /* bwd out [rows][ldv] */
  float* dg;       /* bwd out [rows] */
  int32_t rows, cols, ldv, ldw;
} hold_wn_layer;
typedef struct {
  int32_t n_layers, accumulate;
  hold_wn_layer layers[HOLD_WN_MAX_LAYERS];
} hold_wn_desc;
int hold_weight_norm_fwd(const hold_wn_desc* d, hold_stream_t stream);
int hold_weight_norm_bwd(const hold_wn_desc* d, hold_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Training loss-target geometry without kaolin (hold_amd/csrc/geometry.hip):
 * signed distance (negative inside) of pts [B][P][3] to a closed triangle mesh, replacing
 * kaolin.metrics.trianglemesh.point_to_mesh_distance + kaolin.ops.mesh.check_sign of compute_mano_cano_sdf /
 * check_off_in_surface_points_cano_mesh (code/src/engine/volsdf_utils.py:172-217).
 * verts [B][V][3] (or [V][3] with verts_shared = 1), faces [F][3] int32.  cull_dist > 0 with aabb [B][6]
 * (min xyz, max xyz of the frame's vertices): points farther than cull_dist from the box skip the face loop and get
 * that (lower-bound, positive) distance -- enough for the reference's off-surface test (min distance > threshold).
 * ---------------------------------------------------------------------------------------- */
int hold_mesh_sdf(const float* pts, int32_t B, int64_t P, const float* verts, int32_t verts_shared, int32_t V,
                  const int32_t* faces, int32_t F, float cull_dist, const float* aabb, float* sd, hold_stream_t stream);

/* per-ray off-surface test of check_off_in_surface_points_cano_mesh (code/src/engine/volsdf_utils.py:189-217):
 * off[r] = 1 iff min over the ray's S samples of the signed distance to the (single, canonical) closed mesh > thr, using
 * the node-SDF grid + per-cell triangle lists built by hold_amd/geometry.py:MeshIndex (node_sdf [G^3] x-major at
 * origin + h * index with h * sqrt(3) < thr; cell_start [(G-1)^3 + 1], cell_tris [*] = triangles whose thr-dilated
 * bounding box touches the cell).  xc [n_rays * S][ldx] ray-major canonical points.  Same decisions as evaluating
 * hold_mesh_sdf on every sample (the band the grid cannot decide is resolved by exact point-triangle distances). */
int hold_ray_off_surface(const float* xc, int32_t ldx, int64_t n_rays, int32_t S, const float* node_sdf, int32_t G,
                         float ox, float oy, float oz, float h, float thr, const int32_t* cell_start,
                         const int32_t* cell_tris, const float* verts, const int32_t* faces, uint8_t* off,
                         hold_stream_t stream);

/* ---- step tail: per-pixel loss terms and the optimiser step on one flat fp32 bucket ----
 * hold_pixel_loss_fwd/bwd: the ray-wise terms of Loss.forward (code/src/hold/loss.py:17-93) in one launch each --
 *   sums[0] = sum |rgb - gt| over rows without NaN, sums[2] = number of such rows (loss_terms.get_rgb_loss :14-20)
 *   sums[1] = sum (semantics - onehot(class(gt_mask)))^2     (get_sem_loss :67-98; class bounds 25 / 100 / 200)
 *   sums[3+2i], sums[4+2i] = sum / count of node i's mask_prob over its off-surface rays (get_opacity_sparse_loss :44-56)
 * the host divides by the pixel counts and applies the schedule weights; bwd takes g = dL/d sums and writes
 * d rgb [N,3], d semantics [N,4], d mask_prob_i [N] (rows outside the index get 0).
 * hold_sumsq + hold_adam_step: torch.optim.Adam(eps=1e-8) of code/src/hold/hold.py:79-101 over a flat bucket whose
 * first n_low elements (the per-frame pose tables) use lr_low = 0.1 lr, with clip_grad_norm_(0.5) of code/train.py:30
 * folded in: grads are scaled by grad_mul * min(1, clip_norm / (sqrt(sumsq[0]) * |grad_mul| + 1e-6)); sumsq is a
 * DEVICE scalar (no host sync).  step >= 1 is the 1-based Adam step used for the bias corrections. */
typedef struct hold_loss_nodes {
  const float* mask_prob[3];
  const uint8_t* off[3]; /* bool index_off_surface per ray, NULL = node has no target yet */
  float* d_mask[3];      /* bwd only */
} hold_loss_nodes;
int64_t hold_reduce_workspace_floats(void); /* scratch of the two-pass (deterministic, atomic-free) reductions */
int hold_pixel_loss_fwd(const float* rgb, const float* gt_rgb, const float* sem, const float* gt_mask, int64_t N,
                        int32_t n_nodes, const hold_loss_nodes* nodes, float* sums /* [10], overwritten */,
                        float* workspace, hold_stream_t stream);
int hold_pixel_loss_bwd(const float* rgb, const float* gt_rgb, const float* sem, const float* gt_mask, int64_t N,
                        int32_t n_nodes, const hold_loss_nodes* nodes, const float* g, float* d_rgb, float* d_sem,
                        hold_stream_t stream);
int hold_sumsq(const float* x, int64_t n, float* out /* [1] */, int32_t accumulate, float* workspace,
               hold_stream_t stream);
int hold_adam_step(float* p, const float* g, float* m, float* v, int64_t n, int64_t n_low, float lr_low, float lr,
                   float beta1, float beta2, float eps, int32_t step, float grad_mul, float clip_norm,
                   const float* sumsq, hold_stream_t stream);

/* ---- canonical meshing: marching tetrahedra on a dense n^3 SDF grid (x-major), replacing MISE +
 * skimage marching cubes of code/src/utils/meshing.py:9-72 / code/src/libmise/mise.pyx.  Three passes around two host
 * scans: classify (flags of the 7 lattice edges owned by each grid point, triangle count per cube), vertices (one
 * per flagged edge, id = exclusive scan of the flags), triangles (vertex ids through the edge scan; winding from the
 * host-generated case tables tet_corner [6][4], ntri_tab [6][16], tri_tab [6][16][2][3][2], hold_amd/meshing.py). */
int hold_mt_classify(const float* sdf, int32_t n, float level, const int8_t* ntri_tab, const int8_t* tet_corner,
                     int32_t* edge_flag /* [n^3][7] */, int32_t* cube_ntri /* [n^3] */, hold_stream_t stream);
int hold_mt_vertices(const float* sdf, int32_t n, float level, float ox, float oy, float oz, float h,
                     const int32_t* edge_flag, const int64_t* edge_scan, float* verts /* [V][3] */, hold_stream_t stream);
int hold_mt_triangles(const float* sdf, int32_t n, float level, const int8_t* ntri_tab, const int8_t* tet_corner,
                      const int8_t* tri_tab, const int32_t* cube_ntri, const int64_t* cube_scan,
                      const int64_t* edge_scan, int64_t* faces /* [F][3] */, hold_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Pose-refinement inner loop (hold_amd/csrc/silhouette.hip).
 * Soft silhouette = pytorch3d MeshRenderer(MeshRasterizer(blur_radius, faces_per_pixel=100), SoftSilhouetteShader)
 * as configured by code/src/fitting/utils.py:101-158 and called at code/src/fitting/model.py:136-138:
 *   mask[b][i][j] = 1 - prod_f sigmoid(d_f / sigma) over faces with signed squared NDC distance d_f < blur_radius.
 * v3d_c [B][V][3] camera-space vertices (OpenCV axes), faces [F][3] int32, pin-hole fx, fy, cx, cy in pixels.
 * workspace: hold_silhouette_workspace_floats(B, F) floats.  Backward returns d L / d v3d_c ([B][V][3], overwritten);
 * d_ndc_scratch needs B*V*2 floats.
 * K = 1 nearest neighbour = pytorch3d.ops.knn_points(K=1) of the contact terms (code/src/fitting/loss.py:90,131-136):
 *   d2 [B][Nq] squared distances, idx [B][Nq]; backward: dq overwritten, dt_accum += .
 * ---------------------------------------------------------------------------------------- */
int64_t hold_silhouette_workspace_floats(int32_t B, int32_t F);
int hold_silhouette_fwd(const float* v3d_c, int32_t B, int32_t V, const int32_t* faces, int32_t F, float fx, float fy,
                        float cx, float cy, int32_t H, int32_t W, float sigma, float blur_radius, float* workspace,
                        float* mask, hold_stream_t stream);
int hold_silhouette_bwd(const float* v3d_c, int32_t B, int32_t V, const int32_t* faces, int32_t F, float fx, float fy,
                        float cx, float cy, int32_t H, int32_t W, float sigma, float blur_radius, float* workspace,
                        const float* d_mask, float* d_ndc_scratch, float* d_v3d_c, hold_stream_t stream);
/* the most faces any pixel sees (inside or within the blur radius): pytorch3d keeps only the faces_per_pixel = 100
 * nearest of them (fitting/utils.py:107); while this stays <= 100 the cap is inactive and hold_silhouette_fwd, which
 * multiplies over all faces, equals the capped rasteriser.  max_count: one int32 on the device. */
int hold_silhouette_max_faces(const float* v3d_c, int32_t B, int32_t V, const int32_t* faces, int32_t F, float fx, float fy,
                              float cx, float cy, int32_t H, int32_t W, float blur_radius, float* workspace,
                              int32_t* max_count, hold_stream_t stream);
int hold_knn1_fwd(const float* q, int32_t B, int32_t Nq, const float* t, int32_t Nt, float* d2, int32_t* idx,
                  hold_stream_t stream);
int hold_knn1_bwd(const float* q, int32_t B, int32_t Nq, const float* t, int32_t Nt, const int32_t* idx, const float* g,
                  float* dq, float* dt_accum, hold_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* HOLD_HIP_H */
