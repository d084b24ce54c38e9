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
} hold_gemm_desc;

int hold_gemm_nt(const hold_gemm_desc* d, hold_stream_t stream);

/* Weight gradient: dW[n][k] (+)= sum_p R[p][n] * X[p][k];  db[n] (+)= sum_p R[p][n] (db may be NULL).
 * Split over P into `splits` partial tiles in `workspace` (>= hold_wgrad_workspace_floats floats),
 * reduced deterministically by a second kernel.  Replaces the weight/bias gradient GEMMs of
 * torch autograd for every Linear on the path. */
int64_t hold_wgrad_workspace_floats(int32_t N, int32_t K, int32_t splits);
int hold_wgrad(const float* R, int32_t ldr, const float* X, int32_t ldx, int32_t P, int32_t N, int32_t K,
               float* dW, int32_t lddw, float* db, int32_t accumulate, int32_t splits, float* workspace,
               hold_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* HOLD_HIP_H */
