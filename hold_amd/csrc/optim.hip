// Step tail of the training loop: the per-pixel loss terms of code/src/hold/loss.py:17-93 /
// loss_terms.py:14-111 as one forward and one backward launch, and the optimiser step of code/src/hold/hold.py:79-101
// (Adam, eps 1e-8, pose tables at 0.1 x lr) with Lightning's gradient_clip_val = 0.5 (code/train.py:30) as two
// launches over ONE flat fp32 bucket -- the same bucket the data-parallel all-reduce runs on (hold_amd/parallel.py).
// Everything here is HBM-bound streaming: 2.2 M parameters = 8.8 MB per array, read/written once per step.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/hold_hip.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// block of 256 threads = 4 waves: reduce K values per thread into this block's row of the partials workspace; a second
// single-block launch adds the rows in a fixed order (deterministic sums, no float atomics)
template <int K>
__device__ __forceinline__ void block_accumulate(float (&v)[K], float* partials) {
  __shared__ float part[4][K];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const float s = wave_sum(v[k]);
    if (lane == 0) part[w][k] = s;
  }
  __syncthreads();
  if (threadIdx.x < K)
    partials[(long)blockIdx.x * K + threadIdx.x] =
        part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
}

// out[k] (+)= sum over rows of partials[rows][K] in a fixed tree order
template <int K>
__global__ __launch_bounds__(256) void reduce_partials(const float* __restrict__ partials, int rows, float* __restrict__ out,
                                                       int accumulate) {
  __shared__ float part[4][K];
  float v[K];
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = 0.f;
  for (int r = threadIdx.x; r < rows; r += 256)
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] += partials[(long)r * K + k];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const float s = wave_sum(v[k]);
    if (lane == 0) part[w][k] = s;
  }
  __syncthreads();
  if (threadIdx.x < K) {
    const float s = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
    out[threadIdx.x] = accumulate ? out[threadIdx.x] + s : s;
  }
}

__device__ __forceinline__ int sem_class(float m) {  // loss_terms.get_sem_loss: <25 bg, <100 object, <200 right, else left
  return m < 25.f ? 0 : (m < 100.f ? 1 : (m < 200.f ? 2 : 3));
}

// sums[0] = sum |rgb - gt| over rows without NaN, sums[1] = sum (sem - onehot)^2, sums[2] = rows without NaN,
// sums[3 + 2 i] = sum of node i's mask_prob over its off-surface rays, sums[4 + 2 i] = their count   (10 slots)
__global__ __launch_bounds__(256) void pixel_loss_fwd(const float* __restrict__ rgb, const float* __restrict__ gt_rgb,
                                                      const float* __restrict__ sem, const float* __restrict__ gt_mask,
                                                      long N, int n_nodes, hold_loss_nodes nd, float* __restrict__ partials) {
  const float* const* mask_prob = nd.mask_prob;
  const uint8_t* const* off = nd.off;
  float v[10] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < N; r += (long)gridDim.x * 256) {
    const float a = rgb[r * 3], b = rgb[r * 3 + 1], c = rgb[r * 3 + 2];
    if (!(a != a || b != b || c != c)) {
      v[0] += fabsf(a - gt_rgb[r * 3]) + fabsf(b - gt_rgb[r * 3 + 1]) + fabsf(c - gt_rgb[r * 3 + 2]);
      v[2] += 1.f;
    }
    const int cls = sem_class(gt_mask[r]);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float d = sem[r * 4 + k] - (k == cls ? 1.f : 0.f);
      v[1] += d * d;
    }
    for (int i = 0; i < n_nodes; ++i) {
      if (off[i] && off[i][r]) {
        v[3 + 2 * i] += fabsf(mask_prob[i][r]);
        v[4 + 2 * i] += 1.f;
      }
    }
  }
  block_accumulate<10>(v, partials);
}

// g[0] = dL/d sums[0], g[1] = dL/d sums[1], g[3 + 2 i] = dL/d sums[3 + 2 i]
__global__ __launch_bounds__(256) void pixel_loss_bwd(const float* __restrict__ rgb, const float* __restrict__ gt_rgb,
                                                      const float* __restrict__ sem, const float* __restrict__ gt_mask,
                                                      long N, int n_nodes, hold_loss_nodes nd, const float* __restrict__ g,
                                                      float* __restrict__ d_rgb, float* __restrict__ d_sem) {
  const float* const* mask_prob = nd.mask_prob;
  const uint8_t* const* off = nd.off;
  float* const* d_mask = nd.d_mask;
  const long r = (long)blockIdx.x * 256 + threadIdx.x;
  if (r >= N) return;
  const float g0 = g[0], g1 = g[1];
  const float a = rgb[r * 3], b = rgb[r * 3 + 1], c = rgb[r * 3 + 2];
  const bool ok = !(a != a || b != b || c != c);
  const float x[3] = {a, b, c};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float d = x[k] - gt_rgb[r * 3 + k];
    d_rgb[r * 3 + k] = ok ? (d > 0.f ? g0 : (d < 0.f ? -g0 : 0.f)) : 0.f;
  }
  const int cls = sem_class(gt_mask[r]);
#pragma unroll
  for (int k = 0; k < 4; ++k) d_sem[r * 4 + k] = 2.f * g1 * (sem[r * 4 + k] - (k == cls ? 1.f : 0.f));
  for (int i = 0; i < n_nodes; ++i) {
    if (!d_mask[i]) continue;
    float o = 0.f;
    if (off[i] && off[i][r]) {
      const float m = mask_prob[i][r];
      o = m > 0.f ? g[3 + 2 * i] : (m < 0.f ? -g[3 + 2 * i] : 0.f);
    }
    d_mask[i][r] = o;
  }
}

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, long n, float* __restrict__ partials) {
  float v[1] = {0.f};
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) v[0] += x[i] * x[i];
  block_accumulate<1>(v, partials);
}

// p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps) with g scaled by min(1, clip / (||g|| + 1e-6)) (torch clip_grad_norm_)
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long n, long n_low, float lr_low, float lr,
                                                   float b1, float b2, float eps, float bc1, float bc2, float grad_mul,
                                                   float clip, const float* __restrict__ sumsq) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float scale = grad_mul;
  if (clip > 0.f && sumsq) {
    const float norm = sqrtf(sumsq[0]) * fabsf(grad_mul);
    scale *= fminf(1.f, clip / (norm + 1e-6f));
  }
  const float gi = g[i] * scale;
  const float mi = b1 * m[i] + (1.f - b1) * gi;
  const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
  m[i] = mi;
  v[i] = vi;
  const float step = (i < n_low ? lr_low : lr) / bc1;
  p[i] -= step * mi / (sqrtf(vi) / sqrtf(bc2) + eps);
}

inline int ok() { return hipGetLastError() == hipSuccess ? HOLD_OK : HOLD_E_LAUNCH; }
inline unsigned grid_for(long n) {
  long b = (n + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

}  // namespace

extern "C" int hold_pixel_loss_fwd(const float* rgb, const float* gt_rgb, const float* sem, const float* gt_mask, int64_t N,
                                   int32_t n_nodes, const hold_loss_nodes* nodes, float* sums /* [10] */,
                                   float* workspace /* hold_reduce_workspace_floats() */, hold_stream_t st) {
  if (!rgb || !gt_rgb || !sem || !gt_mask || !sums || !workspace || n_nodes < 0 || n_nodes > 3 || (n_nodes && !nodes))
    return HOLD_E_ARG;
  hold_loss_nodes nd = {};
  if (nodes) nd = *nodes;
  const unsigned grid = grid_for(N);
  hipLaunchKernelGGL(pixel_loss_fwd, dim3(grid), dim3(256), 0, (hipStream_t)st, rgb, gt_rgb, sem, gt_mask, (long)N,
                     n_nodes, nd, workspace);
  hipLaunchKernelGGL(reduce_partials<10>, dim3(1), dim3(256), 0, (hipStream_t)st, workspace, (int)grid, sums, 0);
  return ok();
}

extern "C" int hold_pixel_loss_bwd(const float* rgb, const float* gt_rgb, const float* sem, const float* gt_mask, int64_t N,
                                   int32_t n_nodes, const hold_loss_nodes* nodes, const float* g, float* d_rgb,
                                   float* d_sem, hold_stream_t st) {
  if (!rgb || !gt_rgb || !sem || !gt_mask || !g || !d_rgb || !d_sem || n_nodes < 0 || n_nodes > 3 || (n_nodes && !nodes))
    return HOLD_E_ARG;
  if (N == 0) return HOLD_OK;
  hold_loss_nodes nd = {};
  if (nodes) nd = *nodes;
  hipLaunchKernelGGL(pixel_loss_bwd, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)st, rgb, gt_rgb, sem,
                     gt_mask, (long)N, n_nodes, nd, g, d_rgb, d_sem);
  return ok();
}

extern "C" int64_t hold_reduce_workspace_floats(void) { return 2048 * 10; }

extern "C" int hold_sumsq(const float* x, int64_t n, float* out, int32_t accumulate, float* workspace, hold_stream_t st) {
  if (!x || !out || !workspace || n < 0) return HOLD_E_ARG;
  const unsigned grid = grid_for(n);
  hipLaunchKernelGGL(sumsq_kernel, dim3(grid), dim3(256), 0, (hipStream_t)st, x, (long)n, workspace);
  hipLaunchKernelGGL(reduce_partials<1>, dim3(1), dim3(256), 0, (hipStream_t)st, workspace, (int)grid, out, accumulate);
  return ok();
}

extern "C" int hold_adam_step(float* p, const float* g, float* m, float* v, int64_t n, int64_t n_low, float lr_low, float lr,
                              float beta1, float beta2, float eps, int32_t step, float grad_mul, float clip_norm,
                              const float* sumsq, hold_stream_t st) {
  if (!p || !g || !m || !v || n < 0 || n_low < 0 || n_low > n || step < 1) return HOLD_E_ARG;
  if (n == 0) return HOLD_OK;
  // bias corrections in double, as torch.optim.Adam computes them on the host
  const float bc1 = (float)(1.0 - pow((double)beta1, (double)step)), bc2 = (float)(1.0 - pow((double)beta2, (double)step));
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)st, p, g, m, v, (long)n,
                     (long)n_low, lr_low, lr, beta1, beta2, eps, bc1, bc2, grad_mul, clip_norm, sumsq);
  return ok();
}
