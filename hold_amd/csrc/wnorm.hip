// Weight normalisation of ALL layers of a net in one launch per direction (gfx950).
//
// Every Linear of ImplicitNet / RenderingNet is wrapped in torch.nn.utils.weight_norm (code/src/networks/shape_net.py:79-80,
// texture_net.py:40-41): the matrix the kernels consume is w = v * (g / ||v||_row), and its gradient has to come back as
// (dv, dg).  Per layer and direction that is ~10 tiny launches in torch; the reference's own training batch (1 280 rays)
// re-derives the matrices of 6 nets every step and spent ~220 launches there.  One wave per matrix row, rows of all layers
// of a call in one grid; HBM-streaming (12 bytes per element and direction), a few microseconds per call.
//
//   hold_weight_norm_fwd : w[r][:] = v[r][:] * (g[r] / ||v[r]||)
//   hold_weight_norm_bwd : t = <dw[r], v[r]>, n = ||v[r]||:  dg[r] (+)= t / n,  dv[r][:] (+)= dw[r][:] g[r]/n - v[r][:] t g[r]/n^3
//                          (accumulate != 0: add into dv / dg -- the gradient buckets of the optimiser -- instead of storing;
//                           a layer with dw == NULL contributes nothing)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hold_hip.h"

namespace {

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) x += __shfl_xor(x, d);
  return x;
}

// the layer that owns global row `row`, and the row inside it (n_layers <= HOLD_WN_MAX_LAYERS: a linear scan)
__device__ __forceinline__ int find_layer(const hold_wn_desc& d, long& row) {
  int l = 0;
  while (l + 1 < d.n_layers && row >= d.layers[l].rows) {
    row -= d.layers[l].rows;
    ++l;
  }
  return l;
}

template <bool BWD>
__global__ __launch_bounds__(256) void wnorm_kernel(hold_wn_desc d, long total_rows) {
  const int lane = threadIdx.x & 63;
  long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= total_rows) return;
  const hold_wn_layer& L = d.layers[find_layer(d, row)];
  const int K = L.cols;
  const float* v = L.v + row * (long)L.ldv;
  const float g = L.g[row];
  float ss = 0.f, t = 0.f;
  const float* dw = BWD && L.dw ? L.dw + row * (long)L.ldw : nullptr;
  if (BWD && !dw) return;
  for (int k = lane; k < K; k += 64) {
    const float x = v[k];
    ss = fmaf(x, x, ss);
    if (BWD) t = fmaf(dw[k], x, t);
  }
  ss = wave_sum(ss);
  const float n = sqrtf(ss);
  const float s = g / n;
  if (!BWD) {
    float* w = L.w + row * (long)L.ldw;
    for (int k = lane; k < K; k += 64) w[k] = v[k] * s;
    return;
  }
  t = wave_sum(t);
  const float c = t * g / (n * n * n);
  float* dv = L.dv + row * (long)L.ldv;
  if (d.accumulate) {
    for (int k = lane; k < K; k += 64) dv[k] += dw[k] * s - v[k] * c;
    if (lane == 0) L.dg[row] += t / n;
  } else {
    for (int k = lane; k < K; k += 64) dv[k] = dw[k] * s - v[k] * c;
    if (lane == 0) L.dg[row] = t / n;
  }
}

int check(const hold_wn_desc* d, bool bwd, long* total) {
  if (!d || d->n_layers < 1 || d->n_layers > HOLD_WN_MAX_LAYERS) return HOLD_E_ARG;
  long rows = 0;
  for (int l = 0; l < d->n_layers; ++l) {
    const hold_wn_layer& L = d->layers[l];
    if (!L.v || !L.g || L.rows < 0 || L.cols < 1 || L.ldv < L.cols || L.ldw < L.cols) return HOLD_E_ARG;
    if (!bwd && !L.w) return HOLD_E_ARG;
    if (bwd && L.dw && (!L.dv || !L.dg)) return HOLD_E_ARG;
    rows += L.rows;
  }
  *total = rows;
  return HOLD_OK;
}

}  // namespace

extern "C" int hold_weight_norm_fwd(const hold_wn_desc* d, hold_stream_t st) {
  long rows = 0;
  if (int e = check(d, false, &rows)) return e;
  if (rows == 0) return HOLD_OK;
  hipLaunchKernelGGL((wnorm_kernel<false>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)st, *d, rows);
  return hipGetLastError() == hipSuccess ? HOLD_OK : HOLD_E_LAUNCH;
}

extern "C" int hold_weight_norm_bwd(const hold_wn_desc* d, hold_stream_t st) {
  long rows = 0;
  if (int e = check(d, true, &rows)) return e;
  if (rows == 0) return HOLD_OK;
  hipLaunchKernelGGL((wnorm_kernel<true>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)st, *d, rows);
  return hipGetLastError() == hipSuccess ? HOLD_OK : HOLD_E_LAUNCH;
}
