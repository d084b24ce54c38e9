// Diagnostic micro-kernel: issue-rate ceiling of v_mfma_f32_32x32x2_f32 on this box (no memory traffic),
// used by scripts/bench_gemm.py to calibrate the GEMM roofline fraction against the DVFS-limited clock.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hold_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {
__global__ __launch_bounds__(256, 2) void mfma_peak_kernel(float* out, int iters, float seed) {
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = seed + threadIdx.x * 1e-3f, b = seed * 0.5f - threadIdx.x * 1e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, b, acc[3], 0, 0, 0);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
}  // namespace

extern "C" int hold_diag_mfma_peak(float* out, int32_t blocks, int32_t iters, hold_stream_t st) {
  if (!out || blocks <= 0 || iters <= 0) return HOLD_E_ARG;
  hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)st, out, iters, 0.37f);
  return hipGetLastError() == hipSuccess ? HOLD_OK : HOLD_E_LAUNCH;
}
