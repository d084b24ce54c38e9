// Diagnostic micro-kernel: issue-rate ceiling of v_mfma_f32_32x32x2_f32 on this box (no memory traffic),
// used by scripts/bench_gemm.py to calibrate the GEMM roofline fraction against the DVFS-limited clock.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hold_hip_dev.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {
__global__ __launch_bounds__(256, 2) void mfma_peak_kernel(float* out, int iters, float seed) {
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  if (seed >= 0.f) {  // constant operands: lowest switching power, highest sustained clock
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
  } else {  // 32 different pseudo-random operands per lane, like a real GEMM's operand stream
    float av[16], bv[16];
    unsigned x = 0x9E3779B9u * (threadIdx.x + 1 + blockIdx.x * 256);
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      x ^= x << 13; x ^= x >> 17; x ^= x << 5;
      av[u] = (float)(int)(x & 0xffffff) * (1.0f / 8388608.0f) - 1.0f;
      x ^= x << 13; x ^= x >> 17; x ^= x << 5;
      bv[u] = ((float)(int)(x & 0xffffff) * (1.0f / 8388608.0f) - 1.0f) * 0.05f;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[u], av[(u + 5) & 15], acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[(u + 3) & 15], bv[(u + 7) & 15], acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[(u + 9) & 15], av[(u + 11) & 15], acc[3], 0, 0, 0);
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
// mode 2: A operands from LDS (4 x ds_read_b128 per 16 MFMAs), B from registers
// mode 3: mode 2 + B operand from a global (L2-resident) float4 load per 16 MFMAs
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512, 2) void mfma_lds_kernel(float* out, const float* __restrict__ wsrc, int iters,
                                                         int mode) {
  __shared__ __attribute__((aligned(16))) float act[128 * 260];
  for (int i = threadIdx.x; i < 128 * 260; i += 512) act[i] = (float)((i * 2654435761u) >> 9) * (1.0f / 8388608.0f) - 1.0f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hh = lane >> 5, li = lane & 31;
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const float* arow = act + li * 260 + hh * 4;
  const f32x4* wp = reinterpret_cast<const f32x4*>(wsrc) + wave * 64 + lane;
  f32x4 b = wp[0];
  for (int it = 0; it < iters; ++it) {
    const int kc = it & 31;
    f32x4 bn = b;
    if (mode == 3) bn = wp[(long)((it + 1) & 31) * 512];
    f32x4 av[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) av[m] = *reinterpret_cast<const f32x4*>(arow + m * 32 * 260 + kc * 8);
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m][c], b[c], acc[m], 0, 0, 0);
    b = bn;
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}
}  // namespace

extern "C" int hold_diag_mfma_lds(float* out, const float* wsrc, int32_t blocks, int32_t iters, int32_t mode,
                                  hold_stream_t st) {
  if (!out || !wsrc || blocks <= 0 || iters <= 0) return HOLD_E_ARG;
  hipLaunchKernelGGL(mfma_lds_kernel, dim3(blocks), dim3(512), 0, (hipStream_t)st, out, wsrc, iters, mode);
  return hipGetLastError() == hipSuccess ? HOLD_OK : HOLD_E_LAUNCH;
}

extern "C" int hold_diag_mfma_peak(float* out, int32_t blocks, int32_t iters, int32_t random_operands, hold_stream_t st) {
  if (!out || blocks <= 0 || iters <= 0) return HOLD_E_ARG;
  hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)st, out, iters, random_operands ? -1.0f : 0.37f);
  return hipGetLastError() == hipSuccess ? HOLD_OK : HOLD_E_LAUNCH;
}
