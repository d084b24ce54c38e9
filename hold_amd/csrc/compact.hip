// Exact sample compaction (north_star: "wavefront ballots for early termination and sample compaction"; reference semantics to
// preserve: code/src/engine/volsdf_utils.py:220-251 density2weight, code/src/engine/density.py:21-26).
//
// The reference integrates every sample.  A sample whose SDF lies so far outside the surface that the Laplace density
// 0.5 exp(-sdf / beta) / beta AND the exponential exp(-sdf / beta) of its derivatives are EXACT fp32 zeros (sdf / beta > 104:
// with a trained beta of ~0.005 that is half a scene unit off the surface -- every sample of a ray that misses the object, the
// far tail of the others) has compositing weight 0, so its colour and normal never reach a pixel and receive a zero gradient,
// and d loss / d sdf = (d loss / d density) x 0 and its term of d loss / d beta are zero as well: nothing any stage behind the
// SDF computes for it is ever used.  These kernels build the ordered list of the OTHER samples -- the predicate is evaluated with
// the compositor's own expressions (laplace.h), a wave ballot gives every lane its rank among the live samples of its wave
// (popcount of the ballot below the lane), the waves' totals are scanned in LDS, the blocks' totals by the caller -- and the
// host runs the reverse sweep, the colour net and the whole backward on the compacted rows only (hold_amd/field.py),
// scattering zeros for the rest: outputs bit-identical to the uncompacted path, parameter gradients equal up to the order of
// the (fewer) terms of their sums.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hold_hip.h"
#include "laplace.h"

namespace {

constexpr int CB = 1024;  // samples per block: 4 rounds of 256

__device__ __forceinline__ bool alive(const float* sdf, int ld, long p, long P, float beta) {
  if (p >= P) return false;
  const float s = sdf[p * ld];
  return !(hold_laplace_density(s, beta) == 0.f && hold_laplace_exp(s, beta) == 0.f);
}

__global__ __launch_bounds__(256) void alive_count_kernel(const float* __restrict__ sdf, int ld, long P, float beta,
                                                          int* __restrict__ counts) {
  __shared__ int wsum[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int c = 0;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const long p = (long)blockIdx.x * CB + it * 256 + threadIdx.x;
    c += __popcll(__ballot(alive(sdf, ld, p, P, beta)));
  }
  if (lane == 0) wsum[wave] = c;
  __syncthreads();
  if (threadIdx.x == 0) counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// idx[offsets[block] + rank of p among the block's live samples] = p, ascending
__global__ __launch_bounds__(256) void alive_index_kernel(const float* __restrict__ sdf, int ld, long P, float beta,
                                                          const long* __restrict__ offsets, long* __restrict__ idx) {
  __shared__ int cnt[16];  // [round][wave]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  bool live[4];
  int rank[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const long p = (long)blockIdx.x * CB + it * 256 + threadIdx.x;
    live[it] = alive(sdf, ld, p, P, beta);
    const unsigned long long b = __ballot(live[it]);
    rank[it] = __popcll(b & ((1ull << lane) - 1ull));  // live lanes below this one
    if (lane == 0) cnt[it * 4 + wave] = __popcll(b);
  }
  __syncthreads();
  const long base = offsets[blockIdx.x];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    int before = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q)
      if (q < it * 4 + wave) before += cnt[q];
    const long p = (long)blockIdx.x * CB + it * 256 + threadIdx.x;
    if (live[it]) idx[base + before + rank[it]] = p;
  }
}

// mask[p] = 1 for a live sample, 0 for a dead one (the batch-of-frames compaction ranks the samples frame by frame on the host side)
__global__ __launch_bounds__(256) void alive_mask_kernel(const float* __restrict__ sdf, int ld, long P, float beta,
                                                         uint8_t* __restrict__ mask) {
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  if (p < P) mask[p] = alive(sdf, ld, p, P, beta) ? 1 : 0;
}

inline int ok() { return hipGetLastError() == hipSuccess ? HOLD_OK : HOLD_E_LAUNCH; }

}  // namespace

extern "C" int64_t hold_alive_blocks(int64_t P) { return (P + CB - 1) / CB; }

extern "C" int hold_alive_count(const float* sdf, int32_t ld, int64_t P, float beta, int32_t* block_counts, hold_stream_t st) {
  if (!sdf || !block_counts || ld < 1 || P < 0 || !(beta > 0.f)) return HOLD_E_ARG;
  if (P == 0) return HOLD_OK;
  hipLaunchKernelGGL(alive_count_kernel, dim3((unsigned)hold_alive_blocks(P)), dim3(256), 0, (hipStream_t)st, sdf, ld, (long)P,
                     beta, block_counts);
  return ok();
}

extern "C" int hold_alive_index(const float* sdf, int32_t ld, int64_t P, float beta, const int64_t* block_offsets, int64_t* idx,
                                hold_stream_t st) {
  if (!sdf || !block_offsets || !idx || ld < 1 || P < 0 || !(beta > 0.f)) return HOLD_E_ARG;
  if (P == 0) return HOLD_OK;
  hipLaunchKernelGGL(alive_index_kernel, dim3((unsigned)hold_alive_blocks(P)), dim3(256), 0, (hipStream_t)st, sdf, ld, (long)P,
                     beta, reinterpret_cast<const long*>(block_offsets), reinterpret_cast<long*>(idx));
  return ok();
}

extern "C" int hold_alive_mask(const float* sdf, int32_t ld, int64_t P, float beta, uint8_t* mask, hold_stream_t st) {
  if (!sdf || !mask || ld < 1 || P < 0 || !(beta > 0.f)) return HOLD_E_ARG;
  if (P == 0) return HOLD_OK;
  hipLaunchKernelGGL(alive_mask_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, (hipStream_t)st, sdf, ld, (long)P, beta, mask);
  return ok();
}
