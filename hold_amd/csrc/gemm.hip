// fp32 matrix-core GEMMs for the HOLD MLPs (gfx950 / CDNA4 only).
//
//   hold_gemm_nt : C[P][N] = epi(alpha * A[P][K] . W[N][K]^T + bias)   (layer forward, input-gradient,
//                  backward-data and double-backward sweeps; the caller passes W or W^T)
//   hold_wgrad   : dW[N][K] = R[P][N]^T . X[P][K]                        (weight gradients, reduction over points)
//
// Both run on v_mfma_f32_32x32x2_f32 (exact fp32, 157 TFLOP/s peak).  Roofline: MFMA-bound;
// algorithmic work 2*P*N*K flop per launch, HBM traffic (P*(K+N) + N*K)*4 bytes.
//
// Tiling (hold_gemm_nt): 256 threads = 4 waves as 2(points) x 2(outputs); block tile 128 points x 128
// outputs, wave tile 64x64 = 2x2 MFMA tiles (64 accumulator VGPRs), K stepped by 32 through a
// double-buffered LDS stage (2 x (128+128) x 36 floats = 72 KiB -> two blocks per CU, 2 waves/SIMD).
// The k index is permuted inside a 32-chunk so that each lane fetches 16 contiguous floats with four
// ds_read_b128 (lane half h supplies k = 16h + s to MFMA step s); A and W use the same permutation,
// so products pair up and only the fp32 summation order differs from a sequential dot product.
// LDS row stride 36 floats makes the b128 fragment reads and the b128 staging writes conflict-free.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hold_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BM = 128, BN = 128, BK = 32, LSTR = 36;

__device__ __forceinline__ float softplus100(float y) {
  float z = y * 100.0f;
  return z > 20.0f ? y : log1pf(expf(z)) * 0.01f;
}
// softplus'(x) recovered from h = softplus(x): sigmoid(100x) = 1 - exp(-100h)
__device__ __forceinline__ float dsp_from_h(float h) { return -expm1f(-100.0f * h); }

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(hold_gemm_desc d) {
  __shared__ __attribute__((aligned(16))) float smem[2 * (BM + BN) * LSTR];
  float* sA = smem;
  float* sW = smem + 2 * BM * LSTR;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int hh = lane >> 5, li = lane & 31;
  const int ntn = (d.N + BN - 1) / BN;
  const int mt_blk = blockIdx.x / ntn, nt_blk = blockIdx.x % ntn;
  const long m0 = (long)mt_blk * BM;
  const int n0 = nt_blk * BN;

  // staging assignment: 4 float4 per thread per operand
  const int srow = tid >> 3;        // 0..31 (+32*j)
  const int scol = (tid & 7) * 4;   // 0..28
  f32x4 ra[4], rw[4];

  auto load_tiles = [&](int kt) {
    const int k = kt * BK + scol;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long p = m0 + srow + 32 * j;
      const int n = n0 + srow + 32 * j;
      f32x4 za = {0.f, 0.f, 0.f, 0.f}, zw = {0.f, 0.f, 0.f, 0.f};
      if (p < d.P && k < d.K) za = *reinterpret_cast<const f32x4*>(d.A + p * (long)d.lda + k);
      if (n < d.N && k < d.K) zw = *reinterpret_cast<const f32x4*>(d.W + (long)n * d.ldw + k);
      ra[j] = za;
      rw[j] = zw;
    }
  };
  auto store_tiles = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      *reinterpret_cast<f32x4*>(sA + buf * BM * LSTR + (srow + 32 * j) * LSTR + scol) = ra[j];
      *reinterpret_cast<f32x4*>(sW + buf * BN * LSTR + (srow + 32 * j) * LSTR + scol) = rw[j];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int nk = (d.K + BK - 1) / BK;
  load_tiles(0);
  store_tiles(0);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load_tiles(kt + 1);
    const float* pa = sA + buf * BM * LSTR + (wm * 64 + li) * LSTR + hh * 16;
    const float* pw = sW + buf * BN * LSTR + (wn * 64 + li) * LSTR + hh * 16;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 a0 = *reinterpret_cast<const f32x4*>(pa + 4 * q);
      f32x4 a1 = *reinterpret_cast<const f32x4*>(pa + 32 * LSTR + 4 * q);
      f32x4 b0 = *reinterpret_cast<const f32x4*>(pw + 4 * q);
      f32x4 b1 = *reinterpret_cast<const f32x4*>(pw + 32 * LSTR + 4 * q);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[c], b0[c], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[c], b1[c], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[c], b0[c], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[c], b1[c], acc[1][1], 0, 0, 0);
      }
    }
    if (kt + 1 < nk) store_tiles(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: lane owns output column n (fixed), 16 points per MFMA tile ----
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int n = n0 + wn * 64 + b * 32 + li;
    if (n >= d.N) continue;
    const float bias = d.bias ? d.bias[n] : 0.f;
    const bool raw = n >= d.n_split;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long p = m0 + wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        if (p >= d.P) continue;
        float y = acc[a][b][r] * d.alpha + bias;
        if (raw) {
          float* o = d.C2 + p * (long)d.ldc2 + (n - d.n_split);
          *o = d.accumulate ? *o + y : y;
          continue;
        }
        float* o = d.C + p * (long)d.ldc + n;
        if (EPI == HOLD_EPI_NONE) {
          *o = d.accumulate ? *o + y : y;
        } else if (EPI == HOLD_EPI_SOFTPLUS) {
          *o = softplus100(y);
        } else if (EPI == HOLD_EPI_RELU) {
          *o = fmaxf(y, 0.f);
        } else if (EPI == HOLD_EPI_SIGMOID) {
          *o = 1.0f / (1.0f + expf(-y));
        } else if (EPI == HOLD_EPI_MUL_DSP) {
          const float h = d.aux1[p * (long)d.ldaux1 + n];
          float v = y * dsp_from_h(h);
          if (d.aux2) v += d.aux2[p * (long)d.ldaux2 + n];
          *o = v;
        } else if (EPI == HOLD_EPI_MUL_DRELU) {
          const float h = d.aux1[p * (long)d.ldaux1 + n];
          *o = h > 0.f ? y : 0.f;
        } else if (EPI == HOLD_EPI_DBWD) {
          const float h = d.aux1[p * (long)d.ldaux1 + n];
          const float t = d.aux2[p * (long)d.ldaux2 + n];
          const float e = expf(-100.0f * h);  // 1 - s
          const float s = -expm1f(-100.0f * h);
          *o = y * s;
          d.out2[p * (long)d.ldout2 + n] = 100.0f * y * t * e;
        } else if (EPI == HOLD_EPI_MUL_DSIG) {
          const float sg = d.aux1[p * (long)d.ldaux1 + n];
          *o = y * sg * (1.0f - sg);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// wgrad: partial[split][n][k] = sum_{p in split} R[p][n] X[p][k]; fragments come straight from
// global memory (both operands are contiguous along the non-reduced index, so each half-wave reads
// one 128-byte row segment per load).  Block tile 128(n) x 128(k), wave tile 64 x 64.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const float* __restrict__ R, int ldr,
                                                       const float* __restrict__ X, int ldx, int P, int N, int K,
                                                       int splits, float* __restrict__ part,
                                                       float* __restrict__ part_b) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wk = wave & 1;
  const int hh = lane >> 5, li = lane & 31;
  const int ntn = (N + 127) / 128, ntk = (K + 127) / 128;
  const int tile = blockIdx.x % (ntn * ntk), split = blockIdx.x / (ntn * ntk);
  const int n0 = (tile / ntk) * 128 + wn * 64, k0 = (tile % ntk) * 128 + wk * 64;
  const long chunks = ((long)P + 31) / 32;
  const long cper = (chunks + splits - 1) / splits;
  const long c_begin = (long)split * cper, c_end = min(chunks, c_begin + cper);

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  float bsum[2] = {0.f, 0.f};

  const int na = n0 + li, nb = n0 + 32 + li, ka = k0 + li, kb = k0 + 32 + li;
  const bool va = na < N, vb = nb < N, vka = ka < K, vkb = kb < K;
  const bool do_bias = (part_b != nullptr) && (tile % ntk == 0) && (wk == 0);

  for (long c = c_begin; c < c_end; ++c) {
    const long pb = c * 32 + hh * 16;
    float fa0[16], fa1[16], fb0[16], fb1[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const long p = pb + s;
      const bool vp = p < P;
      fa0[s] = (vp && va) ? R[p * (long)ldr + na] : 0.f;
      fa1[s] = (vp && vb) ? R[p * (long)ldr + nb] : 0.f;
      fb0[s] = (vp && vka) ? X[p * (long)ldx + ka] : 0.f;
      fb1[s] = (vp && vkb) ? X[p * (long)ldx + kb] : 0.f;
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[s], fb0[s], acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[s], fb1[s], acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[s], fb0[s], acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[s], fb1[s], acc[1][1], 0, 0, 0);
      bsum[0] += fa0[s];
      bsum[1] += fa1[s];
    }
  }
  // D[i][j]: i = n (A operand rows), j = k.  lane holds column j = li, rows (r&3)+8(r>>2)+4hh.
  float* out = part + (long)split * N * K;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int k = k0 + b * 32 + li;
      if (k >= K) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        if (n < N) out[(long)n * K + k] = acc[a][b][r];
      }
    }
  if (do_bias) {
    float s0 = bsum[0] + __shfl_xor(bsum[0], 32);
    float s1 = bsum[1] + __shfl_xor(bsum[1], 32);
    if (hh == 0) {
      if (va) part_b[(long)split * N + na] = s0;
      if (vb) part_b[(long)split * N + nb] = s1;
    }
  }
}

__global__ void wgrad_reduce_kernel(const float* __restrict__ part, int splits, long NK, int K, float* __restrict__ dW,
                                    int lddw, int accumulate) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= NK) return;
  float s = 0.f;
  for (int sp = 0; sp < splits; ++sp) s += part[(long)sp * NK + i];
  const long n = i / K, k = i % K;
  float* o = dW + n * lddw + k;
  *o = accumulate ? *o + s : s;
}

}  // namespace

extern "C" int hold_abi_version(void) { return 1; }

extern "C" int hold_gemm_nt(const hold_gemm_desc* dp, hold_stream_t stream) {
  if (!dp) return HOLD_E_ARG;
  hold_gemm_desc d = *dp;
  if (!d.A || !d.W || !d.C || d.P < 0 || d.N <= 0 || d.K <= 0) return HOLD_E_ARG;
  if ((d.lda & 3) || (d.ldw & 3) || (d.K & 3)) return HOLD_E_ARG;
  if (((uintptr_t)d.A & 15) || ((uintptr_t)d.W & 15)) return HOLD_E_ARG;
  if (d.n_split <= 0 || d.n_split > d.N) d.n_split = d.N;
  if (d.n_split < d.N && !d.C2) return HOLD_E_ARG;
  if (d.P == 0) return HOLD_OK;
  const long mt = ((long)d.P + BM - 1) / BM;
  const int nt = (d.N + BN - 1) / BN;
  dim3 grid((unsigned)(mt * nt)), block(256);
  hipStream_t s = (hipStream_t)stream;
  switch (d.epilogue) {
    case HOLD_EPI_NONE: hipLaunchKernelGGL(gemm_nt_kernel<HOLD_EPI_NONE>, grid, block, 0, s, d); break;
    case HOLD_EPI_SOFTPLUS: hipLaunchKernelGGL(gemm_nt_kernel<HOLD_EPI_SOFTPLUS>, grid, block, 0, s, d); break;
    case HOLD_EPI_RELU: hipLaunchKernelGGL(gemm_nt_kernel<HOLD_EPI_RELU>, grid, block, 0, s, d); break;
    case HOLD_EPI_SIGMOID: hipLaunchKernelGGL(gemm_nt_kernel<HOLD_EPI_SIGMOID>, grid, block, 0, s, d); break;
    case HOLD_EPI_MUL_DSP:
      if (!d.aux1) return HOLD_E_ARG;
      hipLaunchKernelGGL(gemm_nt_kernel<HOLD_EPI_MUL_DSP>, grid, block, 0, s, d);
      break;
    case HOLD_EPI_MUL_DRELU:
      if (!d.aux1) return HOLD_E_ARG;
      hipLaunchKernelGGL(gemm_nt_kernel<HOLD_EPI_MUL_DRELU>, grid, block, 0, s, d);
      break;
    case HOLD_EPI_DBWD:
      if (!d.aux1 || !d.aux2 || !d.out2) return HOLD_E_ARG;
      hipLaunchKernelGGL(gemm_nt_kernel<HOLD_EPI_DBWD>, grid, block, 0, s, d);
      break;
    case HOLD_EPI_MUL_DSIG:
      if (!d.aux1) return HOLD_E_ARG;
      hipLaunchKernelGGL(gemm_nt_kernel<HOLD_EPI_MUL_DSIG>, grid, block, 0, s, d);
      break;
    default: return HOLD_E_ARG;
  }
  return hipGetLastError() == hipSuccess ? HOLD_OK : HOLD_E_LAUNCH;
}

extern "C" int64_t hold_wgrad_workspace_floats(int32_t N, int32_t K, int32_t splits) {
  return (int64_t)splits * ((int64_t)N * K + N);
}

extern "C" int hold_wgrad(const float* R, int32_t ldr, const float* X, int32_t ldx, int32_t P, int32_t N, int32_t K,
                          float* dW, int32_t lddw, float* db, int32_t accumulate, int32_t splits, float* workspace,
                          hold_stream_t stream) {
  if (!R || !X || !dW || !workspace || N <= 0 || K <= 0 || P < 0 || splits <= 0) return HOLD_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  const long chunks = ((long)P + 31) / 32;
  if (splits > chunks) splits = (int)(chunks > 0 ? chunks : 1);
  float* part = workspace;
  float* part_b = db ? workspace + (long)splits * N * K : nullptr;
  const int tiles = ((N + 127) / 128) * ((K + 127) / 128);
  hipLaunchKernelGGL(wgrad_kernel, dim3(tiles * splits), dim3(256), 0, s, R, ldr, X, ldx, P, N, K, splits, part,
                     part_b);
  const long NK = (long)N * K;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((NK + 255) / 256)), dim3(256), 0, s, part, splits, NK, K, dW,
                     lddw, accumulate);
  if (db)
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, part_b, splits,
                       (long)N, N, db, N, accumulate);
  return hipGetLastError() == hipSuccess ? HOLD_OK : HOLD_E_LAUNCH;
}
