// Narrow layer: C[P][N] (+)= A[P][256] W[N][256]^T for N <= 64 (gfx950) -- the three GEMMs of a node's step whose output is a few
// dozen columns wide: d sdf / d embedding (t_0 W_0, N = 39, forward and backward) and the non-feature columns of the colour
// net's input gradient (N = 16 / 48).  hold_gemm_nt_x6 runs them on 128 x 128 tiles, i.e. pays the MFMAs and the fragment
// splits of 128 outputs for 39; here the work is what it is: stream A once (1 KiB per point, the HBM floor) and issue
// ceil(N / 32) tiles of MFMAs per 32 points.
//
// Split-precision arithmetic of hold_gemm_nt_x6 (both operands split exactly into three bf16 limbs by truncation, six limb
// products on v_mfma_f32_32x32x16_bf16, fp32 accumulation).
//
// Workgroup = 4 waves, ONE per SIMD is not needed (< 128 registers), but the LDS budget allows one workgroup per CU:
//   * the limbs of W (the MFMA B operand: lane = output n, 8 consecutive k) are prepared once per workgroup and stay in LDS as
//     [limb][n-tile][k-step][lane] x 16 B planes (48 KiB per 32 outputs), read lane-linearly (conflict-free);
//   * every wave works on its own 32-point tiles with NO workgroup barrier in the main loop: the raw rows of a quarter of K
//     ([32 points][64 floats] = 8 KiB) come by LDS-DMA into a wave-private two-stage ring, one stage ahead; the 16-byte chunk
//     index is XOR-swizzled by the row on the SOURCE address, so that the fragment reads (lane = point, two 16-byte chunks of
//     8 consecutive k) of 16 rows hit 16 distinct bank groups;
//   * the A fragment of a k-step is split into limbs as it leaves LDS (36 VALU per 12 MFMAs with two output tiles).
// A finished tile is stored straight from the accumulators: a store instruction covers 32 consecutive output columns of two
// points (128-byte runs); C may be any 4-byte-aligned strided view (d sdf / d embedding lives in columns 217.. of t_3).
// Roofline: HBM, 1 KiB + 4 N B per point (+ 4 N B when accumulating); the MFMAs of two output tiles take half the time of the
// stream at 5 TB/s.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hold_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int WL_NT = 3 * 16 * 1024;  // limb planes of one 32-output tile
constexpr int STAGE = 8 * 1024;       // [32 rows][64 floats]

struct NArgs {
  const float* A; long lda;
  const float* W; int ldw;
  float* C; long ldc;
  long P, n_tiles;
  int N, accumulate;
};

__device__ __forceinline__ uint32_t fbits(float x) { return __builtin_bit_cast(uint32_t, x); }
__device__ __forceinline__ float bitsf(uint32_t x) { return __builtin_bit_cast(float, x); }

// exact truncation split of 8 values into three bf16 limbs (element 2 j in the low half of word j): x = l0 + l1 + l2 + O(2^-24 x)
__device__ __forceinline__ void split8(const float (&x)[8], u32x4 (&l)[3]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float a = x[2 * j], b = x[2 * j + 1];
    l[0][j] = __builtin_amdgcn_perm(fbits(b), fbits(a), 0x07060302u);
    const float a1 = a - bitsf(fbits(a) & 0xffff0000u), b1 = b - bitsf(fbits(b) & 0xffff0000u);
    l[1][j] = __builtin_amdgcn_perm(fbits(b1), fbits(a1), 0x07060302u);
    const float a2 = a1 - bitsf(fbits(a1) & 0xffff0000u), b2 = b1 - bitsf(fbits(b1) & 0xffff0000u);
    l[2][j] = __builtin_amdgcn_perm(fbits(b2), fbits(a2), 0x07060302u);
  }
}

// 64 lanes x 16 bytes: lane l fetches src + voff(l), the bytes land lane-linear at LDS byte address dst (M0)
__device__ __forceinline__ void dma_piece(const char* src, uint32_t voff, uint32_t dst) {
  asm volatile(
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %0, %1"
      :
      : "v"(voff), "s"(src), "s"(dst)
      : "memory");
}
#define RN_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

template <int NT>
__global__ __launch_bounds__(256) void rnarrow_kernel(NArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // the only LDS of the kernel: byte address 0 = smem
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hh = lane >> 5, li = lane & 31;

  // ---- limbs of W, once: unit (nt, ks, l) = outputs n = 32 nt + (l & 31), k = 16 ks + 8 (l >> 5) .. + 7 ----
  for (int u = tid; u < NT * 16 * 64; u += 256) {
    const int l = u & 63, ks = (u >> 6) & 15, nt = u >> 10;
    const int n = 32 * nt + (l & 31), k0 = 16 * ks + 8 * (l >> 5);
    float x[8];
    if (n < a.N) {
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(a.W + (long)n * a.ldw + k0);
      const f32x4 v1 = *reinterpret_cast<const f32x4*>(a.W + (long)n * a.ldw + k0 + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { x[e] = v0[e]; x[4 + e] = v1[e]; }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = 0.f;
    }
    u32x4 lim[3];
    split8(x, lim);
#pragma unroll
    for (int j = 0; j < 3; ++j) *reinterpret_cast<u32x4*>(smem + ((j * NT + nt) * 16 + ks) * 1024 + l * 16) = lim[j];
  }
  __syncthreads();

  const long tstride = (long)gridDim.x * 4;
  long t = (long)blockIdx.x * 4 + wave;
  if (t >= a.n_tiles) return;  // (after the only barrier)
  const uint32_t stg0 = (uint32_t)(NT * WL_NT + wave * 2 * STAGE);
  const long rowb = a.lda * 4;
  // DMA piece i of a stage = rows 4 i + (lane >> 4); LDS position (row, physical chunk lane & 15) receives the logical chunk
  // (lane & 15) ^ (row & 15) of the row's 64-float quarter
  const int prow = lane >> 4, pch = lane & 15;
  auto issue = [&](long tile, int q, int buf) {
    const char* base = reinterpret_cast<const char*>(a.A) + tile * 32 * rowb + q * 256;
    const long last = a.P - 1 - tile * 32;  // rows beyond the matrix re-read the last one (never stored)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = 4 * i + prow;
      const long rr = row < last ? row : last;
      dma_piece(base, (uint32_t)(rr * rowb + ((pch ^ (row & 15)) << 4)), stg0 + buf * STAGE + i * 1024);
    }
  };

  int buf = 0;
  issue(t, 0, 0);
  for (; t < a.n_tiles; t += tstride) {
    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const long tn = q < 3 ? t : t + tstride;
      if (tn < a.n_tiles) {
        issue(tn, (q + 1) & 3, buf ^ 1);
        RN_WAIT_VM(8);  // everything older than the eight pieces just requested has landed (incl. the last tile's stores)
      } else {
        RN_WAIT_VM(0);
      }
      const char* rowp = smem + stg0 + buf * STAGE + li * 256;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(rowp + (((4 * ks + 2 * hh) ^ (li & 15)) << 4));
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(rowp + (((4 * ks + 2 * hh + 1) ^ (li & 15)) << 4));
        float x[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) { x[e] = v0[e]; x[4 + e] = v1[e]; }
        u32x4 la[3];
        split8(x, la);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          u32x4 lb[3];
#pragma unroll
          for (int j = 0; j < 3; ++j)
            lb[j] = *reinterpret_cast<const u32x4*>(smem + ((j * NT + nt) * 16 + 4 * q + ks) * 1024 + lane * 16);
          // (A limb, W limb): 00 01 10 11 02 20
#pragma unroll
          for (int pr = 0; pr < 6; ++pr) {
            const int il = (pr == 2 || pr == 3) ? 1 : (pr == 5 ? 2 : 0);
            const int jl = (pr == 1 || pr == 3) ? 1 : (pr == 4 ? 2 : 0);
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, la[il]),
                                                              __builtin_bit_cast(bf16x8, lb[jl]), acc[nt], 0, 0, 0);
          }
        }
      }
      buf ^= 1;
    }
    // ---- the tile's results: lane (hh, li) holds column 32 nt + li of the points 8 g + 4 hh + r.  Whole tiles run branch-free
    // inside ONE predicated region per output tile (all previous values requested, then all stores): with a branch per element
    // the compiler waits for vmcnt(0) -- i.e. for the PREVIOUS store to complete -- before every store ----
    const long p0 = t * 32;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int col = 32 * nt + li;
      if (col < a.N) {
        float* o = a.C + (p0 + 4 * hh) * a.ldc + col;
        if (p0 + 32 <= a.P) {
          float prev[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) prev[r] = 0.f;
          if (a.accumulate) {
#pragma unroll
            for (int r = 0; r < 16; ++r) prev[r] = o[(8 * (r >> 2) + (r & 3)) * a.ldc];
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) o[(8 * (r >> 2) + (r & 3)) * a.ldc] = acc[nt][r] + prev[r];
        } else {  // the last, partial tile
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = 8 * (r >> 2) + (r & 3);
            if (p0 + 4 * hh + row < a.P) {
              float* e = o + row * a.ldc;
              *e = a.accumulate ? *e + acc[nt][r] : acc[nt][r];
            }
          }
        }
      }
    }
  }
  RN_WAIT_VM(0);  // no LDS-DMA may be in flight when the workgroup's LDS is released
}

}  // namespace

extern "C" int hold_gemm_narrow_x6(const float* A, int32_t lda, const float* W, int32_t ldw, float* C, int32_t ldc, int64_t P,
                                   int32_t N, int32_t accumulate, hold_stream_t stream) {
  if (!A || !W || !C || P < 0 || N <= 0 || N > 64 || lda < 256 || ldw < 256 || ldc < N || (lda & 3) || (ldw & 3) ||
      ((uintptr_t)A & 15) || ((uintptr_t)W & 15) || ((uintptr_t)C & 3))
    return HOLD_E_ARG;
  if (P == 0) return HOLD_OK;
  static int n_cu = 0;
  static bool attr_set = false;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return HOLD_E_LAUNCH;
    n_cu = prop.multiProcessorCount;
  }
  constexpr int LDS1 = WL_NT + 4 * 2 * STAGE, LDS2 = 2 * WL_NT + 4 * 2 * STAGE;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)rnarrow_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS1) != hipSuccess ||
        hipFuncSetAttribute((const void*)rnarrow_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS2) != hipSuccess)
      return HOLD_E_LAUNCH;
    attr_set = true;
  }
  NArgs a;
  a.A = A; a.lda = lda; a.W = W; a.ldw = ldw; a.C = C; a.ldc = ldc; a.P = P; a.N = N; a.accumulate = accumulate;
  a.n_tiles = (P + 31) / 32;
  long g = (a.n_tiles + 3) / 4;
  if (g > n_cu) g = n_cu;
  hipStream_t s = (hipStream_t)stream;
  if (N <= 32)
    hipLaunchKernelGGL(rnarrow_kernel<1>, dim3((unsigned)g), dim3(256), LDS1, s, a);
  else
    hipLaunchKernelGGL(rnarrow_kernel<2>, dim3((unsigned)g), dim3(256), LDS2, s, a);
  return hipGetLastError() == hipSuccess ? HOLD_OK : HOLD_E_LAUNCH;
}
