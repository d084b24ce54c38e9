// Split-precision sampler trunk, second generation (hold_fused_sdf_x6q): the contract and the arithmetic of
// hold_fused_sdf_x6 (fused_sdf.hip: every fp32 operand = exact sum of three bf16 limbs, six of the nine limb products
// on v_mfma_f32_32x32x16_bf16, fp32 accumulation), re-tiled around what limited that kernel.
//
// Measured (round 2, 1 x MI355X): fused_sdf_x6p runs 176 TFLOP/s fp32-equivalent = 42 % of what its MFMA stream alone
// would allow.  Per 16-wide k step a 64-point workgroup issues 96 MFMAs (768 cycles per SIMD), reads 48 KiB of limb
// planes from LDS (192 LDS cycles) -- and pulls 24 KiB of pre-split weight limbs (6 B per weight) from L2.  At the
// measured rate that is 13 B/clk/CU, 8.2 TB/s over the chip: the L2 -> CU weight stream, not the matrix cores or LDS,
// is the wall (the fp32-MFMA trunk needs 3.2 B/clk/CU).  Two changes cut the stream per point by 2.25x:
//   * weights stay fp32 in memory (4 B instead of 6 B) and are split into limbs in registers, by truncation
//     (v = hi16(v) + hi16(v - hi16(v)) + rest, exact: 8 + 8 + 8 significand bits) -- ~44 VALU operations per k step per
//     lane, hidden in the shadow of that step's 18 MFMAs;
//   * a workgroup owns 96 points (three 32-point tiles) instead of 64: the limb planes take 3 x 96 x 264 bf16 = 148.5 KiB
//     of the 160 KiB LDS, which is possible because the Fourier embedding is no longer kept in LDS for the skip layer
//     but recomputed there from the canonical point (39 sin / cos per point, once).
// Wave w = output features [32 w, 32 w + 32) x 3 point tiles (three accumulators, each re-used every 3rd MFMA).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hold_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int NE = 39, SKIP_OUT = 217;
constexpr int QP = 96, QT = 3, QROW = 264, QPLANE = QP * QROW;  // points, tiles, bf16 row stride, bf16 per limb plane
constexpr int Q_L0_STEPS = 3, Q_LK_STEPS = 16;                   // K = 48 (39 zero-padded) and 256 in steps of 16
constexpr int Q_STEP_F4 = 8 * 64 * 2;                            // f32x4 units per k step: [8 waves][64 lanes][2]

struct QArgs {
  const float* xc; int ldx; long P;
  const f32x4* wq;      // fp32 weights, [layer][step][wave][lane][8]
  const float* bias;    // [8][256]
  const float* w8;      // [256] sdf row of the last layer
  float b8;
  const float* barf;    // [39] or null
  float* sdf; int lds;
};

__device__ __forceinline__ float softplus100(float y) {  // same arithmetic as fused_sdf.hip / chain.hip
  const float z = y * 100.0f;
  const float e = __expf(-fabsf(z));
  const float l_log = __logf(1.0f + e);
  const float l_ser = e * (1.0f - e * (0.5f - 0.33333334f * e));
  const float l = (e > 1e-3f) ? l_log : l_ser;
  const float r = (fmaxf(z, 0.f) + l) * 0.01f;
  return (z > 20.0f) ? y : r;
}

__device__ __forceinline__ float embed_value(const float* x3, int j, const float* barf) {
  float v;
  if (j < 3) {
    v = x3[j];
  } else {
    const int q = (j - 3) / 3, dim = (j - 3) % 3, k = q >> 1;
    const float arg = x3[dim] * (float)(1 << k);
    v = (q & 1) ? cosf(arg) : sinf(arg);
  }
  return barf ? v * barf[j] : v;
}

// 8 fp32 weights -> three bf16x8 limbs by truncation (exact): the high halves of v, of v - hi(v), and of the rest
__device__ __forceinline__ void split8(const f32x4& w0, const f32x4& w1, bf16x8 (&b)[3]) {
  unsigned u[8], r1[8], r2[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    u[i] = __builtin_bit_cast(unsigned, w0[i]);
    u[4 + i] = __builtin_bit_cast(unsigned, w1[i]);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float v = __builtin_bit_cast(float, u[i]);
    const float a = v - __builtin_bit_cast(float, u[i] & 0xffff0000u);
    r1[i] = __builtin_bit_cast(unsigned, a);
    r2[i] = __builtin_bit_cast(unsigned, a - __builtin_bit_cast(float, r1[i] & 0xffff0000u));
  }
  u32x4 p0, p1, p2;
#pragma unroll
  for (int i = 0; i < 4; ++i) {  // {hi16(x[2i+1]), hi16(x[2i])} in one dword
    p0[i] = __builtin_amdgcn_perm(u[2 * i + 1], u[2 * i], 0x07060302u);
    p1[i] = __builtin_amdgcn_perm(r1[2 * i + 1], r1[2 * i], 0x07060302u);
    p2[i] = __builtin_amdgcn_perm(r2[2 * i + 1], r2[2 * i], 0x07060302u);
  }
  b[0] = __builtin_bit_cast(bf16x8, p0);
  b[1] = __builtin_bit_cast(bf16x8, p1);
  b[2] = __builtin_bit_cast(bf16x8, p2);
}

// one layer: acc[m] (3 tiles x 16) = W_l (this wave's 32 features) x activations (limb planes in LDS).
// Register budget: 2 waves per SIMD = 256 registers per lane.  Three accumulators (48) + one step's nine limb fragments
// (36) + the weight limbs (12) + next step's fp32 weights (8) + the split's temporaries; the limb planes are read at
// the top of the step they are used in (LDS latency is covered by the other wave of the SIMD, which is half a step
// out of phase), the weights -- L2 latency -- one step ahead.
template <int STEPS>
__device__ __forceinline__ void q_layer(const f32x4* __restrict__ wq, const f32x4* __restrict__ nxt,
                                        const __bf16* __restrict__ prow, f32x16 (&acc)[QT], f32x4 (&wn)[2]) {
#pragma unroll
  for (int m = 0; m < QT; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
#pragma unroll 2
  for (int s = 0; s < STEPS; ++s) {
    bf16x8 a[QT][3], b[3];
    // ---- issue: this step's limb planes (LDS), next step's fp32 weights (L2) ----
#pragma unroll
    for (int m = 0; m < QT; ++m)
#pragma unroll
      for (int t = 0; t < 3; ++t)
        a[m][t] = *reinterpret_cast<const bf16x8*>(prow + t * QPLANE + m * 32 * QROW + s * 16);
    split8(wn[0], wn[1], b);
    const f32x4* src = (s + 1 < STEPS) ? wq + (s + 1) * Q_STEP_F4 : nxt;
    wn[0] = src[0];
    wn[1] = src[1];
    __builtin_amdgcn_sched_barrier(0);
    // ---- compute: (w limb, a limb) = 00 01 10 11 02 20, tile-minor so an accumulator recurs every 3rd MFMA ----
#pragma unroll
    for (int pr = 0; pr < 6; ++pr) {
      const int wl = (pr == 2 || pr == 3) ? 1 : (pr == 5 ? 2 : 0);
      const int al = (pr == 1 || pr == 3) ? 1 : (pr == 4 ? 2 : 0);
#pragma unroll
      for (int m = 0; m < QT; ++m)
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[wl], a[m][al], acc[m], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

__global__ __launch_bounds__(512) void fused_sdf_x6q_kernel(QArgs a) {
  constexpr int NTHR = 512;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __bf16* planes = reinterpret_cast<__bf16*>(smem);   // [3][96][264] bf16
  float* red = smem + 3 * QPLANE / 2;                  // [8 waves][96] partial sdf
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hh = lane >> 5, li = lane & 31;
  const int nb = wave * 32 + 4 * hh;
  const f32x4* w0 = a.wq + (wave * 64 + lane) * 2;

  for (long blk = blockIdx.x; blk * QP < a.P; blk += gridDim.x) {
    const long p0 = blk * QP;
    f32x4 wn[2] = {w0[0], w0[1]};
    for (int e = tid; e < QP * 48; e += NTHR) {  // layer-0 input: Fourier embedding, zero-padded to K = 48, as limbs
      const int p = e / 48, j = e % 48;
      const long gp = p0 + p;
      float v = 0.f;
      if (j < NE && gp < a.P) v = embed_value(a.xc + gp * a.ldx, j, a.barf);
      const __bf16 h1 = (__bf16)v;
      const float r1 = v - (float)h1;
      const __bf16 h2 = (__bf16)r1;
      planes[p * QROW + j] = h1;
      planes[QPLANE + p * QROW + j] = h2;
      planes[2 * QPLANE + p * QROW + j] = (__bf16)(r1 - (float)h2);
    }
    __syncthreads();

    const __bf16* prow = planes + li * QROW + hh * 8;
    const f32x4* wl = w0;
    float part[QT] = {0.f, 0.f, 0.f};
    for (int layer = 0; layer < 8; ++layer) {
      f32x16 acc[QT];
      if (layer == 0) {
        q_layer<Q_L0_STEPS>(wl, wl + Q_L0_STEPS * Q_STEP_F4, prow, acc, wn);
        wl += Q_L0_STEPS * Q_STEP_F4;
      } else {
        q_layer<Q_LK_STEPS>(wl, layer < 7 ? wl + Q_LK_STEPS * Q_STEP_F4 : w0, prow, acc, wn);
        wl += Q_LK_STEPS * Q_STEP_F4;
      }
      __syncthreads();  // every wave has finished READING this layer's input
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n4 = nb + 8 * g;
        const f32x4 bias = *reinterpret_cast<const f32x4*>(a.bias + layer * 256 + n4);
        const f32x4 w8v = *reinterpret_cast<const f32x4*>(a.w8 + n4);
#pragma unroll
        for (int m = 0; m < QT; ++m) {
          const int p = m * 32 + li;
          f32x4 v;
#pragma unroll
          for (int c = 0; c < 4; ++c) v[c] = softplus100(acc[m][4 * g + c] + bias[c]);
          if (layer == 3 && n4 + 3 >= SKIP_OUT) {  // skip connection: columns 217.. are the embedding itself (recomputed)
            const long gp = p0 + p;
            float x3[3] = {0.f, 0.f, 0.f};
            if (gp < a.P) {
              x3[0] = a.xc[gp * a.ldx];
              x3[1] = a.xc[gp * a.ldx + 1];
              x3[2] = a.xc[gp * a.ldx + 2];
            }
#pragma unroll
            for (int c = 0; c < 4; ++c)
              if (n4 + c >= SKIP_OUT) v[c] = (gp < a.P) ? embed_value(x3, n4 + c - SKIP_OUT, a.barf) : 0.f;
          }
          if (layer == 7) {  // sdf row of the last layer straight from the registers
            part[m] += v[0] * w8v[0] + v[1] * w8v[1] + v[2] * w8v[2] + v[3] * w8v[3];
          } else {
            bf16x4 l1, l2, l3;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const __bf16 h1 = (__bf16)v[c];
              const float r1 = v[c] - (float)h1;
              const __bf16 h2 = (__bf16)r1;
              l1[c] = h1;
              l2[c] = h2;
              l3[c] = (__bf16)(r1 - (float)h2);
            }
            __bf16* dst = planes + p * QROW + n4;
            *reinterpret_cast<bf16x4*>(dst) = l1;
            *reinterpret_cast<bf16x4*>(dst + QPLANE) = l2;
            *reinterpret_cast<bf16x4*>(dst + 2 * QPLANE) = l3;
          }
        }
      }
      __syncthreads();
    }
    // ---- sdf = w8 . h7 + b8: lanes hh = 0 / 1 hold complementary features of the same points ----
#pragma unroll
    for (int m = 0; m < QT; ++m) {
      const float s = part[m] + __shfl_xor(part[m], 32);
      if (hh == 0) red[wave * QP + m * 32 + li] = s;
    }
    __syncthreads();
    if (tid < QP) {
      float s = a.b8;
#pragma unroll
      for (int w = 0; w < 8; ++w) s += red[w * QP + tid];
      if (p0 + tid < a.P) a.sdf[(p0 + tid) * a.lds] = s;
    }
    __syncthreads();
  }
}

}  // namespace

extern "C" int64_t hold_fused_sdf_x6q_pack_floats(void) {
  return (int64_t)(Q_L0_STEPS + 7 * Q_LK_STEPS) * Q_STEP_F4 * 4;
}

extern "C" int hold_fused_sdf_x6q(const float* xc, int32_t ldx, int64_t P, const float* wpack_q, const float* bias,
                                  const float* w8, float b8, const float* barf_w, float* sdf, int32_t ld_sdf,
                                  hold_stream_t st) {
  if (!xc || !wpack_q || !bias || !w8 || !sdf || ldx < 3 || ld_sdf < 1 || P < 0) return HOLD_E_ARG;
  if (((uintptr_t)wpack_q & 15) || ((uintptr_t)w8 & 15) || ((uintptr_t)bias & 15)) return HOLD_E_ARG;
  if (P == 0) return HOLD_OK;
  static int n_cu = 0;
  static bool attr_set = false;
  const size_t sh = (size_t)3 * QPLANE * 2 + (size_t)8 * QP * sizeof(float);
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return HOLD_E_LAUNCH;
    n_cu = prop.multiProcessorCount;
  }
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)fused_sdf_x6q_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh) !=
        hipSuccess)
      return HOLD_E_LAUNCH;
    attr_set = true;
  }
  QArgs a = {xc, ldx, (long)P, reinterpret_cast<const f32x4*>(wpack_q), bias, w8, b8, barf_w, sdf, ld_sdf};
  const long blocks = (P + QP - 1) / QP;
  hipLaunchKernelGGL(fused_sdf_x6q_kernel, dim3((unsigned)(blocks < n_cu ? blocks : n_cu)), dim3(512), sh, (hipStream_t)st,
                     a);
  return hipGetLastError() == hipSuccess ? HOLD_OK : HOLD_E_LAUNCH;
}
