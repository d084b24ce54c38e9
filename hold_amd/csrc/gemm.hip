// Matrix-core GEMMs for the HOLD MLPs (gfx950 / CDNA4 only).
//
//   hold_gemm_nt[_x6] : C[P][N] = epi(alpha * A[P][K] . W[N][K]^T + bias [+ row[p] * col[n]])   (layer forward, input
//                       gradient, backward-data and double-backward sweeps; the caller passes W or W^T)
//   hold_wgrad[_x6]   : dW[N][K] = R[P][N]^T . X[P][K]                      (weight gradients, reduction over points)
//   hold_wcolsum      : out[n] = sum_p w[p] X[p][n]                         (rank-1 companion of wgrad, HBM streaming)
//   hold_head3_fwd/bwd: the 3-output colour head and its backward           (HBM streaming)
//
// Arithmetic: v_mfma_f32_32x32x2_f32 (exact fp32 operands, 157 TFLOP/s peak) or, in the _x6 entry points, both operands
// split exactly into three bf16 limbs as their fragments leave LDS and 6 of the 9 limb products on
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  Roofline: MFMA-bound; algorithmic work 2*P*N*K flop per launch, HBM
// traffic (P*(K+N) + N*K)*4 bytes.
//
// Tiling (hold_gemm_nt): 256 threads = 4 waves as 2(points) x 2(outputs); block tile 128 points x 256 outputs (128 x 128
// for N <= 128), wave tile 64 x 128 = 2 x 4 MFMA tiles, K stepped by 16 (32 for the narrow tile) through a
// double-buffered LDS stage filled by global_load_lds (69.6 KiB -> two blocks per CU, 2 waves/SIMD).
// The k index is permuted inside a stage so that each lane fetches contiguous 16-byte chunks with ds_read_b128; A and W
// use the same permutation, so products pair up and only the fp32 summation order differs from a sequential dot product.
// The 16-byte chunk index is XOR-swizzled by the row (on the DMA source address and on the fragment reads): conflict-free.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/hold_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BM = 128;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
struct Limbs3 { bf16x8 l[3]; };
// exact three-limb bf16 decomposition of 8 fp32 values (x = l0 + l1 + l2, each limb the bf16 rounding of the residual)
__device__ __forceinline__ Limbs3 split8s(const float (&x)[8]) {
  Limbs3 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const __bf16 h1 = (__bf16)x[e];
    const float r1 = x[e] - (float)h1;
    const __bf16 h2 = (__bf16)r1;
    o.l[0][e] = h1;
    o.l[1][e] = h2;
    o.l[2][e] = (__bf16)(r1 - (float)h2);
  }
  return o;
}
// the same by truncation (v_and / v_sub / v_perm only), see fused_sdf.hip:split8_trunc
__device__ __forceinline__ Limbs3 split8s_trunc(const float (&x)[8]) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  u32x4 p1, p2, p3;
#pragma unroll
  for (int j = 0; j < 4; ++j) {  // two values at a time: v_pk_add_f32 for the remainders (36 VALU slots per 8 values)
    const f32x2 v = {x[2 * j], x[2 * j + 1]};
    const u32x2 xb = __builtin_bit_cast(u32x2, v);
    const f32x2 r1 = v - __builtin_bit_cast(f32x2, xb & 0xffff0000u);
    const u32x2 r1b = __builtin_bit_cast(u32x2, r1);
    const u32x2 r2b = __builtin_bit_cast(u32x2, r1 - __builtin_bit_cast(f32x2, r1b & 0xffff0000u));
    p1[j] = __builtin_amdgcn_perm(xb[1], xb[0], 0x07060302u);
    p2[j] = __builtin_amdgcn_perm(r1b[1], r1b[0], 0x07060302u);
    p3[j] = __builtin_amdgcn_perm(r2b[1], r2b[0], 0x07060302u);
  }
  Limbs3 o;
  o.l[0] = __builtin_bit_cast(bf16x8, p1);
  o.l[1] = __builtin_bit_cast(bf16x8, p2);
  o.l[2] = __builtin_bit_cast(bf16x8, p3);
  return o;
}

// softplus(y, beta=100, threshold=20) = y for 100y > 20, else (max(z,0) + log1p(exp(-|z|))) / 100, z = 100y.
// exp/log run on the hardware transcendental units (v_exp_f32 / v_log_f32); log1p switches to its series
// for small arguments, so the absolute error stays ~1e-9 (x 1/100), the class of fp32 GEMM reordering noise.
__device__ __forceinline__ float softplus100(float y) {
  const float z = y * 100.0f;
  if (z > 20.0f) return y;
  const float e = __expf(-fabsf(z));
  const float l = (e > 1e-3f) ? __logf(1.0f + e) : e * (1.0f - e * (0.5f - 0.33333334f * e));
  return (fmaxf(z, 0.f) + l) * 0.01f;
}
// softplus'(x) recovered from h = softplus(x): sigmoid(100x) = 1 - exp(-100h) (series near 0 keeps the
// relative accuracy of the small derivatives)
__device__ __forceinline__ float dsp_from_h(float h) {
  const float x = 100.0f * h;
  if (x < 0.05f) return x * (1.0f - x * (0.5f - x * (0.16666667f - 0.041666668f * x)));
  return 1.0f - __expf(-x);
}

// NT = number of 32-wide output tiles per wave: 2 -> block tile 128 x 128 (BK 32), 4 -> 128 x 256 (BK 16).
// The wide tile loads 24 KiB per 64 MFMAs/wave instead of 32 KiB (and streams A once instead of twice):
// the kernel is limited by the per-CU load path (ablation in DESIGN.md), not by MFMA issue.
// X6: 0 = fp32 MFMA; 2 = split precision: both operand fragments are split into three bf16 limbs (truncation split,
// exact) as they leave LDS and 6 of the 9 limb products run on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.
// Staging, swizzle, tile walk and epilogues are shared; a lane's 8 consecutive k of a fragment row are the two
// 16-byte chunks it reads anyway.
// NBUF: operand stages in LDS.  2 = the next stage is requested while this one is computed; 3 = two stages ahead (the
// split-precision product spends 2.7x fewer matrix-core cycles per stage, one stage of compute no longer covers the
// load latency) -- 3 x 24 KiB for the wide tile, still two blocks per CU.
template <int EPI, int NT, int X6 = 0, int NBUF = 2>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(hold_gemm_desc d, int total_tiles, int stagger) {
  constexpr int BNc = 64 * NT;              // block outputs
  constexpr int BKc = (NT == 2) ? 32 : 16;  // k per stage
  constexpr int CH = BKc / 4;               // 16-byte chunks per staged row
  constexpr int RPI = 64 / CH;              // rows per DMA instruction
  constexpr int KSH = (NT == 2) ? 1 : 2;    // swizzle key = (row >> KSH) & (CH - 1)
  constexpr int QN = BKc / 8;               // ds_read_b128 per fragment row per stage
  constexpr int CSTR = 68;
  constexpr int STG = NBUF * (BM + BNc) * BKc, EPS = 4 * 64 * CSTR;  // operand stages / epilogue staging (69.6 KiB)
  __shared__ __attribute__((aligned(16))) float smem[STG > EPS ? STG : EPS];
  float* sA = smem;
  float* sW = smem + NBUF * BM * BKc;
  constexpr int NI = BM / RPI / 4 + BNc / RPI / 4;  // DMA instructions per wave and stage

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int hh = lane >> 5, li = lane & 31;
  const int ntn = (d.N + BNc - 1) / BNc;
  const int nk = (d.K + BKc - 1) / BKc;

  // ---- operand staging: global -> LDS DMA (global_load_lds_dwordx4), no VGPR round trip ----
  // One wave instruction moves 64 x 16 B = RPI tile rows into a linear 1 KiB of LDS.  The LDS image is
  // [rows][CH chunks of 16 B] with the chunk index XOR-swizzled by key(row): the DMA destination is
  // lane-linear, so the swizzle is applied to the per-lane SOURCE address and again on the fragment reads
  // (both sides or neither).  With these keys the ds_read_b128 fragment reads of a 16-lane group hit 16
  // distinct 16-byte bank groups (conflict-free).
  auto key = [](int row) { return (row >> KSH) & (CH - 1); };
  auto stage = [&](long m0, int n0, int kt, int buf) {
    const int k0 = kt * BKc;
    if (k0 + BKc <= d.K) {
#pragma unroll
      for (int j = 0; j < BM / RPI / 4; ++j) {
        const int row = (wave * (BM / RPI / 4) + j) * RPI + lane / CH;
        const int lc = (lane % CH) ^ key(row);
        long p = m0 + row;
        p = p < d.P ? p : (long)d.P - 1;
        const float* ga = d.A + p * (long)d.lda + k0 + lc * 4;
        float* la = sA + buf * (BM * BKc) + (wave * (BM / RPI / 4) + j) * 256;
        if (!(stagger & 2048))
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)ga,
                                           (__attribute__((address_space(3))) void*)la, 16, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < BNc / RPI / 4; ++j) {
        const int row = (wave * (BNc / RPI / 4) + j) * RPI + lane / CH;
        const int lc = (lane % CH) ^ key(row);
        int n = n0 + row;
        n = n < d.N ? n : d.N - 1;
        const float* gw = d.W + (long)n * d.ldw + k0 + lc * 4;
        float* lw = sW + buf * (BNc * BKc) + (wave * (BNc / RPI / 4) + j) * 256;
        if (!(stagger & 4096))
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gw,
                                           (__attribute__((address_space(3))) void*)lw, 16, 0, 0);
      }
    } else {  // ragged last chunk (K % BK != 0): masked register path into the same swizzled image
      for (int e = tid; e < (BM + BNc) * CH; e += 256) {
        const bool isA = e < BM * CH;
        const int ee = isA ? e : e - BM * CH;
        const int row = ee / CH, c = ee % CH;
        const int k = k0 + c * 4;
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
        if (isA) {
          const long p = m0 + row;
          if (p < d.P && k < d.K) z = *reinterpret_cast<const f32x4*>(d.A + p * (long)d.lda + k);
          *reinterpret_cast<f32x4*>(sA + buf * (BM * BKc) + row * BKc + ((c ^ key(row)) << 2)) = z;
        } else {
          const int n = n0 + row;
          if (n < d.N && k < d.K) z = *reinterpret_cast<const f32x4*>(d.W + (long)n * d.ldw + k);
          *reinterpret_cast<f32x4*>(sW + buf * (BNc * BKc) + row * BKc + ((c ^ key(row)) << 2)) = z;
        }
      }
    }
  };
  auto stage_wait = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };
  // all stages but the most recently requested one have landed (that one is a full DMA stage: NI instructions in flight)
  auto stage_wait_keep1 = [&]() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");
    __syncthreads();
  };
  auto full_stage = [&](int kt) { return kt * BKc + BKc <= d.K; };

  // Persistent blocks: the grid is 2 blocks per CU, each walks tiles blockIdx.x, +gridDim.x, ...  The
  // second half of the grid (the blocks that share CUs with the first half) starts half a tile late, so
  // one block's prologue / epilogue overlaps the other's MFMA phase instead of coinciding with it.
  if ((stagger & 255) && blockIdx.x >= (gridDim.x >> 1)) {
    for (int i = 0; i < (stagger & 255); ++i) __builtin_amdgcn_s_sleep(127);
  }

  auto al16 = [](const void* q, int ld) { return (((uintptr_t)q & 15) == 0) && ((ld & 3) == 0); };
  bool vec_ok = al16(d.C, d.ldc) && (!d.bias || al16(d.bias, 0));
  if (EPI == HOLD_EPI_MUL_DSP) vec_ok = vec_ok && al16(d.aux1, d.ldaux1) && (!d.aux2 || al16(d.aux2, d.ldaux2));
  if (EPI == HOLD_EPI_MUL_DRELU || EPI == HOLD_EPI_MUL_DSIG) vec_ok = vec_ok && al16(d.aux1, d.ldaux1);
  if (EPI == HOLD_EPI_DBWD)
    vec_ok = vec_ok && al16(d.aux1, d.ldaux1) && al16(d.aux2, d.ldaux2) && al16(d.out2, d.ldout2);

  if (d.r1_row) vec_ok = vec_ok && al16(d.r1_col, 0);

  auto epi_scalar = [&](long p, int n, float acc_v) {
    float y = acc_v * d.alpha + (d.bias ? d.bias[n] : 0.f);
    if (d.r1_row) y += d.r1_row[p * (long)d.ldr1] * d.r1_col[n];
    if (n >= d.n_split) {
      float* o = d.C2 + p * (long)d.ldc2 + (n - d.n_split);
      *o = d.accumulate ? *o + y : y;
      return;
    }
    float* o = d.C + p * (long)d.ldc + n;
    if (EPI == HOLD_EPI_NONE) {
      *o = d.accumulate ? *o + y : y;
    } else if (EPI == HOLD_EPI_SOFTPLUS) {
      *o = softplus100(y);
    } else if (EPI == HOLD_EPI_RELU) {
      *o = fmaxf(y, 0.f);
    } else if (EPI == HOLD_EPI_SIGMOID) {
      *o = 1.0f / (1.0f + __expf(-y));
    } else if (EPI == HOLD_EPI_MUL_DSP) {
      float v = y * dsp_from_h(d.aux1[p * (long)d.ldaux1 + n]);
      if (d.aux2) v += d.aux2[p * (long)d.ldaux2 + n];
      *o = v;
    } else if (EPI == HOLD_EPI_MUL_DRELU) {
      *o = d.aux1[p * (long)d.ldaux1 + n] > 0.f ? y : 0.f;
    } else if (EPI == HOLD_EPI_DBWD) {
      const float h = d.aux1[p * (long)d.ldaux1 + n];
      const float t = d.aux2[p * (long)d.ldaux2 + n];
      const float e = __expf(-100.0f * h);
      *o = y * dsp_from_h(h);
      d.out2[p * (long)d.ldout2 + n] = 100.0f * y * t * e;
    } else if (EPI == HOLD_EPI_MUL_DSIG) {
      const float sg = d.aux1[p * (long)d.ldaux1 + n];
      *o = y * sg * (1.0f - sg);
    }
  };


  for (int tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
    const long m0 = (long)(tile / ntn) * BM;
    const int n0 = (tile % ntn) * BNc;

    f32x16 acc[2][NT];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < NT; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    stage(m0, n0, 0, 0);
    if (NBUF == 3 && nk > 1) {
      stage(m0, n0, 1, 1);
      if (full_stage(1)) stage_wait_keep1(); else stage_wait();
    } else {
      stage_wait();
    }

    const int rowa0 = wm * 64 + li, roww0 = wn * (32 * NT) + li;
    // 32-wide output tiles of this wave that hold any column < N (wave-uniform): the products of the others are
    // skipped (N = 40 / 48-column tail tiles would otherwise run 4 - 8x their useful MFMAs)
    const int nbv_raw = (d.N - (n0 + wn * (32 * NT)) + 31) / 32;
    const int nbv = nbv_raw < 0 ? 0 : (nbv_raw > NT ? NT : nbv_raw);
    for (int kt = 0; kt < nk; ++kt) {
      const int buf = kt % NBUF;
      if (kt + NBUF - 1 < nk && !(stagger & 256)) stage(m0, n0, kt + NBUF - 1, (kt + NBUF - 1) % NBUF);
      const float* pa = sA + buf * (BM * BKc) + rowa0 * BKc;
      const float* pw = sW + buf * (BNc * BKc) + roww0 * BKc;
      // fragments of the next 16-byte k group are requested before this group's MFMAs (pinned with sched barriers:
      // left alone the scheduler issues them a couple of MFMAs before their use)
      if constexpr (X6 != 0) {
#pragma unroll
        for (int g = 0; g < QN / 2; ++g) {
          const int c = hh * QN + 2 * g;
          Limbs3 la[2];
#pragma unroll
          for (int a = 0; a < 2; ++a) {
            const int kk = key(rowa0 + 32 * a);
            const f32x4 u0 = *reinterpret_cast<const f32x4*>(pa + a * 32 * BKc + ((c ^ kk) << 2));
            const f32x4 u1 = *reinterpret_cast<const f32x4*>(pa + a * 32 * BKc + (((c + 1) ^ kk) << 2));
            const float x[8] = {u0[0], u0[1], u0[2], u0[3], u1[0], u1[1], u1[2], u1[3]};
            la[a] = split8s_trunc(x);
          }
          // output tiles two at a time (4 accumulators in rotation, 12 + 24 limb registers live)
#pragma unroll
          for (int bp = 0; bp < NT; bp += 2) {
            if (bp >= nbv) break;
            Limbs3 lb[2];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
              const int kk = key(roww0 + 32 * (bp + b));
              const f32x4 u0 = *reinterpret_cast<const f32x4*>(pw + (bp + b) * 32 * BKc + ((c ^ kk) << 2));
              const f32x4 u1 = *reinterpret_cast<const f32x4*>(pw + (bp + b) * 32 * BKc + (((c + 1) ^ kk) << 2));
              const float x[8] = {u0[0], u0[1], u0[2], u0[3], u1[0], u1[1], u1[2], u1[3]};
              lb[b] = split8s_trunc(x);
            }
#pragma unroll
            for (int pr = 0; pr < 6; ++pr) {
              const int il = (pr == 2 || pr == 3) ? 1 : (pr == 5 ? 2 : 0);  // (A limb, W limb): 00 01 10 11 02 20
              const int jl = (pr == 1 || pr == 3) ? 1 : (pr == 4 ? 2 : 0);
#pragma unroll
              for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                  acc[a][bp + b] =
                      __builtin_amdgcn_mfma_f32_32x32x16_bf16(la[a].l[il], lb[b].l[jl], acc[a][bp + b], 0, 0, 0);
            }
          }
        }
        if (!(stagger & 512)) {
          // stage kt + 1 must have landed; with three buffers the stage just requested (kt + 2) stays in flight
          if (NBUF == 3 && kt + 2 < nk && full_stage(kt + 2)) stage_wait_keep1(); else stage_wait();
        }
        continue;
      }
      f32x4 avn[2], bvn[NT];
      auto frag = [&](int q) {
        const int c = hh * QN + q;
#pragma unroll
        for (int a = 0; a < 2; ++a)
          avn[a] = *reinterpret_cast<const f32x4*>(pa + a * 32 * BKc + ((c ^ key(rowa0 + 32 * a)) << 2));
#pragma unroll
        for (int b = 0; b < NT; ++b)
          bvn[b] = *reinterpret_cast<const f32x4*>(pw + b * 32 * BKc + ((c ^ key(roww0 + 32 * b)) << 2));
      };
      frag(0);
#pragma unroll
      for (int q = 0; q < QN; ++q) {
        f32x4 av[2], bv[NT];
#pragma unroll
        for (int a = 0; a < 2; ++a) av[a] = avn[a];
#pragma unroll
        for (int b = 0; b < NT; ++b) bv[b] = bvn[b];
        if (q + 1 < QN) frag(q + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int cc = 0; cc < 4; ++cc)
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < NT; ++b)
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a][cc], bv[b][cc], acc[a][b], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (!(stagger & 512)) stage_wait();
    }
    if (stagger & 1024) continue;

    // ---- epilogue ----
    // The accumulator tile is transposed through LDS (the operand stages are free after the last barrier)
    // 64 columns at a time, so that every lane handles 4 CONSECUTIVE output columns of one point: aux reads
    // and the result stores become 16-byte accesses, 16 lanes covering a 256-byte row segment.
    float* sc = smem + wave * (64 * CSTR);
    const int c4 = (lane & 15) * 4;
#pragma unroll
    for (int half = 0; half < NT / 2; ++half) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            sc[(a * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh) * CSTR + b * 32 + li] = acc[a][half * 2 + b][r];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const int n = n0 + wn * (32 * NT) + half * 64 + c4;
      // Fast path (wave-uniform): the wave's 64 x 64 sub-tile lies inside the matrix and every access is a 16-byte one.
      // Branch-free, eight rows at a time: the side inputs of eight rows are requested together, then consumed -- the
      // generic loop below waits out one memory latency per row (load, s_waitcnt, store), 16 times per sub-tile.
      const int nlim = d.N < d.n_split ? d.N : d.n_split;
      const bool fast = vec_ok && (m0 + wm * 64 + 64 <= (long)d.P) && (n0 + wn * (32 * NT) + half * 64 + 64 <= nlim);
      if (fast) {
        f32x4 bias4 = {0.f, 0.f, 0.f, 0.f}, r1c = {0.f, 0.f, 0.f, 0.f};
        if (d.bias) bias4 = *reinterpret_cast<const f32x4*>(d.bias + n);
        if (d.r1_row) r1c = *reinterpret_cast<const f32x4*>(d.r1_col + n);
        constexpr bool AUX1 = EPI == HOLD_EPI_MUL_DSP || EPI == HOLD_EPI_MUL_DRELU || EPI == HOLD_EPI_DBWD ||
                              EPI == HOLD_EPI_MUL_DSIG;
        const bool aux2 = (EPI == HOLD_EPI_DBWD) || (EPI == HOLD_EPI_MUL_DSP && d.aux2 != nullptr);
        const bool prev = (EPI == HOLD_EPI_NONE) && d.accumulate;
        // rows per group: 8 with at most one side input, 4 with two (the other half's accumulators are still live)
        constexpr int GR = (EPI == HOLD_EPI_NONE || EPI == HOLD_EPI_MUL_DSP || EPI == HOLD_EPI_DBWD) ? 4 : 8;
#pragma unroll
        for (int it0 = 0; it0 < 16; it0 += GR) {
          f32x4 v[GR], a1[GR], a2[GR];
          float r1r[GR];
#pragma unroll
          for (int j = 0; j < GR; ++j) {
            const int row = (it0 + j) * 4 + (lane >> 4);
            const long p = m0 + wm * 64 + row;
            v[j] = *reinterpret_cast<const f32x4*>(sc + row * CSTR + c4);
            if (AUX1) a1[j] = *reinterpret_cast<const f32x4*>(d.aux1 + p * (long)d.ldaux1 + n);
            if (aux2) a2[j] = *reinterpret_cast<const f32x4*>(d.aux2 + p * (long)d.ldaux2 + n);
            if (prev) a2[j] = *reinterpret_cast<const f32x4*>(d.C + p * (long)d.ldc + n);
            if (d.r1_row) r1r[j] = d.r1_row[p * (long)d.ldr1];
          }
#pragma unroll
          for (int j = 0; j < GR; ++j) {
            const int row = (it0 + j) * 4 + (lane >> 4);
            const long p = m0 + wm * 64 + row;
            f32x4 y = v[j] * d.alpha + bias4;
            if (d.r1_row) y += r1c * r1r[j];
            f32x4 r = y;
            if (EPI == HOLD_EPI_NONE) {
              if (prev) r += a2[j];
            } else if (EPI == HOLD_EPI_SOFTPLUS) {
#pragma unroll
              for (int c = 0; c < 4; ++c) r[c] = softplus100(y[c]);
            } else if (EPI == HOLD_EPI_RELU) {
#pragma unroll
              for (int c = 0; c < 4; ++c) r[c] = fmaxf(y[c], 0.f);
            } else if (EPI == HOLD_EPI_SIGMOID) {
#pragma unroll
              for (int c = 0; c < 4; ++c) r[c] = 1.0f / (1.0f + __expf(-y[c]));
            } else if (EPI == HOLD_EPI_MUL_DSP) {
#pragma unroll
              for (int c = 0; c < 4; ++c) r[c] = y[c] * dsp_from_h(a1[j][c]);
              if (aux2) r += a2[j];
            } else if (EPI == HOLD_EPI_MUL_DRELU) {
#pragma unroll
              for (int c = 0; c < 4; ++c) r[c] = a1[j][c] > 0.f ? y[c] : 0.f;
            } else if (EPI == HOLD_EPI_DBWD) {
              f32x4 r2;
#pragma unroll
              for (int c = 0; c < 4; ++c) {
                const float e = __expf(-100.0f * a1[j][c]);
                r[c] = y[c] * dsp_from_h(a1[j][c]);
                r2[c] = 100.0f * y[c] * a2[j][c] * e;
              }
              *reinterpret_cast<f32x4*>(d.out2 + p * (long)d.ldout2 + n) = r2;
            } else if (EPI == HOLD_EPI_MUL_DSIG) {
              r = y * a1[j] * (1.0f - a1[j]);
            }
            *reinterpret_cast<f32x4*>(d.C + p * (long)d.ldc + n) = r;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        continue;
      }
#pragma unroll 4
      for (int it = 0; it < 16; ++it) {
        const int row = it * 4 + (lane >> 4);
        const long p = m0 + wm * 64 + row;
        if (p >= d.P || n >= d.N) continue;
        const f32x4 v = *reinterpret_cast<const f32x4*>(sc + row * CSTR + c4);
        if (vec_ok && n + 3 < d.n_split) {
          f32x4 y = v * d.alpha;
          if (d.bias) y += *reinterpret_cast<const f32x4*>(d.bias + n);
          if (d.r1_row) y += *reinterpret_cast<const f32x4*>(d.r1_col + n) * d.r1_row[p * (long)d.ldr1];
          f32x4* o = reinterpret_cast<f32x4*>(d.C + p * (long)d.ldc + n);
          if (EPI == HOLD_EPI_NONE) {
            if (d.accumulate) y += *o;
            *o = y;
          } else if (EPI == HOLD_EPI_SOFTPLUS) {
            f32x4 r;
    #pragma unroll
            for (int j = 0; j < 4; ++j) r[j] = softplus100(y[j]);
            *o = r;
          } else if (EPI == HOLD_EPI_RELU) {
            f32x4 r;
    #pragma unroll
            for (int j = 0; j < 4; ++j) r[j] = fmaxf(y[j], 0.f);
            *o = r;
          } else if (EPI == HOLD_EPI_SIGMOID) {
            f32x4 r;
    #pragma unroll
            for (int j = 0; j < 4; ++j) r[j] = 1.0f / (1.0f + __expf(-y[j]));
            *o = r;
          } else if (EPI == HOLD_EPI_MUL_DSP) {
            const f32x4 h = *reinterpret_cast<const f32x4*>(d.aux1 + p * (long)d.ldaux1 + n);
            f32x4 r;
    #pragma unroll
            for (int j = 0; j < 4; ++j) r[j] = y[j] * dsp_from_h(h[j]);
            if (d.aux2) r += *reinterpret_cast<const f32x4*>(d.aux2 + p * (long)d.ldaux2 + n);
            *o = r;
          } else if (EPI == HOLD_EPI_MUL_DRELU) {
            const f32x4 h = *reinterpret_cast<const f32x4*>(d.aux1 + p * (long)d.ldaux1 + n);
            f32x4 r;
    #pragma unroll
            for (int j = 0; j < 4; ++j) r[j] = h[j] > 0.f ? y[j] : 0.f;
            *o = r;
          } else if (EPI == HOLD_EPI_DBWD) {
            const f32x4 h = *reinterpret_cast<const f32x4*>(d.aux1 + p * (long)d.ldaux1 + n);
            const f32x4 t = *reinterpret_cast<const f32x4*>(d.aux2 + p * (long)d.ldaux2 + n);
            f32x4 r, r2;
    #pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float e = __expf(-100.0f * h[j]);
              r[j] = y[j] * dsp_from_h(h[j]);
              r2[j] = 100.0f * y[j] * t[j] * e;
            }
            *o = r;
            *reinterpret_cast<f32x4*>(d.out2 + p * (long)d.ldout2 + n) = r2;
          } else if (EPI == HOLD_EPI_MUL_DSIG) {
            const f32x4 sg = *reinterpret_cast<const f32x4*>(d.aux1 + p * (long)d.ldaux1 + n);
            *o = y * sg * (1.0f - sg);
          }
        } else {
    #pragma unroll
          for (int j = 0; j < 4; ++j)
            if (n + j < d.N) epi_scalar(p, n + j, v[j]);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    __syncthreads();  // epilogue staging reads done before the next tile overwrites the LDS
  }  // tile loop
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

// LDS-staged weight gradient: the [rows][n] / [rows][k] operand tiles are brought in with global_load_lds
// (16 B per lane, 1 KiB per wave instruction) instead of one global_load_dword per MFMA operand -- the dword
// version is bound by the CU's texture-address path (64 VMEM instructions per 64 MFMAs per wave).  Fragments are
// column reads of the row-major LDS image (ds_read_b32, consecutive lanes = consecutive banks, conflict-free).
// KT = 32-wide k tiles per wave: 2 -> block tile 128 n x 128 k (32 rows per stage), 4 -> 128 n x 256 k (16 rows).
template <int KT, int X6 = 0, int NBUF = 2>  // X6: 0 fp32 MFMA, 1 bf16 x 6 limb products (rounded limbs), 2 (truncated)
__global__ __launch_bounds__(256, 2) void wgrad_lds_kernel(const float* __restrict__ R, int ldr,
                                                           const float* __restrict__ X, int ldx, int P, int N, int K,
                                                           int splits, float* __restrict__ part,
                                                           float* __restrict__ part_b, int remap, int xcols) {
  // xcols = floats of a row of X that exist from the X pointer on (ldx, or less when X points INTO a row: the K - 256 tail
  // of the rendering net's first layer) -- the bound the staged columns are clamped to
  constexpr int BKW = 64 * KT;              // block width along k
  constexpr int PC = (KT == 4) ? 16 : 32;   // rows (reduction index) per stage
  constexpr int STEPS = PC / 2;
  constexpr int RI = PC * 128 / 256;        // DMA instructions per R stage (256 floats each)
  constexpr int XI = PC * BKW / 256;
  __shared__ __attribute__((aligned(16))) float smem[NBUF * PC * (128 + BKW)];
  float* sR = smem;
  float* sX = smem + NBUF * PC * 128;
  constexpr int NI = RI / 4 + XI / 4;  // DMA instructions per wave and stage
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wk = wave & 1;
  const int hh = lane >> 5, li = lane & 31;
  const int ntn = (N + 127) / 128, ntk = (K + BKW - 1) / BKW;
  // blockIdx -> (tile, split).  The tiles of one split read the same rows of R / X (each its own columns of one operand,
  // ALL columns of the other): workgroup b runs on XCD b % 8 and every XCD has its own L2, so the tiles of a split are
  // numbered 8 apart -- same XCD, dispatched back to back -- and the shared operand comes out of that L2 for all tiles
  // but the first instead of crossing the fabric once per tile (PMC, round 2: 1.5x the algorithmic bytes fetched).
  const int T = ntn * ntk;
  int tile = blockIdx.x % T, split = blockIdx.x / T;
  if (remap) {
    const int full = (splits / 8) * 8 * T;
    if ((int)blockIdx.x < full) {
      const int r = blockIdx.x % (8 * T);
      tile = r / 8;
      split = (blockIdx.x / (8 * T)) * 8 + (r % 8);
    } else {
      const int idx = blockIdx.x - full;
      tile = idx % T;
      split = (splits / 8) * 8 + idx / T;
    }
  }
  const int n0 = (tile / ntk) * 128, k0 = (tile % ntk) * BKW;
  const long chunks = ((long)P + PC - 1) / PC;
  const long cper = (chunks + splits - 1) / splits;
  const long c_begin = (long)split * cper, c_end = min(chunks, c_begin + cper);

  auto stage = [&](long c, int buf) {
    const long p0 = c * PC;
    if (p0 + PC <= P) {
#pragma unroll
      for (int j = 0; j < RI / 4; ++j) {
        const int ins = wave * (RI / 4) + j;
        const int row = ins * 2 + (lane >> 5);
        int col = n0 + (lane & 31) * 4;
        col = col <= ldr - 4 ? col : ldr - 4;  // columns >= N are never stored; keep the address inside the row
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(R + (p0 + row) * (long)ldr + col),
                                         (__attribute__((address_space(3))) void*)(sR + buf * PC * 128 + ins * 256), 16, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < XI / 4; ++j) {
        const int ins = wave * (XI / 4) + j;
        const int row = (BKW == 256) ? ins : ins * 2 + (lane >> 5);
        int col = k0 + ((BKW == 256) ? lane : (lane & 31)) * 4;
        col = col <= xcols - 4 ? col : xcols - 4;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(X + (p0 + row) * (long)ldx + col),
                                         (__attribute__((address_space(3))) void*)(sX + buf * PC * BKW + ins * 256), 16, 0, 0);
      }
    } else {  // ragged tail: rows >= P contribute zeros
      for (int e = tid; e < PC * (128 + BKW) / 4; e += 256) {
        const bool isR = e < PC * 32;
        const int ee = isR ? e : e - PC * 32;
        const int w4 = isR ? 32 : BKW / 4;
        const int row = ee / w4, c4 = (ee % w4) * 4;
        const long p = p0 + row;
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
        if (isR) {
          int col = n0 + c4;
          col = col <= ldr - 4 ? col : ldr - 4;
          if (p < P) z = *reinterpret_cast<const f32x4*>(R + p * (long)ldr + col);
          *reinterpret_cast<f32x4*>(sR + buf * PC * 128 + row * 128 + c4) = z;
        } else {
          int col = k0 + c4;
          col = col <= xcols - 4 ? col : xcols - 4;
          if (p < P) z = *reinterpret_cast<const f32x4*>(X + p * (long)ldx + col);
          *reinterpret_cast<f32x4*>(sX + buf * PC * BKW + row * BKW + c4) = z;
        }
      }
    }
  };
  auto stage_wait = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };

  f32x16 acc[2][KT];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < KT; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  float bsum[2] = {0.f, 0.f};
  const bool do_bias = (part_b != nullptr) && (tile % ntk == 0) && (wk == 0);

  auto stage_wait_keep1 = [&]() {  // everything but the most recently requested (full DMA) stage has landed
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");
    __syncthreads();
  };
  auto full_stage = [&](long c) { return c * PC + PC <= P; };
  if (c_begin < c_end) {
    stage(c_begin, 0);
    if (NBUF == 3 && c_begin + 1 < c_end) {
      stage(c_begin + 1, 1);
      if (full_stage(c_begin + 1)) stage_wait_keep1(); else stage_wait();
    } else {
      stage_wait();
    }
  }
  for (long c = c_begin; c < c_end; ++c) {
    const int buf = (int)((c - c_begin) % NBUF);
    if (c + NBUF - 1 < c_end) stage(c + NBUF - 1, (int)((c - c_begin + NBUF - 1) % NBUF));
    const float* pr = sR + buf * PC * 128 + (hh * STEPS) * 128 + wn * 64 + li;
    const float* px = sX + buf * PC * BKW + (hh * STEPS) * BKW + wk * (32 * KT) + li;
    if constexpr (X6 != 0) {
      const int kbv_raw = (K - (k0 + wk * (32 * KT)) + 31) / 32;
      const int kbv = kbv_raw < 0 ? 0 : (kbv_raw > KT ? KT : kbv_raw);
      // split precision (hold_wgrad_x6): both operands are split into three bf16 limbs as they leave LDS, six limb
      // products per 16 reduction rows on v_mfma_f32_32x32x16_bf16.
      // Lane (hh, li) supplies rows 16 step + 8 hh .. + 7 of column li of its n / k tile to both operands.
      const float* pr6 = sR + buf * PC * 128 + (hh * 8) * 128 + wn * 64 + li;
      const float* px6 = sX + buf * PC * BKW + (hh * 8) * BKW + wk * (32 * KT) + li;
#pragma unroll
      for (int st = 0; st < PC / 16; ++st) {
        Limbs3 la[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          float x[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) x[e] = pr6[(st * 16 + e) * 128 + a * 32];
          bsum[a] += ((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]));
          la[a] = (X6 == 2) ? split8s_trunc(x) : split8s(x);
        }
        // k tiles two at a time: 4 accumulators in rotation, 24 + 24 limb registers live; tiles entirely beyond K
        // (wave-uniform) are skipped -- the K = 40 embedding layers would otherwise run 3x their useful MFMAs
#pragma unroll
        for (int bp = 0; bp < KT; bp += 2) {
          if (bp >= kbv) break;
          Limbs3 lb[2];
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            float x[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = px6[(st * 16 + e) * BKW + (bp + b) * 32];
            lb[b] = (X6 == 2) ? split8s_trunc(x) : split8s(x);
          }
#pragma unroll
          for (int pr = 0; pr < 6; ++pr) {
            const int il = (pr == 2 || pr == 3) ? 1 : (pr == 5 ? 2 : 0);   // (R limb, X limb): 00 01 10 11 02 20
            const int jl = (pr == 1 || pr == 3) ? 1 : (pr == 4 ? 2 : 0);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
              for (int b = 0; b < 2; ++b)
                acc[a][bp + b] =
                    __builtin_amdgcn_mfma_f32_32x32x16_bf16(la[a].l[il], lb[b].l[jl], acc[a][bp + b], 0, 0, 0);
          }
        }
      }
    } else {
      // fragments one step ahead, pinned: left alone the scheduler issues them ~2 MFMAs before their use
      float avn[2], bvn[KT];
#pragma unroll
      for (int a = 0; a < 2; ++a) avn[a] = pr[a * 32];
#pragma unroll
      for (int b = 0; b < KT; ++b) bvn[b] = px[b * 32];
#pragma unroll
      for (int st = 0; st < STEPS; ++st) {
        float av[2], bv[KT];
#pragma unroll
        for (int a = 0; a < 2; ++a) av[a] = avn[a];
#pragma unroll
        for (int b = 0; b < KT; ++b) bv[b] = bvn[b];
        if (st + 1 < STEPS) {
#pragma unroll
          for (int a = 0; a < 2; ++a) avn[a] = pr[(st + 1) * 128 + a * 32];
#pragma unroll
          for (int b = 0; b < KT; ++b) bvn[b] = px[(st + 1) * BKW + b * 32];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < KT; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], bv[b], acc[a][b], 0, 0, 0);
        bsum[0] += av[0];
        bsum[1] += av[1];
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (NBUF == 3 && c + 2 < c_end && full_stage(c + 2)) stage_wait_keep1(); else stage_wait();
  }
  float* out = part + (long)split * N * K;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < KT; ++b) {
      const int k = k0 + wk * (32 * KT) + b * 32 + li;
      if (k >= K) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wn * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        if (n < N) out[(long)n * K + k] = acc[a][b][r];
      }
    }
  if (do_bias) {
    const float s0 = bsum[0] + __shfl_xor(bsum[0], 32);
    const float s1 = bsum[1] + __shfl_xor(bsum[1], 32);
    const int na = n0 + wn * 64 + li, nb = na + 32;
    if (hh == 0) {
      if (na < N) part_b[(long)split * N + na] = s0;
      if (nb < N) part_b[(long)split * N + nb] = s1;
    }
  }
}

__global__ void wgrad_reduce_kernel(const float* __restrict__ part, int splits, long NK, int K, float* __restrict__ dW,
                                    int lddw, int accumulate, long pstride) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= NK) return;
  float s = 0.f;
  for (int sp = 0; sp < splits; ++sp) s += part[(long)sp * pstride + i];
  const long n = i / K, k = i % K;
  float* o = dW + n * lddw + k;
  *o = accumulate ? *o + s : s;
}

// The same reduction for K % 4 == 0 with 16-byte accesses and the partial sums spread over 16 thread groups: a block
// owns 64 consecutive elements (16 lanes x float4 = one 256-byte segment per partial tile), thread group g sums the
// tiles g, g + 16, ... (independent loads in flight instead of one dependent chain of `splits` loads per thread),
// then the 16 group sums are added in a fixed order -- deterministic, ~5x faster than the scalar version at 256 splits.
// pstride = floats between the partial tiles (>= NK: the whole-dW kernel writes 256-row tiles of which the first N count)
__global__ __launch_bounds__(256) void wgrad_reduce4_kernel(const float* __restrict__ part, int splits, long NK, int K,
                                                            float* __restrict__ dW, int lddw, int accumulate, long pstride) {
  __shared__ f32x4 red[16][16];
  const int e = threadIdx.x & 15, g = threadIdx.x >> 4;
  const long i = ((long)blockIdx.x * 16 + e) * 4;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (i < NK)
    for (int sp = g; sp < splits; sp += 16) s += *reinterpret_cast<const f32x4*>(part + (long)sp * pstride + i);
  red[g][e] = s;
  __syncthreads();
  if (g != 0 || i >= NK) return;
#pragma unroll
  for (int j = 1; j < 16; ++j) s += red[j][e];
  const long n = i / K, k = i % K;
  float* o = dW + n * lddw + k;
  if (accumulate) s += *reinterpret_cast<const f32x4*>(o);
  *reinterpret_cast<f32x4*>(o) = s;
}

// weighted column sums, first pass: part[block][n] = sum over the block's rows of w[p] * X[p][n].  A wave reads one
// 1 KiB row segment per instruction (64 lanes x 16 B); the four waves of a block take rows r, r+1, r+2, r+3 (mod 4).
__global__ __launch_bounds__(256) void wcolsum_kernel(const float* __restrict__ X, int ldx, int N, long P,
                                                     const float* __restrict__ w, long rows_per_block,
                                                     float* __restrict__ part) {
  __shared__ f32x4 red[4][64];
  const int cg = threadIdx.x & 63, ph = threadIdx.x >> 6;
  const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(P, r0 + rows_per_block);
  for (int c0 = 0; c0 < N; c0 += 256) {
    const int c = c0 + cg * 4;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    if (c < N) {
      long r = r0 + ph;
      for (; r + 4 < r1; r += 8) {
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(X + r * ldx + c);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(X + (r + 4) * ldx + c);
        acc0 += v0 * (w ? w[r] : 1.0f);
        acc1 += v1 * (w ? w[r + 4] : 1.0f);
      }
      if (r < r1) acc0 += *reinterpret_cast<const f32x4*>(X + r * ldx + c) * (w ? w[r] : 1.0f);
    }
    red[ph][cg] = acc0 + acc1;
    __syncthreads();
    if (ph == 0 && c < N)
      *reinterpret_cast<f32x4*>(part + (long)blockIdx.x * N + c) = (red[0][cg] + red[1][cg]) + (red[2][cg] + red[3][cg]);
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// The 3-output colour head (last Linear + sigmoid of the rendering nets, code/src/networks/texture_net.py:95-101):
// HBM-streaming kernels instead of GEMM tiles that would be 3 / 128 full.
// ---------------------------------------------------------------------------------------------
// out[p][c] = act(sum_k A[p][k] W[c][k] + b[c]), c < 3.  16 lanes per row (4 rows per wave): lane l reads the 16-byte
// chunks l, l + 16, ... of the row, so 16 lanes cover 256 contiguous bytes per load.
__global__ __launch_bounds__(256) void head3_fwd_kernel(const float* __restrict__ A, int lda, const float* __restrict__ W,
                                                       int ldw, const float* __restrict__ bias, int K, long P,
                                                       float* __restrict__ out, int ldo, int sigmoid) {
  const int l = threadIdx.x & 15;
  const long p = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f;
  if (p < P) {
    for (int k = l * 4; k < K; k += 64) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(A + p * lda + k);
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(W + k);
      const f32x4 w1 = *reinterpret_cast<const f32x4*>(W + ldw + k);
      const f32x4 w2 = *reinterpret_cast<const f32x4*>(W + 2 * ldw + k);
      a0 += (a[0] * w0[0] + a[1] * w0[1]) + (a[2] * w0[2] + a[3] * w0[3]);
      a1 += (a[0] * w1[0] + a[1] * w1[1]) + (a[2] * w1[2] + a[3] * w1[3]);
      a2 += (a[0] * w2[0] + a[1] * w2[1]) + (a[2] * w2[2] + a[3] * w2[3]);
    }
  }
#pragma unroll
  for (int d = 8; d > 0; d >>= 1) {
    a0 += __shfl_xor(a0, d);
    a1 += __shfl_xor(a1, d);
    a2 += __shfl_xor(a2, d);
  }
  if (p < P && l < 3) {
    float y = (l == 0 ? a0 : (l == 1 ? a1 : a2)) + (bias ? bias[l] : 0.f);
    if (sigmoid) y = 1.0f / (1.0f + __expf(-y));
    out[p * ldo + l] = y;
  }
}

// backward of the head given dy[p][c] = cotangent of the pre-activation (3 columns):
//   rr[p][k]  = (R[p][k] > 0) ? sum_c dy[p][c] W[c][k] : 0      (input gradient through the ReLU that produced R)
//   part[blk][c][k] = sum over the block's rows of dy[p][c] R[p][k];  part_b[blk][c] = sum dy[p][c]
// one pass over R; the partials are reduced deterministically by wgrad_reduce4_kernel.
__global__ __launch_bounds__(256) void head3_bwd_kernel(const float* __restrict__ dy, int ldy, const float* __restrict__ R,
                                                       int ldr, const float* __restrict__ W, int ldw, int K, long P,
                                                       long rows_per_block, float* __restrict__ rr, int ldrr,
                                                       float* __restrict__ part, float* __restrict__ part_b) {
  __shared__ f32x4 red[4][3][64];
  __shared__ float redb[4][4];
  const int cg = threadIdx.x & 63, ph = threadIdx.x >> 6;
  const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(P, r0 + rows_per_block);
  for (int c0 = 0; c0 < K; c0 += 256) {
    const int c = c0 + cg * 4;
    const bool on = c < K;
    f32x4 w0 = {0.f, 0.f, 0.f, 0.f}, w1 = w0, w2 = w0, acc0 = w0, acc1 = w0, acc2 = w0;
    float sb0 = 0.f, sb1 = 0.f, sb2 = 0.f;
    if (on) {
      w0 = *reinterpret_cast<const f32x4*>(W + c);
      w1 = *reinterpret_cast<const f32x4*>(W + ldw + c);
      w2 = *reinterpret_cast<const f32x4*>(W + 2 * ldw + c);
    }
    for (long r = r0 + ph; r < r1; r += 4) {
      const float d0 = dy[r * ldy], d1 = dy[r * ldy + 1], d2 = dy[r * ldy + 2];
      sb0 += d0;
      sb1 += d1;
      sb2 += d2;
      if (on) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(R + r * ldr + c);
        acc0 += x * d0;
        acc1 += x * d1;
        acc2 += x * d2;
        f32x4 g = w0 * d0 + w1 * d1 + w2 * d2;
#pragma unroll
        for (int j = 0; j < 4; ++j) g[j] = x[j] > 0.f ? g[j] : 0.f;
        *reinterpret_cast<f32x4*>(rr + r * ldrr + c) = g;
      }
    }
    red[ph][0][cg] = acc0;
    red[ph][1][cg] = acc1;
    red[ph][2][cg] = acc2;
    if (c0 == 0 && cg == 0) {
      redb[ph][0] = sb0;
      redb[ph][1] = sb1;
      redb[ph][2] = sb2;
      redb[ph][3] = 0.f;
    }
    __syncthreads();
    if (ph < 3 && on)
      *reinterpret_cast<f32x4*>(part + ((long)blockIdx.x * 3 + ph) * K + c) =
          (red[0][ph][cg] + red[1][ph][cg]) + (red[2][ph][cg] + red[3][ph][cg]);
    if (c0 == 0 && threadIdx.x < 4)
      part_b[(long)blockIdx.x * 4 + threadIdx.x] =
          (redb[0][threadIdx.x] + redb[1][threadIdx.x]) + (redb[2][threadIdx.x] + redb[3][threadIdx.x]);
    __syncthreads();
  }
}

}  // namespace

extern "C" int hold_abi_version(void) { return 1; }

constexpr int WCOLSUM_BLOCKS = 2048;
extern "C" int64_t hold_wcolsum_workspace_floats(int32_t N) { return (int64_t)WCOLSUM_BLOCKS * N; }

extern "C" int hold_wcolsum(const float* X, int32_t ldx, int32_t N, int64_t P, const float* w, float* out,
                            int32_t accumulate, float* workspace, hold_stream_t stream) {
  if (!X || !out || !workspace || N <= 0 || N > 1024 || (N & 3) || (ldx & 3) || P < 0) return HOLD_E_ARG;
  if (((uintptr_t)X & 15) || ((uintptr_t)out & 15) || ((uintptr_t)workspace & 15)) return HOLD_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (P == 0) {
    if (!accumulate && hipMemsetAsync(out, 0, sizeof(float) * N, s) != hipSuccess) return HOLD_E_LAUNCH;
    return HOLD_OK;
  }
  long blocks = (P + 15) / 16;
  if (blocks > WCOLSUM_BLOCKS) blocks = WCOLSUM_BLOCKS;
  const long rpb = (P + blocks - 1) / blocks;
  blocks = (P + rpb - 1) / rpb;
  hipLaunchKernelGGL(wcolsum_kernel, dim3((unsigned)blocks), dim3(256), 0, s, X, ldx, N, (long)P, w, rpb, workspace);
  hipLaunchKernelGGL(wgrad_reduce4_kernel, dim3((unsigned)((N / 4 + 15) / 16)), dim3(256), 0, s, workspace, (int)blocks,
                     (long)N, N, out, N, accumulate, (long)N);
  return hipGetLastError() == hipSuccess ? HOLD_OK : HOLD_E_LAUNCH;
}

template <int NT, int X6 = 0, int NBUF = 2>
static int launch_gemm(const hold_gemm_desc& d, hipStream_t s) {
  constexpr int BNc = 64 * NT, BKc = (NT == 2) ? 32 : 16;
  const long mt = ((long)d.P + BM - 1) / BM;
  const int nt = (d.N + BNc - 1) / BNc;
  const long tiles_l = mt * nt;
  if (tiles_l > 0x7fffffffL) return HOLD_E_ARG;
  const int tiles = (int)tiles_l;
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return HOLD_E_LAUNCH;
    n_cu = prop.multiProcessorCount;
  }
  const int resident = 2 * n_cu;  // two 69.6 KiB blocks per CU
  dim3 grid((unsigned)(tiles < resident ? tiles : resident)), block(256);
  // stagger the second half of a full persistent grid by ~half a tile (s_sleep(127) = 8128 cycles each)
  const int nkk = (d.K + BKc - 1) / BKc;
  int stagger = (tiles >= 2 * resident) ? (nkk * BKc >= 192 ? 2 : 1) : 0;
#ifdef HOLD_DEV  // developer build only (hold_amd/build.py, HOLD_DEV=1): timing ablations that skip parts of the kernel
  if (const char* dbg = getenv("HOLD_GEMM_DEBUG")) {
    static bool warned = false;
    if (!warned) fprintf(stderr, "libholdhip: HOLD_GEMM_DEBUG=%s -- timing ablation, hold_gemm_nt results are WRONG\n", dbg);
    warned = true;
    stagger |= atoi(dbg);
  }
#endif
  switch (d.epilogue) {
#define HOLD_CASE(E) \
  case E: hipLaunchKernelGGL((gemm_nt_kernel<E, NT, X6, NBUF>), grid, block, 0, s, d, tiles, stagger); break;
    HOLD_CASE(HOLD_EPI_NONE)
    HOLD_CASE(HOLD_EPI_SOFTPLUS)
    HOLD_CASE(HOLD_EPI_RELU)
    HOLD_CASE(HOLD_EPI_SIGMOID)
    HOLD_CASE(HOLD_EPI_MUL_DSP)
    HOLD_CASE(HOLD_EPI_MUL_DRELU)
    HOLD_CASE(HOLD_EPI_DBWD)
    HOLD_CASE(HOLD_EPI_MUL_DSIG)
#undef HOLD_CASE
    default: return HOLD_E_ARG;
  }
  return hipGetLastError() == hipSuccess ? HOLD_OK : HOLD_E_LAUNCH;
}

static int gemm_nt_impl(const hold_gemm_desc* dp, hold_stream_t stream, int x6) {
  if (!dp) return HOLD_E_ARG;
  hold_gemm_desc d = *dp;
  if (!d.A || !d.W || !d.C || d.P < 0 || d.N <= 0 || d.K <= 0) return HOLD_E_ARG;
  if ((d.lda & 3) || (d.ldw & 3) || (d.K & 3)) return HOLD_E_ARG;
  if (((uintptr_t)d.A & 15) || ((uintptr_t)d.W & 15)) return HOLD_E_ARG;
  if (d.n_split <= 0 || d.n_split > d.N) d.n_split = d.N;
  if (d.n_split < d.N && !d.C2) return HOLD_E_ARG;
  if ((d.epilogue == HOLD_EPI_MUL_DSP || d.epilogue == HOLD_EPI_MUL_DRELU || d.epilogue == HOLD_EPI_MUL_DSIG) && !d.aux1)
    return HOLD_E_ARG;
  if (d.epilogue == HOLD_EPI_DBWD && (!d.aux1 || !d.aux2 || !d.out2)) return HOLD_E_ARG;
  if ((d.r1_row == nullptr) != (d.r1_col == nullptr)) return HOLD_E_ARG;
  if (d.P == 0) return HOLD_OK;
  hipStream_t s = (hipStream_t)stream;
  int wide = d.N > 128;
#ifdef HOLD_DEV
  if (const char* w = getenv("HOLD_GEMM_TILE")) wide = atoi(w) == 256;
#endif
  if (x6) {
#ifdef HOLD_DEV  // three operand stages: measured equal to two (134 vs 133 TF-eq), kept as a developer-build A/B only
    if (const char* nb = getenv("HOLD_GEMM_NBUF"))
      if (wide && atoi(nb) == 3) return launch_gemm<4, 2, 3>(d, s);
#endif
    return wide ? launch_gemm<4, 2>(d, s) : launch_gemm<2, 2>(d, s);
  }
  return wide ? launch_gemm<4>(d, s) : launch_gemm<2>(d, s);
}

extern "C" int hold_gemm_nt(const hold_gemm_desc* dp, hold_stream_t stream) { return gemm_nt_impl(dp, stream, 0); }
extern "C" int hold_gemm_nt_x6(const hold_gemm_desc* dp, hold_stream_t stream) { return gemm_nt_impl(dp, stream, 1); }

extern "C" int64_t hold_wgrad_workspace_floats(int32_t N, int32_t K, int32_t splits) {
  // the whole-dW kernel (K in 256..320, N in 129..256) writes partial tiles of 256 rows whatever N is
  const int64_t Np = (N > 128 && N < 256 && K >= 256 && K <= 320) ? 256 : N;
  return (int64_t)splits * (Np * (int64_t)K + Np);
}

// register-resident 256 x 256 variant (csrc/wgrad_r6.hip): fills <= max_splits partial tiles, returns their number
int hold_wgrad_r6_partials(const float* R, int ldr, const float* X, int ldx, long P, int max_splits, float* part,
                           float* part_b, hipStream_t s);
int hold_wgrad_h3_partials(const float* R, int ldr, const float* X, int ldx, long P, int n_valid, int max_splits, float* part,
                           float* part_b, hipStream_t s);

// mode 0: fp32 MFMA; 1 / 2: split precision (3 bf16 limbs x 6 products, fp32 accumulate) with the round-to-nearest /
// truncating limb split (both decompose the 24-bit significand exactly); 3: as 2, with the whole-dW shapes in the two-limb
// fp16 arithmetic (wgrad_h3_kernel, csrc/wgrad_r6.hip)
static int wgrad_impl(const float* R, int32_t ldr, const float* X, int32_t ldx, int32_t P, int32_t N, int32_t K,
                      float* dW, int32_t lddw, float* db, int32_t accumulate, int32_t splits, float* workspace,
                      hold_stream_t stream, int mode, int xcols = -1) {
  if (!R || !X || !dW || !workspace || N <= 0 || K <= 0 || P < 0 || splits <= 0) return HOLD_E_ARG;
  if (xcols < 0) xcols = ldx;  // columns of a row of X that exist from the X pointer on
  const bool h3 = mode == 3;
  if (h3) mode = 2;  // every shape outside the whole-dW domain: the bf16 three-limb kernels
  hipStream_t s = (hipStream_t)stream;
  const long chunks = ((long)P + 31) / 32;
  if (splits > chunks) splits = (int)(chunks > 0 ? chunks : 1);
  float* part = workspace;
  float* part_b = db ? workspace + (long)splits * N * K : nullptr;
  bool lds_ok = !(ldr & 3) && !(ldx & 3) && !((uintptr_t)R & 15) && !((uintptr_t)X & 15) && ldr >= 4 && ldx >= 4 && xcols >= 4;
#ifdef HOLD_DEV
  if (getenv("HOLD_WGRAD_DIRECT")) lds_ok = false;
#endif
  int remap = 1;
#ifdef HOLD_DEV
  if (const char* w = getenv("HOLD_WGRAD_REMAP")) remap = atoi(w);
#endif
  bool x6_wide = K > 128;  // 128 n x 256 k tiles: 6 fragment splits per 48 MFMAs instead of 4 per 24
#ifdef HOLD_DEV
  if (const char* w = getenv("HOLD_WGRAD_X6_TILE")) x6_wide = atoi(w) == 256;
#endif
  // whole-dW workgroups (csrc/wgrad_r6.hip), one wave per SIMD: each operand row read once, 3 VALU of limb split per MFMA.
  // Taken for 256 columns of X and N in 129..256 columns of R that is at least 256 wide in memory (the kernel reads 256
  // columns of both; rows >= N of its partial tiles are never reduced); K in 257..320: the first 256 columns of X this way,
  // the rest by the tile kernel below through a second call (R is read twice: 1 KiB per point more than the one-kernel form).
  bool r6 = lds_ok && mode == 2 && N > 128 && N <= 256 && ldr >= 256 && K >= 256 && K <= 320 && ldx >= K &&
            (P % 16) == 0 && P >= 4096;
#ifdef HOLD_DEV
  if (const char* w = getenv("HOLD_WGRAD_R6")) r6 = r6 && atoi(w) != 0;
#endif
  long pstride = (long)N * K;  // floats between partial tiles
  int Kred = K;                // columns the reduction below covers
  if (r6) {
    const int g = h3 ? hold_wgrad_h3_partials(R, ldr, X, ldx, (long)P, N, splits, part, db ? part + (long)splits * 65536 : nullptr, s)
                     : hold_wgrad_r6_partials(R, ldr, X, ldx, (long)P, splits, part, db ? part + (long)splits * 65536 : nullptr, s);
    if (g < 0) return g;
    if (K > 256) {  // the remaining K - 256 columns of dW: tile kernel on (R, X + 256), its partials behind the first part's
      float* ws2 = part + (long)splits * (65536 + 256);
      const int rc = wgrad_impl(R, ldr, X + 256, ldx, P, N, K - 256, dW + 256, lddw, nullptr, accumulate, splits, ws2, stream,
                                h3 ? 3 : mode, ldx - 256);
      if (rc != HOLD_OK) return rc;
    }
    part_b = db ? part + (long)splits * 65536 : nullptr;
    splits = g;
    pstride = 65536;
    Kred = 256;
  } else if (lds_ok && mode == 2 && x6_wide) {
    const int tiles = ((N + 127) / 128) * ((K + 255) / 256);
    const long ch = ((long)P + 15) / 16;
    if (splits > ch) splits = (int)(ch > 0 ? ch : 1);
    bool deep = false;
#ifdef HOLD_DEV  // three operand stages: measured slower than two (150 vs 154 TF-eq), developer-build A/B only
    if (const char* nb = getenv("HOLD_WGRAD_NBUF")) deep = atoi(nb) == 3;
    if (deep)
      hipLaunchKernelGGL((wgrad_lds_kernel<4, 2, 3>), dim3(tiles * splits), dim3(256), 0, s, R, ldr, X, ldx, P, N, K,
                         splits, part, part_b, remap, xcols);
#endif
    if (!deep)
      hipLaunchKernelGGL((wgrad_lds_kernel<4, 2>), dim3(tiles * splits), dim3(256), 0, s, R, ldr, X, ldx, P, N, K, splits,
                         part, part_b, remap, xcols);
  } else if (lds_ok && mode != 0) {  // split-precision path (128 x 128 tiles)
    const int tiles = ((N + 127) / 128) * ((K + 127) / 128);
    if (mode == 2)
      hipLaunchKernelGGL((wgrad_lds_kernel<2, 2>), dim3(tiles * splits), dim3(256), 0, s, R, ldr, X, ldx, P, N, K, splits,
                         part, part_b, remap, xcols);
    else
      hipLaunchKernelGGL((wgrad_lds_kernel<2, 1>), dim3(tiles * splits), dim3(256), 0, s, R, ldr, X, ldx, P, N, K, splits,
                         part, part_b, remap, xcols);
  } else if (lds_ok && K > 128) {
    const int tiles = ((N + 127) / 128) * ((K + 255) / 256);
    const long ch = ((long)P + 15) / 16;
    if (splits > ch) splits = (int)(ch > 0 ? ch : 1);
    hipLaunchKernelGGL((wgrad_lds_kernel<4>), dim3(tiles * splits), dim3(256), 0, s, R, ldr, X, ldx, P, N, K, splits,
                       part, part_b, remap, xcols);
  } else if (lds_ok) {
    const int tiles = ((N + 127) / 128) * ((K + 127) / 128);
    hipLaunchKernelGGL((wgrad_lds_kernel<2>), dim3(tiles * splits), dim3(256), 0, s, R, ldr, X, ldx, P, N, K, splits,
                       part, part_b, remap, xcols);
  } else {
    const int tiles = ((N + 127) / 128) * ((K + 127) / 128);
    hipLaunchKernelGGL(wgrad_kernel, dim3(tiles * splits), dim3(256), 0, s, R, ldr, X, ldx, P, N, K, splits, part,
                       part_b);
  }
  const long NK = (long)N * Kred;
  if (!(Kred & 3) && !(lddw & 3) && !((uintptr_t)dW & 15) && !((uintptr_t)part & 15))
    hipLaunchKernelGGL(wgrad_reduce4_kernel, dim3((unsigned)((NK / 4 + 15) / 16)), dim3(256), 0, s, part, splits, NK, Kred,
                       dW, lddw, accumulate, pstride);
  else
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((NK + 255) / 256)), dim3(256), 0, s, part, splits, NK, Kred, dW,
                       lddw, accumulate, pstride);
  if (db) {
    const long bstride = r6 ? 256 : N;
    if (!(N & 3) && !((uintptr_t)db & 15) && !((uintptr_t)part_b & 15))
      hipLaunchKernelGGL(wgrad_reduce4_kernel, dim3((unsigned)((N / 4 + 15) / 16)), dim3(256), 0, s, part_b, splits,
                         (long)N, N, db, N, accumulate, bstride);
    else
      hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, part_b, splits,
                         (long)N, N, db, N, accumulate, bstride);
  }
  return hipGetLastError() == hipSuccess ? HOLD_OK : HOLD_E_LAUNCH;
}

extern "C" int hold_wgrad(const float* R, int32_t ldr, const float* X, int32_t ldx, int32_t P, int32_t N, int32_t K,
                          float* dW, int32_t lddw, float* db, int32_t accumulate, int32_t splits, float* workspace,
                          hold_stream_t stream) {
  return wgrad_impl(R, ldr, X, ldx, P, N, K, dW, lddw, db, accumulate, splits, workspace, stream, 0);
}

extern "C" int hold_wgrad_x6(const float* R, int32_t ldr, const float* X, int32_t ldx, int32_t P, int32_t N, int32_t K,
                             float* dW, int32_t lddw, float* db, int32_t accumulate, int32_t splits, float* workspace,
                             hold_stream_t stream) {
  int mode = 2;
#ifdef HOLD_DEV
  if (const char* sp = getenv("HOLD_X6_SPLIT")) mode = sp[0] == 'r' ? 1 : 2;
#endif
  return wgrad_impl(R, ldr, X, ldx, P, N, K, dW, lddw, db, accumulate, splits, workspace, stream, mode);
}

// hold_wgrad_x6 with the whole-dW shapes (N in 129..256, K in 256..320, P a multiple of 16 >= 4 096) in the two-limb fp16
// arithmetic: three v_mfma_f32_32x32x16_f16 per product, per-workgroup power-of-two operand scales from a sample of the
// workgroup's rows (csrc/wgrad_r6.hip: wgrad_h3_body); every other shape exactly as hold_wgrad_x6
extern "C" int hold_wgrad_h3(const float* R, int32_t ldr, const float* X, int32_t ldx, int32_t P, int32_t N, int32_t K,
                             float* dW, int32_t lddw, float* db, int32_t accumulate, int32_t splits, float* workspace,
                             hold_stream_t stream) {
  return wgrad_impl(R, ldr, X, ldx, P, N, K, dW, lddw, db, accumulate, splits, workspace, stream, 3);
}

constexpr int HEAD3_BLOCKS = 2048;
extern "C" int64_t hold_head3_workspace_floats(int32_t K) { return (int64_t)HEAD3_BLOCKS * (3 * (int64_t)K + 4); }

extern "C" int hold_head3_fwd(const float* A, int32_t lda, const float* W, int32_t ldw, const float* bias, int32_t K,
                              int64_t P, float* out, int32_t ldo, int32_t sigmoid, hold_stream_t stream) {
  if (!A || !W || !out || K <= 0 || (K & 3) || (lda & 3) || (ldw & 3) || ldo < 3 || P < 0) return HOLD_E_ARG;
  if (((uintptr_t)A & 15) || ((uintptr_t)W & 15)) return HOLD_E_ARG;
  if (P == 0) return HOLD_OK;
  hipLaunchKernelGGL(head3_fwd_kernel, dim3((unsigned)((P + 15) / 16)), dim3(256), 0, (hipStream_t)stream, A, lda, W, ldw,
                     bias, K, (long)P, out, ldo, sigmoid);
  return hipGetLastError() == hipSuccess ? HOLD_OK : HOLD_E_LAUNCH;
}

extern "C" int hold_head3_bwd(const float* dy, int32_t ldy, const float* R, int32_t ldr, const float* W, int32_t ldw,
                              int32_t K, int64_t P, float* rr, int32_t ldrr, float* dW, int32_t lddw, float* db4,
                              int32_t accumulate, float* workspace, hold_stream_t stream) {
  if (!dy || !R || !W || !rr || !dW || !db4 || !workspace || K <= 0 || (K & 3) || ldy < 3 || P < 0) return HOLD_E_ARG;
  if ((ldr & 3) || (ldw & 3) || (ldrr & 3) || (lddw & 3)) return HOLD_E_ARG;
  if (((uintptr_t)R & 15) || ((uintptr_t)W & 15) || ((uintptr_t)rr & 15) || ((uintptr_t)dW & 15) || ((uintptr_t)db4 & 15) ||
      ((uintptr_t)workspace & 15))
    return HOLD_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (P == 0) {
    if (!accumulate) {
      for (int c = 0; c < 3; ++c)
        if (hipMemsetAsync(dW + (long)c * lddw, 0, sizeof(float) * K, s) != hipSuccess) return HOLD_E_LAUNCH;
      if (hipMemsetAsync(db4, 0, sizeof(float) * 4, s) != hipSuccess) return HOLD_E_LAUNCH;
    }
    return HOLD_OK;
  }
  long blocks = (P + 15) / 16;
  if (blocks > HEAD3_BLOCKS) blocks = HEAD3_BLOCKS;
  const long rpb = (P + blocks - 1) / blocks;
  blocks = (P + rpb - 1) / rpb;
  float* part = workspace;
  float* part_b = workspace + (long)HEAD3_BLOCKS * 3 * K;
  hipLaunchKernelGGL(head3_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, s, dy, ldy, R, ldr, W, ldw, K, (long)P, rpb,
                     rr, ldrr, part, part_b);
  const long NK = 3L * K;
  hipLaunchKernelGGL(wgrad_reduce4_kernel, dim3((unsigned)((NK / 4 + 15) / 16)), dim3(256), 0, s, part, (int)blocks, NK, K,
                     dW, lddw, accumulate, NK);
  hipLaunchKernelGGL(wgrad_reduce4_kernel, dim3(1), dim3(256), 0, s, part_b, (int)blocks, 4L, 4, db4, 4, accumulate, 4L);
  return hipGetLastError() == hipSuccess ? HOLD_OK : HOLD_E_LAUNCH;
}
