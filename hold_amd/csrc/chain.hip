// LDS-resident layer CHAINS of the ImplicitNet for the training path (gfx950): up to 8 consecutive 256-wide layers of one
// sweep in ONE launch, the running activation never leaving LDS; only the per-layer side inputs (stored h_l / t_l / a2_l)
// are read from and the per-layer results (needed later by wgrad and by the other sweeps) written to HBM, straight
// from / to the accumulator registers.  Same software pipeline as fused_sdf_pipe_kernel (fused_sdf.hip): the 128 points
// of a workgroup are two halves half a layer apart, the epilogue of one half runs under the MFMAs of the other.
//
// Sweeps (hold_amd/field.py; reference: ImplicitNet.forward code/src/networks/shape_net.py:84-130 and what
// torch.autograd derives from it for volsdf_utils.py:51-105 with create_graph=True):
//   HOLD_CHAIN_SOFTPLUS  forward trunk          h_l   = softplus(W_l in_l + b_l)                    stores h_l
//   HOLD_CHAIN_DSP       descending sweeps      v_l-1 = (W_l^T v_l) * sp'(h_l-1) [+ a2_l-1]         stores v_l-1
//                        (d sdf / d a_l of the normal path, and the first-order backward)
//   HOLD_CHAIN_DBWD      ascending 2nd-order    tb = W_l vb_l ; ub_l = tb * sp'(h_l) ; a2_l = 100 tb t_l (1 - sp'(h_l))
//                                                                                                    stores ub_l, a2_l
// Skip layer (the 217 | 39 split of layer 3 -> 4): SOFTPLUS / DBWD take columns 217.. of the next input from the 40-wide
// side matrix (embedding / its cotangent); DSP leaves the raw products (d / d embedding) in columns 217.. of its output.
//
// Roofline: the MFMA pipe -- fp32 (v_mfma_f32_32x32x2_f32, 157 TFLOP/s) for hold_chain, bf16 (six limb products on
// v_mfma_f32_32x32x16_bf16 per algorithmic product, 2.5 PFLOP/s) for hold_chain_x6; algorithmic HBM bytes per point and
// layer = 1 KiB per side input + 1 KiB per stored result (SOFTPLUS 1, DSP 2, DSP + a2 3, DBWD 4 KiB),
// against 1 KiB more for the layer-by-layer GEMM.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/hold_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int ASTR = 260, ESTR = 40, SKIP_OUT = 217;

// same arithmetic as the GEMM epilogues (gemm.hip) so both routes agree to rounding
__device__ __forceinline__ float softplus100(float y) {  // branch-free: both log1p forms are a few VALU ops
  const float z = y * 100.0f;
  const float e = __expf(-fabsf(z));
  const float l_log = __logf(1.0f + e);
  const float l_ser = e * (1.0f - e * (0.5f - 0.33333334f * e));
  const float l = (e > 1e-3f) ? l_log : l_ser;
  const float r = (fmaxf(z, 0.f) + l) * 0.01f;
  return (z > 20.0f) ? y : r;
}
__device__ __forceinline__ float dsp_from_h(float h) {
  const float x = 100.0f * h;
  const float ser = x * (1.0f - x * (0.5f - x * (0.16666667f - 0.041666668f * x)));
  const float ex = 1.0f - __expf(-x);
  return (x < 0.05f) ? ser : ex;
}

// Accumulator layout: D[i = point][j = feature] -- lane (hh, li) of wave w holds FEATURE 32 w + li of the points
// 8 g + 4 hh + k (register 4 g + k) of each 32-point tile, so that for a fixed register the 32 lanes of a half-wave
// address 128 contiguous bytes of one row of the [P][256] matrices: every global side-input load / result store is
// two fully used cache lines per instruction.
//
// Everything one epilogue needs, for the lane's column and the 2 x 16 points of one half.  Global matrices go through
// buffer descriptors (num_records = P * ld * 4 bytes; hold_chain rejects >= 2^32): rows >= P are out of range, so the
// hardware drops those loads (-> 0) and stores and the tail block needs no predication; a NULL matrix gets 0 records.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
struct EpiCtx {
  float bias;         // bias[col]
  rsrc_t a1, a2, o1, o2;
  uint32_t goff;      // byte offset of (point half*64 + 4 hh, column col)
  uint32_t rowb;      // ld * 4
  float* lcol;        // LDS: act + (half*64 + 4 hh) * ASTR + col
  const float* scol;  // LDS: side + (half*64 + 4 hh) * ESTR + (col - 217)   (only read on the skip layer)
  bool skip;          // skip layer (uniform)
  bool special;       // skip layer and col >= 217
  bool wr_lds;
};

__device__ __forceinline__ rsrc_t make_rsrc(const float* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, p ? bytes : 0u, 0x00020000);
}
// per-lane byte offset in voffset, the wave-uniform row offset in soffset (an SGPR: no per-access address VGPRs); the
// raw-buffer range check is voffset >= num_records - soffset, i.e. it covers the sum
__device__ __forceinline__ float ldb(rsrc_t r, uint32_t voff, uint32_t soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ void stb(rsrc_t r, uint32_t voff, uint32_t soff, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, 0);
}

template <int MODE, bool A2>
__device__ __forceinline__ void epi_load(const EpiCtx& x, int u, f32x4& v1, f32x4& v2) {
  const int m = u >> 2, g = u & 3;
  v1 = f32x4{0.f, 0.f, 0.f, 0.f};
  v2 = f32x4{0.f, 0.f, 0.f, 0.f};
  if (MODE == HOLD_CHAIN_SOFTPLUS) return;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t soff = (32 * m + 8 * g + k) * x.rowb;
    v1[k] = ldb(x.a1, x.goff, soff);
    if (A2) v2[k] = ldb(x.a2, x.goff, soff);
  }
}

template <int MODE, bool A2>
__device__ __forceinline__ void epi_exec(const EpiCtx& x, int u, const f32x16 (&acc)[2], const f32x4& v1,
                                         const f32x4& v2) {
  const int m = u >> 2, g = u & 3;
  f32x4 sv = {0.f, 0.f, 0.f, 0.f};
  if (MODE != HOLD_CHAIN_DSP && x.skip) {
#pragma unroll
    for (int k = 0; k < 4; ++k) sv[k] = x.scol[(32 * m + 8 * g + k) * ESTR];
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int o = 32 * m + 8 * g + k;
    const float y = (m == 0) ? acc[0][4 * g + k] : acc[1][4 * g + k];
    float r, r2 = 0.f;
    if (MODE == HOLD_CHAIN_SOFTPLUS) {
      r = softplus100(y + x.bias);
      r = x.special ? sv[k] : r;
    } else if (MODE == HOLD_CHAIN_DSP) {
      r = y * dsp_from_h(v1[k]);
      if (A2) r += v2[k];
      r = x.special ? y : r;  // raw product = d / d(skip input); the next layer's packed weights are zero there
    } else {
      const float e = __expf(-100.0f * v1[k]);
      r = y * dsp_from_h(v1[k]);
      r2 = 100.0f * y * v2[k] * e;
      r = x.special ? sv[k] : r;
      r2 = x.special ? 0.f : r2;
    }
    if (x.wr_lds) x.lcol[o * ASTR] = r;
    const uint32_t soff = o * x.rowb;
    stb(x.o1, x.goff, soff, r);
    if (MODE == HOLD_CHAIN_DBWD) stb(x.o2, x.goff, soff, r2);
  }
}

// chunk after whose MFMAs epilogue unit u runs, and chunk before whose MFMAs its side inputs are requested (two
// register sets: the request of unit u comes after the use of unit u - 2)
template <int CHUNKS>
__device__ constexpr int exec_at(int u) {
  return ((u + 1) * CHUNKS + 7) / 8 - 1 < CHUNKS - 1 ? ((u + 1) * CHUNKS + 7) / 8 - 1 : CHUNKS - 1;
}
template <int CHUNKS>
__device__ constexpr int load_at(int u) { return u < 2 ? 0 : exec_at<CHUNKS>(u - 2) + 1; }

// One pipeline step: MFMAs of (this layer, one half: 2 m-tiles x 32 features of this wave, K = 8 * CHUNKS) interleaved
// with the epilogue of the PREVIOUS step's accumulators.  b-register parity PH: chunk c lives in bq[(c + PH) & 1].
template <int MODE, bool A2, int CHUNKS, int PH, bool EPI>
__device__ __forceinline__ void chain_step(const f32x4* __restrict__ wp, const f32x4* __restrict__ nxt,
                                           const float* __restrict__ arow, f32x16 (&accC)[2], f32x4 (&bq)[2],
                                           const f32x16 (&accP)[2], const EpiCtx& x) {
  f32x4 s1[2], s2[2], avn[2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) accC[m][r] = 0.f;
#pragma unroll
  for (int m = 0; m < 2; ++m) avn[m] = *reinterpret_cast<const f32x4*>(arow + m * 32 * ASTR);
#pragma unroll
  for (int c = 0; c < CHUNKS; ++c) {
    // ---- issue: everything later chunks wait on (the scheduler must not sink these next to their uses) ----
    const f32x4 b = bq[(c + PH) & 1];
    f32x4 av[2] = {avn[0], avn[1]};
    if (c + 1 < CHUNKS) {
#pragma unroll
      for (int m = 0; m < 2; ++m) avn[m] = *reinterpret_cast<const f32x4*>(arow + m * 32 * ASTR + (c + 1) * 8);
    }
    if (c + 2 < CHUNKS) {
      bq[(c + PH) & 1] = wp[(c + 2) * 512];
    } else if (nxt) {
      bq[(c + PH) & 1] = nxt[(c + 2 - CHUNKS) * 512];
    }
    if (EPI) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (load_at<CHUNKS>(u) == c) epi_load<MODE, A2>(x, u, s1[u & 1], s2[u & 1]);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- compute: 8 MFMAs, the epilogue units due at this chunk interleaved by the scheduler ----
#pragma unroll
    for (int cc = 0; cc < 4; ++cc)
#pragma unroll
      for (int m = 0; m < 2; ++m) accC[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m][cc], b[cc], accC[m], 0, 0, 0);
    if (EPI) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (exec_at<CHUNKS>(u) == c) epi_exec<MODE, A2>(x, u, accP, s1[u & 1], s2[u & 1]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int MODE, bool A2, int FIRST>
__global__ __launch_bounds__(512, 2) void chain_kernel(hold_chain_desc d) {
  constexpr int PTS = 128, NTHR = 512, KIN = 8 * FIRST;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* act = smem;                   // [128][260]
  float* side = smem + PTS * ASTR;     // [128][40]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hh = lane >> 5, li = lane & 31;
  const f32x4* w0 = reinterpret_cast<const f32x4*>(d.wpack) + wave * 64 + lane;
  constexpr long LAYER0 = (long)FIRST * 512, LAYERK = 32L * 512;  // f32x4 units
  const int NL = d.n_layers;
  const long ld = d.ld;
  const uint32_t nbytes = (uint32_t)(d.P * ld * 4);

  for (long blk = blockIdx.x; blk * PTS < d.P; blk += gridDim.x) {
    const long p0 = blk * PTS;
    f32x4 bq[2];
    bq[0] = w0[0];
    bq[1] = w0[512];
    // ---- initial activations (and the 40-wide side matrix) -> LDS ----
    for (int e = tid; e < PTS * (KIN / 4); e += NTHR) {
      const int p = e / (KIN / 4), j4 = (e % (KIN / 4)) * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (p0 + p < d.P) v = *reinterpret_cast<const f32x4*>(d.in + (p0 + p) * d.ld_in + j4);
      *reinterpret_cast<f32x4*>(act + p * ASTR + j4) = v;
    }
    if (d.side) {
      for (int e = tid; e < PTS * (ESTR / 4); e += NTHR) {
        const int p = e / (ESTR / 4), j4 = (e % (ESTR / 4)) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (p0 + p < d.P) v = *reinterpret_cast<const f32x4*>(d.side + (p0 + p) * d.ld_side + j4);
        *reinterpret_cast<f32x4*>(side + p * ESTR + j4) = v;
      }
    }
    __syncthreads();

    float* row[2] = {act + li * ASTR, act + (64 + li) * ASTR};
    const int col = wave * 32 + li;
    auto ctx = [&](int layer, int half) {
      EpiCtx x;
      const long prow = p0 + half * 64 + 4 * hh;
      x.bias = (MODE == HOLD_CHAIN_SOFTPLUS) ? d.bias[layer][col] : 0.f;
      x.a1 = make_rsrc(d.aux1[layer], nbytes);
      x.a2 = make_rsrc(d.aux2[layer], nbytes);
      x.o1 = make_rsrc(d.out[layer], nbytes);
      x.o2 = make_rsrc(d.out2[layer], nbytes);
      x.goff = (uint32_t)((prow * ld + col) * 4);
      x.rowb = (uint32_t)(ld * 4);
      x.lcol = act + (half * 64 + 4 * hh) * ASTR + col;
      x.scol = side + (half * 64 + 4 * hh) * ESTR + (col - SKIP_OUT);
      x.skip = layer == d.skip_layer;
      x.special = x.skip && col >= SKIP_OUT;
      x.wr_lds = layer + 1 < NL;
      return x;
    };
    f32x16 accA[2], accB[2];
    const EpiCtx none = ctx(0, 0);
    chain_step<MODE, A2, FIRST, 0, false>(w0, w0, row[0] + hh * 4, accA, bq, accB, none);
    __syncthreads();
    chain_step<MODE, A2, FIRST, FIRST & 1, true>(w0, NL > 1 ? w0 + LAYER0 : nullptr, row[1] + hh * 4, accB, bq, accA,
                                              ctx(0, 0));
    __syncthreads();
    const f32x4* wl = w0 + LAYER0;
    for (int layer = 1; layer < NL; ++layer) {
      chain_step<MODE, A2, 32, 0, true>(wl, wl, row[0] + hh * 4, accA, bq, accB, ctx(layer - 1, 1));
      __syncthreads();
      chain_step<MODE, A2, 32, 0, true>(wl, layer + 1 < NL ? wl + LAYERK : nullptr, row[1] + hh * 4, accB, bq, accA,
                                    ctx(layer, 0));
      __syncthreads();
      wl += LAYERK;
    }
    // epilogue of (last layer, H1): nothing left to hide it under; requests run two units ahead
    {
      const EpiCtx x = ctx(NL - 1, 1);
      f32x4 s1[2], s2[2];
      epi_load<MODE, A2>(x, 0, s1[0], s2[0]);
      epi_load<MODE, A2>(x, 1, s1[1], s2[1]);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        epi_exec<MODE, A2>(x, u, accB, s1[u & 1], s2[u & 1]);
        if (u + 2 < 8) epi_load<MODE, A2>(x, u + 2, s1[u & 1], s2[u & 1]);
      }
    }
    // the next iteration's LDS fill only conflicts with MFMA reads that finished before the last barrier; the last
    // layer's epilogues do not write LDS
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// Split-precision variant (hold_chain_x6): the same sweeps, LDS layout, epilogues and pipeline, with the layer products
// on v_mfma_f32_32x32x16_bf16 -- every fp32 operand is the exact sum of three bf16 limbs, six of the nine limb products
// are issued, fp32 accumulation (the arithmetic of hold_fused_sdf_x6 / hold_wgrad_x6, include/hold_hip.h).  Activations
// stay fp32 in LDS and are split into limbs by truncation as they are fetched (8 consecutive k per lane: the top 16 bits,
// the top 16 bits of the exact remainder, the rest); the weights come pre-split from the limb pack
// [K/16 steps][3 limbs][8 n-tiles][64 lanes] x bf16x8 (hold_amd/field.py:pack_x6_mats), one step ahead.
constexpr int X6_UNITS = 3 * 512;  // 16-byte units per 16-wide k step

struct Limbs3 { bf16x8 l[3]; };

// Two values at a time: the compiler selects v_pk_add_f32 for the two remainders, so 8 values cost 16 v_and + 8 v_pk_add
// + 12 v_perm = 36 VALU issue slots (44 with scalar subtractions).  Measured neutral: neither this nor halving the number
// of splits (a four-wave variant, 64 features per wave) moved the kernel -- it is not VALU-issue-bound.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ Limbs3 split8_trunc(const f32x4& x0, const f32x4& x1) {
  u32x4 p1, p2, p3;
#pragma unroll
  for (int j = 0; j < 4; ++j) {  // dword j = bf16 elements 2j (low half) and 2j + 1 (high half)
    const f32x2 x = j < 2 ? f32x2{x0[2 * j], x0[2 * j + 1]} : f32x2{x1[2 * j - 4], x1[2 * j - 3]};
    const u32x2 xb = __builtin_bit_cast(u32x2, x);
    const f32x2 r1 = x - __builtin_bit_cast(f32x2, xb & 0xffff0000u);
    const u32x2 r1b = __builtin_bit_cast(u32x2, r1);
    const u32x2 r2b = __builtin_bit_cast(u32x2, r1 - __builtin_bit_cast(f32x2, r1b & 0xffff0000u));
    p1[j] = __builtin_amdgcn_perm(xb[1], xb[0], 0x07060302u);   // the high halves = the truncated limbs
    p2[j] = __builtin_amdgcn_perm(r1b[1], r1b[0], 0x07060302u);
    p3[j] = __builtin_amdgcn_perm(r2b[1], r2b[0], 0x07060302u);
  }
  Limbs3 o;
  o.l[0] = __builtin_bit_cast(bf16x8, p1);
  o.l[1] = __builtin_bit_cast(bf16x8, p2);
  o.l[2] = __builtin_bit_cast(bf16x8, p3);
  return o;
}

// One pipeline step (one layer x one 64-point half), K = 16 * STEPS.  Each 16-wide k step is two phases of 6 MFMAs
// (limb products 00 01 10 | 11 02 20 for the two 32-point tiles); the phases play the role of the fp32 kernel's 8-wide
// chunks for the placement of the previous half's epilogue units (exec_at / load_at with CHUNKS = 2 * STEPS).
template <int MODE, bool A2, int STEPS, bool EPI>
__device__ __forceinline__ void chain_step_x6(const bf16x8* __restrict__ wq, const bf16x8* __restrict__ nxt,
                                              const float* __restrict__ arow, f32x16 (&accC)[2], bf16x8 (&bn)[3],
                                              const f32x16 (&accP)[2], const EpiCtx& x) {
  constexpr int CH = 2 * STEPS;
  f32x4 s1[2], s2[2], xn[4];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) accC[m][r] = 0.f;
  auto rd = [&](int s) {  // this lane's rows (points li of the two tiles), k = 16 s + 8 hh .. + 8
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      xn[2 * m] = *reinterpret_cast<const f32x4*>(arow + m * 32 * ASTR + s * 16);
      xn[2 * m + 1] = *reinterpret_cast<const f32x4*>(arow + m * 32 * ASTR + s * 16 + 4);
    }
  };
  rd(0);
  Limbs3 la[2];
  la[0] = split8_trunc(xn[0], xn[1]);
  la[1] = split8_trunc(xn[2], xn[3]);
#pragma unroll
  for (int s = 0; s < STEPS; ++s) {
    // ---- phase A: issue (next step's fp32 fragments, next step's weight limbs, side inputs due), then 00 01 10 ----
    bf16x8 b[3];
    if (s + 1 < STEPS) rd(s + 1);
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      b[t] = bn[t];
      if (s + 1 < STEPS) {
        bn[t] = wq[(s + 1) * X6_UNITS + t * 512];
      } else if (nxt) {
        bn[t] = nxt[t * 512];
      }
    }
    if (EPI) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (load_at<CH>(u) == 2 * s) epi_load<MODE, A2>(x, u, s1[u & 1], s2[u & 1]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < 2; ++m) accC[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(la[m].l[0], b[0], accC[m], 0, 0, 0);
#pragma unroll
    for (int m = 0; m < 2; ++m) accC[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(la[m].l[1], b[0], accC[m], 0, 0, 0);
#pragma unroll
    for (int m = 0; m < 2; ++m) accC[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(la[m].l[0], b[1], accC[m], 0, 0, 0);
    if (EPI) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (exec_at<CH>(u) == 2 * s) epi_exec<MODE, A2>(x, u, accP, s1[u & 1], s2[u & 1]);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- phase B: side inputs due, then 11 02 20 with the next step's split interleaved ----
    if (EPI) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (load_at<CH>(u) == 2 * s + 1) epi_load<MODE, A2>(x, u, s1[u & 1], s2[u & 1]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < 2; ++m) accC[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(la[m].l[1], b[1], accC[m], 0, 0, 0);
#pragma unroll
    for (int m = 0; m < 2; ++m) accC[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(la[m].l[2], b[0], accC[m], 0, 0, 0);
#pragma unroll
    for (int m = 0; m < 2; ++m) accC[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(la[m].l[0], b[2], accC[m], 0, 0, 0);
    if (s + 1 < STEPS) {
      la[0] = split8_trunc(xn[0], xn[1]);
      la[1] = split8_trunc(xn[2], xn[3]);
    }
    if (EPI) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (exec_at<CH>(u) == 2 * s + 1) epi_exec<MODE, A2>(x, u, accP, s1[u & 1], s2[u & 1]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int MODE, bool A2, int FIRST_STEPS>
__global__ __launch_bounds__(512, 2) void chain_x6_kernel(hold_chain_desc d) {
  constexpr int PTS = 128, NTHR = 512;
  constexpr int KIN = (FIRST_STEPS == 3) ? 40 : 256, KPAD = 16 * FIRST_STEPS;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* act = smem;                   // [128][260]
  float* side = smem + PTS * ASTR;     // [128][40]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hh = lane >> 5, li = lane & 31;
  const bf16x8* w0 = reinterpret_cast<const bf16x8*>(d.wpack) + wave * 64 + lane;
  constexpr long LAYER0 = (long)FIRST_STEPS * X6_UNITS, LAYERK = 16L * X6_UNITS;  // 16-byte units
  const int NL = d.n_layers;
  const long ld = d.ld;
  const uint32_t nbytes = (uint32_t)(d.P * ld * 4);

  for (long blk = blockIdx.x; blk * PTS < d.P; blk += gridDim.x) {
    const long p0 = blk * PTS;
    bf16x8 bn[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) bn[t] = w0[t * 512];
    // ---- initial activations (zero-padded to a multiple of 16 columns) and the 40-wide side matrix -> LDS ----
    for (int e = tid; e < PTS * (KPAD / 4); e += NTHR) {
      const int p = e / (KPAD / 4), j4 = (e % (KPAD / 4)) * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (j4 < KIN && p0 + p < d.P) v = *reinterpret_cast<const f32x4*>(d.in + (p0 + p) * d.ld_in + j4);
      *reinterpret_cast<f32x4*>(act + p * ASTR + j4) = v;
    }
    if (d.side) {
      for (int e = tid; e < PTS * (ESTR / 4); e += NTHR) {
        const int p = e / (ESTR / 4), j4 = (e % (ESTR / 4)) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (p0 + p < d.P) v = *reinterpret_cast<const f32x4*>(d.side + (p0 + p) * d.ld_side + j4);
        *reinterpret_cast<f32x4*>(side + p * ESTR + j4) = v;
      }
    }
    __syncthreads();

    float* row[2] = {act + li * ASTR, act + (64 + li) * ASTR};
    const int col = wave * 32 + li;
    auto ctx = [&](int layer, int half) {
      EpiCtx x;
      const long prow = p0 + half * 64 + 4 * hh;
      x.bias = (MODE == HOLD_CHAIN_SOFTPLUS) ? d.bias[layer][col] : 0.f;
      x.a1 = make_rsrc(d.aux1[layer], nbytes);
      x.a2 = make_rsrc(d.aux2[layer], nbytes);
      x.o1 = make_rsrc(d.out[layer], nbytes);
      x.o2 = make_rsrc(d.out2[layer], nbytes);
      x.goff = (uint32_t)((prow * ld + col) * 4);
      x.rowb = (uint32_t)(ld * 4);
      x.lcol = act + (half * 64 + 4 * hh) * ASTR + col;
      x.scol = side + (half * 64 + 4 * hh) * ESTR + (col - SKIP_OUT);
      x.skip = layer == d.skip_layer;
      x.special = x.skip && col >= SKIP_OUT;
      x.wr_lds = layer + 1 < NL;
      return x;
    };
    f32x16 accA[2], accB[2];
    const EpiCtx none = ctx(0, 0);
    chain_step_x6<MODE, A2, FIRST_STEPS, false>(w0, w0, row[0] + hh * 8, accA, bn, accB, none);
    __syncthreads();
    chain_step_x6<MODE, A2, FIRST_STEPS, true>(w0, NL > 1 ? w0 + LAYER0 : nullptr, row[1] + hh * 8, accB, bn, accA,
                                               ctx(0, 0));
    __syncthreads();
    const bf16x8* wl = w0 + LAYER0;
    for (int layer = 1; layer < NL; ++layer) {
      chain_step_x6<MODE, A2, 16, true>(wl, wl, row[0] + hh * 8, accA, bn, accB, ctx(layer - 1, 1));
      __syncthreads();
      chain_step_x6<MODE, A2, 16, true>(wl, layer + 1 < NL ? wl + LAYERK : nullptr, row[1] + hh * 8, accB, bn, accA,
                                        ctx(layer, 0));
      __syncthreads();
      wl += LAYERK;
    }
    {  // epilogue of (last layer, H1): nothing left to hide it under; requests run two units ahead
      const EpiCtx x = ctx(NL - 1, 1);
      f32x4 s1[2], s2[2];
      epi_load<MODE, A2>(x, 0, s1[0], s2[0]);
      epi_load<MODE, A2>(x, 1, s1[1], s2[1]);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        epi_exec<MODE, A2>(x, u, accB, s1[u & 1], s2[u & 1]);
        if (u + 2 < 8) epi_load<MODE, A2>(x, u + 2, s1[u & 1], s2[u & 1]);
      }
    }
  }
}




bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

template <int MODE, bool A2, int FIRST>
int launch(const hold_chain_desc& d, int n_cu, hipStream_t s) {
  const size_t sh = (size_t)(128 * ASTR + 128 * ESTR) * sizeof(float);  // 153 600 B, one block per CU
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)chain_kernel<MODE, A2, FIRST>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)sh) != hipSuccess)
      return HOLD_E_LAUNCH;
    attr_set = true;
  }
  const long blocks = (d.P + 127) / 128;
  hipLaunchKernelGGL((chain_kernel<MODE, A2, FIRST>), dim3((unsigned)(blocks < n_cu ? blocks : n_cu)), dim3(512), sh, s, d);
  return hipGetLastError() == hipSuccess ? HOLD_OK : HOLD_E_LAUNCH;
}

template <int MODE, bool A2, int FIRST_STEPS>
int launch_x6(const hold_chain_desc& d, int n_cu, hipStream_t s) {
  const size_t sh = (size_t)(128 * ASTR + 128 * ESTR) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)chain_x6_kernel<MODE, A2, FIRST_STEPS>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)sh) != hipSuccess)
      return HOLD_E_LAUNCH;
    attr_set = true;
  }
  const long blocks = (d.P + 127) / 128;
  hipLaunchKernelGGL((chain_x6_kernel<MODE, A2, FIRST_STEPS>), dim3((unsigned)(blocks < n_cu ? blocks : n_cu)), dim3(512), sh,
                     s, d);
  return hipGetLastError() == hipSuccess ? HOLD_OK : HOLD_E_LAUNCH;
}

}  // namespace

extern "C" int64_t hold_chain_pack_floats(int32_t first_chunks, int32_t n_layers) {
  if ((first_chunks != 5 && first_chunks != 32) || n_layers < 1 || n_layers > 8) return -1;
  return ((int64_t)first_chunks + 32 * (int64_t)(n_layers - 1)) * 2048;
}

extern "C" int64_t hold_chain_x6_pack_bytes(int32_t first_chunks, int32_t n_layers) {
  if ((first_chunks != 5 && first_chunks != 32) || n_layers < 1 || n_layers > 8) return -1;
  return ((int64_t)(first_chunks == 5 ? 3 : 16) + 16 * (int64_t)(n_layers - 1)) * X6_UNITS * 16;
}

static int chain_impl(const hold_chain_desc* dp, hold_stream_t st, bool x6);

extern "C" int hold_chain(const hold_chain_desc* dp, hold_stream_t st) { return chain_impl(dp, st, false); }
extern "C" int hold_chain_x6(const hold_chain_desc* dp, hold_stream_t st) { return chain_impl(dp, st, true); }

static int chain_impl(const hold_chain_desc* dp, hold_stream_t st, bool x6) {
  if (!dp) return HOLD_E_ARG;
  const hold_chain_desc& d = *dp;
  if (d.P < 0 || d.n_layers < 1 || d.n_layers > 8 || !d.in || !d.wpack) return HOLD_E_ARG;
  if (d.first_chunks != 5 && d.first_chunks != 32) return HOLD_E_ARG;
  if (d.ld < 256 || (d.ld & 3) || d.ld_in < 8 * d.first_chunks || (d.ld_in & 3)) return HOLD_E_ARG;
  if (!al16(d.in) || !al16(d.wpack) || (d.side && (!al16(d.side) || d.ld_side < 40 || (d.ld_side & 3))))
    return HOLD_E_ARG;
  if (d.skip_layer >= d.n_layers - 1) return HOLD_E_ARG;  // the side matrix must not be re-filled under a reader
  if (d.skip_out != 0 && d.skip_out != SKIP_OUT) return HOLD_E_ARG;  // only hold_chain_r6 knows another skip width
  // 32-bit byte offsets inside the kernels (incl. the rows of a partial last block): split larger batches by rows
  if (((uint64_t)d.P + 128) * (uint64_t)d.ld * 4 >= (1ull << 32)) return HOLD_E_ARG;
  const bool has2 = d.aux2[0] != nullptr;
  for (int l = 0; l < d.n_layers; ++l) {
    if ((d.aux2[l] != nullptr) != has2) return HOLD_E_ARG;
    if (!al16(d.aux1[l]) || !al16(d.aux2[l]) || !al16(d.out[l]) || !al16(d.out2[l]) || !al16(d.bias[l]))
      return HOLD_E_ARG;
    if (d.mode == HOLD_CHAIN_SOFTPLUS && !d.bias[l]) return HOLD_E_ARG;
    if (d.mode == HOLD_CHAIN_DSP && !d.aux1[l]) return HOLD_E_ARG;
    if (d.mode == HOLD_CHAIN_DBWD && (!d.aux1[l] || !d.aux2[l] || !d.out2[l])) return HOLD_E_ARG;
  }
  if (d.skip_layer >= 0 && d.mode != HOLD_CHAIN_DSP && !d.side) return HOLD_E_ARG;
  if (d.P == 0) return HOLD_OK;
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return HOLD_E_LAUNCH;
    n_cu = prop.multiProcessorCount;
  }
  hipStream_t s = (hipStream_t)st;
  if (x6) {
    if (d.mode == HOLD_CHAIN_SOFTPLUS && d.first_chunks == 5) return launch_x6<HOLD_CHAIN_SOFTPLUS, false, 3>(d, n_cu, s);
    if (d.mode == HOLD_CHAIN_DSP && d.first_chunks == 32)
      return has2 ? launch_x6<HOLD_CHAIN_DSP, true, 16>(d, n_cu, s) : launch_x6<HOLD_CHAIN_DSP, false, 16>(d, n_cu, s);
    if (d.mode == HOLD_CHAIN_DBWD && d.first_chunks == 5) return launch_x6<HOLD_CHAIN_DBWD, true, 3>(d, n_cu, s);
    return HOLD_E_ARG;
  }
  if (d.mode == HOLD_CHAIN_SOFTPLUS && d.first_chunks == 5) return launch<HOLD_CHAIN_SOFTPLUS, false, 5>(d, n_cu, s);
  if (d.mode == HOLD_CHAIN_DSP && d.first_chunks == 32)
    return has2 ? launch<HOLD_CHAIN_DSP, true, 32>(d, n_cu, s) : launch<HOLD_CHAIN_DSP, false, 32>(d, n_cu, s);
  if (d.mode == HOLD_CHAIN_DBWD && d.first_chunks == 5) return launch<HOLD_CHAIN_DBWD, true, 5>(d, n_cu, s);
  return HOLD_E_ARG;
}
