// Register-resident trunk of the ImplicitNet (gfx950): lin0..lin7 of ImplicitNet.forward (code/src/networks/shape_net.py:84-130)
// for 128 points per workgroup with NO activation traffic through LDS or HBM between layers.
//
// Structure (one wave per SIMD, 4 waves = 256 threads, one workgroup per CU, persistent over point blocks):
//   * a wave owns 32 points for the whole network.  Every layer is D[feature][point] = W (A operand, rows = output
//     features) x act (B operand, columns = points) on v_mfma_f32_32x32x16_bf16 with the exact three-limb bf16 split of
//     both operands (six limb products, fp32 accumulation -- the arithmetic of hold_fused_sdf_x6 / hold_chain_x6).
//   * the 256 x 32 fp32 outputs of a layer stay in the wave's registers (8 accumulator tiles): lane (hh, li) holds, for
//     point li, features 32 nt + 8 g + 4 hh + k in register 4 g + k of tile nt.  Registers 8 q .. 8 q + 7 of tile nt are
//     therefore 8 values of ONE point -- exactly what a B-operand lane holds for one 16-wide k step -- so the next layer
//     contracts over the "virtual" k order  k-step j = (nt, q) = (j / 2, j % 2),  element e of lane half hh  <->
//     feature 32 nt + 16 q + 8 (e / 4) + 4 hh + e % 4,  and the host packs the weight limbs in that same order
//     (hold_amd/field.py:pack_r6).  No transpose, no cross-lane traffic: softplus + limb split of the previous layer's
//     accumulators (pure VALU) is interleaved with the MFMAs of the current k step ("input stationary": each k step's
//     B limbs feed 48 MFMAs into the 8 accumulator tiles of the layer being computed).
//   * the weight limbs are the only stream: 24 KiB per k step (8 n-tiles x 3 limbs x 1 KiB fragments), brought into a
//     5-slot LDS ring by LDS-DMA (global_load_lds_dwordx4, six 1 KiB pieces per wave and k step, issued four steps
//     ahead) and read by all four waves with conflict-free lane-linear ds_read_b128 -- one L2 read of each weight byte
//     per 128 points (the 8-wave kernels read it twice per 128 points), one raw s_barrier per k step, counted vmcnt.
//
// Entry points: hold_fused_sdf_r6 (the sampler's SDF query: embedding in-kernel, sdf = w8 . h7 + b8 out) and
// hold_trunk_r6 (training forward: additionally stores h_0..h_7, the skip layer's columns 217.. = the embedding).
// Roofline: bf16 MFMA pipe (6 limb products issued per algorithmic product); algorithmic HBM bytes per point:
// 16 in + 4 out (sdf) / + 8 KiB of h stores (trunk).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/hold_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int NW = 4;                 // waves per workgroup (one per SIMD)
constexpr int BPTS = 32 * NW;         // points per workgroup pass
constexpr int PIECE = 1024;           // one MFMA A fragment for a wave: 64 lanes x 16 B
constexpr int SLOT = 24 * PIECE;      // one k step: [8 n-tiles][3 limbs] fragments
constexpr int RING = 5;               // LDS slots; NSTEP % RING == 0 keeps slot = step % RING across blocks
constexpr int L0S = 3, LKS = 16;      // k steps of layer 0 (K = 48, 39 used) and of the 256-wide layers
constexpr int NSTEP = L0S + 7 * LKS;  // 115 k steps per block of points
constexpr int NE = 39, EMB_STR = 52, SKIP_OUT = 217;
constexpr int OFF_BIAS = RING * SLOT;               // [8][256] fp32
constexpr int OFF_W8 = OFF_BIAS + 8 * 256 * 4;      // [256] fp32
constexpr int OFF_EMB = OFF_W8 + 256 * 4;           // [4 waves][32 points][EMB_STR] fp32 (wave-private)
constexpr int OFF_BARF = OFF_EMB + NW * 32 * EMB_STR * 4;  // [64] fp32: BARF weights of the 39 embedding columns (or 1)
constexpr int LDS_BYTES = OFF_BARF + 64 * 4;
static_assert(NSTEP % RING == 0, "slot index must not depend on the block iteration");
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");

struct R6Args {
  const float* xc; int ldx; long P;
  const char* wpack;    // hold_trunk_r6_pack_bytes() bytes, [NSTEP][24 pieces][64 lanes][8 bf16]
  const float* bias;    // [8][256]
  const float* w8;      // [256] sdf row of lin8 (HEAD)
  const float* b8;      // device scalar: bias of the sdf row (HEAD)
  const float* barf;    // [39] or null
  float* sdf; int lds;  // HEAD output
  float* h[8]; int ldh; // STORE outputs ([P][ldh], columns 0..255)
  // CONDITIONAL launch (hold_*_r6_if: the fallback of the f16x3 kernels, csrc/rmlp_h3.hip): null = always run.  Otherwise
  // 4 words of device memory: the kernel exits at once unless [0] != 0; if it ran, the last workgroup to finish counts the
  // event in [2] and clears [0] and its own arrival counter [1] -- every workgroup has read [0] by then
  uint32_t* guard;
};

__device__ __forceinline__ uint32_t fbits(float x) { return __builtin_bit_cast(uint32_t, x); }
__device__ __forceinline__ float bitsf(uint32_t x) { return __builtin_bit_cast(float, x); }

// max(y, 0) as ONE v_max_i32 (fmaxf costs a canonicalising v_max y,y first)
__device__ __forceinline__ float relu1(float y) {  // sign bit set <=> negative as an integer
  const int b = __builtin_bit_cast(int, y);
  return __builtin_bit_cast(float, b > 0 ? b : 0);
}

// value-only softplus (sampler queries): max(y,0) + ln2/100 * log2(1 + 2^(-100 log2e |y|)), abs error <= 6e-10
__device__ __forceinline__ float sp_fast(float y) {
  const float e = __builtin_amdgcn_exp2f(-144.26950408889634f * fabsf(y));
  return fmaf(0.0069314718056f, __builtin_amdgcn_logf(1.0f + e), relu1(y));
}
// training softplus: log1p by series where 1 + e would round e away (the backward sweeps recover softplus' from the
// stored h, so small h need relative accuracy); y > 0.2 returns y as the reference's threshold branch does
__device__ __forceinline__ float sp_train(float y) {
  const float e = __builtin_amdgcn_exp2f(-144.26950408889634f * fabsf(y));
  const float lg = 0.0069314718056f * __builtin_amdgcn_logf(1.0f + e);
  const float ser = (0.01f * e) * fmaf(e, fmaf(e, 0.33333334f, -0.5f), 1.0f);
  const float l = (e > 1e-3f) ? lg : ser;
  const float r = relu1(y) + l;
  return (y > 0.2f) ? y : r;
}

typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const float* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, p ? bytes : 0u, 0x00020000);
}

// 16-byte row-fragment store.  The column offset goes into the instruction's immediate field (constant part of
// voffset), NOT into soffset: with an SGPR soffset hipcc assumes the ">64-bit store data overwritten by the next VALU"
// hazard away and re-uses the data registers right behind the store -- on gfx950 that corrupted the stored rows
// (first hardware run of this kernel: the layer-7 block, where results are produced back to back, was wrong).
__device__ __forceinline__ void store4(const f32x4& v, rsrc_t rs, uint32_t voff) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, voff, 0, 0);
  asm volatile("s_nop 3");
}

struct Limbs { u32x4 l[3]; };  // three bf16x8 B fragments, as dwords (dword d = elements 2 d, 2 d + 1)

// exact truncation split of two fp32 values into one dword of each of the three bf16 limb fragments
struct Split3 { uint32_t p1, p2, p3; };
__device__ __forceinline__ Split3 split2(float x0, float x1) {
  const uint32_t b0 = fbits(x0), b1 = fbits(x1);
  const float r0 = x0 - bitsf(b0 & 0xffff0000u), r1 = x1 - bitsf(b1 & 0xffff0000u);
  const uint32_t c0 = fbits(r0), c1 = fbits(r1);
  const float s0 = r0 - bitsf(c0 & 0xffff0000u), s1 = r1 - bitsf(c1 & 0xffff0000u);
  Split3 o;
  o.p1 = __builtin_amdgcn_perm(b1, b0, 0x07060302u);
  o.p2 = __builtin_amdgcn_perm(c1, c0, 0x07060302u);
  o.p3 = __builtin_amdgcn_perm(fbits(s1), fbits(s0), 0x07060302u);
  return o;
}
// the three dwords are made opaque HERE: pure VALU code has no side effects, so without the pin LLVM sinks the whole
// next-step epilogue to its first use (the end of the k step) instead of leaving a quarter of it in each MFMA group
__device__ __forceinline__ void put_limbs(Limbs& out, int c, Split3 s) {
  asm volatile("" : "+v"(s.p1), "+v"(s.p2), "+v"(s.p3));
  out.l[0][c] = s.p1;
  out.l[1][c] = s.p2;
  out.l[2][c] = s.p3;
}

// Six 1 KiB pieces of k step `step` into ring slot `slot` (pieces 6 w .. 6 w + 5 are wave w's) by LDS-DMA.
// Issued from inline assembly on purpose: hipcc models the global_load_lds builtin as a FLAT access that may touch LDS,
// which makes it wait lgkmcnt(0) at every later ds_read use -- the fragment prefetch of the next MFMA group would be
// waited for together with the current one's.  The DMA is invisible to the compiler's counters; its completion is
// counted by hand (R6_WAIT_VM before the rendezvous barrier).  M0 = LDS destination of lane 0; the instruction offset
// advances BOTH the global and the LDS address (pieces 0..3), pieces 4..5 take a second base pair.
template <int DMAV>
__device__ __forceinline__ void dma_step(const char* wpack, uint32_t lane16, int step, int slot, int wave) {
  const char* src = wpack + (long)step * SLOT + wave * (6 * PIECE);  // wave-uniform
  const uint32_t dst = (uint32_t)(slot * SLOT + wave * (6 * PIECE));  // dynamic LDS starts at byte 0 of the allocation
  uint32_t keep;
  if (DMAV == 1) {  // developer-build cross-check: no instruction offsets, M0 and the base pair advanced per piece
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %8\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %3\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %4\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %5\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %6\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %7\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(lane16), "s"(src), "s"(src + PIECE), "s"(src + 2 * PIECE), "s"(src + 3 * PIECE), "s"(src + 4 * PIECE),
          "s"(src + 5 * PIECE), "s"(dst)
        : "memory");
    return;
  }
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %4\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2\n\t"
      "global_load_lds_dwordx4 %1, %2 offset:1024\n\t"
      "global_load_lds_dwordx4 %1, %2 offset:2048\n\t"
      "global_load_lds_dwordx4 %1, %2 offset:3072\n\t"
      "s_mov_b32 m0, %5\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %3\n\t"
      "global_load_lds_dwordx4 %1, %3 offset:1024\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(lane16), "s"(src), "s"(src + 4 * PIECE), "s"(dst), "s"(dst + 4 * PIECE)
      : "memory");
}

// one 1 KiB piece (wave-uniform source, lane offset lane16) to LDS byte `dst`.  M0 is not saved: nothing else in these
// kernels uses it (no LDS-DMA builtin, no movrel), every statement that reads it writes it first.
__device__ __forceinline__ void dma_piece(const char* src, uint32_t lane16, uint32_t dst) {
  asm volatile(
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %0, %1"
      :
      : "v"(lane16), "s"(src), "s"(dst)
      : "memory");
}

#define R6_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

template <bool HEAD, bool STORE, int DMAV = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void rmlp_kernel(R6Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hh = lane >> 5, li = lane & 31;
  const uint32_t lane16 = lane * 16;
  float* embw = reinterpret_cast<float*>(smem + OFF_EMB) + wave * (32 * EMB_STR);
  const char* ring_lane = smem + lane * 16;
  if (a.guard && __hip_atomic_load(a.guard, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;  // wave-uniform
  const float b8 = HEAD ? *a.b8 : 0.f;  // read on the device: a host copy of a trained parameter costs a stream drain

  // ---- once per workgroup: biases (+ the sdf row) into LDS, the first four k steps into the ring ----
  for (int i = tid; i < 8 * 256; i += 256) reinterpret_cast<float*>(smem + OFF_BIAS)[i] = a.bias[i];
  if (HEAD) reinterpret_cast<float*>(smem + OFF_W8)[tid] = a.w8[tid];
  if (tid < 64) reinterpret_cast<float*>(smem + OFF_BARF)[tid] = (a.barf && tid < NE) ? a.barf[tid] : 1.0f;
  __syncthreads();
#pragma unroll
  for (int s = 0; s < 4; ++s) dma_step<DMAV>(a.wpack, lane16, s, s, wave);

  f32x16 P[8], Q[8];
  u32x4 A[2][6];  // weight fragments of two n-tiles x three limbs, double-buffered
  Limbs Bc, Bn;

  auto read_pair = [&](int slot, int pair, u32x4 (&dst)[6]) {
    const char* base = ring_lane + slot * SLOT + pair * (6 * PIECE);
#pragma unroll
    for (int i = 0; i < 6; ++i) dst[i] = *reinterpret_cast<const u32x4*>(base + i * PIECE);
  };
  auto init_bias = [&](int layer) {
    const float* bl = reinterpret_cast<const float*>(smem + OFF_BIAS) + layer * 256 + 4 * hh;
#pragma unroll
    for (int nt = 0; nt < 8; ++nt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(bl + 32 * nt + 8 * g);
#pragma unroll
        for (int k = 0; k < 4; ++k) Q[nt][4 * g + k] = b[k];
      }
  };

  int first = 1;
  for (long blk = blockIdx.x; blk * BPTS < a.P; blk += gridDim.x) {
    const long p0 = blk * BPTS + wave * 32;  // this wave's first point
    long prow = p0 + li;
    const bool prow_ok = prow < a.P;
    prow = prow_ok ? prow : a.P - 1;
    // ---- embedding of this wave's 32 points -> wave-private LDS [32][EMB_STR] (embedders.py:18-50): lane half hh takes
    // the frequencies 3 hh .. 3 hh + 2 of its point (9 sincosf), half 0 also the raw coordinates, half 1 the zero padding
    {
      const float* xr = a.xc + prow * a.ldx;
      const float x3[3] = {xr[0], xr[1], xr[2]};
      float* er = embw + li * EMB_STR;
      const float* bw = reinterpret_cast<const float*>(smem + OFF_BARF);
#pragma unroll
      for (int kk = 0; kk < 3; ++kk) {
        const int k = 3 * hh + kk;
        const float f = (float)(1 << k);
#pragma unroll
        for (int dim = 0; dim < 3; ++dim) {
          float sn, cs;
          sincosf(x3[dim] * f, &sn, &cs);
          const int j = 3 + 6 * k + dim;
          er[j] = sn * bw[j];
          er[j + 3] = cs * bw[j + 3];
        }
      }
#pragma unroll
      for (int i = 0; i < 3; ++i) er[hh ? NE + i : i] = hh ? 0.f : x3[i] * bw[i];
#pragma unroll
      for (int i = 3; i < 9; ++i)
        if (hh) er[NE + i] = 0.f;
    }
    if (first) {  // step 0 of the very first block: everybody's pieces landed
      R6_WAIT_VM(18);
      __builtin_amdgcn_s_barrier();
      read_pair(0, 0, A[0]);
      first = 0;
    }

    // One k step of the layer being accumulated into Q (`t` = step index in the block's stream, ring slot t % RING), with an
    // EXPLICIT schedule: the step is 4 groups x 12 MFMAs, and behind every MFMA stands a fixed slice of everything else
    // -- one fragment read for the next group (gaps 0..5), one DMA piece (gaps 1, 5, 9 of the two groups behind the
    // rendezvous), and cnt[group] / 12 micro-operations of the next step's epilogue (mop(group, k), k = 0 .. cnt - 1, in
    // order) -- closed by a full scheduling barrier.  With one wave per SIMD the wave must be back at the next MFMA within
    // the 32 cycles the current one runs; left to the scheduler the epilogue's VALU clustered in runs of 15 - 30
    // instructions between runs of back-to-back MFMAs and the matrix pipe drained in every one of them.
    auto kstep = [&](int t, const int (&cnt)[4], auto&& mop) {
      const int slot = t % RING;
#pragma unroll
      for (int pair = 0; pair < 4; ++pair) {
        if (pair == 2) {  // mid-step rendezvous: step t + 1 complete in LDS, slot of step t - 1 free
          R6_WAIT_VM(12);
          __builtin_amdgcn_s_barrier();
        }
        const char* rd = ring_lane + (pair < 3 ? slot * SLOT + (pair + 1) * (6 * PIECE) : ((t + 1) % RING) * SLOT);
        const char* src = a.wpack + (long)((t + 4) % NSTEP) * SLOT + wave * (6 * PIECE) + (pair & 1) * (3 * PIECE);
        const uint32_t dst = (uint32_t)(((t + 4) % RING) * SLOT + wave * (6 * PIECE) + (pair & 1) * (3 * PIECE));
#pragma unroll
        for (int m = 0; m < 12; ++m) {
          const int pr = m >> 1, tl = m & 1;
          const int wl = (pr == 2 || pr == 3) ? 1 : (pr == 5 ? 2 : 0);  // (w limb, act limb): 00 01 10 11 02 20
          const int al = (pr == 1 || pr == 3) ? 1 : (pr == 4 ? 2 : 0);
          Q[2 * pair + tl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[pair & 1][3 * tl + wl]),
                                                                    __builtin_bit_cast(bf16x8, Bc.l[al]), Q[2 * pair + tl],
                                                                    0, 0, 0);
          if (m < 6) A[(pair + 1) & 1][m] = *reinterpret_cast<const u32x4*>(rd + m * PIECE);
          if (pair >= 2 && (m & 3) == 1) dma_piece(src + (m >> 2) * PIECE, lane16, dst + (m >> 2) * PIECE);
#pragma unroll
          for (int u = 0; u < 6; ++u) {  // constant trip count (the slice bounds fold once pair and m are unrolled)
            const int k = cnt[pair] * m / 12 + u;
            if (k < cnt[pair] * (m + 1) / 12) mop(pair, k);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      Bc = Bn;
    };
    static constexpr int CNT_NONE[4] = {0, 0, 0, 0};
    auto no_mop = [](int, int) {};

    // ---- layer 0: B limbs straight from the embedding (natural k order 16 j + 8 hh + e) ----
    auto emb_limbs = [&](int j, int c, Limbs& out) {
      const float* er = embw + li * EMB_STR + 16 * j + 8 * hh + 2 * c;
      put_limbs(out, c, split2(er[0], er[1]));
    };
    init_bias(0);
#pragma unroll
    for (int c = 0; c < 4; ++c) emb_limbs(0, c, Bc);
    static constexpr int CNT_EMB[4] = {1, 1, 1, 1};
#pragma unroll
    for (int j = 0; j < L0S; ++j) {
      if (j + 1 < L0S)
        kstep(j, CNT_EMB, [&](int c, int) { emb_limbs(j + 1, c, Bn); });
      else
        kstep(j, CNT_NONE, no_mop);
    }

    // ---- layers 1..7: input = softplus of the previous layer's accumulators ----
    float part = 0.f;  // HEAD: this lane's share of w8 . h7
    // STORE: h rows through a buffer descriptor (rows >= P fall outside num_records: the hardware drops those stores)
    rsrc_t hrs = make_rsrc(nullptr, 0);
    const uint32_t hbytes = (uint32_t)(a.P * a.ldh * 4);
    const uint32_t hvoff = (uint32_t)(((p0 + li) * a.ldh + 4 * hh) * 4);
    // Epilogue of k step j = (nt, q) of the finished layer: its 8 values P[nt][8 q + i] (features 32 nt + 16 q + 8 (i / 4) +
    // 4 hh + i % 4), as FOUR stages (one per MFMA group of the running k step) of micro-operations in ROUND-MAJOR order:
    // micro-operation k of a stage = round k / 8 on value k % 8, so consecutive micro-operations are independent and a
    // dependent pair (exp -> add -> log -> fma -> and -> sub ...) is eight apart -- with one wave per SIMD nothing else
    // would hide a dependent VALU chain's latency.
    //   stage 0: y, |y| c, e = exp2(.), 1 + e (training: + the three operations of the log1p series)
    //   stage 1: log2, max(y, 0), softplus (training: series select, threshold select), skip-layer override  -> r[8]
    //   stage 2 / 3: limb split of r[0..3] / r[4..7] (two dwords each, the two dwords' operations alternating), store
    struct EpiState { float y[8], u[8], e[8], ser[8], r[8]; uint32_t w[2][8]; };
    static constexpr int CNT_SP[4] = {HEAD ? 32 : 64, HEAD ? 24 : 48, 22 + (STORE ? 1 : 0), 22 + (STORE ? 1 : 0)};
    auto epi_mop = [&](int layer, int j, int stage, int k, Limbs& out, EpiState& st) {
      const int nt = j >> 1, q = j & 1;
      const int rd = k >> 3, i = k & 7;
      if (stage == 0) {
        if (rd == 0) st.y[i] = P[nt][8 * q + i];
        else if (rd == 1) st.e[i] = -144.26950408889634f * fabsf(st.y[i]);
        else if (rd == 2) st.e[i] = __builtin_amdgcn_exp2f(st.e[i]);
        else if (rd == 3) st.u[i] = 1.0f + st.e[i];
        else if (rd == 4) st.ser[i] = fmaf(st.e[i], 0.33333334f, -0.5f);
        else if (rd == 5) st.ser[i] = fmaf(st.e[i], st.ser[i], 1.0f);
        else if (rd == 6) st.e[i] = 0.01f * st.e[i];
        else st.ser[i] = st.e[i] * st.ser[i];
      } else if (stage == 1) {
        if (rd == 0) st.u[i] = __builtin_amdgcn_logf(st.u[i]);
        else if (rd == 1) st.r[i] = relu1(st.y[i]);
        else if (HEAD) {
          float r = fmaf(0.0069314718056f, st.u[i], st.r[i]);
          if (j >= 13) {  // skip connection: columns 217.. of layer 3's output are the embedding (shape_net.py:122-123)
            const int m = 32 * nt + 16 * q + 8 * (i >> 2) + 4 * hh + (i & 3) - SKIP_OUT;
            const float ev = embw[li * EMB_STR + (m < 0 ? 0 : m)];
            r = (layer == 4 && m >= 0) ? ev : r;
          }
          st.r[i] = r;
        } else if (rd == 2) {
          st.u[i] = 0.0069314718056f * st.u[i];
        } else if (rd == 3) {  // e > 1e-3 (e was scaled by 0.01 in stage 0): log(1 + e) is accurate; else the series
          st.u[i] = (st.e[i] > 1e-5f) ? st.u[i] : st.ser[i];
        } else if (rd == 4) {
          st.r[i] = st.r[i] + st.u[i];
        } else {
          float r = (st.y[i] > 0.2f) ? st.y[i] : st.r[i];
          if (j >= 13) {
            const int m = 32 * nt + 16 * q + 8 * (i >> 2) + 4 * hh + (i & 3) - SKIP_OUT;
            const float ev = embw[li * EMB_STR + (m < 0 ? 0 : m)];
            r = (layer == 4 && m >= 0) ? ev : r;
          }
          st.r[i] = r;
        }
      } else {
        const int h2 = stage - 2;
        if (k == 22) {  // STORE: the four consecutive features of this half
          const f32x4 v = {st.r[4 * h2], st.r[4 * h2 + 1], st.r[4 * h2 + 2], st.r[4 * h2 + 3]};
          if (DMAV == 2) return;  // developer-build ablation: no activation stores
          if (DMAV == 3) {        // developer-build ablation: the same bytes as lane-linear 1 KiB stores (wrong layout)
            store4(v, hrs, (uint32_t)(p0 * a.ldh * 4) + lane16 + (4 * nt + 2 * q + h2) * 1024);
            return;
          }
          store4(v, hrs, hvoff + (32 * nt + 16 * q + 8 * h2) * 4);
          return;
        }
        const int d = k & 1, op = k >> 1;  // dword d of this half (values 4 h2 + 2 d, + 1), operation op = 0..10
        const float x0 = st.r[4 * h2 + 2 * d], x1 = st.r[4 * h2 + 2 * d + 1];
        uint32_t* w = st.w[d];  // w[0..1] high parts, w[2..3] first remainders, w[4..5] their high parts, w[6..7] second remainders
        if (op == 0) w[0] = fbits(x0) & 0xffff0000u;
        else if (op == 1) w[1] = fbits(x1) & 0xffff0000u;
        else if (op == 2) w[2] = fbits(x0 - bitsf(w[0]));
        else if (op == 3) w[3] = fbits(x1 - bitsf(w[1]));
        else if (op == 4) w[4] = w[2] & 0xffff0000u;
        else if (op == 5) w[5] = w[3] & 0xffff0000u;
        else if (op == 6) w[6] = fbits(bitsf(w[2]) - bitsf(w[4]));
        else if (op == 7) w[7] = fbits(bitsf(w[3]) - bitsf(w[5]));
        else if (op == 8) out.l[0][2 * h2 + d] = __builtin_amdgcn_perm(fbits(x1), fbits(x0), 0x07060302u);
        else if (op == 9) out.l[1][2 * h2 + d] = __builtin_amdgcn_perm(w[3], w[2], 0x07060302u);
        else out.l[2][2 * h2 + d] = __builtin_amdgcn_perm(w[7], w[6], 0x07060302u);
      }
    };
    for (int layer = 1; layer < 8; ++layer) {
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) {
        P[nt] = Q[nt];
        // the finished layer stays in the ACCUMULATOR half of the register file (the epilogue reads each value once,
        // through v_accvgpr_read): the 256 VALU-addressable registers are left to fragments, limbs and temporaries
#pragma unroll
        for (int r = 0; r < 16; ++r) asm volatile("" : "+a"(P[nt][r]));
      }
      init_bias(layer);
      const int t0 = L0S + (layer - 1) * LKS;
      EpiState st;
      if (STORE) hrs = make_rsrc(a.h[layer - 1], hbytes);
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int k = 0; k < 64; ++k)
          if (k < CNT_SP[c]) epi_mop(layer, 0, c, k, Bc, st);
#pragma unroll
      for (int j = 0; j < LKS; ++j) {
        if (j + 1 < LKS)
          kstep(t0 + j, CNT_SP, [&](int c, int k) { epi_mop(layer, j + 1, c, k, Bn, st); });
        else
          kstep(t0 + j, CNT_NONE, no_mop);
      }
    }
    // ---- output of layer 7 ----
    {
      const float* w8l = reinterpret_cast<const float*>(smem + OFF_W8);
      const rsrc_t h7rs = make_rsrc(STORE ? a.h[7] : nullptr, STORE ? hbytes : 0);
#pragma unroll
      for (int nt = 0; nt < 8; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int f0 = 32 * nt + 8 * g + 4 * hh;
          f32x4 v;
#pragma unroll
          for (int k = 0; k < 4; ++k) v[k] = HEAD ? sp_fast(Q[nt][4 * g + k]) : sp_train(Q[nt][4 * g + k]);
          if (HEAD) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(w8l + f0);
            part += v[0] * w[0] + v[1] * w[1] + v[2] * w[2] + v[3] * w[3];
          }
          if (STORE)
            store4(v, h7rs, hvoff + (32 * nt + 8 * g) * 4);
        }
      if (HEAD) {
        const float s = part + __shfl_xor(part, 32) + b8;
        if (hh == 0 && prow_ok) a.sdf[prow * a.lds] = s;
      }
    }
  }
  if (a.guard) {  // the conditional launch ran: count it once and re-arm the guard
    __syncthreads();
    if (tid == 0) {
      __threadfence();
      if (atomicAdd(a.guard + 1, 1u) == gridDim.x - 1) {
        a.guard[1] = 0u;
        atomicAdd(a.guard + 2, 1u);
        __threadfence();
        __hip_atomic_store(a.guard, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

}  // namespace

extern "C" int64_t hold_trunk_r6_pack_bytes(void) { return (int64_t)NSTEP * SLOT; }

static int rmlp_launch(const R6Args& a, bool head, bool store, hipStream_t s) {
  static int n_cu = 0;
  static bool attr_set = false;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return HOLD_E_LAUNCH;
    n_cu = prop.multiProcessorCount;
  }
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)rmlp_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            LDS_BYTES) != hipSuccess ||
        hipFuncSetAttribute((const void*)rmlp_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            LDS_BYTES) != hipSuccess)
      return HOLD_E_LAUNCH;
    attr_set = true;
  }
  const long blocks = (a.P + BPTS - 1) / BPTS;
  const dim3 grid((unsigned)(blocks < n_cu ? blocks : n_cu));
#ifdef HOLD_DEV
  if (const char* v = getenv("HOLD_R6_DMA")) {
    if ((v[0] == '2' || v[0] == '3') && store && !head) {
      static bool set2 = false;
      if (!set2 && (hipFuncSetAttribute((const void*)rmlp_kernel<false, true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        LDS_BYTES) != hipSuccess ||
                    hipFuncSetAttribute((const void*)rmlp_kernel<false, true, 3>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        LDS_BYTES) != hipSuccess))
        return HOLD_E_LAUNCH;
      set2 = true;
      if (v[0] == '2') hipLaunchKernelGGL((rmlp_kernel<false, true, 2>), grid, dim3(256), LDS_BYTES, s, a);
      else hipLaunchKernelGGL((rmlp_kernel<false, true, 3>), grid, dim3(256), LDS_BYTES, s, a);
      return hipGetLastError() == hipSuccess ? HOLD_OK : HOLD_E_LAUNCH;
    }
    if (v[0] == '1' && head && !store) {
      static bool set1 = false;
      if (!set1 && hipFuncSetAttribute((const void*)rmlp_kernel<true, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       LDS_BYTES) != hipSuccess)
        return HOLD_E_LAUNCH;
      set1 = true;
      hipLaunchKernelGGL((rmlp_kernel<true, false, 1>), grid, dim3(256), LDS_BYTES, s, a);
      return hipGetLastError() == hipSuccess ? HOLD_OK : HOLD_E_LAUNCH;
    }
  }
#endif
  if (head && !store)
    hipLaunchKernelGGL((rmlp_kernel<true, false>), grid, dim3(256), LDS_BYTES, s, a);
  else if (store && !head)
    hipLaunchKernelGGL((rmlp_kernel<false, true>), grid, dim3(256), LDS_BYTES, s, a);
  else
    return HOLD_E_ARG;
  return hipGetLastError() == hipSuccess ? HOLD_OK : HOLD_E_LAUNCH;
}

// SDF-only query of the sampler (the contract of hold_fused_sdf_x6 with the register-resident trunk).
extern "C" int hold_fused_sdf_r6_if(const float* xc, int32_t ldx, int64_t P, const void* wpack_r6, const float* bias,
                                    const float* w8, const float* b8, const float* barf_w, float* sdf, int32_t ld_sdf,
                                    uint32_t* guard, hold_stream_t st) {
  if (!xc || !wpack_r6 || !bias || !w8 || !b8 || !sdf || ldx < 3 || ld_sdf < 1 || P < 0) return HOLD_E_ARG;
  if (((uintptr_t)wpack_r6 & 15) || ((uintptr_t)w8 & 15) || ((uintptr_t)bias & 15) || ((uintptr_t)guard & 3)) return HOLD_E_ARG;
  if (P == 0) return HOLD_OK;
  R6Args a = {};
  a.xc = xc; a.ldx = ldx; a.P = (long)P; a.wpack = (const char*)wpack_r6; a.bias = bias; a.w8 = w8; a.b8 = b8;
  a.barf = barf_w; a.sdf = sdf; a.lds = ld_sdf; a.guard = guard;
  return rmlp_launch(a, true, false, (hipStream_t)st);
}
extern "C" int hold_fused_sdf_r6(const float* xc, int32_t ldx, int64_t P, const void* wpack_r6, const float* bias,
                                 const float* w8, const float* b8, const float* barf_w, float* sdf, int32_t ld_sdf,
                                 hold_stream_t st) {
  return hold_fused_sdf_r6_if(xc, ldx, P, wpack_r6, bias, w8, b8, barf_w, sdf, ld_sdf, nullptr, st);
}

// Training forward trunk: h[l] [P][ldh] (l = 0..7) = softplus outputs of lin0..lin7; columns 217..255 of h[3] receive
// the embedding (the skip concat of shape_net.py:122-123, 1/sqrt2 folded into the packed lin4).
extern "C" int hold_trunk_r6_if(const float* xc, int32_t ldx, int64_t P, const void* wpack_r6, const float* bias,
                                const float* barf_w, float* const* h, int32_t ldh, uint32_t* guard, hold_stream_t st) {
  if (!xc || !wpack_r6 || !bias || !h || ldx < 3 || ldh < 256 || (ldh & 3) || P < 0) return HOLD_E_ARG;
  if (((uintptr_t)wpack_r6 & 15) || ((uintptr_t)bias & 15) || ((uintptr_t)guard & 3)) return HOLD_E_ARG;
  R6Args a = {};
  for (int l = 0; l < 8; ++l) {
    if (!h[l] || ((uintptr_t)h[l] & 15)) return HOLD_E_ARG;
    a.h[l] = h[l];
  }
  if (P == 0) return HOLD_OK;
  if (((uint64_t)P + 128) * (uint64_t)ldh * 4 >= (1ull << 32)) return HOLD_E_ARG;  // 32-bit buffer offsets: split by rows
  a.xc = xc; a.ldx = ldx; a.P = (long)P; a.wpack = (const char*)wpack_r6; a.bias = bias; a.barf = barf_w; a.ldh = ldh;
  a.guard = guard;
  return rmlp_launch(a, false, true, (hipStream_t)st);
}
extern "C" int hold_trunk_r6(const float* xc, int32_t ldx, int64_t P, const void* wpack_r6, const float* bias,
                             const float* barf_w, float* const* h, int32_t ldh, hold_stream_t st) {
  return hold_trunk_r6_if(xc, ldx, P, wpack_r6, bias, barf_w, h, ldh, nullptr, st);
}
