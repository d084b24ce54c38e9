// Register-resident BACKWARD sweeps of the ImplicitNet trunk in the TWO-LIMB fp16 arithmetic "f16x3" (gfx950): the kernel of
// csrc/rchain.hip -- the descending sweeps (d sdf / d a_l of the normal path, the first-order backward) and the ascending
// second-order sweep that torch.autograd derives from ImplicitNet.forward (code/src/networks/shape_net.py:84-130) under
// create_graph=True (code/src/engine/volsdf_utils.py:71-96); one wave per SIMD owns 32 points for the whole chain, a layer's
// accumulators become the next layer's MFMA B operand in registers, side inputs and results move as whole 128-byte lines
// through swizzled LDS tiles, a pair-granular weight ring -- with HALF the matrix instructions per product (see csrc/rmlp_h3.hip):
//
//   x = hi + lo,  hi = RN_f16(s x),  lo = RN_f16(s x - hi);   w x ~ hi_w hi_x + hi_w lo_x + lo_w hi_x   on v_mfma_f32_32x32x16_f16
//
//   RC_DSP   v_{l-1} = (M_j v_l) * sp'(aux1_j) [+ aux2_j]                                        7 layers, input v_7 [P][256]
//   RC_DBWD  tb = M_j vb ; out_j = tb * sp'(aux1_j) ; out2_j = 100 tb aux2_j (1 - sp'(aux1_j))   8 layers, input [P][40]
//
// SCALES.  The A operand (weights) scales statically, per matrix, at pack time (s_w[j]: max |M_j| s_w in [2^13, 2^14); c3[j] =
// 1 / s_w[j] comes with the stream).  The B operand is a running loss cotangent -- 1e-9 or 1e+2, heavy-tailed along a ray
// (compositing weights span thirty orders of magnitude), changing from layer to layer INSIDE the kernel.  An MFMA column is a
// POINT (lane li of both lane halves), and the product is linear in it, so every point carries ITS OWN power-of-two scale
// 2^k: the lane multiplies the 8 values it splits per k step by it and the epilogue of the layer un-scales its 128 outputs by
// c3[j] 2^-k (exact).  k for layer l + 1 cannot be known before layer l + 1's values exist (they are finished only by layer l's
// last k step, and the additive side input a2 arrives tile by tile after that), so it is PREDICTED: the exact maximum of the
// point's values in layer l (one v_max3_f32 per value pair, both lane halves combined at the layer boundary) is put at [2^6, 2^7)
// -- 2^9 of headroom up to fp16's largest value, and a point whose values shrink by 2^9 still keeps fp32-class accuracy
// relative to its own maximum (absolute 2^-25 of the scaled value: the dot products it enters are dominated by their largest
// terms).  The chain input's maximum is exact (its rows are in registers before the first k step).  OVERFLOW GUARD as in
// rmlp_h3.hip: the same running maximum says when a scaled value left fp16's range (a point whose cotangent grew 1000 x in one
// layer: an a2 far above the running value can do that); the launch then sets the caller's guard word and the entry point's
// conditional launch of the f32x6 kernel (hold_chain_r6_if) recomputes it -- no host read, no silent infinity.
//
// Per k step: 16 KiB of weight limbs (24 in rchain.hip), 24 MFMAs (48), a limb split of 6 instructions per two values incl.
// scale and maximum (11), 16 fragment reads per wave (24).  Weight ring: units of one pair = 2 n-tiles x 2 limbs = 4 KiB, 8
// units = 32 KiB.  The ascending sweep reads the stream of hold_trunk_h3 (layer 0 = FOUR k steps, K = 64 zero-padded: 464 units
// per block, a multiple of the ring -- no phase flip as in rchain.hip); its fourth input k step gets ZERO limbs.
// LDS: DSP 80 KiB, DSP + a2 112 KiB, DBWD 128 KiB.  Roofline: fp16 MFMA pipe at 3 limb products per product; HBM bytes per point
// and layer as rchain.hip (DSP 2 KiB, DSP + a2 3 KiB, DBWD 4 KiB) -- the HBM floor is now the larger one for all three.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/hold_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int NW = 4, BPTS = 32 * NW, PIECE = 1024, UNIT = 4 * PIECE, SLOT = 4 * UNIT, TILE = 4 * PIECE;
enum { RC_DSP = 1, RC_DBWD = 2 };

struct RCArgs {
  long P;
  const char* wpack;      // DSP: 7 x 16 k steps; DBWD: 4 + 7 x 16 (the stream of hold_trunk_h3)
  const float* c3;        // [L] 1 / s_w of the chain layers (device)
  uint32_t* guard;        // [0] set to 1 when a scaled value left fp16's range (null: unreported)
  const float* in;        // DSP: v_7 [P][ld_in >= 256]; DBWD: [P][ld_in >= 40]
  int ld_in, ld;
  const float* aux1[8];
  const float* aux2[8];
  float* out[8];
  float* out2[8];
};

__device__ __forceinline__ uint32_t fbits(float x) { return __builtin_bit_cast(uint32_t, x); }
__device__ __forceinline__ float bitsf(uint32_t x) { return __builtin_bit_cast(float, x); }

typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const float* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, p ? bytes : 0u, 0x00020000);
}
// column offset as an instruction immediate, soffset = 0 (see rmlp.hip: the store-data hazard hipcc assumes away for
// SGPR soffsets)
__device__ __forceinline__ void store4(const f32x4& v, rsrc_t rs, uint32_t voff) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, voff, 0, 0);
  asm volatile("s_nop 3");
}

struct Limbs { u32x4 l[2]; };  // two f16x8 B fragments (hi, lo), as dwords (dword d = elements 2 d, 2 d + 1)

// round-to-nearest pack of two fp32 values into one dword of f16 (v_cvt_pk_f16_f32)
__device__ __forceinline__ uint32_t cvt_pk(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
}
// x - (float) half SEL of the packed f16 dword hi: ONE v_fma_mix_f32 (exact: hi has 11 significant bits inside x's 24)
template <int SEL>
__device__ __forceinline__ float resid(uint32_t hi, float x) {
  float r;
  if (SEL == 0)
    asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hi), "v"(x));
  else
    asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hi), "v"(x));
  return r;
}
// the whole split of one dword (two values, ALREADY scaled) -- the unscheduled paths (chain input rows)
__device__ __forceinline__ void split_dword(float x0, float x1, Limbs& out, int c, float& mx) {
  asm volatile("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(mx) : "v"(x0), "v"(x1));
  uint32_t hi = cvt_pk(x0, x1);
  asm volatile("" : "+v"(hi));
  uint32_t lo = cvt_pk(resid<0>(hi, x0), resid<1>(hi, x1));
  asm volatile("" : "+v"(lo));
  out.l[0][c] = hi;
  out.l[1][c] = lo;
}
// per-point scale bookkeeping: k = current exponent (values are multiplied by 2^k), mx = exact maximum of the scaled
// magnitudes seen since the last update (both lane halves hold the same point: combined here).  Returns the exponent for the
// NEXT layer -- the unscaled maximum at [2^6, 2^7) -- and reports an overflow of this one.
__device__ __forceinline__ int next_scale(float& mx, int k, bool& ovf) {
  const float m = fmaxf(mx, __shfl_xor(mx, 32));
  ovf = ovf || (m >= 65504.f);
  const int e = (int)((fbits(m) >> 23) & 0xffu);
  int kn = k + (127 + 6 - e);
  kn = kn > 96 ? 96 : (kn < -96 ? -96 : kn);  // 2^-k c3 and 2^k x stay normal fp32 numbers
  mx = 0.f;
  return (m > 0.f && e != 0 && e != 255) ? kn : k;
}
__device__ __forceinline__ float pow2f(int k) { return bitsf((uint32_t)(127 + k) << 23); }

// softplus'(a) recovered from h = softplus(a): 1 - e^{-100 h} (series where the subtraction would cancel); e out
__device__ __forceinline__ float dsp_e(float h, float& e) {
  const float x = 100.0f * h;
  e = __builtin_amdgcn_exp2f(-144.26950408889634f * h);
  const float ser = x * (1.0f - x * (0.5f - x * (0.16666667f - 0.041666668f * x)));
  return (x < 0.05f) ? ser : 1.0f - e;
}

// one 1 KiB LDS-DMA piece: lane L -> 16 bytes from src + voff(L) to LDS byte dst + 16 L (inline assembly: hipcc models
// the builtin as a FLAT access that may touch LDS and puts lgkmcnt(0) in front of every later ds_read, see rmlp.hip; M0 is
// written by every statement that reads it, nothing else uses it)
__device__ __forceinline__ void dma_piece(const char* src, uint32_t voff, uint32_t dst) {
  asm volatile(
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %0, %1"
      :
      : "v"(voff), "s"(src), "s"(dst)
      : "memory");
}

// ... of a SIDE tile: whole 128-byte lines that this launch reads exactly once, requested with the non-temporal hint -- they do not
// displace the weight stream (read by every workgroup) from the L2.  Same-box A/B of the headline (round 6, GPU call 28): DBWD 10.6 ->
// 10.2 ms, DSP + a2 7.8 -> 7.4, DSP 6.27 -> 6.20, the background's sweep 2.17 -> 2.03.  The hint is for once-touched whole lines only:
// on hold_gemm_h3's quarter-line input fragments (each line touched by four requests) it costs 20 %, and on result stores it doubles
// every kernel that stores 32-byte row fragments -- those rely on the L2 merging a line's four partial writes (GPU call 27)
__device__ __forceinline__ void dma_side(const char* src, uint32_t voff, uint32_t dst) {
  asm volatile(
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %0, %1 nt"
      :
      : "v"(voff), "s"(src), "s"(dst)
      : "memory");
}
#define RC_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

template <int MODE, bool A2, int DIST_>
struct RCfg {
  static constexpr bool DB = MODE == RC_DBWD;
  static constexpr int NAUX = (DB || A2) ? 2 : 1;   // side matrices per layer
  static constexpr int NOUT = DB ? 2 : 1;           // result matrices per layer
  static constexpr int L = DB ? 8 : 7;              // chain layers
  static constexpr int L0 = DB ? 4 : 16;            // k steps of chain layer 0 (DBWD: K = 64, the stream of hold_trunk_h3)
  static constexpr int NST = L0 + 16 * (L - 1);     // k steps per block of points
  static constexpr int DIST = DIST_;                // weight groups requested DIST rendezvous ahead: 1 or 3 (LDS budget)
  static constexpr int NU = 4 + 4 * DIST;           // weight ring in units: 8 or 16 -- a power of two dividing the 64 units of
                                                    // a 16-step layer, so every ring address is an instruction immediate
  static constexpr int SLOT_T = NAUX * TILE;        // one side slot: the tile of every side matrix
  static constexpr int WREG = 2 * SLOT_T + NOUT * TILE;  // wave region: two side slots + the result tile(s)
  static constexpr int OFF_SIDE = NU * UNIT;
  static constexpr int OUT_OFF = 2 * SLOT_T;
  static constexpr int LDS = OFF_SIDE + NW * WREG;
  static_assert(LDS <= 160 * 1024, "LDS budget");
};

// ABL (developer build, results garbage): 1 = the side tiles of every block are read from block 0's rows (they stay in L2): what the
// HBM latency of the side requests costs; 2 = additionally no result stores
// SKIP_OUT: first special output column of the skip layer (217: the foreground nets; 172: the background net)
template <int MODE, bool A2, int DIST_, int ABL = 0, int SKIP_OUT = 217>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void rsweep_h3_kernel(RCArgs a) {
  using C = RCfg<MODE, A2, DIST_>;
  constexpr bool DB = C::DB;
  constexpr int NAUX = C::NAUX, NOUT = C::NOUT, L = C::L, L0 = C::L0, NST = C::NST, NSU = 4 * NST;
  constexpr int DIST = C::DIST, NU = C::NU, SLOT_T = C::SLOT_T, WREG = C::WREG, OFF_SIDE = C::OFF_SIDE, OUT_OFF = C::OUT_OFF;
  // Rendezvous waits (DIST 1: the group needed now was requested at the previous rendezvous), from the per-step VMEM queue of THIS
  // file's micro-operation placement (in-order retirement):
  //   even step: [rendezvous] W x 4 (gaps 12..15), S x 4 NAUX (gaps 18..23)  |  odd step: [rendezvous] W x 4, stores x 4 NOUT (gaps >= 16)
  // At an odd step's rendezvous the even step's side pieces are younger than the weights it needs; at an even step's the odd
  // step's stores are.  (rchain.hip issues the stores BEFORE the odd step's weight pieces: its constants differ.)  The first
  // 16-step layer of a block has no stores in front of its first rendezvous: a full wait stands in front of that layer.
  //   DIST 3 (requested three rendezvous ago; 16-unit ring).  Even step g:  .. W(g-3) st(g-3) | W(g-2) S(g-2) | W(g-1) st(g-1) | --
  //           S(g-2) is read from step g + 1 on and must be forced here: only W(g-1) and st(g-1) stay in flight.  Odd step g:
  //           .. W(g-3) S(g-3) | W(g-2) st(g-2) | W(g-1) S(g-1) | -- everything behind W(g-3) may stay in flight.
  constexpr int NW_EVEN = DIST == 1 ? 4 * NOUT : 4 + 4 * NOUT;
  constexpr int NW_ODD = DIST == 1 ? 4 * NAUX : 8 + 8 * NAUX + 4 * NOUT;
  constexpr int NW_IN0 = 8 * NAUX;  // DBWD chain layer 0, step 0: the two side tiles just requested stay in flight
  static_assert(NW_ODD < 64, "vmcnt field");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is3 = wave == 3;
  const int hh = lane >> 5, li = lane & 31;
  const int r8 = lane >> 3, p8 = lane & 7;
  const uint32_t lane16 = lane * 16;
  const char* ring_lane = smem + lane * 16;
  const uint32_t side_dst0 = (uint32_t)(OFF_SIDE + wave * WREG);
  const int fl = ((li >> 1) & 7) ^ ((li & 1) << 2);
  // fragment (q, e4) of this lane's row: byte (frag0 ^ (64 q + 32 e4)) of a tile (the region base is 1 KiB aligned)
  const uint32_t frag0 = (uint32_t)(OFF_SIDE + wave * WREG + li * 128 + ((hh ^ fl) << 4));
  const uint32_t rb0 = (uint32_t)(OFF_SIDE + wave * WREG + OUT_OFF) + lane16;  // read-back: lane-linear
  const int f0 = (r8 >> 1) ^ ((r8 & 1) << 2);  // f(8 i + r8) = f0 ^ ((i & 1) << 2)
  const uint32_t nbytes = (uint32_t)(a.P * a.ld * 4);

  // ---- weight ring addressing.  c = unit index relative to the block's first unit; ring slot = c % NU.  Both streams are a
  // multiple of the ring long (DSP 448 units per block, DBWD 464): slot offsets are compile-time constants.
  static_assert(NSU % NU == 0, "the ring phase must not change from block to block");
  auto ring_off = [&](int c) -> uint32_t { return (uint32_t)((c & (NU - 1)) * UNIT); };
  // this wave's part of the group requested at a rendezvous: units 4 (g + DIST) + 3 .. + 6 -- wave 3 the first (the last
  // unit of step g + DIST), waves 0..2 the units 0..2 of step g + DIST + 1 (an aligned group: + wave never wraps)
  const uint32_t wvU = (uint32_t)(wave * UNIT);
  const long wv_src = (long)(is3 ? 3 : 4 + wave) * UNIT;  // same split on the source side

  f32x16 P[8], Q[8];
  u32x4 A[2][4];
  Limbs Bc, Bn;
  f32x4 rbv[NOUT][4];
  // per-point scale state (see the header): kB = exponent of the B operand's scale in the layer being accumulated, sB = 2^kB,
  // ysc = what un-scales the FINISHED layer's accumulators (c3 of its weights x 2^-k of its operand), mx = running maximum
  int kB = 0;
  float sB = 1.f, ysc = 1.f, mx = 0.f;
  bool ovf = false;

  auto zero_q = [&]() {
#pragma unroll
    for (int nt = 0; nt < 8; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) Q[nt][r] = 0.f;
  };

  // prologue: the units a first rendezvous expects to have been requested, 0 .. 4 DIST + 2 (same wave split)
#pragma unroll
  for (int x = 0; x <= 4 * DIST + 2; ++x)
    if ((x & 3) == wave) {
#pragma unroll
      for (int i = 0; i < 4; ++i) dma_piece(a.wpack + x * UNIT + i * PIECE, lane16, (uint32_t)(x * UNIT + i * PIECE));
    }
  int first = 1;

  for (long blk = blockIdx.x; blk * BPTS < a.P; blk += gridDim.x) {
    const long row = blk * BPTS + wave * 32 + li;  // this lane's point
    const uint32_t st_off = (uint32_t)((row * a.ld + 4 * hh) * 4);  // last layer's exposed epilogue (row fragments)
    // full-line pieces: lane -> (row 8 i + r8, chunk p8 ^ f): DMA sources clamped to the last row, stores range-checked
    uint32_t dvoff[4], svoff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long r = blk * BPTS + wave * 32 + 8 * i + r8;
      const long cr = r < a.P ? r : a.P - 1;
      const int cs = p8 ^ f0 ^ ((i & 1) << 2);
      dvoff[i] = (uint32_t)(((ABL >= 1 ? (long)(wave * 32 + 8 * i + r8) : cr) * a.ld + 4 * cs) * 4);
      svoff[i] = ABL >= 2 ? 0xfffffff0u : (uint32_t)((r * a.ld + 4 * cs) * 4);
    }
    auto ring_rd = [&](int c) -> const char* { return ring_lane + (c & (NU - 1)) * UNIT; };

    // ---- chain input ----
    u32x4 in6[DB ? 6 : 1];
    float amax = 0.f;  // exact maximum of this lane's share of the chain input
    {
      const rsrc_t irs = make_rsrc(a.in, (uint32_t)(a.P * a.ld_in * 4));
      if (!DB) {  // v_7 rows into the accumulator layout: P[nt][4 g + k] = in[row][32 nt + 8 g + 4 hh + k]
        const uint32_t ioff = (uint32_t)((row * a.ld_in + 4 * hh) * 4);
#pragma unroll
        for (int nt = 0; nt < 8; ++nt)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(irs, ioff + (32 * nt + 8 * g) * 4, 0, 0);
#pragma unroll
            for (int k = 0; k < 4; ++k) P[nt][4 * g + k] = bitsf(v[k]);
            amax = fmaxf(fmaxf(amax, fmaxf(fabsf(bitsf(v[0])), fabsf(bitsf(v[1])))), fmaxf(fabsf(bitsf(v[2])), fabsf(bitsf(v[3]))));
          }
      } else {  // [P][40] rows, natural k order 16 j + 8 hh + e of chain layer 0 (K padded to 48: columns >= 40 are zeros)
        const uint32_t ioff = (uint32_t)((row * a.ld_in + 8 * hh) * 4);
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
          for (int e4 = 0; e4 < 2; ++e4) {
            u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(irs, ioff + (16 * j + 4 * e4) * 4, 0, 0);
            if (j == 2) {  // columns 40 .. 47 (lane half 1) lie beyond the row: the packed weights are zero there, the
              // operand must be finite
#pragma unroll
              for (int k = 0; k < 4; ++k) v[k] = hh ? 0u : v[k];
            }
            in6[2 * j + e4] = v;
            amax = fmaxf(fmaxf(amax, fmaxf(fabsf(bitsf(v[0])), fabsf(bitsf(v[1])))), fmaxf(fabsf(bitsf(v[2])), fabsf(bitsf(v[3]))));
          }
      }
    }
    // the scale of the chain input: its exact maximum (both lane halves of the point) at [2^6, 2^7)
    {
      bool dummy = false;
      mx = amax;
      kB = next_scale(mx, 0, dummy);
      sB = pow2f(kB);
      ysc = 1.f;
    }
    if (first) {
      RC_WAIT_VM(0);
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int i = 0; i < 4; ++i) A[0][i] = *reinterpret_cast<const u32x4*>(ring_lane + i * PIECE);
      first = 0;
    }

    const float *lo1 = a.aux1[0], *lo2 = NAUX == 2 ? a.aux2[0] : nullptr, *hi1 = lo1, *hi2 = lo2;

    // piece i of side tile tn of matrix mat -> LDS tile at dst
    auto dma_tile_piece = [&](const float* mat, int tn, int i, uint32_t dst) {
      dma_side(reinterpret_cast<const char*>(mat + 32 * tn), dvoff[i], dst + i * PIECE);
    };

    // One k step with an EXPLICIT schedule (rmlp_h3.hip:kstep): 4 groups ("pairs") x 6 MFMAs (hi hi, hi lo, lo hi for two
    // n-tiles), behind every MFMA a fixed slice of the rest -- one fragment read for the next group (gaps 0..3); in the group
    // behind the rendezvous this wave's four weight pieces (gaps 0..3); with side_req (the EVEN steps of a 16-step layer), in
    // the last group, BEHIND them in the queue, the pieces of side tile jp / 2 + 2 (tiles 8, 9 = tiles 0, 1 of the next epilogue
    // layer) into the slot tile jp / 2 has just left; and n / 24 micro-operations of the next step's epilogue -- closed by a
    // full scheduling barrier.  The n micro-operations of a step are ONE list spread evenly over the 24 gaps (rchain.hip gives
    // each group its own stage: with 6 instead of 12 MFMAs per group the first group's gaps then carry 12 VALU instructions each
    // and the last two groups 2 -- first hardware run of this file: 2 900 cycles per k step for 768 cycles of MFMA).
    // c0 = this step's first unit relative to the block; wsrc = this wave's share of the group requested here; nwait = the
    // rendezvous' vmcnt (one of the NW_* constants).
    auto kstep = [&](int c0, const char* wsrc, int jp, bool side_req, int nwait, int n, auto&& mop) {
      const uint32_t wdst = is3 ? ring_off(c0 + 4 * DIST + 3) : ring_off(c0 + 4 * DIST + 4) + wvU;
      const int tn = (jp >> 1) + 2;
      const float* s1 = tn < 8 ? lo1 : hi1;
      const float* s2 = tn < 8 ? lo2 : hi2;
      const uint32_t sd = side_dst0 + ((jp >> 1) & 1) * SLOT_T;
#pragma unroll
      for (int pair = 0; pair < 4; ++pair) {
        if (pair == 2) {
          // rendezvous: units .. c0 + 6 have landed in every wave; the slots of units .. c0 + 2 are free (the fragments of
          // unit c0 + 2 were read during pair 1: lgkmcnt(0) makes that true for every wave behind the barrier)
          if (nwait == NW_EVEN) RC_WAIT_VM(NW_EVEN);
          else if (nwait == NW_ODD) RC_WAIT_VM(NW_ODD);
          else if (nwait == NW_IN0) RC_WAIT_VM(NW_IN0);
          else RC_WAIT_VM(0);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
        }
        const char* rd = ring_rd(c0 + pair + 1);
#pragma unroll
        for (int m = 0; m < 6; ++m) {
          const int pr = m >> 1, tt = m & 1;           // (w limb, act limb): hi hi, hi lo, lo hi
          const int wl = pr == 2 ? 1 : 0, al = pr == 1 ? 1 : 0;
          Q[2 * pair + tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A[pair & 1][2 * tt + wl]),
                                                                   __builtin_bit_cast(f16x8, Bc.l[al]), Q[2 * pair + tt],
                                                                   0, 0, 0);
          if (m < 4) A[(pair + 1) & 1][m] = *reinterpret_cast<const u32x4*>(rd + m * PIECE);
          if (pair == 2 && m < 4) dma_piece(wsrc + m * PIECE, lane16, wdst + m * PIECE);
          if (pair == 3 && side_req) {
            // 4 NAUX side pieces over the six gaps: matrix 1's piece m in gaps 0..3, matrix 2's piece m - 2 in gaps 2..5
            if (m < 4) dma_tile_piece(s1, tn & 7, m, sd);
            if (NAUX == 2 && m >= 2) dma_tile_piece(s2, tn & 7, m - 2, sd + TILE);
          }
          const int G = 6 * pair + m;
#pragma unroll
          for (int u = 0; u < 8; ++u) {  // constant trip count (the slice bounds fold once pair and m are unrolled)
            const int k = n * G / 24 + u;
            if (k < n * (G + 1) / 24) mop(k);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      Bc = Bn;
    };

    // Epilogue micro-operations of k step j of the finished layer in P (chain layer lw, wave-uniform; raw = DSP chain layer
    // 0 is fed by the chain input itself): 8 values, four stages in round-major order (rmlp.hip: consecutive operations
    // independent).  Side values from tile j / 2 in slot (j / 2) & 1, results into the wave's result tile(s).
    //   stage 0: y, h (and a2 / t), x = 100 h, e = exp(-x), the four operations of the small-x series of 1 - e^{-x}
    //   stage 1: sp' = series or 1 - e, y * sp' (+ a2 | and 100 y t e), raw / skip-layer selects              -> r[8] (, r2[8])
    //   stage 2 / 3: limb split of r[0..3] / r[4..7], 16-byte write(s) into the result tile(s)
    rsrc_t ors = make_rsrc(nullptr, 0), ors2 = make_rsrc(nullptr, 0);
    struct EpiState { float y[8], h[8], x2[8], e[8], ser[8], r[8], r2[8]; float xs[2][2], ra[2], rb[2]; uint32_t hi[2]; };
    constexpr int C1 = DB ? 48 : (A2 ? 40 : 32);
    constexpr int C2 = 12 + NOUT;  // limb split of two dwords (6 operations each: scale, maximum, hi, two residuals, lo) + the tile write(s)
    // the step's micro-operation list: [stage 0 (C0) | read-back of the tile finished last step (odd steps: 4 NOUT) | stage 1 (C1) |
    // its stores (odd steps: 4 NOUT) | stage 2 (C2) | stage 3 (C2)]
    constexpr int C0 = 64;
    constexpr int N_EVEN = C0 + C1 + 2 * C2, N_ODD = N_EVEN + 8 * NOUT, N_LAST = 24, N_IN = 4;
    // the hand-counted queue above needs every store of an odd step BEHIND its weight pieces (gaps 12..15)
    static_assert((C0 + 4 * NOUT + C1) * 24 / N_ODD >= 16, "an odd step's stores would mix with its weight pieces in the VMEM queue");
    // the side tile of step jp / 2 is overwritten by the LDS-DMA of the last group (gaps 18..23) of an even step: stage 0 -- the
    // only reader -- must be over by then
    static_assert((C0 * 24 + N_EVEN - 1) / N_EVEN <= 18, "stage 0 would still read the side slot the last group's DMA refills");
    // Every result is PINNED (an empty volatile asm): the instruction selector is free to place pure VALU code anywhere between
    // its operands and its first user, and without the pins it sinks the whole epilogue to the end of the step -- the second
    // build of this file had 22 .. 32 instructions behind each MFMA of the last group and 1 behind the others (rmlp.hip found
    // the same).  Loads are not pinned (a pin would wait for them); v_exp_f32 as a volatile instruction (rmlp_h3.hip).
#define RC_PIN(x) asm volatile("" : "+v"(x))
    auto epi_mop = [&](int j, bool raw, bool skip, int stage, int k, Limbs& out, EpiState& st) {
      const int nt = j >> 1, q = j & 1;
      const int rd = k >> 3, i = k & 7;
      if (stage == 0) {
        if (rd == 0) {
          st.y[i] = P[nt][8 * q + i] * ysc;  // un-scaled: c3 of the finished layer's weights x 2^-k of its operand
          RC_PIN(st.y[i]);
          if (i < 2) {  // the two 16-byte reads of the side fragment(s) FIRST (values 4 i .. 4 i + 3): their first use is eight
            // micro-operations away
            const char* sp = smem + (frag0 ^ (uint32_t)(64 * q + 32 * i)) + (nt & 1) * SLOT_T;
            const f32x4 hv = *reinterpret_cast<const f32x4*>(sp);
#pragma unroll
            for (int v = 0; v < 4; ++v) st.h[4 * i + v] = hv[v];
            if (NAUX == 2) {
              const f32x4 xv = *reinterpret_cast<const f32x4*>(sp + TILE);
#pragma unroll
              for (int v = 0; v < 4; ++v) st.x2[4 * i + v] = xv[v];
            }
          }
        } else if (rd == 1) { st.e[i] = -144.26950408889634f * st.h[i]; RC_PIN(st.e[i]); }
        else if (rd == 2) { float r_; asm volatile("v_exp_f32 %0, %1" : "=v"(r_) : "v"(st.e[i])); st.e[i] = r_; }
        else if (rd == 3) { st.h[i] = 100.0f * st.h[i]; RC_PIN(st.h[i]); }
        else if (rd == 4) { st.ser[i] = fmaf(st.h[i], -0.041666668f, 0.16666667f); RC_PIN(st.ser[i]); }
        else if (rd == 5) { st.ser[i] = fmaf(-st.h[i], st.ser[i], 0.5f); RC_PIN(st.ser[i]); }
        else if (rd == 6) { st.ser[i] = fmaf(-st.h[i], st.ser[i], 1.0f); RC_PIN(st.ser[i]); }
        else { st.ser[i] = st.h[i] * st.ser[i]; RC_PIN(st.ser[i]); }
      } else if (stage == 1) {
        const int f = 16 * j + 8 * (i >> 2) + 4 * hh + (i & 3);  // this value's feature (skip layer: columns 217.. are special)
        if (!DB) {
          if (rd == 0) { st.e[i] = 1.0f - st.e[i]; RC_PIN(st.e[i]); }
          else if (rd == 1) { st.e[i] = (st.h[i] < 0.05f) ? st.ser[i] : st.e[i]; RC_PIN(st.e[i]); }
          else if (rd == 2) { st.r[i] = st.y[i] * st.e[i]; RC_PIN(st.r[i]); }
          else if (A2 && rd == 3) { st.r[i] = st.r[i] + st.x2[i]; RC_PIN(st.r[i]); }
          else {
            float r = raw ? st.y[i] : st.r[i];
            if (j >= SKIP_OUT / 16) r = (skip && f >= SKIP_OUT) ? st.y[i] : r;  // the raw products (d / d skip input) are stored
            st.r[i] = r;
            RC_PIN(st.r[i]);
          }
        } else {
          if (rd == 0) { st.ser[i] = (st.h[i] < 0.05f) ? st.ser[i] : 1.0f - st.e[i]; RC_PIN(st.ser[i]); }
          else if (rd == 1) { st.r[i] = st.y[i] * st.ser[i]; RC_PIN(st.r[i]); }
          else if (rd == 2) { st.r2[i] = 100.0f * st.y[i]; RC_PIN(st.r2[i]); }
          else if (rd == 3) { st.r2[i] = st.r2[i] * st.x2[i]; RC_PIN(st.r2[i]); }
          else if (rd == 4) { st.r2[i] = st.r2[i] * st.e[i]; RC_PIN(st.r2[i]); }
          else if (j >= SKIP_OUT / 16) {  // skip layer: the next input's columns 217.. are the side columns (in aux2, see the header)
            const bool sp_ = skip && f >= SKIP_OUT;
            st.r[i] = sp_ ? st.x2[i] : st.r[i];
            st.r2[i] = sp_ ? 0.f : st.r2[i];
            RC_PIN(st.r[i]);
            RC_PIN(st.r2[i]);
          }
        }
      } else {
        const int h2 = stage - 2;
        if (k >= 12) {
          const float* src = (k == 12) ? st.r : st.r2;
          const f32x4 v = {src[4 * h2], src[4 * h2 + 1], src[4 * h2 + 2], src[4 * h2 + 3]};
          *reinterpret_cast<f32x4*>(smem + (frag0 ^ (uint32_t)(64 * q + 32 * h2)) + OUT_OFF + (k - 12) * TILE) = v;
          return;
        }
        // two-limb fp16 split of r[4 h2 .. 4 h2 + 3] (dwords d = 0, 1), round-major: scale by the point's 2^kB, exact running
        // maximum (overflow guard + the next layer's scale), hi = RN_f16, the two residuals (v_fma_mix_f32), lo = RN_f16
        const int d = k & 1, op = k >> 1;
        if (op == 0) {
          st.xs[d][0] = st.r[4 * h2 + 2 * d] * sB;
          st.xs[d][1] = st.r[4 * h2 + 2 * d + 1] * sB;
          RC_PIN(st.xs[d][0]);
          RC_PIN(st.xs[d][1]);
        } else if (op == 1) {
          asm volatile("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(mx) : "v"(st.xs[d][0]), "v"(st.xs[d][1]));
        } else if (op == 2) {
          uint32_t hi = cvt_pk(st.xs[d][0], st.xs[d][1]);
          asm volatile("" : "+v"(hi));
          st.hi[d] = hi;
          out.l[0][2 * h2 + d] = hi;
        } else if (op == 3) st.ra[d] = resid<0>(st.hi[d], st.xs[d][0]);
        else if (op == 4) st.rb[d] = resid<1>(st.hi[d], st.xs[d][1]);
        else {
          uint32_t lo = cvt_pk(st.ra[d], st.rb[d]);
          asm volatile("" : "+v"(lo));
          out.l[1][2 * h2 + d] = lo;
        }
      }
    };
    // result tile(s) nt: piece i of result o read back lane-linear (stage 0 extras), stored as whole lines (stage 1 extras)
    auto io_mop = [&](int nt, int stage, int x) {
      const int o = x >> 2, i = x & 3;
      if (stage == 0) rbv[o][i] = *reinterpret_cast<const f32x4*>(smem + rb0 + o * TILE + i * PIECE);
      else store4(rbv[o][i], o ? ors2 : ors, svoff[i] + 128 * nt);
    };

    int l0 = 0;
    if (DB) {  // chain layer 0: K = 48 from the input rows in registers, natural k order 16 j + 8 hh + e
      zero_q();
      // side tiles 0 and 1 of the first epilogue layer (consumed from the start of the next layer on)
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          dma_tile_piece(a.aux1[0], tn, i, side_dst0 + tn * SLOT_T);
          dma_tile_piece(a.aux2[0], tn, i, side_dst0 + tn * SLOT_T + TILE);
        }
      auto in_limbs = [&](int j, int c, Limbs& out) {
        if (j >= 3) {  // the padding k step of K = 64 (columns 48..63: zero weights in the stream): ZERO limbs
          out.l[0][c] = 0u;
          out.l[1][c] = 0u;
          return;
        }
        const u32x4 v = in6[2 * j + (c >> 1)];
        split_dword(bitsf(v[2 * (c & 1)]) * sB, bitsf(v[2 * (c & 1) + 1]) * sB, out, c, mx);
      };
#pragma unroll
      for (int c = 0; c < 4; ++c) in_limbs(0, c, Bc);
      const char* w0 = a.wpack + (long)(4 * DIST) * UNIT + wv_src;
#pragma unroll
      for (int j = 0; j < L0; ++j) {
        // rendezvous waits: step 0 leaves the 16 side pieces just requested in flight, the later ones force them (they sit
        // in front of the weight pieces in the queue)
        if (j + 1 < L0)
          kstep(4 * j, w0 + (long)(4 * j) * UNIT, 1, false, j == 0 ? NW_IN0 : 0, N_IN, [&](int c) { in_limbs(j + 1, c, Bn); });
        else
          kstep(4 * j, w0 + (long)(4 * j) * UNIT, 1, false, 0, 0, [&](int) {});
      }
      l0 = 1;
    }

    for (int l = l0; l < L; ++l) {
      // MFMA layer l consumes P through the epilogue of chain layer lw = l - 1 (DSP l = 0: the raw chain input)
      if (DB || l > 0) {
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) P[nt] = Q[nt];
        // the finished layer's accumulators carry s_w[l - 1] 2^kB; the next operand's scale from this one's exact maximum
        ysc = a.c3[l - 1] * pow2f(-kB);
        kB = next_scale(mx, kB, ovf);
        sB = pow2f(kB);
      }
#pragma unroll
      for (int nt = 0; nt < 8; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) asm volatile("" : "+a"(P[nt][r]));
      zero_q();
      // NW_EVEN counts on an odd step's stores between the weights it waits for and the youngest request: the first 16-step
      // layer of a block has none in front of it (a new block / DBWD's chain layer 0) -- everything requested so far lands here
      if (l == l0) RC_WAIT_VM(0);
      const int lw = l - 1;
      const bool raw = !DB && l == 0;
      const bool skip = lw == 3;
      const int lwc = lw < 0 ? 0 : lw, lhc = l < L - 1 ? l : L - 2;
      lo1 = a.aux1[lwc];
      hi1 = a.aux1[lhc];
      if (NAUX == 2) {
        lo2 = a.aux2[lwc];
        hi2 = a.aux2[lhc];
      }
      ors = make_rsrc(raw ? nullptr : a.out[lwc], nbytes);
      if (DB) ors2 = make_rsrc(a.out2[lwc], nbytes);
      const int t0 = DB ? L0 + 16 * (l - 1) : 16 * l;
      // this wave's share of the group requested at step j of this layer: units 4 (t0 + j + DIST) + 3 (wave 3) / + 4 + wave;
      // past the end of the block's stream it wraps to the start (the next block reads the same weights): in the LAST layer
      // for j + DIST >= 16 (every wave) and for j + DIST == 15 (waves 0..2)
      const char* wl = a.wpack + (long)(4 * (t0 + DIST)) * UNIT + wv_src;
      const bool last = l == L - 1;
      const char* wlw = last ? wl - (long)NSU * UNIT : wl;
      const char* wlm = (last && !is3) ? wl - (long)NSU * UNIT : wl;
      constexpr int CB = DB ? 4 * L0 : 0;  // relative unit index of the layer's first unit (a multiple of the ring)
      EpiState st;
      // flat micro-operation index -> (stage, index in the stage); io = the read-back / store extras of the odd steps
      auto epi_flat = [&](int j, int k, bool odd, Limbs& out, EpiState& st_) {
        const int x = odd ? 4 * NOUT : 0;
        if (k < C0) epi_mop(j, raw, skip, 0, k, out, st_);
        else if (k < C0 + x) io_mop((j - 1) >> 1, 0, k - C0);
        else if (k < C0 + x + C1) epi_mop(j, raw, skip, 1, k - C0 - x, out, st_);
        else if (k < C0 + 2 * x + C1) io_mop((j - 1) >> 1, 1, k - C0 - x - C1);
        else if (k < C0 + 2 * x + C1 + C2) epi_mop(j, raw, skip, 2, k - C0 - 2 * x - C1, out, st_);
        else epi_mop(j, raw, skip, 3, k - C0 - 2 * x - C1 - C2, out, st_);
      };
#pragma unroll
      for (int k = 0; k < N_EVEN; ++k) epi_flat(0, k, false, Bc, st);
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const char* ws = (j + DIST >= 16 ? wlw : (j + DIST == 15 ? wlm : wl)) + (long)(4 * j) * UNIT;
        if (j == 15)  // tile 7, no epilogue: one slot per gap -- the read-back in gaps 0 .., the stores in gaps 16 .. (behind the weights)
          kstep(CB + 4 * j, ws, j, false, NW_ODD, N_LAST, [&](int k) {
            if (k < 4 * NOUT) io_mop(7, 0, k);
            else if (k >= 16 && k - 16 < 4 * NOUT) io_mop(7, 1, k - 16);
          });
        else if (j & 1)  // (the extras of MFMA step j concern tile j / 2 = ((j + 1) - 1) / 2)
          kstep(CB + 4 * j, ws, j, false, NW_ODD, N_ODD, [&](int k) { epi_flat(j + 1, k, true, Bn, st); });
        else
          kstep(CB + 4 * j, ws, j, true, NW_EVEN, N_EVEN, [&](int k) { epi_flat(j + 1, k, false, Bn, st); });
      }
    }

    // ---- epilogue of the last chain layer (exposed): side rows by ordinary buffer loads ----
    {
      const float yl = a.c3[L - 1] * pow2f(-kB);
      (void)next_scale(mx, kB, ovf);  // the last layer's operand: overflow check only
      const rsrc_t a1 = make_rsrc(a.aux1[L - 1], nbytes);
      const rsrc_t a2 = make_rsrc(NAUX == 2 ? a.aux2[L - 1] : nullptr, nbytes);
      const rsrc_t o1 = make_rsrc(a.out[L - 1], nbytes), o2 = make_rsrc(DB ? a.out2[L - 1] : nullptr, nbytes);
#pragma unroll
      for (int nt = 0; nt < 8; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const uint32_t co = (32 * nt + 8 * g) * 4;
          const u32x4 hv = __builtin_amdgcn_raw_buffer_load_b128(a1, st_off + co, 0, 0);
          u32x4 xv = {0u, 0u, 0u, 0u};
          if (NAUX == 2) xv = __builtin_amdgcn_raw_buffer_load_b128(a2, st_off + co, 0, 0);
          f32x4 r, r2;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float y = Q[nt][4 * g + k] * yl;
            float e;
            const float s = dsp_e(bitsf(hv[k]), e);
            if (!DB) {
              r[k] = y * s + (A2 ? bitsf(xv[k]) : 0.f);
              r2[k] = 0.f;
            } else {
              r[k] = y * s;
              r2[k] = 100.0f * y * bitsf(xv[k]) * e;
            }
          }
          store4(r, o1, st_off + co);
          if (DB) store4(r2, o2, st_off + co);
        }
    }
  }
  if (a.guard && ovf) __hip_atomic_store(a.guard, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace

extern "C" int64_t hold_chain_h3_pack_bytes(void) { return (int64_t)(7 * 16) * SLOT; }

template <int MODE, bool A2, int DIST, int SKIP = 217>
static int rsweep_h3_launch(const RCArgs& a, hipStream_t s) {
#ifdef HOLD_DEV
  if (const char* v = getenv("HOLD_H3C_ABL")) {
    constexpr int lds_ = RCfg<MODE, A2, DIST>::LDS;
    const long blocks_ = (a.P + BPTS - 1) / BPTS;
    int dev_ = 0;
    hipDeviceProp_t prop_;
    if (hipGetDevice(&dev_) != hipSuccess || hipGetDeviceProperties(&prop_, dev_) != hipSuccess) return HOLD_E_LAUNCH;
    const dim3 grid_((unsigned)(blocks_ < prop_.multiProcessorCount ? blocks_ : prop_.multiProcessorCount));
    if (v[0] == '1') {
      if (hipFuncSetAttribute((const void*)rsweep_h3_kernel<MODE, A2, DIST, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_) != hipSuccess) return HOLD_E_LAUNCH;
      hipLaunchKernelGGL((rsweep_h3_kernel<MODE, A2, DIST, 1>), grid_, dim3(256), lds_, s, a);
      return hipGetLastError() == hipSuccess ? HOLD_OK : HOLD_E_LAUNCH;
    }
    if (v[0] == '2') {
      if (hipFuncSetAttribute((const void*)rsweep_h3_kernel<MODE, A2, DIST, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_) != hipSuccess) return HOLD_E_LAUNCH;
      hipLaunchKernelGGL((rsweep_h3_kernel<MODE, A2, DIST, 2>), grid_, dim3(256), lds_, s, a);
      return hipGetLastError() == hipSuccess ? HOLD_OK : HOLD_E_LAUNCH;
    }
  }
#endif
  constexpr int lds = RCfg<MODE, A2, DIST>::LDS;
  static int n_cu = 0;
  static bool attr_set = false;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return HOLD_E_LAUNCH;
    n_cu = prop.multiProcessorCount;
  }
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)rsweep_h3_kernel<MODE, A2, DIST, 0, SKIP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      return HOLD_E_LAUNCH;
    attr_set = true;
  }
  const long blocks = (a.P + BPTS - 1) / BPTS;
  hipLaunchKernelGGL((rsweep_h3_kernel<MODE, A2, DIST, 0, SKIP>), dim3((unsigned)(blocks < n_cu ? blocks : n_cu)), dim3(256), lds, s, a);
  return hipGetLastError() == hipSuccess ? HOLD_OK : HOLD_E_LAUNCH;
}

// The descriptor and semantics of hold_chain_r6 (csrc/rchain.hip) in the f16x3 arithmetic:
//   mode DSP  (7 layers, first_chunks 32, skip_layer 3, every out[] optional, aux2 optional; skip_out 217, or 172 without aux2):
//              d->wpack = hold_chain_h3_pack_bytes() bytes of fp16, [7 x 16 k steps][8 nt][2 limbs][2 h][32 i][8 e], the k order
//              of hold_trunk_h3, limb_t of s_w[j] M_j;
//   mode DBWD (8 layers, first_chunks 5, ...): d->wpack = the stream of hold_trunk_h3 (hold_trunk_h3_pack_bytes() bytes).
// c3: [n_layers] = 1 / s_w[j] of the CHAIN layers (device memory).  guard / wpack_r6: the overflow guard and the conditional
// f32x6 fallback of hold_fused_sdf_h3 (include/hold_hip.h): with wpack_r6 (the stream hold_chain_r6 takes for the same
// matrices) the entry point enqueues hold_chain_r6_if behind the kernel.
extern "C" int hold_chain_r6_if(const hold_chain_desc* dp, uint32_t* guard, hold_stream_t st);
extern "C" int hold_chain_h3(const hold_chain_desc* dp, const float* c3, uint32_t* guard, const void* wpack_r6, hold_stream_t st) {
  if (!dp || !c3) return HOLD_E_ARG;
  const hold_chain_desc& d = *dp;
  if (d.P < 0 || !d.in || !d.wpack || d.skip_layer != 3) return HOLD_E_ARG;
  const int so = d.skip_out ? d.skip_out : 217;
  if (so != 217 && so != 172) return HOLD_E_ARG;  // 172: the background net's skip width (DSP without a2 only, as hold_chain_r6)
  if (((uintptr_t)guard & 3) || (wpack_r6 && !guard)) return HOLD_E_ARG;
  const bool db = d.mode == HOLD_CHAIN_DBWD;
  if (d.mode != HOLD_CHAIN_DSP && !db) return HOLD_E_ARG;
  const int nl = db ? 8 : 7;
  if (d.n_layers != nl || d.first_chunks != (db ? 5 : 32) || d.ld_in < (db ? 40 : 256)) return HOLD_E_ARG;
  if (d.ld < 256 || (d.ld & 3) || (d.ld_in & 3)) return HOLD_E_ARG;
  if (((uintptr_t)d.in & 15) || ((uintptr_t)d.wpack & 15)) return HOLD_E_ARG;
  if (((uint64_t)d.P + 128) * (uint64_t)d.ld * 4 >= (1ull << 32)) return HOLD_E_ARG;  // 32-bit byte offsets
  if (((uint64_t)d.P + 128) * (uint64_t)d.ld_in * 4 >= (1ull << 32)) return HOLD_E_ARG;
  RCArgs a = {};
  a.P = (long)d.P; a.wpack = (const char*)d.wpack; a.c3 = c3; a.guard = guard; a.in = d.in; a.ld_in = d.ld_in; a.ld = d.ld;
  const bool has2 = d.aux2[0] != nullptr;
  if (db && !has2) return HOLD_E_ARG;
  for (int l = 0; l < nl; ++l) {
    if (!d.aux1[l] || ((uintptr_t)d.aux1[l] & 15) || ((uintptr_t)d.aux2[l] & 15) || ((uintptr_t)d.out[l] & 15) ||
        ((uintptr_t)d.out2[l] & 15))
      return HOLD_E_ARG;
    if ((d.aux2[l] != nullptr) != has2) return HOLD_E_ARG;
    if (db && (!d.out[l] || !d.out2[l])) return HOLD_E_ARG;
    a.aux1[l] = d.aux1[l]; a.aux2[l] = d.aux2[l]; a.out[l] = d.out[l]; a.out2[l] = d.out2[l];
  }
  if (d.P == 0) return HOLD_OK;
  hipStream_t s = (hipStream_t)st;
  int rc;
  // DIST = how many rendezvous ahead a weight group is requested (ring of 4 + 4 DIST units = 32 / 64 KiB).  Measured (GPU call 8
  // of round 6, 1.6 M points, developer build): DIST 3 against DIST 1: DSP 7.57 -> 6.95 ms, DSP + a2 8.48 -> 8.04, DBWD 11.10 -> 11.17
  // (HBM-bound: 58 GB at 5.2 TB/s) -- the descending sweeps take 3, the ascending one keeps the 32 KiB ring (128 instead of 160 KiB
  // of LDS).  What is left is the side traffic: with the side tiles served from L2-resident rows 6.5 / 7.1 / 9.7 ms, without the
  // result stores as well 5.8 / 6.3 / 6.7 (profiles/r06_rsweep_h3_dist_ablation.log).
  int dist = db ? 1 : 3;
#ifdef HOLD_DEV
  if (const char* v = getenv("HOLD_H3C_DIST")) dist = atoi(v);
#endif
  if (so == 172) {
    if (db || has2) return HOLD_E_ARG;
    rc = rsweep_h3_launch<RC_DSP, false, 3, 172>(a, s);
  } else if (dist == 3) {
    if (db) rc = rsweep_h3_launch<RC_DBWD, true, 3>(a, s);
    else if (has2) rc = rsweep_h3_launch<RC_DSP, true, 3>(a, s);
    else rc = rsweep_h3_launch<RC_DSP, false, 3>(a, s);
  } else {
    if (db) rc = rsweep_h3_launch<RC_DBWD, true, 1>(a, s);
    else if (has2) rc = rsweep_h3_launch<RC_DSP, true, 1>(a, s);
    else rc = rsweep_h3_launch<RC_DSP, false, 1>(a, s);
  }
  if (rc != HOLD_OK || !wpack_r6) return rc;
  hold_chain_desc f = d;
  f.wpack = (const float*)wpack_r6;
  return hold_chain_r6_if(&f, guard, st);
}
