// Register-resident DESCENDING sweeps of the ImplicitNet trunk (gfx950) -- d sdf / d a_l of the normal path and the
// first-order backward that torch.autograd derives from ImplicitNet.forward (code/src/networks/shape_net.py:84-130;
// code/src/engine/volsdf_utils.py:71-96) -- in the structure of csrc/rmlp.hip: one wave per SIMD owns 32 points for the whole chain, a layer's 256 x 32 outputs stay
// in the wave's accumulator registers, and -- after the per-layer epilogue -- ARE the next layer's MFMA B operand (the
// "virtual k order" of rmlp.hip: k step j, element e of lane half hh <-> feature 16 j + 8 (e / 4) + 4 hh + e % 4).
//
//   v_{l-1} = (M_j v_l) * sp'(aux1_j) [+ aux2_j]          7 layers, input v_7 [P][256]
//   (the semantics of hold_chain / hold_chain_x6, mode DSP, skip_layer = 3, include/hold_hip.h)
// Used for the sweep WITHOUT the additive side input (d sdf / d a_l of the normal path; 150 vs 134 TF-eq); with aux2
// (3 KiB per point and layer) it measured 113 vs 124 TF-eq for hold_chain_x6 and the host keeps that sweep there.
// The ascending second-order sweep (mode DBWD: two side inputs and two results per layer) was built in this structure too
// and measured SLOWER than hold_chain_x6 (85 vs 108 TF-eq, matrix pipe 22 % busy, 57 % of the wave time stalled at issue):
// a lane owns a POINT here, so side inputs and results move as 32-byte row fragments -- four L2 requests per 128-byte
// line where the LDS-resident kernel's feature-per-lane layout issues one -- and at 4 KiB per point and layer the L2
// request rate, not the matrix pipe, is the bound.  It was removed again (git history); DBWD stays on hold_chain_x6.
//
// What streams: the weight limbs (24 KiB per 16-wide k step, LDS ring of 3 slots filled by LDS-DMA two steps ahead,
// shared by the four waves) and, new here, the per-layer SIDE inputs -- for every k step the 8 stored h (and a2 / t)
// values a lane needs are two (four) 16-byte row fragments: they are fetched by LDS-DMA too (lane-linear image, wave
// private, ring of 4, requested three steps before use), so no load ever has a register destination in flight and the
// compiler's counters only see ds_reads and stores; completion of all DMA is counted by hand at the one rendezvous per
// k step.  The vector-memory counter retires in order, so the side request is issued right BEHIND the weight DMA of a
// rendezvous: the next rendezvous waits for those weights (L2-resident, one step is ample) without forcing the side
// fragments that follow them in the queue -- these get two full steps of HBM latency.  The results (next layer's input) are stored with
// 16-byte row-fragment stores straight from the epilogue.
// Exposed per block of 128 points: loading the chain input (32 x 16 B per lane) and the epilogue of the LAST layer.
// Roofline: bf16 MFMA pipe; HBM bytes per point and layer: DSP 2 KiB (1 side + 1 out), DSP + a2 3 KiB.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hold_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int NW = 4, BPTS = 32 * NW, PIECE = 1024, SLOT = 24 * PIECE;
constexpr int SKIP_OUT = 217, SIDE_RING = 4;

struct RCArgs {
  long P;
  const char* wpack;      // 7 x 16 k steps
  const float* in;        // v_7 [P][ld_in >= 256]
  int ld_in, ld;
  const float* aux1[8];
  const float* aux2[8];
  float* out[8];
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

struct Limbs { u32x4 l[3]; };
struct Split3 { uint32_t p1, p2, p3; };
__device__ __forceinline__ Split3 split2(float x0, float x1) {  // exact truncation split, two values -> one dword per limb
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
__device__ __forceinline__ void put_limbs(Limbs& out, int c, Split3 s) {  // pinned: see rmlp.hip
  asm volatile("" : "+v"(s.p1), "+v"(s.p2), "+v"(s.p3));
  out.l[0][c] = s.p1;
  out.l[1][c] = s.p2;
  out.l[2][c] = s.p3;
}

// softplus'(a) recovered from h = softplus(a): 1 - e^{-100 h} (series where the subtraction would cancel)
__device__ __forceinline__ float dsp(float h) {
  const float x = 100.0f * h;
  const float e = __builtin_amdgcn_exp2f(-144.26950408889634f * h);
  const float ser = x * (1.0f - x * (0.5f - x * (0.16666667f - 0.041666668f * x)));
  return (x < 0.05f) ? ser : 1.0f - e;
}

// weights: six 1 KiB pieces of k step `step` into ring slot `slot` (inline assembly: see rmlp.hip)
__device__ __forceinline__ void dma_w(const char* wpack, uint32_t lane16, int step, int slot, int wave) {
  const char* src = wpack + (long)step * SLOT + wave * (6 * PIECE);
  const uint32_t dst = (uint32_t)(slot * SLOT + wave * (6 * PIECE));
  uint32_t keep;
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
// side input: the two 16-byte row fragments (columns c0 + 4 hh .. and c0 + 8 + 4 hh ..) of one matrix for this wave's 32
// rows -> two lane-linear 1 KiB pieces at LDS byte `dst`.  `base` = matrix + c0 (wave-uniform), rowoff = this lane's
// (row * ld + 4 hh) * 4.  The instruction offset advances the global AND the LDS address: M0 of the second piece is
// dst + 1024 - 32.
__device__ __forceinline__ void dma_side(const float* base, uint32_t rowoff, uint32_t dst) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2\n\t"
      "s_mov_b32 m0, %4\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2 offset:32\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(rowoff), "s"(base), "s"(dst), "s"(dst + PIECE - 32)
      : "memory");
}

// single 1 KiB pieces (see rmlp.hip:dma_piece: M0 is written by every statement that reads it, nothing else uses it)
__device__ __forceinline__ void dma_piece(const char* src, uint32_t voff, uint32_t dst) {
  asm volatile(
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %0, %1"
      :
      : "v"(voff), "s"(src), "s"(dst)
      : "memory");
}

#define RC_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

// ABL: developer-build timing ablations (results are garbage): 2 no in-loop stores, 3 the stores' bytes as lane-linear
// 1 KiB pieces, 4 the side loads' bytes as lane-linear 1 KiB pieces, 5 both
template <bool A2, int ABL = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void rchain_kernel(RCArgs a) {
  constexpr int NAUX = A2 ? 2 : 1;           // side matrices per layer
  constexpr int NOUT = 1;                    // result matrices per layer
  constexpr int L = 7;                       // chain layers
  constexpr int NST = 16 * L;                // k steps per block of points
  constexpr int R3 = 4;                      // weight ring slots; DMA distance R3 - 1 steps
  constexpr int SIDE_SLOT = NAUX * 2 * PIECE;
  constexpr int OFF_SIDE = R3 * SLOT;
  // VMEM operations issued between the weight DMA of the previous rendezvous and this one's wait: the side DMA that follows
  // that weight DMA, the previous step's stores B, this step's stores A.  Waiting down to this count lands the weights
  // (needed now) and every side fragment issued ONE rendezvous earlier, while the newest side fragments stay in flight:
  // they get two full steps, the L2-resident weights one.
  // The weights needed next were requested TWO rendezvous ago, like the side fragments consumed in the next step (which
  // sit right behind them in the queue): everything younger -- the previous rendezvous' 6 weight pieces and side request,
  // two steps' stores -- may stay in flight; weights and side both get two full steps.
  constexpr int NWAIT = 6 + 2 * NAUX + 3 * NOUT;  // (queue per step: weights x 6, store, side x 2 NAUX, store)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hh = lane >> 5, li = lane & 31;
  const uint32_t lane16 = lane * 16;
  const char* ring_lane = smem + lane * 16;
  const uint32_t side_dst0 = (uint32_t)(OFF_SIDE + wave * (SIDE_RING * SIDE_SLOT));
  const float* side_rd = reinterpret_cast<const float*>(smem + OFF_SIDE + wave * (SIDE_RING * SIDE_SLOT)) + lane * 4;
  const uint32_t nbytes = (uint32_t)(a.P * a.ld * 4);

  f32x16 P[8], Q[8];
  u32x4 A[2][6];
  Limbs Bc, Bn;

  auto read_pair = [&](int slot, int pair, u32x4 (&dst)[6]) {
    const char* base = ring_lane + slot * SLOT + pair * (6 * PIECE);
#pragma unroll
    for (int i = 0; i < 6; ++i) dst[i] = *reinterpret_cast<const u32x4*>(base + i * PIECE);
  };
  auto zero_q = [&]() {
#pragma unroll
    for (int nt = 0; nt < 8; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) Q[nt][r] = 0.f;
  };

  // the first R3 - 1 k steps of the stream; gs counts k steps over all blocks (ring slot = gs % R3)
#pragma unroll
  for (int s0 = 0; s0 < R3 - 1; ++s0) dma_w(a.wpack, lane16, s0, s0, wave);
  int gs = 0;
  int first = 1;

  for (long blk = blockIdx.x; blk * BPTS < a.P; blk += gridDim.x) {
    const long row = blk * BPTS + wave * 32 + li;  // this lane's point
    const long crow = row < a.P ? row : a.P - 1;    // clamped for the DMA reads (rows >= P: results dropped by the stores)
    const uint32_t st_off = (uint32_t)((row * a.ld + 4 * hh) * 4);   // stores: unclamped, the buffer range check drops them
    uint32_t ld_off = (uint32_t)((crow * a.ld + 4 * hh) * 4);  // side DMA source
    const uint32_t lin_off = (uint32_t)((blk * BPTS + wave * 32) * a.ld * 4) + lane16;
    if (ABL >= 4) ld_off = lin_off;

    // ---- chain input: v_7 rows into the accumulator layout, P[nt][4 g + k] = in[row][32 nt + 8 g + 4 hh + k] ----
    {
      const rsrc_t irs = make_rsrc(a.in, (uint32_t)(a.P * a.ld_in * 4));
      const uint32_t ioff = (uint32_t)((row * a.ld_in + 4 * hh) * 4);
#pragma unroll
      for (int nt = 0; nt < 8; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(irs, ioff + (32 * nt + 8 * g) * 4, 0, 0);
#pragma unroll
          for (int k = 0; k < 4; ++k) P[nt][4 * g + k] = bitsf(v[k]);
        }
    }
    if (first) {
      RC_WAIT_VM(0);
      __builtin_amdgcn_s_barrier();
      read_pair(0, 0, A[0]);
      first = 0;
    }

    // side pointers of the epilogue layers lw = l - 1 ("lo", DMA of steps j' < 12) and lw = l ("hi", steps j' >= 12)
    const float *lo1 = a.aux1[0], *lo2 = NAUX == 2 ? a.aux2[0] : nullptr, *hi1 = lo1, *hi2 = lo2;

    // One k step with an EXPLICIT schedule (see rmlp.hip:kstep): 4 groups x 12 MFMAs, behind every MFMA a fixed slice of
    // the rest -- one fragment read for the next group (gaps 0..5); in the group behind the rendezvous the six weight
    // pieces of step tl + R3 - 1 (every second gap), in the last group, BEHIND them in the queue, the side fragments of
    // epilogue k step 16 (l - 1) + jp + 4 (consumed three steps later; side slot = k step % 4); and cnt[group] / 12
    // micro-operations of the next step's epilogue -- closed by a full scheduling barrier.
    auto kstep = [&](int tl, int jp, const int (&cnt)[4], auto&& mop) {
      const int slot = gs % R3, nslot = (gs + 1) % R3, fslot = (gs + R3 - 1) % R3;
      const char* wsrc = a.wpack + (long)((tl + R3 - 1) % NST) * SLOT + wave * (6 * PIECE);
      const uint32_t wdst = (uint32_t)(fslot * SLOT + wave * (6 * PIECE));
      const int jc = (jp + 4) & 15;
      const uint32_t sd = side_dst0 + (jp & 3) * SIDE_SLOT;
      const char* s1 = reinterpret_cast<const char*>((jp < 12 ? lo1 : hi1) + (ABL >= 4 ? 512 : 16) * jc);
      const char* s2 = reinterpret_cast<const char*>((jp < 12 ? lo2 : hi2) + (ABL >= 4 ? 512 : 16) * jc);
      constexpr int SECOND = ABL >= 4 ? 1024 : 32;
#pragma unroll
      for (int pair = 0; pair < 4; ++pair) {
        if (pair == 2) {  // rendezvous: the weights of step gs + 1 have landed in every wave; slot gs - 1 is free
          RC_WAIT_VM(NWAIT);
          __builtin_amdgcn_s_barrier();
        }
        const char* rd = ring_lane + (pair < 3 ? slot * SLOT + (pair + 1) * (6 * PIECE) : nslot * SLOT);
#pragma unroll
        for (int m = 0; m < 12; ++m) {
          const int pr = m >> 1, tt = m & 1;
          const int wl = (pr == 2 || pr == 3) ? 1 : (pr == 5 ? 2 : 0);  // (w limb, act limb): 00 01 10 11 02 20
          const int al = (pr == 1 || pr == 3) ? 1 : (pr == 4 ? 2 : 0);
          Q[2 * pair + tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[pair & 1][3 * tt + wl]),
                                                                    __builtin_bit_cast(bf16x8, Bc.l[al]), Q[2 * pair + tt],
                                                                    0, 0, 0);
          if (m < 6) A[(pair + 1) & 1][m] = *reinterpret_cast<const u32x4*>(rd + m * PIECE);
          if (pair == 2 && (m & 1) == 0) dma_piece(wsrc + (m >> 1) * PIECE, lane16, wdst + (m >> 1) * PIECE);
          if (pair == 3 && (m == 1 || m == 4)) dma_piece(s1 + SECOND * (m == 4), ld_off, sd + (m == 4) * PIECE);
          if (pair == 3 && NAUX == 2 && (m == 7 || m == 10))
            dma_piece(s2 + SECOND * (m == 10), ld_off, sd + 2 * PIECE + (m == 10) * PIECE);
#pragma unroll
          for (int u = 0; u < 8; ++u) {  // constant trip count (the slice bounds fold once pair and m are unrolled)
            const int k = cnt[pair] * m / 12 + u;
            if (k < cnt[pair] * (m + 1) / 12) mop(pair, k);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      Bc = Bn;
      gs += 1;
    };
    static constexpr int CNT_NONE[4] = {0, 0, 0, 0};
    auto no_mop = [](int, int) {};

    // Epilogue of k step j of the finished layer in P (chain layer lw, wave-uniform; raw = chain layer 0 is fed by the chain
    // input itself): 8 values, four stages of micro-operations in round-major order (rmlp.hip: consecutive operations
    // independent).  Side values from this wave's LDS ring slot `ss` (two / four 16-byte reads).
    //   stage 0: y, h (and a2), x = 100 h, e = exp(-x), the four operations of the small-x series of 1 - e^{-x}
    //   stage 1: 1 - e, series select, y * sp', + a2, raw / skip-layer selects                           -> r[8]
    //   stage 2 / 3: limb split of r[0..3] / r[4..7], 16-byte store
    rsrc_t ors = make_rsrc(nullptr, 0);
    struct EpiState { float y[8], h[8], x2[8], e[8], ser[8], r[8]; uint32_t w[2][8]; };
    static constexpr int CNT_DSP[4] = {64, A2 ? 40 : 32, 23, 23};
    auto epi_mop = [&](int j, int ss, bool raw, bool skip, int stage, int k, Limbs& out, EpiState& st) {
      const int nt = j >> 1, q = j & 1;
      const int rd = k >> 3, i = k & 7;
      if (stage == 0) {
        if (rd == 0) {
          st.y[i] = P[nt][8 * q + i];
          if ((i & 3) == 0) {  // this value and the next three: one 16-byte read of the side fragment(s)
            const float* sp = side_rd + ss * (SIDE_SLOT / 4) + (i >> 2) * (PIECE / 4);
            const f32x4 hv = *reinterpret_cast<const f32x4*>(sp);
#pragma unroll
            for (int v = 0; v < 4; ++v) st.h[i + v] = hv[v];
            if (A2) {
              const f32x4 xv = *reinterpret_cast<const f32x4*>(sp + 2 * (PIECE / 4));
#pragma unroll
              for (int v = 0; v < 4; ++v) st.x2[i + v] = xv[v];
            }
          }
        } else if (rd == 1) st.e[i] = -144.26950408889634f * st.h[i];
        else if (rd == 2) st.e[i] = __builtin_amdgcn_exp2f(st.e[i]);
        else if (rd == 3) st.h[i] = 100.0f * st.h[i];
        else if (rd == 4) st.ser[i] = fmaf(st.h[i], -0.041666668f, 0.16666667f);
        else if (rd == 5) st.ser[i] = fmaf(-st.h[i], st.ser[i], 0.5f);
        else if (rd == 6) st.ser[i] = fmaf(-st.h[i], st.ser[i], 1.0f);
        else st.ser[i] = st.h[i] * st.ser[i];
      } else if (stage == 1) {
        if (rd == 0) st.e[i] = 1.0f - st.e[i];
        else if (rd == 1) st.e[i] = (st.h[i] < 0.05f) ? st.ser[i] : st.e[i];
        else if (rd == 2) st.r[i] = st.y[i] * st.e[i];
        else if (A2 && rd == 3) st.r[i] = st.r[i] + st.x2[i];
        else {
          float r = raw ? st.y[i] : st.r[i];
          if (j >= 13) {  // skip layer (chain layer 3), columns 217..: the raw products (d / d skip input) are stored
            const int f = 16 * j + 8 * (i >> 2) + 4 * hh + (i & 3);
            r = (skip && f >= SKIP_OUT) ? st.y[i] : r;
          }
          st.r[i] = r;
        }
      } else {
        const int h2 = stage - 2;
        if (k == 22) {
          const f32x4 v = {st.r[4 * h2], st.r[4 * h2 + 1], st.r[4 * h2 + 2], st.r[4 * h2 + 3]};
          if (ABL == 2) return;
          if (ABL == 3 || ABL == 5) {
            store4(v, ors, lin_off + (2 * j + h2) * 1024);
            return;
          }
          store4(v, ors, st_off + (16 * j + 8 * h2) * 4);
          return;
        }
        const int d = k & 1, op = k >> 1;
        const float x0 = st.r[4 * h2 + 2 * d], x1 = st.r[4 * h2 + 2 * d + 1];
        uint32_t* w = st.w[d];
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

    for (int l = 0; l < L; ++l) {
      // MFMA layer l consumes P through the epilogue of chain layer lw = l - 1 (DSP l = 0: the raw chain input)
      if (l > 0) {
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) P[nt] = Q[nt];
      }
#pragma unroll
      for (int nt = 0; nt < 8; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) asm volatile("" : "+a"(P[nt][r]));
      zero_q();
      const int lw = l - 1;
      const bool raw = l == 0;
      const bool skip = lw == 3;
      const int lwc = lw < 0 ? 0 : lw, lhc = l < L - 1 ? l : L - 2;
      lo1 = a.aux1[lwc];
      hi1 = a.aux1[lhc];
      if (NAUX == 2) {
        lo2 = a.aux2[lwc];
        hi2 = a.aux2[lhc];
      }
      ors = make_rsrc(raw ? nullptr : a.out[lwc], nbytes);
      const int t0 = 16 * l;
      EpiState st;
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int k = 0; k < 64; ++k)
          if (k < CNT_DSP[c]) epi_mop(0, 0, raw, skip, c, k, Bc, st);
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if (j + 1 < 16)
          kstep(t0 + j, j, CNT_DSP, [&](int c, int k) { epi_mop(j + 1, (j + 1) & 3, raw, skip, c, k, Bn, st); });
        else
          kstep(t0 + j, j, CNT_NONE, no_mop);
      }
    }

    // ---- epilogue of the last chain layer (exposed): side rows by ordinary buffer loads ----
    {
      const rsrc_t a1 = make_rsrc(a.aux1[L - 1], nbytes);
      const rsrc_t a2 = make_rsrc(NAUX == 2 ? a.aux2[L - 1] : nullptr, nbytes);
      const rsrc_t o1 = make_rsrc(a.out[L - 1], nbytes);
#pragma unroll
      for (int nt = 0; nt < 8; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const uint32_t co = (32 * nt + 8 * g) * 4;
          const u32x4 hv = __builtin_amdgcn_raw_buffer_load_b128(a1, st_off + co, 0, 0);
          u32x4 xv = {0u, 0u, 0u, 0u};
          if (NAUX == 2) xv = __builtin_amdgcn_raw_buffer_load_b128(a2, st_off + co, 0, 0);
          f32x4 r;
#pragma unroll
          for (int k = 0; k < 4; ++k) r[k] = Q[nt][4 * g + k] * dsp(bitsf(hv[k])) + (A2 ? bitsf(xv[k]) : 0.f);
          store4(r, o1, st_off + co);
        }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// Round 4: the same sweep with its side traffic moved as FULL 128-byte lines (VERDICT r3 #1a, DESIGN.md 4.1 ablation: the
// 32-byte row fragments above cost 4 L2 requests per line; with two side inputs the request rate, not the matrix pipe,
// bounds the kernel).  A lane still owns a POINT, so rows are transposed through LDS, per n-tile = 2 k steps:
//   * side input: the [32 points][32 features] tile of h (and a2) = 4 KiB = four LDS-DMA pieces, piece i = rows 8 i ..
//     8 i + 7, lane L -> row 8 i + L / 8, 16-byte chunk L % 8: eight lanes fetch one whole line.  LDS-DMA writes
//     lane-linear, so the image is row-major [32][128 B]; the swizzle sits on the SOURCE address: position p of row r
//     holds chunk p ^ f(r), f(r) = ((r >> 1) & 7) ^ ((r & 1) << 2).  The epilogue's fragment read (lane (hh, li): row li,
//     chunk c = 4 q + 2 e4 + hh at position c ^ f(li)) then hits 16 distinct 16-byte slots of the 256-byte bank row in each
//     of ds_read_b128's four lane groups, and the result write 8 distinct slots of the 128-byte row in each of
//     ds_write_b128's contiguous 8-lane groups (MI355X_MICROARCH.md, LDS table).
//   * result: the epilogue writes its two fragments per k step into the wave's result tile (same swizzle); once both k
//     steps of a tile are done the tile is read back lane-linear (4 x ds_read_b128) and stored as four row-contiguous
//     16-byte-per-lane stores: eight lanes write one whole line.  DS operations of a wave execute in order, so write ->
//     read-back -> next write need no counter.
// Ring: two side slots (tile nt in slot nt & 1).  Tile nt is read last in pair 0 of MFMA step 2 nt; tile nt + 2 is
// requested in pair 3 of that step (behind the weight pieces) and used from step 2 nt + 3 on.  VMEM queue per two steps:
//   even step: W x 6, S x 4 NAUX | odd step: stores x 4, W x 6
// and the rendezvous waits (in-order retirement) are counted from it: see NW_EVEN / NW_ODD.
// LDS: the result tiles (16 KiB) leave room for a 3-slot weight ring only when there are two side matrices.
template <bool A2>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void rtile_kernel(RCArgs a) {
  constexpr int NAUX = A2 ? 2 : 1;
  constexpr int L = 7;
  constexpr int NST = 16 * L;
  constexpr int R3 = A2 ? 3 : 4;
  constexpr int TILE = 4 * PIECE;              // [32 rows][32 features] fp32
  constexpr int SLOT_T = NAUX * TILE;          // one side slot
  constexpr int WREG = 2 * SLOT_T + TILE;      // wave region: two side slots + the result tile
  constexpr int OFF_SIDE = R3 * SLOT;
  constexpr int OUT_OFF = 2 * SLOT_T;
  // R3 = 3: the weights needed at a rendezvous were requested at the previous one.  Even step: nothing younger exists
  // (the previous odd step issued its stores BEFORE its weight pieces) -> 0; odd step: the even step's side pieces and
  // this step's stores are younger.  R3 = 4: requested two rendezvous ago.  Even step: younger = the side pieces of
  // step - 2 (needed in the next step: force them), the stores and weight pieces of step - 1 -> 4 + 6; odd step:
  // weight pieces and side pieces of step - 1, this step's stores.
  constexpr int NW_EVEN = R3 == 3 ? 0 : 10;
  constexpr int NW_ODD = R3 == 3 ? 4 * NAUX + 4 : 6 + 4 * NAUX + 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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

  f32x16 P[8], Q[8];
  u32x4 A[2][6];
  Limbs Bc, Bn;
  f32x4 rbv[4];

  auto read_pair = [&](int slot, int pair, u32x4 (&dst)[6]) {
    const char* base = ring_lane + slot * SLOT + pair * (6 * PIECE);
#pragma unroll
    for (int i = 0; i < 6; ++i) dst[i] = *reinterpret_cast<const u32x4*>(base + i * PIECE);
  };
  auto zero_q = [&]() {
#pragma unroll
    for (int nt = 0; nt < 8; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) Q[nt][r] = 0.f;
  };

#pragma unroll
  for (int s0 = 0; s0 < R3 - 1; ++s0) dma_w(a.wpack, lane16, s0, s0, wave);
  int gs = 0;
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
      dvoff[i] = (uint32_t)((cr * a.ld + 4 * cs) * 4);
      svoff[i] = (uint32_t)((r * a.ld + 4 * cs) * 4);
    }

    // ---- chain input: v_7 rows into the accumulator layout, P[nt][4 g + k] = in[row][32 nt + 8 g + 4 hh + k] ----
    {
      const rsrc_t irs = make_rsrc(a.in, (uint32_t)(a.P * a.ld_in * 4));
      const uint32_t ioff = (uint32_t)((row * a.ld_in + 4 * hh) * 4);
#pragma unroll
      for (int nt = 0; nt < 8; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(irs, ioff + (32 * nt + 8 * g) * 4, 0, 0);
#pragma unroll
          for (int k = 0; k < 4; ++k) P[nt][4 * g + k] = bitsf(v[k]);
        }
    }
    if (first) {
      RC_WAIT_VM(0);
      __builtin_amdgcn_s_barrier();
      read_pair(0, 0, A[0]);
      first = 0;
    }

    const float *lo1 = a.aux1[0], *lo2 = NAUX == 2 ? a.aux2[0] : nullptr, *hi1 = lo1, *hi2 = lo2;

    // One k step (explicit schedule, see rchain_kernel): jp = step inside the layer.  Even steps request side tile
    // jp / 2 + 2 (tiles 8, 9 = tiles 0, 1 of the next epilogue layer) into the slot tile jp / 2 has just left.
    auto kstep = [&](int tl, int jp, const int (&cnt)[4], auto&& mop) {
      const int slot = gs % R3, nslot = (gs + 1) % R3, fslot = (gs + R3 - 1) % R3;
      const char* wsrc = a.wpack + (long)((tl + R3 - 1) % NST) * SLOT + wave * (6 * PIECE);
      const uint32_t wdst = (uint32_t)(fslot * SLOT + wave * (6 * PIECE));
      const int tn = (jp >> 1) + 2;
      const char* s1 = reinterpret_cast<const char*>((tn < 8 ? lo1 : hi1) + 32 * (tn & 7));
      const char* s2 = reinterpret_cast<const char*>((tn < 8 ? lo2 : hi2) + 32 * (tn & 7));
      const uint32_t sd = side_dst0 + ((jp >> 1) & 1) * SLOT_T;
#pragma unroll
      for (int pair = 0; pair < 4; ++pair) {
        if (pair == 2) {  // rendezvous: the weights of step gs + 1 have landed in every wave; slot gs - 1 is free
          if (jp & 1) RC_WAIT_VM(NW_ODD);
          else RC_WAIT_VM(NW_EVEN);
          __builtin_amdgcn_s_barrier();
        }
        const char* rd = ring_lane + (pair < 3 ? slot * SLOT + (pair + 1) * (6 * PIECE) : nslot * SLOT);
#pragma unroll
        for (int m = 0; m < 12; ++m) {
          const int pr = m >> 1, tt = m & 1;
          const int wl = (pr == 2 || pr == 3) ? 1 : (pr == 5 ? 2 : 0);  // (w limb, act limb): 00 01 10 11 02 20
          const int al = (pr == 1 || pr == 3) ? 1 : (pr == 4 ? 2 : 0);
          Q[2 * pair + tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[pair & 1][3 * tt + wl]),
                                                                    __builtin_bit_cast(bf16x8, Bc.l[al]), Q[2 * pair + tt],
                                                                    0, 0, 0);
          if (m < 6) A[(pair + 1) & 1][m] = *reinterpret_cast<const u32x4*>(rd + m * PIECE);
          if (pair == 2 && (m & 1) == 0) dma_piece(wsrc + (m >> 1) * PIECE, lane16, wdst + (m >> 1) * PIECE);
          if (pair == 3 && (jp & 1) == 0) {
            const int i = (m % 3 == 0) ? -1 : 2 * ((m % 6) / 3) + (m % 3) - 1;  // m = 1 2 4 5 | 7 8 10 11 -> 0 1 2 3
            if (i >= 0 && m < 6) dma_piece(s1, dvoff[i], sd + i * PIECE);
            if (i >= 0 && m >= 6 && NAUX == 2) dma_piece(s2, dvoff[i], sd + TILE + i * PIECE);
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) {  // constant trip count (the slice bounds fold once pair and m are unrolled)
            const int k = cnt[pair] * m / 12 + u;
            if (k < cnt[pair] * (m + 1) / 12) mop(pair, k);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      Bc = Bn;
      gs += 1;
    };

    // Epilogue micro-operations of k step j of the finished layer in P (see rchain_kernel::epi_mop); side values from the
    // tile j / 2 in slot (j / 2) & 1, results into the wave's result tile.
    rsrc_t ors = make_rsrc(nullptr, 0);
    struct EpiState { float y[8], h[8], x2[8], e[8], ser[8], r[8]; uint32_t w[2][8]; };
    constexpr int C1 = A2 ? 40 : 32;
    static constexpr int CNT_EVEN[4] = {64, C1, 23, 23};          // epilogue only
    static constexpr int CNT_ODD[4] = {64 + 4, C1 + 4, 23, 23};   // + read-back and stores of the tile finished last step
    static constexpr int CNT_LAST[4] = {4, 4, 0, 0};              // step 15: tile 7, no epilogue
    auto epi_mop = [&](int j, bool raw, bool skip, int stage, int k, Limbs& out, EpiState& st) {
      const int nt = j >> 1, q = j & 1;
      const int rd = k >> 3, i = k & 7;
      if (stage == 0) {
        if (rd == 0) {
          st.y[i] = P[nt][8 * q + i];
          if ((i & 3) == 0) {  // this value and the next three: one 16-byte read of the side fragment(s)
            const char* sp = smem + (frag0 ^ (uint32_t)(64 * q + 32 * (i >> 2))) + (nt & 1) * SLOT_T;
            const f32x4 hv = *reinterpret_cast<const f32x4*>(sp);
#pragma unroll
            for (int v = 0; v < 4; ++v) st.h[i + v] = hv[v];
            if (A2) {
              const f32x4 xv = *reinterpret_cast<const f32x4*>(sp + TILE);
#pragma unroll
              for (int v = 0; v < 4; ++v) st.x2[i + v] = xv[v];
            }
          }
        } else if (rd == 1) st.e[i] = -144.26950408889634f * st.h[i];
        else if (rd == 2) st.e[i] = __builtin_amdgcn_exp2f(st.e[i]);
        else if (rd == 3) st.h[i] = 100.0f * st.h[i];
        else if (rd == 4) st.ser[i] = fmaf(st.h[i], -0.041666668f, 0.16666667f);
        else if (rd == 5) st.ser[i] = fmaf(-st.h[i], st.ser[i], 0.5f);
        else if (rd == 6) st.ser[i] = fmaf(-st.h[i], st.ser[i], 1.0f);
        else st.ser[i] = st.h[i] * st.ser[i];
      } else if (stage == 1) {
        if (rd == 0) st.e[i] = 1.0f - st.e[i];
        else if (rd == 1) st.e[i] = (st.h[i] < 0.05f) ? st.ser[i] : st.e[i];
        else if (rd == 2) st.r[i] = st.y[i] * st.e[i];
        else if (A2 && rd == 3) st.r[i] = st.r[i] + st.x2[i];
        else {
          float r = raw ? st.y[i] : st.r[i];
          if (j >= 13) {  // skip layer (chain layer 3), columns 217..: the raw products (d / d skip input) are stored
            const int f = 16 * j + 8 * (i >> 2) + 4 * hh + (i & 3);
            r = (skip && f >= SKIP_OUT) ? st.y[i] : r;
          }
          st.r[i] = r;
        }
      } else {
        const int h2 = stage - 2;
        if (k == 22) {
          const f32x4 v = {st.r[4 * h2], st.r[4 * h2 + 1], st.r[4 * h2 + 2], st.r[4 * h2 + 3]};
          *reinterpret_cast<f32x4*>(smem + (frag0 ^ (uint32_t)(64 * q + 32 * h2)) + OUT_OFF) = v;
          return;
        }
        const int d = k & 1, op = k >> 1;
        const float x0 = st.r[4 * h2 + 2 * d], x1 = st.r[4 * h2 + 2 * d + 1];
        uint32_t* w = st.w[d];
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
    // result tile nt: piece i read back lane-linear (stage 0 extras), stored as whole lines (stage 1 extras)
    auto io_mop = [&](int nt, int stage, int i) {
      if (stage == 0) rbv[i] = *reinterpret_cast<const f32x4*>(smem + rb0 + i * PIECE);
      else store4(rbv[i], ors, svoff[i] + 128 * nt);
    };

    for (int l = 0; l < L; ++l) {
      // MFMA layer l consumes P through the epilogue of chain layer lw = l - 1 (DSP l = 0: the raw chain input)
      if (l > 0) {
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) P[nt] = Q[nt];
      }
#pragma unroll
      for (int nt = 0; nt < 8; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) asm volatile("" : "+a"(P[nt][r]));
      zero_q();
      const int lw = l - 1;
      const bool raw = l == 0;
      const bool skip = lw == 3;
      const int lwc = lw < 0 ? 0 : lw, lhc = l < L - 1 ? l : L - 2;
      lo1 = a.aux1[lwc];
      hi1 = a.aux1[lhc];
      if (NAUX == 2) {
        lo2 = a.aux2[lwc];
        hi2 = a.aux2[lhc];
      }
      ors = make_rsrc(raw ? nullptr : a.out[lwc], nbytes);
      const int t0 = 16 * l;
      EpiState st;
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int k = 0; k < 64; ++k)
          if (k < CNT_EVEN[c]) epi_mop(0, raw, skip, c, k, Bc, st);
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if (j == 15)
          kstep(t0 + j, j, CNT_LAST, [&](int c, int k) { io_mop(7, c, k); });
        else if (j & 1)
          kstep(t0 + j, j, CNT_ODD, [&](int c, int k) {
            if (c == 0 && k >= 64) io_mop(j >> 1, 0, k - 64);
            else if (c == 1 && k >= C1) io_mop(j >> 1, 1, k - C1);
            else epi_mop(j + 1, raw, skip, c, k, Bn, st);
          });
        else
          kstep(t0 + j, j, CNT_EVEN, [&](int c, int k) { epi_mop(j + 1, raw, skip, c, k, Bn, st); });
      }
    }

    // ---- epilogue of the last chain layer (exposed): side rows by ordinary buffer loads ----
    {
      const rsrc_t a1 = make_rsrc(a.aux1[L - 1], nbytes);
      const rsrc_t a2 = make_rsrc(NAUX == 2 ? a.aux2[L - 1] : nullptr, nbytes);
      const rsrc_t o1 = make_rsrc(a.out[L - 1], nbytes);
#pragma unroll
      for (int nt = 0; nt < 8; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const uint32_t co = (32 * nt + 8 * g) * 4;
          const u32x4 hv = __builtin_amdgcn_raw_buffer_load_b128(a1, st_off + co, 0, 0);
          u32x4 xv = {0u, 0u, 0u, 0u};
          if (NAUX == 2) xv = __builtin_amdgcn_raw_buffer_load_b128(a2, st_off + co, 0, 0);
          f32x4 r;
#pragma unroll
          for (int k = 0; k < 4; ++k) r[k] = Q[nt][4 * g + k] * dsp(bitsf(hv[k])) + (A2 ? bitsf(xv[k]) : 0.f);
          store4(r, o1, st_off + co);
        }
    }
  }
}

}  // namespace

extern "C" int64_t hold_chain_r6_pack_bytes(void) { return (int64_t)(7 * 16) * SLOT; }

template <bool A2>
static int rchain_launch(const RCArgs& a, hipStream_t s) {
  constexpr int NAUX = A2 ? 2 : 1;
  constexpr int lds_frag = 4 * SLOT + NW * SIDE_RING * NAUX * 2 * PIECE;                  // rchain_kernel
  constexpr int lds_tile = (A2 ? 3 : 4) * SLOT + NW * (2 * NAUX + 1) * 4 * PIECE;         // rtile_kernel
  static_assert(lds_frag <= 160 * 1024 && lds_tile <= 160 * 1024, "LDS budget");
  static int n_cu = 0;
  static bool attr_set = false;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return HOLD_E_LAUNCH;
    n_cu = prop.multiProcessorCount;
  }
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)rtile_kernel<A2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_tile) != hipSuccess)
      return HOLD_E_LAUNCH;
    attr_set = true;
  }
  const long blocks = (a.P + BPTS - 1) / BPTS;
  const dim3 grid((unsigned)(blocks < n_cu ? blocks : n_cu));
#ifdef HOLD_DEV
  if (const char* v = getenv("HOLD_R6_IO")) {  // developer A/B: HOLD_R6_IO=frag -> the round-3 row-fragment kernel
    if (v[0] == 'f') {
      if (hipFuncSetAttribute((const void*)rchain_kernel<A2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_frag) != hipSuccess)
        return HOLD_E_LAUNCH;
      hipLaunchKernelGGL((rchain_kernel<A2>), grid, dim3(256), lds_frag, s, a);
      return hipGetLastError() == hipSuccess ? HOLD_OK : HOLD_E_LAUNCH;
    }
  }
#endif
  hipLaunchKernelGGL((rtile_kernel<A2>), grid, dim3(256), lds_tile, s, a);
  return hipGetLastError() == hipSuccess ? HOLD_OK : HOLD_E_LAUNCH;
}

// The descriptor and semantics of hold_chain_x6 for the descending sweeps (mode DSP, 7 layers, first_chunks 32,
// skip_layer 3, every out[] optional, aux2 optional) with the register-resident structure.  d->wpack =
// hold_chain_r6_pack_bytes() bytes in the k order of hold_trunk_r6.
extern "C" int hold_chain_r6(const hold_chain_desc* dp, hold_stream_t st) {
  if (!dp) return HOLD_E_ARG;
  const hold_chain_desc& d = *dp;
  if (d.P < 0 || !d.in || !d.wpack || d.skip_layer != 3 || d.mode != HOLD_CHAIN_DSP) return HOLD_E_ARG;
  if (d.n_layers != 7 || d.first_chunks != 32 || d.ld_in < 256) return HOLD_E_ARG;
  if (d.ld < 256 || (d.ld & 3) || (d.ld_in & 3)) return HOLD_E_ARG;
  if (((uintptr_t)d.in & 15) || ((uintptr_t)d.wpack & 15)) return HOLD_E_ARG;
  if (((uint64_t)d.P + 128) * (uint64_t)d.ld * 4 >= (1ull << 32)) return HOLD_E_ARG;  // 32-bit byte offsets
  if (((uint64_t)d.P + 128) * (uint64_t)d.ld_in * 4 >= (1ull << 32)) return HOLD_E_ARG;
  RCArgs a = {};
  a.P = (long)d.P; a.wpack = (const char*)d.wpack; a.in = d.in; a.ld_in = d.ld_in; a.ld = d.ld;
  const bool has2 = d.aux2[0] != nullptr;
  for (int l = 0; l < 7; ++l) {
    if (!d.aux1[l] || ((uintptr_t)d.aux1[l] & 15) || ((uintptr_t)d.aux2[l] & 15) || ((uintptr_t)d.out[l] & 15))
      return HOLD_E_ARG;
    if ((d.aux2[l] != nullptr) != has2) return HOLD_E_ARG;
    a.aux1[l] = d.aux1[l]; a.aux2[l] = d.aux2[l]; a.out[l] = d.out[l];
  }
  if (d.P == 0) return HOLD_OK;
  hipStream_t s = (hipStream_t)st;
  return has2 ? rchain_launch<true>(a, s) : rchain_launch<false>(a, s);
}
