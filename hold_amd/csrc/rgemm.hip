// One 256-wide layer per launch with the register-resident recipe of rmlp.hip / rchain.hip (gfx950):
//   C[p][0..255] = epi(A[p][0..K) . W^T + bias)        K = 256 .. 320, epi = none | ReLU | x (aux > 0)
// -- the rendering net's layers (texture_net.py:95-101: lin0 with K = 272 / 304, lin1..3 256 x 256, ReLU), their input
// gradients (x ReLU mask of the stored activation) and lin8's 256 feature rows (shape_net.py:128-130), which hold_gemm_nt_x6
// ran at 124 TF-eq.  Split-precision arithmetic of hold_gemm_nt_x6 (three bf16 limbs of both operands, six products, fp32
// accumulation).
//
// Structure = rchain.hip with successive 128-point BLOCKS in the place of layers: one wave per SIMD owns 32 points,
// D[feature][point] = W (A operand: 24 KiB per k step through a 4-slot LDS ring filled by LDS-DMA, shared by the four waves)
// x input limbs (B operand).  KS = 16 or 20 k steps (K padded to a multiple of 64 with zero weights, so that ring slots and
// k steps stay in phase from block to block).  The B operand of k step e is the wave's own 32 rows x 16 input columns:
// requested four k steps ahead as two lane-linear 1 KiB LDS-DMA pieces into a wave-private side ring (behind the weight
// pieces of the same rendezvous, so the weights' wait never forces the newest input request: rchain.hip), read back as two
// 16-byte fragments and split into limbs behind the MFMAs of k step e - 1.  A lane's 8 values of a k step are the columns
// 16 e + 8 (i / 4) + 4 hh + i % 4 -- the order in which it holds 8 OUTPUT values of an accumulator tile -- and the host packs
// W's columns in that order (field.r6_kmap), so input fragments, mask fragments and result stores share one addressing.
// The finished block stays in the accumulator registers P (AGPRs) while the next one accumulates into Q; its epilogue
// (ReLU / mask from a second side ring) and its 16-byte row-fragment stores, two per k step, are micro-operations in the
// slices behind the next block's MFMAs (explicit per-gap schedule, rmlp.hip).  The first block of a workgroup runs those
// slices against a zero-length buffer descriptor (the hardware drops the stores); the last block's epilogue is exposed.
// Roofline: bf16 MFMA pipe; HBM per point 4 (K + 256) B (+ 1 KiB for the mask operand).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hold_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int NW = 4, BPTS = 32 * NW, PIECE = 1024, SLOT = 24 * PIECE;
constexpr int R3 = 4;         // weight ring slots, DMA distance R3 - 1 k steps
constexpr int SIDE_RING = 4;  // side ring slots: a fragment is requested four k steps before its k step

enum { EPI_NONE = 0, EPI_RELU = 1, EPI_MASK = 2 };

struct RGArgs {
  const float* A; int lda; long P;
  const char* wpack;   // [KS][8 n-tiles][3 limbs][2 halves][32 rows][8 e] bf16, k order of field.r6_kmap
  int KS, K16;         // k steps run (16 or 20), k steps that exist in A (K / 16)
  const float* bias;   // [256] or null (not with EPI_MASK)
  const float* aux; int ld_aux;  // EPI_MASK: C = y * (aux > 0)
  float* C; int ldc;
  float* amax_out;     // [P] max |C[p][:]| (null: not wanted) -- what hold_gemm_h3 (csrc/rgemm_h3.hip) scales its operand rows by
  uint32_t* guard;     // CONDITIONAL launch (hold_gemm_r6_if, the fallback of hold_gemm_h3; protocol of rmlp.hip): null = always run
};

__device__ __forceinline__ uint32_t fbits(float x) { return __builtin_bit_cast(uint32_t, x); }
__device__ __forceinline__ float bitsf(uint32_t x) { return __builtin_bit_cast(float, x); }
__device__ __forceinline__ float relu1(float y) {  // one v_max_i32 (rmlp.hip)
  const int b = __builtin_bit_cast(int, y);
  return __builtin_bit_cast(float, b > 0 ? b : 0);
}

typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const float* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, p ? bytes : 0u, 0x00020000);
}
// column offset in the instruction immediate, soffset = 0 (the gfx950 store-data hazard, rmlp.hip)
__device__ __forceinline__ void store4(const f32x4& v, rsrc_t rs, uint32_t voff) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, voff, 0, 0);
  asm volatile("s_nop 3");
}

struct Limbs { u32x4 l[3]; };

__device__ __forceinline__ void dma_piece(const char* src, uint32_t voff, uint32_t dst) {
  asm volatile(
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %0, %1"
      :
      : "v"(voff), "s"(src), "s"(dst)
      : "memory");
}
#define RG_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

template <int EPI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void rgemm_kernel(RGArgs a) {
  constexpr int NSIDE = EPI == EPI_MASK ? 2 : 1;  // side matrices: the input, and the mask operand
  constexpr int SIDE_SLOT = NSIDE * 2 * PIECE;
  constexpr int OFF_SIDE = R3 * SLOT;
  constexpr int OFF_BIAS = OFF_SIDE + NW * SIDE_RING * SIDE_SLOT;  // (no bias with the mask epilogue: LDS is full there)
  // VMEM operations that may stay in flight at a rendezvous (rchain.hip): the previous rendezvous' six weight pieces and
  // side requests, three stores
  constexpr int NWAIT = 6 + 2 * NSIDE + 3;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (a.guard && __hip_atomic_load(a.guard, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;  // wave-uniform
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hh = lane >> 5, li = lane & 31;
  const uint32_t lane16 = lane * 16;
  float omx = 0.f;   // running maximum of the outputs of the block held in P (amax_out)
  long row_P = -1;   // first row of this wave's share of that block
  const char* ring_lane = smem + lane * 16;
  const uint32_t side_dst0 = (uint32_t)(OFF_SIDE + wave * (SIDE_RING * SIDE_SLOT));
  const float* side_rd = reinterpret_cast<const float*>(smem + OFF_SIDE + wave * (SIDE_RING * SIDE_SLOT)) + lane * 4;
  const int KS = a.KS;

  if (EPI != EPI_MASK) {
    reinterpret_cast<float*>(smem + OFF_BIAS)[tid] = a.bias ? a.bias[tid] : 0.f;
    __syncthreads();
  }

  f32x16 P[8], Q[8];
#pragma unroll
  for (int nt = 0; nt < 8; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) P[nt][r] = 0.f;
  u32x4 A[2][6];
  Limbs Bc, Bn;

  auto init_q = [&]() {  // bias of this lane's rows: features 32 nt + 8 g + 4 hh + k
    if (EPI == EPI_MASK) {
#pragma unroll
      for (int nt = 0; nt < 8; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) Q[nt][r] = 0.f;
      return;
    }
    const float* bl = reinterpret_cast<const float*>(smem + OFF_BIAS) + 4 * hh;
#pragma unroll
    for (int nt = 0; nt < 8; ++nt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(bl + 32 * nt + 8 * g);
#pragma unroll
        for (int k = 0; k < 4; ++k) Q[nt][4 * g + k] = b[k];
      }
  };

  const uint32_t cbytes = (uint32_t)(a.P * a.ldc * 4);
  const uint32_t xbytes = (uint32_t)(a.P * (long)a.ld_aux * 4);

  long blk = blockIdx.x;
  if (blk * BPTS >= a.P) return;
  auto clamp_row = [&](long b) {
    const long r = b * BPTS + wave * 32 + li;
    return r < a.P ? r : a.P - 1;  // reads of rows >= P are redirected (their results are dropped by the stores)
  };
  uint32_t in_off = (uint32_t)((clamp_row(blk) * a.lda + 4 * hh) * 4);  // side-DMA lane offsets of the running block
  uint32_t in_off_next = (uint32_t)((clamp_row(blk + gridDim.x) * a.lda + 4 * hh) * 4);
  uint32_t ax_off_prev = (uint32_t)((clamp_row(blk) * (long)a.ld_aux + 4 * hh) * 4);  // mask rows of the block held in P
  uint32_t ax_off_cur = ax_off_prev;
  const char* Ab = reinterpret_cast<const char*>(a.A);
  const char* Xb = reinterpret_cast<const char*>(a.aux);
  const char* wbase = a.wpack;

  // ---- once per workgroup: the first R3 - 1 weight steps, the input fragments of the k steps 0..3 ----
#pragma unroll
  for (int s0 = 0; s0 < R3 - 1; ++s0)
#pragma unroll
    for (int i = 0; i < 6; ++i)
      dma_piece(a.wpack + (long)s0 * SLOT + wave * (6 * PIECE) + i * PIECE, lane16, (uint32_t)(s0 * SLOT + wave * (6 * PIECE) + i * PIECE));
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      dma_piece(Ab + (16 * e + 8 * h2) * 4, in_off, side_dst0 + e * SIDE_SLOT + h2 * PIECE);
      if (EPI == EPI_MASK) dma_piece(Xb + (16 * e + 8 * h2) * 4, ax_off_prev, side_dst0 + e * SIDE_SLOT + (2 + h2) * PIECE);
    }
  RG_WAIT_VM(0);
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int i = 0; i < 6; ++i) A[0][i] = *reinterpret_cast<const u32x4*>(ring_lane + i * PIECE);
  rsrc_t crs = make_rsrc(nullptr, 0);  // stores of the block held in P (none before the first block is finished)
  const rsrc_t nullrs = make_rsrc(nullptr, 0);
  uint32_t cvoff = 0;

  // Preparation of k step / epilogue unit e (side slot e & 3), four stages of micro-operations:
  //   stage 0: the 8 input values (and mask values) from the side ring, the 8 values of P's unit e
  //   stage 1: epilogue of the 8 values
  //   stage 2 / 3: limb split of the input dwords 0, 1 / 2, 3 (11 operations each, alternating) -> out; one 16-byte store
  struct EpiState { float x[8], y[8], mk[8]; uint32_t w[2][8]; };
  static constexpr int CNT[4] = {8, 12, 23, 23};
  auto mop = [&](int e, bool unit, int stage, int k, Limbs& out, EpiState& st) {
    const int ss = e & 3;
    if (stage == 0) {
      const int i = k;
      if ((i & 3) == 0) {
        const float* sp = side_rd + ss * (SIDE_SLOT / 4) + (i >> 2) * (PIECE / 4);
        const f32x4 xv = *reinterpret_cast<const f32x4*>(sp);
#pragma unroll
        for (int v = 0; v < 4; ++v) st.x[i + v] = xv[v];
        if (EPI == EPI_MASK) {
          const f32x4 mv = *reinterpret_cast<const f32x4*>(sp + 2 * (PIECE / 4));
#pragma unroll
          for (int v = 0; v < 4; ++v) st.mk[i + v] = mv[v];
        }
      }
      if (unit) {
        float t = P[e >> 1][8 * (e & 1) + i];
        asm volatile("" : "+v"(t));
        st.y[i] = t;
      }
    } else if (stage == 1) {
      if (unit && k < 8) {
        float t = st.y[k];
        if (EPI == EPI_RELU) t = relu1(t);
        if (EPI == EPI_MASK) t = st.mk[k] > 0.f ? t : 0.f;
        asm volatile("" : "+v"(t));
        st.y[k] = t;
      } else if (unit) {  // running maximum of the results, one v_max3_f32 per pair
        const int p = k - 8;
        asm volatile("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(omx) : "v"(st.y[2 * p]), "v"(st.y[2 * p + 1]));
      }
    } else {
      const int h2 = stage - 2;
      if (k == 22) {
        // EVERY k step issues its two stores: the rendezvous' vmcnt(NWAIT) counts on the queue's shape -- the k steps without an
        // epilogue unit (e >= 16: the last step of a block, and the four extra steps of K = 272 / 304) used to issue none, and with
        // 8 instead of 12 operations behind the weight pieces it waits for, vmcnt(NWAIT) no longer forced them: in 3 of 3 000
        // launches with K = 304 a workgroup read a ring slot before its pieces had landed (round 6, GPU call 16: the same latent
        // race was in csrc/rgemm.hip since round 3).  Without a unit the store goes to a zero-length descriptor: dropped, counted.
        const f32x4 v = {st.y[4 * h2], st.y[4 * h2 + 1], st.y[4 * h2 + 2], st.y[4 * h2 + 3]};
        store4(v, unit ? crs : nullrs, unit ? cvoff + (16 * e + 8 * h2) * 4 : 0u);
        return;
      }
      const int d = k & 1, op = k >> 1;
      const float x0 = st.x[4 * h2 + 2 * d], x1 = st.x[4 * h2 + 2 * d + 1];
      uint32_t* w = st.w[d];
      uint32_t t;
      if (op == 0) { t = fbits(x0) & 0xffff0000u; asm volatile("" : "+v"(t)); w[0] = t; }
      else if (op == 1) { t = fbits(x1) & 0xffff0000u; asm volatile("" : "+v"(t)); w[1] = t; }
      else if (op == 2) { t = fbits(x0 - bitsf(w[0])); asm volatile("" : "+v"(t)); w[2] = t; }
      else if (op == 3) { t = fbits(x1 - bitsf(w[1])); asm volatile("" : "+v"(t)); w[3] = t; }
      else if (op == 4) { t = w[2] & 0xffff0000u; asm volatile("" : "+v"(t)); w[4] = t; }
      else if (op == 5) { t = w[3] & 0xffff0000u; asm volatile("" : "+v"(t)); w[5] = t; }
      else if (op == 6) { t = fbits(bitsf(w[2]) - bitsf(w[4])); asm volatile("" : "+v"(t)); w[6] = t; }
      else if (op == 7) { t = fbits(bitsf(w[3]) - bitsf(w[5])); asm volatile("" : "+v"(t)); w[7] = t; }
      else if (op == 8) { t = __builtin_amdgcn_perm(fbits(x1), fbits(x0), 0x07060302u); asm volatile("" : "+v"(t)); out.l[0][2 * h2 + d] = t; }
      else if (op == 9) { t = __builtin_amdgcn_perm(w[3], w[2], 0x07060302u); asm volatile("" : "+v"(t)); out.l[1][2 * h2 + d] = t; }
      else { t = __builtin_amdgcn_perm(w[7], w[6], 0x07060302u); asm volatile("" : "+v"(t)); out.l[2][2 * h2 + d] = t; }
    }
  };

  // One k step j (rchain.hip:kstep): 4 groups x 12 MFMAs; behind every MFMA one fragment read (gaps 0..5), in the group
  // behind the rendezvous the six weight pieces of stream step gs + R3 - 1, in the last group -- BEHIND them in the queue --
  // the side fragments of k step j + 4 (of the next block once j + 4 >= KS), and cnt[group] / 12 micro-operations.
  auto kstep = [&](int j, const int (&cnt)[4], auto&& mp) {
    // KS % R3 == 0 and every block starts a new pass over the stream: ring slots are compile-time functions of j
    const int slot = j % R3, nslot = (j + 1) % R3, fslot = (j + R3 - 1) % R3;
    int jw = j + R3 - 1;
    jw = jw >= KS ? jw - KS : jw;
    const char* wsrc = wbase + (long)jw * SLOT + wave * (6 * PIECE);
    const uint32_t wdst = (uint32_t)(fslot * SLOT + wave * (6 * PIECE));
    const int e4 = j + 4;
    const bool wrap = e4 >= KS;
    const int ec = wrap ? e4 - KS : e4;  // k step (and epilogue unit) the side request is for
    const uint32_t sd = side_dst0 + (j & 3) * SIDE_SLOT;
    const char* s1 = Ab + 64 * (ec < a.K16 ? ec : 0);  // padded k steps (zero weights) re-read k step 0: never past a row
    const uint32_t o1 = wrap ? in_off_next : in_off;
    const char* s2 = Xb + 64 * (ec < 16 ? ec : 15);
    const uint32_t o2 = wrap ? ax_off_cur : ax_off_prev;  // unit ec of the block in P, or (wrapped) of the running block
#pragma unroll
    for (int pair = 0; pair < 4; ++pair) {
      if (pair == 2) {  // rendezvous: the weights of stream step gs + 1 have landed in every wave; slot gs - 1 is free
        RG_WAIT_VM(NWAIT);
        __builtin_amdgcn_s_barrier();
      }
      const char* rd = ring_lane + (pair < 3 ? slot * SLOT + (pair + 1) * (6 * PIECE) : nslot * SLOT);
#pragma unroll
      for (int m = 0; m < 12; ++m) {
        const int pr = m >> 1, tt = m & 1;
        const int wl = (pr == 2 || pr == 3) ? 1 : (pr == 5 ? 2 : 0);  // (w limb, act limb): 00 01 10 11 02 20
        const int al = (pr == 1 || pr == 3) ? 1 : (pr == 4 ? 2 : 0);
        Q[2 * pair + tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[pair & 1][3 * tt + wl]),
                                                                  __builtin_bit_cast(bf16x8, Bc.l[al]), Q[2 * pair + tt], 0, 0, 0);
        if (m < 6) A[(pair + 1) & 1][m] = *reinterpret_cast<const u32x4*>(rd + m * PIECE);
        if (pair == 2 && (m & 1) == 0) dma_piece(wsrc + (m >> 1) * PIECE, lane16, wdst + (m >> 1) * PIECE);
        if (pair == 3 && (m == 1 || m == 4)) dma_piece(s1 + 32 * (m == 4), o1, sd + (m == 4) * PIECE);
        if (pair == 3 && NSIDE == 2 && (m == 7 || m == 10)) dma_piece(s2 + 32 * (m == 10), o2, sd + 2 * PIECE + (m == 10) * PIECE);
#pragma unroll
        for (int u = 0; u < 3; ++u) {  // constant trip count (the slice bounds fold once pair and m are unrolled)
          const int k = cnt[pair] * m / 12 + u;
          if (k < cnt[pair] * (m + 1) / 12) mp(pair, k);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    Bc = Bn;
  };
  static constexpr int CNT_NONE[4] = {0, 0, 0, 0};
  auto no_mop = [](int, int) {};

  for (; blk * BPTS < a.P; blk += gridDim.x) {
    // the stream bases are made opaque once per block: loop-invariant, the ~160 DMA source addresses of a block would be
    // hoisted out of the block loop and live in spilled scalar registers (two v_readlane per DMA) instead of two s_add each
    asm volatile("" : "+s"(wbase), "+s"(Ab), "+s"(Xb));
#pragma unroll
    for (int nt = 0; nt < 8; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) asm volatile("" : "+a"(P[nt][r]));  // the finished block lives in the AGPR half
    init_q();
    EpiState st;
    // k step 0 / unit 0: not overlapped (once per block)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int k = 0; k < 23; ++k)
        if (k < CNT[c]) mop(0, true, c, k, Bc, st);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (j + 1 < 16)
        kstep(j, CNT, [&](int c, int k) { mop(j + 1, true, c, k, Bn, st); });
      else  // k step 16 exists only when KS = 20; its input fragment sits in side slot 0 either way
        kstep(j, CNT, [&](int c, int k) { mop(j + 1, false, c, k, Bn, st); });
    }
    if (KS > 16) {  // K padded to 320 (wave-uniform)
#pragma unroll
      for (int j = 16; j < 20; ++j) kstep(j, CNT, [&](int c, int k) { mop(j + 1, false, c, k, Bn, st); });
    }
    // ---- every unit of the block held in P has been stored: its rows' maxima (both lane halves hold the same point) ----
    if (a.amax_out && row_P >= 0) {
      const float o = fmaxf(omx, __shfl_xor(omx, 32));
      if (hh == 0 && row_P + li < a.P) a.amax_out[row_P + li] = o;
    }
    omx = 0.f;
    row_P = blk * BPTS + wave * 32;
    // ---- the block is finished: it becomes P; its stores run behind the next block ----
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) P[nt] = Q[nt];
    const long row = blk * BPTS + wave * 32 + li;  // unclamped: the buffer range check drops rows >= P
    crs = make_rsrc(a.C, cbytes);
    cvoff = (uint32_t)((row * a.ldc + 4 * hh) * 4);
    in_off = in_off_next;
    in_off_next = (uint32_t)((clamp_row(blk + 2 * (long)gridDim.x) * a.lda + 4 * hh) * 4);
    ax_off_prev = ax_off_cur;
    ax_off_cur = (uint32_t)((clamp_row(blk + gridDim.x) * (long)a.ld_aux + 4 * hh) * 4);
  }
  RG_WAIT_VM(0);  // no LDS-DMA in flight when the workgroup's LDS is released
  (void)no_mop; (void)CNT_NONE; (void)xbytes;
  // ---- epilogue of the last block (exposed): mask rows by ordinary buffer loads ----
  {
    const rsrc_t xrs = make_rsrc(EPI == EPI_MASK ? a.aux : nullptr, xbytes);
#pragma unroll
    for (int nt = 0; nt < 8; ++nt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const uint32_t co = (32 * nt + 8 * g) * 4;
        u32x4 mv = {0u, 0u, 0u, 0u};
        if (EPI == EPI_MASK) mv = __builtin_amdgcn_raw_buffer_load_b128(xrs, ax_off_prev + co, 0, 0);
        f32x4 v;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float t = P[nt][4 * g + k];
          if (EPI == EPI_RELU) t = relu1(t);
          if (EPI == EPI_MASK) t = bitsf(mv[k]) > 0.f ? t : 0.f;
          v[k] = t;
          omx = fmaxf(omx, fabsf(t));
        }
        store4(v, crs, cvoff + co);
      }
    if (a.amax_out && row_P >= 0) {
      const float o = fmaxf(omx, __shfl_xor(omx, 32));
      if (hh == 0 && row_P + li < a.P) a.amax_out[row_P + li] = o;
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

template <int EPI>
int launch(const RGArgs& a, hipStream_t s) {
  constexpr int NSIDE = EPI == EPI_MASK ? 2 : 1;
  constexpr int lds = R3 * SLOT + NW * SIDE_RING * NSIDE * 2 * PIECE + (EPI == EPI_MASK ? 0 : 1024);
  static_assert(lds <= 160 * 1024, "LDS budget");
  static int n_cu = 0;
  static bool attr_set = false;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return HOLD_E_LAUNCH;
    n_cu = prop.multiProcessorCount;
  }
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)rgemm_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      return HOLD_E_LAUNCH;
    attr_set = true;
  }
  const long blocks = (a.P + BPTS - 1) / BPTS;
  hipLaunchKernelGGL((rgemm_kernel<EPI>), dim3((unsigned)(blocks < n_cu ? blocks : n_cu)), dim3(256), lds, s, a);
  return hipGetLastError() == hipSuccess ? HOLD_OK : HOLD_E_LAUNCH;
}

}  // namespace

extern "C" int64_t hold_gemm_r6_pack_bytes(int32_t K) { return (int64_t)((K + 63) / 64 * 4) * SLOT; }

// C[P][256] (row stride ldc) = epi(A[P][K] . W^T + bias): epilogue 0 none, 1 ReLU, 2 multiply by (aux > 0) (no bias).
// wpack = hold_gemm_r6_pack_bytes(K) bytes of bf16, [KS k steps j][8 n-tiles nt][3 limbs t][2 halves h][32 rows i][8 e] =
// limb_t(W)[32 nt + i][16 j + 8 (e / 4) + 4 h + e % 4], KS = 4 ceil(K / 64), zero for columns >= K and rows >= N.
// K a multiple of 16 in 256 .. 320 (the padded k steps multiply columns 0..15 of A by the zero weights: no read past a
// row).  32-bit offsets: P * max(lda, ldc, ld_aux) * 4 < 2^32.
extern "C" int hold_gemm_r6_if(const float* A, int32_t lda, int64_t P, const void* wpack, int32_t K, const float* bias,
                               int32_t epilogue, const float* aux, int32_t ld_aux, float* C, int32_t ldc, float* amax_out,
                               uint32_t* guard, hold_stream_t st);
extern "C" int hold_gemm_r6(const float* A, int32_t lda, int64_t P, const void* wpack, int32_t K, const float* bias,
                            int32_t epilogue, const float* aux, int32_t ld_aux, float* C, int32_t ldc, hold_stream_t st) {
  return hold_gemm_r6_if(A, lda, P, wpack, K, bias, epilogue, aux, ld_aux, C, ldc, nullptr, nullptr, st);
}
// ... with the per-row maxima of C as a second output (amax_out [P] or NULL) and as a CONDITIONAL launch (guard != NULL: see
// hold_fused_sdf_r6_if in include/hold_hip.h) -- the fallback of hold_gemm_h3
extern "C" int hold_gemm_r6_if(const float* A, int32_t lda, int64_t P, const void* wpack, int32_t K, const float* bias,
                               int32_t epilogue, const float* aux, int32_t ld_aux, float* C, int32_t ldc, float* amax_out,
                               uint32_t* guard, hold_stream_t st) {
  if (((uintptr_t)amax_out & 3) || ((uintptr_t)guard & 3)) return HOLD_E_ARG;
  if (!A || !wpack || !C || P < 0 || K < 256 || K > 320 || (K & 15) || lda < K || (lda & 3) || ldc < 256 || (ldc & 3)) return HOLD_E_ARG;
  if (((uintptr_t)A & 15) || ((uintptr_t)C & 15) || ((uintptr_t)wpack & 15) || (bias && ((uintptr_t)bias & 15))) return HOLD_E_ARG;
  if (epilogue < 0 || epilogue > 2) return HOLD_E_ARG;
  if (epilogue == 2 && (!aux || bias || ld_aux < 256 || (ld_aux & 3) || ((uintptr_t)aux & 15))) return HOLD_E_ARG;
  if (P == 0) return HOLD_OK;
  const int64_t ldmax = lda > ldc ? (lda > ld_aux ? lda : ld_aux) : (ldc > ld_aux ? ldc : ld_aux);
  if (((uint64_t)P + BPTS) * (uint64_t)ldmax * 4 >= (1ull << 32)) return HOLD_E_ARG;
  RGArgs a;
  a.A = A; a.lda = lda; a.P = (long)P; a.wpack = (const char*)wpack; a.KS = (K + 63) / 64 * 4; a.K16 = K / 16; a.bias = bias; a.aux = aux;
  a.ld_aux = ld_aux; a.C = C; a.ldc = ldc; a.amax_out = amax_out; a.guard = guard;
  hipStream_t s = (hipStream_t)st;
  return epilogue == 0 ? launch<EPI_NONE>(a, s) : epilogue == 1 ? launch<EPI_RELU>(a, s) : launch<EPI_MASK>(a, s);
}
