// One 256-wide layer per launch, register-resident, in the TWO-LIMB fp16 arithmetic "f16x3" (gfx950): the kernel of csrc/rgemm.hip
//   C[p][0..255] = epi(A[p][0..K) . W^T + bias)        K = 256 .. 320, epi = none | ReLU | x (aux > 0)
// -- the rendering net's layers (texture_net.py:95-101), their input gradients and lin8's 256 feature rows (shape_net.py:128-130)
// -- with 24 instead of 48 matrix instructions per k step (x = hi + lo, hi = RN_f16(s x), lo = RN_f16(s x - hi); hi hi + hi lo +
// lo hi on v_mfma_f32_32x32x16_f16; csrc/rmlp_h3.hip), 16 KiB of weight limbs per k step instead of 24.
//
// SCALES.  Weights: per matrix, at pack time (s_w = 2^k, max |W| s_w in [2^13, 2^14); c3 = 1 / s_w comes with the stream).  The B
// operand is a ROW of A per point -- an activation of O(1) in the forward layers, a loss cotangent of any magnitude in the input-
// gradient launches -- and an MFMA column is a point, so every point carries its own power-of-two scale 2^k (csrc/rchain_h3.hip).
// Here the row's magnitude is KNOWN before its first k step: the launch that produced A reports the exact maximum of every row it
// wrote (amax_out, 4 bytes per point, from the values in its epilogue registers), and this launch reads it (amax_in; a lane loads
// its next block's value a whole block ahead) -- max(amax_in, amax_floor) goes to [2^12, 2^13): 2^3 of headroom
// for the columns of A the producer did not write (the rendering net's first layer reads [features | xc | normal | pose | time]:
// amax_in covers the features, amax_floor the rest).  Without amax_in the scale is the constant one of amax_floor (lin8's input, the
// trunk's last softplus output: amax_floor = 64 is the 2^6 of rmlp_h3.hip).  OVERFLOW GUARD as in rmlp_h3.hip: exact running maximum
// of the scaled values, guard word, conditional f32x6 launch (hold_gemm_r6_if) behind the kernel.
// Structure, rings, queue accounting: csrc/rgemm.hip (4-slot weight ring = 64 KiB here, wave-private side rings: side_dist()).
// Roofline: fp16 MFMA pipe at 3 limb products per product; HBM per point 4 (K + 256) B (+ 1 KiB mask operand) + 8 B of maxima.
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

constexpr int NW = 4, BPTS = 32 * NW, PIECE = 1024, SLOT = 16 * PIECE;
constexpr int R3 = 4;         // weight ring slots, DMA distance R3 - 1 k steps
// Side ring (per wave): a k step's input fragment is requested side_dist() k steps ahead into slot (k step % side_ring()).  With ONE
// side matrix: five steps ahead, eight slots (LDS 129 KiB); with the mask operand as a second one: four / four (129 KiB again).
// What the distance buys is set by the in-order vector-memory queue: the rendezvous of step j must have the weights issued in step
// j - 2 (8 + 4 NSIDE younger operations), so at most that many operations stay in flight, and a side request is forced as soon as
// the weight pieces issued right behind it are -- 1.75 k steps after its issue at distance 4 (it is consumed before the NEXT
// rendezvous, which allows 7 + 2 NSIDE), 2.75 at distance 5 (the weights' own bound, 8 + 4 NSIDE = 12).  At ~8 KiB of side requests
// per workgroup and step the bytes in flight are what bounds a launch (256 x 14 KiB / ~1 us = 3.6 TB/s measured before).
constexpr int side_dist(int nside) { return nside == 1 ? 5 : 4; }
constexpr int side_ring(int nside) { return nside == 1 ? 8 : 4; }

// MASKB: the mask as one BIT per element (bits_in) instead of aux; RELUB: ReLU that also writes its mask as bits (bits_out)
enum { EPI_NONE = 0, EPI_RELU = 1, EPI_MASK = 2, EPI_MASKB = 3, EPI_RELUB = 4 };

struct RGArgs {
  const float* A; int lda; long P;
  const char* wpack;   // [KS][8 n-tiles][2 limbs][2 halves][32 rows][8 e] fp16 of s_w W, k order of field.r6_kmap
  const float* c3;     // device scalar 1 / s_w
  const float* amax_in; float amax_floor;  // per-row maximum of A (null: amax_floor alone)
  float* amax_out;     // per-row maximum of |C| (null: not wanted)
  uint32_t* guard;     // overflow guard word (null: unreported)
  int KS, K16;         // k steps run (16 or 20), k steps that exist in A (K / 16)
  const float* bias;   // [256] or null (not with EPI_MASK)
  const float* aux; int ld_aux;  // EPI_MASK: C = y * (aux > 0)
  // ReLU masks as bits (round 6): [P][8] dwords, bit n of a row = (column n of the ReLU launch's result > 0).  EPI_RELU writes them
  // (bits_out, optional) from the values in its epilogue registers; EPI_MASKB reads them (bits_in) instead of streaming the [P][256]
  // fp32 activation a second time -- 32 bytes per point instead of 1 KiB, and the masked launch has ONE side matrix
  uint32_t* bits_out; const uint32_t* bits_in;
  float* C; int ldc;
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

struct Limbs { u32x4 l[2]; };

__device__ __forceinline__ uint32_t cvt_pk(float a, float b) {  // v_cvt_pk_f16_f32 (round to nearest)
  const f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
}
template <int SEL>
__device__ __forceinline__ float resid(uint32_t hi, float x) {  // x - (float) half SEL of hi: one v_fma_mix_f32, exact
  float r;
  if (SEL == 0)
    asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hi), "v"(x));
  else
    asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hi), "v"(x));
  return r;
}
__device__ __forceinline__ float pow2f(int k) { return bitsf((uint32_t)(127 + k) << 23); }
// exponent k that puts a row maximum m at [2^12, 2^13) (|k| <= 96; m = 0, inf, NaN: 0)
__device__ __forceinline__ int row_scale(float m) {
  const int e = (int)((fbits(m) >> 23) & 0xffu);
  int k = 127 + 12 - e;
  k = k > 96 ? 96 : (k < -96 ? -96 : k);
  return (e != 0 && e != 255) ? k : 0;
}
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

// ABL (developer build, results garbage, timing only): 1 = the input / mask pieces are requested in the SHAPE of whole lines (lane L -> row
// 8 i + L / 8, 16-byte chunk L % 8 of a 32-column tile: the four pieces of two k steps cover a [32 rows][128 B] tile, as csrc/rchain_h3.hip's
// side tiles do) instead of 32-byte row fragments; 2 = the result stores too.  Same bytes, same instruction counts, same queue: what the
// access shape alone costs -- the upper bound of what whole-line tiles through LDS could gain this kernel.  3 / 4: column rotation per
// workgroup (without / with whole lines).  5: every k step re-reads the weights of stream step 0 (hot in L1 / L2); 6: every block re-reads
// the input / mask rows of the workgroup's first block (hot in the L2); 7: no result stores -- which of its three streams the kernel waits for
template <int EPI, int ABL = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void rgemm_h3_kernel(RGArgs a) {
  constexpr int NSIDE = EPI == EPI_MASK ? 2 : 1;  // side matrices: the input, and the mask operand
  constexpr int SIDE_SLOT = NSIDE * 2 * PIECE;
  constexpr int SD = side_dist(NSIDE), SIDE_RING = side_ring(NSIDE);
  constexpr int OFF_SIDE = R3 * SLOT;
  constexpr int OFF_BIAS = OFF_SIDE + NW * SIDE_RING * SIDE_SLOT;
  constexpr int OFF_RT = OFF_BIAS + 1024;  // wave-private result tiles, [32 rows][128 B] each (see "result stores" below)
  // VMEM operations that may stay in flight at a rendezvous.  Queue of a k step: W x 4, store, S x 2 NSIDE, store.  The rendezvous of
  // step j needs (a) the weights issued in step j - 2: 8 + 4 NSIDE younger operations; (b) the side fragments consumed before the
  // next rendezvous, i.e. the k step j + 2's, issued in step j + 2 - SD: 7 + 2 NSIDE younger operations at SD = 4, one whole step
  // (6 + 2 NSIDE) more at SD = 5.  NWAIT = the smaller bound.  The two extra operations of a block (the load of the next block's
  // row maxima, the store of the finished block's) only make a needed operation OLDER in the queue: never unsafe
  constexpr int NWAIT = SD == 4 ? 7 + 2 * NSIDE : 8 + 4 * NSIDE;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hh = lane >> 5, li = lane & 31;
  const uint32_t lane16 = lane * 16;
  const char* ring_lane = smem + lane * 16;
  const uint32_t side_dst0 = (uint32_t)(OFF_SIDE + wave * (SIDE_RING * SIDE_SLOT));
  const float* side_rd = reinterpret_cast<const float*>(smem + OFF_SIDE + wave * (SIDE_RING * SIDE_SLOT)) + lane * 4;
  const int KS = a.KS;

  constexpr bool MASKED = EPI == EPI_MASK || EPI == EPI_MASKB;
  constexpr bool RELU = EPI == EPI_RELU || EPI == EPI_RELUB;
  reinterpret_cast<float*>(smem + OFF_BIAS)[tid] = (!MASKED && a.bias) ? a.bias[tid] : 0.f;
  __syncthreads();
  const float c3 = *a.c3, s_w = 1.0f / c3;  // exact powers of two
  // per-point scale state: kB / sB = exponent / scale of the B operand of the block being accumulated (its row maximum, or the
  // floor, at [2^12, 2^13)); yscP = what un-scales the block held in P; mx = exact running maximum of the scaled values (guard);
  // omx = running maximum of the finished block's outputs (amax_out)
  int kB = 0;
  float sB = 1.f, yscP = 0.f, mx = 0.f, omx = 0.f;
  const float* AMb = a.amax_in;

  f32x16 P[8], Q[8];
#pragma unroll
  for (int nt = 0; nt < 8; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) P[nt][r] = 0.f;
  u32x4 A[2][4];
  Limbs Bc, Bn;

  auto init_q = [&]() {  // bias of this lane's rows (features 32 nt + 8 g + 4 hh + k), in the accumulators' scale s_w 2^kB
    const float swB = s_w * sB;
    if (MASKED) {
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
        for (int k = 0; k < 4; ++k) Q[nt][4 * g + k] = b[k] * swB;
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
  // this lane's row maximum of the running block, and -- requested a whole block ahead -- of the next one.  The request is an
  // assembler load straight into an ACCUMULATOR register: a load the compiler knows about made it put s_waitcnt vmcnt(0) behind
  // the request (it cannot count the LDS-DMA pieces in the queue, so it drains it -- once per block, with the load's own HBM
  // latency exposed); the value is read a block later, when the in-order queue has retired it long ago (>= 128 younger
  // operations, every rendezvous leaves at most NWAIT of them in flight).  An earlier version moved the 32 values per wave by a
  // 4-byte LDS-DMA: results within tolerance but not bit-reproducible (the row scales depended on stale LDS) -- GPU call 11.
  long row_P = -1;  // first row of this wave's share of the block held in P (none yet)
  float am_cur = AMb ? AMb[clamp_row(blk)] : 0.f;
  float am_next = 0.f;
  // mask bits.  A lane owns, of every 16-column unit e, the columns 16 e + 8 h2 + 4 hh + k: in dword e / 2 of its row the bits
  // 16 (e % 2) + 8 h2 + 4 hh + k.  Both directions work on the dword shifted by 4 hh, so that every position is an immediate:
  // bo[d] collects the lane's own bits of the block held in P (EPI_RELU; shifted back, OR-ed with the other lane half and stored at the
  // block's end), mb[d] holds the row's dword >> 4 hh (EPI_MASKB; requested a whole block ahead, like the row maxima)
  uint32_t bo[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
  uint32_t mb[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
  u32x4 mbn0 = {0u, 0u, 0u, 0u}, mbn1 = {0u, 0u, 0u, 0u};
  constexpr bool want_bits = EPI == EPI_RELUB;
  auto request_bits = [&](long b) {  // the rows of block b (both lane halves of a point load the same 32 bytes)
    const uint32_t* ptr = a.bits_in + clamp_row(b) * 8;
    asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:16" : "=a"(mbn0), "=a"(mbn1) : "v"(ptr) : "memory");
  };
  auto store_bits = [&]() {  // of the block held in P (rows row_P ..): lane half 0 stores the point's 32 bytes
    u32x4 w0, w1;
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      uint32_t w = bo[d] << (4 * hh);
      w |= (uint32_t)__shfl_xor((int)w, 32);
      if (d < 4) w0[d] = w; else w1[d - 4] = w;
      bo[d] = 0u;
    }
    if (hh == 0 && row_P + li < a.P) {
      u32x4* dst = reinterpret_cast<u32x4*>(a.bits_out + (row_P + li) * 8);
      dst[0] = w0;
      dst[1] = w1;
    }
  };
  auto request_amax = [&](long b) {
    const float* ptr = AMb + clamp_row(b);
    asm volatile("global_load_dword %0, %1, off" : "=a"(am_next) : "v"(ptr) : "memory");
  };
  const char* Ab = reinterpret_cast<const char*>(a.A);
  const char* Xb = reinterpret_cast<const char*>(a.aux);
  const char* wbase = a.wpack;

  // ---- once per workgroup: the first R3 - 1 weight steps, the input fragments of the k steps 0..SD-1 ----
#pragma unroll
  for (int s0 = 0; s0 < R3 - 1; ++s0)
#pragma unroll
    for (int i = 0; i < 4; ++i)
      dma_piece(a.wpack + (long)s0 * SLOT + wave * (4 * PIECE) + i * PIECE, lane16, (uint32_t)(s0 * SLOT + wave * (4 * PIECE) + i * PIECE));
#pragma unroll
  for (int e = 0; e < SD; ++e)
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      dma_piece(Ab + (16 * e + 8 * h2) * 4, in_off, side_dst0 + e * SIDE_SLOT + h2 * PIECE);
      if (EPI == EPI_MASK) dma_piece(Xb + (16 * e + 8 * h2) * 4, ax_off_prev, side_dst0 + e * SIDE_SLOT + (2 + h2) * PIECE);
    }
  RG_WAIT_VM(0);
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int i = 0; i < 4; ++i) A[0][i] = *reinterpret_cast<const u32x4*>(ring_lane + i * PIECE);
  rsrc_t crs = make_rsrc(nullptr, 0);  // stores of the block held in P (none before the first block is finished)
  const rsrc_t nullrs = make_rsrc(nullptr, 0);
  uint32_t cvoff = 0;
  // ---- result stores as WHOLE LINES (csrc/rchain_h3.hip's result tiles).  A unit's two 16-byte fragments per lane -- 32 rows x 32
  // bytes per store instruction, four partial writes per 128-byte line that only the L2 merges -- cost this kernel 20-30 % of its
  // time (developer-build ablations, GPU calls 38 / 39: no stores 0.66-0.92 ms, whole-line stores 0.83-1.07, fragments 0.96-1.34).
  // Now the epilogue writes them into a wave-private tile [32 rows][128 B] of two units (position p of row r holds chunk p ^ f(r):
  // the sweeps' swizzle, conflict-free both ways); one k step after the tile's second unit it is read back lane-linear -- piece i =
  // rows 8 i .. 8 i + 7, eight lanes per row -- and leaves as four stores of eight whole lines each, two per k step, so the queue
  // keeps its shape: unit e >= 2 even: read back tile e / 2 - 1 (stage 0), store its pieces 0, 1; e >= 3 odd: pieces 2, 3.  The last
  // tile of a block is read back with "unit" 16 (the block's last k step); with KS = 16 its pieces 2, 3 leave at the next block's
  // unit 0 with the OLD block's row offsets and descriptor (sv_old, crs_prev), or behind the loop.
  char* rt_lane_w = smem + OFF_RT + wave * 4096 + li * 128;  // + ((chunk ^ fl) << 4)
  const int fl = ((li >> 1) & 7) ^ ((li & 1) << 2);
  const char* rt_lane_r = smem + OFF_RT + wave * 4096 + lane * 16;  // + 1024 i
  const int r8 = lane >> 3, p8 = lane & 7;
  const int f0 = (r8 >> 1) ^ ((r8 & 1) << 2);  // f(8 i + r8) = f0 ^ ((i & 1) << 2)
  f32x4 rb[4];
  uint32_t sv[4] = {0u, 0u, 0u, 0u}, sv_old[2] = {0u, 0u};
  rsrc_t crs_prev = make_rsrc(nullptr, 0);
#pragma unroll
  for (int i = 0; i < 4; ++i) rb[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Preparation of k step / epilogue unit e (side slot e & 3), four stages of micro-operations:
  //   stage 0: the 8 input values (and mask values) from the side ring, the 8 values of P's unit e (un-scaled)
  //   stage 1: epilogue of the 8 values, running maximum of the results (amax_out)
  //   stage 2 / 3: two-limb fp16 split of the input dwords 0, 1 / 2, 3 (6 operations each, alternating: scale by the point's
  //                2^kB, exact running maximum, hi, two residuals, lo) -> out; one 16-byte store
  struct EpiState { float x[8], y[8], mk[8]; float xs[2][2], ra[2], rb[2]; uint32_t hi[2]; };
  static constexpr int CNT[4] = {8, 12, 13, 13};
#define RG_PIN(x) asm volatile("" : "+v"(x))
  // track: does the k step exist (its values count for the guard)?  0 = no (the slot holds the NEXT block's k step 0, split again at
  // its block start with ITS scale), 1 = yes, 2 = only with KS = 20
  auto mop = [&](int e, bool unit, int stage, int k, Limbs& out, EpiState& st, int track = 1) {
    const int ss = (e == 20 ? 0 : e) % SIDE_RING;  // (unit 20 = the next block's k step 0 when KS = 20; unit 16 when KS = 16: slot 0 too)
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
        float t = P[e >> 1][8 * (e & 1) + i] * yscP;
        RG_PIN(t);
        st.y[i] = t;
      }
      if (ABL == 0 && e >= 2 && (e & 1) == 0 && e <= 16 && (i & 1) == 1) {  // i = 1, 3, 5, 7: piece i / 2 of the finished tile
        rb[i >> 1] = *reinterpret_cast<const f32x4*>(rt_lane_r + (i >> 1) * 1024);
      }
    } else if (stage == 1) {
      if (!unit) return;
      if (k < 8) {
        float t = st.y[k];
        if (RELU) t = relu1(t);
        if (EPI == EPI_MASK) t = st.mk[k] > 0.f ? t : 0.f;
        if (EPI == EPI_MASKB) {  // bit 16 (e % 2) + 8 (k / 4) + k % 4 of the shifted dword: all ones or zero
          const int m = __builtin_amdgcn_sbfe((int)mb[e >> 1], 16 * (e & 1) + 8 * (k >> 2) + (k & 3), 1);
          t = bitsf(fbits(t) & (uint32_t)m);
        }
        RG_PIN(t);
        st.y[k] = t;
        if (EPI == EPI_RELUB) {  // (t is +0 or positive after the ReLU: its bits are non-zero exactly when it is > 0)
          const uint32_t one = fbits(t) < 1u ? fbits(t) : 1u;
          bo[e >> 1] |= one << (16 * (e & 1) + 8 * (k >> 2) + (k & 3));
        }
      } else {
        const int p = k - 8;
        asm volatile("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(omx) : "v"(st.y[2 * p]), "v"(st.y[2 * p + 1]));
      }
    } else {
      const int h2 = stage - 2;
      if (k == 12) {
        // EVERY k step issues its two stores: the rendezvous' vmcnt(NWAIT) counts on the queue's shape -- the k steps without an
        // epilogue unit (e >= 16: the last step of a block, and the four extra steps of K = 272 / 304) used to issue none, and with
        // 8 instead of 12 operations behind the weight pieces it waits for, vmcnt(NWAIT) no longer forced them: in 3 of 3 000
        // launches with K = 304 a workgroup read a ring slot before its pieces had landed (round 6, GPU call 16: the same latent
        // race was in csrc/rgemm.hip since round 3).  Without a unit the store goes to a zero-length descriptor: dropped, counted.
        const f32x4 v = {st.y[4 * h2], st.y[4 * h2 + 1], st.y[4 * h2 + 2], st.y[4 * h2 + 3]};
        if (ABL == 7) {  // timing ablation: no result traffic (the store is issued to the zero-length descriptor: dropped, counted)
          store4(v, nullrs, 0u);
          return;
        }
        if ((ABL == 2 || ABL == 8) && unit) {  // (8: whole-line stores with the product's fragment loads)  // timing ablation: the same 1 KiB per instruction as 8 whole lines of the unit pair's [32][128 B] tile
          const long r = row_P + 8 * (2 * (e & 1) + h2) + (lane >> 3);
          store4(v, crs, (uint32_t)((r * a.ldc + 32 * (e >> 1) + 4 * (lane & 7)) * 4));
          return;
        }
        if (ABL != 0) {  // (the ablations keep the fragment stores they were written against)
          store4(v, unit ? crs : nullrs, unit ? cvoff + (16 * e + 8 * h2) * 4 : 0u);
          return;
        }
        // the store of this slot: a piece of the tile finished before (see "result stores" above), or none (zero-length descriptor)
        if (e == 0) store4(rb[2 + h2], KS > 16 ? nullrs : crs_prev, sv_old[h2] + 128 * 7);
        else if (e == 1 || e > 17) store4(rb[0], nullrs, 0u);
        else if ((e & 1) == 0) store4(rb[h2], crs, sv[h2] + 128 * (e / 2 - 1));
        else store4(rb[2 + h2], crs, sv[2 + h2] + 128 * ((e - 3) / 2));
        if (unit)  // this unit's fragment into the tile: chunk 4 (e % 2) + 2 h2 + hh of row li
          *reinterpret_cast<f32x4*>(rt_lane_w + (((4 * (e & 1) + 2 * h2 + hh) ^ fl) << 4)) = v;
        return;
      }
      const int d = k & 1, op = k >> 1;
      if (op == 0) {
        st.xs[d][0] = st.x[4 * h2 + 2 * d] * sB;
        st.xs[d][1] = st.x[4 * h2 + 2 * d + 1] * sB;
        RG_PIN(st.xs[d][0]);
        RG_PIN(st.xs[d][1]);
      } else if (op == 1) {
        if (track == 1) asm volatile("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(mx) : "v"(st.xs[d][0]), "v"(st.xs[d][1]));
        if (track == 2) {
          float t = mx;
          asm volatile("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(t) : "v"(st.xs[d][0]), "v"(st.xs[d][1]));
          mx = KS > 16 ? t : mx;
        }
      } else if (op == 2) {
        uint32_t hi = cvt_pk(st.xs[d][0], st.xs[d][1]);
        RG_PIN(hi);
        st.hi[d] = hi;
        out.l[0][2 * h2 + d] = hi;
      } else if (op == 3) st.ra[d] = resid<0>(st.hi[d], st.xs[d][0]);
      else if (op == 4) st.rb[d] = resid<1>(st.hi[d], st.xs[d][1]);
      else {
        uint32_t lo = cvt_pk(st.ra[d], st.rb[d]);
        RG_PIN(lo);
        out.l[1][2 * h2 + d] = lo;
      }
    }
  };

  // One k step j (rgemm.hip:kstep with 6 instead of 12 MFMAs per group): 4 groups x 6 MFMAs (hi hi, hi lo, lo hi for two n-tiles);
  // behind every MFMA one fragment read (gaps 0..3), in the group behind the rendezvous the four weight pieces of stream step
  // gs + R3 - 1, in the last group -- BEHIND them in the queue -- the side fragments of k step j + SD (of the next block once
  // j + SD >= KS), and cnt[group] / 6 micro-operations.
  constexpr bool WL = ABL == 1 || ABL == 2 || ABL == 4;
  uint32_t abl_o1[2] = {0u, 0u}, abl_o2[2] = {0u, 0u};
  const uint32_t abl_in0 = in_off, abl_ax0 = ax_off_prev;
  auto kstep = [&](int j, const int (&cnt)[4], auto&& mp) {
    // KS % R3 == 0 and every block starts a new pass over the stream: ring slots are compile-time functions of j
    const int slot = j % R3, nslot = (j + 1) % R3, fslot = (j + R3 - 1) % R3;
    int jw = j + R3 - 1;
    jw = jw >= KS ? jw - KS : jw;
    const char* wsrc = wbase + (long)(ABL == 5 ? 0 : jw) * SLOT + wave * (4 * PIECE);  // (ABL 5: every step re-reads stream step 0 -- L1 / L2-hot weights)
    const uint32_t wdst = (uint32_t)(fslot * SLOT + wave * (4 * PIECE));
    const int e4 = j + SD;
    const bool wrap = e4 >= KS;
    const int ec = wrap ? e4 - KS : e4;  // k step (and epilogue unit) the side request is for
    const uint32_t sd = side_dst0 + (uint32_t)(ec % SIDE_RING) * SIDE_SLOT;
    const char* s1 = Ab + 64 * (ec < a.K16 ? ec : 0);  // padded k steps (zero weights) re-read k step 0: never past a row
    uint32_t o1 = wrap ? in_off_next : in_off;
    const char* s2 = Xb + 64 * (ec < 16 ? ec : 15);
    uint32_t o2 = wrap ? ax_off_cur : ax_off_prev;  // unit ec of the block in P, or (wrapped) of the running block
    if (ABL == 6) {  // timing ablation: every block reads the input / mask rows of the workgroup's FIRST block (they stay in the L2)
      o1 = abl_in0;
      o2 = abl_ax0;
    }
    if (ABL == 3 || ABL == 4) {  // timing ablation: every workgroup walks the K columns from its own starting tile (2 (blockIdx % 8) k steps
      // on): do all workgroups reading the same 128-byte column at the same time camp on a few channels?
      const int er = ((ec < a.K16 ? ec : 0) + 2 * (int)(blockIdx.x & 7)) & 15;
      s1 = Ab + 64 * er;
      s2 = Xb + 64 * er;
    }
    if (ABL == 1 || ABL == 2 || ABL == 4) {  // timing ablation: the same bytes as whole lines (see the template's comment); rows of the block, clamped
      const int ecl = ABL == 4 ? (((ec < a.K16 ? ec : 0) + 2 * (int)(blockIdx.x & 7)) & 15) : (ec < a.K16 ? ec : 0);
      const long rb = (wrap ? blk + gridDim.x : blk) * BPTS + wave * 32;
      auto lin = [&](int piece, long ld) {
        long r = rb + 8 * (2 * (ecl & 1) + piece) + (lane >> 3);
        r = r < a.P ? r : a.P - 1;
        return (uint32_t)((r * ld + 32 * (ecl >> 1) + 4 * (lane & 7)) * 4);
      };
      s1 = Ab; s2 = Xb;
      abl_o1[0] = lin(0, a.lda); abl_o1[1] = lin(1, a.lda);
      abl_o2[0] = lin(0, a.ld_aux); abl_o2[1] = lin(1, a.ld_aux);
    }
#pragma unroll
    for (int pair = 0; pair < 4; ++pair) {
      if (pair == 2) {  // rendezvous: the weights of stream step gs + 1 have landed in every wave; slot gs - 1 is free
        RG_WAIT_VM(NWAIT);
        __builtin_amdgcn_s_barrier();
      }
      const char* rd = ring_lane + (pair < 3 ? slot * SLOT + (pair + 1) * (4 * PIECE) : nslot * SLOT);
#pragma unroll
      for (int m = 0; m < 6; ++m) {
        const int pr = m >> 1, tt = m & 1;            // (w limb, act limb): hi hi, hi lo, lo hi
        const int wl = pr == 2 ? 1 : 0, al = pr == 1 ? 1 : 0;
        Q[2 * pair + tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A[pair & 1][2 * tt + wl]),
                                                                 __builtin_bit_cast(f16x8, Bc.l[al]), Q[2 * pair + tt], 0, 0, 0);
        if (m < 4) A[(pair + 1) & 1][m] = *reinterpret_cast<const u32x4*>(rd + m * PIECE);
        if (pair == 2 && m < 4) dma_piece(wsrc + m * PIECE, lane16, wdst + m * PIECE);
        if (pair == 3 && (m == 0 || m == 2)) dma_piece(WL ? s1 : s1 + 32 * (m == 2), WL ? abl_o1[m == 2] : o1, sd + (m == 2) * PIECE);
        if (pair == 3 && NSIDE == 2 && (m == 3 || m == 5))
          dma_piece(WL ? s2 : s2 + 32 * (m == 5), WL ? abl_o2[m == 5] : o2, sd + 2 * PIECE + (m == 5) * PIECE);
#pragma unroll
        for (int u = 0; u < 4; ++u) {  // constant trip count (the slice bounds fold once pair and m are unrolled)
          const int k = cnt[pair] * m / 6 + u;
          if (k < cnt[pair] * (m + 1) / 6) mp(pair, k);
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
    // the scale of this block's rows: their maximum (loaded a block ago) or the floor, at [2^12, 2^13); the next block's maximum is
    // requested now
    {
      float m = a.amax_floor;
      if (AMb) {
        m = fmaxf(m, am_cur);
        request_amax(blk + gridDim.x);
      }
      if (EPI == EPI_MASKB) request_bits(blk);  // this block's mask rows: needed from the next iteration on, when it is P
      kB = row_scale(m);
      sB = pow2f(kB);
    }
    init_q();
    EpiState st;
    // k step 0 / unit 0: not overlapped (once per block)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int k = 0; k < 13; ++k)
        if (k < CNT[c]) mop(0, true, c, k, Bc, st);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (j + 1 < 16)
        kstep(j, CNT, [&](int c, int k) { mop(j + 1, true, c, k, Bn, st); });
      else  // k step 16 exists only when KS = 20; its input fragment sits in side slot 0 either way
        kstep(j, CNT, [&](int c, int k) { mop(j + 1, false, c, k, Bn, st, 2); });
    }
    if (KS > 16) {  // K padded to 320 (wave-uniform)
#pragma unroll
      for (int j = 16; j < 20; ++j) kstep(j, CNT, [&](int c, int k) { mop(j + 1, false, c, k, Bn, st, j + 1 < 20 ? 1 : 0); });
    }
    // ---- every unit of the block held in P has been stored: its rows' maxima (both lane halves hold the same point) ----
    if (a.amax_out && row_P >= 0) {
      const float o = fmaxf(omx, __shfl_xor(omx, 32));
      if (hh == 0 && row_P + li < a.P) a.amax_out[row_P + li] = o;
    }
    omx = 0.f;
    if (want_bits && row_P >= 0) store_bits();
    if (EPI == EPI_MASKB) {
#pragma unroll
      for (int d = 0; d < 8; ++d) mb[d] = (d < 4 ? mbn0[d & 3] : mbn1[d & 3]) >> (4 * hh);
    }
    // ---- the block is finished: it becomes P; its stores run behind the next block ----
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) P[nt] = Q[nt];
    yscP = c3 * pow2f(-kB);
    row_P = blk * BPTS + wave * 32;
    am_cur = am_next;
    const long row = blk * BPTS + wave * 32 + li;  // unclamped: the buffer range check drops rows >= P
    crs_prev = crs;
    sv_old[0] = sv[2];
    sv_old[1] = sv[3];
    crs = make_rsrc(a.C, cbytes);
    cvoff = (uint32_t)((row * a.ldc + 4 * hh) * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i)  // whole-line pieces of this block's rows: lane -> (row 8 i + r8, chunk p8 ^ f); range-checked stores
      sv[i] = (uint32_t)(((row_P + 8 * i + r8) * a.ldc + 4 * (p8 ^ f0 ^ ((i & 1) << 2))) * 4);
    in_off = in_off_next;
    in_off_next = (uint32_t)((clamp_row(blk + 2 * (long)gridDim.x) * a.lda + 4 * hh) * 4);
    ax_off_prev = ax_off_cur;
    ax_off_cur = (uint32_t)((clamp_row(blk + gridDim.x) * (long)a.ld_aux + 4 * hh) * 4);
  }
  RG_WAIT_VM(0);  // no LDS-DMA in flight when the workgroup's LDS is released
  (void)no_mop; (void)CNT_NONE; (void)xbytes;
  if (ABL == 0) {  // the last tile's pieces 2, 3 of the block stored during the last iteration (KS = 20 stored them with unit 17)
    store4(rb[2], KS > 16 ? nullrs : crs_prev, sv_old[0] + 128 * 7);
    store4(rb[3], KS > 16 ? nullrs : crs_prev, sv_old[1] + 128 * 7);
  }
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
          float t = P[nt][4 * g + k] * yscP;
          if (RELU) t = relu1(t);
          if (EPI == EPI_MASK) t = bitsf(mv[k]) > 0.f ? t : 0.f;
          if (EPI == EPI_MASKB) t = bitsf(fbits(t) & (uint32_t)__builtin_amdgcn_sbfe((int)mb[nt], 8 * g + k, 1));
          if (EPI == EPI_RELUB) bo[nt] |= (fbits(t) < 1u ? fbits(t) : 1u) << (8 * g + k);
          v[k] = t;
          omx = fmaxf(omx, fabsf(t));
        }
        store4(v, crs, cvoff + co);
      }
    if (a.amax_out && row_P >= 0) {
      const float o = fmaxf(omx, __shfl_xor(omx, 32));
      if (hh == 0 && row_P + li < a.P) a.amax_out[row_P + li] = o;
    }
    if (want_bits && row_P >= 0) store_bits();
  }
  if (a.guard) {
    const float m2 = fmaxf(mx, __shfl_xor(mx, 32));
    if (m2 >= 65504.f) __hip_atomic_store(a.guard, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

template <int EPI>
int launch_h3(const RGArgs& a, hipStream_t s) {
  constexpr int NSIDE = EPI == EPI_MASK ? 2 : 1;
  constexpr int lds = R3 * SLOT + NW * side_ring(NSIDE) * NSIDE * 2 * PIECE + 1024 + NW * 4096;
  static_assert(lds <= 160 * 1024, "LDS budget");
#ifdef HOLD_DEV
  if (const char* v = getenv("HOLD_RG_ABL")) {  // developer build: access-shape ablations (results garbage)
    int dev_ = 0;
    hipDeviceProp_t prop_;
    if (hipGetDevice(&dev_) != hipSuccess || hipGetDeviceProperties(&prop_, dev_) != hipSuccess) return HOLD_E_LAUNCH;
    const long blocks_ = (a.P + BPTS - 1) / BPTS;
    const dim3 grid_((unsigned)(blocks_ < prop_.multiProcessorCount ? blocks_ : prop_.multiProcessorCount));
    if (v[0] == '1') {
      if (hipFuncSetAttribute((const void*)rgemm_h3_kernel<EPI, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return HOLD_E_LAUNCH;
      hipLaunchKernelGGL((rgemm_h3_kernel<EPI, 1>), grid_, dim3(256), lds, s, a);
      return hipGetLastError() == hipSuccess ? HOLD_OK : HOLD_E_LAUNCH;
    }
    if (v[0] == '2') {
      if (hipFuncSetAttribute((const void*)rgemm_h3_kernel<EPI, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return HOLD_E_LAUNCH;
      hipLaunchKernelGGL((rgemm_h3_kernel<EPI, 2>), grid_, dim3(256), lds, s, a);
      return hipGetLastError() == hipSuccess ? HOLD_OK : HOLD_E_LAUNCH;
    }
    if (v[0] == '3') {
      if (hipFuncSetAttribute((const void*)rgemm_h3_kernel<EPI, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return HOLD_E_LAUNCH;
      hipLaunchKernelGGL((rgemm_h3_kernel<EPI, 3>), grid_, dim3(256), lds, s, a);
      return hipGetLastError() == hipSuccess ? HOLD_OK : HOLD_E_LAUNCH;
    }
    if (v[0] == '4') {
      if (hipFuncSetAttribute((const void*)rgemm_h3_kernel<EPI, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return HOLD_E_LAUNCH;
      hipLaunchKernelGGL((rgemm_h3_kernel<EPI, 4>), grid_, dim3(256), lds, s, a);
      return hipGetLastError() == hipSuccess ? HOLD_OK : HOLD_E_LAUNCH;
    }
    if (v[0] == '5') {
      if (hipFuncSetAttribute((const void*)rgemm_h3_kernel<EPI, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return HOLD_E_LAUNCH;
      hipLaunchKernelGGL((rgemm_h3_kernel<EPI, 5>), grid_, dim3(256), lds, s, a);
      return hipGetLastError() == hipSuccess ? HOLD_OK : HOLD_E_LAUNCH;
    }
    if (v[0] == '6') {
      if (hipFuncSetAttribute((const void*)rgemm_h3_kernel<EPI, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return HOLD_E_LAUNCH;
      hipLaunchKernelGGL((rgemm_h3_kernel<EPI, 6>), grid_, dim3(256), lds, s, a);
      return hipGetLastError() == hipSuccess ? HOLD_OK : HOLD_E_LAUNCH;
    }
    if (v[0] == '7') {
      if (hipFuncSetAttribute((const void*)rgemm_h3_kernel<EPI, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return HOLD_E_LAUNCH;
      hipLaunchKernelGGL((rgemm_h3_kernel<EPI, 7>), grid_, dim3(256), lds, s, a);
      return hipGetLastError() == hipSuccess ? HOLD_OK : HOLD_E_LAUNCH;
    }
    if (v[0] == '8') {
      if (hipFuncSetAttribute((const void*)rgemm_h3_kernel<EPI, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return HOLD_E_LAUNCH;
      hipLaunchKernelGGL((rgemm_h3_kernel<EPI, 8>), grid_, dim3(256), lds, s, a);
      return hipGetLastError() == hipSuccess ? HOLD_OK : HOLD_E_LAUNCH;
    }
  }
#endif
  static int n_cu = 0;
  static bool attr_set = false;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return HOLD_E_LAUNCH;
    n_cu = prop.multiProcessorCount;
  }
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)rgemm_h3_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      return HOLD_E_LAUNCH;
    attr_set = true;
  }
  const long blocks = (a.P + BPTS - 1) / BPTS;
  hipLaunchKernelGGL((rgemm_h3_kernel<EPI>), dim3((unsigned)(blocks < n_cu ? blocks : n_cu)), dim3(256), lds, s, a);
  return hipGetLastError() == hipSuccess ? HOLD_OK : HOLD_E_LAUNCH;
}

// bits[p][d] bit b = (C[p][32 d + b] > 0), rebuilt from C -- only when the overflow guard's fallback count (guard[2]) moved since the
// last such launch (guard[3]): hold_gemm_r6_if has then rewritten C behind an EPI_RELU launch whose bits came from overflowed values.
// The last workgroup to finish publishes guard[3] = guard[2] (guard[1]: the arrival counter of the conditional launches, 0 between them).
__global__ __launch_bounds__(256) void relu_bits_if_kernel(const float* __restrict__ C, int ldc, long P, uint32_t* __restrict__ bits,
                                                           uint32_t* guard) {
  if (__hip_atomic_load(guard + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ==
      __hip_atomic_load(guard + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
    return;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < P * 8; i += (long)gridDim.x * 256) {
    const float* c = C + (i >> 3) * ldc + 32 * (i & 7);
    uint32_t w = 0u;
#pragma unroll
    for (int b4 = 0; b4 < 8; ++b4) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(c + 4 * b4);
#pragma unroll
      for (int k = 0; k < 4; ++k) w |= ((int)fbits(v[k]) > 0 ? 1u : 0u) << (4 * b4 + k);
    }
    bits[i] = w;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    if (atomicAdd(guard + 1, 1u) == gridDim.x - 1) {
      guard[1] = 0u;
      __hip_atomic_store(guard + 3, __hip_atomic_load(guard + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

}  // namespace

extern "C" int64_t hold_gemm_h3_pack_bytes(int32_t K) { return (int64_t)((K + 63) / 64 * 4) * SLOT; }

extern "C" int hold_gemm_r6_if(const float* A, int32_t lda, int64_t P, const void* wpack, int32_t K, const float* bias,
                               int32_t epilogue, const float* aux, int32_t ld_aux, float* C, int32_t ldc, float* amax_out,
                               uint32_t* guard, hold_stream_t st);

// hold_gemm_r6's contract in the f16x3 arithmetic.  wpack_h3 = hold_gemm_h3_pack_bytes(K) bytes of fp16, [KS k steps j][8 n-tiles
// nt][2 limbs t][2 halves h][32 rows i][8 e] = limb_t(s_w W)[32 nt + i][16 j + 8 (e / 4) + 4 h + e % 4]; c3 = 1 / s_w (device
// scalar).  amax_in [P] (or NULL): an upper bound of |A[p][:]| over the columns its producer wrote (exact when it is a producer's
// amax_out); amax_floor >= the magnitude of the other columns (NULL amax_in: of every column); amax_out [P] (or NULL): receives
// max |C[p][:]|.  guard / wpack_r6: overflow guard and conditional f32x6 fallback (hold_gemm_r6_if), as hold_fused_sdf_h3.
// relu_bits_out (epilogue 1, optional): [P][8] dwords, bit n of row p = (C[p][n] > 0).  mask_bits_in (epilogue 2, optional): such a
// matrix; the mask is then taken from it instead of aux (aux is still what the f32x6 fallback reads: required with wpack_r6).
extern "C" int hold_gemm_h3_bits(const float* A, int32_t lda, int64_t P, const void* wpack_h3, const float* c3, int32_t K,
                                 const float* bias, int32_t epilogue, const float* aux, int32_t ld_aux, float* C, int32_t ldc,
                                 const float* amax_in, float amax_floor, float* amax_out, uint32_t* relu_bits_out,
                                 const uint32_t* mask_bits_in, uint32_t* guard, const void* wpack_r6, hold_stream_t st) {
  if (!A || !wpack_h3 || !c3 || !C || P < 0 || K < 256 || K > 320 || (K & 15) || lda < K || (lda & 3) || ldc < 256 || (ldc & 3)) return HOLD_E_ARG;
  if (((uintptr_t)A & 15) || ((uintptr_t)C & 15) || ((uintptr_t)wpack_h3 & 15) || (bias && ((uintptr_t)bias & 15))) return HOLD_E_ARG;
  if (epilogue < 0 || epilogue > 2) return HOLD_E_ARG;
  const bool maskb = epilogue == 2 && mask_bits_in != nullptr;
  if (epilogue == 2 && bias) return HOLD_E_ARG;
  if (epilogue == 2 && (!maskb || wpack_r6) && !aux) return HOLD_E_ARG;  // aux: the mask itself, or what the f32x6 fallback reads
  if (aux && (ld_aux < 256 || (ld_aux & 3) || ((uintptr_t)aux & 15))) return HOLD_E_ARG;
  if ((relu_bits_out && epilogue != 1) || (mask_bits_in && epilogue != 2)) return HOLD_E_ARG;
  if (((uintptr_t)relu_bits_out & 15) || ((uintptr_t)mask_bits_in & 15)) return HOLD_E_ARG;
  if (!(amax_floor >= 0.f) || (!amax_in && !(amax_floor > 0.f))) return HOLD_E_ARG;
  if (((uintptr_t)amax_in & 3) || ((uintptr_t)amax_out & 3) || ((uintptr_t)guard & 3) || (wpack_r6 && !guard)) return HOLD_E_ARG;
  if (P == 0) return HOLD_OK;
  const int64_t ldmax = lda > ldc ? (lda > ld_aux ? lda : ld_aux) : (ldc > ld_aux ? ldc : ld_aux);
  if (((uint64_t)P + BPTS) * (uint64_t)ldmax * 4 >= (1ull << 32)) return HOLD_E_ARG;
  RGArgs a;
  a.A = A; a.lda = lda; a.P = (long)P; a.wpack = (const char*)wpack_h3; a.c3 = c3; a.amax_in = amax_in; a.amax_floor = amax_floor;
  a.amax_out = amax_out; a.guard = guard; a.KS = (K + 63) / 64 * 4; a.K16 = K / 16; a.bias = bias; a.aux = aux;
  a.ld_aux = ld_aux; a.C = C; a.ldc = ldc; a.bits_out = relu_bits_out; a.bits_in = mask_bits_in;
  hipStream_t s = (hipStream_t)st;
  const int rc = epilogue == 0 ? launch_h3<EPI_NONE>(a, s)
               : epilogue == 1 ? (relu_bits_out ? launch_h3<EPI_RELUB>(a, s) : launch_h3<EPI_RELU>(a, s))
               : maskb ? launch_h3<EPI_MASKB>(a, s) : launch_h3<EPI_MASK>(a, s);
  if (rc != HOLD_OK || !wpack_r6) return rc;
#ifdef HOLD_DEV
  if (const char* v = getenv("HOLD_RG_ABL"))  // a timing ablation's garbage results would trip the guard: no conditional relaunch
    if (v[0] != '0') return rc;
#endif
  const int rc2 = hold_gemm_r6_if(A, lda, P, wpack_r6, K, bias, epilogue, aux, ld_aux, C, ldc, amax_out, guard, st);
  if (rc2 != HOLD_OK || !relu_bits_out) return rc2;
  // the f32x6 kernel rewrites C, not the bits: a third conditional launch rebuilds them from C iff the fallback count moved
  hipLaunchKernelGGL(relu_bits_if_kernel, dim3(256), dim3(256), 0, s, (const float*)C, (int)ldc, (long)P, relu_bits_out, guard);
  return hipGetLastError() == hipSuccess ? HOLD_OK : HOLD_E_LAUNCH;
}

extern "C" int hold_gemm_h3(const float* A, int32_t lda, int64_t P, const void* wpack_h3, const float* c3, int32_t K,
                            const float* bias, int32_t epilogue, const float* aux, int32_t ld_aux, float* C, int32_t ldc,
                            const float* amax_in, float amax_floor, float* amax_out, uint32_t* guard, const void* wpack_r6,
                            hold_stream_t st) {
  if (epilogue == 2 && !aux) return HOLD_E_ARG;
  return hold_gemm_h3_bits(A, lda, P, wpack_h3, c3, K, bias, epilogue, aux, ld_aux, C, ldc, amax_in, amax_floor, amax_out, nullptr,
                           nullptr, guard, wpack_r6, st);
}
