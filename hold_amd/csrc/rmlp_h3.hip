// Register-resident trunk of the ImplicitNet in the TWO-LIMB fp16 arithmetic "f16x3" (gfx950): lin0..lin7 of
// ImplicitNet.forward (code/src/networks/shape_net.py:84-130), the structure of csrc/rmlp.hip -- one wave per SIMD owns
// 32 points for the whole network, a layer's accumulator registers become the next layer's MFMA B operand after softplus +
// limb split in registers, the weight limbs are the only stream (LDS-DMA ring shared by the four waves) -- with HALF the
// matrix instructions per product:
//
//   x = hi + lo,  hi = RN_f16(s x),  lo = RN_f16(s x - hi)      (s = an exact power of two per operand)
//   w x ~ hi_w hi_x + hi_w lo_x + lo_w hi_x                      3 x v_mfma_f32_32x32x16_f16, fp32 accumulation
//
// against the six bf16 limb products of rmlp.hip.  The representation keeps 22-23 significand bits wherever the scaled value
// is >= 2^-2 (hi: 11 bits, lo: a signed 11-bit correction of at most half an ulp of hi) and an ABSOLUTE 2^-25 / s below
// (lo becomes an fp16 subnormal: the matrix core takes fp16 subnormals unflushed -- scripts/probes/h3_probe.hip checks
// exactly that on the hardware); the dropped lo_w lo_x term is <= 2^-22 of the product, signed at random (round to nearest),
// where the truncation split of the bf16 scheme drops same-signed terms <= 2^-23.  Measured against fp64 the two schemes
// are indistinguishable (scripts/split_precision_study.py; tests/test_rmlp_gpu.py holds this kernel to <= 1.5 x the error of
// hold_fused_sdf_r6 on 524 288 points).  Scales: weights by s_w[l] = 2^k with max |W_l| s_w in [2^13, 2^14) (host, per matrix,
// at pack time -- hold_amd/field.py:pack_h3); activations by the constant SA = 2^6 (softplus outputs and the embedding:
// full precision from 2^-8 up, absolute 4.7e-10 below).  OVERFLOW GUARD (round 6): an activation >= 1023.5 would round to an
// fp16 infinity -- and an infinity is not guaranteed to stay visible (inf x negative weight -> -inf -> softplus -> 0).  Every
// lane therefore keeps the EXACT maximum of the scaled magnitudes it splits (one v_max3_f32 per value pair, as wgrad_h3_body
// does) and a lane that saw one beyond fp16's largest finite value sets the caller's guard word; the entry points enqueue the
// f32x6 kernel (csrc/rmlp.hip) right behind as a CONDITIONAL launch that exits at once unless that word is set, recomputes the
// whole launch otherwise, counts the event and clears the word: no host read, no silent infinity.
// The accumulators hold s_w SA a_l; the epilogue multiplies by c3[l] = 1 / s_w[l] (exact) and carries SA through softplus.
//
// Per k step: 16 KiB of weight limbs (8 n-tiles x 2 limbs x 1 KiB fragments, 24 KiB in rmlp.hip), 24 MFMAs (48), and a
// limb split of 4 instructions per two values (v_cvt_pk_f16_f32, 2 x v_fma_mix_f32, v_cvt_pk_f16_f32; 11 in rmlp.hip).
//
// Entry points: hold_fused_sdf_h3 (sampler query) and hold_trunk_h3 (training forward: stores h_0..h_7 in fp32, unscaled).
// Roofline: f16 MFMA pipe (3 limb products issued per algorithmic product); HBM bytes as rmlp.hip.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/hold_hip.h"
#include "rmlp_h3_sched.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int NW = 4;                 // waves per workgroup (one per SIMD)
constexpr int BPTS = 32 * NW;         // points per workgroup pass
constexpr int PIECE = 1024;           // one MFMA A fragment for a wave: 64 lanes x 16 B
constexpr int NPC = 16;               // pieces per k step: [8 n-tiles][2 limbs]
constexpr int SLOT = NPC * PIECE;
constexpr int RING = 4;               // LDS slots; L0S % RING == LKS % RING == 0: the slot of a k step is a compile-time constant
constexpr int L0S = 4, LKS = 16;      // k steps of layer 0 (K = 64: 39 used, the last step's weights are zero) and of the 256-wide layers
constexpr int NSTEP = L0S + 7 * LKS;  // 116 k steps per block of points
constexpr int AHEAD = RING - 1;       // the rows of k step t + AHEAD are requested during k step t
constexpr int NE = 39, EMB_STR = 52, SKIP_OUT = 217;
constexpr int NGAP = 24;              // MFMAs per k step
constexpr int OFF_BIAS = RING * SLOT;               // [8][256] fp32, pre-scaled by s_w[l] SA on the host
constexpr int OFF_W8 = OFF_BIAS + 8 * 256 * 4;      // [256] fp32
constexpr int OFF_EMB = OFF_W8 + 256 * 4;           // [4 waves][32 points][EMB_STR] fp32 (wave-private), scaled by SA
constexpr int OFF_BARF = OFF_EMB + NW * 32 * EMB_STR * 4;  // [64] fp32: SA x BARF weights of the 39 embedding columns
constexpr int OFF_RT = (OFF_BARF + 64 * 4 + 1023) & ~1023;  // STORE: wave-private result tiles [32 rows][128 B] (whole-line h stores)
constexpr int LDS_BYTES = OFF_RT + NW * 4096;
static_assert(L0S % RING == 0 && LKS % RING == 0, "the ring slot of a k step must not depend on the layer");
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");

constexpr float SA = 64.0f;                            // activation scale (a power of two)
constexpr float KE = -144.26950408889634f / SA;        // exp2 argument per unit of the SCALED pre-activation
constexpr float CL = 0.0069314718056f * SA;            // ln2 / 100, scaled

struct H3Args {
  const float* xc; int ldx; long P;
  const char* wpack;    // hold_trunk_h3_pack_bytes() bytes, [NSTEP = 116][16 pieces][64 lanes][8 f16]
  const float* bias;    // [8][256], bias_l s_w[l] SA
  const float* c3;      // [8] 1 / s_w[l]
  const float* w8;      // [256] sdf row of lin8 (HEAD)
  const float* b8;      // device scalar: bias of the sdf row (HEAD)
  const float* barf;    // [39] or null
  float* sdf; int lds;  // HEAD output
  float* h[8]; int ldh; // STORE outputs ([P][ldh], columns 0..255)
  uint32_t* guard;      // [0] set to 1 when a scaled activation left fp16's range (null: not reported)
};

__device__ __forceinline__ uint32_t fbits(float x) { return __builtin_bit_cast(uint32_t, x); }
__device__ __forceinline__ float bitsf(uint32_t x) { return __builtin_bit_cast(float, x); }

// max(y, 0) as ONE v_max_i32 (fmaxf costs a canonicalising v_max y,y first)
__device__ __forceinline__ float relu1(float y) {  // sign bit set <=> negative as an integer
  const int b = __builtin_bit_cast(int, y);
  return __builtin_bit_cast(float, b > 0 ? b : 0);
}

// value-only softplus of an UNSCALED pre-activation (the last layer of the sampler query)
__device__ __forceinline__ float sp_fast(float y) {
  const float e = __builtin_amdgcn_exp2f(-144.26950408889634f * fabsf(y));
  return fmaf(0.0069314718056f, __builtin_amdgcn_logf(1.0f + e), relu1(y));
}
// training softplus (see rmlp.hip): log1p by series where 1 + e would round e away; y > 0.2 returns y
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

// 16-byte row-fragment store, column offset in the immediate + s_nop 3 (the gfx950 store-data hazard of rmlp.hip)
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
// x - (float) half `sel` of the packed f16 dword hi: ONE v_fma_mix_f32 (exact: hi has 11 significant bits inside x's 24)
template <int SEL>
__device__ __forceinline__ float resid(uint32_t hi, float x) {
  float r;
#ifdef HOLD_H3_SPLIT_PLAIN
  const f16x2 h = __builtin_bit_cast(f16x2, hi);
  r = x - (float)h[SEL];
#else
  if (SEL == 0)
    asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hi), "v"(x));
  else
    asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hi), "v"(x));
#endif
  return r;
}

// LDS-DMA (inline assembly: hipcc models the builtin as a FLAT access that may touch LDS and would wait lgkmcnt(0) at every
// later ds_read; completion is counted by hand): TWO consecutive 1 KiB pieces (wave-uniform source, lane offset lane16) to
// LDS bytes dst, dst + 1 KiB -- the instruction offset advances both the global and the LDS address (M0 = LDS base).  M0 is
// not saved: nothing else in this kernel uses it.
__device__ __forceinline__ void dma_pair(const char* src, uint32_t lane16, uint32_t dst) {
  asm volatile(
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %0, %1\n\t"
      "global_load_lds_dwordx4 %0, %1 offset:1024"
      :
      : "v"(lane16), "s"(src), "s"(dst)
      : "memory");
}
// transcendentals as opaque volatile instructions: they stay where the schedule puts them without a pin behind them (an
// empty asm right behind a builtin v_exp / v_log made the hazard recogniser pad a wait state -- 9 s_nop per k step); their
// consumers are LAT_TRANS instruction slots away by construction of the schedule (tests/test_r6_pack_cpu.py checks the table)
// packed fp32 arithmetic (two values per instruction) as volatile instructions as well: written as <2 x float> C
// arithmetic the instruction selector scalarises most of it again (operands assembled from separately computed halves)
__device__ __forceinline__ f32x2 pk_mul(f32x2 a, f32x2 b) { f32x2 r; asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) { f32x2 r; asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { f32x2 r; asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float exp2_v(float x) { float r; asm volatile("v_exp_f32 %0, %1" : "=v"(r) : "v"(x)); return r; }
__device__ __forceinline__ float log2_v(float x) { float r; asm volatile("v_log_f32 %0, %1" : "=v"(r) : "v"(x)); return r; }

// Placement of a k step's micro-operations behind its 24 MFMAs: gap G issues the operations op(k), begin(G) <= k < end(G).
struct SchedNone {
  static constexpr int begin(int) { return 0; }
  static constexpr int end(int) { return 0; }
  static constexpr int op(int k) { return k; }
};
template <int N>
struct SchedUniform {  // N operations in index order, evenly
  static constexpr int begin(int G) { return N * G / NGAP; }
  static constexpr int end(int G) { return N * (G + 1) / NGAP; }
  static constexpr int op(int k) { return k; }
};
// the generated tables (scripts/gen_h3_schedule.py): transcendentals one per gap, every gap inside the MFMA's 32 cycles
template <bool HEAD>
struct SchedEpi {
  static constexpr int N = HEAD ? H3_SCHED_HEAD_N : H3_SCHED_STORE_N;
  static constexpr int end(int G) {
    constexpr unsigned char eh[NGAP] = H3_SCHED_HEAD_END;
    constexpr unsigned char es[NGAP] = H3_SCHED_STORE_END;
    return HEAD ? eh[G] : es[G];
  }
  static constexpr int begin(int G) { return G ? end(G - 1) : 0; }
  static constexpr int op(int k) {
    constexpr unsigned char oh[H3_SCHED_HEAD_N] = H3_SCHED_HEAD_ORDER;
    constexpr unsigned char os[H3_SCHED_STORE_N] = H3_SCHED_STORE_ORDER;
    return HEAD ? oh[k < H3_SCHED_HEAD_N ? k : 0] : os[k < H3_SCHED_STORE_N ? k : 0];
  }
};

#define H3_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define H3_WAIT_LDS3() __builtin_amdgcn_s_waitcnt(0xC37F)  // lgkmcnt(3): a real s_waitcnt, which the compiler's own wait insertion sees

// ABL (developer builds only, results garbage): bit 0 = v_exp_f32 replaced by a v_mul_f32, bit 1 = v_log_f32 likewise (what the
// transcendentals cost the k step beyond an ordinary VALU instruction in their place: nothing, GPU call 6); bit 2 = no LDS-DMA
// in the k steps, bit 3 = no fragment reads (the first step's fragments are reused), bit 4 = no rendezvous barrier
template <bool HEAD, bool STORE, int ABL = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void rmlp_h3_kernel(H3Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hh = lane >> 5, li = lane & 31;
  const uint32_t lane16 = lane * 16;
  float* embw = reinterpret_cast<float*>(smem + OFF_EMB) + wave * (32 * EMB_STR);
  const char* ring_lane = smem + lane * 16;
  const float b8 = HEAD ? *a.b8 : 0.f;  // read on the device: a host copy of a trained parameter costs a stream drain

  // ---- once per workgroup: biases (+ the sdf row) into LDS, the first AHEAD k steps into the ring ----
  for (int i = tid; i < 8 * 256; i += 256) reinterpret_cast<float*>(smem + OFF_BIAS)[i] = a.bias[i];
  if (HEAD) reinterpret_cast<float*>(smem + OFF_W8)[tid] = a.w8[tid];
  if (tid < 64) reinterpret_cast<float*>(smem + OFF_BARF)[tid] = SA * ((a.barf && tid < NE) ? a.barf[tid] : 1.0f);
  __syncthreads();
  const char* wsrc0 = a.wpack + wave * (4 * PIECE);                 // this wave's four pieces of k step 0
  const uint32_t wdst0 = (uint32_t)(wave * (4 * PIECE));            // ... and their place in ring slot 0
#pragma unroll
  for (int s = 0; s < AHEAD; ++s)
#pragma unroll
    for (int h = 0; h < 2; ++h) dma_pair(wsrc0 + (long)s * SLOT + h * (2 * PIECE), lane16, wdst0 + s * SLOT + h * (2 * PIECE));

  f32x16 P[8], Q[8];
  u32x4 A[2][4];  // weight fragments of two n-tiles x two limbs, double-buffered
  Limbs Bc, Bn;
  float mx = 0.f;  // exact maximum of the scaled magnitudes this lane has split (the overflow guard)

  auto read_pair = [&](int slot, int pair, u32x4 (&dst)[4]) {
    const char* base = ring_lane + slot * SLOT + pair * (4 * PIECE);
#pragma unroll
    for (int i = 0; i < 4; ++i) dst[i] = *reinterpret_cast<const u32x4*>(base + i * PIECE);
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
    // ---- SA x embedding of this wave's 32 points -> wave-private LDS [32][EMB_STR] (embedders.py:18-50): lane half hh takes
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
      H3_WAIT_VM(4 * (AHEAD - 1));
      __builtin_amdgcn_s_barrier();
      read_pair(0, 0, A[0]);
      A[1][0] = *reinterpret_cast<const u32x4*>(ring_lane + 4 * PIECE);  // group 1's fragments 0, 2, 1 (its 3 follows behind MFMA 0)
      A[1][2] = *reinterpret_cast<const u32x4*>(ring_lane + 6 * PIECE);
      A[1][1] = *reinterpret_cast<const u32x4*>(ring_lane + 5 * PIECE);
      first = 0;
    }

    // One k step of the layer being accumulated into Q.  `ts` = the step's index in its layer (compile time; ring slot
    // ts % RING), `src` = this wave's four pieces of the step AHEAD steps later in the stream.  EXPLICIT schedule: 4 groups
    // (pairs of n-tiles) x 6 MFMAs; behind every MFMA stands a fixed slice of everything else -- one fragment read for the next
    // group (gaps 0..3 of a group), the LDS wait for this group's fragments (gap 0), one DMA pair (gap 1 of the two groups
    // behind the rendezvous) and the micro-operations the schedule S gives the gap -- closed by a full scheduling barrier.
    auto kstep = [&](int ts, const char* src, auto sch, auto&& mop) {
      using S = decltype(sch);
      const int slot = ts % RING;
#pragma unroll
      for (int pair = 0; pair < 4; ++pair) {
        if (pair == 2 && !(ABL & 16)) {  // mid-step rendezvous: the next step complete in LDS, the previous step's slot free
          if (ABL & 4) H3_WAIT_VM(0); else
          H3_WAIT_VM(4 * (AHEAD - 2));
          __builtin_amdgcn_s_barrier();
        }
        // fragment f (0 = tile 0 hi, 1 = tile 0 lo, 2 = tile 1 hi, 3 = tile 1 lo) of group tp of this step (tp >= 4: group
        // tp - 4 of the next step, complete in LDS behind the rendezvous)
        auto frag = [&](int tp, int f) {
          return *reinterpret_cast<const u32x4*>(ring_lane + (tp < 4 ? slot : (ts + 1) % RING) * SLOT + (tp & 3) * (4 * PIECE) + f * PIECE);
        };
#pragma unroll
        for (int m = 0; m < 6; ++m) {
          const int pr = m >> 1, tl = m & 1;           // (w limb, act limb): hi hi, hi lo, lo hi
          const int wl = pr == 2 ? 1 : 0, al = pr == 1 ? 1 : 0;
          Q[2 * pair + tl] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A[pair & 1][2 * tl + wl]),
                                                                   __builtin_bit_cast(f16x8, Bc.l[al]), Q[2 * pair + tl],
                                                                   0, 0, 0);
          // Fragment reads, each into the register quad its last MFMA has just left (the MFMAs of a group use the fragments in
          // the order 0 2 0 2 1 3): fragments 0, 2, 1 of the group AFTER NEXT behind the MFMAs 3, 4, 5, fragment 3 of the next
          // group behind MFMA 0 -- a read is 5 to 9 MFMAs ahead of its wait instead of 2 to 5 (the fragment reads, not the
          // matrix pipe, bounded the first version: without them the k step ran in 541 instead of 717 ns, GPU call 8)
          if (!(ABL & 8)) {
            if (m == 0) A[(pair + 1) & 1][3] = frag(pair + 1, 3);
            if (m == 3) A[pair & 1][0] = frag(pair + 2, 0);
            if (m == 4) A[pair & 1][2] = frag(pair + 2, 2);
            if (m == 5) A[pair & 1][1] = frag(pair + 2, 1);
          }
          if (pair >= 2 && m == 1 && !(ABL & 4))
            dma_pair(src + (pair & 1) * (2 * PIECE), lane16,
                     wdst0 + (uint32_t)(((ts + AHEAD) % RING) * SLOT + (pair & 1) * (2 * PIECE)));
          const int G = 6 * pair + m;
#pragma unroll
          for (int u = 0; u < 12; ++u) {  // constant trip count (the slice bounds fold once pair and m are unrolled)
            const int k = S::begin(G) + u;
            if (k < S::end(G)) mop(S::op(k));
          }
          // ONE LDS wait per group, at the end of the gap in front of it (inside this scheduling region, so that the next
          // group's first MFMA cannot be hoisted above it and get a wait of its own): everything but the three most recent
          // reads (the group after next's) has landed, i.e. the next group's four fragments.  A real s_waitcnt: the compiler's
          // wait insertion accounts for it.
          if (m == 5) H3_WAIT_LDS3();
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      Bc = Bn;
    };
    auto no_mop = [](int) {};

    // limb split of dword d from two (scaled) values, as four micro-operations (op = 0..3); results pinned (empty volatile
    // asm): the instruction selector's list scheduler is free to place pure VALU code anywhere between its operands and its
    // first user, and without the pins it collects the whole epilogue in front of the first volatile instruction
    struct SplitState { uint32_t hi[4]; float ra[4], rb[4]; };
    auto split_op = [&](int op, int d, float x0, float x1, Limbs& out, SplitState& ss, bool track = true) {
      if (op == 0) {
        ss.hi[d] = cvt_pk(x0, x1);
        asm volatile("" : "+v"(ss.hi[d]));
        out.l[0][d] = ss.hi[d];
        if (track) asm volatile("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(mx) : "v"(x0), "v"(x1));
      } else if (op == 1) {
        ss.ra[d] = resid<0>(ss.hi[d], x0);
      } else if (op == 2) {
        ss.rb[d] = resid<1>(ss.hi[d], x1);
      } else {
        uint32_t lo = cvt_pk(ss.ra[d], ss.rb[d]);
        asm volatile("" : "+v"(lo));
        out.l[1][d] = lo;
      }
    };

    // ---- layer 0: B limbs straight from the (scaled) embedding (natural k order 16 j + 8 hh + e; the fourth k step is
    // padding: zero weights, zero B limbs) ----
    SplitState ss;
    auto emb_mop = [&](int j, int k, Limbs& out) {  // k = 0..15: round-major over the four dwords
      const int op = k >> 2, d = k & 3;
      const float* er = embw + li * EMB_STR + 16 * j + 8 * hh + 2 * d;
      // the padding k step (j = 3, columns 48..63) gets ZERO limbs, not what lies behind the embedding in LDS: columns 48..51 of
      // a point's row are never written, and a stale bit pattern there that converts to an fp16 inf / NaN times a zero weight is
      // a NaN in every accumulator (call 4 / 5 of round 5: wrong sdf -> a sampler window out of range -> a memory fault two
      // kernels later; the unit tests, which start from a quiet LDS, passed)
      const bool pad = 16 * j >= 48;
      split_op(op, d, pad ? 0.f : er[0], pad ? 0.f : er[1], out, ss, !pad);
    };
    init_bias(0);
#pragma unroll
    for (int k = 0; k < 16; ++k) emb_mop(0, k, Bc);
#pragma unroll
    for (int j = 0; j < L0S; ++j) {
      const char* src = wsrc0 + (long)(j + AHEAD) * SLOT;
      if (j + 1 < L0S)
        kstep(j, src, SchedUniform<16>(), [&](int k) { emb_mop(j + 1, k, Bn); });
      else
        kstep(j, src, SchedNone(), no_mop);
    }

    // ---- layers 1..7: input = SA x softplus(c3 x the previous layer's accumulators) ----
    float part = 0.f;  // HEAD: this lane's share of w8 . h7
    // STORE: h rows through a buffer descriptor (rows >= P fall outside num_records: the hardware drops those stores)
    rsrc_t hrs = make_rsrc(nullptr, 0);
    const uint32_t hbytes = (uint32_t)(a.P * a.ldh * 4);
    const uint32_t hvoff = (uint32_t)(((p0 + li) * a.ldh + 4 * hh) * 4);
    // ---- STORE: the h rows leave as WHOLE LINES (csrc/rgemm_h3.hip "result stores", csrc/rchain_h3.hip's result tiles).  A unit's
    // two 16-byte fragments per lane are kept in registers (dv) until the NEXT unit's first micro-operations write them into a
    // wave-private tile [32 rows][128 B] of two units (position p of row r holds chunk p ^ f(r)); an even unit j then reads the
    // finished tile back lane-linear (piece i = rows 8 i .. 8 i + 7, eight lanes per row); the two store slots of a unit -- unchanged
    // in number and place, so the vector-memory queue keeps its shape -- send pieces 0, 1 (even unit) and 2, 3 (odd unit) of that
    // tile as eight whole lines each.  Tile 7 of a layer is finished by the next layer's unit 0 and leaves with its units 0 and 1
    // (descriptor of the previous layer: hrs_prev); behind layer 7's last k step it is flushed.
    char* rt_lane_w = smem + OFF_RT + wave * 4096 + li * 128;      // + ((chunk ^ fl) << 4)
    const int fl = ((li >> 1) & 7) ^ ((li & 1) << 2);
    const char* rt_lane_r = smem + OFF_RT + wave * 4096 + lane * 16;  // + 1024 i
    uint32_t sv[4];
    {
      const int r8 = lane >> 3, p8 = lane & 7, f0 = (r8 >> 1) ^ ((r8 & 1) << 2);  // f(8 i + r8) = f0 ^ ((i & 1) << 2)
#pragma unroll
      for (int i = 0; i < 4; ++i) sv[i] = (uint32_t)(((p0 + 8 * i + r8) * a.ldh + 4 * (p8 ^ f0 ^ ((i & 1) << 2))) * 4);
    }
    rsrc_t hrs_prev = make_rsrc(nullptr, 0);
    f32x4 dv[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    f32x4 rb[4] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    // Epilogue of k step j = (nt, q) of the finished layer: its 8 values P[nt][8 q + i] (features 32 nt + 16 q + 8 (i / 4) +
    // 4 hh + i % 4) as a flat list of micro-operations, ROUND-MAJOR: a round works on the 8 values (one instruction each) or, where
    // a packed fp32 instruction exists, on the 4 value PAIRS (2 i, 2 i + 1) -- consecutive micro-operations are independent.
    //   sampler query: y (8), ys = c3 y (4, v_pk_mul_f32), KE |ys| (8), exp2 (8), 1 + e (4, v_pk_add_f32), log2 (8),
    //                  max(ys, 0) (8), CL log2 + max (4, v_pk_fma_f32) [skip override]                         = 52
    //   training:      y (8), ys (4), KE |ys| (8), exp2 (8), 1 + e (4), series (4 + 4 + 4 + 4), log2 (8), CL log2 (4), select (8),
    //                  max (8), + (4), [skip override] (8; the reference's `y > 0.2 -> y` branch needs no instruction: softplus(y) - y
    //                  <= 2.1e-11 there, far below half an ulp of y, so the sum IS y), 1 / SA = the stored value (4)   = 92
    //   then the limb split of the four dwords (16, round-major) and, training, the two 16-byte stores
    struct EpiState { f32x2 y[4], e[4], u[4], ser[4], r[4], o[4]; };
    constexpr int NSP = HEAD ? 52 : 92;
    constexpr int NOPS = NSP + 16 + (STORE ? 2 : 0);
    static_assert(NOPS == SchedEpi<HEAD>::N, "regenerate rmlp_h3_sched.h");
#define H3_PIN(x) asm volatile("" : "+v"(x))
#define H3_PINP(x)  // (no pin behind a transcendental: see exp2_v)
    // constants of the packed instructions as OPAQUE register pairs: with an inline constant or an SGPR operand the
    // instruction selector scalarises a <2 x float> operation into two v_*_f32 (packed fp32 takes no literal per half)
    f32x2 ONE2 = {1.0f, 1.0f}, CL2 = {CL, CL};
    H3_PIN(ONE2);
    H3_PIN(CL2);
    f32x2 THIRD2 = {0.33333334f, 0.33333334f}, MHALF2 = {-0.5f, -0.5f}, E001_2 = {0.01f * SA, 0.01f * SA}, ISA2 = {1.0f / SA, 1.0f / SA};
    if (!HEAD) { H3_PIN(THIRD2); H3_PIN(MHALF2); H3_PIN(E001_2); H3_PIN(ISA2); }
    auto epi_mop = [&](int layer, f32x2 c3, int j, int k, Limbs& out, EpiState& st) {
      const int nt = j >> 1, q = j & 1;
      // round boundaries (compile-time): operations of 8 (per value) or 4 (per pair)
      constexpr int bH[9] = {0, 8, 12, 20, 28, 32, 40, 48, 52};
      constexpr int bS[17] = {0, 8, 12, 20, 28, 32, 36, 40, 44, 48, 56, 60, 68, 76, 80, 88, 92};
      auto skip = [&](float r, int i) {  // skip connection: columns 217.. of layer 3's output are the embedding (shape_net.py:122-123)
        if (j >= 13) {
          const int m = 32 * nt + 16 * q + 8 * (i >> 2) + 4 * hh + (i & 3) - SKIP_OUT;
          const float ev = embw[li * EMB_STR + (m < 0 ? 0 : m)];
          r = (layer == 4 && m >= 0) ? ev : r;
        }
        return r;
      };
      if (k < NSP) {
        // round and position within it (plain arithmetic on k: a search loop over the table is not folded for 16 rounds and
        // the state arrays would be indexed dynamically, i.e. live in scratch)
        const int rd = HEAD ? (k >= 8) + (k >= 12) + (k >= 20) + (k >= 28) + (k >= 32) + (k >= 40) + (k >= 48)
                            : (k >= 8) + (k >= 12) + (k >= 20) + (k >= 28) + (k >= 32) + (k >= 36) + (k >= 40) + (k >= 44) +
                                  (k >= 48) + (k >= 56) + (k >= 60) + (k >= 68) + (k >= 76) + (k >= 80) + (k >= 88);
        const int i = k - (HEAD ? bH[rd] : bS[rd]);  // value 0..7 or pair 0..3 within the round
        const int p = i >> 1, c = i & 1;             // (per-value rounds) pair and component of value i
        if (STORE && k < 2)  // the fragments of the unit before (chunk 4 q' + 2 h2 + hh of row li; unit 0: the previous layer's unit 15)
          *reinterpret_cast<f32x4*>(rt_lane_w + (((4 * ((j + 1) & 1) + 2 * k + hh) ^ fl) << 4)) = dv[k];
        if (STORE && (j & 1) == 0 && k >= 2 && k < 6)  // ... which completed a tile: read it back (pieces 0..3)
          rb[k - 2] = *reinterpret_cast<const f32x4*>(rt_lane_r + (k - 2) * 1024);
        if (rd == 0) { st.y[p][c] = P[nt][8 * q + i]; H3_PIN(st.y[p]); }
        else if (rd == 1) { st.y[i] = pk_mul(st.y[i], c3); }
        else if (rd == 2) { st.e[p][c] = KE * fabsf(st.y[p][c]); H3_PIN(st.e[p]); }
        else if (rd == 3) {
          if (ABL & 1) { float t = st.e[p][c] * 1e-3f; H3_PIN(t); st.e[p][c] = t; }
          else { st.e[p][c] = exp2_v(st.e[p][c]); H3_PINP(st.e[p]); }
        }
        else if (rd == 4) { st.u[i] = pk_add(st.e[i], ONE2); }
        else if (HEAD) {
          if (rd == 5) {
            if (ABL & 2) { float t = st.u[p][c] * 0.5f; H3_PIN(t); st.u[p][c] = t; }
            else { st.u[p][c] = log2_v(st.u[p][c]); H3_PINP(st.u[p]); }
          }
          else if (rd == 6) { st.r[p][c] = relu1(st.y[p][c]); H3_PIN(st.r[p]); }
          else {
            f32x2 r = pk_fma(st.u[i], CL2, st.r[i]);
            if (j >= 13) { r[0] = skip(r[0], 2 * i); r[1] = skip(r[1], 2 * i + 1); H3_PIN(r); }
            st.r[i] = r;
          }
        } else {
          if (rd == 5) { st.ser[i] = pk_fma(st.e[i], THIRD2, MHALF2); }
          else if (rd == 6) { st.ser[i] = pk_fma(st.e[i], st.ser[i], ONE2); }
          else if (rd == 7) { st.e[i] = pk_mul(st.e[i], E001_2); }
          else if (rd == 8) { st.ser[i] = pk_mul(st.e[i], st.ser[i]); }
          else if (rd == 9) { st.u[p][c] = log2_v(st.u[p][c]); H3_PINP(st.u[p]); }
          else if (rd == 10) { st.u[i] = pk_mul(st.u[i], CL2); }
          else if (rd == 11) { st.u[p][c] = (st.e[p][c] > 1e-5f * SA) ? st.u[p][c] : st.ser[p][c]; H3_PIN(st.u[p]); }  // e > 1e-3: log(1 + e) accurate
          else if (rd == 12) { st.r[p][c] = relu1(st.y[p][c]); H3_PIN(st.r[p]); }
          else if (rd == 13) { st.r[i] = pk_add(st.r[i], st.u[i]); }
          else if (rd == 14) { if (j >= 13) { st.r[p][c] = skip(st.r[p][c], i); H3_PIN(st.r[p]); } }
          else { st.o[i] = pk_mul(st.r[i], ISA2); }
        }
      } else if (k < NSP + 16) {
        const int s = k - NSP, op = s >> 2, d = s & 3;
        split_op(op, d, st.r[d][0], st.r[d][1], out, ss);
      } else {  // STORE: the four consecutive features of half h2
        const int h2 = k - (NSP + 16);
        const f32x4 v = {st.o[2 * h2][0], st.o[2 * h2][1], st.o[2 * h2 + 1][0], st.o[2 * h2 + 1][1]};
        if (j == 0) store4(rb[h2], hrs_prev, sv[h2] + 128 * 7);            // the previous layer's last tile
        else if (j == 1) store4(rb[2 + h2], hrs_prev, sv[2 + h2] + 128 * 7);
        else if ((j & 1) == 0) store4(rb[h2], hrs, sv[h2] + 128 * (j / 2 - 1));
        else store4(rb[2 + h2], hrs, sv[2 + h2] + 128 * ((j - 3) / 2));
        dv[h2] = v;
      }
    };
    const char* lsrc = wsrc0 + (long)(L0S + AHEAD) * SLOT;  // this wave's pieces of the step AHEAD behind the layer's first
    for (int layer = 1; layer < 8; ++layer) {
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) {
        P[nt] = Q[nt];
        // the finished layer stays in the ACCUMULATOR half of the register file (read once, through v_accvgpr_read)
#pragma unroll
        for (int r = 0; r < 16; ++r) asm volatile("" : "+a"(P[nt][r]));
      }
      init_bias(layer);
      const float c3s = a.c3[layer - 1];
      f32x2 c3 = {c3s, c3s};
      H3_PIN(c3);
      EpiState st;
      if (STORE) {
        hrs_prev = hrs;
        hrs = make_rsrc(a.h[layer - 1], hbytes);
      }
#pragma unroll
      for (int k = 0; k < NOPS; ++k) epi_mop(layer, c3, 0, k, Bc, st);
#pragma unroll
      for (int j = 0; j < LKS; ++j) {
        // the stream wraps into the NEXT block's first steps behind the last layer
        const char* src = (j + AHEAD >= LKS && layer == 7) ? wsrc0 + (long)(j + AHEAD - LKS) * SLOT : lsrc + (long)j * SLOT;
        if (j + 1 < LKS)
          kstep(j, src, SchedEpi<HEAD>(), [&](int k) { epi_mop(layer, c3, j + 1, k, Bn, st); });
        else
          kstep(j, src, SchedNone(), no_mop);
      }
      lsrc += (long)LKS * SLOT;
    }
    if (STORE) {  // layer 6's last tile: unit 15's fragments, read-back, four whole-line stores (exposed, once per block)
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) *reinterpret_cast<f32x4*>(rt_lane_w + (((4 + 2 * h2 + hh) ^ fl) << 4)) = dv[h2];
#pragma unroll
      for (int i = 0; i < 4; ++i) rb[i] = *reinterpret_cast<const f32x4*>(rt_lane_r + i * 1024);
#pragma unroll
      for (int i = 0; i < 4; ++i) store4(rb[i], hrs, sv[i] + 128 * 7);
    }
    // ---- output of layer 7 ----
    {
      const float c1 = a.c3[7] * (1.0f / SA);
      const float* w8l = reinterpret_cast<const float*>(smem + OFF_W8);
      const rsrc_t h7rs = make_rsrc(STORE ? a.h[7] : nullptr, STORE ? hbytes : 0);
#pragma unroll
      for (int nt = 0; nt < 8; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int f0 = 32 * nt + 8 * g + 4 * hh;
          f32x4 v;
#pragma unroll
          for (int k = 0; k < 4; ++k) v[k] = HEAD ? sp_fast(c1 * Q[nt][4 * g + k]) : sp_train(c1 * Q[nt][4 * g + k]);
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
  H3_WAIT_VM(0);  // the stream runs AHEAD steps past the last block: no LDS-DMA may be in flight when the workgroup's LDS is released
  // 65504 = fp16's largest finite value (RN turns anything from 65520 on into an infinity; the 16 in between are given away)
  if (a.guard && mx >= 65504.f) __hip_atomic_store(a.guard, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace

extern "C" int64_t hold_trunk_h3_pack_bytes(void) { return (int64_t)NSTEP * SLOT; }
extern "C" float hold_trunk_h3_act_scale(void) { return SA; }

static int rmlp_h3_launch(const H3Args& a, bool head, hipStream_t s) {
  static int n_cu = 0;
  static bool attr_set = false;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return HOLD_E_LAUNCH;
    n_cu = prop.multiProcessorCount;
  }
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)rmlp_h3_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            LDS_BYTES) != hipSuccess ||
        hipFuncSetAttribute((const void*)rmlp_h3_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            LDS_BYTES) != hipSuccess)
      return HOLD_E_LAUNCH;
    attr_set = true;
  }
  const long blocks = (a.P + BPTS - 1) / BPTS;
  const dim3 grid((unsigned)(blocks < n_cu ? blocks : n_cu));
#ifdef HOLD_DEV
  if (const char* v = getenv("HOLD_H3_ABL")) {
    const int abl = atoi(v);
    if (head && abl >= 1) {
      auto k = abl == 3 ? rmlp_h3_kernel<true, false, 3> : abl == 4 ? rmlp_h3_kernel<true, false, 4> : abl == 8 ? rmlp_h3_kernel<true, false, 8>
             : abl == 12 ? rmlp_h3_kernel<true, false, 12> : abl == 16 ? rmlp_h3_kernel<true, false, 16> : abl == 28 ? rmlp_h3_kernel<true, false, 28>
             : rmlp_h3_kernel<true, false, 31>;
      if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) return HOLD_E_LAUNCH;
      hipLaunchKernelGGL(k, grid, dim3(256), LDS_BYTES, s, a);
      return hipGetLastError() == hipSuccess ? HOLD_OK : HOLD_E_LAUNCH;
    }
  }
#endif
  if (head)
    hipLaunchKernelGGL((rmlp_h3_kernel<true, false>), grid, dim3(256), LDS_BYTES, s, a);
  else
    hipLaunchKernelGGL((rmlp_h3_kernel<false, true>), grid, dim3(256), LDS_BYTES, s, a);
  return hipGetLastError() == hipSuccess ? HOLD_OK : HOLD_E_LAUNCH;
}

// SDF-only query of the sampler in the f16x3 arithmetic (the contract of hold_fused_sdf_r6 + the per-layer weight scales).
// `guard` (4 words of device memory, zero-initialised once, one per stream; may be null: overflow then goes unreported):
// [0] is set when a scaled activation left fp16's range.  With `wpack_r6` / `bias` (the operands of hold_fused_sdf_r6 for the
// same weights) the f32x6 kernel follows as a conditional launch (hold_fused_sdf_r6_if) that recomputes the query if and only
// if that happened, adds 1 to guard[2] and clears guard[0]: the result is then hold_fused_sdf_r6's, bit for bit.
extern "C" int hold_fused_sdf_h3(const float* xc, int32_t ldx, int64_t P, const void* wpack_h3, const float* bias_scaled,
                                 const float* c3, const float* w8, const float* b8, const float* barf_w, float* sdf,
                                 int32_t ld_sdf, uint32_t* guard, const void* wpack_r6, const float* bias,
                                 hold_stream_t st) {
  if (!xc || !wpack_h3 || !bias_scaled || !c3 || !w8 || !b8 || !sdf || ldx < 3 || ld_sdf < 1 || P < 0) return HOLD_E_ARG;
  if (((uintptr_t)wpack_h3 & 15) || ((uintptr_t)w8 & 15) || ((uintptr_t)bias_scaled & 15) || ((uintptr_t)guard & 3)) return HOLD_E_ARG;
  if ((wpack_r6 != nullptr) != (bias != nullptr) || (wpack_r6 && !guard)) return HOLD_E_ARG;
  if (P == 0) return HOLD_OK;
  H3Args a = {};
  a.xc = xc; a.ldx = ldx; a.P = (long)P; a.wpack = (const char*)wpack_h3; a.bias = bias_scaled; a.c3 = c3; a.w8 = w8;
  a.b8 = b8; a.barf = barf_w; a.sdf = sdf; a.lds = ld_sdf; a.guard = guard;
  const int rc = rmlp_h3_launch(a, true, (hipStream_t)st);
  if (rc != HOLD_OK || !wpack_r6) return rc;
  return hold_fused_sdf_r6_if(xc, ldx, P, wpack_r6, bias, w8, b8, barf_w, sdf, ld_sdf, guard, st);
}

// Training forward trunk in the f16x3 arithmetic: h[l] [P][ldh] (l = 0..7) = softplus outputs of lin0..lin7 (fp32,
// unscaled); columns 217..255 of h[3] receive the embedding (the contract of hold_trunk_r6).
// (`guard`, `wpack_r6`, `bias`: as for hold_fused_sdf_h3; the fallback is hold_trunk_r6_if)
extern "C" int hold_trunk_h3(const float* xc, int32_t ldx, int64_t P, const void* wpack_h3, const float* bias_scaled,
                             const float* c3, const float* barf_w, float* const* h, int32_t ldh, uint32_t* guard,
                             const void* wpack_r6, const float* bias, hold_stream_t st) {
  if (!xc || !wpack_h3 || !bias_scaled || !c3 || !h || ldx < 3 || ldh < 256 || (ldh & 3) || P < 0) return HOLD_E_ARG;
  if (((uintptr_t)wpack_h3 & 15) || ((uintptr_t)bias_scaled & 15) || ((uintptr_t)guard & 3)) return HOLD_E_ARG;
  if ((wpack_r6 != nullptr) != (bias != nullptr) || (wpack_r6 && !guard)) return HOLD_E_ARG;
  H3Args a = {};
  for (int l = 0; l < 8; ++l) {
    if (!h[l] || ((uintptr_t)h[l] & 15)) return HOLD_E_ARG;
    a.h[l] = h[l];
  }
  if (P == 0) return HOLD_OK;
  if (((uint64_t)P + 128) * (uint64_t)ldh * 4 >= (1ull << 32)) return HOLD_E_ARG;  // 32-bit buffer offsets: split by rows
  a.xc = xc; a.ldx = ldx; a.P = (long)P; a.wpack = (const char*)wpack_h3; a.bias = bias_scaled; a.c3 = c3; a.barf = barf_w;
  a.ldh = ldh; a.guard = guard;
  const int rc = rmlp_h3_launch(a, false, (hipStream_t)st);
  if (rc != HOLD_OK || !wpack_r6) return rc;
  return hold_trunk_r6_if(xc, ldx, P, wpack_r6, bias, barf_w, h, ldh, guard, st);
}
