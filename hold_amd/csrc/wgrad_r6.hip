// Weight gradient of a 256 x 256 layer with the register-resident recipe of rmlp.hip (gfx950):
//   part[g][n][k] = sum over workgroup g's points of R[p][n] X[p][k]      (+ part_b[g][n] = sum R[p][n])
// ONE wave per SIMD; the four waves of a workgroup own the four 128 x 128 quadrants of the WHOLE dW (16 accumulator tiles =
// 256 accumulator registers each), so every row of R and X crosses the fabric once (the 128 x 256 tiles of
// wgrad_lds_kernel read X twice) and a fragment's limb split serves 24 MFMAs instead of 12 (3 VALU per MFMA, not 4.5).
// Split-precision arithmetic of hold_wgrad_x6: both operands split exactly into three bf16 limbs (truncation), six limb
// products on v_mfma_f32_32x32x16_bf16, fp32 accumulation.
//
// A step = 16 points.  Raw rows [16][256] of R and of X go to a 4-slot LDS ring by LDS-DMA (32 KiB per step, three steps
// ahead; each wave requests eight 1 KiB rows, the 16-byte chunk index XOR-swizzled by the row half on the SOURCE address so
// that the column reads of the two lane halves hit disjoint banks).  An MFMA operand fragment = one column, 8 consecutive
// points per lane half: 8 ds_read_b32 + 36 VALU (split) per fragment, 8 fragments per wave and step, prepared for step
// t + 1 behind the 96 MFMAs of step t with the explicit per-gap schedule of rmlp.hip (one MFMA, then a fixed slice of
// reads / split micro-operations / DMA pieces, closed by a scheduling barrier); ONE workgroup barrier per step.
// Roofline: bf16 MFMA pipe (6 limb products per algorithmic product); HBM 2 KiB per point (each operand row once).
#include <hip/hip_runtime.h>
#include <stdint.h>
#ifdef HOLD_DEV
#include <stdio.h>
#include <stdlib.h>
#endif

#include "../../include/hold_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int NST = 4;            // ring slots
constexpr int STAGE = 32 * 1024;  // one step: R rows [16][256] fp32, then X rows [16][256]
constexpr int LDS_BYTES = NST * STAGE;
constexpr int LDS_BYTES_H3 = LDS_BYTES + 64;  // + the workgroup reductions of the scale sample and of the exact maxima

struct WArgs {
  const float* R; int ldr;
  const float* X; int ldx;
  long nsteps;   // full 16-point steps (P / 16)
  int spw;       // steps per workgroup
  float* part;   // [G][256][256]
  float* part_b; // [G][256] or null
};

__device__ __forceinline__ uint32_t fbits(float x) { return __builtin_bit_cast(uint32_t, x); }
__device__ __forceinline__ float bitsf(uint32_t x) { return __builtin_bit_cast(float, x); }

// one 1 KiB row piece: lane l fetches 16 bytes at src + voff(l), they land lane-linear at LDS byte dst (M0)
__device__ __forceinline__ void dma_piece(const char* src, uint32_t voff, uint32_t dst) {
  asm volatile(
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %0, %1"
      :
      : "v"(voff), "s"(src), "s"(dst)
      : "memory");
}
#define WG_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

struct Limbs8 { u32x4 l[8][3]; };  // fragments 0..3 = columns of R (MFMA A operand), 4..7 = columns of X (B operand)

// state of one fragment while it is being split: the 8 raw values and the intermediate remainders, as 4 value pairs
struct Frag { float x[8]; u32x2 r1[4], r2[4]; };

// micro-operation k (0..35) of the truncation split of fragment f: operation k / 4 on value pair k % 4 (consecutive
// micro-operations are independent).  Per pair: p1 = high halves of x; r1 = x - hi(x); p2 = high halves of r1;
// r2 = r1 - hi(r1); p3 = high halves of r2.
// Every result is made opaque (empty asm with a "+v" operand): pure VALU code has no side effects, so without the pin LLVM
// sinks the whole split to its first use -- the next step's MFMAs -- instead of leaving it in the slices of this step.
__device__ __forceinline__ void pin(uint32_t& x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void split_mop(Frag& s, u32x4 (&out)[3], int k) {
  const int op = k >> 2, j = k & 3;
  const f32x2 v = {s.x[2 * j], s.x[2 * j + 1]};
  const u32x2 xb = __builtin_bit_cast(u32x2, v);
  uint32_t t0, t1;
  if (op == 0) { t0 = __builtin_amdgcn_perm(xb[1], xb[0], 0x07060302u); pin(t0); out[0][j] = t0; }
  else if (op == 1) { t0 = xb[0] & 0xffff0000u; pin(t0); s.r2[j][0] = t0; }
  else if (op == 2) { t0 = xb[1] & 0xffff0000u; pin(t0); s.r2[j][1] = t0; }
  else if (op == 3) {
    const u32x2 r = __builtin_bit_cast(u32x2, v - __builtin_bit_cast(f32x2, s.r2[j]));
    t0 = r[0]; t1 = r[1]; pin(t0); pin(t1);
    s.r1[j][0] = t0; s.r1[j][1] = t1;
  }
  else if (op == 4) { t0 = __builtin_amdgcn_perm(s.r1[j][1], s.r1[j][0], 0x07060302u); pin(t0); out[1][j] = t0; }
  else if (op == 5) { t0 = s.r1[j][0] & 0xffff0000u; pin(t0); s.r2[j][0] = t0; }
  else if (op == 6) { t0 = s.r1[j][1] & 0xffff0000u; pin(t0); s.r2[j][1] = t0; }
  else if (op == 7) {
    const u32x2 r = __builtin_bit_cast(u32x2, __builtin_bit_cast(f32x2, s.r1[j]) - __builtin_bit_cast(f32x2, s.r2[j]));
    t0 = r[0]; t1 = r[1]; pin(t0); pin(t1);
    s.r2[j][0] = t0; s.r2[j][1] = t1;
  }
  else { t0 = __builtin_amdgcn_perm(s.r2[j][1], s.r2[j][0], 0x07060302u); pin(t0); out[2][j] = t0; }
}

// schedule of the preparation of step t + 1 behind the 96 MFMAs of step t: fragment f's reads occupy the gaps
// [rs(f), rs(f) + 8), its 36 split micro-operations the window [ws(f), ws(f + 1)) -- three gaps behind its last read
constexpr int rs(int f) { return 85 * f / 8; }
constexpr int ws(int f) { return 11 + 85 * f / 8; }

// the steps [s0, s1) of one (R, X) pair by one workgroup -> its partial tile out[256][256] (and out_b[256])
// ABL (developer builds only, results garbage): 1 = no fragment reads / limb splits in the steps (the first step's limbs
// are reused), 2 = no LDS-DMA in the steps, 3 = no MFMAs -- what each of the three streams costs on its own
// DIST = how many steps ahead of the MFMAs the rows are requested (3 or 4 -- with 4 the request made during step t goes to
// the slot of step t itself, whose rows were consumed by the preparation that ran during step t - 1)
template <bool BIAS, int ABL = 0, int DIST = 3>
__device__ __forceinline__ void wgrad_r6_body(const float* R, int ldr, const float* X, int ldx, long s0, long s1,
                                              float* out, float* out_b) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hh = lane >> 5, li = lane & 31;
  const int wn = wave >> 1, wk = wave & 1;  // this wave's quadrant: rows n = 128 wn .., columns k = 128 wk ..

  // ---- LDS-DMA of this wave's eight rows of a step: waves 0, 1 -> R rows 0..7 / 8..15, waves 2, 3 -> X rows
  const float* M = (wave >> 1) ? X : R;
  const long ldm = (wave >> 1) ? ldx : ldr;
  const uint32_t dvoff = (uint32_t)((lane ^ (8 * (wave & 1))) * 16);  // chunk swizzle: rows 8..15 swap 32-column halves
  const long rowb = ldm * 4;                                            // bytes between rows
  const char* mrow0 = reinterpret_cast<const char*>(M) + (wave & 1) * 8 * rowb;  // this wave's first row of step 0
  const uint32_t dst0 = (uint32_t)((wave >> 1) * 16384 + (wave & 1) * 8 * 1024);
  // rows of step u (beyond the range: the last step again, which keeps the vmcnt arithmetic uniform); p walks the rows
  auto step_src = [&](long u) { return mrow0 + (ABL == 4 ? s0 + (u & 1) : (u < s1 ? u : s1 - 1)) * 16 * rowb; };
  // ---- fragment reads: column c = 32 f' + li of the wave's 128 (f' = 0..3), rows 8 hh .. 8 hh + 7 of the 16; the row
  // half's swizzle flips bit 5 of the column = bit 7 of the byte address, so fragment f' is an XOR of the base address
  const uint32_t rdA = (uint32_t)((8 * hh) * 1024 + ((128 * wn + li) ^ (32 * hh)) * 4);
  const uint32_t rdB = (uint32_t)(16384 + (8 * hh) * 1024 + ((128 * wk + li) ^ (32 * hh)) * 4);
  auto frag_addr = [&](int f, uint32_t slot_base) {
    return smem + slot_base + ((f < 4 ? rdA : rdB) ^ (uint32_t)((f & 3) << 7));
  };

  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float bsum[2] = {0.f, 0.f};  // BIAS: column sums of the R fragments 2 wk, 2 wk + 1 of this wave's n range

  if (s0 < s1) {
#pragma unroll
    for (int u = 0; u < DIST; ++u) {
      const char* p = step_src(s0 + u);
#pragma unroll
      for (int i = 0; i < 8; ++i) dma_piece(p + i * rowb, dvoff, (uint32_t)((int)((s0 + u) % NST) * STAGE) + dst0 + i * 1024);
    }
    const char* dptr = step_src(s0 + DIST);  // rows of the step requested during the running one
    WG_WAIT_VM(8 * (DIST - 1));
    __builtin_amdgcn_s_barrier();

    Limbs8 L0, L1;
    Frag fs[2];
    // column sums of R for the bias gradient: wave (wn, wk) keeps those of the fragments 2 wk, 2 wk + 1 -- branch-free (a
    // wave-uniform SELECT: a branch here would end the scheduling region and make the LDS wait counts pessimistic; a 0 / 1
    // FACTOR, as in round 3, let a NaN / Inf of a column the wave does not own -- with N < 256 the columns N..255 of R are
    // whatever the caller left there -- into the sums of the columns it owns: 0 x NaN).  Nothing is kept for the rows
    // prepared during the LAST step: they are the re-read of that step, not new points.
    bool sel[2] = {wk == 0, wk == 1};
    auto bias_add = [&](int f, const Frag& s) {
      if (BIAS && f < 4) {
        const f32x2 p0 = {s.x[0], s.x[1]}, p1 = {s.x[2], s.x[3]}, p2 = {s.x[4], s.x[5]}, p3 = {s.x[6], s.x[7]};
        const f32x2 q = (p0 + p1) + (p2 + p3);
        const float add = sel[f >> 1] ? q[0] + q[1] : 0.f;
        bsum[f & 1] += add;
      }
    };
    {  // limbs of the first step (not overlapped)
      const uint32_t sb = (uint32_t)((int)(s0 % NST) * STAGE);
#pragma unroll
      for (int f = 0; f < 8; ++f) {
        const char* p = frag_addr(f, sb);
#pragma unroll
        for (int e = 0; e < 8; ++e) fs[0].x[e] = *reinterpret_cast<const float*>(p + e * 1024);
        bias_add(f, fs[0]);
#pragma unroll
        for (int k = 0; k < 36; ++k) split_mop(fs[0], L0.l[f], k);
      }
    }

    // One step: 96 MFMAs on the limbs C of step t; behind each a slice of the preparation of step t + 1 into N.
    auto step = [&](long t, Limbs8& C, Limbs8& N) {
      if (ABL == 2) WG_WAIT_VM(0); else
      WG_WAIT_VM(8 * (DIST - 2));  // the rows of step t + 1 have landed (younger: the pieces of the steps t + 2 .. t + DIST - 1)
      __builtin_amdgcn_s_barrier();  // ... in every wave; and every wave is done reading the slot of step t
      const uint32_t sb = (uint32_t)((int)((t + 1) % NST) * STAGE);
      const uint32_t db = (uint32_t)((int)((t + DIST) % NST) * STAGE) + dst0;
      const char* p = dptr;
      dptr = (ABL != 4 && t + DIST + 1 < s1) ? dptr + 16 * rowb : dptr;
      if (BIAS && t + 1 >= s1) sel[0] = sel[1] = false;
#pragma unroll
      for (int m = 0; m < 96; ++m) {
        const int aa = m / 24, pr = (m % 24) / 4, bb = m % 4;  // four accumulators in rotation
        const int il = (pr == 2 || pr == 3) ? 1 : (pr == 5 ? 2 : 0);  // (R limb, X limb): 00 01 10 11 02 20
        const int jl = (pr == 1 || pr == 3) ? 1 : (pr == 4 ? 2 : 0);
        if (ABL != 3)
          acc[aa][bb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, C.l[aa][il]),
                                                                __builtin_bit_cast(bf16x8, C.l[4 + bb][jl]), acc[aa][bb], 0, 0, 0);
        if (ABL != 2 && m % 12 == 5) {
          dma_piece(p, dvoff, db + (m / 12) * 1024);
          p += rowb;
        }
#pragma unroll
        for (int f = 0; f < (ABL == 1 ? 0 : 8); ++f) {
          if (m >= rs(f) && m < rs(f) + 8) {
            const int e = m - rs(f);
            fs[f & 1].x[e] = *reinterpret_cast<const float*>(frag_addr(f, sb) + e * 1024);
          }
          const int w0 = ws(f), w1 = ws(f + 1) < 96 ? ws(f + 1) : 96, len = w1 - w0;
          if (m >= w0 && m < w1) {
            if (m == w0) bias_add(f, fs[f & 1]);
#pragma unroll
            for (int u = 0; u < 5; ++u) {
              const int k = 36 * (m - w0) / len + u;
              if (k < 36 * (m - w0 + 1) / len) split_mop(fs[f & 1], N.l[f], k);
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };

    for (long t = s0; t < s1; t += 2) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) asm volatile("" : "+a"(acc[i][j][r]));  // the accumulators live in the AGPR half
      step(t, L0, ABL == 1 ? L0 : L1);
      if (t + 1 < s1) step(t + 1, ABL == 1 ? L0 : L1, L0);
    }
    WG_WAIT_VM(0);  // no LDS-DMA may still be in flight when the workgroup's LDS is released
  }

  // ---- partial sums of this workgroup: lane (hh, li) holds column k = 128 wk + 32 b + li, rows 8 g + 4 hh + r ----
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = 128 * wn + 32 * i + 8 * (r >> 2) + 4 * hh + (r & 3);
        out[n * 256 + 128 * wk + 32 * j + li] = acc[i][j][r];
      }
  if (BIAS) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const float s = bsum[q] + __shfl_xor(bsum[q], 32);
      if (hh == 0) out_b[128 * wn + 32 * (2 * wk + q) + li] = s;
    }
  }
}


// =====================================================================================================================
// The same weight gradient in the TWO-LIMB fp16 arithmetic "f16x3" (see csrc/rmlp_h3.hip): both operands, scaled by a power
// of two, as hi = RN_f16(s x), lo = RN_f16(s x - hi); hi hi + hi lo + lo hi on v_mfma_f32_32x32x16_f16 -- 48 MFMAs per
// 16-point step instead of 96, a limb split of 5 instructions per two values (v_pk_mul_f32 by the scale, v_cvt_pk_f16_f32,
// 2 x v_fma_mix_f32, v_cvt_pk_f16_f32) instead of 9 + 2 pins, and the fragment's 8 rows read as 4 ds_read2st64_b32.
// SCALES.  The operands are activations (O(1)) and loss cotangents (1e-9 .. 1e-3 and anything else): every WORKGROUP picks
// its own pair of scales from a sample of ITS rows (16 groups of four rows, evenly spread; all 256 -- for R: the first n_valid -- columns)
// so that the sampled maximum lands in [2^6, 2^7): 2^9 of headroom below fp16's largest value for rows the sample did
// not see, full 22-23-bit precision for every value within 2^-8 of the sampled maximum and an absolute 2^-31 of it below.
// Loss cotangents are heavy-tailed (compositing weights span thirty orders of magnitude; a sharp density makes it worse), so a
// row outside the sample CAN exceed that headroom: every pass therefore keeps the exact maximum of the scaled values of each
// operand (one v_max3_f32 per value pair), and a workgroup that saw one beyond 65504 repeats its share once with that
// operand's scale taken from the exact maximum -- no overflow ever reaches dW (GPU call 12 of round 5: the first version,
// without this, returned NaN gradients on the beta = 0.005 scene).  The opposite failure -- every sampled row exactly zero, so a
// scale of 1 for cotangents of 1e-9 in the rows not sampled -- is caught by the same maximum (non-zero but below 2^4) and repeated
// the same way (round 6).  Workgroups need not agree on scales: a partial tile is
// multiplied by 1 / (s_R s_X) (exact) before it is written, the reduction adds unscaled fp32 tiles as before.
struct LimbsH { u32x4 l[8][2]; };
struct FragH { float x[8]; f32x2 xs[4]; uint32_t hi[4]; float ra[4], rb[4]; };

__device__ __forceinline__ uint32_t cvt_pk_h(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
}
// micro-operation k (0..19) of the split of a fragment: operation k / 4 on value pair k % 4
// `mx` collects the EXACT maximum of the scaled magnitudes (one v_max3_f32 per pair): the scales come from a sample, and a row
// the sample did not see may exceed fp16's range -- the workgroup then knows, and repeats its share with exact scales
__device__ __forceinline__ void split_mop_h(FragH& s, u32x4 (&out)[2], int k, float scale, float& mx) {
  const int op = k >> 2, j = k & 3;
  if (op == 0) {
    f32x2 v = {s.x[2 * j], s.x[2 * j + 1]};
    v = v * scale;
    asm volatile("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(mx) : "v"(v[0]), "v"(v[1]));
    asm volatile("" : "+v"(v));
    s.xs[j] = v;
  } else if (op == 1) {
    uint32_t h = cvt_pk_h(s.xs[j][0], s.xs[j][1]);
    pin(h);
    s.hi[j] = h;
    out[0][j] = h;
  } else if (op == 2) {
    asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(s.ra[j]) : "v"(s.hi[j]), "v"(s.xs[j][0]));
  } else if (op == 3) {
    asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(s.rb[j]) : "v"(s.hi[j]), "v"(s.xs[j][1]));
  } else {
    uint32_t lo = cvt_pk_h(s.ra[j], s.rb[j]);
    pin(lo);
    out[1][j] = lo;
  }
}
// preparation of step t + 1 behind the 48 MFMAs of step t: fragment f's four row-pair reads in the gaps [rsh(f), rsh(f) + 4),
// its 20 split micro-operations in [wsh(f), wsh(f + 1)) -- two gaps behind its last read
constexpr int rsh(int f) { return 42 * f / 8; }
constexpr int wsh(int f) { return f >= 8 ? 48 : 6 + 42 * f / 8; }

// power-of-two scale exponent for an operand whose maximum is am: 2^k am in [2^top, 2^(top + 1)); |k| <= 60.  top = 6 for a
// SAMPLED maximum (2^9 of headroom for the rows not sampled), 13 for an exact one
__device__ __forceinline__ int scale_exp(float am, int top = 6) {
  const int e = (int)((fbits(am) >> 23) & 0xffu);
  int k = 127 + top - e;
  k = k > 60 ? 60 : (k < -60 ? -60 : k);
  return am > 0.f ? k : 0;
}

template <bool BIAS>
__device__ __forceinline__ void wgrad_h3_body(const float* R, int ldr, const float* X, int ldx, int n_valid, long s0, long s1,
                                              float* out, float* out_b) {
  constexpr int DIST = 3;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hh = lane >> 5, li = lane & 31;
  const int wn = wave >> 1, wk = wave & 1;

  const float* M = (wave >> 1) ? X : R;
  const long ldm = (wave >> 1) ? ldx : ldr;
  const uint32_t dvoff = (uint32_t)((lane ^ (8 * (wave & 1))) * 16);
  const long rowb = ldm * 4;
  const char* mrow0 = reinterpret_cast<const char*>(M) + (wave & 1) * 8 * rowb;
  const uint32_t dst0 = (uint32_t)((wave >> 1) * 16384 + (wave & 1) * 8 * 1024);
  auto step_src = [&](long u) { return mrow0 + (u < s1 ? u : s1 - 1) * 16 * rowb; };
  const uint32_t rdA = (uint32_t)((8 * hh) * 1024 + ((128 * wn + li) ^ (32 * hh)) * 4);
  const uint32_t rdB = (uint32_t)(16384 + (8 * hh) * 1024 + ((128 * wk + li) ^ (32 * hh)) * 4);
  auto frag_addr = [&](int f, uint32_t slot_base) {
    return smem + slot_base + ((f < 4 ? rdA : rdB) ^ (uint32_t)((f & 3) << 7));
  };

  f32x16 acc[4][4];
  float bsum[2] = {0.f, 0.f};
  float inv = 1.0f;
  float* red = reinterpret_cast<float*>(smem + LDS_BYTES);  // 16 floats behind the ring: workgroup reductions

  auto preload = [&]() {
#pragma unroll
    for (int u = 0; u < DIST; ++u) {
      const char* p = step_src(s0 + u);
#pragma unroll
      for (int i = 0; i < 8; ++i) dma_piece(p + i * rowb, dvoff, (uint32_t)((int)((s0 + u) % NST) * STAGE) + dst0 + i * 1024);
    }
  };
  auto wg_max2 = [&](float& a, float& b) {  // maxima over the workgroup (NaNs are ignored by v_max)
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      a = fmaxf(a, __shfl_xor(a, o));
      b = fmaxf(b, __shfl_xor(b, o));
    }
    __syncthreads();
    if (lane == 0) { red[wave] = a; red[4 + wave] = b; }
    __syncthreads();
    a = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    b = fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]));
  };

  int kR = 0, kX = 0;
  if (s0 < s1) {
    preload();  // the first DIST steps' rows are requested FIRST: the scale sample below runs while they travel
    // ---- this workgroup's scales: a sample of its rows -- at most 16 passes of four consecutive rows (a row = 64 threads x
    // 16 bytes), evenly spread ----
    const long r0 = s0 * 16, nrows = (s1 - s0) * 16;
    long stride = (nrows / 16 + 3) & ~3L;
    stride = stride < 4 ? 4 : stride;
    const int c4 = (tid & 63) * 4, sub = tid >> 6;
    float amR = 0.f, amX = 0.f;
    for (long r = 0; r + 4 <= nrows; r += stride) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(R + (r0 + r + sub) * (long)ldr + c4);
      const f32x4 b = *reinterpret_cast<const f32x4*>(X + (r0 + r + sub) * (long)ldx + c4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (c4 + e < n_valid) amR = fmaxf(amR, fabsf(a[e]));  // columns >= n_valid of R are not the caller's data
        amX = fmaxf(amX, fabsf(b[e]));
      }
    }
    wg_max2(amR, amX);
    kR = __builtin_amdgcn_readfirstlane(scale_exp(amR));
    kX = __builtin_amdgcn_readfirstlane(scale_exp(amX));
  }
  // A pass over the workgroup's steps with the scales 2^kR, 2^kX.  It keeps the EXACT maxima of the scaled operands; if one
  // exceeds fp16's largest value (a row the sample did not see: heavy-tailed loss cotangents do that -- compositing weights
  // span thirty orders of magnitude), the pass is repeated ONCE with that operand's scale taken from its exact maximum.
  for (int attempt = 0; attempt < 2 && s0 < s1; ++attempt) {
    const float sR = bitsf((uint32_t)(127 + kR) << 23), sX = bitsf((uint32_t)(127 + kX) << 23);
    inv = bitsf((uint32_t)(127 - kR - kX) << 23);
    // R's scale per lane and fragment: 0 for the columns >= n_valid (their content is not the caller's data -- NaN / inf
    // included: it then neither counts as an overflow nor reaches a row of dW that is reduced)
    float sRf[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) sRf[f] = (128 * wn + 32 * f + li < n_valid) ? sR : 0.f;
    float mR = 0.f, mX = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    bsum[0] = bsum[1] = 0.f;
    if (attempt > 0) preload();
    const char* dptr = step_src(s0 + DIST);
    WG_WAIT_VM(8 * (DIST - 1));
    __builtin_amdgcn_s_barrier();

    LimbsH L0, L1;
    FragH fs[2];
    bool sel[2] = {wk == 0, wk == 1};
    auto bias_add = [&](int f, const FragH& s) {
      if (BIAS && f < 4) {
        const f32x2 p0 = {s.x[0], s.x[1]}, p1 = {s.x[2], s.x[3]}, p2 = {s.x[4], s.x[5]}, p3 = {s.x[6], s.x[7]};
        const f32x2 q = (p0 + p1) + (p2 + p3);
        const float add = sel[f >> 1] ? q[0] + q[1] : 0.f;
        bsum[f & 1] += add;
      }
    };
    {  // limbs of the first step (not overlapped)
      const uint32_t sb = (uint32_t)((int)(s0 % NST) * STAGE);
#pragma unroll
      for (int f = 0; f < 8; ++f) {
        const char* p = frag_addr(f, sb);
#pragma unroll
        for (int e = 0; e < 8; ++e) fs[0].x[e] = *reinterpret_cast<const float*>(p + e * 1024);
        bias_add(f, fs[0]);
#pragma unroll
        for (int k = 0; k < 20; ++k) split_mop_h(fs[0], L0.l[f], k, f < 4 ? sRf[f & 3] : sX, f < 4 ? mR : mX);
      }
    }

    auto step = [&](long t, LimbsH& C, LimbsH& N) {
      WG_WAIT_VM(8 * (DIST - 2));
      __builtin_amdgcn_s_barrier();
      const uint32_t sb = (uint32_t)((int)((t + 1) % NST) * STAGE);
      const uint32_t db = (uint32_t)((int)((t + DIST) % NST) * STAGE) + dst0;
      const char* p = dptr;
      dptr = (t + DIST + 1 < s1) ? dptr + 16 * rowb : dptr;
      if (BIAS && t + 1 >= s1) sel[0] = sel[1] = false;
#pragma unroll
      for (int m = 0; m < 48; ++m) {
        const int aa = m / 12, pr = (m % 12) / 4, bb = m % 4;  // four accumulators in rotation; (R limb, X limb): hh, hl, lh
        const int il = pr == 2 ? 1 : 0, jl = pr == 1 ? 1 : 0;
        acc[aa][bb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, C.l[aa][il]),
                                                             __builtin_bit_cast(f16x8, C.l[4 + bb][jl]), acc[aa][bb], 0, 0, 0);
        if (m % 6 == 2) {
          dma_piece(p, dvoff, db + (m / 6) * 1024);
          p += rowb;
        }
#pragma unroll
        for (int f = 0; f < 8; ++f) {
          if (m >= rsh(f) && m < rsh(f) + 4) {  // two rows per gap: adjacent loads 1 KiB apart = one ds_read2st64_b32
            const int e = 2 * (m - rsh(f));
            const char* q = frag_addr(f, sb);
            fs[f & 1].x[e] = *reinterpret_cast<const float*>(q + e * 1024);
            fs[f & 1].x[e + 1] = *reinterpret_cast<const float*>(q + (e + 1) * 1024);
          }
          const int w0 = wsh(f), w1 = wsh(f + 1), len = w1 - w0;
          if (m >= w0 && m < w1) {
            if (m == w0) {
              // ONE wait for the fragment's four reads (a real s_waitcnt, which the compiler's own wait insertion takes into
              // account: it otherwise puts a wait in front of every first use, 33 per step): LDS reads return in order, so
              // only the reads of fragment f + 1 issued in the meantime may stay outstanding
              constexpr int nx = 0;
              const int after = f < 7 ? (w0 - rsh(f + 1) < 0 ? 0 : (w0 - rsh(f + 1) > 4 ? 4 : w0 - rsh(f + 1))) : 0;
              if (after == 0) __builtin_amdgcn_s_waitcnt(0xC07F);
              else if (after == 1) __builtin_amdgcn_s_waitcnt(0xC17F);
              else if (after == 2) __builtin_amdgcn_s_waitcnt(0xC27F);
              else if (after == 3) __builtin_amdgcn_s_waitcnt(0xC37F);
              else __builtin_amdgcn_s_waitcnt(0xC47F);
              (void)nx;
              bias_add(f, fs[f & 1]);
            }
#pragma unroll
            for (int u = 0; u < 5; ++u) {
              const int k = 20 * (m - w0) / len + u;
              if (k < 20 * (m - w0 + 1) / len) split_mop_h(fs[f & 1], N.l[f], k, f < 4 ? sRf[f & 3] : sX, f < 4 ? mR : mX);
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };

    for (long t = s0; t < s1; t += 2) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) asm volatile("" : "+a"(acc[i][j][r]));
      step(t, L0, L1);
      if (t + 1 < s1) step(t + 1, L1, L0);
    }
    WG_WAIT_VM(0);  // no LDS-DMA in flight when the ring is refilled (or released)
    // ---- did a scaled value leave fp16's range?  (65504 is the largest finite value; a NaN operand is not an overflow) ----
    // ... or stay far BELOW the range the sample aimed at?  The sample only sees a subset of the rows: when every sampled row of
    // an operand is exactly zero (rays that miss the node, alpha == 0) its scale is 1, and cotangents of 1e-9 .. 1e-4 in the rows
    // it did not see would be split at that scale -- lo in fp16's subnormal range, hi with <= 11 bits, anything below 3e-8 flushed
    // (advisor, round 5).  A non-zero exact maximum below 2^4 (the sampled maximum lands in [2^6, 2^7), and the exact one is never
    // below the sampled one) can only mean that: the operand is re-scaled from its exact maximum like an overflowing one.
    wg_max2(mR, mX);
    const bool ovR = mR > 65504.f || (mR > 0.f && mR < 16.f), ovX = mX > 65504.f || (mX > 0.f && mX < 16.f);
    if (!(ovR || ovX)) break;
    if (ovR) kR = __builtin_amdgcn_readfirstlane(scale_exp(mR * bitsf((uint32_t)(127 - kR) << 23), 13));
    if (ovX) kX = __builtin_amdgcn_readfirstlane(scale_exp(mX * bitsf((uint32_t)(127 - kX) << 23), 13));
  }
  if (!(s0 < s1)) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  }

#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = 128 * wn + 32 * i + 8 * (r >> 2) + 4 * hh + (r & 3);
        out[n * 256 + 128 * wk + 32 * j + li] = acc[i][j][r] * inv;
      }
  if (BIAS) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const float s = bsum[q] + __shfl_xor(bsum[q], 32);
      if (hh == 0) out_b[128 * wn + 32 * (2 * wk + q) + li] = s;
    }
  }
}

template <bool BIAS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void wgrad_h3_kernel(WArgs a, int n_valid) {
  const long s0 = (long)blockIdx.x * a.spw;
  const long s1 = s0 + a.spw < a.nsteps ? s0 + a.spw : a.nsteps;
  wgrad_h3_body<BIAS>(a.R, a.ldr, a.X, a.ldx, n_valid, s0, s1, a.part + (long)blockIdx.x * 65536,
                      BIAS ? a.part_b + (long)blockIdx.x * 256 : nullptr);
}

template <bool BIAS, int ABL = 0, int DIST = 3>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void wgrad_r6_kernel(WArgs a) {
  const long s0 = (long)blockIdx.x * a.spw;
  const long s1 = s0 + a.spw < a.nsteps ? s0 + a.spw : a.nsteps;
  wgrad_r6_body<BIAS, ABL, DIST>(a.R, a.ldr, a.X, a.ldx, s0, s1, a.part + (long)blockIdx.x * 65536,
                      BIAS ? a.part_b + (long)blockIdx.x * 256 : nullptr);
}

// ---- several (R, X) pairs over the same points in ONE launch (hold_wgrad_group_x6): workgroup b works on pair
// b / gper, its share b % gper of the steps.  With n pairs a pair's points are split over gper = CUs / n workgroups
// instead of over all of them, so a launch writes CUs partial tiles for ALL pairs together (64 MiB) where n single
// launches write n x 64 MiB -- at 125k points (the two-hand configuration's chunks) the partial tiles of a single launch
// are half the operand bytes -- and one reduction pass replaces 2 n.
constexpr int WG_MAX_ITEMS = 24;
struct WPair { const float* R; const float* X; int ldr; int ldx; int n; };
struct WGroupArgs {
  WPair it[WG_MAX_ITEMS];
  long nsteps;
  int gper, spw;
  float* part;    // [n x gper][256][256]
  float* part_b;  // [n x gper][256]
};

template <bool H3>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void wgrad_r6_group_kernel(WGroupArgs a) {
  const int item = blockIdx.x / a.gper, share = blockIdx.x % a.gper;
  // a select chain, not a[item]: a dynamic index into the by-value argument would send the table through scratch
  const float* R = a.it[0].R;
  const float* X = a.it[0].X;
  int ldr = a.it[0].ldr, ldx = a.it[0].ldx, nv = a.it[0].n;
#pragma unroll
  for (int i = 1; i < WG_MAX_ITEMS; ++i)
    if (item == i) { R = a.it[i].R; X = a.it[i].X; ldr = a.it[i].ldr; ldx = a.it[i].ldx; nv = a.it[i].n; }
  const long s0 = (long)share * a.spw;
  const long s1 = s0 + a.spw < a.nsteps ? s0 + a.spw : a.nsteps;
  if (H3)
    wgrad_h3_body<true>(R, ldr, X, ldx, nv, s0, s1, a.part + (long)blockIdx.x * 65536, a.part_b + (long)blockIdx.x * 256);
  else
    wgrad_r6_body<true>(R, ldr, X, ldx, s0, s1, a.part + (long)blockIdx.x * 65536, a.part_b + (long)blockIdx.x * 256);
}

// destination d of the group = the pairs [first, first + count) (consecutive in the list): dW[:N][:256] (+)= the sum of
// their count x gper partial tiles in a fixed order (16 strided chains, then the 16 chain sums in order: deterministic);
// block x = 1024 reduces the bias sums of the pairs whose bit is set in bias_items.
struct WDst { float* dW; float* db; int lddw, N, first, count, accumulate; unsigned bias_items; };
struct WReduceArgs { WDst d[WG_MAX_ITEMS]; int gper; const float* part; const float* part_b; };

__global__ __launch_bounds__(256) void wgrad_group_reduce_kernel(WReduceArgs a) {
  __shared__ f32x4 red[16][16];
  const int di = blockIdx.y;
  WDst d = a.d[0];
#pragma unroll
  for (int i = 1; i < WG_MAX_ITEMS; ++i)
    if (di == i) d = a.d[i];
  if (blockIdx.x == 1024) {
    const int n = threadIdx.x;
    if (!d.db || n >= d.N) return;
    float s = 0.f;
    for (int it = 0; it < d.count; ++it)
      if ((d.bias_items >> it) & 1u)
        for (int sp = 0; sp < a.gper; ++sp) s += a.part_b[((long)(d.first + it) * a.gper + sp) * 256 + n];
    d.db[n] = d.accumulate ? d.db[n] + s : s;
    return;
  }
  const int e = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int i = (blockIdx.x * 16 + e) * 4, n = i >> 8, k = i & 255;
  if (n >= d.N) return;  // block-uniform: a block's 64 elements lie in one row
  const float* p = a.part + (long)d.first * a.gper * 65536 + i;
  const int np = d.count * a.gper;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  for (int sp = g; sp < np; sp += 16) s += *reinterpret_cast<const f32x4*>(p + (long)sp * 65536);
  red[g][e] = s;
  __syncthreads();
  if (g != 0) return;
#pragma unroll
  for (int j = 1; j < 16; ++j) s += red[j][e];
  float* o = d.dW + (long)n * d.lddw + k;
  if (d.accumulate) s += *reinterpret_cast<const f32x4*>(o);
  *reinterpret_cast<f32x4*>(o) = s;
}

}  // namespace

// Called by hold_wgrad_x6 (gemm.hip) for 256 columns of R and of X, P a multiple of 16: fills part[G][256][256] (and
// part_b[G][256]) with G <= max_splits workgroup partials and returns G (< 0: error); the caller reduces them -- all 256
// rows, or the first N when R has fewer meaningful columns (rows are independent: what columns N..255 of R hold, NaN
// included, only reaches the rows >= N of a partial tile).
// compute units of the current device (0: query failed); the kernels' LDS size attribute is set on the first call
static int wgrad_r6_setup() {
  static int n_cu = 0;
  static bool attr_set = false;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    n_cu = prop.multiProcessorCount;
  }
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)wgrad_r6_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) !=
            hipSuccess ||
        hipFuncSetAttribute((const void*)wgrad_r6_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) !=
            hipSuccess ||
        hipFuncSetAttribute((const void*)wgrad_r6_group_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) !=
            hipSuccess ||
        hipFuncSetAttribute((const void*)wgrad_r6_group_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES_H3) !=
            hipSuccess ||
        hipFuncSetAttribute((const void*)wgrad_h3_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES_H3) !=
            hipSuccess ||
        hipFuncSetAttribute((const void*)wgrad_h3_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES_H3) !=
            hipSuccess)
      return 0;
    attr_set = true;
  }
  return n_cu;
}

int hold_wgrad_r6_partials(const float* R, int ldr, const float* X, int ldx, long P, int max_splits, float* part,
                           float* part_b, hipStream_t s) {
  const int n_cu = wgrad_r6_setup();
  if (n_cu <= 0) return HOLD_E_LAUNCH;
  WArgs a;
  a.R = R; a.ldr = ldr; a.X = X; a.ldx = ldx; a.nsteps = P / 16; a.part = part; a.part_b = part_b;
  long G = n_cu < max_splits ? n_cu : max_splits;
  if (G > a.nsteps) G = a.nsteps;
  a.spw = (int)((a.nsteps + G - 1) / G);
  G = (a.nsteps + a.spw - 1) / a.spw;  // every workgroup owns at least one step
#ifdef HOLD_DEV  // timing ablations (developer build only): what the limb preparation / the LDS-DMA / the MFMAs cost alone
  if (const char* ab = getenv("HOLD_WGRAD_ABL")) {
    const int v = atoi(ab);
    static bool warned = false;
    if (v && !warned) fprintf(stderr, "libholdhip: HOLD_WGRAD_ABL=%d -- timing ablation, hold_wgrad_x6 results are WRONG\n", v);
    warned = true;
    if (v >= 1 && v <= 6) {  // 4: rows from two L2-resident steps; 5: requests four steps ahead; 6: both
      auto k = v == 1 ? wgrad_r6_kernel<false, 1> : v == 2 ? wgrad_r6_kernel<false, 2> : v == 3 ? wgrad_r6_kernel<false, 3>
             : v == 4 ? wgrad_r6_kernel<false, 4> : v == 5 ? wgrad_r6_kernel<false, 0, 4> : wgrad_r6_kernel<false, 4, 4>;
      if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess)
        return HOLD_E_LAUNCH;
      hipLaunchKernelGGL(k, dim3((unsigned)G), dim3(256), LDS_BYTES, s, a);
      return hipGetLastError() == hipSuccess ? (int)G : HOLD_E_LAUNCH;
    }
  }
#endif
  if (part_b)
    hipLaunchKernelGGL((wgrad_r6_kernel<true>), dim3((unsigned)G), dim3(256), LDS_BYTES, s, a);
  else
    hipLaunchKernelGGL((wgrad_r6_kernel<false>), dim3((unsigned)G), dim3(256), LDS_BYTES, s, a);
  return hipGetLastError() == hipSuccess ? (int)G : HOLD_E_LAUNCH;
}

// the f16x3 twin of hold_wgrad_r6_partials; n_valid = the columns of R that are the caller's data (the scale sample reads
// only those)
int hold_wgrad_h3_partials(const float* R, int ldr, const float* X, int ldx, long P, int n_valid, int max_splits, float* part,
                           float* part_b, hipStream_t s) {
  const int n_cu = wgrad_r6_setup();
  if (n_cu <= 0) return HOLD_E_LAUNCH;
  WArgs a;
  a.R = R; a.ldr = ldr; a.X = X; a.ldx = ldx; a.nsteps = P / 16; a.part = part; a.part_b = part_b;
  long G = n_cu < max_splits ? n_cu : max_splits;
  if (G > a.nsteps) G = a.nsteps;
  a.spw = (int)((a.nsteps + G - 1) / G);
  G = (a.nsteps + a.spw - 1) / a.spw;
  if (part_b)
    hipLaunchKernelGGL((wgrad_h3_kernel<true>), dim3((unsigned)G), dim3(256), LDS_BYTES_H3, s, a, n_valid);
  else
    hipLaunchKernelGGL((wgrad_h3_kernel<false>), dim3((unsigned)G), dim3(256), LDS_BYTES_H3, s, a, n_valid);
  return hipGetLastError() == hipSuccess ? (int)G : HOLD_E_LAUNCH;
}

extern "C" int64_t hold_wgrad_group_workspace_floats(void) {
  const int n_cu = wgrad_r6_setup();
  return n_cu > 0 ? (int64_t)(n_cu > WG_MAX_ITEMS ? n_cu : WG_MAX_ITEMS) * (65536 + 256) : -1;
}

static int wgrad_group_impl(const hold_wgrad_item* items, int32_t n_items, int64_t P, float* workspace, hold_stream_t stream,
                            bool h3) {
  if (!items || n_items <= 0 || n_items > WG_MAX_ITEMS || !workspace || P < 16 || (P % 16)) return HOLD_E_ARG;
  const int n_cu = wgrad_r6_setup();
  if (n_cu <= 0) return HOLD_E_LAUNCH;
  WGroupArgs a;
  WReduceArgs r;
  int nd = 0;
  for (int i = 0; i < n_items; ++i) {
    const hold_wgrad_item& it = items[i];
    if (!it.R || !it.X || !it.dW || it.N <= 0 || it.N > 256 || it.ldr < 256 || it.ldx < 256 || it.lddw < 256 ||
        (it.ldr & 3) || (it.ldx & 3) || (it.lddw & 3) || ((uintptr_t)it.R & 15) || ((uintptr_t)it.X & 15) ||
        ((uintptr_t)it.dW & 15))
      return HOLD_E_ARG;
    a.it[i] = WPair{it.R, it.X, it.ldr, it.ldx, it.N};
    if (nd > 0 && r.d[nd - 1].dW == it.dW) {  // same destination as the previous pair: one reduction over both
      WDst& d = r.d[nd - 1];
      if (d.N != it.N || d.lddw != it.lddw || (it.db && d.db && it.db != d.db)) return HOLD_E_ARG;
      if (it.db) { d.db = it.db; d.bias_items |= 1u << d.count; }
      ++d.count;
    } else {
      for (int j = 0; j < nd; ++j)
        if (r.d[j].dW == it.dW) return HOLD_E_ARG;  // pairs of one destination must be adjacent
      r.d[nd++] = WDst{it.dW, it.db, it.lddw, it.N, i, 1, it.accumulate, it.db ? 1u : 0u};
    }
  }
  for (int i = n_items; i < WG_MAX_ITEMS; ++i) a.it[i] = a.it[0];
  for (int i = nd; i < WG_MAX_ITEMS; ++i) r.d[i] = r.d[0];
  a.nsteps = P / 16;
  long gper = n_cu / n_items > 0 ? n_cu / n_items : 1;
  if (gper > a.nsteps) gper = a.nsteps;
  a.spw = (int)((a.nsteps + gper - 1) / gper);
  gper = (a.nsteps + a.spw - 1) / a.spw;
  a.gper = (int)gper;
  a.part = workspace;
  a.part_b = workspace + (long)n_items * gper * 65536;
  r.gper = a.gper; r.part = a.part; r.part_b = a.part_b;
  hipStream_t s = (hipStream_t)stream;
  if (h3)
    hipLaunchKernelGGL(wgrad_r6_group_kernel<true>, dim3((unsigned)(n_items * gper)), dim3(256), LDS_BYTES_H3, s, a);
  else
    hipLaunchKernelGGL(wgrad_r6_group_kernel<false>, dim3((unsigned)(n_items * gper)), dim3(256), LDS_BYTES, s, a);
  hipLaunchKernelGGL(wgrad_group_reduce_kernel, dim3(1025, nd), dim3(256), 0, s, r);
  return hipGetLastError() == hipSuccess ? HOLD_OK : HOLD_E_LAUNCH;
}

extern "C" int hold_wgrad_group_x6(const hold_wgrad_item* items, int32_t n_items, int64_t P, float* workspace,
                                   hold_stream_t stream) {
  return wgrad_group_impl(items, n_items, P, workspace, stream, false);
}

// the same grouped launch in the two-limb fp16 arithmetic (wgrad_h3_body: per-workgroup operand scales)
extern "C" int hold_wgrad_group_h3(const hold_wgrad_item* items, int32_t n_items, int64_t P, float* workspace,
                                   hold_stream_t stream) {
  return wgrad_group_impl(items, n_items, P, workspace, stream, true);
}
