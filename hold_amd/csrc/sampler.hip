// VolSDF error-bounded hierarchical ray sampler (Algorithm 1) for gfx950.
// Reference: code/src/engine/ray_sampler.py:54-80 (uniform), :128-352 (ErrorBoundSampler.get_z_vals),
// :354-366 (get_error_bound), code/src/engine/density.py:21-26 (Laplace density).
//
// One 64-lane wavefront owns one ray.  The ray's sample window (<= 640 + 128 z values, sdf, section
// bounds, pdf/cdf) lives in LDS; every lane owns a contiguous chunk of the window, runs the serial
// part of each cumulative sum in registers and the cross-lane part with wave shuffles.  The 10-step
// beta bisection re-evaluates the error bound entirely on-chip; the only HBM traffic is the window
// read (z, sdf) and the new samples written back (algorithmic bytes/ray/round = S*8 + 128*8 + 8).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hold_hip.h"

namespace {

constexpr int MAXS = 768;            // 640 merged + 128 new
constexpr int LPAD = MAXS + 32;      // + one pad word per 32 elements
constexpr int WAVES = 4;

__device__ __forceinline__ int pi_(int e) { return e + (e >> 5); }

// orders one wave's LDS traffic across lanes (LDS executes a wave's requests in issue order; the
// fences keep the compiler from caching or reordering across the point)
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float wave_incl_scan(float v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const float o = __shfl_up(v, d);
    if (lane >= d) v += o;
  }
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v = fmaxf(v, __shfl_xor(v, d));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
  return v;
}

__device__ __forceinline__ float laplace_density(float sdf, float beta) {
  const float sgn = (sdf > 0.f) ? 1.f : ((sdf < 0.f) ? -1.f : 0.f);
  return (1.0f / beta) * (0.5f + 0.5f * sgn * expm1f(-fabsf(sdf) / beta));
}

// in-place inclusive scan of arr[0..n) (padded LDS), lane-contiguous chunks of size C
__device__ __forceinline__ void scan_inplace(float* arr, int n, int C, int lane) {
  const int e0 = lane * C, e1 = min(n, e0 + C);
  float run = 0.f;
  for (int e = e0; e < e1; ++e) {
    run += arr[pi_(e)];
    arr[pi_(e)] = run;
  }
  const float incl = wave_incl_scan(run, lane);
  const float off = incl - run;
  for (int e = e0; e < e1; ++e) arr[pi_(e)] += off;
}

struct RayLds {
  float z[LPAD], sdf[LPAD], dist[LPAD], dstar[LPAD], a[LPAD], b[LPAD];
};

// error bound for one beta (ray_sampler.py:354-366); uses r.a / r.b as scratch.
__device__ float error_bound(RayLds& r, int S, int C, int lane, float beta) {
  const int n = S - 1;
  const int e0 = lane * C, e1 = min(n, e0 + C);
  const float ib = 1.0f / beta, q = 1.0f / (4.0f * beta * beta);
  float runE = 0.f, runI = 0.f;
  for (int e = e0; e < e1; ++e) {
    const float d = r.dist[pi_(e)];
    runE += expf(-r.dstar[pi_(e)] * ib) * (d * d) * q;
    r.a[pi_(e)] = runE;                 // inclusive error integral (lane-local)
    r.b[pi_(e)] = runI;                 // exclusive density integral (lane-local)
    runI += d * laplace_density(r.sdf[pi_(e)], beta);
  }
  const float offE = wave_incl_scan(runE, lane) - runE;
  const float offI = wave_incl_scan(runI, lane) - runI;
  float m = -3.0e38f;
  for (int e = e0; e < e1; ++e) {
    const float E = r.a[pi_(e)] + offE, I = r.b[pi_(e)] + offI;
    const float bo = (fminf(expf(E), 1.0e6f) - 1.0f) * expf(-I);
    m = fmaxf(m, bo);
  }
  return wave_max(m);
}

__device__ void load_window(RayLds& r, const float* z, const float* sdf, int S, int lane) {
  for (int e = lane; e < S; e += 64) {
    r.z[pi_(e)] = z[e];
    r.sdf[pi_(e)] = sdf[e];
  }
  wave_sync();
}

// dists + d* (Theorem 1 triangle bound, ray_sampler.py:191-206)
__device__ void section_bounds(RayLds& r, int S, int lane) {
  for (int e = lane; e < S - 1; e += 64) {
    const float z0 = r.z[pi_(e)], z1 = r.z[pi_(e + 1)];
    const float s0 = r.sdf[pi_(e)], s1 = r.sdf[pi_(e + 1)];
    const float a = z1 - z0, b = fabsf(s0), c = fabsf(s1);
    const bool first = a * a + b * b <= c * c;
    const bool second = a * a + c * c <= b * b;
    float ds = 0.f;
    if (first) ds = b;
    if (second) ds = c;
    const float s = (a + b + c) * 0.5f;
    const float area = s * (s - a) * (s - b) * (s - c);
    if (!first && !second && (b + c - a > 0.f)) ds = (2.0f * sqrtf(area)) / a;
    const float sg0 = (s0 > 0.f) ? 1.f : ((s0 < 0.f) ? -1.f : 0.f);
    const float sg1 = (s1 > 0.f) ? 1.f : ((s1 < 0.f) ? -1.f : 0.f);
    if (sg0 * sg1 != 1.0f) ds = 0.f;
    r.dist[pi_(e)] = a;
    r.dstar[pi_(e)] = ds;
  }
  wave_sync();
}

// ------------------------------------------------------------------------------------------
// init: far = sphere exit, z = uniform (stratified in training), beta = sqrt(bound)   (:54-80,:152-156)
// ------------------------------------------------------------------------------------------
__global__ void sampler_init_kernel(const float* __restrict__ cam, const float* __restrict__ dirs, long N, float R,
                                    float near, int n0, float eps, const float* __restrict__ t_rand,
                                    float* __restrict__ z, int ldz, float* __restrict__ beta,
                                    float* __restrict__ far_out, int* __restrict__ err_flag) {
  const int lane = threadIdx.x & 63;
  const long ray = (long)blockIdx.x * WAVES + (threadIdx.x >> 6);
  if (ray >= N) return;
  const float ox = cam[ray * 3], oy = cam[ray * 3 + 1], oz = cam[ray * 3 + 2];
  const float dx = dirs[ray * 3], dy = dirs[ray * 3 + 1], dz = dirs[ray * 3 + 2];
  const float rcd = dx * ox + dy * oy + dz * oz;
  const float under = rcd * rcd - ((ox * ox + oy * oy + oz * oz) - R * R);
  if (under <= 0.f && lane == 0) atomicExch(err_flag, 1);  // "BOUNDING SPHERE PROBLEM" (ray_sampler.py:16-18)
  const float far = fmaxf(sqrtf(fmaxf(under, 0.f)) - rcd, 0.f);
  if (lane == 0) far_out[ray] = far;
  __shared__ float zl[WAVES][LPAD];
  float* zw = zl[threadIdx.x >> 6];
  const float step = 1.0f / (float)(n0 - 1);
  float sumsq = 0.f;
  auto zlin = [&](int i) {
    // torch.linspace(0,1,n): start + i*step for the first half, end - (n-1-i)*step for the second
    const float t = (i < n0 / 2) ? (float)i * step : 1.0f - (float)(n0 - 1 - i) * step;
    return near * (1.0f - t) + far * t;
  };
  for (int e = lane; e < n0; e += 64) {
    float v = zlin(e);
    if (t_rand) {
      const float lo = (e == 0) ? v : 0.5f * (v + zlin(e - 1));
      const float up = (e == n0 - 1) ? v : 0.5f * (zlin(e + 1) + v);
      v = lo + (up - lo) * t_rand[ray * n0 + e];
    }
    zw[pi_(e)] = v;
    z[ray * ldz + e] = v;
  }
  wave_sync();
  for (int e = lane; e < n0 - 1; e += 64) {
    const float d = zw[pi_(e + 1)] - zw[pi_(e)];
    sumsq += d * d;
  }
  sumsq = wave_sum(sumsq);
  if (lane == 0) beta[ray] = sqrtf((1.0f / (4.0f * logf(eps + 1.0f))) * sumsq);
}

// ------------------------------------------------------------------------------------------
// beta line search for one round (:191-220).  Optionally scatters the freshly evaluated sdf of the
// previous round's new samples into their merged slots first (:179-189).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64 * WAVES) void sampler_beta_kernel(const float* __restrict__ z, float* __restrict__ sdf,
                                                                  int ld, int S, long N,
                                                                  const float* __restrict__ sdf_new,
                                                                  const int* __restrict__ slot, int n_new,
                                                                  float* __restrict__ beta, float beta0, float eps,
                                                                  int beta_iters, unsigned* __restrict__ maxbeta) {
  __shared__ RayLds lds[WAVES];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long ray = (long)blockIdx.x * WAVES + wv;
  if (ray >= N) return;
  RayLds& r = lds[wv];
  load_window(r, z + ray * ld, sdf + ray * ld, S, lane);
  if (sdf_new) {
    for (int j = lane; j < n_new; j += 64) {
      const int sl = slot[ray * n_new + j];
      const float v = sdf_new[ray * n_new + j];
      r.sdf[pi_(sl)] = v;
      sdf[ray * ld + sl] = v;
    }
    wave_sync();
  }
  section_bounds(r, S, lane);
  const int C = (S - 1 + 63) / 64;
  float b = beta[ray];
  float cur = error_bound(r, S, C, lane, beta0);
  if (cur <= eps) b = beta0;
  float bmin = beta0, bmax = b;
  for (int it = 0; it < beta_iters; ++it) {
    const float mid = (bmin + bmax) * 0.5f;
    cur = error_bound(r, S, C, lane, mid);
    if (cur <= eps) bmax = mid;
    if (cur > eps) bmin = mid;
  }
  if (lane == 0) {
    beta[ray] = bmax;
    atomicMax(maxbeta, __float_as_uint(bmax));
  }
}

// ------------------------------------------------------------------------------------------
// upsampling (:223-311): build the pdf (error-bound pdf when `more`, opacity-weight pdf for the final
// set), invert the CDF at u, and when `more` merge the new samples into the sorted window.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64 * WAVES) void sampler_sample_kernel(float* __restrict__ z, float* __restrict__ sdf,
                                                                    int ld, int S, long N,
                                                                    const float* __restrict__ beta, int more,
                                                                    float add_tiny, const float* __restrict__ u,
                                                                    long u_stride, int n_new,
                                                                    float* __restrict__ samples_out,
                                                                    int* __restrict__ slot_out) {
  __shared__ RayLds lds[WAVES];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long ray = (long)blockIdx.x * WAVES + wv;
  if (ray >= N) return;
  RayLds& r = lds[wv];
  load_window(r, z + ray * ld, sdf + ray * ld, S, lane);
  section_bounds(r, S, lane);
  const float bt = beta[ray];
  const int n = S - 1;  // number of sections / pdf entries
  const int C = (n + 63) / 64;
  const int e0 = lane * C, e1 = min(n, e0 + C);
  // transmittance T_e = exp(-sum_{i<e} dist_i * density_i), stored in r.b; weights / bound pdf in r.a
  {
    const float ib = 1.0f / bt, q = 1.0f / (4.0f * bt * bt);
    float runE = 0.f, runI = 0.f;
    for (int e = e0; e < e1; ++e) {
      const float d = r.dist[pi_(e)];
      runE += expf(-r.dstar[pi_(e)] * ib) * (d * d) * q;
      r.a[pi_(e)] = runE;
      r.b[pi_(e)] = runI;
      runI += d * laplace_density(r.sdf[pi_(e)], bt);
    }
    const float offE = wave_incl_scan(runE, lane) - runE;
    const float offI = wave_incl_scan(runI, lane) - runI;
    float psum = 0.f;
    for (int e = e0; e < e1; ++e) {
      const float T = expf(-(r.b[pi_(e)] + offI));
      float pdf;
      if (more) {
        pdf = (fminf(expf(r.a[pi_(e)] + offE), 1.0e6f) - 1.0f) * T + add_tiny;
      } else {
        const float fe = r.dist[pi_(e)] * laplace_density(r.sdf[pi_(e)], bt);
        pdf = (1.0f - expf(-fe)) * T + 1e-5f;
      }
      r.a[pi_(e)] = pdf;
      psum += pdf;
    }
    const float tot = wave_sum(psum);
    for (int e = e0; e < e1; ++e) r.a[pi_(e)] = r.a[pi_(e)] / tot;
  }
  scan_inplace(r.a, n, C, lane);  // r.a[e] = cdf[e+1]; cdf[0] = 0
  wave_sync();
  // ---- inverse CDF (:295-307); new samples -> r.b[0..n_new) ----
  for (int j = lane; j < n_new; j += 64) {
    const float uu = u[ray * u_stride + j];
    // inds = #{k in [0,S) : cdf[k] <= uu}, cdf[0] = 0, cdf[k] = r.a[k-1]
    int lo = 0, hi = S;  // count of elements <= uu via binary search on the non-decreasing cdf
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      const float cv = (mid == 0) ? 0.f : r.a[pi_(mid - 1)];
      if (cv <= uu) lo = mid + 1; else hi = mid;
    }
    const int inds = lo;
    const int below = max(inds - 1, 0), above = min(inds, S - 1);
    const float c0 = (below == 0) ? 0.f : r.a[pi_(below - 1)];
    const float c1 = (above == 0) ? 0.f : r.a[pi_(above - 1)];
    const float b0 = r.z[pi_(below)], b1 = r.z[pi_(above)];
    float denom = c1 - c0;
    if (denom < 1e-5f) denom = 1.0f;
    const float t = (uu - c0) / denom;
    const float smp = b0 + t * (b1 - b0);
    r.dist[pi_(j)] = smp;  // stash (dist no longer needed)
    samples_out[ray * n_new + j] = smp;
  }
  if (!more) return;
  wave_sync();
  // ---- stable merge of [old z (S) | new samples (n_new)] (:311): old first on ties ----
  // new sample j goes to j + #{old <= s_j}; old element i goes to i + #{new < z_i}
  for (int j = lane; j < n_new; j += 64) {
    const float s = r.dist[pi_(j)];
    int lo = 0, hi = S;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (r.z[pi_(mid)] <= s) lo = mid + 1; else hi = mid;
    }
    const int pos = j + lo;
    slot_out[ray * n_new + j] = pos;
    z[ray * ld + pos] = s;
  }
  for (int i = lane; i < S; i += 64) {
    const float zi = r.z[pi_(i)];
    int lo = 0, hi = n_new;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (r.dist[pi_(mid)] < zi) lo = mid + 1; else hi = mid;
    }
    const int pos = i + lo;
    z[ray * ld + pos] = zi;
    sdf[ray * ld + pos] = r.sdf[pi_(i)];
  }
}

// ------------------------------------------------------------------------------------------
// final set (:313-336): sort([z_samples (ns), near, far, z[:, idx_extra] (nx)]) ascending
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64 * WAVES) void sampler_final_kernel(const float* __restrict__ zs, int ns,
                                                                   const float* __restrict__ z, int ld,
                                                                   const int* __restrict__ idx_extra, int nx,
                                                                   const float* __restrict__ far, float near, long N,
                                                                   float* __restrict__ out, int ldo) {
  __shared__ float buf[WAVES][256];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long ray = (long)blockIdx.x * WAVES + wv;
  if (ray >= N) return;
  float* b = buf[wv];
  const int tot = ns + 2 + nx;  // <= 256
  for (int e = lane; e < 256; e += 64) {
    float v = 3.0e38f;
    if (e < ns) v = zs[ray * ns + e];
    else if (e == ns) v = near;
    else if (e == ns + 1) v = far[ray];
    else if (e < tot) v = z[ray * ld + idx_extra[e - ns - 2]];
    b[e] = v;
  }
  wave_sync();
  // bitonic sort of 256 keys, 4 per lane (values only: equal keys are interchangeable)
  for (int k = 2; k <= 256; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int e = lane; e < 256; e += 64) {
        const int ixj = e ^ j;
        if (ixj > e) {
          const float x = b[e], y = b[ixj];
          const bool up = (e & k) == 0;
          if ((x > y) == up) {
            b[e] = y;
            b[ixj] = x;
          }
        }
      }
      wave_sync();
    }
  }
  for (int e = lane; e < tot; e += 64) out[ray * ldo + e] = b[e];
}

inline int ok() { return hipGetLastError() == hipSuccess ? HOLD_OK : HOLD_E_LAUNCH; }

}  // namespace

extern "C" int hold_sampler_init(const float* cam_loc, const float* ray_dirs, int64_t n_rays, float R, float near,
                                 int32_t n0, float eps, const float* t_rand, float* z, int32_t ldz, float* beta,
                                 float* far_out, int32_t* err_flag, hold_stream_t st) {
  if (!cam_loc || !ray_dirs || !z || !beta || !far_out || !err_flag || n0 < 2 || n0 > MAXS) return HOLD_E_ARG;
  if (n_rays == 0) return HOLD_OK;
  hipLaunchKernelGGL(sampler_init_kernel, dim3((unsigned)((n_rays + WAVES - 1) / WAVES)), dim3(64 * WAVES), 0,
                     (hipStream_t)st, cam_loc, ray_dirs, (long)n_rays, R, near, n0, eps, t_rand, z, ldz, beta, far_out,
                     err_flag);
  return ok();
}

extern "C" int hold_sampler_beta(const float* z, float* sdf, int32_t ld, int32_t S, int64_t n_rays,
                                 const float* sdf_new, const int32_t* slot, int32_t n_new, float* beta, float beta0,
                                 float eps, int32_t beta_iters, uint32_t* maxbeta_bits, hold_stream_t st) {
  if (!z || !sdf || !beta || !maxbeta_bits || S < 2 || S > MAXS || (sdf_new && !slot)) return HOLD_E_ARG;
  if (n_rays == 0) return HOLD_OK;
  hipLaunchKernelGGL(sampler_beta_kernel, dim3((unsigned)((n_rays + WAVES - 1) / WAVES)), dim3(64 * WAVES), 0,
                     (hipStream_t)st, z, sdf, ld, S, (long)n_rays, sdf_new, slot, n_new, beta, beta0, eps, beta_iters,
                     maxbeta_bits);
  return ok();
}

extern "C" int hold_sampler_sample(float* z, float* sdf, int32_t ld, int32_t S, int64_t n_rays, const float* beta,
                                   int32_t more, float add_tiny, const float* u, int64_t u_stride, int32_t n_new,
                                   float* samples_out, int32_t* slot_out, hold_stream_t st) {
  if (!z || !sdf || !beta || !u || !samples_out || S < 2 || n_new < 1 || n_new > MAXS || (more && !slot_out) ||
      (more && S + n_new > MAXS) || (more && S + n_new > ld))
    return HOLD_E_ARG;
  if (n_rays == 0) return HOLD_OK;
  hipLaunchKernelGGL(sampler_sample_kernel, dim3((unsigned)((n_rays + WAVES - 1) / WAVES)), dim3(64 * WAVES), 0,
                     (hipStream_t)st, z, sdf, ld, S, (long)n_rays, beta, more, add_tiny, u, (long)u_stride, n_new,
                     samples_out, slot_out);
  return ok();
}

extern "C" int hold_sampler_final(const float* z_samples, int32_t ns, const float* z, int32_t ld,
                                  const int32_t* idx_extra, int32_t nx, const float* far, float near, int64_t n_rays,
                                  float* out, int32_t ldo, hold_stream_t st) {
  if (!z_samples || !z || !far || !out || ns + 2 + nx > 256 || (nx > 0 && !idx_extra)) return HOLD_E_ARG;
  if (n_rays == 0) return HOLD_OK;
  hipLaunchKernelGGL(sampler_final_kernel, dim3((unsigned)((n_rays + WAVES - 1) / WAVES)), dim3(64 * WAVES), 0,
                     (hipStream_t)st, z_samples, ns, z, ld, idx_extra, nx, far, near, (long)n_rays, out, ldo);
  return ok();
}
