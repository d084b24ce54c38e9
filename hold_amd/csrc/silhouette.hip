// Pose-refinement inner loop kernels (gfx950): differentiable soft-silhouette rasteriser and K=1 nearest neighbour.
// Reference call sites: code/src/fitting/model.py:109-144 (fwd_params: MeshRenderer(SoftSilhouetteShader)),
// code/src/fitting/utils.py:101-158 (BlendParams sigma 1e-6, blur_radius log(1/1e-4 - 1) * sigma, faces_per_pixel 100,
// PerspectiveCameras in_ndc=False, R = diag(-1,-1,1)), code/src/fitting/loss.py:84-165 (knn_points K=1 contact terms).
// The rasteriser itself is pytorch3d 0.7.4 (not in the reference tree): restated from its documented behaviour, see
// oracle/fitting_oracle.py -- parity of that dependency is unpinned.
//
// alpha(pixel) = 1 - prod_f sigmoid(d_f / sigma), d_f = signed squared NDC distance of the pixel centre to face f
// (negative inside), over faces with d_f < blur_radius in front of the camera.  One thread per pixel, 16x16 pixel tiles;
// faces are set up once per launch (projection, bbox) and culled per tile with a wave ballot into an LDS list.
// VALU-bound: ~60 flop per (pixel, surviving face).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hold_hip.h"

namespace {

struct FaceSetup {  // 12 floats
  float ax, ay, bx, by, cx, cy, xmin, xmax, ymin, ymax, valid, pad;
};

__global__ void face_setup_kernel(const float* __restrict__ v3d, int V, const int* __restrict__ faces, int F, int B,
                                  float fx, float fy, float cx, float cy, int H, int W, float blur,
                                  FaceSetup* __restrict__ fs) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * F) return;
  const int b = (int)(i / F), f = (int)(i % F);
  const float s = 0.5f * (float)(H < W ? H : W);
  float x[3], y[3], z[3];
  for (int k = 0; k < 3; ++k) {
    const float* p = v3d + ((long)b * V + faces[f * 3 + k]) * 3;
    z[k] = p[2];
    x[k] = -((fx * p[0] / p[2] + cx) - 0.5f * W) / s;
    y[k] = -((fy * p[1] / p[2] + cy) - 0.5f * H) / s;
  }
  const float r = sqrtf(blur);
  FaceSetup o;
  o.ax = x[0]; o.ay = y[0]; o.bx = x[1]; o.by = y[1]; o.cx = x[2]; o.cy = y[2];
  o.xmin = fminf(fminf(x[0], x[1]), x[2]) - r;
  o.xmax = fmaxf(fmaxf(x[0], x[1]), x[2]) + r;
  o.ymin = fminf(fminf(y[0], y[1]), y[2]) - r;
  o.ymax = fmaxf(fmaxf(y[0], y[1]), y[2]) + r;
  const float area = (x[1] - x[0]) * (y[2] - y[0]) - (y[1] - y[0]) * (x[2] - x[0]);
  const float zmin = fminf(fminf(z[0], z[1]), z[2]), zmax = fmaxf(fmaxf(z[0], z[1]), z[2]);
  o.valid = (fabsf(area) > 1e-8f && zmin > 0.f) ? 1.f : 0.f;
  // a face that STRADDLES the image plane (some vertices behind the camera, some in front): pytorch3d drops it per pixel
  // (interpolated depth < 0, rasterize_meshes.cu), this kernel as a whole -- the two agree only while no such face exists;
  // hold_silhouette_max_faces reports them (pad = 1) so that the caller can refuse the configuration
  o.pad = (zmin <= 0.f && zmax > 0.f && fabsf(area) > 1e-8f) ? 1.f : 0.f;
  fs[i] = o;
}

__device__ __forceinline__ float seg_d2(float px, float py, float ax, float ay, float bx, float by, float& t) {
  const float abx = bx - ax, aby = by - ay;
  const float den = fmaxf(abx * abx + aby * aby, 1e-20f);
  t = fminf(fmaxf(((px - ax) * abx + (py - ay) * aby) / den, 0.f), 1.f);
  const float qx = ax + t * abx, qy = ay + t * aby;
  return (px - qx) * (px - qx) + (py - qy) * (py - qy);
}

// signed squared distance; returns false when the face does not contribute
__device__ __forceinline__ bool face_dist(const FaceSetup& f, float px, float py, float blur, float& d, int& edge,
                                          float& t) {
  const float e0 = (f.bx - f.ax) * (py - f.ay) - (f.by - f.ay) * (px - f.ax);
  const float e1 = (f.cx - f.bx) * (py - f.by) - (f.cy - f.by) * (px - f.bx);
  const float e2 = (f.ax - f.cx) * (py - f.cy) - (f.ay - f.cy) * (px - f.cx);
  const bool inside = (e0 >= 0.f && e1 >= 0.f && e2 >= 0.f) || (e0 <= 0.f && e1 <= 0.f && e2 <= 0.f);
  float t0, t1, t2;
  const float d0 = seg_d2(px, py, f.ax, f.ay, f.bx, f.by, t0);
  const float d1 = seg_d2(px, py, f.bx, f.by, f.cx, f.cy, t1);
  const float d2 = seg_d2(px, py, f.cx, f.cy, f.ax, f.ay, t2);
  float dm = d0; edge = 0; t = t0;
  if (d1 < dm) { dm = d1; edge = 1; t = t1; }
  if (d2 < dm) { dm = d2; edge = 2; t = t2; }
  if (!inside && dm >= blur) return false;
  d = inside ? -dm : dm;
  return true;
}

constexpr int TILE = 16, LIST = 1024;
constexpr int HOLD_SILHOUETTE_STRADDLE = 1 << 30;  // hold_silhouette_max_faces: "a face straddles the image plane"

template <bool BWD>
__global__ __launch_bounds__(256) void silhouette_kernel(const FaceSetup* __restrict__ fs, const int* __restrict__ faces,
                                                        int F, int H, int W, float sigma, float blur,
                                                        float* __restrict__ mask, const float* __restrict__ dmask,
                                                        float* __restrict__ dndc, int V) {
  __shared__ FaceSetup sf[LIST];
  __shared__ int sidx[LIST];
  __shared__ int scount;
  const int b = blockIdx.z;
  const int tx = threadIdx.x % TILE, ty = threadIdx.x / TILE;
  const int j = blockIdx.x * TILE + tx, i = blockIdx.y * TILE + ty;
  const float s = 0.5f * (float)(H < W ? H : W);
  const float px = -((j + 0.5f) - 0.5f * W) / s, py = -((i + 0.5f) - 0.5f * H) / s;
  // tile bounds in NDC (x decreases with j)
  const float txmax = -((blockIdx.x * TILE + 0.5f) - 0.5f * W) / s, txmin = -((blockIdx.x * TILE + TILE - 0.5f) - 0.5f * W) / s;
  const float tymax = -((blockIdx.y * TILE + 0.5f) - 0.5f * H) / s, tymin = -((blockIdx.y * TILE + TILE - 0.5f) - 0.5f * H) / s;
  const bool live = (i < H && j < W);
  float prod = 1.f;
  const FaceSetup* fb = fs + (long)b * F;
  // pass structure: gather up to LIST overlapping faces, process, repeat
  for (int f0 = 0; f0 < F;) {
    if (threadIdx.x == 0) scount = 0;
    __syncthreads();
    int fend = f0;
    // scan faces in strides of 256 until the list could overflow
    for (; fend < F; fend += 256) {
      if (scount + 256 > LIST) break;
      const int f = fend + threadIdx.x;
      bool hit = false;
      FaceSetup cur;
      if (f < F) {
        cur = fb[f];
        hit = cur.valid != 0.f && cur.xmax >= txmin && cur.xmin <= txmax && cur.ymax >= tymin && cur.ymin <= tymax;
      }
      if (hit) {
        const int pos = atomicAdd(&scount, 1);
        sf[pos] = cur;
        sidx[pos] = f;
      }
      __syncthreads();
    }
    const int n = scount;
    if (!BWD) {
      if (live)
        for (int q = 0; q < n; ++q) {
          float d, t;
          int e;
          if (face_dist(sf[q], px, py, blur, d, e, t)) prod *= 1.0f / (1.0f + __expf(-d / sigma));
        }
    } else {
      // backward needs the full product first: accumulate here, distribute in the second sweep below
      if (live)
        for (int q = 0; q < n; ++q) {
          float d, t;
          int e;
          if (face_dist(sf[q], px, py, blur, d, e, t)) prod *= 1.0f / (1.0f + __expf(-d / sigma));
        }
    }
    __syncthreads();
    f0 = fend;
  }
  if (!BWD) {
    if (live) mask[((long)b * H + i) * W + j] = 1.0f - prod;
    return;
  }
  // ---- backward: second sweep distributes d mask / d d_f = -(1/sigma) p_f prod to the nearest edge's vertices ----
  const float g = live ? dmask[((long)b * H + i) * W + j] : 0.f;
  const float coef = -g * prod / sigma;
  for (int f0 = 0; f0 < F;) {
    if (threadIdx.x == 0) scount = 0;
    __syncthreads();
    int fend = f0;
    for (; fend < F; fend += 256) {
      if (scount + 256 > LIST) break;
      const int f = fend + threadIdx.x;
      bool hit = false;
      FaceSetup cur;
      if (f < F) {
        cur = fb[f];
        hit = cur.valid != 0.f && cur.xmax >= txmin && cur.xmin <= txmax && cur.ymax >= tymin && cur.ymin <= tymax;
      }
      if (hit) {
        const int pos = atomicAdd(&scount, 1);
        sf[pos] = cur;
        sidx[pos] = f;
      }
      __syncthreads();
    }
    const int n = scount;
    if (live && coef != 0.f)
      for (int q = 0; q < n; ++q) {
        float d, t;
        int e;
        const FaceSetup& fc = sf[q];
        if (!face_dist(fc, px, py, blur, d, e, t)) continue;
        const float one_m_p = 1.0f / (1.0f + __expf(-d / sigma));
        const float dd = coef * (1.0f - one_m_p);  // dL/d d_f
        if (dd == 0.f) continue;
        const float ax = (e == 0) ? fc.ax : (e == 1) ? fc.bx : fc.cx, ay = (e == 0) ? fc.ay : (e == 1) ? fc.by : fc.cy;
        const float bx = (e == 0) ? fc.bx : (e == 1) ? fc.cx : fc.ax, by = (e == 0) ? fc.by : (e == 1) ? fc.cy : fc.ay;
        const float qx = ax + t * (bx - ax), qy = ay + t * (by - ay);
        const float sg = (d < 0.f) ? -1.f : 1.f;
        const float gx = sg * dd * (-2.0f) * (px - qx), gy = sg * dd * (-2.0f) * (py - qy);
        const int f = sidx[q];
        const int va = faces[f * 3 + e], vb = faces[f * 3 + (e + 1) % 3];
        float* pa = dndc + ((long)b * V + va) * 2;
        float* pb = dndc + ((long)b * V + vb) * 2;
        atomicAdd(pa, (1.0f - t) * gx);
        atomicAdd(pa + 1, (1.0f - t) * gy);
        atomicAdd(pb, t * gx);
        atomicAdd(pb + 1, t * gy);
      }
    __syncthreads();
    f0 = fend;
  }
}

__global__ void ndc_bwd_kernel(const float* __restrict__ v3d, const float* __restrict__ dndc, long n, float fx, float fy,
                               int H, int W, float* __restrict__ dv3d) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float s = 0.5f * (float)(H < W ? H : W);
  const float x = v3d[i * 3], y = v3d[i * 3 + 1], z = v3d[i * 3 + 2];
  const float gx = dndc[i * 2], gy = dndc[i * 2 + 1];
  // x_ndc = -(fx x / z + cx - W/2) / s
  dv3d[i * 3] = -gx * fx / (z * s);
  dv3d[i * 3 + 1] = -gy * fy / (z * s);
  dv3d[i * 3 + 2] = (gx * fx * x + gy * fy * y) / (z * z * s);
}

// ---- K = 1 nearest neighbour (pytorch3d knn_points K=1 as used by the contact terms) ----
__global__ __launch_bounds__(256) void knn1_kernel(const float* __restrict__ q, int Nq, const float* __restrict__ t,
                                                  int Nt, float* __restrict__ d2, int* __restrict__ idx) {
  __shared__ float st[256 * 3];
  const int b = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  float qx = 0, qy = 0, qz = 0;
  if (i < Nq) {
    const float* p = q + ((long)b * Nq + i) * 3;
    qx = p[0]; qy = p[1]; qz = p[2];
  }
  float best = 3.0e38f;
  int bi = 0;
  for (int c0 = 0; c0 < Nt; c0 += 256) {
    const int n = min(256, Nt - c0);
    __syncthreads();
    for (int e = threadIdx.x; e < n * 3; e += 256) st[e] = t[((long)b * Nt + c0) * 3 + e];
    __syncthreads();
    for (int k = 0; k < n; ++k) {
      const float dx = qx - st[k * 3], dy = qy - st[k * 3 + 1], dz = qz - st[k * 3 + 2];
      const float d = dx * dx + dy * dy + dz * dz;
      if (d < best) { best = d; bi = c0 + k; }
    }
  }
  if (i < Nq) {
    d2[(long)b * Nq + i] = best;
    idx[(long)b * Nq + i] = bi;
  }
}

__global__ void knn1_bwd_kernel(const float* __restrict__ q, int Nq, const float* __restrict__ t, int Nt,
                                const int* __restrict__ idx, const float* __restrict__ g, long n,
                                float* __restrict__ dq, float* __restrict__ dt) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const long b = i / Nq;
  const float* tp = t + (b * Nt + idx[i]) * 3;
  float* dtp = dt + (b * Nt + idx[i]) * 3;
  for (int k = 0; k < 3; ++k) {
    const float v = 2.0f * g[i] * (q[i * 3 + k] - tp[k]);
    dq[i * 3 + k] = v;
    atomicAdd(dtp + k, -v);
  }
}

inline int ok() { return hipGetLastError() == hipSuccess ? HOLD_OK : HOLD_E_LAUNCH; }

}  // namespace

// number of faces contributing to a pixel (inside, or within the blur radius), maximum over the frame batch: the check
// behind pytorch3d's faces_per_pixel = 100 cap (fitting/utils.py:107) -- while no pixel sees more than K faces the cap
// is inactive and the product over ALL faces (what silhouette_kernel computes) is what the capped rasteriser returns
__global__ __launch_bounds__(256) void silhouette_count_kernel(const FaceSetup* __restrict__ fs, int F, int H, int W,
                                                               float blur, int* __restrict__ max_count) {
  const int b = blockIdx.z;
  const int tx = threadIdx.x % TILE, ty = threadIdx.x / TILE;
  const int j = blockIdx.x * TILE + tx, i = blockIdx.y * TILE + ty;
  const float s = 0.5f * (float)(H < W ? H : W);
  const float px = -((j + 0.5f) - 0.5f * W) / s, py = -((i + 0.5f) - 0.5f * H) / s;
  int cnt = 0;
  if (i < H && j < W) {
    const FaceSetup* fb = fs + (long)b * F;
    for (int f = 0; f < F; ++f) {
      const FaceSetup cur = fb[f];
      if (cur.pad != 0.f) cnt = HOLD_SILHOUETTE_STRADDLE;  // see face_setup_kernel
      if (cur.valid == 0.f || cur.xmax < px || cur.xmin > px || cur.ymax < py || cur.ymin > py) continue;
      float d, t;
      int e;
      if (face_dist(cur, px, py, blur, d, e, t) && cnt < HOLD_SILHOUETTE_STRADDLE) ++cnt;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cnt = max(cnt, __shfl_down(cnt, o, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(max_count, cnt);
}

extern "C" int64_t hold_silhouette_workspace_floats(int32_t B, int32_t F) { return (int64_t)B * F * 12; }

extern "C" int hold_silhouette_max_faces(const float* v3d_c, int32_t B, int32_t V, const int32_t* faces, int32_t F, float fx,
                                         float fy, float cx, float cy, int32_t H, int32_t W, float blur_radius,
                                         float* workspace, int32_t* max_count, hold_stream_t st) {
  if (!v3d_c || !faces || !workspace || !max_count || B <= 0 || V <= 0 || F <= 0 || H <= 0 || W <= 0) return HOLD_E_ARG;
  hipStream_t s = (hipStream_t)st;
  FaceSetup* fs = reinterpret_cast<FaceSetup*>(workspace);
  hipLaunchKernelGGL(face_setup_kernel, dim3((unsigned)(((long)B * F + 255) / 256)), dim3(256), 0, s, v3d_c, V, faces, F,
                     B, fx, fy, cx, cy, H, W, blur_radius, fs);
  if (hipMemsetAsync(max_count, 0, sizeof(int32_t), s) != hipSuccess) return HOLD_E_LAUNCH;
  hipLaunchKernelGGL(silhouette_count_kernel, dim3((W + TILE - 1) / TILE, (H + TILE - 1) / TILE, B), dim3(256), 0, s, fs, F,
                     H, W, blur_radius, max_count);
  return ok();
}

extern "C" int hold_silhouette_fwd(const float* v3d_c, int32_t B, int32_t V, const int32_t* faces, int32_t F, float fx,
                                   float fy, float cx, float cy, int32_t H, int32_t W, float sigma, float blur_radius,
                                   float* workspace, float* mask, hold_stream_t st) {
  if (!v3d_c || !faces || !workspace || !mask || B <= 0 || V <= 0 || F <= 0 || H <= 0 || W <= 0 || sigma <= 0.f)
    return HOLD_E_ARG;
  hipStream_t s = (hipStream_t)st;
  FaceSetup* fs = reinterpret_cast<FaceSetup*>(workspace);
  hipLaunchKernelGGL(face_setup_kernel, dim3((unsigned)(((long)B * F + 255) / 256)), dim3(256), 0, s, v3d_c, V, faces, F,
                     B, fx, fy, cx, cy, H, W, blur_radius, fs);
  hipLaunchKernelGGL((silhouette_kernel<false>), dim3((W + TILE - 1) / TILE, (H + TILE - 1) / TILE, B), dim3(256), 0, s,
                     fs, faces, F, H, W, sigma, blur_radius, mask, nullptr, nullptr, V);
  return ok();
}

extern "C" int hold_silhouette_bwd(const float* v3d_c, int32_t B, int32_t V, const int32_t* faces, int32_t F, float fx,
                                   float fy, float cx, float cy, int32_t H, int32_t W, float sigma, float blur_radius,
                                   float* workspace, const float* d_mask, float* d_ndc_scratch, float* d_v3d_c,
                                   hold_stream_t st) {
  if (!v3d_c || !faces || !workspace || !d_mask || !d_ndc_scratch || !d_v3d_c || B <= 0 || V <= 0 || F <= 0)
    return HOLD_E_ARG;
  hipStream_t s = (hipStream_t)st;
  FaceSetup* fs = reinterpret_cast<FaceSetup*>(workspace);
  hipLaunchKernelGGL(face_setup_kernel, dim3((unsigned)(((long)B * F + 255) / 256)), dim3(256), 0, s, v3d_c, V, faces, F,
                     B, fx, fy, cx, cy, H, W, blur_radius, fs);
  if (hipMemsetAsync(d_ndc_scratch, 0, (size_t)B * V * 2 * sizeof(float), s) != hipSuccess) return HOLD_E_LAUNCH;
  hipLaunchKernelGGL((silhouette_kernel<true>), dim3((W + TILE - 1) / TILE, (H + TILE - 1) / TILE, B), dim3(256), 0, s, fs,
                     faces, F, H, W, sigma, blur_radius, nullptr, d_mask, d_ndc_scratch, V);
  hipLaunchKernelGGL(ndc_bwd_kernel, dim3((unsigned)(((long)B * V + 255) / 256)), dim3(256), 0, s, v3d_c, d_ndc_scratch,
                     (long)B * V, fx, fy, H, W, d_v3d_c);
  return ok();
}

extern "C" int hold_knn1_fwd(const float* q, int32_t B, int32_t Nq, const float* t, int32_t Nt, float* d2, int32_t* idx,
                             hold_stream_t st) {
  if (!q || !t || !d2 || !idx || B <= 0 || Nq <= 0 || Nt <= 0) return HOLD_E_ARG;
  hipLaunchKernelGGL(knn1_kernel, dim3((Nq + 255) / 256, B), dim3(256), 0, (hipStream_t)st, q, Nq, t, Nt, d2, idx);
  return ok();
}

extern "C" int hold_knn1_bwd(const float* q, int32_t B, int32_t Nq, const float* t, int32_t Nt, const int32_t* idx,
                             const float* g, float* dq, float* dt_accum, hold_stream_t st) {
  if (!q || !t || !idx || !g || !dq || !dt_accum) return HOLD_E_ARG;
  const long n = (long)B * Nq;
  hipLaunchKernelGGL(knn1_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)st, q, Nq, t, Nt, idx,
                     g, n, dq, dt_accum);
  return ok();
}
