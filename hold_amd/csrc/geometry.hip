// Training loss-target geometry without kaolin: signed distance of query points to a closed triangle
// mesh, replacing kaolin.metrics.trianglemesh.point_to_mesh_distance + kaolin.ops.mesh.check_sign as used by
// compute_mano_cano_sdf / check_off_in_surface_points_cano_mesh (code/src/engine/volsdf_utils.py:172-217) on the sealed,
// once-subdivided canonical MANO (3 110 vertices, 6 216 faces; mano_node.py:126-135).
//   unsigned distance: min over faces of the distance to the closest point of the triangle (Ericson RTCD 5.1.5)
//   sign: inside <=> |sum of signed solid angles| > 2 pi (generalised winding number; Van Oosterom & Strackee)
// One thread per point, the frame's triangles staged through LDS 128 at a time (9 floats each, gathered through the
// index list once per block): VALU-bound, ~110 flop + 1 atan2 + 3 sqrt per (point, face).  Points farther than `cull`
// from the mesh's bounding box (when cull > 0) skip the face loop and return that distance: the off-surface test of the
// reference only needs min distance > 0.01, and most ray samples are far from the hand.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hold_hip.h"

namespace {

constexpr int TILE = 128;

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ V3 madd(V3 a, V3 d, float t) { return {a.x + d.x * t, a.y + d.y * t, a.z + d.z * t}; }

__device__ __forceinline__ float tri_d2(V3 p, V3 a, V3 b, V3 c) {
  const V3 ab = sub(b, a), ac = sub(c, a), ap = sub(p, a);
  const float d1 = dot(ab, ap), d2 = dot(ac, ap);
  V3 q;
  if (d1 <= 0.f && d2 <= 0.f) {
    q = a;
  } else {
    const V3 bp = sub(p, b);
    const float d3 = dot(ab, bp), d4 = dot(ac, bp);
    const float vc = d1 * d4 - d3 * d2;
    if (d3 >= 0.f && d4 <= d3) {
      q = b;
    } else if (vc <= 0.f && d1 >= 0.f && d3 <= 0.f) {
      q = madd(a, ab, d1 / fmaxf(d1 - d3, 1e-30f));
    } else {
      const V3 cp = sub(p, c);
      const float d5 = dot(ab, cp), d6 = dot(ac, cp);
      const float vb = d5 * d2 - d1 * d6, va = d3 * d6 - d5 * d4;
      if (d6 >= 0.f && d5 <= d6) {
        q = c;
      } else if (vb <= 0.f && d2 >= 0.f && d6 <= 0.f) {
        q = madd(a, ac, d2 / fmaxf(d2 - d6, 1e-30f));
      } else if (va <= 0.f && (d4 - d3) >= 0.f && (d5 - d6) >= 0.f) {
        q = madd(b, sub(c, b), (d4 - d3) / fmaxf((d4 - d3) + (d5 - d6), 1e-30f));
      } else {
        float den = va + vb + vc;
        den = fabsf(den) < 1e-30f ? 1e-30f : den;
        q = madd(madd(a, ab, vb / den), ac, vc / den);
      }
    }
  }
  const V3 r = sub(p, q);
  return dot(r, r);
}

__device__ __forceinline__ float solid_angle(V3 p, V3 a, V3 b, V3 c) {
  const V3 ra = sub(a, p), rb = sub(b, p), rc = sub(c, p);
  const float la = sqrtf(dot(ra, ra)), lb = sqrtf(dot(rb, rb)), lc = sqrtf(dot(rc, rc));
  const float num = dot(ra, cross(rb, rc));
  const float den = la * lb * lc + dot(ra, rb) * lc + dot(rb, rc) * la + dot(rc, ra) * lb;
  return 2.0f * atan2f(num, den);
}

__global__ __launch_bounds__(256) void mesh_sdf_kernel(const float* __restrict__ pts, long P,
                                                       const float* __restrict__ verts, long vstride, int V,
                                                       const int* __restrict__ faces, int F, float cull,
                                                       const float* __restrict__ aabb, float* __restrict__ sd) {
  __shared__ float tri[TILE * 9];
  const int b = blockIdx.y;
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  const float* vb = verts + (long)b * vstride;
  bool active = p < P;
  V3 x = {0.f, 0.f, 0.f};
  float far_d = 0.f;
  if (active) {
    const float* q = pts + ((long)b * P + p) * 3;
    x = {q[0], q[1], q[2]};
    if (cull > 0.f && aabb) {
      const float* bb = aabb + b * 6;
      const float dx = fmaxf(fmaxf(bb[0] - x.x, x.x - bb[3]), 0.f), dy = fmaxf(fmaxf(bb[1] - x.y, x.y - bb[4]), 0.f),
                  dz = fmaxf(fmaxf(bb[2] - x.z, x.z - bb[5]), 0.f);
      far_d = sqrtf(dx * dx + dy * dy + dz * dz);
      if (far_d > cull) active = false;  // outside, at least far_d away
    }
  }
  float best = 3.0e38f, omega = 0.f;
  for (int f0 = 0; f0 < F; f0 += TILE) {
    const int nt = min(TILE, F - f0);
    for (int e = threadIdx.x; e < nt * 3; e += 256) {
      const int vid = faces[(long)(f0 + e / 3) * 3 + (e % 3)];
      const float* v = vb + (long)(vid < V ? (vid < 0 ? 0 : vid) : V - 1) * 3;
      float* t = tri + (e / 3) * 9 + (e % 3) * 3;
      t[0] = v[0];
      t[1] = v[1];
      t[2] = v[2];
    }
    __syncthreads();
    if (active) {
      for (int j = 0; j < nt; ++j) {
        const float* t = tri + j * 9;  // uniform address: LDS broadcast
        const V3 a = {t[0], t[1], t[2]}, bb = {t[3], t[4], t[5]}, c = {t[6], t[7], t[8]};
        best = fminf(best, tri_d2(x, a, bb, c));
        omega += solid_angle(x, a, bb, c);
      }
    }
    __syncthreads();
  }
  if (p < P) {
    float r;
    if (active) {
      r = sqrtf(best) * (fabsf(omega) > 6.2831853f ? -1.f : 1.f);
    } else {
      r = far_d;
    }
    sd[(long)b * P + p] = r;
  }
}

// Few query points (the 10 x 307 samples of compute_mano_cano_sdf in a training step): one thread per point would put
// 20 blocks on 256 CUs.  Here a wavefront owns WPP points and its 64 lanes share the faces of a tile (lane j takes
// triangles j, j + 64, ...), min / sum reduced by a fixed butterfly: the same minimum, the winding sum in another
// (deterministic) order.
constexpr int WPP = 2;
__global__ __launch_bounds__(256) void mesh_sdf_wave_kernel(const float* __restrict__ pts, long P,
                                                            const float* __restrict__ verts, long vstride, int V,
                                                            const int* __restrict__ faces, int F, float cull,
                                                            const float* __restrict__ aabb, float* __restrict__ sd) {
  __shared__ float tri[TILE * 9];
  const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* vb = verts + (long)b * vstride;
  V3 x[WPP];
  float far_d[WPP], best[WPP], omega[WPP];
  bool active[WPP];
#pragma unroll
  for (int i = 0; i < WPP; ++i) {
    const long p = ((long)blockIdx.x * 4 + wave) * WPP + i;
    active[i] = p < P;
    x[i] = {0.f, 0.f, 0.f};
    far_d[i] = 0.f;
    best[i] = 3.0e38f;
    omega[i] = 0.f;
    if (active[i]) {
      const float* q = pts + ((long)b * P + p) * 3;
      x[i] = {q[0], q[1], q[2]};
      if (cull > 0.f && aabb) {
        const float* bb = aabb + b * 6;
        const float dx = fmaxf(fmaxf(bb[0] - x[i].x, x[i].x - bb[3]), 0.f),
                    dy = fmaxf(fmaxf(bb[1] - x[i].y, x[i].y - bb[4]), 0.f),
                    dz = fmaxf(fmaxf(bb[2] - x[i].z, x[i].z - bb[5]), 0.f);
        far_d[i] = sqrtf(dx * dx + dy * dy + dz * dz);
        if (far_d[i] > cull) active[i] = false;
      }
    }
  }
  for (int f0 = 0; f0 < F; f0 += TILE) {
    const int nt = min(TILE, F - f0);
    for (int e = threadIdx.x; e < nt * 3; e += 256) {
      const int vid = faces[(long)(f0 + e / 3) * 3 + (e % 3)];
      const float* v = vb + (long)(vid < V ? (vid < 0 ? 0 : vid) : V - 1) * 3;
      float* t = tri + (e / 3) * 9 + (e % 3) * 3;
      t[0] = v[0];
      t[1] = v[1];
      t[2] = v[2];
    }
    __syncthreads();
    for (int j = lane; j < nt; j += 64) {
      const float* t = tri + j * 9;  // stride 9 floats: conflict-free over the 32 banks
      const V3 a = {t[0], t[1], t[2]}, bb = {t[3], t[4], t[5]}, c = {t[6], t[7], t[8]};
#pragma unroll
      for (int i = 0; i < WPP; ++i)
        if (active[i]) {
          best[i] = fminf(best[i], tri_d2(x[i], a, bb, c));
          omega[i] += solid_angle(x[i], a, bb, c);
        }
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < WPP; ++i) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
      best[i] = fminf(best[i], __shfl_xor(best[i], d));
      omega[i] += __shfl_xor(omega[i], d);
    }
    const long p = ((long)blockIdx.x * 4 + wave) * WPP + i;
    if (lane == 0 && p < P)
      sd[(long)b * P + p] = active[i] ? sqrtf(best[i]) * (fabsf(omega[i]) > 6.2831853f ? -1.f : 1.f) : far_d[i];
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Per-ray off-surface test of check_off_in_surface_points_cano_mesh (volsdf_utils.py:189-217) without visiting every
// face for every sample: off(ray) <=> min over the ray's samples of the signed distance > thr.  Built on two structures
// made once per canonical-mesh update (hold_amd/geometry.py:MeshIndex):
//   node_sdf [G^3]  exact signed distances at the nodes of a uniform grid (spacing h, h*sqrt(3) < thr) around the mesh
//   cell lists      for every grid cell, the triangles whose thr-dilated bounding box touches it (CSR)
// For a sample x with nearest node g, a = |x - g|, v = sd(g), the 1-Lipschitz bound sd(x) in [v - a, v + a] decides most
// samples: v - a > thr -> farther than thr outside; v + a <= thr -> sd <= thr.  Only samples in the band in between look
// at their cell's triangles (exact closest-point distance, early exit at d <= thr); if none is within thr, |sd(x)| > thr
// > h*sqrt(3) >= 2a, so the segment x-g cannot cross the surface and sign(sd(x)) = sign(v).  The decisions are those of
// the brute-force kernel above (same distance routine), at a few hundred point-triangle tests per ray instead of
// samples x faces.  One wavefront per ray: lanes = samples, a ballot ends the ray at the first sample with sd <= thr.
struct MeshGrid {
  const float* node_sdf;
  int G;
  float ox, oy, oz, h, thr;
  const int* cell_start;
  const int* cell_tris;
  const float* verts;
  const int* faces;
};

__global__ __launch_bounds__(256) void ray_off_surface_kernel(const float* __restrict__ xc, int ldx, long n_rays, int S,
                                                              MeshGrid m, uint8_t* __restrict__ off) {
  const int lane = threadIdx.x & 63;
  const long ray = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= n_rays) return;
  const float inv_h = 1.0f / m.h;
  const float hi = (float)(m.G - 1);
  bool any_in = false;
  for (int s0 = 0; s0 < S && !any_in; s0 += 64) {
    const int s = s0 + lane;
    int state = 0;  // 0 = farther than thr outside, 1 = sd <= thr, 2 = undecided
    float px = 0.f, py = 0.f, pz = 0.f, v = 0.f;
    int cx = 0, cy = 0, cz = 0;
    if (s < S) {
      const float* q = xc + (ray * S + s) * (long)ldx;
      px = q[0]; py = q[1]; pz = q[2];
      const float gx = (px - m.ox) * inv_h, gy = (py - m.oy) * inv_h, gz = (pz - m.oz) * inv_h;
      if (gx >= 0.f && gy >= 0.f && gz >= 0.f && gx <= hi && gy <= hi && gz <= hi) {
        const int ix = (int)(gx + 0.5f), iy = (int)(gy + 0.5f), iz = (int)(gz + 0.5f);
        v = m.node_sdf[((long)ix * m.G + iy) * m.G + iz];
        const float ax = (gx - (float)ix) * m.h, ay = (gy - (float)iy) * m.h, az = (gz - (float)iz) * m.h;
        const float a = sqrtf(ax * ax + ay * ay + az * az);
        state = (v - a > m.thr) ? 0 : ((v + a <= m.thr) ? 1 : 2);
        cx = min((int)gx, m.G - 2); cy = min((int)gy, m.G - 2); cz = min((int)gz, m.G - 2);
      }
    }
    if (__ballot(state == 1)) { any_in = true; break; }
    if (state == 2) {
      const long cell = ((long)cx * (m.G - 1) + cy) * (m.G - 1) + cz;
      const int b = m.cell_start[cell], e = m.cell_start[cell + 1];
      const V3 x = {px, py, pz};
      const float thr2 = m.thr * m.thr;
      bool within = false;
      for (int k = b; k < e; ++k) {
        const int* f = m.faces + (long)m.cell_tris[k] * 3;
        const float *pa = m.verts + (long)f[0] * 3, *pb = m.verts + (long)f[1] * 3, *pc = m.verts + (long)f[2] * 3;
        const V3 a3 = {pa[0], pa[1], pa[2]}, b3 = {pb[0], pb[1], pb[2]}, c3 = {pc[0], pc[1], pc[2]};
        if (tri_d2(x, a3, b3, c3) <= thr2) { within = true; break; }
      }
      state = (within || v < 0.f) ? 1 : 0;
    }
    if (__ballot(state == 1)) any_in = true;
  }
  if (lane == 0) off[ray] = any_in ? 0 : 1;
}

}  // namespace

extern "C" int hold_ray_off_surface(const float* xc, int32_t ldx, int64_t n_rays, int32_t S, const float* node_sdf, int32_t G,
                                    float ox, float oy, float oz, float h, float thr, const int32_t* cell_start,
                                    const int32_t* cell_tris, const float* verts, const int32_t* faces, uint8_t* off,
                                    hold_stream_t st) {
  if (!xc || !node_sdf || !cell_start || !cell_tris || !verts || !faces || !off || ldx < 3 || S <= 0 || G < 2 || h <= 0.f ||
      thr <= 0.f || n_rays < 0)
    return HOLD_E_ARG;
  if (!(h * 1.7320508f < thr)) return HOLD_E_ARG;  // the sign inference needs h * sqrt(3) < thr
  if (n_rays == 0) return HOLD_OK;
  MeshGrid m = {node_sdf, G, ox, oy, oz, h, thr, cell_start, cell_tris, verts, faces};
  hipLaunchKernelGGL(ray_off_surface_kernel, dim3((unsigned)((n_rays + 3) / 4)), dim3(256), 0, (hipStream_t)st, xc, ldx,
                     (long)n_rays, S, m, off);
  return hipGetLastError() == hipSuccess ? HOLD_OK : HOLD_E_LAUNCH;
}

extern "C" int hold_mesh_sdf(const float* pts, int32_t B, int64_t P, const float* verts, int32_t verts_shared, int32_t V,
                             const int32_t* faces, int32_t F, float cull_dist, const float* aabb, float* sd,
                             hold_stream_t st) {
  if (!pts || !verts || !faces || !sd || B < 0 || P < 0 || V <= 0 || F <= 0) return HOLD_E_ARG;
  if (cull_dist > 0.f && !aabb) return HOLD_E_ARG;
  if (B == 0 || P == 0) return HOLD_OK;
  if ((long)B * P <= 32768) {  // too few points to fill the chip one thread per point: lanes over faces instead
    const dim3 grid((unsigned)((P + 4 * WPP - 1) / (4 * WPP)), (unsigned)B);
    hipLaunchKernelGGL(mesh_sdf_wave_kernel, grid, dim3(256), 0, (hipStream_t)st, pts, (long)P, verts,
                       verts_shared ? 0L : (long)V * 3, V, faces, F, cull_dist, aabb, sd);
    return hipGetLastError() == hipSuccess ? HOLD_OK : HOLD_E_LAUNCH;
  }
  const dim3 grid((unsigned)((P + 255) / 256), (unsigned)B);
  hipLaunchKernelGGL(mesh_sdf_kernel, grid, dim3(256), 0, (hipStream_t)st, pts, (long)P, verts,
                     verts_shared ? 0L : (long)V * 3, V, faces, F, cull_dist, aabb, sd);
  return hipGetLastError() == hipSuccess ? HOLD_OK : HOLD_E_LAUNCH;
}
