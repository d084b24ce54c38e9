// Canonical meshing on the GPU: iso-surface extraction from a dense SDF grid, replacing the CPU
// path of code/src/utils/meshing.py:9-72 (MISE octree refinement in code/src/libmise/mise.pyx + skimage's Lewiner
// marching cubes).  The grid values come from one batched ImplicitNet query (hold_fused_sdf); this file turns the
// (n x n x n) samples into an indexed, welded, consistently oriented triangle mesh without leaving the device.
//
// Method: marching tetrahedra on the Kuhn (6 tetrahedra per cube, all sharing the (0,0,0)-(1,1,1) diagonal)
// decomposition.  Every tetrahedron edge is one of 7 lattice directions {x, y, z, xy, yz, xz, xyz} owned by its
// lower grid point, so a surface vertex has the global id 7 * point + direction: vertices are welded by construction
// (no sort / hash), the surface is watertight wherever it does not leave the grid, and the face diagonals of
// neighbouring cubes agree.  Triangle orientation comes from a host-generated case table (6 tets x 16 sign cases)
// whose winding was fixed on a prototype; it cannot flip within a sign case.
//   pass 1  mt_classify : per grid point, flag its 7 owned edges that cross the level set; per cube, count triangles
//   (host)  exclusive scans of the flags and of the counts (torch.cumsum)
//   pass 2  mt_vertices : one vertex per flagged edge, linear interpolation, written in world coordinates
//   pass 3  mt_triangles: per cube, emit its triangles as vertex indices through the edge scan
// HBM-bound and tiny next to the grid query: 129^3 points = 8.6 MB of SDF, ~60 MB of flags.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hold_hip.h"

namespace {

// lattice directions of the 7 owned edges, as corner-bit offsets (bit0 = x, bit1 = y, bit2 = z)
__device__ __constant__ int kDirBits[7] = {1, 2, 4, 3, 6, 5, 7};

struct Grid {
  int n;            // points per axis
  float level;
  float ox, oy, oz; // world position of grid point (0,0,0)
  float h;          // world spacing
};

__device__ __forceinline__ long gidx(int n, int ix, int iy, int iz) { return ((long)ix * n + iy) * n + iz; }

__global__ __launch_bounds__(256) void mt_classify(const float* __restrict__ sdf, Grid g,
                                                   const int8_t* __restrict__ ntri_tab /*[6][16]*/,
                                                   const int8_t* __restrict__ tet_corner /*[6][4]*/,
                                                   int32_t* __restrict__ edge_flag, int32_t* __restrict__ cube_ntri) {
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  const long np = (long)g.n * g.n * g.n;
  if (p >= np) return;
  const int iz = (int)(p % g.n), iy = (int)((p / g.n) % g.n), ix = (int)(p / ((long)g.n * g.n));
  const float v0 = sdf[p];
  const bool in0 = v0 < g.level;
#pragma unroll
  for (int d = 0; d < 7; ++d) {
    const int b = kDirBits[d];
    const int jx = ix + (b & 1), jy = iy + ((b >> 1) & 1), jz = iz + ((b >> 2) & 1);
    int f = 0;
    if (jx < g.n && jy < g.n && jz < g.n) f = (in0 != (sdf[gidx(g.n, jx, jy, jz)] < g.level)) ? 1 : 0;
    edge_flag[p * 7 + d] = f;
  }
  int cnt = 0;
  if (ix + 1 < g.n && iy + 1 < g.n && iz + 1 < g.n) {
    int mask = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c)
      mask |= (sdf[gidx(g.n, ix + (c & 1), iy + ((c >> 1) & 1), iz + ((c >> 2) & 1))] < g.level ? 1 : 0) << c;
    if (mask != 0 && mask != 255) {
      for (int t = 0; t < 6; ++t) {
        int m = 0;
        for (int k = 0; k < 4; ++k) m |= ((mask >> tet_corner[t * 4 + k]) & 1) << k;
        cnt += ntri_tab[t * 16 + m];
      }
    }
  }
  cube_ntri[p] = cnt;  // indexed by the cube's origin point (zero on the far faces)
}

__global__ __launch_bounds__(256) void mt_vertices(const float* __restrict__ sdf, Grid g,
                                                   const int32_t* __restrict__ edge_flag,
                                                   const int64_t* __restrict__ edge_scan, float* __restrict__ verts) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  const long ne = (long)g.n * g.n * g.n * 7;
  if (e >= ne || !edge_flag[e]) return;
  const long p = e / 7;
  const int d = (int)(e % 7), b = kDirBits[d];
  const int iz = (int)(p % g.n), iy = (int)((p / g.n) % g.n), ix = (int)(p / ((long)g.n * g.n));
  const float v0 = sdf[p], v1 = sdf[gidx(g.n, ix + (b & 1), iy + ((b >> 1) & 1), iz + ((b >> 2) & 1))];
  const float t = (g.level - v0) / (v1 - v0);
  float* o = verts + edge_scan[e] * 3;
  o[0] = g.ox + g.h * ((float)ix + t * (float)(b & 1));
  o[1] = g.oy + g.h * ((float)iy + t * (float)((b >> 1) & 1));
  o[2] = g.oz + g.h * ((float)iz + t * (float)((b >> 2) & 1));
}

__global__ __launch_bounds__(256) void mt_triangles(const float* __restrict__ sdf, Grid g,
                                                    const int8_t* __restrict__ ntri_tab, const int8_t* __restrict__ tet_corner,
                                                    const int8_t* __restrict__ tri_tab /*[6][16][2][3][2] corner pairs*/,
                                                    const int32_t* __restrict__ cube_ntri,
                                                    const int64_t* __restrict__ cube_scan,
                                                    const int64_t* __restrict__ edge_scan, int64_t* __restrict__ faces) {
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  const long np = (long)g.n * g.n * g.n;
  if (p >= np || cube_ntri[p] == 0) return;
  const int iz = (int)(p % g.n), iy = (int)((p / g.n) % g.n), ix = (int)(p / ((long)g.n * g.n));
  int mask = 0;
#pragma unroll
  for (int c = 0; c < 8; ++c)
    mask |= (sdf[gidx(g.n, ix + (c & 1), iy + ((c >> 1) & 1), iz + ((c >> 2) & 1))] < g.level ? 1 : 0) << c;
  int64_t* out = faces + cube_scan[p] * 3;
  for (int t = 0; t < 6; ++t) {
    int m = 0;
    for (int k = 0; k < 4; ++k) m |= ((mask >> tet_corner[t * 4 + k]) & 1) << k;
    const int nt = ntri_tab[t * 16 + m];
    for (int j = 0; j < nt; ++j) {
      for (int v = 0; v < 3; ++v) {
        const int8_t* pr = tri_tab + ((((t * 16 + m) * 2 + j) * 3 + v) * 2);
        const int ca = pr[0], cb = pr[1];  // cube corners, ca a bit-subset of cb
        const long owner = gidx(g.n, ix + (ca & 1), iy + ((ca >> 1) & 1), iz + ((ca >> 2) & 1));
        const int bits = cb ^ ca;
        int d = 0;
#pragma unroll
        for (int q = 0; q < 7; ++q) d = (kDirBits[q] == bits) ? q : d;
        out[v] = edge_scan[owner * 7 + d];
      }
      out += 3;
    }
  }
}

inline Grid make_grid(int n, float level, const float* origin, float h) {
  Grid g;
  g.n = n; g.level = level; g.ox = origin[0]; g.oy = origin[1]; g.oz = origin[2]; g.h = h;
  return g;
}

}  // namespace

extern "C" int hold_mt_classify(const float* sdf, int32_t n, float level, const int8_t* ntri_tab,
                                const int8_t* tet_corner, int32_t* edge_flag, int32_t* cube_ntri, hold_stream_t st) {
  if (!sdf || !ntri_tab || !tet_corner || !edge_flag || !cube_ntri || n < 2 || n > 1024) return HOLD_E_ARG;
  const float o[3] = {0, 0, 0};
  const long np = (long)n * n * n;
  hipLaunchKernelGGL(mt_classify, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, (hipStream_t)st, sdf,
                     make_grid(n, level, o, 1.f), ntri_tab, tet_corner, edge_flag, cube_ntri);
  return hipGetLastError() == hipSuccess ? HOLD_OK : HOLD_E_LAUNCH;
}

extern "C" int hold_mt_vertices(const float* sdf, int32_t n, float level, float ox, float oy, float oz, float h,
                                const int32_t* edge_flag, const int64_t* edge_scan, float* verts, hold_stream_t st) {
  if (!sdf || !edge_flag || !edge_scan || !verts || n < 2 || n > 1024) return HOLD_E_ARG;
  const float o[3] = {ox, oy, oz};
  const long ne = (long)n * n * n * 7;
  hipLaunchKernelGGL(mt_vertices, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, (hipStream_t)st, sdf,
                     make_grid(n, level, o, h), edge_flag, edge_scan, verts);
  return hipGetLastError() == hipSuccess ? HOLD_OK : HOLD_E_LAUNCH;
}

extern "C" int hold_mt_triangles(const float* sdf, int32_t n, float level, const int8_t* ntri_tab,
                                 const int8_t* tet_corner, const int8_t* tri_tab, const int32_t* cube_ntri,
                                 const int64_t* cube_scan, const int64_t* edge_scan, int64_t* faces, hold_stream_t st) {
  if (!sdf || !ntri_tab || !tet_corner || !tri_tab || !cube_ntri || !cube_scan || !edge_scan || !faces || n < 2 || n > 1024)
    return HOLD_E_ARG;
  const float o[3] = {0, 0, 0};
  const long np = (long)n * n * n;
  hipLaunchKernelGGL(mt_triangles, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, (hipStream_t)st, sdf,
                     make_grid(n, level, o, 1.f), ntri_tab, tet_corner, tri_tab, cube_ntri, cube_scan, edge_scan, faces);
  return hipGetLastError() == hipSuccess ? HOLD_OK : HOLD_E_LAUNCH;
}
