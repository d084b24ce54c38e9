// MANO forward LBS + HOLD's server post-processing, forward and backward, one workgroup per frame (gfx950).
// Reference: lbs() code/src/utils/external/lbs.py:139-251 (blend_shapes :274-295, vertices2joints :254-271,
// batch_rodrigues :298-330 incl. norm(r + 1e-8), batch_rigid_transform :345-399), MANO.forward
// code/src/utils/external/body_models.py:601-685 (pose += pose_mean, fingertip joints), GenericServer.forward
// code/src/model/mano/server.py:62-99 (scene scale / translation, relative-to-canonical tfs . tfs_c_inv).
//
// ~1.3 MFLOP per frame: latency-bound; the point of the kernel is ONE launch instead of ~60 tiny ones (300-600
// times per batch in the pose-refinement loop) and keeping vertices / transforms on chip in between.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hold_hip.h"

namespace {

constexpr int NV = 778, NJ = 16, NPF = 135, NB = 10;

struct ManoConst {
  const float* v_template;   // [778][3]
  const float* shapedirs;    // [778][3][10]
  const float* posedirs;     // [135][2334]
  const float* J_regressor;  // [16][778]
  const int* parents;        // [16], parents[0] < 0
  const float* lbs_weights;  // [778][16]
  const float* pose_mean;    // [48]
  const float* tfs_c_inv;    // [16][16] or null (absolute)
};

struct Sh {
  float vsh[NV * 3];     // v_shaped, later d v_shaped
  float vpo[NV * 3];     // v_posed, later d v_posed
  float J[NJ * 3];
  float R[NJ * 9];       // local rotations
  float rel[NJ * 3];
  float Gr[NJ * 9];      // world rotations
  float Gt[NJ * 3];      // world translations
  float A[NJ * 12];      // relative transforms (3x4)
  float pf[NPF + 1];
  float pose[48];
  float beta[NB];
  float red[256];
  // backward scratch
  float dA[NJ * 12], dGr[NJ * 9], dGt[NJ * 3], dJ[NJ * 3], dR[NJ * 9], dpf[NPF + 1];
};

__device__ void rodrigues(const float* r, float* R) {
  const float x = r[0] + 1e-8f, y = r[1] + 1e-8f, z = r[2] + 1e-8f;
  const float a = sqrtf(x * x + y * y + z * z);
  const float dx = r[0] / a, dy = r[1] / a, dz = r[2] / a;
  const float s = sinf(a), c = cosf(a);
  const float K[9] = {0.f, -dz, dy, dz, 0.f, -dx, -dy, dx, 0.f};
  float KK[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) KK[i * 3 + j] = K[i * 3] * K[j] + K[i * 3 + 1] * K[3 + j] + K[i * 3 + 2] * K[6 + j];
  for (int i = 0; i < 9; ++i) R[i] = ((i % 4 == 0) ? 1.f : 0.f) + s * K[i] + (1.f - c) * KK[i];
}

// gradient of R = I + sin(a) K + (1-cos a) K K w.r.t. the axis-angle vector r
__device__ void rodrigues_bwd(const float* r, const float* dR, float* dr) {
  const float ex = r[0] + 1e-8f, ey = r[1] + 1e-8f, ez = r[2] + 1e-8f;
  const float a = sqrtf(ex * ex + ey * ey + ez * ez);
  const float d[3] = {r[0] / a, r[1] / a, r[2] / a};
  const float s = sinf(a), c = cosf(a);
  const float K[9] = {0.f, -d[2], d[1], d[2], 0.f, -d[0], -d[1], d[0], 0.f};
  float KK[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) KK[i * 3 + j] = K[i * 3] * K[j] + K[i * 3 + 1] * K[3 + j] + K[i * 3 + 2] * K[6 + j];
  float ds = 0.f, dc = 0.f;
  for (int i = 0; i < 9; ++i) {
    ds += dR[i] * K[i];
    dc -= dR[i] * KK[i];
  }
  // dK = s dR + (1-c) (dR K^T + K^T dR)
  float dK[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      float t = 0.f;
      for (int k = 0; k < 3; ++k) t += dR[i * 3 + k] * K[j * 3 + k] + K[k * 3 + i] * dR[k * 3 + j];
      dK[i * 3 + j] = s * dR[i * 3 + j] + (1.f - c) * t;
    }
  const float dd[3] = {dK[7] - dK[5], dK[2] - dK[6], dK[3] - dK[1]};
  float da = ds * c - dc * s;
  da -= (dd[0] * r[0] + dd[1] * r[1] + dd[2] * r[2]) / (a * a);
  dr[0] = dd[0] / a + da * ex / a;
  dr[1] = dd[1] / a + da * ey / a;
  dr[2] = dd[2] / a + da * ez / a;
}

__device__ void forward_core(Sh& S, const ManoConst& m, const float* betas, const float* full_pose, int tid) {
  if (tid < 48) S.pose[tid] = full_pose[tid] + m.pose_mean[tid];
  if (tid < NB) S.beta[tid] = betas[tid];
  __syncthreads();
  for (int c = tid; c < NV * 3; c += 256) {
    float v = m.v_template[c];
    const float* sd = m.shapedirs + (long)c * NB;
#pragma unroll
    for (int l = 0; l < NB; ++l) v += S.beta[l] * sd[l];
    S.vsh[c] = v;
  }
  if (tid < NJ) rodrigues(S.pose + tid * 3, S.R + tid * 9);
  __syncthreads();
  if (tid < NJ * 3) {
    const int j = tid / 3, k = tid % 3;
    float acc = 0.f;
    for (int v = 0; v < NV; ++v) acc += m.J_regressor[j * NV + v] * S.vsh[v * 3 + k];
    S.J[tid] = acc;
  }
  if (tid >= 64 && tid < 64 + NPF) {
    const int q = tid - 64, j = 1 + q / 9, e = q % 9;
    S.pf[q] = S.R[j * 9 + e] - ((e % 4 == 0) ? 1.f : 0.f);
  }
  __syncthreads();
  for (int c = tid; c < NV * 3; c += 256) {
    float v = S.vsh[c];
    for (int q = 0; q < NPF; ++q) v += S.pf[q] * m.posedirs[(long)q * (NV * 3) + c];
    S.vpo[c] = v;
  }
  if (tid < NJ * 3) {
    const int j = tid / 3, k = tid % 3;
    const int p = m.parents[j];
    S.rel[tid] = (j == 0 || p < 0) ? S.J[tid] : S.J[tid] - S.J[p * 3 + k];
  }
  __syncthreads();
  if (tid == 0) {  // kinematic chain: 16 sequential 3x4 products
    for (int i = 0; i < 9; ++i) S.Gr[i] = S.R[i];
    for (int i = 0; i < 3; ++i) S.Gt[i] = S.rel[i];
    for (int j = 1; j < NJ; ++j) {
      const int p = m.parents[j];
      for (int a = 0; a < 3; ++a) {
        for (int b = 0; b < 3; ++b)
          S.Gr[j * 9 + a * 3 + b] = S.Gr[p * 9 + a * 3] * S.R[j * 9 + b] + S.Gr[p * 9 + a * 3 + 1] * S.R[j * 9 + 3 + b] +
                                    S.Gr[p * 9 + a * 3 + 2] * S.R[j * 9 + 6 + b];
        S.Gt[j * 3 + a] = S.Gr[p * 9 + a * 3] * S.rel[j * 3] + S.Gr[p * 9 + a * 3 + 1] * S.rel[j * 3 + 1] +
                          S.Gr[p * 9 + a * 3 + 2] * S.rel[j * 3 + 2] + S.Gt[p * 3 + a];
      }
    }
  }
  __syncthreads();
  if (tid < NJ * 3) {
    const int j = tid / 3, a = tid % 3;
    float t = S.Gt[tid];
    for (int b = 0; b < 3; ++b) {
      S.A[j * 12 + a * 4 + b] = S.Gr[j * 9 + a * 3 + b];
      t -= S.Gr[j * 9 + a * 3 + b] * S.J[j * 3 + b];
    }
    S.A[j * 12 + a * 4 + 3] = t;
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void mano_fwd_kernel(ManoConst m, const float* __restrict__ betas,
                                                      const float* __restrict__ full_pose,
                                                      const float* __restrict__ scale, const float* __restrict__ transl,
                                                      float* __restrict__ verts, float* __restrict__ jnts,
                                                      float* __restrict__ tfs, float* __restrict__ v_posed) {
  __shared__ Sh S;
  const int f = blockIdx.x, tid = threadIdx.x;
  forward_core(S, m, betas + f * NB, full_pose + f * 48, tid);
  const float s = scale[f], tx = transl[f * 3], ty = transl[f * 3 + 1], tz = transl[f * 3 + 2];
  const float tt[3] = {tx, ty, tz};
  for (int v = tid; v < NV; v += 256) {
    float T[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) T[e] = 0.f;
    for (int j = 0; j < NJ; ++j) {
      const float w = m.lbs_weights[v * NJ + j];
#pragma unroll
      for (int e = 0; e < 12; ++e) T[e] += w * S.A[j * 12 + e];
    }
    const float px = S.vpo[v * 3], py = S.vpo[v * 3 + 1], pz = S.vpo[v * 3 + 2];
    for (int a = 0; a < 3; ++a) {
      const float o = T[a * 4] * px + T[a * 4 + 1] * py + T[a * 4 + 2] * pz + T[a * 4 + 3];
      verts[((long)f * NV + v) * 3 + a] = o * s + tt[a] * s;
    }
    if (v_posed) {
      v_posed[((long)f * NV + v) * 3] = px;
      v_posed[((long)f * NV + v) * 3 + 1] = py;
      v_posed[((long)f * NV + v) * 3 + 2] = pz;
    }
  }
  if (jnts && tid < NJ * 3) jnts[(long)f * 63 + tid] = S.Gt[tid] * s + tt[tid % 3] * s;
  if (tid < NJ * 16) {
    const int j = tid / 16, a = (tid % 16) / 4, b = tid % 4;
    // tf = [s*A | s*A_t + s*t ; 0 0 0 1] . Cinv_j
    float o;
    if (a == 3) {
      o = m.tfs_c_inv ? m.tfs_c_inv[j * 16 + 12 + b] : (b == 3 ? 1.f : 0.f);
    } else {
      const float r0 = s * S.A[j * 12 + a * 4], r1 = s * S.A[j * 12 + a * 4 + 1], r2 = s * S.A[j * 12 + a * 4 + 2];
      const float r3 = s * S.A[j * 12 + a * 4 + 3] + s * tt[a];
      if (m.tfs_c_inv) {
        const float* C = m.tfs_c_inv + j * 16;
        o = r0 * C[b] + r1 * C[4 + b] + r2 * C[8 + b] + r3 * C[12 + b];
      } else {
        o = (b == 0) ? r0 : (b == 1) ? r1 : (b == 2) ? r2 : r3;
      }
    }
    tfs[(long)f * 256 + tid] = o;
  }
  if (jnts && tid >= 64 && tid < 64 + 15) {  // fingertip joints = posed vertices 744, 320, 443, 554, 671
    const int tips[5] = {744, 320, 443, 554, 671};
    const int q = (tid - 64) / 3, a = (tid - 64) % 3, v = tips[q];
    float o = 0.f;
    for (int j = 0; j < NJ; ++j) {
      const float w = m.lbs_weights[v * NJ + j];
      o += w * (S.A[j * 12 + a * 4] * S.vpo[v * 3] + S.A[j * 12 + a * 4 + 1] * S.vpo[v * 3 + 1] +
                S.A[j * 12 + a * 4 + 2] * S.vpo[v * 3 + 2] + S.A[j * 12 + a * 4 + 3]);
    }
    jnts[(long)f * 63 + 48 + (tid - 64)] = o * s + tt[a] * s;
  }
}

__device__ float block_sum(float v, float* red, int tid) {
  red[tid] = v;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) red[tid] += red[tid + s];
    __syncthreads();
  }
  const float r = red[0];
  __syncthreads();
  return r;
}

__global__ __launch_bounds__(256) void mano_bwd_kernel(ManoConst m, const float* __restrict__ betas,
                                                      const float* __restrict__ full_pose,
                                                      const float* __restrict__ scale, const float* __restrict__ transl,
                                                      const float* __restrict__ d_tfs, const float* __restrict__ d_verts,
                                                      float* __restrict__ d_pose, float* __restrict__ d_betas,
                                                      float* __restrict__ d_transl) {
  __shared__ Sh S;
  const int f = blockIdx.x, tid = threadIdx.x;
  forward_core(S, m, betas + f * NB, full_pose + f * 48, tid);
  const float s = scale[f];
  // ---- d tfs -> dA (3x4 per joint), d transl ----
  float dt_loc = 0.f;
  if (tid < NJ * 12) {
    const int j = tid / 12, a = (tid % 12) / 4, b = tid % 4;
    float g = 0.f;
    if (d_tfs) {
      const float* D = d_tfs + (long)f * 256 + j * 16 + a * 4;
      if (m.tfs_c_inv) {
        const float* C = m.tfs_c_inv + j * 16 + b * 4;  // (D . Cinv^T)[a][b] = sum_k D[a][k] Cinv[b][k]
        g = D[0] * C[0] + D[1] * C[1] + D[2] * C[2] + D[3] * C[3];
      } else {
        g = D[b];
      }
    }
    S.dA[tid] = s * g;
    if (b == 3) dt_loc = s * g;
  }
  __syncthreads();
  // translation gradient: sum over joints of the translation column, per axis
  float dtr[3] = {0.f, 0.f, 0.f};
  for (int a = 0; a < 3; ++a) {
    const bool mine = tid < NJ * 12 && (tid % 4 == 3) && ((tid % 12) / 4 == a);
    dtr[a] = block_sum(mine ? dt_loc : 0.f, S.red, tid);
  }
  // ---- d verts path: skinning backward ----
  if (d_verts) {
    // d transl += s * sum_v d_out_v
    for (int a = 0; a < 3; ++a) {
      float loc = 0.f;
      for (int v = tid; v < NV; v += 256) loc += d_verts[((long)f * NV + v) * 3 + a];
      dtr[a] += s * block_sum(loc, S.red, tid);
    }
    // dA_j += sum_v W_vj (s d_out_v) [v_posed_v; 1]^T ; thread (j,e) loops over vertices
    if (tid < NJ * 12) {
      const int j = tid / 12, a = (tid % 12) / 4, b = tid % 4;
      float g = 0.f;
      for (int v = 0; v < NV; ++v) {
        const float dv = s * d_verts[((long)f * NV + v) * 3 + a];
        const float x = (b == 3) ? 1.f : S.vpo[v * 3 + b];
        g += m.lbs_weights[v * NJ + j] * dv * x;
      }
      S.dA[tid] += g;
    }
    __syncthreads();
    // d v_posed_v = sum_j W_vj A_jR^T (s d_out_v)  -> stored over S.vsh later; keep in registers via vpo overwrite
    for (int v = tid; v < NV; v += 256) {
      float T[9];
#pragma unroll
      for (int e = 0; e < 9; ++e) T[e] = 0.f;
      for (int j = 0; j < NJ; ++j) {
        const float w = m.lbs_weights[v * NJ + j];
        for (int a = 0; a < 3; ++a)
          for (int b = 0; b < 3; ++b) T[a * 3 + b] += w * S.A[j * 12 + a * 4 + b];
      }
      const float g0 = s * d_verts[((long)f * NV + v) * 3], g1 = s * d_verts[((long)f * NV + v) * 3 + 1],
                  g2 = s * d_verts[((long)f * NV + v) * 3 + 2];
      S.vpo[v * 3 + 0] = T[0] * g0 + T[3] * g1 + T[6] * g2;
      S.vpo[v * 3 + 1] = T[1] * g0 + T[4] * g1 + T[7] * g2;
      S.vpo[v * 3 + 2] = T[2] * g0 + T[5] * g1 + T[8] * g2;
    }
  } else {
    for (int c = tid; c < NV * 3; c += 256) S.vpo[c] = 0.f;
  }
  __syncthreads();
  // S.vpo now holds d v_posed.  d pf = posedirs . d v_posed
  if (tid < NPF) {
    float g = 0.f;
    if (d_verts)
      for (int c = 0; c < NV * 3; ++c) g += m.posedirs[(long)tid * (NV * 3) + c] * S.vpo[c];
    S.dpf[tid] = g;
  }
  // ---- A -> world transforms and joints ----
  if (tid < NJ) {
    const int j = tid;
    for (int a = 0; a < 3; ++a) {
      const float dat = S.dA[j * 12 + a * 4 + 3];
      for (int b = 0; b < 3; ++b) S.dGr[j * 9 + a * 3 + b] = S.dA[j * 12 + a * 4 + b] - dat * S.J[j * 3 + b];
      S.dGt[j * 3 + a] = dat;
    }
    for (int b = 0; b < 3; ++b) {
      float g = 0.f;
      for (int a = 0; a < 3; ++a) g -= S.Gr[j * 9 + a * 3 + b] * S.dA[j * 12 + a * 4 + 3];
      S.dJ[j * 3 + b] = g;
    }
  }
  __syncthreads();
  if (tid == 0) {  // chain backward, children first
    for (int j = NJ - 1; j >= 1; --j) {
      const int p = m.parents[j];
      float drel[3];
      for (int b = 0; b < 3; ++b) {
        for (int c = 0; c < 3; ++c) {
          float g = 0.f;
          for (int a = 0; a < 3; ++a) g += S.Gr[p * 9 + a * 3 + b] * S.dGr[j * 9 + a * 3 + c];
          S.dR[j * 9 + b * 3 + c] = g;  // dR_j = Gr_p^T dGr_j
        }
        float g = 0.f;
        for (int a = 0; a < 3; ++a) g += S.Gr[p * 9 + a * 3 + b] * S.dGt[j * 3 + a];
        drel[b] = g;
      }
      for (int a = 0; a < 3; ++a) {
        for (int b = 0; b < 3; ++b) {
          float g = S.dGt[j * 3 + a] * S.rel[j * 3 + b];
          for (int c = 0; c < 3; ++c) g += S.dGr[j * 9 + a * 3 + c] * S.R[j * 9 + b * 3 + c];
          S.dGr[p * 9 + a * 3 + b] += g;  // dGr_p += dGr_j R_j^T + dGt_j rel_j^T
        }
        S.dGt[p * 3 + a] += S.dGt[j * 3 + a];
      }
      for (int b = 0; b < 3; ++b) {
        S.dJ[j * 3 + b] += drel[b];
        S.dJ[p * 3 + b] -= drel[b];
      }
    }
    for (int i = 0; i < 9; ++i) S.dR[i] = S.dGr[i];
    for (int b = 0; b < 3; ++b) S.dJ[b] += S.dGt[b];
  }
  __syncthreads();
  // pose-feature path adds to dR_1..15
  if (tid < NPF) {
    const int j = 1 + tid / 9, e = tid % 9;
    S.dR[j * 9 + e] += S.dpf[tid];
  }
  __syncthreads();
  if (tid < NJ) {
    float dr[3];
    rodrigues_bwd(S.pose + tid * 3, S.dR + tid * 9, dr);
    for (int a = 0; a < 3; ++a) d_pose[(long)f * 48 + tid * 3 + a] = dr[a];
  }
  // d v_shaped = d v_posed + J_regressor^T dJ ; d betas = shapedirs^T d v_shaped
  float db[NB];
#pragma unroll
  for (int l = 0; l < NB; ++l) db[l] = 0.f;
  for (int c = tid; c < NV * 3; c += 256) {
    const int v = c / 3, k = c % 3;
    float g = S.vpo[c];
    for (int j = 0; j < NJ; ++j) g += m.J_regressor[j * NV + v] * S.dJ[j * 3 + k];
    const float* sd = m.shapedirs + (long)c * NB;
#pragma unroll
    for (int l = 0; l < NB; ++l) db[l] += g * sd[l];
  }
  for (int l = 0; l < NB; ++l) {
    const float r = block_sum(db[l], S.red, tid);
    if (tid == 0) d_betas[(long)f * NB + l] = r;
  }
  if (tid < 3) d_transl[(long)f * 3 + tid] = dtr[tid];
}

inline int ok() { return hipGetLastError() == hipSuccess ? HOLD_OK : HOLD_E_LAUNCH; }

}  // namespace

extern "C" int hold_mano_lbs_fwd(const hold_mano_model* mm, int32_t n_frames, const float* betas,
                                 const float* full_pose, const float* scene_scale, const float* transl, float* verts,
                                 float* jnts, float* tfs, float* v_posed, hold_stream_t st) {
  if (!mm || !betas || !full_pose || !scene_scale || !transl || !verts || !tfs || n_frames < 0) return HOLD_E_ARG;
  if (n_frames == 0) return HOLD_OK;
  ManoConst m = {mm->v_template, mm->shapedirs, mm->posedirs, mm->J_regressor, mm->parents,
                 mm->lbs_weights, mm->pose_mean, mm->tfs_c_inv};
  hipLaunchKernelGGL(mano_fwd_kernel, dim3(n_frames), dim3(256), 0, (hipStream_t)st, m, betas, full_pose, scene_scale,
                     transl, verts, jnts, tfs, v_posed);
  return ok();
}

extern "C" int hold_mano_lbs_bwd(const hold_mano_model* mm, int32_t n_frames, const float* betas,
                                 const float* full_pose, const float* scene_scale, const float* transl,
                                 const float* d_tfs, const float* d_verts, float* d_pose, float* d_betas,
                                 float* d_transl, hold_stream_t st) {
  if (!mm || !betas || !full_pose || !scene_scale || !transl || (!d_tfs && !d_verts) || !d_pose || !d_betas ||
      !d_transl || n_frames < 0)
    return HOLD_E_ARG;
  if (n_frames == 0) return HOLD_OK;
  ManoConst m = {mm->v_template, mm->shapedirs, mm->posedirs, mm->J_regressor, mm->parents,
                 mm->lbs_weights, mm->pose_mean, mm->tfs_c_inv};
  hipLaunchKernelGGL(mano_bwd_kernel, dim3(n_frames), dim3(256), 0, (hipStream_t)st, m, betas, full_pose, scene_scale,
                     transl, d_tfs, d_verts, d_pose, d_betas, d_transl);
  return ok();
}
