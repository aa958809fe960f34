// Rigid-frame geometry of Invariant Point Attention around the attention core (reference
// src/model/ipa_pytorch_dynamic.py:363-390 points -> global frame, :470-488 o_pt -> local frame + norms), forward and
// backward incl. the gradient with respect to the frames (quaternion via dL/dR, translation).  The reference runs
// these as dozens of tiny aten ops on [F,N,H*P,3] tensors (chunk/stack/unsqueeze/mul/sum/...); here one workgroup
// handles one residue, its 3x3 gradient accumulator is a block reduction.  All fp32.
#include "dfold_common.h"
#include "../../include/dfold_hip.h"

#define GH 8
#define GPQ 8
#define GPV 12
#define NQ (GH * GPQ)            // 64 query points per residue
#define NKV (GH * (GPQ + GPV))   // 160 key+value points per residue
#define NPT (NQ + NKV)           // 224
#define NOV (GH * GPV)           // 96 output points

__device__ __forceinline__ void quat_rot(const float* t7, float* R) {
  const float a = t7[0], b = t7[1], c = t7[2], d = t7[3];
  R[0] = a * a + b * b - c * c - d * d; R[1] = 2 * (b * c - a * d);         R[2] = 2 * (b * d + a * c);
  R[3] = 2 * (b * c + a * d);         R[4] = a * a - b * b + c * c - d * d; R[5] = 2 * (c * d - a * b);
  R[6] = 2 * (b * d - a * c);         R[7] = 2 * (c * d + a * b);         R[8] = a * a - b * b - c * c + d * d;
}

// dL/dq from G = dL/dR (row-major 3x3) for the quadratic-form rotation matrix above
__device__ __forceinline__ void drot_to_dquat(const float* t7, const float* G, float* dq) {
  const float a = t7[0], b = t7[1], c = t7[2], d = t7[3];
  dq[0] = 2.f * (a * (G[0] + G[4] + G[8]) + d * (G[3] - G[1]) + c * (G[2] - G[6]) + b * (G[7] - G[5]));
  dq[1] = 2.f * (b * (G[0] - G[4] - G[8]) + c * (G[1] + G[3]) + d * (G[2] + G[6]) + a * (G[7] - G[5]));
  dq[2] = 2.f * (c * (-G[0] + G[4] - G[8]) + b * (G[1] + G[3]) + a * (G[2] - G[6]) + d * (G[5] + G[7]));
  dq[3] = 2.f * (d * (-G[0] - G[4] + G[8]) + a * (G[3] - G[1]) + b * (G[2] + G[6]) + c * (G[5] + G[7]));
}

// block reduction of NV per-thread values over 256 threads -> result valid on thread 0
template <int NV>
__device__ __forceinline__ void block_sum(float* v, float* red /* [4][NV] */) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const float s = wave_sum(v[k]);
    if (lane == 0) red[w * NV + k] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = red[k] + red[NV + k] + red[2 * NV + k] + red[3 * NV + k];
  }
}

// raw_q [P][3*NQ], raw_kv [P][3*NKV] (x | y | z blocks, index h*npts + pt) -> q_pts [P][H][PQ][3], k_pts, v_pts [P][H][PV][3]
__global__ __launch_bounds__(256) void ipa_points_fwd_kernel(const float* __restrict__ raw_q, const float* __restrict__ raw_kv,
                                                             const float* __restrict__ t7, float* __restrict__ q_pts,
                                                             float* __restrict__ k_pts, float* __restrict__ v_pts) {
  const long p = blockIdx.x;
  const int i = threadIdx.x;
  if (i >= NPT) return;
  float R[9];
  quat_rot(t7 + p * 7, R);
  const float tx = t7[p * 7 + 4], ty = t7[p * 7 + 5], tz = t7[p * 7 + 6];
  float x, y, z;
  float* dst;
  if (i < NQ) {
    const float* r = raw_q + p * 3 * NQ;
    x = r[i]; y = r[NQ + i]; z = r[2 * NQ + i];
    dst = q_pts + (p * NQ + i) * 3;
  } else {
    const int k = i - NQ;
    const float* r = raw_kv + p * 3 * NKV;
    x = r[k]; y = r[NKV + k]; z = r[2 * NKV + k];
    const int h = k / (GPQ + GPV), pt = k - h * (GPQ + GPV);
    dst = pt < GPQ ? k_pts + ((p * GH + h) * GPQ + pt) * 3 : v_pts + ((p * GH + h) * GPV + pt - GPQ) * 3;
  }
  dst[0] = R[0] * x + R[1] * y + R[2] * z + tx;
  dst[1] = R[3] * x + R[4] * y + R[5] * z + ty;
  dst[2] = R[6] * x + R[7] * y + R[8] * z + tz;
}

extern "C" int dfold_ipa_points_fwd(const float* raw_q, const float* raw_kv, const float* t7, float* q_pts, float* k_pts,
                                    float* v_pts, int64_t P, void* stream) {
  if (!raw_q || !raw_kv || !t7 || !q_pts || !k_pts || !v_pts || P <= 0) return DFOLD_EINVAL;
  DFOLD_LAUNCH(ipa_points_fwd_kernel, dim3((unsigned)P), dim3(256), 0, (hipStream_t)stream, raw_q, raw_kv, t7, q_pts, k_pts, v_pts);
  return dfold_check_launch();
}

__global__ __launch_bounds__(256) void ipa_points_bwd_kernel(const float* __restrict__ raw_q, const float* __restrict__ raw_kv,
                                                             const float* __restrict__ t7, const float* __restrict__ dq_pts,
                                                             const float* __restrict__ dk_pts, const float* __restrict__ dv_pts,
                                                             float* __restrict__ draw_q, float* __restrict__ draw_kv,
                                                             float* __restrict__ dt7) {
  __shared__ float red[4 * 12];
  const long p = blockIdx.x;
  const int i = threadIdx.x;
  float R[9];
  quat_rot(t7 + p * 7, R);
  float acc[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) acc[k] = 0.f;
  if (i < NPT) {
    float x, y, z;
    const float* g;
    float* dx;
    int stride;
    if (i < NQ) {
      const float* r = raw_q + p * 3 * NQ;
      x = r[i]; y = r[NQ + i]; z = r[2 * NQ + i];
      g = dq_pts + (p * NQ + i) * 3;
      dx = draw_q + p * 3 * NQ + i;
      stride = NQ;
    } else {
      const int k = i - NQ;
      const float* r = raw_kv + p * 3 * NKV;
      x = r[k]; y = r[NKV + k]; z = r[2 * NKV + k];
      const int h = k / (GPQ + GPV), pt = k - h * (GPQ + GPV);
      g = pt < GPQ ? dk_pts + ((p * GH + h) * GPQ + pt) * 3 : dv_pts + ((p * GH + h) * GPV + pt - GPQ) * 3;
      dx = draw_kv + p * 3 * NKV + k;
      stride = NKV;
    }
    const float g0 = g[0], g1 = g[1], g2 = g[2];
    dx[0] = R[0] * g0 + R[3] * g1 + R[6] * g2;            // R^T g
    dx[stride] = R[1] * g0 + R[4] * g1 + R[7] * g2;
    dx[2 * stride] = R[2] * g0 + R[5] * g1 + R[8] * g2;
    acc[0] = g0 * x; acc[1] = g0 * y; acc[2] = g0 * z;    // dL/dR_ij += g_i x_j
    acc[3] = g1 * x; acc[4] = g1 * y; acc[5] = g1 * z;
    acc[6] = g2 * x; acc[7] = g2 * y; acc[8] = g2 * z;
    acc[9] = g0; acc[10] = g1; acc[11] = g2;
  }
  block_sum<12>(acc, red);
  if (threadIdx.x == 0) {
    float dq[4];
    drot_to_dquat(t7 + p * 7, acc, dq);
#pragma unroll
    for (int k = 0; k < 4; ++k) dt7[p * 7 + k] = dq[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) dt7[p * 7 + 4 + k] = acc[9 + k];
  }
}

extern "C" int dfold_ipa_points_bwd(const float* raw_q, const float* raw_kv, const float* t7, const float* dq_pts,
                                    const float* dk_pts, const float* dv_pts, float* draw_q, float* draw_kv, float* dt7, int64_t P,
                                    void* stream) {
  if (!raw_q || !raw_kv || !t7 || !dq_pts || !dk_pts || !dv_pts || !draw_q || !draw_kv || !dt7 || P <= 0) return DFOLD_EINVAL;
  DFOLD_LAUNCH(ipa_points_bwd_kernel, dim3((unsigned)P), dim3(256), 0, (hipStream_t)stream, raw_q, raw_kv, t7, dq_pts, dk_pts,
               dv_pts, draw_q, draw_kv, dt7);
  return dfold_check_launch();
}

// o_pt_g [P][H][PV][3] (global frame) -> geo_l bf16 [P][4*NOV] = [l_x | l_y | l_z | |l|], geo_g bf16 [P][4*NOV] likewise for g
//   l = R^T (g - t);  norms sqrt(|.|^2 + eps)
__global__ __launch_bounds__(128) void ipa_outfeat_fwd_kernel(const float* __restrict__ o_pt, const float* __restrict__ t7,
                                                              bf16_t* __restrict__ geo_l, bf16_t* __restrict__ geo_g, long ld,
                                                              float eps) {
  const long p = blockIdx.x;
  const int i = threadIdx.x;
  if (i >= NOV) return;
  float R[9];
  quat_rot(t7 + p * 7, R);
  const float* g = o_pt + (p * NOV + i) * 3;
  const float g0 = g[0], g1 = g[1], g2 = g[2];
  const float u0 = g0 - t7[p * 7 + 4], u1 = g1 - t7[p * 7 + 5], u2 = g2 - t7[p * 7 + 6];
  const float l0 = R[0] * u0 + R[3] * u1 + R[6] * u2;
  const float l1 = R[1] * u0 + R[4] * u1 + R[7] * u2;
  const float l2 = R[2] * u0 + R[5] * u1 + R[8] * u2;
  bf16_t* ol = geo_l + p * ld;
  bf16_t* og = geo_g + p * ld;
  ol[i] = f2bf(l0); ol[NOV + i] = f2bf(l1); ol[2 * NOV + i] = f2bf(l2);
  ol[3 * NOV + i] = f2bf(sqrtf(l0 * l0 + l1 * l1 + l2 * l2 + eps));
  og[i] = f2bf(g0); og[NOV + i] = f2bf(g1); og[2 * NOV + i] = f2bf(g2);
  og[3 * NOV + i] = f2bf(sqrtf(g0 * g0 + g1 * g1 + g2 * g2 + eps));
}

extern "C" int dfold_ipa_outfeat_fwd(const float* o_pt, const float* t7, void* geo_l, void* geo_g, int64_t ld, int64_t P,
                                     float eps, void* stream) {
  if (!o_pt || !t7 || !geo_l || !geo_g || P <= 0 || ld < 4 * NOV) return DFOLD_EINVAL;
  DFOLD_LAUNCH(ipa_outfeat_fwd_kernel, dim3((unsigned)P), dim3(128), 0, (hipStream_t)stream, o_pt, t7, (bf16_t*)geo_l,
               (bf16_t*)geo_g, (long)ld, eps);
  return dfold_check_launch();
}

__global__ __launch_bounds__(256) void ipa_outfeat_bwd_kernel(const float* __restrict__ o_pt, const float* __restrict__ t7,
                                                              const bf16_t* __restrict__ dgeo_l, const bf16_t* __restrict__ dgeo_g,
                                                              float* __restrict__ do_pt, float* __restrict__ dt7, float eps) {
  __shared__ float red[4 * 12];
  const long p = blockIdx.x;
  const int i = threadIdx.x;
  float R[9];
  quat_rot(t7 + p * 7, R);
  float acc[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) acc[k] = 0.f;
  if (i < NOV) {
    const float* g = o_pt + (p * NOV + i) * 3;
    const float g0 = g[0], g1 = g[1], g2 = g[2];
    const float u0 = g0 - t7[p * 7 + 4], u1 = g1 - t7[p * 7 + 5], u2 = g2 - t7[p * 7 + 6];
    const float l0 = R[0] * u0 + R[3] * u1 + R[6] * u2;
    const float l1 = R[1] * u0 + R[4] * u1 + R[7] * u2;
    const float l2 = R[2] * u0 + R[5] * u1 + R[8] * u2;
    const bf16_t* gl = dgeo_l + p * 4 * NOV;
    const bf16_t* gg = dgeo_g + p * 4 * NOV;
    const float inl = bf2f(gl[3 * NOV + i]) / sqrtf(l0 * l0 + l1 * l1 + l2 * l2 + eps);
    const float ing = bf2f(gg[3 * NOV + i]) / sqrtf(g0 * g0 + g1 * g1 + g2 * g2 + eps);
    const float dl0 = bf2f(gl[i]) + inl * l0, dl1 = bf2f(gl[NOV + i]) + inl * l1, dl2 = bf2f(gl[2 * NOV + i]) + inl * l2;
    // du = R dl ; dg = du + direct global-frame terms ; dt = -sum du ; dL/dR_ji += dl_i u_j  (l_i = sum_j R_ji u_j)
    const float du0 = R[0] * dl0 + R[1] * dl1 + R[2] * dl2;
    const float du1 = R[3] * dl0 + R[4] * dl1 + R[5] * dl2;
    const float du2 = R[6] * dl0 + R[7] * dl1 + R[8] * dl2;
    float* dg = do_pt + (p * NOV + i) * 3;
    dg[0] = du0 + bf2f(gg[i]) + ing * g0;
    dg[1] = du1 + bf2f(gg[NOV + i]) + ing * g1;
    dg[2] = du2 + bf2f(gg[2 * NOV + i]) + ing * g2;
    acc[0] = u0 * dl0; acc[1] = u0 * dl1; acc[2] = u0 * dl2;   // G[j][i] = u_j dl_i
    acc[3] = u1 * dl0; acc[4] = u1 * dl1; acc[5] = u1 * dl2;
    acc[6] = u2 * dl0; acc[7] = u2 * dl1; acc[8] = u2 * dl2;
    acc[9] = -du0; acc[10] = -du1; acc[11] = -du2;
  }
  block_sum<12>(acc, red);
  if (threadIdx.x == 0) {
    float dq[4];
    drot_to_dquat(t7 + p * 7, acc, dq);
#pragma unroll
    for (int k = 0; k < 4; ++k) dt7[p * 7 + k] = dq[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) dt7[p * 7 + 4 + k] = acc[9 + k];
  }
}

extern "C" int dfold_ipa_outfeat_bwd(const float* o_pt, const float* t7, const void* dgeo_l, const void* dgeo_g, float* do_pt,
                                     float* dt7, int64_t P, float eps, void* stream) {
  if (!o_pt || !t7 || !dgeo_l || !dgeo_g || !do_pt || !dt7 || P <= 0) return DFOLD_EINVAL;
  DFOLD_LAUNCH(ipa_outfeat_bwd_kernel, dim3((unsigned)P), dim3(256), 0, (hipStream_t)stream, o_pt, t7, (const bf16_t*)dgeo_l,
               (const bf16_t*)dgeo_g, do_pt, dt7, eps);
  return dfold_check_launch();
}

// ---------------------------------------------------------------------------------------------
// Backbone frame update (reference Rigid.compose_q_update_vec, openfold/utils/rigid_utils.py:1039-1063 ->
// Rotation.compose_q_update_vec :587-616 -> quat_multiply_by_vec :266-275, normalisation :331-332; called at
// src/model/ipa_pytorch_dynamic.py:871 with the masked 6-vector of BackboneUpdate):
//   q' = normalize(q + m * (q (x) (0, u)));   t' = t + m * R(q) v        (R = quadratic form of the UNnormalised q)
// One thread per frame, forward and analytic backward (the torch composition of this op was ~90 elementwise launches
// per call and ~230 in its backward).  t7 [P][7], upd [P][6] = (u, v), mask [P] or NULL.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void compose_fwd_kernel(const float* __restrict__ t7, const float* __restrict__ upd,
                                                          const float* __restrict__ mask, float* __restrict__ out, long P) {
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
  const float* s = t7 + p * 7;
  const float a = s[0], b = s[1], c = s[2], d = s[3];
  const float* w = upd + p * 6;
  const float m = mask != nullptr ? mask[p] : 1.f;
  const float u0 = w[0], u1 = w[1], u2 = w[2];
  float qs[4];
  qs[0] = a + m * (-b * u0 - c * u1 - d * u2);
  qs[1] = b + m * (a * u0 + c * u2 - d * u1);
  qs[2] = c + m * (a * u1 - b * u2 + d * u0);
  qs[3] = d + m * (a * u2 + b * u1 - c * u0);
  const float inv = 1.f / sqrtf(qs[0] * qs[0] + qs[1] * qs[1] + qs[2] * qs[2] + qs[3] * qs[3]);
  float R[9];
  quat_rot(s, R);
  float* o = out + p * 7;
#pragma unroll
  for (int k = 0; k < 4; ++k) o[k] = qs[k] * inv;
  o[4] = s[4] + m * (R[0] * w[3] + R[1] * w[4] + R[2] * w[5]);
  o[5] = s[5] + m * (R[3] * w[3] + R[4] * w[4] + R[5] * w[5]);
  o[6] = s[6] + m * (R[6] * w[3] + R[7] * w[4] + R[8] * w[5]);
}

__global__ __launch_bounds__(256) void compose_bwd_kernel(const float* __restrict__ t7, const float* __restrict__ upd,
                                                          const float* __restrict__ mask, const float* __restrict__ g,
                                                          float* __restrict__ dt7, float* __restrict__ dupd, long P) {
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
  const float* s = t7 + p * 7;
  const float a = s[0], b = s[1], c = s[2], d = s[3];
  const float* w = upd + p * 6;
  const float m = mask != nullptr ? mask[p] : 1.f;
  const float u0 = w[0], u1 = w[1], u2 = w[2];
  float qs[4];
  qs[0] = a + m * (-b * u0 - c * u1 - d * u2);
  qs[1] = b + m * (a * u0 + c * u2 - d * u1);
  qs[2] = c + m * (a * u1 - b * u2 + d * u0);
  qs[3] = d + m * (a * u2 + b * u1 - c * u0);
  const float inv = 1.f / sqrtf(qs[0] * qs[0] + qs[1] * qs[1] + qs[2] * qs[2] + qs[3] * qs[3]);
  const float* gp = g + p * 7;
  // through the normalisation: gqs = (gq - qn (qn . gq)) / |qs|
  float qn[4], gq[4], dot = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    qn[k] = qs[k] * inv;
    dot += qn[k] * gp[k];
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) gq[k] = (gp[k] - qn[k] * dot) * inv;
  // qs = q + m (q (x) w), w = (0, u):  dL/dq = gqs + m (gqs (x) w*),  dL/dw = m (q* (x) gqs)
  float dq[4];
  dq[0] = gq[0] + m * (gq[1] * u0 + gq[2] * u1 + gq[3] * u2);
  dq[1] = gq[1] + m * (-gq[0] * u0 - gq[2] * u2 + gq[3] * u1);
  dq[2] = gq[2] + m * (-gq[0] * u1 + gq[1] * u2 - gq[3] * u0);
  dq[3] = gq[3] + m * (-gq[0] * u2 - gq[1] * u1 + gq[2] * u0);
  float* du = dupd + p * 6;
  du[0] = m * (a * gq[1] - b * gq[0] - c * gq[3] + d * gq[2]);
  du[1] = m * (a * gq[2] + b * gq[3] - c * gq[0] - d * gq[1]);
  du[2] = m * (a * gq[3] - b * gq[2] + c * gq[1] - d * gq[0]);
  // t' = t + m R(q) v
  float R[9], G[9];
  quat_rot(s, R);
  const float gt0 = gp[4], gt1 = gp[5], gt2 = gp[6];
  du[3] = m * (R[0] * gt0 + R[3] * gt1 + R[6] * gt2);
  du[4] = m * (R[1] * gt0 + R[4] * gt1 + R[7] * gt2);
  du[5] = m * (R[2] * gt0 + R[5] * gt1 + R[8] * gt2);
  G[0] = m * gt0 * w[3]; G[1] = m * gt0 * w[4]; G[2] = m * gt0 * w[5];
  G[3] = m * gt1 * w[3]; G[4] = m * gt1 * w[4]; G[5] = m * gt1 * w[5];
  G[6] = m * gt2 * w[3]; G[7] = m * gt2 * w[4]; G[8] = m * gt2 * w[5];
  float dqr[4];
  drot_to_dquat(s, G, dqr);
  float* o = dt7 + p * 7;
#pragma unroll
  for (int k = 0; k < 4; ++k) o[k] = dq[k] + dqr[k];
  o[4] = gt0; o[5] = gt1; o[6] = gt2;
}

extern "C" int dfold_compose_fwd(const float* t7, const float* upd6, const float* mask, float* out, int64_t P, void* stream) {
  if (!t7 || !upd6 || !out || P <= 0) return DFOLD_EINVAL;
  DFOLD_LAUNCH(compose_fwd_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, (hipStream_t)stream, t7, upd6, mask, out,
               (long)P);
  return dfold_check_launch();
}

extern "C" int dfold_compose_bwd(const float* t7, const float* upd6, const float* mask, const float* g, float* dt7, float* dupd6,
                                 int64_t P, void* stream) {
  if (!t7 || !upd6 || !g || !dt7 || !dupd6 || P <= 0) return DFOLD_EINVAL;
  DFOLD_LAUNCH(compose_bwd_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, (hipStream_t)stream, t7, upd6, mask, g, dt7,
               dupd6, (long)P);
  return dfold_check_launch();
}

// ---------------------------------------------------------------------------------------------
// Pair-side projections of an IPA block in ONE pass over the pair tensor (round 4): linear_b (:396, 8 heads; its bias drops
// out of the softmax) and down_z (:498, 32 channels; its bias is added after the aggregation) were three GEMM launches, each
// reading z [B,N,N,128] (134 MB at config 3) for a 8- / 32-wide output: 427 us per block.  Here a wave takes 16 consecutive
// cells of a pair-tensor row, multiplies them with the 48 (8 + 8 zero + 32) weight rows on the matrix cores (12 MFMA
// 16x16x32) and writes all three consumers' layouts: bias_t fp32 [B][8][N][N] (head-major planes), pz bf16 [B][N][N][32]
// and pzT bf16 [B][N][32][N] (key-contiguous).  N % 8 == 0.
// ---------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) unsigned ppu32x4;
__global__ __launch_bounds__(256) void ipa_pair_proj_kernel(const bf16_t* __restrict__ z, const bf16_t* __restrict__ wb,
                                                            const bf16_t* __restrict__ wdz, float* __restrict__ bias_t,
                                                            bf16_t* __restrict__ pz, bf16_t* __restrict__ pzT, int B, int N) {
  __shared__ __attribute__((aligned(16))) bf16_t stage[4][16][40];       // wave-private [16 cells][32 channels] (+ pad)
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  // weight fragments (B operand: [n = output row l15][k = l4*8 ..]): tile 0 = linear_b (rows 8..15 zero), 1 / 2 = down_z
  bf16x8 wf[3][4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const ppu32x4 zero = {0u, 0u, 0u, 0u};
    wf[0][ks] = l15 < 8 ? *(const bf16x8*)(wb + l15 * 128 + ks * 32 + l4 * 8) : __builtin_bit_cast(bf16x8, zero);
    wf[1][ks] = *(const bf16x8*)(wdz + l15 * 128 + ks * 32 + l4 * 8);
    wf[2][ks] = *(const bf16x8*)(wdz + (16 + l15) * 128 + ks * 32 + l4 * 8);
  }
  const int tpr = (N + 15) >> 4;                      // tiles per pair-tensor row
  const long ntiles = (long)B * N * tpr;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  for (long t = (long)blockIdx.x * 4 + w; t < ntiles; t += (long)gridDim.x * 4) {
    const int jt = (int)(t % tpr);
    const long bi = t / tpr;                          // b * N + i
    const int j0 = jt * 16;
    const int cell = min(j0 + l15, N - 1);            // (cells past the end of the row: a valid address, results dropped)
    const bf16_t* zr = z + (bi * N + cell) * 128 + l4 * 8;
    bf16x8 a[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) a[ks] = *(const bf16x8*)(zr + ks * 32);
    f32x4 acc[3] = {zero4, zero4, zero4};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int c = 0; c < 3; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ks], wf[c][ks], acc[c], 0, 0, 0);
    // accumulator: rows = cells j0 + l4*4 + r, column = output row l15 of the tile
    const int jq = j0 + l4 * 4;                       // N % 4 == 0: the four cells of a lane are in or out together
    if (jq < N) {
      const long b = bi / N;
      const long i = bi - b * N;
      if (l15 < 8) *(f32x4*)(bias_t + ((b * 8 + l15) * N + i) * (long)N + jq) = acc[0];
#pragma unroll
      for (int c = 1; c < 3; ++c)
        *(uint2*)(pzT + (bi * 32 + (c - 1) * 16 + l15) * (long)N + jq) = make_uint2(pack2bf_hw(acc[c][0], acc[c][1]), pack2bf_hw(acc[c][2], acc[c][3]));
    }
    // pz rows through the wave's LDS tile: [cell][channel] -> 64-byte rows, 16 bytes per lane
#pragma unroll
    for (int c = 1; c < 3; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) stage[w][l4 * 4 + r][(c - 1) * 16 + l15] = f2bf_hw(acc[c][r]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    {
      const int row = lane >> 2, ch = (lane & 3) * 8;
      if (j0 + row < N) *(uint4*)(pz + (bi * N + j0 + row) * 32 + ch) = *(const uint4*)&stage[w][row][ch];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  }
}

extern "C" int dfold_ipa_pair_proj(const void* z_bf16, const void* w_b_bf16, const void* w_dz_bf16, float* bias_t, void* pz_bf16,
                                   void* pzT_bf16, int32_t B, int32_t N, void* stream) {
  if (!z_bf16 || !w_b_bf16 || !w_dz_bf16 || !bias_t || !pz_bf16 || !pzT_bf16 || B <= 0 || N <= 0 || (N & 7)) return DFOLD_EINVAL;
  const long ntiles = (long)B * N * ((N + 15) / 16);
  long grid = (ntiles + 3) / 4;
  if (grid > 4096) grid = 4096;
  DFOLD_LAUNCH(ipa_pair_proj_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)z_bf16,
               (const bf16_t*)w_b_bf16, (const bf16_t*)w_dz_bf16, bias_t, (bf16_t*)pz_bf16, (bf16_t*)pzT_bf16, B, N);
  return dfold_check_launch();
}
