// IGSO(3) score series of the rotation head (reference SO3Diffuser.torch_score, src/data/so3_diffuser.py:274-305
// -> igso3_expansion :9-49 and score :71-117, with use_cached_score=False):
//
//   f(w)    = sum_{l<L} env_l * sin((l+1/2) w) / sin(w/2)            env_l = (2l+1) exp(-l(l+1) sigma^2 / 2)
//   dsig(w) = sum_{l<L} env_l * (lo*dhi_l - hi_l*dlo) / lo^2        (= df/dw)
//   sc(w)   = dsig / (f + 1e-4)
//
// The reference's mixed precision is reproduced: w and every trig term are fp32 (the argument (l+1/2)*w is
// one fp32 product), env_l is fp64 (sigma comes from numpy), products/sums promote to fp64.  The kernel also
// returns d sc / d w (analytic, fp64) so the backward needs no [P,1000] temporaries (the reference's autograd
// keeps ~8 of them per call).  env is a per-window table computed on the host exactly as the reference does.
#include "dfold_common.h"
#include "../../include/dfold_hip.h"

// Eight lanes per position, each walking every eighth term of the series; the per-term fp64 DIVISIONS are hoisted out of the
// sum (f = (sum e hi) / lo etc.: the same value up to fp64 rounding, 1e-16): three fp64 FMAs per term instead of three fp64
// divides, and 8 x the parallelism -- one thread per position ran the 1536 positions of a config-1 sampler forward on 6 CUs for
// 0.43 ms (profiles/r5_eval_forward_kernel_stats.csv).
#define IG_LPP 8
__global__ __launch_bounds__(256) void igso3_series_kernel(const float* __restrict__ omega, const double* __restrict__ env,
                                                           double* __restrict__ sc, double* __restrict__ dsc, long P,
                                                           long per_window, int L) {
  extern __shared__ double envs[];
  const int w = blockIdx.y;
  for (int l = threadIdx.x; l < L; l += 256) envs[l] = env[(long)w * L + l];
  __syncthreads();
  const int sub = threadIdx.x & (IG_LPP - 1);
  const long loc = (long)blockIdx.x * (256 / IG_LPP) + (threadIdx.x / IG_LPP);
  const bool live = loc < per_window && (long)w * per_window + loc < P;
  const long p = live ? (long)w * per_window + loc : (long)w * per_window;     // (dead lanes compute on a valid address: the shuffles below need every lane)
  const float om = omega[p];
  const float lo = sinf(om * 0.5f);
  const float dlo = 0.5f * cosf(om * 0.5f);
  double s1 = 0.0, s2 = 0.0, s3 = 0.0;       // sum e hi,  sum e (lo dhi - hi dlo),  sum e hi (1/4 - lh^2)
  for (int l = sub; l < L; l += IG_LPP) {
    const float lh = (float)l + 0.5f;
    const float arg = om * lh;
    const float hi = sinf(arg);
    const float dhi = lh * cosf(arg);
    const double e = envs[l];
    const float num = lo * dhi - hi * dlo;
    const double eh = e * (double)hi;
    s1 += eh;
    s2 += e * (double)num;
    s3 += eh * (0.25 - (double)lh * (double)lh);
  }
#pragma unroll
  for (int o = 1; o < IG_LPP; o <<= 1) {
    s1 += __shfl_xor(s1, o, 64);
    s2 += __shfl_xor(s2, o, 64);
    s3 += __shfl_xor(s3, o, 64);
  }
  if (!live || sub != 0) return;
  const double dlo_d = (double)dlo, lod = (double)lo, lo2 = (double)(lo * lo);
  const double f = s1 / lod, ds = s2 / lo2;
  // d/dw (num / lo^2) = lo hi (1/4 - lh^2) / lo^2 - 2 num dlo / lo^3
  const double dds = lod * s3 / lo2 - 2.0 * dlo_d * s2 / (lo2 * lod);
  const double den = f + 1e-4;
  sc[p] = ds / den;
  dsc[p] = (dds * den - ds * ds) / (den * den);
}

extern "C" int dfold_igso3_series(const float* omega, const double* env, double* sc, double* dsc, int64_t P,
                                  int64_t per_window, int32_t L, void* stream) {
  if (!omega || !env || !sc || !dsc || P <= 0 || per_window <= 0 || L <= 0 || L > 4096) return DFOLD_EINVAL;
  if (P % per_window) return DFOLD_EINVAL;
  const long ppb = 256 / IG_LPP;      // positions per block
  dim3 grid((unsigned)((per_window + ppb - 1) / ppb), (unsigned)(P / per_window));
  DFOLD_LAUNCH(igso3_series_kernel, grid, dim3(256), (size_t)L * sizeof(double), (hipStream_t)stream, omega, env, sc,
                     dsc, (long)P, (long)per_window, L);
  return dfold_check_launch();
}

// ------------------------------------------------------------------------------------------------------------------
// The rest of the rotation-score head around the series (round 6): SE3Diffuser.calc_rot_score (src/data/se3_diffuser.py:
// 119-125) = igso3 score of rotvec(q_0^-1 q_t) -- quaternion inverse and product (openfold/utils/rigid_utils.py:230-286),
// quat_to_rotvec (src/data/utils.py:589-606), SO3Diffuser.torch_score's omega = |v| + eps and sc v / (omega + eps)
// (so3_diffuser.py:274-305).  As an aten graph this was ~60 launches forward and ~120 backward on [P,4] tensors.
//   rot_head_pre:  q_t, q_0 -> v = rotvec (fp32, the reference's fp32 arithmetic), omega = |v| + 1e-6
//   (dfold_igso3_series: omega -> sc, dsc)
//   rot_head_post: score = sc v / (omega + 1e-6)   (float64, as the reference's promotion gives)
//   rot_head_bwd:  d score -> d q_0 (analytic chain rule through all of the above, float64 inside)
// ------------------------------------------------------------------------------------------------------------------
struct RotVec {
  float q[4];        // q_0^-1 q_t after the sign flip (w >= 0)
  float sign, n, angle, scale;
  bool small;
};
__device__ __forceinline__ RotVec rot_head_vec(const float* __restrict__ qt, const float* __restrict__ q0) {
  const float s = q0[0] * q0[0] + q0[1] * q0[1] + q0[2] * q0[2] + q0[3] * q0[3];
  const float a1 = q0[0] / s, b1 = -q0[1] / s, c1 = -q0[2] / s, d1 = -q0[3] / s;
  const float a2 = qt[0], b2 = qt[1], c2 = qt[2], d2 = qt[3];
  RotVec r;
  r.q[0] = a1 * a2 - b1 * b2 - c1 * c2 - d1 * d2;
  r.q[1] = a1 * b2 + b1 * a2 + c1 * d2 - d1 * c2;
  r.q[2] = a1 * c2 - b1 * d2 + c1 * a2 + d1 * b2;
  r.q[3] = a1 * d2 + b1 * c2 - c1 * b2 + d1 * a2;
  r.sign = r.q[0] < 0.f ? -1.f : 1.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) r.q[k] *= r.sign;
  r.n = sqrtf(r.q[1] * r.q[1] + r.q[2] * r.q[2] + r.q[3] * r.q[3]);
  r.angle = 2.f * atan2f(r.n, r.q[0]);
  const float a2s = r.angle * r.angle;
  r.small = r.angle <= 1e-3f;
  r.scale = r.small ? 2.f + a2s / 12.f + 7.f * a2s * a2s / 2880.f : r.angle / sinf(r.angle * 0.5f + 1e-6f);
  return r;
}

__global__ __launch_bounds__(256) void rot_head_pre_kernel(const float* __restrict__ qt, const float* __restrict__ q0,
                                                           float* __restrict__ vec, float* __restrict__ omega, long P) {
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
  const RotVec r = rot_head_vec(qt + 4 * p, q0 + 4 * p);
  const float v0 = r.scale * r.q[1], v1 = r.scale * r.q[2], v2 = r.scale * r.q[3];
  vec[3 * p] = v0;
  vec[3 * p + 1] = v1;
  vec[3 * p + 2] = v2;
  omega[p] = sqrtf(v0 * v0 + v1 * v1 + v2 * v2) + 1e-6f;
}

__global__ __launch_bounds__(256) void rot_head_post_kernel(const float* __restrict__ vec, const float* __restrict__ omega,
                                                            const double* __restrict__ sc, double* __restrict__ score, long P) {
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
  const double den = (double)(omega[p] + 1e-6f), s = sc[p];
#pragma unroll
  for (int k = 0; k < 3; ++k) score[3 * p + k] = s * (double)vec[3 * p + k] / den;
}

__global__ __launch_bounds__(256) void rot_head_bwd_kernel(const double* __restrict__ g, const float* __restrict__ qt,
                                                           const float* __restrict__ q0, const float* __restrict__ vec,
                                                           const float* __restrict__ omega, const double* __restrict__ sc,
                                                           const double* __restrict__ dsc, float* __restrict__ d_q0, long P) {
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
  // score_i = sc(om) v_i / (om + eps), om = |v| + eps
  const double v[3] = {(double)vec[3 * p], (double)vec[3 * p + 1], (double)vec[3 * p + 2]};
  const double den = (double)(omega[p] + 1e-6f), s = sc[p];
  const double nv = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  const double gdotv = g[3 * p] * v[0] + g[3 * p + 1] * v[1] + g[3 * p + 2] * v[2];
  const double kr = nv > 0.0 ? gdotv * (dsc[p] / den - s / (den * den)) / nv : 0.0;
  double gv[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) gv[k] = s / den * g[3 * p + k] + kr * v[k];
  // v_k = scale(angle) q'_k, angle = 2 atan2(n, q'_w), n = |q'_xyz|
  const RotVec r = rot_head_vec(qt + 4 * p, q0 + 4 * p);
  const double qw = r.q[0], qx = r.q[1], qy = r.q[2], qz = r.q[3], n = r.n, a = r.angle;
  const double A = gv[0] * qx + gv[1] * qy + gv[2] * qz;
  double ds;
  if (r.small) {
    ds = a / 6.0 + 7.0 * a * a * a / 720.0;
  } else {
    const double S = sin(a * 0.5 + 1e-6), C = cos(a * 0.5 + 1e-6);
    ds = 1.0 / S - a * C / (2.0 * S * S);
  }
  const double r2 = n * n + qw * qw;
  const double da_dn = r2 > 0.0 ? 2.0 * qw / r2 : 0.0, da_dw = r2 > 0.0 ? -2.0 * n / r2 : 0.0;
  const double kn = n > 0.0 ? A * ds * da_dn / n : 0.0;
  double gq[4];
  gq[0] = A * ds * da_dw;
  gq[1] = (double)r.scale * gv[0] + kn * qx;
  gq[2] = (double)r.scale * gv[1] + kn * qy;
  gq[3] = (double)r.scale * gv[2] + kn * qz;
#pragma unroll
  for (int k = 0; k < 4; ++k) gq[k] *= (double)r.sign;
  // q = p (x) q_t, p = conj(q_0) / |q_0|^2
  const double a2 = qt[4 * p], b2 = qt[4 * p + 1], c2 = qt[4 * p + 2], d2 = qt[4 * p + 3];
  const double gp[4] = {a2 * gq[0] + b2 * gq[1] + c2 * gq[2] + d2 * gq[3], -b2 * gq[0] + a2 * gq[1] - d2 * gq[2] + c2 * gq[3],
                        -c2 * gq[0] + d2 * gq[1] + a2 * gq[2] - b2 * gq[3], -d2 * gq[0] - c2 * gq[1] + b2 * gq[2] + a2 * gq[3]};
  const double w0 = q0[4 * p], x0 = q0[4 * p + 1], y0 = q0[4 * p + 2], z0 = q0[4 * p + 3];
  const double ss = w0 * w0 + x0 * x0 + y0 * y0 + z0 * z0;
  const double cj[4] = {w0, -x0, -y0, -z0};
  const double gs = -(gp[0] * cj[0] + gp[1] * cj[1] + gp[2] * cj[2] + gp[3] * cj[3]) / (ss * ss);
  d_q0[4 * p] = (float)(gp[0] / ss + 2.0 * w0 * gs);
  d_q0[4 * p + 1] = (float)(-gp[1] / ss + 2.0 * x0 * gs);
  d_q0[4 * p + 2] = (float)(-gp[2] / ss + 2.0 * y0 * gs);
  d_q0[4 * p + 3] = (float)(-gp[3] / ss + 2.0 * z0 * gs);
}

extern "C" int dfold_rot_head_pre(const float* quats_t, const float* quats_0, float* vec, float* omega, int64_t P, void* stream) {
  if (!quats_t || !quats_0 || !vec || !omega || P <= 0) return DFOLD_EINVAL;
  DFOLD_LAUNCH(rot_head_pre_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, (hipStream_t)stream, quats_t, quats_0, vec, omega,
               (long)P);
  return dfold_check_launch();
}
extern "C" int dfold_rot_head_post(const float* vec, const float* omega, const double* sc, double* score, int64_t P, void* stream) {
  if (!vec || !omega || !sc || !score || P <= 0) return DFOLD_EINVAL;
  DFOLD_LAUNCH(rot_head_post_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, (hipStream_t)stream, vec, omega, sc, score, (long)P);
  return dfold_check_launch();
}
extern "C" int dfold_rot_head_bwd(const double* g_score, const float* quats_t, const float* quats_0, const float* vec, const float* omega,
                                  const double* sc, const double* dsc, float* d_quats_0, int64_t P, void* stream) {
  if (!g_score || !quats_t || !quats_0 || !vec || !omega || !sc || !dsc || !d_quats_0 || P <= 0) return DFOLD_EINVAL;
  DFOLD_LAUNCH(rot_head_bwd_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g_score, quats_t, quats_0, vec,
               omega, sc, dsc, d_quats_0, (long)P);
  return dfold_check_launch();
}
