// One reverse-SDE (denoise) step of the SE(3) diffusion on device (reference SE3Diffuser.reverse,
// src/data/se3_diffuser.py:160-215 -> SO3Diffuser.reverse src/data/so3_diffuser.py:329-365 (geodesic random walk,
// right-multiplied) and R3Diffuser.reverse src/data/r3_diffuser.py:106-157 (Euler-Maruyama + per-frame centring)).
// The reference does this on the host per sampling step: device->host copies, scipy matrix<->rotvec round trips,
// numpy RNG, host->device copy and a CPU eigh to get quaternions back (openfold/utils/rigid_utils.py:226).  Here the
// step is one launch on tensor_7 frames: the rotation update is a quaternion product q_t (x) exp(perturb) (identical
// to R(rot_t) R(perturb)), arithmetic in fp64 like the reference's numpy path, the normal draws z are INPUTS (host
// numpy draws for parity with the reference's RNG stream, or device draws for speed).
// One workgroup per (window, frame) row: the centre of mass over residues is a block reduction.
#include "dfold_common.h"
#include "../../include/dfold_hip.h"

__global__ __launch_bounds__(256) void se3_reverse_kernel(const float* __restrict__ t7, const double* __restrict__ rot_score,
                                                          const float* __restrict__ trans_score,
                                                          const double* __restrict__ z_rot, const double* __restrict__ z_trans,
                                                          const float* __restrict__ mask, float* __restrict__ out, int N,
                                                          double g_rot, double b_t, double dt, double noise_scale, double cs,
                                                          int center) {
  extern __shared__ double xs[];  // [N][3] perturbed (scaled) translations
  __shared__ double red[4][3];
  const long row = blockIdx.x;
  const double sdt = sqrt(dt), g_r3 = sqrt(b_t);
  double s0 = 0, s1 = 0, s2 = 0;
  for (int n = threadIdx.x; n < N; n += blockDim.x) {
    const long p = row * N + n;
    // ---- rotation: q' = normalize(q) (x) exp(perturb) ----
    double q[4];
    double qn = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      q[c] = t7[p * 7 + c];
      qn += q[c] * q[c];
    }
    qn = 1.0 / sqrt(qn);
    double v[3];
    double th2 = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      v[c] = g_rot * g_rot * rot_score[p * 3 + c] * dt + g_rot * sdt * noise_scale * z_rot[p * 3 + c];
      th2 += v[c] * v[c];
    }
    const double th = sqrt(th2);
    const double sh = th > 1e-12 ? sin(0.5 * th) / th : 0.5;   // sin(th/2)/th
    const double e0 = cos(0.5 * th), e1 = sh * v[0], e2 = sh * v[1], e3 = sh * v[2];
    const double a = q[0] * qn, b = q[1] * qn, c_ = q[2] * qn, d = q[3] * qn;
    double o[4];
    o[0] = a * e0 - b * e1 - c_ * e2 - d * e3;
    o[1] = a * e1 + b * e0 + c_ * e3 - d * e2;
    o[2] = a * e2 - b * e3 + c_ * e0 + d * e1;
    o[3] = a * e3 + b * e2 - c_ * e1 + d * e0;
    const bool keep = mask != nullptr && mask[p] == 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) out[p * 7 + c] = keep ? t7[p * 7 + c] : (float)o[c];
    // ---- translation ----
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const double x = cs * (double)t7[p * 7 + 4 + c];
      const double f = -0.5 * b_t * x;
      const double pert = (f - g_r3 * g_r3 * (double)trans_score[p * 3 + c]) * dt + g_r3 * sdt * noise_scale * z_trans[p * 3 + c];
      xs[n * 3 + c] = x - pert;
    }
    s0 += xs[n * 3];
    s1 += xs[n * 3 + 1];
    s2 += xs[n * 3 + 2];
  }
  s0 = wave_sum_d(s0);
  s1 = wave_sum_d(s1);
  s2 = wave_sum_d(s2);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) {
    red[w][0] = s0;
    red[w][1] = s1;
    red[w][2] = s2;
  }
  __syncthreads();
  double com[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) com[c] = center ? (red[0][c] + red[1][c] + red[2][c] + red[3][c]) / (double)N : 0.0;
  for (int n = threadIdx.x; n < N; n += blockDim.x) {
    const long p = row * N + n;
    const bool keep = mask != nullptr && mask[p] == 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) out[p * 7 + 4 + c] = keep ? t7[p * 7 + 4 + c] : (float)((xs[n * 3 + c] - com[c]) / cs);
  }
}

extern "C" int dfold_se3_reverse(const float* t7, const double* rot_score, const float* trans_score, const double* z_rot,
                                 const double* z_trans, const float* mask, float* out, int64_t rows, int32_t N, double g_rot,
                                 double b_t, double dt, double noise_scale, double coordinate_scaling, int32_t center,
                                 void* stream) {
  if (!t7 || !rot_score || !trans_score || !z_rot || !z_trans || !out || rows <= 0 || N <= 0 || N > 6000) return DFOLD_EINVAL;
  if (!(dt > 0) || !(coordinate_scaling > 0) || !(b_t >= 0)) return DFOLD_EINVAL;
  DFOLD_LAUNCH(se3_reverse_kernel, dim3((unsigned)rows), dim3(256), (size_t)N * 3 * sizeof(double), (hipStream_t)stream, t7,
               rot_score, trans_score, z_rot, z_trans, mask, out, N, g_rot, b_t, dt, noise_scale, coordinate_scaling, center);
  return dfold_check_launch();
}
