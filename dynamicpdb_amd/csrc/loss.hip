// Training loss of the reference (Experiment.loss_fn, train_DFOLD_dynamics.py:1182-1400, live terms only; torsion term
// openfold/utils/loss.py:52-76 as called at :1219-1224) for the LAST frame of every window -- the only frame the live terms
// read -- values and gradients in one launch (round 6).  As an aten graph the loss was ~55 launches forward and ~80 backward
// on [B,F,N,..] tensors whose every frame but one is multiplied by nothing: launch-bound glue between the last forward kernel
// and the first backward kernel of the step.
//
// Per window w (last frame; F frames, nb_w of them with any residue):  coef_w = F / (nb_w + 1e-10)
//   torsion_w = w_tor * sum_{n,k} min(|a^ - g|^2, |a^ - h|^2) m_nk / (sum m + 1e-2),   a^ = a / (|a| + 1e-8)
//   trans_w   = w_tr  * mean_{n,c} (x0_gt - x0_pred)^2
//   rot_w     = w_rot * sum_{n,c} (s_gt - s_pred d_n)^2 l_n / rss_w^2 / (sum_n l_n + 1e-10) * [t_w > thr]
//   gate_w = [trans_w < 100];  final_w = coef_w (gate rot_w + gate trans_w + torsion_w);  loss = mean_w final_w
// (the repetition of the per-window value over the F frames and the division by the number of live frames, :1388-1396, is
// coef_w).  One workgroup per window: sums in double, then the gradients of `loss` w.r.t. the predicted torsions (fp32),
// translations (fp32) and rotation scores (double, as the IGSO(3) score head produces them).
#include "dfold_common.h"
#include "../../include/dfold_hip.h"
#include <math.h>

__device__ __forceinline__ double loss_block_sum(double v, double* sh, int tid) {
  v = wave_sum_d(v);
  __syncthreads();
  if ((tid & 63) == 0) sh[tid >> 6] = v;
  __syncthreads();
  double s = 0.0;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += sh[i];
  return s;
}

__global__ __launch_bounds__(256) void loss_last_frame_kernel(
    const float* __restrict__ ang, const float* __restrict__ ang_gt, const float* __restrict__ ang_alt, const float* __restrict__ tmask,
    const float* __restrict__ tr_pred, const float* __restrict__ tr_gt, const double* __restrict__ rot_pred,
    const double* __restrict__ rot_gt, const float* __restrict__ dmask, const float* __restrict__ lmask,
    const double* __restrict__ rss, const float* __restrict__ t, const float* __restrict__ live_frames, double* __restrict__ terms,
    float* __restrict__ d_ang, float* __restrict__ d_tr, double* __restrict__ d_rot, int B, int N, int F, float w_tr, float w_rot,
    float w_tor, float rot_t_thr) {
  __shared__ double sh[4];
  const int w = blockIdx.x, tid = threadIdx.x;
  const long a0 = (long)w * N * 14, m0 = (long)w * N * 7, x0 = (long)w * N * 3, n0 = (long)w * N;
  // ---- sums ----
  double s_t = 0.0, m_t = 0.0, s_x = 0.0, s_r = 0.0, m_l = 0.0;
  for (int i = tid; i < N * 7; i += blockDim.x) {
    const float x = ang[a0 + 2 * i], y = ang[a0 + 2 * i + 1];
    const float r = sqrtf(x * x + y * y), inv = 1.f / (r + 1e-8f);
    const float ax = x * inv, ay = y * inv;
    const float gx = ax - ang_gt[a0 + 2 * i], gy = ay - ang_gt[a0 + 2 * i + 1];
    const float hx = ax - ang_alt[a0 + 2 * i], hy = ay - ang_alt[a0 + 2 * i + 1];
    const float dg = gx * gx + gy * gy, dh = hx * hx + hy * hy;
    const float mk = tmask[m0 + i];
    s_t += (double)(fminf(dg, dh) * mk);
    m_t += (double)mk;
  }
  const double rs = rss[w];
  for (int i = tid; i < N * 3; i += blockDim.x) {
    const float d = tr_gt[x0 + i] - tr_pred[x0 + i];
    s_x += (double)(d * d);
    const int n = i / 3;
    const double dr = rot_gt[x0 + i] - rot_pred[x0 + i] * (double)dmask[n0 + n];
    s_r += dr * dr * (double)lmask[n0 + n] / (rs * rs);
  }
  for (int n = tid; n < N; n += blockDim.x) m_l += (double)lmask[n0 + n];
  s_t = loss_block_sum(s_t, sh, tid);
  m_t = loss_block_sum(m_t, sh, tid);
  s_x = loss_block_sum(s_x, sh, tid);
  s_r = loss_block_sum(s_r, sh, tid);
  m_l = loss_block_sum(m_l, sh, tid);
  const double coef = (double)F / ((double)live_frames[w] + 1e-10);
  const double tors = (double)w_tor * s_t / (m_t + 1e-2);
  const float trans_f = (float)((double)w_tr * s_x / (3.0 * N));          // (the gate compares the fp32 value, as the aten graph does)
  const double gate = trans_f < 100.f ? 1.0 : 0.0;
  const double tsel = t[w] > rot_t_thr ? 1.0 : 0.0;
  const double rot = (double)w_rot * s_r / (m_l + 1e-10) * tsel * gate;
  const double trans_g = (double)trans_f * gate;
  if (tid == 0) {
    terms[4 * w + 0] = coef * (rot + trans_g + tors);
    terms[4 * w + 1] = coef * rot;
    terms[4 * w + 2] = coef * trans_g;
    terms[4 * w + 3] = coef * tors;
  }
  // ---- gradients of loss = mean_w final_w ----
  const double cw = coef / (double)B;
  const float k_t = (float)(cw * (double)w_tor / (m_t + 1e-2));
  for (int i = tid; i < N * 7; i += blockDim.x) {
    const float x = ang[a0 + 2 * i], y = ang[a0 + 2 * i + 1];
    const float r = sqrtf(x * x + y * y), inv = 1.f / (r + 1e-8f);
    const float ax = x * inv, ay = y * inv;
    const float gx = ax - ang_gt[a0 + 2 * i], gy = ay - ang_gt[a0 + 2 * i + 1];
    const float hx = ax - ang_alt[a0 + 2 * i], hy = ay - ang_alt[a0 + 2 * i + 1];
    const float dg = gx * gx + gy * gy, dh = hx * hx + hy * hy;
    // d min / d a^: the smaller branch; equal distances: half of each (torch.minimum's rule)
    float vx, vy;
    if (dg < dh) {
      vx = 2.f * gx; vy = 2.f * gy;
    } else if (dh < dg) {
      vx = 2.f * hx; vy = 2.f * hy;
    } else {
      vx = gx + hx; vy = gy + hy;
    }
    const float s = k_t * tmask[m0 + i];
    vx *= s;
    vy *= s;
    // a^ = a / (r + eps):  J^T v = v / (r + eps) - a (a . v) / (r (r + eps)^2);  r = 0: the norm's subgradient is 0
    const float av = x * vx + y * vy;
    const float c2 = r > 0.f ? av * inv * inv / r : 0.f;
    d_ang[a0 + 2 * i] = vx * inv - x * c2;
    d_ang[a0 + 2 * i + 1] = vy * inv - y * c2;
  }
  const float k_x = (float)(cw * gate * (double)w_tr * 2.0 / (3.0 * N));
  const double k_r = cw * gate * tsel * (double)w_rot * 2.0 / ((m_l + 1e-10) * rs * rs);
  for (int i = tid; i < N * 3; i += blockDim.x) {
    d_tr[x0 + i] = k_x * (tr_pred[x0 + i] - tr_gt[x0 + i]);
    const int n = i / 3;
    const double dm = (double)dmask[n0 + n];
    d_rot[x0 + i] = k_r * (rot_pred[x0 + i] * dm - rot_gt[x0 + i]) * dm * (double)lmask[n0 + n];
  }
}

extern "C" int dfold_loss_last_frame(const float* ang, const float* ang_gt, const float* ang_alt, const float* ang_mask,
                                     const float* trans_pred, const float* trans_gt, const double* rot_pred, const double* rot_gt,
                                     const float* diffuse_mask, const float* loss_mask, const double* rot_score_scaling, const float* t,
                                     const float* live_frames, double* terms, float* d_ang, float* d_trans, double* d_rot, int32_t B,
                                     int32_t N, int32_t F, float trans_weight, float rot_weight, float torsion_weight,
                                     float rot_t_threshold, void* stream) {
  if (!ang || !ang_gt || !ang_alt || !ang_mask || !trans_pred || !trans_gt || !rot_pred || !rot_gt || !diffuse_mask || !loss_mask ||
      !rot_score_scaling || !t || !live_frames || !terms || !d_ang || !d_trans || !d_rot || B <= 0 || N <= 0 || F <= 0)
    return DFOLD_EINVAL;
  DFOLD_LAUNCH(loss_last_frame_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, ang, ang_gt, ang_alt, ang_mask, trans_pred,
               trans_gt, rot_pred, rot_gt, diffuse_mask, loss_mask, rot_score_scaling, t, live_frames, terms, d_ang, d_trans, d_rot, B, N,
               F, trans_weight, rot_weight, torsion_weight, rot_t_threshold);
  return dfold_check_launch();
}
