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

__global__ __launch_bounds__(256) void igso3_series_kernel(const float* __restrict__ omega, const double* __restrict__ env,
                                                           double* __restrict__ sc, double* __restrict__ dsc, long P,
                                                           long per_window, int L) {
  extern __shared__ double envs[];
  const int w = blockIdx.y;
  for (int l = threadIdx.x; l < L; l += 256) envs[l] = env[(long)w * L + l];
  __syncthreads();
  const long loc = (long)blockIdx.x * 256 + threadIdx.x;
  if (loc >= per_window) return;
  const long p = (long)w * per_window + loc;
  if (p >= P) return;
  const float om = omega[p];
  const float lo = sinf(om * 0.5f);
  const float dlo = 0.5f * cosf(om * 0.5f);
  const float lo2 = lo * lo;
  double f = 0.0, ds = 0.0, dds = 0.0;
  for (int l = 0; l < L; ++l) {
    const float lh = (float)l + 0.5f;
    const float arg = om * lh;
    const float hi = sinf(arg);
    const float dhi = lh * cosf(arg);
    const double e = envs[l];
    f += e * (double)hi / (double)lo;
    const float num = lo * dhi - hi * dlo;
    ds += e * (double)num / (double)lo2;
    // d/dw (num / lo^2) = lo*hi*(1/4 - lh^2)/lo^2 - 2 num dlo / lo^3
    const double nprime = (double)lo * (double)hi * (0.25 - (double)lh * (double)lh);
    dds += e * (nprime / (double)lo2 - 2.0 * (double)num * (double)dlo / ((double)lo2 * (double)lo));
  }
  const double den = f + 1e-4;
  sc[p] = ds / den;
  dsc[p] = (dds * den - ds * ds) / (den * den);
}

extern "C" int dfold_igso3_series(const float* omega, const double* env, double* sc, double* dsc, int64_t P,
                                  int64_t per_window, int32_t L, void* stream) {
  if (!omega || !env || !sc || !dsc || P <= 0 || per_window <= 0 || L <= 0 || L > 4096) return DFOLD_EINVAL;
  if (P % per_window) return DFOLD_EINVAL;
  dim3 grid((unsigned)((per_window + 255) / 256), (unsigned)(P / per_window));
  DFOLD_LAUNCH(igso3_series_kernel, grid, dim3(256), (size_t)L * sizeof(double), (hipStream_t)stream, omega, env, sc,
                     dsc, (long)P, (long)per_window, L);
  return dfold_check_launch();
}
