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
