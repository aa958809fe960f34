// MyLayerNorm (reference src/model/ipa_pytorch_dynamic.py:709-724): statistics over the WHOLE
// [F,N,C] tensor of one window, unbiased variance, eps inside the sqrt, no affine -- optionally fused
// with the SiLU that follows it inside every embedder (:757-796).  HBM-bound: one reduction pass
// (fp64 partial sums, one atomic pair per workgroup) + one vectorised apply pass.
#include "dfold_common.h"
#include "../../include/dfold_hip.h"

__device__ __forceinline__ float silu_f(float y) { return y / (1.f + expf(-y)); }
__device__ __forceinline__ float silu_grad_f(float y) {
  const float s = 1.f / (1.f + expf(-y));
  return s * (1.f + y * (1.f - s));
}

// stats[w] = {sum x, sum x^2} (double), must be zeroed by the caller.  x fp32 [W][n]
__global__ __launch_bounds__(256) void gln_stats_kernel(const float* __restrict__ x, double* __restrict__ stats, long n) {
  __shared__ double red[2][4];
  const int w = blockIdx.y;
  const float* xw = x + (long)w * n;
  double s = 0.0, s2 = 0.0;
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long)gridDim.x * 1024) {
    if (i + 3 < n) {
      const float4 v = *(const float4*)(xw + i);
      s += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
      s2 += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    } else {
      for (long k = i; k < n; ++k) {
        s += xw[k];
        s2 += (double)xw[k] * xw[k];
      }
    }
  }
  s = wave_sum_d(s);
  s2 = wave_sum_d(s2);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) {
    red[0][wv] = s;
    red[1][wv] = s2;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(stats + 2 * w, red[0][0] + red[0][1] + red[0][2] + red[0][3]);
    atomicAdd(stats + 2 * w + 1, red[1][0] + red[1][1] + red[1][2] + red[1][3]);
  }
}

// y = (x - mean) * rstd  [-> silu];  out bf16.  mr[w] = {mean, rstd} (float) is written for backward.
__global__ __launch_bounds__(256) void gln_apply_kernel(const float* __restrict__ x, const double* __restrict__ stats,
                                                        bf16_t* __restrict__ y, float* __restrict__ mr, long n, float eps,
                                                        int silu) {
  const int w = blockIdx.y;
  const double mean_d = stats[2 * w] / (double)n;
  const double var_d = (stats[2 * w + 1] - (double)n * mean_d * mean_d) / (double)(n - 1);
  const float mean = (float)mean_d;
  const float rstd = (float)(1.0 / sqrt(var_d + (double)eps));
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    mr[2 * w] = mean;
    mr[2 * w + 1] = rstd;
  }
  const float* xw = x + (long)w * n;
  bf16_t* yw = y + (long)w * n;
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long)gridDim.x * 1024) {
    if (i + 3 < n) {
      const float4 v = *(const float4*)(xw + i);
      float a = (v.x - mean) * rstd, b = (v.y - mean) * rstd, c = (v.z - mean) * rstd, d = (v.w - mean) * rstd;
      if (silu) {
        a = silu_f(a); b = silu_f(b); c = silu_f(c); d = silu_f(d);
      }
      uint2 o;
      o.x = pack2bf(a, b);
      o.y = pack2bf(c, d);
      *(uint2*)(yw + i) = o;
    } else {
      for (long k = i; k < n; ++k) {
        float a = (xw[k] - mean) * rstd;
        yw[k] = f2bf(silu ? silu_f(a) : a);
      }
    }
  }
}

extern "C" int dfold_gln_fwd(const float* x, double* stats, void* y_bf16, float* mean_rstd, int32_t W, int64_t n, float eps,
                             int32_t silu, void* stream) {
  if (!x || !stats || !y_bf16 || !mean_rstd || W <= 0 || n < 2 || (n & 3)) return DFOLD_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(stats, 0, sizeof(double) * 2 * W, st) != hipSuccess) return DFOLD_ELAUNCH;
  long bx = (n / 4 + 255) / 256;
  if (bx > 512) bx = 512;
  dim3 grid((unsigned)bx, W);
  // the two fp64 atomics per workgroup all land on the window's stats pair and serialise in L2: ~1024 workgroups in all
  long sx = bx;
  const long cap = 1024 / W > 0 ? 1024 / W : 1;
  if (sx > cap) sx = cap;
  DFOLD_LAUNCH(gln_stats_kernel, dim3((unsigned)sx, W), dim3(256), 0, st, x, stats, (long)n);
  DFOLD_LAUNCH(gln_apply_kernel, grid, dim3(256), 0, st, x, (const double*)stats, (bf16_t*)y_bf16, mean_rstd, (long)n,
                     eps, silu);
  return dfold_check_launch();
}

// backward: with y = (x-mean)*rstd, out = silu ? silu(y) : y, g = dL/dout (bf16):
//   gy = g * (silu ? silu'(y) : 1);   dx = rstd * (gy - mean(gy) - y * sum(gy*y)/(n-1))
__global__ __launch_bounds__(256) void gln_bwd_stats_kernel(const float* __restrict__ x, const bf16_t* __restrict__ g,
                                                            const float* __restrict__ mr, double* __restrict__ stats, long n,
                                                            int silu) {
  __shared__ double red[2][4];
  const int w = blockIdx.y;
  const float mean = mr[2 * w], rstd = mr[2 * w + 1];
  const float* xw = x + (long)w * n;
  const bf16_t* gw = g + (long)w * n;
  double s = 0.0, s2 = 0.0;
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long)gridDim.x * 1024) {
    if (i + 3 < n) {
      const float4 xv = *(const float4*)(xw + i);
      const uint2 gv = *(const uint2*)(gw + i);
      const float xe[4] = {xv.x, xv.y, xv.z, xv.w};
      const float ge[4] = {bf_lo(gv.x), bf_hi(gv.x), bf_lo(gv.y), bf_hi(gv.y)};
      float ps = 0.f, ps2 = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float y = (xe[j] - mean) * rstd;
        float gy = ge[j];
        if (silu) gy *= silu_grad_f(y);
        ps += gy;
        ps2 += gy * y;
      }
      s += ps;
      s2 += ps2;
    } else {
      for (long k = i; k < n; ++k) {
        const float y = (xw[k] - mean) * rstd;
        float gy = bf2f(gw[k]);
        if (silu) gy *= silu_grad_f(y);
        s += gy;
        s2 += (double)gy * y;
      }
    }
  }
  s = wave_sum_d(s);
  s2 = wave_sum_d(s2);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) {
    red[0][wv] = s;
    red[1][wv] = s2;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(stats + 2 * w, red[0][0] + red[0][1] + red[0][2] + red[0][3]);
    atomicAdd(stats + 2 * w + 1, red[1][0] + red[1][1] + red[1][2] + red[1][3]);
  }
}

__global__ __launch_bounds__(256) void gln_bwd_apply_kernel(const float* __restrict__ x, const bf16_t* __restrict__ g,
                                                            const float* __restrict__ mr, const double* __restrict__ stats,
                                                            bf16_t* __restrict__ dx, long n, int silu) {
  const int w = blockIdx.y;
  const float mean = mr[2 * w], rstd = mr[2 * w + 1];
  const float mg = (float)(stats[2 * w] / (double)n);
  const float cg = (float)(stats[2 * w + 1] / (double)(n - 1));
  const float* xw = x + (long)w * n;
  const bf16_t* gw = g + (long)w * n;
  bf16_t* dw = dx + (long)w * n;
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long)gridDim.x * 1024) {
    if (i + 3 < n) {
      const float4 xv = *(const float4*)(xw + i);
      const uint2 gv = *(const uint2*)(gw + i);
      const float xe[4] = {xv.x, xv.y, xv.z, xv.w};
      const float ge[4] = {bf_lo(gv.x), bf_hi(gv.x), bf_lo(gv.y), bf_hi(gv.y)};
      float o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float y = (xe[j] - mean) * rstd;
        float gy = ge[j];
        if (silu) gy *= silu_grad_f(y);
        o[j] = rstd * (gy - mg - y * cg);
      }
      uint2 ov;
      ov.x = pack2bf(o[0], o[1]);
      ov.y = pack2bf(o[2], o[3]);
      *(uint2*)(dw + i) = ov;
    } else {
      for (long k = i; k < n; ++k) {
        const float y = (xw[k] - mean) * rstd;
        float gy = bf2f(gw[k]);
        if (silu) gy *= silu_grad_f(y);
        dw[k] = f2bf(rstd * (gy - mg - y * cg));
      }
    }
  }
}

extern "C" int dfold_gln_bwd(const float* x, const void* g_bf16, const float* mean_rstd, double* stats, void* dx_bf16,
                             int32_t W, int64_t n, int32_t silu, void* stream) {
  if (!x || !g_bf16 || !mean_rstd || !stats || !dx_bf16 || W <= 0 || n < 2 || (n & 3)) return DFOLD_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(stats, 0, sizeof(double) * 2 * W, st) != hipSuccess) return DFOLD_ELAUNCH;
  long bx = (n / 4 + 255) / 256;
  if (bx > 512) bx = 512;
  dim3 grid((unsigned)bx, W);
  long sx = bx;
  const long cap = 1024 / W > 0 ? 1024 / W : 1;
  if (sx > cap) sx = cap;
  DFOLD_LAUNCH(gln_bwd_stats_kernel, dim3((unsigned)sx, W), dim3(256), 0, st, x, (const bf16_t*)g_bf16, mean_rstd, stats, (long)n, silu);
  DFOLD_LAUNCH(gln_bwd_apply_kernel, grid, dim3(256), 0, st, x, (const bf16_t*)g_bf16, mean_rstd,
                     (const double*)stats, (bf16_t*)dx_bf16, (long)n, silu);
  return dfold_check_launch();
}
