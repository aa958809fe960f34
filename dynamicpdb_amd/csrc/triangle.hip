// Pointwise / row kernels of the triangle pair operators (reference openfold/model/
// triangular_multiplicative_update.py:26-126 and triangular_attention.py:31-139, primitives.py:180-243).
// All of them are HBM-bound passes over the [N*N, c] pair tensor: one wave per pair-row for the LayerNorms
// and the softmax (wave64 shuffle reductions, no LDS), 16-byte vector access for the gates.
// The dense parts (projections, the ik,jk->ij contraction, q.k / a.v products) run on the MFMA engine.
#include "dfold_common.h"
#include "../../include/dfold_hip.h"

__device__ __forceinline__ float sigm(float y) { return 1.f / (1.f + expf(-y)); }

#define LN_MAXE 8  // elements per lane: C <= 512

// ---------------------------------------------------------------------------------------------
// Row LayerNorm with affine (F.layer_norm, eps inside sqrt, biased variance), one wave per row.
//   x: fp32 or bf16 [R][C] -> y bf16 [R][C]; stats[r] = {mean, rstd}
// ---------------------------------------------------------------------------------------------
template <bool XBF16>
__global__ __launch_bounds__(256) void row_ln_fwd_kernel(const void* __restrict__ xv, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, bf16_t* __restrict__ y,
                                                         float* __restrict__ stats, long R, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long nw = ((long)gridDim.x * blockDim.x) >> 6;
  for (long r = wave; r < R; r += nw) {
    float v[LN_MAXE];
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < LN_MAXE; ++e) {
      const int c = lane + 64 * e;
      v[e] = 0.f;
      if (c < C) {
        v[e] = XBF16 ? bf2f(((const bf16_t*)xv)[r * C + c]) : ((const float*)xv)[r * C + c];
        s += v[e];
      }
    }
    const float mean = wave_sum(s) / C;
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < LN_MAXE; ++e)
      if (lane + 64 * e < C) q += (v[e] - mean) * (v[e] - mean);
    const float rstd = rsqrtf(wave_sum(q) / C + eps);
#pragma unroll
    for (int e = 0; e < LN_MAXE; ++e) {
      const int c = lane + 64 * e;
      if (c < C) y[r * C + c] = f2bf((v[e] - mean) * rstd * gamma[c] + beta[c]);
    }
    if (lane == 0) {
      stats[2 * r] = mean;
      stats[2 * r + 1] = rstd;
    }
  }
}

// C = 128 (the pair tensor's width): 16 lanes per row -- lane l15 holds channels 8 l15 .. + 8 as 16-byte vectors --, four rows
// per wave and pass, two passes in flight; the two reductions of a row are 4 DPP steps each and serve four rows at once
// (one wave per row with 4-byte accesses and two 6-step wave reductions per row ran the LayerNorm_in backward of the triangle
// multiplication at 1.9 TB/s, profiles/r5_trimul_bwd_kernel_stats.csv).
typedef __attribute__((ext_vector_type(4))) unsigned lnu32x4;
__device__ __forceinline__ float ln_row16_sum(float v) {
  v += dpp_mov_f<0xb1>(0.f, v);
  v += dpp_mov_f<0x4e>(0.f, v);
  v += dpp_mov_f<0x124>(0.f, v);
  v += dpp_mov_f<0x128>(0.f, v);
  return v;
}
template <bool XBF16>
__device__ __forceinline__ void ln128_load(const void* xv, long r, int l15, float (&x)[8]) {
  if (XBF16) {
    const lnu32x4 u = *(const lnu32x4*)((const bf16_t*)xv + r * 128 + l15 * 8);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      x[2 * k] = bf_lo(u[k]);
      x[2 * k + 1] = bf_hi(u[k]);
    }
  } else {
    const f32x4 a = *(const f32x4*)((const float*)xv + r * 128 + l15 * 8), b = *(const f32x4*)((const float*)xv + r * 128 + l15 * 8 + 4);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      x[k] = a[k];
      x[4 + k] = b[k];
    }
  }
}

template <bool XBF16>
__global__ __launch_bounds__(256) void row_ln_fwd128_kernel(const void* __restrict__ xv, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, bf16_t* __restrict__ y,
                                                            float* __restrict__ stats, long R, float eps) {
  const int lane = threadIdx.x & 63, l15 = lane & 15, l4 = lane >> 4;
  const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long nw = ((long)gridDim.x * blockDim.x) >> 6;
  float gm[8], bt[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    gm[k] = gamma[l15 * 8 + k];
    bt[k] = beta[l15 * 8 + k];
  }
  for (long r0 = wave * 8; r0 < R; r0 += nw * 8) {
    float x[2][8];
    long rr[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      rr[u] = r0 + u * 4 + l4;
      ln128_load<XBF16>(xv, rr[u] < R ? rr[u] : R - 1, l15, x[u]);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      float (&v)[8] = x[u];
      const float mean = ln_row16_sum(((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]))) * (1.f / 128.f);
      float q = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        v[k] -= mean;
        q = __builtin_fmaf(v[k], v[k], q);
      }
      const float rstd = rsqrtf(ln_row16_sum(q) * (1.f / 128.f) + eps);
      if (rr[u] < R) {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = __builtin_fmaf(v[k] * rstd, gm[k], bt[k]);
        *(lnu32x4*)(y + rr[u] * 128 + l15 * 8) = (lnu32x4){pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])};
        if (l15 == 0) {
          stats[2 * rr[u]] = mean;
          stats[2 * rr[u] + 1] = rstd;
        }
      }
    }
  }
}

template <bool XBF16, bool DXBF16>
__global__ __launch_bounds__(256) void row_ln_bwd128_kernel(const void* __restrict__ xv, const float* __restrict__ stats,
                                                            const float* __restrict__ gamma, const bf16_t* __restrict__ g,
                                                            void* __restrict__ dxv, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, long R) {
  __shared__ float red[4][2][128];
  const int lane = threadIdx.x & 63, l15 = lane & 15, l4 = lane >> 4, w = threadIdx.x >> 6;
  const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long nw = ((long)gridDim.x * blockDim.x) >> 6;
  float gm[8], ag[8], ab[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    gm[k] = gamma[l15 * 8 + k];
    ag[k] = ab[k] = 0.f;
  }
  for (long r0 = wave * 8; r0 < R; r0 += nw * 8) {
    float x[2][8], gy[2][8], mean[2], rstd[2];
    long rr[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      rr[u] = r0 + u * 4 + l4;
      const long rc = rr[u] < R ? rr[u] : R - 1;
      ln128_load<XBF16>(xv, rc, l15, x[u]);
      ln128_load<true>(g, rc, l15, gy[u]);
      mean[u] = stats[2 * rc];
      rstd[u] = stats[2 * rc + 1];
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const bool live = rr[u] < R;
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float xh = (x[u][k] - mean[u]) * rstd[u];
        const float gg = live ? gy[u][k] : 0.f;
        ag[k] = __builtin_fmaf(gg, xh, ag[k]);
        ab[k] += gg;
        x[u][k] = xh;
        gy[u][k] = gg * gm[k];
        s1 += gy[u][k];
        s2 = __builtin_fmaf(gy[u][k], xh, s2);
      }
      s1 = ln_row16_sum(s1) * (1.f / 128.f);
      s2 = ln_row16_sum(s2) * (1.f / 128.f);
      if (live) {
        float d[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) d[k] = rstd[u] * (gy[u][k] - s1 - x[u][k] * s2);
        if (DXBF16) {
          *(lnu32x4*)((bf16_t*)dxv + rr[u] * 128 + l15 * 8) = (lnu32x4){pack2bf(d[0], d[1]), pack2bf(d[2], d[3]), pack2bf(d[4], d[5]), pack2bf(d[6], d[7])};
        } else {
          *(f32x4*)((float*)dxv + rr[u] * 128 + l15 * 8) = (f32x4){d[0], d[1], d[2], d[3]};
          *(f32x4*)((float*)dxv + rr[u] * 128 + l15 * 8 + 4) = (f32x4){d[4], d[5], d[6], d[7]};
        }
      }
    }
  }
  // the four l4 groups of a wave hold the same channels; then the block's four waves; one atomic per channel and block
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    float a = ag[k], b = ab[k];
    a += __shfl_xor(a, 16, 64);
    a += __shfl_xor(a, 32, 64);
    b += __shfl_xor(b, 16, 64);
    b += __shfl_xor(b, 32, 64);
    if (l4 == 0) {
      red[w][0][l15 * 8 + k] = a;
      red[w][1][l15 * 8 + k] = b;
    }
  }
  __syncthreads();
  const int c = threadIdx.x & 127, which = threadIdx.x >> 7;
  atomicAdd((which ? dbeta : dgamma) + c, (red[0][which][c] + red[1][which][c]) + (red[2][which][c] + red[3][which][c]));
}

extern "C" int dfold_row_ln_fwd(const void* x, int32_t x_is_bf16, const float* gamma, const float* beta, void* y_bf16,
                                float* stats, int64_t R, int32_t C, float eps, void* stream) {
  if (!x || !gamma || !beta || !y_bf16 || !stats || R <= 0 || C <= 0 || C > 64 * LN_MAXE) return DFOLD_EINVAL;
  if (C == 128 && (((uintptr_t)x | (uintptr_t)y_bf16) & 15) == 0) {
    long blocks128 = (R + 31) / 32;
    if (blocks128 > 2048) blocks128 = 2048;
    if (x_is_bf16)
      DFOLD_LAUNCH(row_ln_fwd128_kernel<true>, dim3((unsigned)blocks128), dim3(256), 0, (hipStream_t)stream, x, gamma, beta,
                   (bf16_t*)y_bf16, stats, (long)R, eps);
    else
      DFOLD_LAUNCH(row_ln_fwd128_kernel<false>, dim3((unsigned)blocks128), dim3(256), 0, (hipStream_t)stream, x, gamma, beta,
                   (bf16_t*)y_bf16, stats, (long)R, eps);
    return dfold_check_launch();
  }
  long blocks = (R + 3) / 4;
  if (blocks > 4096) blocks = 4096;
  if (x_is_bf16)
    DFOLD_LAUNCH(row_ln_fwd_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, gamma, beta,
                 (bf16_t*)y_bf16, stats, (long)R, C, eps);
  else
    DFOLD_LAUNCH(row_ln_fwd_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, gamma, beta,
                 (bf16_t*)y_bf16, stats, (long)R, C, eps);
  return dfold_check_launch();
}

// backward: g = dL/dy (bf16).  dx = rstd * (g*gamma - mean(g*gamma) - xhat * mean(g*gamma*xhat));
// dgamma += sum_r g*xhat, dbeta += sum_r g (fp32 atomics, caller zeroes).  dx: fp32 or bf16.
template <bool XBF16, bool DXBF16>
__global__ __launch_bounds__(256) void row_ln_bwd_kernel(const void* __restrict__ xv, const float* __restrict__ stats,
                                                         const float* __restrict__ gamma, const bf16_t* __restrict__ g,
                                                         void* __restrict__ dxv, float* __restrict__ dgamma,
                                                         float* __restrict__ dbeta, long R, int C) {
  const int lane = threadIdx.x & 63;
  const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long nw = ((long)gridDim.x * blockDim.x) >> 6;
  float ag[LN_MAXE], ab[LN_MAXE], gm[LN_MAXE];
#pragma unroll
  for (int e = 0; e < LN_MAXE; ++e) {
    ag[e] = ab[e] = 0.f;
    gm[e] = (lane + 64 * e < C) ? gamma[lane + 64 * e] : 0.f;
  }
  for (long r = wave; r < R; r += nw) {
    const float mean = stats[2 * r], rstd = stats[2 * r + 1];
    float xh[LN_MAXE], gy[LN_MAXE];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int e = 0; e < LN_MAXE; ++e) {
      const int c = lane + 64 * e;
      xh[e] = gy[e] = 0.f;
      if (c < C) {
        const float xv_ = XBF16 ? bf2f(((const bf16_t*)xv)[r * C + c]) : ((const float*)xv)[r * C + c];
        const float gg = bf2f(g[r * C + c]);
        xh[e] = (xv_ - mean) * rstd;
        gy[e] = gg * gm[e];
        ag[e] += gg * xh[e];
        ab[e] += gg;
        s1 += gy[e];
        s2 += gy[e] * xh[e];
      }
    }
    s1 = wave_sum(s1) / C;
    s2 = wave_sum(s2) / C;
#pragma unroll
    for (int e = 0; e < LN_MAXE; ++e) {
      const int c = lane + 64 * e;
      if (c < C) {
        const float d = rstd * (gy[e] - s1 - xh[e] * s2);
        if (DXBF16)
          ((bf16_t*)dxv)[r * C + c] = f2bf(d);
        else
          ((float*)dxv)[r * C + c] = d;
      }
    }
  }
#pragma unroll
  for (int e = 0; e < LN_MAXE; ++e) {
    const int c = lane + 64 * e;
    if (c < C) {
      atomicAdd(dgamma + c, ag[e]);
      atomicAdd(dbeta + c, ab[e]);
    }
  }
}

extern "C" int dfold_row_ln_bwd(const void* x, int32_t x_is_bf16, const float* stats, const float* gamma, const void* g_bf16,
                                void* dx, int32_t dx_is_bf16, float* dgamma, float* dbeta, int64_t R, int32_t C,
                                void* stream) {
  if (!x || !stats || !gamma || !g_bf16 || !dx || !dgamma || !dbeta || R <= 0 || C <= 0 || C > 64 * LN_MAXE) return DFOLD_EINVAL;
  long blocks = (R + 3) / 4;
  if (blocks > 1024) blocks = 1024;
  dim3 grid((unsigned)blocks), blk(256);
  hipStream_t st = (hipStream_t)stream;
  const bf16_t* g = (const bf16_t*)g_bf16;
  if (C == 128 && (((uintptr_t)x | (uintptr_t)g_bf16 | (uintptr_t)dx) & 15) == 0) {
    long b128 = (R + 31) / 32;
    if (b128 > 2048) b128 = 2048;
    dim3 g128((unsigned)b128);
    if (x_is_bf16 && dx_is_bf16)
      DFOLD_LAUNCH((row_ln_bwd128_kernel<true, true>), g128, blk, 0, st, x, stats, gamma, g, dx, dgamma, dbeta, (long)R);
    else if (x_is_bf16)
      DFOLD_LAUNCH((row_ln_bwd128_kernel<true, false>), g128, blk, 0, st, x, stats, gamma, g, dx, dgamma, dbeta, (long)R);
    else if (dx_is_bf16)
      DFOLD_LAUNCH((row_ln_bwd128_kernel<false, true>), g128, blk, 0, st, x, stats, gamma, g, dx, dgamma, dbeta, (long)R);
    else
      DFOLD_LAUNCH((row_ln_bwd128_kernel<false, false>), g128, blk, 0, st, x, stats, gamma, g, dx, dgamma, dbeta, (long)R);
    return dfold_check_launch();
  }
  if (x_is_bf16 && dx_is_bf16)
    DFOLD_LAUNCH((row_ln_bwd_kernel<true, true>), grid, blk, 0, st, x, stats, gamma, g, dx, dgamma, dbeta, (long)R, C);
  else if (x_is_bf16)
    DFOLD_LAUNCH((row_ln_bwd_kernel<true, false>), grid, blk, 0, st, x, stats, gamma, g, dx, dgamma, dbeta, (long)R, C);
  else if (dx_is_bf16)
    DFOLD_LAUNCH((row_ln_bwd_kernel<false, true>), grid, blk, 0, st, x, stats, gamma, g, dx, dgamma, dbeta, (long)R, C);
  else
    DFOLD_LAUNCH((row_ln_bwd_kernel<false, false>), grid, blk, 0, st, x, stats, gamma, g, dx, dgamma, dbeta, (long)R, C);
  return dfold_check_launch();
}

// ---------------------------------------------------------------------------------------------
// Triangle multiplication gates (triangular_multiplicative_update.py:97-104).
//   proj bf16 [R][5c] = [a_p | a_g | b_p | b_g | g] -> ab bf16 [R][2c]:  a = a_p*sigmoid(a_g)*mask, b likewise
// backward: dab -> dproj[:, 0:4c]
// ---------------------------------------------------------------------------------------------
__global__ void trimul_gate_fwd_kernel(const bf16_t* __restrict__ proj, const float* __restrict__ mask,
                                       bf16_t* __restrict__ ab, long R, int c) {
  const long total = R * 2 * c;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / (2 * c);
    const int e = (int)(i - r * 2 * c);
    const int half = e / c, cc = e - half * c;
    const bf16_t* pr = proj + r * 5 * c + half * 2 * c;
    ab[i] = f2bf(bf2f(pr[cc]) * sigm(bf2f(pr[c + cc])) * mask[r]);
  }
}

__global__ void trimul_gate_bwd_kernel(const bf16_t* __restrict__ proj, const float* __restrict__ mask,
                                       const bf16_t* __restrict__ dab, bf16_t* __restrict__ dproj, long R, int c) {
  const long total = R * 2 * c;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / (2 * c);
    const int e = (int)(i - r * 2 * c);
    const int half = e / c, cc = e - half * c;
    const bf16_t* pr = proj + r * 5 * c + half * 2 * c;
    bf16_t* dp = dproj + r * 5 * c + half * 2 * c;
    const float p = bf2f(pr[cc]), s = sigm(bf2f(pr[c + cc])), d = bf2f(dab[i]) * mask[r];
    dp[cc] = f2bf(d * s);
    dp[c + cc] = f2bf(d * p * s * (1.f - s));
  }
}

extern "C" int dfold_trimul_gate_fwd(const void* proj, const float* mask, void* ab, int64_t R, int32_t c, void* stream) {
  if (!proj || !mask || !ab || R <= 0 || c <= 0) return DFOLD_EINVAL;
  long blocks = (R * 2 * c + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  DFOLD_LAUNCH(trimul_gate_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)proj, mask,
               (bf16_t*)ab, (long)R, c);
  return dfold_check_launch();
}

extern "C" int dfold_trimul_gate_bwd(const void* proj, const float* mask, const void* dab, void* dproj, int64_t R, int32_t c,
                                     void* stream) {
  if (!proj || !mask || !dab || !dproj || R <= 0 || c <= 0) return DFOLD_EINVAL;
  long blocks = (R * 2 * c + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  DFOLD_LAUNCH(trimul_gate_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)proj, mask,
               (const bf16_t*)dab, (bf16_t*)dproj, (long)R, c);
  return dfold_check_launch();
}

// ---------------------------------------------------------------------------------------------
// Output gate  out = y * sigmoid(g)   (triangular_multiplicative_update.py:122-124; Attention gate primitives.py:385-390)
//   y fp32 [R][c] (row stride ldy); g bf16 taken from a wider row (row stride ldg, column offset applied by caller)
// backward: dy bf16 = dout * sigmoid(g);  dg bf16 (row stride ldg) = dout * y * s(1-s)
// ---------------------------------------------------------------------------------------------
__global__ void gate_mul_fwd_kernel(const float* __restrict__ y, const bf16_t* __restrict__ g, float* __restrict__ out, long R,
                                    int c, long ldg) {
  const long total = R * c;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / c;
    const int cc = (int)(i - r * c);
    out[i] = y[i] * sigm(bf2f(g[r * ldg + cc]));
  }
}

__global__ void gate_mul_bwd_kernel(const float* __restrict__ y, const bf16_t* __restrict__ g, const float* __restrict__ dout,
                                    bf16_t* __restrict__ dy, bf16_t* __restrict__ dg, long R, int c, long ldg) {
  const long total = R * c;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / c;
    const int cc = (int)(i - r * c);
    const float s = sigm(bf2f(g[r * ldg + cc]));
    const float d = dout[i];
    dy[i] = f2bf(d * s);
    dg[r * ldg + cc] = f2bf(d * y[i] * s * (1.f - s));
  }
}

extern "C" int dfold_gate_mul_fwd(const float* y, const void* g, float* out, int64_t R, int32_t c, int64_t ldg, void* stream) {
  if (!y || !g || !out || R <= 0 || c <= 0 || ldg < c) return DFOLD_EINVAL;
  long blocks = (R * c + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  DFOLD_LAUNCH(gate_mul_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, y, (const bf16_t*)g, out, (long)R,
               c, (long)ldg);
  return dfold_check_launch();
}

extern "C" int dfold_gate_mul_bwd(const float* y, const void* g, const float* dout, void* dy, void* dg, int64_t R, int32_t c,
                                  int64_t ldg, void* stream) {
  if (!y || !g || !dout || !dy || !dg || R <= 0 || c <= 0 || ldg < c) return DFOLD_EINVAL;
  long blocks = (R * c + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  DFOLD_LAUNCH(gate_mul_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, y, (const bf16_t*)g, dout,
               (bf16_t*)dy, (bf16_t*)dg, (long)R, c, (long)ldg);
  return dfold_check_launch();
}

// ---------------------------------------------------------------------------------------------
// Triangle attention row softmax (triangular_attention.py:105-113, primitives.py:219-243):
//   P[i,h,q,:] = softmax_k( S[i,h,q,k] + inf*(mask[i,k]-1) + tri[h,q,k] ),  S fp32 [I][H][Q][K] in place -> P, bf16 copy
// backward: dS = P (dP - sum_k P dP) in place; dtri[h,q,k] = sum_i dS (separate reduction kernel)
// ---------------------------------------------------------------------------------------------
#define TA_MAXT 16
__global__ __launch_bounds__(256) void triatt_softmax_fwd_kernel(float* S, const float* __restrict__ mask,
                                                                 const float* __restrict__ tri, bf16_t* __restrict__ Pb, int I,
                                                                 int H, int N, float inf) {
  const int lane = threadIdx.x & 63;
  const long row = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;  // (i, h, q)
  if (row >= (long)I * H * N) return;
  const int q = (int)(row % N);
  const int h = (int)((row / N) % H);
  const int i = (int)(row / ((long)N * H));
  float* s = S + row * N;
  const float* tr = tri + ((long)h * N + q) * N;
  const float* mk = mask + (long)i * N;
  float v[TA_MAXT];
  float mx = -3.0e38f;
#pragma unroll
  for (int t = 0; t < TA_MAXT; ++t) {
    const int k = lane + 64 * t;
    v[t] = -3.0e38f;
    if (k < N) {
      v[t] = s[k] + inf * (mk[k] - 1.f) + tr[k];
      mx = fmaxf(mx, v[t]);
    }
  }
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < TA_MAXT; ++t)
    if (lane + 64 * t < N) {
      v[t] = expf(v[t] - mx);
      sum += v[t];
    }
  const float inv = 1.f / wave_sum(sum);
#pragma unroll
  for (int t = 0; t < TA_MAXT; ++t) {
    const int k = lane + 64 * t;
    if (k < N) {
      const float p = v[t] * inv;
      s[k] = p;
      Pb[row * N + k] = f2bf(p);
    }
  }
}

extern "C" int dfold_triatt_softmax_fwd(float* S, const float* mask, const float* tri, void* P_bf16, int32_t I, int32_t H,
                                        int32_t N, float inf, void* stream) {
  if (!S || !mask || !tri || !P_bf16 || I <= 0 || H <= 0 || N <= 0 || N > 64 * TA_MAXT) return DFOLD_EINVAL;
  const long rows = (long)I * H * N;
  DFOLD_LAUNCH(triatt_softmax_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, S, mask, tri,
               (bf16_t*)P_bf16, I, H, N, inf);
  return dfold_check_launch();
}

__global__ __launch_bounds__(256) void triatt_softmax_bwd_kernel(const float* __restrict__ P, float* dP, bf16_t* __restrict__ dSb,
                                                                 long rows, int N) {
  const int lane = threadIdx.x & 63;
  const long row = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (row >= rows) return;
  const float* p = P + row * N;
  float* d = dP + row * N;
  float pv[TA_MAXT], gv[TA_MAXT];
  float dot = 0.f;
#pragma unroll
  for (int t = 0; t < TA_MAXT; ++t) {
    const int k = lane + 64 * t;
    pv[t] = gv[t] = 0.f;
    if (k < N) {
      pv[t] = p[k];
      gv[t] = d[k];
      dot += pv[t] * gv[t];
    }
  }
  dot = wave_sum(dot);
#pragma unroll
  for (int t = 0; t < TA_MAXT; ++t) {
    const int k = lane + 64 * t;
    if (k < N) {
      const float ds = pv[t] * (gv[t] - dot);
      d[k] = ds;
      dSb[row * N + k] = f2bf(ds);
    }
  }
}

extern "C" int dfold_triatt_softmax_bwd(const float* P, float* dP, void* dS_bf16, int64_t rows, int32_t N, void* stream) {
  if (!P || !dP || !dS_bf16 || rows <= 0 || N <= 0 || N > 64 * TA_MAXT) return DFOLD_EINVAL;
  DFOLD_LAUNCH(triatt_softmax_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, P, dP,
               (bf16_t*)dS_bf16, (long)rows, N);
  return dfold_check_launch();
}

// out[e] = sum_{i<I} x[i*stride + e]  for e < n   (dtri = sum over rows i of dS; fp32)
__global__ void sum_leading_kernel(const float* __restrict__ x, float* __restrict__ out, int I, long n, long stride) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int i = 0; i < I; ++i) s += x[(long)i * stride + e];
    out[e] = s;
  }
}

extern "C" int dfold_sum_leading(const float* x, float* out, int32_t I, int64_t n, int64_t stride, void* stream) {
  if (!x || !out || I <= 0 || n <= 0) return DFOLD_EINVAL;
  long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  DFOLD_LAUNCH(sum_leading_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, out, I, (long)n, (long)stride);
  return dfold_check_launch();
}
