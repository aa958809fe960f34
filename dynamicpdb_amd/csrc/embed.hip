// First layer of the five feature embedders (reference DFOLDIpaScore force/vel/index/rigid/angle_embeder,
// src/model/ipa_pytorch_dynamic.py:757-796):   h = SiLU(x W^T + b),  x fp32 [P, k] with k = 1, 3, 7 or 14.
// K is far too small for the matrix cores (and for a vendor GEMM: hipBLASLt spent 2.2 ms per call on it), so
// this is a VALU kernel: one thread per output column keeps its W row in registers, rows stream through LDS.
#include "dfold_common.h"
#include "../../include/dfold_hip.h"

#define EMB_MAXK 16
#define EMB_ROWS 64

__device__ __forceinline__ float sigmoid_f(float y) { return 1.f / (1.f + expf(-y)); }

// out bf16 [P][D] (D == blockDim.x == 256)
__global__ __launch_bounds__(256) void embed_in_fwd_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                           const float* __restrict__ b, bf16_t* __restrict__ out, long P,
                                                           int k, int D) {
  __shared__ float xs[EMB_ROWS][EMB_MAXK];
  const int o = threadIdx.x;
  float w[EMB_MAXK];
#pragma unroll
  for (int c = 0; c < EMB_MAXK; ++c) w[c] = c < k ? W[(long)o * k + c] : 0.f;
  const float bo = b[o];
  for (long r0 = (long)blockIdx.x * EMB_ROWS; r0 < P; r0 += (long)gridDim.x * EMB_ROWS) {
    const int rows = (int)min((long)EMB_ROWS, P - r0);
    __syncthreads();
    for (int e = threadIdx.x; e < rows * k; e += 256) xs[e / k][e % k] = x[r0 * k + e];
    __syncthreads();
    for (int r = 0; r < rows; ++r) {
      float acc = bo;
#pragma unroll
      for (int c = 0; c < EMB_MAXK; ++c)
        if (c < k) acc += xs[r][c] * w[c];
      out[(r0 + r) * D + o] = f2bf(acc * sigmoid_f(acc));
    }
  }
}

extern "C" int dfold_embed_in_fwd(const float* x, const float* W, const float* b, void* out_bf16, int64_t P, int32_t k,
                                  int32_t D, void* stream) {
  if (!x || !W || !b || !out_bf16 || P <= 0 || k <= 0 || k > EMB_MAXK || D != 256) return DFOLD_EINVAL;
  long blocks = (P + EMB_ROWS - 1) / EMB_ROWS;
  if (blocks > 2048) blocks = 2048;
  DFOLD_LAUNCH(embed_in_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, W, b, (bf16_t*)out_bf16,
               (long)P, k, D);
  return dfold_check_launch();
}

// backward: g bf16 [P][D] = dL/dh.  gpre = g * silu'(pre) with pre recomputed;
//   dW[o][c] += sum_r gpre[r][o] x[r][c];  db[o] += sum_r gpre[r][o];  dx[r][c] = sum_o gpre[r][o] W[o][c] (optional)
// dW / db are accumulated with fp32 atomics (zero them first).
__global__ __launch_bounds__(256) void embed_in_bwd_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                           const float* __restrict__ b, const bf16_t* __restrict__ g,
                                                           float* __restrict__ dW, float* __restrict__ db,
                                                           float* __restrict__ dx, long P, int k, int D) {
  __shared__ float xs[EMB_ROWS][EMB_MAXK];
  __shared__ float dxs[EMB_ROWS][EMB_MAXK];
  const int o = threadIdx.x, lane = threadIdx.x & 63;
  float w[EMB_MAXK], aw[EMB_MAXK];
#pragma unroll
  for (int c = 0; c < EMB_MAXK; ++c) {
    w[c] = c < k ? W[(long)o * k + c] : 0.f;
    aw[c] = 0.f;
  }
  const float bo = b[o];
  float ab = 0.f;
  for (long r0 = (long)blockIdx.x * EMB_ROWS; r0 < P; r0 += (long)gridDim.x * EMB_ROWS) {
    const int rows = (int)min((long)EMB_ROWS, P - r0);
    __syncthreads();
    for (int e = threadIdx.x; e < rows * k; e += 256) xs[e / k][e % k] = x[r0 * k + e];
    if (dx != nullptr)
      for (int e = threadIdx.x; e < EMB_ROWS * EMB_MAXK; e += 256) dxs[e / EMB_MAXK][e % EMB_MAXK] = 0.f;
    __syncthreads();
    // rows in groups of 8 with the 8 gradient loads issued together (the row loop is otherwise one dependent
    // global load per iteration)
    for (int rb = 0; rb < rows; rb += 8) {
      float gv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) gv[u] = bf2f(g[(r0 + min(rb + u, rows - 1)) * D + o]);     // unconditional (rows past the end re-read
                                                                                              // the last one, unused): under a condition
                                                                                              // each load was its own branch + wait
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int r = rb + u;
        if (r < rows) {
          float pre = bo;
#pragma unroll
          for (int c = 0; c < EMB_MAXK; ++c)
            if (c < k) pre += xs[r][c] * w[c];
          const float sg = sigmoid_f(pre);
          const float gp = gv[u] * sg * (1.f + pre * (1.f - sg));
          ab += gp;
#pragma unroll
          for (int c = 0; c < EMB_MAXK; ++c)
            if (c < k) aw[c] += gp * xs[r][c];
          if (dx != nullptr) {
#pragma unroll
            for (int c = 0; c < EMB_MAXK; ++c)
              if (c < k) {
                const float s = wave_sum(gp * w[c]);
                if (lane == 0) atomicAdd(&dxs[r][c], s);
              }
          }
        }
      }
    }
    if (dx != nullptr) {
      __syncthreads();
      for (int e = threadIdx.x; e < rows * k; e += 256) dx[r0 * k + e] = dxs[e / k][e % k];
    }
  }
  atomicAdd(db + o, ab);
#pragma unroll
  for (int c = 0; c < EMB_MAXK; ++c)
    if (c < k) atomicAdd(dW + (long)o * k + c, aw[c]);
}

// The same with dx for k <= 8 (the rigid embedder: the frames carry a gradient): the row-wise sums over the 256 channels were
// k wave reductions + an LDS atomic per row and wave (0.39 ms per call at config 3, 4 calls per step).  Here the
// pre-activation gradients of a 32-row tile are parked in LDS (pitch 257: conflict-free for both phases) and the 32 x 8
// (row, component) sums run as one thread each over the 256 channels.
#define EMB_DXR 32
__global__ __launch_bounds__(256) void embed_in_bwd_dx_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                              const float* __restrict__ b, const bf16_t* __restrict__ g,
                                                              float* __restrict__ dW, float* __restrict__ db,
                                                              float* __restrict__ dx, long P, int k, int D) {
  __shared__ float xs[EMB_DXR][8];
  __shared__ float wt[256][8];
  __shared__ float gps[EMB_DXR][257];
  const int o = threadIdx.x;
  float w[8], aw[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    w[c] = c < k ? W[(long)o * k + c] : 0.f;
    wt[o][c] = w[c];
    aw[c] = 0.f;
  }
  const float bo = b[o];
  float ab = 0.f;
  const int pr = threadIdx.x >> 3, pc = threadIdx.x & 7;        // phase 2: (row of the tile, component)
  for (long r0 = (long)blockIdx.x * EMB_DXR; r0 < P; r0 += (long)gridDim.x * EMB_DXR) {
    const int rows = (int)min((long)EMB_DXR, P - r0);
    __syncthreads();                                              // the previous tile's phase 2 is done with gps / xs
    for (int e = threadIdx.x; e < EMB_DXR * 8; e += 256) {
      const int r = e >> 3, c = e & 7;
      xs[r][c] = (r < rows && c < k) ? x[(r0 + r) * k + c] : 0.f;
    }
    __syncthreads();
    for (int rb = 0; rb < EMB_DXR; rb += 8) {
      float gv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) gv[u] = bf2f(g[(r0 + min(rb + u, rows - 1)) * D + o]);     // unconditional, see embed_in_bwd_kernel
#pragma unroll
      for (int u = 0; u < 8; ++u) gv[u] = rb + u < rows ? gv[u] : 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int r = rb + u;
        float pre = bo;
#pragma unroll
        for (int c = 0; c < 8; ++c) pre += xs[r][c] * w[c];
        const float sg = sigmoid_f(pre);
        const float gp = gv[u] * sg * (1.f + pre * (1.f - sg));      // rows past the end: gv = 0
        ab += gp;
#pragma unroll
        for (int c = 0; c < 8; ++c) aw[c] += gp * xs[r][c];
        gps[r][o] = gp;
      }
    }
    __syncthreads();
    if (pr < rows && pc < k) {
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 4
      for (int oo = 0; oo < 256; oo += 4) {
        a0 += gps[pr][oo] * wt[oo][pc];
        a1 += gps[pr][oo + 1] * wt[oo + 1][pc];
        a2 += gps[pr][oo + 2] * wt[oo + 2][pc];
        a3 += gps[pr][oo + 3] * wt[oo + 3][pc];
      }
      dx[(r0 + pr) * k + pc] = (a0 + a1) + (a2 + a3);
    }
  }
  atomicAdd(db + o, ab);
#pragma unroll
  for (int c = 0; c < 8; ++c)
    if (c < k) atomicAdd(dW + (long)o * k + c, aw[c]);
}

extern "C" int dfold_embed_in_bwd(const float* x, const float* W, const float* b, const void* g_bf16, float* dW, float* db,
                                  float* dx, int64_t P, int32_t k, int32_t D, void* stream) {
  if (!x || !W || !b || !g_bf16 || !dW || !db || P <= 0 || k <= 0 || k > EMB_MAXK || D != 256) return DFOLD_EINVAL;
  // 512 workgroups (measured: 128 workgroups of 512 rows each ran 2.6x slower -- the row loop, not the closing
  // 256 x (k + 1) fp32 atomics per workgroup, sets the time)
  if (dx != nullptr && k <= 8) {
    long nb = (P + EMB_DXR - 1) / EMB_DXR;
    if (nb > 1024) nb = 1024;
    DFOLD_LAUNCH(embed_in_bwd_dx_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, W, b, (const bf16_t*)g_bf16, dW, db,
                 dx, (long)P, k, D);
    return dfold_check_launch();
  }
  long blocks = (P + EMB_ROWS - 1) / EMB_ROWS;
  if (blocks > 512) blocks = 512;
  DFOLD_LAUNCH(embed_in_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, W, b, (const bf16_t*)g_bf16,
               dW, db, dx, (long)P, k, D);
  return dfold_check_launch();
}
