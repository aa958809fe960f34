// Layout / reduction helpers around the implicit-GEMM conv tower (reference ConvNet,
// src/model/ipa_pytorch_dynamic.py:664-706).  All HBM-bound; 16-B vector access where the layout allows.
#include "dfold_common.h"
#include "../../include/dfold_hip.h"

// ---------------------------------------------------------------------------------------------
// fp32 -> bf16 cast (n multiple of 4 handled vectorised, tail scalar)
// ---------------------------------------------------------------------------------------------
__global__ void cast_f32_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, long n) {
  long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const long stride = (long)gridDim.x * blockDim.x * 4;
  for (; i + 3 < n; i += stride) {
    const float4 v = *(const float4*)(src + i);
    uint2 o;
    o.x = pack2bf(v.x, v.y);
    o.y = pack2bf(v.z, v.w);
    *(uint2*)(dst + i) = o;
  }
  if (i < n && i + 3 >= n)
    for (long j = i; j < n; ++j) dst[j] = f2bf(src[j]);
}

// any alignment (views into larger tensors, e.g. one frame of a ragged [F, N, 6] gradient)
__global__ void cast_f32_bf16_scalar_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) dst[i] = f2bf(src[i]);
}

extern "C" int dfold_cast_f32_bf16(const float* src, void* dst, int64_t n, void* stream) {
  if (!src || !dst || n < 0) return DFOLD_EINVAL;
  if (n == 0) return DFOLD_OK;
  if (((uintptr_t)src & 15) || ((uintptr_t)dst & 7)) {
    long sb = (n + 255) / 256;
    if (sb > 4096) sb = 4096;
    DFOLD_LAUNCH(cast_f32_bf16_scalar_kernel, dim3((unsigned)sb), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, (long)n);
    return dfold_check_launch();
  }
  long blocks = (n / 4 + 255) / 256 + 1;
  if (blocks > 4096) blocks = 4096;
  DFOLD_LAUNCH(cast_f32_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src,
                     (bf16_t*)dst, (long)n);
  return dfold_check_launch();
}

__global__ void cast_bf16_f32_kernel(const bf16_t* __restrict__ src, float* __restrict__ dst, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) dst[i] = bf2f(src[i]);
}

extern "C" int dfold_cast_bf16_f32(const void* src, float* dst, int64_t n, void* stream) {
  if (!src || !dst || n < 0) return DFOLD_EINVAL;
  if (n == 0) return DFOLD_OK;
  long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  DFOLD_LAUNCH(cast_bf16_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)src, dst, (long)n);
  return dfold_check_launch();
}

// ---------------------------------------------------------------------------------------------
// Conv weight pack: W fp32 [CO][CI][5][5] (reference OIHW) ->
//   Wf bf16 [CO][25][CI]          forward implicit-GEMM B operand
//   Wd bf16 [CI][25][CO], tap-flipped: Wd[ci][t][co] = W[co][ci][24-t]   dgrad B operand
// One block = 32 co x 32 ci through LDS (bf16, 25 taps).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_weight_pack_kernel(const float* __restrict__ W, bf16_t* __restrict__ Wf,
                                                               bf16_t* __restrict__ Wd, int CO, int CI) {
  // rows of 32 x 26 + 2 elements: 417 dwords apart, so that the Wd phase (lanes along co) hits 32 different banks
  __shared__ bf16_t t[32][32 * 26 + 2];
  const int co0 = blockIdx.y * 32, ci0 = blockIdx.x * 32;
  // load: for each co row, 32 ci x 25 taps = 800 contiguous floats
  for (int r = 0; r < 32; ++r) {
    const int co = co0 + r;
    if (co >= CO) break;
    const float* src = W + ((long)co * CI + ci0) * 25;
    const int lim = min(32, CI - ci0) * 25;
    for (int e = threadIdx.x; e < lim; e += 256) t[r][(e / 25) * 26 + e % 25] = f2bf(src[e]);
  }
  __syncthreads();
  // Wf rows (co, tap): 32 contiguous ci
  for (int e = threadIdx.x; e < 32 * 25 * 32; e += 256) {
    const int ci = e & 31, tap = (e >> 5) % 25, r = e / (32 * 25);
    if (co0 + r < CO && ci0 + ci < CI) Wf[((long)(co0 + r) * 25 + tap) * CI + ci0 + ci] = t[r][ci * 26 + tap];
  }
  // Wd rows (ci, tap'): 32 contiguous co
  for (int e = threadIdx.x; e < 32 * 25 * 32; e += 256) {
    const int r = e & 31, tap = (e >> 5) % 25, ci = e / (32 * 25);
    if (co0 + r < CO && ci0 + ci < CI) Wd[((long)(ci0 + ci) * 25 + tap) * CO + co0 + r] = t[r][ci * 26 + 24 - tap];
  }
}

// the same for CO, CI multiples of 32 (the tower's 1280 / 640): whole 32 x 32 x 25 blocks, 16-byte stores (the scalar form
// above issues 51200 two-byte stores per block and ran at 1.1 TB/s: 152 us per 82 MB weight, 1.2 ms per step)
__global__ __launch_bounds__(256) void conv_weight_pack_v8_kernel(const float* __restrict__ W, bf16_t* __restrict__ Wf,
                                                                  bf16_t* __restrict__ Wd, int CO, int CI) {
  __shared__ bf16_t t[32][32 * 26 + 2];
  const int co0 = blockIdx.y * 32, ci0 = blockIdx.x * 32;
  // the block's 32 rows x 800 floats as 6400 16-byte loads, 25 per thread, all in flight before the first LDS store (one
  // 4-byte load per thread and round trip ran this kernel at 1.2 TB/s: 133 us per 82 MB weight, 1.06 ms per step)
  {
    const f32x4* src4 = (const f32x4*)(W + ((long)co0 * CI + ci0) * 25);
    const long row4 = (long)CI * 25 / 4;
    f32x4 v[25];
#pragma unroll
    for (int it = 0; it < 25; ++it) {
      const int e4 = threadIdx.x + it * 256, r = e4 / 200, q = e4 - r * 200;
      v[it] = src4[r * row4 + q];
    }
#pragma unroll
    for (int it = 0; it < 25; ++it) {
      const int e4 = threadIdx.x + it * 256, r = e4 / 200, q = e4 - r * 200;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int e = q * 4 + k;
        t[r][(e / 25) * 26 + e % 25] = f2bf_hw(v[it][k]);
      }
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 32 * 25 * 4; e += 256) {       // Wf rows (co, tap): 4 chunks of 8 ci
    const int ch = e & 3, tap = (e >> 2) % 25, r = e / 100;
    const bf16_t* s0 = &t[r][ch * 8 * 26 + tap];
    const uint4 v = make_uint4(s0[0] | ((uint32_t)s0[26] << 16), s0[52] | ((uint32_t)s0[78] << 16), s0[104] | ((uint32_t)s0[130] << 16),
                               s0[156] | ((uint32_t)s0[182] << 16));
    *(uint4*)(Wf + ((long)(co0 + r) * 25 + tap) * CI + ci0 + ch * 8) = v;
  }
  for (int e = threadIdx.x; e < 32 * 25 * 4; e += 256) {       // Wd rows (ci, flipped tap): 4 chunks of 8 co
    const int ch = e & 3, tap = (e >> 2) % 25, ci = e / 100;
    const int col = ci * 26 + 24 - tap;
    const uint4 v = make_uint4(t[ch * 8][col] | ((uint32_t)t[ch * 8 + 1][col] << 16), t[ch * 8 + 2][col] | ((uint32_t)t[ch * 8 + 3][col] << 16),
                               t[ch * 8 + 4][col] | ((uint32_t)t[ch * 8 + 5][col] << 16), t[ch * 8 + 6][col] | ((uint32_t)t[ch * 8 + 7][col] << 16));
    *(uint4*)(Wd + ((long)(ci0 + ci) * 25 + tap) * CO + co0 + ch * 8) = v;
  }
}

extern "C" int dfold_conv_weight_pack(const float* W, void* Wf, void* Wd, int32_t CO, int32_t CI, void* stream) {
  if (!W || !Wf || !Wd || CO <= 0 || CI <= 0) return DFOLD_EINVAL;
  dim3 grid((CI + 31) / 32, (CO + 31) / 32);
  if ((CO % 32) == 0 && (CI % 32) == 0 && (((uintptr_t)Wf | (uintptr_t)Wd) & 15) == 0)
    DFOLD_LAUNCH(conv_weight_pack_v8_kernel, grid, dim3(256), 0, (hipStream_t)stream, W, (bf16_t*)Wf, (bf16_t*)Wd, CO, CI);
  else
    DFOLD_LAUNCH(conv_weight_pack_kernel, grid, dim3(256), 0, (hipStream_t)stream, W, (bf16_t*)Wf, (bf16_t*)Wd,
                       CO, CI);
  return dfold_check_launch();
}

// dWg fp32 [CO][25][CI] (GEMM layout) -> G fp32 [CO][CI][25] (reference layout); accumulate != 0 adds.
__global__ __launch_bounds__(256) void conv_wgrad_unpack_kernel(const float* __restrict__ dWg, float* __restrict__ G,
                                                                int CO, int CI, int accumulate) {
  __shared__ float t[25][65];
  const int co = blockIdx.y, ci0 = blockIdx.x * 64;
  const int w = min(64, CI - ci0);
  for (int e = threadIdx.x; e < 25 * 64; e += 256) {
    const int ci = e & 63, tap = e >> 6;
    if (ci < w) t[tap][ci] = dWg[((long)co * 25 + tap) * CI + ci0 + ci];
  }
  __syncthreads();
  float* dst = G + ((long)co * CI + ci0) * 25;
  for (int e = threadIdx.x; e < w * 25; e += 256) {
    const float v = t[e % 25][e / 25];
    dst[e] = accumulate ? dst[e] + v : v;
  }
}

// transposed accumulator layout dWgT fp32 [CI][25][CO] -> G fp32 [CO][CI][25]; block = 16 ci x 32 co
__global__ __launch_bounds__(256) void conv_wgrad_unpack_t_kernel(const float* __restrict__ dWgT, float* __restrict__ G,
                                                                  int CO, int CI, int accumulate) {
  __shared__ float t[16][25][33];
  const int ci0 = blockIdx.x * 16, co0 = blockIdx.y * 32;
  for (int e = threadIdx.x; e < 16 * 25 * 32; e += 256) {
    const int co = e & 31, tap = (e >> 5) % 25, ci = e / (32 * 25);
    if (ci0 + ci < CI && co0 + co < CO) t[ci][tap][co] = dWgT[((long)(ci0 + ci) * 25 + tap) * CO + co0 + co];
  }
  __syncthreads();
  const int cw = min(16, CI - ci0);
  for (int co = 0; co < 32 && co0 + co < CO; ++co) {
    float* dst = G + ((long)(co0 + co) * CI + ci0) * 25;
    for (int e = threadIdx.x; e < cw * 25; e += 256) {
      const float v = t[e / 25][e % 25][co];
      dst[e] = accumulate ? dst[e] + v : v;
    }
  }
}

extern "C" int dfold_conv_wgrad_unpack(const float* dWg, float* G, int32_t CO, int32_t CI, int32_t accumulate,
                                       int32_t transposed, void* stream) {
  if (!dWg || !G || CO <= 0 || CI <= 0) return DFOLD_EINVAL;
  if (transposed) {
    dim3 grid((CI + 15) / 16, (CO + 31) / 32);
    DFOLD_LAUNCH(conv_wgrad_unpack_t_kernel, grid, dim3(256), 0, (hipStream_t)stream, dWg, G, CO, CI, accumulate);
  } else {
    dim3 grid((CI + 63) / 64, CO);
    DFOLD_LAUNCH(conv_wgrad_unpack_kernel, grid, dim3(256), 0, (hipStream_t)stream, dWg, G, CO, CI, accumulate);
  }
  return dfold_check_launch();
}

// ---------------------------------------------------------------------------------------------
// Transposed, column-shifted copies of a padded grid tensor for the conv wgrad:
//   X bf16 [W][Fp][Wp][C]  ->  T bf16 [nd][C][W][Fp][NP],  T[d][c][w][f][n] = X[w][f][n + d0 + d][c]  for n < N
// (NP >= N = row pitch of T, a multiple of 8 so that every frame row is a 16-byte aligned K run of the wgrad GEMM;
// the tail n in [N, NP) is never written -- the caller keeps it zero)
// for the padded frame rows f in [f0, f0 + nf) only (the rest of T is left untouched: a frame sub-range of the wgrad
// reduction only reads those rows).  block = (n tile of 64, c tile of 64, (w, f - f0) row)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void grid_transpose_shift_kernel(const bf16_t* __restrict__ X, bf16_t* __restrict__ T,
                                                                   int Wn, int Fp, int Wp, int C, int N, int NP, int d0,
                                                                   int nd, int f0, int nf, float* __restrict__ colsum) {
  __shared__ bf16_t t[68][66];
  const int n0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int wz = blockIdx.z / nf;
  const int wf = wz * Fp + f0 + (blockIdx.z - wz * nf);  // w*Fp + f
  const int cols = min(64 + nd - 1, Wp - (n0 + d0));  // padded columns available from n0+d0
  const int cw = min(64, C - c0);
  const bf16_t* src = X + ((long)wf * Wp + n0 + d0) * C + c0;
  // 8 bf16 (16 B) per thread per row when the c tile is full and aligned
  for (int e = threadIdx.x; e < 68 * 8; e += 256) {
    const int col = e >> 3, cc = (e & 7) * 8;
    if (col < cols) {
      if (cw == 64 && (C & 7) == 0) {
        const uint4 v = *(const uint4*)(src + (long)col * C + cc);
        const bf16_t* pv = (const bf16_t*)&v;
#pragma unroll
        for (int j = 0; j < 8; ++j) t[col][cc + j] = pv[j];
      } else {
        for (int j = 0; j < 8; ++j)
          if (cc + j < cw) t[col][cc + j] = src[(long)col * C + cc + j];
      }
    }
  }
  __syncthreads();
  const long plane = (long)Wn * Fp * NP;  // elements per (d, c)
  const int nw = min(64, N - n0);
  if (colsum != nullptr) {
    // fused bias gradient: per-channel sum over the interior cells of this tile (border rows / columns are zero)
    const int c = threadIdx.x & 63, part = threadIdx.x >> 6, sh = 2 - d0;
    float sacc = 0.f;
    if (c < cw)
      for (int n = part; n < nw; n += 4) sacc += bf2f(t[n + sh][c]);
    __shared__ float cs[4][64];
    cs[part][c] = sacc;
    __syncthreads();
    if (part == 0 && c < cw) atomicAdd(colsum + c0 + c, cs[0][c] + cs[1][c] + cs[2][c] + cs[3][c]);
  }
  for (int d = 0; d < nd; ++d) {
    for (int e = threadIdx.x; e < 64 * 64; e += 256) {
      const int n = e & 63, c = e >> 6;
      if (n < nw && c < cw) T[((long)d * C + c0 + c) * plane + (long)wf * NP + n0 + n] = t[n + d][c];
    }
  }
}

extern "C" int dfold_grid_transpose_shift(const void* X, void* T, int32_t Wn, int32_t Fp, int32_t Wp, int32_t C,
                                          int32_t N, int32_t NP, int32_t d0, int32_t nd, int32_t f0, int32_t nf,
                                          float* colsum, void* stream) {
  if (!X || !T || Wn <= 0 || Fp <= 0 || Wp <= 0 || C <= 0 || N <= 0 || nd <= 0 || nd > 5 || d0 < 0) return DFOLD_EINVAL;
  if (NP < N) return DFOLD_EINVAL;
  if (f0 < 0 || nf <= 0 || f0 + nf > Fp) return DFOLD_EINVAL;
  if (N + d0 + nd - 1 > Wp) return DFOLD_EINVAL;
  if (colsum && (d0 > 2 || 2 - d0 + 64 > 68)) return DFOLD_EINVAL;
  dim3 grid((N + 63) / 64, (C + 63) / 64, Wn * nf);
  DFOLD_LAUNCH(grid_transpose_shift_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)X,
                     (bf16_t*)T, Wn, Fp, Wp, C, N, NP, d0, nd, f0, nf, colsum);
  return dfold_check_launch();
}

// ---------------------------------------------------------------------------------------------
// Column sums of a bf16 [R][C] matrix into fp32 [C] (bias gradients); out must be zeroed (or hold the
// running sum) by the caller: partial sums are added atomically.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void colsum_bf16_kernel(const bf16_t* __restrict__ X, float* __restrict__ out, long R,
                                                          int C, long ld) {
  __shared__ float part[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  const long rows_per = (R + gridDim.y - 1) / gridDim.y;
  const long r0 = (long)blockIdx.y * rows_per, r1 = min(R, r0 + rows_per);
  float s = 0.f;
  if (c < C)
    for (long r = r0 + rl; r < r1; r += 4) s += bf2f(X[r * ld + c]);
  part[rl][threadIdx.x & 63] = s;
  __syncthreads();
  if (rl == 0 && c < C) atomicAdd(out + c, part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]);
}

// 16-byte form: a thread owns 8 adjacent columns, a block covers TILE column groups x (256/TILE) row lanes; ~2 blocks
// per CU so that the fp32 atomics per output address stay in the hundreds (same-address atomics serialise in L2).
template <int TILE>
__global__ __launch_bounds__(256) void colsum_bf16_v8_kernel(const bf16_t* __restrict__ X, float* __restrict__ out, long R,
                                                             int C, long ld, long bstride) {
  constexpr int RL = 256 / TILE;
  X += (long)blockIdx.z * bstride;        // batch item (the same frame range of another window of a padded grid)
  __shared__ float part[RL][TILE * 8 + 1];
  const int tx = threadIdx.x % TILE, ty = threadIdx.x / TILE;
  const int cg = blockIdx.x * TILE + tx;
  const long rows_per = (R + gridDim.y - 1) / gridDim.y;
  const long r0 = (long)blockIdx.y * rows_per, r1 = min(R, r0 + rows_per);
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (cg * 8 < C) {
    const bf16_t* p = X + (long)cg * 8;
    long r = r0 + ty;
    for (; r + 3 * RL < r1; r += 4 * RL) {
      uint4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *(const uint4*)(p + (r + (long)u * RL) * ld);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bf16_t* e = (const bf16_t*)&v[u];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += bf2f(e[j]);
      }
    }
    for (; r < r1; r += RL) {
      const uint4 v = *(const uint4*)(p + r * ld);
      const bf16_t* e = (const bf16_t*)&v;
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += bf2f(e[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) part[ty][tx * 8 + j] = acc[j];
  __syncthreads();
  for (int c = threadIdx.x; c < TILE * 8; c += 256) {
    const int col = blockIdx.x * TILE * 8 + c;
    if (col < C) {
      float s = 0.f;
#pragma unroll 8
      for (int y = 0; y < RL; ++y) s += part[y][c];
      atomicAdd(out + col, s);
    }
  }
}

static int colsum_launch(const void* X, float* out, int64_t R, int32_t C, int64_t ld, int32_t nbatch, int64_t bstride, hipStream_t st) {
  if ((C % 8) == 0 && (ld % 8) == 0 && (bstride % 8) == 0 && ((uintptr_t)X % 16) == 0) {
    const int cgs = C / 8;
    const int tile = cgs >= 64 ? 64 : cgs >= 32 ? 32 : cgs >= 16 ? 16 : cgs >= 8 ? 8 : 4;
    const int bx = (cgs + tile - 1) / tile;
    const int rl = 256 / tile;
    long by = (R + (long)rl * 16 - 1) / ((long)rl * 16);
    const long cap = max(1L, 512L / ((long)bx * nbatch));
    if (by > cap) by = cap;
    dim3 grid(bx, (unsigned)by, (unsigned)nbatch);
    switch (tile) {
      case 64: DFOLD_LAUNCH(colsum_bf16_v8_kernel<64>, grid, dim3(256), 0, st, (const bf16_t*)X, out, (long)R, C, (long)ld, (long)bstride); break;
      case 32: DFOLD_LAUNCH(colsum_bf16_v8_kernel<32>, grid, dim3(256), 0, st, (const bf16_t*)X, out, (long)R, C, (long)ld, (long)bstride); break;
      case 16: DFOLD_LAUNCH(colsum_bf16_v8_kernel<16>, grid, dim3(256), 0, st, (const bf16_t*)X, out, (long)R, C, (long)ld, (long)bstride); break;
      case 8: DFOLD_LAUNCH(colsum_bf16_v8_kernel<8>, grid, dim3(256), 0, st, (const bf16_t*)X, out, (long)R, C, (long)ld, (long)bstride); break;
      default: DFOLD_LAUNCH(colsum_bf16_v8_kernel<4>, grid, dim3(256), 0, st, (const bf16_t*)X, out, (long)R, C, (long)ld, (long)bstride); break;
    }
    return dfold_check_launch();
  }
  for (int z = 0; z < nbatch; ++z) {      // unaligned shapes: the scalar kernel, one launch per batch item
    long chunks = (R + 511) / 512;
    if (chunks > 1024) chunks = 1024;
    dim3 grid((C + 63) / 64, (unsigned)chunks);
    DFOLD_LAUNCH(colsum_bf16_kernel, grid, dim3(256), 0, st, (const bf16_t*)X + (long)z * bstride, out, (long)R, C, (long)ld);
  }
  return dfold_check_launch();
}

extern "C" int dfold_colsum_bf16(const void* X, float* out, int64_t R, int32_t C, int64_t ld, void* stream) {
  if (!X || !out || R <= 0 || C <= 0 || ld < C) return DFOLD_EINVAL;
  return colsum_launch(X, out, R, C, ld, 1, 0, (hipStream_t)stream);
}

extern "C" int dfold_colsum_bf16_batched(const void* X, float* out, int64_t R, int32_t C, int64_t ld, int32_t nbatch, int64_t bstride,
                                         void* stream) {
  if (!X || !out || R <= 0 || C <= 0 || ld < C || nbatch <= 0 || nbatch > 65535 || bstride < 0) return DFOLD_EINVAL;
  return colsum_launch(X, out, R, C, ld, nbatch, bstride, (hipStream_t)stream);
}

// out = (v > 0) ? g : 0   (ReLU backward on bf16 tensors; n multiple of 8 fast path)
__global__ void relu_mask_bf16_kernel(const bf16_t* __restrict__ g, const bf16_t* __restrict__ v, bf16_t* __restrict__ out,
                                      long n) {
  long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
  const long stride = (long)gridDim.x * blockDim.x * 8;
  for (; i + 7 < n; i += stride) {
    uint4 a = *(const uint4*)(g + i);
    const uint4 b = *(const uint4*)(v + i);
    bf16_t* pa = (bf16_t*)&a;
    const bf16_t* pb = (const bf16_t*)&b;
#pragma unroll
    for (int j = 0; j < 8; ++j) pa[j] = bf2f(pb[j]) > 0.f ? pa[j] : (bf16_t)0;
    *(uint4*)(out + i) = a;
  }
  if (i < n)
    for (long j = i; j < n; ++j) out[j] = bf2f(v[j]) > 0.f ? g[j] : (bf16_t)0;
}

extern "C" int dfold_relu_mask_bf16(const void* g, const void* v, void* out, int64_t n, void* stream) {
  if (!g || !v || !out || n < 0) return DFOLD_EINVAL;
  if (n == 0) return DFOLD_OK;
  long blocks = (n / 8 + 255) / 256 + 1;
  if (blocks > 4096) blocks = 4096;
  DFOLD_LAUNCH(relu_mask_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)g,
                     (const bf16_t*)v, (bf16_t*)out, (long)n);
  return dfold_check_launch();
}

// ---------------------------------------------------------------------------------------------
// Batched 2-D transpose of bf16 matrices: dst[b][c][r] = src[b][r][c]
//   src row stride lds_, batch stride sbs; dst row stride ldd, batch stride sbd (elements).
// 64x64 tiles through LDS; reads and writes both coalesced along their contiguous axis.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst,
                                                             int R, int C, long lds_, long ldd, long sbs0, long sbs1,
                                                             long sbd0, long sbd1, int nb1) {
  __shared__ bf16_t t[64][66];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int z0 = blockIdx.z / nb1, z1 = blockIdx.z - z0 * nb1;
  const bf16_t* s = src + z0 * sbs0 + z1 * sbs1;
  bf16_t* d = dst + z0 * sbd0 + z1 * sbd1;
  for (int e = threadIdx.x; e < 64 * 64; e += 256) {
    const int c = e & 63, r = e >> 6;
    if (r0 + r < R && c0 + c < C) t[r][c] = s[(long)(r0 + r) * lds_ + c0 + c];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 64 * 64; e += 256) {
    const int r = e & 63, c = e >> 6;
    if (r0 + r < R && c0 + c < C) d[(long)(c0 + c) * ldd + r0 + r] = t[r][c];
  }
}

// 16-byte form (R, C, leading dimensions, batch strides all multiples of 8 and 16-byte aligned bases): a 64x64 tile is
// read as 16-byte row segments, parked in LDS (row pitch 66: the transposed 2-byte gathers then spread over all
// banks) and written as 16-byte segments of the destination rows.
__global__ __launch_bounds__(256) void transpose_bf16_v8_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst,
                                                                int R, int C, long lds_, long ldd, long sbs0, long sbs1,
                                                                long sbd0, long sbd1, int nb1) {
  __shared__ bf16_t t[64][66];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int z0 = blockIdx.z / nb1, z1 = blockIdx.z - z0 * nb1;
  const bf16_t* s = src + z0 * sbs0 + z1 * sbs1;
  bf16_t* d = dst + z0 * sbd0 + z1 * sbd1;
  const int ch = threadIdx.x & 7, rr = threadIdx.x >> 3;
  uint4 v[2];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int r = r0 + rr + 32 * it, c = c0 + ch * 8;
    v[it] = (r < R && c < C) ? *(const uint4*)(s + (long)r * lds_ + c) : make_uint4(0, 0, 0, 0);
  }
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    uint32_t* row = (uint32_t*)&t[rr + 32 * it][ch * 8];   // pitch 132 B: 4-byte aligned
    row[0] = v[it].x; row[1] = v[it].y; row[2] = v[it].z; row[3] = v[it].w;
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int cl = rr + 32 * it, c = c0 + cl, r = r0 + ch * 8;
    if (c < C && r < R) {
      uint32_t o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        o[j] = (uint32_t)t[ch * 8 + 2 * j][cl] | ((uint32_t)t[ch * 8 + 2 * j + 1][cl] << 16);
      *(uint4*)(d + (long)c * ldd + r) = make_uint4(o[0], o[1], o[2], o[3]);
    }
  }
}

extern "C" int dfold_transpose_bf16(const void* src, void* dst, int32_t R, int32_t C, int64_t ld_src, int64_t ld_dst,
                                    int32_t nbatch, int32_t nb1, int64_t bs_src0, int64_t bs_src1, int64_t bs_dst0,
                                    int64_t bs_dst1, void* stream) {
  if (!src || !dst || R <= 0 || C <= 0 || nbatch <= 0 || nb1 <= 0 || ld_src < C || ld_dst < R) return DFOLD_EINVAL;
  if (nbatch > 65535) return DFOLD_EINVAL;
  dim3 grid((C + 63) / 64, (R + 63) / 64, nbatch);
  const bool vec = ((R | C) % 8) == 0 && ((ld_src | ld_dst | bs_src0 | bs_src1 | bs_dst0 | bs_dst1) % 8) == 0 &&
                   ((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0;
  if (vec)
    DFOLD_LAUNCH(transpose_bf16_v8_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, (bf16_t*)dst, R,
                 C, (long)ld_src, (long)ld_dst, (long)bs_src0, (long)bs_src1, (long)bs_dst0, (long)bs_dst1, nb1);
  else
    DFOLD_LAUNCH(transpose_bf16_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, (bf16_t*)dst, R,
                 C, (long)ld_src, (long)ld_dst, (long)bs_src0, (long)bs_src1, (long)bs_dst0, (long)bs_dst1, nb1);
  return dfold_check_launch();
}

// ---------------------------------------------------------------------------------------------
// dense [W][nf][N][C] -> interior of the padded grid, plus the non-zero frame flags of the tower's backward (round 6)
// ---------------------------------------------------------------------------------------------
// One frame row of the interior is ONE contiguous run of N*C elements in both tensors.  A block copies 32 KiB pieces of a row
// (16-byte vectors, 8 loads in flight per thread), ORs what it saw and raises the row's flag with one device-scope atomic per
// wave that met a non-zero (none at all on a zero frame); a second, single-block launch turns the flags into per-window prefix
// sums and leaves the scratch zero.  (First version: the last block to arrive did that -- 5120 blocks taking a ticket from ONE
// counter serialised the launch at 0.29 ms for 336 MB; profiles/r6_kernel_stats.csv.)
#define GLF_VPT 8
__global__ __launch_bounds__(256) void grid_load_flags_kernel(const uint4* __restrict__ src, uint4* __restrict__ grid,
                                                              int* __restrict__ scratch, int W, int F, int N, int C, int f_off, int nf,
                                                              long nvec_row) {
  const int Fp = F + 4, Wp = N + 4;
  const int f = blockIdx.y, w = blockIdx.z;
  const long row_g = (((long)w * Fp + f_off + f + 2) * Wp + 2) * C / 8;         // vector offset of the row's interior in the grid
  const uint4* s = src ? src + ((long)w * nf + f) * nvec_row : grid + row_g;
  uint4* d = grid + row_g;
  const long v0 = (long)blockIdx.x * (256 * GLF_VPT) + threadIdx.x;
  uint4 v[GLF_VPT];
#pragma unroll
  for (int k = 0; k < GLF_VPT; ++k) {
    const long i = v0 + k * 256;
    v[k] = i < nvec_row ? s[i] : make_uint4(0, 0, 0, 0);
  }
  unsigned any = 0;
#pragma unroll
  for (int k = 0; k < GLF_VPT; ++k) {
    const long i = v0 + k * 256;
    // (-0.0 counts as zero: the products it would enter are zeros either way)
    any |= (v[k].x | v[k].y | v[k].z | v[k].w) & 0x7fff7fffu;
    if (src && i < nvec_row) d[i] = v[k];
  }
  if (__ballot(any != 0) != 0 && (threadIdx.x & 63) == 0) atomicOr(scratch + w * Fp + f_off + f + 2, 1);
}

__global__ __launch_bounds__(256) void frame_flags_prefix_kernel(int* __restrict__ ps, int* __restrict__ scratch, int W, int Fp) {
  for (int ww = threadIdx.x; ww < W; ww += 256) {
    int run = 0;
    int* o = ps + (long)ww * (Fp + 1);
    o[0] = 0;
    for (int j = 0; j < Fp; ++j) {
      run += scratch[ww * Fp + j] != 0 ? 1 : 0;
      scratch[ww * Fp + j] = 0;
      o[j + 1] = run;
    }
  }
}

extern "C" int dfold_grid_load_flags(const void* src, void* grid, int32_t* ps, int32_t* scratch, int32_t W, int32_t F, int32_t N,
                                     int32_t C, int32_t f_off, int32_t nf, void* stream) {
  if (!grid || !ps || !scratch || W <= 0 || F <= 0 || N <= 0 || C <= 0 || f_off < 0 || nf <= 0 || f_off + nf > F) return DFOLD_EINVAL;
  if (((long)N * C) % 8 || (C % 8) || (((uintptr_t)src | (uintptr_t)grid) & 15) || W > 65535 || nf > 65535) return DFOLD_EINVAL;
  const long nvec = (long)N * C / 8;
  const unsigned bx = (unsigned)((nvec + 256 * GLF_VPT - 1) / (256 * GLF_VPT));
  DFOLD_LAUNCH(grid_load_flags_kernel, dim3(bx, (unsigned)nf, (unsigned)W), dim3(256), 0, (hipStream_t)stream, (const uint4*)src,
               (uint4*)grid, scratch, W, F, N, C, f_off, nf, nvec);
  if (dfold_check_launch() != DFOLD_OK) return DFOLD_ELAUNCH;
  DFOLD_LAUNCH(frame_flags_prefix_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, ps, scratch, W, F + 4);
  return dfold_check_launch();
}

// ---------------------------------------------------------------------------------------------
// row-block flags of a dense fp32 gradient (round 6): one wave per block of rows, then the prefix sums
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void row_block_flags_kernel(const float* __restrict__ G, int* __restrict__ scratch, long R, int C, long ld,
                                                              int block, int nblocks) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= nblocks) return;
  const long r0 = (long)b * block;
  const long rows = r0 + block <= R ? block : R - r0;
  unsigned any = 0;
  for (long i = lane; i < rows * C; i += 64) {
    const long r = i / C, c = i - r * C;
    any |= __float_as_uint(G[(r0 + r) * ld + c]) & 0x7fffffffu;
  }
  if (__ballot(any != 0) != 0 && lane == 0) scratch[b] = 1;
}

extern "C" int dfold_row_block_flags(const float* G, int32_t* ps, int32_t* scratch, int64_t R, int32_t C, int64_t ld, int32_t block,
                                     void* stream) {
  if (!G || !ps || !scratch || R <= 0 || C <= 0 || ld < C || block <= 0) return DFOLD_EINVAL;
  const long nb = (R + block - 1) / block;
  if (nb > (1L << 24)) return DFOLD_EINVAL;
  DFOLD_LAUNCH(row_block_flags_kernel, dim3((unsigned)((nb + 3) / 4)), dim3(256), 0, (hipStream_t)stream, G, scratch, (long)R, C, (long)ld,
               block, (int)nb);
  if (dfold_check_launch() != DFOLD_OK) return DFOLD_ELAUNCH;
  DFOLD_LAUNCH(frame_flags_prefix_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, ps, scratch, 1, (int)nb);
  return dfold_check_launch();
}

extern "C" int dfold_abi_version(void) { return DFOLD_ABI_VERSION; }
