// Pair-value side of the IPA attention (src/model/ipa_pytorch_dynamic.py:498-502 and its autograd), gfx950 (MI355X), round 6.
//
//   forward   o_pair[b,f,i,h,c] = sum_j P[b,f,h,i,j] * pz[b,i,j,c] + b_dz[c]            (c < 32, pz = down_z(z) without its bias)
//   backward  dP[b,f,h,i,j]     = sum_c do_pair[b,f,i,h,c] * pz[b,i,j,c]                 (the pair-value term of dL/dP)
//
// Per (window b, query residue i) these are a [F H, N] x [N, 32] and a [F H, 32] x [32, N] product -- 4 MFLOP against 150 KB of
// operands: streaming work.  On the contraction engine (gemm_bf16.hip) they ran as batched GEMMs with 32-wide / 32-deep tiles at
// 25 - 37 TFLOP/s, 230 and 340 us per trunk block at config 3, i.e. at a fifth of what their bytes cost.  Here one workgroup takes
// one (b, i): every MFMA operand fragment is a 16-byte global load straight into the register layout v_mfma_f32_32x32x16_bf16
// wants (a row of P is contiguous in j, a row of pz^T in j, a row of pz / do_pair in c) -- no LDS, no transposed copies -- and
// the products are formed TRANSPOSED (D = B^T-side operand first) so that a lane ends up with four consecutive output elements
// of one row: 8-byte stores.
#include "dfold_common.h"
#include "../../include/dfold_hip.h"

typedef __attribute__((ext_vector_type(4))) unsigned ip_u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned ip_u32x2;
#define IP_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)

__device__ __forceinline__ bf16x8 ip_load8(const bf16_t* p) {
  const ip_u32x4 v = *(const ip_u32x4*)p;
  return __builtin_bit_cast(bf16x8, v);
}

// ---- forward: out[(b f i)][c_off + h * 32 + c] --------------------------------------------------------------------------
// grid: (B * N, ceil(F H / 256)); 4 waves, wave w: rows r = blockIdx.y * 256 + 64 w + {0 .. 63} of the (f, h) axis.
// D[c][r] = sum_j pzT[c][j] P[r][j]: A operand = pz^T (32 channels x 16 keys), B operand = P rows.
__global__ __launch_bounds__(256) void ipa_pair_value_fwd_kernel(const bf16_t* __restrict__ P, const bf16_t* __restrict__ pzT,
                                                                 const float* __restrict__ b_dz, bf16_t* __restrict__ out, int B, int F,
                                                                 int N, int H, long ld, long c_off) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int bi = blockIdx.x, b = bi / N, i = bi - b * N;
  const int FH = F * H;
  const int l31 = lane & 31, kh = lane >> 5;
  const long NN = (long)N * N;
  const bf16_t* pa = pzT + ((long)bi * 32 + l31) * N + kh * 8;                   // A fragments: channel l31, keys j0 + 8 kh ..
  const int r_base = blockIdx.y * 256 + w * 64;
  const bf16_t* pb[2];
  int rr[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int r = r_base + 32 * t + l31;
    rr[t] = r;
    const int rc = r < FH ? r : FH - 1;                                           // rows past F H: any valid row, never stored
    pb[t] = P + ((long)b * FH + rc) * NN + (long)i * N + kh * 8;
  }
  f32x16 acc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
  for (int j0 = 0; j0 < N; j0 += 64) {
    // four K16 steps at a time: all twelve loads in flight before the first MFMA
    bf16x8 fa[4], fb[2][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int j = j0 + 16 * s < N ? j0 + 16 * s : 0;                            // (N % 64 != 0: the surplus steps re-read step 0 ...
      fa[s] = ip_load8(pa + j);
      fb[0][s] = ip_load8(pb[0] + j);
      fb[1][s] = ip_load8(pb[1] + j);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (j0 + 16 * s < N) {                                                      //  ... and are not accumulated)
        acc[0] = IP_MFMA(fa[s], fb[0][s], acc[0]);
        acc[1] = IP_MFMA(fa[s], fb[1][s], acc[1]);
      }
    }
  }
  // D layout: column = r (lane & 31), rows c = 8 g + 4 kh + {0..3} for register group g = e >> 2: four consecutive channels
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int r = rr[t];
    if (r >= FH) continue;
    const int f = r / H, h = r - f * H;
    bf16_t* dst = out + (((long)b * F + f) * N + i) * ld + c_off + h * 32 + 4 * kh;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int c = 8 * g + 4 * kh;
      const float v0 = acc[t][4 * g] + b_dz[c], v1 = acc[t][4 * g + 1] + b_dz[c + 1];
      const float v2 = acc[t][4 * g + 2] + b_dz[c + 2], v3 = acc[t][4 * g + 3] + b_dz[c + 3];
      ip_u32x2 o = {pack2bf_hw(v0, v1), pack2bf_hw(v2, v3)};
      *(ip_u32x2*)(dst + 8 * g) = o;
    }
  }
}

// ---- backward: dP[(b f h)][i][j] ------------------------------------------------------------------------------------------
// grid: (B * N, ceil(F H / 256)); wave w: rows r = blockIdx.y * 256 + 64 w + {0 .. 63}; all keys j in tiles of 32.
// D[j][r] = sum_c pz[j][c] dop[r][c]: A operand = pz rows (keys), B operand = do_pair rows; K = 32 = two K16 steps.
__global__ __launch_bounds__(256) void ipa_pair_value_bwd_kernel(const bf16_t* __restrict__ dop, const bf16_t* __restrict__ pz,
                                                                 bf16_t* __restrict__ dP, int B, int F, int N, int H, long ld_dop) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int bi = blockIdx.x, b = bi / N, i = bi - b * N;
  const int FH = F * H;
  const int l31 = lane & 31, kh = lane >> 5;
  const long NN = (long)N * N;
  const int r_base = blockIdx.y * 256 + w * 64;
  bf16x8 fb[2][2];
  int rr[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int r = r_base + 32 * t + l31;
    rr[t] = r;
    const int rc = r < FH ? r : FH - 1;
    const int f = rc / H, h = rc - f * H;
    const bf16_t* src = dop + (((long)b * F + f) * N + i) * ld_dop + h * 32 + kh * 8;
    fb[t][0] = ip_load8(src);
    fb[t][1] = ip_load8(src + 16);
  }
  const bf16_t* pa = pz + ((long)bi * N + l31) * 32 + kh * 8;                     // A fragments: key jt * 32 + l31, channels 8 kh ..
  const int ntile = (N + 31) / 32;
  for (int jt = 0; jt < ntile; jt += 2) {
    // two key tiles at a time: their four A loads in flight together
    bf16x8 fa[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      int jrow = (jt + u) * 32 + l31;
      jrow = jrow < N ? jrow : N - 1;                                             // (keys past N: any valid row, never stored)
      const bf16_t* src = pz + ((long)bi * N + jrow) * 32 + kh * 8;
      fa[u][0] = ip_load8(src);
      fa[u][1] = ip_load8(src + 16);
    }
    (void)pa;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (jt + u >= ntile) continue;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        acc = IP_MFMA(fa[u][0], fb[t][0], acc);
        acc = IP_MFMA(fa[u][1], fb[t][1], acc);
        const int r = rr[t];
        if (r >= FH) continue;
        // D layout: column = r, rows j = 8 g + 4 kh + {0..3}
        bf16_t* dst = dP + ((long)b * FH + r) * NN + (long)i * N + (jt + u) * 32 + 4 * kh;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int j = (jt + u) * 32 + 8 * g + 4 * kh;
          if (j + 3 < N) {
            ip_u32x2 o = {pack2bf_hw(acc[4 * g], acc[4 * g + 1]), pack2bf_hw(acc[4 * g + 2], acc[4 * g + 3])};
            *(ip_u32x2*)(dst + 8 * g) = o;
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (j + q < N) dst[8 * g + q] = f2bf_hw(acc[4 * g + q]);
          }
        }
      }
    }
  }
}

extern "C" int dfold_ipa_pair_value_fwd(const void* P_bf16, const void* pzT_bf16, const float* b_dz, void* out_bf16, int32_t B,
                                        int32_t F, int32_t N, int32_t H, int64_t ld, int64_t c_off, void* stream) {
  if (!P_bf16 || !pzT_bf16 || !b_dz || !out_bf16 || B <= 0 || F <= 0 || N <= 0 || H <= 0) return DFOLD_EINVAL;
  if ((N & 15) || (ld & 3) || (c_off & 3) || ld < c_off + (long)H * 32) return DFOLD_EINVAL;
  if ((((uintptr_t)P_bf16 | (uintptr_t)pzT_bf16) & 15) || ((uintptr_t)out_bf16 & 7)) return DFOLD_EINVAL;
  dim3 grid((unsigned)((long)B * N), (unsigned)((F * H + 255) / 256));
  DFOLD_LAUNCH(ipa_pair_value_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)P_bf16, (const bf16_t*)pzT_bf16, b_dz,
               (bf16_t*)out_bf16, B, F, N, H, (long)ld, (long)c_off);
  return dfold_check_launch();
}

extern "C" int dfold_ipa_pair_value_bwd(const void* do_pair_bf16, const void* pz_bf16, void* dP_bf16, int32_t B, int32_t F, int32_t N,
                                        int32_t H, int64_t ld_dop, void* stream) {
  if (!do_pair_bf16 || !pz_bf16 || !dP_bf16 || B <= 0 || F <= 0 || N <= 0 || H <= 0) return DFOLD_EINVAL;
  if ((N & 3) || (ld_dop & 7) || ld_dop < (long)H * 32) return DFOLD_EINVAL;
  if ((((uintptr_t)do_pair_bf16 | (uintptr_t)pz_bf16) & 15) || ((uintptr_t)dP_bf16 & 7)) return DFOLD_EINVAL;
  dim3 grid((unsigned)((long)B * N), (unsigned)((F * H + 255) / 256));
  DFOLD_LAUNCH(ipa_pair_value_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)do_pair_bf16, (const bf16_t*)pz_bf16,
               (bf16_t*)dP_bf16, B, F, N, H, (long)ld_dop);
  return dfold_check_launch();
}
