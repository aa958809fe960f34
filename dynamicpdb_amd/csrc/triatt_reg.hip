// Triangle attention (openfold/model/triangular_attention.py:31-139, Attention openfold/model/primitives.py:219-243,
// 299-448) for c_in = 128, 4 heads x 32, N_res <= 512 -- third form (round 6): one wave owns 64 cells of a pair-tensor row.
//
// The whole-row kernel (triatt_fused.hip) keeps a 64 KB LayerNorm tile + the four projection tiles of a head in LDS: 141 KB,
// one workgroup per CU, every phase (LayerNorm loads | projections | attention | stores) behind a workgroup barrier of eight
// waves -- its counters say latency (51 % of the wave cycles waiting, matrix pipe busy 26 %, profiles/r4b_triatt_*).  Here:
//
//   * LayerNorm(x[i, 64 w .. 64 w + 63, :]) of wave w lives in REGISTERS, already in the MFMA operand layout (16 cells x 4
//     lane groups of 8 channels: the A and the B operand of v_mfma_f32_16x16x32_bf16 hold the same lane layout) -- 64 VGPRs;
//   * per head the wave projects its own cells four ways without leaving the register file: Q^T = W_q XN^T and G^T = W_g XN^T
//     come out of the matrix pipe with a lane holding (query, 4 + 4 channels) -- exactly a B operand of S^T = K Q^T once the
//     K tile uses the same channel order, and exactly the layout of O^T = V^T P^T for the gate; K^T and V go to LDS in the
//     operand layouts of the two attention products (one 16-byte write per lane and tile);
//   * the head's weights (q | k | v | g rows of the head: 32 KB) wait in LDS as ready-made operand fragments, requested by LDS-DMA
//     a whole attention phase ahead (W_o the same way behind the last head): a projection unit costs its 16 products, not an
//     L2 round trip per unit (the first version loaded fragments from global memory: 17 - 27 k cycles of projections per head,
//     7 k now);
//   * og_h = gate * O_h / l never touches LDS either: its accumulator layout is a B operand of out^T = W_o og^T -- V and G rows
//     are taken in the channel order pi(rho) = 8 ((rho & 15) >> 2) + 4 (rho >> 4) + (rho & 3), so that the k slots of og are
//     eight consecutive channels and a W_o fragment is 16 contiguous bytes; the four heads wait in 64 VGPRs for one product at
//     the end of the row;
//   * LDS = K [keys][32] + V^T [32][keys] + 32 weight fragments + mask bias = 66 KB at N_res <= 256: TWO workgroups (rows) per CU
//     whose phases interleave freely -- the loads of one row under the softmax of the other; two barriers per head (weights /
//     K, V^T visible; K, V^T complete / weight buffer free).
//
// Softmax: exp2 domain (W_q's product is scaled by scale * log2 e before rounding, the triangle bias arrives x log2 e),
// online over chunks of 64 keys (16 accumulator registers per 16-query tile + 16 that hold the bias blocks of the next chunk,
// requested before the chunk's arithmetic; the register budget of two waves per SIMD is 256, 128 of them LayerNorm output +
// gated O).  Pass 0 (dfold_triatt_bias_blocked, triatt_fused.hip) writes the blocked triangle bias as before.
// What was measured on the way (serialised loads, elimination builds, phase clock, two-tile form): HISTORY.md, round 6.
#include "dfold_common.h"
#include "../../include/dfold_hip.h"
#include <math.h>

typedef __attribute__((ext_vector_type(4))) unsigned tgu32x4;
typedef __attribute__((ext_vector_type(2))) unsigned tgu32x2;
#define TG_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#define TG_LOG2E 1.44269504088896341f
#ifndef TG_CT
#define TG_CT 4      // key tiles per softmax chunk
#endif

struct TriAttRegParams {
  const void* x;        // [B][N][N][128] fp32 | bf16
  const float* mask;    // [B][N][N] (coordinates of x)
  const float* gamma;   // LayerNorm
  const float* beta;
  const bf16_t* W;      // [512][128]: q | k | v | g rows
  const float* bcat;    // [512] biases of q | k | v | g
  const float* tri;     // [B][4][NP/16][NP/16][64][4] fp32, x log2(e): 16 x 16 blocks in accumulator order (pass 0)
  const bf16_t* Wo;     // [128][128]
  const float* bo;      // [128]
  void* out;            // [B][N][N][128] fp32 | bf16
  float* dbg;           // optional: row 0 of item 0 -> q|k|v|g of head 0 as fp32 [4][N][32] (tests)
  int B, N, NP, ending, x_bf16, out_bf16;
  float inf, scale, eps;
  int xflags;
};

template <int NW>
struct TGCfg {
  static constexpr int NK = 64 * NW;              // cells = keys = queries a workgroup covers
  static constexpr int KT = 4 * NW;               // key tiles of 16
  static constexpr int CT = TG_CT;                // key tiles per softmax chunk
  static constexpr int NCH = KT / CT;
  static constexpr int KBYTES = NK * 64;          // K [key][32 ch]: 64-byte rows, 16-byte chunk l4 = channels 4 l4 .. + 4 | 16 + 4 l4 .. + 4
  static constexpr int VPITCH = NK * 2 + 16;      // V^T [ch][keys]: per 32 keys, 16-byte chunk l4 = keys 4 l4 .. + 4 | 16 + 4 l4 .. + 4
  static constexpr int LDS_V = KBYTES;
  static constexpr int LDS_W = LDS_V + 32 * VPITCH;   // 32 operand fragments of 1 KB: the head's q | k | v | g weights, then W_o
  static constexpr int LDS_MB = LDS_W + 32768;
  static constexpr int LDS = LDS_MB + NK * 4;
  static_assert(KT % CT == 0, "whole chunks");
};

__device__ __forceinline__ float tg_sigm(float y) { return __builtin_amdgcn_rcpf(1.f + __expf(-y)); }
// combine the four lane groups (lanes l15, l15 + 16, l15 + 32, l15 + 48) of a value: gfx950's v_permlane32_swap / v_permlane16_swap
// (VALU; __shfl_xor is a ds_bpermute: an LDS round trip with an s_waitcnt behind it, twice per 128-key chunk)
__device__ __forceinline__ float tg_xmax(float v) {
  auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float tg_xsum(float v) {
  auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ tgu32x4 tg_pack8(const f32x4 a, const f32x4 b) {
  return (tgu32x4){pack2bf_hw(a[0], a[1]), pack2bf_hw(a[2], a[3]), pack2bf_hw(b[0], b[1]), pack2bf_hw(b[2], b[3])};
}

typedef __attribute__((address_space(3))) void* tg_lds_ptr_t;
// weight fragments of projection pj, channel tile ct from the LDS weight buffer: fragment (pj, ct, ks) is 1 KB in lane order
__device__ __forceinline__ void tg_load_w(const char* ldsW, int pj, int ct, int lane, bf16x8 (&wf)[4]) {
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) wf[ks] = *(const bf16x8*)(ldsW + ((pj * 2 + ct) * 4 + ks) * 1024 + lane * 16);
}
// acc[t] (lane: column = cell 16 t + l15, rows = channels 16 ct + 4 l4 + r) = W_h XN^T
__device__ __forceinline__ void tg_proj_rows(const bf16x8 (&wf)[4], const bf16x8 (&xn)[4][4], f32x4 (&acc)[4]) {
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = TG_MFMA(wf[ks], xn[t][ks], acc[t]);
}
// acc[t] (lane: column = channel 16 ct + l15, rows = cells 16 t + 4 l4 + r) = XN W_h^T
__device__ __forceinline__ void tg_proj_cols(const bf16x8 (&wf)[4], const bf16x8 (&xn)[4][4], f32x4 (&acc)[4]) {
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = TG_MFMA(xn[t][ks], wf[ks], acc[t]);
}
__device__ __forceinline__ tgu32x2 tg_pack4(const f32x4 a) { return (tgu32x2){pack2bf_hw(a[0], a[1]), pack2bf_hw(a[2], a[3])}; }

template <int NW, bool TAP>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(2, 2))) void triatt_reg_kernel(const TriAttRegParams p) {
  using C = TGCfg<NW>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* const ldsMB = (float*)(smem + C::LDS_MB);
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, l4 = lane >> 4;
  const int N = p.N, nt16 = p.NP >> 4;
  // XCD-aware work ids (blockIdx round-robins over the 8 XCDs): the rows of one item share its triangle bias in one L2
  const unsigned nwg = gridDim.x, bid = blockIdx.x;
  const unsigned xq = nwg >> 3, xr = nwg & 7, xcd = bid & 7, xidx = bid >> 3;
  const unsigned lid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + xidx;
  const int i = (int)(lid % (unsigned)N), b = (int)(lid / (unsigned)N);
  // cell (i, j) of the operator's coordinates x' (= x, or x^T for the ending node) in the memory of x / mask / out
  const long cs = p.ending ? (long)N : 1L;                                   // cell stride along j
  const long c0 = p.ending ? ((long)b * N) * N + i : ((long)b * N + i) * N;  // cell (i, 0)
  const float sl2 = p.scale * TG_LOG2E;
  const bool tap = TAP && lid == 0;      // (tests: the instantiation with the debug tap)

  // phase clock of the tap instantiation (xflags & 64; scripts/triatt_phase_times.py): s_memtime at every phase boundary of the
  // waves of one workgroup in mid grid -> dbg[4 N 32 + 64 w + k]
  long tstamp0 = 0;
  int tsn = 0;
  const bool prof = TAP && (p.xflags & 64) && lid == 777 && lane == 0;
#define TG_STAMP()                                                                   \
  do {                                                                               \
    if (TAP && (p.xflags & 64)) {                                                    \
      const long tt = (long)__builtin_amdgcn_s_memtime();                            \
      if (tsn == 0) tstamp0 = tt;                                                    \
      if (prof) p.dbg[4 * N * 32 + w * 64 + tsn] = (float)(tt - tstamp0);           \
      ++tsn;                                                                         \
    }                                                                                \
  } while (0)
  TG_STAMP();
  float mb_own = -INFINITY;                 // keys past the end of the row
  if (tid < N) mb_own = p.inf * (p.mask[c0 + tid * cs] - 1.f) * TG_LOG2E;
  ldsMB[tid] = mb_own;

  char* const bufK = smem;
  char* const bufV = smem + C::LDS_V;
  char* const ldsW = smem + C::LDS_W;
  const int kswz = (l15 >> 2) & 3;
  const int pi15 = 8 * (l15 >> 2) + (l15 & 3);          // pi(ct * 16 + l15) - 4 ct
  // request the weights of head h (h == 4: W_o) into the LDS weight buffer: fragment f of this wave's share, 1 KB per
  // instruction; source = scalar base + one of two per-lane byte offsets (plain rows | the pi order of the V and G rows)
  const unsigned voff_plain = (unsigned)(l15 * 128 + l4 * 8) * 2u, voff_pi = (unsigned)(pi15 * 128 + l4 * 8) * 2u;
  auto request_weights = [&](int h) {
#pragma unroll
    for (int f0 = 0; f0 < 32; f0 += NW) {
      const int f = f0 + w;                 // (wave-uniform)
      if (32 % NW == 0 || f < 32) {
        const char* base;
        unsigned voff = voff_plain;
        if (h < 4) {
          const int pj = f >> 3, ct = (f >> 2) & 1, ks = f & 3;
          base = (const char*)(p.W + (long)(pj * 128 + h * 32 + (pj >= 2 ? 4 * ct : 16 * ct)) * 128 + ks * 32);
          voff = pj >= 2 ? voff_pi : voff_plain;
        } else {
          const int hh = f >> 3, nb = f & 7;              // fragment (head, out-channel tile): W_o[16 nb + l15][32 hh + 8 l4 .. + 8]
          base = (const char*)(p.Wo + (long)(nb * 16) * 128 + hh * 32);
        }
        __builtin_amdgcn_global_load_lds((const void*)(base + voff), (tg_lds_ptr_t)(ldsW + f * 1024), 16, 0, 0);
      }
    }
  };
  request_weights(0);      // lands under the LayerNorm loads

  // ---- LayerNorm of the wave's 64 cells -> operand fragments xn[t][ks]: lane (l15, l4) = cell 64 w + 16 t + l15, channels
  //      32 ks + 8 l4 .. + 8.  Cells past the end of the row: zero rows (K, V finite; their keys carry a -inf mask bias) ----
  bf16x8 xn[4][4];
  {
    f32x4 gam[4][2], bet[4][2];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        gam[ks][hh] = *(const f32x4*)(p.gamma + ks * 32 + l4 * 8 + hh * 4);
        bet[ks][hh] = *(const f32x4*)(p.beta + ks * 32 + l4 * 8 + hh * 4);
      }
    f32x4 v[4][4][2];
    if (!p.x_bf16) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int cell = w * 64 + t * 16 + l15;
        const float* src = (const float*)p.x + (c0 + (long)min(cell, N - 1) * cs) * 128 + l4 * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          v[t][ks][0] = *(const f32x4*)(src + ks * 32);
          v[t][ks][1] = *(const f32x4*)(src + ks * 32 + 4);
        }
      }
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int cell = w * 64 + t * 16 + l15;
        const bf16_t* src = (const bf16_t*)p.x + (c0 + (long)min(cell, N - 1) * cs) * 128 + l4 * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const uint4 u = *(const uint4*)(src + ks * 32);
          v[t][ks][0] = (f32x4){bf_lo(u.x), bf_hi(u.x), bf_lo(u.y), bf_hi(u.y)};
          v[t][ks][1] = (f32x4){bf_lo(u.z), bf_hi(u.z), bf_lo(u.w), bf_hi(u.w)};
        }
      }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const bool live = w * 64 + t * 16 + l15 < N;
      float s1 = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) s1 += (v[t][ks][hh][0] + v[t][ks][hh][1]) + (v[t][ks][hh][2] + v[t][ks][hh][3]);
      const float mean = tg_xsum(s1) * (1.f / 128.f);
      float q2 = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[t][ks][hh][e] -= mean;
            q2 = __builtin_fmaf(v[t][ks][hh][e], v[t][ks][hh][e], q2);
          }
      const float rstd = rsqrtf(tg_xsum(q2) * (1.f / 128.f) + p.eps);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        f32x4 y[2];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            y[hh][e] = live ? __builtin_fmaf(v[t][ks][hh][e] * rstd, gam[ks][hh][e], bet[ks][hh][e]) : 0.f;
        xn[t][ks] = __builtin_bit_cast(bf16x8, tg_pack8(y[0], y[1]));
      }
    }
  }
  TG_STAMP();     // 1: LayerNorm done

  // ---- projections.  The head's weights wait in LDS as ready-made operand fragments (DMA: no registers, requested a whole
  //      attention phase ahead), eight units of 16 output channels each: K | V -> LDS, Q | G -> registers.
  //      K rows in LDS: the 16-byte chunk l4 of key row k sits at chunk l4 ^ ((k >> 2) & 3) (any 16 consecutive keys read
  //      conflict-free).  V^T row rho (and G's, O's accumulator row rho) holds head channel pi(rho) = 8 ((rho & 15) >> 2) +
  //      4 (rho >> 4) + (rho & 3): the k slots of og then run over channels 8 l4 .. + 8 and a W_o fragment is 16 contiguous bytes ----
  tgu32x4 qf[4], gf[4];
  auto unit_k = [&](int h, int ct, const bf16x8 (&wf)[4]) {
    f32x4 acc[4];
    tg_proj_rows(wf, xn, acc);
    const f32x4 bk = *(const f32x4*)(p.bcat + 128 + h * 32 + ct * 16 + l4 * 4);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const f32x4 a = acc[t] + bk;
      const int key = w * 64 + t * 16 + l15;
      *(tgu32x2*)(bufK + key * 64 + ((l4 ^ kswz) << 4) + ct * 8) = tg_pack4(a);
      if (tap && h == 0 && key < N) {
#pragma unroll
        for (int r = 0; r < 4; ++r) p.dbg[((long)N + key) * 32 + ct * 16 + l4 * 4 + r] = a[r];
      }
    }
  };
  auto unit_v = [&](int h, int ct, const bf16x8 (&wf)[4]) {
    f32x4 acc[4];
    tg_proj_cols(wf, xn, acc);
    const float bv = p.bcat[256 + h * 32 + pi15 + 4 * ct];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      f32x4 a0 = acc[2 * j], a1 = acc[2 * j + 1];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        a0[r] += bv;
        a1[r] += bv;
      }
      *(tgu32x4*)(bufV + (ct * 16 + l15) * C::VPITCH + (2 * w + j) * 64 + l4 * 16) = tg_pack8(a0, a1);
      if (tap && h == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int k0 = w * 64 + (2 * j) * 16 + l4 * 4 + r, k1 = k0 + 16;
          if (k0 < N) p.dbg[((long)2 * N + k0) * 32 + pi15 + 4 * ct] = a0[r];
          if (k1 < N) p.dbg[((long)2 * N + k1) * 32 + pi15 + 4 * ct] = a1[r];
        }
      }
    }
  };
  auto unit_q = [&](int h, int ct, const bf16x8 (&wf)[4]) {
    f32x4 acc[4];
    tg_proj_rows(wf, xn, acc);
    const f32x4 bq = *(const f32x4*)(p.bcat + h * 32 + ct * 16 + l4 * 4);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const f32x4 a = acc[t] + bq;
      const tgu32x2 pk = tg_pack4(a * sl2);
      qf[t][2 * ct] = pk.x;
      qf[t][2 * ct + 1] = pk.y;
      const int q = w * 64 + t * 16 + l15;
      if (tap && h == 0 && q < N) {
#pragma unroll
        for (int r = 0; r < 4; ++r) p.dbg[(long)q * 32 + ct * 16 + l4 * 4 + r] = a[r];
      }
    }
  };
  auto unit_g = [&](int h, int ct, const bf16x8 (&wf)[4]) {      // accumulator rows 16 ct + 4 l4 + r = channels 8 l4 + 4 ct + r
    f32x4 acc[4];
    tg_proj_rows(wf, xn, acc);
    const f32x4 bg = *(const f32x4*)(p.bcat + 384 + h * 32 + 8 * l4 + 4 * ct);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      f32x4 a = acc[t] + bg;
#pragma unroll
      for (int r = 0; r < 4; ++r) a[r] = tg_sigm(a[r]);
      const tgu32x2 pk = tg_pack4(a);
      gf[t][2 * ct] = pk.x;
      gf[t][2 * ct + 1] = pk.y;
      const int q = w * 64 + t * 16 + l15;
      if (tap && h == 0 && q < N) {
#pragma unroll
        for (int r = 0; r < 4; ++r) p.dbg[((long)3 * N + q) * 32 + 8 * l4 + 4 * ct + r] = a[r];
      }
    }
  };

  tgu32x4 ogf[4][4];                       // [query tile][head]: gated, normalised O of the head as a B operand (k = head channels)
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int hh = 0; hh < 4; ++hh) ogf[t][hh] = (tgu32x4){0u, 0u, 0u, 0u};
  bool masked = false;

#pragma unroll 1
  for (int h = 0; h < 4; ++h) {
    TG_STAMP();   // 2 + 5 h: head start
    // this wave's share of the head's weights has landed; behind the barrier everybody's has, and every wave has left K / V^T
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (h == 0)
      masked = __syncthreads_or(mb_own != 0.f) != 0;   // no masked key, no key past the end: the mask bias is never added
    else
      __syncthreads();
    TG_STAMP();   // 3 + 5 h: weights in LDS
    {
      bf16x8 wa[4], wb[4];
      tg_load_w(ldsW, 1, 0, lane, wa);
      tg_load_w(ldsW, 1, 1, lane, wb);
      unit_k(h, 0, wa);
      tg_load_w(ldsW, 2, 0, lane, wa);
      unit_k(h, 1, wb);
      tg_load_w(ldsW, 2, 1, lane, wb);
      unit_v(h, 0, wa);
      tg_load_w(ldsW, 0, 0, lane, wa);
      unit_v(h, 1, wb);
      tg_load_w(ldsW, 0, 1, lane, wb);
      unit_q(h, 0, wa);
      tg_load_w(ldsW, 3, 0, lane, wa);
      unit_q(h, 1, wb);
      tg_load_w(ldsW, 3, 1, lane, wb);
      unit_g(h, 0, wa);
      unit_g(h, 1, wb);
    }
    TG_STAMP();   // 4 + 5 h: projections done
    __syncthreads();        // K / V^T of head h complete; the weight buffer is free
    request_weights(h + 1);
    TG_STAMP();   // 5 + 5 h: barrier passed, next weights requested

    // ---- attention of head h, one 16-query tile at a time; the bias blocks of the NEXT chunk (of the next tile behind the
    //      tile's last chunk) are requested before the chunk's own arithmetic ----
    f32x4 tn[C::CT];
    auto tri_request = [&](const float* trow_, int c_) {
#pragma unroll
      for (int kb = 0; kb < C::CT; ++kb)     // one contiguous KB per wave instruction
        tn[kb] = *(const f32x4*)(trow_ + (long)min(c_ * C::CT + kb, nt16 - 1) * 256 + lane * 4);
    };
    auto tri_row = [&](int t_) {             // (wave-uniform: scalar base + lane offset; query tiles past the end: any valid block)
      return p.tri + ((((long)b * 4 + h) * nt16 + min(w * 4 + t_, nt16 - 1)) * nt16) * 256;
    };
    tri_request(tri_row(0), 0);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float* const trow = tri_row(t);
      const float* const trow_next = tri_row(t < 3 ? t + 1 : t);
      const bf16x8 qb = __builtin_bit_cast(bf16x8, qf[t]);
      float m_run = -INFINITY, l_run = 0.f;                 // l_run: this lane's share (keys 4 l4 .. + 4 of every tile)
      f32x4 oacc[2];
      oacc[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
      oacc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
      for (int c = 0; c < C::NCH; ++c) {
        f32x4 s[C::CT];
#pragma unroll
        for (int kb = 0; kb < C::CT; ++kb) s[kb] = tn[kb];
        {
          const bool last = c + 1 == C::NCH;
          tri_request(last ? trow_next : trow, last ? 0 : c + 1);
        }
        {
          bf16x8 kf[C::CT];
#pragma unroll
          for (int kb = 0; kb < C::CT; ++kb) kf[kb] = *(const bf16x8*)(bufK + ((c * C::CT + kb) * 16 + l15) * 64 + ((l4 ^ kswz) << 4));
#pragma unroll
          for (int kb = 0; kb < C::CT; ++kb) s[kb] = TG_MFMA(kf[kb], qb, s[kb]);
        }
        // V^T fragments of the chunk: requested now, they arrive under the exponentials
        bf16x8 vf[C::CT / 2][2];
#pragma unroll
        for (int k2 = 0; k2 < C::CT / 2; ++k2)
#pragma unroll
          for (int ct = 0; ct < 2; ++ct)
            vf[k2][ct] = *(const bf16x8*)(bufV + (ct * 16 + l15) * C::VPITCH + (c * (C::CT / 2) + k2) * 64 + l4 * 16);
        if (masked) {
#pragma unroll
          for (int kb = 0; kb < C::CT; ++kb) s[kb] += *(const f32x4*)(ldsMB + (c * C::CT + kb) * 16 + l4 * 4);
        }
        float mxk[C::CT];                    // (trees, not chains: the waves are latency-bound)
#pragma unroll
        for (int kb = 0; kb < C::CT; ++kb) mxk[kb] = fmaxf(fmaxf(s[kb][0], s[kb][1]), fmaxf(s[kb][2], s[kb][3]));
        float mx = mxk[0];
#pragma unroll
        for (int kb = 1; kb < C::CT; ++kb) mx = fmaxf(mx, mxk[kb]);
        const float m_new = fmaxf(m_run, tg_xmax(mx));      // finite from chunk 0 on: a real key's mask bias is finite
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        float smk[C::CT];
#pragma unroll
        for (int kb = 0; kb < C::CT; ++kb) {
#pragma unroll
          for (int r = 0; r < 4; ++r) s[kb][r] = __builtin_amdgcn_exp2f(s[kb][r] - m_new);
          smk[kb] = (s[kb][0] + s[kb][1]) + (s[kb][2] + s[kb][3]);
        }
        float sum = smk[0];
#pragma unroll
        for (int kb = 1; kb < C::CT; ++kb) sum += smk[kb];
        l_run = l_run * alpha + sum;
        m_run = m_new;
        oacc[0] *= alpha;
        oacc[1] *= alpha;
        // O^T[rho][q] += V^T[rho][keys] P^T[keys][q]: k slot e of lane group l4 <-> key (2 ks + (e >> 2)) * 16 + 4 l4 + (e & 3)
#pragma unroll
        for (int k2 = 0; k2 < C::CT / 2; ++k2) {
          const bf16x8 pb = __builtin_bit_cast(bf16x8, tg_pack8(s[2 * k2], s[2 * k2 + 1]));
#pragma unroll
          for (int ct = 0; ct < 2; ++ct) oacc[ct] = TG_MFMA(vf[k2][ct], pb, oacc[ct]);
        }
      }
      const float inv = __builtin_amdgcn_rcpf(tg_xsum(l_run));
      f32x4 o0, o1;
      o0[0] = oacc[0][0] * inv * bf_lo(gf[t][0]); o0[1] = oacc[0][1] * inv * bf_hi(gf[t][0]);
      o0[2] = oacc[0][2] * inv * bf_lo(gf[t][1]); o0[3] = oacc[0][3] * inv * bf_hi(gf[t][1]);
      o1[0] = oacc[1][0] * inv * bf_lo(gf[t][2]); o1[1] = oacc[1][1] * inv * bf_hi(gf[t][2]);
      o1[2] = oacc[1][2] * inv * bf_lo(gf[t][3]); o1[3] = oacc[1][3] * inv * bf_hi(gf[t][3]);
      ogf[t][0] = ogf[t][1];                // (a shift register: static indices under the rolled head loop)
      ogf[t][1] = ogf[t][2];
      ogf[t][2] = ogf[t][3];
      ogf[t][3] = tg_pack8(o0, o1);         // k slot e of lane group l4 = head channel 8 l4 + e
    }
    TG_STAMP();   // 6 + 5 h: attention done
  }

  // ---- out^T = W_o og^T + b_o: A = W_o fragments from LDS (rows = out channels), B = og; accumulator lane (l15, l4) = query
  //      l15, out channels 16 nb + 4 l4 .. + 4.  Rows leave through a wave-private LDS tile (K / V^T are idle) as whole
  //      256-byte half cells: a lane-per-query store touches sixteen 128-byte lines per instruction ----
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  TG_STAMP();     // 22: W_o in LDS
  {
    constexpr int OP = 272;                 // staged half row: 64 fp32 + 16 bytes
    char* const st = smem + w * (16 * OP);
#pragma unroll
    for (int half = 0; half < 2; ++half) {   // (out channels in two halves of 64: 64 weight registers at a time)
      bf16x8 wo[4][4];
#pragma unroll
      for (int hh = 0; hh < 4; ++hh)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) wo[hh][nb] = *(const bf16x8*)(ldsW + (hh * 8 + half * 4 + nb) * 1024 + lane * 16);
      f32x4 bo[4];
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) bo[nb] = *(const f32x4*)(p.bo + (half * 4 + nb) * 16 + l4 * 4);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        f32x4 acc[4];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) acc[nb] = bo[nb];
#pragma unroll
        for (int hh = 0; hh < 4; ++hh)
#pragma unroll
          for (int nb = 0; nb < 4; ++nb) acc[nb] = TG_MFMA(wo[hh][nb], __builtin_bit_cast(bf16x8, ogf[t][hh]), acc[nb]);
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) *(f32x4*)(st + l15 * OP + (nb * 16 + l4 * 4) * 4) = acc[nb];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        // 16 rows x 256 bytes of this half: lane -> (row = id >> 4, 16-byte piece id & 15), four instructions
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int id = lane + 64 * j, row = id >> 4, pc = id & 15;
          const f32x4 vv = *(const f32x4*)(st + row * OP + pc * 16);
          const int q = w * 64 + t * 16 + row;
          if (q < N) {
            const long cell = c0 + (long)q * cs;
            if (!p.out_bf16)
              *(f32x4*)((float*)p.out + cell * 128 + half * 64 + pc * 4) = vv;
            else
              *(tgu32x2*)((bf16_t*)p.out + cell * 128 + half * 64 + pc * 4) = (tgu32x2){pack2bf_hw(vv[0], vv[1]), pack2bf_hw(vv[2], vv[3])};
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
  TG_STAMP();     // 23: out written
}

template <int NW>
static int tg_launch(const TriAttRegParams& p, hipStream_t stream) {
  if (p.dbg != nullptr) {
    DFOLD_MAX_LDS_ONCE((triatt_reg_kernel<NW, true>), TGCfg<NW>::LDS);
    DFOLD_LAUNCH((triatt_reg_kernel<NW, true>), dim3((unsigned)((long)p.B * p.N)), dim3(64 * NW), TGCfg<NW>::LDS, stream, p);
  } else {
    DFOLD_MAX_LDS_ONCE((triatt_reg_kernel<NW, false>), TGCfg<NW>::LDS);
    DFOLD_LAUNCH((triatt_reg_kernel<NW, false>), dim3((unsigned)((long)p.B * p.N)), dim3(64 * NW), TGCfg<NW>::LDS, stream, p);
  }
  return dfold_check_launch();
}

extern "C" int dfold_triatt_reg_fwd(const void* x, int32_t x_is_bf16, const float* mask, const float* ln_gamma,
                                    const float* ln_beta, const void* w_cat_bf16, const float* bias_cat, const float* tri,
                                    const void* w_o_bf16, const float* b_o, void* out, int32_t out_is_bf16, float* dbg,
                                    int32_t dbg_phase_clock, int32_t B, int32_t N, int32_t NP, int32_t ending, float inf, float scale,
                                    float eps, void* stream) {
  if (!x || !mask || !ln_gamma || !ln_beta || !w_cat_bf16 || !bias_cat || !tri || !w_o_bf16 || !b_o || !out) return DFOLD_EINVAL;
  if (B <= 0 || N <= 0 || N > 512 || NP < N || (NP & 63) || (long)B * N > 0x7fffffffL) return DFOLD_EINVAL;
  TriAttRegParams p;
  p.x = x; p.mask = mask; p.gamma = ln_gamma; p.beta = ln_beta; p.W = (const bf16_t*)w_cat_bf16; p.bcat = bias_cat; p.tri = tri;
  p.Wo = (const bf16_t*)w_o_bf16; p.bo = b_o; p.out = out; p.dbg = dbg; p.B = B; p.N = N; p.NP = NP; p.ending = ending ? 1 : 0;
  p.x_bf16 = x_is_bf16 ? 1 : 0; p.out_bf16 = out_is_bf16 ? 1 : 0; p.inf = inf; p.scale = scale; p.eps = eps;
  p.xflags = (dbg != nullptr && dbg_phase_clock) ? 64 : 0;
  if (N <= 128) return tg_launch<2>(p, (hipStream_t)stream);
  if (N <= 256) return tg_launch<4>(p, (hipStream_t)stream);
  if (N <= 384) return tg_launch<6>(p, (hipStream_t)stream);
  return tg_launch<8>(p, (hipStream_t)stream);
}
