// Triangle attention (openfold/model/triangular_attention.py:31-139, Attention openfold/model/primitives.py:219-243,
// 299-448) for c_in = 128, 4 heads x 32, N_res <= 256: LayerNorm + q|k|v|g projections + gated attention + linear_o of one
// pair-tensor ROW in one workgroup -- q, k, v, g never exist in HBM.
//
// The two-kernel form (pair_fused.hip: projection kernel -> core kernel) moves x (fp32) + q|k|v|g (4 x bf16) + out per
// call: 1.5 x the pair tensor out and back in between the two launches.  The projections of row i are consumed by row i
// only, so one workgroup per (batch item, row) can keep them on chip:
//
//   pass 0 (dfold_triatt_proj_fwd with null q/k/v/gate): the triangle bias tri[h][q][k] = w_tri[h] . LN(x[q][k]) needs every
//           cell before any row can start -- one streaming pass over x that writes 4 N^2 floats;
//   this kernel, per (b, i): LN(x[i,:,:]) -> bf16 tile in LDS (N x 128);  per head h:  q_h | k_h | v_h | g_h = LN . W_h^T on
//           MFMA (wave w computes 16 of the 128 output channels of the head for ALL cells: its weight fragments are loaded
//           once) -> K [N][32], V^T [32][N], Q, sigmoid(G) tiles in LDS;  S^T = K Q^T (+ mask bias as accumulator init, +
//           triangle bias), exact softmax over the row's keys in registers (N <= 256: 16 key tiles x 4 registers per
//           16-query tile), O^T = V^T P^T with the probabilities straight from the accumulators (B-operand layout);
//           og = O * g;  out += og_h W_o[:, h]^T accumulated over the heads in registers;  + b_o -> out row.
//
// HBM bytes per call: x once here + once in pass 0, out once, tri written once and read (from L2: the rows of an item are
// placed on one XCD) once per row.  LDS: 64 KB (LN tile) + 16.5 x 4 KB (K, V^T, Q, G) + 8 KB (og) + 1 KB = 141 KB, one
// workgroup (8 waves) per CU.
#include "dfold_common.h"
#include "../../include/dfold_hip.h"
#include <math.h>

typedef __attribute__((ext_vector_type(4))) unsigned tfu32x4;
typedef __attribute__((ext_vector_type(2))) unsigned tfu32x2;
#define TF_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)

#define TF_NMAX 256
#define TF_VPITCH 528                      // V^T rows: 256 keys x 2 B + 16
#define TF_GPITCH 72                       // gate rows: 32 ch x 2 B + 8
#define TF_LDS_XN 0                        // 4 blocks of [64 rows][128 bf16], 16-byte chunks XORed with (row & 15)
#define TF_LDS_K (TF_LDS_XN + 65536)       // [256 keys][32 ch] 64-byte rows, chunk ^ ((-(row >> 2)) & 3)
#define TF_LDS_Q (TF_LDS_K + 16384)        // same layout, queries
#define TF_LDS_V (TF_LDS_Q + 16384)        // [32 ch][TF_VPITCH]
#define TF_LDS_G (TF_LDS_V + 32 * TF_VPITCH)      // [256 cells][TF_GPITCH]
#define TF_LDS_OG (TF_LDS_G + 256 * TF_GPITCH)    // 8 waves x [16 q][32 ch] 64-byte rows (swizzled like K)
#define TF_LDS_MB (TF_LDS_OG + 8 * 1024)          // 256 floats: mask bias of the row's keys
#define TF_LDS (TF_LDS_MB + 1024)
#define TF_OPITCH 528                      // out staging (aliases the LN tile): 16 q x (128 ch x 4 B + 16)

__device__ __forceinline__ int tf_a_tile_off(int row, int chunk) {
  return (row >> 6) * 16384 + (row & 63) * 256 + ((chunk ^ (row & 15)) << 4);
}
__device__ __forceinline__ int tf_k_off(int row, int chunk) { return row * 64 + ((chunk ^ ((-(row >> 2)) & 3)) << 4); }
__device__ __forceinline__ float tf_row16_sum(float v) {
  v += dpp_mov_f<0xb1>(0.f, v);
  v += dpp_mov_f<0x4e>(0.f, v);
  v += dpp_mov_f<0x124>(0.f, v);
  v += dpp_mov_f<0x128>(0.f, v);
  return v;
}
__device__ __forceinline__ float tf_xmax(float v) {
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float tf_xsum(float v) {
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}
__device__ __forceinline__ float tf_sigm(float y) { return __builtin_amdgcn_rcpf(1.f + __expf(-y)); }

struct TriAttFusedParams {
  const void* x;        // [B][N][N][128] fp32 | bf16
  const float* mask;    // [B][N][N] (coordinates of x)
  const float* gamma;   // LayerNorm
  const float* beta;
  const bf16_t* W;      // [512][128]: q | k | v | g rows
  const float* bcat;    // [512] biases of q | k | v | g (OpenFold: zeros for q, k, v; OmegaFold's attention has all four)
  const float* tri;     // [B][4][NP/16][NP/16][64][4] fp32, x log2(e): 16 x 16 blocks in accumulator order (pass 0)
  const bf16_t* Wo;     // [128][128]
  const float* bo;      // [128]
  void* out;            // [B][N][N][128] fp32 | bf16
  float* dbg;           // optional: row 0 of item 0 -> q|k|v|g of head 0 as fp32 [4][N][32] (tests)
  int B, N, NP, ending, x_bf16, out_bf16;
  float inf, scale, eps;
};

__global__ __launch_bounds__(512) void triatt_fused_kernel(const TriAttFusedParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const ldsXN = smem + TF_LDS_XN;
  char* const ldsK = smem + TF_LDS_K;
  char* const ldsQ = smem + TF_LDS_Q;
  char* const ldsV = smem + TF_LDS_V;
  char* const ldsG = smem + TF_LDS_G;
  float* const ldsMB = (float*)(smem + TF_LDS_MB);
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, l4 = lane >> 4;
  char* const ldsOG = smem + TF_LDS_OG + w * 1024;
  const int N = p.N, NP = p.NP;
  // XCD-aware work ids (blockIdx round-robins over the 8 XCDs): consecutive rows of one item share its triangle bias in
  // one L2
  const unsigned nwg = gridDim.x, bid = blockIdx.x;
  const unsigned xq = nwg >> 3, xr = nwg & 7, xcd = bid & 7, xidx = bid >> 3;
  const unsigned lid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + xidx;
  const int i = (int)(lid % (unsigned)N), b = (int)(lid / (unsigned)N);
  // cell (i, j) of the operator's coordinates x' (= x, or x^T for the ending node) in the memory of x / mask / out
  const long cs = p.ending ? (long)N : 1L;                                   // cell stride along j
  const long c0 = p.ending ? ((long)b * N) * N + i : ((long)b * N + i) * N;  // cell (i, 0)
  const float sl2 = p.scale * 1.44269504088896341f, inv_sl2 = 1.f / sl2;
  const int kswz = (-(l15 >> 2)) & 3;

  // ---- LayerNorm of the row's cells -> bf16 A tile; mask bias.  16 lanes per cell (8 channels each), 4 cells per pass ----
  {
    float gam[8], bet[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      gam[e] = p.gamma[l15 * 8 + e];
      bet[e] = p.beta[l15 * 8 + e];
    }
    const unsigned esz = p.x_bf16 ? 2u : 4u;
#pragma unroll 2
    for (int ps = 0; ps < 8; ++ps) {
      const int cell = w * 32 + ps * 4 + l4;
      float v[8];
      if (cell < N) {
        const char* src = (const char*)p.x + (c0 + cell * cs) * (128 * (long)esz) + (unsigned)l15 * (8u * esz);
        if (p.x_bf16) {
          const uint4 u = *(const uint4*)src;
          v[0] = bf_lo(u.x); v[1] = bf_hi(u.x); v[2] = bf_lo(u.y); v[3] = bf_hi(u.y);
          v[4] = bf_lo(u.z); v[5] = bf_hi(u.z); v[6] = bf_lo(u.w); v[7] = bf_hi(u.w);
        } else {
          const f32x4 a = *(const f32x4*)src, c = *(const f32x4*)(src + 16);
          v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
          v[4] = c[0]; v[5] = c[1]; v[6] = c[2]; v[7] = c[3];
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.f;
      }
      const float mean = tf_row16_sum(((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]))) * (1.f / 128.f);
      float q2 = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[e] -= mean;
        q2 = __builtin_fmaf(v[e], v[e], q2);
      }
      const float rstd = rsqrtf(tf_row16_sum(q2) * (1.f / 128.f) + p.eps);
      const bool live = cell < N;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = live ? __builtin_fmaf(v[e] * rstd, gam[e], bet[e]) : 0.f;   // pad cells: zero rows
      *(uint4*)(ldsXN + tf_a_tile_off(cell, l15)) =
          make_uint4(pack2bf_hw(v[0], v[1]), pack2bf_hw(v[2], v[3]), pack2bf_hw(v[4], v[5]), pack2bf_hw(v[6], v[7]));
    }
    if (tid < TF_NMAX) {
      float mb = -INFINITY;                 // keys past the end of the row
      if (tid < N) mb = p.inf * (p.mask[c0 + tid * cs] - 1.f) * inv_sl2;
      ldsMB[tid] = mb;
    }
  }
  __syncthreads();

  const int pj = w >> 1, nt = w & 1;        // this wave's slice of the head's projections: projection pj, channels nt*16 + l15
  f32x4 oacc_out[2][8];                     // linear_o accumulators of the wave's 2 x 16 queries: [q tile][16-channel block]
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) oacc_out[qt][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
  for (int h = 0; h < 4; ++h) {
    // ---- projections of head h: wave (pj, nt) -> 16 output channels for all cells ----
    {
      bf16x8 wf[4];
      const bf16_t* wrow = p.W + (long)(pj * 128 + h * 32 + nt * 16 + l15) * 128 + l4 * 8;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) wf[ks] = *(const bf16x8*)(wrow + ks * 32);
      const int ch = nt * 16 + l15;         // channel within the head
      const float bgv = p.bcat[pj * 128 + h * 32 + ch];
#pragma unroll 1
      for (int rt0 = 0; rt0 < TF_NMAX / 16; rt0 += 4) {
        // four row tiles at a time: four independent accumulator chains (a lone chain of 4 dependent MFMAs waits out the
        // matrix pipe's latency on every step).  All 16 row tiles whatever N is: the rows of cells >= N are zero (LN pass),
        // so the K / V^T / Q tiles are finite everywhere
        f32x4 acc4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) acc4[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int u = 0; u < 4; ++u)
            acc4[u] = TF_MFMA(*(const bf16x8*)(ldsXN + tf_a_tile_off((rt0 + u) * 16 + l15, ks * 4 + l4)), wf[ks], acc4[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const f32x4 acc = acc4[u];
          const int cell0 = (rt0 + u) * 16 + l4 * 4;         // accumulator: column = channel l15, rows = cells cell0 + r
          if (pj == 2) {
            *(uint2*)(ldsV + ch * TF_VPITCH + cell0 * 2) = make_uint2(pack2bf_hw(acc[0] + bgv, acc[1] + bgv), pack2bf_hw(acc[2] + bgv, acc[3] + bgv));
          } else if (pj == 3) {
#pragma unroll
            for (int r = 0; r < 4; ++r) *(bf16_t*)(ldsG + (cell0 + r) * TF_GPITCH + ch * 2) = f2bf_hw(tf_sigm(acc[r] + bgv));
          } else {
            char* const dst = pj == 0 ? ldsQ : ldsK;
#pragma unroll
            for (int r = 0; r < 4; ++r) *(bf16_t*)(dst + tf_k_off(cell0 + r, ch >> 3) + (ch & 7) * 2) = f2bf_hw(acc[r] + bgv);
          }
          if (p.dbg != nullptr && h == 0 && lid == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (cell0 + r < N) p.dbg[((long)pj * N + cell0 + r) * 32 + ch] = pj == 3 ? tf_sigm(acc[r] + bgv) : acc[r] + bgv;
          }
        }
      }
    }
    __syncthreads();

    // ---- attention of head h for the wave's 32 queries (two tiles of 16) ----
    bf16x8 wo[8];                           // W_o[:, h*32 .. +32] as B fragments [n = out channel][k = head channel]
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) wo[nb] = *(const bf16x8*)(p.Wo + (long)(nb * 16 + l15) * 128 + h * 32 + l4 * 8);
#pragma unroll 1
    for (int qt = 0; qt < 2; ++qt) {
      const int q = w * 32 + qt * 16 + l15;
      const bool qok = q < N;
      const int nt16 = NP >> 4;
      const int qtile = min(w * 2 + qt, nt16 - 1);          // (query tiles past the end of the row: any valid block, rows discarded)
      const float* tblk = p.tri + ((((long)b * 4 + h) * nt16 + qtile) * nt16) * 256 + lane * 4;
      // accumulator init = (triangle bias + mask bias) / (scale log2 e): logit * log2 e = acc * (scale log2 e)
      f32x4 s[16];
#pragma unroll
      for (int kb = 0; kb < 16; ++kb) {
        const int key0 = kb * 16 + l4 * 4;
        const f32x4 tb = *(const f32x4*)(tblk + (kb < nt16 ? kb : 0) * 256);     // one contiguous KB per wave instruction
        const f32x4 mb = *(const f32x4*)(ldsMB + key0);
#pragma unroll
        for (int r = 0; r < 4; ++r) s[kb][r] = __builtin_fmaf(tb[r], inv_sl2, mb[r]);
      }
      const bf16x8 qf = *(const bf16x8*)(ldsQ + q * 64 + ((l4 ^ kswz) << 4));
#pragma unroll
      for (int kb = 0; kb < 16; ++kb)
        s[kb] = TF_MFMA(*(const bf16x8*)(ldsK + (kb * 16 + l15) * 64 + ((l4 ^ kswz) << 4)), qf, s[kb]);
      float mx = -INFINITY;
#pragma unroll
      for (int kb = 0; kb < 16; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          s[kb][r] *= sl2;
          mx = fmaxf(mx, s[kb][r]);
        }
      mx = tf_xmax(mx);
      float sum = 0.f;
#pragma unroll
      for (int kb = 0; kb < 16; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          s[kb][r] = __builtin_amdgcn_exp2f(s[kb][r] - mx);
          sum += s[kb][r];
        }
      sum = tf_xsum(sum);
      // O^T[c][q] = V^T[c][keys] P^T[keys][q]; MFMA k-slot e of lane group l4 <-> key (2 ks + (e >> 2)) * 16 + l4 * 4 + (e & 3)
      f32x4 oacc[2];
      oacc[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
      oacc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const tfu32x4 pb = {pack2bf_hw(s[2 * ks][0], s[2 * ks][1]), pack2bf_hw(s[2 * ks][2], s[2 * ks][3]),
                            pack2bf_hw(s[2 * ks + 1][0], s[2 * ks + 1][1]), pack2bf_hw(s[2 * ks + 1][2], s[2 * ks + 1][3])};
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          const char* vp = ldsV + (cb * 16 + l15) * TF_VPITCH + ks * 64 + l4 * 8;
          const tfu32x2 lo = *(const tfu32x2*)vp, hi = *(const tfu32x2*)(vp + 32);
          const tfu32x4 av = {lo.x, lo.y, hi.x, hi.y};
          oacc[cb] = TF_MFMA(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, pb), oacc[cb]);
        }
      }
      // normalise, gate, stage og_h [16 q][32 ch] (wave-private), out += og_h W_o[:, h]^T
      const float inv = __builtin_amdgcn_rcpf(sum);
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        const tfu32x2 gg = *(const tfu32x2*)(ldsG + (qok ? q : 0) * TF_GPITCH + (cb * 16 + l4 * 4) * 2);
        const float o0 = oacc[cb][0] * inv * bf_lo(gg.x), o1 = oacc[cb][1] * inv * bf_hi(gg.x);
        const float o2 = oacc[cb][2] * inv * bf_lo(gg.y), o3 = oacc[cb][3] * inv * bf_hi(gg.y);
        // channel block cb*16 + l4*4 .. +4 -> 16-byte chunk cb*2 + (l4 >> 1), 8-byte half (l4 & 1)
        *(uint2*)(ldsOG + l15 * 64 + (((cb * 2 + (l4 >> 1)) ^ kswz) << 4) + ((l4 & 1) << 3)) =
            make_uint2(pack2bf_hw(o0, o1), pack2bf_hw(o2, o3));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      const bf16x8 ogf = *(const bf16x8*)(ldsOG + l15 * 64 + ((l4 ^ kswz) << 4));
      if (qt == 0) {          // (wave-uniform branch with static accumulator indices: a runtime index would put them in scratch)
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) oacc_out[0][nb] = TF_MFMA(ogf, wo[nb], oacc_out[0][nb]);
      } else {
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) oacc_out[1][nb] = TF_MFMA(ogf, wo[nb], oacc_out[1][nb]);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();        // every wave has left the K / V^T / Q / G tiles of this head
  }

  // ---- out rows: accumulator column = out channel l15 (+16 nb), rows = queries l4*4 + r; + b_o; staged through the (idle)
  //      LN tile region as fp32 rows so that the stores are whole 512-byte cells ----
  char* const st = smem + TF_LDS_XN + w * (16 * TF_OPITCH);
#pragma unroll 1
  for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) {
      const float bo = p.bo[nb * 16 + l15];
#pragma unroll
      for (int r = 0; r < 4; ++r) *(float*)(st + (l4 * 4 + r) * TF_OPITCH + (nb * 16 + l15) * 4) = (qt == 0 ? oacc_out[0][nb][r] : oacc_out[1][nb][r]) + bo;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    const int q0 = w * 32 + qt * 16;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int id = lane + 64 * j, row = id >> 5, c = id & 31;      // 16 rows x 32 chunks of 4 channels
      const int qq = q0 + row;
      const uint2 v0 = *(const uint2*)(st + row * TF_OPITCH + c * 16);
      const uint2 v1 = *(const uint2*)(st + row * TF_OPITCH + c * 16 + 8);
      if (qq < N) {
        const long cell = c0 + qq * cs;
        if (!p.out_bf16)
          *(uint4*)((float*)p.out + cell * 128 + c * 4) = make_uint4(v0.x, v0.y, v1.x, v1.y);
        else
          *(uint2*)((bf16_t*)p.out + cell * 128 + c * 4) =
              make_uint2(pack2bf_hw(__uint_as_float(v0.x), __uint_as_float(v0.y)), pack2bf_hw(__uint_as_float(v1.x), __uint_as_float(v1.y)));
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  }
}

// ------------------------------------------------------------------------------------------------------------------
// pass 0: triangle bias only.  tri[b][h][q][k] = log2(e) * w_tri[h] . LN(x'[q][k]) in the blocked layout read above -- a
// pure streaming pass (x in, 4 floats per cell out): 16 lanes per cell, four cells per pass, one wave per 64-key tile.
// For the query-block row kernel (csrc/triatt_rows.hip, any N_res) the same pass also writes the LayerNorm output itself
// as bf16 in the operator's coordinates (xn != nullptr: the ending node's transpose happens in this copy).  (bf16 bias
// blocks were tried with that kernel: half the per-row re-read of [4, N, N], 1 % of its time -- not kept.)
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tri_bias_kernel(const void* __restrict__ x, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const float* __restrict__ wtri,
                                                       float* __restrict__ tri, bf16_t* __restrict__ xn, int B, int N,
                                                       int NP, int ending, int x_bf16, float eps) {
  __shared__ __attribute__((aligned(16))) float stage[4][4][64];      // [wave][head][key]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  float gam[8], bet[8], wt[4][8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    gam[e] = gamma[l15 * 8 + e];
    bet[e] = beta[l15 * 8 + e];
#pragma unroll
    for (int h = 0; h < 4; ++h) wt[h][e] = wtri[h * 128 + l15 * 8 + e] * 1.44269504088896341f;
  }
  const int tpl = NP >> 6, nt16 = NP >> 4;
  const long ntiles = (long)B * N * tpl;
  const unsigned esz = x_bf16 ? 2u : 4u;
  for (long t = (long)blockIdx.x * 4 + w; t < ntiles; t += (long)gridDim.x * 4) {
    const int pt = (int)(t % tpl);
    const long bl = t / tpl;
    const int line = (int)(bl % N), b = (int)(bl / N);
#pragma unroll 4
    for (int ps = 0; ps < 16; ++ps) {
      const int pos = pt * 64 + ps * 4 + l4;
      const int pc = pos < N ? pos : N - 1;                 // (keys past the end: a valid cell; their bias meets a -inf mask)
      const long cell = ending ? ((long)b * N + pc) * N + line : ((long)b * N + line) * N + pc;
      const char* src = (const char*)x + cell * (128 * (long)esz) + (unsigned)l15 * (8u * esz);
      float v[8];
      if (x_bf16) {
        const uint4 u = *(const uint4*)src;
        v[0] = bf_lo(u.x); v[1] = bf_hi(u.x); v[2] = bf_lo(u.y); v[3] = bf_hi(u.y);
        v[4] = bf_lo(u.z); v[5] = bf_hi(u.z); v[6] = bf_lo(u.w); v[7] = bf_hi(u.w);
      } else {
        const f32x4 a = *(const f32x4*)src, c = *(const f32x4*)(src + 16);
        v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
        v[4] = c[0]; v[5] = c[1]; v[6] = c[2]; v[7] = c[3];
      }
      const float mean = tf_row16_sum(((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]))) * (1.f / 128.f);
      float q2 = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[e] -= mean;
        q2 = __builtin_fmaf(v[e], v[e], q2);
      }
      const float rstd = rsqrtf(tf_row16_sum(q2) * (1.f / 128.f) + eps);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = __builtin_fmaf(v[e] * rstd, gam[e], bet[e]);     // fp32, as the two-kernel form's bias
      if (xn != nullptr && pos < N)         // the row kernel's A operand: the same rounding as triatt_fused_kernel's LDS tile
        *(uint4*)(xn + ((((long)b * N + line) * N + pos) * 128 + l15 * 8)) =
            make_uint4(pack2bf_hw(v[0], v[1]), pack2bf_hw(v[2], v[3]), pack2bf_hw(v[4], v[5]), pack2bf_hw(v[6], v[7]));
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        float th = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) th = __builtin_fmaf(v[e], wt[h][e], th);
        th = tf_row16_sum(th);
        if (l15 == 0) stage[w][h][ps * 4 + l4] = th;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    {
      const int h = lane >> 4, v16 = lane & 15;            // keys pt*64 + 4 v16 .. +4 of query `line`, head h
      float* dst = tri + (((((long)b * 4 + h) * nt16 + (line >> 4)) * nt16 + pt * 4 + (v16 >> 2)) * 64 + (v16 & 3) * 16 + (line & 15)) * 4;
      *(f32x4*)dst = *(const f32x4*)&stage[w][h][v16 * 4];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  }
}

extern "C" int dfold_triatt_bias_blocked(const void* x, int32_t x_is_bf16, const float* ln_gamma, const float* ln_beta,
                                         const float* w_tri, float* tri, int32_t B, int32_t N, int32_t NP, int32_t ending, float eps,
                                         void* stream) {
  if (!x || !ln_gamma || !ln_beta || !w_tri || !tri || B <= 0 || N <= 0 || NP < N || (NP & 63)) return DFOLD_EINVAL;
  const long ntiles = (long)B * N * (NP >> 6);
  long grid = (ntiles + 3) / 4;
  if (grid > 2048) grid = 2048;
  DFOLD_LAUNCH(tri_bias_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, x, ln_gamma, ln_beta, w_tri, tri,
               (bf16_t*)nullptr, B, N, NP, ending ? 1 : 0, x_is_bf16 ? 1 : 0, eps);
  return dfold_check_launch();
}

extern "C" int dfold_triatt_ln_bias(const void* x, int32_t x_is_bf16, const float* ln_gamma, const float* ln_beta, const float* w_tri,
                                    float* tri, void* xn_bf16, int32_t B, int32_t N, int32_t NP, int32_t ending,
                                    float eps, void* stream) {
  if (!x || !ln_gamma || !ln_beta || !w_tri || !tri || !xn_bf16 || B <= 0 || N <= 0 || NP < N || (NP & 63)) return DFOLD_EINVAL;
  const long ntiles = (long)B * N * (NP >> 6);
  long grid = (ntiles + 3) / 4;
  if (grid > 2048) grid = 2048;
  DFOLD_LAUNCH(tri_bias_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, x, ln_gamma, ln_beta, w_tri, tri,
               (bf16_t*)xn_bf16, B, N, NP, ending ? 1 : 0, x_is_bf16 ? 1 : 0, eps);
  return dfold_check_launch();
}

extern "C" int dfold_triatt_fused_fwd(const void* x, int32_t x_is_bf16, const float* mask, const float* ln_gamma,
                                      const float* ln_beta, const void* w_cat_bf16, const float* bias_cat, const float* tri,
                                      const void* w_o_bf16, const float* b_o, void* out, int32_t out_is_bf16, float* dbg, int32_t B,
                                      int32_t N, int32_t NP, int32_t ending, float inf, float scale, float eps, void* stream) {
  if (!x || !mask || !ln_gamma || !ln_beta || !w_cat_bf16 || !bias_cat || !tri || !w_o_bf16 || !b_o || !out) return DFOLD_EINVAL;
  if (B <= 0 || N <= 0 || N > TF_NMAX || NP < N || (NP & 63) || (long)B * N > 0x7fffffffL) return DFOLD_EINVAL;
  TriAttFusedParams p;
  p.x = x; p.mask = mask; p.gamma = ln_gamma; p.beta = ln_beta; p.W = (const bf16_t*)w_cat_bf16; p.bcat = bias_cat; p.tri = tri;
  p.Wo = (const bf16_t*)w_o_bf16; p.bo = b_o; p.out = out; p.dbg = dbg; p.B = B; p.N = N; p.NP = NP; p.ending = ending ? 1 : 0;
  p.x_bf16 = x_is_bf16 ? 1 : 0; p.out_bf16 = out_is_bf16 ? 1 : 0; p.inf = inf; p.scale = scale; p.eps = eps;
  DFOLD_MAX_LDS_ONCE((triatt_fused_kernel), TF_LDS);
  DFOLD_LAUNCH(triatt_fused_kernel, dim3((unsigned)((long)B * N)), dim3(512), TF_LDS, (hipStream_t)stream, p);
  return dfold_check_launch();
}
