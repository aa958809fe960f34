// Triangle attention (openfold/model/triangular_attention.py:31-139, Attention openfold/model/primitives.py:219-243,
// 299-448) for c_in = 128, 4 heads x 32 and ANY N_res, projections kept on chip: the query-block form of the row kernel.
//
// csrc/triatt_fused.hip holds the LayerNorm tile of a whole pair-tensor row in LDS (N_res <= 256).  Here
//
//   pass 0 (dfold_triatt_ln_bias) is the one streaming pass over x: LayerNorm of every cell, written once as bf16 in the
//           operator's coordinates (xn; the ending node's transpose happens in this copy), and the triangle bias in 16 x 16
//           accumulator-order blocks;
//   this kernel, one workgroup per (item, row i, block of 256 queries), per head h and per chunk of 256 keys (the block's
//           own chunk first): the chunk's 256 xn rows arrive in LDS by LDS-DMA (global_load_lds, issued one phase ahead:
//           they land under the previous chunk's attention; N_res <= 256: loaded once, resident for the four heads) and are
//           projected on MFMA into K [256][32], V^T [32][256] (every chunk) and Q, sigmoid(G) (own chunk only: they stay
//           for the other chunks);
//           S^T = K Q^T (+ mask and triangle bias as accumulator init), ONLINE softmax over the key chunks (running max /
//           sum per query, the O^T accumulators rescaled when a chunk arrives; one chunk = the exact softmax of the
//           N_res <= 256 kernel), O^T += V^T P^T with the probabilities straight from the accumulators, og = O * g,
//           out += og_h W_o[:, h]^T accumulated over the heads in registers.
//
// q, k, v, g never exist in HBM (the two-kernel form of pair_fused.hip moves 4 x bf16 of the pair tensor out and back in).
// HBM bytes per call: x read once and xn written once (pass 0), xn read once (the two query blocks of a row at N_res 512
// meet in L2), out written once.  K / V of a row are projected once per query block (N_res 512: twice, + 15 % MFMA work).
// LDS: 64 KB (xn chunk) + 16 + 16 + 16.5 + 18 KB (K, Q, V^T, G) + 8 KB (og) + 2 KB (mask) = 140.5 KB, one workgroup per CU.
// (First version: 64-cell tiles through registers into two 16 KB buffers with a barrier per tile -- every tile waited out an
// L2 round trip for 16 MFMAs of work: 0.70 ms at batch 8 x N_res 256 against the whole-row kernel's 0.52.)
#include "dfold_common.h"
#include "../../include/dfold_hip.h"
#include <math.h>

typedef __attribute__((ext_vector_type(4))) unsigned tru32x4;
typedef __attribute__((ext_vector_type(2))) unsigned tru32x2;
typedef __attribute__((address_space(3))) char* tr_lds_ptr_t;
// LDS-DMA as inline assembly: with the builtin hipcc orders every LDS read it can see behind ALL outstanding DMA
// (s_waitcnt vmcnt(0) in front of the first ds_read after a global_load_lds) -- the chunk in flight would be waited for at
// the first instruction of the attention it is meant to hide under.  The waits for these transfers are explicit below.
// lds_addr: wave-uniform LDS byte address of the wave's 64 x 16 (x 4) bytes.
// Source = wave-uniform base + a 32-bit byte offset per lane (one VGPR instead of an address pair).
__device__ __forceinline__ void tr_dma16(const void* base, unsigned voff, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(lds_addr) : "memory", "m0");
}
__device__ __forceinline__ void tr_dma4(const void* base, unsigned voff, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(voff), "s"(base), "s"(lds_addr) : "memory", "m0");
}
#define TR_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)

#define TR_VPITCH 528                      // V^T rows: 256 keys x 2 B + 16
#define TR_GPITCH 72                       // gate rows: 32 ch x 2 B + 8
#define TR_LDS_A 0                         // [256 rows][128 bf16], 16-byte chunks XORed with (row & 15)
#define TR_LDS_K (TR_LDS_A + 65536)        // [256 keys][32 ch] 64-byte rows, chunk ^ ((-(row >> 2)) & 3)
#define TR_LDS_Q (TR_LDS_K + 16384)        // same layout, the block's queries
#define TR_LDS_V (TR_LDS_Q + 16384)        // [32 ch][TR_VPITCH]
#define TR_LDS_G (TR_LDS_V + 32 * TR_VPITCH)      // [256 queries][TR_GPITCH]
#define TR_LDS_OG (TR_LDS_G + 256 * TR_GPITCH)    // 8 waves x [16 q][32 ch] 64-byte rows (swizzled like K)
#define TR_LDS_MB (TR_LDS_OG + 8 * 1024)          // 256 floats: mask bias of the chunk's keys
#define TR_LDS_MR (TR_LDS_MB + 1024)              // 256 floats: the next chunk's mask values as they arrive (LDS-DMA)
#define TR_LDS (TR_LDS_MR + 1024)
#define TR_OPITCH 528                      // out staging (aliases the tiles): 16 q x (128 ch x 4 B + 16)

__device__ __forceinline__ int tr_a_off(int row, int chunk) { return row * 256 + ((chunk ^ (row & 15)) << 4); }
__device__ __forceinline__ int tr_k_off(int row, int chunk) { return row * 64 + ((chunk ^ ((-(row >> 2)) & 3)) << 4); }
__device__ __forceinline__ float tr_xmax(float v) {
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float tr_xsum(float v) {
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}
__device__ __forceinline__ float tr_sigm(float y) { return __builtin_amdgcn_rcpf(1.f + __expf(-y)); }

struct TriAttRowsParams {
  const bf16_t* xn;     // [B][N][N][128] bf16: LayerNorm output in the operator's coordinates (pass 0)
  const float* mask;    // [B][N][N] (coordinates of x)
  const bf16_t* W;      // [512][128]: q | k | v | g rows
  const float* bcat;    // [512] biases of q | k | v | g
  const float* tri;     // [B][4][NP/16][NP/16][64][4] fp32, x log2(e): 16 x 16 blocks in accumulator order (pass 0)
  const bf16_t* Wo;     // [128][128]
  const float* bo;      // [128]
  void* out;            // [B][N][N][128] fp32 | bf16 (coordinates of x)
  float* dbg;           // optional: row 0 of item 0 -> q|k|v|g of head 0 as fp32 [4][N][32] (tests)
  int B, N, NP, QB, ending, out_bf16;
  float inf, scale;
};

// projections of NU row tiles of 16 cells starting at row tile u0 of the chunk, for projection pj, channels nt*16 + l15 of
// head h (NU independent accumulator chains: a lone chain of 4 dependent MFMAs waits out the matrix pipe's latency on
// every step).  K / Q / V^T / G tiles are indexed by the cell's position inside its 256-cell chunk.
template <int NU>
__device__ __forceinline__ void tr_project_tile(const char* ldsA, char* ldsQ, char* ldsK, char* ldsV, char* ldsG, const bf16x8 (&wf)[4],
                                                float bgv, int pj, int ch, int u0, int l15, int l4, float* dbg, int N, int gcell0) {
  f32x4 acc[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u) acc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
#pragma unroll
    for (int u = 0; u < NU; ++u)
      acc[u] = TR_MFMA(*(const bf16x8*)(ldsA + tr_a_off((u0 + u) * 16 + l15, ks * 4 + l4)), wf[ks], acc[u]);
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const f32x4 a = acc[u];
    const int cell0 = (u0 + u) * 16 + l4 * 4;        // accumulator: column = channel l15, rows = cells cell0 + r
    if (pj == 2) {
      *(uint2*)(ldsV + ch * TR_VPITCH + cell0 * 2) = make_uint2(pack2bf_hw(a[0] + bgv, a[1] + bgv), pack2bf_hw(a[2] + bgv, a[3] + bgv));
    } else if (pj == 3) {
#pragma unroll
      for (int r = 0; r < 4; ++r) *(bf16_t*)(ldsG + (cell0 + r) * TR_GPITCH + ch * 2) = f2bf_hw(tr_sigm(a[r] + bgv));
    } else {
      char* const dst = pj == 0 ? ldsQ : ldsK;
#pragma unroll
      for (int r = 0; r < 4; ++r) *(bf16_t*)(dst + tr_k_off(cell0 + r, ch >> 3) + (ch & 7) * 2) = f2bf_hw(a[r] + bgv);
    }
    if (dbg != nullptr) {
      const int g0 = gcell0 + (u0 + u) * 16 + l4 * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (g0 + r < N) dbg[((long)pj * N + g0 + r) * 32 + ch] = pj == 3 ? tr_sigm(a[r] + bgv) : a[r] + bgv;
    }
  }
}

// one chunk of 256 keys for one tile of 16 queries: logits, online-softmax update, O^T += V^T P^T
__device__ __forceinline__ void tr_attend(const char* ldsK, const char* ldsV, const float* ldsMB, const bf16x8 qf, const float* tri,
                                          unsigned toff, int kc, int nt16, float sl2, float inv_sl2, int kswz, int l15,
                                          int l4, float& m, float& l, f32x4& oa0, f32x4& oa1) {
  // accumulator init = (triangle bias + mask bias) / (scale log2 e): logit * log2 e = acc * (scale log2 e).  All 16 bias
  // loads go out before the first one is consumed (two loops + a scheduling group: left to itself under this register
  // pressure hipcc issued load - wait - 4 FMAs sixteen times: 0.68 instead of 0.50 ms per call at batch 8 x N_res 256)
  f32x4 s[16];
#pragma unroll
  for (int kb = 0; kb < 16; ++kb) {
    const int kbg = kc * 16 + kb;
    const long boff = (long)(kbg < nt16 ? kbg : 0) * 256;          // (wave-uniform: folds into the scalar base; the lane part stays 32-bit)
    s[kb] = *(const f32x4*)(tri + boff + toff);                    // one contiguous KB per wave instruction
  }
  __builtin_amdgcn_sched_group_barrier(0x020, 16, 0);              // 16 VMEM reads first
#pragma unroll
  for (int kb = 0; kb < 16; ++kb) {
    const f32x4 mb = *(const f32x4*)(ldsMB + kb * 16 + l4 * 4);
#pragma unroll
    for (int r = 0; r < 4; ++r) s[kb][r] = __builtin_fmaf(s[kb][r], inv_sl2, mb[r]);
  }
#pragma unroll
  for (int kb = 0; kb < 16; ++kb)
    s[kb] = TR_MFMA(*(const bf16x8*)(ldsK + (kb * 16 + l15) * 64 + ((l4 ^ kswz) << 4)), qf, s[kb]);
  float mx = -INFINITY;                     // in accumulator units (scale log2 e > 0 is applied inside the exponent's FMA)
#pragma unroll
  for (int kb = 0; kb < 16; ++kb)
#pragma unroll
    for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kb][r]);
  mx = fmaxf(m, tr_xmax(mx) * sl2);         // every chunk holds at least one key < N: finite
  const float alpha = __builtin_amdgcn_exp2f(m - mx);      // first chunk: exp2(-inf) = 0
  m = mx;
  const float nmx = -mx;
  float sum = 0.f;
#pragma unroll
  for (int kb = 0; kb < 16; ++kb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      s[kb][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kb][r], sl2, nmx));
      sum += s[kb][r];
    }
  l = __builtin_fmaf(l, alpha, tr_xsum(sum));
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    oa0[r] *= alpha;
    oa1[r] *= alpha;
  }
  // O^T[c][q] += V^T[c][keys] P^T[keys][q]; MFMA k-slot e of lane group l4 <-> key (2 ks + (e >> 2)) * 16 + l4 * 4 + (e & 3)
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    const tru32x4 pb = {pack2bf_hw(s[2 * ks][0], s[2 * ks][1]), pack2bf_hw(s[2 * ks][2], s[2 * ks][3]),
                        pack2bf_hw(s[2 * ks + 1][0], s[2 * ks + 1][1]), pack2bf_hw(s[2 * ks + 1][2], s[2 * ks + 1][3])};
    {
      const char* vp = ldsV + l15 * TR_VPITCH + ks * 64 + l4 * 8;
      const tru32x2 lo = *(const tru32x2*)vp, hi = *(const tru32x2*)(vp + 32);
      const tru32x4 av = {lo.x, lo.y, hi.x, hi.y};
      oa0 = TR_MFMA(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, pb), oa0);
    }
    {
      const char* vp = ldsV + (16 + l15) * TR_VPITCH + ks * 64 + l4 * 8;
      const tru32x2 lo = *(const tru32x2*)vp, hi = *(const tru32x2*)(vp + 32);
      const tru32x4 av = {lo.x, lo.y, hi.x, hi.y};
      oa1 = TR_MFMA(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, pb), oa1);
    }
  }
}

// normalise, gate, stage og_h [16 q][32 ch] (wave-private), out += og_h W_o[:, h]^T
__device__ __forceinline__ void tr_finish_head(const char* ldsG, char* ldsOG, const bf16x8 (&wo)[8], int qrow, float l, const f32x4 oa0,
                                               const f32x4 oa1, int kswz, int l15, int l4, f32x4 (&oout)[8]) {
  const float inv = __builtin_amdgcn_rcpf(l);
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    const f32x4 oa = cb == 0 ? oa0 : oa1;
    const tru32x2 gg = *(const tru32x2*)(ldsG + qrow * TR_GPITCH + (cb * 16 + l4 * 4) * 2);
    const float o0 = oa[0] * inv * bf_lo(gg.x), o1 = oa[1] * inv * bf_hi(gg.x);
    const float o2 = oa[2] * inv * bf_lo(gg.y), o3 = oa[3] * inv * bf_hi(gg.y);
    // channel block cb*16 + l4*4 .. +4 -> 16-byte chunk cb*2 + (l4 >> 1), 8-byte half (l4 & 1)
    *(uint2*)(ldsOG + l15 * 64 + (((cb * 2 + (l4 >> 1)) ^ kswz) << 4) + ((l4 & 1) << 3)) = make_uint2(pack2bf_hw(o0, o1), pack2bf_hw(o2, o3));
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  const bf16x8 ogf = *(const bf16x8*)(ldsOG + l15 * 64 + ((l4 ^ kswz) << 4));
#pragma unroll
  for (int nb = 0; nb < 8; ++nb) oout[nb] = TR_MFMA(ogf, wo[nb], oout[nb]);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

__global__ __launch_bounds__(512) void triatt_rows_kernel(const TriAttRowsParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const ldsA = smem + TR_LDS_A;
  char* const ldsK = smem + TR_LDS_K;
  char* const ldsQ = smem + TR_LDS_Q;
  char* const ldsV = smem + TR_LDS_V;
  char* const ldsG = smem + TR_LDS_G;
  float* const ldsMB = (float*)(smem + TR_LDS_MB);
  float* const ldsMR = (float*)(smem + TR_LDS_MR);
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, l4 = lane >> 4;
  char* const ldsOG = smem + TR_LDS_OG + w * 1024;
  const int N = p.N, NP = p.NP, QB = p.QB, nt16 = NP >> 4;
  // XCD-aware work ids (blockIdx round-robins over the 8 XCDs): the query blocks of a row and consecutive rows of an item
  // share that row's xn / the item's triangle bias in one L2
  const unsigned nwg = gridDim.x, bid = blockIdx.x;
  const unsigned xq = nwg >> 3, xr = nwg & 7, xcd = bid & 7, xidx = bid >> 3;
  const unsigned lid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + xidx;
  const int qb = (int)(lid % (unsigned)QB);
  const unsigned rowid = lid / (unsigned)QB;
  const int i = (int)(rowid % (unsigned)N), b = (int)(rowid / (unsigned)N);
  // cell (i, j) of the operator's coordinates x' (= x, or x^T for the ending node) in the memory of mask / out
  const long cs = p.ending ? (long)N : 1L;                                   // cell stride along j
  const long c0 = p.ending ? ((long)b * N) * N + i : ((long)b * N + i) * N;  // cell (i, 0)
  const bf16_t* const xrow = p.xn + ((long)b * N + i) * N * 128;             // xn is stored in the operator's coordinates
  const float sl2 = p.scale * 1.44269504088896341f, inv_sl2 = 1.f / sl2;
  const int kswz = (-(l15 >> 2)) & 3;
  float* const dbg = (p.dbg != nullptr && lid == 0) ? p.dbg : nullptr;

  // xn chunk staging by LDS-DMA: wave w, instruction j writes the 1 KB of chunk rows (j*8 + w)*4 .. +4 (lane -> row + (lane >> 4),
  // 16-byte slot lane & 15); the XOR swizzle sits on the SOURCE side (slot s of row r receives chunk s ^ (r & 15), and
  // r & 15 = (w & 3)*4 + (lane >> 4) for every j).  Cells past the end of the row: the last cell's values (finite; their
  // keys carry a -inf mask bias, their query rows are discarded)
  const int srow = w * 4 + (lane >> 4);
  const int schunk = (lane & 15) ^ (((w & 3) << 2) + (lane >> 4));
  const unsigned ldsA_addr = (unsigned)(unsigned long)(tr_lds_ptr_t)ldsA, ldsMR_addr = (unsigned)(unsigned long)(tr_lds_ptr_t)(smem + TR_LDS_MR);
  const float* const mrow = p.mask + c0;
  auto issue_chunk = [&](int cb) __attribute__((always_inline)) {
    int sr = srow;
    asm volatile("" : "+v"(sr));            // (opaque: the eight offsets are recomputed where they are used -- hoisted out of
                                            //  the attention loop they cost 16 registers there and the projections spilled)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int cell = min(cb + j * 32 + sr, N - 1);
      tr_dma16(xrow, (unsigned)(cell * 256 + schunk * 16), ldsA_addr + (j * 8 + w) * 1024);
    }
    // the chunk's mask values ride along (waves 0-3: key cb + tid, four bytes per lane); turned into the mask bias at the
    // start of the chunk's phase by the wave that fetched them -- no register lives through the attention for them
    if (w < 4) {
      int tt = tid;
      asm volatile("" : "+v"(tt));
      const int key = min(cb + tt, N - 1);
      tr_dma4(mrow, (unsigned)(key * (int)cs) * 4u, ldsMR_addr + w * 256);
    }
  };
  issue_chunk(qb * 256);                    // the first phase's chunk (the block's own)

  f32x4 oacc_out[2][8];                     // linear_o accumulators of the wave's 2 x 16 queries: [q tile][16-channel block]
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) oacc_out[qt][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int qt0 = min(qb * 16 + w * 2, nt16 - 1), qt1 = min(qb * 16 + w * 2 + 1, nt16 - 1);   // (tiles past the end: any valid block, rows discarded)

#pragma unroll 1
  for (int h = 0; h < 4; ++h) {
    float mA = -INFINITY, mB = -INFINITY, lA = 0.f, lB = 0.f;           // online-softmax state of the wave's two query tiles
    f32x4 oaA0 = {0.f, 0.f, 0.f, 0.f}, oaA1 = oaA0, oaB0 = oaA0, oaB1 = oaA0;
    // bias blocks of (item, head): uniform base + 32-bit lane offsets (elements)
    const float* const tri_h = p.tri + (((long)b * 4 + h) * nt16 * nt16) * 256;
    const unsigned tb0 = (unsigned)(qt0 * nt16) * 256u + lane * 4, tb1 = (unsigned)(qt1 * nt16) * 256u + lane * 4;
#pragma unroll 1
    for (int it = 0; it < QB; ++it) {
      int kc = qb + it;
      if (kc >= QB) kc -= QB;
      const bool own = it == 0;
      const int cbase = kc * 256;
      // ---- projections of head h for the chunk's cells: own chunk q | k | v | g (wave -> projection w >> 1, 16 channels
      //      w & 1, all 16 row tiles), other chunks k | v (wave -> projection 1 + (w >> 2), 16 channels (w >> 1) & 1, 8 of the
      //      16 row tiles) ----
      {
        const int pj = own ? (w >> 1) : 1 + (w >> 2);
        const int nt = own ? (w & 1) : ((w >> 1) & 1);
        const int u0 = own ? 0 : (w & 1) * 8;          // first of the wave's row tiles (other chunks: 8 of the 16)
        bf16x8 wf[4];
        const bf16_t* wrow = p.W + (long)(pj * 128 + h * 32 + nt * 16 + l15) * 128 + l4 * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) wf[ks] = *(const bf16x8*)(wrow + ks * 32);
        const int ch = nt * 16 + l15;         // channel within the head
        const float bgv = p.bcat[pj * 128 + h * 32 + ch];
        // the chunk's rows are in flight since before the previous phase's attention (N_res <= 256: resident since head 0)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // mask bias of the chunk's keys from the values this wave's DMA delivered (a load -> wait -> write chain here was one
        // more exposed memory round trip per phase in front of the barrier)
        if (w < 4) {
          const float mraw = ldsMR[tid];
          ldsMB[tid] = (cbase + tid < N) ? p.inf * (mraw - 1.f) * inv_sl2 : -INFINITY;       // keys past the end of the row
        }
        __syncthreads();
        float* const dtap = (h == 0) ? dbg : nullptr;
        if (own) {
#pragma unroll 1
          for (int rt0 = 0; rt0 < 16; rt0 += 4)
            tr_project_tile<4>(ldsA, ldsQ, ldsK, ldsV, ldsG, wf, bgv, pj, ch, rt0, l15, l4, dtap, N, cbase);
        } else {
#pragma unroll 1
          for (int rt0 = u0; rt0 < u0 + 8; rt0 += 4)
            tr_project_tile<4>(ldsA, ldsQ, ldsK, ldsV, ldsG, wf, bgv, pj, ch, rt0, l15, l4, dtap, N, cbase);
        }
        __syncthreads();
      }

      // ---- attention of head h over this key chunk for the wave's 32 queries (two tiles of 16).  A rolled loop: the 64
      //      logit registers of a tile exist once; the tiles' softmax states trade places at the end of every pass (static
      //      register indices: a runtime index would put them in scratch) ----
#pragma unroll 1
      for (int qt = 0; qt < 2; ++qt) {
        // next phase's chunk (one chunk per row: it stays).  Issued HERE, behind whatever wait the compiler places at the
        // head of this loop and in front of the first tile's bias loads: the transfers and those loads are in flight together
        // (memory returns in order, so the first bias wait also covers the chunk -- one round trip, not two)
        if (qt == 0 && QB > 1 && !((h == 3) && (it == QB - 1))) {
          int kn = kc + 1;
          if (kn >= QB) kn -= QB;                         // (after the last chunk of a head: the own chunk of the next head)
          issue_chunk(kn * 256);
        }
        const int qrow = w * 32 + qt * 16 + l15;           // the lane's query inside the block
        const bf16x8 qf = *(const bf16x8*)(ldsQ + qrow * 64 + ((l4 ^ kswz) << 4));
        tr_attend(ldsK, ldsV, ldsMB, qf, tri_h, qt == 0 ? tb0 : tb1, kc, nt16, sl2, inv_sl2, kswz, l15, l4, mA, lA, oaA0, oaA1);
        if (it == QB - 1) {
          bf16x8 wo[8];                       // W_o[:, h*32 .. +32] as B fragments [n = out channel][k = head channel]
#pragma unroll
          for (int nb = 0; nb < 8; ++nb) wo[nb] = *(const bf16x8*)(p.Wo + (long)(nb * 16 + l15) * 128 + h * 32 + l4 * 8);
          if (qt == 0)
            tr_finish_head(ldsG, ldsOG, wo, qrow, lA, oaA0, oaA1, kswz, l15, l4, oacc_out[0]);
          else
            tr_finish_head(ldsG, ldsOG, wo, qrow, lA, oaA0, oaA1, kswz, l15, l4, oacc_out[1]);
        }
        { const float t = mA; mA = mB; mB = t; }
        { const float t = lA; lA = lB; lB = t; }
        { const f32x4 t = oaA0; oaA0 = oaB0; oaB0 = t; }
        { const f32x4 t = oaA1; oaA1 = oaB1; oaB1 = t; }
      }
      __syncthreads();        // every wave has left the K / V^T (/ Q / G) tiles of this chunk
    }
  }

  // ---- out rows: accumulator column = out channel l15 (+16 nb), rows = queries l4*4 + r; + b_o; staged through the (idle)
  //      tile region as fp32 rows so that the stores are whole 512-byte cells ----
  char* const st = smem + w * (16 * TR_OPITCH);
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) {
      const float bo = p.bo[nb * 16 + l15];
#pragma unroll
      for (int r = 0; r < 4; ++r) *(float*)(st + (l4 * 4 + r) * TR_OPITCH + (nb * 16 + l15) * 4) = oacc_out[qt][nb][r] + bo;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    const int q0 = qb * 256 + w * 32 + qt * 16;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int id = lane + 64 * j, row = id >> 5, c = id & 31;      // 16 rows x 32 chunks of 4 channels
      const int qq = q0 + row;
      const uint2 v0 = *(const uint2*)(st + row * TR_OPITCH + c * 16);
      const uint2 v1 = *(const uint2*)(st + row * TR_OPITCH + c * 16 + 8);
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

extern "C" int dfold_triatt_rows_fwd(const void* xn_bf16, const float* mask, const void* w_cat_bf16, const float* bias_cat,
                                     const float* tri, const void* w_o_bf16, const float* b_o, void* out,
                                     int32_t out_is_bf16, float* dbg, int32_t B, int32_t N, int32_t NP, int32_t ending, float inf,
                                     float scale, void* stream) {
  if (!xn_bf16 || !mask || !w_cat_bf16 || !bias_cat || !tri || !w_o_bf16 || !b_o || !out) return DFOLD_EINVAL;
  if (B <= 0 || N <= 0 || NP < N || (NP & 63)) return DFOLD_EINVAL;
  const int QB = (N + 255) / 256;
  if ((long)B * N * QB > 0x7fffffffL) return DFOLD_EINVAL;
  TriAttRowsParams p;
  p.xn = (const bf16_t*)xn_bf16; p.mask = mask; p.W = (const bf16_t*)w_cat_bf16; p.bcat = bias_cat; p.tri = tri;
  p.Wo = (const bf16_t*)w_o_bf16; p.bo = b_o; p.out = out; p.dbg = dbg; p.B = B; p.N = N; p.NP = NP; p.QB = QB;
  p.ending = ending ? 1 : 0; p.out_bf16 = out_is_bf16 ? 1 : 0; p.inf = inf; p.scale = scale;
  DFOLD_MAX_LDS_ONCE((triatt_rows_kernel), TR_LDS);
  DFOLD_LAUNCH(triatt_rows_kernel, dim3((unsigned)((long)B * N * QB)), dim3(512), TR_LDS, (hipStream_t)stream, p);
  return dfold_check_launch();
}
