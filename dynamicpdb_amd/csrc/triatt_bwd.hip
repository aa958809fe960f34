// Streaming backward of the triangle attention core (openfold/model/triangular_attention.py:78-139, Attention
// openfold/model/primitives.py:219-243, 377-448; AF2 Alg. 13 / 14) for 4 heads x 32 channels, N_res <= 512.
//
// The round-1..3 backward re-ran the intermediate-keeping chain: fp32 logits, probabilities, dP and bf16 dS as
// [B N, H, N, N] tensors in HBM (2.1 GB per item at N_res 512) plus their transposes.  Here the logits are recomputed per
// pair-tensor row on the matrix cores and never leave the chip; what crosses HBM is pair-sized only (the recomputed
// q | k | v | g projections, their gradients, `do`, three floats of softmax statistics per (row, head, query)) plus the
// [H, N, N] triangle-bias gradient.
//
//   forward of row i, head h:  S = a q k^T + mask bias + tri_h,  P = softmax_k S,  o = P v,  og = o * sigmoid(g),
//                              out = og W_o^T + b_o
//   backward:                  dog = dout W_o,  do = dog * sigmoid(g),  dg = dog * o * sigmoid'(g),
//                              dP = do v^T,  D = rowsum(P * dP) = <do, o>,  dS = P * (dP - D),
//                              dq = a dS k,  dk = a dS^T q,  dv = P^T do,  dtri_h = sum_i dS
//
// Two launches, both with the MFMA-accumulator-as-B-operand trick of the forward kernels (a 16 x 16 tile of P or dS in
// accumulator layout IS the B operand of the next product once the reduction index of the A operand is permuted to match):
//
//   triatt_bwd_q_kernel  (b, i, h): a wave owns 16 queries and ALL keys of the row: S^T = K Q^T, exact softmax in
//       registers, O^T = V^T P^T, dog^T = W_o^T dout^T, dP^T = V do^T, dS^T, dQ^T = K^T dS^T.  Writes og, dg, do, dq and
//       the row statistics (max, 1 / sum, D).  K, V of the row sit in LDS in both orientations.
//   triatt_bwd_k_kernel  (chunk of rows, b, h, 128 keys): a wave owns 16 keys and loops over ALL queries of a row and over
//       the rows of its chunk: S = Q K^T and P from the stored statistics, dP = do V^T, dS, dK^T += Q^T dS,
//       dV^T += do^T P; dS is summed over the chunk's rows IN REGISTERS (the triangle-bias gradient: no atomics, no
//       [I, H, N, N] tensor; the few chunk partials are added by dfold_sum_leading).  Q, do of the row sit in LDS in both
//       orientations.
#include "dfold_common.h"
#include "../../include/dfold_hip.h"
#include <math.h>

typedef __attribute__((ext_vector_type(4))) unsigned tbu32x4;
typedef __attribute__((ext_vector_type(2))) unsigned tbu32x2;
#define TB_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#define TB_L2E 1.44269504088896341f

struct TriAttBwdParams {
  const bf16_t* proj;   // [B N N][512] bf16: q | k | v | g (pre-activation) of every cell, head h at columns h*32 of each block
  const bf16_t* projT;  // [512][B N N] the same, channel-major (the projection GEMM run with swapped operands): K^T, V^T, Q^T tiles
  const float* tri;     // [B][4][N][N] triangle bias (unscaled)
  const float* mask;    // [B][N][N]
  const bf16_t* dob;    // [B N N][128] dL/dout
  const bf16_t* WoT;    // [128 (h*32 + c)][128 (out channel)] = W_o^T
  bf16_t* dproj;        // [B N N][512] dq | dk | dv | dg
  bf16_t* og;           // [B N N][128] o * sigmoid(g) (operand of dW_o)
  bf16_t* dos;          // [B N N][128] scratch: do
  bf16_t* dosT;         // [128][B N N] scratch: do, channel-major
  long R;               // B N N
  float* stats;         // [B N][4][3][N]: row max (log2 domain), 1 / sum, D
  float* dtri_part;     // [IC][B][4][N][N]
  int B, N, IC, rpc, KB;
  float inf, scale;
};

__device__ __forceinline__ int tb_k_off(int row, int chunk) { return row * 64 + ((chunk ^ ((-(row >> 2)) & 3)) << 4); }
__device__ __forceinline__ float tb_xmax(float v) {
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float tb_xsum(float v) {
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}
__device__ __forceinline__ float tb_sigm(float y) { return 1.f / (1.f + __expf(-y)); }

// [rows][32] bf16 head slice (row stride `ld` elements) -> LDS, row-major 64-byte rows (16-byte chunks XOR-swizzled like the
// forward kernels' K tile); rows >= nvalid are zero.
template <int NMAX>
__device__ __forceinline__ void tb_load_rows(const bf16_t* __restrict__ src, long ld, int nvalid, char* ldsR, int tid) {
#pragma unroll
  for (int it = 0; it < NMAX * 4 / 256; ++it) {
    const int id = it * 256 + tid, row = id >> 2, c = id & 3;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (row < nvalid) v = *(const uint4*)(src + (long)row * ld + c * 8);
    *(uint4*)(ldsR + tb_k_off(row, c)) = v;
  }
}
// the same head slice from the channel-major copy (32 rows of `nvalid` consecutive cells, row stride ldT) -> LDS
// [32][NMAX * 2 + 16 bytes]; cells >= nvalid are zero (nvalid % 8 == 0)
template <int NMAX>
__device__ __forceinline__ void tb_load_planes(const bf16_t* __restrict__ srcT, long ldT, int nvalid, char* ldsT, int tid) {
  constexpr int TP = NMAX * 2 + 16, CPR = NMAX / 8;
#pragma unroll
  for (int it = 0; it < NMAX * 4 / 256; ++it) {
    const int id = it * 256 + tid, c = id / CPR, cell0 = (id - c * CPR) * 8;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (cell0 < nvalid) v = *(const uint4*)(srcT + (long)c * ldT + cell0);
    *(uint4*)(ldsT + c * TP + cell0 * 2) = v;
  }
}

// The four LDS tiles of a row -- two [rows][32] head slices (row-major) and the same two channel-major -- staged TOGETHER:
// every global load is issued before the first LDS write, unconditionally (rows / cells past the end re-read the last valid
// one and are zeroed by a select).  As four calls of tb_load_rows / tb_load_planes with `if (row < nvalid) v = load` each
// iteration was a branch with load - s_waitcnt vmcnt(0) - LDS write inside: 16 (N_res <= 256: 8 ...) memory round trips in
// a row per pair-tensor row in kernel K, whose arithmetic per row is 32 MFMAs per wave (hipcc -S, scripts/isa_audit.py).
template <int NMAX, int NT>
__device__ __forceinline__ void tb_stage_row(const bf16_t* __restrict__ r0, long ld0, const bf16_t* __restrict__ r1, long ld1,
                                             const bf16_t* __restrict__ t0, const bf16_t* __restrict__ t1, long ldT, int nvalid,
                                             char* lds0, char* lds1, char* ldsT0, char* ldsT1, int tid) {
  constexpr int IT = NMAX * 4 / NT, TP = NMAX * 2 + 16, CPR = NMAX / 8;
  tbu32x4 a[IT], b[IT], c[IT], d[IT];
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int id = it * NT + tid, row = id >> 2, cc = id & 3;
    const int rc = row < nvalid ? row : nvalid - 1;
    a[it] = *(const tbu32x4*)(r0 + (long)rc * ld0 + cc * 8);
    b[it] = *(const tbu32x4*)(r1 + (long)rc * ld1 + cc * 8);
    const int ch = id / CPR, cell0 = (id - ch * CPR) * 8;
    const int cl = cell0 < nvalid ? cell0 : nvalid - 8;                  // nvalid % 8 == 0
    c[it] = *(const tbu32x4*)(t0 + (long)ch * ldT + cl);
    d[it] = *(const tbu32x4*)(t1 + (long)ch * ldT + cl);
  }
  const tbu32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int id = it * NT + tid, row = id >> 2, cc = id & 3;
    const bool rin = row < nvalid;
    *(tbu32x4*)(lds0 + tb_k_off(row, cc)) = rin ? a[it] : z;
    *(tbu32x4*)(lds1 + tb_k_off(row, cc)) = rin ? b[it] : z;
    const int ch = id / CPR, cell0 = (id - ch * CPR) * 8;
    const bool cin = cell0 < nvalid;
    *(tbu32x4*)(ldsT0 + ch * TP + cell0 * 2) = cin ? c[it] : z;
    *(tbu32x4*)(ldsT1 + ch * TP + cell0 * 2) = cin ? d[it] : z;
  }
}

// A operand whose reduction index is permuted to the accumulator-derived B operand: 4 + 4 elements 32 bytes apart
__device__ __forceinline__ bf16x8 tb_frag_perm(const char* p) {
  const tbu32x2 lo = *(const tbu32x2*)p, hi = *(const tbu32x2*)(p + 32);
  const tbu32x4 v = {lo.x, lo.y, hi.x, hi.y};
  return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ bf16x8 tb_pack8(const f32x4& a, const f32x4& b) {
  const tbu32x4 v = {pack2bf_hw(a[0], a[1]), pack2bf_hw(a[2], a[3]), pack2bf_hw(b[0], b[1]), pack2bf_hw(b[2], b[3])};
  return __builtin_bit_cast(bf16x8, v);
}

template <int NMAX>
__global__ __launch_bounds__(256, NMAX == 256 ? 2 : 1) void triatt_bwd_q_kernel(const TriAttBwdParams p) {
  constexpr int NKT = NMAX / 16, ROWT = NMAX * 64, TP = NMAX * 2 + 16, TRT = 32 * TP;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const ldsK = smem;
  char* const ldsV = smem + ROWT;
  char* const ldsKT = smem + 2 * ROWT;
  char* const ldsVT = smem + 2 * ROWT + TRT;
  float* const ldsMB = (float*)(smem + 2 * ROWT + 2 * TRT);
  char* const ldsWO = smem + 2 * ROWT + 2 * TRT + 3 * NMAX * 4;     // W_o^T rows of the head: [32][128 bf16], chunks XORed with (row & 15)
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, l4 = lane >> 4;
  const int N = p.N;
  const int h = blockIdx.x & 3;
  const long bi = blockIdx.x >> 2;                    // b * N + i
  const int b = (int)(bi / N);
  const long row0 = bi * N;                           // first cell of the row
  {
    // the head's 32 rows of W_o^T (8 KB) once per workgroup: fetched per query tile from global memory every one of the 8
    // fragments was a load - wait - MFMA step
    tbu32x4 wo2[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int id = tid + 256 * j, row = id >> 4, c = id & 15;
      wo2[j] = *(const tbu32x4*)(p.WoT + (long)(h * 32 + row) * 128 + c * 8);
    }
    float mraw[NMAX / 256];
#pragma unroll
    for (int u = 0; u < NMAX / 256; ++u) mraw[u] = p.mask[row0 + min(tid + 256 * u, N - 1)];      // (in flight with the tiles)
    tb_stage_row<NMAX, 256>(p.proj + row0 * 512 + 128 + h * 32, 512, p.proj + row0 * 512 + 256 + h * 32, 512,
                            p.projT + (long)(128 + h * 32) * p.R + row0, p.projT + (long)(256 + h * 32) * p.R + row0, p.R, N, ldsK, ldsV,
                            ldsKT, ldsVT, tid);
#pragma unroll
    for (int u = 0; u < NMAX / 256; ++u) ldsMB[tid + 256 * u] = tid + 256 * u < N ? p.inf * (mraw[u] - 1.f) * TB_L2E : -INFINITY;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int id = tid + 256 * j, row = id >> 4, c = id & 15;
      *(tbu32x4*)(ldsWO + row * 256 + ((c ^ (row & 15)) << 4)) = wo2[j];
    }
  }
  __syncthreads();

  const float sl2 = p.scale * TB_L2E;
  const int kswz = (-(l15 >> 2)) & 3;
  const int nqt = (N + 15) >> 4;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int qt = w; qt < nqt; qt += 4) {
    asm volatile("" ::: "memory");      // the K / V tiles are loop-invariant: without this every fragment of the row is hoisted out
                                        // of the loop and parked in scratch
    const int q = qt * 16 + l15;
    const bool qok = q < N;
    const long qrow = row0 + (qok ? q : N - 1);
    const bf16_t* prow = p.proj + qrow * 512 + h * 32;
    // everything this query tile reads from global memory is requested here, together: q, the four dout fragments, the gate
    // pre-activations and the NKT triangle-bias vectors (unconditional; keys past the end re-read the last four and meet a
    // -inf mask bias) -- as `if (key0 < N) tb = load` behind the MFMAs each bias vector was its own load - wait step
    const bf16x8 qf = *(const bf16x8*)(prow + l4 * 8);
    bf16x8 dof[4];
    tbu32x2 ggv[2];
    f32x4 s[NKT];
    float mx = -INFINITY;
    if constexpr (NMAX == 256) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) dof[ks] = *(const bf16x8*)(p.dob + qrow * 128 + ks * 32 + l4 * 8);
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) ggv[cb] = *(const tbu32x2*)(prow + 384 + cb * 16 + l4 * 4);
    // ---- S^T = K Q^T with (triangle bias log2 e + mask bias) / (scale log2 e) as accumulator init; exact softmax over the
    //      row's keys (accumulator: rows = keys kb*16 + l4*4 + r, column = query l15)
    const float* trow = p.tri + (((long)b * 4 + h) * N + (qok ? q : N - 1)) * N;
#pragma unroll
    for (int kb = 0; kb < NKT; ++kb) {
      const int key0 = kb * 16 + l4 * 4;
      s[kb] = *(const f32x4*)(trow + (key0 < N ? key0 : N - 4));            // N % 4 == 0
    }
    __builtin_amdgcn_sched_group_barrier(0x020, NKT + 7, 0);
    const float inv_sl2 = 1.f / sl2;
#pragma unroll
    for (int kb = 0; kb < NKT; ++kb) {
      const f32x4 mb = *(const f32x4*)(ldsMB + kb * 16 + l4 * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) s[kb][r] = __builtin_fmaf(s[kb][r], TB_L2E, mb[r]) * inv_sl2;
    }
#pragma unroll
    for (int kb = 0; kb < NKT; ++kb) s[kb] = TB_MFMA(*(const bf16x8*)(ldsK + (kb * 16 + l15) * 64 + ((l4 ^ kswz) << 4)), qf, s[kb]);
#pragma unroll
    for (int kb = 0; kb < NKT; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s[kb][r] *= sl2;
        mx = fmaxf(mx, s[kb][r]);
      }
    } else {
      // (N_res <= 512 instance, one wave per SIMD with half its values in AGPRs: the batched form measured 3-6 % slower there --
      //  hipcc serialises the loads it cannot place -- so it keeps the products first and the bias loads behind them)
    // ---- S^T = K Q^T; logits * log2 e; exact softmax over the row's keys (accumulator: rows = keys kb*16 + l4*4 + r, column = query l15)
#pragma unroll
    for (int kb = 0; kb < NKT; ++kb) s[kb] = TB_MFMA(*(const bf16x8*)(ldsK + (kb * 16 + l15) * 64 + ((l4 ^ kswz) << 4)), qf, zero4);
    const float* trow = p.tri + (((long)b * 4 + h) * N + (qok ? q : N - 1)) * N;
#pragma unroll
    for (int kb = 0; kb < NKT; ++kb) {
      const int key0 = kb * 16 + l4 * 4;
      f32x4 tb = zero4;
      if (key0 < N) tb = *(const f32x4*)(trow + key0);           // N % 4 == 0
      const f32x4 mb = *(const f32x4*)(ldsMB + key0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s[kb][r] = __builtin_fmaf(s[kb][r], sl2, __builtin_fmaf(tb[r], TB_L2E, mb[r]));
        mx = fmaxf(mx, s[kb][r]);
      }
    }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) dof[ks] = *(const bf16x8*)(p.dob + qrow * 128 + ks * 32 + l4 * 8);
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) ggv[cb] = *(const tbu32x2*)(prow + 384 + cb * 16 + l4 * 4);
    }
    mx = tb_xmax(mx);
    float sum = 0.f;
#pragma unroll
    for (int kb = 0; kb < NKT; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s[kb][r] = __builtin_amdgcn_exp2f(s[kb][r] - mx);
        sum += s[kb][r];
      }
    sum = tb_xsum(sum);
    const float inv = 1.f / sum;
    // ---- O^T = V^T P^T (k-slot e of lane group l4 <-> key (2 ks + (e >> 2)) * 16 + l4 * 4 + (e & 3))
    f32x4 oacc[2] = {zero4, zero4};
#pragma unroll
    for (int ks = 0; ks < NKT / 2; ++ks) {
      const bf16x8 pb = tb_pack8(s[2 * ks], s[2 * ks + 1]);
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) oacc[cb] = TB_MFMA(tb_frag_perm(ldsVT + (cb * 16 + l15) * TP + ks * 64 + l4 * 8), pb, oacc[cb]);
    }
    // ---- dog^T = W_o^T dout^T for the head's 32 channels (rows = channel cb*16 + l4*4 + r, column = query)
    f32x4 dg[2] = {zero4, zero4};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
        dg[cb] = TB_MFMA(*(const bf16x8*)(ldsWO + (cb * 16 + l15) * 256 + (((ks * 4 + l4) ^ l15) << 4)), dof[ks], dg[cb]);
    }
    // ---- gate; do = dog * sigma, dg = dog * o * sigma (1 - sigma), og = o * sigma, D = <do, o>
    f32x4 dov[2];
    float Dp = 0.f;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      const tbu32x2 gg = ggv[cb];
      const float gp[4] = {bf_lo(gg.x), bf_hi(gg.x), bf_lo(gg.y), bf_hi(gg.y)};
      float ogv[4], dgv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float sg = tb_sigm(gp[r]);
        const float o = oacc[cb][r] * inv;
        const float dog = dg[cb][r];
        dov[cb][r] = dog * sg;
        Dp = __builtin_fmaf(dov[cb][r], o, Dp);
        dgv[r] = dog * o * sg * (1.f - sg);
        ogv[r] = o * sg;
      }
      if (qok) {
        const long cell = row0 + q;
        const int col = h * 32 + cb * 16 + l4 * 4;
        *(uint2*)(p.og + cell * 128 + col) = make_uint2(pack2bf_hw(ogv[0], ogv[1]), pack2bf_hw(ogv[2], ogv[3]));
        *(uint2*)(p.dos + cell * 128 + col) = make_uint2(pack2bf_hw(dov[cb][0], dov[cb][1]), pack2bf_hw(dov[cb][2], dov[cb][3]));
        *(uint2*)(p.dproj + cell * 512 + 384 + col) = make_uint2(pack2bf_hw(dgv[0], dgv[1]), pack2bf_hw(dgv[2], dgv[3]));
#pragma unroll
        for (int r = 0; r < 4; ++r) p.dosT[(long)(col + r) * p.R + cell] = f2bf_hw(dov[cb][r]);      // channel-major copy (kernel K's do^T tile)
      }
    }
    const float D = tb_xsum(Dp);
    if (l4 == 0 && qok) {
      float* st = p.stats + ((bi * 4 + h) * 3) * N;
      st[q] = mx;
      st[N + q] = inv;
      st[2 * N + q] = D;
    }
    // ---- dP^T = V do^T (reduction over the head's channels, permuted to the accumulator order of do), dS^T, dQ^T = K^T dS^T
    const bf16x8 dofrag = tb_pack8(dov[0], dov[1]);
    f32x4 dq[2] = {zero4, zero4};
#pragma unroll
    for (int ks = 0; ks < NKT / 2; ++ks) {
      f32x4 ds[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int kb = 2 * ks + u;
        const char* vr = ldsV + (kb * 16 + l15) * 64 + (l4 & 1) * 8;
        const tbu32x2 lo = *(const tbu32x2*)(vr + (((l4 >> 1) ^ kswz) << 4)), hi = *(const tbu32x2*)(vr + (((2 + (l4 >> 1)) ^ kswz) << 4));
        const tbu32x4 av = {lo.x, lo.y, hi.x, hi.y};
        const f32x4 dp = TB_MFMA(__builtin_bit_cast(bf16x8, av), dofrag, zero4);
#pragma unroll
        for (int r = 0; r < 4; ++r) ds[u][r] = s[kb][r] * inv * (dp[r] - D);
      }
      const bf16x8 dsb = tb_pack8(ds[0], ds[1]);
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) dq[cb] = TB_MFMA(tb_frag_perm(ldsKT + (cb * 16 + l15) * TP + ks * 64 + l4 * 8), dsb, dq[cb]);
    }
    if (qok) {
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
        *(uint2*)(p.dproj + (row0 + q) * 512 + h * 32 + cb * 16 + l4 * 4) =
            make_uint2(pack2bf_hw(dq[cb][0] * p.scale, dq[cb][1] * p.scale), pack2bf_hw(dq[cb][2] * p.scale, dq[cb][3] * p.scale));
    }
  }
}

// Kernel K: 8 waves = KT key tiles x QS query ranges (QS = NMAX / 128: every wave owns 16 keys and 128 queries, so that its
// slice of the triangle bias and of the bias gradient -- 2 x 32 registers -- stays in registers for the whole chunk of rows;
// the dk / dv partial sums of the QS waves of a key tile meet in LDS once per row).
template <int NMAX, int NT>
__device__ __forceinline__ void tb_load_rows_nt(const bf16_t* __restrict__ src, long ld, int nvalid, char* ldsR, int tid) {
#pragma unroll
  for (int it = 0; it < NMAX * 4 / NT; ++it) {
    const int id = it * NT + tid, row = id >> 2, c = id & 3;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (row < nvalid) v = *(const uint4*)(src + (long)row * ld + c * 8);
    *(uint4*)(ldsR + tb_k_off(row, c)) = v;
  }
}
template <int NMAX, int NT>
__device__ __forceinline__ void tb_load_planes_nt(const bf16_t* __restrict__ srcT, long ldT, int nvalid, char* ldsT, int tid) {
  constexpr int TP = NMAX * 2 + 16, CPR = NMAX / 8;
#pragma unroll
  for (int it = 0; it < NMAX * 4 / NT; ++it) {
    const int id = it * NT + tid, c = id / CPR, cell0 = (id - c * CPR) * 8;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (cell0 < nvalid) v = *(const uint4*)(srcT + (long)c * ldT + cell0);
    *(uint4*)(ldsT + c * TP + cell0 * 2) = v;
  }
}

template <int NMAX>
__global__ __launch_bounds__(512) void triatt_bwd_k_kernel(const TriAttBwdParams p) {
  constexpr int QS = NMAX / 128, KT = 8 / QS, NQW = 8;            // query tiles per wave (128 queries)
  constexpr int ROWT = NMAX * 64, TP = NMAX * 2 + 16, TRT = 32 * TP;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const ldsQ = smem;
  char* const ldsDO = smem + ROWT;
  char* const ldsQT = smem + 2 * ROWT;
  char* const ldsDOT = smem + 2 * ROWT + TRT;
  float* const ldsST = (float*)(smem + 2 * ROWT + 2 * TRT);      // [3][NMAX]: max, 1 / sum, D of the row's queries
  float* const ldsRED = ldsST + 3 * NMAX;                        // [KT][QS - 1][16][64]: dk / dv partial sums
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, l4 = lane >> 4;
  const int N = p.N, B = p.B;
  unsigned bid = blockIdx.x;
  const int kblk = (int)(bid % (unsigned)p.KB);
  bid /= (unsigned)p.KB;
  const int h = (int)(bid & 3u);
  bid >>= 2;
  const int b = (int)(bid % (unsigned)B), ic = (int)(bid / (unsigned)B);
  const int i0 = ic * p.rpc, i1 = min(N, i0 + p.rpc);
  const int ktl = w % KT, qs = w / KT;                            // key tile of the block, query range
  const int kt = kblk * KT + ktl;
  const int key = kt * 16 + l15;
  const int qt0 = qs * NQW;                                       // first query tile of the wave
  const bool kok = key < N, wave_on = kt * 16 < N && qt0 * 16 < N;
  const int kc = kok ? key : N - 1;
  const float sl2 = p.scale * TB_L2E;
  const int kswz = (-(l15 >> 2)) & 3;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  // the wave's slice of the triangle bias (x log2 e) -- the same for every row -- and of its gradient, in registers for the
  // whole chunk of rows: [query tile][r] <-> tri[h][(qt0 + t)*16 + l4*4 + r][key]
  f32x4 trv[NQW], dt[NQW];
  {
    const float* tbase = p.tri + ((long)b * 4 + h) * N * N + kc;
#pragma unroll
    for (int t = 0; t < NQW; ++t) {
      dt[t] = zero4;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qq = (qt0 + t) * 16 + l4 * 4 + r;
        trv[t][r] = tbase[(long)min(qq, N - 1) * N] * TB_L2E;
      }
    }
  }
  const char* const aq = ldsQ + (qt0 * 16 + l15) * 64 + ((l4 ^ kswz) << 4);       // A fragments: + t * 1024
  const char* const ad = aq + ROWT;
  const float* const stq = ldsST + qt0 * 16 + l4 * 4;                                // statistics: + t * 16 floats
  const char* const tq = ldsQT + l15 * TP + qt0 * 32 + l4 * 8;                       // permuted A fragments: + cb * 16 TP + tp * 64
  const char* const td = tq + TRT;

#pragma unroll 1
  for (int i = i0; i < i1; ++i) {
    const long bi = (long)b * N + i;
    const long row0 = bi * N;
    __syncthreads();            // the previous row's readers are done with the tiles and the reduction buffer
    {
      // statistics of the row's queries (3 x NMAX floats), requested together with the tiles
      const float* st = p.stats + ((bi * 4 + h) * 3) * N;
      constexpr int SI = (3 * NMAX + 511) / 512;
      float sv[SI];
#pragma unroll
      for (int u = 0; u < SI; ++u) {
        const int t = min(tid + 512 * u, 3 * NMAX - 1), which = t / NMAX, qq = t - which * NMAX;
        sv[u] = st[which * N + min(qq, N - 1)];
      }
      tb_stage_row<NMAX, 512>(p.proj + row0 * 512 + h * 32, 512, p.dos + row0 * 128 + h * 32, 128, p.projT + (long)(h * 32) * p.R + row0,
                              p.dosT + (long)(h * 32) * p.R + row0, p.R, N, ldsQ, ldsDO, ldsQT, ldsDOT, tid);
#pragma unroll
      for (int u = 0; u < SI; ++u) {
        const int t = tid + 512 * u, which = t / NMAX, qq = t - which * NMAX;
        if (t < 3 * NMAX) ldsST[t] = qq < N ? sv[u] : 0.f;
      }
    }
    __syncthreads();
    f32x4 dk[2] = {zero4, zero4}, dv[2] = {zero4, zero4};
    if (wave_on) {
      const bf16x8 kf = *(const bf16x8*)(p.proj + (row0 + kc) * 512 + 128 + h * 32 + l4 * 8);
      const bf16x8 vf = *(const bf16x8*)(p.proj + (row0 + kc) * 512 + 256 + h * 32 + l4 * 8);
      const float mbk = kok ? p.inf * (p.mask[row0 + key] - 1.f) * TB_L2E : -INFINITY;
#pragma unroll
      for (int tp = 0; tp < NQW / 2; ++tp) {
        f32x4 pr[2], ds[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int t = 2 * tp + u;
          const f32x4 sa = TB_MFMA(*(const bf16x8*)(aq + t * 1024), kf, zero4);       // rows = queries (qt0+t)*16 + l4*4 + r, column = key
          const f32x4 dp = TB_MFMA(*(const bf16x8*)(ad + t * 1024), vf, zero4);
          const f32x4 m4 = *(const f32x4*)(stq + t * 16), il4 = *(const f32x4*)(stq + NMAX + t * 16), D4 = *(const f32x4*)(stq + 2 * NMAX + t * 16);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            // (queries >= N: Q / do rows, the statistics and so 1 / sum are zero; the clamp keeps an overflowing exp2 from
            // meeting that zero -- for real queries the argument is <= 0 by construction)
            const float e = __builtin_amdgcn_exp2f(fminf(__builtin_fmaf(sa[r], sl2, trv[t][r] + mbk) - m4[r], 64.f));
            const float pv = e * il4[r];
            const float dsv = pv * (dp[r] - D4[r]);
            pr[u][r] = pv;
            ds[u][r] = dsv;
            dt[t][r] += dsv;
          }
        }
        const bf16x8 pb = tb_pack8(pr[0], pr[1]), dsb = tb_pack8(ds[0], ds[1]);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          dk[cb] = TB_MFMA(tb_frag_perm(tq + cb * 16 * TP + tp * 64), dsb, dk[cb]);
          dv[cb] = TB_MFMA(tb_frag_perm(td + cb * 16 * TP + tp * 64), pb, dv[cb]);
        }
      }
    }
    // dk / dv of the key tile: the QS - 1 upper query ranges hand their partial sums to the wave of range 0
    if (qs > 0) {
      float* red = ldsRED + ((ktl * (QS - 1) + qs - 1) * 16) * 64 + lane;
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          red[(cb * 8 + r) * 64] = dk[cb][r];
          red[(cb * 8 + 4 + r) * 64] = dv[cb][r];
        }
    }
    __syncthreads();
    if (qs == 0 && kt * 16 < N) {
#pragma unroll
      for (int o = 0; o < QS - 1; ++o) {
        const float* red = ldsRED + ((ktl * (QS - 1) + o) * 16) * 64 + lane;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            dk[cb][r] += red[(cb * 8 + r) * 64];
            dv[cb][r] += red[(cb * 8 + 4 + r) * 64];
          }
      }
      if (kok) {
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          bf16_t* dst = p.dproj + (row0 + key) * 512 + h * 32 + cb * 16 + l4 * 4;
          *(uint2*)(dst + 128) = make_uint2(pack2bf_hw(dk[cb][0] * p.scale, dk[cb][1] * p.scale), pack2bf_hw(dk[cb][2] * p.scale, dk[cb][3] * p.scale));
          *(uint2*)(dst + 256) = make_uint2(pack2bf_hw(dv[cb][0], dv[cb][1]), pack2bf_hw(dv[cb][2], dv[cb][3]));
        }
      }
    }
  }
  // triangle-bias gradient of this chunk of rows
  if (wave_on && kok) {
    float* dst = p.dtri_part + ((((long)ic * B + b) * 4 + h) * N) * N + key;
#pragma unroll
    for (int t = 0; t < NQW; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qq = (qt0 + t) * 16 + l4 * 4 + r;
        if (qq < N) dst[(long)qq * N] = dt[t][r];
      }
  }
}

extern "C" int dfold_triatt_bwd_core(const void* proj_bf16, const void* proj_t_bf16, const float* tri, const float* mask,
                                     const void* dout_bf16, const void* w_o_t_bf16, void* dproj_bf16, void* og_bf16,
                                     void* do_scratch_bf16, void* do_t_scratch_bf16, float* stats, float* dtri_part, int32_t B,
                                     int32_t N, int32_t n_chunks, float inf, float scale, void* stream) {
  if (!proj_bf16 || !proj_t_bf16 || !tri || !mask || !dout_bf16 || !w_o_t_bf16 || !dproj_bf16 || !og_bf16 || !do_scratch_bf16 ||
      !do_t_scratch_bf16 || !stats || !dtri_part)
    return DFOLD_EINVAL;
  if (B <= 0 || N <= 0 || N > 512 || (N & 7) || n_chunks <= 0 || n_chunks > N || (long)B * N * 4 > 0x3fffffffL) return DFOLD_EINVAL;
  TriAttBwdParams p;
  p.proj = (const bf16_t*)proj_bf16; p.projT = (const bf16_t*)proj_t_bf16; p.tri = tri; p.mask = mask; p.dob = (const bf16_t*)dout_bf16;
  p.WoT = (const bf16_t*)w_o_t_bf16; p.dproj = (bf16_t*)dproj_bf16; p.og = (bf16_t*)og_bf16; p.dos = (bf16_t*)do_scratch_bf16;
  p.dosT = (bf16_t*)do_t_scratch_bf16; p.stats = stats; p.dtri_part = dtri_part; p.R = (long)B * N * N;
  p.B = B; p.N = N; p.IC = n_chunks; p.rpc = (N + n_chunks - 1) / n_chunks; p.KB = N <= 256 ? (N + 63) / 64 : (N + 31) / 32;      /* key tiles per block of kernel K: 8 waves / (NMAX / 128) */ p.inf = inf; p.scale = scale;
  const unsigned gq = (unsigned)((long)B * N * 4), gk = (unsigned)((long)n_chunks * B * 4 * p.KB);
  if (N <= 256) {
    constexpr int LDS = 2 * 256 * 64 + 2 * 32 * (256 * 2 + 16) + 3 * 256 * 4, LDSK = LDS + 4 * 1 * 16 * 64 * 4;
    DFOLD_MAX_LDS_ONCE((triatt_bwd_q_kernel<256>), LDS + 8192);
    DFOLD_MAX_LDS_ONCE((triatt_bwd_k_kernel<256>), LDSK);
    DFOLD_LAUNCH(triatt_bwd_q_kernel<256>, dim3(gq), dim3(256), LDS + 8192, (hipStream_t)stream, p);
    DFOLD_LAUNCH(triatt_bwd_k_kernel<256>, dim3(gk), dim3(512), LDSK, (hipStream_t)stream, p);
  } else {
    constexpr int LDS = 2 * 512 * 64 + 2 * 32 * (512 * 2 + 16) + 3 * 512 * 4, LDSK = LDS + 2 * 3 * 16 * 64 * 4;
    DFOLD_MAX_LDS_ONCE((triatt_bwd_q_kernel<512>), LDS + 8192);
    DFOLD_MAX_LDS_ONCE((triatt_bwd_k_kernel<512>), LDSK);
    DFOLD_LAUNCH(triatt_bwd_q_kernel<512>, dim3(gq), dim3(256), LDS + 8192, (hipStream_t)stream, p);
    DFOLD_LAUNCH(triatt_bwd_k_kernel<512>, dim3(gk), dim3(512), LDSK, (hipStream_t)stream, p);
  }
  return dfold_check_launch();
}
