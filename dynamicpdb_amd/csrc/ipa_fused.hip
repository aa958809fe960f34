// Fused forward of the Invariant Point Attention core for gfx950 (reference: InvariantPointAttention.forward,
// src/model/ipa_pytorch_dynamic.py:396-469): logits, softmax, o = P v and o_pt = P v_pts of one (window, frame, head,
// block of 128 queries) in ONE launch -- no [B,F,H,N,N] fp32 logits in HBM, the probabilities leave the chip once (bf16,
// for the o_pair product and the backward).
//
// Everything that is quadratic in N_res runs on the matrix cores as TWO augmented products:
//
//   logit[i,j] = alpha q_i.k_j + bias_scale bias[i,j] - hw/2 |qp_i - kp_j|^2 + inf (m_i m_j - 1)           (:402-443)
//              = alpha (Q'_i . K'_j) + bias_scale bias[i,j] + kn_j + m_i inf (m_j - 1)   + (terms constant along a row)
//       Q'_i = [ q_i (256) | (hw/alpha) (qp_i - ctr) as bf16 pieces ],   K'_j = [ k_j (256) | (kp_j - ctr) as bf16 pieces ],
//       kn_j = -hw/2 |kp_j - ctr|^2
//     -hw/2 |qp - kp|^2 = hw qp.kp - hw/2 |kp|^2 - hw/2 |qp|^2: the last term is constant along a softmax row and drops
//     out.  The 24-coordinate product qp.kp needs fp32-grade accuracy (coordinates of tens of Angstrom, differences of a
//     few): every coordinate is split into three bf16 pieces x = xh + xm + xl (24 mantissa bits) and the six products
//     hh, hm, mh, hl, lh, mm (everything above 2^-24 |q||k|) are laid out as 6 x 24 = 144 (+16 zero) extra K columns;
//     products of bf16 pieces are exact in fp32, the MFMA accumulates in fp32.  `ctr` = a per-(window, frame) centre
//     subtracted from all points (the logits only depend on differences): it halves the magnitudes that cancel.
//
//   [o | o_pt - ctr] = P [v | (vp - ctr)]:  O^T = V'^T P^T with V'^T = [v^T (256 rows) | vp pieces (3 x 48 rows)] and the
//     probabilities as TWO bf16 pieces P = Ph + Pl for the point rows (Ph (Vh+Vm+Vl) + Pl (Vh+Vm): error 2^-17 |vp|, i.e.
//     ~1e-4 A; with one bf16 piece the 2^-9 relative error of P becomes ~0.1 A on points that are mapped back into the local
//     frame afterwards -- measured in round 2) and Ph alone for the 256 scalar channels (as the unfused chain).
//
// S^T = K' Q'^T is computed (A rows = keys, B columns = the wave's 16 queries) so that the probabilities sit in the
// accumulators exactly in the B-operand layout of O^T = V'^T P^T: no cross-lane shuffle, no LDS round trip.  A wave owns
// 16 queries and ALL keys of its row (N_res <= 512: <= 128 accumulator registers), so the softmax is exact (no online
// rescaling) and needs two 2-step cross-lane reductions per row.  Key / value tiles are streamed through LDS in 64-key
// chunks shared by the 8 waves of a workgroup, register-staged one chunk ahead, two buffers.
#include "dfold_common.h"
#include "../../include/dfold_hip.h"
#include <math.h>

typedef __attribute__((ext_vector_type(4))) unsigned ifu32x4;
typedef __attribute__((ext_vector_type(2))) unsigned ifu32x2;
#define IF_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)

#define IF_C 256        // scalar channels per head
#define IF_PK 160       // point columns of Q' / K': 6 products x 24 coordinates + 16 zeros
#define IF_VROWS 400    // rows of V'^T: 256 value channels + 3 pieces x 48 (36 point components + 12 zeros)
#define IF_KC 64        // keys per LDS chunk
#define IF_VPITCH 144   // bytes per V'^T row in LDS: 64 keys x 2 B + 16
#define IF_BUF (IF_VROWS * IF_VPITCH)      // 57600 B per buffer (>= the 53248 B of a K' chunk)

__device__ __forceinline__ int if_a_tile_off(int row, int chunk) { return row * 256 + ((chunk ^ (row & 15)) << 4); }

__device__ __forceinline__ void if_split3(float x, bf16_t& h, bf16_t& m, bf16_t& l) {
  h = f2bf_hw(x);
  float r = x - bf2f(h);
  m = f2bf_hw(r);
  r -= bf2f(m);
  l = f2bf_hw(r);
}

// ------------------------------------------------------------------------------------------------------------------
// operand preparation (elementwise, one thread per (window*frame, residue, head))
// ------------------------------------------------------------------------------------------------------------------
// q_pts, k_pts fp32 [B*F, N, H, 8, 3] -> QP, KP bf16 [B*F, H, N, 160], kn fp32 [B*F, H, N]
__global__ __launch_bounds__(256) void ipa_aug_prep_qk_kernel(const float* __restrict__ q_pts, const float* __restrict__ k_pts,
                                                              const float* __restrict__ hw, const float* __restrict__ ctr,
                                                              bf16_t* __restrict__ QP, bf16_t* __restrict__ KP,
                                                              float* __restrict__ kn, long total, int N, int H, float inv_alpha) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int h = (int)(idx % H);
  const long rn = idx / H;
  const int n = (int)(rn % N);
  const long bf = rn / N;
  const float* q = q_pts + idx * 24;
  const float* k = k_pts + idx * 24;
  const float c0 = ctr[bf * 3], c1 = ctr[bf * 3 + 1], c2 = ctr[bf * 3 + 2];
  const float hwh = hw[h], s = hwh * inv_alpha;
  // outputs as packed pairs (one v_cvt_pk_bf16_f32 per pair and piece; coordinate pairs never straddle a product block)
  __attribute__((aligned(16))) uint32_t qo[IF_PK / 2], ko[IF_PK / 2];
  float k2 = 0.f;
#pragma unroll
  for (int c = 0; c < 24; c += 2) {
    const float ca = (c % 3 == 0) ? c0 : ((c % 3 == 1) ? c1 : c2);
    const float cb = ((c + 1) % 3 == 0) ? c0 : (((c + 1) % 3 == 1) ? c1 : c2);
    const float qa = (q[c] - ca) * s, qb = (q[c + 1] - cb) * s, ka = k[c] - ca, kb = k[c + 1] - cb;
    k2 = __builtin_fmaf(ka, ka, k2);
    k2 = __builtin_fmaf(kb, kb, k2);
    const uint32_t qh = pack2bf_hw(qa, qb), kh = pack2bf_hw(ka, kb);
    const float qra = qa - bf_lo(qh), qrb = qb - bf_hi(qh), kra = ka - bf_lo(kh), krb = kb - bf_hi(kh);
    const uint32_t qm = pack2bf_hw(qra, qrb), km = pack2bf_hw(kra, krb);
    const uint32_t ql = pack2bf_hw(qra - bf_lo(qm), qrb - bf_hi(qm)), kl = pack2bf_hw(kra - bf_lo(km), krb - bf_hi(km));
    const int i2 = c >> 1;
    // products hh, hm, mh, hl, lh, mm
    qo[i2] = qh;       ko[i2] = kh;
    qo[12 + i2] = qh;  ko[12 + i2] = km;
    qo[24 + i2] = qm;  ko[24 + i2] = kh;
    qo[36 + i2] = qh;  ko[36 + i2] = kl;
    qo[48 + i2] = ql;  ko[48 + i2] = kh;
    qo[60 + i2] = qm;  ko[60 + i2] = km;
  }
#pragma unroll
  for (int c = 72; c < IF_PK / 2; ++c) {
    qo[c] = 0u;
    ko[c] = 0u;
  }
  const long row = (bf * H + h) * N + n;
  uint4* qd = (uint4*)(QP + row * IF_PK);
  uint4* kd = (uint4*)(KP + row * IF_PK);
#pragma unroll
  for (int v = 0; v < IF_PK / 8; ++v) {
    qd[v] = make_uint4(qo[4 * v], qo[4 * v + 1], qo[4 * v + 2], qo[4 * v + 3]);
    kd[v] = make_uint4(ko[4 * v], ko[4 * v + 1], ko[4 * v + 2], ko[4 * v + 3]);
  }
  kn[row] = -0.5f * hwh * k2;
}

// v_pts fp32 [B*F, N, H, 12, 3] -> rows 256 .. 399 of VT bf16 [B*F, H, 400, NP] (three pieces x 48 rows, key-contiguous)
__global__ __launch_bounds__(256) void ipa_aug_prep_v_kernel(const float* __restrict__ v_pts, const float* __restrict__ ctr,
                                                             bf16_t* __restrict__ VT, long total, int N, int H, int NP) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;      // ((bf * H + h) * N + j): lanes walk the keys
  if (idx >= total) return;
  const int j = (int)(idx % N);
  const long bh = idx / N;
  const int h = (int)(bh % H);
  const long bf = bh / H;
  const float* v = v_pts + ((bf * N + j) * H + h) * 36;
  const float c0 = ctr[bf * 3], c1 = ctr[bf * 3 + 1], c2 = ctr[bf * 3 + 2];
  bf16_t* base = VT + (bh * IF_VROWS + IF_C) * (long)NP + j;
#pragma unroll
  for (int c = 0; c < 36; ++c) {
    const float cc = (c % 3 == 0) ? c0 : ((c % 3 == 1) ? c1 : c2);
    bf16_t vh, vm, vl;
    if_split3(v[c] - cc, vh, vm, vl);
    base[(long)c * NP] = vh;
    base[(long)(48 + c) * NP] = vm;
    base[(long)(96 + c) * NP] = vl;
  }
}

extern "C" int dfold_ipa_aug_prep(const float* q_pts, const float* k_pts, const float* v_pts, const float* hw, const float* ctr,
                                  void* QP_bf16, void* KP_bf16, float* kn, void* VT_bf16, int32_t B, int32_t F, int32_t N,
                                  int32_t H, int32_t NP, float alpha, void* stream) {
  if (!q_pts || !k_pts || !v_pts || !hw || !ctr || !QP_bf16 || !KP_bf16 || !kn || !VT_bf16) return DFOLD_EINVAL;
  if (B <= 0 || F <= 0 || N <= 0 || H <= 0 || NP < N || (NP % IF_KC) || !(alpha > 0.f)) return DFOLD_EINVAL;
  const long total = (long)B * F * N * H;
  const unsigned grid = (unsigned)((total + 255) / 256);
  hipStream_t st = (hipStream_t)stream;
  DFOLD_LAUNCH(ipa_aug_prep_qk_kernel, dim3(grid), dim3(256), 0, st, q_pts, k_pts, hw, ctr, (bf16_t*)QP_bf16, (bf16_t*)KP_bf16, kn,
               total, N, H, 1.f / alpha);
  if (dfold_check_launch() != DFOLD_OK) return DFOLD_ELAUNCH;
  DFOLD_LAUNCH(ipa_aug_prep_v_kernel, dim3(grid), dim3(256), 0, st, v_pts, ctr, (bf16_t*)VT_bf16, total, N, H, NP);
  return dfold_check_launch();
}

// ------------------------------------------------------------------------------------------------------------------
// the attention kernel
// ------------------------------------------------------------------------------------------------------------------
struct IpaFusedParams {
  const bf16_t* q;      // [B*F, N, H*256]
  const bf16_t* kv;     // [B*F, N, H*512]  (k | v per head)
  const bf16_t* QP;     // [B*F, H, N, 160]
  const bf16_t* KP;     // [B*F, H, N, 160]
  const bf16_t* VT;     // [B*F, H, 400, NP]
  const float* kn;      // [B*F, H, N]
  const float* bias;    // [B, H, N, N]
  const float* mask;    // [B*F, N]
  const float* ctr;     // [B*F, 3]
  bf16_t* o;            // [B*F, N, o_ld]: head h at columns h*256 (o_ld >= H*256: the row of a wider feature matrix)
  long o_ld;
  float* o_pt;          // [B*F, N, H, 36]
  bf16_t* Pb;           // [B*F, H, N, N]
  float* P;             // optional fp32 copy of the probabilities (null: not written)
  int BF, F, N, H, NP;
  float alpha, bias_scale, inf;
};

__device__ __forceinline__ float if_xmax(float v) {
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float if_xsum(float v) {
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}

// A wave's [16 queries x (16 MT) columns] tile, held as MT accumulator tiles (lane: query l15, columns m*16 + l4*4 + r), leaves
// as bf16 through a wave-private LDS staging area so that every global store instruction writes whole rows (64 lanes x
// 16 bytes = two 512-byte rows at 256 columns): the direct form -- 8 bytes per lane, 16 rows x 32 bytes per instruction --
// wrote partial cache lines only.  rows_ok: number of valid query rows of this wave; ncols: valid columns (multiple of 8).
template <int MT>
__device__ __forceinline__ void if_store_rows_bf16(char* stage, const f32x4 (&t)[MT], bf16_t* dst, long row_stride, int rows_ok,
                                                   int ncols, int lane) {
  constexpr int PITCH = MT * 32 + 16;                  // bytes per staged row: 16 MT columns x 2 B + 16 (bank spread)
  const int l15 = lane & 15, l4 = lane >> 4;
#pragma unroll
  for (int m = 0; m < MT; ++m)
    *(uint2*)(stage + l15 * PITCH + m * 32 + l4 * 8) = make_uint2(pack2bf_hw(t[m][0], t[m][1]), pack2bf_hw(t[m][2], t[m][3]));
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  const int cpr = ncols >> 3;                          // 16-byte chunks per row
  const int total = 16 * cpr;
#pragma unroll
  for (int i = 0; i < (16 * MT * 2 + 63) / 64; ++i) {
    const int id = lane + 64 * i;
    if (id < total) {
      const int r = id / cpr, c = id - r * cpr;
      if (r < rows_ok) *(uint4*)(dst + (long)r * row_stride + c * 8) = *(const uint4*)(stage + r * PITCH + c * 16);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// NT: key tiles of 16 held in registers (N <= 16 NT); NW waves x 16 queries per workgroup.  N <= 256: 8 waves (two per
// SIMD, <= 256 registers each); N <= 512: 4 waves (one per SIMD: 128 logit + 76 output accumulators per lane).
template <int NT, int NW>
__global__ __launch_bounds__(NW * 64) void ipa_fused_fwd_kernel(const IpaFusedParams p) {
  constexpr int NTHR = NW * 64;
  constexpr int IF_SLOTS = (3328 + NTHR - 1) / NTHR;     // 16-byte register slots per thread for the chunk in flight
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, l4 = lane >> 4;
  const int N = p.N, H = p.H, NP = p.NP;
  const int nqb = (N + NW * 16 - 1) / (NW * 16);
  // XCD-aware work ids: the query blocks of one (window, frame, head) share K' / V'^T -- consecutive logical ids sit on
  // one XCD (blockIdx round-robins over the 8 XCDs); bijective for any grid size
  const unsigned nwg = gridDim.x, bid = blockIdx.x;
  const unsigned xq = nwg >> 3, xr = nwg & 7, xcd = bid & 7, xidx = bid >> 3;
  const unsigned lid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + xidx;
  const int qb = (int)(lid % (unsigned)nqb);
  const unsigned bh = lid / (unsigned)nqb;
  const int h = (int)(bh % (unsigned)H);
  const long bf = bh / (unsigned)H;
  const long b = bf / p.F;
  const int q0 = qb * (NW * 16) + w * 16, myq = q0 + l15;
  const bool qok = myq < N;
  const int qrow = qok ? myq : N - 1;
  const int nch = (N + IF_KC - 1) / IF_KC;
  const int kswz = (-(l15 >> 2)) & 3;
  const long headrow = (bf * H + h) * (long)N;          // first row of this (window, frame, head) in QP / KP / kn / Pb

  // ---- Q' fragments of this lane's query: 8 scalar + 5 point k-steps of 32 (B operand: [n = query][k]) ----
  bf16x8 qf[13];
  {
    const bf16_t* qs = p.q + ((bf * N + qrow) * H + h) * (long)IF_C + l4 * 8;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qf[ks] = *(const bf16x8*)(qs + ks * 32);
    const bf16_t* qp = p.QP + (headrow + qrow) * IF_PK + l4 * 8;
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) qf[8 + ks] = *(const bf16x8*)(qp + ks * 32);
  }

  // ---- chunk staging: global -> registers (one chunk ahead) -> LDS ----
  ifu32x4 st[IF_SLOTS];
  auto load_k = [&](int kc) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < IF_SLOTS; ++i) {
      const int id = tid + NTHR * i;
      if (id < 2048) {                                   // scalar part: 64 keys x 32 chunks of 8 channels
        const int r = id >> 5, c = id & 31;
        int key = kc * IF_KC + r;
        key = key < N ? key : N - 1;                     // (rows past the end: finite values, masked by -inf below)
        st[i] = *(const ifu32x4*)(p.kv + ((bf * N + key) * H + h) * (long)(2 * IF_C) + c * 8);
      } else if (id < 2048 + 1280) {                     // point part: 64 keys x 20 chunks
        const int id2 = id - 2048, r = id2 / 20, c = id2 - r * 20;
        int key = kc * IF_KC + r;
        key = key < N ? key : N - 1;
        st[i] = *(const ifu32x4*)(p.KP + (headrow + key) * IF_PK + c * 8);
      }
    }
  };
  auto commit_k = [&](char* buf) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < IF_SLOTS; ++i) {
      const int id = tid + NTHR * i;
      if (id < 2048) {
        const int r = id >> 5, c = id & 31;
        *(ifu32x4*)(buf + (c >> 4) * 16384 + if_a_tile_off(r, c & 15)) = st[i];
      } else if (id < 2048 + 1280) {
        const int id2 = id - 2048, r = id2 / 20, c = id2 - r * 20;
        if (c < 16)
          *(ifu32x4*)(buf + 32768 + if_a_tile_off(r, c)) = st[i];
        else
          *(ifu32x4*)(buf + 49152 + r * 64 + (((c - 16) ^ ((-(r >> 2)) & 3)) << 4)) = st[i];
      }
    }
  };
  auto load_v = [&](int kc) __attribute__((always_inline)) {
    const bf16_t* vb = p.VT + bh * (long)IF_VROWS * NP + kc * IF_KC;
#pragma unroll
    for (int i = 0; i < IF_SLOTS; ++i) {
      const int id = tid + NTHR * i;
      if (id < IF_VROWS * 8) st[i] = *(const ifu32x4*)(vb + (long)(id >> 3) * NP + (id & 7) * 8);
    }
  };
  auto commit_v = [&](char* buf) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < IF_SLOTS; ++i) {
      const int id = tid + NTHR * i;
      if (id < IF_VROWS * 8) *(ifu32x4*)(buf + (id >> 3) * IF_VPITCH + (id & 7) * 16) = st[i];
    }
  };

  // ---- phase 1: S^T = K' Q'^T, all keys of the row in accumulators ----
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  load_k(0);
  // per-key terms of the logits, shared by every query of the (window, frame, head): |k_pts - ctr|^2 bias and mask - 1
  float* const ldsKN = (float*)(smem + 2 * IF_BUF);
  float* const ldsM1 = ldsKN + NT * 16;
  for (int t = tid; t < NT * 16; t += NTHR) {
    ldsKN[t] = t < N ? p.kn[headrow + t] : 0.f;
    ldsM1[t] = t < N ? p.mask[bf * N + t] - 1.f : 0.f;
  }
  commit_k(smem);
  __syncthreads();
#pragma unroll
  for (int c = 0; c < NT / 4; ++c) {
    if (c < nch) {
      char* const buf = smem + (c & 1) * IF_BUF;
      if (c + 1 < nch) load_k(c + 1);
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        const int row = tt * 16 + l15;
        f32x4 a = acc[4 * c + tt];
#pragma unroll
        for (int ks = 0; ks < 12; ++ks)
          a = IF_MFMA(*(const bf16x8*)(buf + (ks >> 2) * 16384 + if_a_tile_off(row, (ks & 3) * 4 + l4)), qf[ks], a);
        a = IF_MFMA(*(const bf16x8*)(buf + 49152 + row * 64 + ((l4 ^ kswz) << 4)), qf[12], a);
        acc[4 * c + tt] = a;
      }
      if (c + 1 < nch) commit_k(smem + ((c + 1) & 1) * IF_BUF);
      __syncthreads();
    }
  }

  // ---- softmax over the keys of each query (exact: the whole row is in registers) ----
  load_v(0);                                            // first V'^T chunk in flight under the softmax
  {
    const float mi_inf = p.mask[bf * N + qrow] * p.inf;
    const float* brow = p.bias + ((b * H + h) * (long)N + qrow) * N;
    // All pair-bias loads of the row go out before the first one is consumed, unconditionally (keys past the end re-read the
    // last four): under `if (key0 < N)` every key tile became its own branch with load - wait - arithmetic inside, NT memory
    // round trips in a row with nothing to overlap them.  The per-key terms (|k_pts|^2 bias, mask) are the same for every
    // query: staged in LDS once per workgroup (below the chunk buffers' end) instead of two more loads per tile and lane.
    f32x4 bvs[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int key0 = t * 16 + l4 * 4;
      bvs[t] = *(const f32x4*)(brow + (key0 < N ? key0 : N - 4));       // N % 4 == 0
    }
    __builtin_amdgcn_sched_group_barrier(0x020, NT, 0);
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int key0 = t * 16 + l4 * 4;
      const f32x4 kv = *(const f32x4*)(ldsKN + key0);
      const f32x4 mv = *(const f32x4*)(ldsM1 + key0);
      const bool kin = key0 < N;                         // the four keys of a lane are in or out together
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float L = __builtin_fmaf(p.alpha, acc[t][r], __builtin_fmaf(p.bias_scale, bvs[t][r], kv[r])) + mi_inf * mv[r];
        acc[t][r] = kin ? L : -INFINITY;
        mx = fmaxf(mx, acc[t][r]);
      }
    }
    mx = if_xmax(mx);
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = __expf(acc[t][r] - mx);
        acc[t][r] = e;
        sum += e;
      }
    sum = if_xsum(sum);
    const float inv = 1.f / sum;
    float* prow = p.P ? p.P + (headrow + qrow) * (long)N : nullptr;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int key0 = t * 16 + l4 * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[t][r] *= inv;
      if (prow && qok && key0 < N) *(f32x4*)(prow + key0) = acc[t];      // (diagnostic copy: partial-line stores)
    }
    // the LDS buffers are idle here (the phase-1 loop ended with a barrier, the first V'^T chunk is still in registers)
    if (q0 < N)
      if_store_rows_bf16<NT>(smem + w * (16 * (NT * 32 + 16)), acc, p.Pb + (headrow + q0) * (long)N, N, min(16, N - q0), N, lane);
  }
  __syncthreads();                                      // staging areas are read before the V'^T chunk lands in them

  // ---- phase 2: O^T = V'^T P^T ----
  f32x4 oacc[16], pacc[3];
#pragma unroll
  for (int m = 0; m < 16; ++m) oacc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int m = 0; m < 3; ++m) pacc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
  commit_v(smem);
  __syncthreads();
#pragma unroll
  for (int c = 0; c < NT / 4; ++c) {
    if (c < nch) {
      char* const buf = smem + (c & 1) * IF_BUF;
      if (c + 1 < nch) load_v(c + 1);
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int t0 = 4 * c + 2 * s2;
        // MFMA k-slot e of lane group l4 <-> key (t0 + (e >> 2)) * 16 + l4 * 4 + (e & 3): what the accumulators hold
        const f32x4 x0 = acc[t0], x1 = acc[t0 + 1];
        const ifu32x4 phv = {pack2bf_hw(x0[0], x0[1]), pack2bf_hw(x0[2], x0[3]), pack2bf_hw(x1[0], x1[1]), pack2bf_hw(x1[2], x1[3])};
        const ifu32x4 plv = {pack2bf_hw(x0[0] - bf_lo(phv.x), x0[1] - bf_hi(phv.x)), pack2bf_hw(x0[2] - bf_lo(phv.y), x0[3] - bf_hi(phv.y)),
                             pack2bf_hw(x1[0] - bf_lo(phv.z), x1[1] - bf_hi(phv.z)), pack2bf_hw(x1[2] - bf_lo(phv.w), x1[3] - bf_hi(phv.w))};
        const bf16x8 ph = __builtin_bit_cast(bf16x8, phv), pl = __builtin_bit_cast(bf16x8, plv);
        const char* vcol = buf + s2 * 64 + l4 * 8;
        auto afrag = [&](int row) __attribute__((always_inline)) {
          const char* vp = vcol + row * IF_VPITCH;
          const ifu32x2 lo = *(const ifu32x2*)vp, hi = *(const ifu32x2*)(vp + 32);
          const ifu32x4 av = {lo.x, lo.y, hi.x, hi.y};
          return __builtin_bit_cast(bf16x8, av);
        };
#pragma unroll
        for (int m = 0; m < 16; ++m) oacc[m] = IF_MFMA(afrag(m * 16 + l15), ph, oacc[m]);
#pragma unroll
        for (int m = 0; m < 3; ++m) {
          const bf16x8 vh = afrag(IF_C + m * 16 + l15), vm = afrag(IF_C + 48 + m * 16 + l15), vl = afrag(IF_C + 96 + m * 16 + l15);
          f32x4 a = pacc[m];
          a = IF_MFMA(vl, ph, a);
          a = IF_MFMA(vm, pl, a);
          a = IF_MFMA(vm, ph, a);
          a = IF_MFMA(vh, pl, a);
          a = IF_MFMA(vh, ph, a);
          pacc[m] = a;
        }
      }
      if (c + 1 < nch) commit_v(smem + ((c + 1) & 1) * IF_BUF);
      __syncthreads();
    }
  }

  // ---- epilogue: lane holds query l15, rows (channels / point components) m*16 + l4*4 + r ----
  if (q0 < N)       // (the phase-2 loop ended with a barrier: the LDS buffers are idle)
    if_store_rows_bf16<16>(smem + w * (16 * (16 * 32 + 16)), oacc, p.o + (bf * N + q0) * p.o_ld + h * (long)IF_C, p.o_ld,
                           min(16, N - q0), IF_C, lane);
  if (qok) {
    float* prow = p.o_pt + ((bf * N + myq) * H + h) * 36L;
    const float c0 = p.ctr[bf * 3], c1 = p.ctr[bf * 3 + 1], c2 = p.ctr[bf * 3 + 2];
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      const int cb = m * 16 + l4 * 4;                    // component index of r = 0; (cb + r) % 3 selects x / y / z
      if (cb < 36) {
        f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int k3 = (cb + r) % 3;
          v[r] = pacc[m][r] + (k3 == 0 ? c0 : (k3 == 1 ? c1 : c2));
        }
        *(f32x4*)(prow + cb) = v;
      }
    }
  }
}

extern "C" int dfold_ipa_fused_fwd(const void* q_bf16, const void* kv_bf16, const void* QP_bf16, const void* KP_bf16,
                                   const void* VT_bf16, const float* kn, const float* bias, const float* mask, const float* ctr,
                                   void* o_bf16, int64_t o_ld, float* o_pt, void* P_bf16, float* P_f32, int32_t B, int32_t F,
                                   int32_t N, int32_t H, int32_t NP, float alpha, float bias_scale, float inf, void* stream) {
  if (!q_bf16 || !kv_bf16 || !QP_bf16 || !KP_bf16 || !VT_bf16 || !kn || !bias || !mask || !ctr || !o_bf16 || !o_pt || !P_bf16)
    return DFOLD_EINVAL;
  if (B <= 0 || F <= 0 || N <= 0 || H <= 0 || (N & 7) || N > 512 || NP < N || (NP % IF_KC)) return DFOLD_EINVAL;
  if (o_ld < (int64_t)H * IF_C || (o_ld & 7) || ((uintptr_t)o_bf16 & 15)) return DFOLD_EINVAL;
  const int qpw = N <= 256 ? 128 : 64;                   // queries per workgroup (8 resp. 4 waves)
  const long nwg = (long)B * F * H * ((N + qpw - 1) / qpw);
  if (nwg > 0x7fffffffL) return DFOLD_EINVAL;
  IpaFusedParams p;
  p.q = (const bf16_t*)q_bf16; p.kv = (const bf16_t*)kv_bf16; p.QP = (const bf16_t*)QP_bf16; p.KP = (const bf16_t*)KP_bf16;
  p.VT = (const bf16_t*)VT_bf16; p.kn = kn; p.bias = bias; p.mask = mask; p.ctr = ctr; p.o = (bf16_t*)o_bf16; p.o_ld = o_ld; p.o_pt = o_pt;
  p.Pb = (bf16_t*)P_bf16; p.P = P_f32; p.BF = B * F; p.F = F; p.N = N; p.H = H; p.NP = NP; p.alpha = alpha;
  p.bias_scale = bias_scale; p.inf = inf;
  hipStream_t st = (hipStream_t)stream;
  if (N <= 256) {
    DFOLD_MAX_LDS_ONCE((ipa_fused_fwd_kernel<16, 8>), 2 * IF_BUF + 2 * 256 * 4);
    DFOLD_LAUNCH((ipa_fused_fwd_kernel<16, 8>), dim3((unsigned)nwg), dim3(512), 2 * IF_BUF + 2 * 256 * 4, st, p);
  } else {
    DFOLD_MAX_LDS_ONCE((ipa_fused_fwd_kernel<32, 4>), 2 * IF_BUF + 2 * 512 * 4);
    DFOLD_LAUNCH((ipa_fused_fwd_kernel<32, 4>), dim3((unsigned)nwg), dim3(256), 2 * IF_BUF + 2 * 512 * 4, st, p);
  }
  return dfold_check_launch();
}
