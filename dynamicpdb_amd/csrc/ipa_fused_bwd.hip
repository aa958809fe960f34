// Fused row pass of the Invariant Point Attention backward for gfx950 (reference: the autograd of
// InvariantPointAttention.forward, src/model/ipa_pytorch_dynamic.py:396-469).  One launch per IPA block replaces the
// dP = dO V^T product, the accumulation of the pair-value term into it, the VALU row pass (csrc/ipa_attn.hip) and the
// dQ = dS K product with its transposed copy of K:
//
//   g[i,j]  = do_i . v_j + do_pt_i . (vp_j - ctr) + dP_pair[i,j]            (the part of do_pt_i . ctr is constant along a row
//                                                                            of g and drops out of dS below)
//   dS[i,j] = P[i,j] (g[i,j] - sum_j' P[i,j'] g[i,j'] / sum_j' P[i,j'])      P = the forward's bf16 probabilities
//   dq_i    = alpha sum_j dS[i,j] k_j,   A_i = sum_j dS[i,j] (kp_j - ctr),   dq_pts_i = hw A_i,
//   dhw    += -1/2 sum_i ( sum_j dS[i,j] |kp_j - ctr|^2 - 2 (qp_i - ctr) . A_i )
//
// The structure mirrors csrc/ipa_fused.hip (whose comments explain the layouts): G^T = V'' dO''^T with the keys as the A
// rows and the wave's 16 queries as the B columns, V'' = [v (256) | (vp - ctr) as bf16 pieces], dO'' = [do (256) | do_pt as
// bf16 pieces] (three pieces each, the six products hh, hm, mh, hl, lh, mm as 6 x 36 (+ 8 zero) extra K columns); a wave
// holds ALL keys of its 16 queries in accumulators, so the row sums are exact; dS sits in the accumulators in the B-operand
// layout of the second product  dQ''^T = K''^T dS^T,  K''^T = [k^T (256 rows) | (kp - ctr, |kp - ctr|^2) pieces (3 x 32
// rows)], with dS as TWO bf16 pieces for the point rows (dq_pts leans on sum_j dS = 0 to 2^-17, not 2^-9).
#include "dfold_common.h"
#include "../../include/dfold_hip.h"
#include <math.h>

typedef __attribute__((ext_vector_type(4))) unsigned ibu32x4;
typedef __attribute__((ext_vector_type(2))) unsigned ibu32x2;
#define IB_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)

#define IB_C 256        // scalar channels per head
#define IB_PK 224       // point columns of dO'' / V'': 6 products x 36 components + 8 zeros
#define IB_KROWS 352    // rows of K''^T: 256 key channels + 3 pieces x 32 (24 point components, |kp|^2, 7 zeros)
#define IB_KC 64        // keys per LDS chunk
#define IB_KPITCH 144   // bytes per K''^T row in LDS: 64 keys x 2 B + 16
#define IB_BUF 65536    // bytes per buffer: 4 A tiles of 16 KiB (phase 1) >= 352 x 144 (phase 2)

__device__ __forceinline__ int ib_a_tile_off(int row, int chunk) { return row * 256 + ((chunk ^ (row & 15)) << 4); }

// ------------------------------------------------------------------------------------------------------------------
// operand preparation
// ------------------------------------------------------------------------------------------------------------------
// do_pt, v_pts fp32 [B*F, N, H, 12, 3] -> DOP, VP bf16 [B*F, H, N, 224] (one thread per (window*frame, residue, head))
__global__ __launch_bounds__(256) void ipa_bwd_prep_dov_kernel(const float* __restrict__ do_pt, const float* __restrict__ v_pts,
                                                               const float* __restrict__ ctr, bf16_t* __restrict__ DOP,
                                                               bf16_t* __restrict__ VP, long total, int N, int H) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int h = (int)(idx % H);
  const long rn = idx / H;
  const int n = (int)(rn % N);
  const long bf = rn / N;
  const float* d = do_pt + idx * 36;
  const float* v = v_pts + idx * 36;
  const float c0 = ctr[bf * 3], c1 = ctr[bf * 3 + 1], c2 = ctr[bf * 3 + 2];
  __attribute__((aligned(16))) uint32_t dq_[IB_PK / 2], vq_[IB_PK / 2];
#pragma unroll
  for (int c = 0; c < 36; c += 2) {
    const float ca = (c % 3 == 0) ? c0 : ((c % 3 == 1) ? c1 : c2);
    const float cb = ((c + 1) % 3 == 0) ? c0 : (((c + 1) % 3 == 1) ? c1 : c2);
    const float da = d[c], db = d[c + 1], va = v[c] - ca, vb = v[c + 1] - cb;
    const uint32_t dh = pack2bf_hw(da, db), vh = pack2bf_hw(va, vb);
    const float dra = da - bf_lo(dh), drb = db - bf_hi(dh), vra = va - bf_lo(vh), vrb = vb - bf_hi(vh);
    const uint32_t dm = pack2bf_hw(dra, drb), vm = pack2bf_hw(vra, vrb);
    const uint32_t dl = pack2bf_hw(dra - bf_lo(dm), drb - bf_hi(dm)), vl = pack2bf_hw(vra - bf_lo(vm), vrb - bf_hi(vm));
    const int i2 = c >> 1;
    // products hh, hm, mh, hl, lh, mm
    dq_[i2] = dh;       vq_[i2] = vh;
    dq_[18 + i2] = dh;  vq_[18 + i2] = vm;
    dq_[36 + i2] = dm;  vq_[36 + i2] = vh;
    dq_[54 + i2] = dh;  vq_[54 + i2] = vl;
    dq_[72 + i2] = dl;  vq_[72 + i2] = vh;
    dq_[90 + i2] = dm;  vq_[90 + i2] = vm;
  }
#pragma unroll
  for (int c = 108; c < IB_PK / 2; ++c) {
    dq_[c] = 0u;
    vq_[c] = 0u;
  }
  const long row = (bf * H + h) * N + n;
  uint4* dd = (uint4*)(DOP + row * IB_PK);
  uint4* vd = (uint4*)(VP + row * IB_PK);
#pragma unroll
  for (int u = 0; u < IB_PK / 8; ++u) {
    dd[u] = make_uint4(dq_[4 * u], dq_[4 * u + 1], dq_[4 * u + 2], dq_[4 * u + 3]);
    vd[u] = make_uint4(vq_[4 * u], vq_[4 * u + 1], vq_[4 * u + 2], vq_[4 * u + 3]);
  }
}

__device__ __forceinline__ void ib_split3(float x, bf16_t& h, bf16_t& m, bf16_t& l) {
  h = f2bf_hw(x);
  float r = x - bf2f(h);
  m = f2bf_hw(r);
  r -= bf2f(m);
  l = f2bf_hw(r);
}

// k_pts fp32 [B*F, N, H, 8, 3] -> rows 256 .. 351 of KT bf16 [B*F, H, 352, NP]: three pieces x 32 rows (component c < 24:
// kp_c - ctr; row 24: |kp - ctr|^2; rows 25..31 stay zero), key-contiguous
__global__ __launch_bounds__(256) void ipa_bwd_prep_k_kernel(const float* __restrict__ k_pts, const float* __restrict__ ctr,
                                                             bf16_t* __restrict__ KT, long total, int N, int H, int NP) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;      // ((bf * H + h) * N + j): lanes walk the keys
  if (idx >= total) return;
  const int j = (int)(idx % N);
  const long bh = idx / N;
  const int h = (int)(bh % H);
  const long bf = bh / H;
  const float* k = k_pts + ((bf * N + j) * H + h) * 24;
  const float c0 = ctr[bf * 3], c1 = ctr[bf * 3 + 1], c2 = ctr[bf * 3 + 2];
  bf16_t* base = KT + (bh * IB_KROWS + IB_C) * (long)NP + j;
  float k2 = 0.f;
#pragma unroll
  for (int c = 0; c < 24; ++c) {
    const float cc = (c % 3 == 0) ? c0 : ((c % 3 == 1) ? c1 : c2);
    const float x = k[c] - cc;
    k2 = __builtin_fmaf(x, x, k2);
    bf16_t xh, xm, xl;
    ib_split3(x, xh, xm, xl);
    base[(long)c * NP] = xh;
    base[(long)(32 + c) * NP] = xm;
    base[(long)(64 + c) * NP] = xl;
  }
  bf16_t xh, xm, xl;
  ib_split3(k2, xh, xm, xl);
  base[24L * NP] = xh;
  base[(32L + 24) * NP] = xm;
  base[(64L + 24) * NP] = xl;
}

extern "C" int dfold_ipa_bwd_prep(const float* do_pt, const float* v_pts, const float* k_pts, const float* ctr, void* DOP_bf16,
                                  void* VP_bf16, void* KT_bf16, int32_t B, int32_t F, int32_t N, int32_t H, int32_t NP,
                                  void* stream) {
  if (!do_pt || !v_pts || !k_pts || !ctr || !DOP_bf16 || !VP_bf16 || !KT_bf16) return DFOLD_EINVAL;
  if (B <= 0 || F <= 0 || N <= 0 || H <= 0 || NP < N || (NP % IB_KC)) return DFOLD_EINVAL;
  const long total = (long)B * F * N * H;
  const unsigned grid = (unsigned)((total + 255) / 256);
  hipStream_t st = (hipStream_t)stream;
  DFOLD_LAUNCH(ipa_bwd_prep_dov_kernel, dim3(grid), dim3(256), 0, st, do_pt, v_pts, ctr, (bf16_t*)DOP_bf16, (bf16_t*)VP_bf16, total, N, H);
  if (dfold_check_launch() != DFOLD_OK) return DFOLD_ELAUNCH;
  DFOLD_LAUNCH(ipa_bwd_prep_k_kernel, dim3(grid), dim3(256), 0, st, k_pts, ctr, (bf16_t*)KT_bf16, total, N, H, NP);
  return dfold_check_launch();
}

// ------------------------------------------------------------------------------------------------------------------
// the row-pass kernel
// ------------------------------------------------------------------------------------------------------------------
struct IpaBwdParams {
  const bf16_t* dout;   // do  [B*F, N, H*256]
  const bf16_t* kv;     // [B*F, N, H*512]  (k | v per head): the v half is the A operand of the first product
  const bf16_t* DOP;    // [B*F, H, N, 224]
  const bf16_t* VP;     // [B*F, H, N, 224]
  const bf16_t* KT;     // [B*F, H, 352, NP]
  const bf16_t* Pb;     // [B*F, H, N, N]
  const bf16_t* dPp;    // [B*F, H, N, N] pair-value term of dP (bf16) or null
  const float* q_pts;   // [B*F, N, H, 24]
  const float* hw;      // [H]
  const float* ctr;     // [B*F, 3]
  float* dS;            // [B*F, H, N, N]
  bf16_t* dSb;          // [B*F, H, N, N]
  bf16_t* dq;           // [B*F, N, H*256]
  float* dq_pts;        // [B*F, N, H, 24]
  float* dhw;           // [H] (atomics)
  int BF, N, H, NP;
  float alpha;
};

__device__ __forceinline__ float ib_xsum(float v) {
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}

// A wave's [16 queries x (16 MT) columns] tile (lane: query l15, columns m*16 + l4*4 + r) leaves through a wave-private
// LDS staging area so that every global store instruction writes whole 16-byte chunks of whole rows (see ipa_fused.hip)
template <int MT>
__device__ __forceinline__ void ib_store_rows_bf16(char* stage, const f32x4 (&t)[MT], float scale, bf16_t* dst, long row_stride,
                                                   int rows_ok, int ncols, int lane) {
  constexpr int PITCH = MT * 32 + 16;
  const int l15 = lane & 15, l4 = lane >> 4;
#pragma unroll
  for (int m = 0; m < MT; ++m)
    *(uint2*)(stage + l15 * PITCH + m * 32 + l4 * 8) =
        make_uint2(pack2bf_hw(t[m][0] * scale, t[m][1] * scale), pack2bf_hw(t[m][2] * scale, t[m][3] * scale));
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  const int cpr = ncols >> 3;
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
// fp32 form for the MT tiles T0 .. T0 + MT - 1 of the wave's NTT (called once per column half: the staging area keeps the
// size of the bf16 form)
template <int MT, int T0, int NTT>
__device__ __forceinline__ void ib_store_rows_f32(char* stage, const f32x4 (&t)[NTT], float* dst, long row_stride, int rows_ok,
                                                  int ncols, int lane) {
  constexpr int PITCH = MT * 64 + 16;
  const int l15 = lane & 15, l4 = lane >> 4;
#pragma unroll
  for (int m = 0; m < MT; ++m) *(f32x4*)(stage + l15 * PITCH + m * 64 + l4 * 16) = t[T0 + m];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  const int cpr = ncols >> 2;                          // 16-byte chunks (4 floats) per row
  const int total = 16 * cpr;
#pragma unroll
  for (int i = 0; i < (16 * MT * 4 + 63) / 64; ++i) {
    const int id = lane + 64 * i;
    if (id < total) {
      const int r = id / cpr, c = id - r * cpr;
      if (r < rows_ok) *(uint4*)(dst + (long)r * row_stride + c * 4) = *(const uint4*)(stage + r * PITCH + c * 16);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// NT: key tiles of 16 held in registers (N <= 16 NT); NW waves x 16 queries per workgroup
template <int NT, int NW>
__global__ __launch_bounds__(NW * 64) void ipa_fused_bwd_kernel(const IpaBwdParams p) {
  constexpr int NTHR = NW * 64;
  constexpr int IB_SLOTS = (3840 + NTHR - 1) / NTHR;     // 16-byte register slots per thread for the chunk in flight
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, l4 = lane >> 4;
  const int N = p.N, H = p.H, NP = p.NP;
  const int nqb = (N + NW * 16 - 1) / (NW * 16);
  const unsigned nwg = gridDim.x, bid = blockIdx.x;
  const unsigned xq = nwg >> 3, xr = nwg & 7, xcd = bid & 7, xidx = bid >> 3;
  const unsigned lid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + xidx;
  const int qb = (int)(lid % (unsigned)nqb);
  const unsigned bh = lid / (unsigned)nqb;
  const int h = (int)(bh % (unsigned)H);
  const long bf = bh / (unsigned)H;
  const int q0 = qb * (NW * 16) + w * 16, myq = q0 + l15;
  const bool qok = myq < N;
  const int qrow = qok ? myq : N - 1;
  const int nch = (N + IB_KC - 1) / IB_KC;
  const long headrow = (bf * H + h) * (long)N;          // first row of this (window, frame, head) in DOP / VP / Pb / dS

  // ---- dO'' fragments of this lane's query: 8 scalar + 7 point k-steps of 32 (B operand: [n = query][k]) ----
  bf16x8 df[15];
  {
    const bf16_t* ds = p.dout + ((bf * N + qrow) * H + h) * (long)IB_C + l4 * 8;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) df[ks] = *(const bf16x8*)(ds + ks * 32);
    const bf16_t* dp = p.DOP + (headrow + qrow) * IB_PK + l4 * 8;
#pragma unroll
    for (int ks = 0; ks < 7; ++ks) df[8 + ks] = *(const bf16x8*)(dp + ks * 32);
  }

  // ---- chunk staging: global -> registers (one chunk ahead) -> LDS ----
  // (addresses: a wave-uniform 64-bit base per operand + 32-bit lane offsets -- per-slot 64-bit pointers were what the
  //  <32, 4> instance parked in scratch)
  ibu32x4 st[IB_SLOTS];
  const bf16_t* const vbase = p.kv + ((bf * N) * H + h) * (long)(2 * IB_C) + IB_C;      // v half of (window, frame, head)
  const bf16_t* const vpbase = p.VP + headrow * IB_PK;
  auto load_v = [&](int kc) __attribute__((always_inline)) {      // phase 1: 64 keys x (32 scalar + 28 point) chunks
#pragma unroll
    for (int i = 0; i < IB_SLOTS; ++i) {
      const int id = tid + NTHR * i;
      if (id < 2048) {
        const int r = id >> 5, c = id & 31;
        int key = kc * IB_KC + r;
        key = key < N ? key : N - 1;                     // (rows past the end: finite values; their P is zero)
        st[i] = *(const ibu32x4*)(vbase + (unsigned)(key * H * (2 * IB_C) + c * 8));
      } else if (id < 2048 + 1792) {
        const int id2 = id - 2048, r = id2 / 28, c = id2 - r * 28;
        int key = kc * IB_KC + r;
        key = key < N ? key : N - 1;
        st[i] = *(const ibu32x4*)(vpbase + (unsigned)(key * IB_PK + c * 8));
      }
    }
  };
  auto commit_v = [&](char* buf) __attribute__((always_inline)) {   // four A tiles of [64 keys][128 columns]
#pragma unroll
    for (int i = 0; i < IB_SLOTS; ++i) {
      const int id = tid + NTHR * i;
      if (id < 2048) {
        const int r = id >> 5, c = id & 31;
        *(ibu32x4*)(buf + (c >> 4) * 16384 + ib_a_tile_off(r, c & 15)) = st[i];
      } else if (id < 2048 + 1792) {
        const int id2 = id - 2048, r = id2 / 28, c = id2 - r * 28;
        *(ibu32x4*)(buf + 32768 + (c >> 4) * 16384 + ib_a_tile_off(r, c & 15)) = st[i];
      }
    }
  };
  auto load_k = [&](int kc) __attribute__((always_inline)) {      // phase 2: 352 rows x 8 chunks of 8 keys
    const bf16_t* kb = p.KT + bh * (long)IB_KROWS * NP + kc * IB_KC;
#pragma unroll
    for (int i = 0; i < IB_SLOTS; ++i) {
      const int id = tid + NTHR * i;
      if (id < IB_KROWS * 8) st[i] = *(const ibu32x4*)(kb + (unsigned)((id >> 3) * NP + (id & 7) * 8));
    }
  };
  auto commit_k = [&](char* buf) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < IB_SLOTS; ++i) {
      const int id = tid + NTHR * i;
      if (id < IB_KROWS * 8) *(ibu32x4*)(buf + (id >> 3) * IB_KPITCH + (id & 7) * 16) = st[i];
    }
  };

  // ---- phase 1: G^T = V'' dO''^T, all keys of the row in accumulators ----
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  load_v(0);
  commit_v(smem);
  __syncthreads();
#pragma unroll
  for (int c = 0; c < NT / 4; ++c) {
    if (c < nch) {
      char* const buf = smem + (c & 1) * IB_BUF;
      if (c + 1 < nch) load_v(c + 1);
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        const int row = tt * 16 + l15;
        f32x4 a = acc[4 * c + tt];
#pragma unroll
        for (int ks = 0; ks < 15; ++ks)
          a = IB_MFMA(*(const bf16x8*)(buf + (ks >> 2) * 16384 + ib_a_tile_off(row, (ks & 3) * 4 + l4)), df[ks], a);
        acc[4 * c + tt] = a;
      }
      if (c + 1 < nch) commit_v(smem + ((c + 1) & 1) * IB_BUF);
      __syncthreads();
    }
  }

  // ---- dS = P (g - <g>_P) over the keys of each query (exact: the whole row is in registers) ----
  load_k(0);                                            // first K''^T chunk in flight under the row arithmetic
  {
    const bf16_t* const pbase = p.Pb + headrow * (long)N;
    const bf16_t* const ppbase = p.dPp ? p.dPp + headrow * (long)N : nullptr;
    const unsigned prow = (unsigned)(qrow * N);
    // The probabilities stay PACKED (two registers per key tile instead of four) between the row sums and the product: with
    // the unpacked copy next to the accumulators and the K''^T chunk in flight the <32, 4> instance spilled 204 bytes per lane
    ibu32x2 praw[NT];
    float dot = 0.f, psum = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int key0 = t * 16 + l4 * 4;
      praw[t] = (ibu32x2){0u, 0u};
      if (key0 < N) {                                    // N % 4 == 0: the four keys of a lane are in or out together
        praw[t] = *(const ibu32x2*)(pbase + (prow + key0));
        if (ppbase) {
          const ibu32x2 gv = *(const ibu32x2*)(ppbase + (prow + key0));
          acc[t][0] += bf_lo(gv.x);
          acc[t][1] += bf_hi(gv.x);
          acc[t][2] += bf_lo(gv.y);
          acc[t][3] += bf_hi(gv.y);
        }
      }
      const float p0 = bf_lo(praw[t].x), p1 = bf_hi(praw[t].x), p2 = bf_lo(praw[t].y), p3 = bf_hi(praw[t].y);
      dot = __builtin_fmaf(p0, acc[t][0], dot);
      dot = __builtin_fmaf(p1, acc[t][1], dot);
      dot = __builtin_fmaf(p2, acc[t][2], dot);
      dot = __builtin_fmaf(p3, acc[t][3], dot);
      psum += p0;
      psum += p1;
      psum += p2;
      psum += p3;
    }
    dot = ib_xsum(dot);
    psum = ib_xsum(psum);
    const float mean = dot / psum;                       // a row always holds probability mass (psum ~ 1)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      unsigned px = praw[t].x, py = praw[t].y;
      asm volatile("" : "+v"(px), "+v"(py));             // (opaque: the unpacked values of the first pass must not be kept alive)
      acc[t][0] = bf_lo(px) * (acc[t][0] - mean);
      acc[t][1] = bf_hi(px) * (acc[t][1] - mean);
      acc[t][2] = bf_lo(py) * (acc[t][2] - mean);
      acc[t][3] = bf_hi(py) * (acc[t][3] - mean);
    }
    // the LDS buffers are idle here (the phase-1 loop ended with a barrier, the first K''^T chunk is still in registers)
    if (q0 < N) {
      char* stage = smem + w * (16 * (NT * 32 + 16));
      const int rows_ok = min(16, N - q0);
      ib_store_rows_bf16<NT>(stage, acc, 1.f, p.dSb + (headrow + q0) * (long)N, N, rows_ok, N, lane);
      float* drow = p.dS + (headrow + q0) * (long)N;
      ib_store_rows_f32<NT / 2, 0, NT>(stage, acc, drow, N, rows_ok, min(N, NT * 8), lane);
      if (N > NT * 8) ib_store_rows_f32<NT / 2, NT / 2, NT>(stage, acc, drow + NT * 8, N, rows_ok, N - NT * 8, lane);
    }
  }
  __syncthreads();                                      // staging areas are read before the K''^T chunk lands in them

  // ---- phase 2: dQ''^T = K''^T dS^T ----
  f32x4 oacc[16], pacc[2];
#pragma unroll
  for (int m = 0; m < 16; ++m) oacc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int m = 0; m < 2; ++m) pacc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
  commit_k(smem);
  __syncthreads();
#pragma unroll
  for (int c = 0; c < NT / 4; ++c) {
    if (c < nch) {
      char* const buf = smem + (c & 1) * IB_BUF;
      if (c + 1 < nch) load_k(c + 1);
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int t0 = 4 * c + 2 * s2;
        // MFMA k-slot e of lane group l4 <-> key (t0 + (e >> 2)) * 16 + l4 * 4 + (e & 3): what the accumulators hold
        const f32x4 x0 = acc[t0], x1 = acc[t0 + 1];
        const ibu32x4 phv = {pack2bf_hw(x0[0], x0[1]), pack2bf_hw(x0[2], x0[3]), pack2bf_hw(x1[0], x1[1]), pack2bf_hw(x1[2], x1[3])};
        const ibu32x4 plv = {pack2bf_hw(x0[0] - bf_lo(phv.x), x0[1] - bf_hi(phv.x)), pack2bf_hw(x0[2] - bf_lo(phv.y), x0[3] - bf_hi(phv.y)),
                             pack2bf_hw(x1[0] - bf_lo(phv.z), x1[1] - bf_hi(phv.z)), pack2bf_hw(x1[2] - bf_lo(phv.w), x1[3] - bf_hi(phv.w))};
        const bf16x8 ph = __builtin_bit_cast(bf16x8, phv), pl = __builtin_bit_cast(bf16x8, plv);
        const char* kcol = buf + s2 * 64 + l4 * 8;
        auto afrag = [&](int row) __attribute__((always_inline)) {
          const char* kp = kcol + row * IB_KPITCH;
          const ibu32x2 lo = *(const ibu32x2*)kp, hi = *(const ibu32x2*)(kp + 32);
          const ibu32x4 av = {lo.x, lo.y, hi.x, hi.y};
          return __builtin_bit_cast(bf16x8, av);
        };
#pragma unroll
        for (int m = 0; m < 16; ++m) oacc[m] = IB_MFMA(afrag(m * 16 + l15), ph, oacc[m]);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const bf16x8 kh = afrag(IB_C + m * 16 + l15), km = afrag(IB_C + 32 + m * 16 + l15), kl = afrag(IB_C + 64 + m * 16 + l15);
          f32x4 a = pacc[m];
          a = IB_MFMA(kl, ph, a);
          a = IB_MFMA(km, pl, a);
          a = IB_MFMA(km, ph, a);
          a = IB_MFMA(kh, pl, a);
          a = IB_MFMA(kh, ph, a);
          pacc[m] = a;
        }
      }
      if (c + 1 < nch) commit_k(smem + ((c + 1) & 1) * IB_BUF);
      __syncthreads();
    }
  }

  // ---- epilogue: lane holds query l15, rows (channels / point components) m*16 + l4*4 + r ----
  if (q0 < N)       // (the phase-2 loop ended with a barrier: the LDS buffers are idle)
    ib_store_rows_bf16<16>(smem + w * (16 * (16 * 32 + 16)), oacc, p.alpha, p.dq + ((bf * N + q0) * H + h) * (long)IB_C, (long)H * IB_C,
                           min(16, N - q0), IB_C, lane);
  float dh = 0.f;
  if (qok) {
    const float hwh = p.hw[h];
    float* drow = p.dq_pts + ((bf * N + myq) * H + h) * 24L;
    const float* qrowp = p.q_pts + ((bf * N + myq) * H + h) * 24L;
    const float c0 = p.ctr[bf * 3], c1 = p.ctr[bf * 3 + 1], c2 = p.ctr[bf * 3 + 2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int cb = m * 16 + l4 * 4;                    // component index of r = 0; (cb + r) % 3 selects x / y / z
      if (cb < 24) {
        const f32x4 qv = *(const f32x4*)(qrowp + cb);
        f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int k3 = (cb + r) % 3;
          v[r] = hwh * pacc[m][r];
          dh += (qv[r] - (k3 == 0 ? c0 : (k3 == 1 ? c1 : c2))) * pacc[m][r];      // (q - ctr) . A
        }
        *(f32x4*)(drow + cb) = v;
      } else if (cb == 24) {
        dh -= 0.5f * pacc[m][0];                         // - 1/2 sum_j dS |kp - ctr|^2
      }
    }
  }
  // dhw = sum over queries of (q . A - 1/2 akn): one atomic per wave
  dh = wave_sum(dh);
  if (lane == 0 && dh != 0.f) atomicAdd(p.dhw + h, dh);
}

extern "C" int dfold_ipa_fused_bwd(const void* do_bf16, const void* kv_bf16, const void* DOP_bf16, const void* VP_bf16,
                                   const void* KT_bf16, const void* P_bf16, const void* dP_pair_bf16, const float* q_pts,
                                   const float* hw, const float* ctr, float* dS, void* dS_bf16, void* dq_bf16, float* dq_pts,
                                   float* dhw, int32_t B, int32_t F, int32_t N, int32_t H, int32_t NP, float alpha, void* stream) {
  if (!do_bf16 || !kv_bf16 || !DOP_bf16 || !VP_bf16 || !KT_bf16 || !P_bf16 || !q_pts || !hw || !ctr || !dS || !dS_bf16 || !dq_bf16 ||
      !dq_pts || !dhw)
    return DFOLD_EINVAL;
  if (B <= 0 || F <= 0 || N <= 0 || H <= 0 || (N & 7) || N > 512 || NP < N || (NP % IB_KC)) return DFOLD_EINVAL;
  const int qpw = N <= 256 ? 128 : 64;                   // queries per workgroup (8 resp. 4 waves)
  const long nwg = (long)B * F * H * ((N + qpw - 1) / qpw);
  if (nwg > 0x7fffffffL) return DFOLD_EINVAL;
  IpaBwdParams p;
  p.dout = (const bf16_t*)do_bf16; p.kv = (const bf16_t*)kv_bf16; p.DOP = (const bf16_t*)DOP_bf16; p.VP = (const bf16_t*)VP_bf16;
  p.KT = (const bf16_t*)KT_bf16; p.Pb = (const bf16_t*)P_bf16; p.dPp = (const bf16_t*)dP_pair_bf16; p.q_pts = q_pts; p.hw = hw;
  p.ctr = ctr; p.dS = dS; p.dSb = (bf16_t*)dS_bf16; p.dq = (bf16_t*)dq_bf16; p.dq_pts = dq_pts; p.dhw = dhw;
  p.BF = B * F; p.N = N; p.H = H; p.NP = NP; p.alpha = alpha;
  hipStream_t st = (hipStream_t)stream;
  if (N <= 256) {
    // LDS: two chunk buffers; the fp32 row staging of 8 waves (16 x (8 x 64 + 16) bytes each) fits inside them
    DFOLD_MAX_LDS_ONCE((ipa_fused_bwd_kernel<16, 8>), 2 * IB_BUF);
    DFOLD_LAUNCH((ipa_fused_bwd_kernel<16, 8>), dim3((unsigned)nwg), dim3(512), 2 * IB_BUF, st, p);
  } else {
    DFOLD_MAX_LDS_ONCE((ipa_fused_bwd_kernel<32, 4>), 2 * IB_BUF);
    DFOLD_LAUNCH((ipa_fused_bwd_kernel<32, 4>), dim3((unsigned)nwg), dim3(256), 2 * IB_BUF, st, p);
  }
  return dfold_check_launch();
}
