// Invariant Point Attention core for gfx950 -- the part of the reference's
// InvariantPointAttention.forward (src/model/ipa_pytorch_dynamic.py:396-469) that the reference
// materialises as [F,N,N,H,Pq,3] / [F,H,3,N,N,Pv] broadcast tensors:
//
//   logits[b,f,h,i,j] = S (= scaled q.k, from the MFMA engine) + sqrt(1/3) * bias[b,h,i,j]
//                       - 0.5 * hw[h] * sum_p |q_pts[i,p] - k_pts[j,p]|^2 + inf * (m_i m_j - 1)      (:402-443)
//   P = softmax_j(logits)                                                                             (:444)
//   o_pt[b,f,i,h,p,:] = sum_j P[i,j] v_pts[j,p,:]   (global frame)                                    (:460-469)
//
// and the matching backward.  Point terms stay in fp32 on the VALU (coordinates are O(100 A); the
// difference form |q-k|^2 is kept, never the bf16-hostile |q|^2+|k|^2-2qk expansion).  One wave owns
// one attention row: the row lives in registers, max / sum / dot reductions are wave64 DPP reductions; the
// per-(window,frame,head) key/value point tables are staged once per workgroup in LDS.
#include "dfold_common.h"
#include "../../include/dfold_hip.h"

#define IPA_PQ 8
#define IPA_PV 12
#define KP (IPA_PQ * 3)  // 24 floats per (residue, head)
#define VP (IPA_PV * 3)  // 36
// LDS row strides of the point tables, in floats.  Rows are read as 16-byte vectors (ds_read_b128: 256 B/clk/CU vs
// 128 for 4-byte reads, a quarter of the instructions); 28 = 4*7 and 36 = 4*9 keep every row 16-byte aligned and, 7 and
// 9 being odd, spread any 16 lanes with distinct j mod 16 over all 64 banks (conflict free for lane <-> j access).
#define KPS 28
#define VPS 36
#define MAXT 16          // columns per lane (N <= 1024)
#define ROWS_PER_BLOCK 64   // attention rows per workgroup (16 per wave): amortises the per-block point-table staging

struct IpaDims {
  int B, F, N, H;
};

// q_pts [B,F,N,H,PQ,3], k_pts [B,F,N,H,PQ,3], v_pts [B,F,N,H,PV,3] fp32; S/P [B,F,H,N,N]; bias [B,H,N,N]; mask [B,F,N]
// One workgroup per (window, frame, head): the key point table is staged once (16-byte vectors) for all N rows; a wave
// owns every 4th row, its q point vector arrives through the scalar cache, and the logits / pair bias of the NEXT row are
// requested before the current row is reduced (the row-serial load -> max -> exp -> sum -> store chain was latency bound).
template <int MT>
__global__ __launch_bounds__(256) void ipa_softmax_fwd_kernel(const float* S, const float* __restrict__ bias,
                                                              const float* __restrict__ q_pts,
                                                              const float* __restrict__ k_pts,
                                                              const float* __restrict__ mask, const float* __restrict__ hw,
                                                              float* P, bf16_t* __restrict__ Pb, IpaDims d,
                                                              float bias_scale, float inf) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* kp = sm;  // [N][KPS]
  const int N = d.N, H = d.H;
  const int bf = blockIdx.z, h = blockIdx.y, b = bf / d.F;
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const float* kbase = k_pts + ((long)bf * N * H + h) * KP;
  for (int e = threadIdx.x; e < N * (KP / 4); e += 256) {
    const int j = e / (KP / 4), q = e - j * (KP / 4);
    *(float4*)(kp + j * KPS + 4 * q) = *(const float4*)(kbase + (long)j * H * KP + 4 * q);
  }
  __syncthreads();
  const float hwh = hw[h];
  float mj[MT];                                   // key mask of this lane's columns (row independent)
#pragma unroll
  for (int t = 0; t < MT; ++t) mj[t] = (lane + 64 * t < N) ? mask[(long)bf * N + lane + 64 * t] : 0.f;
  const long baseS = ((long)bf * H + h) * N * N, baseB = ((long)b * H + h) * N * N;
  float sc[MT], bc[MT];                           // logits / pair bias of the row in flight
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int j = lane + 64 * t;
    const bool ok = j < N && w < N;
    sc[t] = ok ? S[baseS + (long)w * N + j] : 0.f;
    bc[t] = ok ? bias[baseB + (long)w * N + j] : 0.f;
  }
  for (int i = w; i < N; i += 4) {
    float qv[KP];
    const float* qb = q_pts + (((long)bf * N + i) * H + h) * KP;      // wave-uniform row: scalar loads
#pragma unroll
    for (int c = 0; c < KP; ++c) qv[c] = qb[c];
    const float mi = mask[(long)bf * N + i];
    const long rowS = baseS + (long)i * N;
    float sn[MT], bn[MT];                         // next row of this wave
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const int j = lane + 64 * t;
      const bool ok = j < N && i + 4 < N;
      sn[t] = ok ? S[rowS + 4L * N + j] : 0.f;
      bn[t] = ok ? bias[baseB + (long)(i + 4) * N + j] : 0.f;
    }
    float lg[MT];
    float mx = -3.0e38f;
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const int j = lane + 64 * t;
      lg[t] = -3.0e38f;
      if (j < N) {
        float d2 = 0.f;
#pragma unroll
        for (int c4 = 0; c4 < KP / 4; ++c4) {
          const float4 kv = *(const float4*)(kp + j * KPS + 4 * c4);
          const float d0 = qv[4 * c4] - kv.x, d1 = qv[4 * c4 + 1] - kv.y, d2_ = qv[4 * c4 + 2] - kv.z, d3 = qv[4 * c4 + 3] - kv.w;
          d2 += d0 * d0 + d1 * d1 + d2_ * d2_ + d3 * d3;
        }
        float v = sc[t] + bias_scale * bc[t] - 0.5f * hwh * d2;
        v += inf * (mi * mj[t] - 1.f);
        lg[t] = v;
        mx = fmaxf(mx, v);
      }
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const int j = lane + 64 * t;
      if (j < N) {
        lg[t] = expf(lg[t] - mx);
        sum += lg[t];
      }
    }
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const int j = lane + 64 * t;
      if (j < N) {
        const float p = lg[t] * inv;
        P[rowS + j] = p;
        Pb[rowS + j] = f2bf(p);
      }
      sc[t] = sn[t];
      bc[t] = bn[t];
    }
  }
}

extern "C" int dfold_ipa_softmax_fwd(const float* S, const float* bias, const float* q_pts, const float* k_pts,
                                     const float* mask, const float* hw, float* P, void* P_bf16, int32_t B, int32_t F,
                                     int32_t N, int32_t H, float bias_scale, float inf, void* stream) {
  if (!S || !bias || !q_pts || !k_pts || !mask || !hw || !P || !P_bf16) return DFOLD_EINVAL;
  if (B <= 0 || F <= 0 || N <= 0 || H <= 0 || N > 64 * MAXT || (long)B * F > 65535) return DFOLD_EINVAL;
  IpaDims d{B, F, N, H};
  dim3 grid(1, H, B * F);
  const size_t lds = (size_t)N * KPS * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  if (N <= 256)
    DFOLD_LAUNCH(ipa_softmax_fwd_kernel<4>, grid, dim3(256), lds, st, S, bias, q_pts, k_pts, mask, hw, P, (bf16_t*)P_bf16, d, bias_scale, inf);
  else if (N <= 512)
    DFOLD_LAUNCH(ipa_softmax_fwd_kernel<8>, grid, dim3(256), lds, st, S, bias, q_pts, k_pts, mask, hw, P, (bf16_t*)P_bf16, d, bias_scale, inf);
  else {
    DFOLD_MAX_LDS_ONCE((ipa_softmax_fwd_kernel<MAXT>), 160 * 1024);
    DFOLD_LAUNCH(ipa_softmax_fwd_kernel<MAXT>, grid, dim3(256), lds, st, S, bias, q_pts, k_pts, mask, hw, P, (bf16_t*)P_bf16, d, bias_scale, inf);
  }
  return dfold_check_launch();
}

// o_pt[b,f,i,h,c] = sum_j P[b,f,h,i,j] * v_pts[b,f,j,h,c]   (c = 36 point components, fp32)
// A workgroup stages `rows` attention rows of P (pitch N+4) next to the value-point table in LDS; a thread owns a
// 2-row x 4-component register tile and walks j four at a time: 6 ds_read_b128 feed 32 FMAs (the one-output-per-thread
// form issued 2 four-byte LDS reads per FMA and ran at the LDS instruction rate).  N % 4 == 0.
#define OPT_THREADS 192
__global__ __launch_bounds__(OPT_THREADS) void ipa_opt_fwd_kernel(const float* __restrict__ P, const float* __restrict__ v_pts,
                                                                  float* __restrict__ o_pt, IpaDims d, int rows_per_block) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int N = d.N, H = d.H, NP = N + 4;
  float* vp = sm;             // [N][VPS]
  float* pt = sm + N * VPS;   // [rows][NP]
  const int bf = blockIdx.z, h = blockIdx.y, i0 = blockIdx.x * rows_per_block;
  const float* vbase = v_pts + ((long)bf * N * H + h) * VP;
  for (int e = threadIdx.x; e < N * (VP / 4); e += OPT_THREADS) {
    const int j = e / (VP / 4), q = e - j * (VP / 4);
    *(float4*)(vp + j * VPS + 4 * q) = *(const float4*)(vbase + (long)j * H * VP + 4 * q);
  }
  const int rows = min(rows_per_block, N - i0);
  const float* pbase = P + (((long)bf * H + h) * N + i0) * N;
  const int n4 = N >> 2;
  for (int e = threadIdx.x; e < rows_per_block * n4; e += OPT_THREADS) {
    const int r = e / n4, q = e - r * n4;
    const float4 v = r < rows ? *(const float4*)(pbase + (long)r * N + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    *(float4*)(pt + r * NP + 4 * q) = v;
  }
  __syncthreads();
  const int rg = threadIdx.x / 9, cq = threadIdx.x - rg * 9;
  if (2 * rg >= rows) return;
  const float* p0 = pt + (2 * rg) * NP;
  const float* p1 = p0 + NP;
  const float* vq = vp + 4 * cq;
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
#pragma unroll 2
  for (int j = 0; j < N; j += 4) {
    const float4 x0 = *(const float4*)(p0 + j), x1 = *(const float4*)(p1 + j);
    const float4 v0 = *(const float4*)(vq + (j + 0) * VPS), v1 = *(const float4*)(vq + (j + 1) * VPS),
                 v2 = *(const float4*)(vq + (j + 2) * VPS), v3 = *(const float4*)(vq + (j + 3) * VPS);
    a0.x += x0.x * v0.x + x0.y * v1.x + x0.z * v2.x + x0.w * v3.x;
    a0.y += x0.x * v0.y + x0.y * v1.y + x0.z * v2.y + x0.w * v3.y;
    a0.z += x0.x * v0.z + x0.y * v1.z + x0.z * v2.z + x0.w * v3.z;
    a0.w += x0.x * v0.w + x0.y * v1.w + x0.z * v2.w + x0.w * v3.w;
    a1.x += x1.x * v0.x + x1.y * v1.x + x1.z * v2.x + x1.w * v3.x;
    a1.y += x1.x * v0.y + x1.y * v1.y + x1.z * v2.y + x1.w * v3.y;
    a1.z += x1.x * v0.z + x1.y * v1.z + x1.z * v2.z + x1.w * v3.z;
    a1.w += x1.x * v0.w + x1.y * v1.w + x1.z * v2.w + x1.w * v3.w;
  }
  const int i = i0 + 2 * rg;
  *(float4*)(o_pt + (((long)bf * N + i) * H + h) * VP + 4 * cq) = a0;
  if (2 * rg + 1 < rows) *(float4*)(o_pt + (((long)bf * N + i + 1) * H + h) * VP + 4 * cq) = a1;
}

// scalar form (any N): one output per thread, P rows and the value-point table in LDS
__global__ __launch_bounds__(256) void ipa_opt_fwd_scalar_kernel(const float* __restrict__ P, const float* __restrict__ v_pts,
                                                                 float* __restrict__ o_pt, IpaDims d, int rows_per_block) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int N = d.N, H = d.H;
  float* vp = sm;             // [N][VPS + 1]
  float* pt = sm + N * (VPS + 1);   // [rows][N]
  const int bf = blockIdx.z, h = blockIdx.y, i0 = blockIdx.x * rows_per_block;
  const float* vbase = v_pts + ((long)bf * N * H + h) * VP;
  for (int e = threadIdx.x; e < N * VP; e += 256) {
    const int j = e / VP, c = e - j * VP;
    vp[j * (VPS + 1) + c] = vbase[(long)j * H * VP + c];
  }
  const int rows = min(rows_per_block, N - i0);
  const float* pbase = P + (((long)bf * H + h) * N + i0) * N;
  for (int e = threadIdx.x; e < rows * N; e += 256) pt[e] = pbase[e];
  __syncthreads();
  for (int o = threadIdx.x; o < rows * VP; o += 256) {
    const int r = o / VP, c = o - r * VP;
    float acc = 0.f;
    for (int j = 0; j < N; ++j) acc += pt[r * N + j] * vp[j * (VPS + 1) + c];
    o_pt[(((long)bf * N + i0 + r) * H + h) * VP + c] = acc;
  }
}

extern "C" int dfold_ipa_opt_fwd(const float* P, const float* v_pts, float* o_pt, int32_t B, int32_t F, int32_t N, int32_t H,
                                 void* stream) {
  if (!P || !v_pts || !o_pt || B <= 0 || F <= 0 || N <= 0 || H <= 0 || (long)B * F > 65535) return DFOLD_EINVAL;
  IpaDims d{B, F, N, H};
  if ((N & 3) || (((uintptr_t)P | (uintptr_t)v_pts | (uintptr_t)o_pt) & 15)) {
    int rows = (int)((76 * 1024 - (long)N * (VPS + 1) * 4) / ((long)N * 4));
    if (rows > 64) rows = 64;
    if (rows < 4) rows = (int)((160 * 1024 - (long)N * (VPS + 1) * 4) / ((long)N * 4));
    if (rows < 1) return DFOLD_EINVAL;
    const size_t lds = ((size_t)N * (VPS + 1) + (size_t)rows * N) * sizeof(float);
    DFOLD_MAX_LDS_ONCE((ipa_opt_fwd_scalar_kernel), 160 * 1024);
    DFOLD_LAUNCH(ipa_opt_fwd_scalar_kernel, dim3((N + rows - 1) / rows, H, B * F), dim3(256), lds, (hipStream_t)stream, P, v_pts,
                 o_pt, d, rows);
    return dfold_check_launch();
  }
  // rows per block: up to 32 (16 row pairs x 9 column quads = 144 of the 192 threads), fewer when the value-point
  // table of a long chain leaves less LDS; two blocks per CU at N <= 256 so that staging overlaps compute
  const long table = (long)N * VPS * 4, row_bytes = (long)(N + 4) * 4;
  long rows = (78 * 1024 - table) / row_bytes;
  if (rows < 8) rows = (160 * 1024 - table) / row_bytes;
  if (rows > 32) rows = 32;
  rows &= ~1L;
  if (rows < 2) return DFOLD_EINVAL;
  const size_t lds = (size_t)(table + rows * row_bytes);
  dim3 grid((unsigned)((N + rows - 1) / rows), H, B * F);
  DFOLD_MAX_LDS_ONCE((ipa_opt_fwd_kernel), 160 * 1024);
  DFOLD_LAUNCH(ipa_opt_fwd_kernel, grid, dim3(OPT_THREADS), lds, (hipStream_t)stream, P, v_pts, o_pt, d, (int)rows);
  return dfold_check_launch();
}

// ---------------------------------------------------------------------------------------------
// Backward, row pass.  Per attention row (b,f,h,i):
//   g_ij  = dP_ij + sum_c do_pt[i,c] v_pts[j,c]
//   dS_ij = P_ij (g_ij - sum_j' P_ij' g_ij')
//   dq_pts[i,c] = -hw sum_j dS_ij (q_ic - k_jc);   dhw[h] += sum_j dS_ij * (-0.5 |q_i - k_j|^2)
// writes dS (fp32, and bf16 for the MFMA products dQ/dK).
// ---------------------------------------------------------------------------------------------
// One workgroup per (window, frame, head): the key / value point tables are staged ONCE (16-byte vectors) and serve all N
// attention rows (8 waves x N/8 rows); the row's own q / do_pt vectors are wave-uniform and come through the scalar cache.
// 64 KB of LDS -> two workgroups (16 waves) per CU.  (With 64-row workgroups re-staging the tables for every row block,
// 81 KB of LDS and one workgroup per CU, the row-serial load -> reduce -> store chain ran at 1.4 TB/s.)
template <int MT>
__global__ __launch_bounds__(512) void ipa_softmax_bwd_kernel(const bf16_t* __restrict__ P, const float* dP,
                                                              const float* __restrict__ q_pts,
                                                              const float* __restrict__ k_pts,
                                                              const float* __restrict__ v_pts,
                                                              const float* __restrict__ do_pt, const float* __restrict__ hw,
                                                              float* dS, bf16_t* __restrict__ dSb,
                                                              float* __restrict__ dq_pts, float* __restrict__ dhw,
                                                              const float* __restrict__ ctr, IpaDims d) {
  // The rows of dS sum to zero (sum_j P_ij (g_ij - sum_j' P_ij' g_ij') = 0), so the q-dependent parts of the point
  // gradients drop out of the per-pair work:
  //   dq_pts[i,c] = -hw sum_j dS_ij (q_ic - k_jc)                      = +hw A_ic,           A_ic = sum_j dS_ij k_jc
  //   dhw[h]     += sum_j dS_ij (-0.5 |q_i - k_j|^2) = -0.5 (sum_j dS_ij |k_j|^2 - 2 q_i . A_i)
  // i.e. 25 FMAs per (i, j) pair instead of 24 differences + 48 FMAs, and 7 instead of 12 LDS vector reads; the q_i . A_i
  // product is once per row.  (fp32: the dropped terms are eps * |q|^2 * sum_j |dS_ij|, five orders below the kept ones.)
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int N = d.N, H = d.H;
  float* kp = sm;                          // [N][KPS]  (column 24 of a row: |k_j|^2)
  float* vp = kp + N * KPS;                // [N][VPS]
  const int bf = blockIdx.z, h = blockIdx.y;
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const float* kbase = k_pts + ((long)bf * N * H + h) * KP;
  const float* vbase = v_pts + ((long)bf * N * H + h) * VP;
  // Key points relative to a per-(window, frame) centre (the gradients below are invariant under a common shift of q and
  // k: they depend on differences, and sum_j dS_ij = 0): |k_j|^2 and q_i . A_i cancel against each other in dhw, and with
  // coordinates of tens of Angstrom the centred form loses an order of magnitude less to that cancellation.
  const float c3[3] = {ctr ? ctr[bf * 3] : 0.f, ctr ? ctr[bf * 3 + 1] : 0.f, ctr ? ctr[bf * 3 + 2] : 0.f};
  for (int e = threadIdx.x; e < N * (KP / 4); e += 512) {
    const int j = e / (KP / 4), q = e - j * (KP / 4);
    float4 v = *(const float4*)(kbase + (long)j * H * KP + 4 * q);
    // component index 4q + {0,1,2,3}; its axis is (4q + r) % 3 = (q + r) % 3
    v.x -= c3[q % 3];
    v.y -= c3[(q + 1) % 3];
    v.z -= c3[(q + 2) % 3];
    v.w -= c3[q % 3];
    *(float4*)(kp + j * KPS + 4 * q) = v;
  }
  for (int e = threadIdx.x; e < N * (VP / 4); e += 512) {
    const int j = e / (VP / 4), q = e - j * (VP / 4);
    *(float4*)(vp + j * VPS + 4 * q) = *(const float4*)(vbase + (long)j * H * VP + 4 * q);
  }
  __syncthreads();
  for (int j = threadIdx.x; j < N; j += 512) {      // |k_j|^2 into the pad column of the key table
    float kn = 0.f;
#pragma unroll
    for (int c = 0; c < KP; ++c) kn += kp[j * KPS + c] * kp[j * KPS + c];
    kp[j * KPS + KP] = kn;
  }
  __syncthreads();
  const float hwh = hw[h];
  float dhw_acc = 0.f;
  const long base = ((long)bf * H + h) * N * N;
  // The row's own q_pts (24) / do_pt (36) vectors are the same for every lane.  They used to come through the scalar unit
  // (s_load -> SGPR operands): the waves sat parked on s_waitcnt for 57 % of their cycles (profiles/r3_ipa_pmc_sq.txt).
  // Now lane c < 60 fetches element c with one coalesced vector load, one row AHEAD, and the wave reads it back from a
  // private 256-byte LDS slot as broadcast float4s.
  float* const rowbuf = sm + (size_t)N * (KPS + VPS) + w * 128;      // two slots of 64 floats per wave
  const float* const qrow0 = q_pts + ((long)bf * N * H + h) * KP;
  const float* const drow0 = do_pt + ((long)bf * N * H + h) * VP;
  auto row_elem = [&](int i) __attribute__((always_inline)) -> float {
    const int ii = i < N ? i : N - 1;
    if (lane < KP) return qrow0[(long)ii * H * KP + lane];
    if (lane < KP + VP) return drow0[(long)ii * H * VP + (lane - KP)];
    return 0.f;
  };
  float rv = row_elem(w);
  int par = 0;
  float pc[MT], dc[MT];                       // P / dP of the row in flight (the next row is requested one row ahead)
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int j = lane + 64 * t;
    const bool ok = j < N && w < N;
    pc[t] = ok ? bf2f(P[base + (long)w * N + j]) : 0.f;
    dc[t] = ok ? dP[base + (long)w * N + j] : 0.f;
  }
  for (int i = w; i < N; i += 8) {
    const long pix = ((long)bf * N + i) * H + h;
    float* const rb = rowbuf + par * 64;       // [q (24) | do_pt (36)] of row i
    rb[lane] = rv;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    rv = row_elem(i + 8);
    par ^= 1;
    const float* qv = rb;
    const float* dov = rb + KP;
    const long row = base + (long)i * N;
    float pn[MT], dn[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const int j = lane + 64 * t;
      const bool ok = j < N && i + 8 < N;
      pn[t] = ok ? bf2f(P[row + 8L * N + j]) : 0.f;
      dn[t] = ok ? dP[row + 8L * N + j] : 0.f;
    }
    float pv[MT], gv[MT];
    float dot = 0.f, psum = 0.f;
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      pv[t] = (lane + 64 * t < N) ? pc[t] : 0.f;
      gv[t] = (lane + 64 * t < N) ? dc[t] : 0.f;
      pc[t] = pn[t];
      dc[t] = dn[t];
    }
#pragma unroll
    for (int c4 = 0; c4 < VP / 4; ++c4) {
      const float4 dv = *(const float4*)(dov + 4 * c4);          // broadcast
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        const int j = lane + 64 * t;
        if (j < N) {
          const float4 vv = *(const float4*)(vp + j * VPS + 4 * c4);
          gv[t] += dv.x * vv.x + dv.y * vv.y + dv.z * vv.z + dv.w * vv.w;
        }
      }
    }
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      dot += pv[t] * gv[t];
      psum += pv[t];
    }
    // The probabilities are the forward's bf16 copy (what its o / o_pair products consumed); their rows sum to 1 only to
    // 2^-9, so the row mean of g is taken with respect to THEM (dot / psum): sum_j dS_ij = 0 then holds to fp32 rounding,
    // which is what the dropped q-dependent terms below rely on (g carries a large common mode, do_pt . v_pts).
    dot = wave_sum(dot) / wave_sum(psum);
    float ak[KP];            // A_ic partial sums of this lane
    float akn = 0.f;         // sum_j dS_ij |k_j|^2
#pragma unroll
    for (int c = 0; c < KP; ++c) ak[c] = 0.f;
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const int j = lane + 64 * t;
      if (j < N) {
        const float ds = pv[t] * (gv[t] - dot);
        dS[row + j] = ds;
        dSb[row + j] = f2bf(ds);
#pragma unroll
        for (int c4 = 0; c4 < KP / 4; ++c4) {
          const float4 kv = *(const float4*)(kp + j * KPS + 4 * c4);
          ak[4 * c4] += ds * kv.x;
          ak[4 * c4 + 1] += ds * kv.y;
          ak[4 * c4 + 2] += ds * kv.z;
          ak[4 * c4 + 3] += ds * kv.w;
        }
        akn += ds * kp[j * KPS + KP];
      }
    }
    float qa = 0.f;          // q_i . A_i, accumulated per component after the wave reduction
#pragma unroll
    for (int c = 0; c < KP; ++c) {
      const float sres = wave_sum(ak[c]);
      qa += (qv[c] - c3[c % 3]) * sres;
      if (lane == 0) dq_pts[pix * KP + c] = hwh * sres;
    }
    akn = wave_sum(akn);
    if (lane == 0) dhw_acc += -0.5f * (akn - 2.f * qa);
  }
  if (lane == 0 && dhw_acc != 0.f) atomicAdd(dhw + h, dhw_acc);
}

extern "C" int dfold_ipa_softmax_bwd(const void* P_bf16, const float* dP, const float* q_pts, const float* k_pts,
                                     const float* v_pts, const float* do_pt, const float* hw, float* dS, void* dS_bf16,
                                     float* dq_pts, float* dhw, const float* ctr, int32_t B, int32_t F, int32_t N, int32_t H,
                                     void* stream) {
  if (!P_bf16 || !dP || !q_pts || !k_pts || !v_pts || !do_pt || !hw || !dS || !dS_bf16 || !dq_pts || !dhw) return DFOLD_EINVAL;
  if (B <= 0 || F <= 0 || N <= 0 || H <= 0 || N > 64 * MAXT || (long)B * F > 65535) return DFOLD_EINVAL;
  IpaDims d{B, F, N, H};
  const bf16_t* P = (const bf16_t*)P_bf16;
  const size_t lds = ((size_t)N * (KPS + VPS) + 8 * 128) * sizeof(float);      // point tables + 8 wave-private row slots
  if (lds > 160 * 1024) return DFOLD_EINVAL;
  dim3 grid(1, H, B * F);
  hipStream_t st = (hipStream_t)stream;
  bf16_t* dsb = (bf16_t*)dS_bf16;
  if (N <= 256) {
    DFOLD_MAX_LDS_ONCE((ipa_softmax_bwd_kernel<4>), 160 * 1024);
    DFOLD_LAUNCH(ipa_softmax_bwd_kernel<4>, grid, dim3(512), lds, st, P, dP, q_pts, k_pts, v_pts, do_pt, hw, dS, dsb, dq_pts, dhw, ctr, d);
  } else if (N <= 512) {
    DFOLD_MAX_LDS_ONCE((ipa_softmax_bwd_kernel<8>), 160 * 1024);
    DFOLD_LAUNCH(ipa_softmax_bwd_kernel<8>, grid, dim3(512), lds, st, P, dP, q_pts, k_pts, v_pts, do_pt, hw, dS, dsb, dq_pts, dhw, ctr, d);
  } else {
    DFOLD_MAX_LDS_ONCE((ipa_softmax_bwd_kernel<MAXT>), 160 * 1024);
    DFOLD_LAUNCH(ipa_softmax_bwd_kernel<MAXT>, grid, dim3(512), lds, st, P, dP, q_pts, k_pts, v_pts, do_pt, hw, dS, dsb, dq_pts, dhw, ctr, d);
  }
  return dfold_check_launch();
}

// ---------------------------------------------------------------------------------------------
// Backward, column pass.  Per key/value residue (b,f,h,j), lane <-> j, loop over rows i:
//   dk_pts[j,c] = hw (sum_i dS_ij q_ic - k_jc sum_i dS_ij);    dv_pts[j,c] = sum_i P_ij do_pt[i,c]
// ---------------------------------------------------------------------------------------------
// lane <-> KPL adjacent key/value residues; the loop runs over query rows i.  The per-row vectors q_pts[i] (24 floats) and
// do_pt[i] (36) are the same for every lane: tiles of 32 rows are staged in LDS (coalesced 16-byte loads, one tile ahead)
// and read back as broadcast float4s.  (Round 2 fetched them through the scalar unit -- s_load into SGPR operands: with 32
// waves per CU each streaming 61 KB the scalar cache thrashed, and rocprofv3 showed the waves parked on s_waitcnt for 77 %
// of their cycles, SQ_WAIT_ANY / SQ_WAVE_CYCLES, profiles/r3_ipa_pmc_sq.txt: ~2500 cycles per row.)
#define CB_TR 32                 // rows per LDS tile
#define CB_RP 64                 // floats per staged row: q (24) | do_pt (36) | pad (4)
template <int KPL, int NTHR>
__global__ __launch_bounds__(NTHR) void ipa_col_bwd_kernel(const bf16_t* __restrict__ P, const float* __restrict__ dS,
                                                          const float* __restrict__ q_pts, const float* __restrict__ k_pts,
                                                          const float* __restrict__ do_pt, const float* __restrict__ hw,
                                                          float* __restrict__ dk_pts, float* __restrict__ dv_pts, IpaDims d) {
  __shared__ __attribute__((aligned(16))) float tile[2][CB_TR][CB_RP];
  const int N = d.N, H = d.H;
  const int bf = blockIdx.z, h = blockIdx.y, tid = threadIdx.x;
  constexpr int nthr = NTHR;
  const float* qbase = q_pts + ((long)bf * N * H + h) * KP;
  const float* dbase = do_pt + ((long)bf * N * H + h) * VP;
  const int j0 = (blockIdx.x * nthr + tid) * KPL;            // first key of this lane (N % KPL == 0)
  const bool ok = j0 < N;
  float aq[KPL][KP], av[KPL][VP], cs[KPL];
#pragma unroll
  for (int e = 0; e < KPL; ++e) {
    cs[e] = 0.f;
#pragma unroll
    for (int c = 0; c < KP; ++c) aq[e][c] = 0.f;
#pragma unroll
    for (int c = 0; c < VP; ++c) av[e][c] = 0.f;
  }
  const long base = ((long)bf * H + h) * N * N + (ok ? j0 : 0);
  // tile staging: CB_TR rows x 15 float4 (6 of q, 9 of do_pt); thread t fetches chunks t, t + nthr, ... of the tile.
  // First-class vector values and UNCONDITIONAL loads (slots past the tile's last chunk re-fetch it and rewrite the same
  // bytes): as `float4 stg[]` under `if (id < NCH)` the staged vectors lived in scratch memory -- every global load was
  // waited for on the spot (scratch_store right behind it), four memory round trips per tile in a row, and came back
  // through scratch_load before the LDS write (visible in hipcc -S; scripts/isa_audit.py flags both).
  constexpr int NCH = CB_TR * 15;
  constexpr int MAXS = (NCH + NTHR - 1) / NTHR;
  f32x4 stg[MAXS];
  auto fetch = [&](int t0) __attribute__((always_inline)) {
#pragma unroll
    for (int s_ = 0; s_ < MAXS; ++s_) {
      int id = tid + s_ * nthr;
      id = id < NCH ? id : NCH - 1;
      const int r = id / 15, c = id - r * 15;
      int i = t0 + r;
      i = i < N ? i : N - 1;
      const float* src = c < 6 ? qbase + (long)i * H * KP + 4 * c : dbase + (long)i * H * VP + 4 * (c - 6);
      stg[s_] = *(const f32x4*)src;
    }
  };
  auto commit = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int s_ = 0; s_ < MAXS; ++s_) {
      int id = tid + s_ * nthr;
      id = id < NCH ? id : NCH - 1;
      const int r = id / 15, c = id - r * 15;
      *(f32x4*)&tile[buf][r][4 * c] = stg[s_];
    }
  };
  // this lane's dS / P values of 8 consecutive rows: independent loads; the NEXT group of 8 is requested before the current
  // one is consumed (two register sets, static indices through the unrolled group loop) -- with one set every group of 8
  // rows waited out a memory round trip in front of its 480 packed FMAs
  auto load8 = [&](int row0, float (&ds)[8][KPL], float (&pp)[8][KPL]) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      int i = row0 + u;
      i = i < N ? i : N - 1;
      if (KPL == 2) {
        const float2 dd = *(const float2*)(dS + base + (long)i * N);
        const uint32_t pr = *(const uint32_t*)(P + base + (long)i * N);
        ds[u][0] = dd.x; ds[u][KPL - 1] = dd.y;
        pp[u][0] = bf_lo(pr); pp[u][KPL - 1] = bf_hi(pr);
      } else {
        ds[u][0] = dS[base + (long)i * N];
        pp[u][0] = bf2f(P[base + (long)i * N]);
      }
    }
  };
  auto fma8 = [&](int row0, const float* rows, const float (&ds)[8][KPL], const float (&pp)[8][KPL]) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (row0 + u < N) {                             // wave-uniform
        const float* row = rows + u * CB_RP;
#pragma unroll
        for (int e = 0; e < KPL; ++e) cs[e] += ds[u][e];
#pragma unroll
        for (int c4 = 0; c4 < KP / 4; ++c4) {
          const float4 qv = *(const float4*)(row + 4 * c4);          // broadcast
#pragma unroll
          for (int e = 0; e < KPL; ++e) {
            aq[e][4 * c4] = __builtin_fmaf(ds[u][e], qv.x, aq[e][4 * c4]);
            aq[e][4 * c4 + 1] = __builtin_fmaf(ds[u][e], qv.y, aq[e][4 * c4 + 1]);
            aq[e][4 * c4 + 2] = __builtin_fmaf(ds[u][e], qv.z, aq[e][4 * c4 + 2]);
            aq[e][4 * c4 + 3] = __builtin_fmaf(ds[u][e], qv.w, aq[e][4 * c4 + 3]);
          }
        }
#pragma unroll
        for (int c4 = 0; c4 < VP / 4; ++c4) {
          const float4 dv = *(const float4*)(row + KP + 4 * c4);
#pragma unroll
          for (int e = 0; e < KPL; ++e) {
            av[e][4 * c4] = __builtin_fmaf(pp[u][e], dv.x, av[e][4 * c4]);
            av[e][4 * c4 + 1] = __builtin_fmaf(pp[u][e], dv.y, av[e][4 * c4 + 1]);
            av[e][4 * c4 + 2] = __builtin_fmaf(pp[u][e], dv.z, av[e][4 * c4 + 2]);
            av[e][4 * c4 + 3] = __builtin_fmaf(pp[u][e], dv.w, av[e][4 * c4 + 3]);
          }
        }
      }
    }
  };
  fetch(0);
  float dsA[8][KPL], pA[8][KPL], dsB[8][KPL], pB[8][KPL];
  load8(0, dsA, pA);
  commit(0);
  __syncthreads();
  const int ntile = (N + CB_TR - 1) / CB_TR;
  static_assert(CB_TR == 32, "the group loop below is written for four groups of 8 rows per tile");
  for (int t = 0; t < ntile; ++t) {
    const int t0 = t * CB_TR, buf = t & 1;
    fetch(t + 1 < ntile ? t0 + CB_TR : t0);          // (unconditional: after the last tile it re-fetches that tile)
    const float* rows = &tile[buf][0][0];
    load8(t0 + 8, dsB, pB);
    fma8(t0, rows, dsA, pA);
    load8(t0 + 16, dsA, pA);
    fma8(t0 + 8, rows + 8 * CB_RP, dsB, pB);
    load8(t0 + 24, dsB, pB);
    fma8(t0 + 16, rows + 16 * CB_RP, dsA, pA);
    load8(t0 + 32, dsA, pA);                         // first group of the next tile (rows past the end: clamped, unused)
    fma8(t0 + 24, rows + 24 * CB_RP, dsB, pB);
    if (t + 1 < ntile) commit(buf ^ 1);
    __syncthreads();
  }
  if (!ok) return;
  const float hwh = hw[h];
#pragma unroll
  for (int e = 0; e < KPL; ++e) {
    const long pix = ((long)bf * N + j0 + e) * H + h;
#pragma unroll
    for (int c = 0; c < KP; ++c) dk_pts[pix * KP + c] = hwh * (aq[e][c] - cs[e] * k_pts[pix * KP + c]);
#pragma unroll
    for (int c = 0; c < VP; ++c) dv_pts[pix * VP + c] = av[e][c];
  }
}

extern "C" int dfold_ipa_col_bwd(const void* P_bf16, const float* dS, const float* q_pts, const float* k_pts, const float* do_pt,
                                 const float* hw, float* dk_pts, float* dv_pts, int32_t B, int32_t F, int32_t N, int32_t H,
                                 void* stream) {
  if (!P_bf16 || !dS || !q_pts || !k_pts || !do_pt || !hw || !dk_pts || !dv_pts) return DFOLD_EINVAL;
  if (B <= 0 || F <= 0 || N <= 0 || H <= 0 || (long)B * F > 65535) return DFOLD_EINVAL;
  IpaDims d{B, F, N, H};
  const bf16_t* P = (const bf16_t*)P_bf16;
  if ((N & 1) == 0 && (((uintptr_t)P_bf16 & 3) | ((uintptr_t)dS & 7)) == 0) {       // two keys per lane: 8- / 4-byte loads
    dim3 grid((N / 2 + 127) / 128, H, B * F);
    DFOLD_LAUNCH((ipa_col_bwd_kernel<2, 128>), grid, dim3(128), 0, (hipStream_t)stream, P, dS, q_pts, k_pts, do_pt, hw, dk_pts, dv_pts, d);
  } else {
    dim3 grid((N + 255) / 256, H, B * F);
    DFOLD_LAUNCH((ipa_col_bwd_kernel<1, 256>), grid, dim3(256), 0, (hipStream_t)stream, P, dS, q_pts, k_pts, do_pt, hw, dk_pts, dv_pts, d);
  }
  return dfold_check_launch();
}

// dbias[b,h,i,j] = scale * sum_f dS[b,f,h,i,j]; written as bf16 in two layouts:
//   out_hn [B][H][N*N]  (K-contiguous over (i,j): A operand of dW_b)  and  out_nh [B][N*N] rows of 8 (H padded to 8, for dz)
//   with a row pitch of nh_pitch elements (a multiple of 8: the rows may be the tail columns of a wider operand matrix)
__global__ __launch_bounds__(256) void ipa_bias_grad_kernel(const float* __restrict__ dS, bf16_t* __restrict__ out_hn,
                                                            bf16_t* __restrict__ out_nh, IpaDims d, float scale, long nh_pitch) {
  const long NN = (long)d.N * d.N;
  const long ij = (long)blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (ij >= NN) return;
  __attribute__((aligned(16))) bf16_t row[8];
#pragma unroll
  for (int h = 0; h < 8; ++h) row[h] = 0;
  for (int h = 0; h < d.H; ++h) {
    float s = 0.f;
    for (int f = 0; f < d.F; ++f) s += dS[(((long)(b * d.F + f)) * d.H + h) * NN + ij];
    const bf16_t v = f2bf(s * scale);
    out_hn[((long)b * d.H + h) * NN + ij] = v;
    row[h] = v;
  }
  *(uint4*)(out_nh + ((long)b * NN + ij) * nh_pitch) = *(const uint4*)row;
}

extern "C" int dfold_ipa_bias_grad(const float* dS, void* out_hn, void* out_nh, int64_t nh_pitch, int32_t B, int32_t F, int32_t N,
                                   int32_t H, float scale, void* stream) {
  if (!dS || !out_hn || !out_nh || B <= 0 || F <= 0 || N <= 0 || H <= 0 || H > 8 || nh_pitch < 8 || (nh_pitch & 7)) return DFOLD_EINVAL;
  IpaDims d{B, F, N, H};
  dim3 grid((unsigned)(((long)N * N + 255) / 256), B);
  DFOLD_LAUNCH(ipa_bias_grad_kernel, grid, dim3(256), 0, (hipStream_t)stream, dS, (bf16_t*)out_hn, (bf16_t*)out_nh, d,
                     scale, (long)nh_pitch);
  return dfold_check_launch();
}
