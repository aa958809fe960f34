// Shared pieces of the bf16 MFMA contraction engine (gemm_bf16.hip, conv_fwd_w4.hip): launch parameters and the epilogues.
#pragma once
#include "dfold_common.h"
typedef __attribute__((ext_vector_type(4))) unsigned gu32x4;
#include "../../include/dfold_hip.h"

#define BM 128
#define BN 128
#define BK 64
#define TILE_BYTES (BM * BK * 2)

struct GemmParams {
  const bf16_t* A;
  const bf16_t* B;
  void* C;
  void* C2;
  const float* bias;
  const bf16_t* R;
  const bf16_t* R2;
  const bf16_t* zeros;
  // element offset of K segment g = (hi, mid, lo), lo = g % seg_div, mid = (g / seg_div) % seg_div_mid, hi = the rest:
  //   seg0 + hi * seg_s0 + mid * seg_s1 + lo * seg_s2
  long a_seg0, a_seg_s0, a_seg_s1, a_seg_s2;
  long b_seg0, b_seg_s0, b_seg_s1, b_seg_s2;
  int seg_div, seg_div_mid;
  RowMap am, cm;
  long ldb;
  long sa0, sa1, sb0, sb1, sc0, sc1;
  int M, N, nseg, seglen, nb1, flags;
  float alpha;
  int prio;              // wave-priority scheme of the 256x320 kernel (DFOLD_GEMM_PRIO, see the kernel)
  int conv_f0, conv_F;   // 5x5 conv: grid frame of logical frame 0 / frames of the grid (conv_F = 0: no tap skipping)
  float* ws;    // split-K partial tiles (nullptr: no split)
  int* cnt;     // split-K arrival counters, one per output tile
  int sk_per, sk_tiles;   // stream-K form of conv_fwd_w4.hip: (tile, K group) units per workgroup, output tiles
  int sk_fence;           // conv_fwd_w4.hip: the partial-tile workspace is ordinary (L2-cached) memory: hand-overs need fences
  const int* nz_ps;       // zero-frame skipping (dfold_gemm_desc.nz_ps; nullptr: none): prefix sums [window][fp + 1]
  int nz_radius, nz_f0;
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// Shared epilogue: 2x2 MFMA 32x32 accumulator tiles of one wave -> C (C/D layout: col = lane&31,
// row = (e&3) + 8*(e>>2) + 4*(lane>>5)).
template <int NJ, int MI = 2>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x16 (&acc)[MI][NJ], long mbase, int nbase, long coff,
                                              int lane) {
  // Straight-line, fully unrolled (accumulators stay in registers): per accumulator row the side loads (residual /
  // mask rows) are issued as one batch of NJ independent loads, bias values are loaded once per lane; stores are
  // predicated instead of branching around the body (dependent load -> wait -> store chains made the epilogue
  // latency-bound: ~110 us per 256x320 tile before).
  const int fl = p.flags;
  const int frow = lane & 31, fhalf = lane >> 5;
  const bool has_r = (fl & (DFOLD_GEMM_RESID | DFOLD_GEMM_RELUMASK)) != 0;
  const bool c2_relu = p.C2 != nullptr && (fl & DFOLD_GEMM_C2RELU) != 0;
  const bool mask2 = (fl & DFOLD_GEMM_MASK2) != 0;
  const bool c2_pre = p.C2 != nullptr && p.R2 == nullptr && !c2_relu;
  const bool c2_mask = p.C2 != nullptr && p.R2 != nullptr && !mask2;
  float bias_v[NJ];
  bool nok[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int n = nbase + j * 32 + frow;
    nok[j] = n < p.N;
    bias_v[j] = ((fl & DFOLD_GEMM_BIAS) && nok[j]) ? p.bias[n] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < MI; ++i) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const long m = mbase + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fhalf;
      const bool mok = m < p.M && row_valid(p.cm, m);
      const long ro = row_off(p.cm, mok ? m : 0) + coff + nbase + frow;
      float rv[NJ], r2v[NJ], cv[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const bool ok = mok && nok[j];
        rv[j] = (has_r && ok) ? bf2f(p.R[ro + j * 32]) : 0.f;
        r2v[j] = ((c2_mask || mask2) && ok) ? bf2f(p.R2[ro + j * 32]) : 0.f;
        cv[j] = ((fl & DFOLD_GEMM_ACCUM) && ok) ? ((const float*)p.C)[ro + j * 32] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const bool ok = mok && nok[j];
        const long off = ro + j * 32;
        float v = acc[i][j][e] * p.alpha + bias_v[j];
        if (fl & DFOLD_GEMM_RELU) v = fmaxf(v, 0.f);
        if (c2_pre && ok) ((bf16_t*)p.C2)[off] = f2bf(v);
        if (mask2) v = r2v[j] > 0.f ? v : 0.f;
        if (fl & DFOLD_GEMM_RESID) v += rv[j];
        if (fl & DFOLD_GEMM_RELUMASK) v = rv[j] > 0.f ? v : 0.f;
        if (c2_relu && ok) ((bf16_t*)p.C2)[off] = f2bf(fmaxf(bf2f(f2bf(v)), 0.f));
        if (fl & DFOLD_GEMM_OUT_BF16) {
          if (ok) ((bf16_t*)p.C)[off] = f2bf(v);
        } else if (fl & DFOLD_GEMM_ATOMIC) {
          if (ok) atomicAdd((float*)p.C + off, v);
        } else {
          if (ok) ((float*)p.C)[off] = v + cv[j];
        }
        if (c2_mask && ok) ((bf16_t*)p.C2)[off] = r2v[j] > 0.f ? f2bf(v) : (bf16_t)0;
      }
    }
  }
}

// bf16 epilogue of the 256x320 kernel staged through LDS: the wave's 64x160 tile is written to LDS (two 32-row
// halves, bias / ReLU applied, rounded to bf16), then streamed out in whole 16-byte chunks -- residual / mask rows are
// read and C / C2 written as dwordx4 per lane (full 320-byte row segments per wave) instead of 2-byte scattered
// accesses.  Requires N % 320 == 0 (no column tail) and 16-byte aligned rows; the pre-residual value is rounded to
// bf16 before the residual add (C2 is exactly that value).
#define EPI_ROWB(NJ) ((NJ) * 64 + 16)  // NJ*64 B of data + 16 B pad per staged row
template <int NJ>
__device__ __forceinline__ void gemm_epilogue_lds_bf16(const GemmParams& p, f32x16 (&acc)[2][NJ], long mbase, int nbase,
                                                       long coff, int lane, char* wave_lds) {
  constexpr int ROWB = EPI_ROWB(NJ);
  constexpr int CPR = NJ * 4;  // 16-byte chunks per row
  const int fl = p.flags;
  const int frow = lane & 31, fhalf = lane >> 5;
  const bool c2_relu = p.C2 != nullptr && (fl & DFOLD_GEMM_C2RELU) != 0;
  const bool mask2 = (fl & DFOLD_GEMM_MASK2) != 0;
  const bool c2_pre = p.C2 != nullptr && p.R2 == nullptr && !c2_relu;
  const bool c2_mask = p.C2 != nullptr && p.R2 != nullptr && !mask2;
  // (every global load of this epilogue is issued in a batch in front of its uses: written as `cond ? load : 0` / inside the
  //  store loop each one became its own branch with an s_waitcnt vmcnt(0) behind it -- 5 + 2 x 10 memory round trips per tile
  //  and wave on the launches with a residual or a ReLU mask, scripts/isa_audit.py)
  float bias_v[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) bias_v[j] = 0.f;
  if (fl & DFOLD_GEMM_BIAS) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) bias_v[j] = p.bias[nbase + j * 32 + frow];
  }
  const bool need_r = (fl & (DFOLD_GEMM_RESID | DFOLD_GEMM_RELUMASK)) != 0;
  long* rowtab = (long*)(wave_lds + 32 * ROWB);  // 32 row offsets (-1 = row past M)
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    if (lane < 32) {
      const long m = mbase + i * 32 + lane;
      rowtab[lane] = (m < p.M && row_valid(p.cm, m)) ? row_off(p.cm, m) + coff + nbase : -1;
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int r = (e & 3) + 8 * (e >> 2) + 4 * fhalf;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        float v = acc[i][j][e] * p.alpha + bias_v[j];
        if (fl & DFOLD_GEMM_RELU) v = fmaxf(v, 0.f);
        *(bf16_t*)(wave_lds + r * ROWB + (j * 32 + frow) * 2) = f2bf(v);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's LDS writes are done (wave-private region)
    __builtin_amdgcn_wave_barrier();
    // chunk c = lane + 64 k of the 32 rows x CPR chunks of 8 bf16: element offsets first, then the residual / mask vectors
    // of all chunks in flight together (rows past M read offset 0 of the operand and are not stored)
    long offs[CPR / 2];
    gu32x4 rr[CPR / 2], r2[CPR / 2];
#pragma unroll
    for (int k = 0; k < CPR / 2; ++k) {
      const int c = lane + 64 * k;
      const int r = c / CPR, c16 = c - r * CPR;
      const long ro = rowtab[r];
      offs[k] = ro < 0 ? -1 : ro + c16 * 8;
    }
    if (need_r) {
#pragma unroll
      for (int k = 0; k < CPR / 2; ++k) rr[k] = *(const gu32x4*)(p.R + (offs[k] < 0 ? 0 : offs[k]));
    }
    if (c2_mask || mask2) {
#pragma unroll
      for (int k = 0; k < CPR / 2; ++k) r2[k] = *(const gu32x4*)(p.R2 + (offs[k] < 0 ? 0 : offs[k]));
    }
#pragma unroll
    for (int k = 0; k < CPR / 2; ++k) {
      const int c = lane + 64 * k;
      const int r = c / CPR, c16 = c - r * CPR;
      gu32x4 val = *(const gu32x4*)(wave_lds + r * ROWB + c16 * 16);
      const long off = offs[k];
      const bool live = off >= 0;
      if (c2_pre && live) *(gu32x4*)((bf16_t*)p.C2 + off) = val;
      if (mask2) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const uint32_t rq = r2[k][q];
          val[q] = (bf_lo(rq) > 0.f ? (val[q] & 0xffffu) : 0u) | (bf_hi(rq) > 0.f ? (val[q] & 0xffff0000u) : 0u);
        }
      }
      if (need_r) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          uint32_t v2 = val[q];
          const uint32_t rq = rr[k][q];
          if (fl & DFOLD_GEMM_RESID) v2 = (uint32_t)f2bf(bf_lo(v2) + bf_lo(rq)) | ((uint32_t)f2bf(bf_hi(v2) + bf_hi(rq)) << 16);
          if (fl & DFOLD_GEMM_RELUMASK) v2 = (bf_lo(rq) > 0.f ? (v2 & 0xffffu) : 0u) | (bf_hi(rq) > 0.f ? (v2 & 0xffff0000u) : 0u);
          val[q] = v2;
        }
      }
      if (live) *(gu32x4*)((bf16_t*)p.C + off) = val;
      if (c2_relu) {
        gu32x4 rl;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          rl[q] = (bf_lo(val[q]) > 0.f ? (val[q] & 0xffffu) : 0u) | (bf_hi(val[q]) > 0.f ? (val[q] & 0xffff0000u) : 0u);
        if (live) *(gu32x4*)((bf16_t*)p.C2 + off) = rl;
      }
      if (c2_mask) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const uint32_t rq = r2[k][q];
          val[q] = (bf_lo(rq) > 0.f ? (val[q] & 0xffffu) : 0u) | (bf_hi(rq) > 0.f ? (val[q] & 0xffff0000u) : 0u);
        }
        if (live) *(gu32x4*)((bf16_t*)p.C2 + off) = val;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// conv_fwd_w4.hip: the one-wave-per-SIMD 512 x 160 form of the 5x5 conv launch (dispatched from dfold_gemm_bf16)
int dfold_conv_w4_launch(const GemmParams& p, int splitk, hipStream_t stream);
int dfold_conv_w4_launch_streamk(const GemmParams& p, int n_wg, hipStream_t stream);
// gemm_k256.hip: dense K = 256 products with a wide bf16 output (A panel in registers, weights streamed through LDS)
int dfold_gemm_k256_launch(const GemmParams& p, hipStream_t stream);
