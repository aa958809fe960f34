// "TN" bf16 MFMA GEMM for gfx950 (MI355X): both operands stored with the REDUCTION index as the slow axis.
//
//   C[m, n] (+)= alpha * sum_k A[k, m] * B[k, n]          A bf16 [K][lda], B bf16 [K][ldb]  (rows = k, channels contiguous)
//
// This is the shape of every weight gradient of a dense layer (dW = dY^T X: reference torch.nn.Linear autograd, used all
// over src/model/ipa_pytorch_dynamic.py) and of the "transposed" attention products of the IPA backward (dK = dS^T Q,
// dV = P^T dO, :396-469).  The NT engine of gemm_bf16.hip needs both operands K-contiguous and therefore runs these on
// transposed COPIES (dfold_transpose_bf16, two per product); this kernel reads the operands as they lie: K tiles of
// 64 rows x 256 channels go HBM -> LDS by LDS-DMA and the MFMA fragments are gathered with ds_read_b64_tr_b16 (the LDS
// transpose read; layout, XOR keys and the hand-counted waits exactly as in conv_wgrad_tn.hip, whose comments explain them).
//
// 256 x 256 x 64 tile, 8 waves as 4 (M) x 2 (N), 2 x 4 v_mfma_f32_32x32x16_bf16 tiles per wave (rows (wm + 4i) * 32,
// columns (wn + 2j) * 32), two LDS stages of 64 KiB, the second wave of every SIMD half a K step behind the first.
// Split-K (a long reduction with a small output: the K range is cut over blockIdx.z, partial tiles meet in fp32 atomics)
// and a two-level batch.  K (per split) a multiple of 64; M, N multiples of 8: a tile that hangs over the edge of the output
// re-reads chunk 0 of its row for the columns past M (N) -- finite values whose products land in accumulator columns that are
// never stored -- so the 128-, 192-, 480-, 640-wide weight gradients run here too instead of on transposed copies.
#include "dfold_common.h"
#include "../../include/dfold_hip.h"

#define GBK 64
#define GBT 256
#define G_PITCH (GBT * 2)
#define G_TILE (GBK * G_PITCH)     // 32 KiB per operand tile
#define G_STAGE (2 * G_TILE)
#define GNJ 4

typedef __attribute__((address_space(3))) void* tg_lds_ptr_t;
typedef __attribute__((address_space(3))) char tg_lchar;
typedef __attribute__((ext_vector_type(4))) short tg_s16x4;
typedef __attribute__((ext_vector_type(8))) short tg_s16x8;

struct TnGemmParams {
  const char* A;
  const char* B;
  void* C;
  long lda, ldb, ldc;                 // elements
  long sa0, sa1, sb0, sb1, sc0, sc1;  // batch strides (elements): z = (z0, z1), z1 = z % nb1
  long ksplit;                        // rows of the reduction per blockIdx.z
  int tiles_n, nb1, nsteps, flags;
  int M, N;
  float alpha;
  const int* rz_ps;                   // row-block flags of A (prefix sums over blocks of rz_block reduction rows) or nullptr
  int rz_block;
};

template <int OFF>
__device__ __forceinline__ bf16x8 tg_frag(unsigned base) {
  tg_s16x4 lo, hi;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(base), "n"(OFF));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(base), "n"(OFF + 4 * G_PITCH));
  const tg_s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return *(const bf16x8*)&v;
}
// the 6 fragments (2 row tiles of A, 4 column tiles of B) of K16 block KB; column tiles j and j + 2 share a key (one base each)
template <int KB>
__device__ __forceinline__ void tg_ldfrag(bf16x8 (&af)[2], bf16x8 (&bfr)[GNJ], unsigned ba, unsigned bb0, unsigned bb1) {
  af[0] = tg_frag<KB * 16 * G_PITCH>(ba);
  af[1] = tg_frag<KB * 16 * G_PITCH + 256>(ba);
  bfr[0] = tg_frag<G_TILE + KB * 16 * G_PITCH>(bb0);
  bfr[1] = tg_frag<G_TILE + KB * 16 * G_PITCH>(bb1);
  bfr[2] = tg_frag<G_TILE + KB * 16 * G_PITCH + 256>(bb0);
  bfr[3] = tg_frag<G_TILE + KB * 16 * G_PITCH + 256>(bb1);
}
template <int N>
__device__ __forceinline__ void tg_pin() {   // (2 MFMA, 1 DMA) pairs inside a block of 8 MFMAs
#pragma unroll
  for (int k = 0; k < N; ++k) {
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
    __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
  }
  if (2 * GNJ - 2 * N > 0) __builtin_amdgcn_sched_group_barrier(0x008, 2 * GNJ - 2 * N, 0);
}
// LDS reads return in order: "at most N outstanding" = everything but the youngest N has landed (12 reads per K16 block)
#define TG_WAIT(N, set)                                                                                         \
  asm volatile("s_waitcnt lgkmcnt(" #N ")"                                                                      \
               : "+v"(af[set][0]), "+v"(af[set][1]), "+v"(bfr[set][0]), "+v"(bfr[set][1]), "+v"(bfr[set][2]),   \
                 "+v"(bfr[set][3]))

// RZ (round 6): row-block flags for A.  A split-K part walks only the blocks of rz_block reduction rows in which A can hold a
// non-zero (one 64-bit mask per part, built with a ballot, walked on the scalar unit: lowest set bit = next block); a part
// without any adds nothing and exits.  The rows left out contribute exact zeros.
template <bool RZ>
__global__ __launch_bounds__(512, 2) void dfold_tn_gemm_kernel(const TnGemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char tl[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 1, wn = w & 1;
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q = nwg >> 3, r8 = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  const int lid = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + idx;
  const int m0 = (lid / p.tiles_n) * GBT, n0 = (lid % p.tiles_n) * GBT;
  const int z = blockIdx.y, z0 = z / p.nb1, z1 = z - z0 * p.nb1;
  const long k0 = (long)blockIdx.z * p.ksplit;
  const long pitchA = p.lda * 2, pitchB = p.ldb * 2;
  unsigned long rz_mask = 0;           // RZ: live blocks of this part still ahead of the K cursor
  int rz_in = 0, rz_spb = 1, rz_steps = 0;
  if (RZ) {
    const int b0 = (int)(k0 / p.rz_block), nblk = (int)(p.ksplit / p.rz_block);     // (host: ksplit a multiple of rz_block, <= 64 blocks)
    const bool lv = lane < nblk && p.rz_ps[b0 + lane + 1] - p.rz_ps[b0 + lane] > 0;
    rz_mask = __ballot(lv);
    if (rz_mask == 0) return;
    rz_spb = p.rz_block / GBK;
    rz_steps = __builtin_popcountl(rz_mask) * rz_spb;
  }
  const char* pa = p.A + (z0 * p.sa0 + z1 * p.sa1 + m0) * 2 + k0 * pitchA;
  const char* pb = p.B + (z0 * p.sb0 + z1 * p.sb1 + n0) * 2 + k0 * pitchB;

  // staging: piece t*8 + w of a tile holds rows 2 (t*8 + w), +1 (32 chunks each); LDS (row, pc) <- logical chunk pc ^ key(row)
  unsigned aoff0, boff0;
  {
    const int row = 2 * w + (lane >> 5);
    const int lc = (lane & 31) ^ ((row & 3) << 2);
    // columns past the edge of the output: chunk 0 of the tile instead (in range: m0 < M, n0 < N)
    aoff0 = (unsigned)(row * pitchA + (m0 + lc * 8 < p.M ? lc : 0) * 16);
    boff0 = (unsigned)(row * pitchB + (n0 + lc * 8 < p.N ? lc : 0) * 16);
  }
  const long dA = GBK * pitchA, dB = GBK * pitchB;
  int st_left = RZ ? rz_steps : p.nsteps;
  const char *pa0 = pa, *pb0 = pb;
  if (RZ) {
    const int cb = __builtin_ctzl(rz_mask);
    rz_mask &= rz_mask - 1;
    pa = pa0 + (long)cb * rz_spb * dA;
    pb = pb0 + (long)cb * rz_spb * dB;
  }
  auto stage_a = [&](int buf) {
    char* la = tl + buf * G_STAGE;
#pragma unroll
    for (int t = 0; t < 4; ++t)
      __builtin_amdgcn_global_load_lds((const void*)(pa + t * 16 * pitchA + aoff0), (tg_lds_ptr_t)(la + (t * 8 + w) * 1024), 16, 0, 0);
  };
  auto stage_b = [&](int buf) {          // second half of a tile's staging: also moves the K cursor on
    char* lb = tl + buf * G_STAGE + G_TILE;
#pragma unroll
    for (int t = 0; t < 4; ++t)
      __builtin_amdgcn_global_load_lds((const void*)(pb + t * 16 * pitchB + boff0), (tg_lds_ptr_t)(lb + (t * 8 + w) * 1024), 16, 0, 0);
    if (RZ) {
      if (st_left > 1) {               // a tile after this one exists (else: re-read it, harmlessly)
        --st_left;
        if (++rz_in < rz_spb) {
          pa += dA;
          pb += dB;
        } else {
          rz_in = 0;
          const int nb = __builtin_ctzl(rz_mask);
          rz_mask &= rz_mask - 1;
          pa = pa0 + (long)nb * rz_spb * dA;
          pb = pb0 + (long)nb * rz_spb * dB;
        }
      }
    } else {
    const unsigned adv = (unsigned)(1 - st_left) >> 31;        // a tile after this one exists (else: re-read it, harmlessly)
    st_left -= (int)adv;
    pa += dA & -(long)adv;
    pb += dB & -(long)adv;
    }
  };

  f32x16 acc[2][GNJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < GNJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // fragment addresses (stage 0): row tile i = chunks wm*4 + 16 i + c0; column tile j = chunks wn*4 + 8 j + c0: bit 3 of the
  // chunk index (j & 1) meets the XOR key, so even and odd j get a base each, j >> 1 is an immediate of 256 bytes
  const int p16 = lane & 15, g = lane >> 4;
  const int row_l = (g >> 1) * 8 + (p16 >> 2);
  const int c0 = (g & 1) * 2 + ((p16 >> 1) & 1);
  const int key = (p16 >> 2) << 2;
  const unsigned tls = (unsigned)(uintptr_t)(tg_lchar*)tl;
  const unsigned fa0 = tls + row_l * G_PITCH + (((wm * 4 + c0) ^ key) << 4) + (p16 & 1) * 8;
  const unsigned fb00 = tls + row_l * G_PITCH + (((wn * 4 + c0) ^ key) << 4) + (p16 & 1) * 8;
  const unsigned fb10 = tls + row_l * G_PITCH + (((wn * 4 + 8 + c0) ^ key) << 4) + (p16 & 1) * 8;
  bf16x8 af[2][2], bfr[2][GNJ];
  auto mma = [&](int set) {
#pragma unroll
    for (int j = 0; j < GNJ; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[set][i], bfr[set][j], acc[i][j], 0, 0, 0);
  };

  stage_a(0);
  stage_b(0);
  const int nsteps = RZ ? rz_steps : p.nsteps;
  if (w >= 4) __builtin_amdgcn_s_setprio(1);
  if (w < 4) {
    for (int s = 0; s < nsteps; ++s) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const unsigned so = (s & 1) * G_STAGE;
      const unsigned fa = fa0 + so, fb0 = fb00 + so, fb1 = fb10 + so;
      tg_ldfrag<0>(af[0], bfr[0], fa, fb0, fb1);
      tg_ldfrag<1>(af[1], bfr[1], fa, fb0, fb1);
      TG_WAIT(12, 0);
      stage_a((s + 1) & 1);
      mma(0);
      tg_pin<4>();
      tg_ldfrag<2>(af[0], bfr[0], fa, fb0, fb1);
      TG_WAIT(12, 1);
      stage_b((s + 1) & 1);
      mma(1);
      tg_pin<4>();
      tg_ldfrag<3>(af[1], bfr[1], fa, fb0, fb1);
      TG_WAIT(12, 0);
      mma(0);
      TG_WAIT(0, 1);
      mma(1);
    }
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    tg_ldfrag<0>(af[0], bfr[0], fa0, fb00, fb10);
    tg_ldfrag<1>(af[1], bfr[1], fa0, fb00, fb10);
    TG_WAIT(12, 0);
    stage_a(1);
    mma(0);
    tg_pin<4>();
    tg_ldfrag<2>(af[0], bfr[0], fa0, fb00, fb10);
    TG_WAIT(12, 1);
    stage_b(1);
    mma(1);
    tg_pin<4>();
    tg_ldfrag<3>(af[1], bfr[1], fa0, fb00, fb10);
    for (int s = 1; s < nsteps; ++s) {
      // the fragment reads of the previous tile still in flight (tg_ldfrag<3>) must have left the LDS before this barrier:
      // behind it the leading waves' LDS-DMA overwrites exactly that stage (ordered by a wait, not by latency)
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const unsigned so = (s & 1) * G_STAGE;
      const unsigned fa = fa0 + so, fb0 = fb00 + so, fb1 = fb10 + so;
      TG_WAIT(12, 0);
      stage_a((s + 1) & 1);
      mma(0);                 // previous tile, K16 block 2
      tg_pin<4>();
      TG_WAIT(0, 1);
      stage_b((s + 1) & 1);
      mma(1);                 // previous tile, K16 block 3
      tg_pin<4>();
      tg_ldfrag<0>(af[0], bfr[0], fa, fb0, fb1);
      tg_ldfrag<1>(af[1], bfr[1], fa, fb0, fb1);
      TG_WAIT(12, 0);
      mma(0);
      tg_ldfrag<2>(af[0], bfr[0], fa, fb0, fb1);
      TG_WAIT(12, 1);
      mma(1);
      tg_ldfrag<3>(af[1], bfr[1], fa, fb0, fb1);
    }
    TG_WAIT(12, 0);
    mma(0);
    TG_WAIT(0, 1);
    mma(1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the surplus prefetch must have landed before the LDS is reused / released

  const int frow = lane & 31, fhalf = lane >> 5;
  const long cbase = z0 * p.sc0 + z1 * p.sc1;
  if (p.flags & DFOLD_GEMM_OUT_BF16) {
    // bf16 output through LDS: the wave's 32 x 128 block of row tile i (4 column tiles of 32) is staged, then written as
    // 16-byte chunks (64-byte runs per column tile)
    __syncthreads();   // every wave is done with the operand stages
    char* stg = tl + w * (32 * 272);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int r = (e & 3) + 8 * (e >> 2) + 4 * fhalf;
#pragma unroll
        for (int j = 0; j < GNJ; ++j) *(bf16_t*)(stg + r * 272 + (j * 32 + frow) * 2) = f2bf(acc[i][j][e] * p.alpha);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int c = lane + 64 * k;          // 32 rows x 16 chunks
        const int r = c >> 4, ch = c & 15, j = ch >> 2, qq = ch & 3;
        const uint4 v = *(const uint4*)(stg + r * 272 + ch * 16);
        const long m = (long)m0 + (wm + 4 * i) * 32 + r;
        const long n = (long)n0 + (wn + 2 * j) * 32 + qq * 8;
        if (m < p.M && n < p.N) *(uint4*)((bf16_t*)p.C + cbase + m * p.ldc + n) = v;
      }
      __builtin_amdgcn_wave_barrier();
    }
    return;
  }
  float* cb = (float*)p.C + cbase + n0 + wn * 32 + frow;
  const bool atomic = (p.flags & DFOLD_GEMM_ATOMIC) != 0, accum = (p.flags & DFOLD_GEMM_ACCUM) != 0;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const long m = (long)m0 + (wm + 4 * i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * fhalf;
      float* row = cb + m * p.ldc;
      const int ncol = n0 + wn * 32 + frow;              // column of tile j: ncol + 64 j
      if (m >= p.M) continue;
      if (atomic) {
#pragma unroll
        for (int j = 0; j < GNJ; ++j)
          if (ncol + j * 64 < p.N) atomicAdd(row + j * 64, acc[i][j][e] * p.alpha);
      } else {
        float cv[GNJ];
#pragma unroll
        for (int j = 0; j < GNJ; ++j) cv[j] = (accum && ncol + j * 64 < p.N) ? row[j * 64] : 0.f;
#pragma unroll
        for (int j = 0; j < GNJ; ++j)
          if (ncol + j * 64 < p.N) row[j * 64] = acc[i][j][e] * p.alpha + cv[j];
      }
    }
  }
}

static int tn_gemm_launch(const void* A, const void* B, void* C, int32_t M, int32_t N, int64_t K, int64_t lda,
                          int64_t ldb, int64_t ldc, int32_t nbatch, int32_t nb1, int64_t sa0, int64_t sa1,
                          int64_t sb0, int64_t sb1, int64_t sc0, int64_t sc1, int32_t splitk, int32_t flags,
                          float alpha, const int32_t* rz_ps, int32_t rz_block, void* stream) {
  if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || nbatch <= 0 || splitk <= 0) return DFOLD_EINVAL;
  if ((M & 7) || (N & 7) || (K % ((long)splitk * GBK))) return DFOLD_EINVAL;
  if (lda < M || ldb < N || ldc < N || (lda & 7) || (ldb & 7) || ((sa0 | sa1 | sb0 | sb1) & 7)) return DFOLD_EINVAL;
  if (((uintptr_t)A | (uintptr_t)B) & 15) return DFOLD_EINVAL;
  if (flags & ~(DFOLD_GEMM_OUT_BF16 | DFOLD_GEMM_ATOMIC | DFOLD_GEMM_ACCUM)) return DFOLD_EINVAL;
  if ((flags & DFOLD_GEMM_OUT_BF16) && ((flags & (DFOLD_GEMM_ATOMIC | DFOLD_GEMM_ACCUM)) || splitk > 1 || (ldc & 7) || ((sc0 | sc1) & 7) ||
                                        ((uintptr_t)C & 15)))
    return DFOLD_EINVAL;
  if (splitk > 1 && !(flags & DFOLD_GEMM_ATOMIC)) return DFOLD_EINVAL;
  if (rz_ps && (rz_block <= 0 || (rz_block % GBK) || nbatch != 1 || !(flags & (DFOLD_GEMM_ATOMIC | DFOLD_GEMM_ACCUM)) ||
                ((K / splitk) % rz_block) || (K / splitk) / rz_block > 64))
    return DFOLD_EINVAL;
  if ((long)(GBK + 2) * lda * 2 >= (1L << 31) || (long)(GBK + 2) * ldb * 2 >= (1L << 31)) return DFOLD_EINVAL;   // 32-bit lane offsets
  TnGemmParams p;
  p.A = (const char*)A; p.B = (const char*)B; p.C = C;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.sa0 = sa0; p.sa1 = sa1; p.sb0 = sb0; p.sb1 = sb1; p.sc0 = sc0; p.sc1 = sc1;
  p.ksplit = K / splitk;
  p.tiles_n = (N + GBT - 1) / GBT; p.nb1 = nb1 > 0 ? nb1 : 1; p.nsteps = (int)(p.ksplit / GBK); p.flags = flags; p.alpha = alpha;
  p.M = M; p.N = N;
  p.rz_ps = rz_ps; p.rz_block = rz_block;
  dim3 grid((unsigned)(((M + GBT - 1) / GBT) * p.tiles_n), (unsigned)nbatch, (unsigned)splitk);
  if (rz_ps) {
    DFOLD_MAX_LDS_ONCE(dfold_tn_gemm_kernel<true>, 2 * G_STAGE);
    DFOLD_LAUNCH(dfold_tn_gemm_kernel<true>, grid, dim3(512), (size_t)(2 * G_STAGE), (hipStream_t)stream, p);
  } else {
    DFOLD_MAX_LDS_ONCE(dfold_tn_gemm_kernel<false>, 2 * G_STAGE);
    DFOLD_LAUNCH(dfold_tn_gemm_kernel<false>, grid, dim3(512), (size_t)(2 * G_STAGE), (hipStream_t)stream, p);
  }
  return dfold_check_launch();
}

extern "C" int dfold_gemm_tn_bf16(const void* A, const void* B, void* C, int32_t M, int32_t N, int64_t K, int64_t lda,
                                  int64_t ldb, int64_t ldc, int32_t nbatch, int32_t nb1, int64_t sa0, int64_t sa1,
                                  int64_t sb0, int64_t sb1, int64_t sc0, int64_t sc1, int32_t splitk, int32_t flags,
                                  float alpha, void* stream) {
  return tn_gemm_launch(A, B, C, M, N, K, lda, ldb, ldc, nbatch, nb1, sa0, sa1, sb0, sb1, sc0, sc1, splitk, flags, alpha, nullptr, 0,
                        stream);
}

extern "C" int dfold_gemm_tn_bf16_rowflags(const void* A, const void* B, void* C, int32_t M, int32_t N, int64_t K, int64_t lda,
                                           int64_t ldb, int64_t ldc, int32_t nbatch, int32_t nb1, int64_t sa0, int64_t sa1,
                                           int64_t sb0, int64_t sb1, int64_t sc0, int64_t sc1, int32_t splitk, int32_t flags,
                                           float alpha, const int32_t* ps, int32_t block, void* stream) {
  if (!ps) return DFOLD_EINVAL;
  return tn_gemm_launch(A, B, C, M, N, K, lda, ldb, ldc, nbatch, nb1, sa0, sa1, sb0, sb1, sc0, sc1, splitk, flags, alpha, ps, block,
                        stream);
}
