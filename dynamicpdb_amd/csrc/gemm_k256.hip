// C[M][N] = A[M][256] B[N][256]^T (+ bias), bf16 in / bf16 or fp32 out: the K = 256 projections of the trunk (linear_q / linear_kv /
// linear_q_points / linear_kv_points of InvariantPointAttention, src/model/ipa_pytorch_dynamic.py:350-396, from the 256-wide node
// features: N = 4096 / 3072 / 2048 at M = windows x frames x residues = 65536) -- round 6.
//
// On the tile engine (gemm_bf16.hip, 256 x 256 tiles) these launches ran at 0.38 - 0.41 PFLOP/s: four K steps per tile, so a
// workgroup is all prologue and epilogue, one workgroup per CU, and the output (537 MB at N = 4096) leaves at 1.6 TB/s.  Here the
// A PANEL of a wave is loaded once and stays in REGISTERS as MFMA operand fragments for the whole row of the output (32 rows x
// 256 = 64 registers: the register-resident scheme of triatt_reg.hip), the weights stream through LDS in chunks of 64 output
// channels as ready-made fragments (requested two chunks ahead into registers, written to the other buffer a chunk ahead),
// and every chunk's 32 x 64 result of a wave leaves through a wave-private LDS tile as whole 128-byte row segments.
//   workgroup = 8 waves x 32 rows; per chunk and wave 64 MFMAs (16x16x32) against 32 fragment reads: matrix pipe and LDS are
//   both at their rate, the launch is bound by the output stream.
#include "gemm_engine.h"
#include <type_traits>

typedef __attribute__((ext_vector_type(4))) unsigned k2u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned k2u32x2;
#define K2_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#define K2_WBUF 32768                      // 32 fragments of 1 KB: (channel tile ct, k step ks) -> fragment ct * 8 + ks
#define K2_SPITCH(F32) ((F32) ? 272 : 144)    // staged row: 64 bf16 | fp32 + 16 bytes
#define K2_STAGE(F32) (32 * K2_SPITCH(F32))
#define K2_LDS(F32) (2 * K2_WBUF + 8 * K2_STAGE(F32))

// RAGGED: the (single) row block that straddles M, launched on its own -- predicated stores hide their number from the compiler's
// wait-count pass, which then waits for them wherever it waits for a weight fragment request
// KS = K steps of 32: 8 (K = 256 exactly) or 2 (8 <= K <= 64, K % 8 == 0: the groups of 8 past K are loaded as zeros -- the pair-side
// product dz = [dpz | dbias] [W_dz | W_b]^T of the IPA backward, K = 40, ran at 28 TFLOP/s on the tile engine)
template <bool BIAS, bool RAGGED, bool F32, int KS, bool RMASK = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_k256_kernel(const GemmParams p, const int block0) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, l4 = lane >> 4;
  const int M = p.M, N = p.N, NCF = N >> 6, NC = NCF + ((N >> 5) & 1);      // chunks of 64 output channels (+ one of 32)
  const long m0 = (long)(block0 + (int)blockIdx.x) * 256 + w * 32;
  const bf16_t* const A = p.A + p.am.base;
  char* const C = (char*)p.C + p.cm.base * (F32 ? 4 : 2);
  // ---- the wave's A panel: xa[t][ks], lane (l15, l4) = row m0 + 16 t + l15, k = 32 ks + 8 l4 .. + 8 (rows past M: the last row,
  //      computed and never stored) ----
  if (RMASK && p.nz_ps != nullptr) {
    // row-block flags of A (dfold_row_block_flags: prefix sums over blocks of p.nz_f0 rows): a workgroup whose 256 rows of A are all
    // zero stores the zeros its masked product would be and leaves (the angle head's backward: 31 of 32 blocks, functional.AngleResnetFn)
    const long mb = (long)(block0 + (int)blockIdx.x) * 256;
    const long ml = (mb + 256 < M ? mb + 256 : M) - 1;
    if (p.nz_ps[ml / p.nz_f0 + 1] - p.nz_ps[mb / p.nz_f0] <= 0) {
      const int n16 = N >> 3;                                   // 16-byte pieces per row
      for (long id = tid; id < (ml - mb + 1) * n16; id += 512) {
        const long row = id / n16, pc = id - row * n16;
        *(k2u32x4*)(C + ((mb + row) * p.cm.ld) * 2 + pc * 16) = (k2u32x4){0u, 0u, 0u, 0u};
      }
      return;
    }
  }
  const int K = p.seglen;
  const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  bf16x8 xa[2][KS];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    long m = m0 + t * 16 + l15;
    if (RAGGED) m = m < M ? m : M - 1;
    const bf16_t* src = A + m * p.am.ld + l4 * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) xa[t][ks] = (KS == 8 || ks * 32 + l4 * 8 < K) ? *(const bf16x8*)(src + ks * 32) : zero8;
  }
  // ---- weights: fragment f = ct * KS + ks of a chunk holds, in lane (l15, l4), B[n0 + 16 ct + l15][32 ks + 8 l4 .. + 8]; wave w
  //      moves fragments FPW w .. FPW w + FPW - 1 (KS = 8: four, ct = w >> 1, ks = 4 (w & 1) + j; KS = 2: one, ct = w >> 1, ks = w & 1) ----
  constexpr int FPW = KS / 2;
  const int ks0 = (w * FPW) % KS;
  const int wrow = (w >> 1) * 16 + l15;         // row of the chunk this lane fetches (rows past N -- the tail chunk -- repeat row N - 1)
  auto wsrc_of = [&](int c_) { return p.B + (long)min(c_ * 64 + wrow, N - 1) * p.ldb + ks0 * 32 + l4 * 8; };
  bool wok[FPW];
#pragma unroll
  for (int j = 0; j < FPW; ++j) wok[j] = KS == 8 || (ks0 + j) * 32 + l4 * 8 < K;
  bf16x8 wst[FPW];
#pragma unroll
  for (int j = 0; j < FPW; ++j) wst[j] = wok[j] ? *(const bf16x8*)(wsrc_of(0) + j * 32) : zero8;
#pragma unroll
  for (int j = 0; j < FPW; ++j) *(bf16x8*)(smem + (w * FPW + j) * 1024 + lane * 16) = wst[j];
  if (NC > 1) {
#pragma unroll
    for (int j = 0; j < FPW; ++j) wst[j] = wok[j] ? *(const bf16x8*)(wsrc_of(1) + j * 32) : zero8;
  }
  __syncthreads();
  __builtin_amdgcn_s_waitcnt(0x0F70);       // vmcnt(0): the A panel is in (said here, the loop below would otherwise wait for it mid-chunk)
  constexpr int SP = K2_SPITCH(F32);
  char* const st = smem + 2 * K2_WBUF + w * K2_STAGE(F32);

  auto chunk = [&](auto tail_tag, const int c) {
    constexpr bool TAIL = decltype(tail_tag)::value;      // the last chunk of an N that is 32 mod 64: channels 0 .. 31 exist
    const char* const wb = smem + (c & 1) * K2_WBUF;
    // the fragments of chunk c + 1 (requested a whole chunk ago) go into the other buffer -- every wave left it before the
    // barrier that ended chunk c - 1 --, those of chunk c + 2 are requested: in vmcnt order they are OLDER than this chunk's
    // stores, so waiting for them next time round does not wait for stores
    if (c + 1 < NC) {
      char* const wn = smem + ((c + 1) & 1) * K2_WBUF;
#pragma unroll
      for (int j = 0; j < FPW; ++j) *(bf16x8*)(wn + (w * FPW + j) * 1024 + lane * 16) = wst[j];
      if (c + 2 < NC) {
#pragma unroll
        for (int j = 0; j < FPW; ++j) wst[j] = wok[j] ? *(const bf16x8*)(wsrc_of(c + 2) + j * 32) : zero8;
      }
    }
    k2u32x4 rm[4];          // RMASK: the ReLU-mask rows of this chunk's stores (same cells as C), requested before the products
    if (RMASK) {
      constexpr int PPR_ = 8 / (TAIL ? 2 : 1);
#pragma unroll
      for (int j = 0; j < 32 * PPR_ / 64; ++j) {
        const int id = lane + 64 * j, row = id / PPR_, pc = id % PPR_;
        long m = m0 + row;
        if (RAGGED) m = m < M ? m : M - 1;
        rm[j] = *(const k2u32x4*)((const char*)(p.R + p.cm.base) + (m * p.cm.ld + c * 64) * 2 + pc * 16);
      }
    }
    f32x4 bv[4];            // (a compile-time switch: as a run-time one every channel tile of the epilogue got its own branch, load and
                            //  s_waitcnt vmcnt(0) -- which on gfx9 also waits for the previous chunk's stores)
    if (BIAS) {
#pragma unroll
      for (int ct = 0; ct < (TAIL ? 2 : 4); ++ct) bv[ct] = *(const f32x4*)(p.bias + c * 64 + ct * 16 + l4 * 4);
    }
    // acc[t][ct]: lane (l15, l4) = row 16 t + l15, output channels 64 c + 16 ct + 4 l4 .. + 4   (A operand = weight fragment)
    f32x4 acc[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) acc[t][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      bf16x8 bf[4];
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) bf[ct] = *(const bf16x8*)(wb + (ct * KS + ks) * 1024 + lane * 16);
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[t][ct] = K2_MFMA(bf[ct], xa[t][ks], acc[t][ct]);
    }
    // ---- epilogue of the chunk: (+ bias) -> bf16 -> the wave's LDS tile [32 rows][64 channels] -> whole 128-byte row segments ----
#pragma unroll
    for (int ct = 0; ct < (TAIL ? 2 : 4); ++ct) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        f32x4 v = acc[t][ct];
        if (BIAS) v += bv[ct];
        if (F32)
          *(f32x4*)(st + (t * 16 + l15) * SP + ct * 64 + l4 * 16) = v;
        else
          *(k2u32x2*)(st + (t * 16 + l15) * SP + ct * 32 + l4 * 8) = (k2u32x2){pack2bf_hw(v[0], v[1]), pack2bf_hw(v[2], v[3])};
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    // 32 rows x 128 (bf16) | 256 (fp32) bytes: lane -> (row, 16-byte piece), 4 | 8 instructions
    constexpr int PPR = (F32 ? 16 : 8) / (TAIL ? 2 : 1);      // 16-byte pieces per row (tail chunk: half a row)
#pragma unroll
    for (int j = 0; j < 32 * PPR / 64; ++j) {
      const int id = lane + 64 * j, row = id / PPR, pc = id % PPR;
      k2u32x4 v = *(const k2u32x4*)(st + row * SP + pc * 16);
      if (RMASK) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          v[q] = (bf_lo(rm[j][q]) > 0.f ? (v[q] & 0xffffu) : 0u) | (bf_hi(rm[j][q]) > 0.f ? (v[q] & 0xffff0000u) : 0u);
      }
      const long m = m0 + row;
      if (!RAGGED || m < M) *(k2u32x4*)(C + (m * p.cm.ld + c * 64) * (F32 ? 4 : 2) + pc * 16) = v;
    }
    __syncthreads();        // the next chunk's fragments are complete; every wave has left this chunk's buffer and its own tile
  };
#pragma unroll 1
  for (int c = 0; c < NCF; ++c) chunk(std::false_type{}, c);
  if (NC > NCF) chunk(std::true_type{}, NCF);
}

// the launch qualifies (checked by the caller, dfold_gemm_bf16): plain row maps, one K segment of 256, N % 64 == 0,
// flags within {OUT_BF16, BIAS}, alpha == 1
template <bool BIAS, bool RAGGED, bool F32, int KS>
static void k256_launch(const GemmParams& p, int block0, int blocks, hipStream_t stream) {
  DFOLD_MAX_LDS_ONCE((gemm_k256_kernel<BIAS, RAGGED, F32, KS>), K2_LDS(F32));
  DFOLD_LAUNCH((gemm_k256_kernel<BIAS, RAGGED, F32, KS>), dim3((unsigned)blocks), dim3(512), (size_t)K2_LDS(F32), stream, p, block0);
}
template <bool RAGGED, int KS>
static void k256_pick(const GemmParams& p, int block0, int blocks, hipStream_t stream) {
  const bool bias = (p.flags & DFOLD_GEMM_BIAS) != 0, f32 = !(p.flags & DFOLD_GEMM_OUT_BF16);
  if (bias && f32) k256_launch<true, RAGGED, true, KS>(p, block0, blocks, stream);
  else if (bias) k256_launch<true, RAGGED, false, KS>(p, block0, blocks, stream);
  else if (f32) k256_launch<false, RAGGED, true, KS>(p, block0, blocks, stream);
  else k256_launch<false, RAGGED, false, KS>(p, block0, blocks, stream);
}
template <bool RAGGED, int KS>
static void k256_rmask(const GemmParams& p, int block0, int blocks, hipStream_t stream) {
  DFOLD_MAX_LDS_ONCE((gemm_k256_kernel<false, RAGGED, false, KS, true>), K2_LDS(false));
  DFOLD_LAUNCH((gemm_k256_kernel<false, RAGGED, false, KS, true>), dim3((unsigned)blocks), dim3(512), (size_t)K2_LDS(false), stream, p, block0);
}
// K = p.seglen: 256, or 8 ... 64 in whole groups of 8.  DFOLD_GEMM_RELUMASK (bf16 out, no bias): the ReLU backward in the epilogue,
// with A's row-block flags (p.nz_ps) dead workgroups store zeros.
int dfold_gemm_k256_launch(const GemmParams& p, hipStream_t stream) {
  const int full = p.M / 256;
  const bool k8 = p.seglen == 256;
  if (p.flags & DFOLD_GEMM_RELUMASK) {
    if (full > 0) {
      if (k8) k256_rmask<false, 8>(p, 0, full, stream);
      else k256_rmask<false, 2>(p, 0, full, stream);
      if (dfold_check_launch() != DFOLD_OK) return DFOLD_ELAUNCH;
    }
    if (p.M % 256) {
      if (k8) k256_rmask<true, 8>(p, full, 1, stream);
      else k256_rmask<true, 2>(p, full, 1, stream);
    }
    return dfold_check_launch();
  }
  if (full > 0) {
    if (k8) k256_pick<false, 8>(p, 0, full, stream);
    else k256_pick<false, 2>(p, 0, full, stream);
    if (dfold_check_launch() != DFOLD_OK) return DFOLD_ELAUNCH;
  }
  if (p.M % 256) {
    if (k8) k256_pick<true, 8>(p, full, 1, stream);
    else k256_pick<true, 2>(p, full, 1, stream);
  }
  return dfold_check_launch();
}
