// bf16 MFMA GEMM / implicit-GEMM engine for gfx950 (MI355X).
//
//   C[m, n] = epilogue( alpha * sum_k A[row(m), k] * B[n, k] )        ("NT": both operands K-contiguous)
//
// One kernel serves every dense contraction of the DFOLDv2 hot path:
//   * the 5x5 conv tower (reference src/model/ipa_pytorch_dynamic.py:664-706) as an implicit GEMM over
//     the zero-padded channels-last [window, frame+4, residue+4, C] activation grid: K is split into 25
//     tap segments, each adding a constant element offset to the gathered A row (no im2col, no bounds
//     checks: the border of the grid is zero) -- forward and dgrad (flipped/transposed weights);
//   * its wgrad (K = frames x residues, per-window segments over transposed activations);
//   * all nn.Linear layers of IPA / AngleResnet / embedders / triangle ops, forward and backward;
//   * batched attention products (two-level batch strides).
//
// Structure: 128x128x64 tile, 4 waves (2x2), each wave 2x2 v_mfma_f32_32x32x16_bf16 tiles (fp32
// accumulate); operands go HBM -> LDS with global_load_lds_dwordx4 (no VGPR round trip), double
// buffered, one barrier per K step; LDS rows are 128 B with a 16-B-chunk XOR swizzle applied on the
// *source* address (the LDS-DMA image is lane-linear) and on the ds_read_b128 side, which makes the
// fragment reads bank-conflict free; the blockIdx -> tile map is XCD-aware (tiles that share an A row
// panel sit on one XCD's L2).
#include "gemm_engine.h"
#include <stdlib.h>

// ROLE only selects the kernel symbol (so that profiles attribute time to the conv tower separately):
//   0 = generic dense / batched GEMM, 1 = 5x5 conv implicit GEMM over the padded grid (forward & dgrad),
//   2 = 5x5 conv wgrad (25 tap batches over transposed activations)
template <int ROLE>
__global__ __launch_bounds__(256, 2) void dfold_mfma_gemm_kernel(const GemmParams p) {
  __shared__ __attribute__((aligned(16))) char lds[2 * 2 * TILE_BYTES];  // [buf][A|B][128 rows][128 B]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = w >> 1, wn = w & 1;

  // ---- XCD-aware tile id (bijective for any grid size) ----
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q = nwg >> 3, r8 = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  const int lid = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + idx;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int m0 = (lid / tiles_n) * BM, n0 = (lid % tiles_n) * BN;
  const int z = blockIdx.y, z0 = z / p.nb1, z1 = z - z0 * p.nb1;
  const bf16_t* A = p.A + z0 * p.sa0 + z1 * p.sa1;
  const bf16_t* B = p.B + z0 * p.sb0 + z1 * p.sb1;
  const long coff = z0 * p.sc0 + z1 * p.sc1;

  // ---- staging assignment: wave w, pass t covers tile rows (t*4+w)*8 .. +8, lane -> (row, 16-B chunk) ----
  const int cphys = lane & 7;
  const int rsub = lane >> 3;
  const int clog = cphys ^ ((((w & 1) << 2) + (lane >> 4)) & 7);  // logical k-chunk this lane fetches
  const int kofs = clog * 8;
  const bf16_t* arow[4];
  const bf16_t* brow[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int r = (t * 4 + w) * 8 + rsub;
    const long m = (long)m0 + r;
    arow[t] = m < p.M ? A + row_off(p.am, m) + kofs : nullptr;
    const long n = (long)n0 + r;
    brow[t] = n < p.N ? B + n * p.ldb + kofs : nullptr;
  }
  const int sps = (p.seglen + BK - 1) / BK;  // K steps per segment
  const int nsteps = p.nseg * sps;

  // K-step cursor of the NEXT tile to stage (stages are issued in increasing step order): segment (hi, lo) and
  // offset kk inside it -- pure scalar arithmetic, no table loads (a VMEM load here would force vmcnt(0)).
  int st_hi = 0, st_mid = 0, st_lo = 0, st_kk = 0;
  auto stage = [&](int buf, int) {
    const long ao = p.a_seg0 + st_hi * p.a_seg_s0 + st_mid * p.a_seg_s1 + st_lo * p.a_seg_s2 + st_kk;
    const long bo = p.b_seg0 + st_hi * p.b_seg_s0 + st_mid * p.b_seg_s1 + st_lo * p.b_seg_s2 + st_kk;
    const int kk = st_kk;
    const bool kin = (kk + kofs) < p.seglen;
    st_kk += BK;
    if (st_kk >= p.seglen) {
      st_kk = 0;
      if (++st_lo == p.seg_div) {
        st_lo = 0;
        if (++st_mid == p.seg_div_mid) {
          st_mid = 0;
          ++st_hi;
        }
      }
    }
    char* la = lds + buf * 2 * TILE_BYTES;
    char* lb = la + TILE_BYTES;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const bf16_t* sa = (arow[t] != nullptr && kin) ? arow[t] + ao : p.zeros;
      __builtin_amdgcn_global_load_lds((const void*)sa, (lds_ptr_t)(la + (t * 4 + w) * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const bf16_t* sb = (brow[t] != nullptr && kin) ? brow[t] + bo : p.zeros;
      __builtin_amdgcn_global_load_lds((const void*)sb, (lds_ptr_t)(lb + (t * 4 + w) * 1024), 16, 0, 0);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int frow = lane & 31;
  const int fsw = (lane >> 1) & 7;  // swizzle key of this lane's fragment rows ((row>>1)&7, row = 32*x + frow)
  const int fhalf = lane >> 5;

  stage(0, 0);
  for (int s = 0; s < nsteps; ++s) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (s + 1 < nsteps) stage((s + 1) & 1, s + 1);
    const char* la = lds + (s & 1) * 2 * TILE_BYTES;
    const char* lb = la + TILE_BYTES;
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) {
      const int ch = ((k4 * 2 + fhalf) ^ fsw) << 4;
      bf16x8 af[2], bfr[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = *(const bf16x8*)(la + (wm * 64 + i * 32 + frow) * 128 + ch);
#pragma unroll
      for (int j = 0; j < 2; ++j) bfr[j] = *(const bf16x8*)(lb + (wn * 64 + j * 32 + frow) * 128 + ch);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
  }

  gemm_epilogue<2>(p, acc, (long)m0 + wm * 64, n0 + wn * 64, coff, lane);
}


// ------------------------------------------------------------------------------------------------
// Large-problem variant: 256(M) x 128(N) x 64 tile, 8 waves (4 x 2, each 64x64 = 2x2 MFMA 32x32x16),
// THREE LDS stages (3 x 48 KiB) filled by LDS-DMA with a counted s_waitcnt: the loads of tile s+1 stay in
// flight across the barrier of step s (raw s_barrier -- __syncthreads() would drain vmcnt to 0), so HBM/L2
// latency is hidden behind two MFMA phases.  Per K step a wave issues 6 LDS-DMA pieces (4 A + 2 B) and 16 MFMAs;
// L2->LDS bytes per flop are 0.75x the 128x128 kernel's.
// ------------------------------------------------------------------------------------------------
#define BM2 256
#define A2_BYTES (BM2 * BK * 2)
#define STAGE2_BYTES (A2_BYTES + TILE_BYTES)
#define NSTAGE2 3

template <int ROLE>
__global__ __launch_bounds__(512, 2) void dfold_mfma_gemm256_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char lds2[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = w >> 1, wn = w & 1;
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q = nwg >> 3, r8 = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  const int lid = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + idx;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int m0 = (lid / tiles_n) * BM2, n0 = (lid % tiles_n) * BN;
  const int z = blockIdx.y, z0 = z / p.nb1, z1 = z - z0 * p.nb1;
  const bf16_t* A = p.A + z0 * p.sa0 + z1 * p.sa1;
  const bf16_t* B = p.B + z0 * p.sb0 + z1 * p.sb1;
  const long coff = z0 * p.sc0 + z1 * p.sc1;

  const int cphys = lane & 7;
  const int rsub = lane >> 3;
  const int clog = cphys ^ ((((w & 1) << 2) + (lane >> 4)) & 7);
  const int kofs = clog * 8;
  const bf16_t* arow[4];
  const bf16_t* brow[2];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const long m = (long)m0 + (t * 8 + w) * 8 + rsub;
    arow[t] = m < p.M ? A + row_off(p.am, m) + kofs : nullptr;
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const long n = (long)n0 + (t * 8 + w) * 8 + rsub;
    brow[t] = n < p.N ? B + n * p.ldb + kofs : nullptr;
  }
  const int sps = (p.seglen + BK - 1) / BK;
  const int nsteps = p.nseg * sps;

  // K-step cursor of the NEXT tile to stage (stages are issued in increasing step order): segment (hi, lo) and
  // offset kk inside it -- pure scalar arithmetic, no table loads (a VMEM load here would force vmcnt(0)).
  int st_hi = 0, st_mid = 0, st_lo = 0, st_kk = 0;
  auto stage = [&](int buf, int) {
    const long ao = p.a_seg0 + st_hi * p.a_seg_s0 + st_mid * p.a_seg_s1 + st_lo * p.a_seg_s2 + st_kk;
    const long bo = p.b_seg0 + st_hi * p.b_seg_s0 + st_mid * p.b_seg_s1 + st_lo * p.b_seg_s2 + st_kk;
    const int kk = st_kk;
    const bool kin = (kk + kofs) < p.seglen;
    st_kk += BK;
    if (st_kk >= p.seglen) {
      st_kk = 0;
      if (++st_lo == p.seg_div) {
        st_lo = 0;
        if (++st_mid == p.seg_div_mid) {
          st_mid = 0;
          ++st_hi;
        }
      }
    }
    char* la = lds2 + buf * STAGE2_BYTES;
    char* lb = la + A2_BYTES;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const bf16_t* sa = (arow[t] != nullptr && kin) ? arow[t] + ao : p.zeros;
      __builtin_amdgcn_global_load_lds((const void*)sa, (lds_ptr_t)(la + (t * 8 + w) * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const bf16_t* sb = (brow[t] != nullptr && kin) ? brow[t] + bo : p.zeros;
      __builtin_amdgcn_global_load_lds((const void*)sb, (lds_ptr_t)(lb + (t * 8 + w) * 1024), 16, 0, 0);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int frow = lane & 31;
  const int fsw = (lane >> 1) & 7;
  const int fhalf = lane >> 5;

  stage(0, 0);
  if (nsteps > 1) stage(1, 1);
  int cur = 0;
  for (int s = 0; s < nsteps; ++s) {
    // tile s must have landed; tile s+1 (6 pieces per wave) may stay in flight across the barrier
    if (s + 1 < nsteps)
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (s + 2 < nsteps) {
      int nb = cur + 2;
      if (nb >= NSTAGE2) nb -= NSTAGE2;
      stage(nb, s + 2);
    }
    const char* la = lds2 + cur * STAGE2_BYTES;
    const char* lb = la + A2_BYTES;
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) {
      const int ch = ((k4 * 2 + fhalf) ^ fsw) << 4;
      bf16x8 af[2], bfr[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = *(const bf16x8*)(la + (wm * 64 + i * 32 + frow) * 128 + ch);
#pragma unroll
      for (int j = 0; j < 2; ++j) bfr[j] = *(const bf16x8*)(lb + (wn * 64 + j * 32 + frow) * 128 + ch);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    cur = cur + 1 == NSTAGE2 ? 0 : cur + 1;
  }

  const bool vec_ok = (p.flags & DFOLD_GEMM_OUT_BF16) && (p.N % BN) == 0 && ((p.cm.ld | p.cm.base | coff) & 7) == 0;
  if (vec_ok) {
    __syncthreads();
    gemm_epilogue_lds_bf16<2>(p, acc, (long)m0 + wm * 64, n0 + wn * 64, coff, lane, lds2 + w * (32 * EPI_ROWB(2) + 256));
  } else {
    gemm_epilogue<2>(p, acc, (long)m0 + wm * 64, n0 + wn * 64, coff, lane);
  }
}


// ------------------------------------------------------------------------------------------------
// Conv-tower variant: 256(M) x 320(N) x 64 tile -- 640 and 1280 output channels tile exactly (no padding waste),
// 8 waves as 4(M) x 2(N), each wave 64 x 160 = 2 x 5 MFMA 32x32x16 tiles (160 accumulator registers).
// Per K16 step a wave reads 7 fragments for 10 MFMAs (0.7 ds_read_b128 per MFMA vs 1.0 in the 64x64 wave tile) and a
// K step moves 72 KiB HBM/L2 -> LDS for 10.5 MFLOP (vs 48 KiB for 4.2): the LDS port stops being the limiter.
// Two LDS stages (2 x 72 KiB): wait tile s, barrier, launch the LDS-DMA of tile s+1, run 40 MFMAs per wave on tile s.
// ------------------------------------------------------------------------------------------------
#define BM3 256
#define BN3 320
#define A3_BYTES (BM3 * BK * 2)
#define B3_BYTES (BN3 * BK * 2)
#define STAGE3_BYTES (A3_BYTES + B3_BYTES)
// "halo" form of the 5x5 conv launch: one A tile of 256 + 4 consecutive residues (8-row DMA pieces: 264 rows) serves the five
// residue taps of a (channel chunk, frame tap) group
#define HALO_PIECES 33
#define HALO_BYTES (HALO_PIECES * 1024)
#define HALO_LDS (2 * HALO_BYTES + 2 * B3_BYTES)

// HALO (5x5 conv launches whose 256-row tiles are runs of consecutive residues of one frame row, N_res % 256 == 0): the five
// residue taps dn of a (channel chunk, frame tap) group read the SAME 260 activation rows shifted by one row each, so the
// A operand is staged once per group as a 264-row halo tile and the fragment reads of tap dn start dn rows further down;
// only the weight tile is staged per K step.  LDS-DMA pieces per wave and K step: 6 instead of 9 (their issue cost inside
// the MFMA stream is the largest loss of the K loop: scripts/exp_conv_variants.py, NODMA +45 %).  Same K order, same
// accumulation order: results are bit-identical to the per-tap form.
// RZ (round 6): row-block flags for A (dfold_gemm_desc.nz_ps with a plain row map): a tile whose 256 A rows are all zero by the
// flags leaves out its K walk (zero accumulators, the usual epilogue).
template <int ROLE, int NJ, bool HALO = false, bool RZ = false>
__global__ __launch_bounds__(512, 2) void dfold_mfma_gemm320_kernel(const GemmParams p) {
  constexpr int BNW = 64 * NJ;                 // N tile: 320 (NJ = 5) or 256 (NJ = 4)
  constexpr int BW_BYTES = BNW * BK * 2;
  constexpr int STAGE_BYTES = A3_BYTES + BW_BYTES;
  extern __shared__ __attribute__((aligned(16))) char lds3[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave id in an SGPR: LDS-DMA targets and the role branch stay scalar
  const int wm = w >> 1, wn = w & 1;
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q = nwg >> 3, r8 = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  const int lid = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + idx;
  const int tiles_n = (p.N + BNW - 1) / BNW;
  int m0, n0, z0, z1;
  if (ROLE == 2) {
    // conv wgrad: 1-D grid over (n tile, dn, df, m tile), m tile fastest.  The ~31 consecutive ids that the XCD map
    // puts on one XCD then share ONE column-shifted operand copy (dn) of one n panel and the 5 m panels, and the 5
    // row shifts (df) of a panel touch the same cache lines a few K steps apart: operands stream from HBM about
    // once per XCD instead of once per workgroup.
    const int tiles_m = (p.M + BM3 - 1) / BM3;
    int r = lid;
    m0 = (r % tiles_m) * BM3;
    r /= tiles_m;
    z0 = r % 5;   // df
    r /= 5;
    z1 = r % 5;   // dn
    n0 = (r / 5) * BNW;
  } else {
    m0 = (lid / tiles_n) * BM3;
    n0 = (lid % tiles_n) * BNW;
    const int z = blockIdx.y;
    z0 = z / p.nb1;
    z1 = z - z0 * p.nb1;
  }
  const char* A = (const char*)(p.A + z0 * p.sa0 + z1 * p.sa1);
  const char* B = (const char*)(p.B + z0 * p.sb0 + z1 * p.sb1);
  const long coff = z0 * p.sc0 + z1 * p.sc1;

  // Per-lane staging offsets are 32-bit byte offsets from the (wave-uniform) operand base, so every LDS-DMA is
  // "SGPR base + VGPR offset" and 9 registers hold all row addresses.  Rows past M / N are clamped (their products
  // are never stored); the host only dispatches here when seglen % 64 == 0, so there is no K tail to zero.
  const int cphys = lane & 7;
  const int rsub = lane >> 3;
  const int clog = cphys ^ ((((w & 1) << 2) + (lane >> 4)) & 7);
  const int kofs = clog * 8;
  unsigned aoff[4], boff[NJ];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    long m = (long)m0 + (t * 8 + w) * 8 + rsub;
    if (m >= p.M) m = p.M - 1;
    aoff[t] = (unsigned)((row_off(p.am, m) + kofs) * 2);
  }
  // HALO: byte offset of the tile's first halo row (tap (0,0) corner of row m0; the 264 rows follow at the row pitch)
  const unsigned h_row0 = HALO ? (unsigned)(row_off(p.am, m0) * 2) : 0u;
  const unsigned h_ld2 = (unsigned)(p.am.ld * 2);
  // piece j holds LDS rows 8j + rsub; swizzle key of that row = ((8j + rsub) >> 1) & 7 = (4 (j & 1) + (rsub >> 1)) & 7
  const unsigned h_ce = (unsigned)((cphys ^ ((rsub >> 1) & 7)) * 16), h_co = (unsigned)((cphys ^ ((4 + (rsub >> 1)) & 7)) * 16);
  auto halo_off = [&](int j) {   // lane's source byte offset of halo piece j (rows past the 260th re-read the last one)
    unsigned r = (unsigned)(j * 8 + rsub);
    r = r < 259u ? r : 259u;
    return h_row0 + r * h_ld2 + ((j & 1) ? h_co : h_ce);
  };
#pragma unroll
  for (int t = 0; t < NJ; ++t) {
    long n = (long)n0 + (t * 8 + w) * 8 + rsub;
    if (n >= p.N) n = p.N - 1;
    boff[t] = (unsigned)((n * p.ldb + kofs) * 2);
  }
  // 5x5 conv: a tile whose rows all lie in one window only needs the frame taps df with 0 <= frame + df - 2 < F for some
  // row of the tile; the others read nothing but the zero border of the grid.  Pure scalar set-up (the tile origin is
  // wave-uniform): the K walk below simply starts df_lo taps in and wraps after ndf of them.
  int df_lo = 0, ndf = p.seg_div_mid;
  if (ROLE == 1 && p.conv_F > 0) {
    const unsigned mlast = (unsigned)((m0 + BM3 - 1 < p.M ? m0 + BM3 - 1 : p.M - 1));
    const unsigned wf0 = (unsigned)m0 / (unsigned)p.am.n, wf1 = mlast / (unsigned)p.am.n;
    const unsigned w0 = wf0 / (unsigned)p.am.f, w1 = wf1 / (unsigned)p.am.f;
    if (w0 == w1) {
      const int f0 = p.conv_f0 + (int)(wf0 - w0 * (unsigned)p.am.f), f1 = p.conv_f0 + (int)(wf1 - w1 * (unsigned)p.am.f);
      const int lo = 2 - f1 > 0 ? 2 - f1 : 0, hi = p.conv_F + 2 - f0 < 5 ? p.conv_F + 2 - f0 : 5;
      if (hi > lo) {
        df_lo = lo;
        ndf = hi - lo;
      }
    }
  }
  const int nsteps = (ROLE == 1 && p.conv_F > 0) ? (p.nseg / p.seg_div_mid) * ndf * (p.seglen / BK) : p.nseg * (p.seglen / BK);

  // Tile cursor: (pa, pb) = operand byte pointers of the tile staged next, advanced INCREMENTALLY and branch-free with
  // integer masks (sign-bit tricks keep every step on the scalar ALU: the closed form seg0 + hi*s0 + mid*s1 + lo*s2 + kk
  // cost ~85 scalar 64-bit multiply/add instructions plus VALU round trips per K step, issued by every wave right
  // before the barrier with the matrix pipe idle).  Step deltas in bytes:
  //   inside a segment +BK;  kk wraps: +E1;  lo wraps too: +E2;  mid wraps too: +E3   (differences, accumulated)
  // After the last tile the pointers stay put: the final, unused prefetch re-reads valid memory into the idle buffer.
  const long BK2 = BK * 2;
  const long ea1 = (p.a_seg_s2 - p.seglen) * 2, ea2 = (p.a_seg_s1 - (long)p.seg_div * p.a_seg_s2) * 2,
             ea3 = (p.a_seg_s0 - (long)ndf * p.a_seg_s1) * 2;
  const long eb1 = (p.b_seg_s2 - p.seglen) * 2, eb2 = (p.b_seg_s1 - (long)p.seg_div * p.b_seg_s2) * 2,
             eb3 = (p.b_seg_s0 - (long)ndf * p.b_seg_s1) * 2;
  const char* pa = A + (p.a_seg0 + (long)df_lo * p.a_seg_s1) * 2;
  const char* pb = B + (p.b_seg0 + (long)df_lo * p.b_seg_s1) * 2;
  int st_mid = 0, st_lo = 0, st_kk = 0, st_left = nsteps;
  // HALO prefetch cursor: h_next = base of the group being prefetched (the one after the group of the step in flight),
  // h_mid its frame tap, h_gleft = groups from it to the end (it stays on the last group when there is none left: the
  // surplus loads go to the idle buffer), h_dn = position inside the current group, h_buf = halo buffer of the current group
  const int h_groups = nsteps / 5;
  const char* h_next = A + (p.a_seg0 + (long)df_lo * p.a_seg_s1) * 2;
  int h_mid = 0, h_gleft = h_groups, h_dn = 0, h_buf = 0;
  if (HALO && h_groups > 1) {       // the prefetched group starts as group 1
    h_gleft = h_groups - 1;
    const bool wrap = ndf == 1;
    h_mid = wrap ? 0 : 1;
    h_next += wrap ? p.a_seg_s0 * 2 : p.a_seg_s1 * 2;
  }
  auto stage = [&](int buf) {
    const char* sa = pa;
    const char* sb = pb;
    const unsigned adv = (unsigned)(1 - st_left) >> 31;            // st_left > 1
    st_left -= (int)adv;
    int kk = st_kk + BK;
    const unsigned w0 = (unsigned)(p.seglen - 1 - kk) >> 31;       // kk >= seglen
    kk &= (int)(w0 - 1u);
    int lo = st_lo + (int)w0;
    const unsigned w1 = (unsigned)(p.seg_div - 1 - lo) >> 31;      // lo >= seg_div
    lo &= (int)(w1 - 1u);
    int mid = st_mid + (int)w1;
    const unsigned w2 = (unsigned)(ndf - 1 - mid) >> 31;           // mid >= ndf (= seg_div_mid unless frame taps are skipped)
    mid &= (int)(w2 - 1u);
    st_kk = kk;
    st_lo = lo;
    st_mid = mid;
    const long k0 = -(long)w0, k1 = -(long)w1, k2 = -(long)w2, ka = -(long)adv;
    pa += (BK2 + (ea1 & k0) + (ea2 & k1) + (ea3 & k2)) & ka;
    pb += (BK2 + (eb1 & k0) + (eb2 & k1) + (eb3 & k2)) & ka;
    char* la = lds3 + buf * STAGE_BYTES;
    char* lb = la + A3_BYTES;
    if (HALO) {
      // one piece of the NEXT group's halo per wave and step (5 steps x 8 waves = 40 slots for 33 pieces: the surplus
      // slots re-load piece 32, so every wave issues the same instruction count and the loop stays branch-free)
      int j = h_dn * 8 + w;
      j = j < HALO_PIECES - 1 ? j : HALO_PIECES - 1;
      la = lds3 + (h_buf ^ 1) * HALO_BYTES;
      lb = lds3 + 2 * HALO_BYTES + buf * BW_BYTES;
      __builtin_amdgcn_global_load_lds((const void*)(h_next + halo_off(j)), (lds_ptr_t)(la + j * 1024), 16, 0, 0);
      // advance (dn, group) -- scalar, branch-free: at the end of a group the prefetched group becomes current
      const int dn1 = h_dn + 1;
      const unsigned gw = (unsigned)(4 - dn1) >> 31;                 // dn1 == 5
      h_dn = dn1 & (int)(gw - 1u);
      h_buf ^= (int)gw;
      const unsigned more = ((unsigned)(1 - h_gleft) >> 31) & gw;    // a group after the prefetched one exists
      h_gleft -= (int)more;
      int mid = h_mid + (int)more;
      const unsigned mw = ((unsigned)(ndf - 1 - mid) >> 31) & more;  // frame tap wraps: next channel chunk
      mid &= (int)(mw - 1u);
      h_mid = mid;
      h_next += ((p.a_seg_s1 * 2) & -(long)more) + (ea3 & -(long)mw);
    }
    if (!HALO) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
        __builtin_amdgcn_global_load_lds((const void*)(sa + aoff[t]), (lds_ptr_t)(la + (t * 8 + w) * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < NJ; ++t)
      __builtin_amdgcn_global_load_lds((const void*)(sb + boff[t]), (lds_ptr_t)(lb + (t * 8 + w) * 1024), 16, 0, 0);
  };

  f32x16 acc[2][NJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int frow = lane & 31;
  const int fsw = (lane >> 1) & 7;
  const int fhalf = lane >> 5;
  const int fa = (wm * 64 + frow) * 128;
  const int fb = (HALO ? 0 : A3_BYTES) + (wn * (32 * NJ) + frow) * 128;

  bf16x8 af[2][2], bfr[2][NJ];
  // HALO: the A fragments of residue tap dn sit dn rows further down the halo tile; their swizzle key follows the row
  int c_dn = 0, c_hb = 0;          // residue tap / halo buffer of the tile being read (advanced once per K step)
  const char* c_ha = lds3;         // halo tile + dn rows
  int c_sw = fsw;
  auto halo_step = [&]() {          // called when a wave moves on to the next tile
    const int d1 = c_dn + 1;
    const unsigned gw = (unsigned)(4 - d1) >> 31;
    c_dn = d1 & (int)(gw - 1u);
    c_hb ^= (int)gw;
    c_ha = lds3 + c_hb * HALO_BYTES + c_dn * 128;
    c_sw = ((frow + c_dn) >> 1) & 7;
  };
  auto ldfrag = [&](int set, const char* base, int k4) {
    const int ch = ((k4 * 2 + fhalf) ^ fsw) << 4;
    if (HALO) {
      const int cha = ((k4 * 2 + fhalf) ^ c_sw) << 4;
#pragma unroll
      for (int i = 0; i < 2; ++i) af[set][i] = *(const bf16x8*)(c_ha + fa + i * 32 * 128 + cha);
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i) af[set][i] = *(const bf16x8*)(base + fa + i * 32 * 128 + ch);
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) bfr[set][j] = *(const bf16x8*)(base + fb + j * 32 * 128 + ch);
  };
  auto mma = [&](int set) {
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[set][i], bfr[set][j], acc[i][j], 0, 0, 0);
  };

  bool rz_live = true;
  if (RZ) {
    const int bs = p.nz_f0;
    const int m_last = (m0 + BM3 < p.M ? m0 + BM3 : p.M) - 1;
    rz_live = p.nz_ps[m_last / bs + 1] - p.nz_ps[m0 / bs] > 0;
  }
  if (!RZ || rz_live) {
  if (HALO) {
    // prologue: the whole halo tile of group 0 (33 pieces over 8 waves, surplus slots re-load piece 32), then weight tile 0
    const char* h0 = A + (p.a_seg0 + (long)df_lo * p.a_seg_s1) * 2;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      int j = w + 8 * i;
      j = j < HALO_PIECES - 1 ? j : HALO_PIECES - 1;
      __builtin_amdgcn_global_load_lds((const void*)(h0 + halo_off(j)), (lds_ptr_t)(lds3 + j * 1024), 16, 0, 0);
    }
    {
      const int keep_dn = h_dn, keep_buf = h_buf, keep_mid = h_mid, keep_left = h_gleft;
      const char* keep_next = h_next;
      stage(0);                    // issues weight tile 0 (and one harmless piece of group 1); undo the halo cursor advance
      h_dn = keep_dn; h_buf = keep_buf; h_mid = keep_mid; h_gleft = keep_left; h_next = keep_next;
    }
  } else {
    stage(0);
  }
  const int bbase = HALO ? 2 * HALO_BYTES : 0, bstride = HALO ? BW_BYTES : STAGE_BYTES;
  constexpr int BLK2 = HALO ? 1 : 4;   // LDS-DMA pieces beyond the NJ weight pieces per wave and step
  // Static wave priority (s_setprio is scalar and ignores EXEC: the guard is wave-uniform).  1 (default): the late group
  // B outranks group A for the whole K loop (+2.3 % on the conv launches, scripts/exp_conv_prio.py); 2: the reverse
  // (-0.3 %); 0: none.  Per-cluster flips inside the loop need scalar branches there, which break the pinned
  // MFMA / DMA / LDS-read interleave (3x slower).
  if ((p.prio == 1 && w >= 4) || (p.prio == 2 && w < 4)) __builtin_amdgcn_s_setprio(1);
  if (w < 4) {
    // ---- group A (one wave per SIMD): per K step [fragment reads][40 MFMAs], DMA pieces between the MFMAs ----
    for (int s = 0; s < nsteps; ++s) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const char* base = lds3 + bbase + (s & 1) * bstride;
      ldfrag(0, base, 0);
      ldfrag(1, base, 1);
      stage((s + 1) & 1);
      mma(0);
      ldfrag(0, base, 2);
      mma(1);
      ldfrag(1, base, 3);
      mma(0);
      mma(1);
      if (HALO) halo_step();
      // issue order: [2(2+NJ) reads] [NJ x (2 MFMA, 1 DMA)] [2+NJ reads] [BLK2 x (2 MFMA, 1 DMA)] [rest of mma(1)] [2+NJ reads] [4NJ MFMA]
      __builtin_amdgcn_sched_group_barrier(0x100, 2 * (2 + NJ), 0);
#pragma unroll
      for (int g = 0; g < NJ; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x100, 2 + NJ, 0);
#pragma unroll
      for (int g = 0; g < BLK2; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
      }
      if (2 * NJ - 2 * BLK2 > 0) __builtin_amdgcn_sched_group_barrier(0x008, 2 * NJ - 2 * BLK2, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2 + NJ, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 4 * NJ, 0);
    }
  } else {
    // ---- group B (the second wave of every SIMD) runs HALF A K STEP BEHIND: it enters each step with the fragments of
    // the previous tile's second half already in registers and issues those 20 MFMAs while group A is still reading its
    // fragments; it reads the current tile late in the step (after A).  The two waves of a SIMD therefore alternate on
    // the matrix pipe instead of both stalling on LDS right after the barrier.  Same barriers, same DMA share. ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    ldfrag(0, lds3 + bbase, 0);
    ldfrag(1, lds3 + bbase, 1);
    stage(1);
    mma(0);
    ldfrag(0, lds3 + bbase, 2);
    mma(1);
    ldfrag(1, lds3 + bbase, 3);
    if (HALO) halo_step();
    for (int s = 1; s < nsteps; ++s) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const char* base = lds3 + bbase + (s & 1) * bstride;
      stage((s + 1) & 1);
      mma(0);                 // previous tile, K16 blocks 2 and 3
      mma(1);
      ldfrag(0, base, 0);
      ldfrag(1, base, 1);
      mma(0);
      ldfrag(0, base, 2);
      mma(1);
      ldfrag(1, base, 3);
      if (HALO) halo_step();
#pragma unroll
      for (int g = 0; g < BLK2 + NJ; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
      }
      if (2 * NJ - 2 * BLK2 > 0) __builtin_amdgcn_sched_group_barrier(0x008, 2 * NJ - 2 * BLK2, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2 * (2 + NJ), 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 2 * NJ, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2 + NJ, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 2 * NJ, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2 + NJ, 0);
    }
    mma(0);
    mma(1);
  }
  }   // rz_live
  if (ROLE == 1 && p.ws != nullptr) {
    // Deterministic split-K: gridDim.y workgroups hold partial sums of this output tile (each walked its own range of
    // channel chunks).  Every one parks its fp32 partial tile in the workspace; the last to arrive adds the partials in
    // the fixed order z = 0 .. S-1 and goes on to the epilogue, the others are done.
    const int S = gridDim.y;
    const long tile_elems = (long)BM3 * BNW;
    // partial tiles as 16-byte vectors per lane ([accumulator tile][quad of registers][lane][4]): with 4-byte elements the
    // reduction below was 160 S dependent dword loads per lane -- the last arriver of a 20-way split took ~0.2 ms per launch
    float* slot = p.ws + ((long)blockIdx.y * nwg + lid) * tile_elems + (long)w * (2 * NJ * 1024) + lane * 4;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4)
          *(f32x4*)(slot + ((i * NJ + j) * 4 + q4) * 256) =
              (f32x4){acc[i][j][4 * q4], acc[i][j][4 * q4 + 1], acc[i][j][4 * q4 + 2], acc[i][j][4 * q4 + 3]};
    __threadfence();
    __syncthreads();
    __shared__ int s_last;
    if (tid == 0) s_last = atomicAdd(p.cnt + lid, 1) == S - 1;
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    const float* part = p.ws + (long)lid * tile_elems + (long)w * (2 * NJ * 1024) + lane * 4;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    for (int z = 0; z < S; ++z) {
      const float* pz = part + (long)z * nwg * tile_elems;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) {
            const f32x4 v = *(const f32x4*)(pz + ((i * NJ + j) * 4 + q4) * 256);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][4 * q4 + r] += v[r];
          }
    }
    if (tid == 0) p.cnt[lid] = 0;   // counters are left clean for the next launch
  }
  const bool vec_ok = (p.flags & DFOLD_GEMM_OUT_BF16) && (p.N % BNW) == 0 && ((p.cm.ld | p.cm.base | coff) & 7) == 0;
  if (vec_ok) {
    __syncthreads();   // every wave is done with the operand stages: reuse the LDS as per-wave output staging
    gemm_epilogue_lds_bf16<NJ>(p, acc, (long)m0 + wm * 64, n0 + wn * (32 * NJ), coff, lane, lds3 + w * (32 * EPI_ROWB(NJ) + 256));
  } else {
    gemm_epilogue<NJ>(p, acc, (long)m0 + wm * 64, n0 + wn * (32 * NJ), coff, lane);
  }
}



static long row_vw_host(const RowMap& r) { return ((long)r.f * r.wp + 255) / 256 * 256; }
static long row_off_host(const RowMap& r, long m) {
  if (r.mode == 0) return r.base + m * r.ld;
  if (r.mode == 2) {
    const long vw = row_vw_host(r), w = m / vw, v = m - w * vw;
    return r.base + (w * ((long)r.fp * r.wp) + v) * r.ld;
  }
  const long wf = m / r.n, n = m - wf * r.n, w = wf / r.f, f = wf - w * r.f;
  return r.base + (((w * r.fp + f) * r.wp) + n) * r.ld;
}

static RowMap to_rowmap(const dfold_rowmap* r) {
  RowMap o;
  o.base = r->base; o.ld = r->ld; o.mode = r->mode; o.n = r->n; o.f = r->f; o.fp = r->fp; o.wp = r->wp;
  return o;
}

extern "C" int dfold_gemm_bf16(const dfold_gemm_desc* d, void* stream) {
  if (!d || !d->A || !d->B || !d->C || !d->zeros) return DFOLD_EINVAL;
  if (d->M <= 0 || d->N <= 0 || d->nseg <= 0 || d->seglen <= 0 || d->nbatch <= 0) return DFOLD_EINVAL;
  if ((d->seglen & 7) || (d->ldb & 7) || (d->a_rows.ld & 7) || (d->a_rows.base & 7)) return DFOLD_EINVAL;  // 16-B chunks
  if ((d->a_seg0 | d->a_seg_s0 | d->a_seg_s1 | d->a_seg_s2 | d->b_seg0 | d->b_seg_s0 | d->b_seg_s1 | d->b_seg_s2) & 7) return DFOLD_EINVAL;
  if ((d->flags & DFOLD_GEMM_BIAS) && !d->bias) return DFOLD_EINVAL;
  if ((d->flags & (DFOLD_GEMM_RESID | DFOLD_GEMM_RELUMASK)) && !d->R) return DFOLD_EINVAL;
  if ((d->flags & (DFOLD_GEMM_ACCUM | DFOLD_GEMM_ATOMIC)) && (d->flags & DFOLD_GEMM_OUT_BF16)) return DFOLD_EINVAL;
  if ((d->flags & DFOLD_GEMM_C2RELU) && (!d->C2 || d->R2 || !(d->flags & DFOLD_GEMM_OUT_BF16))) return DFOLD_EINVAL;
  if ((d->flags & DFOLD_GEMM_MASK2) && (!d->R2 || d->C2 || !(d->flags & DFOLD_GEMM_OUT_BF16))) return DFOLD_EINVAL;
  if (d->a_rows.mode != 0 && (d->a_rows.n <= 0 || d->a_rows.f <= 0 || d->a_rows.mode > 2 || d->a_rows.mode < 0)) return DFOLD_EINVAL;
  if (d->c_rows.mode != 0 && (d->c_rows.n <= 0 || d->c_rows.f <= 0 || d->c_rows.mode > 2 || d->c_rows.mode < 0)) return DFOLD_EINVAL;
  // mode 2 (cells of a window as one line, any N_res): conv launches of the 512 x 160 kernel only, both maps over the same grid
  const bool lin = d->a_rows.mode == 2 || d->c_rows.mode == 2;
  if (lin && (d->a_rows.mode != 2 || d->c_rows.mode != 2 || d->a_rows.n != d->c_rows.n || d->a_rows.f != d->c_rows.f ||
              d->a_rows.fp != d->c_rows.fp || d->a_rows.wp != d->c_rows.wp || d->a_rows.wp < d->a_rows.n + 4 ||
              d->a_rows.fp < d->a_rows.f + 4 || (long)d->a_rows.f * d->a_rows.wp >= (1L << 30)))
    return DFOLD_EINVAL;
  GemmParams p;
  p.A = (const bf16_t*)d->A; p.B = (const bf16_t*)d->B; p.C = d->C; p.C2 = d->C2;
  p.bias = d->bias; p.R = (const bf16_t*)d->R; p.R2 = (const bf16_t*)d->R2; p.zeros = (const bf16_t*)d->zeros;
  p.a_seg0 = d->a_seg0; p.a_seg_s0 = d->a_seg_s0; p.a_seg_s1 = d->a_seg_s1; p.a_seg_s2 = d->a_seg_s2;
  p.b_seg0 = d->b_seg0; p.b_seg_s0 = d->b_seg_s0; p.b_seg_s1 = d->b_seg_s1; p.b_seg_s2 = d->b_seg_s2;
  p.seg_div = d->seg_div > 0 ? d->seg_div : 1;
  p.seg_div_mid = d->seg_div_mid > 0 ? d->seg_div_mid : 0x7fffffff;
  p.am = to_rowmap(&d->a_rows); p.cm = to_rowmap(&d->c_rows);
  p.ldb = d->ldb;
  p.sa0 = d->sa0; p.sa1 = d->sa1; p.sb0 = d->sb0; p.sb1 = d->sb1; p.sc0 = d->sc0; p.sc1 = d->sc1;
  p.M = d->M; p.N = d->N; p.nseg = d->nseg; p.seglen = d->seglen;
  p.nb1 = d->nb1 > 0 ? d->nb1 : 1; p.flags = d->flags; p.alpha = d->alpha;
  p.ws = nullptr; p.cnt = nullptr; p.sk_per = 0; p.sk_tiles = 0; p.sk_fence = 1;
  p.conv_f0 = (d->conv_frames >> 16) & 0x7fff; p.conv_F = d->conv_frames & 0xffff;
  p.nz_ps = d->nz_ps; p.nz_radius = d->nz_radius; p.nz_f0 = d->nz_f0;
  if (p.nz_ps && (p.nz_radius < 0 || p.nz_f0 < (d->a_rows.mode == 0 ? 1 : 0))) return DFOLD_EINVAL;
  if (lin && (d->M % (int)row_vw_host(p.am))) return DFOLD_EINVAL;
  static int prio_mode = -1;
  if (prio_mode < 0) {
    const char* e = getenv("DFOLD_GEMM_PRIO");
    prio_mode = e ? atoi(e) : 1;
  }
  p.prio = prio_mode;
  const int role = (d->a_rows.mode != 0 && d->seg_div == 5 && d->seg_div_mid == 5) ? 1 : (d->nbatch == 25 && d->nb1 == 5) ? 2 : 0;
  const long steps = (long)d->nseg * ((d->seglen + BK - 1) / BK);
  const long tiles256 = (long)((d->M + BM2 - 1) / BM2) * ((d->N + BN - 1) / BN);
  // K = 256 projections (round 6, gemm_k256.hip; DFOLD_GEMM_K256=0: the tile kernels below, 2: only outputs of 1024 columns and more, 3: K = 256 only)
  {
    static int k256_mode = -1;
    if (k256_mode < 0) {
      const char* e = getenv("DFOLD_GEMM_K256");
      k256_mode = e ? atoi(e) : 1;
    }
    if (k256_mode && role == 0 && d->nbatch == 1 && d->nseg == 1 && d->a_seg0 == 0 && d->b_seg0 == 0 &&
        (d->seglen == 256 || (k256_mode != 3 && d->seglen >= 8 && d->seglen <= 64 && (d->seglen & 7) == 0)) &&
        d->a_rows.mode == 0 && d->c_rows.mode == 0 &&
        ((d->flags & ~(DFOLD_GEMM_OUT_BF16 | DFOLD_GEMM_BIAS)) == 0 ||
         (d->flags == (DFOLD_GEMM_OUT_BF16 | DFOLD_GEMM_RELUMASK) && k256_mode == 1 && (!d->nz_ps || d->nz_f0 > 0))) &&
        d->alpha == 1.f && !d->C2 && d->splitk <= 1 && (d->N % 32) == 0 && d->N >= (k256_mode == 2 ? 1024 : 128) &&
        d->M >= 4096 && ((p.am.ld | p.am.base | p.cm.ld | p.cm.base | d->ldb) & 7) == 0)
      return dfold_gemm_k256_launch(p, (hipStream_t)stream);
  }
  static int variant = -1;   // DFOLD_GEMM_VARIANT=128 forces the 128x128 kernel (A/B measurements)
  if (variant < 0) {
    const char* e = getenv("DFOLD_GEMM_VARIANT");
    variant = e ? atoi(e) : 256;   // 128: 128x128 only; 2560: no 256x320 kernel; default: all variants
  }
  const long tiles320 = (long)((d->M + BM3 - 1) / BM3) * ((d->N + BN3 - 1) / BN3);
  const long a_extent = row_off_host(p.am, d->M - 1) + d->a_rows.ld;   // elements spanned by the A rows of one batch
  int S = 1;
  // splitk == -1: the stream-K form of the one-wave-per-SIMD conv kernel (conv_fwd_w4.hip) when the launch qualifies for that
  // kernel, otherwise an ordinary unsplit launch (a performance-only fallback: the caller's cost model assumed the kernel)
  if (d->splitk == -1 && role == 1 && d->nbatch == 1 && d->splitk_ws && d->splitk_cnt && (d->nseg % 25) == 0 &&
      !(d->flags & (DFOLD_GEMM_ACCUM | DFOLD_GEMM_ATOMIC)) && (d->N % 160) == 0 && d->seglen == BK && d->a_rows.mode != 0 &&
      (lin || (d->a_rows.n % BM3) == 0) && (d->M % BM3) == 0 && d->seg_div == 5 && d->a_seg_s2 == d->a_rows.ld && p.conv_F == 0 &&
      d->a_seg_s0 == BK && d->b_seg_s0 == BK && (d->flags & DFOLD_GEMM_OUT_BF16) && ((p.cm.ld | p.cm.base) & 7) == 0 &&
      a_extent < (1L << 31) && (long)d->N * d->ldb < (1L << 31)) {
    static int sk_env = -1;
    static int n_cus[64] = {0};             // per device (0: not asked yet)
    if (sk_env < 0) {
      const char* e = getenv("DFOLD_CONV_W4");
      const char* h = getenv("DFOLD_CONV_HALO");
      sk_env = (!e || atoi(e) != 0) && (!h || atoi(h) != 0) ? 1 : 0;
    }
    int dev = 0, n_cu = 0;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
      if (n_cus[dev] == 0 && hipDeviceGetAttribute(&n_cus[dev], hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n_cus[dev] = 0;
      n_cu = n_cus[dev];
    }
    if (sk_env && n_cu > 0) {
      p.ws = d->splitk_ws; p.cnt = d->splitk_cnt;
      return dfold_conv_w4_launch_streamk(p, n_cu, (hipStream_t)stream);
    }
  }
  // splitk <= -2: zero-frame-flagged 5x5 conv launch with a split factor chosen on the device, at most -splitk parts per tile
  // (conv_fwd_w4.hip; round 6).  Only the one-wave-per-SIMD kernel has it: anything else is an argument error.
  int nz_adapt = 0;
  if (d->splitk <= -2) {
    if (!d->nz_ps || -d->splitk > 8 || role != 1 || d->nbatch != 1 || !d->splitk_cnt || (d->nseg % 25) ||
        (d->flags & (DFOLD_GEMM_ACCUM | DFOLD_GEMM_ATOMIC)))
      return DFOLD_EINVAL;
    int dev = 0, n_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0)
      return DFOLD_ELAUNCH;
    nz_adapt = -d->splitk;
    p.cnt = d->splitk_cnt;
    p.sk_tiles = n_cu;
  }
  if (d->splitk > 1) {
    const int chunks = d->nseg / 25;
    if (role != 1 || d->nbatch != 1 || !d->splitk_ws || !d->splitk_cnt || (d->nseg % 25) || (chunks % d->splitk) ||
        (lin ? (d->N % 160) : (d->N % BN3)) || (d->seglen % BK) || (d->flags & (DFOLD_GEMM_ACCUM | DFOLD_GEMM_ATOMIC)))
      return DFOLD_EINVAL;
    S = d->splitk;
    p.nseg = d->nseg / S;                                 // every part walks chunks/S channel chunks x 25 taps
    p.sa0 = (long)(chunks / S) * d->a_seg_s0; p.sa1 = 0;  // part z starts z * chunks/S chunks further along K
    p.sb0 = (long)(chunks / S) * d->b_seg_s0; p.sb1 = 0;
    p.sc0 = 0; p.sc1 = 0; p.nb1 = 1;
    p.ws = d->splitk_ws; p.cnt = d->splitk_cnt;
  }
  if (S > 1 && !(a_extent < (1L << 31) && (long)d->N * d->ldb < (1L << 31) && steps / S >= 2)) return DFOLD_EINVAL;
  if (lin) {
    // mode-2 row maps: the 512 x 160 kernel is the only reader (any launch size; split-K workspace: S x ceil(M / 512) x N / 160
    // tiles of 512 x 160 fp32, as many counters as tiles)
    if (role != 1 || d->nbatch != 1 || (d->N % 160) || d->seglen != BK || d->seg_div != 5 || d->a_seg_s2 != d->a_rows.ld ||
        (d->nseg % (25 * S)) || p.conv_F != 0 || d->a_seg_s0 != BK || d->b_seg_s0 != BK || !(d->flags & DFOLD_GEMM_OUT_BF16) ||
        ((p.cm.ld | p.cm.base) & 7) || !(a_extent < (1L << 31)) || !((long)d->N * d->ldb < (1L << 31)))
      return DFOLD_EINVAL;
    return dfold_conv_w4_launch(p, nz_adapt ? -nz_adapt : S, (hipStream_t)stream);
  }
  if (S > 1 || (variant >= 256 && variant != 2560 && (d->N % BN3) == 0 && (d->seglen % BK) == 0 && (d->M >= 2048 || role == 2) && steps >= 4 &&
      tiles320 * d->nbatch >= 128 && a_extent < (1L << 31) && (long)d->N * d->ldb < (1L << 31))) {
    static bool attr3_done = false;
    if (!attr3_done) {
      hipFuncSetAttribute((const void*)dfold_mfma_gemm320_kernel<0, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE3_BYTES);
      hipFuncSetAttribute((const void*)dfold_mfma_gemm320_kernel<1, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE3_BYTES);
      hipFuncSetAttribute((const void*)dfold_mfma_gemm320_kernel<2, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE3_BYTES);
      hipFuncSetAttribute((const void*)dfold_mfma_gemm320_kernel<1, 5, true>, hipFuncAttributeMaxDynamicSharedMemorySize, HALO_LDS);
      attr3_done = true;
    }
    dim3 grid3((unsigned)tiles320, S > 1 ? S : d->nbatch, 1);
    const size_t lds = 2 * STAGE3_BYTES;
    // halo form (DFOLD_CONV_HALO=0 turns it off): every 256-row tile is a run of consecutive residues of one frame row, one
    // K step per (chunk, tap) segment, the residue tap moves the operand by exactly one row
    static int halo_mode = -1;
    if (halo_mode < 0) {
      const char* e = getenv("DFOLD_CONV_HALO");
      halo_mode = e ? atoi(e) : 1;
    }
    const bool halo = halo_mode && role == 1 && d->a_rows.mode != 0 && (lin || (d->a_rows.n % BM3) == 0) && (d->M % BM3) == 0 && d->seglen == BK &&
                      d->seg_div == 5 && d->a_seg_s2 == d->a_rows.ld && (d->nseg % (25 * S)) == 0 && p.conv_F == 0;
    // one-wave-per-SIMD 512 x 160 form (conv_fwd_w4.hip; DFOLD_CONV_W4=0 keeps the 256 x 320 halo kernel): unsplit launches
    // with a bf16 LDS-staged epilogue
    static int w4_mode = -1;
    if (w4_mode < 0) {
      const char* e = getenv("DFOLD_CONV_W4");
      w4_mode = e ? atoi(e) : 1;
    }
    // (split launches too, as long as the 512 x 160 tiling needs no more partial-tile slots / counters than the 256 x 320 one
    //  the caller sized the workspace for: an odd number of 256-row runs stays on the kernel below)
    const long tiles_w4 = (long)((d->M + 511) / 512) * (d->N / 160);
    if (halo && w4_mode && d->nbatch == 1 && (d->N % 160) == 0 && d->a_seg_s0 == BK && d->b_seg_s0 == BK &&
        (d->flags & DFOLD_GEMM_OUT_BF16) && ((p.cm.ld | p.cm.base) & 7) == 0 && (S == 1 || tiles_w4 <= tiles320 || w4_mode == 2 || lin))
      return dfold_conv_w4_launch(p, nz_adapt ? -nz_adapt : S, (hipStream_t)stream);
    // (a flagged launch that does not qualify for the kernel that can choose its split runs unsplit below: performance only)

    if (halo)
      DFOLD_LAUNCH((dfold_mfma_gemm320_kernel<1, 5, true>), grid3, dim3(512), (size_t)HALO_LDS, (hipStream_t)stream, p);
    else if (role == 1)
      DFOLD_LAUNCH((dfold_mfma_gemm320_kernel<1, 5>), grid3, dim3(512), lds, (hipStream_t)stream, p);
    else if (role == 2)
      DFOLD_LAUNCH((dfold_mfma_gemm320_kernel<2, 5>), dim3((unsigned)(tiles320 * 25), 1, 1), dim3(512), lds, (hipStream_t)stream, p);
    else if (p.nz_ps && d->a_rows.mode == 0 && d->nbatch == 1) {
      DFOLD_MAX_LDS_ONCE((dfold_mfma_gemm320_kernel<0, 5, false, true>), 2 * STAGE3_BYTES);
      DFOLD_LAUNCH((dfold_mfma_gemm320_kernel<0, 5, false, true>), grid3, dim3(512), lds, (hipStream_t)stream, p);
    } else
      DFOLD_LAUNCH((dfold_mfma_gemm320_kernel<0, 5>), grid3, dim3(512), lds, (hipStream_t)stream, p);
    return dfold_check_launch();
  }
  // 256 x 256 form of the same kernel (wave tile 64 x 128): N a multiple of 256 -- the per-(window,frame,head) attention
  // products (M = N = 256: one workgroup per batch item reads each operand once) and the 256-wide projections.
  const long tilesq = (long)((d->M + BM3 - 1) / BM3) * (d->N / 256);
  // many small batch items whose 256 x 256 tiles would not even fill the chip once (the triangle contraction at
  // N_res = 256: 128 channel planes = 128 tiles for 256 CUs): 256 x 128 tiles instead (19.5 -> 15.2 us there)
  const bool half_tiles = tilesq * d->nbatch < 256 && d->M >= 256 && d->nbatch >= 64 && steps >= 4 && tiles256 * d->nbatch >= 192;
  // (a plain dense product with fp32 output does not take this form: its fp32 epilogue is not staged through LDS, and the
  // 256 x 128 kernel below ran the step's fp32-output projections 1.4 - 2.4 x faster -- 65536 x 256 x 3072: 235 -> 171 us,
  // 65536 x 256 x 256: 115 -> 47 us, scripts/bench_gemm_shapes.py, round 4)
  const bool dense_f32 = d->nbatch == 1 && !(d->flags & DFOLD_GEMM_OUT_BF16);
  if (!half_tiles && !dense_f32 && variant >= 256 && variant != 2560 && variant != 3201 && role == 0 && (d->N % 256) == 0 && (d->seglen % BK) == 0 && d->M >= 256 &&
      steps >= 2 && tilesq * d->nbatch >= 128 && a_extent < (1L << 31) && (long)d->N * d->ldb < (1L << 31)) {
    static bool attrq_done = false;
    const size_t lds = 2 * (A3_BYTES + 256 * BK * 2);
    if (!attrq_done) {
      hipFuncSetAttribute((const void*)dfold_mfma_gemm320_kernel<0, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attrq_done = true;
    }
    DFOLD_LAUNCH((dfold_mfma_gemm320_kernel<0, 4>), dim3((unsigned)tilesq, d->nbatch, 1), dim3(512), lds, (hipStream_t)stream, p);
    return dfold_check_launch();
  }
  if (variant >= 256 && (d->M >= 1024 || (d->M >= 256 && d->nbatch >= 64)) && steps >= 4 && tiles256 * d->nbatch >= 192) {
    static bool attr_done = false;
    if (!attr_done) {
      hipFuncSetAttribute((const void*)dfold_mfma_gemm256_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, NSTAGE2 * STAGE2_BYTES);
      hipFuncSetAttribute((const void*)dfold_mfma_gemm256_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, NSTAGE2 * STAGE2_BYTES);
      hipFuncSetAttribute((const void*)dfold_mfma_gemm256_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, NSTAGE2 * STAGE2_BYTES);
      attr_done = true;
    }
    dim3 grid2((unsigned)tiles256, d->nbatch, 1);
    const size_t lds = NSTAGE2 * STAGE2_BYTES;
    if (role == 1)
      DFOLD_LAUNCH(dfold_mfma_gemm256_kernel<1>, grid2, dim3(512), lds, (hipStream_t)stream, p);
    else if (role == 2)
      DFOLD_LAUNCH(dfold_mfma_gemm256_kernel<2>, grid2, dim3(512), lds, (hipStream_t)stream, p);
    else
      DFOLD_LAUNCH(dfold_mfma_gemm256_kernel<0>, grid2, dim3(512), lds, (hipStream_t)stream, p);
    return dfold_check_launch();
  }
  const int tiles = ((d->M + BM - 1) / BM) * ((d->N + BN - 1) / BN);
  dim3 grid(tiles, d->nbatch, 1);
  if (role == 1)
    DFOLD_LAUNCH(dfold_mfma_gemm_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, p);
  else if (role == 2)
    DFOLD_LAUNCH(dfold_mfma_gemm_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, p);
  else
    DFOLD_LAUNCH(dfold_mfma_gemm_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, p);
  return dfold_check_launch();
}
