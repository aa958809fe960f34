// 5x5 conv weight gradient straight from the channels-last activation grids (gfx950, MI355X).
//
//   dW[a][tap][b] (+)= sum over cells  A[cell][a] * B[cell + tap shift][b]       (reference: the autograd of nn.Conv2d,
//                                                                                src/model/ipa_pytorch_dynamic.py:669-690)
//
// Both operands of this contraction are stored with the REDUCTION index (the cell) as the slow axis and the channel as the
// fast one ([window][frame + 4][residue + 4][C] grids), whereas v_mfma_f32_32x32x16_bf16 wants 8 consecutive k values of
// one row / column per lane.  The engine of gemm_bf16.hip therefore runs this product on transposed, column-shifted COPIES
// of the grids (dfold_grid_transpose_shift: 6 copies per weight gradient).  This kernel needs no copies: K tiles of
// 64 cells of the grids (all channels of the tile per cell) go HBM -> LDS as they lie (LDS-DMA, 16-byte chunks, rows stay channel-contiguous) and the
// MFMA fragments are read with ds_read_b64_tr_b16, the LDS transpose read of gfx950: a 16-lane group fetches a
// [4 cells][16 channels] block and every lane receives the 4 cells of ITS channel.  Two such reads make one 8-deep
// fragment.  Which 8 cells of a K16 block a lane holds is a free choice as long as the A and the B fragment agree (a
// permutation of the reduction index): lanes 0-31 take cells 0-3 and 4-7, lanes 32-63 cells 8-11 and 12-15.
//
// Work unit: one workgroup per (256-channel A tile, frame shift z0, 64-channel B tile) computes ALL FIVE residue shifts z1 of
// its tap row: the five shifted B operands are the same 64 + 4 consecutive cells read 0..4 rows further down, so the B tile
// of a K step is a 68-cell x 64-channel "halo" tile (8.5 KiB instead of 5 x 8 KiB) and the 256 x 320 accumulator tile is
// 256 A channels x (5 shifts x 64 B channels).  8 waves as 4 (M) x 2 (N), 2 x 5 MFMA tiles per wave: rows wm*32 + i*128
// (interleaved with the neighbours: the chunk-index bits that the XOR key touches are then the same for both tiles, one
// address register + immediates), columns = shift j, channels wn*32..+32 of the B tile.
//
// LDS image, three stages of 41 KiB: A tile [64 cells][256 ch] (512-byte rows), B tile [68 (+4 pad) cells][64 ch] (128-byte
// rows).  One transpose read touches, per 32-lane half, 4 cells x 64 bytes: the 16-byte chunks of a row are XOR-permuted on
// the DMA *source* side (the LDS-DMA image itself is lane-linear) so that those 4 x 64 bytes fall into four different
// 64-byte bank groups:   A: chunk ^= (cell & 3) << 2        B (rows 128 bytes apart): chunk ^= ((cell >> 1) & 1) << 2
// (any 4 consecutive cells, whatever the shift, cover the four (parity, key) combinations).
//
// K pipeline: the DMA of tile s + 2 is issued during step s (6 pieces per wave: 4 A, the wave's B piece, the shared halo
// piece), the wait in front of step s + 1 leaves exactly those 6 outstanding (s_waitcnt vmcnt(6): loads retire in order), so
// a tile has two full K steps to arrive.  The second wave of every SIMD runs half a K step behind the first.  Workgroup ids
// are XCD-aware with the m tile fastest: the workgroups of one XCD share few B streams and the five A streams.
#include "dfold_common.h"
#include "../../include/dfold_hip.h"

#define TBK 64
#define TBM 256
#define TBC 64                         // B channels per workgroup (x 5 residue shifts = 320 accumulator columns)
#define TA_PITCH (TBM * 2)            // bytes per cell row of the A tile
#define TB_PITCH (TBC * 2)
#define TA_BYTES (TBK * TA_PITCH)     // 32 KiB
#define TB_ROWS 68                    // 64 cells + 4 halo cells
#define TB_BYTES (9 * 1024)           // 9 DMA pieces of 8 rows (rows 68..71 are padding)
#define TSTAGE (TA_BYTES + TB_BYTES)  // 41 KiB
#define TNSTAGE 3
#define TNJ 5

typedef __attribute__((address_space(3))) void* tn_lds_ptr_t;
typedef __attribute__((address_space(3))) char tn_lchar;   // LDS byte pointer: 32-bit address arithmetic
typedef __attribute__((ext_vector_type(4))) short tn_s16x4;
typedef __attribute__((ext_vector_type(8))) short tn_s16x8;

struct WgradTnParams {
  const char* A;       // unshifted operand: first cell of the reduction (window 0, first frame row, residue 0), channel 0
  const char* B;       // shifted operand: the cell that pairs with it for shift (0, 0)
  float* C;            // accumulators [CA][25][CB]
  int CA, CB;          // channels = cell pitch of the two grids (elements)
  long rowA, rowB;     // bytes from one frame row of the grid to the next
  long winA, winB;     // bytes from one window to the next
  int nchunk, nF, nW;  // K walk: 64-cell chunks per frame row, frame rows, windows
  int flip, accumulate;
  const int* nz_ps;    // NZ: frame flags of the gradient operand (prefix sums [window][Fp + 1]), see dfold_conv_wgrad_tn
  int nz_radius, nz_f0, Fp;   // nz_f0: padded frame row of the gradient cell that K row (w, 0) pairs with at frame shift 0
};

// One MFMA operand fragment: cells (k) 0-3 and 4-7 of this lane's K-half, its own channel.  The reads are inline assembly on
// purpose: the compiler orders every LDS load it can see behind ALL outstanding LDS-DMA transfers (s_waitcnt vmcnt(0) in
// front of the first read after a global_load_lds -- it cannot tell the two stages apart), which would park the waves on
// the next tile's HBM latency in the middle of every K step (measured: 1.94 ms per launch against 1.74 ms for the copy
// form).  The price: the compiler does not count these reads either, so the lgkmcnt waits below are written by hand
// (TN_WAIT names the fragment registers it guards as read-write operands, which pins the MFMAs behind it).
template <int OFF, int PITCH>
__device__ __forceinline__ bf16x8 tn_frag(unsigned base) {
  tn_s16x4 lo, hi;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(base), "n"(OFF));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(base), "n"(OFF + 4 * PITCH));
  const tn_s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return *(const bf16x8*)&v;
}
// the 7 fragments (2 A row tiles, 5 residue shifts of the B tile) of K16 block KB into one register set
template <int KB>
__device__ __forceinline__ void tn_ldfrag(bf16x8 (&af)[2], bf16x8 (&bfr)[TNJ], unsigned ba, const unsigned (&bb)[TNJ]) {
  af[0] = tn_frag<KB * 16 * TA_PITCH, TA_PITCH>(ba);
  af[1] = tn_frag<KB * 16 * TA_PITCH + 256, TA_PITCH>(ba);
  bfr[0] = tn_frag<KB * 16 * TB_PITCH, TB_PITCH>(bb[0]);
  bfr[1] = tn_frag<KB * 16 * TB_PITCH, TB_PITCH>(bb[1]);
  bfr[2] = tn_frag<KB * 16 * TB_PITCH, TB_PITCH>(bb[2]);
  bfr[3] = tn_frag<KB * 16 * TB_PITCH, TB_PITCH>(bb[3]);
  bfr[4] = tn_frag<KB * 16 * TB_PITCH, TB_PITCH>(bb[4]);
}
// MFMA / LDS-DMA interleave of a block of 10 MFMAs that carries N DMA pieces: (2 MFMA, 1 DMA) pairs, then the rest
template <int N>
__device__ __forceinline__ void tn_pin() {
#pragma unroll
  for (int k = 0; k < N; ++k) {
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
    __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
  }
  if (2 * TNJ - 2 * N > 0) __builtin_amdgcn_sched_group_barrier(0x008, 2 * TNJ - 2 * N, 0);
}
// LDS reads return in order: "at most N outstanding" = everything but the youngest N has landed
#define TN_WAIT(N, set)                                                                                              \
  asm volatile("s_waitcnt lgkmcnt(" #N ")"                                                                           \
               : "+v"(af[set][0]), "+v"(af[set][1]), "+v"(bfr[set][0]), "+v"(bfr[set][1]), "+v"(bfr[set][2]),        \
                 "+v"(bfr[set][3]), "+v"(bfr[set][4]))

// NZ = true (round 6, zero-frame skipping): the K walk visits only the frame rows whose gradient cells can be non-zero by the
// frame flags.  Every wave builds the same per-window 64-bit masks with ballots (lane = frame) and walks them on the scalar unit
// (lowest set bit = next frame row; the masks of the following windows wait in a shift register of SGPR pairs) -- no memory
// operation inside the hand-counted K loop.  The rows left out contribute exact zeros: the sum is unchanged bit for bit.  With
// the flipped operand order the gradient frame of a K row depends on the workgroup's frame shift z0, so workgroups of different
// shifts skip different rows (their counts differ by at most the shift at the edges of the non-zero band).
template <bool NZ>
__global__ __launch_bounds__(512, 2) void conv_wgrad_tn_kernel(const WgradTnParams p) {
  extern __shared__ __attribute__((aligned(16))) char tl[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 1, wn = w & 1;
  // XCD-aware workgroup id (bijective for any grid size), then (m tile fastest, frame shift, B channel tile)
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q = nwg >> 3, r8 = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  const int lid = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + idx;
  const int tiles_m = p.CA / TBM;
  int r = lid;
  const int m0 = (r % tiles_m) * TBM;
  r /= tiles_m;
  const int z0 = r % 5;   // frame shift of the B operand
  const int n0 = (r / 5) * TBC;
  const long pitchA = (long)p.CA * 2, pitchB = (long)p.CB * 2;
  const char* pa = p.A + (long)m0 * 2;
  const char* pb = p.B + (long)z0 * p.rowB + (long)n0 * 2;

  // ---- staging: A pieces j = t*8 + w hold cells 2j, 2j+1 (32 chunks each); B piece w holds cells 8w..8w+7 (8 chunks each),
  // B piece 8 (issued by every wave: same bytes, same place) the halo cells 64..67 (+ 4 padding rows that re-read cell 67).
  // LDS position (cell, physical chunk pc) receives logical chunk pc ^ key(cell).
  unsigned aoff0, boff0, boff8;  // A: piece t of this wave lies 16 cells below piece t - 1 (a scalar step on the pointer)
  {
    const int cell = 2 * w + (lane >> 5);
    const int lc = (lane & 31) ^ ((cell & 3) << 2);
    aoff0 = (unsigned)(cell * pitchA + lc * 16);
    const int cb = 8 * w + (lane >> 3);
    boff0 = (unsigned)(cb * pitchB + (((lane & 7) ^ (((cb >> 1) & 1) << 2)) << 4));
    const int ch = 64 + (lane >> 3), chs = ch < TB_ROWS ? ch : TB_ROWS - 1;
    boff8 = (unsigned)(chs * pitchB + (((lane & 7) ^ (((ch >> 1) & 1) << 2)) << 4));
  }
  // K walk (window, frame row, 64-cell chunk), chunk fastest: byte deltas, branch-free on the scalar unit.  After the last
  // tile the pointers stay put: the surplus prefetches re-read it into a stage nobody reads any more.
  const long dA = TBK * pitchA, dB = TBK * pitchB;
  const long eA1 = p.rowA - (long)p.nchunk * dA, eB1 = p.rowB - (long)p.nchunk * dB;
  const long eA2 = p.winA - (long)p.nF * p.rowA, eB2 = p.winB - (long)p.nF * p.rowB;
  int nsteps = p.nchunk * p.nF * p.nW;
  // NZ: live frame rows of the current window (bit f) and of the windows behind it; wa / wb: frame row 0 of the current window
  unsigned long cur = 0, q1 = 0, q2 = 0, q3 = 0, q4 = 0, q5 = 0, q6 = 0, q7 = 0;
  const char *wa = pa, *wb = pb;
  auto next_row = [&]() {                // (a live row remains: the caller counted)
    while (cur == 0) {
      cur = q1; q1 = q2; q2 = q3; q3 = q4; q4 = q5; q5 = q6; q6 = q7; q7 = 0;
      wa += p.winA;
      wb += p.winB;
    }
    const int f = __builtin_ctzl(cur);
    cur &= cur - 1;
    pa = wa + (long)f * p.rowA;
    pb = wb + (long)f * p.rowB;
  };
  if (NZ) {
    unsigned long m[8];
    int rows = 0;
#pragma unroll
    for (int ww = 0; ww < 8; ++ww) {
      bool lv = false;
      if (ww < p.nW && lane < p.nF) {
        const int fr = p.nz_f0 + lane + (p.flip ? z0 : 0);
        int a = fr - p.nz_radius, b = fr + p.nz_radius;
        a = a < 0 ? 0 : a;
        b = b > p.Fp - 1 ? p.Fp - 1 : b;
        const int* row = p.nz_ps + (long)ww * (p.Fp + 1);
        lv = row[b + 1] - row[a] > 0;
      }
      m[ww] = __ballot(lv);
      rows += __builtin_popcountl(m[ww]);
    }
    cur = m[0]; q1 = m[1]; q2 = m[2]; q3 = m[3]; q4 = m[4]; q5 = m[5]; q6 = m[6]; q7 = m[7];
    nsteps = p.nchunk * rows;
    if (nsteps == 0 && p.accumulate) return;      // nothing to add
    if (nsteps > 0) next_row();
  }
  int st_c = 0, st_f = 0, st_left = nsteps;
  const char* sa_keep = pa;
  auto stage_1 = [&](int soff) {         // advance the K walk; the two B pieces and the first A piece of the tile
    sa_keep = pa;
    const char* sb = pb;
    if (NZ) {
      if (st_left > 1) {                 // a tile after this one exists
        --st_left;
        const int c = st_c + 1;
        if (c < p.nchunk) {
          st_c = c;
          pa += dA;
          pb += dB;
        } else {
          st_c = 0;
          next_row();
        }
      }
    } else {
    const unsigned adv = (unsigned)(1 - st_left) >> 31;        // a tile after this one exists
    st_left -= (int)adv;
    int c = st_c + 1;
    const unsigned w0 = (unsigned)(p.nchunk - 1 - c) >> 31;    // frame row finished
    c &= (int)(w0 - 1u);
    int f = st_f + (int)w0;
    const unsigned w1 = (unsigned)(p.nF - 1 - f) >> 31;        // window finished
    f &= (int)(w1 - 1u);
    st_c = c;
    st_f = f;
    const long k0 = -(long)w0, k1 = -(long)w1, ka = -(long)adv;
    pa += (dA + (eA1 & k0) + (eA2 & k1)) & ka;
    pb += (dB + (eB1 & k0) + (eB2 & k1)) & ka;
    }
    char* la = tl + soff;
    char* lb = la + TA_BYTES;
    __builtin_amdgcn_global_load_lds((const void*)(sb + boff0), (tn_lds_ptr_t)(lb + w * 1024), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const void*)(sb + boff8), (tn_lds_ptr_t)(lb + 8 * 1024), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const void*)(sa_keep + aoff0), (tn_lds_ptr_t)(la + w * 1024), 16, 0, 0);
  };
  auto stage_2 = [&](int soff) {         // the other three A pieces
    char* la = tl + soff;
#pragma unroll
    for (int t = 1; t < 4; ++t)
      __builtin_amdgcn_global_load_lds((const void*)(sa_keep + t * 16 * pitchA + aoff0), (tn_lds_ptr_t)(la + (t * 8 + w) * 1024), 16, 0, 0);
  };

  f32x16 acc[2][TNJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TNJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // ---- fragment addresses: 16-lane group g reads the [4 cells][16 channels] block of K-half g >> 1, channel half g & 1;
  // lane p16 of the group points at cell p16 >> 2, channel quad p16 & 3 of that block (8 bytes) and receives channel p16.
  // A tile i of the wave = rows (wm + 4i) * 32 (chunks wm*4 + 16i + ..); B fragment j = the wave's 32 channels (chunks wn*4 + ..)
  // read j cells further down.  fa / fb hold the addresses for the stage being read and move on by one stage per K step.
  const int p16 = lane & 15, g = lane >> 4;
  const int cell_l = (g >> 1) * 8 + (p16 >> 2);
  const int c0 = (g & 1) * 2 + ((p16 >> 1) & 1);
  const unsigned tls = (unsigned)(uintptr_t)(tn_lchar*)tl;     // LDS byte address of the dynamic region
  unsigned fa = tls + cell_l * TA_PITCH + (((wm * 4 + c0) ^ ((p16 >> 2) << 2)) << 4) + (p16 & 1) * 8;
  unsigned fb[TNJ];
#pragma unroll
  for (int j = 0; j < TNJ; ++j) {
    const int cell = cell_l + j;
    fb[j] = tls + TA_BYTES + cell * TB_PITCH + (((wn * 4 + c0) ^ (((cell >> 1) & 1) << 2)) << 4) + (p16 & 1) * 8;
  }
  bf16x8 af[2][2], bfr[2][TNJ];
  auto mma = [&](int set) {
#pragma unroll
    for (int j = 0; j < TNJ; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[set][i], bfr[set][j], acc[i][j], 0, 0, 0);
  };
  int rd = 0, wr = 2 * TSTAGE;          // byte offsets of the stage being read / the stage the next prefetch goes to
  auto next_stage = [&]() {             // called once per K step by every wave
    const int wrap = rd == (TNSTAGE - 1) * TSTAGE;
    const int d = wrap ? -(TNSTAGE - 1) * TSTAGE : TSTAGE;
    rd += d;
    fa += d;
#pragma unroll
    for (int j = 0; j < TNJ; ++j) fb[j] += d;
    wr = wr == (TNSTAGE - 1) * TSTAGE ? 0 : wr + TSTAGE;
  };

  if (!NZ || nsteps > 0) {
  stage_1(0);
  stage_2(0);
  stage_1(TSTAGE);
  stage_2(TSTAGE);
  if (w >= 4) __builtin_amdgcn_s_setprio(1);
  if (w < 4) {
    // group A (one wave per SIMD): per K step [28 fragment reads][4 x 10 MFMAs], the 6 DMA pieces of tile s + 2 between the
    // first 20 MFMAs, the reads of K16 blocks 2 / 3 behind the MFMAs that free their register set
    for (int s = 0; s < nsteps; ++s) {
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // 6 = DMA pieces per wave and K step
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      tn_ldfrag<0>(af[0], bfr[0], fa, fb);
      tn_ldfrag<1>(af[1], bfr[1], fa, fb);
      TN_WAIT(14, 0);
      stage_1(wr);
      mma(0);
      tn_pin<3>();
      tn_ldfrag<2>(af[0], bfr[0], fa, fb);
      TN_WAIT(14, 1);
      stage_2(wr);
      mma(1);
      tn_pin<3>();
      tn_ldfrag<3>(af[1], bfr[1], fa, fb);
      TN_WAIT(14, 0);
      mma(0);
      TN_WAIT(0, 1);
      mma(1);
      next_stage();
    }
  } else {
    // group B (the second wave of every SIMD) runs half a K step behind: it issues the second half of the previous tile's
    // MFMAs while group A reads its fragments, and reads the current tile after A
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // 6 = DMA pieces per wave and K step
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    tn_ldfrag<0>(af[0], bfr[0], fa, fb);
    tn_ldfrag<1>(af[1], bfr[1], fa, fb);
    TN_WAIT(14, 0);
    stage_1(wr);
    mma(0);
    tn_pin<3>();
    tn_ldfrag<2>(af[0], bfr[0], fa, fb);
    TN_WAIT(14, 1);
    stage_2(wr);
    mma(1);
    tn_pin<3>();
    tn_ldfrag<3>(af[1], bfr[1], fa, fb);
    next_stage();
    for (int s = 1; s < nsteps; ++s) {
      // the previous tile's fragment reads still in flight (tn_ldfrag<2>, <3>) read the stage that group A's LDS-DMA of
      // this step overwrites: they must have left the LDS before the barrier (ordered by a wait, not by latency)
      asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");   // 6 = DMA pieces per wave and K step
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      TN_WAIT(14, 0);
      stage_1(wr);
      mma(0);                 // previous tile, K16 block 2
      tn_pin<3>();
      TN_WAIT(0, 1);
      stage_2(wr);
      mma(1);                 // previous tile, K16 block 3
      tn_pin<3>();
      tn_ldfrag<0>(af[0], bfr[0], fa, fb);
      tn_ldfrag<1>(af[1], bfr[1], fa, fb);
      TN_WAIT(14, 0);
      mma(0);
      tn_ldfrag<2>(af[0], bfr[0], fa, fb);
      TN_WAIT(14, 1);
      mma(1);
      tn_ldfrag<3>(af[1], bfr[1], fa, fb);
      next_stage();
    }
    TN_WAIT(14, 0);
    mma(0);
    TN_WAIT(0, 1);
    mma(1);
  }
  }   // nsteps > 0

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the surplus prefetches must have landed before the LDS allocation goes away
  // ---- epilogue: fp32 accumulators [CA][25][CB]; every element belongs to exactly one workgroup (plain read-modify-write)
  const int frow = lane & 31, fhalf = lane >> 5;
  const long ldc = 25L * p.CB;
  const long tstep = p.flip ? -(long)p.CB : (long)p.CB;       // residue shift j -> tap 5 z0 + j, or 24 - (5 z0 + j)
  float* cb = p.C + (long)(p.flip ? 24 - 5 * z0 : 5 * z0) * p.CB + n0 + wn * 32 + frow;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const long m = (long)m0 + wm * 32 + i * 128 + (e & 3) + 8 * (e >> 2) + 4 * fhalf;
      float* row = cb + m * ldc;
      float cv[TNJ];
#pragma unroll
      for (int j = 0; j < TNJ; ++j) cv[j] = p.accumulate ? row[j * tstep] : 0.f;
#pragma unroll
      for (int j = 0; j < TNJ; ++j) row[j * tstep] = acc[i][j][e] + cv[j];
    }
  }
}

extern "C" int dfold_conv_wgrad_tn(const void* a_grid, const void* b_grid, float* dwg, int32_t CA, int32_t CB, int32_t W,
                                   int32_t Fp, int32_t Wp, int32_t N, int32_t f0, int32_t nf, int32_t flip,
                                   int32_t accumulate, const int32_t* nz_ps, int32_t nz_radius, void* stream) {
  if (!a_grid || !b_grid || !dwg) return DFOLD_EINVAL;
  if (CA <= 0 || CB <= 0 || (CA % TBM) || (CB % TBC) || W <= 0 || N <= 0 || N + 4 > Wp) return DFOLD_EINVAL;
  // N_res that is not a multiple of the 64-cell K chunk: the frame range of a window is walked as ONE line of nf * Wp cells (pad
  // columns included: the gradient grid is zero there, and behind its last frame) in ceil(nf * Wp / 64) chunks.  The last chunk
  // of the last window reads up to 67 cells past the end of the shifted operand: the caller keeps them readable and finite.
  const bool linear = (N % TBK) != 0;
  if (linear && nz_ps) return DFOLD_EINVAL;
  if (f0 < 0 || nf <= 0 || f0 + nf + 4 > Fp) return DFOLD_EINVAL;
  if (((uintptr_t)a_grid | (uintptr_t)b_grid) & 15) return DFOLD_EINVAL;
  if ((long)(TBK + 8) * CA * 2 >= (1L << 31) || (long)(TBK + 8) * CB * 2 >= (1L << 31)) return DFOLD_EINVAL;   // 32-bit lane offsets
  WgradTnParams p;
  p.A = (const char*)a_grid + (((long)(2 + f0) * Wp + 2) * CA) * 2;
  p.B = (const char*)b_grid + ((long)f0 * Wp * CB) * 2;
  p.C = dwg;
  p.CA = CA; p.CB = CB;
  p.rowA = (long)Wp * CA * 2; p.rowB = (long)Wp * CB * 2;
  p.winA = (long)Fp * p.rowA; p.winB = (long)Fp * p.rowB;
  p.nchunk = linear ? (nf * Wp + TBK - 1) / TBK : N / TBK; p.nF = linear ? 1 : nf; p.nW = W;
  p.flip = flip ? 1 : 0; p.accumulate = accumulate ? 1 : 0;
  p.nz_ps = nz_ps; p.nz_radius = nz_radius; p.nz_f0 = flip ? f0 : f0 + 2; p.Fp = Fp;
  const unsigned nwg = (unsigned)((CA / TBM) * 5 * (CB / TBC));
  if (nz_ps) {
    if (nz_radius < 0 || nf > 64 || W > 8) return DFOLD_EINVAL;
    DFOLD_MAX_LDS_ONCE(conv_wgrad_tn_kernel<true>, TNSTAGE * TSTAGE);
    DFOLD_LAUNCH(conv_wgrad_tn_kernel<true>, dim3(nwg), dim3(512), (size_t)(TNSTAGE * TSTAGE), (hipStream_t)stream, p);
  } else {
    DFOLD_MAX_LDS_ONCE(conv_wgrad_tn_kernel<false>, TNSTAGE * TSTAGE);
    DFOLD_LAUNCH(conv_wgrad_tn_kernel<false>, dim3(nwg), dim3(512), (size_t)(TNSTAGE * TSTAGE), (hipStream_t)stream, p);
  }
  return dfold_check_launch();
}
