// 5x5 conv implicit GEMM (forward and data gradient), one wave per SIMD (gfx950, MI355X).
//
//   out[cell, co] = epilogue( sum over (df, dn, ci)  x[cell + (df, dn), ci] * W[co, (df, dn), ci] )
//   (reference: ConvNet, src/model/ipa_pytorch_dynamic.py:664-706; the data gradient is the same product with the
//    tap-flipped transposed weights)
//
// Same operands, descriptor and epilogues as the 256 x 320 "halo" kernel of gemm_bf16.hip; what changes is how much of
// every byte that passes through the LDS is turned into matrix work (that kernel's K loop loses a
// third of its matrix-core cycles to the issue cost of its LDS-DMA pieces and fragment reads):
//
//   * tile 512 (cells) x 160 (output channels), FOUR waves, each 128 x 160 = 4 x 5 MFMA 32x32x16 tiles: 9 fragment reads
//     per 20 MFMAs (0.45 per MFMA; the 64 x 160 wave tile: 0.7).  The 320 accumulator registers of a wave are placed by
//     hand: 16 tiles in the 256 AGPRs, 4 tiles in VGPRs (inline-assembly MFMAs with "a" / "v" register-class constraints;
//     left to itself hipcc shuttles accumulators between the two files inside the K loop);
//   * the M tile is TWO runs of 256 consecutive residues (two frame rows at N_res 256): the weight tile of a K step is
//     shared by twice as many cells as in the 256-row tile -- 33.2 instead of 46.6 KB of LDS-DMA per 64 channels of a tap;
//   * K steps of 32 channels: a (64-channel chunk, frame tap, channel half) GROUP stages the activations once as two
//     272-row x 64-byte halo tiles which the five residue taps read shifted by one row each; weights [160][32] per step
//     in a ring of three stages, prefetched two steps ahead with a counted s_waitcnt vmcnt.
//
// One wave per SIMD has no partner wave to cover its stalls, so the loop is software-pipelined across the barrier: a step
// is [MFMAs of K16 block 0 | fragment reads of block 1] -- barrier -- [MFMAs of block 1 | reads of the NEXT tile's block 0],
// the first instruction behind every barrier is an MFMA whose operands are already in registers.  LDS-DMA pieces and
// fragment reads are inline assembly placed one per MFMA gap (the compiler neither reorders them nor sees them: waits
// are written by hand).
//
// Accumulation order per output element: (chunk, df, half, dn, k).  The 256 x 320 kernel sums (chunk, df, dn, k): results
// of the two kernels differ by fp32 summation order (bit-identical for one kernel across tile positions and launches).
#include "gemm_engine.h"
#include <stdlib.h>
#include <type_traits>

#define W4_BM 512
#define W4_BN 160
#define W4_HROWS 272                          // rows of one run in a halo buffer: 17 LDS-DMA pieces of 16 rows x 64 B
#define W4_HRUN_BYTES (W4_HROWS * 64)
#define W4_HPIECES 34                         // pieces per halo buffer (2 runs)
#define W4_HALO_BYTES (2 * W4_HRUN_BYTES)     // 34 KiB
#define W4_BPIECES 10                         // pieces per weight tile: 160 rows x 64 B
#define W4_BT_BYTES (W4_BN * 64)
#define W4_NB 3
#define W4_LDS_BYTES (2 * W4_HALO_BYTES + W4_NB * W4_BT_BYTES)   // 98 KiB

typedef __attribute__((address_space(3))) char w4_lchar;

__device__ __forceinline__ void w4_dma(const char* base, unsigned voff, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(lds_addr) : "memory", "m0");
}
#define W4_READ(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF))
#define W4_MFMA_A(acc, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))
#define W4_MFMA_V(acc, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))

// SK = false: one workgroup per (tile, split-K part).  SK = true ("stream-K", the thin launches: dependency cones, small eval
// windows): a PERSISTENT grid of one workgroup per CU; the launch's work is the sequence of (tile, K group) units in tile-major
// order, workgroup s takes units [s per, (s + 1) per) -- a run of pieces (tile, [g_begin, g_end)).  A piece that covers its
// whole tile goes straight to the epilogue; the pieces of a shared tile park their fp32 partial tiles and the last one to arrive
// adds them in the fixed order of the workgroup ids (deterministic: the association of the sums depends on the launch shape
// only) -- the chip is busy for ceil(tiles x groups / CUs) group times instead of whole rounds of whole (or 1/S) tiles.
//
// NZ = true (round 6, zero-frame skipping; dfold_gemm_desc.nz_ps): before anything else a workgroup asks the frame flags whether
// any of the input frame rows of its two runs can hold a non-zero.  If not, the K walk is left out: the accumulators stay zero
// and the epilogue runs as it would have (the products left out are exact zeros -- same bits).  The data-gradient launches of
// the tower's backward meet gradients that are zero outside the dependency cone of the frames the loss read (a quarter of the
// tiles at 32 frames when the loss reads the last frame); nothing about the loss is known to the host, the decision is made
// per tile on the device.
template <bool SK, bool NZ = false>
__global__ __launch_bounds__(256, 1) void dfold_conv_w4_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char wl[];
  __shared__ int s_last;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-aware tile id (bijective for any grid size); the n tiles of one m tile are consecutive ids = one XCD
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  const int tiles_n = p.N / W4_BN;
  const int M = p.M;
  const int ngroups = (p.nseg / 25) * 10;       // K groups of a whole tile (of this split-K part)
  int u = 0, u_end = 1;                         // SK: this workgroup's unit range
  if (SK) {
    u = lid * p.sk_per;
    const int U = p.sk_tiles * ngroups;
    u_end = u + p.sk_per < U ? u + p.sk_per : U;
    if (u >= u_end) return;
  }
  bool first_piece = true;
  for (;;) {
  const int tile = SK ? u / ngroups : lid;
  int g_begin = SK ? u - tile * ngroups : 0;
  int g_end = SK ? (g_begin + (u_end - u) < ngroups ? g_begin + (u_end - u) : ngroups) : ngroups;
  const int m0 = (tile / tiles_n) * W4_BM, n0 = (tile % tiles_n) * W4_BN;
  bool live = true;
  int S_eff = (int)gridDim.y;       // parts per tile of a split-K launch
  if (NZ) {
    const int fp1 = p.am.fp + 1;
    // can the 256-row run that starts at GEMM row m read a non-zero?  (two loads of the window's frame prefix sums)
    auto run_live = [&](int m) -> bool {
      if (m >= M) return false;
      unsigned ww, ff, fl;          // window, first and last logical frame of the run's output cells
      if (p.am.mode == 2) {
        const unsigned vw = row_vw(p.am), v = (unsigned)m - ((unsigned)m / vw) * vw;
        ww = (unsigned)m / vw;
        ff = v / (unsigned)p.am.wp;
        fl = (v + 255u) / (unsigned)p.am.wp;
        if (ff >= (unsigned)p.am.f) return false;          // a run behind the last frame: nothing of it is stored
        fl = fl < (unsigned)p.am.f ? fl : (unsigned)p.am.f - 1u;
      } else {
        const unsigned wf = (unsigned)m / (unsigned)p.am.n;
        ww = wf / (unsigned)p.am.f;
        ff = fl = wf - ww * (unsigned)p.am.f;
      }
      int a = p.nz_f0 + (int)ff - p.nz_radius, b = p.nz_f0 + (int)fl + 4 + p.nz_radius;
      a = a < 0 ? 0 : a;
      b = b > p.am.fp - 1 ? p.am.fp - 1 : b;
      const int* row = p.nz_ps + (long)ww * fp1;
      return row[b + 1] - row[a] > 0;
    };
    live = run_live(m0) || run_live(m0 + 256);
    if (!live && (blockIdx.y != 0 || (p.flags & DFOLD_GEMM_NZ_KEEP))) return;     // (split-K: part 0 alone writes the zero tile -- unless
                                                                                   //  the caller says it is there already)
    if (!SK && p.sk_per == -1) {
      // ---- split factor chosen on the DEVICE from the flags (round 6): the launch carries S_max parts per tile; how many of
      // them walk K is decided here, the same way by every workgroup of a window.  The live tiles of a skipped launch run as
      // whole rounds on the CUs (64 .. 576 live tiles of 512 / 1024: 1 .. 3 rounds for 0.25 .. 2.25 rounds of work); L live
      // tiles x S parts spread evenly.  L is estimated from the tile rows of this workgroup's own window (all windows alike:
      // the loss reads the same frames of each) -- a function of the data alone, so the fp32 association, too, is the same
      // in every step that meets the same gradient pattern.  Cost model = ops.conv_splitk's.
      const unsigned rows_w = p.am.mode == 2 ? row_vw(p.am) : (unsigned)p.am.n * (unsigned)p.am.f;
      S_eff = 1;
      if (rows_w % W4_BM == 0) {
        const unsigned tpw = rows_w / W4_BM, w_first = ((unsigned)m0 / rows_w) * rows_w;
        int cnt = 0;
        for (unsigned j0 = 0; j0 < tpw; j0 += 64) {
          const unsigned j = j0 + (unsigned)lane;
          const bool lv = j < tpw && (run_live((int)(w_first + j * W4_BM)) || run_live((int)(w_first + j * W4_BM + 256)));
          cnt += __builtin_popcountll(__ballot(lv));
        }
        cnt = __builtin_amdgcn_readfirstlane(cnt);
        const int L = cnt * (int)((unsigned)M / rows_w) * tiles_n, n_cu = p.sk_tiles, chunks = p.nseg / 25, steps = 25 * chunks;
        int best_cost = ((L + n_cu - 1) / n_cu) * (steps + 16);
        for (int S = 2; S <= (int)gridDim.y; ++S) {
          // (measured, same-box A/B of the step: splitting pays where all parts of all live tiles run at once -- 64 live tiles
          //  0.79 -> 0.50 ms, 128: 0.80 -> 0.64 -- and loses a few per cent beyond one round, where the parts of a tile sit at
          //  different K positions next to each other and the weight slabs of 4 - 5 K ranges share the L2: 320 live tiles
          //  1.00 -> 1.08 ms, 192 of the wider launch 0.83 -> 0.97)
          if (S == 3 || chunks % S || L * S > n_cu) continue;
          const int cost = ((L * S + n_cu - 1) / n_cu) * (steps / S + 16 + 2 * S);
          if (10 * cost < 9 * best_cost) {
            best_cost = cost;
            S_eff = S;
          }
        }
      }
      if ((int)blockIdx.y >= S_eff) return;
      const int gpp = ngroups / S_eff;
      g_begin = (int)blockIdx.y * gpp;
      g_end = g_begin + gpp;
    }
  }
  // (split-K launches: part blockIdx.y walks its own range of channel chunks -- p.nseg, p.sa0, p.sb0 are per part)
  const char* A = (const char*)p.A + (long)blockIdx.y * p.sa0 * 2;
  const char* B = (const char*)p.B + (long)blockIdx.y * p.sb0 * 2;

  // ---- K walk: groups (chunk c, frame tap df, channel half h), h fastest; five residue taps dn inside a group ----
  const long a_s0 = p.a_seg_s0, a_s1 = p.a_seg_s1, b_s0 = p.b_seg_s0, b_s1 = p.b_seg_s1;
  const long a_0 = p.a_seg0, b_0 = p.b_seg0;
  const unsigned b_dn2 = (unsigned)(p.b_seg_s2 * 2);        // bytes from one residue tap of the weights to the next
  auto grp_a = [&](int g) -> const char* {
    g = g < g_end - 1 ? g : g_end - 1;
    const unsigned h = (unsigned)g & 1u, t = (unsigned)g >> 1, c = t / 5u, df = t - 5u * c;
    return A + (a_0 + (long)c * a_s0 + (long)df * a_s1 + (long)h * 32) * 2;
  };
  auto grp_b = [&](int g) -> const char* {
    g = g < g_end - 1 ? g : g_end - 1;
    const unsigned h = (unsigned)g & 1u, t = (unsigned)g >> 1, c = t / 5u, df = t - 5u * c;
    return B + (b_0 + (long)c * b_s0 + (long)df * b_s1 + (long)h * 32) * 2;
  };

  // (SK: the lane-dependent addresses below are recomputed per piece from an opaque copy of the lane id -- hoisted out of the
  //  piece loop they would stay live across the epilogue, where the 320 accumulator registers leave no room: 376 B of scratch)
  int lane_p = lane;
  if (SK) asm volatile("" : "+v"(lane_p));
  // ---- LDS-DMA lane offsets.  A piece is 16 rows x 64 B (4 chunks of 16 B): lane l fills (row l >> 2, physical chunk l & 3)
  // with the logical chunk (l & 3) ^ key(row), key(row) = (row >> 2) & 3 = (l >> 4) & 3 (every piece starts at a multiple of
  // 16 rows).  Halo run r starts at the tap (0, 0) corner of GEMM row m0 + 256 r (a run past M repeats run 0: computed,
  // never stored); rows 260 .. 271 of a run are padding (they re-read row 259 .. never read back).
  const unsigned ld2 = (unsigned)(p.am.ld * 2);
  const unsigned ldb2 = (unsigned)(p.ldb * 2);
  const unsigned lch = (unsigned)(((lane_p & 3) ^ ((lane_p >> 4) & 3)) << 4);
  const unsigned hl_norm = (unsigned)(lane_p >> 2) * ld2 + lch;
  const unsigned hl_last = (unsigned)((lane_p >> 2) < 3 ? (lane_p >> 2) : 3) * ld2 + lch;
  const unsigned b_lane = (unsigned)(lane_p >> 2) * ldb2 + lch;
  unsigned hrow[2];
  hrow[0] = (unsigned)(row_off(p.am, m0) * 2);
  hrow[1] = (unsigned)(row_off(p.am, m0 + 256 < M ? m0 + 256 : m0) * 2);
  const char* Bn = B + (long)n0 * p.ldb * 2;      // (folded into grp_b below through this delta)
  const long bn_delta = Bn - B;
  const unsigned lds0 = (unsigned)(uintptr_t)(w4_lchar*)wl;
  const unsigned lds_b = lds0 + 2 * W4_HALO_BYTES;

  auto halo_piece = [&](const char* gbase, int q, unsigned hbuf) {        // piece q of a group's halo tile -> buffer at hbuf
    const int r = q >= 17 ? 1 : 0, pr = q - 17 * r;
    const char* src = gbase + hrow[r] + (unsigned)pr * 16u * ld2;
    w4_dma(src, pr == 16 ? hl_last : hl_norm, hbuf + (unsigned)q * 1024u);
  };
  auto b_piece = [&](const char* tbase, int pp, unsigned stage) {           // piece pp of a weight tile
    w4_dma(tbase + bn_delta + (unsigned)pp * 16u * ldb2, b_lane, stage + (unsigned)pp * 1024u);
  };

  // ---- fragment read addresses (LDS byte addresses, relative to the halo buffer / the weight stage).  MFMA A operand:
  // lane_p -> (row lane_p & 31, k half lane_p >> 5); the wave's rows are run w >> 1, rows (w & 1) * 128 + i * 32 + frow + dn
  const int frow = lane_p & 31, fhalf = lane_p >> 5;
  unsigned a_lane[5][2], bf_lane[2];
#pragma unroll
  for (int dn = 0; dn < 5; ++dn)
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const int row = (w & 1) * 128 + frow + dn;
      a_lane[dn][kb] = (unsigned)((w >> 1) * W4_HRUN_BYTES + row * 64 + (((kb * 2 + fhalf) ^ ((row >> 2) & 3)) << 4));
    }
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) bf_lane[kb] = (unsigned)(frow * 64 + (((kb * 2 + fhalf) ^ ((frow >> 2) & 3)) << 4));

  f32x16 acc[2][2][5];        // [row pair][i][j]: rows w*128 + (2*pair + i)*32, columns j*32; j == 4 lives in VGPRs
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[a][i][j][e] = 0.f;
  bf16x8 fa[2][4], fb[2][5];   // [register set][fragment]

  // one block of 20 MFMAs on register set S; `fill(k)` is called in the gap behind MFMA k (k = 0 .. 19)
#define W4_BLOCK(S, FILL)                                                         \
  _Pragma("unroll") for (int k_ = 0; k_ < 20; ++k_) {                              \
    const int j_ = k_ >> 2, i_ = k_ & 3;                                          \
    if (j_ < 4) W4_MFMA_A(acc[i_ >> 1][i_ & 1][j_], fa[S][i_], fb[S][j_]);        \
    else W4_MFMA_V(acc[i_ >> 1][i_ & 1][j_], fa[S][i_], fb[S][j_]);               \
    FILL(k_);                                                                     \
  }
  // the 9 fragment reads of one K16 block into register set S: read number k (0 .. 8)
#define W4_FRAG(S, k, aaddr, baddr)                                 \
  do {                                                             \
    if ((k) == 0) W4_READ(fa[S][0], aaddr, 0);                      \
    if ((k) == 1) W4_READ(fb[S][0], baddr, 0);                      \
    if ((k) == 2) W4_READ(fa[S][1], aaddr, 2048);                   \
    if ((k) == 3) W4_READ(fa[S][2], aaddr, 4096);                   \
    if ((k) == 4) W4_READ(fa[S][3], aaddr, 6144);                   \
    if ((k) == 5) W4_READ(fb[S][1], baddr, 2048);                   \
    if ((k) == 6) W4_READ(fb[S][2], baddr, 4096);                   \
    if ((k) == 7) W4_READ(fb[S][3], baddr, 6144);                   \
    if ((k) == 8) W4_READ(fb[S][4], baddr, 8192);                   \
  } while (0)

  if (live) {
  // ---- prologue: halo tile of group 0, weight tiles 0 .. 2 ----
  const char* pa_n = grp_a(g_begin);      // halo source of the group being prefetched
  const char* pb_c = grp_b(g_begin);      // weight base of the current group (dn = 0)
  const char* pb_n = grp_b(g_begin + 1);  // ... of the next group
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int q = w + 4 * t;
    if (q < W4_HPIECES) halo_piece(pa_n, q, lds0);
  }
#pragma unroll
  for (int v = 0; v < 3; ++v)
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int pp = w + 4 * t;
      if (pp < W4_BPIECES) b_piece(pb_c + v * b_dn2, pp, lds_b + v * W4_BT_BYTES);
    }
  pa_n = grp_a(g_begin + 1);
  unsigned hb_c = lds0, hb_n = lds0 + W4_HALO_BYTES;      // halo buffer of the current / the prefetched group
  unsigned st_c = lds_b, st_1 = lds_b + W4_BT_BYTES, st_2 = lds_b + 2 * W4_BT_BYTES;   // stages of tiles u, u+1, u+2
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("s_barrier" ::: "memory");
  {
    const unsigned aa = a_lane[0][0] + hb_c, ba = bf_lane[0] + st_c;
#pragma unroll
    for (int k = 0; k < 9; ++k) W4_FRAG(0, k, aa, ba);
  }

  for (int g = g_begin; g < g_end; ++g) {
    // one group = five K32 steps (residue taps), fully unrolled: DN is a compile-time constant of each step
    auto step = [&](auto dnc) {
      constexpr int DN = decltype(dnc)::value;
      // ---- first half: MFMAs of K16 block 0 (set 0); reads of block 1 -> set 1; halo pieces of the next group ----
      {
        const unsigned aa = a_lane[DN][1] + hb_c, ba = bf_lane[1] + st_c;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        constexpr int HB0 = DN == 0 ? 0 : DN == 1 ? 9 : DN == 2 ? 18 : DN == 3 ? 26 : 34;
#define W4_FILL_Y(k)                                                              \
        do {                                                                      \
          if ((k) < 9) W4_FRAG(1, (k), aa, ba);                                   \
          if (DN < 4 && ((k) == 10 || (k) == 13)) halo_piece(pa_n, HB0 + w + 4 * (((k) - 10) / 3), hb_n);  \
          if (DN < 2 && (k) == 16) {          /* ninth piece of a 9-piece step: wave 0 */                  \
            int wq = w;                                                           \
            asm volatile("" : "+s"(wq));      /* (opaque: a hoisted predicate costs an SGPR pair per slot) */ \
            if (wq == 0) halo_piece(pa_n, HB0 + 8, hb_n);                         \
          }                                                                       \
        } while (0)
        W4_BLOCK(0, W4_FILL_Y)
#undef W4_FILL_Y
      }
      // ---- tile u + 1 has landed (everything but this interval's own pieces: at least 4, or 2 without halo pieces);
      // every wave is done reading tile u's stage and (DN == 4) the current halo buffer ----
      if (DN < 4)
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
      else
        asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
      asm volatile("s_barrier" ::: "memory");
      // ---- second half: MFMAs of block 1 (set 1); reads of the next tile's block 0 -> set 0; weight tile u + 3 ----
      {
        constexpr int DN1 = DN == 4 ? 0 : DN + 1;
        const unsigned aa = a_lane[DN1][0] + (DN == 4 ? hb_n : hb_c), ba = bf_lane[0] + st_1;
        const char* tb = DN + 3 < 5 ? pb_c + (DN + 3) * b_dn2 : pb_n + (DN - 2) * b_dn2;
        const int rot = (w + DN) & 3;
#define W4_FILL_X(k)                                                              \
        do {                                                                      \
          if ((k) < 9) W4_FRAG(0, (k), aa, ba);                                   \
          if ((k) == 10 || (k) == 13) b_piece(tb, rot + 4 * (((k) - 10) / 3), st_c);  \
          if ((k) == 16) {                                                        \
            int rq = rot;                                                         \
            asm volatile("" : "+s"(rq));                                          \
            if (rq < 2) b_piece(tb, rq + 8, st_c);                                \
          }                                                                       \
        } while (0)
        W4_BLOCK(1, W4_FILL_X)
#undef W4_FILL_X
      }
      // rotate the weight ring: tile u + 3 went into the stage tile u just left
      const unsigned t0 = st_c;
      st_c = st_1;
      st_1 = st_2;
      st_2 = t0;
    };
    step(std::integral_constant<int, 0>{});
    step(std::integral_constant<int, 1>{});
    step(std::integral_constant<int, 2>{});
    step(std::integral_constant<int, 3>{});
    step(std::integral_constant<int, 4>{});
    // next group
    const unsigned hb = hb_c;
    hb_c = hb_n;
    hb_n = hb;
    pb_c = pb_n;
    pb_n = grp_b(g + 2);
    pa_n = grp_a(g + 2);
  }
  }   // live
#undef W4_BLOCK
#undef W4_FRAG

  // the surplus prefetches must have landed before the LDS is reused; MFMA results are read by VALU below
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
  __syncthreads();
  const long tile_elems = (long)W4_BM * W4_BN;
  const long lane_off = (long)w * (20 * 1024) + lane * 4;
  // Partial tiles travel between workgroups on DIFFERENT XCDs (one L2 each, not coherent with each other for ordinary device
  // memory inside a kernel).  The workspace is FINE-GRAINED device memory (hipDeviceMallocFinegrained: not cached in the L2s;
  // w4_partials() below), written and read with agent-scope (sc1) accesses and ordered with s_waitcnt alone.  With ordinary
  // memory (p.sk_fence, the fallback when that allocation fails) every hand-over needs __threadfence() = buffer_wbl2 +
  // buffer_inv: a write-back and an invalidate of the WHOLE L2 of the XCD, under the 31 other workgroups that are in the middle
  // of their K walks (split-K cone launches of the step: 21.6 -> 19.7 ms without the fences).
  auto park = [&](float* slot) {          // the wave's accumulators as 16-byte vectors per lane: [tile][q4][lane][4]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) {
          float* dst = slot + ((a * 2 + i) * 5 + j) * 1024;
          // (the data operand straight from the register file the tile lives in -- AGPRs for j < 4 -- as a sub-register range)
#define W4_PARK(Q4, OFF)                                                                                           \
          do {                                                                                                     \
            const f32x4 v = __builtin_shufflevector(acc[a][i][j], acc[a][i][j], 4 * Q4, 4 * Q4 + 1, 4 * Q4 + 2, 4 * Q4 + 3); \
            if (j < 4) asm volatile("global_store_dwordx4 %0, %1, off offset:" #OFF " sc1\n\ts_nop 1" ::"v"(dst), "a"(v) : "memory"); \
            else asm volatile("global_store_dwordx4 %0, %1, off offset:" #OFF " sc1\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");       \
          } while (0)
          W4_PARK(0, 0);
          W4_PARK(1, 1024);
          W4_PARK(2, 2048);
          W4_PARK(3, 3072);
#undef W4_PARK
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  auto clear = [&]() {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[a][i][j][e] = 0.f;
  };
  // (plain loads: the compiler tracks their completion itself.  An inline-assembly load hands back a register the compiler may
  //  copy or spill BEFORE the data has arrived -- it did, under this much register pressure.  The workspace is not cached in
  //  the L2s and each slot is read exactly once per launch by one workgroup, so there is nothing stale to hit.)
  auto add = [&](const float* pz) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) {
            const f32x4 v = *(const f32x4*)(pz + (((a * 2 + i) * 5 + j) * 4 + q4) * 256);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[a][i][j][4 * q4 + r] += v[r];
          }
  };
  bool finish = true;                     // this workgroup runs the tile's epilogue
  if (SK) {
    if (g_begin != 0 || g_end != ngroups) {
      // a shared tile: workgroups s_lo .. s_hi hold a piece of it; a workgroup's FIRST piece parks in slot 2 s, a later one
      // (which can only be its last) in slot 2 s + 1
      const int s_lo = (tile * ngroups) / p.sk_per, s_hi = ((tile + 1) * ngroups - 1) / p.sk_per;
      park(p.ws + (long)(2 * lid + (first_piece ? 0 : 1)) * tile_elems + lane_off);
      if (p.sk_fence) __threadfence();
      __syncthreads();
      if (tid == 0) s_last = atomicAdd(p.cnt + tile, 1) == s_hi - s_lo;
      __syncthreads();
      finish = s_last != 0;
      if (finish) {
        if (p.sk_fence) __threadfence();
        clear();
        for (int sw = s_lo; sw <= s_hi; ++sw)
          add(p.ws + (long)(2 * sw + ((sw * p.sk_per) / ngroups == tile ? 0 : 1)) * tile_elems + lane_off);
        if (tid == 0) p.cnt[tile] = 0;   // counters are left clean for the next launch
      }
    }
  } else if (p.ws != nullptr && live && S_eff > 1) {
    // Deterministic split-K, as in the 256 x 320 kernel: every part parks its fp32 partial tile in the workspace, the last part
    // to arrive adds them in the fixed order z = 0 .. S-1 and runs the epilogue.  The 512 x 160 tile has as many elements as a
    // 256 x 320 one.
    const int S = S_eff;
    park(p.ws + ((long)blockIdx.y * nwg + lid) * tile_elems + lane_off);
    if (p.sk_fence) __threadfence();
    __syncthreads();
    if (tid == 0) s_last = atomicAdd(p.cnt + lid, 1) == S - 1;
    __syncthreads();
    if (!s_last) return;
    if (p.sk_fence) __threadfence();
    clear();
    for (int z = 0; z < S; ++z) add(p.ws + ((long)z * nwg + lid) * tile_elems + lane_off);
    if (tid == 0) p.cnt[lid] = 0;   // counters are left clean for the next launch
  }
  if (finish) {
    char* wave_lds = wl + w * (32 * EPI_ROWB(5) + 256);
    gemm_epilogue_lds_bf16<5>(p, acc[0], (long)m0 + w * 128, n0, 0, lane, wave_lds);
    gemm_epilogue_lds_bf16<5>(p, acc[1], (long)m0 + w * 128 + 64, n0, 0, lane, wave_lds);
  }
  if (!SK) break;
  u += g_end - g_begin;
  if (u >= u_end) break;
  first_piece = false;
  __syncthreads();            // the epilogue staged through the LDS the next piece's prologue fills
  }
}

// host side: called by dfold_gemm_bf16 for the conv launches that qualify (see there)
// Fine-grained workspace for the partial tiles (see the kernel): one buffer per (device, stream), grown on demand and kept for
// the life of the process -- launches in flight on two streams of a device (two towers, a side-stream data gradient, callers on
// several threads) never share slots (round-5 review: one buffer per device did).  The table is guarded by a mutex.  Growing
// waits for the launches of THAT stream which still read the old buffer; while the stream is being captured into a graph nothing
// may be synchronised or freed, so a capture that needs more than the stream already has falls back to the caller's ordinary
// workspace with fences (nullptr), as does a failed allocation or a full table.  DFOLD_CONV_FINE_WS=0 forces that path (A/B
// measurements and the parity test of the two protocols).
#include <mutex>
static float* w4_partials(size_t bytes, hipStream_t stream) {
  struct Entry { int dev; hipStream_t stream; float* ptr; size_t cap; };
  static Entry tab[64];
  static int n_tab = 0;
  static int mode = -1;
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  if (mode < 0) {
    const char* e = getenv("DFOLD_CONV_FINE_WS");
    mode = e ? atoi(e) : 1;
  }
  int dev = 0;
  if (!mode || hipGetDevice(&dev) != hipSuccess) return nullptr;
  Entry* en = nullptr;
  for (int i = 0; i < n_tab; ++i)
    if (tab[i].dev == dev && tab[i].stream == stream) en = &tab[i];
  if (en && en->cap >= bytes) return en->ptr;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &cs) != hipSuccess) (void)hipGetLastError();
  if (cs != hipStreamCaptureStatusNone) return nullptr;
  if (!en) {
    if (n_tab == 64) return nullptr;
    en = &tab[n_tab++];
    en->dev = dev; en->stream = stream; en->ptr = nullptr; en->cap = 0;
  }
  if (en->ptr) {
    if (hipStreamSynchronize(stream) != hipSuccess || hipFree(en->ptr) != hipSuccess) {
      (void)hipGetLastError();
      return nullptr;                     // (the old buffer stays: too small for this launch, still right for smaller ones)
    }
    en->ptr = nullptr;
    en->cap = 0;
  }
  void* q = nullptr;
  if (hipExtMallocWithFlags(&q, bytes, hipDeviceMallocFinegrained) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  en->ptr = (float*)q;
  en->cap = bytes;
  return en->ptr;
}

// splitk <= -2 (zero-frame-flagged launches only): up to -splitk parts per tile, how many of them walk K is decided on the
// device (see the kernel); p.sk_tiles = compute units.  The partial tiles of such a launch need -splitk x tiles slots, which
// only the library's own fine-grained buffer provides: without it the launch runs unsplit.
int dfold_conv_w4_launch(const GemmParams& p0, int splitk, hipStream_t stream) {
  GemmParams p = p0;
  const unsigned tiles = (unsigned)(((p.M + W4_BM - 1) / W4_BM) * (p.N / W4_BN));
  p.sk_fence = 1;
  if (splitk <= -2) {
    float* fine = p.nz_ps ? w4_partials((size_t)(-splitk) * tiles * W4_BM * W4_BN * sizeof(float), stream) : nullptr;
    if (fine) {
      p.ws = fine;
      p.sk_fence = 0;
      p.sk_per = -1;
      splitk = -splitk;
    } else {
      p.ws = nullptr;
      p.sk_per = 0;
      splitk = 1;
    }
  } else if (splitk > 1) {
    float* fine = w4_partials((size_t)splitk * tiles * W4_BM * W4_BN * sizeof(float), stream);
    if (fine) {
      p.ws = fine;
      p.sk_fence = 0;
    }
  }
  if (p.nz_ps) {
    DFOLD_MAX_LDS_ONCE((dfold_conv_w4_kernel<false, true>), W4_LDS_BYTES);
    DFOLD_LAUNCH((dfold_conv_w4_kernel<false, true>), dim3(tiles, splitk > 1 ? splitk : 1), dim3(256), (size_t)W4_LDS_BYTES, stream, p);
  } else {
    DFOLD_MAX_LDS_ONCE((dfold_conv_w4_kernel<false, false>), W4_LDS_BYTES);
    DFOLD_LAUNCH((dfold_conv_w4_kernel<false, false>), dim3(tiles, splitk > 1 ? splitk : 1), dim3(256), (size_t)W4_LDS_BYTES, stream, p);
  }
  return dfold_check_launch();
}

// stream-K form: `p` describes the UNSPLIT launch plus workspace (>= 2 n_wg partial tiles) and counters (>= tiles)
int dfold_conv_w4_launch_streamk(const GemmParams& p0, int n_wg, hipStream_t stream) {
  DFOLD_MAX_LDS_ONCE((dfold_conv_w4_kernel<true, false>), W4_LDS_BYTES);
  GemmParams p = p0;
  const int tiles = (int)(((p.M + W4_BM - 1) / W4_BM) * (p.N / W4_BN));
  const long units = (long)tiles * ((p.nseg / 25) * 10);
  p.sk_tiles = tiles;
  p.sk_per = (int)((units + n_wg - 1) / n_wg);
  p.sk_fence = 1;
  float* fine = w4_partials((size_t)2 * n_wg * W4_BM * W4_BN * sizeof(float), stream);
  if (fine) {
    p.ws = fine;
    p.sk_fence = 0;
  }
  DFOLD_LAUNCH((dfold_conv_w4_kernel<true, false>), dim3((unsigned)n_wg, 1), dim3(256), (size_t)W4_LDS_BYTES, stream, p);
  return dfold_check_launch();
}
