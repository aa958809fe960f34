// Fused forward kernels of the triangle pair operators for c_z = c_hidden = 128 (triangle multiplication) and
// c_in = 128, 4 heads x 32 (triangle attention) -- the configuration the reference instantiates
// (openfold/config.py:347-350; operators openfold/model/triangular_multiplicative_update.py:26-126,
// openfold/model/triangular_attention.py:31-139, Attention openfold/model/primitives.py:299-448).
//
// The pair tensor [B, N, N, 128] is streamed through the chip ONCE per stage; everything between two HBM passes
// happens in registers / LDS:
//
//   pair_proj_kernel<0>   LayerNorm_in -> 5 projections (a_p|a_g|b_p|b_g|g: one 640-wide MFMA product per 64-cell
//                         tile) -> sigmoid gates * mask -> a|b written as bf16 K-contiguous PLANES [B][N][256][NP]
//                         (the layout the ik,jk->ij contraction consumes: no transposes), output gate sigmoid(g) bf16.
//   (contraction)         x_c = a_c b_c^T per (batch, channel) on the bf16 MFMA engine (dfold_gemm_bf16, batched).
//   trimul_out_kernel     x planes -> LDS transpose -> LayerNorm_out -> linear_z (MFMA) -> * gate -> out.
//
//   pair_proj_kernel<1>   LayerNorm -> q|k|v|g projections + the 4-wide triangle-bias projection; q, k, sigmoid(g)
//                         channel-last bf16, v as key-contiguous planes [B][I][128][NP], bias fp32 [B][4][N][NP].
//   triatt_core_kernel    per (batch, row i, 128-query block): flash-style attention over the keys of row i for the 4
//                         heads (S^T = K Q^T on MFMA 16x16x32 so that the probabilities come out of the accumulators
//                         directly in B-operand layout for O^T = V^T P^T; online softmax in fp32, triangle bias and
//                         mask bias streamed), output gate, linear_o (MFMA) -> out.  No [I,H,N,N] logits in HBM.
//
// Work decomposition of the projection / output kernels: persistent workgroups (one per CU, 8 waves) loop over
// 64-cell tiles; the projection WEIGHTS LIVE IN REGISTERS for the whole kernel (wave w owns output channels
// [16w, 16w+16) of every 128-wide group as MFMA B fragments: 5 x 4 x 4 = 80 VGPRs), so a tile costs only its own
// 32 KB of pair-tensor reads; the next tile's rows are prefetched into registers while the current tile's MFMAs
// run.  Results are staged through LDS and leave as 16-byte vectors (128-byte plane segments / whole channel rows).
#include "dfold_common.h"
#include "../../include/dfold_hip.h"
#include <math.h>

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)

__device__ __forceinline__ float sigm_f(float y) { return __builtin_amdgcn_rcpf(1.f + __expf(-y)); }

static int pf_num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
      n = v;
    else
      n = 256;
  }
  return n;
}

// ------------------------------------------------------------------------------------------------------------------
// LDS tile of MFMA A rows: [rows][128 bf16] = 256 B per row, the 16-byte chunk index XORed with (row & 15):
// conflict-free for the 16x16x32 fragment reads (ds_read_b128 lane groups {0-3,12-15,20-27} ... cover 16 distinct
// rows x 2 adjacent chunks) and for the 4-byte LayerNorm writes.
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int a_tile_off(int row, int chunk) { return row * 256 + ((chunk ^ (row & 15)) << 4); }

#define PP_TILE 64
#define PP_SPITCH 144   // plane staging: 64 cells * 2 B + 16
#define PP_GPITCH 272   // channel-last staging: 128 ch * 2 B + 16

struct PairProjParams {
  const void* x;
  const float* mask;
  const float* gamma;
  const float* beta;
  const bf16_t* W;
  const float* bias;
  const float* wtri;
  bf16_t* o0;
  bf16_t* o1;
  bf16_t* o2;
  bf16_t* o3;
  float* f0;
  int B, N, NP, swap;
  float eps;
  int tri_blocked;   // MODE 1: triangle bias in 16 x 16 blocks of MFMA-accumulator order (triatt_fused.hip) instead of rows
};

// MODE 0: triangle multiplication (o0 = planes [B][N][256][NP], o1 = gate [B][N][N][128], f0 = LN stats or null)
// MODE 1: triangle attention      (o0 = q, o1 = k, o3 = gate: [B][N][N][128]; o2 = vT [B][N][128][NP]; f0 = tri [B][4][N][NP])
#define PP_STAGE0 (256 * PP_SPITCH + 64 * PP_GPITCH)
#define PP_LDS0 (2 * 16384 + 512 + 2 * PP_STAGE0 + 1024)      // (+ 1 KB: LayerNorm gamma | beta)
#define PP_LDS1 (2 * 16384 + 3 * 64 * PP_GPITCH + 128 * PP_SPITCH + 2048 + 1024)

// sum over the 16 lanes of a DPP row (all 16 lanes receive it): quad butterflies + two row rotations, 4 VALU ops
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_mov_f<0xb1>(0.f, v);    // quad_perm [1,0,3,2]
  v += dpp_mov_f<0x4e>(0.f, v);    // quad_perm [2,3,0,1]
  v += dpp_mov_f<0x124>(0.f, v);   // row_ror 4
  v += dpp_mov_f<0x128>(0.f, v);   // row_ror 8
  return v;
}

template <int MODE, bool XBF16>
__global__ __launch_bounds__(512) void pair_proj_kernel(const PairProjParams p) {
  constexpr int NG = MODE == 0 ? 5 : 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const ldsA2 = smem;                        // two A tiles (tile parity)
  // MODE 0
  float* const ldsM2 = (float*)(smem + 32768);     // two mask rows
  char* const ldsS2 = smem + 32768 + 512;                       // two staging areas (tile parity): planes + gate
  // MODE 1
  char* const ldsQ = smem + 32768;
  char* const ldsK = ldsQ + 64 * PP_GPITCH;
  char* const ldsG1 = ldsK + 64 * PP_GPITCH;
  char* const ldsV = ldsG1 + 64 * PP_GPITCH;
  float* const ldsT = (float*)(ldsV + 128 * PP_SPITCH);

  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, l4 = lane >> 4;
  const int N = p.N, NP = p.NP;

  // projection weights of this wave as MFMA B fragments (B[n][k], k contiguous): resident for the whole kernel
  bf16x8 wf[NG][4];
  float bv[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int n = g * 128 + 16 * w + l15;
    bv[g] = p.bias[n];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) wf[g][ks] = *(const bf16x8*)(p.W + (long)n * 128 + ks * 32 + l4 * 8);
  }
  // LayerNorm layout: 16 lanes per cell (lane l15 holds channels 8*l15 .. +8), four cells per pass (row l4 of the quad):
  // the two reductions of a cell are 4 DPP steps each and serve four cells at once
  // (gamma / beta of the LayerNorm sit in LDS and are re-read per tile: 16 registers that the pipelined MFMA phase needs)
  float* const ldsGB = (float*)(smem + (MODE == 0 ? PP_LDS0 : PP_LDS1) - 1024);
  if (tid < 128) {
    ldsGB[tid] = p.gamma[tid];
    ldsGB[128 + tid] = p.beta[tid];
  }
  __syncthreads();
  float wt[4][8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
#pragma unroll
    for (int h = 0; h < 4; ++h) wt[h][i] = MODE == 1 ? p.wtri[h * 128 + l15 * 8 + i] * 1.44269504088896341f : 0.f;   // bias consumed in the log2 domain
  }

  const int tpl = NP / PP_TILE;
  const unsigned ntiles = (unsigned)p.B * (unsigned)N * (unsigned)tpl;   // < 2^31, checked by the launcher
  const unsigned esz = XBF16 ? 2u : 4u;
  const unsigned rstride_in = (p.swap ? (unsigned)N : 1u) * 128u * esz;   // bytes between consecutive cells of a tile
  // per-thread byte offsets of the stream-out vectors relative to the tile's first element (tile-invariant)
  unsigned voff_pl[4], voff_cl[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int id = tid + 512 * i;
    voff_pl[i] = ((unsigned)(id >> 3) * (unsigned)NP + (unsigned)(id & 7) * 8u) * 2u;   // plane row `id >> 3` of this line
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int id = tid + 512 * i;
    voff_cl[i] = (unsigned)(id >> 4) * ((MODE == 0 && p.swap) ? (unsigned)N : 1u) * 256u + (unsigned)(id & 15) * 16u;
  }

  // one register set: the rows of tile t+1 are requested while tile t computes (a second set / prefetch distance 2 was
  // measured: no gain, +16 VGPRs)
  f32x4 zrA[2][2];
  float mkA[2] = {0.f, 0.f};
  auto issue = [&](unsigned t, f32x4 (&zr)[2][2], float (&mk)[2], int only = -1) __attribute__((always_inline)) {
    const int pt = (int)(t % (unsigned)tpl);
    const unsigned bl = t / (unsigned)tpl;
    const int line = (int)(bl % (unsigned)N), b = (int)(bl / (unsigned)N);
    const long cell0 = p.swap ? ((long)b * N + pt * PP_TILE) * N + line : ((long)b * N + line) * N + pt * PP_TILE;
    const char* base = (const char*)p.x + cell0 * (128 * (long)esz);     // wave-uniform
#pragma unroll
    for (int qd = 0; qd < 2; ++qd) {
      if (only >= 0 && qd != only) continue;       // (MODE 0 spreads the two quads over the MFMA phase)
      // cells past the row end re-read the last valid cell: their mask is zero (a = b = 0) and their rows are not stored
      int r = w * 8 + qd * 4 + l4;
      const int over = pt * PP_TILE + r - (N - 1);
      r -= over > 0 ? over : 0;
      const char* src = base + (unsigned)r * rstride_in + (unsigned)l15 * (8u * esz);
      if (XBF16) {
        const uint4 u = *(const uint4*)src;
        zr[qd][0] = (f32x4){bf_lo(u.x), bf_hi(u.x), bf_lo(u.y), bf_hi(u.y)};
        zr[qd][1] = (f32x4){bf_lo(u.z), bf_hi(u.z), bf_lo(u.w), bf_hi(u.w)};
      } else {
        zr[qd][0] = *(const f32x4*)src;
        zr[qd][1] = *(const f32x4*)(src + 16);
      }
      if (MODE == 0) {
        const int pos = pt * PP_TILE + w * 8 + qd * 4 + l4;
        // unconditional (the 16 lanes of a cell read one address; positions past the row end read its last cell): a load under
        // a lane condition merges with the 0.f through a copy that hipcc places right behind the load, with s_waitcnt
        // vmcnt(0) in front of it -- every prefetched row of the tile was waited for before the MFMA phase it should hide
        // under.  The consumer selects (S0 of the next tile).
        const int posc = pos < N ? w * 8 + qd * 4 + l4 : N - 1 - pt * PP_TILE;
        mk[qd] = p.mask[cell0 + (long)posc * (p.swap ? N : 1)];
      }
    }
  };

  // ---- stream a staged tile out as 16-byte vectors (128-byte plane segments / whole channel rows) ----
  auto stream_out = [&](int b, int line, int pt, int par) __attribute__((always_inline)) {
    const int pos0 = pt * PP_TILE;
    if (MODE == 0) {
      const char* const ldsS = ldsS2 + par * PP_STAGE0;
      const char* const ldsG0 = ldsS + 256 * PP_SPITCH;
      char* const pbase = (char*)p.o0 + ((((long)b * N + line) * 256) * NP + pos0) * 2;          // wave-uniform
      // all six staged vectors first, then the six stores (as read - wait - store per vector hipcc serialised six LDS round trips)
      u32x4 sv[6];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int id = tid + 512 * i, pl = id >> 3, v = id & 7;
        sv[i] = *(const u32x4*)(ldsS + pl * PP_SPITCH + v * 16);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int id = tid + 512 * i, cr = id >> 4, v = id & 15;
        sv[4 + i] = *(const u32x4*)(ldsG0 + cr * PP_GPITCH + v * 16);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i) *(u32x4*)(pbase + voff_pl[i]) = sv[i];
      const long cell0 = p.swap ? ((long)b * N + pos0) * N + line : ((long)b * N + line) * N + pos0;
      char* const gbase = (char*)p.o1 + cell0 * 256;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int id = tid + 512 * i, cr = id >> 4;
        if (pos0 + cr < N) *(u32x4*)(gbase + voff_cl[i]) = sv[4 + i];
      }
    } else {
      const long cell0 = ((long)b * N + line) * N + pos0;
      const bool planes_out = p.o0 != nullptr;      // null q / k / v / gate: only the triangle bias is wanted (triatt_fused.hip)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int id = tid + 512 * i, cr = id >> 4, v = id & 15;
        if (planes_out && pos0 + cr < N) {
          *(uint4*)((char*)p.o0 + cell0 * 256 + voff_cl[i]) = *(const uint4*)(ldsQ + cr * PP_GPITCH + v * 16);
          *(uint4*)((char*)p.o1 + cell0 * 256 + voff_cl[i]) = *(const uint4*)(ldsK + cr * PP_GPITCH + v * 16);
          *(uint4*)((char*)p.o3 + cell0 * 256 + voff_cl[i]) = *(const uint4*)(ldsG1 + cr * PP_GPITCH + v * 16);
        }
      }
      char* const vbase = (char*)p.o2 + ((((long)b * N + line) * 128) * NP + pos0) * 2;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int id = tid + 512 * i, pl = id >> 3, v = id & 7;
        if (planes_out) *(uint4*)(vbase + voff_pl[i]) = *(const uint4*)(ldsV + pl * PP_SPITCH + v * 16);
      }
      if (tid < 64) {
        const int h = tid >> 4, v = tid & 15;        // keys pos0 + 4v .. +4 of line (= query) `line`
        float* dst = p.f0 + (((long)b * 4 + h) * N + line) * NP + pos0 + v * 4;
        if (p.tri_blocked) {
          // [B][4][NP/16 query tiles][NP/16 key tiles][64 lanes][4]: element (q, key) sits where the lane that holds it in a
          // 16 x 16 S^T accumulator tile (lane = ((key & 15) >> 2) * 16 + (q & 15), register key & 3) finds it with ONE
          // contiguous 16-byte load per lane (1 KB per wave instruction instead of 16 rows x 64 bytes)
          const int nt16 = NP >> 4;
          dst = p.f0 + (((((long)b * 4 + h) * nt16 + (line >> 4)) * nt16 + (pos0 >> 4) + (v >> 2)) * 64 + (v & 3) * 16 + (line & 15)) * 4;
        }
        *(uint4*)dst = *(const uint4*)(ldsT + par * 256 + h * 64 + v * 4);
      }
    }
  };

  // MODE 0: the same stream-out in three parts (plane vectors 0-1, 2-3, the two gate vectors), staged reads and stores as separate
  // calls, so that the tile loop can spread them over the MFMA / gate phase of the next tile (with everything in one burst
  // behind the barrier the CU's one memory path and its eight waves took turns: memory-only 0.135 ms, compute-only 0.123 ms,
  // together 0.177 ms per call at batch 8 x N_res 256, scripts/make_pairproj_lab.py)
  u32x4 spv[2];
  auto stream_read = [&](int part, int par) __attribute__((always_inline)) {
    const char* const ldsS = ldsS2 + par * PP_STAGE0;
    const char* const ldsG0 = ldsS + 256 * PP_SPITCH;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int id = tid + 512 * (part < 2 ? 2 * part + k : k);
      spv[k] = part < 2 ? *(const u32x4*)(ldsS + (id >> 3) * PP_SPITCH + (id & 7) * 16) : *(const u32x4*)(ldsG0 + (id >> 4) * PP_GPITCH + (id & 15) * 16);
    }
  };
  auto stream_store = [&](int part, int b, int line, int pt) __attribute__((always_inline)) {
    const int pos0 = pt * PP_TILE;
    if (part < 2) {
      char* const pbase = (char*)p.o0 + ((((long)b * N + line) * 256) * NP + pos0) * 2;
#pragma unroll
      for (int k = 0; k < 2; ++k) *(u32x4*)(pbase + voff_pl[2 * part + k]) = spv[k];
    } else {
      const long cell0 = p.swap ? ((long)b * N + pos0) * N + line : ((long)b * N + line) * N + pos0;
      char* const gbase = (char*)p.o1 + cell0 * 256;
#pragma unroll
      for (int k = 0; k < 2; ++k)
        if (pos0 + ((tid + 512 * k) >> 4) < N) *(u32x4*)(gbase + voff_cl[k]) = spv[k];
    }
  };

  // Per tile t: [LayerNorm(t) -> A[par]] | barrier | [prefetch rows of t+1] [stores of tile t-1 from the staging
  // area] | barrier | [MFMA + gates(t) -> staging].  Loads and stores are both issued right before the MFMA phase and
  // nothing waits on the memory counter until the next tile's LayerNorm, so neither latency sits on the critical path
  // (the compiler waits vmcnt(0) wherever loads and stores are mixed).  A / mask / bias tiles alternate by tile parity.
  int pb = 0, pline = 0, ppt = 0, par = 0;
  bool have_prev = false;
  auto tile = [&](unsigned t, f32x4 (&zr)[2][2], float (&mk)[2]) __attribute__((always_inline)) {
    const int pt = (int)(t % (unsigned)tpl);
    const unsigned bl = t / (unsigned)tpl;
    const int line = (int)(bl % (unsigned)N), b = (int)(bl / (unsigned)N);
    char* const ldsA = ldsA2 + par * 16384;
    float* const ldsM = ldsM2 + par * 64;

    // ---- S0: LayerNorm of this wave's 8 cells, four at a time -> bf16 A tile (one 16-byte chunk per lane) ----
#pragma unroll
    for (int qd = 0; qd < 2; ++qd) {
      const int row = w * 8 + qd * 4 + l4;
      float x[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        x[i] = zr[qd][0][i];
        x[4 + i] = zr[qd][1][i];
      }
      const float mean = row16_sum(((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]))) * (1.f / 128.f);
      float q2 = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        x[i] -= mean;
        q2 = __builtin_fmaf(x[i], x[i], q2);
      }
      const float rstd = rsqrtf(row16_sum(q2) * (1.f / 128.f) + p.eps);
#pragma unroll
      for (int i = 0; i < 8; ++i) x[i] = __builtin_fmaf(x[i] * rstd, ldsGB[l15 * 8 + i], ldsGB[128 + l15 * 8 + i]);
      *(uint4*)(ldsA + a_tile_off(row, l15)) =
          make_uint4(pack2bf_hw(x[0], x[1]), pack2bf_hw(x[2], x[3]), pack2bf_hw(x[4], x[5]), pack2bf_hw(x[6], x[7]));
      if (MODE == 1) {
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          float th = 0.f;
#pragma unroll
          for (int i = 0; i < 8; ++i) th = __builtin_fmaf(x[i], wt[h][i], th);
          th = row16_sum(th);
          if (l15 == 0) ldsT[par * 256 + h * 64 + row] = th;
        }
      }
      if (MODE == 0) {
        if (l15 == 0) ldsM[row] = (pt * PP_TILE + row < N) ? mk[qd] : 0.f;
        if (p.f0 != nullptr) {
          const int pos = pt * PP_TILE + row;
          if (l15 == 0 && pos < N) {
            const long cell = p.swap ? ((long)b * N + pos) * N + line : ((long)b * N + line) * N + pos;
            p.f0[2 * cell] = mean;
            p.f0[2 * cell + 1] = rstd;
          }
        }
      }
    }
    __syncthreads();

    // ---- S1: prefetch the next tile's rows, stream the previous tile out (both in flight during the MFMA phase) ----
    // (this tile's rows have been consumed above.  Unconditional: after the last tile it re-reads that tile -- under
    //  `if (t + gridDim.x < ntiles)` the loaded registers reach the loop-carried ones through copies, and the wait for those
    //  copies sat HERE, in front of the MFMA phase: the prefetch never overlapped anything)
    const unsigned tnext = t + gridDim.x < ntiles ? t + gridDim.x : t;
    if (MODE == 1) {
      issue(tnext, zr, mk);
      if (have_prev) stream_out(pb, pline, ppt, par ^ 1);
      __syncthreads();
    }
    // MODE 0: the staging area alternates with the tile parity, so this tile's results go where tile t - 2's were read from
    // before every wave's barrier above -- no second barrier per tile; its loads and stores are spread over S2 (mem_slot)
    auto mem_slot = [&](int rt, int q) __attribute__((always_inline)) {
      if (MODE != 0) return;
      if (rt == 0 && q == 0) issue(tnext, zr, mk, 0);
      if (rt == 0 && q == 1) issue(tnext, zr, mk, 1);
      if (!have_prev) return;
      if (rt == 0 && q == 2) stream_read(0, par ^ 1);
      if (rt == 1 && q == 0) stream_store(0, pb, pline, ppt);
      if (rt == 1 && q == 1) stream_read(1, par ^ 1);
      if (rt == 1 && q == 3) stream_store(1, pb, pline, ppt);
      if (rt == 2 && q == 0) stream_read(2, par ^ 1);
      if (rt == 2 && q == 2) stream_store(2, pb, pline, ppt);
    };
    char* const ldsS = ldsS2 + par * PP_STAGE0;
    char* const ldsG0 = ldsS + 256 * PP_SPITCH;

    // ---- S2: projections on MFMA 16x16x32, gates, staging ----
    // Software-pipelined over the four 16-cell row tiles: the 20 MFMAs of row tile rt + 1 are issued BETWEEN the gate
    // arithmetic of row tile rt (sched_group_barrier: 1 MFMA per few VALU).  Written as [all MFMAs of rt][all gates of rt] the
    // two waves of a SIMD ran the same phase at the same time -- matrix pipe and VALU took turns instead of overlapping
    // (hipcc -S: ~20 MFMAs, then ~100 VALU, four times).  The bias enters as the accumulator init of the first K step.
    // fragments of a row tile (four K steps), one K step of its products, the gate arithmetic of ONE of its four cell rows
    auto frags_rt = [&](int rt, bf16x8 (&af)[4]) __attribute__((always_inline)) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) af[ks] = *(const bf16x8*)(ldsA + a_tile_off(rt * 16 + l15, ks * 4 + l4));
    };
    auto mma_ks = [&](int ks, const bf16x8 (&af)[4], f32x4 (&acc)[NG]) __attribute__((always_inline)) {
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const f32x4 c0 = {bv[g], bv[g], bv[g], bv[g]};
        acc[g] = MFMA16(af[ks], wf[g][ks], ks == 0 ? c0 : acc[g]);
      }
    };
    // accumulator layout: column (channel) = l15, rows (cells) = l4*4 + r
    const int ch = 16 * w + l15;
    float ga[4], gb[4];
    auto gate_row = [&](int rt, int r, const f32x4 (&acc)[NG]) __attribute__((always_inline)) {
      const int cell0 = rt * 16 + l4 * 4;
      if (MODE == 0) {
        const float m = ldsM[cell0 + r];
        ga[r] = acc[0][r] * sigm_f(acc[1][r]) * m;
        gb[r] = acc[2][r] * sigm_f(acc[3][r]) * m;
        *(bf16_t*)(ldsG0 + (cell0 + r) * PP_GPITCH + ch * 2) = f2bf_hw(sigm_f(acc[4][r]));
        if (r == 3) {
          *(uint2*)(ldsS + ch * PP_SPITCH + cell0 * 2) = make_uint2(pack2bf_hw(ga[0], ga[1]), pack2bf_hw(ga[2], ga[3]));
          *(uint2*)(ldsS + (128 + ch) * PP_SPITCH + cell0 * 2) = make_uint2(pack2bf_hw(gb[0], gb[1]), pack2bf_hw(gb[2], gb[3]));
        }
      } else {
        *(bf16_t*)(ldsQ + (cell0 + r) * PP_GPITCH + ch * 2) = f2bf_hw(acc[0][r]);
        *(bf16_t*)(ldsK + (cell0 + r) * PP_GPITCH + ch * 2) = f2bf_hw(acc[1][r]);
        *(bf16_t*)(ldsG1 + (cell0 + r) * PP_GPITCH + ch * 2) = f2bf_hw(sigm_f(acc[3][r]));
        ga[r] = acc[2][r];
        if (r == 3) *(uint2*)(ldsV + ch * PP_SPITCH + cell0 * 2) = make_uint2(pack2bf_hw(ga[0], ga[1]), pack2bf_hw(ga[2], ga[3]));
      }
    };
    f32x4 accA[NG], accB[NG];
    bf16x8 afA[4], afB[4];
    frags_rt(0, afA);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) mma_ks(ks, afA, accA);
    frags_rt(1, afB);
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      f32x4 (&cur)[NG] = (rt & 1) ? accB : accA;
      f32x4 (&nxt)[NG] = (rt & 1) ? accA : accB;
      bf16x8 (&afn)[4] = (rt & 1) ? afA : afB;        // fragments of row tile rt + 1 (requested one row tile ahead)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        // chunk q: the NG products of K step q of row tile rt + 1 go to the matrix pipe, then the gates of cell row q of row
        // tile rt issue on the VALU while they run
        if (rt + 1 < 4) mma_ks(q, afn, nxt);
        gate_row(rt, q, cur);
        mem_slot(rt, q);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (rt + 2 < 4) frags_rt(rt + 2, (rt & 1) ? afB : afA);
    }
    pb = b; pline = line; ppt = pt; par ^= 1; have_prev = true;
  };
  unsigned t = blockIdx.x;
  if (t < ntiles) issue(t, zrA, mkA);
  for (; t < ntiles; t += gridDim.x) tile(t, zrA, mkA);
  __syncthreads();
  if (have_prev) stream_out(pb, pline, ppt, par ^ 1);
}

static int pair_proj_launch(int mode, const PairProjParams& p, int x_is_bf16, hipStream_t st) {
  const long ntiles = (long)p.B * p.N * (p.NP / PP_TILE);
  long grid = ntiles < pf_num_cus() ? ntiles : pf_num_cus();
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)pair_proj_kernel<0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS0);
    (void)hipFuncSetAttribute((const void*)pair_proj_kernel<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS0);
    (void)hipFuncSetAttribute((const void*)pair_proj_kernel<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS1);
    (void)hipFuncSetAttribute((const void*)pair_proj_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS1);
    attr_done = true;
  }
  if (mode == 0) {
    if (x_is_bf16)
      DFOLD_LAUNCH((pair_proj_kernel<0, true>), dim3((unsigned)grid), dim3(512), PP_LDS0, st, p);
    else
      DFOLD_LAUNCH((pair_proj_kernel<0, false>), dim3((unsigned)grid), dim3(512), PP_LDS0, st, p);
  } else {
    if (x_is_bf16)
      DFOLD_LAUNCH((pair_proj_kernel<1, true>), dim3((unsigned)grid), dim3(512), PP_LDS1, st, p);
    else
      DFOLD_LAUNCH((pair_proj_kernel<1, false>), dim3((unsigned)grid), dim3(512), PP_LDS1, st, p);
  }
  return dfold_check_launch();
}

static bool pf_dims_ok(int B, int N, int NP) {
  // 32-bit per-thread byte offsets inside one batch item's planes (256 N NP bf16) and inside a tile of transposed cells
  return B > 0 && N > 0 && NP >= N && (NP % PP_TILE) == 0 && (long)B * N * (NP / PP_TILE) < (1L << 31) && (long)B * N * N < (1L << 40) &&
         512L * NP < (1L << 32) && 64L * N * 512 < (1L << 32);
}

extern "C" int dfold_trimul_proj_fwd(const void* z, int32_t z_is_bf16, const float* mask, const float* ln_gamma,
                                     const float* ln_beta, const void* w_cat_bf16, const float* bias_cat, void* planes_bf16,
                                     void* gate_bf16, float* stats, int32_t B, int32_t N, int32_t NP, int32_t incoming,
                                     float eps, void* stream) {
  if (!z || !mask || !ln_gamma || !ln_beta || !w_cat_bf16 || !bias_cat || !planes_bf16 || !gate_bf16 || !pf_dims_ok(B, N, NP))
    return DFOLD_EINVAL;
  PairProjParams p;
  p.x = z; p.mask = mask; p.gamma = ln_gamma; p.beta = ln_beta; p.W = (const bf16_t*)w_cat_bf16; p.bias = bias_cat;
  p.wtri = nullptr; p.o0 = (bf16_t*)planes_bf16; p.o1 = (bf16_t*)gate_bf16; p.o2 = nullptr; p.o3 = nullptr; p.f0 = stats;
  p.B = B; p.N = N; p.NP = NP; p.swap = incoming ? 1 : 0; p.eps = eps; p.tri_blocked = 0;
  return pair_proj_launch(0, p, z_is_bf16, (hipStream_t)stream);
}

extern "C" int dfold_triatt_proj_fwd(const void* x, int32_t x_is_bf16, const float* ln_gamma, const float* ln_beta,
                                     const void* w_cat_bf16, const float* bias_cat, const float* w_tri, void* q_bf16,
                                     void* k_bf16, void* vT_bf16, void* gate_bf16, float* tri, int32_t B, int32_t N,
                                     int32_t NP, int32_t ending, float eps, void* stream) {
  const bool none = !q_bf16 && !k_bf16 && !vT_bf16 && !gate_bf16;      // triangle bias only
  const bool all = q_bf16 && k_bf16 && vT_bf16 && gate_bf16;
  if (!x || !ln_gamma || !ln_beta || !w_cat_bf16 || !bias_cat || !w_tri || !(none || all) || !tri || !pf_dims_ok(B, N, NP))
    return DFOLD_EINVAL;
  PairProjParams p;
  p.x = x; p.mask = nullptr; p.gamma = ln_gamma; p.beta = ln_beta; p.W = (const bf16_t*)w_cat_bf16; p.bias = bias_cat;
  p.wtri = w_tri; p.o0 = (bf16_t*)q_bf16; p.o1 = (bf16_t*)k_bf16; p.o2 = (bf16_t*)vT_bf16; p.o3 = (bf16_t*)gate_bf16;
  p.f0 = tri; p.B = B; p.N = N; p.NP = NP; p.swap = ending ? 1 : 0; p.eps = eps;
  p.tri_blocked = none ? 1 : 0;          // the row kernel's layout (tri then needs [B][4][NP][NP] floats)
  return pair_proj_launch(1, p, x_is_bf16, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------------------------
// Triangle multiplication, output stage: x planes [B][N][128][NP] (bf16, from the contraction) -> LayerNorm_out over
// the 128 channels of every cell -> linear_z -> * gate -> out [B][N][N][128] (fp32 or bf16)
// (triangular_multiplicative_update.py:119-124).
// ------------------------------------------------------------------------------------------------------------------
#define TO_XPITCH 260   // transposed x tile: 128 ch * 2 B + 4 (2-way conflicts on the dword writes = free)
#define TO_OPITCH 528   // fp32 out staging: 128 * 4 + 16
#define TO_LDS (64 * TO_XPITCH + 16384 + 2 * 64 * PP_GPITCH + 64 * TO_OPITCH)

struct TriMulOutParams {
  const bf16_t* xpl;
  const bf16_t* gate;
  const float* gamma;
  const float* beta;
  const bf16_t* Wz;
  const float* bz;
  void* out;
  int B, N, NP, out_bf16;
  float eps;
};

__global__ __launch_bounds__(512) void trimul_out_kernel(const TriMulOutParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const ldsX = smem;
  char* const ldsA = smem + 64 * TO_XPITCH;
  char* const ldsG2 = ldsA + 16384;               // two gate tiles (tile parity)
  char* const ldsO = ldsG2 + 2 * 64 * PP_GPITCH;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, l4 = lane >> 4;
  const int N = p.N, NP = p.NP;

  bf16x8 wz[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) wz[ks] = *(const bf16x8*)(p.Wz + (long)(16 * w + l15) * 128 + ks * 32 + l4 * 8);
  const float bz = p.bz[16 * w + l15];
  float gam[8], bet[8];            // LayerNorm layout as in pair_proj_kernel: 16 lanes per cell, 8 channels per lane
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    gam[i] = p.gamma[l15 * 8 + i];
    bet[i] = p.beta[l15 * 8 + i];
  }

  const int tpl = NP / PP_TILE;
  const unsigned ntiles = (unsigned)p.B * (unsigned)N * (unsigned)tpl;
  const int pr = tid >> 3, xv8 = tid & 7;   // x planes: channel pair (2pr, 2pr+1), cells 8*xv8 .. +8

  // first-class vector values, not arrays of HIP's uint4 struct: the struct arrays were kept in scratch memory once the
  // prefetch became unconditional (scratch_store right behind the loads = a wait for them)
  typedef __attribute__((ext_vector_type(4))) unsigned tou32x4;
  tou32x4 xv0, xv1, gv0, gv1;
  auto issue = [&](unsigned t) __attribute__((always_inline)) {
    const int jt = (int)(t % (unsigned)tpl);
    const unsigned bl = t / (unsigned)tpl;
    const int i = (int)(bl % (unsigned)N), b = (int)(bl / (unsigned)N);
    const bf16_t* xb = p.xpl + (((long)b * N + i) * 128 + 2 * pr) * NP + jt * PP_TILE + xv8 * 8;
    xv0 = *(const tou32x4*)xb;
    xv1 = *(const tou32x4*)(xb + NP);
    // unconditional, positions past the row end re-read its last cell (their output rows are not stored): see pair_proj_kernel
    const int cr0 = tid >> 4, v = tid & 15;
    const int pos0 = jt * PP_TILE + cr0, pos1 = pos0 + 32;
    const bf16_t* gb = p.gate + (((long)b * N + i) * N) * 128 + v * 8;
    gv0 = *(const tou32x4*)(gb + (long)(pos0 < N ? pos0 : N - 1) * 128);
    gv1 = *(const tou32x4*)(gb + (long)(pos1 < N ? pos1 : N - 1) * 128);
  };

  auto stream_out = [&](int b, int i, int jt) __attribute__((always_inline)) {   // whole 512-byte rows
    if (!p.out_bf16) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int id = tid + 512 * k, cr = id >> 5, v = id & 31;
        const int pos = jt * PP_TILE + cr;
        if (pos < N)
          *(uint4*)((float*)p.out + (((long)b * N + i) * N + pos) * 128 + v * 4) = *(const uint4*)(ldsO + cr * TO_OPITCH + v * 16);
      }
    } else {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int id = tid + 512 * k, cr = id >> 4, v = id & 15;
        const int pos = jt * PP_TILE + cr;
        if (pos < N) {
          const f32x4 lo = *(const f32x4*)(ldsO + cr * TO_OPITCH + v * 32);
          const f32x4 hi = *(const f32x4*)(ldsO + cr * TO_OPITCH + v * 32 + 16);
          *(uint4*)((bf16_t*)p.out + (((long)b * N + i) * N + pos) * 128 + v * 8) =
              make_uint4(pack2bf_hw(lo[0], lo[1]), pack2bf_hw(lo[2], lo[3]), pack2bf_hw(hi[0], hi[1]), pack2bf_hw(hi[2], hi[3]));
        }
      }
    }
  };

  // Per tile t: [x / gate tiles(t) -> LDS] | barrier | [LayerNorm -> A] [prefetch t+1] [stores of tile t-1] | barrier |
  // [MFMA + gate -> fp32 staging]; see pair_proj_kernel for why the stores ride one tile behind.
  unsigned t = blockIdx.x;
  if (t < ntiles) issue(t);
  int pb = 0, pi = 0, pjt = 0, par = 0;
  bool have_prev = false;
  for (; t < ntiles; t += gridDim.x) {
    const int jt = (int)(t % (unsigned)tpl);
    const unsigned bl = t / (unsigned)tpl;
    const int i = (int)(bl % (unsigned)N), b = (int)(bl / (unsigned)N);
    char* const ldsG = ldsG2 + par * 64 * PP_GPITCH;

    // ---- S0a: x tile transposed into [cell][channel] (two channels per dword), gate tile ----
    {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const uint32_t a = xv0[q >> 1], bq = xv1[q >> 1];
        *(uint32_t*)(ldsX + (xv8 * 8 + q) * TO_XPITCH + pr * 4) = (q & 1) ? ((a >> 16) | (bq & 0xffff0000u)) : ((a & 0xffffu) | (bq << 16));
      }
      {
        const int cr = tid >> 4, v = tid & 15;       // ids tid and tid + 512: rows cr and cr + 32
        *(tou32x4*)(ldsG + cr * PP_GPITCH + v * 16) = gv0;
        *(tou32x4*)(ldsG + (cr + 32) * PP_GPITCH + v * 16) = gv1;
      }
    }
    __syncthreads();
    // ---- S0b: LayerNorm over channels, four cells per pass -> bf16 A tile ----
#pragma unroll
    for (int qd = 0; qd < 2; ++qd) {
      const int row = w * 8 + qd * 4 + l4;
      float x[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint32_t u = *(const uint32_t*)(ldsX + row * TO_XPITCH + l15 * 16 + i * 4);
        x[2 * i] = bf_lo(u);
        x[2 * i + 1] = bf_hi(u);
      }
      const float mean = row16_sum(((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]))) * (1.f / 128.f);
      float q2 = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        x[i] -= mean;
        q2 = __builtin_fmaf(x[i], x[i], q2);
      }
      const float rstd = rsqrtf(row16_sum(q2) * (1.f / 128.f) + p.eps);
#pragma unroll
      for (int i = 0; i < 8; ++i) x[i] = __builtin_fmaf(x[i] * rstd, gam[i], bet[i]);
      *(uint4*)(ldsA + a_tile_off(row, l15)) =
          make_uint4(pack2bf_hw(x[0], x[1]), pack2bf_hw(x[2], x[3]), pack2bf_hw(x[4], x[5]), pack2bf_hw(x[6], x[7]));
    }
    issue(t + gridDim.x < ntiles ? t + gridDim.x : t);         // unconditional: see pair_proj_kernel
    if (have_prev) stream_out(pb, pi, pjt);
    __syncthreads();

    // ---- S2: linear_z on MFMA, gate, fp32 staging ----
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) acc = MFMA16(*(const bf16x8*)(ldsA + a_tile_off(rt * 16 + l15, ks * 4 + l4)), wz[ks], acc);
      const int cell0 = rt * 16 + l4 * 4, ch = 16 * w + l15;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float g = bf2f(*(const bf16_t*)(ldsG + (cell0 + r) * PP_GPITCH + ch * 2));
        *(float*)(ldsO + (cell0 + r) * TO_OPITCH + ch * 4) = (acc[r] + bz) * g;
      }
    }
    pb = b; pi = i; pjt = jt; par ^= 1; have_prev = true;
  }
  __syncthreads();
  if (have_prev) stream_out(pb, pi, pjt);
}

extern "C" int dfold_trimul_out_fwd(const void* x_planes_bf16, const void* gate_bf16, const float* ln_gamma,
                                    const float* ln_beta, const void* w_z_bf16, const float* b_z, void* out,
                                    int32_t out_is_bf16, int32_t B, int32_t N, int32_t NP, float eps, void* stream) {
  if (!x_planes_bf16 || !gate_bf16 || !ln_gamma || !ln_beta || !w_z_bf16 || !b_z || !out || !pf_dims_ok(B, N, NP)) return DFOLD_EINVAL;
  TriMulOutParams p;
  p.xpl = (const bf16_t*)x_planes_bf16; p.gate = (const bf16_t*)gate_bf16; p.gamma = ln_gamma; p.beta = ln_beta;
  p.Wz = (const bf16_t*)w_z_bf16; p.bz = b_z; p.out = out; p.B = B; p.N = N; p.NP = NP; p.out_bf16 = out_is_bf16 ? 1 : 0;
  p.eps = eps;
  const long ntiles = (long)B * N * (NP / PP_TILE);
  const long grid = ntiles < pf_num_cus() ? ntiles : pf_num_cus();
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)trimul_out_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, TO_LDS);
    attr_done = true;
  }
  DFOLD_LAUNCH(trimul_out_kernel, dim3((unsigned)grid), dim3(512), TO_LDS, (hipStream_t)stream, p);
  return dfold_check_launch();
}

// ------------------------------------------------------------------------------------------------------------------
// Triangle attention core: one workgroup per (batch, row i, block of 64 queries), 4 waves x 16 queries; 50 KB of LDS
// and <= 256 VGPRs so that two workgroups share a CU (their memory phases overlap each other's compute).
//   logits[q, key] = scale * q_h . k_h + tri[h, q, key] + inf * (mask[i, key] - 1)      (triangular_attention.py:105-113)
//   o = softmax_key(logits) v_h ;  og = o * gate ;  out = og W_o^T + b_o                (primitives.py:219-243, 385-448)
// Keys are walked in chunks of 256 with an online softmax (any N); per (head, chunk) step K [256][32] and V^T [32][256]
// sit in LDS, shared by the 4 waves; the next step's tiles are fetched into registers while the current one computes.
// ------------------------------------------------------------------------------------------------------------------
#define TC_VPITCH 528
#define TC_LDS_K 0
#define TC_LDS_V 16384
#define TC_LDS_OG (16384 + 32 * TC_VPITCH)           // 33280
#define TC_LDS_MB (TC_LDS_OG + 16384)                // 49664
#define TC_LDS (TC_LDS_MB + 1024)                    // 50688
#define TC_WSTAGE 8320                               // per-wave out staging inside the K|V region (33280 / 4)
#define TC_OPITCH 520                                // 16 query rows x (128 ch * 4 B + 8)

struct TriAttCoreParams {
  const bf16_t* q;
  const bf16_t* k;
  const bf16_t* vT;
  const bf16_t* gate;
  const float* tri;
  const float* mask;
  const bf16_t* Wo;
  const float* bo;
  void* out;
  int B, N, NP, ending, out_bf16;
  float inf, scale;
};

__device__ __forceinline__ float xor16_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float xor16_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}

__global__ __launch_bounds__(256, 2) void triatt_core_kernel(const TriAttCoreParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const ldsK = smem + TC_LDS_K;
  char* const ldsV = smem + TC_LDS_V;
  char* const ldsOG = smem + TC_LDS_OG;
  float* const ldsMB = (float*)(smem + TC_LDS_MB);
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, l4 = lane >> 4;
  const int N = p.N, NP = p.NP;
  const unsigned qblocks = (unsigned)(N + 63) / 64u;
  // XCD-aware work id: consecutive blockIdx values are dealt round-robin to the 8 XCDs (each with its own L2), so the
  // query blocks of one row -- which share that row's K / V tiles -- and neighbouring rows -- which share the triangle
  // bias -- are renumbered to sit on ONE XCD (bijective for any grid size).  Without it every query block re-fetched
  // its row's K / V through the fabric: 2.2 GB instead of 0.27 GB per call at N_res = 512 (profiles/r2_triangle_pmc_*).
  const unsigned nwg = gridDim.x, bid = blockIdx.x;
  const unsigned xq = nwg >> 3, xr = nwg & 7, xcd = bid & 7, xidx = bid >> 3;
  const unsigned lid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + xidx;
  const int qb = (int)(lid % qblocks);
  const unsigned bi = lid / qblocks;
  const int i = (int)(bi % (unsigned)N), b = (int)(bi / (unsigned)N);
  const long rowbase = ((long)b * N + i) * N;      // first cell of row i (x' coordinates)
  const long vbase = ((long)b * N + i) * 128;      // first V^T row of row i
  const int q0 = qb * 64 + w * 16;
  const int myq = q0 + l15;
  const bool qok = myq < N;
  const int kswz = (-(l15 >> 2)) & 3;
  const int nchunks = (N + 255) / 256;
  const float sl2 = p.scale * 1.44269504088896341f, inv_sl2 = 1.f / sl2;

  // register-staged prefetch of the next (head, chunk) step: K / V^T tiles, the mask row, and -- when the step opens a
  // new head -- this lane's query fragment and gate values.  Branch-free (clamped addresses; whatever a clamped lane
  // reads is finite and meets a -inf mask bias / a zero probability) and with no arithmetic on the loaded values, so
  // that nothing waits on the memory counter before the commit one whole compute phase later.
  u32x4 kr[4], vr[4], qn;
  u32x2 gn[2];
  float mvr;
  auto fetch = [&](int h, int kc) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int id = tid + 256 * j;
      const int r = id >> 2, c = id & 3, key = kc * 256 + r;           // K tile: 256 keys x 4 chunks of 8 channels
      kr[j] = *(const u32x4*)(p.k + (rowbase + (key < N ? key : 0)) * 128 + h * 32 + c * 8);
      const int cr = id >> 5, v8 = id & 31, key8 = kc * 256 + v8 * 8;  // V^T tile: 32 channels x 32 chunks of 8 keys
      vr[j] = *(const u32x4*)(p.vT + (vbase + h * 32 + cr) * NP + (key8 < NP ? key8 : 0));
    }
    const int key = kc * 256 + tid;
    const int kk = key < N ? key : 0;
    mvr = p.ending ? p.mask[((long)b * N + kk) * N + i] : p.mask[rowbase + kk];
    if (kc == 0) {
      const long qrow = (rowbase + (qok ? myq : 0)) * 128 + h * 32;
      qn = *(const u32x4*)(p.q + qrow + l4 * 8);
      gn[0] = *(const u32x2*)(p.gate + qrow + l4 * 4);
      gn[1] = *(const u32x2*)(p.gate + qrow + 16 + l4 * 4);
    }
  };
  bf16x8 qf;
  u32x2 g2[2];
  auto commit = [&](int kc) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int id = tid + 256 * j;
      const int r = id >> 2, c = id & 3;
      *(u32x4*)(ldsK + r * 64 + ((c ^ ((-(r >> 2)) & 3)) << 4)) = kr[j];
      const int cr = id >> 5, v8 = id & 31;
      *(u32x4*)(ldsV + cr * TC_VPITCH + v8 * 16) = vr[j];
    }
    ldsMB[tid] = (kc * 256 + tid < N) ? p.inf * (mvr - 1.f) * inv_sl2 : -INFINITY;
    if (kc == 0) {
      qf = __builtin_bit_cast(bf16x8, qn);
      g2[0] = gn[0];
      g2[1] = gn[1];
    }
  };

  fetch(0, 0);
  commit(0);

#pragma unroll 1
  for (int h = 0; h < 4; ++h) {
    const float* trow = p.tri + (((long)b * 4 + h) * N + (qok ? myq : 0)) * NP;
    float m_run = -INFINITY, l_run = 0.f;
    f32x4 oacc[2];
    oacc[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    oacc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bf16x8 qh = qf;          // this head's query fragment / gates (qf, g2 are overwritten by the last commit)
    const u32x2 gh0 = g2[0], gh1 = g2[1];
#pragma unroll 1
    for (int kc = 0; kc < nchunks; ++kc) {
      __syncthreads();   // the tiles of this step are in LDS
      // triangle bias of this step first (16 independent 16-byte loads per lane, L2 resident), then the next step's
      // tiles: the counter is in order, so waiting for the bias below leaves the prefetch in flight
      f32x4 tb[16];
#pragma unroll
      for (int kb = 0; kb < 16; ++kb) {
        const int keyb = kc * 256 + kb * 16 + l4 * 4;
        // pad keys (>= NP only when NP is not a multiple of 256) read a valid finite bias; their mask bias is -inf
        tb[kb] = *(const f32x4*)(trow + (keyb < NP ? keyb : 0));
      }
      const bool more = (kc + 1 < nchunks) || (h + 1 < 4);
      const int nh = (kc + 1 < nchunks) ? h : ((h + 1) & 3), nkc = (kc + 1 < nchunks) ? kc + 1 : 0;
      fetch(nh, nkc);   // unconditional (the last step re-reads head 0 and drops it): behind a branch the compiler would
                        // have to assume the loads were NOT issued and wait vmcnt(0) for the bias instead of vmcnt(12)
      __builtin_amdgcn_sched_barrier(0);
      // S^T[key][q] = K_h Q_h^T : A rows = keys, B columns = this wave's 16 queries, K = 32 channels (one MFMA per 16 keys).
      // The accumulator starts at the mask bias of its 4 keys (stored pre-divided by scale*log2e), so that
      // logit*log2e = acc * (scale*log2e) + tri*log2e costs one fma per element (tri is stored pre-multiplied).
      f32x4 s[16];
#pragma unroll
      for (int kb = 0; kb < 16; ++kb) {
        const bf16x8 a = *(const bf16x8*)(ldsK + (kb * 16 + l15) * 64 + ((l4 ^ kswz) << 4));
        s[kb] = MFMA16(a, qh, *(const f32x4*)(ldsMB + kb * 16 + l4 * 4));
      }
      __builtin_amdgcn_sched_barrier(0);
      // lane holds query l15, keys kb*16 + l4*4 + r; four independent max / sum chains (no fp reassociation by the compiler)
      f32x4 mc4 = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
      for (int kb = 0; kb < 16; ++kb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          s[kb][r] = __builtin_fmaf(s[kb][r], sl2, tb[kb][r]);
          mc4[r] = fmaxf(mc4[r], s[kb][r]);
        }
      }
      const float mc = xor16_max(fmaxf(fmaxf(mc4[0], mc4[1]), fmaxf(mc4[2], mc4[3])));
      const float mn = fmaxf(m_run, mc);
      const float resc = __builtin_amdgcn_exp2f(m_run - mn);
      f32x4 ps4 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < 16; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          s[kb][r] = __builtin_amdgcn_exp2f(s[kb][r] - mn);
          ps4[r] += s[kb][r];
        }
      float ps = (ps4[0] + ps4[1]) + (ps4[2] + ps4[3]);
      ps = xor16_sum(ps);
      l_run = l_run * resc + ps;
      m_run = mn;
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) oacc[cb][r] *= resc;
      // O^T[c][q] += V^T[c][keys] P^T[keys][q]; MFMA k-slot e of lane group l4 <-> key (2ks + (e>>2))*16 + l4*4 + (e&3)
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const u32x4 pb = {pack2bf_hw(s[2 * ks][0], s[2 * ks][1]), pack2bf_hw(s[2 * ks][2], s[2 * ks][3]),
                          pack2bf_hw(s[2 * ks + 1][0], s[2 * ks + 1][1]), pack2bf_hw(s[2 * ks + 1][2], s[2 * ks + 1][3])};
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          const char* vp = ldsV + (cb * 16 + l15) * TC_VPITCH + ks * 64 + l4 * 8;
          const u32x2 lo = *(const u32x2*)vp;
          const u32x2 hi = *(const u32x2*)(vp + 32);
          const u32x4 av = {lo.x, lo.y, hi.x, hi.y};
          oacc[cb] = MFMA16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, pb), oacc[cb]);
        }
      }
      __syncthreads();   // everybody is done with this step's tiles
      if (more) commit(nkc);
    }
    // head epilogue: normalise, gate, stage into this wave's rows of the linear_o operand tile
    const float inv = __builtin_amdgcn_rcpf(l_run);
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      const u32x2 gg = cb == 0 ? gh0 : gh1;
      const float o0 = oacc[cb][0] * inv * bf_lo(gg.x), o1 = oacc[cb][1] * inv * bf_hi(gg.x);
      const float o2 = oacc[cb][2] * inv * bf_lo(gg.y), o3 = oacc[cb][3] * inv * bf_hi(gg.y);
      const int row = w * 16 + l15, chunk = h * 4 + cb * 2 + (l4 >> 1);
      *(uint2*)(ldsOG + a_tile_off(row, chunk) + ((l4 & 1) << 3)) = make_uint2(pack2bf_hw(o0, o1), pack2bf_hw(o2, o3));
    }
  }
  // the last step ended with a barrier: every wave has left the K / V tiles (reused as output staging below); the og
  // rows are wave-private (written and read by the same wave)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();

  // linear_o: out[q][o] = sum_hc og[q][hc] W_o[o][hc] + b_o[o]; W_o fragments straight from global (32 KB, L2 resident)
  f32x4 oa[8];
#pragma unroll
  for (int nb = 0; nb < 8; ++nb) oa[nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const bf16x8 a = *(const bf16x8*)(ldsOG + a_tile_off(w * 16 + l15, ks * 4 + l4));
#pragma unroll
    for (int nb = 0; nb < 8; ++nb)
      oa[nb] = MFMA16(a, *(const bf16x8*)(p.Wo + (nb * 16 + l15) * 128 + ks * 32 + l4 * 8), oa[nb]);
  }
  char* const st = smem + w * TC_WSTAGE;   // 16 queries x 128 channels fp32, wave-private
#pragma unroll
  for (int nb = 0; nb < 8; ++nb) {
    const float bo = p.bo[nb * 16 + l15];
#pragma unroll
    for (int r = 0; r < 4; ++r) *(float*)(st + (l4 * 4 + r) * TC_OPITCH + (nb * 16 + l15) * 4) = oa[nb][r] + bo;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int id = lane + 64 * j, row = id >> 5, c = id & 31;      // 16 rows x 32 chunks of 4 channels
    const int qq = q0 + row;
    const uint2 v0 = *(const uint2*)(st + row * TC_OPITCH + c * 16);
    const uint2 v1 = *(const uint2*)(st + row * TC_OPITCH + c * 16 + 8);
    if (qq < N) {
      const long cell = p.ending ? ((long)b * N + qq) * N + i : rowbase + qq;
      if (!p.out_bf16)
        *(uint4*)((float*)p.out + cell * 128 + c * 4) = make_uint4(v0.x, v0.y, v1.x, v1.y);
      else
        *(uint2*)((bf16_t*)p.out + cell * 128 + c * 4) =
            make_uint2(pack2bf_hw(__uint_as_float(v0.x), __uint_as_float(v0.y)), pack2bf_hw(__uint_as_float(v1.x), __uint_as_float(v1.y)));
    }
  }
}

extern "C" int dfold_triatt_core_fwd(const void* q_bf16, const void* k_bf16, const void* vT_bf16, const void* gate_bf16,
                                     const float* tri, const float* mask, const void* w_o_bf16, const float* b_o, void* out,
                                     int32_t out_is_bf16, int32_t B, int32_t N, int32_t NP, int32_t ending, float inf,
                                     float scale, void* stream) {
  if (!q_bf16 || !k_bf16 || !vT_bf16 || !gate_bf16 || !tri || !mask || !w_o_bf16 || !b_o || !out || !pf_dims_ok(B, N, NP))
    return DFOLD_EINVAL;
  const long nwg = (long)B * N * ((N + 63) / 64);
  if (nwg > 0x7fffffffL) return DFOLD_EINVAL;
  TriAttCoreParams p;
  p.q = (const bf16_t*)q_bf16; p.k = (const bf16_t*)k_bf16; p.vT = (const bf16_t*)vT_bf16; p.gate = (const bf16_t*)gate_bf16;
  p.tri = tri; p.mask = mask; p.Wo = (const bf16_t*)w_o_bf16; p.bo = b_o; p.out = out;
  p.B = B; p.N = N; p.NP = NP; p.ending = ending ? 1 : 0; p.out_bf16 = out_is_bf16 ? 1 : 0; p.inf = inf; p.scale = scale;
  DFOLD_LAUNCH(triatt_core_kernel, dim3((unsigned)nwg), dim3(256), TC_LDS, (hipStream_t)stream, p);
  return dfold_check_launch();
}
