// Fused backward kernels of the triangle multiplicative update (gfx950, MI355X), c_z = c_hidden = 128
// (reference: the autograd of openfold/model/triangular_multiplicative_update.py:97-126).
//
// The backward mirrors the three passes of the forward (csrc/pair_fused.hip) instead of re-running the chain that keeps every
// intermediate as its own tensor (~30 launches):
//
//   (recompute)            planes a | b, output gate sigmoid(g) and x planes with the forward's own kernels (nothing pair-sized
//                          is kept between forward and backward)
//   trimul_out_bwd_kernel  THIS FILE.  One pass over (x planes, gate, dout): LayerNorm_out recomputed, y = xn W_z^T + b_z,
//                          d(gate pre-activation) = dout y g (1 - g), dy = dout g, dxn = dy W_z, LayerNorm backward -> dx written
//                          as bf16 PLANES (the layout the two contraction gradients consume), plus xn / dy channel-last (operands
//                          of dW_z) and the gradients of b_z, gamma_out, beta_out accumulated in registers
//   (contraction pair)     da_c = dx_c b_c, db_c = dx_c^T a_c on the reduction-major MFMA kernel (csrc/tn_gemm.hip)
//   pair_proj_kernel<2>    csrc/pair_fused.hip: LayerNorm_in and the four gated projections recomputed per 64-cell tile, gate
//                          backward against the da | db planes -> pre-activation gradients [cells][a_p a_g b_p b_g | g]
//   (dense tail)           dzn = d5 W_cat (MFMA engine), LayerNorm_in backward, weight gradients as reduction-major products
//
// Work decomposition as trimul_out_kernel: persistent workgroups (one per CU, 8 waves) over 64-cell tiles (row i, 64 columns),
// 16 lanes per cell for the LayerNorm passes, wave w owns channels [16w, 16w + 16) in the two MFMA products.
#include "dfold_common.h"
#include "../../include/dfold_hip.h"
#include <math.h>

typedef __attribute__((ext_vector_type(4))) unsigned tbu32x4;
#define TB2_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)

#define TB2_TILE 64
#define TB2_XPITCH 260    // transposed x tile [cell][128 ch bf16] + 4
#define TB2_GPITCH 272    // channel-last bf16 rows + 16
#define TB2_FPITCH 528    // fp32 rows [cell][128] + 16
#define TB2_PPITCH 144    // plane staging [ch][64 cells bf16] + 16
#define TB2_LDS_X 0
#define TB2_LDS_G (TB2_LDS_X + 64 * TB2_XPITCH)            // 16640
#define TB2_LDS_D (TB2_LDS_G + 64 * TB2_GPITCH)            // + 17408
#define TB2_LDS_A (TB2_LDS_D + 64 * TB2_FPITCH)            // + 33792
#define TB2_LDS_DY (TB2_LDS_A + 16384)
#define TB2_LDS_DG (TB2_LDS_DY + 16384)
#define TB2_LDS_P (TB2_LDS_DG + 64 * TB2_GPITCH)
#define TB2_LDS_ST (TB2_LDS_P + 128 * TB2_PPITCH)
#define TB2_LDS_RED (TB2_LDS_ST + 512)                     // end-of-kernel reduction of the parameter gradients: 3 x 128 floats x 8 waves
#define TB2_LDS (TB2_LDS_RED + 8 * 384 * 4)

__device__ __forceinline__ int tb2_a_off(int row, int chunk) { return row * 256 + ((chunk ^ (row & 15)) << 4); }
__device__ __forceinline__ float tb2_row16_sum(float v) {
  v += dpp_mov_f<0xb1>(0.f, v);
  v += dpp_mov_f<0x4e>(0.f, v);
  v += dpp_mov_f<0x124>(0.f, v);
  v += dpp_mov_f<0x128>(0.f, v);
  return v;
}

struct TriMulOutBwdParams {
  const bf16_t* xpl;
  const bf16_t* gate;
  const void* dout;
  const float* gamma;
  const float* beta;
  const bf16_t* Wz;
  const bf16_t* WzT;
  const float* bz;
  bf16_t* dxpl;
  bf16_t* dg;
  bf16_t* dy;
  bf16_t* xn;
  float* dgamma;
  float* dbeta;
  float* dbz;
  float* dbg;
  long dg_ld;
  int B, N, NP, dout_bf16;
  float eps;
};

static int tb2_num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0, v = 0;
    n = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
  }
  return n;
}

template <bool DBF16>
__global__ __launch_bounds__(512) void trimul_out_bwd_kernel(const TriMulOutBwdParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const ldsX = smem + TB2_LDS_X;
  char* const ldsG = smem + TB2_LDS_G;
  char* const ldsD = smem + TB2_LDS_D;       // dout tile (fp32), later the dxn tile
  char* const ldsA = smem + TB2_LDS_A;       // xn as MFMA A rows
  char* const ldsDY = smem + TB2_LDS_DY;     // dy as MFMA A rows
  char* const ldsDG = smem + TB2_LDS_DG;     // gate pre-activation gradient, channel-last
  char* const ldsP = smem + TB2_LDS_P;       // dx as planes [ch][cell]
  float* const ldsST = (float*)(smem + TB2_LDS_ST);
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, l4 = lane >> 4;
  const int N = p.N, NP = p.NP;

  // wave w owns output channels [16w, 16w + 16) of both products: B fragments B[n][k] (k contiguous)
  bf16x8 wz[4], wzt[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    wz[ks] = *(const bf16x8*)(p.Wz + (long)(16 * w + l15) * 128 + ks * 32 + l4 * 8);
    wzt[ks] = *(const bf16x8*)(p.WzT + (long)(16 * w + l15) * 128 + ks * 32 + l4 * 8);
  }
  const float bz = p.bz[16 * w + l15];
  float gam[8], bet[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    gam[i] = p.gamma[l15 * 8 + i];
    bet[i] = p.beta[l15 * 8 + i];
  }
  float dgam[8], dbet[8], dbz_acc = 0.f, dbg_acc = 0.f;      // parameter gradients of this lane's channels, summed over all its cells
#pragma unroll
  for (int i = 0; i < 8; ++i) dgam[i] = dbet[i] = 0.f;

  const int tpl = NP / TB2_TILE;
  const unsigned ntiles = (unsigned)p.B * (unsigned)N * (unsigned)tpl;
  const int pr = tid >> 3, xv8 = tid & 7;   // x planes: channel pair (2pr, 2pr + 1), cells 8 xv8 .. + 8

  tbu32x4 xv0, xv1, gv0, gv1, dv[4];
  auto issue = [&](unsigned t) __attribute__((always_inline)) {
    const int jt = (int)(t % (unsigned)tpl);
    const unsigned bl = t / (unsigned)tpl;
    const int i = (int)(bl % (unsigned)N), b = (int)(bl / (unsigned)N);
    const bf16_t* xb = p.xpl + (((long)b * N + i) * 128 + 2 * pr) * NP + jt * TB2_TILE + xv8 * 8;
    xv0 = *(const tbu32x4*)xb;
    xv1 = *(const tbu32x4*)(xb + NP);
    const long row0 = ((long)b * N + i) * N;
    {
      const int cr0 = tid >> 4, v = tid & 15;
      const int pos0 = jt * TB2_TILE + cr0, pos1 = pos0 + 32;
      const bf16_t* gb = p.gate + row0 * 128 + v * 8;
      gv0 = *(const tbu32x4*)(gb + (long)(pos0 < N ? pos0 : N - 1) * 128);
      gv1 = *(const tbu32x4*)(gb + (long)(pos1 < N ? pos1 : N - 1) * 128);
    }
    if (DBF16) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int id = tid + 512 * k, cr = id >> 4, v = id & 15;
        const int pos = jt * TB2_TILE + cr;
        dv[k] = *(const tbu32x4*)((const bf16_t*)p.dout + (row0 + (pos < N ? pos : N - 1)) * 128 + v * 8);
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int id = tid + 512 * k, cr = id >> 5, v = id & 31;
        const int pos = jt * TB2_TILE + cr;
        dv[k] = *(const tbu32x4*)((const float*)p.dout + (row0 + (pos < N ? pos : N - 1)) * 128 + v * 4);
      }
    }
  };

  unsigned t = blockIdx.x;
  if (t < ntiles) issue(t);
  for (; t < ntiles; t += gridDim.x) {
    const int jt = (int)(t % (unsigned)tpl);
    const unsigned bl = t / (unsigned)tpl;
    const int i = (int)(bl % (unsigned)N), b = (int)(bl / (unsigned)N);
    const long row0 = ((long)b * N + i) * N;
    const int pos0 = jt * TB2_TILE;

    // ---- A: the prefetched tile into LDS: x transposed to [cell][channel] (two channels per dword), gate, dout (fp32) ----
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const uint32_t a = xv0[q >> 1], bq = xv1[q >> 1];
      *(uint32_t*)(ldsX + (xv8 * 8 + q) * TB2_XPITCH + pr * 4) = (q & 1) ? ((a >> 16) | (bq & 0xffff0000u)) : ((a & 0xffffu) | (bq << 16));
    }
    {
      const int cr = tid >> 4, v = tid & 15;
      *(tbu32x4*)(ldsG + cr * TB2_GPITCH + v * 16) = gv0;
      *(tbu32x4*)(ldsG + (cr + 32) * TB2_GPITCH + v * 16) = gv1;
    }
    if (DBF16) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int id = tid + 512 * k, cr = id >> 4, v = id & 15;
        float* d = (float*)(ldsD + cr * TB2_FPITCH + v * 32);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          d[2 * q] = bf_lo(dv[k][q]);
          d[2 * q + 1] = bf_hi(dv[k][q]);
        }
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int id = tid + 512 * k, cr = id >> 5, v = id & 31;
        *(tbu32x4*)(ldsD + cr * TB2_FPITCH + v * 16) = dv[k];
      }
    }
    __syncthreads();

    // ---- B: LayerNorm_out over the channels, four cells per pass -> xn as bf16 A rows; statistics kept for phase E ----
#pragma unroll
    for (int qd = 0; qd < 2; ++qd) {
      const int row = w * 8 + qd * 4 + l4;
      float x[8];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint32_t u = *(const uint32_t*)(ldsX + row * TB2_XPITCH + l15 * 16 + k * 4);
        x[2 * k] = bf_lo(u);
        x[2 * k + 1] = bf_hi(u);
      }
      const float mean = tb2_row16_sum(((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]))) * (1.f / 128.f);
      float q2 = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        x[k] -= mean;
        q2 = __builtin_fmaf(x[k], x[k], q2);
      }
      const float rstd = rsqrtf(tb2_row16_sum(q2) * (1.f / 128.f) + p.eps);
#pragma unroll
      for (int k = 0; k < 8; ++k) x[k] = __builtin_fmaf(x[k] * rstd, gam[k], bet[k]);
      *(uint4*)(ldsA + tb2_a_off(row, l15)) =
          make_uint4(pack2bf_hw(x[0], x[1]), pack2bf_hw(x[2], x[3]), pack2bf_hw(x[4], x[5]), pack2bf_hw(x[6], x[7]));
      if (l15 == 0) {
        ldsST[2 * row] = mean;
        ldsST[2 * row + 1] = rstd;
      }
    }
    issue(t + gridDim.x < ntiles ? t + gridDim.x : t);      // unconditional (see pair_proj_kernel): the next tile's rows
    __syncthreads();

    // ---- C: y = xn W_z^T + b_z on MFMA; gate backward: d(gate pre-activation) = dout y g (1 - g), dy = dout g ----
    const int ch = 16 * w + l15;
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      f32x4 acc = (f32x4){bz, bz, bz, bz};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) acc = TB2_MFMA(*(const bf16x8*)(ldsA + tb2_a_off(rt * 16 + l15, ks * 4 + l4)), wz[ks], acc);
      const int cell0 = rt * 16 + l4 * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int cell = cell0 + r;
        const float g = bf2f(*(const bf16_t*)(ldsG + cell * TB2_GPITCH + ch * 2));
        const float d = *(const float*)(ldsD + cell * TB2_FPITCH + ch * 4);
        const float dyv = d * g;
        const float dgp = d * acc[r] * g * (1.f - g);
        *(bf16_t*)(ldsDG + cell * TB2_GPITCH + ch * 2) = f2bf_hw(dgp);
        dbg_acc += pos0 + cell < N ? dgp : 0.f;
        *(bf16_t*)(ldsDY + tb2_a_off(cell, ch >> 3) + (ch & 7) * 2) = f2bf_hw(dyv);
        dbz_acc += pos0 + cell < N ? dyv : 0.f;
      }
    }
    __syncthreads();

    // ---- D: dxn = dy W_z on MFMA (B fragments from W_z^T) -> fp32 tile in the dout region ----
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) acc = TB2_MFMA(*(const bf16x8*)(ldsDY + tb2_a_off(rt * 16 + l15, ks * 4 + l4)), wzt[ks], acc);
      const int cell0 = rt * 16 + l4 * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r) *(float*)(ldsD + (cell0 + r) * TB2_FPITCH + ch * 4) = acc[r];
    }
    __syncthreads();

    // ---- E: LayerNorm backward per cell (16 lanes per cell): dx = rstd (g dxn - mean(g dxn) - xhat mean(g dxn xhat)) -> planes ----
#pragma unroll
    for (int qd = 0; qd < 2; ++qd) {
      const int row = w * 8 + qd * 4 + l4;
      const float mean = ldsST[2 * row], rstd = ldsST[2 * row + 1];
      const bool live = pos0 + row < N;
      float xh[8], u[8];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint32_t v = *(const uint32_t*)(ldsX + row * TB2_XPITCH + l15 * 16 + k * 4);
        xh[2 * k] = (bf_lo(v) - mean) * rstd;
        xh[2 * k + 1] = (bf_hi(v) - mean) * rstd;
      }
      const f32x4 t0 = *(const f32x4*)(ldsD + row * TB2_FPITCH + l15 * 32);
      const f32x4 t1 = *(const f32x4*)(ldsD + row * TB2_FPITCH + l15 * 32 + 16);
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float tk = k < 4 ? t0[k] : t1[k - 4];
        const float tl = live ? tk : 0.f;              // (selects, not branches: one basic block per cell row)
        dgam[k] = __builtin_fmaf(tl, xh[k], dgam[k]);
        dbet[k] += tl;
        u[k] = tk * gam[k];
        s1 += u[k];
        s2 = __builtin_fmaf(u[k], xh[k], s2);
      }
      const float m1 = tb2_row16_sum(s1) * (1.f / 128.f), m2 = tb2_row16_sum(s2) * (1.f / 128.f);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float dx = live ? rstd * (u[k] - m1 - xh[k] * m2) : 0.f;
        *(bf16_t*)(ldsP + (l15 * 8 + k) * TB2_PPITCH + row * 2) = f2bf_hw(dx);
      }
    }
    __syncthreads();

    // ---- F: stream out: dx planes (128-byte segments), gate pre-activation gradient / dy / xn rows (16-byte vectors) ----
    {
      tbu32x4 sv[4];
      char* const pbase = (char*)p.dxpl + ((((long)b * N + i) * 128) * NP + pos0) * 2;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int id = tid + 512 * k, pl = id >> 3, v = id & 7, cr = id >> 4, v16 = id & 15;
        sv[k] = *(const tbu32x4*)(ldsP + pl * TB2_PPITCH + v * 16);
        sv[2 + k] = *(const tbu32x4*)(ldsDG + cr * TB2_GPITCH + v16 * 16);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int id = tid + 512 * k, pl = id >> 3, v = id & 7, cr = id >> 4, v16 = id & 15;
        *(tbu32x4*)(pbase + ((long)pl * NP + v * 8) * 2) = sv[k];
        if (pos0 + cr < N) *(tbu32x4*)(p.dg + (row0 + pos0 + cr) * p.dg_ld + v16 * 8) = sv[2 + k];
      }
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int id = tid + 512 * k, cr = id >> 4, v = id & 15;
        sv[k] = *(const tbu32x4*)(ldsDY + tb2_a_off(cr, v));
        sv[2 + k] = *(const tbu32x4*)(ldsA + tb2_a_off(cr, v));
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int id = tid + 512 * k, cr = id >> 4, v = id & 15;
        if (pos0 + cr < N) {
          *(tbu32x4*)(p.dy + (row0 + pos0 + cr) * 128 + v * 8) = sv[k];
          *(tbu32x4*)(p.xn + (row0 + pos0 + cr) * 128 + v * 8) = sv[2 + k];
        }
      }
    }
    // (the next tile's phase A writes ldsX / ldsG / ldsD only: every wave left their last readers -- phases C and E -- before
    //  the barrier in front of phase F; the regions phase F reads are next written behind the barrier after phase A)
  }

  // ---- parameter gradients: lanes of a wave that hold the same channels (the four l4 groups), then the 8 waves, then atomics ----
  __syncthreads();
  float* const red = (float*)(smem + TB2_LDS_RED);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    float a = dgam[k], c = dbet[k];
    a += __shfl_xor(a, 16, 64);
    a += __shfl_xor(a, 32, 64);
    c += __shfl_xor(c, 16, 64);
    c += __shfl_xor(c, 32, 64);
    if (l4 == 0) {
      red[w * 384 + l15 * 8 + k] = a;
      red[w * 384 + 128 + l15 * 8 + k] = c;
    }
  }
  {
    float a = dbz_acc;
    a += __shfl_xor(a, 16, 64);
    a += __shfl_xor(a, 32, 64);
    float c = dbg_acc;
    c += __shfl_xor(c, 16, 64);
    c += __shfl_xor(c, 32, 64);
    if (l4 == 0) {
      red[w * 384 + 256 + l15] = a;       // channel 16 w + l15: slot l15 of this wave's row
      red[w * 384 + 272 + l15] = c;
    }
  }
  __syncthreads();
  if (tid < 256) {
    float s = 0.f;
#pragma unroll
    for (int ww = 0; ww < 8; ++ww) s += red[ww * 384 + tid];
    atomicAdd((tid < 128 ? p.dgamma : p.dbeta) + (tid & 127), s);
  } else if (tid < 384) {
    const int c = tid - 256;                            // channel c lives in wave c >> 4, slot c & 15
    atomicAdd(p.dbz + c, red[(c >> 4) * 384 + 256 + (c & 15)]);
  } else {
    const int c = tid - 384;
    atomicAdd(p.dbg + c, red[(c >> 4) * 384 + 272 + (c & 15)]);
  }
}

extern "C" int dfold_trimul_out_bwd(const void* x_planes_bf16, const void* gate_bf16, const void* dout, int32_t dout_is_bf16,
                                    const float* ln_gamma, const float* ln_beta, const void* w_z_bf16, const void* w_z_t_bf16,
                                    const float* b_z, void* dx_planes_bf16, void* dgate_pre_bf16, int64_t dgate_ld, void* dy_bf16,
                                    void* xn_bf16, float* d_gamma, float* d_beta, float* d_bz, float* d_bg, int32_t B, int32_t N,
                                    int32_t NP, float eps, void* stream) {
  if (!x_planes_bf16 || !gate_bf16 || !dout || !ln_gamma || !ln_beta || !w_z_bf16 || !w_z_t_bf16 || !b_z || !dx_planes_bf16 ||
      !dgate_pre_bf16 || !dy_bf16 || !xn_bf16 || !d_gamma || !d_beta || !d_bz || !d_bg)
    return DFOLD_EINVAL;
  if (B <= 0 || N <= 0 || NP < N || (NP % TB2_TILE) || dgate_ld < 128 || (dgate_ld & 7) || (long)B * N * (NP / TB2_TILE) >= (1L << 31))
    return DFOLD_EINVAL;
  TriMulOutBwdParams p;
  p.xpl = (const bf16_t*)x_planes_bf16; p.gate = (const bf16_t*)gate_bf16; p.dout = dout; p.gamma = ln_gamma; p.beta = ln_beta;
  p.Wz = (const bf16_t*)w_z_bf16; p.WzT = (const bf16_t*)w_z_t_bf16; p.bz = b_z; p.dxpl = (bf16_t*)dx_planes_bf16;
  p.dg = (bf16_t*)dgate_pre_bf16; p.dy = (bf16_t*)dy_bf16; p.xn = (bf16_t*)xn_bf16; p.dgamma = d_gamma; p.dbeta = d_beta; p.dbz = d_bz; p.dbg = d_bg;
  p.dg_ld = dgate_ld; p.B = B; p.N = N; p.NP = NP; p.dout_bf16 = dout_is_bf16 ? 1 : 0; p.eps = eps;
  const long ntiles = (long)B * N * (NP / TB2_TILE);
  const long grid = ntiles < tb2_num_cus() ? ntiles : tb2_num_cus();
  if (dout_is_bf16) {
    DFOLD_MAX_LDS_ONCE((trimul_out_bwd_kernel<true>), TB2_LDS);
    DFOLD_LAUNCH((trimul_out_bwd_kernel<true>), dim3((unsigned)grid), dim3(512), (size_t)TB2_LDS, (hipStream_t)stream, p);
  } else {
    DFOLD_MAX_LDS_ONCE((trimul_out_bwd_kernel<false>), TB2_LDS);
    DFOLD_LAUNCH((trimul_out_bwd_kernel<false>), dim3((unsigned)grid), dim3(512), (size_t)TB2_LDS, (hipStream_t)stream, p);
  }
  return dfold_check_launch();
}

// ------------------------------------------------------------------------------------------------------------------
// Projection stage backward: per 64-cell tile (line, 64 positions; the incoming form reads z transposed like the forward)
//   recompute   zn = LayerNorm_in(z),  [a_p a_g b_p b_g] = zn W^T + b  (one 512-wide product on MFMA 16x16x32, weights in registers)
//   gates       a = a_p s(a_g) m,  b = b_p s(b_g) m:   d a_p = da s(a_g) m,   d a_g = da a_p m s(a_g) (1 - s(a_g))   (same for b)
//   out         d4 = [d a_p | d a_g | d b_p | d b_g] as bf16 rows of the [cells][640] pre-activation gradient matrix (its last 128
//               columns, the output gate's, are written by trimul_out_bwd_kernel), zn rows (operand of the weight gradient),
//               LayerNorm statistics (for the LayerNorm_in backward), bias gradients accumulated in registers.
// da | db arrive as the PLANES the contraction gradients wrote ([B][N][256][NP], line-major like the forward's a | b planes):
// a tile's 256 plane rows x 128 bytes go HBM -> LDS by LDS-DMA one tile ahead (two buffers; 16-byte chunks XOR-permuted on the
// source side so that the 8-byte reads of 16 consecutive plane rows hit different banks).
// Per tile: [LayerNorm -> A] | barrier | [prefetch z rows + DMA of tile t + 1] [stores: d4 of tile t - 1, zn of tile t] | barrier |
// [MFMA + gate backward -> staging] | barrier (the A tile is single-buffered: 148 KB of LDS).
// ------------------------------------------------------------------------------------------------------------------
#define TB3_SPITCH 1040                      // staging rows: 512 ch bf16 + 16
#define TB3_LDS_A 0
#define TB3_LDS_DP 16384                     // two da | db tiles of 32 KiB
#define TB3_LDS_S (TB3_LDS_DP + 2 * 32768)
#define TB3_LDS_M (TB3_LDS_S + 64 * TB3_SPITCH)
#define TB3_LDS_GB (TB3_LDS_M + 256)
#define TB3_LDS_RED (TB3_LDS_GB + 1024)
#define TB3_LDS (TB3_LDS_RED + 2048)

typedef __attribute__((address_space(3))) char tb3_lchar;

struct TriMulProjBwdParams {
  const void* x;
  const float* mask;
  const float* gamma;
  const float* beta;
  const bf16_t* W;         // [>= 512][128]: a_p | a_g | b_p | b_g
  const float* bias;
  const bf16_t* dpl;       // da | db planes [B][N][256][NP]
  bf16_t* d5;              // [B N N][ld5]: columns 0 .. 511 written here
  bf16_t* zn;              // [B N N][128]
  float* stats;            // [B N N][2]
  float* dbias;            // [512], atomically accumulated
  long ld5;
  int B, N, NP, swap;
  float eps;
};

__device__ __forceinline__ float tb3_sigm(float y) { return __builtin_amdgcn_rcpf(1.f + __expf(-y)); }
__device__ __forceinline__ void tb3_dma(const char* base, unsigned voff, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(lds_addr) : "memory", "m0");
}

template <bool XBF16>
__global__ __launch_bounds__(512) void trimul_proj_bwd_kernel(const TriMulProjBwdParams p) {
  constexpr int NG = 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const ldsA = smem + TB3_LDS_A;
  char* const ldsDP = smem + TB3_LDS_DP;
  char* const ldsS = smem + TB3_LDS_S;
  float* const ldsM = (float*)(smem + TB3_LDS_M);
  float* const ldsGB = (float*)(smem + TB3_LDS_GB);
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, l4 = lane >> 4;
  const int N = p.N, NP = p.NP;

  bf16x8 wf[NG][4];
  float bv[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int n = g * 128 + 16 * w + l15;
    bv[g] = p.bias[n];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) wf[g][ks] = *(const bf16x8*)(p.W + (long)n * 128 + ks * 32 + l4 * 8);
  }
  if (tid < 128) {
    ldsGB[tid] = p.gamma[tid];
    ldsGB[128 + tid] = p.beta[tid];
  }
  __syncthreads();
  float dbacc[NG] = {0.f, 0.f, 0.f, 0.f};

  const int tpl = NP / TB2_TILE;
  const unsigned ntiles = (unsigned)p.B * (unsigned)N * (unsigned)tpl;
  const unsigned esz = XBF16 ? 2u : 4u;
  const unsigned rstride_in = (p.swap ? (unsigned)N : 1u) * 128u * esz;
  const long cstride = p.swap ? (long)N : 1L;                // cells of a tile are `cstride` rows of the cell matrix apart
  // LDS-DMA of the da | db tile: piece q = w + 8k holds plane rows 8q .. 8q + 7; lane -> (row 8q + (lane >> 3), physical chunk
  // lane & 7) receives the logical chunk (lane & 7) ^ key(row), key(row) = (row >> 1) & 7 = (4 (w & 1) + (lane >> 4)) & 7
  const unsigned dp_lane = (unsigned)(lane >> 3) * (unsigned)NP * 2u + (unsigned)((((lane & 7) ^ ((4 * (w & 1) + (lane >> 4)) & 7))) << 4);
  const unsigned lds_dp = (unsigned)(uintptr_t)(tb3_lchar*)ldsDP;

  f32x4 zr[2][2];
  float mk[2] = {0.f, 0.f};
  auto coords = [&](unsigned t, int& pt, int& line, int& b) __attribute__((always_inline)) {
    pt = (int)(t % (unsigned)tpl);
    const unsigned bl = t / (unsigned)tpl;
    line = (int)(bl % (unsigned)N);
    b = (int)(bl / (unsigned)N);
  };
  auto issue = [&](unsigned t, int buf) __attribute__((always_inline)) {
    int pt, line, b;
    coords(t, pt, line, b);
    const long cell0 = p.swap ? ((long)b * N + pt * TB2_TILE) * N + line : ((long)b * N + line) * N + pt * TB2_TILE;
    const char* base = (const char*)p.x + cell0 * (128 * (long)esz);
#pragma unroll
    for (int qd = 0; qd < 2; ++qd) {
      int r = w * 8 + qd * 4 + l4;
      const int over = pt * TB2_TILE + r - (N - 1);
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
      const int pos = pt * TB2_TILE + w * 8 + qd * 4 + l4;
      const int posc = pos < N ? w * 8 + qd * 4 + l4 : N - 1 - pt * TB2_TILE;
      mk[qd] = p.mask[cell0 + (long)posc * cstride];
    }
    const char* dbase = (const char*)p.dpl + ((((long)b * N + line) * 256) * NP + pt * TB2_TILE) * 2;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int q = w + 8 * k;
      tb3_dma(dbase + (long)q * 8 * NP * 2, dp_lane, lds_dp + (unsigned)buf * 32768u + (unsigned)q * 1024u);
    }
  };

  int pb = 0, pline = 0, ppt = 0, par = 0;
  bool have_prev = false;
  unsigned t = blockIdx.x;
  if (t < ntiles) issue(t, 0);
  for (; t < ntiles; t += gridDim.x) {
    int pt, line, b;
    coords(t, pt, line, b);
    const int pos0 = pt * TB2_TILE;
    const long cell00 = p.swap ? ((long)b * N + pos0) * N + line : ((long)b * N + line) * N + pos0;

    // ---- P1: LayerNorm of this wave's 8 cells -> bf16 A tile; statistics to HBM; mask row ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the z rows AND the LDS-DMA pieces of this tile (issued one tile ago)
#pragma unroll
    for (int qd = 0; qd < 2; ++qd) {
      const int row = w * 8 + qd * 4 + l4;
      float x[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        x[i] = zr[qd][0][i];
        x[4 + i] = zr[qd][1][i];
      }
      const float mean = tb2_row16_sum(((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]))) * (1.f / 128.f);
      float q2 = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        x[i] -= mean;
        q2 = __builtin_fmaf(x[i], x[i], q2);
      }
      const float rstd = rsqrtf(tb2_row16_sum(q2) * (1.f / 128.f) + p.eps);
#pragma unroll
      for (int i = 0; i < 8; ++i) x[i] = __builtin_fmaf(x[i] * rstd, ldsGB[l15 * 8 + i], ldsGB[128 + l15 * 8 + i]);
      *(uint4*)(ldsA + tb2_a_off(row, l15)) =
          make_uint4(pack2bf_hw(x[0], x[1]), pack2bf_hw(x[2], x[3]), pack2bf_hw(x[4], x[5]), pack2bf_hw(x[6], x[7]));
      const bool live = pos0 + row < N;
      if (l15 == 0) {
        ldsM[row] = live ? mk[qd] : 0.f;
        if (live) {
          const long cell = cell00 + (long)row * cstride;
          p.stats[2 * cell] = mean;
          p.stats[2 * cell + 1] = rstd;
        }
      }
    }
    __syncthreads();

    // ---- P2: next tile's rows + da | db tile; stores: d4 rows of the previous tile (staging), zn rows of this tile (A) ----
    issue(t + gridDim.x < ntiles ? t + gridDim.x : t, par ^ 1);
    {
      tbu32x4 sv[4];
      if (have_prev) {
        const long pc0 = p.swap ? ((long)pb * N + ppt * TB2_TILE) * N + pline : ((long)pb * N + pline) * N + ppt * TB2_TILE;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int id = tid + 512 * (4 * h + k), cr = id >> 6, v = id & 63;
            sv[k] = *(const tbu32x4*)(ldsS + cr * TB3_SPITCH + v * 16);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int id = tid + 512 * (4 * h + k), cr = id >> 6, v = id & 63;
            if (ppt * TB2_TILE + cr < N) *(tbu32x4*)(p.d5 + (pc0 + (long)cr * cstride) * p.ld5 + v * 8) = sv[k];
          }
        }
      }
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int id = tid + 512 * k, cr = id >> 4, v = id & 15;
        const tbu32x4 z4 = *(const tbu32x4*)(ldsA + tb2_a_off(cr, v));
        if (pos0 + cr < N) *(tbu32x4*)(p.zn + (cell00 + (long)cr * cstride) * 128 + v * 8) = z4;
      }
    }
    __syncthreads();

    // ---- P3: projections on MFMA (pipelined over the row tiles like the forward), gate backward, staging ----
    const char* const dp = ldsDP + par * 32768;
    auto frags_rt = [&](int rt, bf16x8 (&af)[4]) __attribute__((always_inline)) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) af[ks] = *(const bf16x8*)(ldsA + tb2_a_off(rt * 16 + l15, ks * 4 + l4));
    };
    auto mma_ks = [&](int ks, const bf16x8 (&af)[4], f32x4 (&acc)[NG]) __attribute__((always_inline)) {
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const f32x4 c0 = {bv[g], bv[g], bv[g], bv[g]};
        acc[g] = TB2_MFMA(af[ks], wf[g][ks], ks == 0 ? c0 : acc[g]);
      }
    };
    const int ch = 16 * w + l15;
    // da / db of this lane's channel, cells rt * 16 + l4 * 4 .. + 4: 8 bytes of plane row ch (a) / 128 + ch (b)
    const int keya = (ch >> 1) & 7;                 // (row 128 + ch has the same key: 128 >> 1 is a multiple of 8)
    auto gate_rt = [&](int rt, const f32x4 (&acc)[NG]) __attribute__((always_inline)) {
      const int cell0 = rt * 16 + l4 * 4;
      const int coff = (((rt * 2 + (l4 >> 1)) ^ keya) << 4) + (l4 & 1) * 8;
      const uint2 ua = *(const uint2*)(dp + ch * 128 + coff);
      const uint2 ub = *(const uint2*)(dp + (128 + ch) * 128 + coff);
      const float da[4] = {bf_lo(ua.x), bf_hi(ua.x), bf_lo(ua.y), bf_hi(ua.y)};
      const float db[4] = {bf_lo(ub.x), bf_hi(ub.x), bf_lo(ub.y), bf_hi(ub.y)};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float m = ldsM[cell0 + r];
        const float sa = tb3_sigm(acc[1][r]), sb = tb3_sigm(acc[3][r]);
        const float g0 = da[r] * sa * m, g1 = da[r] * acc[0][r] * m * sa * (1.f - sa);
        const float g2 = db[r] * sb * m, g3 = db[r] * acc[2][r] * m * sb * (1.f - sb);
        char* row = ldsS + (cell0 + r) * TB3_SPITCH + ch * 2;
        *(bf16_t*)(row) = f2bf_hw(g0);
        *(bf16_t*)(row + 256) = f2bf_hw(g1);
        *(bf16_t*)(row + 512) = f2bf_hw(g2);
        *(bf16_t*)(row + 768) = f2bf_hw(g3);
        const float lv = pos0 + cell0 + r < N ? 1.f : 0.f;
        dbacc[0] = __builtin_fmaf(lv, g0, dbacc[0]);
        dbacc[1] = __builtin_fmaf(lv, g1, dbacc[1]);
        dbacc[2] = __builtin_fmaf(lv, g2, dbacc[2]);
        dbacc[3] = __builtin_fmaf(lv, g3, dbacc[3]);
      }
    };
    {
      f32x4 accA[NG], accB[NG];
      bf16x8 af[4];
      frags_rt(0, af);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) mma_ks(ks, af, accA);
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        f32x4 (&cur)[NG] = (rt & 1) ? accB : accA;
        f32x4 (&nxt)[NG] = (rt & 1) ? accA : accB;
        if (rt + 1 < 4) {          // the products of row tile rt + 1 run on the matrix pipe under the gate arithmetic of row tile rt
          frags_rt(rt + 1, af);
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) mma_ks(ks, af, nxt);
        }
        gate_rt(rt, cur);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    pb = b; pline = line; ppt = pt; par ^= 1; have_prev = true;
    __syncthreads();           // the A tile is single-buffered: the next tile's LayerNorm overwrites it
  }
  // last tile's d4 rows
  if (have_prev) {
    const long pc0 = p.swap ? ((long)pb * N + ppt * TB2_TILE) * N + pline : ((long)pb * N + pline) * N + ppt * TB2_TILE;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int id = tid + 512 * k, cr = id >> 6, v = id & 63;
      if (ppt * TB2_TILE + cr < N) *(tbu32x4*)(p.d5 + (pc0 + (long)cr * cstride) * p.ld5 + v * 8) = *(const tbu32x4*)(ldsS + cr * TB3_SPITCH + v * 16);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the surplus prefetch (z rows, DMA pieces) of the last iteration
  // bias gradients: the four l4 groups of a wave hold the same channel
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    float a = dbacc[g];
    a += __shfl_xor(a, 16, 64);
    a += __shfl_xor(a, 32, 64);
    if (l4 == 0) atomicAdd(p.dbias + g * 128 + 16 * w + l15, a);
  }
}

extern "C" int dfold_trimul_proj_bwd(const void* z, int32_t z_is_bf16, const float* mask, const float* ln_gamma, const float* ln_beta,
                                     const void* w_cat_bf16, const float* bias_cat, const void* dplanes_bf16, void* d5_bf16,
                                     int64_t d5_ld, void* zn_bf16, float* stats, float* d_bias, int32_t B, int32_t N, int32_t NP,
                                     int32_t incoming, float eps, void* stream) {
  if (!z || !mask || !ln_gamma || !ln_beta || !w_cat_bf16 || !bias_cat || !dplanes_bf16 || !d5_bf16 || !zn_bf16 || !stats || !d_bias)
    return DFOLD_EINVAL;
  if (B <= 0 || N <= 0 || NP < N || (NP % TB2_TILE) || d5_ld < 512 || (d5_ld & 7) || (long)B * N * (NP / TB2_TILE) >= (1L << 31) ||
      512L * NP >= (1L << 32) || 64L * N * 512 >= (1L << 32))
    return DFOLD_EINVAL;
  TriMulProjBwdParams p;
  p.x = z; p.mask = mask; p.gamma = ln_gamma; p.beta = ln_beta; p.W = (const bf16_t*)w_cat_bf16; p.bias = bias_cat;
  p.dpl = (const bf16_t*)dplanes_bf16; p.d5 = (bf16_t*)d5_bf16; p.zn = (bf16_t*)zn_bf16; p.stats = stats; p.dbias = d_bias;
  p.ld5 = d5_ld; p.B = B; p.N = N; p.NP = NP; p.swap = incoming ? 1 : 0; p.eps = eps;
  const long ntiles = (long)B * N * (NP / TB2_TILE);
  const long grid = ntiles < tb2_num_cus() ? ntiles : tb2_num_cus();
  if (z_is_bf16) {
    DFOLD_MAX_LDS_ONCE((trimul_proj_bwd_kernel<true>), TB3_LDS);
    DFOLD_LAUNCH((trimul_proj_bwd_kernel<true>), dim3((unsigned)grid), dim3(512), (size_t)TB3_LDS, (hipStream_t)stream, p);
  } else {
    DFOLD_MAX_LDS_ONCE((trimul_proj_bwd_kernel<false>), TB3_LDS);
    DFOLD_LAUNCH((trimul_proj_bwd_kernel<false>), dim3((unsigned)grid), dim3(512), (size_t)TB3_LDS, (hipStream_t)stream, p);
  }
  return dfold_check_launch();
}
