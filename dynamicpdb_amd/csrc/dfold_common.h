// Shared device helpers for the DFOLDv2 gfx950 kernels (wave64, CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;  // raw bfloat16 bits; all bf16 tensors cross the C ABI as uint16_t*
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define DFOLD_OK 0
#define DFOLD_EINVAL (-1)
#define DFOLD_ELAUNCH (-2)

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round-to-nearest-even fp32 -> bf16 (NaN stays NaN)
__device__ __forceinline__ bf16_t f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}
// hardware conversions (gfx950 v_cvt_pk_bf16_f32, round-to-nearest-even: same bits as f2bf for finite inputs)
typedef __attribute__((ext_vector_type(2))) __bf16 dfold_bf16x2;
typedef __attribute__((ext_vector_type(2))) float dfold_f32x2;
__device__ __forceinline__ uint32_t pack2bf_hw(float lo, float hi) {
  dfold_f32x2 v = {lo, hi};
  dfold_bf16x2 b = __builtin_convertvector(v, dfold_bf16x2);
  return *(uint32_t*)&b;
}
__device__ __forceinline__ bf16_t f2bf_hw(float f) {
  __bf16 b = (__bf16)f;
  return *(bf16_t*)&b;
}
__device__ __forceinline__ float bf_lo(uint32_t pair) { return __uint_as_float(pair << 16); }
__device__ __forceinline__ float bf_hi(uint32_t pair) { return __uint_as_float(pair & 0xffff0000u); }
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
}

// Wave64 reductions on the DPP path (VALU cross-lane moves: quad_perm, row_shr, row_bcast), result broadcast from
// lane 63 with v_readlane -- ~6 dependent VALU ops instead of 6 dependent ds_bpermute round trips through the LDS
// crossbar (which made kernels with dozens of reductions per row latency-bound).
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_mov_f(float oldv, float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(oldv), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_mov_f<0xb1>(0.f, v);        // quad_perm [1,0,3,2]
  v += dpp_mov_f<0x4e>(0.f, v);        // quad_perm [2,3,0,1]
  v += dpp_mov_f<0x114>(0.f, v);       // row_shr 4
  v += dpp_mov_f<0x118>(0.f, v);       // row_shr 8
  v += dpp_mov_f<0x142, 0xa>(0.f, v);  // row_bcast 15 -> rows 1,3
  v += dpp_mov_f<0x143, 0xc>(0.f, v);  // row_bcast 31 -> rows 2,3
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_mov_f<0xb1>(v, v));
  v = fmaxf(v, dpp_mov_f<0x4e>(v, v));
  v = fmaxf(v, dpp_mov_f<0x114>(v, v));
  v = fmaxf(v, dpp_mov_f<0x118>(v, v));
  v = fmaxf(v, dpp_mov_f<0x142, 0xa>(v, v));
  v = fmaxf(v, dpp_mov_f<0x143, 0xc>(v, v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Row map: logical GEMM row m -> element offset of that row in a tensor.
//   mode 0: base + m*ld
//   mode 1: frame x residue grid stored zero-padded for the 5x5 conv tower:
//           m = (w*F + f)*N + n  ->  base + (((w*Fp + f)*Wp) + n)*ld   (ld = channels)
//   mode 2 (round 6, any N_res): the same grid walked as ONE line of cells per window -- row m = (w, v), v < VW =
//           f*wp rounded up to whole 256-row runs, is cell v of window w counted along the padded frame rows (pad columns
//           included): base + ((w*fp*wp) + v)*ld.  A run of 256 consecutive rows is 256 consecutive cells whatever N is (the conv
//           taps stay constant offsets); rows that fall on a pad column (v % wp >= n) or behind the last frame (v >= f*wp) are
//           computed and never stored (row_valid).
struct RowMap {
  long base;
  long ld;
  int mode, n, f, fp, wp;
};
__device__ __forceinline__ unsigned row_vw(const RowMap& r) { return ((unsigned)(r.f * r.wp) + 255u) & ~255u; }
__device__ __forceinline__ bool row_valid(const RowMap& r, long m) {
  if (r.mode != 2) return true;
  const unsigned vw = row_vw(r), um = (unsigned)m;
  const unsigned v = um - (um / vw) * vw;
  return v < (unsigned)(r.f * r.wp) && v - (v / (unsigned)r.wp) * (unsigned)r.wp < (unsigned)r.n;
}
__device__ __forceinline__ long row_off(const RowMap& r, long m) {
  if (r.mode == 0) return r.base + m * r.ld;
  if (r.mode == 2) {
    const unsigned vw = row_vw(r), um = (unsigned)m;
    const unsigned w = um / vw, v = um - w * vw;
    return r.base + ((long)w * (r.fp * r.wp) + v) * r.ld;
  }
  // logical row counts are < 2^31 (int32 M at the ABI): 32-bit divisions (a 64-bit one costs ~10x more VALU)
  const unsigned um = (unsigned)m;
  const unsigned wf = um / (unsigned)r.n;
  const unsigned n = um - wf * (unsigned)r.n;
  const unsigned w = wf / (unsigned)r.f;
  const unsigned f = wf - w * (unsigned)r.f;
  return r.base + (((long)(w * (unsigned)r.fp + f) * r.wp) + n) * r.ld;
}

// hipGetLastError() reports the last error of ANY earlier runtime call on this thread (e.g. a benign
// hipErrorNotReady from an event query made by the host framework): clear it before every launch so that
// dfold_check_launch() reflects this launch only.
#define DFOLD_LAUNCH(...)          \
  do {                             \
    (void)hipGetLastError();       \
    hipLaunchKernelGGL(__VA_ARGS__); \
  } while (0)

// Raise a kernel's dynamic-LDS limit ONCE per process (a driver call: not on the per-launch path).  One static flag per
// expansion site, i.e. per kernel instantiation named there.
#define DFOLD_MAX_LDS_ONCE(kernel, bytes)                                                                 \
  do {                                                                                                    \
    static bool dfold_attr_done_ = false;                                                                 \
    if (!dfold_attr_done_) {                                                                              \
      (void)hipFuncSetAttribute((const void*)(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (bytes)); \
      dfold_attr_done_ = true;                                                                            \
    }                                                                                                     \
  } while (0)

static inline int dfold_check_launch() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? DFOLD_OK : DFOLD_ELAUNCH;
}
