/* dfold_hip.h -- C ABI of libdfold_hip.so, the MI355X (gfx950) engine for the DFOLDv2
 * trajectory-prediction hot path of fudan-generative-vision/dynamicPDB.
 *
 * The reference has NO native/FFI layer (it is pure PyTorch); its operator
 * boundary is a set of Python nn.Module / diffuser methods.  Each entry point below cites the
 * reference operator (path:line under the reference tree) whose device arithmetic it replaces.
 * The Python mirror of those operators (dynamicpdb_amd/model, dynamicpdb_amd/data) binds these
 * symbols with ctypes (dynamicpdb_amd/_lib.py); INTEGRATION.md shows the stub a reference
 * maintainer would add.
 *
 * Conventions
 *   - plain pointers + sizes only; every pointer is a DEVICE pointer owned by the caller (incl. all
 *     workspace); nothing is allocated, freed or synchronised inside the library.
 *   - `stream` is a hipStream_t (NULL = default stream); every call is asynchronous and stream-ordered.
 *   - return 0 on success, DFOLD_EINVAL (-1) for a rejected argument, DFOLD_ELAUNCH (-2) if the
 *     launch failed.  The Python layer maps non-zero to ValueError / RuntimeError.
 *   - "bf16" tensors are raw uint16 bfloat16 bits; geometry (rigid frames, points, scores) is fp32
 *     (fp64 accumulation inside the IGSO(3) series like the reference, src/data/so3_diffuser.py:301).
 *   - rigid frames are tensor_7 rows [qw,qx,qy,qz,tx,ty,tz] (openfold/utils/rigid_utils.py:1200-1230).
 */
#ifndef DFOLD_HIP_H
#define DFOLD_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define DFOLD_ABI_VERSION 1
int dfold_abi_version(void);

/* ------------------------------------------------------------------------------------------------
 * Dense contraction engine (bf16 MFMA, fp32 accumulate).
 *   C[m,n] = epi( alpha * sum_{seg,k} A[arow(m) + a_seg_off[seg] + k] * B[n*ldb + b_seg_off[seg] + k] )
 * replaces aten conv2d / linear / matmul at: ConvNet src/model/ipa_pytorch_dynamic.py:664-706
 * (implicit GEMM over the zero-padded [window, F+4, N+4, C] grid, fwd / dgrad / wgrad),
 * IPA projections :350-396,:498-514, AngleResnet openfold/model/structure_module.py:114-158,
 * BackboneUpdate :600, expand_node/edge src/model/Dfold_network_dynamic.py:473-474, the triangle
 * operators' Linear layers openfold/model/triangular_multiplicative_update.py:97-126 and
 * triangular_attention.py:105-139, and the batched attention products ipa_pytorch_dynamic.py:402,452,499.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  int64_t base; /* element offset of logical row 0 */
  int64_t ld;   /* mode 0: row stride; mode 1: channels per grid cell */
  int32_t mode; /* 0: off = base + m*ld
                   1: m = (w*f + fr)*n + res  ->  off = base + (((w*fp + fr)*wp) + res)*ld */
  int32_t n, f, fp, wp;
} dfold_rowmap;

#define DFOLD_GEMM_BIAS 1      /* v += bias[n] */
#define DFOLD_GEMM_RELU 2      /* v = max(v,0)            (after bias) */
#define DFOLD_GEMM_RESID 4     /* v += R[off]             (after relu; R bf16, laid out like C) */
#define DFOLD_GEMM_RELUMASK 8  /* v = R[off] > 0 ? v : 0  (ReLU backward) */
#define DFOLD_GEMM_OUT_BF16 16 /* C is bf16 (else fp32) */
#define DFOLD_GEMM_ACCUM 32    /* fp32 C += v */

typedef struct {
  const void* A;        /* bf16 */
  const void* B;        /* bf16 [N][ldb], K-contiguous */
  void* C;              /* fp32 or bf16 */
  void* C2;             /* optional second bf16 output laid out like C: if R2==NULL the value before the
                           residual add, else (R2[off] > 0 ? v : 0) */
  const float* bias;    /* [N] */
  const void* R;        /* bf16, see flags */
  const void* R2;       /* bf16 mask for C2 */
  const void* zeros;    /* >= 16 bytes of zeros, 16-B aligned (source for out-of-range tile cells) */
  const int64_t* a_seg_off; /* [nseg] element offsets added to every A row, NULL -> seg*seglen */
  const int64_t* b_seg_off; /* [nseg] element offsets added to every B row, NULL -> seg*seglen */
  dfold_rowmap a_rows, c_rows;
  int64_t ldb;
  int64_t sa0, sa1, sb0, sb1, sc0, sc1; /* batch strides (elements): batch z -> (z / nb1, z % nb1) */
  int32_t M, N, nseg, seglen, nbatch, nb1, flags;
  float alpha;
} dfold_gemm_desc;

int dfold_gemm_bf16(const dfold_gemm_desc* desc, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Layout helpers of the conv tower (ConvNet, src/model/ipa_pytorch_dynamic.py:664-706; its backward is
 * aten convolution_backward in the reference).
 * ---------------------------------------------------------------------------------------------- */
int dfold_cast_f32_bf16(const float* src, void* dst_bf16, int64_t n, void* stream);
int dfold_cast_bf16_f32(const void* src_bf16, float* dst, int64_t n, void* stream);
/* W fp32 [CO][CI][5][5] (nn.Conv2d OIHW, :669-690) -> Wf bf16 [CO][25][CI], Wd bf16 [CI][25][CO] (taps flipped) */
int dfold_conv_weight_pack(const float* W, void* Wf, void* Wd, int32_t CO, int32_t CI, void* stream);
/* dWg fp32 [CO][25][CI] -> G fp32 [CO][CI][5][5]; accumulate != 0: G += */
int dfold_conv_wgrad_unpack(const float* dWg, float* G, int32_t CO, int32_t CI, int32_t accumulate, void* stream);
/* X bf16 [W][Fp][Wp][C] -> T bf16 [nd][C][W][Fp][N], T[d][c][w][f][n] = X[w][f][n+d0+d][c] */
int dfold_grid_transpose_shift(const void* X, void* T, int32_t W, int32_t Fp, int32_t Wp, int32_t C, int32_t N,
                               int32_t d0, int32_t nd, void* stream);
/* out[c] += sum_r X[r*ld + c]   (X bf16, out fp32, atomics) */
int dfold_colsum_bf16(const void* X, float* out, int64_t R, int32_t C, int64_t ld, void* stream);
/* out = v > 0 ? g : 0  (bf16) */
int dfold_relu_mask_bf16(const void* g, const void* v, void* out, int64_t n, void* stream);
/* batched 2-D transpose dst[b][c][r] = src[b][r][c] (bf16; strides in elements) */
int dfold_transpose_bf16(const void* src, void* dst, int32_t R, int32_t C, int64_t ld_src, int64_t ld_dst,
                         int32_t nbatch, int64_t bs_src, int64_t bs_dst, void* stream);

#ifdef __cplusplus
}
#endif
#endif
