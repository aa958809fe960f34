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
 *     workspace); nothing is allocated, freed or synchronised inside the library -- with one exception:
 *     dfold_gemm_bf16 with splitk != 1 on a 5x5 conv launch keeps a fine-grained (L2-uncached,
 *     hipDeviceMallocFinegrained) buffer per (device, stream) for the partial tiles that travel between XCDs
 *     (launches in flight on two streams never share slots; the table is guarded by a mutex), allocated on
 *     first use and grown -- with a synchronise of THAT stream -- when a launch needs more; never inside a
 *     graph capture.  If that is not possible (capture, allocation failure, more than 64 streams) the caller's
 *     splitk_ws is used with device-wide fences.  The hand-over through that buffer relies on gfx950 not
 *     caching fine-grained memory in the L2s / vector L1 (each slot is read once per launch); no other target.
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

/* Bumped whenever a signature, a pointer's element type or a struct layout changes (2: round-3 changes of
 * dfold_ipa_softmax_bwd / dfold_ipa_col_bwd (bf16 probabilities, `ctr`) and dfold_ipa_bias_grad (`nh_pitch`); the
 * round-4 additions; 3: round 6, the zero-frame fields at the end of dfold_gemm_desc, the two extra arguments of
 * dfold_conv_wgrad_tn, dfold_grid_load_flags).  The Python binding reads the number from THIS header and refuses a library that reports
 * another one (dynamicpdb_amd/_lib.py), so a stale or variant .so cannot be called with shifted arguments. */
#define DFOLD_ABI_VERSION 3
int dfold_abi_version(void);

/* ------------------------------------------------------------------------------------------------
 * Dense contraction engine (bf16 MFMA, fp32 accumulate).
 *   C[m,n] = epi( alpha * sum_{seg,k} A[arow(m) + a_off(seg) + k] * B[n*ldb + b_off(seg) + k] )
 * replaces aten conv2d / linear / matmul at: ConvNet src/model/ipa_pytorch_dynamic.py:664-706
 * (implicit GEMM over the zero-padded [window, F+4, N+4, C] grid, fwd / dgrad / wgrad),
 * IPA projections :350-396,:498-514, AngleResnet openfold/model/structure_module.py:114-158,
 * BackboneUpdate :600, expand_node/edge src/model/Dfold_network_dynamic.py:473-474, the triangle
 * operators' Linear layers openfold/model/triangular_multiplicative_update.py:97-126 and
 * triangular_attention.py:105-139, and the batched attention products ipa_pytorch_dynamic.py:402,452,499.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  int64_t base; /* element offset of logical row 0 */
  int64_t ld;   /* mode 0: row stride; mode 1 / 2: channels per grid cell */
  int32_t mode; /* 0: off = base + m*ld
                   1: m = (w*f + fr)*n + res  ->  off = base + (((w*fp + fr)*wp) + res)*ld
                   2 (round 6; 5x5 conv launches, any N_res): m = w*VW + v, VW = f*wp rounded up to a multiple of 256; row v is
                      cell v of window w counted along the padded frame rows, pad columns included:
                      off = base + (w*fp*wp + v)*ld.  Rows with v >= f*wp or v % wp >= n are computed and not stored.  Both maps
                      of a launch must be mode 2 over the same grid geometry, M = windows * VW; the A tensor must stay readable
                      (finite values) for 272 cells + 4 frame rows behind its last window (rows that are never stored read there) */
  int32_t n, f, fp, wp;
} dfold_rowmap;

#define DFOLD_GEMM_BIAS 1      /* v += bias[n] */
#define DFOLD_GEMM_RELU 2      /* v = max(v,0)            (after bias) */
#define DFOLD_GEMM_RESID 4     /* v += R[off]             (after relu; R bf16, laid out like C) */
#define DFOLD_GEMM_RELUMASK 8  /* v = R[off] > 0 ? v : 0  (ReLU backward) */
#define DFOLD_GEMM_OUT_BF16 16 /* C is bf16 (else fp32) */
#define DFOLD_GEMM_ACCUM 32    /* fp32 C += v */
#define DFOLD_GEMM_ATOMIC 64   /* fp32 atomicAdd(C, v): split-K slices (batches that share C) */
#define DFOLD_GEMM_C2RELU 128  /* bf16 C with a second output C2 (R2 == NULL): C2 = max(value written to C, 0) -- the next layer's
                                  ReLU'd operand leaves with the residual stream (AngleResnet, structure_module.py:64-71) */
#define DFOLD_GEMM_MASK2 256   /* v = R2[off] > 0 ? v : 0 BEFORE the residual add (R2 bf16; no C2): the ReLU backward of a
                                  branch that joins a residual gradient stream in the same launch */
#define DFOLD_GEMM_NZ_KEEP 512 /* zero-frame-flagged 5x5 conv launch (nz_ps): the caller guarantees that C (and C2) already hold, on
                                * every tile the flags call dead, what the epilogue of a zero product would write there -- those
                                * tiles are left alone (round 6: the tower's backward, whose scratch grids are zeroed per call and
                                * whose dead region only shrinks from stage to stage; a dead tile's epilogue is a pure copy of zeros
                                * at a quarter of the HBM rate) */

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
  /* element offset added to every A (B) row for K segment g = (hi, mid, lo), lo = g % seg_div,
     mid = (g / seg_div) % seg_div_mid (seg_div_mid <= 0: unbounded), hi = the rest:
         seg0 + hi*seg_s0 + mid*seg_s1 + lo*seg_s2
     conv implicit GEMM: hi = 64-channel chunk, mid = df, lo = dn (channel-chunk-outer order keeps the 25 shifted
     re-reads of an activation chunk in the XCD's L2); plain contiguous K: seg_s1 = seglen, seg_div = 1 */
  int64_t a_seg0, a_seg_s0, a_seg_s1, a_seg_s2;
  int64_t b_seg0, b_seg_s0, b_seg_s1, b_seg_s2;
  int32_t seg_div, seg_div_mid;
  dfold_rowmap a_rows, c_rows;
  int64_t ldb;
  int64_t sa0, sa1, sb0, sb1, sc0, sc1; /* batch strides (elements): batch z -> (z / nb1, z % nb1) */
  int32_t M, N, nseg, seglen, nbatch, nb1, flags;
  float alpha;
  /* Optional deterministic split-K of the conv implicit GEMM (a_rows.mode == 1, seg_div == seg_div_mid == 5, nbatch == 1):
     splitk = S > 1 splits the channel-chunk axis (nseg / 25 chunks, a multiple of S) over S workgroups per output tile.
     Each writes its fp32 partial tile to splitk_ws, the last one to arrive (splitk_cnt, one int32 per tile, zero before
     and after the call) sums the S partials in fixed order and runs the epilogue -- narrow launches (few output rows,
     long K) then fill the 256 CUs.  splitk_ws: >= S * tiles * 256 * Ntile fp32 (tiles = ceil(M/256) * N/Ntile, Ntile = 320).
     splitk = -1 ("stream-K", 5x5 conv launches that take the 512 x 160 one-wave-per-SIMD kernel): one persistent workgroup per
     CU, each an equal share of the (tile, K group) units in tile-major order; shared tiles are added up by their last arriver in
     workgroup order (deterministic).  splitk_ws >= 2 * CUs * 512 * 160 fp32, splitk_cnt >= tiles int32 (zero before the first
     launch, left zero).  A launch that does not qualify for that kernel runs unsplit. */
  int32_t splitk;
  /* 5x5 conv launches (a_rows.mode = 1, seg_div = seg_div_mid = 5) may declare where the frame axis of the grid ends:
     conv_frames = (f_first << 16) | F_total, f_first = grid frame of logical frame 0 of a_rows, F_total = frames of the grid.
     A 256-row output tile then skips the frame taps that only read the zero border above / below the grid (their
     products are exact zeros).  0 = walk all 25 taps. */
  int32_t conv_frames;
  float* splitk_ws;
  int32_t* splitk_cnt;
  /* Zero-frame skipping of a 5x5 conv launch (optional; round 6).  nz_ps = the per-window prefix sums that
     dfold_grid_load_flags writes for some grid G0: int32 [windows][fp + 1], nz_ps[w][i] = number of padded frame rows j < i of
     window w in which G0 holds a non-zero.  The caller asserts that A is exactly zero on every padded frame row that is more
     than nz_radius rows away from all non-zero rows of G0 (true by construction when A was produced from G0 by nz_radius / 2
     5x5 convolutions, element-wise masks and sums with grids of the same property: the data-gradient chain of the tower).
     An output tile all of whose input frame rows are zero by that statement does not walk K: its accumulators are zero and
     the epilogue runs as usual (results identical to the full walk: the skipped products are exact zeros).  Decided on the
     device per tile; no host synchronisation.  nz_f0 = padded frame row of the first tap row of logical frame 0 of a_rows.
     Launches that do not take the 512 x 160 kernel, and stream-K launches, ignore the fields (same results).
     Plain row maps (a_rows.mode == 0, round 6): nz_ps = the prefix sums of dfold_row_block_flags over blocks of nz_f0 rows of A
     (nz_radius unused): a 256-row output tile whose A rows are all zero by them skips its K walk (accumulators zero, epilogue
     as usual) -- the dx products of a dense layer whose incoming gradient lives on a few frames.  Honoured by the 256 x 320
     kernel's dense form, ignored elsewhere. */
  const int32_t* nz_ps;
  int32_t nz_radius, nz_f0;
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
/* dWg fp32 [CO][25][CI] (transposed != 0: [CI][25][CO]) -> G fp32 [CO][CI][5][5]; accumulate != 0: G += */
int dfold_conv_wgrad_unpack(const float* dWg, float* G, int32_t CO, int32_t CI, int32_t accumulate, int32_t transposed,
                            void* stream);
/* X bf16 [W][Fp][Wp][C] -> T bf16 [nd][C][W][Fp][NP], T[d][c][w][f][n] = X[w][f][n+d0+d][c] (n < N <= NP; the tail
   n in [N,NP) is not written and must be zero) for the padded frame rows f in [f0, f0+nf) (other rows of T untouched);
   colsum (optional, fp32 [C], atomics): += per-channel sum over the interior cells of those rows (the conv bias
   gradient, fused) */
int dfold_grid_transpose_shift(const void* X, void* T, int32_t W, int32_t Fp, int32_t Wp, int32_t C, int32_t N, int32_t NP,
                               int32_t d0, int32_t nd, int32_t f0, int32_t nf, float* colsum, void* stream);
/* "TN" product on the bf16 MFMA engine, both operands with the reduction index as the slow axis (csrc/tn_gemm.hip; no
   transposed copies -- ds_read_b64_tr_b16 fragments):
     C[z][m][n] (+)= alpha * sum_{k < K} A[z][k][m] * B[z][k][n]      A bf16 rows of lda, B bf16 rows of ldb elements
   = the weight gradient of a dense layer (dW = dY^T X, torch.nn.Linear autograd) and the transposed attention products of
   the IPA backward (dK = dS^T Q, dV = P^T dO; src/model/ipa_pytorch_dynamic.py:396-469).  Batch z -> (z / nb1, z % nb1)
   with element strides (sa0, sa1) / (sb0, sb1) / (sc0, sc1).  flags: DFOLD_GEMM_OUT_BF16 (C bf16, stored), DFOLD_GEMM_ACCUM
   (C fp32 += ), DFOLD_GEMM_ATOMIC (C fp32, atomics: required when splitk > 1 -- the K range is cut into splitk parts, C must
   hold the running sum / zeros), else C fp32 stored.  M, N multiples of 8 (256 x 256 output tiles; a tile over the edge stores only its valid part); K a multiple of 64 * splitk; lda, ldb, ldc and
   the A / B strides multiples of 8, A / B 16-byte aligned. */
int dfold_gemm_tn_bf16(const void* A, const void* B, void* C, int32_t M, int32_t N, int64_t K, int64_t lda, int64_t ldb,
                       int64_t ldc, int32_t nbatch, int32_t nb1, int64_t sa0, int64_t sa1, int64_t sb0, int64_t sb1,
                       int64_t sc0, int64_t sc1, int32_t splitk, int32_t flags, float alpha, void* stream);
/* dfold_gemm_tn_bf16 with row-block flags for A (the incoming gradient of a dense layer's weight-gradient product): ps = the
   prefix sums of dfold_row_block_flags over blocks of `block` reduction rows; a split-K part whose rows are all zero by them
   adds nothing and exits (needs DFOLD_GEMM_ATOMIC or DFOLD_GEMM_ACCUM: C already holds what the part would have left) */
int dfold_gemm_tn_bf16_rowflags(const void* A, const void* B, void* C, int32_t M, int32_t N, int64_t K, int64_t lda, int64_t ldb,
                                int64_t ldc, int32_t nbatch, int32_t nb1, int64_t sa0, int64_t sa1, int64_t sb0, int64_t sb1,
                                int64_t sc0, int64_t sc1, int32_t splitk, int32_t flags, float alpha, const int32_t* ps,
                                int32_t block, void* stream);
/* Which blocks of `block` consecutive rows of G (fp32 [R][C], rows of ld elements) hold a non-zero: ps int32 [nblocks + 1],
   ps[i] = number of non-zero blocks j < i (nblocks = ceil(R / block)); scratch: nblocks int32, zero before the call and left
   zero.  The gradient that reaches a per-position head from a loss that reads some frames only (train_DFOLD_dynamics.py:
   1219-1340: the last one) is zero on all other rows; the dense backward launches skip them on the device. */
int dfold_row_block_flags(const float* G, int32_t* ps, int32_t* scratch, int64_t R, int32_t C, int64_t ld, int32_t block,
                          void* stream);
/* 5x5 conv weight gradient straight from the zero-padded channels-last grids (the autograd of nn.Conv2d,
   src/model/ipa_pytorch_dynamic.py:669-690), no operand copies (csrc/conv_wgrad_tn.hip, ds_read_b64_tr_b16 fragments):
     dWg[a][tap][b] (+)= sum_{w < W, f0 <= f < f0+nf, n < N}  A[w][2+f][2+n][a] * B[w][f+z0][n+z1][b],   z0, z1 = 0..4,
     tap = 5 z0 + z1, or 24 - (5 z0 + z1) when flip != 0
   A bf16 [W][Fp][Wp][CA], B bf16 [W][Fp][Wp][CB], dWg fp32 [CA][25][CB] (overwritten unless accumulate != 0).
   With A = dL/dy (CA = CO), B = x: dWg = dW in the [CO][25][CI] layout of dfold_conv_wgrad_unpack; with A = x, B = dL/dy and
   flip: its transposed [CI][25][CO] form.  CA % 256 == 0, CB % 64 == 0, grids 16-byte aligned.  N % 64 != 0 (round 6): the frame
   range of a window is walked as one line of nf * Wp cells -- both grids must have zero pad columns / border rows (they do when
   the engine wrote them) and B must stay readable (finite values) for 68 cells behind its last window.
   nz_ps / nz_radius (optional, NULL / 0 = none): the frame flags of dfold_grid_load_flags for the GRADIENT operand (A, or B
   when flip != 0), as in dfold_gemm_desc: frame rows of the reduction whose gradient cells are zero by that statement are
   left out of the K walk (exact zeros: the sum is unchanged bit for bit); needs nf <= 64, at most 8 windows per call. */
int dfold_conv_wgrad_tn(const void* A, const void* B, float* dWg, int32_t CA, int32_t CB, int32_t W, int32_t Fp, int32_t Wp,
                        int32_t N, int32_t f0, int32_t nf, int32_t flip, int32_t accumulate, const int32_t* nz_ps,
                        int32_t nz_radius, void* stream);
/* Copies src bf16 [W][nf][N][C] (dense) into the interior cells of frames [f_off, f_off + nf) of the zero-padded grid
   bf16 [W][F+4][N+4][C] (src NULL: the grid is read as it lies, nothing is written) and records which of those frame rows
   hold a non-zero: ps int32 [W][F+5], ps[w][i] = number of padded frame rows j < i of window w with a non-zero cell (rows
   outside the range count as zero -- the caller's statement).  scratch: (F+4) * W + 1 int32, zero before the call and left
   zero (two launches: copy + flags, then one block of prefix sums).  The gradient that enters the tower's backward (aten convolution_backward of ConvNet, ipa_pytorch_dynamic.py:692-706)
   is zero on every frame no loss term reads; the flags let the data / weight gradient launches skip those frames without a
   contract with the loss and without a device -> host copy.  N * C a multiple of 8, 16-byte aligned pointers. */
int dfold_grid_load_flags(const void* src, void* grid, int32_t* ps, int32_t* scratch, int32_t W, int32_t F, int32_t N,
                          int32_t C, int32_t f_off, int32_t nf, void* stream);
/* out[c] += sum_r X[r*ld + c]   (X bf16, out fp32, atomics) */
int dfold_colsum_bf16(const void* X, float* out, int64_t R, int32_t C, int64_t ld, void* stream);
/* the same over nbatch row blocks of R rows each, block z starting bstride elements after block z - 1 (one frame range of every
   window of a padded conv grid: the conv bias gradient of a cone launch, reference nn.Conv2d bias autograd) */
int dfold_colsum_bf16_batched(const void* X, float* out, int64_t R, int32_t C, int64_t ld, int32_t nbatch, int64_t bstride,
                              void* stream);
/* out = v > 0 ? g : 0  (bf16) */
int dfold_relu_mask_bf16(const void* g, const void* v, void* out, int64_t n, void* stream);
/* batched 2-D transpose dst[z][c][r] = src[z][r][c] (bf16; strides in elements; batch z -> (z / nb1, z % nb1)) */
int dfold_transpose_bf16(const void* src, void* dst, int32_t R, int32_t C, int64_t ld_src, int64_t ld_dst,
                         int32_t nbatch, int32_t nb1, int64_t bs_src0, int64_t bs_src1, int64_t bs_dst0,
                         int64_t bs_dst1, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Rigid-frame geometry of IPA (src/model/ipa_pytorch_dynamic.py:363-390 and :470-488), H=8, Pq=8, Pv=12; all fp32, one
 * workgroup per residue.  raw_q [P][3*64], raw_kv [P][3*160]: linear_q_points / linear_kv_points outputs (x|y|z blocks).
 * dt7 [P][7] = gradient w.r.t. the tensor_7 frame (quaternion via dL/dR of the quadratic-form matrix, translation).
 * ---------------------------------------------------------------------------------------------- */
int dfold_ipa_points_fwd(const float* raw_q, const float* raw_kv, const float* t7, float* q_pts, float* k_pts, float* v_pts,
                         int64_t P, void* stream);
int dfold_ipa_points_bwd(const float* raw_q, const float* raw_kv, const float* t7, const float* dq_pts, const float* dk_pts,
                         const float* dv_pts, float* draw_q, float* draw_kv, float* dt7, int64_t P, void* stream);
/* Pair-side projections of an IPA block in one pass over z (round 4): linear_b (ipa_pytorch_dynamic.py:396) and down_z (:498),
 * biases excluded (linear_b's drops out of the softmax, down_z's is added after the aggregation).  z bf16 [B][N][N][128],
 * w_b bf16 [8][128], w_dz bf16 [32][128] -> bias_t fp32 [B][8][N][N], pz bf16 [B][N][N][32], pzT bf16 [B][N][32][N].  N % 8 == 0. */
int dfold_ipa_pair_proj(const void* z_bf16, const void* w_b_bf16, const void* w_dz_bf16, float* bias_t, void* pz_bf16, void* pzT_bf16,
                        int32_t B, int32_t N, void* stream);
/* Pair-value side of the IPA attention as streaming kernels (round 6; ipa_pytorch_dynamic.py:498-502 and its autograd), one
 * workgroup per (window, query residue), every MFMA fragment a 16-byte global load, no transposed copies:
 *   fwd: out[((b F + f) N + i) ld + c_off + h 32 + c] = sum_j P[b][f][h][i][j] pzT[b][i][c][j] + b_dz[c]   (bf16; N % 16 == 0)
 *   bwd: dP[b][f][h][i][j] = sum_c do_pair[((b F + f) N + i) ld_dop + h 32 + c] pz[b][i][j][c]              (bf16; N % 4 == 0)
 * P, dP bf16 [B][F][H][N][N]; pz bf16 [B][N][N][32], pzT bf16 [B][N][32][N] (dfold_ipa_pair_proj); b_dz fp32 [32]. */
int dfold_ipa_pair_value_fwd(const void* P_bf16, const void* pzT_bf16, const float* b_dz, void* out_bf16, int32_t B, int32_t F,
                             int32_t N, int32_t H, int64_t ld, int64_t c_off, void* stream);
int dfold_ipa_pair_value_bwd(const void* do_pair_bf16, const void* pz_bf16, void* dP_bf16, int32_t B, int32_t F, int32_t N, int32_t H,
                             int64_t ld_dop, void* stream);
/* o_pt [P][8][12][3] global frame -> geo_l / geo_g bf16 [P][384] = [x|y|z|norm] of R^T(o_pt - t) and of o_pt (:470-488,504);
 * ld: row stride of geo_l / geo_g in elements (384, or the row of the concatenated feature matrix they are columns of) */
int dfold_ipa_outfeat_fwd(const float* o_pt, const float* t7, void* geo_l, void* geo_g, int64_t ld, int64_t P, float eps,
                          void* stream);
int dfold_ipa_outfeat_bwd(const float* o_pt, const float* t7, const void* dgeo_l, const void* dgeo_g, float* do_pt, float* dt7,
                          int64_t P, float eps, void* stream);

/* ------------------------------------------------------------------------------------------------
 * MyLayerNorm (src/model/ipa_pytorch_dynamic.py:709-724): whole-window statistics, unbiased variance,
 * eps inside the sqrt, no affine; optional fused SiLU (embedders :757-796).  x fp32 [W][n] -> y bf16.
 * stats: caller workspace of 2*W doubles; mean_rstd: 2*W floats kept for the backward.
 * ---------------------------------------------------------------------------------------------- */
int dfold_gln_fwd(const float* x, double* stats, void* y_bf16, float* mean_rstd, int32_t W, int64_t n, float eps,
                  int32_t silu, void* stream);
int dfold_gln_bwd(const float* x, const void* g_bf16, const float* mean_rstd, double* stats, void* dx_bf16, int32_t W,
                  int64_t n, int32_t silu, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Invariant Point Attention core (InvariantPointAttention.forward, src/model/ipa_pytorch_dynamic.py:396-469).
 * Layouts: S / P / dP / dS fp32 [B,F,H,N,N]; bias fp32 [B,H,N,N] (linear_b(z) :396, head-major);
 * q_pts,k_pts [B,F,N,H,8,3], v_pts / o_pt [B,F,N,H,12,3] fp32 in the GLOBAL frame (:363-390);
 * mask [B,F,N]; hw[H] = softplus(head_weights) * sqrt(1/(3*(Pq*9/2))) (:415-421).
 * ---------------------------------------------------------------------------------------------- */
/* P = softmax_j(S + bias_scale*bias - 0.5*hw*|q_pts_i - k_pts_j|^2 + inf*(m_i m_j - 1))  (:407-444) */
int dfold_ipa_softmax_fwd(const float* S, const float* bias, const float* q_pts, const float* k_pts, const float* mask,
                          const float* hw, float* P, void* P_bf16, int32_t B, int32_t F, int32_t N, int32_t H,
                          float bias_scale, float inf, void* stream);
/* o_pt[b,f,i,h,:] = sum_j P[b,f,h,i,j] v_pts[b,f,j,h,:]  (:460-469) */
int dfold_ipa_opt_fwd(const float* P, const float* v_pts, float* o_pt, int32_t B, int32_t F, int32_t N, int32_t H,
                      void* stream);
/* row pass of the backward: dS, dq_pts, dhw (dhw accumulated atomically: zero it first).  P_bf16: the forward's bf16
 * probabilities [B,F,H,N,N] (renormalised per row inside, so that the rows of dS sum to zero); ctr [B,F,3] or NULL: a
 * per-(window, frame) centre the points are taken relative to (the result is invariant; the fp32 cancellation in dhw is not) */
int dfold_ipa_softmax_bwd(const void* P_bf16, const float* dP, const float* q_pts, const float* k_pts, const float* v_pts,
                          const float* do_pt, const float* hw, float* dS, void* dS_bf16, float* dq_pts, float* dhw,
                          const float* ctr, int32_t B, int32_t F, int32_t N, int32_t H, void* stream);
/* Fused row pass of the backward (csrc/ipa_fused_bwd.hip; N % 8 == 0, N <= 512, 256 channels per head, 8 / 12 points): one launch
 * for  g = do v^T + do_pt (v_pts - ctr)^T + dP_pair,  dS = P (g - <g>_P)  (fp32 and bf16),  dq = alpha dS k (bf16),
 * dq_pts = hw dS (k_pts - ctr),  dhw (atomics: zero it first) -- the point terms ride on the matrix cores as bf16-split columns.
 * dfold_ipa_bwd_prep writes the augmented operands: DOP, VP bf16 [B*F,H,N,224] (do_pt / v_pts - ctr as three bf16 pieces, six
 * product blocks of 36) and rows 256..351 of KT bf16 [B*F,H,352,NP] (k_pts - ctr and |k_pts - ctr|^2, three pieces x 32 rows,
 * key-contiguous; rows 0..255 = k^T of the head are the caller's: dfold_transpose_bf16; NP = N rounded up to 64, pad columns and
 * rows 25..31 of each piece zero).  dP_pair_bf16 [B,F,H,N,N] or NULL: the pair-value term of dP. */
int dfold_ipa_bwd_prep(const float* do_pt, const float* v_pts, const float* k_pts, const float* ctr, void* DOP_bf16, void* VP_bf16,
                       void* KT_bf16, int32_t B, int32_t F, int32_t N, int32_t H, int32_t NP, void* stream);
int dfold_ipa_fused_bwd(const void* do_bf16, const void* kv_bf16, const void* DOP_bf16, const void* VP_bf16, const void* KT_bf16,
                        const void* P_bf16, const void* dP_pair_bf16, const float* q_pts, const float* hw, const float* ctr,
                        float* dS, void* dS_bf16, void* dq_bf16, float* dq_pts, float* dhw, int32_t B, int32_t F, int32_t N,
                        int32_t H, int32_t NP, float alpha, void* stream);
/* column pass of the backward: dk_pts, dv_pts (P_bf16 as above) */
int dfold_ipa_col_bwd(const void* P_bf16, const float* dS, const float* q_pts, const float* k_pts, const float* do_pt,
                      const float* hw, float* dk_pts, float* dv_pts, int32_t B, int32_t F, int32_t N, int32_t H,
                      void* stream);
/* dbias = scale * sum_f dS, bf16, as out_hn [B][H][N*N] and as out_nh: rows of 8 (H zero-padded to 8) per pair cell with a
 * row pitch of nh_pitch elements (multiple of 8; 8 = dense [B][N*N][8]) */
int dfold_ipa_bias_grad(const float* dS, void* out_hn, void* out_nh, int64_t nh_pitch, int32_t B, int32_t F, int32_t N,
                        int32_t H, float scale, void* stream);

/* Fused forward of the attention core (src/model/ipa_pytorch_dynamic.py:402-469: logits, softmax, o = a v, o_pt = a v_pts)
 * for one launch per IPA block: csrc/ipa_fused.hip.  N % 8 == 0, N <= 512, 8 query / 12 value points, 256 channels per head.
 * dfold_ipa_aug_prep builds the augmented MFMA operands from the global-frame points (ctr [B,F,3]: any per-(window, frame)
 * centre, e.g. the mean key point): QP, KP bf16 [B,F,H,N,160] (three-piece bf16 splits of the 24 point coordinates laid out as
 * the six products hh, hm, mh, hl, lh, mm; QP pre-scaled by hw / alpha), kn fp32 [B,F,H,N] = -hw/2 |k_pts - ctr|^2, and rows
 * 256..399 of VT bf16 [B,F,H,400,NP] (three pieces x 48 rows of v_pts - ctr, key-contiguous; rows 0..255 = v^T are the
 * caller's, e.g. dfold_transpose_bf16; NP % 64 == 0, pad columns zero).
 * dfold_ipa_fused_fwd: q [B,F,N,H*256], kv [B,F,N,H*512] (k | v per head) bf16; bias fp32 [B,H,N,N] (linear_b(z), :396);
 * mask [B,F,N] -> o bf16 [B,F,N,o_ld] (head h at columns h*256; o_ld = H*256, or the row length of the concatenated feature
 * matrix whose first columns o is), o_pt fp32 [B,F,N,H,12,3] (global frame), P bf16 [B,F,H,N,N] and, if P_f32 != NULL,
 * the same probabilities in fp32. */
int dfold_ipa_aug_prep(const float* q_pts, const float* k_pts, const float* v_pts, const float* hw, const float* ctr,
                       void* QP_bf16, void* KP_bf16, float* kn, void* VT_bf16, int32_t B, int32_t F, int32_t N, int32_t H,
                       int32_t NP, float alpha, void* stream);
int dfold_ipa_fused_fwd(const void* q_bf16, const void* kv_bf16, const void* QP_bf16, const void* KP_bf16, const void* VT_bf16,
                        const float* kn, const float* bias, const float* mask, const float* ctr, void* o_bf16, int64_t o_ld, float* o_pt,
                        void* P_bf16, float* P_f32, int32_t B, int32_t F, int32_t N, int32_t H, int32_t NP, float alpha,
                        float bias_scale, float inf, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Triangle pair operators (openfold/model/triangular_multiplicative_update.py:26-126 TriangleMultiplication
 * Outgoing/Incoming; openfold/model/triangular_attention.py:31-139 TriangleAttentionStarting/EndingNode with
 * Attention / _attention openfold/model/primitives.py:219-243,299-448; LayerNorm primitives.py:180-198).
 * Row / pointwise passes; the projections, the ik,jk->ij contraction and the q.k / a.v products use dfold_gemm_bf16.
 * ---------------------------------------------------------------------------------------------- */
/* y = LayerNorm_C(x) * gamma + beta (biased variance, eps inside sqrt); x fp32|bf16 [R][C]; stats[r] = {mean, rstd} */
int dfold_row_ln_fwd(const void* x, int32_t x_is_bf16, const float* gamma, const float* beta, void* y_bf16, float* stats,
                     int64_t R, int32_t C, float eps, void* stream);
/* dx (fp32|bf16), dgamma += , dbeta += (fp32 atomics; caller zeroes) from g = dL/dy (bf16) */
int dfold_row_ln_bwd(const void* x, int32_t x_is_bf16, const float* stats, const float* gamma, const void* g_bf16, void* dx,
                     int32_t dx_is_bf16, float* dgamma, float* dbeta, int64_t R, int32_t C, void* stream);
/* proj bf16 [R][5c] = [a_p|a_g|b_p|b_g|g] -> ab bf16 [R][2c]: a = a_p*sigmoid(a_g)*mask[r], b likewise (:97-104) */
int dfold_trimul_gate_fwd(const void* proj, const float* mask, void* ab, int64_t R, int32_t c, void* stream);
int dfold_trimul_gate_bwd(const void* proj, const float* mask, const void* dab, void* dproj, int64_t R, int32_t c, void* stream);
/* out = y * sigmoid(g): y fp32 [R][c], g bf16 with row stride ldg (:122-124; primitives.py:385-390) */
int dfold_gate_mul_fwd(const float* y, const void* g, float* out, int64_t R, int32_t c, int64_t ldg, void* stream);
int dfold_gate_mul_bwd(const float* y, const void* g, const float* dout, void* dy_bf16, void* dg_bf16, int64_t R, int32_t c,
                       int64_t ldg, void* stream);
/* S fp32 [I][H][N][N] -> P in place (+bf16 copy): softmax_k(S + inf*(mask[i,k]-1) + tri[h,q,k]) (triangular_attention.py:105-113) */
int dfold_triatt_softmax_fwd(float* S, const float* mask, const float* tri, void* P_bf16, int32_t I, int32_t H, int32_t N,
                             float inf, void* stream);
/* dP -> dS = P*(dP - sum_k P dP) in place (+bf16 copy) */
int dfold_triatt_softmax_bwd(const float* P, float* dP, void* dS_bf16, int64_t rows, int32_t N, void* stream);
/* out[e] = sum_{i<I} x[i*stride + e], e < n (fp32) */
int dfold_sum_leading(const float* x, float* out, int32_t I, int64_t n, int64_t stride, void* stream);

/* Streaming backward of the triangle attention core (round 4; csrc/triatt_bwd.hip): gated attention of
 * openfold/model/triangular_attention.py:78-139 / primitives.py:219-243,377-448 for 4 heads x 32 channels, in the operator's
 * own coordinates (the ending node passes transposed tensors), N % 8 == 0, N <= 512.  Logits, probabilities and their
 * gradients are recomputed per pair-tensor row on the matrix cores and never reach HBM.
 *   proj bf16 [B N N][512]: q | k | v | g (gate PRE-activation) of every cell (the recomputed projections) and proj_t
 *   bf16 [512][B N N], the same values channel-major (the projection GEMM with swapped operands: the K^T / V^T / Q^T tiles);
 *   tri fp32 [B][4][N][N] triangle bias; mask fp32 [B][N][N]; dout bf16 [B N N][128]; w_o_t bf16 [128][128] = W_o^T;
 * -> dproj bf16 [B N N][512]: dq | dk | dv | dg;  og bf16 [B N N][128] = o * sigmoid(g) (operand of dW_o);
 *    dtri_part fp32 [n_chunks][B][4][N][N]: the triangle-bias gradient summed over the rows of each chunk (add the chunks
 *    with dfold_sum_leading);  workspace: do_scratch bf16 [B N N][128], do_t_scratch bf16 [128][B N N], stats fp32
 *    [B N][4][3][N].
 * inf: mask bias = inf * (mask - 1); scale = 1 / sqrt(32). */
int dfold_triatt_bwd_core(const void* proj_bf16, const void* proj_t_bf16, const float* tri, const float* mask,
                          const void* dout_bf16, const void* w_o_t_bf16, void* dproj_bf16, void* og_bf16, void* do_scratch_bf16,
                          void* do_t_scratch_bf16, float* stats, float* dtri_part, int32_t B, int32_t N, int32_t n_chunks, float inf,
                          float scale, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused forward of the triangle operators for c_z = c_hidden = 128 / c_in = 128, 4 heads x 32 (csrc/pair_fused.hip).
 * Pair tensors are [B][N][N][128]; NP = N rounded up to a multiple of 64 is the pitch of the K-contiguous planes.
 * ---------------------------------------------------------------------------------------------- */
/* Triangle multiplication, stage 1 (triangular_multiplicative_update.py:92-104,122): LayerNorm_in, the five projections
 * (w_cat rows a_p|a_g|b_p|b_g|g, [640][128] bf16; bias_cat [640]), gates and mask ->
 *   planes bf16 [B][N][256][NP]: plane[i][ch][k] = a[i,k,ch] (outgoing) or a[k,i,ch] (incoming), ch < 128; b in 128..255
 *   (line-major: the 256 channel rows of one line i share a 256*NP*2-byte region, so a tile's 128-byte segments stay in
 *   one DRAM neighbourhood; pad columns k >= N zero-filled); gate bf16 [B][N][N][128] = sigmoid(linear_g(LN(z)));
 *   stats [B*N*N][2] or NULL */
int dfold_trimul_proj_fwd(const void* z, int32_t z_is_bf16, const float* mask, const float* ln_gamma, const float* ln_beta,
                          const void* w_cat_bf16, const float* bias_cat, void* planes_bf16, void* gate_bf16, float* stats,
                          int32_t B, int32_t N, int32_t NP, int32_t incoming, float eps, void* stream);
/* stage 3 (:119-124): x planes bf16 [B][N][128][NP] (x[i][c][j] = sum_k a_c[i,k] b_c[j,k] from dfold_gemm_bf16) -> LayerNorm_out -> linear_z
 * -> * gate -> out [B][N][N][128] fp32 | bf16 */
int dfold_trimul_out_fwd(const void* x_planes_bf16, const void* gate_bf16, const float* ln_gamma, const float* ln_beta,
                         const void* w_z_bf16, const float* b_z, void* out, int32_t out_is_bf16, int32_t B, int32_t N,
                         int32_t NP, float eps, void* stream);
/* Backward of stage 3 (the autograd of triangular_multiplicative_update.py:119-124), one pass over (x planes, gate, dout):
 * LayerNorm_out recomputed, y = xn W_z^T + b_z, d(gate pre-activation) = dout y g (1 - g) written as bf16 rows of pitch dgate_ld
 * (the last 128 columns of the [cells][640] pre-activation gradient matrix), dy = dout g and xn as bf16 [B N N][128] (operands of
 * dW_z), dxn = dy W_z (w_z_t = W_z^T bf16 [128][128]), LayerNorm backward -> dx as bf16 planes [B][N][128][NP] (what the two
 * contraction gradients consume); d_gamma / d_beta / d_bz / d_bg (the output gate's bias gradient) fp32 [128] are ADDED to
 * (atomics): zero them first. */
int dfold_trimul_out_bwd(const void* x_planes_bf16, const void* gate_bf16, const void* dout, int32_t dout_is_bf16,
                         const float* ln_gamma, const float* ln_beta, const void* w_z_bf16, const void* w_z_t_bf16,
                         const float* b_z, void* dx_planes_bf16, void* dgate_pre_bf16, int64_t dgate_ld, void* dy_bf16,
                         void* xn_bf16, float* d_gamma, float* d_beta, float* d_bz, float* d_bg, int32_t B, int32_t N,
                         int32_t NP, float eps, void* stream);
/* Backward of stage 1 (:97-111): LayerNorm_in and the four gated projections a_p | a_g | b_p | b_g recomputed per 64-cell tile
 * (w_cat rows 0 .. 511), gate backward against dplanes = the gradients of the a | b planes (bf16 [B][N][256][NP], the layout of
 * dfold_trimul_proj_fwd's planes) -> d5 columns 0 .. 511 (bf16 rows of pitch d5_ld, true cell order), zn bf16 [B N N][128],
 * LayerNorm statistics fp32 [B N N][2] (mean, 1/std: inputs of dfold_row_ln_bwd); d_bias fp32 [512] is ADDED to. */
int dfold_trimul_proj_bwd(const void* z, int32_t z_is_bf16, const float* mask, const float* ln_gamma, const float* ln_beta,
                          const void* w_cat_bf16, const float* bias_cat, const void* dplanes_bf16, void* d5_bf16,
                          int64_t d5_ld, void* zn_bf16, float* stats, float* d_bias, int32_t B, int32_t N, int32_t NP,
                          int32_t incoming, float eps, void* stream);
/* Triangle attention, stage 1 (triangular_attention.py:92-113, primitives.py:363-383): LayerNorm, q|k|v|g projections
 * (w_cat [512][128] bf16, bias_cat [512] = 0|0|0|b_g), triangle bias (w_tri fp32 [4][128]).  ending != 0: the operator
 * acts on x' = x^T (cell (i,j) of every output = cell (j,i) of x).  q, k, gate(=sigmoid) bf16 [B][N][N][128];
 * vT bf16 [B][N][128][NP] (keys contiguous); tri fp32 [B][4][N][NP] = log2(e) * bias (the core's softmax runs on exp2);
 * q_bf16 = k_bf16 = vT_bf16 = gate_bf16 = NULL: only tri is produced (pass 0 of dfold_triatt_fused_fwd), and in that kernel's
 * layout: fp32 [B][4][NP/16][NP/16][64][4], 16 x 16 (query, key) blocks in MFMA-accumulator order (element (q, key) at lane
 * ((key & 15) >> 2) * 16 + (q & 15), register key & 3 of block (q >> 4, key >> 4)); tri must then hold B*4*NP*NP floats. */
int dfold_triatt_proj_fwd(const void* x, int32_t x_is_bf16, const float* ln_gamma, const float* ln_beta, const void* w_cat_bf16,
                          const float* bias_cat, const float* w_tri, void* q_bf16, void* k_bf16, void* vT_bf16, void* gate_bf16,
                          float* tri, int32_t B, int32_t N, int32_t NP, int32_t ending, float eps, void* stream);
/* stage 2 (primitives.py:219-243,385-448): per row i gated multi-head attention over the keys of that row with the
 * triangle bias (tri as written by dfold_triatt_proj_fwd, i.e. pre-multiplied by log2 e) and inf*(mask-1), flash-style
 * (no logits in HBM), then linear_o; out [B][N][N][128] fp32 | bf16 in the
 * coordinates of x (transposed back for ending != 0); mask fp32 [B][N][N] in the coordinates of x */
int dfold_triatt_core_fwd(const void* q_bf16, const void* k_bf16, const void* vT_bf16, const void* gate_bf16, const float* tri,
                          const float* mask, const void* w_o_bf16, const float* b_o, void* out, int32_t out_is_bf16, int32_t B,
                          int32_t N, int32_t NP, int32_t ending, float inf, float scale, void* stream);

/* Triangle attention with the projections kept on chip (N <= 256): per (batch item, row) LayerNorm + q|k|v|g projections +
 * gated attention over the row's keys (exact softmax) + linear_o in one workgroup; x is read once here and once by pass 0
 * (dfold_triatt_bias_blocked: LayerNorm + the 4-wide bias projection only, a streaming pass that writes tri in the blocked
 * layout described at dfold_triatt_proj_fwd; that entry point with null q/k/v/gate produces the same).  w_cat [512][128] bf16 (q|k|v|g), bias_cat [512], tri as written by pass 0,
 * mask [B][N][N] in the coordinates of x, out [B][N][N][128] fp32 | bf16.  dbg (tests, may be NULL): fp32 [4][N][32] =
 * q | k | v | sigmoid(g) of head 0 of row 0 of item 0. */
int dfold_triatt_bias_blocked(const void* x, int32_t x_is_bf16, const float* ln_gamma, const float* ln_beta, const float* w_tri,
                              float* tri, int32_t B, int32_t N, int32_t NP, int32_t ending, float eps, void* stream);
int dfold_triatt_fused_fwd(const void* x, int32_t x_is_bf16, const float* mask, const float* ln_gamma, const float* ln_beta,
                           const void* w_cat_bf16, const float* bias_cat, const float* tri, const void* w_o_bf16, const float* b_o,
                           void* out, int32_t out_is_bf16, float* dbg, int32_t B, int32_t N, int32_t NP, int32_t ending,
                           float inf, float scale, float eps, void* stream);
/* The same operator for N <= 512 with every projection kept in REGISTERS (round 6, csrc/triatt_reg.hip;
 * triangular_attention.py:78-139, primitives.py:219-243,377-448): one wave owns 64 cells of the row -- their LayerNorm output
 * stays in the MFMA operand layout in its registers, Q^T / sigmoid(G^T) / gated O leave the matrix pipe in the operand layout
 * of the next product, only K and V^T of a head go through LDS; the head's weights (and W_o) wait there as ready-made operand
 * fragments, requested by LDS-DMA a whole attention phase ahead (66 KB at N <= 256 =
 * two rows per CU).  Same arguments and pass 0 (dfold_triatt_bias_blocked) as dfold_triatt_fused_fwd, plus dbg_phase_clock
 * (measurement, scripts/triatt_phase_times.py; needs dbg with room for 4 N 32 + 512 floats): the waves of one workgroup in mid
 * grid write s_memtime at every phase boundary to dbg[4 N 32 + 64 wave + k]. */
int dfold_triatt_reg_fwd(const void* x, int32_t x_is_bf16, const float* mask, const float* ln_gamma, const float* ln_beta,
                         const void* w_cat_bf16, const float* bias_cat, const float* tri, const void* w_o_bf16, const float* b_o,
                         void* out, int32_t out_is_bf16, float* dbg, int32_t dbg_phase_clock, int32_t B, int32_t N, int32_t NP,
                         int32_t ending, float inf, float scale, float eps, void* stream);
/* The same operator for ANY N (round 4, csrc/triatt_rows.hip; triangular_attention.py:78-139, primitives.py:219-243,377-448):
 * the query-block form of the row kernel.  Pass 0, dfold_triatt_ln_bias: one streaming pass over x that writes
 *   xn bf16 [B][N][N][128] = LayerNorm(x') in the operator's coordinates (x' = x, or x^T for ending != 0) and
 *   tri fp32 [B][4][NP/16][NP/16][64][4] = log2(e) * w_tri . LN(x'), the blocked layout above.
 * dfold_triatt_rows_fwd: one workgroup per (item, row, block of 256 queries); per head and per chunk of 256 keys the
 * q | k | v | g projections of 64-cell xn tiles on the matrix cores into LDS tiles, gated attention with an online softmax
 * over the key chunks (one chunk: the exact softmax), linear_o accumulated over the heads; q, k, v, g never reach HBM.
 * mask [B][N][N] and out [B][N][N][128] fp32 | bf16 in the coordinates of x.  dbg (tests, may be NULL): fp32 [4][N][32] =
 * q | k | v | sigmoid(g) of head 0 of row 0 of item 0 (q / g rows: the first query block only). */
int dfold_triatt_ln_bias(const void* x, int32_t x_is_bf16, const float* ln_gamma, const float* ln_beta, const float* w_tri,
                         float* tri, void* xn_bf16, int32_t B, int32_t N, int32_t NP, int32_t ending, float eps,
                         void* stream);
int dfold_triatt_rows_fwd(const void* xn_bf16, const float* mask, const void* w_cat_bf16, const float* bias_cat, const float* tri,
                          const void* w_o_bf16, const float* b_o, void* out, int32_t out_is_bf16, float* dbg,
                          int32_t B, int32_t N, int32_t NP, int32_t ending, float inf, float scale, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Training loss (Experiment.loss_fn, train_DFOLD_dynamics.py:1182-1400, live terms; torsion term openfold/utils/loss.py:52-76
 * as called at :1219-1224) on the LAST frame of every window, values and gradients in one launch (round 6, csrc/loss.hip).
 * All inputs are the last-frame slices, contiguous: ang / ang_gt / ang_alt fp32 [B][N][7][2], ang_mask fp32 [B][N][7],
 * trans_pred / trans_gt fp32 [B][N][3], rot_pred / rot_gt float64 [B][N][3] (the IGSO(3) score head works in float64),
 * diffuse_mask = 1 - fixed_mask and loss_mask = res_mask * diffuse_mask fp32 [B][N], rot_score_scaling float64 [B], t fp32 [B],
 * live_frames fp32 [B] = frames of the window with any residue (the normalisation of :1388-1396; F = frames of a window).
 * terms float64 [B][4] = per-window (final, rot, gated trans, torsion), already repeated over the F frames and divided by
 * live_frames as the reference does; loss = mean over windows of terms[:,0].  d_ang / d_trans (fp32) / d_rot (float64):
 * d loss / d (ang, trans_pred, rot_pred).  Gate trans < 100 and t > rot_t_threshold as :1338-1340, :1304. */
int dfold_loss_last_frame(const float* ang, const float* ang_gt, const float* ang_alt, const float* ang_mask,
                          const float* trans_pred, const float* trans_gt, const double* rot_pred, const double* rot_gt,
                          const float* diffuse_mask, const float* loss_mask, const double* rot_score_scaling, const float* t,
                          const float* live_frames, double* terms, float* d_ang, float* d_trans, double* d_rot, int32_t B,
                          int32_t N, int32_t F, float trans_weight, float rot_weight, float torsion_weight,
                          float rot_t_threshold, void* stream);

/* ------------------------------------------------------------------------------------------------
 * First layer of the feature embedders (force/vel/index/rigid/angle_embeder[0:2], src/model/ipa_pytorch_dynamic.py:
 * 757-796): h = SiLU(x W^T + b), x fp32 [P,k] (k <= 16), W fp32 [256,k], h bf16 [P,256].  Backward accumulates
 * dW / db with fp32 atomics (caller zeroes them); dx (fp32 [P,k]) may be NULL.
 * ---------------------------------------------------------------------------------------------- */
int dfold_embed_in_fwd(const float* x, const float* W, const float* b, void* out_bf16, int64_t P, int32_t k, int32_t D,
                       void* stream);
int dfold_embed_in_bwd(const float* x, const float* W, const float* b, const void* g_bf16, float* dW, float* db, float* dx,
                       int64_t P, int32_t k, int32_t D, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Backbone frame + 7 torsions -> atom14 / atom37 (feats.torsion_angles_to_frames openfold/utils/feats.py:165-228,
 * all_atom.frames_to_atom14_pos src/data/all_atom.py:114-154, atom14_to_atom37 src/model/Dfold_network_dynamic.py:574-594).
 * t7 fp32 [P][7], angles fp32 [P][7][2] (sin,cos), aatype int64 [P] in 0..20; residue tables as in residue_tables.npz.
 * ---------------------------------------------------------------------------------------------- */
int dfold_frames_to_atoms(const float* t7, const float* angles, const int64_t* aatype, const float* default_frames,
                          const int64_t* atom14_group, const float* atom14_mask, const float* atom14_pos,
                          const int64_t* atom37_to_atom14, const float* atom37_mask, float* atom14, float* atom37, int64_t P,
                          void* stream);
/* its backward (round 4): the reference builds the atoms inside autograd (src/model/Dfold_network_dynamic.py:532-538; the
 * bb-atom / dist-mat loss terms read them, train_DFOLD_dynamics.py:1317-1364).  g_atom14 [P][14][3] and / or g_atom37
 * [P][37][3] (either may be NULL) -> d_t7 [P][7] (quaternion of the un-normalised quadratic form, translation),
 * d_angles [P][7][2]. */
int dfold_frames_to_atoms_bwd(const float* t7, const float* angles, const int64_t* aatype, const float* default_frames,
                              const int64_t* atom14_group, const float* atom14_mask, const float* atom14_pos,
                              const int64_t* atom37_to_atom14, const float* atom37_mask, const float* g_atom14,
                              const float* g_atom37, float* d_t7, float* d_angles, int64_t P, void* stream);

/* ------------------------------------------------------------------------------------------------
 * IGSO(3) score series (SO3Diffuser.torch_score src/data/so3_diffuser.py:274-305, igso3_expansion :9-49,
 * score :71-117): sc[p] = dsig(omega_p)/(f(omega_p)+1e-4) with the reference's fp32-trig / fp64-envelope
 * mixed precision, plus dsc = d sc / d omega for the backward.  env fp64 [windows][L] = (2l+1)exp(-l(l+1)s^2/2);
 * element p belongs to window p / per_window (P % per_window == 0).
 * ---------------------------------------------------------------------------------------------- */
int dfold_igso3_series(const float* omega, const double* env, double* sc, double* dsc, int64_t P, int64_t per_window,
                       int32_t L, void* stream);

/* The rest of the rotation-score head around the series (round 6; SE3Diffuser.calc_rot_score src/data/se3_diffuser.py:119-125 =
 * SO3Diffuser.torch_score so3_diffuser.py:274-305 of quat_to_rotvec(q_0^-1 q_t), src/data/utils.py:589-606, quaternion inverse /
 * product openfold/utils/rigid_utils.py:230-286).  quats fp32 [P][4] (w, x, y, z).
 *   pre:  vec fp32 [P][3] = rotvec(q_0^-1 q_t) in the reference's fp32 arithmetic, omega fp32 [P] = |vec| + 1e-6
 *   (dfold_igso3_series: omega -> sc, dsc)
 *   post: score float64 [P][3] = sc vec / (omega + 1e-6)
 *   bwd:  g_score float64 [P][3] -> d_quats_0 fp32 [P][4] (the chain rule through all of it, float64 inside). */
int dfold_rot_head_pre(const float* quats_t, const float* quats_0, float* vec, float* omega, int64_t P, void* stream);
int dfold_rot_head_post(const float* vec, const float* omega, const double* sc, double* score, int64_t P, void* stream);
int dfold_rot_head_bwd(const double* g_score, const float* quats_t, const float* quats_0, const float* vec, const float* omega,
                       const double* sc, const double* dsc, float* d_quats_0, int64_t P, void* stream);

/* ------------------------------------------------------------------------------------------------
 * One reverse-SDE (denoise) step on tensor_7 frames (SE3Diffuser.reverse src/data/se3_diffuser.py:160-215,
 * SO3Diffuser.reverse so3_diffuser.py:329-365, R3Diffuser.reverse r3_diffuser.py:106-157).  t7/out fp32 [rows][N][7]
 * (rows = windows*frames; centring is per row), rot_score / z_rot / z_trans fp64 [rows][N][3], trans_score fp32,
 * mask fp32 [rows][N] or NULL (0 = keep the frame), g_rot = SO3 diffusion coefficient at t, b_t = R3 beta(t).
 * ---------------------------------------------------------------------------------------------- */
int dfold_se3_reverse(const float* t7, const double* rot_score, const float* trans_score, const double* z_rot,
                      const double* z_trans, const float* mask, float* out, int64_t rows, int32_t N, double g_rot, double b_t,
                      double dt, double noise_scale, double coordinate_scaling, int32_t center, void* stream);

/* ----------------------------------------------------------------------------------------------
 * Device draws for the diffusion steps (replace the host numpy draws inside the sampling loop, src/data/so3_diffuser.py:
 * 347-349 and r3_diffuser.py:140-147, and of the forward noising, so3_diffuser.py:233-248 / r3_diffuser.py:96-99):
 * Philox4x32-10, key = seed, counter = (element / 4, subseq); element e = word e % 4 of its counter block.
 * uniform: (x + 0.5) * 2^-32 in (0,1); normal: fp64 Box-Muller on word pairs (0,1) and (2,3).  out fp64 [n].
 * ---------------------------------------------------------------------------------------------- */
int dfold_philox_normal_f64(double* out, int64_t n, uint64_t seed, uint64_t subseq, void* stream);
int dfold_philox_uniform_f64(double* out, int64_t n, uint64_t seed, uint64_t subseq, void* stream);

/* ----------------------------------------------------------------------------------------------
 * Forward noising q(x_t | x_0) of tensor_7 frames (replaces SE3Diffuser.forward_marginal, src/data/se3_diffuser.py:43-110
 * -> SO3Diffuser.forward_marginal so3_diffuser.py:311-327 (sample :233-248, sample_igso3 :215-231, compose_rotvec
 * src/data/utils.py:184-195) and R3Diffuser.forward_marginal r3_diffuser.py:81-101).  P = windows*frames*residues frames,
 * per_window = frames*residues; window w uses cdf row cdf_idx[w] of cdf [num_sigma][num_omega] (fp64, device) and R3
 * marginal beta b_t[w].  u fp64 [P] uniform(0,1), z_dir / z_trans fp64 [P][3] standard normal: draws are inputs.
 * mask fp32 [P] or NULL (0 = frame stays at x_0 with zero scores).  out fp32 [P][7]; rotvec_out fp64 [P][3] = the sampled
 * rotation vector (its IGSO(3) score comes from dfold_igso3_series); trans_score fp32 [P][3] (unscaled, as the reference).
 * ---------------------------------------------------------------------------------------------- */
int dfold_se3_forward_marginal(const float* t7, const double* u, const double* z_dir, const double* z_trans,
                               const float* mask, const double* cdf, const double* omega_grid, const int32_t* cdf_idx,
                               const double* b_t, float* out, double* rotvec_out, float* trans_score, int64_t P,
                               int64_t per_window, int32_t num_omega, double coordinate_scaling, void* stream);

/* ----------------------------------------------------------------------------------------------
 * Backbone frame update (replaces Rigid.compose_q_update_vec, openfold/utils/rigid_utils.py:1039-1063 ->
 * Rotation.compose_q_update_vec :587-616, quat_multiply_by_vec :266-275, normalisation :331-332; call site
 * src/model/ipa_pytorch_dynamic.py:871):  q' = normalize(q + m (q (x) (0,u))),  t' = t + m R(q) v.
 * t7 fp32 [P][7], upd6 fp32 [P][6] = (u, v), mask fp32 [P] or NULL -> out [P][7];  backward: g = dL/dout ->
 * dt7 [P][7], dupd6 [P][6].
 * ---------------------------------------------------------------------------------------------- */
int dfold_compose_fwd(const float* t7, const float* upd6, const float* mask, float* out, int64_t P, void* stream);
int dfold_compose_bwd(const float* t7, const float* upd6, const float* mask, const float* g, float* dt7, float* dupd6,
                      int64_t P, void* stream);

/* ----------------------------------------------------------------------------------------------
 * Dataset-side geometry (replaces openfold/data/data_transforms.py:755-893 atom37_to_frames and :923-1088
 * atom37_to_torsion_angles as called by src/data/Dfold_data_loader_dynamic.py:237-240).  P = rows*N residues, chains are
 * rows of N residues (the previous residue of a row's first residue does not exist).  Inputs float64 like the loader's;
 * tables = the reference's residue constants (dynamicpdb_amd/data/residue_tables.npz).  Pass gt_frames == NULL to skip
 * the frame block, torsion_sin_cos == NULL to skip the torsion block.
 *   gt_frames / alt_gt_frames fp32 [P][8][4][4];  gt_exists, group_exists, group_is_ambiguous fp64 [P][8];
 *   torsion_sin_cos / alt_torsion_sin_cos fp64 [P][7][2];  torsion_mask fp64 [P][7].
 * ---------------------------------------------------------------------------------------------- */
int dfold_atom37_geometry(const int64_t* aatype, const double* all_atom_positions, const double* all_atom_mask,
                          const int64_t* group_base_atom37, const float* group_mask, const float* group_ambiguous,
                          const int64_t* chi_atom37, const float* chi_mask, const float* chi_pi_periodic,
                          float* gt_frames, float* alt_gt_frames, double* gt_exists, double* group_exists,
                          double* group_is_ambiguous, double* torsion_sin_cos, double* alt_torsion_sin_cos,
                          double* torsion_mask, int64_t P, int32_t N, double eps, void* stream);

/* ----------------------------------------------------------------------------------------------
 * Fused Adam(amsgrad=True) step over a list of fp32 tensors in one launch (replaces the optimizer step of
 * train_DFOLD_dynamics.py:412 / :666, torch.optim.Adam foreach path).  table: n_tensors device records; chunk_start:
 * device int32 [n_tensors + 1], exclusive prefix sum of ceil(n / dfold_adam_chunk()) per tensor; n_chunks = its last
 * entry.  step >= 1 is the 1-based step count used for the bias corrections; no weight decay, not maximising.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  void* p;                /* fp32 parameter, updated in place */
  const void* g;          /* fp32 gradient */
  void* exp_avg;          /* fp32 states, updated in place */
  void* exp_avg_sq;
  void* max_exp_avg_sq;
  int64_t n;              /* elements */
} dfold_adam_tensor;
int dfold_adam_amsgrad(const dfold_adam_tensor* table, const int32_t* chunk_start, int32_t n_tensors, int32_t n_chunks,
                       double lr, double beta1, double beta2, double eps, int64_t step, void* stream);
int dfold_adam_chunk(void);

#ifdef __cplusplus
}
#endif
#endif
