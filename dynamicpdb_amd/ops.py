"""Thin Python launchers over the C ABI (include/dfold_hip.h).  Tensors are torch CUDA tensors used
purely as device buffers; all arithmetic happens in libdfold_hip.so.  bf16 tensors are torch.bfloat16."""
import ctypes
import os
from ctypes import byref, c_int32, c_int64, c_void_p

import torch

from . import _lib
from ._lib import (GEMM_ACCUM, GEMM_ATOMIC, GEMM_BIAS, GEMM_OUT_BF16, GEMM_RELU, GEMM_RELUMASK, GEMM_RESID, GemmDesc,
                   RowMap, check, stream)

GEMM_C2RELU, GEMM_MASK2, GEMM_NZ_KEEP = 128, 256, 512          # include/dfold_hip.h

BF16 = torch.bfloat16
_zero_pages = {}
_seg_tables = {}


def zeros_page(device):
    z = _zero_pages.get(device)
    if z is None:
        z = torch.zeros(256, dtype=torch.uint8, device=device)
        _zero_pages[device] = z
    return z


def _p(t, off=0):
    """device pointer of tensor t advanced by `off` ELEMENTS (None -> NULL)."""
    if t is None:
        return c_void_p(0)
    return c_void_p(t.data_ptr() + off * t.element_size())


def rows_plain(ld, base=0):
    return RowMap(base, ld, 0, 0, 0, 0, 0)


def rows_grid(C, N, F, Fp, Wp, base=0):
    return RowMap(base, C, 1, N, F, Fp, Wp)


def rows_lin(C, N, F, Fp, Wp, base=0):
    """RowMap mode 2: the cells of a window as one line (dfold_common.h): any N_res on the 512 x 160 conv kernel"""
    return RowMap(base, C, 2, N, F, Fp, Wp)


def grid_slack(Wp, C):
    """elements kept readable (and zero) behind a padded conv grid: the mode-2 conv launches and the linear weight-gradient walk
    read up to 272 cells + 4 frame rows past the last window's end for rows they never store / products with zero gradients"""
    return (4 * Wp + 288) * C


def has_grid_slack(g, t):
    """does the padded grid tensor t own grid_slack() elements of storage behind its last window?"""
    C = t.shape[-1]
    behind = t.untyped_storage().nbytes() // t.element_size() - (t.storage_offset() + g.Wn * g.Fp * g.Wp * C)
    return behind >= grid_slack(g.Wp, C)


def seg_table(values, device):
    key = (tuple(values), device)
    t = _seg_tables.get(key)
    if t is None:
        t = torch.tensor(list(values), dtype=torch.int64, device=device)
        _seg_tables[key] = t
    return t


def gemm(A, B, C, M, N, seglen, *, a_rows, c_rows, ldb, nseg=1, bias=None, R=None, C2=None, R2=None,
         a_seg=None, b_seg=None, seg_div=1, seg_div_mid=0, nbatch=1, nb1=1, sa=(0, 0), sb=(0, 0), sc=(0, 0), flags=0, alpha=1.0,
         a_off=0, b_off=0, c_off=0, splitk=1, splitk_ws=None, splitk_cnt=None, conv_frames=0, nz=None):
    """C = epi(alpha * A @ B^T) on the bf16 MFMA engine; see dfold_gemm_desc in include/dfold_hip.h."""
    assert A.dtype == BF16 and B.dtype == BF16
    if C.dtype == BF16:
        flags |= GEMM_OUT_BF16
    else:
        assert C.dtype == torch.float32
    if bias is not None:
        flags |= GEMM_BIAS
        assert bias.dtype == torch.float32
    d = GemmDesc()
    d.A, d.B, d.C, d.C2 = _p(A, a_off), _p(B, b_off), _p(C, c_off), _p(C2, c_off)
    d.bias, d.R, d.R2 = _p(bias), _p(R, c_off), _p(R2, c_off)
    d.zeros = _p(zeros_page(A.device))
    # K-segment offsets (seg0, s1, s2): seg0 + (g // seg_div)*s1 + (g % seg_div)*s2; default = contiguous K
    # (seg0, s_mid, s_lo) or (seg0, s_hi, s_mid, s_lo)
    a_seg = a_seg or (0, seglen * seg_div, seglen)
    b_seg = b_seg or (0, seglen * seg_div, seglen)
    a_seg = a_seg if len(a_seg) == 4 else (a_seg[0], 0, a_seg[1], a_seg[2])
    b_seg = b_seg if len(b_seg) == 4 else (b_seg[0], 0, b_seg[1], b_seg[2])
    d.a_seg0, d.a_seg_s0, d.a_seg_s1, d.a_seg_s2 = a_seg
    d.b_seg0, d.b_seg_s0, d.b_seg_s1, d.b_seg_s2 = b_seg
    d.seg_div, d.seg_div_mid = seg_div, seg_div_mid
    d.a_rows, d.c_rows = a_rows, c_rows
    d.ldb = ldb
    d.sa0, d.sa1, d.sb0, d.sb1, d.sc0, d.sc1 = sa[0], sa[1], sb[0], sb[1], sc[0], sc[1]
    d.M, d.N, d.nseg, d.seglen, d.nbatch, d.nb1, d.flags, d.alpha = M, N, nseg, seglen, nbatch, nb1, flags, alpha
    d.splitk, d.conv_frames = splitk, conv_frames
    split = splitk > 1 or splitk <= -1
    d.splitk_ws, d.splitk_cnt = _p(splitk_ws if split else None), _p(splitk_cnt if split else None)
    if nz is not None:           # (frame-flag prefix sums, radius, padded frame row of logical frame 0): dfold_gemm_desc.nz_ps
        d.nz_ps, d.nz_radius, d.nz_f0 = _p(nz[0]), nz[1], nz[2]
    check(_lib.lib().dfold_gemm_bf16(byref(d), stream()), "dfold_gemm_bf16")
    return C


def gemm_reduce_rows(AT, BT, M, N, K, out=None):
    """out[M,N] fp32 = AT[M,K] @ BT[N,K]^T for a LONG reduction axis K (weight gradients: K = windows*frames*residues)
    and a small output: the K axis is split over workgroups (split-K) whose partial tiles are combined with fp32
    atomics, so that a [256 x 256] gradient still fills the 256 CUs."""
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    S = max(1, min(64, 512 // tiles))
    while S > 1 and (K % (S * 64)):
        S -= 1
    if out is None:
        out = torch.zeros((M, N), dtype=torch.float32, device=AT.device)
    if S == 1:
        return gemm(AT, BT, out, M, N, K, a_rows=rows_plain(K), c_rows=rows_plain(N), ldb=K, flags=GEMM_ACCUM)
    ks = K // S
    return gemm(AT, BT, out, M, N, ks, a_rows=rows_plain(K), c_rows=rows_plain(N), ldb=K, nbatch=S, nb1=1,
                sa=(ks, 0), sb=(ks, 0), sc=(0, 0), flags=GEMM_ATOMIC)


def gemm_tn(A, B, C, M, N, K, lda, ldb, ldc, *, nbatch=1, nb1=1, sa=(0, 0), sb=(0, 0), sc=(0, 0), splitk=1, flags=0, alpha=1.0,
            a_off=0, b_off=0, c_off=0):
    """C (+)= alpha A^T B with both operands reduction-major (dfold_gemm_tn_bf16, include/dfold_hip.h)."""
    assert A.dtype == BF16 and B.dtype == BF16
    if C.dtype == BF16:
        flags |= GEMM_OUT_BF16
    check(_lib.lib().dfold_gemm_tn_bf16(_p(A, a_off), _p(B, b_off), _p(C, c_off), c_int32(M), c_int32(N), c_int64(K), c_int64(lda),
                                        c_int64(ldb), c_int64(ldc), c_int32(nbatch), c_int32(nb1), c_int64(sa[0]), c_int64(sa[1]),
                                        c_int64(sb[0]), c_int64(sb[1]), c_int64(sc[0]), c_int64(sc[1]), c_int32(splitk),
                                        c_int32(flags), ctypes.c_float(alpha), stream()), "dfold_gemm_tn_bf16")
    return C


def gemm_tn_ok(M, N, K, ragged=False):
    """shapes the reduction-major product covers (256 x 256 output tiles, 64-row K steps).  ragged: output extents that are
    only multiples of 8 (tiles hanging over the edge compute on re-read columns and store nothing there): the weight
    gradients of the 128- / 192- / 480- / 640-wide layers, whose alternative is two transposed operand copies."""
    if os.environ.get("DFOLD_GEMM_TN", "1") == "0" or K % 64:
        return False
    if ragged and os.environ.get("DFOLD_GEMM_TN_RAGGED", "1") != "0":
        return M % 8 == 0 and N % 8 == 0
    return M % 256 == 0 and N % 256 == 0


def weight_grad_tn(g2d, x2d, M, N, K, out=None, rz=None):
    """dW fp32 [N, K] (+)= g2d^T x2d for g2d bf16 [M, N], x2d bf16 [M, K]: the long row axis M is cut into split-K parts so
    that the few 256 x 256 output tiles still fill the chip.  rz = (ps, block): row-block flags of g2d (row_block_flags): parts
    whose rows are all zero exit at once -- the split is then as fine as the flags (parts of `block` rows, at most 256 of them)."""
    tiles = ((N + 255) // 256) * ((K + 255) // 256)
    S = max(1, min(64, 512 // tiles))
    while S > 1 and (M % (S * 64)):
        S -= 1
    if rz is not None and (rz[1] % 64 or (M // S) % rz[1] or (M // S) // rz[1] > 64):
        rz = None                        # (a part must be whole flag blocks, at most 64 of them: else every row is walked)
    if out is None:
        out = torch.zeros((N, K), dtype=torch.float32, device=g2d.device)
    if rz is None:
        return gemm_tn(g2d, x2d, out, N, K, M, N, K, K, splitk=S, flags=GEMM_ATOMIC)
    check(_lib.lib().dfold_gemm_tn_bf16_rowflags(_p(g2d), _p(x2d), _p(out), c_int32(N), c_int32(K), c_int64(M), c_int64(N), c_int64(K),
                                                 c_int64(K), c_int32(1), c_int32(1), c_int64(0), c_int64(0), c_int64(0), c_int64(0),
                                                 c_int64(0), c_int64(0), c_int32(S), c_int32(GEMM_ATOMIC), ctypes.c_float(1.0),
                                                 _p(rz[0]), c_int32(rz[1]), stream()), "dfold_gemm_tn_bf16_rowflags")
    return out


def row_block_flags(g2d, block=256):
    """prefix sums of the non-zero flags of blocks of `block` rows of the fp32 matrix g2d [R, C] (dfold_row_block_flags)"""
    R, C = g2d.shape
    nb = (R + block - 1) // block
    ps = torch.empty(nb + 1, dtype=torch.int32, device=g2d.device)
    scratch = torch.zeros(nb, dtype=torch.int32, device=g2d.device)      # (per call: callers on different streams share nothing)
    check(_lib.lib().dfold_row_block_flags(_p(g2d), _p(ps), _p(scratch), c_int64(R), c_int32(C), c_int64(g2d.stride(0)), c_int32(block),
                                           stream()), "dfold_row_block_flags")
    return ps



def cast_bf16(x):
    """fp32 -> bf16 copy (HIP kernel)."""
    x = x.contiguous()
    out = torch.empty(x.shape, dtype=BF16, device=x.device)
    check(_lib.lib().dfold_cast_f32_bf16(_p(x), _p(out), c_int64(x.numel()), stream()), "dfold_cast_f32_bf16")
    return out


def cast_f32(x):
    x = x.contiguous()
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    check(_lib.lib().dfold_cast_bf16_f32(_p(x), _p(out), c_int64(x.numel()), stream()), "dfold_cast_bf16_f32")
    return out


def transpose_bf16(src, R, C, ld_src=None, out=None, nbatch=1, nb1=1, bs_src=(0, 0), bs_dst=None, ld_dst=None,
                   src_off=0):
    """dst[z][c][r] = src[z][r][c]; batch z -> (z // nb1, z % nb1) with element strides bs_src / bs_dst."""
    ld_src = C if ld_src is None else ld_src
    ld_dst = R if ld_dst is None else ld_dst
    if isinstance(bs_src, int):
        bs_src = (bs_src * nb1, bs_src)
    if out is None:
        out = torch.empty((nbatch, C, R) if nbatch > 1 else (C, R), dtype=BF16, device=src.device)
        bs_dst = (C * R * nb1, C * R)
    check(_lib.lib().dfold_transpose_bf16(_p(src, src_off), _p(out), c_int32(R), c_int32(C), c_int64(ld_src),
                                          c_int64(ld_dst), c_int32(nbatch), c_int32(nb1), c_int64(bs_src[0]),
                                          c_int64(bs_src[1]), c_int64(bs_dst[0]), c_int64(bs_dst[1]), stream()),
          "dfold_transpose_bf16")
    return out


def colsum_bf16(x, out, R, C, ld):
    check(_lib.lib().dfold_colsum_bf16(_p(x), _p(out), c_int64(R), c_int32(C), c_int64(ld), stream()),
          "dfold_colsum_bf16")


def relu_mask_bf16(g, v, out):
    check(_lib.lib().dfold_relu_mask_bf16(_p(g), _p(v), _p(out), c_int64(g.numel()), stream()), "dfold_relu_mask_bf16")
    return out


# ------------------------------------------------------------------------------------------------
# dense layers on [rows, K] bf16 matrices
# ------------------------------------------------------------------------------------------------

def linear_fwd(x, w, bias=None, out_dtype=BF16, relu=False, out=None, x_rows=None, M=None, out_rows=None):
    """y = x @ w^T + bias.  x bf16 [M,K] (or any tensor addressed through x_rows), w bf16 [N,K]."""
    N, K = w.shape
    if M is None:
        M = x.numel() // K
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=w.device)
    return gemm(x, w, out, M, N, K, a_rows=x_rows or rows_plain(K), c_rows=out_rows or rows_plain(N), ldb=K,
                bias=bias, flags=GEMM_RELU if relu else 0)


def linear_bwd(x, w_t, gy, need_dx=True, dw_out=None, accumulate=False, M=None):
    """x bf16 [M,K], w_t bf16 [K,N] (= w transposed), gy bf16 [M,N]  ->  dx bf16 [M,K], dW fp32 [N,K]."""
    K, N = w_t.shape
    if M is None:
        M = gy.numel() // N
    dx = None
    if need_dx:
        dx = torch.empty((M, K), dtype=BF16, device=gy.device)
        gemm(gy, w_t, dx, M, K, N, a_rows=rows_plain(N), c_rows=rows_plain(K), ldb=N)
    gyT = transpose_bf16(gy, M, N)      # [N][M]
    xT = transpose_bf16(x, M, K)        # [K][M]
    if dw_out is None:
        dw_out = torch.empty((N, K), dtype=torch.float32, device=gy.device)
        accumulate = False
    gemm(gyT, xT, dw_out, N, K, M, a_rows=rows_plain(M), c_rows=rows_plain(K), ldb=M,
         flags=GEMM_ACCUM if accumulate else 0)
    return dx, dw_out


# ------------------------------------------------------------------------------------------------
# 5x5 conv tower on the zero-padded frame x residue grid
# ------------------------------------------------------------------------------------------------

class Grid:
    """Geometry of the zero-padded channels-last activation grid [windows, F+4, N+4, C]."""

    def __init__(self, Wn, F, N, device):
        self.Wn, self.F, self.N, self.Fp, self.Wp, self.device = Wn, F, N, F + 4, N + 4, device
        self.M = Wn * F * N
        # transposed [channel][window][frame][residue] copies of the weight-gradient product: frame rows are padded to a
        # multiple of 8 residues (16-byte aligned K runs for any N_res; the pad columns stay zero)
        self.NP = (N + 7) // 8 * 8
        self.plane = Wn * self.Fp * self.NP

    def alloc(self, C):
        n = self.Wn * self.Fp * self.Wp * C
        return torch.zeros(n + grid_slack(self.Wp, C), dtype=BF16, device=self.device)[:n].view(self.Wn, self.Fp, self.Wp, C)

    def vw(self, nf):
        """mode-2 rows per window: the nf * Wp cells of the frame range rounded up to whole 256-row runs"""
        return (nf * self.Wp + 255) // 256 * 256

    def rows_in_lin(self, C, f_lo=0, nf=None):
        return rows_lin(C, self.N, self.F if nf is None else nf, self.Fp, self.Wp, f_lo * self.Wp * C)

    def rows_center_lin(self, C, f_lo=0, nf=None):
        return rows_lin(C, self.N, self.F if nf is None else nf, self.Fp, self.Wp, ((2 + f_lo) * self.Wp + 2) * C)

    def interior(self, t):
        return t[:, 2:-2, 2:-2, :]

    # Row maps take an optional frame sub-range [f_lo, f_lo + nf): GEMM row m <-> (window, frame f_lo + f', residue).
    def rows_in(self, C, f_lo=0, nf=None):          # conv input rows: top-left corner of the 5x5 window
        return rows_grid(C, self.N, self.F if nf is None else nf, self.Fp, self.Wp, f_lo * self.Wp * C)

    def rows_center(self, C, ch_off=0, f_lo=0, nf=None):  # the cell itself
        return rows_grid(C, self.N, self.F if nf is None else nf, self.Fp, self.Wp,
                         ((2 + f_lo) * self.Wp + 2) * C + ch_off)

    def tap_offsets(self, C, ck=64):
        """K segment (chunk, df, dn) of the implicit GEMM: ck channels starting at chunk*ck of the cell at (+df rows,
        +dn columns).  Channel chunk OUTERMOST: the 25 shifted re-reads of one activation chunk stay in the XCD's L2."""
        return (0, ck, self.Wp * C, C)

    def seg_shifted(self, f_lo=0):
        """wgrad K segment = window w of a transposed [c][w][f'][n] copy, starting at padded frame row f_lo ..."""
        return (f_lo * self.NP, self.Fp * self.NP, 0)

    def seg_center(self, f_lo=0):
        """... or at padded frame row 2 + f_lo (the un-shifted operand)."""
        return ((2 + f_lo) * self.NP, self.Fp * self.NP, 0)


_N_CU = {}


def cu_count(device):
    """compute units of `device` (asked once per device)"""
    n = _N_CU.get(device)
    if n is None:
        n = _N_CU[device] = torch.cuda.get_device_properties(device).multi_processor_count
    return n


_SPLITK_CAP = int(os.environ.get("DFOLD_SPLITK_CAP", "4"))      # partial tiles in flight per CU the workspace is sized for


def conv_splitk(M, CO, CI, device, tiles=None):
    """Split factor S for a narrow conv launch (few output rows, long K = 25 taps x CI): the 256x320 tile kernel runs
    one workgroup per CU, so tiles x S should fill the CUs in whole rounds.  Cost model in K steps: rounds x (steps/S + a
    fixed ~16 steps for prologue, partial-tile store and the reduction).  S divides the number of 64-channel chunks."""
    if (CO % 320 and tiles is None) or CI % 64 or os.environ.get("DFOLD_CONV_SPLITK", "1") == "0":
        return 1
    n_cu = cu_count(device)
    tiles = ((M + 255) // 256) * (CO // 320) if tiles is None else tiles
    chunks, steps = CI // 64, 25 * (CI // 64)
    best, best_cost = 1, -(-tiles // n_cu) * (steps + 16)
    for S in (2, 4, 5, 10, 20):
        if chunks % S or steps // S < 25 or tiles * S > _SPLITK_CAP * n_cu:
            continue
        # (+ 2 S: the last arriver adds the S partial tiles alone, ~2 K steps' worth of time per partial)
        cost = -(-tiles * S // n_cu) * (steps // S + 16 + 2 * S)
        if cost < 0.9 * best_cost:
            best, best_cost = S, cost
    # stream-K form of the one-wave-per-SIMD kernel (conv_fwd_w4.hip, -1): one persistent workgroup per CU walks an equal share
    # of the (tile, K step) units.  Measured on the cone launches of config 3 (round 5, same-box A/B): the workgroups of a
    # stream-K launch sit at different K positions, so the weight slab / halo rows that lock-stepped workgroups share in the
    # L2 are fetched per workgroup -- K steps run ~1.4 x slower than in a whole-tile launch.  It pays only where the best
    # whole-tile alternative (by the cost model above) runs at <= 0.70 of the evenly-spread time AND a share is at least ~2/3
    # of a tile: 288 tiles: 890 -> 780 us, 544: 1268 -> 1221 (and 176 tiles, model 0.77: 918 -> 807, left alone because
    # 416 tiles, model 0.76, lost: 844 -> 950; 160 tiles, share 0.63: 385 -> 473).
    if _STREAMK and 2 * n_cu <= _SPLITK_CAP * n_cu and tiles <= _SPLITK_CAP * n_cu:
        if tiles * steps / n_cu <= 0.70 * best_cost and tiles >= 0.65 * n_cu:
            return -1
    return best


_STREAMK = os.environ.get("DFOLD_CONV_STREAMK", "1") != "0"
# zero-frame skipping in the tower's backward (ConvTower.backward, functional.ConvTowerFn.backward); 0: every launch walks all of K
CONV_NZ = os.environ.get("DFOLD_CONV_NZ", "1") != "0"
# RowMap mode 2 for N_res % 256 != 0 (conv5x5_fwd, conv5x5_wgrad_tn): 0 never, 1 by conv_lin_wins(), 2 always
_CONV_LIN = int(os.environ.get("DFOLD_CONV_LIN", "1"))


def conv_lin_wins(Wn, nf, N, CO, n_cu, nz=False):
    """Does a conv launch at an N_res that is not a multiple of 256 go to the 512 x 160 kernel through the mode-2 row map?
    Measured (same-box A/B, profiles/r6_conv_lin_ab.txt): sending EVERY launch there loses -- BASELINE config 2 (4 x 32 x 128)
    51.6 -> 58.1 ms per step, a config-1 sampler forward (16 x 96) 8.1 -> 12.6 ms.  The line of cells has (N + 4) / N more rows
    than the launch has cells plus a ragged run per window, and what decides at these sizes is the tile count against the 256
    CUs, not the kernel's 6 % per FLOP: config 2 with 1280 output channels is exactly 256 tiles of 256 x 320 -- one round -- and
    272 tiles of 512 x 160: a second, almost empty round; the thin launches of small windows split K either way and the 512-row
    tile wastes up to a run per launch on top.  So the new path takes: launches that carry zero-frame flags (only this kernel
    can skip tiles; the backward of blocks 0 and 3), output widths the 256 x 320 kernel cannot tile (CO % 320), and dense
    launches that fill at least half the chip in the same number of rounds."""
    if nz or CO % 320:
        return True
    t_old = -(-(Wn * nf * N) // 256) * (CO // 320)
    t_new = -(-(Wn * (-(-(nf * (N + 4)) // 256))) // 2) * (CO // 160)
    return t_old >= n_cu // 2 and -(-t_new // n_cu) <= -(-t_old // n_cu)


_TAIL_SPLIT = os.environ.get("DFOLD_CONV_TAIL_SPLIT", "1") != "0"     # conv5x5_fwd: whole rounds unsplit + the remainder's frames split
# zero-frame-flagged launches: up to this many split-K parts per tile, the number that walks K is chosen on the device from the
# flags (conv_fwd_w4.hip; 0 / DFOLD_CONV_SPLITK=0: never split, bit-identical with the unflagged launch)
_NZ_SPLIT = int(os.environ.get("DFOLD_CONV_NZ_SPLIT", "5"))
# flagged tower backward: dead tiles are left alone instead of being written with the zeros they hold already (ConvTower.backward)
_NZ_KEEP = os.environ.get("DFOLD_CONV_NZ_KEEP", "1") != "0"


def nz_split_parts(CI):
    """parts per tile a zero-frame-flagged conv launch carries (0: none): the largest of 5, 4, 2 that divides the 64-channel
    chunks and is allowed by DFOLD_CONV_NZ_SPLIT"""
    if _NZ_SPLIT < 2 or CI % 64 or os.environ.get("DFOLD_CONV_SPLITK", "1") == "0":
        return 0
    for S in (5, 4, 2):
        if S <= _NZ_SPLIT and (CI // 64) % S == 0:
            return S
    return 0


def conv_tail_frames(nf, tiles_per_frame, n_cu):
    """Frames to cut off the end of a thin conv launch of nf frames (512 x 160 tiles, tiles_per_frame of them per frame) so that
    what stays is whole rounds of n_cu tiles: the remainder must be whole frames and at most a quarter of a round (a larger
    remainder fills the chip well enough on its own).  0: leave the launch alone."""
    tiles = nf * tiles_per_frame
    rem = tiles % n_cu if n_cu > 0 else 0
    if tiles_per_frame <= 0 or tiles <= n_cu or rem == 0 or rem % tiles_per_frame or 4 * rem > n_cu:
        return 0
    return rem // tiles_per_frame


# DFOLD_CONV_SKIP_PAD=1: let the edge tiles of a conv launch skip their all-padding frame taps (dfold_gemm_desc.conv_frames).
# Off by default: the skipped K steps (3.75 %) are exact zeros and the results are bit-identical, but the edge tiles then
# fall out of step with the other workgroups of their XCD, which all stream the same weight K-slab at the same time -- the
# launch gains 0.5 % and its HBM-side fetch goes from 1.27 GB to 3.4 GB (profiles/r2_pmc_conv.json, scripts/pmc_ab.sh).
_SKIP_PAD_TAPS = os.environ.get("DFOLD_CONV_SKIP_PAD", "0") == "1"


def conv5x5_fwd(g, x, wf, bias, out, *, relu=True, resid=None, pre_resid_out=None, relu_mask=None, C2=None, R2=None,
                f_lo=0, nf=None, ws=None, nz=None, nz_keep=False):
    """out[cell] = epi(sum_taps x[cell+tap] @ wf[:, tap, :]^T).  x [Wn,Fp,Wp,CI], wf [CO,25,CI], out [Wn,Fp,Wp,CO].
    f_lo / nf: only the output cells of frames [f_lo, f_lo + nf) are computed (they read x frames f_lo-2 .. f_lo+nf+1).
    nz = (ps, radius): frame flags of grid_load_flags and the distance within which x can be non-zero around the flagged
    frames; output tiles whose input frames are all zero by that statement skip their K walk on the device."""
    CO, _, CI = wf.shape
    flags = GEMM_RELU if relu else 0
    if nz is not None and nz_keep:
        flags |= GEMM_NZ_KEEP     # out / C2 already hold the epilogue of a zero product on the tiles the flags call dead
    R = None
    if resid is not None:
        flags |= GEMM_RESID
        R = resid
    if relu_mask is not None:
        flags |= GEMM_RELUMASK
        R = relu_mask
    if pre_resid_out is not None:
        C2, R2 = pre_resid_out, None
    ck = 64 if CI % 64 == 0 else CI            # K chunk per segment (one MFMA K step when channels allow)
    nf = g.F - f_lo if nf is None else nf
    M = g.Wn * nf * g.N
    # N_res that is not a multiple of 256 (every real protein; BASELINE configs 1 and 2): the window's cells as one line (RowMap
    # mode 2) so that the 512 x 160 one-wave-per-SIMD kernel takes the launch -- 256-row runs of consecutive cells, pad columns
    # computed and not stored ((N + 4) / N of the FLOPs).  DFOLD_CONV_LIN=0: the per-tap 256 x 320 kernel of rounds 2-5.
    lin = _CONV_LIN and g.N % 256 != 0 and CO % 160 == 0 and ck == 64 and out.dtype == BF16 and x.dtype == BF16
    if lin:
        # the launch reads up to grid_slack() elements behind the last window (rows that are never stored): only tensors that
        # own that much storage behind them take this path (Grid.alloc / Workspace.get provide it; a bare torch tensor does not)
        lin = has_grid_slack(g, x)
    if lin and _CONV_LIN == 1:
        lin = conv_lin_wins(g.Wn, nf, g.N, CO, cu_count(x.device), nz is not None)
    if lin:
        Ml = g.Wn * g.vw(nf)
        tiles = ((Ml + 511) // 512) * (CO // 160)
        S, sk = 1, {}
        if ws is not None:
            S = conv_splitk(Ml, CO, CI, x.device, tiles=tiles)
            if S != 1:
                sk = dict(splitk=S, splitk_ws=ws.get("splitk_ws", (_SPLITK_CAP * cu_count(x.device) * 256 * 320,), torch.float32),
                          splitk_cnt=ws.get("splitk_cnt", (_SPLITK_CAP * cu_count(x.device),), torch.int32))
            elif nz is not None and nz_split_parts(CI) and tiles <= _SPLITK_CAP * cu_count(x.device):
                sk = dict(splitk=-nz_split_parts(CI), splitk_ws=ws.get("splitk_ws", (_SPLITK_CAP * cu_count(x.device) * 256 * 320,), torch.float32),
                          splitk_cnt=ws.get("splitk_cnt", (_SPLITK_CAP * cu_count(x.device),), torch.int32))
        return gemm(x, wf, out, Ml, CO, ck, nseg=25 * (CI // ck), a_rows=g.rows_in_lin(CI, f_lo, nf),
                    c_rows=g.rows_center_lin(CO, f_lo, nf), ldb=25 * CI, **sk, bias=bias, R=R, C2=C2, R2=R2,
                    a_seg=g.tap_offsets(CI, ck), b_seg=(0, ck, 5 * CI, CI), seg_div=5, seg_div_mid=5, flags=flags,
                    nz=None if nz is None else (nz[0], nz[1], f_lo))
    if ws is not None and _TAIL_SPLIT and g.N % 256 == 0 and (g.Wn * g.N) % 512 == 0 and CO % 160 == 0 and conv_splitk(M, CO, CI, x.device) != 1:
        # a launch of whole rounds of tiles plus a small remainder (288 = 256 + 32 tiles, 544 = 512 + 32 for 256 CUs): the
        # remainder's FRAMES go into a launch of their own, which splits K; the whole rounds run unsplit and in lock-step.
        # (The policy is a function of the launch shape alone -- not of which launches came before it in the process: step 1
        #  takes the same routes, i.e. the same fp32 association, as every later step.)
        nf_b = conv_tail_frames(nf, (g.Wn * g.N // 512) * (CO // 160), cu_count(x.device))
        if nf_b:
            kw = dict(relu=relu, resid=resid, pre_resid_out=pre_resid_out, relu_mask=relu_mask,
                      C2=None if pre_resid_out is not None else C2, R2=None if pre_resid_out is not None else R2, ws=ws, nz=nz,
                      nz_keep=nz_keep)
            conv5x5_fwd(g, x, wf, bias, out, f_lo=f_lo, nf=nf - nf_b, **kw)
            return conv5x5_fwd(g, x, wf, bias, out, f_lo=f_lo + nf - nf_b, nf=nf_b, **kw)
    S, sk = 1, {}
    if ws is not None:        # thin grids split K: the frame sub-range launches of the last-frame mode, and every launch of a small
        # window -- BASELINE config 1 (16 x 96: 12 / 24 output tiles for 256 CUs) ran its 32 conv launches at 5 % of the chip,
        # 13 of the 15.6 ms of a sampler forward (conv_splitk returns 1 wherever the tiles fill the CUs: the headline shapes)
        S = conv_splitk(M, CO, CI, x.device)
        if S == -1 and not (g.N % 256 == 0 and ck == 64 and out.dtype == BF16):
            S = 1                     # (the launch would not take the kernel that has the stream-K form)
        if S > 1 or S == -1:
            sk = dict(splitk=S, splitk_ws=ws.get("splitk_ws", (_SPLITK_CAP * cu_count(x.device) * 256 * 320,), torch.float32),
                      splitk_cnt=ws.get("splitk_cnt", (_SPLITK_CAP * cu_count(x.device),), torch.int32))
        elif (nz is not None and nz_split_parts(CI) and g.N % 256 == 0 and M % 512 == 0 and CO % 160 == 0 and ck == 64 and out.dtype == BF16
              and (M // 512) * (CO // 160) <= _SPLITK_CAP * cu_count(x.device) and not _SKIP_PAD_TAPS):
            # a flagged launch on the 512 x 160 kernel: the device decides how many parts per live tile walk K (nz_split_parts)
            sk = dict(splitk=-nz_split_parts(CI), splitk_ws=ws.get("splitk_ws", (_SPLITK_CAP * cu_count(x.device) * 256 * 320,), torch.float32),
                      splitk_cnt=ws.get("splitk_cnt", (_SPLITK_CAP * cu_count(x.device),), torch.int32))
    return gemm(x, wf, out, M, CO, ck, nseg=25 * (CI // ck), a_rows=g.rows_in(CI, f_lo, nf),
                c_rows=g.rows_center(CO, 0, f_lo, nf), ldb=25 * CI, **sk, bias=bias, R=R, C2=C2, R2=R2, a_seg=g.tap_offsets(CI, ck), b_seg=(0, ck, 5 * CI, CI),
                seg_div=5, seg_div_mid=5, flags=flags, conv_frames=(f_lo << 16) | g.F if _SKIP_PAD_TAPS else 0,
                nz=None if nz is None else (nz[0], nz[1], f_lo))


def grid_transpose_shift(g, x, C, d0, nd, out, colsum=None, f0=0, nf=None):
    """transposed (column-shifted) copies of the padded frame rows [f0, f0+nf) of x (default: all F+4 rows)."""
    nf = g.Fp - f0 if nf is None else nf
    check(_lib.lib().dfold_grid_transpose_shift(_p(x), _p(out), c_int32(g.Wn), c_int32(g.Fp), c_int32(g.Wp), c_int32(C),
                                                c_int32(g.N), c_int32(g.NP), c_int32(d0), c_int32(nd), c_int32(f0), c_int32(nf),
                                                _p(colsum), stream()),
          "dfold_grid_transpose_shift")
    return out


# DFOLD_WGRAD_TN=0: weight gradients through the transposed, column-shifted copies (dfold_grid_transpose_shift + the NT engine)
# also where the direct form applies.  Default: csrc/conv_wgrad_tn.hip reads the channels-last grids as they lie.
_WGRAD_TN = os.environ.get("DFOLD_WGRAD_TN", "1") != "0"


def wgrad_tn_ok(g, CI, CO):
    """shapes the direct (transpose-read) weight-gradient kernel covers: frame rows of whole 64-cell K chunks, the wider
    channel count a multiple of the 256-row tile, the narrower of the 64-channel column tile"""
    return (g.N % 64 == 0 or _CONV_LIN) and max(CI, CO) % 256 == 0 and min(CI, CO) % 64 == 0


def grid_load_flags(g, src, grid, ps, scratch, f_off=0):
    """src bf16 [Wn,nf,N,C] (None: read the grid) -> interior frames [f_off, f_off+nf) of the padded grid; ps int32 [Wn, F+5]:
    prefix sums of the non-zero frame flags (dfold_grid_load_flags).  scratch int32 [Wn*(F+4)+1], zero on entry and exit."""
    C = grid.shape[-1]
    nf = g.F - f_off if src is None else src.shape[1]
    check(_lib.lib().dfold_grid_load_flags(_p(src), _p(grid), _p(ps), _p(scratch), c_int32(g.Wn), c_int32(g.F), c_int32(g.N),
                                           c_int32(C), c_int32(f_off), c_int32(nf), stream()), "dfold_grid_load_flags")
    return ps


def conv5x5_wgrad_tn(g, x, gy, dwg, accumulate=True, bias_grad=None, f_lo=0, nf=None, nz=None):
    """conv5x5_wgrad without operand copies (same accumulator layouts, same frame-range semantics).
    nz = (ps, radius): frame flags for gy (grid_load_flags): frame rows in which gy is zero by them leave the reduction."""
    CI, CO = x.shape[-1], gy.shape[-1]
    F = g.F - f_lo if nf is None else nf
    if bias_grad is not None:      # the bias gradient used to ride on the transposing copy of gy: one column-sum pass now
        if F == g.F:
            colsum_bf16(gy, bias_grad, g.Wn * g.Fp * g.Wp, CO, CO)          # the border of the grid is zero
        else:          # the frame range of every window in one launch (eight launches of ~7 us each per cone weight gradient before)
            check(_lib.lib().dfold_colsum_bf16_batched(_p(gy, (2 + f_lo) * g.Wp * CO), _p(bias_grad), c_int64(F * g.Wp), c_int32(CO),
                                                       c_int64(CO), c_int32(g.Wn), c_int64(g.Fp * g.Wp * CO), stream()),
                  "dfold_colsum_bf16_batched")
    if CI <= CO:     # rows = gy channels, columns = x channels shifted by the tap
        a, b, flip = gy, x, 0
    else:            # rows = x channels over the gy range widened by 2 frames, columns = gy shifted by the flipped tap
        lo = max(0, f_lo - 2)
        F = min(g.F, f_lo + F + 2) - lo
        f_lo = lo
        a, b, flip = x, gy, 1
    if nz is not None and (F > 64 or g.N % 64):
        nz = None                    # (the kernel's frame masks are 64 bits per window; the linear walk of a ragged N_res has no frames)
    # with frame flags the kernel takes at most 8 windows per call (its masks live in scalar registers)
    step = g.Wn if nz is None else 8
    for w0 in range(0, g.Wn, step):
        wn = min(step, g.Wn - w0)
        check(_lib.lib().dfold_conv_wgrad_tn(_p(a, w0 * g.Fp * g.Wp * a.shape[-1]), _p(b, w0 * g.Fp * g.Wp * b.shape[-1]), _p(dwg),
                                             c_int32(a.shape[-1]), c_int32(b.shape[-1]), c_int32(wn), c_int32(g.Fp), c_int32(g.Wp),
                                             c_int32(g.N), c_int32(f_lo), c_int32(F), c_int32(flip),
                                             c_int32(1 if (accumulate or w0 > 0) else 0),
                                             _p(None if nz is None else nz[0], w0 * (g.Fp + 1)), c_int32(0 if nz is None else nz[1]),
                                             stream()), "dfold_conv_wgrad_tn")
    return dwg


def conv5x5_wgrad(g, x, gy, dwg, ws, accumulate=True, bias_grad=None, f_lo=0, nf=None, tn=None, nz=None):
    """dW (+)= sum_cells gy[cell] (x) x[cell+tap].  x [.., CI], gy [.., CO] padded grids.  The narrower operand gets the 5
    column-shifted transposed copies, the wider one a single copy; the WIDER operand is always the GEMM's M side
    (1280 = 5 x 256 rows, the narrow 640 = 2 x 320 columns: both tile exactly), so
      CI <= CO: dwg fp32 [CO][25][CI]            (gy rows x shifted-x columns)
      CI >  CO: dwg fp32 [CI][25][CO] TRANSPOSED (x rows x shifted-gy columns, taps flipped).
    bias_grad (fp32 [CO], accumulated): the conv bias gradient, summed while gy is transposed (no extra pass).
    f_lo / nf: gy is zero outside frames [f_lo, f_lo + nf): only those cells enter the reduction."""
    CI, CO = x.shape[-1], gy.shape[-1]
    if (_WGRAD_TN if tn is None else tn) and wgrad_tn_ok(g, CI, CO) and (g.N % 64 == 0 or (has_grid_slack(g, x) and has_grid_slack(g, gy))):
        return conv5x5_wgrad_tn(g, x, gy, dwg, accumulate, bias_grad, f_lo, nf, nz)
    plane, N = g.plane, g.NP        # N: K-run length of one frame row in the transposed copies
    tag = "" if g.N == g.NP else "_n%d" % g.N   # ragged N: own scratch (its pad columns must never have been written)
    F = g.F - f_lo if nf is None else nf
    fl = GEMM_ACCUM if accumulate else 0
    if CI <= CO:   # shift x
        # the K range of window w reads padded frame rows f_lo .. f_lo+F+3 of the shifted copies, f_lo+2 .. f_lo+F+1 of the
        # un-shifted one: only those rows are (re)written
        tS = grid_transpose_shift(g, x, CI, 0, 5, ws.get("tS" + tag, (5 * CI * plane + 64,)), f0=f_lo, nf=F + 4)
        tU = grid_transpose_shift(g, gy, CO, 2, 1, ws.get("tU" + tag, (CO * plane + 64,)), colsum=bias_grad, f0=f_lo + 2, nf=F)
        gemm(tU, tS, dwg, CO, CI, F * N, nseg=g.Wn, a_rows=rows_plain(plane), c_rows=rows_plain(25 * CI), ldb=plane,
             a_seg=g.seg_center(f_lo), b_seg=g.seg_shifted(f_lo), nbatch=25, nb1=5, sb=(N, CI * plane),
             sc=(5 * CI, CI), flags=fl)
    else:          # shift gy, taps flipped, transposed accumulator
        # here the reduction runs over the cells of x (the un-shifted operand): a cell pairs with gy up to 2 frames away,
        # so the x range is the gy range widened by 2 frames on both sides (clipped to the grid)
        lo = max(0, f_lo - 2)
        F = min(g.F, f_lo + F + 2) - lo
        f_lo = lo
        tS = grid_transpose_shift(g, gy, CO, 0, 5, ws.get("tS" + tag, (5 * CO * plane + 64,)), colsum=bias_grad, f0=f_lo, nf=F + 4)
        tU = grid_transpose_shift(g, x, CI, 2, 1, ws.get("tU" + tag, (CI * plane + 64,)), f0=f_lo + 2, nf=F)
        gemm(tU, tS, dwg, CI, CO, F * N, nseg=g.Wn, a_rows=rows_plain(plane), c_rows=rows_plain(25 * CO), ldb=plane,
             a_seg=g.seg_center(f_lo), b_seg=g.seg_shifted(f_lo), nbatch=25, nb1=5, sb=(N, CO * plane),
             sc=(-5 * CO, -CO), c_off=24 * CO, flags=fl)
    return dwg


class Workspace:
    """Named scratch buffers reused across calls (caller-owned workspace, as the C ABI requires)."""

    def __init__(self, device):
        self.device, self.bufs = device, {}

    def get(self, name, shape, dtype=BF16, zero=False):
        n = 1
        for s in shape:
            n *= s
        key = (name, tuple(shape), dtype)
        b = self.bufs.get(key)
        if b is None:
            # (padded conv grids [Wn, Fp, Wp, C] get the zero slack behind them that grid_slack() describes)
            slack = grid_slack(shape[2], shape[3]) if (len(shape) == 4 and dtype == BF16) else 0
            b = torch.zeros(n + slack, dtype=dtype, device=self.device)
            self.bufs[key] = b
        elif zero:
            b[:n].zero_()
        return b[:n].view(*shape)


class ConvTower:
    """The shared conv_0 residual tower (reference ConvNet, src/model/ipa_pytorch_dynamic.py:664-706):
    4 x [h = relu(conv2(relu(conv1(h)))) + h], conv1 C->C/2, conv2 C/2->C, 5x5, zero pad 2, bias.
    Holds bf16 packed weights (refreshed by pack()) and fp32 gradient accumulators in GEMM layout."""

    def __init__(self, weights, biases):
        # weights: list of 8 fp32 [CO,CI,5,5] tensors in order conv1.0, conv1.2, conv2.0, ...; biases likewise
        self.weights, self.biases = weights, biases
        dev = weights[0].device
        self.wf = [torch.empty((w.shape[0], 25, w.shape[1]), dtype=BF16, device=dev) for w in weights]
        self.wd = [torch.empty((w.shape[1], 25, w.shape[0]), dtype=BF16, device=dev) for w in weights]
        # fp32 gradient accumulators in GEMM layout: [CO][25][CI], or TRANSPOSED [CI][25][CO] when CI > CO (conv5x5_wgrad)
        self.dwg = [torch.zeros((max(w.shape[:2]), 25, min(w.shape[:2])), dtype=torch.float32, device=dev) for w in weights]
        self.db = [torch.zeros_like(b, dtype=torch.float32) for b in biases]
        self.ws = Workspace(dev)
        self.pool = Workspace(dev)   # persistent activation grids of the applications of a step (grid())
        self.pool_bytes = 0
        self.slot_gen = {}
        self._apps = []           # weak references to the tokens of tracked applications whose backward has not run yet
        self.fresh = [True] * len(weights)   # accumulator j holds nothing yet this step: its first wgrad overwrites
        self.on_final = None      # callback(param) after a layer's gradient has been handed to .grad (dp.GradReducer)
        self._stamp = None

    def refresh(self):
        """re-pack the bf16 operands if any fp32 parameter changed (optimizer step / load_state_dict)."""
        stamp = tuple((w.data_ptr(), w._version) for w in self.weights)
        if stamp != self._stamp:
            self.pack()
            self._stamp = stamp

    def finalize_layer(self, j):
        """Hand the accumulated gradient of conv layer j to the parameters' .grad (reference layout [CO,CI,5,5]; added to
        an existing .grad like autograd's own accumulation) and re-arm its accumulators.  Called once per step and layer,
        right after the LAST weight-gradient product into accumulator j -- for the layers of the top of the tower that is
        long before the backward of the step ends, which is what lets a data-parallel reducer overlap their all-reduce
        with the remaining backward work (dp.GradReducer, `on_final`)."""
        w, b, dwg, db = self.weights[j], self.biases[j], self.dwg[j], self.db[j]
        if self.fresh[j]:
            return                                   # nothing was accumulated (layer not reached this step)
        acc = w.grad is not None
        if not acc:
            w.grad = torch.empty_like(w)
        check(_lib.lib().dfold_conv_wgrad_unpack(_p(dwg), _p(w.grad), c_int32(w.shape[0]), c_int32(w.shape[1]),
                                                 c_int32(1 if acc else 0), c_int32(1 if w.shape[1] > w.shape[0] else 0),
                                                 stream()), "dfold_conv_wgrad_unpack")
        if b.grad is None:
            b.grad = db.clone()
        else:
            b.grad.add_(db)
        db.zero_()
        self.fresh[j] = True
        if self.on_final is not None:
            self.on_final(w)
            self.on_final(b)

    def collect_grads(self):
        """The accumulated gradients as fresh tensors in the reference layout, interleaved [w0, b0, w1, b1, ...] (None for
        a layer that accumulated nothing), accumulators re-armed: what ConvTowerFn.backward returns through autograd."""
        out = []
        for j, (w, b, dwg, db) in enumerate(zip(self.weights, self.biases, self.dwg, self.db)):
            if self.fresh[j]:
                out += [None, None]
                continue
            gw = torch.empty_like(w)
            check(_lib.lib().dfold_conv_wgrad_unpack(_p(dwg), _p(gw), c_int32(w.shape[0]), c_int32(w.shape[1]), c_int32(0),
                                                     c_int32(1 if w.shape[1] > w.shape[0] else 0), stream()),
                  "dfold_conv_wgrad_unpack")
            out += [gw, db.clone()]
            db.zero_()
            self.fresh[j] = True
        return out

    # ---- live tracked applications (functional.ConvTowerFn) ----
    class _App:
        __slots__ = ("group", "__weakref__")

    def _live(self, group=None):
        self._apps = [r for r in self._apps if r() is not None]
        if group is None:
            return len(self._apps)
        return sum(1 for r in self._apps if (lambda t: t is not None and t.group == group)(r()))

    @property
    def pending(self):
        """tracked applications whose backward has not run yet and whose graph is still alive"""
        return self._live()

    def register_application(self, new_group):
        """A tracked application joins the group of applications of ONE forward pass of the model (new_group: it is the
        first one of such a pass).  The gradients of a group are delivered by whichever of its applications runs its
        backward last.  Tokens die with their graph, so a forward whose loss was dropped, a validation pass without
        no_grad or an exception between forward and backward leaves no stale count behind; a graph that is merely kept
        alive (outputs stored somewhere) is its own group and does not block the delivery of later ones."""
        if self._live() == 0 and not all(self.fresh):
            self.zero_grad()          # partial sums of a graph that died between two of its backward applications
        if new_group or not self._apps:
            self._group = getattr(self, "_group", 0) + 1
        import weakref
        tok = ConvTower._App()
        tok.group = self._group
        self._apps.append(weakref.ref(tok))
        return tok

    def pending_in_group(self, tok):
        return self._live(tok.group)

    def complete_application(self, tok):
        self._apps = [r for r in self._apps if r() is not None and r() is not tok]

    def pack(self):
        L = _lib.lib()
        for w, wf, wd in zip(self.weights, self.wf, self.wd):
            check(L.dfold_conv_weight_pack(_p(w.detach()), _p(wf), _p(wd), c_int32(w.shape[0]), c_int32(w.shape[1]), stream()),
                  "dfold_conv_weight_pack")

    def zero_grad(self):
        for t in self.db:
            t.zero_()
        self.fresh = [True] * len(self.weights)

    def reset_step(self):
        """Start of a training step: forget applications whose backward never ran (an exception between forward and
        backward, a forward whose loss was dropped) so that their count and partial sums cannot leak into this step."""
        if self._apps or not all(self.fresh):
            self._apps = []
            self.zero_grad()

    @staticmethod
    def cone(F, i):
        """Frame ranges of residual block i (0..3) when only the LAST frame of the tower output is consumed (training:
        the frame update and every live loss term read frame F-1 only, reference ipa_pytorch_dynamic.py:869 and
        train_DFOLD_dynamics.py:1219-1340).  Each 5x5 conv widens the dependency cone by 2 frames, so block i must
        produce its output on the last 4*(3-i)+1 frames and its inner activation on 2 more; everything below the cone
        has exactly zero gradient and no influence on the loss.  Returns ((f_lo1, nf1), (f_lo2, nf2))."""
        r = 4 * (3 - i)
        nf1, nf2 = min(F, r + 3), min(F, r + 1)
        return (F - nf1, nf1), (F - nf2, nf2)

    POOL_SLOTS = 8       # tracked applications per step served from the pool (the reference model has 4)
    POOL_CAP_BYTES = int(float(os.environ.get("DFOLD_POOL_GB", "64")) * (1 << 30))   # of 288 GB; cfg3 uses 8.4 GB per mode

    def grid(self, g, C, slot, name, last_frame_only=False):
        """A zero-bordered activation grid [Wn,Fp,Wp,C].  slot None: a fresh zero-filled tensor.  Otherwise a persistent
        buffer of the pool, keyed by (slot, name, mode, shape): the conv launches only ever write interior cells (all of
        them in the all-frames mode, the same dependency cone every time in the last-frame mode), so the border -- and,
        in the last-frame mode, everything below the cone -- stays the zero it was allocated as, and the 191 MB fill per
        grid and application disappears from the step.  A slot is one application of the tower inside a step (its
        activations live until that application's backward); passes without a backward share the slot "nograd"."""
        if slot is None:
            return g.alloc(C)
        shape = (g.Wn, g.Fp, g.Wp, C)
        key = "%s/%s/%d" % (slot, name, 1 if last_frame_only else 0)
        if (key, shape, BF16) not in self.pool.bufs:
            # a caller that keeps changing shapes (sampling proteins of many lengths) must not grow the pool without bound:
            # past the cap the pool is dropped (tensors already handed out stay alive through their references; nothing of
            # a pending application is in the dict's sole custody because `saved` holds them)
            need = 2 * g.Wn * g.Fp * g.Wp * C
            if self.pool_bytes + need > self.POOL_CAP_BYTES:
                self.pool.bufs.clear()
                self.pool_bytes = 0
            self.pool_bytes += need
        return self.pool.get(key, shape)

    def slot(self, track):
        """pool slot of the application about to run (None: beyond the pool, fresh tensors).  Taking a slot starts a new
        generation of it: a backward that still holds activations of an earlier generation (a graph kept across
        reset_step) must not run on the overwritten buffers -- check_slot() raises."""
        if not track:
            return "nograd"
        if self.pending >= self.POOL_SLOTS:
            return None
        self.slot_gen[self.pending] = self.slot_gen.get(self.pending, 0) + 1
        return self.pending

    def check_slot(self, slot, gen):
        if slot is not None and slot != "nograd" and self.slot_gen.get(slot) != gen:
            raise RuntimeError("ConvTower: the activations of this application were overwritten by a later forward pass "
                               "(backward of a graph from before reset_step / a previous step)")

    def forward(self, g, h0, save=True, last_frame_only=False, slot=None):
        """h0: padded grid [Wn,Fp,Wp,C] bf16.  Returns (h4, saved).  last_frame_only: compute the dependency cone of
        the last output frame only (h4 is then valid on frame F-1 alone, zero elsewhere).  slot: see grid()."""
        C = h0.shape[-1]
        saved = [h0] if save else None
        h = h0
        for i in range(4):
            (l1, n1), (l2, n2) = self.cone(g.F, i) if last_frame_only else ((0, g.F), (0, g.F))
            u = self.grid(g, C // 2, slot, "u%d" % i, last_frame_only)
            ws = self.ws          # thin launches split K: the cone of the training-step mode, and whole small windows (eval)
            conv5x5_fwd(g, h, self.wf[2 * i], self.biases[2 * i], u, relu=True, f_lo=l1, nf=n1, ws=ws)
            hn, v = self.grid(g, C, slot, "h%d" % i, last_frame_only), self.grid(g, C, slot, "v%d" % i, last_frame_only)
            conv5x5_fwd(g, u, self.wf[2 * i + 1], self.biases[2 * i + 1], hn, relu=True, resid=h, pre_resid_out=v,
                        f_lo=l2, nf=n2, ws=ws)
            if save:
                saved += [u, v, hn]
            h = hn
        return h, saved

    def _wgrad(self, g, j, x, gy, f_lo, nf, finalize, nz=None):
        conv5x5_wgrad(g, x, gy, self.dwg[j], self.ws, accumulate=not self.fresh[j], bias_grad=self.db[j], f_lo=f_lo, nf=nf, nz=nz)
        self.fresh[j] = False
        if finalize:
            self.finalize_layer(j)

    def backward(self, g, saved, gtop, last_frame_only=False, finalize=False, nz_ps=None):
        """gtop: dL/dh4 on the padded grid (border zero).  Accumulates dwg/db, returns dL/dh0.
        nz_ps (all-frames mode): the frame flags of gtop (grid_load_flags).  The gradient of stage k of the chain below can
        only be non-zero within 2k frames of a flagged frame (each 5x5 conv widens the set by two frames, masks and residual
        sums do not widen it): every launch is handed the flags with its radius and skips, on the device, the output tiles
        (data gradient) / reduction rows (weight gradient) that are zero by that statement -- loss-agnostic, no host sync,
        results bit-identical to the full launches (the skipped terms are exact zeros).
        finalize: this is the last application of the step whose backward runs (ConvTowerFn counts them): every layer's
        gradient is handed to .grad right after its weight-gradient product here (finalize_layer).
        last_frame_only: gtop is nonzero on frame F-1 only (see cone()); gradients are propagated inside the cone,
        where they are the only nonzero ones.  The scratch grids are re-zeroed first: below the (growing) frame range of
        each step they must read as the zeros the full computation would produce, not as a previous call's values."""
        C = gtop.shape[-1]
        gi = gtop
        ws = self.ws
        full = ((0, g.F), (0, g.F))
        # flagged backward: dead tiles are not written at all (DFOLD_GEMM_NZ_KEEP).  What a dead tile's epilogue would store is a
        # zero -- du = mask . 0; gn = gi + 0 and dv' = mask . gn where gi, the previous stage's result, is itself zero that far from
        # a flagged frame -- and the scratch grids hold zeros there already: zeroed below, and a tile that is dead at one stage
        # was dead at every earlier one (the radius only grows), i.e. never written since.  dv starts as mask . gtop: zero wherever
        # gtop is.  (A dead tile's epilogue read and wrote 0.3 - 0.65 MB at a fraction of the HBM rate: 0.25 ms of a 0.65 - 1.4 ms
        # launch.)
        keep = _NZ_KEEP and nz_ps is not None and not last_frame_only
        if last_frame_only or keep:
            for name, ch in (("du", C // 2), ("g0", C), ("g1", C)):
                ws.get(name, (g.Wn, g.Fp, g.Wp, ch), zero=True)
        dv = relu_mask_bf16(gi, saved[3 * 3 + 2], ws.get("dv", tuple(gtop.shape)))
        for i in (3, 2, 1, 0):
            hprev, u, v = saved[3 * i], saved[3 * i + 1], saved[3 * i + 2]
            (l1, n1), (l2, n2) = self.cone(g.F, i) if last_frame_only else full
            (_, _), (l0, n0) = self.cone(g.F, i - 1) if (last_frame_only and i > 0) else full   # range of dL/d(block input)
            if last_frame_only and i == 0:
                n0 = min(g.F, 17)
                l0 = g.F - n0
            r0 = 4 * (3 - i)            # gi and dv: within r0 frames of a flagged frame of gtop; du: r0 + 2
            nz0 = None if (nz_ps is None or last_frame_only) else (nz_ps, r0)
            nz2 = None if nz0 is None else (nz_ps, r0 + 2)
            self._wgrad(g, 2 * i + 1, u, dv, l2, n2, finalize, nz0)
            du = ws.get("du", tuple(u.shape))
            sws = ws
            conv5x5_fwd(g, dv, self.wd[2 * i + 1], None, du, relu=False, relu_mask=u, f_lo=l1, nf=n1, ws=sws, nz=nz0, nz_keep=keep)
            self._wgrad(g, 2 * i, hprev, du, l1, n1, finalize, nz2)
            gn = ws.get("g%d" % (i & 1), tuple(gtop.shape))
            if i > 0:
                conv5x5_fwd(g, du, self.wd[2 * i], None, gn, relu=False, resid=gi, C2=dv, R2=saved[3 * (i - 1) + 2],
                            f_lo=l0, nf=n0, ws=sws, nz=nz2, nz_keep=keep)
            else:
                conv5x5_fwd(g, du, self.wd[2 * i], None, gn, relu=False, resid=gi, f_lo=l0, nf=n0, ws=sws, nz=nz2, nz_keep=keep)
            gi = gn
        return gi

    def finalize_grads(self):
        """GEMM-layout fp32 accumulators -> .grad of the reference-layout parameters (all layers)."""
        for j in range(len(self.weights)):
            self.finalize_layer(j)
