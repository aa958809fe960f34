"""Autograd nodes of the DFOLDv2 trunk.  Each node's forward AND backward arithmetic runs in
libdfold_hip.so (bf16 MFMA contraction engine + fused HIP kernels); torch only chains the nodes and owns
the device buffers.  Activations crossing nodes are bf16, geometry (frames, points, scores) fp32."""
import math
import os

import torch
from torch.autograd import Function

from .. import _lib, ops
from .._lib import check, stream
from ..ops import BF16, GEMM_ACCUM, _p, c_int32, c_int64, gemm, rows_grid, rows_plain


# ------------------------------------------------------------------------------------------------
# bf16 weight cache (parameters stay fp32 in the reference layout; the engine consumes bf16 copies)
# ------------------------------------------------------------------------------------------------

class WeightCache:
    """bf16 (and transposed bf16) copies of fp32 parameters.  Entries are validated by object identity (weak reference)
    plus (data_ptr, version): `id()` and device addresses are both recycled after a model is garbage collected."""

    def __init__(self):
        self._w, self._wt = {}, {}

    @staticmethod
    def _stamp(p):
        return (p.data_ptr(), p._version)

    def _lookup(self, table, p):
        e = table.get(id(p))
        if e is not None and e[0]() is p and e[1] == self._stamp(p):
            return e[2]
        return None

    def _store(self, table, p, value):
        import weakref
        key = id(p)
        table[key] = (weakref.ref(p, lambda _r, t=table, k=key: t.pop(k, None)), self._stamp(p), value)
        return value

    def w(self, p):
        v = self._lookup(self._w, p)
        return v if v is not None else self._store(self._w, p, ops.cast_bf16(p.detach()))

    def wt(self, p, pad_rows_to=8):
        """transposed bf16 copy [K][N8] (N zero-padded to a multiple of 8 so it can be a GEMM K axis)."""
        v = self._lookup(self._wt, p)
        if v is not None:
            return v
        w = self.w(p)
        N, K = w.shape
        N8 = (N + pad_rows_to - 1) // pad_rows_to * pad_rows_to
        if N8 != N:
            wp = torch.zeros((N8, K), dtype=BF16, device=w.device)
            wp[:N].copy_(w)
            w = wp
        return self._store(self._wt, p, ops.transpose_bf16(w, N8, K))


CACHE = WeightCache()


def _pad_cols8(g2d):
    """bf16 [M,N] -> [M,N8] zero padded (only for the few narrow heads: N = 6, 14)."""
    M, N = g2d.shape
    N8 = (N + 7) // 8 * 8
    if N8 == N:
        return g2d
    out = torch.zeros((M, N8), dtype=BF16, device=g2d.device)
    out[:, :N].copy_(g2d)
    return out


def _linear_backward(x2d, weight, g2d, need_dx=True, rz=None):
    """x2d bf16 [M,K], g2d bf16 [M,N]  ->  dx bf16 [M,K] | None, dW fp32 [N,K], db fp32 [N].  rz: row-block flags of g2d
    (ops.row_block_flags) for the weight-gradient product"""
    M, K = x2d.shape
    N = weight.shape[0]
    g8 = _pad_cols8(g2d)
    N8 = g8.shape[1]
    dx = None
    if need_dx:
        wt = CACHE.wt(weight)                       # [K][N8]
        dx = torch.empty((M, K), dtype=BF16, device=x2d.device)
        gemm(g8, wt, dx, M, K, N8, a_rows=rows_plain(N8), c_rows=rows_plain(K), ldb=N8)
    if ops.gemm_tn_ok(N8, K, M, ragged=True) and K % 8 == 0 and g8.data_ptr() % 16 == 0 and x2d.data_ptr() % 16 == 0:
        # both operands have the reduction index (the rows) as their slow axis: csrc/tn_gemm.hip reads them as they lie
        dW = ops.weight_grad_tn(g8, x2d, M, N8, K, rz=rz)
        db = torch.zeros(N8, dtype=torch.float32, device=x2d.device)
        ops.colsum_bf16(g8, db, M, N8, N8)
        return dx, dW[:N], db[:N]
    M8 = (M + 7) // 8 * 8
    if M8 == M:
        gT = ops.transpose_bf16(g8, M, N8)          # [N8][M]
        xT = ops.transpose_bf16(x2d, M, K)          # [K][M]
    else:   # the row axis becomes the GEMM K axis: pad it with zero columns to a multiple of 8 (16-byte bf16 chunks)
        gT = torch.zeros((N8, M8), dtype=BF16, device=x2d.device)
        xT = torch.zeros((K, M8), dtype=BF16, device=x2d.device)
        ops.transpose_bf16(g8, M, N8, out=gT, bs_dst=(0, 0), ld_dst=M8)
        ops.transpose_bf16(x2d, M, K, out=xT, bs_dst=(0, 0), ld_dst=M8)
    dW = ops.gemm_reduce_rows(gT, xT, N8, K, M8)
    db = torch.zeros(N8, dtype=torch.float32, device=x2d.device)
    ops.colsum_bf16(g8, db, M, N8, N8)
    return dx, dW[:N], db[:N]


class LinearFn(Function):
    """y = x W^T + b (+ReLU).  x bf16 [..,K]; y bf16 or fp32.  Reference: nn.Linear / openfold Linear."""

    @staticmethod
    def forward(ctx, x, weight, bias, out_fp32, relu):
        K = weight.shape[1]
        x2d = x.reshape(-1, K)
        if not x2d.is_contiguous():
            x2d = x2d.contiguous()
        y = ops.linear_fwd(x2d, CACHE.w(weight), bias.detach() if bias is not None else None,
                           out_dtype=torch.float32 if out_fp32 else BF16, relu=relu)
        ctx.save_for_backward(x2d, weight, y if relu else None)
        ctx.has_bias, ctx.xshape = bias is not None, x.shape
        return y.view(*x.shape[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, gy):
        x2d, weight, y = ctx.saved_tensors
        N = weight.shape[0]
        g2d = gy.reshape(-1, N)
        g2d = ops.cast_bf16(g2d) if g2d.dtype == torch.float32 else g2d.contiguous()
        if y is not None:
            g2d = ops.relu_mask_bf16(g2d, y if y.dtype == BF16 else ops.cast_bf16(y), torch.empty_like(g2d))
        dx, dW, db = _linear_backward(x2d, weight, g2d, need_dx=ctx.needs_input_grad[0])
        return (dx.view(ctx.xshape) if dx is not None else None, dW, db if ctx.has_bias else None, None, None)


def linear(x, weight, bias=None, out_fp32=False, relu=False):
    return LinearFn.apply(x, weight, bias, out_fp32, relu)


class LinearGLNFn(Function):
    """[Linear ->] MyLayerNorm [-> SiLU] with per-window statistics (reference MyLayerNorm,
    src/model/ipa_pytorch_dynamic.py:709-724; embedder tails :757-796; post-IPA :858).
    x bf16 [W, ..., K] -> bf16 [W, ..., N]; the fp32 pre-norm activations never leave the node."""

    @staticmethod
    def forward(ctx, x, weight, bias, silu):
        W = x.shape[0]
        K = weight.shape[1]
        x2d = x.reshape(-1, K)
        if not x2d.is_contiguous():
            x2d = x2d.contiguous()
        h = ops.linear_fwd(x2d, CACHE.w(weight), bias.detach(), out_dtype=torch.float32)
        n = h.numel() // W
        y = torch.empty(h.shape, dtype=BF16, device=h.device)
        stats = torch.empty(2 * W, dtype=torch.float64, device=h.device)
        mr = torch.empty(2 * W, dtype=torch.float32, device=h.device)
        check(_lib.lib().dfold_gln_fwd(_p(h), _p(stats), _p(y), _p(mr), c_int32(W), c_int64(n),
                                       ctypes_float(1e-4), c_int32(1 if silu else 0), stream()), "dfold_gln_fwd")
        ctx.save_for_backward(x2d, weight, h, mr)
        ctx.silu, ctx.W, ctx.n, ctx.xshape = silu, W, n, x.shape
        return y.view(*x.shape[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, gy):
        x2d, weight, h, mr = ctx.saved_tensors
        g = gy.reshape(h.shape).contiguous()
        dh = torch.empty(h.shape, dtype=BF16, device=h.device)
        stats = torch.empty(2 * ctx.W, dtype=torch.float64, device=h.device)
        check(_lib.lib().dfold_gln_bwd(_p(h), _p(g), _p(mr), _p(stats), _p(dh), c_int32(ctx.W), c_int64(ctx.n),
                                       c_int32(1 if ctx.silu else 0), stream()), "dfold_gln_bwd")
        dx, dW, db = _linear_backward(x2d, weight, dh, need_dx=ctx.needs_input_grad[0])
        return (dx.view(ctx.xshape) if dx is not None else None, dW, db, None)


class EmbedInFn(Function):
    """h = SiLU(x W^T + b) for the k <= 16 wide embedder inputs (src/model/ipa_pytorch_dynamic.py:757-796, layers 0-1).
    x fp32 [..., k] -> bf16 [..., 256]."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        k = weight.shape[1]
        x2d = x.reshape(-1, k).contiguous().float()
        out = torch.empty((x2d.shape[0], weight.shape[0]), dtype=BF16, device=x.device)
        check(_lib.lib().dfold_embed_in_fwd(_p(x2d), _p(weight.detach()), _p(bias.detach()), _p(out), c_int64(x2d.shape[0]),
                                            c_int32(k), c_int32(weight.shape[0]), stream()), "dfold_embed_in_fwd")
        ctx.save_for_backward(x2d, weight, bias)
        ctx.xshape = x.shape
        return out.view(*x.shape[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, g):
        x2d, weight, bias = ctx.saved_tensors
        k = weight.shape[1]
        g2d = g.reshape(-1, weight.shape[0]).contiguous()
        dW = torch.zeros(weight.shape, dtype=torch.float32, device=g.device)
        db = torch.zeros(bias.shape, dtype=torch.float32, device=g.device)
        dx = torch.empty_like(x2d) if ctx.needs_input_grad[0] else None
        check(_lib.lib().dfold_embed_in_bwd(_p(x2d), _p(weight.detach()), _p(bias.detach()), _p(g2d), _p(dW), _p(db), _p(dx),
                                            c_int64(x2d.shape[0]), c_int32(k), c_int32(weight.shape[0]), stream()),
              "dfold_embed_in_bwd")
        return (dx.view(ctx.xshape) if dx is not None else None, dW, db)


class AngleResnetFn(Function):
    """AngleResnet up to its raw 2-vectors (openfold/model/structure_module.py:114-151: linear_in(relu s) + linear_initial(relu
    s_initial), two residual blocks a += linear_2(relu(linear_1(relu a))), linear_out(relu a)) as ONE node (round 6; before: a chain
    of LinearFn nodes with aten ReLUs / adds between them).  Every ReLU and residual add rides in a GEMM epilogue: the residual
    stream leaves a launch together with its ReLU'd copy, the next layer's operand (DFOLD_GEMM_RESID + DFOLD_GEMM_C2RELU); in the
    backward the ReLU masks are epilogue masks (DFOLD_GEMM_RELUMASK) and a branch gradient joins the residual gradient stream in
    the launch that produces it (DFOLD_GEMM_MASK2 + DFOLD_GEMM_RESID).  The incoming gradient of a per-position head is zero on
    every row no loss term reads (the reference's loss: all frames but the last): the node flags the 256-row blocks that hold a
    non-zero (dfold_row_block_flags) and its seven dx launches and seven weight-gradient products skip the rest on the device --
    loss-agnostic like the conv tower's zero-frame skipping (a loss that reads every row flags every block).
    s, s_initial bf16 [..., C]; returns fp32 [..., no_angles * 2]."""

    @staticmethod
    def forward(ctx, s, s_initial, *params):
        (w_in, b_in, w_init, b_init, w11, b11, w12, b12, w21, b21, w22, b22, w_out, b_out) = params
        C = w_in.shape[1]
        H = w_in.shape[0]
        s2, si2 = s.reshape(-1, C).contiguous(), s_initial.reshape(-1, C).contiguous()
        M = s2.shape[0]
        dev = s2.device
        new = lambda n: torch.empty((M, n), dtype=BF16, device=dev)
        lin = lambda x, w, b, out, **kw: gemm(x, CACHE.w(w), out, M, w.shape[0], w.shape[1], a_rows=rows_plain(w.shape[1]),
                                              c_rows=rows_plain(w.shape[0]), ldb=w.shape[1], bias=b.detach(), **kw)
        r_s, r_i = ops.relu_mask_bf16(s2, s2, new(C)), ops.relu_mask_bf16(si2, si2, new(C))
        a, r_a0 = new(H), new(H)
        lin(r_s, w_in, b_in, a)
        lin(r_i, w_init, b_init, a, R=a, C2=r_a0, flags=ops.GEMM_RESID | ops.GEMM_C2RELU)
        h0, r_a1 = new(H), new(H)
        lin(r_a0, w11, b11, h0, flags=ops.GEMM_RELU)
        lin(h0, w12, b12, a, R=a, C2=r_a1, flags=ops.GEMM_RESID | ops.GEMM_C2RELU)
        h1, r_a2 = new(H), new(H)
        lin(r_a1, w21, b21, h1, flags=ops.GEMM_RELU)
        lin(h1, w22, b22, a, R=a, C2=r_a2, flags=ops.GEMM_RESID | ops.GEMM_C2RELU)
        out = torch.empty((M, w_out.shape[0]), dtype=torch.float32, device=dev)
        lin(r_a2, w_out, b_out, out)
        ctx.save_for_backward(r_s, r_i, r_a0, h0, r_a1, h1, r_a2, *params)
        ctx.sshape = s.shape
        return out.view(*s.shape[:-1], w_out.shape[0])

    @staticmethod
    def backward(ctx, gy):
        r_s, r_i, r_a0, h0, r_a1, h1, r_a2, *params = ctx.saved_tensors
        (w_in, b_in, w_init, b_init, w11, b11, w12, b12, w21, b21, w22, b22, w_out, b_out) = params
        M, H = r_a2.shape
        C = r_s.shape[1]
        dev = r_s.device
        g = gy.reshape(M, -1)
        g = g.contiguous() if g.dtype == torch.float32 else g.float().contiguous()
        block = 256
        rz = (ops.row_block_flags(g, block), block) if (_ANGLE_RZ and M % block == 0) else None
        nz = None if rz is None else (rz[0], 0, block)
        g8 = _pad_cols8(ops.cast_bf16(g))
        new = lambda n: torch.empty((M, n), dtype=BF16, device=dev)

        def dx(gr, w, out, **kw):        # out = epi(gr @ W): gr bf16 [M, N8], W [N, K] -> [M, K]
            wt = CACHE.wt(w)             # [K][N8]
            return gemm(gr, wt, out, M, w.shape[1], wt.shape[1], a_rows=rows_plain(gr.shape[1]), c_rows=rows_plain(w.shape[1]),
                        ldb=wt.shape[1], nz=nz, **kw)

        def dwb(x, w, gr, with_bias=True):
            _, dW, db = _linear_backward(x, w, gr, need_dx=False, rz=rz)
            return dW, (db if with_bias else None)

        da = dx(g8, w_out, new(H), R=r_a2, flags=ops.GEMM_RELUMASK)
        dW_out, db_out = dwb(r_a2, w_out, g8)
        grads = {}
        for (wa, wb_, h, r_prev, tag) in ((w21, w22, h1, r_a1, 2), (w11, w12, h0, r_a0, 1)):
            dW2, db2 = dwb(h, wb_, da)
            dh = dx(da, wb_, new(H), R=h, flags=ops.GEMM_RELUMASK)
            dW1, db1 = dwb(r_prev, wa, dh)
            da_new = new(H)
            dx(dh, wa, da_new, R=da, R2=r_prev, flags=ops.GEMM_RESID | ops.GEMM_MASK2)
            da = da_new
            grads[tag] = (dW1, db1, dW2, db2)
        dW_in, db_in = dwb(r_s, w_in, da)
        dW_init, _ = dwb(r_i, w_init, da, with_bias=False)
        need = ctx.needs_input_grad
        ds = dx(da, w_in, new(C), R=r_s, flags=ops.GEMM_RELUMASK).view(ctx.sshape) if need[0] else None
        dsi = dx(da, w_init, new(C), R=r_i, flags=ops.GEMM_RELUMASK).view(ctx.sshape) if need[1] else None
        return (ds, dsi, dW_in, db_in, dW_init, db_in.clone(), grads[1][0], grads[1][1], grads[1][2], grads[1][3],
                grads[2][0], grads[2][1], grads[2][2], grads[2][3], dW_out, db_out)


# DFOLD_ANGLE_RZ=0: the angle head's backward launches walk every row (A/B runs of the row-block skipping)
_ANGLE_RZ = os.environ.get("DFOLD_ANGLE_RZ", "1") != "0"


def ctypes_float(v):
    import ctypes
    return ctypes.c_float(v)


def linear_gln(x, weight, bias, silu):
    return LinearGLNFn.apply(x, weight, bias, silu)


# ------------------------------------------------------------------------------------------------
# conv tower node
# ------------------------------------------------------------------------------------------------

class ConvTowerFn(Function):
    """ConvNet (src/model/ipa_pytorch_dynamic.py:664-706) on bf16 [W,F,N,C].  `tower` is the shared
    ops.ConvTower; weight gradients of all applications of a graph are accumulated inside it (GEMM layout, fp32) and
    leave it in the backward of the application that runs last: as ordinary autograd outputs (AccumulateGrad nodes and
    their hooks run, torch.autograd.grad() and a DistributedDataParallel wrapper see them), or -- only when a
    dp.GradReducer has registered itself on the tower (`on_final`) -- layer by layer straight into .grad while the
    remaining layers still compute, so that the reducer can start the all-reduce of the top layers early."""

    @staticmethod
    def forward(ctx, tower, last_frame_only, track, n_parts, *args):
        """args = the n_parts channel slices of the input (bf16 [W,F,N,C_k], concatenated along the channel axis in the
        padded grid itself -- the producers' outputs are copied once, straight into the grid interior; no torch.cat and no
        second copy), then the conv parameters (only there so that autograd sees the dependency).
        last_frame_only: the caller consumes frame F-1 of the output only (training step): the tower evaluates the
        dependency cone of that frame (ops.ConvTower.cone) and returns that frame alone, [W,1,N,C] (the incoming gradient has
        the same shape).
        track (bit 0): a backward will follow (grad mode on and something requires grad, decided by the caller -- inside
        Function.forward grad mode is always off).  Only then are the activations kept and the application counted in
        `tower.pending`; a no_grad pass (sampling, self-conditioning, evaluation) leaves no trace in the tower.
        track bit 1: this application is the first one of a forward pass of the model (ops.ConvTower.register_application)."""
        new_group, track = bool(int(track) & 2), bool(int(track) & 1)
        xs = args[:n_parts]
        Wn, F, N, _ = xs[0].shape
        widths = [x.shape[-1] for x in xs]
        C = sum(widths)
        g = ops.Grid(Wn, F, N, xs[0].device)
        tower.refresh()
        slot = tower.slot(track)
        h0 = tower.grid(g, C, slot, "in")
        inner, off = g.interior(h0), 0
        for x, c in zip(xs, widths):
            inner[..., off:off + c].copy_(x)
            off += c
        h4, saved = tower.forward(g, h0, save=track, last_frame_only=last_frame_only, slot=slot)
        ctx.tower, ctx.g, ctx.saved, ctx.last, ctx.track, ctx.widths = tower, g, saved, last_frame_only, track, widths
        ctx.n_params = len(args) - n_parts
        ctx.slot, ctx.gen = slot, tower.slot_gen.get(slot)
        # a tracked application is alive as long as its graph is: the tower counts live tokens, so a forward whose loss was
        # dropped / a validation pass without no_grad / an exception between forward and backward cannot leave a stale
        # count behind (the token dies with the graph)
        ctx.token = tower.register_application(new_group) if track else None
        if last_frame_only:
            # only frame F-1 of the tower output is defined in this mode: hand back that frame alone ([W,1,N,C]) -- a full-size
            # tensor of zeros around it cost an 84 MB strided copy here and a zero-filled full-size gradient in autograd
            return g.interior(h4)[:, -1:].contiguous()
        return g.interior(h4).contiguous()

    @staticmethod
    def backward(ctx, gy):
        tower, g = ctx.tower, ctx.g
        if not ctx.track or ctx.saved is None:
            raise RuntimeError("ConvTowerFn.backward: forward ran without activation tracking (no_grad) or twice")
        tower.check_slot(ctx.slot, ctx.gen)
        gt = tower.ws.get("gtop", (g.Wn, g.Fp, g.Wp, gy.shape[-1]), zero=ctx.last)
        nz_ps = None
        if ctx.last:
            g.interior(gt)[:, -1:].copy_(gy)            # (gy is [W,1,N,C] in this mode)
        elif ops.CONV_NZ and gy.dtype == ops.BF16 and gy.is_contiguous():
            # one pass: the copy into the padded grid + which frames of the incoming gradient hold a non-zero at all.  The
            # reference's loss reads the last frame (train_DFOLD_dynamics.py:1219-1340), a loss that reads every frame flags
            # every frame: the launches below decide on the device, per tile, what is left to compute (ops.ConvTower.backward)
            nz_ps = ops.grid_load_flags(g, gy, gt, tower.ws.get("nz_ps", (g.Wn, g.Fp + 1), torch.int32),
                                        tower.ws.get("nz_scratch", (g.Wn * g.Fp + 1,), torch.int32))
        else:
            g.interior(gt).copy_(gy)
        # The application whose backward runs last delivers the summed gradients.  With a data-parallel reducer registered
        # (tower.on_final) each layer's gradient goes to .grad as soon as that layer's last weight-gradient product is
        # done (ops.ConvTower.finalize_layer), so that the reducer can start on the top layers while the bottom ones still
        # compute; otherwise the gradients are returned through autograd like any other node's.
        last = tower.pending_in_group(ctx.token) <= 1
        early = last and tower.on_final is not None
        g0 = tower.backward(g, ctx.saved, gt, last_frame_only=ctx.last, finalize=early, nz_ps=nz_ps)
        ctx.saved = None
        tower.complete_application(ctx.token)
        ctx.token = None
        pgrads = [None] * ctx.n_params
        if last and not early:
            pgrads = [t if ctx.needs_input_grad[4 + len(ctx.widths) + k] else None
                      for k, t in enumerate(tower.collect_grads())][:ctx.n_params]
        # one compact copy per input slice (g0 is scratch of the tower, overwritten by the next application's backward;
        # the slices are what the producers' backward nodes read -- no full-width copy in between)
        inner, off, grads = g.interior(g0), 0, []
        for k, c in enumerate(ctx.widths):
            grads.append(inner[..., off:off + c].contiguous() if ctx.needs_input_grad[4 + k] else None)
            off += c
        return (None, None, None, None, *grads, *pgrads)


# ------------------------------------------------------------------------------------------------
# IPA attention core
# ------------------------------------------------------------------------------------------------

def _zT(z2d):
    """transposed copy of the pair tensor rows [R,128] -> [128][R], cached on the tensor (z is reused by all
    four trunk blocks of a step)."""
    c = getattr(z2d, "_dfold_T", None)
    if c is None:
        c = ops.transpose_bf16(z2d, z2d.shape[0], z2d.shape[1])
        z2d._dfold_T = c
    return c


_IPA_FUSED = os.environ.get("DFOLD_IPA_FUSED", "1") != "0"     # A/B switch: "0" = the unfused round-2 chain
_IPA_BWD_FUSED = os.environ.get("DFOLD_IPA_BWD_FUSED", "1") != "0"   # "0": the product / VALU row pass chain instead of csrc/ipa_fused_bwd.hip
_PAIR_PROJ_FUSED = os.environ.get("DFOLD_PAIR_PROJ_FUSED", "1") != "0"   # "0": linear_b / down_z as three GEMM launches (A/B runs)
_IPA_PAIR_STREAM = os.environ.get("DFOLD_IPA_PAIR_STREAM", "1") != "0"   # "0": the pair-value products as batched GEMMs (rounds 1-5)
_IPA_PAIR_TN = os.environ.get("DFOLD_IPA_PAIR_TN", "1") != "0"       # "0": dz's pair product through two transposed copies (rounds 1-5)
_IPA_PAIR_WTN = os.environ.get("DFOLD_IPA_PAIR_WTN", "1") != "0"     # "0": the pair-side weight gradients through transposed copies of z / dpz
_IPA_KEEP_P32 = False      # diagnostic: also write the fp32 copy of the probabilities (nothing reads it)
_IPA_WS = {}


def _ipa_workspace(dev):
    ws = _IPA_WS.get(dev)
    if ws is None:
        ws = _IPA_WS[dev] = ops.Workspace(dev)
    return ws


_WT_CAT = {}


def _wt_cat(w_a, w_b):
    """[K_in][N_a + 8] bf16: the transposed weights of two Linears with the same input width side by side (the second
    zero-padded to 8 rows) -- the B operand of a product whose K axis is the concatenation of their outputs; cached per
    parameter version."""
    key = (id(w_a), id(w_b))
    stamp = (w_a.data_ptr(), w_a._version, w_b.data_ptr(), w_b._version)
    e = _WT_CAT.get(key)
    if e is None or e[0] != stamp:
        e = _WT_CAT[key] = (stamp, torch.cat([CACHE.wt(w_a), CACHE.wt(w_b)], 1).contiguous())
    return e[1]


def _ipa_centre(k_pts):
    """[B*F, 3]: mean key point of every (window, frame), snapped to a 1/8 A grid (any centre is valid)"""
    BF = k_pts.shape[0] * k_pts.shape[1]
    return (torch.round(k_pts.reshape(BF, -1, 3).double().mean(1) * 8) / 8).float().contiguous()


def _ipa_fused_ok(N, C, q_pts, v_pts):
    return _IPA_FUSED and N % 8 == 0 and N <= 512 and C == 256 and q_pts.shape[-2] == 8 and v_pts.shape[-2] == 12


class IpaCoreFn(Function):
    """Attention core of InvariantPointAttention.forward (src/model/ipa_pytorch_dynamic.py:396-502).
      q [B,F,N,H*C] bf16, kv [B,F,N,H*2C] bf16 (k | v per head), q_pts/k_pts [B,F,N,H,8,3], v_pts [B,F,N,H,12,3]
      fp32 global frame, z [B,N,N,c_z] bf16, mask [B,F,N], hw [H]
      -> o [B,F,N,H*C] bf16, o_pt [B,F,N,H,12,3] fp32 (global frame), o_pair [B,F,N,H*c_z/4] bf16."""

    @staticmethod
    def forward(ctx, q, kv, q_pts, k_pts, v_pts, z, w_b, w_dz, b_dz, mask, hw, *rest):
        """rest = (feat_ld,) -- only from IpaFeatFn, which calls this body directly with ITS context: `o` and `o_pair` are
        written straight into the concatenated IPA feature matrix [B,F,N,feat_ld] = [o (H*C) | geo_l (384) | o_pair (H*PZ) |
        geo_g (384)] (:504), returned as a fourth value; the tensors to save are handed back in `ctx._to_save`."""
        L = _lib.lib()
        feat_ld = int(rest[0]) if rest else 0
        B, F, N, HC = q.shape
        H = hw.shape[0]
        C = HC // H
        CZ = z.shape[-1]
        PZ = w_dz.shape[0]
        dev = q.device
        q, kv, z = q.contiguous(), kv.contiguous(), z.contiguous()
        q_pts, k_pts, v_pts = q_pts.contiguous(), k_pts.contiguous(), v_pts.contiguous()
        mask = mask.contiguous().float()
        hwc = hw.detach().contiguous()
        wb, wdz = CACHE.w(w_b), CACHE.w(w_dz)
        NN = N * N
        # pair projections: bias_t [B,H,N,N] fp32, pzT [B,N,PZ,N] bf16, pz [B,N,N,PZ] bf16 (bias of linear_b drops out
        # of the softmax; bias of down_z is added after the aggregation since sum_j P = 1)
        bias_t = torch.empty((B, H, N, N), dtype=torch.float32, device=dev)
        pzT = torch.empty((B, N, PZ, N), dtype=BF16, device=dev)
        pz = torch.empty((B, N, N, PZ), dtype=BF16, device=dev)
        if H == 8 and PZ == 32 and CZ == 128 and N % 8 == 0 and _PAIR_PROJ_FUSED:
            # one pass over z for all three (csrc/ipa_geom.hip): three GEMM launches each read the 134 MB pair tensor for an
            # 8- / 32-wide output (427 us per block at config 3)
            check(L.dfold_ipa_pair_proj(_p(z), _p(wb), _p(wdz), _p(bias_t), _p(pz), _p(pzT), c_int32(B), c_int32(N), stream()),
                  "dfold_ipa_pair_proj")
        else:
            gemm(wb, z, bias_t, H, NN, CZ, a_rows=rows_plain(CZ), c_rows=rows_plain(NN), ldb=CZ, nbatch=B,
                 sb=(NN * CZ, 0), sc=(H * NN, 0))
            gemm(wdz, z, pzT, PZ, N, CZ, a_rows=rows_plain(CZ), c_rows=rows_plain(N), ldb=CZ, nbatch=B * N, nb1=N,
                 sb=(NN * CZ, N * CZ), sc=(N * PZ * N, PZ * N))
            gemm(z, wdz, pz, B * NN, PZ, CZ, a_rows=rows_plain(CZ), c_rows=rows_plain(PZ), ldb=CZ)
        Pb = torch.empty((B, F, H, N, N), dtype=BF16, device=dev)
        direct = feat_ld >= HC + 768 + H * PZ and feat_ld % PZ == 0 and _ipa_fused_ok(N, C, q_pts, v_pts)
        feats = torch.empty((B, F, N, feat_ld), dtype=BF16, device=dev) if direct else None
        o = feats[..., :HC] if direct else torch.empty((B, F, N, HC), dtype=BF16, device=dev)
        o_ld = feat_ld if direct else HC
        o_pt = torch.empty((B, F, N, H, 12, 3), dtype=torch.float32, device=dev)
        alpha = math.sqrt(1.0 / (3 * C))
        if _ipa_fused_ok(N, C, q_pts, v_pts):
            # one launch for logits + softmax + o + o_pt (csrc/ipa_fused.hip): the point terms ride on the matrix cores as
            # bf16-split extra columns of the two attention products, no [B,F,H,N,N] fp32 logits in HBM
            P = torch.empty((B, F, H, N, N), dtype=torch.float32, device=dev) if _IPA_KEEP_P32 else None
            NPv = (N + 63) // 64 * 64
            ws = _ipa_workspace(dev)
            QP = ws.get("QP/%d" % N, (B * F, H, N, 160))
            KP = ws.get("KP/%d" % N, (B * F, H, N, 160))
            kn = ws.get("kn/%d" % N, (B * F, H, N), torch.float32)
            VT = ws.get("VT/%d" % N, (B * F, H, 400, NPv))       # zero-filled once: pad rows / columns are never written
            # any per-(window, frame) centre will do: the mean key point, snapped to a 1/8 A grid so that its value (and with it
            # every rounding downstream) does not depend on how the reduction was tiled for this batch shape
            ctr = _ipa_centre(k_pts)
            check(L.dfold_ipa_aug_prep(_p(q_pts), _p(k_pts), _p(v_pts), _p(hwc), _p(ctr), _p(QP), _p(KP), _p(kn), _p(VT),
                                       c_int32(B), c_int32(F), c_int32(N), c_int32(H), c_int32(NPv), ctypes_float(alpha),
                                       stream()), "dfold_ipa_aug_prep")
            ops.transpose_bf16(kv, N, C, ld_src=2 * HC, out=VT, nbatch=B * F * H, nb1=H, bs_src=(N * 2 * HC, 2 * C), src_off=C,
                               bs_dst=(H * 400 * NPv, 400 * NPv), ld_dst=NPv)                      # rows 0..255 = v^T
            check(L.dfold_ipa_fused_fwd(_p(q), _p(kv), _p(QP), _p(KP), _p(VT), _p(kn), _p(bias_t), _p(mask), _p(ctr), _p(o),
                                        c_int64(o_ld), _p(o_pt), _p(Pb), _p(P), c_int32(B), c_int32(F), c_int32(N), c_int32(H), c_int32(NPv),
                                        ctypes_float(alpha), ctypes_float(math.sqrt(1.0 / 3)), ctypes_float(1e5), stream()),
                  "dfold_ipa_fused_fwd")
        else:
            # unfused chain (N_res > 512): logits S = sqrt(1/(3C)) q k^T (:402-406) on the GEMM engine, row softmax with the
            # fp32 point distances, o = P v, o_pt on the VALU
            P = torch.empty((B, F, H, N, N), dtype=torch.float32, device=dev)
            gemm(q, kv, P, N, N, C, a_rows=rows_plain(HC), c_rows=rows_plain(N), ldb=2 * HC, nbatch=B * F * H, nb1=H,
                 sa=(N * HC, C), sb=(N * 2 * HC, 2 * C), sc=(H * NN, NN), alpha=alpha)
            check(L.dfold_ipa_softmax_fwd(_p(P), _p(bias_t), _p(q_pts), _p(k_pts), _p(mask), _p(hwc), _p(P), _p(Pb),
                                          c_int32(B), c_int32(F), c_int32(N), c_int32(H), ctypes_float(math.sqrt(1.0 / 3)),
                                          ctypes_float(1e5), stream()), "dfold_ipa_softmax_fwd")
            vT = ops.transpose_bf16(kv, N, C, ld_src=2 * HC, nbatch=B * F * H, nb1=H, bs_src=(N * 2 * HC, 2 * C), src_off=C)
            gemm(Pb, vT, o, N, C, N, a_rows=rows_plain(N), c_rows=rows_plain(HC), ldb=N, nbatch=B * F * H, nb1=H,
                 sa=(H * NN, NN), sb=(H * C * N, C * N), sc=(N * HC, C))
            check(L.dfold_ipa_opt_fwd(_p(P), _p(v_pts), _p(o_pt), c_int32(B), c_int32(F), c_int32(N), c_int32(H), stream()),
                  "dfold_ipa_opt_fwd")
        # o_pair[b,f,i,h,:] = sum_j P[b,f,h,i,j] pz[b,i,j,:] + b_dz   (:498-502): per (b,i) a [F*H, N] x [N, PZ] product
        stream_pair = _IPA_PAIR_STREAM and PZ == 32 and N % 16 == 0
        if stream_pair:
            # streaming kernel (csrc/ipa_pair.hip, round 6): one workgroup per (b, i), fragments straight from global memory
            if direct:
                o_pair, dst, ldo, co = feats[..., HC + 384:HC + 384 + H * PZ], feats, feat_ld, HC + 384
            else:
                o_pair = torch.empty((B, F, N, H * PZ), dtype=BF16, device=dev)
                dst, ldo, co = o_pair, H * PZ, 0
            check(L.dfold_ipa_pair_value_fwd(_p(Pb), _p(pzT), _p(b_dz.detach().float().contiguous()), _p(dst), c_int32(B), c_int32(F),
                                             c_int32(N), c_int32(H), c_int64(ldo), c_int64(co), stream()), "dfold_ipa_pair_value_fwd")
        elif direct:
            # the same product written into columns [HC + 384, HC + 384 + H PZ) of the feature matrix: row (f, h) of batch
            # (b, i) lands at ((b F + f) N + i) feat_ld + h PZ -- the grid row map with a "padded width" of N feat_ld / PZ cells
            o_pair = feats[..., HC + 384:HC + 384 + H * PZ]
            gemm(Pb, pzT, feats, F * H, PZ, N, a_rows=rows_plain(NN), c_rows=rows_grid(PZ, H, F, F, N * (feat_ld // PZ)), ldb=N,
                 bias=b_dz.detach(), nbatch=B * N, nb1=N, sa=(F * H * NN, N), sb=(N * PZ * N, PZ * N),
                 sc=(F * N * feat_ld, feat_ld), c_off=HC + 384)
        else:
            o_pair = torch.empty((B, F, N, H * PZ), dtype=BF16, device=dev)
            gemm(Pb, pzT, o_pair, F * H, PZ, N, a_rows=rows_plain(NN), c_rows=rows_grid(PZ, H, F, F, N * H), ldb=N,
                 bias=b_dz.detach(), nbatch=B * N, nb1=N, sa=(F * H * NN, N), sb=(N * PZ * N, PZ * N),
                 sc=(F * N * H * PZ, H * PZ))
        del P            # the backward reads the bf16 probabilities (what the forward products consumed)
        ctx.dims = (B, F, N, H, C, CZ, PZ)
        ctx.ctr = ctr if _ipa_fused_ok(N, C, q_pts, v_pts) else None      # the backward's point terms use the same centre
        saved = (q, kv, q_pts, k_pts, v_pts, z, w_b, w_dz, hwc, Pb, pz)
        if rest:
            ctx._to_save = saved
            return o, o_pt, o_pair, feats
        ctx.save_for_backward(*saved)
        return o, o_pt, o_pair

    @staticmethod
    def backward(ctx, do, do_pt, do_pair):
        return IpaCoreFn._backward(ctx, do, do_pt, do_pair)

    @staticmethod
    def _backward(ctx, do, do_pt, do_pair):
        L = _lib.lib()
        q, kv, q_pts, k_pts, v_pts, z, w_b, w_dz, hw, Pb, pz = ctx.saved_tensors[:11]
        B, F, N, H, C, CZ, PZ = ctx.dims
        HC, NN, dev = H * C, N * N, q.device
        alpha = math.sqrt(1.0 / (3 * C))
        do, do_pair, do_pt = do.contiguous(), do_pair.contiguous(), do_pt.contiguous()
        if _IPA_BWD_FUSED and _ipa_fused_ok(N, C, q_pts, v_pts):
            return IpaCoreFn._backward_fused(ctx, L, do, do_pt, do_pair)
        # dP = do v^T + do_pair pz^T
        dP = torch.empty((B, F, H, N, N), dtype=torch.float32, device=dev)
        gemm(do, kv, dP, N, N, C, a_rows=rows_plain(HC), c_rows=rows_plain(N), ldb=2 * HC, nbatch=B * F * H, nb1=H,
             sa=(N * HC, C), sb=(N * 2 * HC, 2 * C), sc=(H * NN, NN), b_off=C)
        gemm(do_pair, pz, dP, F * H, N, PZ, a_rows=rows_grid(PZ, H, F, F, N * H), c_rows=rows_plain(NN), ldb=PZ,
             nbatch=B * N, nb1=N, sa=(F * N * H * PZ, H * PZ), sb=(NN * PZ, N * PZ), sc=(F * H * NN, N),
             flags=GEMM_ACCUM)
        dSb = torch.empty((B, F, H, N, N), dtype=BF16, device=dev)
        dq_pts = torch.empty_like(q_pts)
        dk_pts = torch.empty_like(k_pts)
        dv_pts = torch.empty_like(v_pts)
        dhw = torch.zeros(H, dtype=torch.float32, device=dev)
        check(L.dfold_ipa_softmax_bwd(_p(Pb), _p(dP), _p(q_pts), _p(k_pts), _p(v_pts), _p(do_pt), _p(hw), _p(dP), _p(dSb),
                                      _p(dq_pts), _p(dhw), _p(ctx.ctr if ctx.ctr is not None else _ipa_centre(k_pts)), c_int32(B), c_int32(F), c_int32(N), c_int32(H),
                                      stream()),
              "dfold_ipa_softmax_bwd")
        dS = dP
        check(L.dfold_ipa_col_bwd(_p(Pb), _p(dS), _p(q_pts), _p(k_pts), _p(do_pt), _p(hw), _p(dk_pts), _p(dv_pts),
                                  c_int32(B), c_int32(F), c_int32(N), c_int32(H), stream()), "dfold_ipa_col_bwd")
        nb = B * F * H
        # dq = alpha dS k
        kT = ops.transpose_bf16(kv, N, C, ld_src=2 * HC, nbatch=nb, nb1=H, bs_src=(N * 2 * HC, 2 * C))
        dq = torch.empty_like(q)
        gemm(dSb, kT, dq, N, C, N, a_rows=rows_plain(N), c_rows=rows_plain(HC), ldb=N, nbatch=nb, nb1=H,
             sa=(H * NN, NN), sb=(H * C * N, C * N), sc=(N * HC, C), alpha=alpha)
        dkv = IpaCoreFn._backward_dkv(ctx, dSb, do)
        return IpaCoreFn._backward_pair_side(ctx, L, do_pair, dS, dq, dkv, dq_pts, dk_pts, dv_pts, dhw)

    @staticmethod
    def _backward_dkv(ctx, dSb, do):
        """dk = alpha dS^T q ; dv = P^T do  into one [B,F,N,H*2C] tensor"""
        q, kv, q_pts, k_pts, v_pts, z, w_b, w_dz, hw, Pb, pz = ctx.saved_tensors[:11]
        B, F, N, H, C, CZ, PZ = ctx.dims
        HC, NN = H * C, N * N
        alpha = math.sqrt(1.0 / (3 * C))
        nb = B * F * H
        dkv = torch.empty_like(kv)
        if ops.gemm_tn_ok(N, C, N):
            # the reduction runs over the query index, the slow axis of dS / P and of q / do: no transposed copies
            ops.gemm_tn(dSb, q, dkv, N, C, N, N, HC, 2 * HC, nbatch=nb, nb1=H, sa=(H * NN, NN), sb=(N * HC, C),
                        sc=(N * 2 * HC, 2 * C), alpha=alpha)
            ops.gemm_tn(Pb, do, dkv, N, C, N, N, HC, 2 * HC, nbatch=nb, nb1=H, sa=(H * NN, NN), sb=(N * HC, C),
                        sc=(N * 2 * HC, 2 * C), c_off=C)
            return dkv
        dSbT = ops.transpose_bf16(dSb, N, N, nbatch=nb, nb1=1, bs_src=(NN, 0))
        qT = ops.transpose_bf16(q, N, C, ld_src=HC, nbatch=nb, nb1=H, bs_src=(N * HC, C))
        gemm(dSbT, qT, dkv, N, C, N, a_rows=rows_plain(N), c_rows=rows_plain(2 * HC), ldb=N, nbatch=nb, nb1=H,
             sa=(H * NN, NN), sb=(H * C * N, C * N), sc=(N * 2 * HC, 2 * C), alpha=alpha)
        del dSbT, qT
        PbT = ops.transpose_bf16(Pb, N, N, nbatch=nb, nb1=1, bs_src=(NN, 0))
        doT = ops.transpose_bf16(do, N, C, ld_src=HC, nbatch=nb, nb1=H, bs_src=(N * HC, C))
        gemm(PbT, doT, dkv, N, C, N, a_rows=rows_plain(N), c_rows=rows_plain(2 * HC), ldb=N, nbatch=nb, nb1=H,
             sa=(H * NN, NN), sb=(H * C * N, C * N), sc=(N * 2 * HC, 2 * C), c_off=C)
        return dkv

    @staticmethod
    def _backward_fused(ctx, L, do, do_pt, do_pair):
        """row pass in one launch (csrc/ipa_fused_bwd.hip): no fp32 dP, no transposed copy of k for dq; the pair-value term of dP
        leaves its product once as bf16"""
        q, kv, q_pts, k_pts, v_pts, z, w_b, w_dz, hw, Pb, pz = ctx.saved_tensors[:11]
        B, F, N, H, C, CZ, PZ = ctx.dims
        HC, NN, dev = H * C, N * N, q.device
        alpha = math.sqrt(1.0 / (3 * C))
        nb = B * F * H
        ws = _ipa_workspace(dev)
        NPv = (N + 63) // 64 * 64
        ctr = ctx.ctr if ctx.ctr is not None else _ipa_centre(k_pts)
        dPp = torch.empty((B, F, H, N, N), dtype=BF16, device=dev)
        if _IPA_PAIR_STREAM and PZ == 32 and N % 4 == 0:
            check(L.dfold_ipa_pair_value_bwd(_p(do_pair), _p(pz), _p(dPp), c_int32(B), c_int32(F), c_int32(N), c_int32(H),
                                             c_int64(H * PZ), stream()), "dfold_ipa_pair_value_bwd")
        else:
            gemm(do_pair, pz, dPp, F * H, N, PZ, a_rows=rows_grid(PZ, H, F, F, N * H), c_rows=rows_plain(NN), ldb=PZ,
                 nbatch=B * N, nb1=N, sa=(F * N * H * PZ, H * PZ), sb=(NN * PZ, N * PZ), sc=(F * H * NN, N))
        DOP = ws.get("DOP/%d" % N, (B * F, H, N, 224))
        VP = ws.get("VP/%d" % N, (B * F, H, N, 224))
        KT = ws.get("KT/%d" % N, (B * F, H, 352, NPv))       # zero-filled once: pad rows / columns are never written
        check(L.dfold_ipa_bwd_prep(_p(do_pt), _p(v_pts), _p(k_pts), _p(ctr), _p(DOP), _p(VP), _p(KT), c_int32(B), c_int32(F),
                                   c_int32(N), c_int32(H), c_int32(NPv), stream()), "dfold_ipa_bwd_prep")
        ops.transpose_bf16(kv, N, C, ld_src=2 * HC, out=KT, nbatch=nb, nb1=H, bs_src=(N * 2 * HC, 2 * C),
                           bs_dst=(H * 352 * NPv, 352 * NPv), ld_dst=NPv)                          # rows 0..255 = k^T
        dS = torch.empty((B, F, H, N, N), dtype=torch.float32, device=dev)
        dSb = torch.empty((B, F, H, N, N), dtype=BF16, device=dev)
        dq = torch.empty_like(q)
        dq_pts = torch.empty_like(q_pts)
        dhw = torch.zeros(H, dtype=torch.float32, device=dev)
        check(L.dfold_ipa_fused_bwd(_p(do), _p(kv), _p(DOP), _p(VP), _p(KT), _p(Pb), _p(dPp), _p(q_pts), _p(hw), _p(ctr), _p(dS),
                                    _p(dSb), _p(dq), _p(dq_pts), _p(dhw), c_int32(B), c_int32(F), c_int32(N), c_int32(H),
                                    c_int32(NPv), ctypes_float(alpha), stream()), "dfold_ipa_fused_bwd")
        del dPp
        dk_pts = torch.empty_like(k_pts)
        dv_pts = torch.empty_like(v_pts)
        check(L.dfold_ipa_col_bwd(_p(Pb), _p(dS), _p(q_pts), _p(k_pts), _p(do_pt), _p(hw), _p(dk_pts), _p(dv_pts),
                                  c_int32(B), c_int32(F), c_int32(N), c_int32(H), stream()), "dfold_ipa_col_bwd")
        dkv = IpaCoreFn._backward_dkv(ctx, dSb, do)
        return IpaCoreFn._backward_pair_side(ctx, L, do_pair, dS, dq, dkv, dq_pts, dk_pts, dv_pts, dhw)

    @staticmethod
    def _backward_pair_side(ctx, L, do_pair, dS, dq, dkv, dq_pts, dk_pts, dv_pts, dhw):
        """gradients of the pair-side inputs (z, linear_b, down_z) from dS and the bf16 probabilities"""
        q, kv, q_pts, k_pts, v_pts, z, w_b, w_dz, hw, Pb, pz = ctx.saved_tensors[:11]
        B, F, N, H, C, CZ, PZ = ctx.dims
        NN, dev = N * N, q.device
        FH = F * H
        dop = do_pair.view(B, F, N, H, PZ).permute(0, 2, 1, 3, 4).contiguous()                          # [B,N,F,H,PZ]
        # dz = dpz W_dz + dbias W_b as ONE product: dpz (PZ columns) and the bias gradient (8 columns) are written side by
        # side into the [B*N*N, PZ + 8] operand and meet the concatenated transposed weights -- one bf16 pass over dz instead
        # of an fp32 product, an accumulating K = 8 product (268 MB read + written for 0.1 GFLOP) and a cast
        KZ = PZ + 8
        dpzb = torch.empty((B * NN, KZ), dtype=BF16, device=dev)
        if _IPA_PAIR_TN and FH % 64 == 0 and N % 8 == 0 and ops.gemm_tn_ok(N, PZ, FH, ragged=True):
            # dpz[b,i,j,c] = sum_(f,h) P[b,f,h,i,j] do_pair[b,f,i,h,c]: the reduction index (f, h) is the slow axis of BOTH operands as
            # they lie (rows of P are N^2 apart, rows of the regrouped do_pair 32 apart): the reduction-major product reads them in
            # place (round 6; before: a 134 MB transposed copy of the probabilities + one of do_pair per block)
            ops.gemm_tn(Pb, dop, dpzb, N, PZ, FH, NN, PZ, KZ, nbatch=B * N, nb1=N, sa=(FH * NN, N), sb=(N * FH * PZ, FH * PZ),
                        sc=(N * N * KZ, N * KZ))
        else:
            PbT2 = ops.transpose_bf16(Pb, FH, N, ld_src=NN, nbatch=B * N, nb1=N, bs_src=(FH * NN, N))       # [B,N(i),N(j),FH]
            dopT = ops.transpose_bf16(dop, FH, PZ, nbatch=B * N, nb1=1, bs_src=(FH * PZ, 0))                 # [B,N,PZ,FH]
            gemm(PbT2, dopT, dpzb, N, PZ, FH, a_rows=rows_plain(FH), c_rows=rows_plain(KZ), ldb=FH, nbatch=B * N, nb1=1,
                 sa=(N * FH, 0), sb=(PZ * FH, 0), sc=(N * KZ, 0))
            del PbT2, dopT
        db_hn = torch.empty((B, H, NN), dtype=BF16, device=dev)
        check(L.dfold_ipa_bias_grad(_p(dS), _p(db_hn), _p(dpzb, PZ), c_int64(KZ), c_int32(B), c_int32(F), c_int32(N), c_int32(H),
                                    ctypes_float(math.sqrt(1.0 / 3)), stream()), "dfold_ipa_bias_grad")
        dz = torch.empty((B * NN, CZ), dtype=BF16, device=dev)
        gemm(dpzb, _wt_cat(w_dz, w_b), dz, B * NN, CZ, KZ, a_rows=rows_plain(KZ), c_rows=rows_plain(CZ), ldb=KZ)
        dz = dz.view(z.shape)
        R = B * NN
        if _IPA_PAIR_WTN and z.dtype == BF16 and z.is_contiguous() and ops.gemm_tn_ok(KZ, CZ, R, ragged=True):
            # dW_dz | dW_b = [dpz | dbias]^T z as ONE reduction-major product straight from the operands as they lie (round 6,
            # second session): dpzb [R, PZ + 8] and z [R, CZ] both have the reduction index as their row.  Before: a transposed copy
            # of z per block (134 MB), one of dpz, and two products with the transposed operands (0.18 + 0.07 + 0.08 + 0.17 ms).
            S = 512
            while S > 1 and R % (S * 64):
                S //= 2
            dw_cat = torch.zeros((KZ, CZ), dtype=torch.float32, device=dev)
            ops.gemm_tn(dpzb, z.view(R, CZ), dw_cat, KZ, CZ, R, KZ, CZ, CZ, splitk=S, flags=ops.GEMM_ATOMIC)
            dw_dz, dw_b = dw_cat[:PZ], dw_cat[PZ:PZ + H]
        else:
            zT = _zT(z.view(R, CZ))                                                                      # [CZ][B*NN]
            dpzT = ops.transpose_bf16(dpzb, R, PZ, ld_src=KZ)                                            # [PZ][B*NN]
            dw_dz = ops.gemm_reduce_rows(dpzT, zT, PZ, CZ, R)
            dw_b = torch.zeros((H, CZ), dtype=torch.float32, device=dev)
            S2 = max(1, min(64, 256 // B))
            while S2 > 1 and NN % (S2 * 64):
                S2 -= 1
            ks = NN // S2                                    # split-K over (window, K slice): 1 output tile otherwise
            gemm(db_hn, zT, dw_b, H, CZ, ks, a_rows=rows_plain(NN), c_rows=rows_plain(CZ), ldb=R, nbatch=B * S2,
                 nb1=S2, sa=(H * NN, ks), sb=(NN, ks), sc=(0, 0), flags=ops.GEMM_ATOMIC)
        db_dz = torch.zeros(PZ, dtype=torch.float32, device=dev)
        ops.colsum_bf16(do_pair, db_dz, do_pair.numel() // PZ, PZ, PZ)
        return dq, dkv, dq_pts, dk_pts, dv_pts, dz, dw_b, dw_dz, db_dz, None, dhw


class IpaPointsFn(Function):
    """raw point projections -> global-frame points (src/model/ipa_pytorch_dynamic.py:363-390).
    raw_q fp32 [..,3*H*Pq], raw_kv fp32 [..,3*H*(Pq+Pv)], t7 fp32 [..,7] -> q_pts [..,H,Pq,3], k_pts, v_pts [..,H,Pv,3]"""

    @staticmethod
    def forward(ctx, raw_q, raw_kv, t7):
        lead = t7.shape[:-1]
        P = t7.numel() // 7
        rq, rkv, t = raw_q.contiguous(), raw_kv.contiguous(), t7.contiguous()
        dev = t7.device
        q_pts = torch.empty(lead + (8, 8, 3), dtype=torch.float32, device=dev)
        k_pts = torch.empty(lead + (8, 8, 3), dtype=torch.float32, device=dev)
        v_pts = torch.empty(lead + (8, 12, 3), dtype=torch.float32, device=dev)
        check(_lib.lib().dfold_ipa_points_fwd(_p(rq), _p(rkv), _p(t), _p(q_pts), _p(k_pts), _p(v_pts), c_int64(P), stream()),
              "dfold_ipa_points_fwd")
        ctx.save_for_backward(rq, rkv, t)
        return q_pts, k_pts, v_pts

    @staticmethod
    def backward(ctx, dq, dk, dv):
        rq, rkv, t = ctx.saved_tensors
        P = t.numel() // 7
        drq, drkv, dt = torch.empty_like(rq), torch.empty_like(rkv), torch.empty_like(t)
        dq, dk, dv = dq.contiguous(), dk.contiguous(), dv.contiguous()   # keep the copies alive across the launch
        check(_lib.lib().dfold_ipa_points_bwd(_p(rq), _p(rkv), _p(t), _p(dq), _p(dk), _p(dv), _p(drq), _p(drkv), _p(dt),
                                              c_int64(P), stream()),
              "dfold_ipa_points_bwd")
        return drq, drkv, dt


class IpaOutFeatFn(Function):
    """global-frame attended points -> the 2 x 384 geometry columns of the IPA output features (:470-488, :504):
    [x|y|z|norm] of R^T(o_pt - t) (local frame) and of o_pt (global frame), bf16."""

    @staticmethod
    def forward(ctx, o_pt, t7, eps):
        lead = t7.shape[:-1]
        P = t7.numel() // 7
        o, t = o_pt.contiguous(), t7.contiguous()
        geo_l = torch.empty(lead + (384,), dtype=BF16, device=t7.device)
        geo_g = torch.empty(lead + (384,), dtype=BF16, device=t7.device)
        check(_lib.lib().dfold_ipa_outfeat_fwd(_p(o), _p(t), _p(geo_l), _p(geo_g), c_int64(384), c_int64(P), ctypes_float(eps),
                                               stream()), "dfold_ipa_outfeat_fwd")
        ctx.save_for_backward(o, t)
        ctx.eps = eps
        return geo_l, geo_g

    @staticmethod
    def backward(ctx, dl, dg):
        o, t = ctx.saved_tensors
        do, dt = IpaOutFeatFn._bwd(o, t, dl, dg, ctx.eps)
        return do, dt, None

    @staticmethod
    def _bwd(o, t, dl, dg, eps):
        P = t.numel() // 7
        do, dt = torch.empty_like(o), torch.empty_like(t)
        dl, dg = dl.contiguous(), dg.contiguous()     # (slices of the concatenated feature gradient: real copies)
        check(_lib.lib().dfold_ipa_outfeat_bwd(_p(o), _p(t), _p(dl), _p(dg), _p(do), _p(dt),
                                               c_int64(P), ctypes_float(eps), stream()), "dfold_ipa_outfeat_bwd")
        return do, dt


class IpaFeatFn(Function):
    """Attention core + output features + cat([o, geo_l, o_pair, geo_g], -1) (:396-504) as ONE node whose output is the
    operand of linear_out: the fused attention kernel, the pair-value product and the output-feature kernel write their
    column blocks straight into it (no torch.cat: 403 MB read + written per block at config 3).  Same kernels and the same
    backward as IpaCoreFn + IpaOutFeatFn (their bodies are called with this node's context); only for shapes the fused
    attention kernel covers (`ipa_feat_direct_ok`)."""

    @staticmethod
    def forward(ctx, q, kv, q_pts, k_pts, v_pts, z, w_b, w_dz, b_dz, mask, hw, t7, eps):
        H, PZ = hw.shape[0], w_dz.shape[0]
        HC = q.shape[-1]
        ld = HC + 768 + H * PZ
        o, o_pt, o_pair, feats = IpaCoreFn.forward(ctx, q, kv, q_pts, k_pts, v_pts, z, w_b, w_dz, b_dz, mask, hw, ld)
        if feats is None:
            raise RuntimeError("IpaFeatFn: shape not covered by the fused attention kernel (use IpaCoreFn + IpaOutFeatFn)")
        t = t7.contiguous()
        P = t.numel() // 7
        check(_lib.lib().dfold_ipa_outfeat_fwd(_p(o_pt), _p(t), _p(feats, HC), _p(feats, HC + 384 + H * PZ), c_int64(ld), c_int64(P),
                                               ctypes_float(eps), stream()), "dfold_ipa_outfeat_fwd")
        ctx.save_for_backward(*ctx._to_save, o_pt, t)
        ctx._to_save = None
        ctx.eps, ctx.cols = eps, (HC, H * PZ)
        return feats

    @staticmethod
    def backward(ctx, g):
        o_pt, t = ctx.saved_tensors[11:13]
        HC, HP = ctx.cols
        do_pt, dt7 = IpaOutFeatFn._bwd(o_pt, t, g[..., HC:HC + 384], g[..., HC + 384 + HP:], ctx.eps)
        grads = IpaCoreFn._backward(ctx, g[..., :HC], do_pt, g[..., HC + 384:HC + 384 + HP])
        return tuple(grads) + (dt7, None)


def ipa_feat_direct_ok(N, C, q_pts, v_pts, H=None, PZ=None):
    """the FULL predicate of IpaCoreFn.forward's direct-write path: with H / PZ given, also the row width of the feature
    matrix as a multiple of the pair-value width (the cell trick of its grid row map) -- a model width outside it takes the
    IpaCoreFn + IpaOutFeatFn + cat path instead of raising inside IpaFeatFn"""
    if os.environ.get("DFOLD_IPA_FEAT_DIRECT", "1") == "0" or not _ipa_fused_ok(N, C, q_pts, v_pts):
        return False
    return H is None or PZ is None or (H * C + 768 + H * PZ) % PZ == 0


# ------------------------------------------------------------------------------------------------
# IGSO(3) score series node
# ------------------------------------------------------------------------------------------------

class Igso3SeriesFn(Function):
    """sc = dsig(omega)/(f(omega)+1e-4) (src/data/so3_diffuser.py:71-117); omega fp32 [W, ...], env fp64 [W, L]."""

    @staticmethod
    def forward(ctx, omega, env):
        om = omega.contiguous()
        W, L = env.shape
        sc = torch.empty(om.shape, dtype=torch.float64, device=om.device)
        dsc = torch.empty_like(sc)
        check(_lib.lib().dfold_igso3_series(_p(om), _p(env), _p(sc), _p(dsc), c_int64(om.numel()),
                                            c_int64(om.numel() // W), c_int32(L), stream()), "dfold_igso3_series")
        ctx.save_for_backward(dsc)
        return sc

    @staticmethod
    def backward(ctx, g):
        (dsc,) = ctx.saved_tensors
        return (g * dsc).float(), None


class RotScoreHeadFn(Function):
    """The whole rotation-score head on the device in four launches (csrc/score.hip: dfold_rot_head_pre -> dfold_igso3_series ->
    dfold_rot_head_post; backward dfold_rot_head_bwd): score = igso3_score(quat_to_rotvec(q_0^-1 q_t)) as float64 [..., 3] and its
    gradient w.r.t. the predicted quaternion q_0 (q_t comes from the batch).  quats fp32 [W, ..., 4], env fp64 [W, L]."""

    @staticmethod
    def forward(ctx, quats_t, quats_0, env):
        L_ = _lib.lib()
        qt, q0 = quats_t.detach().float().contiguous(), quats_0.detach().float().contiguous()
        W, L = env.shape
        P = q0.numel() // 4
        dev = q0.device
        vec = torch.empty(q0.shape[:-1] + (3,), dtype=torch.float32, device=dev)
        om = torch.empty(q0.shape[:-1], dtype=torch.float32, device=dev)
        sc, dsc = torch.empty(om.shape, dtype=torch.float64, device=dev), torch.empty(om.shape, dtype=torch.float64, device=dev)
        score = torch.empty(vec.shape, dtype=torch.float64, device=dev)
        check(L_.dfold_rot_head_pre(_p(qt), _p(q0), _p(vec), _p(om), c_int64(P), stream()), "dfold_rot_head_pre")
        check(L_.dfold_igso3_series(_p(om), _p(env), _p(sc), _p(dsc), c_int64(P), c_int64(P // W), c_int32(L), stream()),
              "dfold_igso3_series")
        check(L_.dfold_rot_head_post(_p(vec), _p(om), _p(sc), _p(score), c_int64(P), stream()), "dfold_rot_head_post")
        ctx.save_for_backward(qt, q0, vec, om, sc, dsc)
        ctx.q0_dtype = quats_0.dtype
        return score

    @staticmethod
    def backward(ctx, g):
        qt, q0, vec, om, sc, dsc = ctx.saved_tensors
        gd = g.to(torch.float64).contiguous()
        d_q0 = torch.empty_like(q0)
        check(_lib.lib().dfold_rot_head_bwd(_p(gd), _p(qt), _p(q0), _p(vec), _p(om), _p(sc), _p(dsc), _p(d_q0), c_int64(q0.numel() // 4),
                                            stream()), "dfold_rot_head_bwd")
        return None, d_q0.to(ctx.q0_dtype), None


# ------------------------------------------------------------------------------------------------
# backbone frame update
# ------------------------------------------------------------------------------------------------

class ComposeFn(Function):
    """Rigid.compose_q_update_vec on tensor_7 frames (openfold/utils/rigid_utils.py:1039-1063; call site
    src/model/ipa_pytorch_dynamic.py:871): one HIP launch forward, one backward (csrc/ipa_geom.hip)."""

    @staticmethod
    def forward(ctx, t7, upd6, mask):
        t, u = t7.contiguous().float(), upd6.contiguous().float()
        m = mask.expand(t.shape[:-1] + (1,)).contiguous().float() if mask is not None else None
        out = torch.empty_like(t)
        P = t.numel() // 7
        check(_lib.lib().dfold_compose_fwd(_p(t), _p(u), _p(m), _p(out), c_int64(P), stream()), "dfold_compose_fwd")
        ctx.save_for_backward(t, u, m)
        return out

    @staticmethod
    def backward(ctx, g):
        t, u, m = ctx.saved_tensors
        gc = g.contiguous().float()
        dt, du = torch.empty_like(t), torch.empty_like(u)
        check(_lib.lib().dfold_compose_bwd(_p(t), _p(u), _p(m), _p(gc), _p(dt), _p(du), c_int64(t.numel() // 7), stream()),
              "dfold_compose_bwd")
        return dt, du, None


def compose_q_update_vec(t7, upd6, mask=None):
    return ComposeFn.apply(t7, upd6, mask)
