"""Pair-stack neighbours of the triangle operators: drop-ins for the vendored OpenFold modules
`OuterProductMean` (openfold/model/outer_product_mean.py:26-129, Algorithm 10), `MSATransition`
(openfold/model/evoformer.py:41-117, Algorithm 9), `Dropout` / `DropoutRowwise` / `DropoutColumnwise`
(openfold/model/dropout.py:22-78) and the block that chains them with the triangle operators, `EvoformerBlockCore`
(openfold/model/evoformer.py:120-212).  Same class names, constructor arguments, forward signatures and state_dict
keys; every contraction runs on the bf16 MFMA engine, LayerNorms on the wave-per-row kernel, the triangle operators on
the fused kernels of csrc/pair_fused.hip.  Device tensors only (no CPU fallback)."""
from functools import partialmethod

import torch
import torch.nn as nn
from torch.autograd import Function

from .. import ops
from ..ops import BF16, gemm, rows_plain
from . import functional as F_
from .triangle import (PairTransition, RowLayerNormFn, TriangleAttentionEndingNode, TriangleAttentionStartingNode,
                       TriangleMultiplicationIncoming, TriangleMultiplicationOutgoing)


def _require_cuda(x):
    if not x.is_cuda:
        raise RuntimeError("dynamicpdb_amd pair-stack operators need an MI355X device tensor (no CPU fallback)")


class OuterGemmFn(Function):
    """a, b bf16 [S, N, C] (S a multiple of 8) -> bf16 [N, N, C*C]:  outer[i, j, c*C + e] = sum_s a[s,i,c] b[s,j,e]
    (outer_product_mean.py:51-56).  One [N*C] x [N*C] x S product on the MFMA engine over the transposed operands; the
    backward needs the same product against the incoming gradient in its two orientations."""

    @staticmethod
    def forward(ctx, a, b):
        S, N, C = a.shape
        a2, b2 = a.contiguous().view(S, N * C), b.contiguous().view(S, N * C)
        aT, bT = ops.transpose_bf16(a2, S, N * C), ops.transpose_bf16(b2, S, N * C)          # [(i,c)][s], [(j,e)][s]
        o1 = torch.empty((N * C, N * C), dtype=BF16, device=a.device)
        gemm(aT, bT, o1, N * C, N * C, S, a_rows=rows_plain(S), c_rows=rows_plain(N * C), ldb=S)
        ctx.save_for_backward(a2, b2)
        ctx.dims = (S, N, C)
        return o1.view(N, C, N, C).permute(0, 2, 1, 3).contiguous().view(N, N, C * C)

    @staticmethod
    def backward(ctx, g):
        a2, b2 = ctx.saved_tensors
        S, N, C = ctx.dims
        g = (g if g.dtype == BF16 else ops.cast_bf16(g.contiguous())).view(N, N, C, C)
        g_ic = g.permute(0, 2, 1, 3).contiguous().view(N * C, N * C)          # [(i,c)][(j,e)]
        g_je = g.permute(1, 3, 0, 2).contiguous().view(N * C, N * C)          # [(j,e)][(i,c)]
        daT = torch.empty((N * C, S), dtype=BF16, device=g.device)            # da[(i,c)][s] = sum_(j,e) g b[s,(j,e)]
        dbT = torch.empty((N * C, S), dtype=BF16, device=g.device)
        gemm(g_ic, b2, daT, N * C, S, N * C, a_rows=rows_plain(N * C), c_rows=rows_plain(S), ldb=N * C)
        gemm(g_je, a2, dbT, N * C, S, N * C, a_rows=rows_plain(N * C), c_rows=rows_plain(S), ldb=N * C)
        da = ops.transpose_bf16(daT, N * C, S).view(S, N, C)
        db = ops.transpose_bf16(dbT, N * C, S).view(S, N, C)
        return da, db


class OuterProductMean(nn.Module):
    def __init__(self, c_m, c_z, c_hidden, eps=1e-3):
        super().__init__()
        if c_m % 8 or c_hidden % 8 or c_m > 512:
            raise ValueError("c_m, c_hidden must be multiples of 8 (c_m <= 512)")
        self.c_m, self.c_z, self.c_hidden, self.eps = c_m, c_z, c_hidden, eps
        self.layer_norm = nn.LayerNorm(c_m)
        self.linear_1 = nn.Linear(c_m, c_hidden)
        self.linear_2 = nn.Linear(c_m, c_hidden)
        self.linear_out = nn.Linear(c_hidden ** 2, c_z)

    def _one(self, m, mask):
        S, N, _ = m.shape
        S8 = (S + 7) // 8 * 8
        if S8 != S:          # the sequence axis becomes a GEMM K axis (16-byte bf16 chunks): pad it with masked-out rows
            m = torch.cat([m, m.new_zeros((S8 - S, N, self.c_m))], 0)
            mask = torch.cat([mask, mask.new_zeros((S8 - S, N))], 0)
        x = RowLayerNormFn.apply(m.reshape(-1, self.c_m), self.layer_norm.weight, self.layer_norm.bias, self.layer_norm.eps)
        mk = mask.to(BF16)[..., None]
        a = F_.linear(x, self.linear_1.weight, self.linear_1.bias).view(S8, N, self.c_hidden) * mk
        b = F_.linear(x, self.linear_2.weight, self.linear_2.bias).view(S8, N, self.c_hidden) * mk
        outer = OuterGemmFn.apply(a, b)                                                        # [N, N, C*C]
        y = F_.linear(outer, self.linear_out.weight, self.linear_out.bias, out_fp32=True)      # [N, N, c_z]
        # norm[i,j] = sum_s mask[s,i] mask[s,j] (:125): 0/1 values are exact in bf16, the sum in fp32
        mT = mask.to(BF16).t().contiguous()                                                    # [N][S8]
        norm = torch.empty((N, N), dtype=torch.float32, device=m.device)
        gemm(mT, mT, norm, N, N, S8, a_rows=rows_plain(S8), c_rows=rows_plain(N), ldb=S8)
        return y / (self.eps + norm)[..., None]

    def forward(self, m, mask=None, chunk_size=None):
        """m [*, N_seq, N_res, c_m], mask [*, N_seq, N_res] -> [*, N_res, N_res, c_z]; chunk_size is accepted and ignored."""
        _require_cuda(m)
        if mask is None:
            mask = m.new_ones(m.shape[:-1])
        m, mask = m.float(), mask.float()
        if m.dim() == 3:
            return self._one(m, mask)
        lead = m.shape[:-3]
        ms, ks = m.reshape((-1,) + m.shape[-3:]), mask.reshape((-1,) + mask.shape[-2:])
        y = torch.stack([self._one(ms[i], ks[i]) for i in range(ms.shape[0])])
        return y.reshape(lead + y.shape[1:])


class MSATransition(nn.Module):
    """m -> Linear_2(ReLU(Linear_1(LayerNorm(m)))) * mask (evoformer.py:41-117): the MSA-side twin of PairTransition."""

    def __init__(self, c_m, n):
        super().__init__()
        if c_m % 8 or c_m > 512:
            raise ValueError("c_m must be a multiple of 8 (<= 512)")
        self.c_m, self.n = c_m, n
        self.layer_norm = nn.LayerNorm(c_m)
        self.linear_1 = nn.Linear(c_m, n * c_m)
        self.linear_2 = nn.Linear(n * c_m, c_m)

    def forward(self, m, mask=None, chunk_size=None):
        _require_cuda(m)
        if mask is None:
            mask = m.new_ones(m.shape[:-1])
        x = RowLayerNormFn.apply(m.reshape(-1, self.c_m), self.layer_norm.weight, self.layer_norm.bias, self.layer_norm.eps)
        h = F_.linear(x, self.linear_1.weight, self.linear_1.bias, relu=True)
        y = F_.linear(h, self.linear_2.weight, self.linear_2.bias, out_fp32=True)
        return y.view(m.shape) * mask.unsqueeze(-1).to(y.dtype)


class Dropout(nn.Module):
    """Dropout whose keep-mask is shared along `batch_dim` (dropout.py:22-62); identity outside training mode.  The mask
    is drawn with torch's device generator (as the reference's nn.Dropout does), one draw per kept row / column."""

    def __init__(self, r, batch_dim):
        super().__init__()
        self.r = r
        self.batch_dim = [batch_dim] if isinstance(batch_dim, int) else batch_dim

    def forward(self, x):
        if not self.training or self.r == 0.0:
            return x
        shape = list(x.shape)
        for bd in self.batch_dim or []:
            shape[bd] = 1
        keep = (torch.rand(shape, device=x.device) >= self.r).to(x.dtype) * (1.0 / (1.0 - self.r))
        return x * keep


class DropoutRowwise(Dropout):
    __init__ = partialmethod(Dropout.__init__, batch_dim=-3)


class DropoutColumnwise(Dropout):
    __init__ = partialmethod(Dropout.__init__, batch_dim=-2)


class EvoformerBlockCore(nn.Module):
    """MSA transition, outer product mean and the five pair updates of one Evoformer block (evoformer.py:120-212)."""

    def __init__(self, c_m, c_z, c_hidden_opm, c_hidden_mul, c_hidden_pair_att, no_heads_msa, no_heads_pair, transition_n,
                 pair_dropout, inf, eps, _is_extra_msa_stack=False):
        super().__init__()
        self.msa_transition = MSATransition(c_m=c_m, n=transition_n)
        self.outer_product_mean = OuterProductMean(c_m, c_z, c_hidden_opm)
        self.tri_mul_out = TriangleMultiplicationOutgoing(c_z, c_hidden_mul)
        self.tri_mul_in = TriangleMultiplicationIncoming(c_z, c_hidden_mul)
        self.tri_att_start = TriangleAttentionStartingNode(c_z, c_hidden_pair_att, no_heads_pair, inf=inf)
        self.tri_att_end = TriangleAttentionEndingNode(c_z, c_hidden_pair_att, no_heads_pair, inf=inf)
        self.pair_transition = PairTransition(c_z, transition_n)
        self.ps_dropout_row_layer = DropoutRowwise(pair_dropout)
        self.ps_dropout_col_layer = DropoutColumnwise(pair_dropout)

    def forward(self, m, z, msa_mask, pair_mask, chunk_size=None, _mask_trans=True):
        _require_cuda(z)
        msa_trans_mask = msa_mask if _mask_trans else None
        pair_trans_mask = pair_mask if _mask_trans else None
        m = m + self.msa_transition(m, mask=msa_trans_mask, chunk_size=chunk_size)
        z = z + self.outer_product_mean(m, mask=msa_mask, chunk_size=chunk_size)
        z = z + self.ps_dropout_row_layer(self.tri_mul_out(z, mask=pair_mask))
        z = z + self.ps_dropout_row_layer(self.tri_mul_in(z, mask=pair_mask))
        z = z + self.ps_dropout_row_layer(self.tri_att_start(z, mask=pair_mask, chunk_size=chunk_size))
        z = z + self.ps_dropout_col_layer(self.tri_att_end(z, mask=pair_mask, chunk_size=chunk_size))
        z = z + self.pair_transition(z, mask=pair_trans_mask, chunk_size=chunk_size)
        return m, z
