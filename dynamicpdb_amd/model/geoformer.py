"""Drop-ins for the two pair-track operators of OmegaFold's GeoFormer block that feed the engine's embeddings
(`node_repr` / `edge_repr`; SURVEY 8f rank 3, offline `extract_embedding.py` path): `Node2Edge`
(src/toolbox/OmegaFold/omegafold/modules.py:320-351) and `GeometricAttention` (modules.py:568-723, with the stacked
two-axis `Attention` of modules.py:354-481).  Same class names, constructor arguments, parameter names and shapes (an
OmegaFold state_dict loads unchanged) and forward signatures; inference only, like the reference's use of them
(`torch.no_grad()`, fp32 -- SURVEY a17).

Nothing new runs on the device: both operators are re-parameterisations of kernels the engine already has.
  * `GeometricAttention` = triangle attention around the starting node + around the ending node (axis 0 / axis 1 of the
    stacked weights; modules.py:616-652) + two gated triangle multiplications, "outgoing" on the edge tensor and on its
    transpose (modules.py:654-689) -- four passes of the fused kernels of csrc/pair_fused.hip with the stacked weights
    sliced per axis.  `utils.normalize` is a LayerNorm without affine (torch_utils.py:53-83): unit gain, zero shift.  The
    per-head constant `linear_b_bias` shifts every logit of a softmax row by the same amount and drops out.
  * `Node2Edge` = outer product mean with the 3-index output weight read as a [f, d*e] matrix
    (`einsum('sid,def,sje->ijf')` == Linear(flatten(outer))), same 1e-3 regulariser.
Device tensors only (no CPU fallback)."""
import torch
import torch.nn as nn

from ..ops import BF16
from .pair_stack import OuterProductMean
from .triangle import _triatt_fused, _trimul_fused


def _require_inference(mod, *tensors):
    if not tensors[0].is_cuda:
        raise RuntimeError("dynamicpdb_amd GeoFormer operators need an MI355X device tensor (no CPU fallback)")
    if torch.is_grad_enabled() and (any(t.requires_grad for t in tensors) or any(p.requires_grad for p in mod.parameters())):
        raise RuntimeError(f"{type(mod).__name__} is an inference-only drop-in (the reference runs it under no_grad): "
                           "call it inside torch.no_grad()")


class Node2Edge(nn.Module):
    def __init__(self, in_dim, proj_dim, out_dim):
        super().__init__()
        if in_dim % 8 or proj_dim % 8 or in_dim > 512:
            raise ValueError("in_dim, proj_dim must be multiples of 8 (in_dim <= 512)")
        self.input_proj = nn.Linear(in_dim, proj_dim * 2)
        self.proj_dim = proj_dim
        self.out_weights = nn.Parameter(torch.empty(proj_dim, proj_dim, out_dim))
        self.out_bias = nn.Parameter(torch.empty(out_dim))
        nn.init.normal_(self.out_weights, std=proj_dim ** -1.0)
        nn.init.zeros_(self.out_bias)
        self._opm = None          # (stamp, OuterProductMean holding the re-read weights); not a registered submodule

    def _engine(self):
        ps = [self.input_proj.weight, self.input_proj.bias, self.out_weights, self.out_bias]
        stamp = tuple((p.data_ptr(), p._version) for p in ps)
        if self._opm is None or self._opm[0] != stamp:
            d, dev = self.proj_dim, self.out_weights.device
            opm = OuterProductMean(self.input_proj.in_features, self.out_weights.shape[-1], d, eps=1e-3).to(dev)
            with torch.no_grad():
                opm.layer_norm.weight.fill_(1.0)
                opm.layer_norm.bias.zero_()
                opm.linear_1.weight.copy_(self.input_proj.weight[:d])          # l, r = act.split(proj_dim) (:343)
                opm.linear_1.bias.copy_(self.input_proj.bias[:d])
                opm.linear_2.weight.copy_(self.input_proj.weight[d:])
                opm.linear_2.bias.copy_(self.input_proj.bias[d:])
                opm.linear_out.weight.copy_(self.out_weights.reshape(d * d, -1).t())   # [f, d*e]
                opm.linear_out.bias.copy_(self.out_bias)
            self._opm = (stamp, opm)
        return self._opm[1]

    def forward(self, node_repr, mask):
        """node_repr [*, S, N, in_dim], mask [*, S, N] -> [*, N, N, out_dim]"""
        _require_inference(self, node_repr)
        with torch.no_grad():
            return self._engine()(node_repr, mask=mask)


class Attention(nn.Module):
    """Parameter container of the stacked gated attention (modules.py:354-398); GeometricAttention reads it per axis."""

    def __init__(self, q_dim, kv_dim, n_head, gating, c, out_dim, n_axis):
        super().__init__()
        self.c, self.n_head, self.gating, self.q_dim, self.n_axis = c, n_head, gating, q_dim, n_axis
        self.qg_weights = nn.Parameter(torch.empty(q_dim, n_axis, n_head, (gating + 1) * c))
        self.kv_weights = nn.Parameter(torch.empty(kv_dim, n_axis, n_head, 2 * c))
        self.qg_bias = nn.Parameter(torch.empty(n_axis, n_head, 1, c * (1 + gating)))
        self.kv_bias = nn.Parameter(torch.empty(n_axis, n_head, 1, c * 2))
        self.o_weights = nn.Parameter(torch.empty(n_axis, n_head, c, out_dim))
        self.o_bias = nn.Parameter(torch.empty([out_dim, n_axis]))
        for p in (self.qg_weights, self.kv_weights, self.o_weights):
            nn.init.normal_(p, std=q_dim ** -0.5)
        for p in (self.qg_bias, self.kv_bias, self.o_bias):
            nn.init.zeros_(p)

    def forward(self, *a, **kw):
        raise NotImplementedError("the stacked Attention is evaluated by GeometricAttention (fused triangle-attention kernels)")


class GeometricAttention(nn.Module):
    def __init__(self, d_edge, c, n_head, n_axis):
        super().__init__()
        if (d_edge, c, n_head, n_axis) != (128, 32, 4, 2):
            raise ValueError("the fused pair kernels are built for d_edge 128, 4 heads x 32, 2 axes (OmegaFold's GeoFormer)")
        self.d_edge, self.n_axis, self.n_head = d_edge, n_axis, n_head
        self.linear_b_weights = nn.Parameter(torch.empty([d_edge, n_axis, n_head]))
        self.linear_b_bias = nn.Parameter(torch.empty([n_axis, n_head, 1, 1]))
        self.act_w = nn.Parameter(torch.empty([d_edge, n_axis, d_edge * 5]))
        self.act_b = nn.Parameter(torch.empty([n_axis, d_edge * 5]))
        self.out_proj_w = nn.Parameter(torch.empty([n_axis, d_edge, d_edge]))
        self.out_proj_b = nn.Parameter(torch.empty([n_axis, d_edge]))
        self.attention = Attention(q_dim=d_edge, kv_dim=d_edge, n_head=n_head, c=c, gating=True, out_dim=d_edge, n_axis=n_axis)
        for p in (self.linear_b_weights, self.act_w, self.out_proj_w):
            nn.init.normal_(p, std=d_edge ** -0.5)
        for p in (self.linear_b_bias, self.act_b, self.out_proj_b):
            nn.init.zeros_(p)
        self._pack = None
        self._ws = [{} for _ in range(4)]

    def _packed(self):
        """per axis r: the (wcat, bcat, ...) tuples of triangle._triatt_fused / _trimul_fused, cut out of the stacked
        OmegaFold parameters; rebuilt when a parameter changes"""
        ps = list(self.parameters())
        stamp = tuple((p.data_ptr(), p._version) for p in ps)
        if self._pack is None or self._pack[0] != stamp:
            d, c, H = self.d_edge, self.attention.c, self.n_head
            at = self.attention
            dev = self.act_w.device
            ones, zeros = torch.ones(d, device=dev), torch.zeros(d, device=dev)
            f32 = lambda t: t.detach().float().contiguous()
            att, mul = [], []
            with torch.no_grad():
                for r in range(2):
                    # Linear-style [out = h*c + cc, in] blocks of q | k | v | gate (split order modules.py:444-452,473-475)
                    wq = at.qg_weights[:, r, :, :c].permute(1, 2, 0).reshape(H * c, d)
                    wg = at.qg_weights[:, r, :, c:].permute(1, 2, 0).reshape(H * c, d)
                    wk = at.kv_weights[:, r, :, :c].permute(1, 2, 0).reshape(H * c, d)
                    wv = at.kv_weights[:, r, :, c:].permute(1, 2, 0).reshape(H * c, d)
                    bq, bg = at.qg_bias[r, :, 0, :c].reshape(-1), at.qg_bias[r, :, 0, c:].reshape(-1)
                    bk, bv = at.kv_bias[r, :, 0, :c].reshape(-1), at.kv_bias[r, :, 0, c:].reshape(-1)
                    wo = at.o_weights[r].reshape(H * c, d).t()                       # [o, h*c + cc]  (:429)
                    att.append((torch.cat([wq, wk, wv, wg], 0).to(BF16).contiguous(), f32(torch.cat([bq, bk, bv, bg])),
                                wo.to(BF16).contiguous(), ones, zeros, f32(self.linear_b_weights[:, r, :].t()),
                                f32(at.o_bias[:, r])))
                    # act_w[:, r] = [row_p | col_p | row_g | col_g | gate] blocks of d columns (_get_sliced_weight :691-695)
                    blk = lambda t, k: t[..., k * d:(k + 1) * d]
                    aw, ab = self.act_w[:, r], self.act_b[r]
                    order = (0, 2, 1, 3, 4)                                         # a_p, a_g, b_p, b_g, g
                    wcat = torch.cat([blk(aw, k).t() for k in order], 0).to(BF16).contiguous()
                    bcat = f32(torch.cat([blk(ab, k) for k in order]))
                    mul.append((wcat, bcat, self.out_proj_w[r].t().to(BF16).contiguous(), ones, zeros, ones, zeros,
                                f32(self.out_proj_b[r])))
            self._pack = (stamp, att, mul)
        return self._pack[1], self._pack[2]

    def forward(self, edge_repr, mask, fwd_cfg=None):
        """edge_repr [N, N, d_edge] (or [B, N, N, d_edge]), mask [N] (or [B, N]) -> same shape as edge_repr;
        fwd_cfg (sub-batching hints of the reference) is accepted and ignored."""
        _require_inference(self, edge_repr)
        with torch.no_grad():
            single = edge_repr.dim() == 3
            e = (edge_repr[None] if single else edge_repr).float().contiguous()
            m = (mask[None] if single else mask).float()
            B, N = e.shape[0], e.shape[1]
            m_row = m[:, :, None].expand(B, N, N).contiguous()      # mask of the first index
            m_one = torch.ones_like(m_row)
            att, mul = self._packed()
            ws = self._ws
            # axis 0: rows attend along their own row (:616-650); axis 1: the same on the transpose, transposed back (:652)
            # == attention around the ending node.  No key is masked: the reference overwrites the mask bias it starts
            # `b` from when it assigns the pair bias into it (:627 vs :645-647), and a drop-in reproduces that.
            out = _triatt_fused(e, m_one, True, 1e9, att[0], ws[0])[0]
            out = out + _triatt_fused(e, m_one, False, 1e9, att[1], ws[1])[0]
            # gated products (:654-689): act_row[i,k] m_i, act_col[j,k] m_j, sum over k -- "outgoing" on e and on e^T;
            # the axis-1 term is NOT transposed back (gated.sum(-2), :689)
            out = out + _trimul_fused(e, m_row, True, mul[0], ws[2])[0]
            eT = e.transpose(1, 2).contiguous()
            out = out + _trimul_fused(eT, m_row, True, mul[1], ws[3])[0]
            out = out.to(edge_repr.dtype)
            return out[0] if single else out
