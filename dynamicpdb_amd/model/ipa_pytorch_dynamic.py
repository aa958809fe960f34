"""Drop-in operators for the reference's src/model/ipa_pytorch_dynamic.py: same class names, constructor
arguments, forward signatures, output keys and state_dict key names / shapes -- backed by the MI355X engine
(libdfold_hip.so).  Every module also accepts a leading window (batch) axis, which the reference lacks
(its per-rank batch is one window, train_DFOLD_dynamics.py:551,680-684); statistics stay per window.

Operators: InvariantPointAttention (:242-516), ConvNet (:664-706), BackboneUpdate (:575-602),
MyLayerNorm (:709-724), AngleResnet (openfold/model/structure_module.py:47-158), DFOLDIpaScore (:726-907).
"""
import math
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as Fn

from .. import _lib, ops
from ..ops import BF16
from ..rigid import Rigid
from . import functional as F_
from . import geometry as G


def stamp_t_host(feats):
    """record which tensor (identity by weak reference, and version counter) feats['t_host'] is the host copy of"""
    import weakref
    t = feats['t']
    feats['t_host_stamp'] = (weakref.ref(t), t._version)
    return feats


def t_host_valid(feats):
    """feats['t_host'] may stand in for feats['t'] only if 't' is still the very tensor, unwritten, it was copied from"""
    st = feats.get('t_host_stamp')
    t = feats.get('t')
    return feats.get('t_host') is not None and st is not None and torch.is_tensor(t) and st[0]() is t and st[1] == t._version


def _require_cuda(t):
    if not t.is_cuda:
        raise RuntimeError("dynamicpdb_amd operators run on an MI355X device tensor (no CPU fallback); "
                           "the CPU restatement lives in oracle/ and is test-only")


_TRUNC_UNIT_STD = None      # standard deviation of a unit normal truncated at +-2 (scipy, computed once)


class Linear(nn.Linear):
    """nn.Linear with the reference's initialisers (src/model/ipa_pytorch_dynamic.py:107-172 and the identical
    openfold/model/primitives.py:115-175 that AngleResnet uses): zero bias; weight by `init` --
      "default": fan-in scaled normal truncated at two standard deviations (variance 1 / fan_in after truncation),
      "relu": the same with variance 2 / fan_in, "final": zeros.
    A fresh drop-in model therefore starts from the reference's distribution -- and from the reference's very numbers under
    the same seeds: like the reference the constructor first runs nn.Linear's own initialisation (it consumes torch's
    generator) and then draws the truncated normals with scipy from numpy's global generator, in the reference's module
    order (tests/test_reference_experiment.py::test_fresh_model_equals_the_reference_fresh_model)."""

    def __init__(self, in_dim, out_dim, bias=True, init="default"):
        super().__init__(in_dim, out_dim, bias=bias)
        with torch.no_grad():
            if bias:
                self.bias.zero_()
            if init == "final":
                self.weight.zero_()
            elif init in ("default", "relu"):
                self.weight.copy_(self._truncated_normal(self.weight.shape, 2.0 if init == "relu" else 1.0))
            else:
                raise ValueError(f"dynamicpdb_amd Linear: init must be 'default', 'relu' or 'final' (the ones the DFOLDv2 path "
                                 f"uses), got {init!r}")

    @staticmethod
    def _truncated_normal(shape, gain):
        global _TRUNC_UNIT_STD
        from scipy.stats import truncnorm
        if _TRUNC_UNIT_STD is None:
            _TRUNC_UNIT_STD = float(truncnorm.std(a=-2, b=2, loc=0, scale=1))
        sigma = math.sqrt(gain / max(1, shape[1])) / _TRUNC_UNIT_STD
        draws = truncnorm.rvs(a=-2, b=2, loc=0, scale=sigma, size=int(np.prod(shape)))
        return torch.tensor(draws.reshape(tuple(shape)))


class MyLayerNorm(nn.Module):
    """(x - mean_all) / sqrt(var_all_unbiased + 1e-4) over one window's whole [F,N,C] tensor."""

    def __init__(self):
        super().__init__()
        self.eps = 1e-4

    def forward(self, x):
        _require_cuda(x)
        batched = x.dim() == 4
        xb = x if batched else x[None]
        eye = torch.eye(xb.shape[-1], device=x.device)
        zero = torch.zeros(xb.shape[-1], device=x.device)
        y = F_.linear_gln(xb.to(BF16), eye, zero, False)   # standalone use: identity projection + fused norm
        y = y.to(x.dtype)
        return y if batched else y[0]


class ConvNet(nn.Module):
    """4 x [x = ReLU(conv5x5(ReLU(conv5x5(x)))) + x] over the frame x residue grid (1280 <-> 640 channels)."""

    def __init__(self, dim):
        super().__init__()
        for i in (1, 2, 3, 4):
            setattr(self, f"conv{i}", nn.Sequential(
                nn.Conv2d(dim, dim // 2, kernel_size=5, stride=1, padding=2, bias=True), nn.ReLU(),
                nn.Conv2d(dim // 2, dim, kernel_size=5, stride=1, padding=2, bias=True), nn.ReLU()))
        self._tower = None

    def _params(self):
        ws, bs = [], []
        for i in (1, 2, 3, 4):
            seq = getattr(self, f"conv{i}")
            ws += [seq[0].weight, seq[2].weight]
            bs += [seq[0].bias, seq[2].bias]
        return ws, bs

    def tower(self):
        ws, bs = self._params()
        if self._tower is None or self._tower.weights[0].data_ptr() != ws[0].data_ptr():
            self._tower = ops.ConvTower(ws, bs)
        return self._tower

    def run(self, x, last_frame_only=False, first_of_pass=True):
        """x bf16 [W,F,N,C], or a list of channel slices [W,F,N,C_k] that are concatenated inside the padded conv grid
        (no torch.cat) -> bf16 [W,F,N,C]; with last_frame_only (see functional.ConvTowerFn: the caller consumes the last frame
        alone) -> bf16 [W,1,N,C], that frame.
        first_of_pass: this is the first application of the shared tower in a forward pass of the enclosing model (the
        weight gradients of the applications of one pass are summed inside the tower and delivered together)."""
        xs = list(x) if isinstance(x, (list, tuple)) else [x]
        ws, bs = self._params()
        inter = [p for pair in zip(ws, bs) for p in pair]
        track = torch.is_grad_enabled() and (any(t.requires_grad for t in xs) or any(p.requires_grad for p in inter))
        flags = (1 if track else 0) | (2 if first_of_pass else 0)
        return F_.ConvTowerFn.apply(self.tower(), bool(last_frame_only), flags, len(xs), *xs, *inter)

    def forward(self, x):
        _require_cuda(x)
        batched = x.dim() == 4
        xb = x if batched else x[None]
        y = self.run(xb.to(BF16)).to(x.dtype)
        return y if batched else y[0]


class BackboneUpdate(nn.Module):
    def __init__(self, c_s):
        super().__init__()
        self.c_s = c_s
        self.linear = Linear(c_s, 6, init="final")                       # :590

    def forward(self, s):
        _require_cuda(s)
        return F_.linear(s.to(BF16), self.linear.weight, self.linear.bias, out_fp32=True)


_relu = torch.relu


class AngleResnetBlock(nn.Module):
    def __init__(self, c_hidden):
        super().__init__()
        self.linear_1 = Linear(c_hidden, c_hidden, init="relu")          # openfold/model/structure_module.py:58-59
        self.linear_2 = Linear(c_hidden, c_hidden, init="final")


class AngleResnet(nn.Module):
    """openfold/model/structure_module.py:75-158"""

    def __init__(self, c_in, c_hidden, no_blocks, no_angles, epsilon):
        super().__init__()
        self.c_in, self.c_hidden, self.no_blocks, self.no_angles, self.eps = c_in, c_hidden, no_blocks, no_angles, epsilon
        self.linear_in = Linear(c_in, c_hidden)                          # structure_module.py:100-108
        self.linear_initial = Linear(c_in, c_hidden)
        self.layers = nn.ModuleList([AngleResnetBlock(c_hidden) for _ in range(no_blocks)])
        self.linear_out = Linear(c_hidden, no_angles * 2)

    def forward(self, s, s_initial):
        _require_cuda(s)
        s, s_initial = s.to(BF16), s_initial.to(BF16)
        l0, l1 = self.layers[0], self.layers[1]
        if (self.no_blocks == 2 and self.c_in % 8 == 0 and self.c_hidden % 8 == 0 and _ANGLE_FUSED):
            # one autograd node: ReLUs, residual adds and the ReLU backward ride in the GEMM epilogues (functional.AngleResnetFn)
            out = F_.AngleResnetFn.apply(s, s_initial, self.linear_in.weight, self.linear_in.bias, self.linear_initial.weight,
                                         self.linear_initial.bias, l0.linear_1.weight, l0.linear_1.bias, l0.linear_2.weight,
                                         l0.linear_2.bias, l1.linear_1.weight, l1.linear_1.bias, l1.linear_2.weight, l1.linear_2.bias,
                                         self.linear_out.weight, self.linear_out.bias)
        else:
            a = F_.linear(_relu(s), self.linear_in.weight, self.linear_in.bias) + \
                F_.linear(_relu(s_initial), self.linear_initial.weight, self.linear_initial.bias)
            for l in self.layers:
                h = F_.linear(_relu(a), l.linear_1.weight, l.linear_1.bias, relu=True)
                a = a + F_.linear(h, l.linear_2.weight, l.linear_2.bias)
            out = F_.linear(_relu(a), self.linear_out.weight, self.linear_out.bias, out_fp32=True)
        out = out.view(out.shape[:-1] + (-1, 2))
        unnorm = out
        denom = torch.sqrt(torch.clamp(torch.sum(out ** 2, dim=-1, keepdim=True), min=self.eps))
        return unnorm, out / denom


# DFOLD_ANGLE_FUSED=0: AngleResnet as the chain of single-layer nodes with aten ReLUs / adds of rounds 1-5
_ANGLE_FUSED = os.environ.get("DFOLD_ANGLE_FUSED", "1") != "0"


class InvariantPointAttention(nn.Module):
    def __init__(self, ipa_conf, inf: float = 1e5, eps: float = 1e-8):
        super().__init__()
        self._ipa_conf = ipa_conf
        self.c_s, self.c_z, self.c_hidden = ipa_conf.c_s, ipa_conf.c_z, ipa_conf.c_hidden
        self.no_heads, self.no_qk_points, self.no_v_points = ipa_conf.no_heads, ipa_conf.no_qk_points, ipa_conf.no_v_points
        self.inf, self.eps = inf, eps
        if self.no_qk_points != 8 or self.no_v_points != 12 or inf != 1e5:
            raise ValueError("dynamicpdb_amd.InvariantPointAttention supports the configuration of config/train_DFOLDv2.yaml "
                             "(no_qk_points=8, no_v_points=12, any c_s / c_z / c_hidden / no_heads that are multiples of 8) "
                             f"with inf=1e5; got no_qk_points={self.no_qk_points}, no_v_points={self.no_v_points}, inf={inf}: "
                             "the point tables of csrc/ipa_attn.hip are sized for 8 query / 12 value points")
        hc = self.c_hidden * self.no_heads
        self.linear_q = Linear(self.c_s, hc)                             # :284-311, the reference's order and initialisers
        self.linear_kv = Linear(self.c_s, 2 * hc)
        self.linear_q_points = Linear(self.c_s, self.no_heads * self.no_qk_points * 3)
        self.linear_kv_points = Linear(self.c_s, self.no_heads * (self.no_qk_points + self.no_v_points) * 3)
        self.linear_b = Linear(self.c_z, self.no_heads)
        self.down_z = Linear(self.c_z, self.c_z // 4)
        self.head_weights = nn.Parameter(torch.full((self.no_heads,), 0.541324854612918))
        concat_out_dim = self.c_z // 4 + self.c_hidden + self.no_v_points * 4
        self.linear_out = Linear(self.no_heads * (concat_out_dim + self.no_v_points * 4), self.c_s, init="final")
        self.linear_rbf = Linear(20, 1)         # unused in the reference forward (:311); kept for state_dict parity

    def features(self, s, z, t7, mask):
        """s bf16 [B,F,N,c_s], z bf16 [B,N,N,c_z], t7 fp32 [B,F,N,7], mask [B,F,N] -> bf16 [B,F,N,H*(...)]  (:350-504).
        The attention products use the residue axis as a GEMM K axis (16-byte bf16 rows): when N_res is not a multiple
        of 8 the inputs are padded with masked-out residues (identity frames, zero features, mask 0 -> exactly zero
        attention weight as keys, their own rows discarded), which leaves every real row unchanged."""
        N = s.shape[2]
        pad = (-N) % 8
        if pad == 0:
            return self._features(s, z, t7, mask)
        ident = t7.new_zeros(t7.shape[:2] + (pad, 7))
        ident[..., 0] = 1.0
        out = self._features(Fn.pad(s, (0, 0, 0, pad)), Fn.pad(z, (0, 0, 0, pad, 0, pad)), torch.cat([t7, ident], 2),
                             Fn.pad(mask, (0, pad)))
        return out[:, :, :N]

    def _features(self, s, z, t7, mask):
        B, Fr, N, _ = s.shape
        H, PQ, PV = self.no_heads, self.no_qk_points, self.no_v_points
        q = F_.linear(s, self.linear_q.weight, self.linear_q.bias)
        kv = F_.linear(s, self.linear_kv.weight, self.linear_kv.bias)
        qp = F_.linear(s, self.linear_q_points.weight, self.linear_q_points.bias, out_fp32=True)
        kvp = F_.linear(s, self.linear_kv_points.weight, self.linear_kv_points.bias, out_fp32=True)
        q_pts, k_pts, v_pts = F_.IpaPointsFn.apply(qp, kvp, t7)                       # global frame (:363-390)
        hw = Fn.softplus(self.head_weights) * math.sqrt(1.0 / (3 * (PQ * 9.0 / 2)))
        if F_.ipa_feat_direct_ok(N, self.c_hidden, q_pts, v_pts, self.no_heads, self.down_z.weight.shape[0]):
            # one node: attention core + output features, every block written straight into the operand of linear_out
            return F_.IpaFeatFn.apply(q, kv, q_pts, k_pts, v_pts, z, self.linear_b.weight, self.down_z.weight, self.down_z.bias,
                                      mask, hw, t7, self.eps)                             # :396-504
        o, o_pt_g, o_pair = F_.IpaCoreFn.apply(q, kv, q_pts, k_pts, v_pts, z, self.linear_b.weight, self.down_z.weight,
                                               self.down_z.bias, mask, hw)
        geo_l, geo_g = F_.IpaOutFeatFn.apply(o_pt_g, t7, self.eps)                       # :470-488
        return torch.cat([o, geo_l, o_pair, geo_g], -1)                                   # :504

    def forward(self, s, z, r, mask, _offload_inference=False, _z_reference_list=None):
        """Reference signature: s [*,N,c_s], z [N,N,c_z] (no frame axis), r Rigid [*,N], mask [*,N] -> [*,N,c_s]."""
        _require_cuda(s)
        t7 = r.to_tensor_7() if isinstance(r, Rigid) else r
        batched = s.dim() == 4
        if not batched:
            s, z, t7, mask = s[None], z[None], t7[None], mask[None]
        feats = self.features(s.to(BF16), z.to(BF16), t7.float(), mask.float())
        out = F_.linear(feats, self.linear_out.weight, self.linear_out.bias, out_fp32=True)
        return out if batched else out[0]


def _embedder(k, d):
    return nn.Sequential(nn.Linear(k, d), nn.SiLU(), nn.Linear(d, d), MyLayerNorm(), nn.SiLU())


class DFOLDIpaScore(nn.Module):
    def __init__(self, model_conf, diffuser):
        super().__init__()
        self._model_conf = model_conf
        ipa_conf = model_conf.ipa
        self._ipa_conf = ipa_conf
        self.diffuser = diffuser
        self.scale_pos = lambda x: x * ipa_conf.coordinate_scaling
        self.unscale_pos = lambda x: x / ipa_conf.coordinate_scaling
        self.trunk = nn.ModuleDict()
        for b in range(ipa_conf.num_blocks):
            self.trunk[f'ipa_{b}'] = InvariantPointAttention(ipa_conf)
            self.trunk[f'ln_{b}'] = MyLayerNorm()
            self.trunk[f'bb_update_{b}'] = BackboneUpdate(ipa_conf.c_s * 5)
        self.trunk['conv_0'] = ConvNet(ipa_conf.c_s * 5)
        self.angle_resnet = AngleResnet(c_in=ipa_conf.c_s * 5, c_hidden=ipa_conf.c_s * 5, no_blocks=2, no_angles=7,
                                        epsilon=1e-12)
        # Dead-code elimination inside the trunk (default on, DFOLD_TRUNK_DCE=0 / attribute False: every position, like the
        # reference's eager graph).  The node features of the INNER blocks (0 < b < num_blocks - 1) feed nothing but
        # bb_update_b, whose output on every frame but the last is multiplied by 0.0 (:858-869); only blocks 0 and
        # num_blocks - 1 reach the angle head on all frames (:871-873).  So for the inner blocks the shared conv tower is
        # evaluated on the dependency cone of the last frame alone: every output of the forward (all keys, all frames) and
        # every gradient is the one the all-positions evaluation gives (the conv results inside the cone are the same sums --
        # bit-identical unless a thin cone launch splits K --; outside it they had no consumer and exactly zero gradient).
        self.trunk_dce = os.environ.get("DFOLD_TRUNK_DCE", "1") != "0"
        d = model_conf.node_embed_size
        self.force_embeder = _embedder(3, d)
        self.vel_embeder = _embedder(3, d)
        self.index_embeder = _embedder(1, d)
        self.rigid_embeder = _embedder(7, d)
        self.angle_embeder = _embedder(14, d)

    def unscale_rigids(self, t7):
        return torch.cat([t7[..., :4], self.unscale_pos(t7[..., 4:])], -1)

    @staticmethod
    def _embed(seq, x):
        """Linear-SiLU-Linear-MyLayerNorm-SiLU (:757-796); x fp32 [B,F',N,k] -> bf16 [B,F',N,d]."""
        h = F_.EmbedInFn.apply(x, seq[0].weight, seq[0].bias)              # k <= 14: VALU kernel
        return F_.linear_gln(h, seq[2].weight, seq[2].bias, True)

    @staticmethod
    def _shift_last(x):
        """cat([x[:-1], x[-2:-1]]) along the frame axis (axis 1 of the batched layout) (:819,822,826,842)"""
        return torch.cat([x[:, :-1], x[:, -2:-1]], 1)

    def forward(self, init_node_embed, edge_embed, input_feats, drop_ref=False, last_frame_only=False):
        """Batched mirror of the reference forward (:798-907).  last_frame_only (engine extension, default off): the
        caller only consumes frame F-1 of the per-frame outputs -- true for the training step, whose live loss terms
        and frame updates read the last frame alone (:869, train_DFOLD_dynamics.py:1219-1340) -- so the shared conv
        tower evaluates only the dependency cone of that frame (4x less conv work at F=32; loss and gradients are
        unchanged, outputs of the other frames ('angles', 'rigid_update') are then NOT the reference's).  All per-window tensors in `input_feats` carry a
        leading window axis here ([B,F,N,..], node/edge repr [B,N,..], t [B]); FullScoreNetwork adds it for
        reference-shaped inputs.  init_node_embed / edge_embed are ignored exactly like the reference (:829-834)."""
        f32 = torch.float32
        node_mask = input_feats['res_mask'].to(f32)
        diffuse_mask = (1 - input_feats['fixed_mask'].to(f32)) * node_mask
        rigids_t = input_feats['rigids_t'].to(f32)
        rig0 = input_feats['rigids_0'].to(f32)
        B, Fr, N = node_mask.shape
        curr_rigids = self._shift_last(rig0)
        force_embed = self._embed(self.force_embeder, self._shift_last(input_feats['force'].to(f32)))
        vel_embed = self._embed(self.vel_embeder, self._shift_last(input_feats['vel'].to(f32)))
        idx = input_feats['seq_idx'][:, 0:1].unsqueeze(-1).to(f32)                         # [B,1,N,1]
        idx_embed = self._embed(self.index_embeder, idx)                                   # [B,1,N,d]
        node_embed = (idx_embed.float() + input_feats['expand_node_repr'].float()[:, None]).to(BF16)
        node_embed = node_embed.expand(B, Fr, N, node_embed.shape[-1]).contiguous()
        edge = input_feats['expand_edge_repr']                                             # bf16 [B,N,N,c_z]
        angle = input_feats['torsion_angles_sin_cos'].to(f32) * input_feats['torsion_angles_mask'].to(f32).unsqueeze(-1)
        angle = self._shift_last(angle).reshape(B, Fr, N, -1)
        angle_embed = self._embed(self.angle_embeder, angle)
        conv = self.trunk['conv_0']
        node_feat = init_node_feat = rigid_update = None
        for b in range(self._ipa_conf.num_blocks):
            rigids_embed = self._embed(self.rigid_embeder, curr_rigids)
            ipa = self.trunk[f'ipa_{b}']
            feats = ipa.features(node_embed, edge, curr_rigids, node_mask)
            ipa_embed = F_.linear_gln(feats, ipa.linear_out.weight, ipa.linear_out.bias, False)   # linear_out + ln_b
            # cat([rigids, ipa, force, vel, angle], -1) (:846) happens inside the padded conv grid
            inner_block = self.trunk_dce and 0 < b < self._ipa_conf.num_blocks - 1      # node_feat[:, :-1] has no consumer
            node_feat = conv.run([rigids_embed, ipa_embed, force_embed, vel_embed, angle_embed], last_frame_only or inner_block,
                                 first_of_pass=(b == 0))
            # The reference evaluates BackboneUpdate on every frame and multiplies all but the last by 0.0 (:869): those
            # products -- and their gradient, which is exactly zero -- are never anything but zero, so the head runs on the
            # last frame alone in both step modes (same outputs: zeros elsewhere; in the last-frame mode that frame is also
            # the only one the tower defines).  At config 3 the dead 65536 x 1280 x 8 dx product alone cost 0.18 ms per block.
            upd_last = self.trunk[f'bb_update_{b}'](node_feat[:, -1:])
            rigid_update = torch.cat([upd_last.new_zeros(B, Fr - 1, N, 6), upd_last], 1)              # :869
            curr_rigids = F_.compose_q_update_vec(curr_rigids, rigid_update, diffuse_mask[..., None])
            if b == 0:
                init_node_feat = node_feat
        if last_frame_only:
            un_l, an_l = self.angle_resnet(node_feat[:, -1:], init_node_feat[:, -1:])
            pad = un_l.new_zeros((B, Fr - 1) + tuple(un_l.shape[2:]))
            unorm_angles, angles = torch.cat([pad, un_l], 1), torch.cat([pad, an_l], 1)
        else:
            unorm_angles, angles = self.angle_resnet(node_feat, init_node_feat)
        t = input_feats['t'].reshape(B)
        # 't_host' is a second copy of the diffusion times; it is trusted only while it can still be the copy of THIS 't': it
        # must have one entry per window and have been set together with the tensor (experiment.set_t_feats / synthetic.device_batch
        # record the tensor's identity and version) -- a batch whose 't' was reassigned or written afterwards falls back to the
        # device -> host copy instead of a stale sigma
        t_host = input_feats.get('t_host') if t_host_valid(input_feats) and np.size(input_feats['t_host']) == B else None
        rot_score = self.diffuser.calc_rot_score_t7(rigids_t[..., :4], curr_rigids[..., :4], t,
                                                    t_host=t_host) * node_mask[..., None]
        curr_rigids = self.unscale_rigids(curr_rigids)
        trans_score = self.diffuser.calc_trans_score(rigids_t[..., 4:], curr_rigids[..., 4:], t[:, None, None, None],
                                                     use_torch=True) * node_mask[..., None]
        return {'angles': angles, 'unorm_angles': unorm_angles, 'rot_score': rot_score, 'trans_score': trans_score,
                'final_rigids': curr_rigids, 'rigid_update': rigid_update}
