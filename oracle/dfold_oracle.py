"""CPU ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A from-scratch functional restatement (plain torch on CPU, dtype-parametric so it
can also run in float64) of the reference's DFOLDv2 hot path.  It is the checker
for the HIP path: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import it.  The product path (dynamicpdb_amd/) never does.

Parity status: PINNED against the reference's own code executed in the build
container -- tests/golden/*.npz hold outputs of the reference modules imported from
/root/reference (oracle/ref_harness), minted by tests/golden/make_golden.py, and
tests/test_oracle_golden.py compares every function here with them.
The reference ships no tests / golden vectors of its own.

Every function cites the reference file:line it restates (paths relative to
/root/reference).  State is a flat dict P of tensors keyed by the reference's
state_dict names.
"""
import math
import os

import numpy as np
import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------
# quaternion / rigid algebra  (openfold/utils/rigid_utils.py)
# ----------------------------------------------------------------------------


def quat_to_rotmat(q):
    """openfold/utils/rigid_utils.py:185-205 (quadratic form, NO normalisation)."""
    a, b, c, d = q.unbind(-1)
    rows = [
        a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c),
        2 * (b * c + a * d), a * a - b * b + c * c - d * d, 2 * (c * d - a * b),
        2 * (b * d - a * c), 2 * (c * d + a * b), a * a - b * b - c * c + d * d,
    ]
    return torch.stack(rows, -1).reshape(q.shape[:-1] + (3, 3))


def quat_mul(p, q):
    """Hamilton product, openfold/utils/rigid_utils.py:230-263."""
    a1, b1, c1, d1 = p.unbind(-1)
    a2, b2, c2, d2 = q.unbind(-1)
    return torch.stack([
        a1 * a2 - b1 * b2 - c1 * c2 - d1 * d2,
        a1 * b2 + b1 * a2 + c1 * d2 - d1 * c2,
        a1 * c2 - b1 * d2 + c1 * a2 + d1 * b2,
        a1 * d2 + b1 * c2 - c1 * b2 + d1 * a2,
    ], -1)


def quat_mul_vec(q, v):
    """q (x) (0, v): openfold/utils/rigid_utils.py:266-275."""
    zero = torch.zeros_like(v[..., :1])
    return quat_mul(q, torch.cat([zero, v], -1))


def quat_invert(q):
    """openfold/utils/rigid_utils.py:282-286."""
    conj = q * q.new_tensor([1.0, -1.0, -1.0, -1.0])
    return conj / (q * q).sum(-1, keepdim=True)


def rot_apply(R, x):
    """openfold/utils/rigid_utils.py:82-106."""
    return (R * x[..., None, :]).sum(-1)


def rigid_apply(t7, pts):
    """Rigid.apply with quaternion rotation: rigid_utils.py:1104-1116."""
    R = quat_to_rotmat(t7[..., :4])
    return rot_apply(R, pts) + t7[..., 4:]


def rigid_invert_apply(t7, pts):
    """Rigid.invert_apply: rigid_utils.py:1118-1130."""
    R = quat_to_rotmat(t7[..., :4])
    return rot_apply(R.transpose(-1, -2), pts - t7[..., 4:])


def compose_q_update_vec(t7, upd6, mask):
    """Rigid.compose_q_update_vec rigid_utils.py:1039-1063 (+ Rotation :587-616,
    normalisation in Rotation.__init__ :331-332).  mask broadcasts as [...,1]."""
    q, t = t7[..., :4], t7[..., 4:]
    dq = quat_mul_vec(q, upd6[..., :3]) * mask
    qn = q + dq
    qn = qn / torch.linalg.norm(qn, dim=-1, keepdim=True)
    dt = rot_apply(quat_to_rotmat(q), upd6[..., 3:]) * mask
    return torch.cat([qn, t + dt], -1)


def quat_to_rotvec(quat, eps=1e-6):
    """src/data/utils.py:589-606."""
    flip = (quat[..., :1] < 0).to(quat.dtype)
    quat = quat * (1 - 2 * flip)
    angle = 2 * torch.atan2(torch.linalg.norm(quat[..., 1:], dim=-1), quat[..., 0])
    a2 = angle * angle
    small = 2 + a2 / 12 + 7 * a2 * a2 / 2880
    large = angle / torch.sin(angle / 2 + eps)
    is_small = (angle <= 1e-3).to(quat.dtype)
    scale = small * is_small + (1 - is_small) * large
    return scale[..., None] * quat[..., 1:]


# ----------------------------------------------------------------------------
# small layers
# ----------------------------------------------------------------------------


# Optional emulation of the engine's storage precision (bf16 operands into every dense contraction, fp32
# accumulate).  Used by the GPU gradient-parity test to separate kernel defects from the ReLU-mask flips that any
# reduced-precision activation storage causes ("gradient parity note").  Off = the reference's fp32 math.
EMULATE_BF16_OPERANDS = False


def _q(x):
    """round-trip through bf16 with a straight-through gradient"""
    if not EMULATE_BF16_OPERANDS:
        return x
    return x + (x.detach().to(torch.bfloat16).to(x.dtype) - x.detach())


class _GradRound(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


def _qg(x):
    """identity whose GRADIENT is rounded to bf16: the engine stores activation gradients in bf16 between kernels
    (conv tower dgrad outputs, dx of every dense layer, the incoming gradient of every dense layer's output)."""
    if not EMULATE_BF16_OPERANDS or not x.requires_grad:
        return x
    return _GradRound.apply(x)


def _qs(x):
    """a bf16 storage point of the engine: value rounded on the way forward, gradient rounded on the way back"""
    return _q(_qg(x))


# Optional ReLU-mask feed (GPU gradient-parity test): a list of 0/1 tensors, consumed in call order by every ReLU of the
# path (conv tower: inner / outer activation of each residual pair; AngleResnet).  With a feed, relu(x) = x * mask --
# the engine's own stored masks -- so that a pre-activation within bf16 rounding of zero takes the same branch on both
# sides and the comparison measures the kernels, not which side of zero a rounding fell.  None = plain ReLU.
RELU_MASK_FEED = None


def _relu(x):
    if RELU_MASK_FEED is None:
        return F.relu(x)
    m = RELU_MASK_FEED.pop(0)
    return x * m.to(x.dtype).reshape(x.shape)


def linear(P, name, x, quant=True):
    """quant=False: the engine evaluates this layer on fp32 operands (the k <= 14 wide embedder inputs, csrc/embed.hip);
    only matters under EMULATE_BF16_OPERANDS."""
    b = P.get(name + ".bias")
    w = P[name + ".weight"].to(x.dtype)
    y = F.linear(_qs(x), _q(w), None if b is None else b.to(x.dtype)) if quant else F.linear(x, w, None if b is None else b.to(x.dtype))
    return _qg(y)


def my_layer_norm(x, eps=1e-4):
    """MyLayerNorm src/model/ipa_pytorch_dynamic.py:709-724: statistics over the
    WHOLE [F,N,C] tensor, unbiased variance, eps inside the sqrt, no affine."""
    mean = x.mean()
    var = x.var(unbiased=True)
    return (x - mean) / torch.sqrt(var + eps)


def embedder(P, name, x):
    """Linear-SiLU-Linear-MyLayerNorm-SiLU, ipa_pytorch_dynamic.py:757-796."""
    h = F.silu(linear(P, name + ".0", x, quant=False))
    h = linear(P, name + ".2", h)
    return F.silu(my_layer_norm(h))


def convnet(P, name, x):
    """ConvNet ipa_pytorch_dynamic.py:664-706 on x[F,N,C] (channels-last view of
    the reference's [1,C,F,N]); 4 residual pairs of 5x5 convs, zero pad 2."""
    h = x.permute(2, 0, 1).unsqueeze(0)
    for i in (1, 2, 3, 4):
        w0, b0 = P[f"{name}.conv{i}.0.weight"].to(x.dtype), P[f"{name}.conv{i}.0.bias"].to(x.dtype)
        w2, b2 = P[f"{name}.conv{i}.2.weight"].to(x.dtype), P[f"{name}.conv{i}.2.bias"].to(x.dtype)
        h = _qs(h)          # block input: a stored bf16 grid; its gradient (conv path + skip path) is stored in bf16 too
        y = _relu(_qg(F.conv2d(h, _q(w0), b0, padding=2)))
        y = _relu(_qg(F.conv2d(_q(y), _q(w2), b2, padding=2)))
        h = y + h
    return _q(h).squeeze(0).permute(1, 2, 0)


def angle_resnet(P, name, s, s_initial, eps=1e-12):
    """AngleResnet openfold/model/structure_module.py:114-158 (2 blocks, 7 angles)."""
    a = linear(P, name + ".linear_in", _relu(s)) + linear(P, name + ".linear_initial", _relu(s_initial))
    for l in (0, 1):
        h = linear(P, f"{name}.layers.{l}.linear_1", _relu(a))
        h = linear(P, f"{name}.layers.{l}.linear_2", _relu(h))
        a = a + h
    out = linear(P, name + ".linear_out", _relu(a))
    out = out.reshape(out.shape[:-1] + (-1, 2))
    denom = torch.sqrt(torch.clamp((out * out).sum(-1, keepdim=True), min=eps))
    return out, out / denom


# ----------------------------------------------------------------------------
# Invariant point attention  (src/model/ipa_pytorch_dynamic.py:242-516)
# ----------------------------------------------------------------------------

IPA_H, IPA_C, IPA_PQ, IPA_PV = 8, 256, 8, 12


def ipa(P, name, s, z, t7, mask, inf=1e5, eps=1e-8, return_attn=False):
    """s[F,N,c_s], z[N,N,c_z] (no frame axis; broadcast), t7[F,N,7], mask[F,N]."""
    H, C, PQ, PV = IPA_H, IPA_C, IPA_PQ, IPA_PV
    Fr, N = s.shape[0], s.shape[1]
    q = _q(linear(P, name + ".linear_q", s)).reshape(Fr, N, H, C)                   # :350-354
    kv = _q(linear(P, name + ".linear_kv", s)).reshape(Fr, N, H, 2 * C)             # :351-360
    k, v = kv[..., :C], kv[..., C:]

    def points(lin, npts):                                                          # :363-390
        raw = linear(P, name + lin, s)                     # [F,N,3*H*npts], xyz-major split
        xyz = torch.stack(torch.chunk(raw, 3, dim=-1), -1)  # [F,N,H*npts,3]
        glob = rigid_apply(t7[..., None, :], xyz)
        return glob.reshape(Fr, N, H, npts, 3)

    q_pts = points(".linear_q_points", PQ)
    kv_pts = points(".linear_kv_points", PQ + PV)
    k_pts, v_pts = kv_pts[..., :PQ, :], kv_pts[..., PQ:, :]

    b = linear(P, name + ".linear_b", z)                                            # :396  [N,N,H]
    a = torch.einsum("fihc,fjhc->fhij", q, k) * math.sqrt(1.0 / (3 * C))            # :402-406
    a = a + math.sqrt(1.0 / 3) * b.permute(2, 0, 1)                                 # :407
    d2 = ((q_pts[:, :, None] - k_pts[:, None, :]) ** 2).sum(-1)                     # :410-414 [F,i,j,H,PQ]
    hw = F.softplus(P[name + ".head_weights"].to(s.dtype)) * math.sqrt(1.0 / (3 * (PQ * 9.0 / 2)))  # :415-421
    pt = (d2 * hw[:, None]).sum(-1) * (-0.5)                                        # :422-424 [F,i,j,H]
    a = a + pt.permute(0, 3, 1, 2)
    sq_mask = inf * (mask[:, :, None] * mask[:, None, :] - 1)                       # :426-427
    a = torch.softmax(a + sq_mask[:, None], dim=-1)                                 # :443-444

    o = torch.einsum("fhij,fjhc->fihc", _q(a), v).reshape(Fr, N, H * C)             # :452-457
    o_pt_g = torch.einsum("fhij,fjhpx->fihpx", a, v_pts)                            # :460-469 (global frame)
    o_pt_l = rigid_invert_apply(t7[:, :, None, None, :], o_pt_g)                    # :481
    n_l = torch.sqrt((o_pt_l ** 2).sum(-1) + eps).reshape(Fr, N, H * PV)            # :484-486
    n_g = torch.sqrt((o_pt_g ** 2).sum(-1) + eps).reshape(Fr, N, H * PV)            # :487-488
    o_pt_l = o_pt_l.reshape(Fr, N, H * PV, 3)
    o_pt_g = o_pt_g.reshape(Fr, N, H * PV, 3)
    pair_z = linear(P, name + ".down_z", z)                                         # :498 [N,N,32]
    # (the engine adds down_z's bias after the aggregation -- rows of `a` sum to 1 -- so the bf16 operand is bias-free)
    bz = P[name + ".down_z.bias"].to(s.dtype)
    o_pair = (torch.einsum("fhij,ijc->fihc", _q(a), _q(pair_z - bz)) + bz * a.sum(-1).permute(0, 2, 1)[..., None]
              if EMULATE_BF16_OPERANDS else torch.einsum("fhij,ijc->fihc", a, pair_z)).reshape(Fr, N, -1)   # :499-502
    feats = [o, o_pt_l[..., 0], o_pt_l[..., 1], o_pt_l[..., 2], n_l, o_pair,
             o_pt_g[..., 0], o_pt_g[..., 1], o_pt_g[..., 2], n_g]                   # :504
    out = linear(P, name + ".linear_out", torch.cat(feats, -1))                     # :510-514
    if return_attn:
        return out, a
    return out


# ----------------------------------------------------------------------------
# IGSO(3) / VP-SDE score heads  (src/data/so3_diffuser.py, r3_diffuser.py)
# ----------------------------------------------------------------------------


class Schedules:
    """Scalar schedule functions of SO3Diffuser / R3Diffuser (so3_diffuser.py:
    176-213, r3_diffuser.py:26-43,159-167) with the yaml defaults."""

    def __init__(self, min_sigma=0.1, max_sigma=1.5, num_sigma=1000, min_b=0.1, max_b=20.0,
                 coordinate_scaling=1.0):
        self.min_sigma, self.max_sigma, self.num_sigma = min_sigma, max_sigma, num_sigma
        self.min_b, self.max_b, self.cs = min_b, max_b, coordinate_scaling
        self.discrete_sigma = self.sigma(np.linspace(0.0, 1.0, num_sigma))

    def sigma(self, t):
        return np.log(t * np.exp(self.max_sigma) + (1 - t) * np.exp(self.min_sigma))

    def t_to_idx(self, t):
        return np.digitize(self.sigma(t), self.discrete_sigma) - 1

    def marginal_b_t(self, t):
        return t * self.min_b + 0.5 * (t ** 2) * (self.max_b - self.min_b)


def igso3_score_torch(sched, vec, t, L=1000, eps=1e-6):
    """SO3Diffuser.torch_score so3_diffuser.py:274-305 with use_cached_score=False
    -> igso3_expansion :9-49 and score :71-117.  Reproduces the reference's mixed
    precision: omega and the trig terms stay in vec.dtype (fp32), the Gaussian
    envelope is float64 (sigma comes from numpy), the product promotes to float64."""
    t_np = np.asarray(t.detach().cpu().numpy())
    sigma = torch.tensor(sched.discrete_sigma[sched.t_to_idx(t_np)])      # float64 [1]
    omega = torch.linalg.norm(vec, dim=-1) + eps                           # [F,N]
    ls = torch.arange(L)
    om = omega[..., None]
    sg = sigma[:, None][..., None]                                         # [1,1,1]
    env = (2 * ls + 1) * torch.exp(-ls * (ls + 1) * sg ** 2 / 2)           # float64
    lo = torch.sin(om / 2)
    hi = torch.sin(om * (ls + 1 / 2))
    f = (env * hi / lo).sum(-1)
    dhi = (ls + 1 / 2) * torch.cos(om * (ls + 1 / 2))
    dlo = 1 / 2 * torch.cos(om / 2)
    dsig = (env * (lo * dhi - hi * dlo) / lo ** 2).sum(-1)
    sc = dsig / (f + 1e-4)
    return sc[..., None] * vec / (omega[..., None] + eps)


def calc_rot_score(sched, quats_t, quats_0, t):
    """SE3Diffuser.calc_rot_score src/data/se3_diffuser.py:119-125."""
    q0t = quat_mul(quat_invert(quats_0), quats_t)
    return igso3_score_torch(sched, quat_to_rotvec(q0t), t)


def calc_trans_score(sched, x_t, x_0, t):
    """R3Diffuser.score (use_torch, scale=True) r3_diffuser.py:169-177; t is a tensor
    broadcastable against [F,N,3]."""
    x_t, x_0 = x_t * sched.cs, x_0 * sched.cs
    bt = sched.marginal_b_t(t)
    return -(x_t - torch.exp(-0.5 * bt) * x_0) / (1 - torch.exp(-bt))


# ----------------------------------------------------------------------------
# frames -> atoms   (openfold/utils/feats.py:165-228, src/data/all_atom.py:114-154,
#                    src/model/Dfold_network_dynamic.py:574-594)
# ----------------------------------------------------------------------------

_TABLES = None


def residue_tables():
    global _TABLES
    if _TABLES is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dynamicpdb_amd",
                            "data", "residue_tables.npz")
        d = np.load(path)
        _TABLES = {k: torch.tensor(d[k]) for k in d.files}
    return _TABLES


def _compose(Ra, ta, Rb, tb):
    return Ra @ Rb, rot_apply(Ra, tb) + ta


def frames_to_atoms(t7, angles, aatype):
    """t7[F,N,7] (unit quats), angles[F,N,7,2] (sin,cos), aatype[F,N] int64 ->
    atom14[F,N,14,3], atom37[F,N,37,3]."""
    T = residue_tables()
    dt = t7.dtype
    d44 = T["default_frames"][aatype].to(dt)                         # [F,N,8,4,4]
    Rd, td = d44[..., :3, :3], d44[..., :3, 3]
    bb = torch.zeros(angles.shape[:-2] + (1, 2), dtype=dt)
    bb[..., 1] = 1
    al = torch.cat([bb, angles], -2)                                 # [F,N,8,2]
    Rt = torch.zeros(al.shape[:-1] + (3, 3), dtype=dt)
    Rt[..., 0, 0] = 1
    Rt[..., 1, 1] = al[..., 1]
    Rt[..., 1, 2] = -al[..., 0]
    Rt[..., 2, 1] = al[..., 0]
    Rt[..., 2, 2] = al[..., 1]
    Rf, tf = Rd @ Rt, td                                             # default_r.compose(all_rots) (trans None -> 0)
    R_l, t_l = [Rf[..., i, :, :] for i in range(8)], [tf[..., i, :] for i in range(8)]
    for i in (5, 6, 7):                                              # chi2..4 chained onto chi1
        R_l[i], t_l[i] = _compose(R_l[i - 1], t_l[i - 1], R_l[i], t_l[i])
    Rb, tb = torch.stack(R_l, -3), torch.stack(t_l, -2)              # [F,N,8,3,3],[F,N,8,3]
    Rg = quat_to_rotmat(t7[..., :4])[..., None, :, :]
    Rall, tall = Rg @ Rb, rot_apply(Rg, tb) + t7[..., None, 4:]
    grp = T["atom14_group"][aatype]                                  # [F,N,14]
    idx = grp[..., None, None].expand(grp.shape + (3, 3))
    Ra = torch.gather(Rall, -3, idx)
    ta = torch.gather(tall, -2, grp[..., None].expand(grp.shape + (3,)))
    pos = rot_apply(Ra, T["atom14_pos"][aatype].to(dt)) + ta
    atom14 = pos * T["atom14_mask"][aatype].to(dt)[..., None]
    i37 = T["atom37_to_atom14"][aatype]                              # [F,N,37]
    atom37 = torch.gather(atom14, -2, i37[..., None].expand(i37.shape + (3,)))
    atom37 = atom37 * T["atom37_mask"][aatype].to(dt)[..., None]
    return atom14, atom37


# ----------------------------------------------------------------------------
# DFOLDIpaScore + FullScoreNetwork
# ----------------------------------------------------------------------------


def _shift_last(x):
    """cat([x[:-1], x[-2:-1]]) -- history + copy of frame F-2 as the guess for F-1
    (ipa_pytorch_dynamic.py:819,822,826,842)."""
    return torch.cat([x[:-1], x[-2:-1]], 0)


def full_score_network(P, sched, feats, dtype=torch.float32, num_blocks=4, return_intermediates=False):
    """FullScoreNetwork.forward Dfold_network_dynamic.py:450-546 + DFOLDIpaScore.forward
    ipa_pytorch_dynamic.py:798-907 for ONE window (leading axis = frames).
    The dead DFOLDv2_Embeder branch (outputs unused) is not evaluated."""
    c = lambda k: feats[k].to(dtype)
    N = feats["node_repr"].shape[0]
    node_repr = linear(P, "expand_node", c("node_repr"))                                  # :473
    edge = linear(P, "expand_edge", c("edge_repr").reshape(N * N, -1)).reshape(N, N, -1)  # :474
    S = "score_model."
    node_mask = c("res_mask")
    diffuse_mask = (1 - c("fixed_mask")) * node_mask
    rig0 = c("rigids_0")
    Fr = rig0.shape[0]
    curr = _shift_last(rig0)
    force_e = embedder(P, S + "force_embeder", _shift_last(c("force")))
    vel_e = embedder(P, S + "vel_embeder", _shift_last(c("vel")))
    idx_e = embedder(P, S + "index_embeder", feats["seq_idx"][0:1].unsqueeze(-1).to(dtype))
    node_embed = idx_e.expand(Fr, -1, -1) + node_repr
    ang = c("torsion_angles_sin_cos") * c("torsion_angles_mask").unsqueeze(-1)
    ang_e = embedder(P, S + "angle_embeder", _shift_last(ang).reshape(Fr, N, 14))
    inter = {}
    node_feat = init_feat = upd = None
    for b in range(num_blocks):
        rig_e = embedder(P, S + "rigid_embeder", curr)
        ipa_e = my_layer_norm(ipa(P, f"{S}trunk.ipa_{b}", node_embed, edge, curr, node_mask))
        node_feat = convnet(P, S + "trunk.conv_0", torch.cat([rig_e, ipa_e, force_e, vel_e, ang_e], -1))
        upd = linear(P, f"{S}trunk.bb_update_{b}.linear", node_feat)
        upd = torch.cat([upd[:-1] * 0.0, upd[-1:]], 0)                                    # :869
        curr = compose_q_update_vec(curr, upd, diffuse_mask[..., None])
        if b == 0:
            init_feat = node_feat
        if return_intermediates:
            inter[f"ipa_ln_{b}"], inter[f"node_feat_{b}"], inter[f"rigids_{b}"] = ipa_e, node_feat, curr
    unorm, angles = angle_resnet(P, S + "angle_resnet", node_feat, init_feat)
    rig_t = c("rigids_t")
    rot_score = calc_rot_score(sched, rig_t[..., :4], curr[..., :4], feats["t"]) * node_mask[..., None]
    tt = feats["t"].to(dtype)[:, None, None]
    trans_score = calc_trans_score(sched, rig_t[..., 4:], curr[..., 4:], tt) * node_mask[..., None]
    gt = c("torsion_angles_sin_cos")
    fm = (1 - c("fixed_mask"))[..., None, None]
    angles_out = fm * angles + (1 - fm) * gt                                              # :515-519
    unorm_out = fm * unorm + (1 - fm) * gt
    atom14, atom37 = frames_to_atoms(curr, angles_out, feats["aatype"].long())
    out = dict(angles=angles_out, unorm_angles=unorm_out, rot_score=rot_score, trans_score=trans_score,
               rigids=curr, atom37=atom37, atom14=atom14, rigid_update=upd)
    if return_intermediates:
        out["_inter"] = inter
    return out


# ----------------------------------------------------------------------------
# training loss  (train_DFOLD_dynamics.py:1182-1400, openfold/utils/loss.py:52-76)
# ----------------------------------------------------------------------------


def torsion_angle_loss(a, a_gt, a_alt_gt, mask):
    """openfold/utils/loss.py:52-76 (this fork: +1e-8 in the normalisation, masked
    sum / (sum(mask)+1e-2), angle-norm weight 0.0) as used at train:1219-1224."""
    norm = torch.linalg.norm(a, dim=-1)
    a = a / (norm.unsqueeze(-1) + 1e-8)
    d_gt = ((a - a_gt) ** 2).sum(-1)
    d_alt = ((a - a_alt_gt) ** 2).sum(-1)
    m = torch.minimum(d_gt, d_alt)
    return (m * mask).sum(dim=(-1, -2)) / (mask.sum(dim=(-1, -2)) + 1e-2)


def loss_fn(out, batch, trans_w=100.0, rot_w=7.0, torsion_w=1.0, rot_t_threshold=0.0):
    """The three live terms of Experiment.loss_fn (last frame only, each repeated F
    times; gate trans_loss<100) -> (scalar loss, aux dict)."""
    dt = out["rigids"].dtype
    bb_mask = batch["res_mask"].to(dt)
    diffuse_mask = 1 - batch["fixed_mask"].to(dt)
    loss_mask = bb_mask * diffuse_mask
    Fr = bb_mask.shape[0]
    tl = torsion_angle_loss(out["angles"], batch["torsion_angles_sin_cos"].to(dt),
                            batch["alt_torsion_angles_sin_cos"].to(dt),
                            batch["torsion_angles_mask"].to(dt)) * torsion_w
    torsion = tl[-1:].repeat(Fr)
    gt_x0, pr_x0 = batch["rigids_0"][..., 4:].to(dt), out["rigids"][..., 4:]
    trans = ((gt_x0[-1:] - pr_x0[-1:]) ** 2).mean(dim=(-1, -2)).repeat(Fr) * trans_w
    pr_rot = out["rot_score"] * diffuse_mask[..., None]
    rot_mse = (batch["rot_score"] - pr_rot) ** 2 * loss_mask[..., None]
    rot = (rot_mse / batch["rot_score_scaling"][:, None, None] ** 2).sum(dim=(-1, -2)) / (loss_mask.sum(-1) + 1e-10)
    rot = rot * rot_w
    rot = rot * (batch["t"] > rot_t_threshold)
    rot = rot[-1:].repeat(Fr)
    gate = (trans < 100.0)
    rot = rot * gate.to(rot.dtype)
    trans = trans * gate.to(trans.dtype)
    torsion = torsion * (trans < 100.0).to(torsion.dtype)
    final = rot + trans + torsion
    bmask = torch.any(bb_mask > 0, dim=-1)
    norm = lambda x: x.sum() / (bmask.sum() + 1e-10)
    return norm(final), dict(rot_loss=norm(rot), trans_loss=norm(trans), torsion_loss=norm(torsion))


# ----------------------------------------------------------------------------
# triangle operators  (openfold/model/triangular_multiplicative_update.py:26-126,
#                      openfold/model/triangular_attention.py:31-139, primitives.py:219-448)
# ----------------------------------------------------------------------------


def _ln(P, name, x, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), P[name + ".weight"].to(x.dtype), P[name + ".bias"].to(x.dtype), eps)


def triangle_multiplication(P, z, mask=None, outgoing=True):
    """z[N,N,c_z] -> [N,N,c_z]; mask[N,N]."""
    if mask is None:
        mask = z.new_ones(z.shape[:-1])
    m = mask[..., None]
    zn = _ln(P, "layer_norm_in", z)
    a = linear(P, "linear_a_p", zn) * torch.sigmoid(linear(P, "linear_a_g", zn)) * m
    b = linear(P, "linear_b_p", zn) * torch.sigmoid(linear(P, "linear_b_g", zn)) * m
    x = torch.einsum("ikc,jkc->ijc", a, b) if outgoing else torch.einsum("kic,kjc->ijc", a, b)
    x = linear(P, "linear_z", _ln(P, "layer_norm_out", x))
    return x * torch.sigmoid(linear(P, "linear_g", zn))


def triangle_attention(P, x, mask=None, starting=True, no_heads=4, inf=1e9):
    """x[I,J,c] -> [I,J,c]; gated MHA over each row with pair bias."""
    if mask is None:
        mask = x.new_ones(x.shape[:-1])
    if not starting:
        x, mask = x.transpose(0, 1), mask.transpose(0, 1)
    I, J, _ = x.shape
    xn = _ln(P, "layer_norm", x)
    tri_bias = linear(P, "linear", xn).permute(2, 0, 1)                   # [H,I,J] -> bias over (q=I?) see below
    H = no_heads
    q = linear(P, "mha.linear_q", xn).reshape(I, J, H, -1)
    k = linear(P, "mha.linear_k", xn).reshape(I, J, H, -1)
    v = linear(P, "mha.linear_v", xn).reshape(I, J, H, -1)
    ch = q.shape[-1]
    q = q / math.sqrt(ch)
    # logits[i,h,q,k] = q[i,q,h]·k[i,k,h] + mask_bias[i,k] + tri_bias[h,q,k]
    a = torch.einsum("iqhc,ikhc->ihqk", q, k)
    a = a + (inf * (mask - 1))[:, None, None, :] + tri_bias[None]
    a = torch.softmax(a, -1)
    o = torch.einsum("ihqk,ikhc->iqhc", a, v)
    g = torch.sigmoid(linear(P, "mha.linear_g", xn)).reshape(I, J, H, -1)
    o = linear(P, "mha.linear_o", (o * g).reshape(I, J, -1))
    if not starting:
        o = o.transpose(0, 1)
    return o


# ----------------------------------------------------------------------------
# SE(3) noise / denoise with injected randomness
# (src/data/se3_diffuser.py:43-110,160-215; so3_diffuser.py:311-365; r3_diffuser.py:81-157)
# ----------------------------------------------------------------------------


def rotvec_to_rotmat(v):
    """Rodrigues; equals scipy Rotation.from_rotvec(v).as_matrix() (src/data/utils.py:191-192)."""
    v = np.asarray(v, np.float64)
    th = np.linalg.norm(v, axis=-1, keepdims=True)
    small = th < 1e-12
    k = v / np.where(small, 1.0, th)
    K = np.zeros(v.shape[:-1] + (3, 3))
    K[..., 0, 1], K[..., 0, 2] = -k[..., 2], k[..., 1]
    K[..., 1, 0], K[..., 1, 2] = k[..., 2], -k[..., 0]
    K[..., 2, 0], K[..., 2, 1] = -k[..., 1], k[..., 0]
    s, c = np.sin(th)[..., None], np.cos(th)[..., None]
    return np.eye(3) + s * K + (1 - c) * (K @ K)


def rotmat_to_rotvec(R):
    """Log map via quaternion (robust near pi); equals scipy as_rotvec up to fp rounding."""
    from scipy.spatial.transform import Rotation
    shp = R.shape[:-2]
    return Rotation.from_matrix(R.reshape(-1, 3, 3)).as_rotvec().reshape(shp + (3,))


def r3_reverse(sched, x_t, score_t, t, dt, z, center=True):
    """R3Diffuser.reverse r3_diffuser.py:106-157 with the Gaussian draw z injected."""
    x = x_t * sched.cs
    b_t = sched.min_b + t * (sched.max_b - sched.min_b)
    g = np.sqrt(b_t)
    f = -0.5 * b_t * x
    x1 = x - ((f - g ** 2 * score_t) * dt + g * np.sqrt(dt) * z)
    if center:
        x1 = x1 - x1.sum(-2, keepdims=True) / x1.shape[-2]
    return x1 / sched.cs


def so3_reverse(sched, rotvec_t, score_t, t, dt, z):
    """SO3Diffuser.reverse so3_diffuser.py:329-365 (geodesic random walk, right-multiply)."""
    sg = sched.sigma(t)
    g = np.sqrt(2 * (np.exp(sched.max_sigma) - np.exp(sched.min_sigma)) * sg / np.exp(sg))
    perturb = g ** 2 * score_t * dt + g * np.sqrt(dt) * z
    R = rotvec_to_rotmat(rotvec_t) @ rotvec_to_rotmat(perturb)
    return rotmat_to_rotvec(R)


# ----------------------------------------------------------------------------
# dataset-side geometry: atom37 coordinates -> rigid-group frames and torsion angles
# (reference openfold/data/data_transforms.py:755-893 atom37_to_frames, :923-1088 atom37_to_torsion_angles,
#  Rigid.from_3_points openfold/utils/rigid_utils.py:1233-1275; called per item by
#  src/data/Dfold_data_loader_dynamic.py:237-240 on float64 tensors)
# ----------------------------------------------------------------------------

def from_3_points(p_neg_x, origin, p_xy, eps=1e-8):
    """Gram-Schmidt frame (rigid_utils.py:1233-1275): returns (R [...,3,3] with columns e0,e1,e2, origin)."""
    e0 = origin - p_neg_x
    e1 = p_xy - origin
    e0 = e0 / torch.sqrt((e0 * e0).sum(-1, keepdim=True) + eps)
    e1 = e1 - e0 * (e0 * e1).sum(-1, keepdim=True)
    e1 = e1 / torch.sqrt((e1 * e1).sum(-1, keepdim=True) + eps)
    e2 = torch.cross(e0, e1, dim=-1)
    return torch.stack([e0, e1, e2], -1), origin


def _to_4x4(R, t):
    out = torch.zeros(R.shape[:-2] + (4, 4), dtype=R.dtype)
    out[..., :3, :3] = R
    out[..., :3, 3] = t
    out[..., 3, 3] = 1
    return out


def atom37_to_frames(aatype, pos, mask, eps=1e-8):
    """aatype [...,N] int64, pos [...,N,37,3], mask [...,N,37] -> dict with rigidgroups_gt_frames [...,N,8,4,4],
    rigidgroups_gt_exists, rigidgroups_group_exists, rigidgroups_group_is_ambiguous [...,N,8],
    rigidgroups_alt_gt_frames (data_transforms.py:755-893)."""
    T = residue_tables()
    idx = T["group_base_atom37"][aatype]                                          # [...,N,8,3]
    base = torch.gather(pos[..., None, :, :].expand(pos.shape[:-2] + (8, 37, 3)), -2,
                        idx[..., None].expand(idx.shape + (3,)))                  # [...,N,8,3(atoms),3]
    R, t = from_3_points(base[..., 0, :], base[..., 1, :], base[..., 2, :], eps)
    group_exists = T["group_mask"][aatype].to(mask.dtype)
    atoms_exist = torch.gather(mask[..., None, :].expand(mask.shape[:-1] + (8, 37)), -1, idx)
    gt_exists = atoms_exist.min(-1)[0] * group_exists
    flip = torch.ones(8, 3, dtype=pos.dtype)
    flip[0, 0] = -1
    flip[0, 2] = -1                                                               # group 0: diag(-1, 1, -1)
    R = R * flip[:, None, :]                                                      # right-multiply by the diagonal matrix
    amb = T["group_ambiguous"][aatype].to(pos.dtype)
    sgn = torch.ones(amb.shape + (3,), dtype=pos.dtype)
    sgn[..., 1] = 1 - 2 * amb
    sgn[..., 2] = 1 - 2 * amb                                                     # ambiguous groups: diag(1, -1, -1)
    R_alt = R * sgn[..., None, :]
    return {"rigidgroups_gt_frames": _to_4x4(R, t), "rigidgroups_gt_exists": gt_exists,
            "rigidgroups_group_exists": group_exists, "rigidgroups_group_is_ambiguous": amb.to(mask.dtype),
            "rigidgroups_alt_gt_frames": _to_4x4(R_alt, t)}


def atom37_to_torsion_angles(aatype, pos, mask):
    """-> torsion_angles_sin_cos [...,N,7,2], alt_torsion_angles_sin_cos, torsion_angles_mask [...,N,7]
    (data_transforms.py:923-1088): pre-omega, phi, psi, chi1-4; the previous residue along the chain supplies CA, C."""
    T = residue_tables()
    aatype = torch.clamp(aatype, max=20)
    prev_pos = torch.cat([torch.zeros_like(pos[..., :1, :, :]), pos[..., :-1, :, :]], -3)
    prev_mask = torch.cat([torch.zeros_like(mask[..., :1, :]), mask[..., :-1, :]], -2)
    pre_omega = torch.cat([prev_pos[..., 1:3, :], pos[..., :2, :]], -2)
    phi = torch.cat([prev_pos[..., 2:3, :], pos[..., :3, :]], -2)
    psi = torch.cat([pos[..., :3, :], pos[..., 4:5, :]], -2)
    m_omega = prev_mask[..., 1:3].prod(-1) * mask[..., :2].prod(-1)
    m_phi = prev_mask[..., 2] * mask[..., :3].prod(-1)
    m_psi = mask[..., :3].prod(-1) * mask[..., 4]
    ci = T["chi_atom37"][aatype]                                                  # [...,N,4,4]
    chis = torch.gather(pos[..., None, :, :].expand(pos.shape[:-2] + (4, 37, 3)), -2, ci[..., None].expand(ci.shape + (3,)))
    chi_atoms_mask = torch.gather(mask[..., None, :].expand(mask.shape[:-1] + (4, 37)), -1, ci).prod(-1)
    m_chi = T["chi_mask"][aatype].to(mask.dtype) * chi_atoms_mask
    atoms = torch.cat([pre_omega[..., None, :, :], phi[..., None, :, :], psi[..., None, :, :], chis], -3)   # [...,N,7,4,3]
    tmask = torch.cat([m_omega[..., None], m_phi[..., None], m_psi[..., None], m_chi], -1)
    R, t = from_3_points(atoms[..., 1, :], atoms[..., 2, :], atoms[..., 0, :], 1e-8)
    rel = ((atoms[..., 3, :] - t)[..., None, :] * R.transpose(-1, -2)).sum(-1)    # R^T (x - t)
    sc = torch.stack([rel[..., 2], rel[..., 1]], -1)
    sc = sc / torch.sqrt((sc * sc).sum(-1, keepdim=True) + 1e-8)
    sc = sc * torch.tensor([1.0, 1.0, -1.0, 1.0, 1.0, 1.0, 1.0], dtype=pos.dtype)[:, None]
    amb = T["chi_pi_periodic"][aatype].to(pos.dtype)
    mirror = torch.cat([torch.ones(aatype.shape + (3,), dtype=pos.dtype), 1.0 - 2.0 * amb], -1)
    return {"torsion_angles_sin_cos": sc, "alt_torsion_angles_sin_cos": sc * mirror[..., None], "torsion_angles_mask": tmask}


# ----------------------------------------------------------------------------
# pair transition (openfold/model/pair_transition.py:24-99, Algorithm 15)
# ----------------------------------------------------------------------------

def pair_transition(P, z, mask=None):
    """z [*,N,N,c_z] -> update [*,N,N,c_z]: LayerNorm, Linear(c_z -> n c_z), ReLU, Linear(n c_z -> c_z), times mask."""
    if mask is None:
        mask = z.new_ones(z.shape[:-1])
    x = torch.nn.functional.layer_norm(z, (z.shape[-1],), P["layer_norm.weight"], P["layer_norm.bias"], 1e-5)
    h = torch.relu(linear(P, "linear_1", x))
    return linear(P, "linear_2", h) * mask[..., None]


def _sub(P, prefix):
    return {k[len(prefix) + 1:]: v for k, v in P.items() if k.startswith(prefix + ".")}


def msa_transition(P, m, mask=None):
    """Algorithm 9 (openfold/model/evoformer.py:41-117): Linear_2(ReLU(Linear_1(LayerNorm(m)))) * mask."""
    return pair_transition(P, m, mask)        # same arithmetic on the MSA tensor (its own parameter names coincide)


def outer_product_mean(P, m, mask=None, eps=1e-3):
    """Algorithm 10 (openfold/model/outer_product_mean.py:26-129): m [S,N,c_m], mask [S,N] -> [N,N,c_z]."""
    if mask is None:
        mask = m.new_ones(m.shape[:-1])
    x = torch.nn.functional.layer_norm(m, (m.shape[-1],), P["layer_norm.weight"], P["layer_norm.bias"], 1e-5)
    a = linear(P, "linear_1", x) * mask[..., None]                      # [S,N,C]
    b = linear(P, "linear_2", x) * mask[..., None]
    outer = torch.einsum("sic,sje->ijce", a, b)                          # :51-56 (a, b transposed to [N,S,C] there)
    outer = linear(P, "linear_out", outer.reshape(outer.shape[:2] + (-1,)))
    norm = torch.einsum("si,sj->ij", mask, mask)                         # :125
    return outer / (eps + norm)[..., None]


def evoformer_block_core(P, m, z, msa_mask, pair_mask):
    """EvoformerBlockCore.forward in eval mode (dropout = identity), openfold/model/evoformer.py:172-212."""
    m = m + msa_transition(_sub(P, "msa_transition"), m, msa_mask)
    z = z + outer_product_mean(_sub(P, "outer_product_mean"), m, msa_mask)
    z = z + triangle_multiplication(_sub(P, "tri_mul_out"), z, pair_mask, outgoing=True)
    z = z + triangle_multiplication(_sub(P, "tri_mul_in"), z, pair_mask, outgoing=False)
    z = z + triangle_attention(_sub(P, "tri_att_start"), z, pair_mask, starting=True)
    z = z + triangle_attention(_sub(P, "tri_att_end"), z, pair_mask, starting=False)
    z = z + pair_transition(_sub(P, "pair_transition"), z, pair_mask)
    return m, z


def _unit_norm(x):
    """utils.normalize (omegafold/utils/torch_utils.py:53-83): LayerNorm over the last axis, no gain / shift, eps 1e-5"""
    return torch.nn.functional.layer_norm(x, (x.shape[-1],), None, None, 1e-5)


def omegafold_node2edge(P, node_repr, mask):
    """Node2Edge.forward (src/toolbox/OmegaFold/omegafold/modules.py:336-351): node_repr [S,N,in], mask [S,N] ->
    [N,N,out].  l, r = halves of the masked input projection; out_ijf = sum_s sum_de l_sid W_def r_sje + bias, divided
    by (number of rows where both residues are present + 1e-3)."""
    x = _unit_norm(node_repr)
    act = (x @ P["input_proj.weight"].t() + P["input_proj.bias"]) * mask[..., None]
    d = P["out_weights"].shape[0]
    left, right = act[..., :d], act[..., d:]
    pair = torch.einsum("sid,sje->ijde", left, right)
    out = torch.einsum("ijde,def->ijf", pair, P["out_weights"]) + P["out_bias"]
    both = torch.einsum("si,sj->ij", mask, mask)
    return out / (both + 1e-3)[..., None]


def omegafold_geometric_attention(P, edge_repr, mask):
    """GeometricAttention.forward (modules.py:716-723) on one [N,N,d] edge tensor with residue mask [N]; no sub-batching.
    Axis r = 0 works on the edge tensor, r = 1 on its transpose (`_get_sharded_stacked`, :550-565).
      attended (:616-652): per row s of X_r a gated multi-head attention over the row (`Attention`, :400-481) whose logits
        get the per-pair bias X_r[q,k] . linear_b_weights[:,r,h] + linear_b_bias[r,h]; the axis-1 result is transposed back
        and added.  NOTE: the vendored code builds `b` from the mask bias (:627) and then ASSIGNS the pair bias into every
        row block of it (:645-647), so the mask bias is overwritten and no key is masked in this term; restated as is.
      gated (:654-689): GLU projections of X_r (even / odd quarter blocks of act_w for the first / second factor,
        :691-715), masked by the residue of their FIRST index, contracted over the second index, normalised, projected
        and gated by sigmoid of the last act_w block; summed over the axes WITHOUT transposing the axis-1 term back."""
    e = _unit_norm(edge_repr)
    d = e.shape[-1]
    H, c = P["attention.qg_weights"].shape[2], P["attention.kv_weights"].shape[-1] // 2
    out = torch.zeros_like(e)
    for r in range(2):
        X = e if r == 0 else e.transpose(0, 1)
        # ---- attention along the rows of X
        qg = torch.einsum("sqa,ahc->shqc", X, P["attention.qg_weights"][:, r]) + P["attention.qg_bias"][r]
        kv = torch.einsum("ska,ahc->shkc", X, P["attention.kv_weights"][:, r]) + P["attention.kv_bias"][r]
        q, g = qg[..., :c], qg[..., c:]
        k, v = kv[..., :c], kv[..., c:]
        pair_bias = torch.einsum("qka,ah->hqk", X, P["linear_b_weights"][:, r]) + P["linear_b_bias"][r]
        logits = torch.einsum("shqc,shkc->shqk", q, k) * c ** -0.5 + pair_bias[None]
        att = torch.einsum("shqk,shkc->shqc", torch.softmax(logits, -1), v) * torch.sigmoid(g)
        o = torch.einsum("shqc,hco->sqo", att, P["attention.o_weights"][r]) + P["attention.o_bias"][:, r]
        out = out + (o if r == 0 else o.transpose(0, 1))
        # ---- gated product
        w, b = P["act_w"][:, r], P["act_b"][r]
        blk = lambda t, i: t[..., i * d:(i + 1) * d]
        glu = lambda i_p, i_g: (X @ blk(w, i_p) + blk(b, i_p)) * torch.sigmoid(X @ blk(w, i_g) + blk(b, i_g))
        first = glu(0, 2) * mask[:, None, None]                             # [i, k, d]
        second = glu(1, 3) * mask[:, None, None]                            # [j, k, d]
        ab = _unit_norm(torch.einsum("ikd,jkd->ijd", first, second))
        gate = torch.sigmoid(X @ blk(w, 4) + blk(b, 4))
        out = out + (ab @ P["out_proj_w"][r] + P["out_proj_b"][r]) * gate
    return out


def make_atom14(aatype, pos37, mask37):
    """make_atom14_masks + make_atom14_positions (data_transforms.py:572-643, :653-752), restated with the reference's
    permutation-matrix formulation (einsum with the 14x14 renaming matrices)."""
    T = residue_tables()
    exists = T["atom14_exists"][aatype]
    idx = T["atom14_to_atom37"][aatype]
    gt_mask = exists.to(mask37.dtype) * torch.gather(mask37, -1, idx)
    gt_pos = gt_mask[..., None] * torch.gather(pos37, -2, idx[..., None].expand(idx.shape + (3,)))
    M = torch.zeros(21, 14, 14, dtype=mask37.dtype)
    M[torch.arange(21)[:, None], torch.arange(14)[None, :], T["atom14_rename"]] = 1.0
    Mr = M[aatype]
    return {"atom14_atom_exists": exists, "residx_atom14_to_atom37": idx, "residx_atom37_to_atom14": T["atom37_to_atom14"][aatype],
            "atom37_atom_exists": T["atom37_mask"][aatype], "atom14_gt_exists": gt_mask, "atom14_gt_positions": gt_pos,
            "atom14_alt_gt_positions": torch.einsum("...rac,...rab->...rbc", gt_pos, Mr),
            "atom14_alt_gt_exists": torch.einsum("...ra,...rab->...rb", gt_mask, Mr),
            "atom14_atom_is_ambiguous": T["atom14_is_ambiguous"][aatype].to(mask37.dtype)}


# ----------------------------------------------------------------------------
# Philox4x32-10 (Salmon et al., SC'11) -- checker of the device draws (csrc/rng.hip); pinned in tests/test_host_cpu.py to
# the known-answer vectors of the Random123 distribution.  The reference itself draws with numpy on the host
# (so3_diffuser.py:347-349, r3_diffuser.py:140-147): the device stream is an engine extension, not a reference stream.
# ----------------------------------------------------------------------------

def philox4x32_10(ctr, key):
    c, k = [int(x) for x in ctr], [int(x) for x in key]
    for _ in range(10):
        p0, p1 = 0xD2511F53 * c[0], 0xCD9E8D57 * c[2]
        c = [((p1 >> 32) ^ c[1] ^ k[0]) & 0xffffffff, p1 & 0xffffffff, ((p0 >> 32) ^ c[3] ^ k[1]) & 0xffffffff, p0 & 0xffffffff]
        k = [(k[0] + 0x9E3779B9) & 0xffffffff, (k[1] + 0xBB67AE85) & 0xffffffff]
    return c


def philox_stream(seed, subseq, n, normal):
    """first n elements of stream (seed, subseq) as csrc/rng.hip defines it: fp64 uniforms (x + 0.5) 2^-32 or Box-Muller
    normals on the word pairs (0,1), (2,3) of each counter block"""
    out = np.empty((n + 3) // 4 * 4, np.float64)
    for b in range((n + 3) // 4):
        x = philox4x32_10([b & 0xffffffff, b >> 32, subseq & 0xffffffff, subseq >> 32], [seed & 0xffffffff, seed >> 32])
        u = (np.array(x, np.float64) + 0.5) * 2.0 ** -32
        if normal:
            r0, r1 = np.sqrt(-2.0 * np.log(u[0])), np.sqrt(-2.0 * np.log(u[2]))
            u = np.array([r0 * np.cos(2 * np.pi * u[1]), r0 * np.sin(2 * np.pi * u[1]),
                          r1 * np.cos(2 * np.pi * u[3]), r1 * np.sin(2 * np.pi * u[3])])
        out[4 * b:4 * b + 4] = u
    return out[:n]
