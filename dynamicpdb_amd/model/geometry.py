"""Small-tensor geometry of the trunk (quaternion / rigid algebra, torsion frames -> atoms, score-head
chain rules).  These act on [.., 3|4|7]-sized fp32 tensors (a few hundred KB per window); they are chained
by autograd around the HIP nodes.  Semantics follow the reference files cited per function (paths under the
reference tree)."""
import math
import os

import numpy as np
import torch


def quat_to_rot(q):
    """openfold/utils/rigid_utils.py:185-205 -- quadratic form, no normalisation."""
    a, b, c, d = q.unbind(-1)
    m = torch.stack([
        a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c),
        2 * (b * c + a * d), a * a - b * b + c * c - d * d, 2 * (c * d - a * b),
        2 * (b * d - a * c), 2 * (c * d + a * b), a * a - b * b - c * c + d * d], -1)
    return m.reshape(q.shape[:-1] + (3, 3))


def quat_mul(p, q):
    """openfold/utils/rigid_utils.py:230-263"""
    a1, b1, c1, d1 = p.unbind(-1)
    a2, b2, c2, d2 = q.unbind(-1)
    return torch.stack([a1 * a2 - b1 * b2 - c1 * c2 - d1 * d2, a1 * b2 + b1 * a2 + c1 * d2 - d1 * c2,
                        a1 * c2 - b1 * d2 + c1 * a2 + d1 * b2, a1 * d2 + b1 * c2 - c1 * b2 + d1 * a2], -1)


def quat_invert(q):
    """openfold/utils/rigid_utils.py:282-286"""
    # (no constant tensor built from a Python list: that is a host -> device copy in every forward, and not capturable)
    return torch.cat([q[..., :1], -q[..., 1:]], -1) / (q * q).sum(-1, keepdim=True)


def rot_apply(R, x):
    """openfold/utils/rigid_utils.py:82-106"""
    return (R * x[..., None, :]).sum(-1)


def compose_q_update_vec(t7, upd6, mask):
    """Rigid.compose_q_update_vec (openfold/utils/rigid_utils.py:1039-1063, Rotation :587-616, normalisation :331-332)."""
    q, t = t7[..., :4], t7[..., 4:]
    zero = torch.zeros_like(upd6[..., :1])
    dq = quat_mul(q, torch.cat([zero, upd6[..., :3]], -1)) * mask
    qn = q + dq
    qn = qn / torch.linalg.norm(qn, dim=-1, keepdim=True)
    dt = rot_apply(quat_to_rot(q), upd6[..., 3:]) * mask
    return torch.cat([qn, t + dt], -1)


def quat_to_rotvec(quat, eps=1e-6):
    """src/data/utils.py:589-606"""
    flip = (quat[..., :1] < 0).to(quat.dtype)
    quat = quat * (1 - 2 * flip)
    angle = 2 * torch.atan2(torch.linalg.norm(quat[..., 1:], dim=-1), quat[..., 0])
    a2 = angle * angle
    small = 2 + a2 / 12 + 7 * a2 * a2 / 2880
    large = angle / torch.sin(angle / 2 + eps)
    is_small = (angle <= 1e-3).to(quat.dtype)
    scale = small * is_small + (1 - is_small) * large
    return scale[..., None] * quat[..., 1:]


_TABLES = {}


def residue_tables(device):
    t = _TABLES.get(device)
    if t is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "data", "residue_tables.npz")
        d = np.load(path)
        t = {k: torch.tensor(d[k]).to(device).contiguous() for k in d.files}
        _TABLES[device] = t
    return t


def _atoms_launch(t7c, ang, aa, P):
    from ctypes import c_int64
    from .. import _lib
    from ..ops import _p
    T = residue_tables(t7c.device)
    a14 = torch.empty((P, 14, 3), dtype=torch.float32, device=t7c.device)
    a37 = torch.empty((P, 37, 3), dtype=torch.float32, device=t7c.device)
    _lib.check(_lib.lib().dfold_frames_to_atoms(_p(t7c), _p(ang), _p(aa), _p(T["default_frames"]), _p(T["atom14_group"]),
                                                _p(T["atom14_mask"]), _p(T["atom14_pos"]), _p(T["atom37_to_atom14"]),
                                                _p(T["atom37_mask"]), _p(a14), _p(a37), c_int64(P), _lib.stream()),
               "dfold_frames_to_atoms")
    return a14, a37


class FramesToAtomsFn(torch.autograd.Function):
    """frames + torsions -> (atom14, atom37) as an autograd node: the reference builds the atoms inside the graph
    (src/model/Dfold_network_dynamic.py:532-538), so its bb-atom / dist-mat loss terms (train_DFOLD_dynamics.py:1317-1364,
    weights `bb_atom_loss_weight` / `dist_mat_loss_weight`) reach the frames and the torsions through them.  One HIP launch
    each way (csrc/atoms.hip)."""

    @staticmethod
    def forward(ctx, t7, angles, aatype):
        lead = t7.shape[:-1]
        P = t7.numel() // 7
        t7c, ang = t7.detach().reshape(P, 7).float().contiguous(), angles.detach().reshape(P, 14).float().contiguous()
        aa = aatype.reshape(P).long().contiguous()
        a14, a37 = _atoms_launch(t7c, ang, aa, P)
        ctx.save_for_backward(t7c, ang, aa)
        ctx.shapes = (t7.shape, angles.shape, t7.dtype, angles.dtype)
        return a14.view(lead + (14, 3)), a37.view(lead + (37, 3))

    @staticmethod
    def backward(ctx, g14, g37):
        from ctypes import c_int64
        from .. import _lib
        from ..ops import _p
        t7c, ang, aa = ctx.saved_tensors
        P = t7c.shape[0]
        T = residue_tables(t7c.device)
        g14 = None if g14 is None else g14.reshape(P, 14, 3).float().contiguous()
        g37 = None if g37 is None else g37.reshape(P, 37, 3).float().contiguous()
        dt7 = torch.empty((P, 7), dtype=torch.float32, device=t7c.device)
        dang = torch.empty((P, 14), dtype=torch.float32, device=t7c.device)
        _lib.check(_lib.lib().dfold_frames_to_atoms_bwd(_p(t7c), _p(ang), _p(aa), _p(T["default_frames"]), _p(T["atom14_group"]),
                                                        _p(T["atom14_mask"]), _p(T["atom14_pos"]), _p(T["atom37_to_atom14"]),
                                                        _p(T["atom37_mask"]), _p(g14), _p(g37), _p(dt7), _p(dang), c_int64(P),
                                                        _lib.stream()), "dfold_frames_to_atoms_bwd")
        s7, sa, d7, da = ctx.shapes
        return dt7.view(s7).to(d7), dang.view(sa).to(da), None


def frames_to_atoms_hip(t7, angles, aatype):
    """device path of frames_to_atoms: one HIP launch (csrc/atoms.hip).  Inside autograd when the frames or the torsions
    carry a graph (like the reference); a plain launch otherwise."""
    if torch.is_grad_enabled() and (t7.requires_grad or angles.requires_grad):
        return FramesToAtomsFn.apply(t7, angles, aatype)
    lead = t7.shape[:-1]
    P = t7.numel() // 7
    t7c, ang = t7.detach().reshape(P, 7).float().contiguous(), angles.detach().reshape(P, 14).float().contiguous()
    a14, a37 = _atoms_launch(t7c, ang, aatype.reshape(P).long().contiguous(), P)
    return a14.view(lead + (14, 3)), a37.view(lead + (37, 3))
