"""Minimal Rigid / Rotation carriers mirroring the slice of the reference's
openfold/utils/rigid_utils.py API that the DFOLDv2 hot path touches
(Rigid.from_tensor_7 :1213, to_tensor_7 :1200, get_rots/get_trans, apply :1104,
invert_apply :1118, compose_q_update_vec :1039; Rotation.get_quats/get_rot_mats,
invert).  They are thin views over a [...,7] tensor (qw,qx,qy,qz,tx,ty,tz); the
arithmetic on device goes through the HIP ops in dynamicpdb_amd.ops, the tiny
host-side helpers below are plain torch (used by the numpy-side diffuser only).
"""
import torch


def quat_to_rot(q: torch.Tensor) -> torch.Tensor:
    """Quadratic-form rotation matrix of a (not necessarily unit) quaternion
    (reference semantics: rigid_utils.py:185-205, no normalisation)."""
    a, b, c, d = q.unbind(-1)
    m = torch.stack([
        a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c),
        2 * (b * c + a * d), a * a - b * b + c * c - d * d, 2 * (c * d - a * b),
        2 * (b * d - a * c), 2 * (c * d + a * b), a * a - b * b - c * c + d * d], -1)
    return m.reshape(q.shape[:-1] + (3, 3))


def rot_to_quat(R: torch.Tensor) -> torch.Tensor:
    """Rotation matrix -> unit quaternion (w,x,y,z), branch on the largest of
    (trace, Rxx, Ryy, Rzz).  Replaces the reference's CPU torch.linalg.eigh
    (rigid_utils.py:208-227); result is defined up to sign exactly like the
    eigenvector the reference returns -- compare after sign canonicalisation."""
    if R.shape[-2:] != (3, 3):
        raise ValueError("Input rotation is incorrectly shaped")
    xx, yy, zz = R[..., 0, 0], R[..., 1, 1], R[..., 2, 2]
    xy, xz, yx, yz, zx, zy = (R[..., 0, 1], R[..., 0, 2], R[..., 1, 0], R[..., 1, 2], R[..., 2, 0], R[..., 2, 1])
    cands = torch.stack([
        torch.stack([1 + xx + yy + zz, zy - yz, xz - zx, yx - xy], -1),
        torch.stack([zy - yz, 1 + xx - yy - zz, xy + yx, xz + zx], -1),
        torch.stack([xz - zx, xy + yx, 1 - xx + yy - zz, yz + zy], -1),
        torch.stack([yx - xy, xz + zx, yz + zy, 1 - xx - yy + zz], -1)], -2)   # [...,4,4]
    diag = torch.stack([cands[..., i, i] for i in range(4)], -1)
    best = diag.argmax(-1)
    q = torch.gather(cands, -2, best[..., None, None].expand(best.shape + (1, 4))).squeeze(-2)
    return q / torch.linalg.norm(q, dim=-1, keepdim=True)


class Rotation:
    def __init__(self, rot_mats=None, quats=None, normalize_quats=True):
        if (rot_mats is None) == (quats is None):
            raise ValueError("Exactly one input argument must be specified")
        if (rot_mats is not None and rot_mats.shape[-2:] != (3, 3)) or (quats is not None and quats.shape[-1] != 4):
            raise ValueError("Incorrectly shaped rotation matrix or quaternion")
        if quats is not None:
            quats = quats.to(torch.float32)
            if normalize_quats:
                quats = quats / torch.linalg.norm(quats, dim=-1, keepdim=True)
        if rot_mats is not None:
            rot_mats = rot_mats.to(torch.float32)
        self._rot_mats, self._quats = rot_mats, quats

    @property
    def shape(self):
        return self._quats.shape[:-1] if self._quats is not None else self._rot_mats.shape[:-2]

    def get_quats(self):
        return self._quats if self._quats is not None else rot_to_quat(self._rot_mats)

    def get_rot_mats(self):
        return self._rot_mats if self._rot_mats is not None else quat_to_rot(self._quats)

    def invert(self):
        if self._rot_mats is not None:
            return Rotation(rot_mats=self._rot_mats.transpose(-1, -2))
        q = self._quats
        conj = q * q.new_tensor([1.0, -1.0, -1.0, -1.0])
        return Rotation(quats=conj / (q * q).sum(-1, keepdim=True), normalize_quats=False)

    def apply(self, pts):
        return (self.get_rot_mats() * pts[..., None, :]).sum(-1)

    def invert_apply(self, pts):
        return (self.get_rot_mats().transpose(-1, -2) * pts[..., None, :]).sum(-1)


class Rigid:
    def __init__(self, rots: Rotation, trans: torch.Tensor):
        if trans is None:
            trans = torch.zeros(tuple(rots.shape) + (3,), dtype=torch.float32)
        self._rots, self._trans = rots, trans.to(torch.float32)

    @property
    def shape(self):
        return self._trans.shape[:-1]

    @property
    def device(self):
        return self._trans.device

    def get_rots(self):
        return self._rots

    def get_trans(self):
        return self._trans

    @staticmethod
    def from_tensor_7(t, normalize_quats=False):
        if t.shape[-1] != 7:
            raise ValueError("Incorrectly shaped input tensor")
        return Rigid(Rotation(quats=t[..., :4], normalize_quats=normalize_quats), t[..., 4:])

    def to_tensor_7(self):
        return torch.cat([self._rots.get_quats(), self._trans], -1)

    def apply(self, pts):
        return self._rots.apply(pts) + self._trans

    def invert_apply(self, pts):
        return self._rots.invert_apply(pts - self._trans)

    def apply_trans_fn(self, fn):
        return Rigid(self._rots, fn(self._trans))

    def compose_q_update_vec(self, q_update_vec, update_mask=None):
        from .model import geometry
        if update_mask is None:
            update_mask = torch.ones_like(q_update_vec[..., :1])
        return Rigid.from_tensor_7(geometry.compose_q_update_vec(self.to_tensor_7(), q_update_vec, update_mask))
