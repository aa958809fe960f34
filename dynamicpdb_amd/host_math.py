"""Tiny host-side (CPU tensor) quaternion helpers used only when the diffuser is
called on CPU tensors (dataset workers, as in the reference); device tensors go
through the HIP kernels in dynamicpdb_amd.ops instead.  Semantics follow
openfold/utils/rigid_utils.py:254-286 and src/data/utils.py:589-606 of the reference."""
import torch


def quat_multiply(p, q):
    a1, b1, c1, d1 = p.unbind(-1)
    a2, b2, c2, d2 = q.unbind(-1)
    return torch.stack([a1 * a2 - b1 * b2 - c1 * c2 - d1 * d2, a1 * b2 + b1 * a2 + c1 * d2 - d1 * c2,
                        a1 * c2 - b1 * d2 + c1 * a2 + d1 * b2, a1 * d2 + b1 * c2 - c1 * b2 + d1 * a2], -1)


def invert_quat(q):
    return q * q.new_tensor([1.0, -1.0, -1.0, -1.0]) / (q * q).sum(-1, keepdim=True)


def quat_to_rotvec(quat, eps=1e-6):
    quat = torch.where(quat[..., :1] < 0, -quat, quat)
    angle = 2 * torch.atan2(torch.linalg.norm(quat[..., 1:], dim=-1), quat[..., 0])
    a2 = angle * angle
    small = 2 + a2 / 12 + 7 * a2 * a2 / 2880
    large = angle / torch.sin(angle / 2 + eps)
    scale = torch.where(angle <= 1e-3, small, large)
    return scale[..., None] * quat[..., 1:]
