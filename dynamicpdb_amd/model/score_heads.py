"""Device-side score heads evaluated inside the network forward (reference SE3Diffuser.calc_rot_score
src/data/se3_diffuser.py:119-125, SO3Diffuser.torch_score src/data/so3_diffuser.py:274-305, R3Diffuser.score
src/data/r3_diffuser.py:169-177).  The 1000-term IGSO(3) series is the HIP kernel dfold_igso3_series; the
[.., 3|4]-sized quaternion / rotation-vector chain around it is autograd glue."""
import numpy as np
import torch

from . import geometry as G
from .functional import Igso3SeriesFn


def series_envelope(sigma, L=1000):
    """(2l+1) exp(-l(l+1) sigma^2 / 2), float64 on the host exactly like the reference (so3_diffuser.py:45)."""
    sigma = np.atleast_1d(np.asarray(sigma, dtype=np.float64))
    ls = np.arange(L, dtype=np.float64)
    return (2 * ls + 1)[None, :] * np.exp(-ls[None, :] * (ls[None, :] + 1) * sigma[:, None] ** 2 / 2)


_ENV_ROWS = {}        # (sigma, L, device) -> float64 [L] device row; at most one per discretised sigma (1000) and device


def _envelope_on_device(sigma, L, device):
    """the series envelope of the windows' sigmas as a float64 [W, L] device tensor.  Rows are cached PER SIGMA (the schedule has
    1000 discrete values, so3_diffuser.py:188-190: the cache is bounded at 8 MB per device however the windows of a batch
    combine them; a cache keyed by the whole tuple of a batch almost never hit under per-window random t and kept up to 4096
    [W, L] tensors -- ADVICE r5): the host series + upload happens once per distinct value, a batch stacks its rows on the device."""
    sig = np.atleast_1d(np.asarray(sigma, dtype=np.float64))
    rows = []
    for v in sig.tolist():
        key = (v, L, str(device))
        r = _ENV_ROWS.get(key)
        if r is None:
            if len(_ENV_ROWS) > 8192:
                _ENV_ROWS.clear()
            r = _ENV_ROWS[key] = torch.tensor(series_envelope(v, L)[0], dtype=torch.float64, device=device)
        rows.append(r)
    if len(rows) == 1:
        return rows[0][None]
    if all(r is rows[0] for r in rows):
        return rows[0][None].expand(len(rows), L).contiguous()
    return torch.stack(rows)


def igso3_score(vec, sigma, eps=1e-6, L=1000):
    """vec [W, ..., 3] fp32 (window axis first), sigma [W] float64 (host) -> float64 score vectors."""
    env = _envelope_on_device(sigma, L, vec.device)
    W = env.shape[0]
    omega = torch.linalg.norm(vec, dim=-1) + eps
    if omega.numel() % W:
        raise ValueError("igso3_score: leading axis must enumerate the windows of `sigma`")
    sc = Igso3SeriesFn.apply(omega, env)
    return sc[..., None] * vec / (omega[..., None] + eps)


_HEAD_FUSED = __import__("os").environ.get("DFOLD_SCORE_FUSED", "1") != "0"


def rot_score(quats_t, quats_0, sigma, L=1000):
    """quats [W, ..., 4] (window axis first), sigma [W] float64 (host) -> float64 score vectors [W, ..., 3].  Device tensors whose
    q_t needs no gradient: four HIP launches (functional.RotScoreHeadFn); otherwise (DFOLD_SCORE_FUSED=0, a differentiable q_t) the
    aten chain of the same formulas."""
    if _HEAD_FUSED and quats_0.is_cuda and not quats_t.requires_grad:
        from .functional import RotScoreHeadFn
        env = _envelope_on_device(sigma, L, quats_0.device)
        if (quats_0.numel() // 4) % env.shape[0]:
            raise ValueError("rot_score: leading axis must enumerate the windows of `sigma`")
        return RotScoreHeadFn.apply(quats_t, quats_0, env)
    q0t = G.quat_mul(G.quat_invert(quats_0), quats_t)
    return igso3_score(G.quat_to_rotvec(q0t), sigma)


def r3_score(x_t, x_0, t, min_b, max_b, s):
    x_t, x_0 = x_t * s, x_0 * s
    bt = t * min_b + 0.5 * (t ** 2) * (max_b - min_b)
    return -(x_t - torch.exp(-0.5 * bt) * x_0) / (1 - torch.exp(-bt))
