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


_ENV_CACHE = {}


def _envelope_on_device(sigma, L, device):
    """the series envelope of a tuple of sigmas as a device tensor, cached: the sampler visits num_t diffusion times and a
    training run a discretised schedule of 1000 sigmas, so the host series + its upload (a synchronous copy in every forward,
    which also keeps the forward out of HIP graphs) happens once per distinct value"""
    key = (tuple(np.atleast_1d(np.asarray(sigma, dtype=np.float64)).tolist()), L, str(device))
    env = _ENV_CACHE.get(key)
    if env is None:
        if len(_ENV_CACHE) > 4096:
            _ENV_CACHE.clear()
        env = _ENV_CACHE[key] = torch.tensor(series_envelope(sigma, L), dtype=torch.float64, device=device)
    return env


def igso3_score(vec, sigma, eps=1e-6, L=1000):
    """vec [W, ..., 3] fp32 (window axis first), sigma [W] float64 (host) -> float64 score vectors."""
    env = _envelope_on_device(sigma, L, vec.device)
    W = env.shape[0]
    omega = torch.linalg.norm(vec, dim=-1) + eps
    if omega.numel() % W:
        raise ValueError("igso3_score: leading axis must enumerate the windows of `sigma`")
    sc = Igso3SeriesFn.apply(omega, env)
    return sc[..., None] * vec / (omega[..., None] + eps)


def rot_score(quats_t, quats_0, sigma):
    q0t = G.quat_mul(G.quat_invert(quats_0), quats_t)
    return igso3_score(G.quat_to_rotvec(q0t), sigma)


def r3_score(x_t, x_0, t, min_b, max_b, s):
    x_t, x_0 = x_t * s, x_0 * s
    bt = t * min_b + 0.5 * (t ** 2) * (max_b - min_b)
    return -(x_t - torch.exp(-0.5 * bt) * x_0) / (1 - torch.exp(-bt))
