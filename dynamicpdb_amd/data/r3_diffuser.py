"""Variance-preserving SDE over residue translations.

Boundary: the method names, argument meaning and ValueErrors of the reference's R3Diffuser
(src/data/r3_diffuser.py:7-177) -- SE3Diffuser and the parity tests call it exactly like the reference's.
Inside, everything is expressed through the two closed-form quantities of the linear-beta schedule

    B(t)     = t b_min + t^2 (b_max - b_min) / 2         integrated noise rate      (reference :42-43)
    alpha(t) = exp(-B(t) / 2),  var(t) = 1 - alpha(t)^2   signal scale / marginal variance

so that q(x_t | x_0) = N(alpha x_0, var), score = (alpha x_0 - x_t) / var, and one reverse Euler-Maruyama step is the
affine map x <- x (1 + beta dt / 2) + beta dt score - sqrt(beta dt) eps.  Device tensors take the HIP path
(model.score_heads / csrc/diffusion.hip via SE3Diffuser.*_t7); the numpy forms below serve host callers (the loader's
fork workers in the reference, the golden-vector tests here)."""
import numpy as np
import torch


def _xp(x, use_torch=False):
    return torch if (use_torch or torch.is_tensor(x)) else np


class R3Diffuser:
    def __init__(self, r3_conf):
        self._r3_conf = r3_conf
        self.min_b, self.max_b = r3_conf.min_b, r3_conf.max_b

    # ---- schedule ----------------------------------------------------------------------------
    def _check_t(self, t):
        tt = t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)
        if np.any(tt < 0) or np.any(tt > 1):
            raise ValueError(f'Invalid t={t}')

    def b_t(self, t):
        """instantaneous noise rate beta(t)"""
        self._check_t(t)
        return self.min_b + (self.max_b - self.min_b) * t

    def marginal_b_t(self, t):
        """B(t), the integral of beta over [0, t]"""
        return self.min_b * t + (self.max_b - self.min_b) * (t ** 2) / 2

    def _alpha(self, t, use_torch=False):
        return _xp(t, use_torch).exp(-0.5 * self.marginal_b_t(t))

    def conditional_var(self, t, use_torch=False):
        return 1 - _xp(t, use_torch).exp(-self.marginal_b_t(t))

    def diffusion_coef(self, t):
        return np.sqrt(self.b_t(t))

    def drift_coef(self, x, t):
        return x * (-0.5 * self.b_t(t))

    def score_scaling(self, t: float):
        return 1 / np.sqrt(self.conditional_var(t))

    def _scale(self, x):
        return self._r3_conf.coordinate_scaling * x

    def _unscale(self, x):
        return x / self._r3_conf.coordinate_scaling

    # ---- marginal, score ----------------------------------------------------------------------
    def sample_ref(self, n_samples: float = 1):
        return np.random.normal(size=(n_samples, 3))

    def score(self, x_t, x_0, t, use_torch=False, scale=False):
        """grad log q(x_t | x_0) = (alpha x_0 - x_t) / var"""
        if use_torch and torch.is_tensor(x_t) and x_t.is_cuda:
            from ..model import score_heads
            return score_heads.r3_score(x_t, x_0, t, self.min_b, self.max_b, self._r3_conf.coordinate_scaling if scale else 1.0)
        if scale:
            x_t, x_0 = self._scale(x_t), self._scale(x_0)
        return -(x_t - self._alpha(t, use_torch) * x_0) / self.conditional_var(t, use_torch=use_torch)

    def calc_trans_0(self, score_t, x_t, t, use_torch=True):
        """invert the score for x_0: (var score + x_t) / alpha"""
        B = self.marginal_b_t(t)[..., None, None]
        xp = _xp(B, use_torch)
        return (x_t + (1 - xp.exp(-B)) * score_t) / xp.exp(-0.5 * B)

    def forward_marginal(self, x_0: np.ndarray, t: float):
        if not np.isscalar(t):
            raise ValueError(f'{t} must be a scalar.')
        x0s = self._scale(x_0)
        a = self._alpha(t)
        x_t = np.random.normal(loc=a * x0s, scale=np.sqrt(self.conditional_var(t)))   # one draw per coordinate, the reference's order
        return self._unscale(x_t), self.score(x_t, x0s, t)

    # ---- reverse time -------------------------------------------------------------------------
    def _reverse_mean_shift(self, xs, score_t, t, dt):
        """(f - g^2 score) dt of the reverse SDE on scaled coordinates"""
        beta = self.b_t(t)
        return -(0.5 * xs + score_t) * (beta * dt), np.sqrt(beta * dt)

    def distribution(self, x_t, score_t, t, mask, dt):
        xs = self._scale(x_t)
        shift, std = self._reverse_mean_shift(xs, score_t, t, dt)
        mu = xs - shift
        if mask is not None:
            mu = mu * mask[..., None]
        return mu, std

    def reverse(self, *, x_t, score_t, t, dt, mask=None, center=True, noise_scale=1.0, z=None):
        """One Euler-Maruyama step backwards in time.  `z` injects the standard-normal draw (parity with the device path);
        by default it comes from numpy's global RNG like the reference's."""
        if not np.isscalar(t):
            raise ValueError(f'{t} must be a scalar.')
        xs = self._scale(x_t)
        shift, std = self._reverse_mean_shift(xs, score_t, t, dt)
        eps = np.random.normal(size=score_t.shape) if z is None else z
        step = shift + std * (noise_scale * eps)
        live = np.ones(xs.shape[:-1]) if mask is None else mask
        if mask is not None:
            step = step * mask[..., None]
        out = xs - step
        if center:                                   # per frame: remove the centre of mass of the live residues
            out = out - (out.sum(axis=-2) / live.sum(axis=-1)[..., None])[..., None, :]
        return self._unscale(out)
