"""VP-SDE over translations.  API mirror of the reference's
src/data/r3_diffuser.py (R3Diffuser :7-177): same method names, argument meaning
and ValueError behaviour; numpy on the host, HIP (dynamicpdb_amd.ops) when
use_torch=True is requested on device tensors."""
import numpy as np
import torch


class R3Diffuser:
    def __init__(self, r3_conf):
        self._r3_conf = r3_conf
        self.min_b = r3_conf.min_b
        self.max_b = r3_conf.max_b

    # -- schedule (reference :20-43) --------------------------------------------
    def _scale(self, x):
        return x * self._r3_conf.coordinate_scaling

    def _unscale(self, x):
        return x / self._r3_conf.coordinate_scaling

    def b_t(self, t):
        if np.any(t < 0) or np.any(t > 1):
            raise ValueError(f'Invalid t={t}')
        return self.min_b + t * (self.max_b - self.min_b)

    def diffusion_coef(self, t):
        return np.sqrt(self.b_t(t))

    def drift_coef(self, x, t):
        return -0.5 * self.b_t(t) * x

    def marginal_b_t(self, t):
        return t * self.min_b + 0.5 * (t ** 2) * (self.max_b - self.min_b)

    def conditional_var(self, t, use_torch=False):
        if use_torch:
            return 1 - torch.exp(-self.marginal_b_t(t))
        return 1 - np.exp(-self.marginal_b_t(t))

    def score_scaling(self, t: float):
        return 1 / np.sqrt(self.conditional_var(t))

    # -- sampling (reference :39-40, :81-101, :106-157) ---------------------------
    def sample_ref(self, n_samples: float = 1):
        return np.random.normal(size=(n_samples, 3))

    def score(self, x_t, x_0, t, use_torch=False, scale=False):
        """-(x_t - e^{-b/2} x_0) / (1 - e^{-b}) (reference :169-177)."""
        if use_torch and torch.is_tensor(x_t) and x_t.is_cuda:
            from ..model import score_heads
            s = self._r3_conf.coordinate_scaling if scale else 1.0
            return score_heads.r3_score(x_t, x_0, t, self.min_b, self.max_b, s)
        exp_fn = torch.exp if use_torch else np.exp
        if scale:
            x_t, x_0 = self._scale(x_t), self._scale(x_0)
        return -(x_t - exp_fn(-0.5 * self.marginal_b_t(t)) * x_0) / self.conditional_var(t, use_torch=use_torch)

    def calc_trans_0(self, score_t, x_t, t, use_torch=True):
        beta_t = self.marginal_b_t(t)[..., None, None]
        exp_fn = torch.exp if use_torch else np.exp
        return (score_t * (1 - exp_fn(-beta_t)) + x_t) / exp_fn(-0.5 * beta_t)

    def forward_marginal(self, x_0: np.ndarray, t: float):
        if not np.isscalar(t):
            raise ValueError(f'{t} must be a scalar.')
        x_0 = self._scale(x_0)
        bt = self.marginal_b_t(t)
        x_t = np.random.normal(loc=np.exp(-0.5 * bt) * x_0, scale=np.sqrt(1 - np.exp(-bt)))
        score_t = self.score(x_t, x_0, t)
        return self._unscale(x_t), score_t

    def distribution(self, x_t, score_t, t, mask, dt):
        x_t = self._scale(x_t)
        g_t = self.diffusion_coef(t)
        f_t = self.drift_coef(x_t, t)
        mu = x_t - (f_t - g_t ** 2 * score_t) * dt
        if mask is not None:
            mu *= mask[..., None]
        return mu, g_t * np.sqrt(dt)

    def reverse(self, *, x_t, score_t, t, dt, mask=None, center=True, noise_scale=1.0, z=None):
        """One Euler-Maruyama step of the reverse VP-SDE.  `z` (optional) injects the
        standard-normal draw so device / host paths can be compared bit-for-bit in
        their randomness; default draws from numpy's global RNG like the reference."""
        if not np.isscalar(t):
            raise ValueError(f'{t} must be a scalar.')
        x_t = self._scale(x_t)
        g_t = self.diffusion_coef(t)
        f_t = self.drift_coef(x_t, t)
        if z is None:
            z = np.random.normal(size=score_t.shape)
        perturb = (f_t - g_t ** 2 * score_t) * dt + g_t * np.sqrt(dt) * (noise_scale * z)
        if mask is not None:
            perturb *= mask[..., None]
        else:
            mask = np.ones(x_t.shape[:-1])
        x_t_1 = x_t - perturb
        if center:
            com = np.sum(x_t_1, axis=-2) / np.sum(mask, axis=-1)[..., None]
            x_t_1 -= com[..., None, :]
        return self._unscale(x_t_1)
