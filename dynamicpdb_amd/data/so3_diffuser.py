"""IGSO(3) diffusion over rotations.  API mirror of the reference's
src/data/so3_diffuser.py (module functions igso3_expansion :9, density :52,
score :71; class SO3Diffuser :120-365): same names / arguments / ValueErrors and
the same on-disk cache layout, but the 1000x1000 tables are built as two dense
float64 contractions over the series index (seconds instead of ~1 minute) and
torch_score on device tensors runs the fused HIP series kernel.
"""
import logging
import os

import numpy as np
import torch


def _series_terms(omega, L):
    """sin / d-sin factors of the truncated series, shape [L, n_omega] (float64)."""
    ls = np.arange(L, dtype=np.float64)[:, None]
    arg = omega[None, :] * (ls + 0.5)
    return np.sin(arg), (ls + 0.5) * np.cos(arg)


def _series_sums(omega, sigma, L):
    """Host evaluation (float64) of the truncated IGSO(3) series and of the numerator of its omega-derivative:
        f    = sum_l c_l(sigma) sin((l+1/2) omega) / sin(omega/2),      c_l = (2l+1) exp(-l(l+1) sigma^2 / 2)
        dnum = sum_l c_l(sigma) d/domega [sin((l+1/2) omega) / sin(omega/2)]
    omega and sigma broadcast against each other; the l axis is contracted in blocks so that a [n_sigma, n_omega]
    request never materialises more than ~16 MB of terms."""
    omega = np.asarray(omega, dtype=np.float64)
    sigma = np.asarray(sigma, dtype=np.float64)
    shape = np.broadcast(omega, sigma).shape
    om = np.broadcast_to(omega, shape).reshape(-1)
    sg = np.broadcast_to(sigma, shape).reshape(-1)
    ls = np.arange(L, dtype=np.float64)
    lo, dlo = np.sin(om / 2), 0.5 * np.cos(om / 2)
    f, dnum = np.empty_like(om), np.empty_like(om)
    step = max(1, (1 << 21) // max(L, 1))
    for i in range(0, om.size, step):
        o, s_ = om[i:i + step, None], sg[i:i + step, None]
        c = (2 * ls + 1) * np.exp(-ls * (ls + 1) * s_ ** 2 / 2)
        hi, dhi = np.sin(o * (ls + 0.5)), (ls + 0.5) * np.cos(o * (ls + 0.5))
        l0, d0 = lo[i:i + step, None], dlo[i:i + step, None]
        f[i:i + step] = (c * hi).sum(-1) / l0[:, 0]
        dnum[i:i + step] = (c * (l0 * dhi - hi * d0)).sum(-1) / (l0[:, 0] ** 2)
    return f.reshape(shape), dnum.reshape(shape)


def _host(x):
    return x.detach().cpu().numpy() if torch.is_tensor(x) else np.asarray(x)


def _check_rank(omega):
    if np.ndim(omega) not in (1, 2):
        raise ValueError("Omega must be 1D or 2D.")


def igso3_expansion(omega, eps, L=1000, use_torch=False):
    """Truncated power series of the IGSO(3) density at rotation angle(s) `omega` for scale(s) `eps`
    (reference module function src/data/so3_diffuser.py:9-49, same signature).  Host helper: evaluated in float64 numpy
    whatever the container; `use_torch` only selects the return type (no autograd -- the differentiable, on-device
    series is csrc/score.hip behind SO3Diffuser.torch_score)."""
    _check_rank(omega)
    f, _ = _series_sums(_host(omega), _host(eps), L)
    return torch.as_tensor(f) if use_torch else f


def density(expansion, omega, marginal=True):
    """IGSO(3) density from its series (reference :52-68): over the angle (marginal) or over SO(3)."""
    return expansion * (1 - np.cos(omega)) / np.pi if marginal else expansion / (8 * np.pi ** 2)


def score(exp, omega, eps, L=1000, use_torch=False):
    """d/d omega log f(omega; eps) with the reference's +1e-4 regulariser in the denominator (reference :71-117)."""
    if np.ndim(omega) > 2:
        raise ValueError("Omega must be 1D or 2D.")
    _, dnum = _series_sums(_host(omega), _host(eps), L)
    out = dnum / (_host(exp) + 1e-4)
    return torch.as_tensor(out) if use_torch else out


def _rotvec_to_matrix(v):
    from scipy.spatial.transform import Rotation
    return Rotation.from_rotvec(v).as_matrix()


def compose_rotvec(r1, r2):
    """Right-compose two rotation vectors (reference src/data/utils.py:184-195)."""
    from scipy.spatial.transform import Rotation
    R = np.einsum('...ij,...jk->...ik', _rotvec_to_matrix(r1), _rotvec_to_matrix(r2))
    return Rotation.from_matrix(R).as_rotvec()


class SO3Diffuser:
    def __init__(self, so3_conf):
        self.schedule = so3_conf.schedule
        self.min_sigma = so3_conf.min_sigma
        self.max_sigma = so3_conf.max_sigma
        self.num_sigma = so3_conf.num_sigma
        self.use_cached_score = so3_conf.use_cached_score
        self._log = logging.getLogger(__name__)
        self.discrete_omega = np.linspace(0, np.pi, so3_conf.num_omega + 1)[1:]

        tag = lambda x: str(x).replace('.', '_')
        cache_dir = os.path.join(
            so3_conf.cache_dir,
            f'eps_{so3_conf.num_sigma}_omega_{so3_conf.num_omega}_min_sigma_{tag(so3_conf.min_sigma)}'
            f'_max_sigma_{tag(so3_conf.max_sigma)}_schedule_{so3_conf.schedule}')
        names = [os.path.join(cache_dir, n) for n in ('pdf_vals.npy', 'cdf_vals.npy', 'score_norms.npy')]
        loaded = None
        if all(os.path.exists(n) for n in names):
            try:    # several ranks may start at once on a fresh box: a file another rank is still writing is ignored
                loaded = tuple(np.load(n) for n in names)
                shape = (self.num_sigma, so3_conf.num_omega)
                if any(a.shape != shape for a in loaded):
                    loaded = None
            except (OSError, ValueError, EOFError):
                loaded = None
        if loaded is not None:
            self._pdf, self._cdf, self._score_norms = loaded
        else:
            self._pdf, self._cdf, self._score_norms = self._build_tables(so3_conf.num_omega)
            try:
                os.makedirs(cache_dir, exist_ok=True)
                for n, a in zip(names, (self._pdf, self._cdf, self._score_norms)):
                    tmp = f"{n}.{os.getpid()}.tmp.npy"       # written whole, then renamed: readers never see a partial file
                    np.save(tmp, a)
                    os.replace(tmp, n)
            except OSError:
                self._log.warning(f'could not write IGSO3 cache to {cache_dir}')
        self._score_scaling = np.sqrt(np.abs(
            np.sum(self._score_norms ** 2 * self._pdf, axis=-1) / np.sum(self._pdf, axis=-1))) / np.sqrt(3)

    def _build_tables(self, num_omega, L=1000):
        """pdf / cdf / score-norm tables [num_sigma, num_omega]: the series sum over l is a
        [sigma,l] x [l,omega] contraction (reference loops igso3_expansion / score per sigma,
        so3_diffuser.py:150-168)."""
        om = self.discrete_omega
        sig = self.discrete_sigma
        ls = np.arange(L, dtype=np.float64)
        coef = (2 * ls + 1)[None, :] * np.exp(-ls[None, :] * (ls[None, :] + 1) * sig[:, None] ** 2 / 2)  # [S,L]
        hi, dhi = _series_terms(om, L)
        lo, dlo = np.sin(om / 2), 0.5 * np.cos(om / 2)
        exp_vals = (coef @ hi) / lo[None, :]
        pdf = exp_vals * (1 - np.cos(om))[None, :] / np.pi
        cdf = np.cumsum(pdf, axis=-1) / num_omega * np.pi
        dsig = (coef @ (lo[None, :] * dhi - hi * dlo[None, :])) / (lo ** 2)[None, :]
        return pdf, cdf, dsig / (exp_vals + 1e-4)

    def device_tables(self, device):
        """cdf [num_sigma,num_omega] and the omega grid as fp64 device tensors (uploaded once per device)."""
        cache = self.__dict__.setdefault('_dev_tables', {})
        key = str(device)
        if key not in cache:
            cache[key] = {'cdf': torch.as_tensor(np.ascontiguousarray(self._cdf, dtype=np.float64)).to(device),
                          'omega': torch.as_tensor(np.ascontiguousarray(self.discrete_omega, dtype=np.float64)).to(device)}
        return cache[key]

    @property
    def discrete_sigma(self):
        return self.sigma(np.linspace(0.0, 1.0, self.num_sigma))

    def sigma_idx(self, sigma):
        return np.digitize(sigma, self.discrete_sigma) - 1

    def sigma(self, t):
        if np.any(t < 0) or np.any(t > 1):
            raise ValueError(f'Invalid t={t}')
        if self.schedule == 'logarithmic':
            return np.log(t * np.exp(self.max_sigma) + (1 - t) * np.exp(self.min_sigma))
        raise ValueError(f'Unrecognize schedule {self.schedule}')

    def diffusion_coef(self, t):
        if self.schedule == 'logarithmic':
            s = self.sigma(t)
            return np.sqrt(2 * (np.exp(self.max_sigma) - np.exp(self.min_sigma)) * s / np.exp(s))
        raise ValueError(f'Unrecognize schedule {self.schedule}')

    def t_to_idx(self, t):
        return self.sigma_idx(self.sigma(t))

    def sample_igso3(self, t: float, n_samples: float = 1):
        if not np.isscalar(t):
            raise ValueError(f'{t} must be a scalar.')
        x = np.random.rand(n_samples)
        return np.interp(x, self._cdf[self.t_to_idx(t)], self.discrete_omega)

    def sample(self, t: float, n_samples: float = 1):
        x = np.random.randn(n_samples, 3)
        x /= np.linalg.norm(x, axis=-1, keepdims=True)
        return x * self.sample_igso3(t, n_samples=n_samples)[:, None]

    def sample_ref(self, n_samples: float = 1):
        return self.sample(1, n_samples=n_samples)

    def score(self, vec: np.ndarray, t: float, eps: float = 1e-6):
        if not np.isscalar(t):
            raise ValueError(f'{t} must be a scalar.')
        return self.torch_score(torch.tensor(vec), torch.tensor(t)[None]).numpy()

    def torch_score(self, vec, t, eps: float = 1e-6):
        """score(omega) * vec / (omega + eps), omega = |vec| + eps, sigma looked up by
        np.digitize on the discretised schedule (reference :274-305)."""
        t_np = t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)
        if vec.is_cuda and not self.use_cached_score:
            from ..model import score_heads
            sigma = np.atleast_1d(self.discrete_sigma[self.t_to_idx(t_np)])
            if len(sigma) == 1:
                return score_heads.igso3_score(vec[None], sigma, eps)[0]
            return score_heads.igso3_score(vec, sigma, eps)
        # host tensors (dataset-side noising, SO3Diffuser.score): float64 numpy series, no autograd
        vec64 = vec.detach().to(torch.float64)
        omega = torch.linalg.norm(vec64, dim=-1) + eps
        idx = self.t_to_idx(t_np)
        if self.use_cached_score:
            score_norms_t = torch.as_tensor(self._score_norms[idx]).to(vec.device)
            omega_idx = torch.bucketize(omega, torch.as_tensor(self.discrete_omega[:-1]).to(vec.device))
            omega_scores_t = torch.gather(score_norms_t, 1, omega_idx)
        else:
            sigma = np.atleast_1d(self.discrete_sigma[idx])[:, None]
            om = omega.cpu().numpy().reshape(sigma.shape[0], -1)
            f, dnum = _series_sums(om, sigma, 1000)
            omega_scores_t = torch.as_tensor((dnum / (f + 1e-4)).reshape(omega.shape)).to(vec.device)
        return omega_scores_t[..., None] * vec64 / (omega[..., None] + eps)

    def score_scaling(self, t):
        return self._score_scaling[self.t_to_idx(t)]

    def forward_marginal(self, rot_0: np.ndarray, t: float):
        n_samples = np.cumprod(rot_0.shape[:-1])[-1]
        sampled = self.sample(t, n_samples=n_samples)
        rot_score = self.score(sampled, t).reshape(rot_0.shape)
        rot_t = compose_rotvec(rot_0.reshape(-1, 3), sampled).reshape(rot_0.shape)
        return rot_t, rot_score

    def reverse(self, rot_t, score_t, t, dt, mask=None, noise_scale=1.0, z=None):
        """Geodesic random-walk step, right-multiplied (reference :329-365); `z` optionally
        injects the standard-normal draw."""
        if not np.isscalar(t):
            raise ValueError(f'{t} must be a scalar.')
        g_t = self.diffusion_coef(t)
        if z is None:
            z = np.random.normal(size=score_t.shape)
        perturb = (g_t ** 2) * score_t * dt + g_t * np.sqrt(dt) * (noise_scale * z)
        if mask is not None:
            perturb *= mask[..., None]
        n = np.cumprod(rot_t.shape[:-1])[-1]
        return compose_rotvec(rot_t.reshape(n, 3), perturb.reshape(n, 3)).reshape(rot_t.shape)
