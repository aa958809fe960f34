"""Device random draws for the diffusion steps (csrc/rng.hip, Philox4x32-10): the production replacement of the
reference's host numpy draws inside the sampling loop (src/data/so3_diffuser.py:347-349, r3_diffuser.py:140-147) and of
the loader's forward-noising draws.  Every call consumes one subsequence of the (seed) stream, so a run is reproducible
from its seed alone and independent of launch geometry; the numpy-injection path of SE3Diffuser.reverse_t7 /
forward_marginal_t7 (draws as inputs) stays for parity with the reference's RNG stream."""
from ctypes import c_int64, c_uint64

import torch

from . import _lib
from .ops import _p


class DeviceRNG:
    def __init__(self, seed, device="cuda:0", subseq=0):
        self.seed, self.device, self.subseq = int(seed) & (2 ** 64 - 1), torch.device(device), int(subseq)

    def _fill(self, shape, fn, name):
        out = torch.empty(tuple(shape), dtype=torch.float64, device=self.device)
        if not out.is_cuda:
            raise RuntimeError("DeviceRNG draws on the MI355X (no CPU fallback)")
        _lib.check(fn(_p(out), c_int64(out.numel()), c_uint64(self.seed), c_uint64(self.subseq), _lib.stream()), name)
        self.subseq += 1
        return out

    def normal(self, shape):
        """standard normal fp64 tensor; consumes one subsequence"""
        return self._fill(shape, _lib.lib().dfold_philox_normal_f64, "dfold_philox_normal_f64")

    def uniform(self, shape):
        """uniform (0, 1) fp64 tensor; consumes one subsequence"""
        return self._fill(shape, _lib.lib().dfold_philox_uniform_f64, "dfold_philox_uniform_f64")

    def state(self):
        return dict(seed=self.seed, subseq=self.subseq)
