"""Optimizer of the training step: torch.optim.Adam(amsgrad=True) (reference train_DFOLD_dynamics.py:412) whose
step() is ONE HIP launch over all parameters (csrc/adam.hip) instead of torch's ~10 foreach passes.  State layout and
hyper-parameters are torch's own (exp_avg, exp_avg_sq, max_exp_avg_sq, step), so optimizer state_dicts of reference
checkpoints load unchanged (src/data/utils.py:353-362 saves 'optimizer')."""
from ctypes import c_double, c_int32, c_int64

import torch

from . import _lib
from .ops import _p


class FusedAdam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=True, foreach=False)
        self._chunk = None
        self._prefix = {}
        self._tables = {}

    def _init_state(self, p):
        st = self.state[p]
        if len(st) == 0:
            st["step"] = torch.tensor(0.0, dtype=torch.float32)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["max_exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _lib.lib()
        if self._chunk is None:
            self._chunk = int(L.dfold_adam_chunk())
        for group in self.param_groups:
            if group.get("weight_decay", 0) != 0 or group.get("maximize", False) or not group["amsgrad"]:
                raise ValueError("FusedAdam implements Adam(amsgrad=True, weight_decay=0, maximize=False)")
            by_step = {}
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.dtype != torch.float32 or p.grad.dtype != torch.float32 or not p.is_cuda:
                    raise ValueError("FusedAdam needs fp32 parameters and gradients on the GPU")
                st = self._init_state(p)
                by_step.setdefault(int(st["step"]), []).append(p)
            for step0, ps in by_step.items():       # normally one entry: every parameter has seen the same number of steps
                dev = ps[0].device
                rows, sizes = [], []
                keep = []
                for p in ps:
                    st = self.state[p]
                    g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                    if not (p.is_contiguous() and st["exp_avg"].is_contiguous()):
                        raise ValueError("FusedAdam needs contiguous parameters")
                    keep.append(g)
                    rows.append((p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                                 st["max_exp_avg_sq"].data_ptr(), p.numel()))
                    sizes.append(p.numel())
                key = (dev, tuple(sizes))
                pre = self._prefix.get(key)
                if pre is None:
                    acc, cs = 0, [0]
                    for n in sizes:
                        acc += (n + self._chunk - 1) // self._chunk
                        cs.append(acc)
                    pre = self._prefix[key] = (torch.tensor(cs, dtype=torch.int32).to(dev), acc)
                # the device pointer table only changes when a tensor is re-allocated (grads with set_to_none=True come
                # back from the caching allocator at the same addresses in steady state): re-upload on change only
                tkey = tuple(rows)
                ent = self._tables.get(key)
                if ent is None or ent[0] != tkey:
                    host = torch.tensor(rows, dtype=torch.int64).pin_memory()
                    ent = self._tables[key] = (tkey, host.to(dev, non_blocking=True), host)
                table = ent[1]
                b1, b2 = group["betas"]
                _lib.check(L.dfold_adam_amsgrad(_p(table), _p(pre[0]), c_int32(len(rows)), c_int32(pre[1]), c_double(group["lr"]),
                                                c_double(b1), c_double(b2), c_double(group["eps"]), c_int64(step0 + 1),
                                                _lib.stream()), "dfold_adam_amsgrad")
                for p in ps:
                    self.state[p]["step"] += 1
                # the kernel writes the parameters through raw pointers: tell autograd / the bf16 weight caches
                # (functional.WeightCache, ops.ConvTower.refresh key on (data_ptr, _version)) that they changed
                torch.autograd.graph.increment_version(ps)
                del keep
        return loss
