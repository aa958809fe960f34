import os
import sys

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_golden(name):
    d = np.load(os.path.join(GOLDEN, name))
    return {k: d[k] for k in d.files}


def rel_l2(a, b):
    a = torch.as_tensor(a).detach().cpu().double().reshape(-1)
    b = torch.as_tensor(b).detach().cpu().double().reshape(-1)
    return float((a - b).norm() / (b.norm() + 1e-30))


def max_abs(a, b):
    return float((torch.as_tensor(a).detach().cpu().double() - torch.as_tensor(b).detach().cpu().double()).abs().max())


def canon_quat(t7):
    """q == -q: flip so that w >= 0."""
    t7 = torch.as_tensor(t7).clone()
    s = torch.where(t7[..., :1] < 0, -1.0, 1.0).to(t7.dtype)
    t7[..., :4] = t7[..., :4] * s
    return t7


def window_from_golden(g, device="cpu"):
    w = {}
    for k, v in g.items():
        if k.startswith("in_"):
            w[k[3:]] = torch.tensor(v).to(device)
    return w
