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


DIFFUSER_KEYS = ("rigids_t", "rot_score", "trans_score", "rot_score_scaling", "trans_score_scaling")


def compact_window(g):
    """Inputs of a compact (BASELINE-sized) golden: everything but the diffuser-dependent tensors is regenerated from
    the seed by dynamicpdb_amd.synthetic.synthetic_window and pinned by the stored float64 checksums.
    Returns (window dict of CPU tensors, (F, N, seed_w, grad_stride))."""
    from dynamicpdb_amd import synthetic
    F, N, seed_w, seed_x, stride = [int(v) for v in g["meta"]]
    holes = float(g["holes"][0]) if "holes" in g else 0.0
    w = synthetic.synthetic_window(seed_x, F, N, t=float(g["t"][0]), diffuser=None, holes=holes)
    sums = np.array([float(np.asarray(w[k].numpy(), dtype=np.float64).sum()) for k in sorted(w)])
    assert np.array_equal(sums, g["in_checksum"]), "synthetic_window no longer reproduces the minted inputs"
    for k in DIFFUSER_KEYS:
        w[k] = torch.tensor(g["in_" + k])
    w["rigids_t"] = w["rigids_t"].float()
    return w, (F, N, seed_w, stride)


def golden_window(g):
    """(window of CPU tensors, (F, N, seed_w, grad_stride)) of a network golden of either form: compact (inputs
    regenerated from the seed) or with every input stored."""
    if "in_checksum" in g:
        return compact_window(g)
    meta = [int(v) for v in g["meta"]]
    F, N, seed_w = meta[:3]
    stride = meta[4] if len(meta) > 4 else 9973
    return window_from_golden(g), (F, N, seed_w, stride)
