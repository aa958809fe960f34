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


import contextlib


@contextlib.contextmanager
def record_relu_masks():
    """Test-side taps on every ReLU site of the path, in execution order -- the conv tower (inner / outer activation of
    each residual pair, as [windows, C, F, N]) and the dense ReLUs of AngleResnet: the 0/1 masks the engine's backward
    will use, fed back to the oracle (oracle.RELU_MASK_FEED) by the gradient-parity tests.  The product carries no hook
    for this: the tower's saved activations and the outputs of the patched functions are read from outside."""
    from dynamicpdb_amd import ops
    from dynamicpdb_amd.model import functional as F_
    from dynamicpdb_amd.model import ipa_pytorch_dynamic as ipa_mod
    log = []
    tower_fwd, relu, linear = ops.ConvTower.forward, ipa_mod._relu, F_.linear

    def tower_forward(self, g, h0, save=True, last_frame_only=False, slot=None):
        h, saved = tower_fwd(self, g, h0, True, last_frame_only, slot)
        for i in range(4):
            log.extend((g.interior(t) > 0).permute(0, 3, 1, 2).float().cpu() for t in (saved[1 + 3 * i], saved[2 + 3 * i]))
        return h, (saved if save else None)

    def relu_logged(x):
        y = relu(x)
        log.append((y > 0).float().cpu())
        return y

    def linear_logged(*a, **k):
        y = linear(*a, **k)
        if k.get("relu"):
            log.append((y > 0).float().cpu())
        return y

    # the fused angle head (functional.AngleResnetFn, round 6): its seven ReLU'd tensors are what it saves for its backward, in the
    # order of the reference's ReLU calls (s, s_initial, a, h, a, h, a) -- read from the node's save list as the forward returns
    angle_fwd = F_.AngleResnetFn.forward

    def angle_forward(ctx, *a):
        out = angle_fwd(ctx, *a)
        log.extend((t > 0).float().cpu() for t in ctx.to_save[:7])
        return out

    ops.ConvTower.forward, ipa_mod._relu, F_.linear = tower_forward, relu_logged, linear_logged
    F_.AngleResnetFn.forward = staticmethod(angle_forward)
    try:
        yield log
    finally:
        ops.ConvTower.forward, ipa_mod._relu, F_.linear = tower_fwd, relu, linear
        F_.AngleResnetFn.forward = staticmethod(angle_fwd)


def free_port():
    """a TCP port nobody listens on right now (rendezvous of the multi-process tests: a pid-derived port can collide with a socket
    an earlier test of the same session left in TIME_WAIT / a store that is still being torn down)"""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]
