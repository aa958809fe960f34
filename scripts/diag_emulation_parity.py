#!/usr/bin/env python
"""how far from the reference's fp32 gradients does the ORACLE land when it merely rounds where the
engine rounds (oracle.EMULATE_BF16_OPERANDS: bf16 operands of every dense contraction, bf16 activation gradients between
kernels) -- with NO mask feeding (every ReLU branch free) and the REAL loss (torsion term included)?  CPU only.

    python scripts/diag_emulation_parity.py network_F32_N256.npz [network_F8_N512.npz ...]

Prints, per golden, the per-tensor rel-L2 (median / max / worst tensors) of (a) the emulation and (b) nothing else: the
engine's own numbers for the same goldens are printed by tests/test_parity_baseline_gpu.py (`-s`).  If (a) reproduces the
engine's distance the engine's rounding points explain it; what the engine has beyond (a) is its own."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import golden_window, load_golden, rel_l2  # noqa: E402
from oracle import dfold_oracle as O  # noqa: E402
from dynamicpdb_amd import synthetic  # noqa: E402


def run(name, emulate, torsion_w=None):
    g = load_golden(name)
    w, (F, N, seed_w, stride) = golden_window(g)
    P = {k: v.clone().requires_grad_(True) for k, v in synthetic.seeded_state_dict(seed_w).items()}
    O.EMULATE_BF16_OPERANDS = emulate
    try:
        out = O.full_score_network(P, O.Schedules(), w)
        loss, aux = O.loss_fn(out, w) if torsion_w is None else O.loss_fn(out, w, torsion_w=torsion_w)
        loss.backward()
    finally:
        O.EMULATE_BF16_OPERANDS = False
    sub, nrm = ("gsub_", "gnorm_") if torsion_w is None else ("g0sub_", "g0norm_")
    stats = {}
    for k in g:
        if not k.startswith(sub):
            continue
        n = k[len(sub):]
        gr, ref_norm = P[n].grad, float(g[nrm + n])
        if ref_norm < 1e-6 or gr is None:
            continue
        ref = torch.tensor(g[k]).double()
        mine = (gr.reshape(-1)[::stride] if gr.numel() > 70000 else gr).double().reshape(ref.shape)
        stats[n] = (abs(float(gr.double().norm()) - ref_norm) / ref_norm, float((mine - ref).norm() / (ref.norm() + 1e-30)))
    fwd = {k: rel_l2(out[k], g["out_" + k]) for k in ("unorm_angles", "angles", "rigid_update", "trans_score", "rot_score")}
    return float(loss), float(g["loss" if torsion_w is None else "loss_notorsion"]), stats, fwd


def main():
    res = {}
    for name in sys.argv[1:] or ["network_F32_N256.npz"]:
        for tag, tw in (("full", None), ("notorsion", 0.0)):
            t0 = time.time()
            loss, ref_loss, stats, fwd = run(name, True, tw)
            rel = sorted(v[1] for v in stats.values())
            worst = sorted(stats.items(), key=lambda kv: -kv[1][1])[:6]
            print(f"[{name} {tag}] emulation vs reference fp32: loss {loss:.5f} / {ref_loss:.5f}; grad rel-L2 median "
                  f"{rel[len(rel) // 2]:.4f} max {rel[-1]:.4f}; forward {json.dumps({k: round(v, 5) for k, v in fwd.items()})} "
                  f"({time.time() - t0:.0f} s)", flush=True)
            print("    worst:", ", ".join(f"{k.replace('score_model.', '')} {v[1]:.3f}" for k, v in worst), flush=True)
            res[f"{name}/{tag}"] = {"median": rel[len(rel) // 2], "max": rel[-1], "worst": [(k, v[1]) for k, v in worst],
                                    "forward": fwd, "rel": {k: v[1] for k, v in stats.items()}}
    out = os.path.join(ROOT, "profiles", "r6_emulation_parity.json")
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1)
    print("wrote", out)


if __name__ == "__main__":
    main()
