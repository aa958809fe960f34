#!/usr/bin/env python
"""How stable is the per-tensor MAXIMUM of the full-loss gradient distance for an implementation of the engine's storage
class?  The bf16-emulating oracle (scripts/diag_emulation_parity.py) is run several times with its operands perturbed by a
relative 1e-6 (fp32-level noise: what a different summation order inside a kernel does) before every bf16 rounding; each run
is the same storage class, a different realisation of the rounding noise.  CPU only.
    python scripts/diag_emulation_spread.py network_F32_N256.npz 4"""
import json
import os
import sys
import time

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import diag_emulation_parity as D  # noqa: E402
from oracle import dfold_oracle as O  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "network_F32_N256.npz"
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
gen = torch.Generator().manual_seed(0)
q_orig = O._q


def q_jitter(x):
    if not O.EMULATE_BF16_OPERANDS:
        return x
    xd = x.detach()
    xj = xd * (1.0 + 1e-6 * torch.randn(xd.shape, generator=gen, dtype=xd.dtype))
    return x + (xj.to(torch.bfloat16).to(x.dtype) - xd)


res = []
for r in range(runs):
    O._q = q_orig if r == 0 else q_jitter
    t0 = time.time()
    loss, ref_loss, stats, fwd = D.run(name, True, None)
    rel = sorted(v[1] for v in stats.values())
    worst = max(stats.items(), key=lambda kv: kv[1][1])
    # norm-weighted aggregate over all tensors (the whole-gradient relative error; cosine ~ 1 - e^2 / 2)
    g = D.load_golden(name)
    num = sum((v[1] * float(g["gnorm_" + k])) ** 2 for k, v in stats.items())
    den = sum(float(g["gnorm_" + k]) ** 2 for k in stats)
    e = (num / den) ** 0.5
    print(f"[{name} realisation {r}] median {rel[len(rel) // 2]:.4f} max {rel[-1]:.4f} ({worst[0].replace('score_model.', '')}) "
          f"whole-gradient rel {e:.4f} fwd raw {fwd['unorm_angles']:.5f} ({time.time() - t0:.0f} s)", flush=True)
    res.append({"median": rel[len(rel) // 2], "max": rel[-1], "worst": worst[0], "whole": e})
O._q = q_orig
with open(os.path.join(ROOT, "profiles", "r6_emulation_spread_" + name.replace(".npz", ".json")), "w") as fh:
    json.dump(res, fh, indent=1)
