#!/usr/bin/env python
"""Where does the engine's full-loss gradient distance at a BASELINE-sized golden come from -- the forward or the backward?
  GPU.   python scripts/diag_grad_split.py network_F32_N256.npz

 (a) the engine's own step: per-tensor rel-L2 against the reference's fp32 gradients (what test_parity_baseline_gpu prints);
 (b) HYBRID: the engine's forward graph, but the gradient of the loss with respect to the network outputs is evaluated at the
     REFERENCE's outputs (golden out_*) and injected into the engine's backward.  (b) removes the amplification of the engine's
     forward difference by the loss (the 1/|raw| torsion normalisation); what is left is the backward itself on the engine's
     activations.  If (b) sits at the level of the torsion-free run, the excess of (a) is forward difference x ill-conditioned
     read-out and the lever is forward precision; if (b) is as far out as (a), a backward kernel is at fault."""
import os
import sys
import json

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import golden_window, load_golden, rel_l2  # noqa: E402
from test_parity_baseline_gpu import _build  # noqa: E402
from dynamicpdb_amd import experiment  # noqa: E402

dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "network_F32_N256.npz"
g = load_golden(name)
w, (F, N, seed_w, stride) = golden_window(g)
model, _ = _build(F, seed_w, dev)
wd = {k: v.to(dev) for k, v in w.items()}
batch = {k: v[None] for k, v in wd.items()}
batch["t"] = wd["t"].reshape(1)
P = dict(model.named_parameters())


def stats(sub="gsub_", nrm="gnorm_"):
    rows = {}
    for k in g:
        if not k.startswith(sub):
            continue
        n = k[len(sub):]
        gr, rn = P[n].grad, float(g[nrm + n])
        if rn < 1e-6 or gr is None:
            continue
        ref = torch.tensor(g[k]).double()
        mine = (gr.reshape(-1)[::stride] if gr.numel() > 70000 else gr).double().cpu().reshape(ref.shape)
        rows[n] = float((mine - ref).norm() / (ref.norm() + 1e-30))
    return rows


def show(tag, rows):
    num = sum((x * float(g["gnorm_" + k])) ** 2 for k, x in rows.items() if "gnorm_" + k in g)
    den = sum(float(g["gnorm_" + k]) ** 2 for k in rows if "gnorm_" + k in g)
    print(f"[{name} {tag}] whole-gradient relative error (norm-weighted over the tensors) {(num / max(den, 1e-30)) ** 0.5:.4f}")
    v = sorted(rows.values())
    top = sorted(rows.items(), key=lambda kv: -kv[1])[:8]
    print(f"[{name} {tag}] median {v[len(v) // 2]:.4f} max {v[-1]:.4f}")
    print("    worst:", ", ".join(f"{k.replace('score_model.', '')} {x:.3f}" for k, x in top), flush=True)
    return {"median": v[len(v) // 2], "max": v[-1], "rel": rows}


res = {}
out = model({k: v.clone() for k, v in wd.items()})
for k in ("unorm_angles", "angles", "rigid_update", "rot_score", "trans_score"):
    ref = torch.tensor(g["out_" + k]).to(dev)
    print(f"forward {k}: all frames {rel_l2(out[k], ref):.5f}  last frame {rel_l2(out[k][-1], ref[-1]):.5f}")
loss, aux = experiment.loss_fn({k: v[None] for k, v in out.items()}, batch)
loss.backward()
res["engine"] = show("engine step, full loss", stats())

# (b) hybrid: d loss / d outputs evaluated at the reference's outputs
keys = [k for k in out if torch.is_tensor(out[k]) and out[k].requires_grad]
gold = {k: torch.tensor(g["out_" + k]).to(dev).to(out[k].dtype).reshape(out[k].shape).requires_grad_(k in keys) for k in out if "out_" + k in g}
lg, _ = experiment.loss_fn({k: v[None] for k, v in gold.items()}, batch)
print("loss at the reference outputs %.5f (golden %.5f), engine %.5f" % (float(lg), float(g["loss"]), float(loss)))
gk = [k for k in keys if k in gold]
grads = torch.autograd.grad(lg, [gold[k] for k in gk], allow_unused=True)
model.zero_grad(set_to_none=True)
out = model({k: v.clone() for k, v in wd.items()})          # (the tower's activations are consumed by a backward: run the forward again)
pairs = [(out[k], gr) for k, gr in zip(gk, grads) if gr is not None]
torch.autograd.backward([a for a, _ in pairs], [b for _, b in pairs])
res["hybrid"] = show("engine backward on the reference's output gradients", stats())

# (c) torsion-free
if "loss_notorsion" in g:
    model.zero_grad(set_to_none=True)
    out0 = model({k: v.clone() for k, v in wd.items()})
    l0, _ = experiment.loss_fn({k: v[None] for k, v in out0.items()}, batch, torsion_w=0.0)
    l0.backward()
    res["notorsion"] = show("engine step, torsion_loss_weight = 0", stats("g0sub_", "g0norm_"))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "r6_grad_split_" + name.replace(".npz", ".json")), "w") as fh:
    json.dump(res, fh, indent=1)
