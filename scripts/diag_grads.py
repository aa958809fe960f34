import sys; sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch, numpy as np
from util import load_golden, window_from_golden, rel_l2
from test_network_gpu import _build
from dynamicpdb_amd import experiment
dev = torch.device("cuda:0")
g = load_golden("network_F3_N16.npz")
F, N, seed_w, _ = [int(v) for v in g["meta"]]
model, _ = _build(F, seed_w, dev)
w = window_from_golden(g, dev)
out = model({k: v.clone() for k, v in w.items()})
batch = {k: v[None] for k, v in w.items()}; batch["t"] = w["t"].reshape(1)
loss, aux = experiment.loss_fn({k: v[None] for k, v in out.items()}, batch)
loss.backward()
P = dict(model.named_parameters())
rows = []
for k in g:
    if not k.startswith("gsub_"): continue
    name = k[5:]; gr = P[name].grad; rn = float(g["gnorm_" + name])
    if rn < 1e-6 or gr is None: continue
    ref = torch.tensor(g[k]).double()
    mine = (gr.reshape(-1)[::9973] if gr.numel() > 70000 else gr).double().cpu().reshape(ref.shape)
    rows.append((float((mine - ref).norm() / (ref.norm() + 1e-30)), abs(float(gr.double().norm()) - rn) / rn, name, rn))
rows.sort(reverse=True)
for r in rows: print("%.4f %.4f %-60s %.3e" % r)
for k in ("angles","unorm_angles","rigid_update","rot_score","trans_score"):
    print(k, rel_l2(out[k], g["out_"+k]))

# same comparison against the oracle with bf16-operand emulation (separates kernel defects from ReLU-mask flips)
from oracle import dfold_oracle as O
from dynamicpdb_amd import synthetic
O.EMULATE_BF16_OPERANDS = True
Pq = {k: v.clone().requires_grad_(True) for k, v in synthetic.seeded_state_dict(seed_w).items()}
wc = {k: v.cpu() for k, v in w.items()}
outq = O.full_score_network(Pq, O.Schedules(), wc)
lq, _ = O.loss_fn(outq, wc)
lq.backward()
rows = []
for name, p in P.items():
    if p.grad is None or Pq[name].grad is None: continue
    a, b = p.grad.double().cpu(), Pq[name].grad.double()
    if float(b.norm()) < 1e-6: continue
    rows.append((float((a - b).norm() / b.norm()), name))
rows.sort(reverse=True)
print("vs bf16-emulating oracle: loss", float(loss), float(lq))
for r in rows[:12]: print("%.4f %s" % r)
print("median %.4f" % sorted(r[0] for r in rows)[len(rows)//2])
