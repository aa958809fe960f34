"""diagnostic: ConvTower last-frame cone vs full, per layer / per tap."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from util import rel_l2
from dynamicpdb_amd import ops
dev = torch.device("cuda:0")
Wn, F, N, C = 2, 20, 16, 128
gen = torch.Generator(device="cpu").manual_seed(9)
ws = [(torch.randn(co, ci, 5, 5, generator=gen) * (2.0 / (25 * ci)) ** 0.5).to(dev) for _ in range(4) for (co, ci) in ((C // 2, C), (C, C // 2))]
bs = [(torch.randn(w.shape[0], generator=gen) * 0.1).to(dev) for w in ws]
x = torch.randn(Wn, F, N, C, generator=gen).to(dev).to(torch.bfloat16)
gy = torch.zeros(Wn, F, N, C, device=dev, dtype=torch.bfloat16)
gy[:, -1] = torch.randn(Wn, N, C, generator=gen).to(dev).to(torch.bfloat16)
g = ops.Grid(Wn, F, N, dev)
res = {}
for mode in (False, True):
    tower = ops.ConvTower(ws, bs)
    tower.pack(); tower.zero_grad()
    h0 = g.alloc(C); g.interior(h0).copy_(x)
    h4, saved = tower.forward(g, h0, last_frame_only=mode)
    gt = g.alloc(C); g.interior(gt).copy_(gy)
    g0 = tower.backward(g, saved, gt, last_frame_only=mode)
    res[mode] = ([s.float().clone() for s in saved], [d.clone() for d in tower.dwg], [d.clone() for d in tower.db],
                 {k[0]: v.float().clone().view(k[1]) for k, v in tower.ws.bufs.items() if k[0] in ("dv", "du", "g0", "g1")})
full, last = res[False], res[True]
for i in range(8):
    a, b = last[1][i], full[1][i]
    print("layer", i, "dW rel", rel_l2(a, b), "db rel", rel_l2(last[2][i], full[2][i]), tuple(a.shape))
    if rel_l2(a, b) > 1e-3:
        for t in range(25):
            print("   tap", t // 5, t % 5, "%.4f" % rel_l2(a[:, t], b[:, t]), end=";")
        print()
for i in range(4):
    (l1, n1), (l2, n2) = ops.ConvTower.cone(F, i)
    u_l, u_f = last[0][3 * i + 1], full[0][3 * i + 1]
    print("block", i, "u in-range equal:", torch.equal(u_l[:, 2 + l1:2 + F], u_f[:, 2 + l1:2 + F]), "hn in-range:",
          torch.equal(last[0][3 * i + 3][:, 2 + l2:2 + F], full[0][3 * i + 3][:, 2 + l2:2 + F]))
for k in ("dv", "du", "g0", "g1"):
    a, b = last[3][k], full[3][k]
    nzf = [int(f) for f in range(F + 4) if float(b[:, f].abs().max()) > 0]
    nzl = [int(f) for f in range(F + 4) if float(a[:, f].abs().max()) > 0]
    print(k, "full nonzero padded frames", nzf[:1], nzf[-1:], "last", nzl[:1], nzl[-1:], "rel", rel_l2(a, b))
