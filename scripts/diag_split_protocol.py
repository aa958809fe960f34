"""split-K / stream-K of the one-wave-per-SIMD conv kernel against its unsplit launch: where do the results differ (diagnostic)"""
import os
import sys
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from ctypes import c_int32  # noqa: E402
from dynamicpdb_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
gen = torch.Generator(device="cpu").manual_seed(21)
Wn, F, N, CI, CO = 8, 6, 256, 640, 640
g = ops.Grid(Wn, F, N, dev)
w = (torch.randn(CO, CI, 5, 5, generator=gen) * (2.0 / (25 * CI)) ** 0.5).to(dev)
wf = torch.empty((CO, 25, CI), dtype=torch.bfloat16, device=dev)
wd = torch.empty((CI, 25, CO), dtype=torch.bfloat16, device=dev)
_lib.check(_lib.lib().dfold_conv_weight_pack(ops._p(w), ops._p(wf), ops._p(wd), c_int32(CO), c_int32(CI), _lib.stream()), "pack")
bias = (0.1 * torch.randn(CO, generator=gen)).to(dev)
x = g.alloc(CI)
g.interior(x).copy_(torch.randn(Wn, F, N, CI, generator=gen).to(dev).to(torch.bfloat16))
ws = ops.Workspace(dev)
real = ops.conv_splitk
real(65536, 1280, 640, dev)
outs = {}
for S in (1, 5, -1):
    ops.conv_splitk = lambda *a, _S=S, **k: _S
    o = g.alloc(CO)
    ops.conv5x5_fwd(g, x, wf, bias, o, relu=False, f_lo=F - 2, nf=2, ws=ws)
    torch.cuda.synchronize()
    outs[S] = g.interior(o)[:, F - 2:].float().reshape(-1, CO)       # [4096 rows, 640]
ref = outs[1]
for S in (5, -1):
    d = (outs[S] - ref).abs()
    rel = float(d.norm() / ref.norm())
    bad = d > 0.05
    rows = bad.any(1).nonzero().flatten()
    cols = bad.any(0).nonzero().flatten()
    print("FINE_WS", os.environ.get("DFOLD_CONV_FINE_WS", "1"), "S", S, "rel", round(rel, 5), "bad rows", len(rows), "cols", len(cols),
          "row%128 hist", torch.bincount((rows % 128) // 32, minlength=4).tolist() if len(rows) else None,
          "col%160 //32 hist", torch.bincount((cols % 160) // 32, minlength=5).tolist() if len(cols) else None,
          "ratio", float((outs[S][bad] / ref[bad]).median()) if bad.any() else None)
