"""Is the conv launch power / clock limited?  Same launch on random operands, on zero activations, and on all-zero
operands (identical instruction stream and memory traffic; only the toggling in the datapaths changes).
    python scripts/exp_conv_dvfs.py"""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from dynamicpdb_amd import ops  # noqa: E402
from scripts.bench_conv import timeit  # noqa: E402

dev = torch.device("cuda:0")
Wn, F, N = 8, 32, 256
g = ops.Grid(Wn, F, N, dev)
for CI, CO in ((1280, 640), (640, 1280)):
    fl = 2.0 * g.M * CO * 25 * CI
    res = {}
    for name, xs, ws in (("random", 1.0, 1.0), ("zero_x", 0.0, 1.0), ("zero_all", 0.0, 0.0), ("random_again", 1.0, 1.0)):
        x = g.alloc(CI)
        g.interior(x).copy_((torch.randn(Wn, F, N, CI, device=dev) * xs).to(torch.bfloat16))
        wf = (torch.randn(CO, 25, CI, device=dev) * ws / np.sqrt(25 * CI)).to(torch.bfloat16)
        out = g.alloc(CO)
        t = timeit(lambda: ops.conv5x5_fwd(g, x, wf, torch.zeros(CO, device=dev), out, relu=True), iters=20, warm=5)
        res[name] = round(fl / t / 1e12, 1)
    print(json.dumps({"conv": f"{CI}->{CO}", "tflops_issued": res}), flush=True)
