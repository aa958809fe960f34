"""A/B of the conv kernel's wave-priority schemes (DFOLD_GEMM_PRIO, read once per process): one subprocess per value."""
import json, os, subprocess, sys
CODE = r'''
import json, sys, numpy as np, torch
sys.path.insert(0, ".")
from dynamicpdb_amd import ops
from scripts.bench_conv import timeit
dev = torch.device("cuda:0")
Wn, F, N, C = 8, 32, 256, 1280
g = ops.Grid(Wn, F, N, dev)
T = lambda L: 5 * L - 6
flop = 2 * C * (C // 2) * T(F) * T(N) * Wn
res = {}
for CI, CO in ((1280, 640), (640, 1280)):
    x = g.alloc(CI); g.interior(x).copy_(torch.randn(Wn, F, N, CI, device=dev, generator=torch.Generator(device=dev).manual_seed(1)).to(torch.bfloat16))
    wf = (torch.randn(CO, 25, CI, device=dev, generator=torch.Generator(device=dev).manual_seed(2)) / np.sqrt(25 * CI)).to(torch.bfloat16)
    out = g.alloc(CO)
    t = timeit(lambda: ops.conv5x5_fwd(g, x, wf, torch.zeros(CO, device=dev), out, relu=True), iters=10, warm=3)
    res[f"{CI}->{CO}"] = round(flop / t / 1e12, 1)
    res[f"sum{CI}"] = int(out.view(torch.int16).to(torch.int64).sum())       # bit-level checksum of the output grid
print(json.dumps(res))
'''
for prio in sys.argv[1:] or ["0", "1", "2"]:
    env = dict(os.environ, DFOLD_GEMM_PRIO=prio)
    r = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True, timeout=120)
    print("prio", prio, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)
