"""Times the two forms of the 5x5 conv weight gradient on the config-3 grid (8 windows x 32 frames x N_res 256):
the copy form (dfold_grid_transpose_shift x 2 + dfold_mfma_gemm320_kernel role 2) and the direct form
(conv_wgrad_tn_kernel, + one column-sum pass for the bias gradient).   python scripts/bench_wgrad.py [--reps 10]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dynamicpdb_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--n", type=int, default=256)
    ap.add_argument("--frames", type=int, default=32)
    ap.add_argument("--windows", type=int, default=8)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    g = ops.Grid(a.windows, a.frames, a.n, dev)
    gen = torch.Generator(device="cpu").manual_seed(0)
    res = {}
    for (CI, CO) in ((1280, 640), (640, 1280)):
        x, gy = g.alloc(CI), g.alloc(CO)
        g.interior(x).copy_(torch.randn(a.windows, a.frames, a.n, CI, generator=gen).clamp_min(0).to(torch.bfloat16))
        g.interior(gy).copy_((torch.randn(a.windows, a.frames, a.n, CO, generator=gen) * 0.1).to(torch.bfloat16))
        dwg = torch.zeros((1280, 25, 640), dtype=torch.float32, device=dev)
        db = torch.zeros(CO, dtype=torch.float32, device=dev)
        ws = ops.Workspace(dev)
        flop = 2.0 * 25 * CI * CO * g.M
        for tn in (False, True):
            for _ in range(2):
                ops.conv5x5_wgrad(g, x, gy, dwg, ws, accumulate=True, bias_grad=db, tn=tn)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                ops.conv5x5_wgrad(g, x, gy, dwg, ws, accumulate=True, bias_grad=db, tn=tn)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.reps
            res[f"{CI}->{CO} {'direct' if tn else 'copies'}"] = {"ms": round(ms, 4), "tflops": round(flop / ms / 1e9, 1)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
