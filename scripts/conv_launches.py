"""The two production conv launches (BASELINE config 3 grid: 8 x 32 x 256 cells, 1280 -> 640 and 640 -> 1280 channels) on dense random
operands, ten times each: the workload of the SQ counter passes of scripts/pmc_conv_sq.sh."""
import sys
import os
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dynamicpdb_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = ops.Grid(8, 32, 256, dev)
for CI, CO in ((1280, 640), (640, 1280)):
    x = g.alloc(CI)
    g.interior(x).copy_(torch.randn(8, 32, 256, CI, device=dev).to(torch.bfloat16))
    wf = (torch.randn(CO, 25, CI, device=dev) / np.sqrt(25 * CI)).to(torch.bfloat16)
    out = g.alloc(CO)
    for _ in range(10):
        ops.conv5x5_fwd(g, x, wf, torch.zeros(CO, device=dev), out, relu=True)
torch.cuda.synchronize()
print("done")
