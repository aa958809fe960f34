#!/bin/bash
# what does the epilogue of the 512 x 160 conv kernel cost?  the two production launches with and without it (diagnostic library:
#   SRC=conv_fwd_w4 bash scripts/build_variant.sh noepi -DW4_NO_EPI   in the build container, before the gpurun call)
set -u
cd "${GRAFT_REPO_ROOT:-$PWD}"
cat > /tmp/tconv.py <<'PY'
import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
from dynamicpdb_amd import ops
dev = torch.device("cuda:0")
g = ops.Grid(8, 32, 256, dev)
for CI, CO in ((1280, 640), (640, 1280)):
    x = g.alloc(CI)
    g.interior(x).copy_(torch.relu(torch.randn(8, 32, 256, CI, device=dev)).to(torch.bfloat16))
    wf = (torch.randn(CO, 25, CI, device=dev) / np.sqrt(25 * CI)).to(torch.bfloat16)
    out = g.alloc(CO)
    r = g.alloc(CO)
    b = torch.zeros(CO, device=dev)
    for name, kw in (("bias+relu", dict(relu=True)), ("resid", dict(relu=False, resid=r))):
        f = lambda: ops.conv5x5_fwd(g, x, wf, b, out, **kw)
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        print(os.environ.get("DFOLD_LIB", "production")[-20:], CI, CO, name, round(e0.elapsed_time(e1) / 10, 4), "ms")
PY
for k in 1 2; do
python /tmp/tconv.py
DFOLD_LIB=$PWD/dynamicpdb_amd/csrc/variants/libdfold_noepi.so python /tmp/tconv.py
done
