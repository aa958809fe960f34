"""Phase timestamps (s_memtime) of one workgroup of the register-resident triangle attention (csrc/triatt_reg.hip, argument
dbg_phase_clock of dfold_triatt_reg_fwd): cycles since the wave's start at every phase boundary, per wave."""
import os, sys
os.environ["DFOLD_TRIATT_ROW"] = "3"
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from dynamicpdb_amd.model import triangle as T
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = T.TriangleAttentionStartingNode(128, 32, 4).to(dev)
x = torch.randn(8, N, N, 128, device=dev) * 1.5
mask = torch.ones(8, N, N, device=dev)
dbg = torch.zeros(4 * N * 32 + 8 * 64, device=dev)
T._TRIATT_DBG = dbg
T._TRIATT_PHASE_CLOCK = True
with torch.no_grad():
    for _ in range(3):
        m(x, mask=mask)
torch.cuda.synchronize()
ts = dbg[4 * N * 32:].view(8, 64).cpu()
names = ["start", "LN"] + sum([[f"h{h} start", f"h{h} wts", f"h{h} proj", f"h{h} barrier", f"h{h} att"] for h in range(4)], []) + ["Wo", "out"]
for w in range(N // 64):
    row = ts[w, :len(names)].tolist()
    print("wave", w, " ".join(f"{n}={int(v)}" for n, v in zip(names, row)))
    d = [row[i] - row[i - 1] for i in range(1, len(row))]
    print("   deltas", " ".join(f"{n}:{int(v)}" for n, v in zip(names[1:], d)))
