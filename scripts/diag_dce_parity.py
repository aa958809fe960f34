"""rel-L2 of the forward outputs against the reference-minted goldens with the trunk's dead-code elimination on and off (diagnostic)."""
import os
import sys
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import load_golden, golden_window, rel_l2  # noqa: E402
import test_parity_baseline_gpu as T  # noqa: E402

dev = torch.device("cuda:0")
for name in sys.argv[1:] or ["network_F16_N96.npz", "network_F3_N16.npz"]:
    g = load_golden(name)
    w, (F, N, seed_w, stride) = golden_window(g)
    model, _ = T._build(F, seed_w, dev)
    wd = {k: v.to(dev) for k, v in w.items()}
    for dce in (True, False):
        model.score_model.trunk_dce = dce
        with torch.no_grad():
            out = model({k: v.clone() for k, v in wd.items()})
        raw = torch.tensor(g["out_unorm_angles"]).to(dev)
        nrm = raw.norm(dim=-1)
        well = nrm > 0.5 * nrm.pow(2).mean().sqrt()
        a, ga = out["angles"], torch.tensor(g["out_angles"]).to(dev)
        print(name, "dce", dce, {k: round(rel_l2(out[k], g["out_" + k]), 5) for k in ("angles", "unorm_angles", "rigid_update", "rot_score")},
              "angles on well-conditioned torsions", round(rel_l2(a[well], ga[well].cpu().numpy()), 5), float(well.float().mean()))
