"""One small invocation of the hot path on cuda:0 (forward + loss + backward of a 1-window step), checked
against the CPU oracle.  Called by __graft_entry__.smoke()."""
import os
import sys

import torch


def run(F=3, N=16, seed_w=0):
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    if root not in sys.path:
        sys.path.insert(0, root)
    from oracle import dfold_oracle as O          # checker only
    from . import experiment, synthetic
    from .data.se3_diffuser import SE3Diffuser
    from .model.Dfold_network_dynamic import FullScoreNetwork
    dev = torch.device("cuda:0")
    conf = synthetic.default_conf(F, cache_dir="/tmp/dfold_igso3_cache/")
    diffuser = SE3Diffuser(conf.diffuser)
    model = FullScoreNetwork(conf.model, diffuser)
    sd = synthetic.seeded_state_dict(seed_w)
    model.load_state_dict(sd, strict=True)
    model.to(dev)
    w = synthetic.synthetic_window(1, F, N, t=0.5, diffuser=diffuser)
    batch = {k: v[None].to(dev) for k, v in w.items()}
    batch["t"] = w["t"].to(dev)
    out = model(batch)
    loss, aux = experiment.loss_fn(out, batch)
    loss.backward()
    torch.cuda.synchronize()
    loss = loss.detach()
    ref = O.full_score_network(sd, O.Schedules(), w)
    ref_loss, _ = O.loss_fn(ref, w)
    ref_loss = ref_loss.detach()
    err = float((out["rigids"][0, ..., 4:].detach().cpu() - ref["rigids"][..., 4:].detach()).abs().max())
    rel = abs(float(loss) - float(ref_loss)) / abs(float(ref_loss))
    gn = sum(float(p.grad.double().norm() ** 2) for p in model.parameters() if p.grad is not None) ** 0.5
    print(f"smoke: loss {float(loss):.5f} (oracle {float(ref_loss):.5f}, rel {rel:.2e}), max |dtrans| {err:.2e} A, "
          f"grad norm {gn:.4e}")
    assert rel < 3e-2 and err < 5e-2 and gn > 0 and gn == gn
    return float(loss)
