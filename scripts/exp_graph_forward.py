"""Can the drop-in network's forward be captured in a HIP graph (torch.cuda.CUDAGraph)?  Eager vs replay time at BASELINE config 1."""
import sys
import time
import numpy as np
import torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dynamicpdb_amd import experiment, synthetic
from dynamicpdb_amd.data.se3_diffuser import SE3Diffuser
from dynamicpdb_amd.model.Dfold_network_dynamic import FullScoreNetwork

dev = torch.device("cuda:0")
F, N = 16, 96
conf = synthetic.default_conf(F, cache_dir="/tmp/dfold_igso3_cache/")
diffuser = SE3Diffuser(conf.diffuser)
model = FullScoreNetwork(conf.model, diffuser)
model.load_state_dict(synthetic.seeded_state_dict(31), strict=True)
model.to(dev).eval()
w = synthetic.synthetic_window(32, F, N, t=1.0, diffuser=None)
np.random.seed(90)
prior = diffuser.sample_ref(n_samples=F * N, as_tensor_7=True)["rigids_t"].reshape(F, N, 7).float()
feats = {k: v.to(dev) for k, v in w.items()}
feats["rigids_t"] = prior.to(dev)
feats = experiment.set_t_feats(diffuser, feats, 0.5, torch.ones(1, device=dev))
feats["sc_ca_t"] = feats["rigids_t"][..., 4:].clone()


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


with torch.no_grad():
    ref = model(dict(feats))
    print("eager forward ms", round(timed(lambda: model(dict(feats))), 3))
    static = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in feats.items()}
    print("t_host" in static)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            model(dict(static))
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g):
            out = model(dict(static))
    except Exception as e:
        import traceback
        tb = traceback.format_exc().splitlines()
        print("capture failed:", type(e).__name__, str(e)[:200])
        print("\n".join(l[:200] for l in tb if "File" in l or "dynamicpdb" in l)[-3000:])
        sys.exit(0)
    g.replay()
    torch.cuda.synchronize()
    for k in ("rigids", "rot_score", "trans_score", "angles", "atom37"):
        print(k, "replay vs eager max abs", float((out[k].double() - ref[k].double()).abs().max()))
    print("graph replay ms", round(timed(g.replay), 3))
    # new inputs through the static buffers
    static["rigids_t"].copy_(static["rigids_t"].roll(1, 0))
    g.replay()
    chk = model({**feats, "rigids_t": feats["rigids_t"].roll(1, 0)})
    torch.cuda.synchronize()
    print("after input update: out nan", bool(torch.isnan(out["rigids"]).any()), "eager nan", bool(torch.isnan(chk["rigids"]).any()),
          "max abs vs eager", float((out["rigids"] - chk["rigids"]).abs().max()))
    rolled = {**feats, "rigids_t": feats["rigids_t"].roll(1, 0)}
    e1 = model(dict(rolled))["rigids"].clone()
    e2 = model(dict(rolled))["rigids"].clone()
    g.replay()
    r1 = out["rigids"].clone()
    g.replay()
    r2 = out["rigids"].clone()
    torch.cuda.synchronize()
    print("eager vs eager", float((e1 - e2).abs().max()), "replay vs replay", float((r1 - r2).abs().max()), "replay vs eager", float((r1 - e1).abs().max()))
    print("static rigids_t == rolled", bool(torch.equal(static["rigids_t"], rolled["rigids_t"])))
    for k in static:
        if torch.is_tensor(static[k]) and k != "rigids_t" and not torch.equal(static[k], feats[k]):
            print("static differs from feats:", k)
