"""Attribute the torch-native ("glue") kernels of one update_fn to the Python lines that launch them.

    python scripts/glue_trace.py [--mode all_frames|last_frame] > gpurun_out/glue_trace.txt

One warm-up step, then one step under torch.profiler (record_shapes, with_stack).  Every aten op that launched a device
kernel is listed with its total device time, call count, input shapes and the innermost frame inside dynamicpdb_amd/."""
import argparse
import collections
import os
import sys

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="all_frames")
    ap.add_argument("--windows", type=int, default=8)
    ap.add_argument("--frames", type=int, default=32)
    ap.add_argument("--nres", type=int, default=256)
    a = ap.parse_args()
    import bench
    from dynamicpdb_amd import experiment, synthetic
    from dynamicpdb_amd.data.se3_diffuser import SE3Diffuser
    from dynamicpdb_amd.model.Dfold_network_dynamic import FullScoreNetwork
    dev = torch.device("cuda:0")
    conf = synthetic.default_conf(a.frames, cache_dir="/tmp/dfold_igso3_cache/")
    diffuser = SE3Diffuser(conf.diffuser)
    model = FullScoreNetwork(conf.model, diffuser)
    model.load_state_dict(synthetic.seeded_state_dict(0), strict=True)
    model.to(dev)
    trainer = experiment.Trainer(model, lr=1e-4, last_frame_only=(a.mode == "last_frame"))
    batch = bench.make_batch(synthetic, diffuser, a.windows, a.frames, a.nres, 0, dev)
    for _ in range(2):
        trainer.update_fn(batch)
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        trainer.update_fn(batch)
        torch.cuda.synchronize()
    agg = collections.defaultdict(lambda: [0.0, 0])
    for ev in prof.events():
        dt = getattr(ev, "self_device_time_total", None)
        if dt is None:
            dt = getattr(ev, "self_cuda_time_total", 0.0)
        if not dt or not ev.name.startswith("aten::"):
            continue
        where = "?"
        for fr in (ev.stack or []):
            if "dynamicpdb_amd" in fr or "bench.py" in fr:
                where = fr.replace(ROOT + "/", "")
                break
        shapes = str(ev.input_shapes)[:90]
        k = (ev.name, where, shapes)
        agg[k][0] += dt
        agg[k][1] += 1
    rows = sorted(agg.items(), key=lambda kv: -kv[1][0])
    tot = sum(v[0] for _, v in rows)
    print(f"aten device time in one step: {tot / 1e3:.2f} ms")
    for (name, where, shapes), (t, n) in rows[:70]:
        print(f"{t / 1e3:8.3f} ms {n:5d}  {name:28s} {where[:80]:80s} {shapes}")


if __name__ == "__main__":
    main()
