"""Per-shape time of every contraction launched by one update_fn (HIP events around ops.gemm; diagnostic).
    python scripts/gemm_trace.py [--mode all_frames|last_frame]"""
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
    a = ap.parse_args()
    import bench
    from dynamicpdb_amd import experiment, ops, synthetic
    from dynamicpdb_amd.data.se3_diffuser import SE3Diffuser
    from dynamicpdb_amd.model.Dfold_network_dynamic import FullScoreNetwork
    dev = torch.device("cuda:0")
    conf = synthetic.default_conf(32, cache_dir="/tmp/dfold_igso3_cache/")
    diffuser = SE3Diffuser(conf.diffuser)
    model = FullScoreNetwork(conf.model, diffuser)
    model.load_state_dict(synthetic.seeded_state_dict(0), strict=True)
    model.to(dev)
    trainer = experiment.Trainer(model, lr=1e-4, last_frame_only=(a.mode == "last_frame"))
    batch = bench.make_batch(synthetic, diffuser, 8, 32, 256, 0, dev)
    for _ in range(2):
        trainer.update_fn(batch)
    torch.cuda.synchronize()
    ev = []
    orig = ops.gemm

    def timed(A, B, C, M, N, K, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = orig(A, B, C, M, N, K, **kw)
        e1.record()
        nseg = kw.get("nseg", 1)
        key = (M, N, K * nseg, kw.get("nbatch", 1), str(C.dtype).replace("torch.", ""), "conv" if kw.get("seg_div_mid", 0) == 5 else
               ("wgrad" if kw.get("nb1", 1) == 5 and kw.get("nbatch", 1) == 25 else ""), kw.get("flags", 0))
        ev.append((key, e0, e1))
        return r
    ops.gemm = timed
    orig_tn = ops.gemm_tn

    def timed_tn(A, B, C, M, N, K, *a_, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = orig_tn(A, B, C, M, N, K, *a_, **kw)
        e1.record()
        ev.append(((M, N, K, kw.get("nbatch", 1), str(C.dtype).replace("torch.", ""), "tn", kw.get("flags", 0)), e0, e1))
        return r
    ops.gemm_tn = timed_tn
    import dynamicpdb_amd.model.functional as F_
    import dynamicpdb_amd.model.triangle as T_
    for mod in (F_, T_):
        if getattr(mod, "gemm", None) is orig:
            mod.gemm = timed
    # the bf16 helpers around the contractions (transposes, column sums, casts), keyed by shape and CALL SITE
    import inspect
    ev2 = []

    def wrap_helper(name, keyfn):
        orig_h = getattr(ops, name)

        def timed_h(*a_, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            fr = inspect.stack()[1]
            e0.record()
            r = orig_h(*a_, **kw)
            e1.record()
            ev2.append(((name, keyfn(*a_, **kw), f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.function}"), e0, e1))
            return r
        setattr(ops, name, timed_h)
        for mod in (F_, T_):
            if getattr(mod, name, None) is orig_h:
                setattr(mod, name, timed_h)
    wrap_helper("transpose_bf16", lambda src, R, C, **kw: (R, C, kw.get("nbatch", 1)))
    wrap_helper("colsum_bf16", lambda x, out, R, C, ld: (R, C, ld))
    wrap_helper("cast_bf16", lambda x: tuple(x.shape))
    wrap_helper("cast_f32", lambda x: tuple(x.shape))
    trainer.update_fn(batch)
    torch.cuda.synchronize()
    agg2 = collections.defaultdict(lambda: [0.0, 0])
    for key, e0, e1 in ev2:
        agg2[key][0] += e0.elapsed_time(e1)
        agg2[key][1] += 1
    print(f"helpers in one step: {sum(v[0] for v in agg2.values()):.2f} ms over {len(ev2)} launches")
    for key, (t, n) in sorted(agg2.items(), key=lambda kv: -kv[1][0])[:30]:
        print(f"{t:7.3f} {n:4d}  {key[0]:15s} {str(key[1]):28s} {key[2]}")
    agg = collections.defaultdict(lambda: [0.0, 0])
    for key, e0, e1 in ev:
        agg[key][0] += e0.elapsed_time(e1)
        agg[key][1] += 1
    tot = sum(v[0] for v in agg.values())
    print(f"contractions in one step: {tot:.2f} ms over {len(ev)} launches")
    print("    ms  calls   us/call  TFLOP/s   M      N      K     batch dtype kind flags")
    for key, (t, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:45]:
        M, N, K, nb, dt, kind, fl = key
        tf = 2.0 * M * N * K * nb * n / (t * 1e-3) / 1e12
        print(f"{t:7.2f} {n:5d} {t / n * 1e3:9.1f} {tf:8.1f}  {M:6d} {N:6d} {K:6d} {nb:5d} {dt:9s} {kind:5s} {fl}")


if __name__ == "__main__":
    main()
