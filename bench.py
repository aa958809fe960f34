#!/usr/bin/env python
"""Bench contract: `python bench.py --gpus N --steps K --warmup W`, one rank per GPU over RCCL: for N>1 either launched
by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (RANK / WORLD_SIZE in the environment) or
started bare, in which case it re-executes itself under that launcher with N ranks.  A "step" is one update_fn of the DFOLDv2 trajectory-prediction path (zero_grad, forward,
loss, backward, gradient all-reduce, Adam/amsgrad step) on one batch of synthetic trajectory windows per rank
(weak scaling).  Rank 0 prints ONE JSON line:

  metric = trajectory frames/s (fwd+bwd) at N_res=256; value = windows*frames of all ranks / max-over-ranks time;
  roofline = the dominant kernel (5x5 conv implicit GEMM, bf16 MFMA) vs the dense bf16 MFMA peak;
  cpu_baseline = the CPU oracle (port of the reference path) timed on this box's host cores on a bounded sample;
  last_frame_mode = extra, NOT the headline: the same update_fn in the engine's training-step mode (conv tower evaluated
                    on the dependency cone of the last frame only; identical loss / gradients).
The headline `value` always runs every frame through the conv tower, i.e. the work the reference does.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BF16_MFMA_PEAK_TFLOPS = 2500.0      # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"


def T(L):
    return 5 * L - 6


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--windows", type=int, default=8, help="trajectory windows per GPU (BASELINE config 3: 8)")
    ap.add_argument("--frames", type=int, default=32)
    ap.add_argument("--nres", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-frames", type=int, default=8)
    ap.add_argument("--selftest-dist", action="store_true",
                    help="only rendezvous (nccl with GPUs, gloo without), all-reduce one number, report n_gpus")
    ap.add_argument("--no-last-frame-mode", action="store_true", help="skip the second timed region (profiling runs)")
    ap.add_argument("--no-triangle", action="store_true", help="skip the triangle-operator extra object")
    ap.add_argument("--mode", choices=("all_frames", "last_frame"), default="all_frames",
                    help="what the MAIN timed region runs (profiling aid; the contract's headline is all_frames)")
    return ap.parse_args()


def make_batch(synthetic, diffuser, B, F, N, rank, dev):
    ws = [synthetic.synthetic_window(1000 * rank + i, F, N, t=0.5, diffuser=diffuser) for i in range(B)]
    batch = {k: torch.stack([w[k] for w in ws]).to(dev) for k in ws[0] if k != "t"}
    batch["t"] = torch.cat([w["t"] for w in ws]).to(dev)
    return batch


def conv_kernel_roofline(model, trainer, batch, B, F, N):
    """Average duration of the 5x5 conv implicit-GEMM launches (forward + dgrad: kernel dfold_mfma_gemm320_kernel<1>)
    of ONE extra, instrumented step, measured with HIP events on the stream the kernels are launched on."""
    from dynamicpdb_amd import ops
    events = []
    orig = ops.gemm

    def timed_gemm(*a, **kw):
        rows = kw.get("a_rows")
        is_conv = rows is not None and rows.mode == 1 and kw.get("seg_div_mid", 0) == 5
        if not is_conv:
            return orig(*a, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = orig(*a, **kw)
        e1.record()
        events.append((e0, e1))
        return r

    ops.gemm = timed_gemm
    try:
        trainer.update_fn(batch, step_optimizer=False)
        torch.cuda.synchronize()
    finally:
        ops.gemm = orig
    ms = [e0.elapsed_time(e1) for e0, e1 in events]
    avg_s = sum(ms) / len(ms) * 1e-3
    flops = 2.0 * 1280 * 640 * T(F) * T(N) * B        # algorithmic (non-padding taps), SURVEY 8(d): same for all 8 convs
    achieved = flops / avg_s / 1e12
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "r2_pmc_conv.json")     # HBM-side bytes per launch from the committed PMC passes
    if os.path.exists(pmc) and (B, F, N) == (8, 32, 256):
        with open(pmc) as fh:
            c = json.load(fh)
        traffic = int((2.0 * c["FETCH_SIZE_kb"] + c["WRITE_SIZE_kb"]) * 1024)
    return {"bound": "mfma", "achieved": round(achieved, 1), "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / BF16_MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
            "kernel": "dfold_mfma_gemm320_kernel<1, 5, true> (5x5 conv implicit GEMM, halo form, forward + dgrad launches)", "launches": len(ms),
            "avg_launch_ms": round(avg_s * 1e3, 4), "flop_per_launch": flops,
            "note": "peak = nominal dense bf16 MFMA rate at 2.4 GHz; the launch is power-limited on real operands: the same "
                    "binary on all-zero operands runs 1.85 PFLOP/s = 0.74 of the peak (scripts/exp_conv_dvfs.py), hipBLASLt "
                    "on the materialised GEMM of the same size 1.08 / 1.51 PFLOP/s (scripts/bench_conv.py library); "
                    "DESIGN.md section 5"}


def triangle_roofline(dev, reps=10):
    """Extra object (not the headline): forward of the north-star-named triangle operators at N_res 256 / 512 on this
    GPU: whole-call time (HIP events) vs the algorithmic bytes of SURVEY 8d (read z + write out + mask) over the 8 TB/s
    HBM peak.  Per-stage numbers and counters: scripts/bench_triangle.py, profiles/r2_triangle_*."""
    from dynamicpdb_amd.model import triangle as T_
    out = {}
    for name, ctor in (("tri_mul_out", lambda: T_.TriangleMultiplicationOutgoing(128, 128)),
                       ("tri_att_start", lambda: T_.TriangleAttentionStartingNode(128, 32, 4))):
        for n in (256, 512):
            torch.manual_seed(0)
            m = ctor().to(dev)
            z = torch.randn(1, n, n, 128, device=dev) * 1.5
            mask = (torch.rand(1, n, n, device=dev) > 0.05).float()
            with torch.no_grad():
                for _ in range(3):
                    m(z, mask=mask)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    m(z, mask=mask)
                e1.record()
                torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / reps * 1e-3
            alg = n * n * (2 * 128 * 4 + 4)
            out[f"{name}_n{n}"] = {"ms": round(t * 1e3, 4), "algorithmic_bytes": alg, "GBps": round(alg / t / 1e9, 1),
                                   "hbm_frac": round(alg / t / 8.0e12, 4)}
    out["note"] = ("forward, fp32 pair tensor, batch 1, fused kernels of csrc/pair_fused.hip; bound = hbm (8 TB/s); the calls are "
                   "VALU / issue bound, not HBM bound: counters in profiles/r2_triangle_pmc_*.txt, DESIGN.md section 4")
    return out


def cpu_baseline(F, N, seed_w=0):
    """The oracle (CPU port of the reference path; kind "port" -- the reference itself is not on the GPU box) running the
    reference's update_fn on host cores: zero_grad + forward + loss + backward + Adam(amsgrad) step
    (train_DFOLD_dynamics.py:660-667), ONE window of F frames x N_res = N.  Bounded sample: one cheap 2-frame iteration to
    warm up the thread pool / primitive caches, then one timed iteration at F frames."""
    from oracle import dfold_oracle as O
    from dynamicpdb_amd import synthetic
    from dynamicpdb_amd.data.se3_diffuser import SE3Diffuser
    try:
        cores = len(os.sched_getaffinity(0))      # cores this process may actually use (cgroup / affinity aware)
    except AttributeError:
        cores = os.cpu_count() or 1
    threads = max(1, min(cores, 128))
    torch.set_num_threads(threads)
    sd = synthetic.seeded_state_dict(seed_w)
    P = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    opt = torch.optim.Adam(list(P.values()), lr=1e-4, amsgrad=True)
    times = {}
    for frames in (2, F):
        conf = synthetic.default_conf(frames, cache_dir="/tmp/dfold_igso3_cache/")
        w = synthetic.synthetic_window(7, frames, N, t=0.5, diffuser=SE3Diffuser(conf.diffuser))
        t0 = time.time()
        opt.zero_grad(set_to_none=True)
        out = O.full_score_network(P, O.Schedules(), w)
        loss, _ = O.loss_fn(out, w)
        loss.backward()
        opt.step()
        times[frames] = time.time() - t0
        print(f"[bench cpu_baseline] {frames}-frame window: {times[frames]:.2f} s ({threads} threads)", file=sys.stderr, flush=True)
    t = times[F]
    return {"value": round(F / t, 4), "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"oracle update_fn (zero_grad+fwd+loss+bwd+Adam amsgrad), 1 window of {F} frames x N_res={N}, one timed "
                      f"iteration ({t:.2f} s) after a 2-frame warm-up iteration ({times[2]:.2f} s); torch CPU threads={threads}, "
                      f"{cores} usable cores"}


def respawn(args):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: become `python -m torch.distributed.run
    --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <same arguments>` (one rank per GPU)."""
    port = os.environ.get("MASTER_PORT", str(29400 + os.getpid() % 1000))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    os.execvpe(sys.executable, cmd, env)


def selftest_dist(args, world, rank):
    backend = "nccl" if torch.cuda.is_available() else "gloo"
    if not dist.is_initialized():
        dist.init_process_group(backend=backend)
    assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
    x = torch.tensor([float(rank + 1)], dtype=torch.float64)
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        x = x.cuda()
    dist.all_reduce(x)
    if rank == 0:
        print(json.dumps({"selftest": True, "n_gpus": dist.get_world_size(), "backend": backend, "allreduce_sum": float(x)}),
              flush=True)
    dist.barrier()
    dist.destroy_process_group()


def main():
    args = parse()
    T0 = time.perf_counter()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        respawn(args)                      # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} but the launcher started {world} rank(s)")
    if args.selftest_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world > 1:
            return selftest_dist(args, world, rank)
        print(json.dumps({"selftest": True, "n_gpus": 1}), flush=True)
        return
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl")
        assert dist.get_world_size() == args.gpus
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import __graft_entry__
    if rank == 0:
        __graft_entry__.build()
    if world > 1:
        dist.barrier()
    from dynamicpdb_amd import experiment, synthetic
    from dynamicpdb_amd.data.se3_diffuser import SE3Diffuser
    from dynamicpdb_amd.model.Dfold_network_dynamic import FullScoreNetwork
    B, F, N = args.windows, args.frames, args.nres
    tlog = lambda msg: print(f"[bench rank{rank} +{time.perf_counter() - T0:.1f}s] {msg}", file=sys.stderr, flush=True)
    conf = synthetic.default_conf(F, cache_dir="/tmp/dfold_igso3_cache/")
    diffuser = SE3Diffuser(conf.diffuser)
    model = FullScoreNetwork(conf.model, diffuser)
    model.load_state_dict(synthetic.seeded_state_dict(0), strict=True)     # same weights on every rank
    model.to(dev)
    # headline: every frame through the conv tower, the work the reference does (SURVEY 8d FLOP model)
    trainer = experiment.Trainer(model, lr=1e-4, last_frame_only=(args.mode == "last_frame"))
    tlog("model ready")
    batch = make_batch(synthetic, diffuser, B, F, N, rank, dev)
    tlog("batch ready")

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    first_loss = None
    for _ in range(args.warmup):
        l0, _ = trainer.update_fn(batch)
        torch.cuda.synchronize()
        first_loss = float(l0) if first_loss is None else first_loss
        tlog("warm-up step done")
    sync()
    t0 = time.perf_counter()
    losses = []
    for _ in range(args.steps):
        loss, aux = trainer.update_fn(batch)
        losses.append(loss)                      # device scalars: no host sync inside the timed region
    sync()
    elapsed = time.perf_counter() - t0
    if first_loss is None:
        first_loss = float(losses[0])
    tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    elapsed = float(tt)
    tlog(f"timed region done: {elapsed / args.steps * 1e3:.1f} ms/step")
    # the instrumented extra step contains the gradient all-reduce: EVERY rank runs it (a collective issued by rank 0
    # alone would pair with the other ranks' next step and hang the job at the end); rank 0 reports its own timings
    roof = conv_kernel_roofline(model, trainer, batch, B, F, N) if args.mode == "all_frames" else None
    # second timed region: the engine's training-step mode (Trainer default).  Loss, gradients and the optimizer update
    # are identical (tests/test_network_gpu.py::test_last_frame_only_training_mode_equals_full); the conv tower only
    # evaluates the dependency cone of the last frame, the one frame the live loss terms and frame updates read.
    el2 = None
    if not args.no_last_frame_mode and args.mode == "all_frames":
        trainer.last_frame_only = True
        trainer.update_fn(batch)
        sync()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            loss_l, _ = trainer.update_fn(batch)
        sync()
        el2 = torch.tensor([time.perf_counter() - t1], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(el2, op=dist.ReduceOp.MAX)
        el2 = float(el2)
        tlog(f"training-step mode (last-frame dependency cone): {el2 / args.steps * 1e3:.1f} ms/step")
    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = world * B * F * args.steps / elapsed
        tlog("roofline: %s" % json.dumps(roof))
        line = {
            "metric": "trajectory_frames_per_sec_fwd_bwd_nres%d" % N, "value": round(value, 2), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "BASELINE config 3: synthetic N_res=%d, %d-frame windows, %d windows/GPU, full "
                                   "update_fn (fwd+loss+bwd+grad all-reduce+Adam amsgrad), random-init seeded weights"
                                   % (N, F, B),
                       "windows_per_gpu": B, "frames": F, "n_res": N, "parallelism": "dp%d" % world, "mode": args.mode},
            # the same batch every step: the loss of the first step taken (warm-up included) and of the last timed step
            # show the optimizer descending (every forward sees the parameters the previous step wrote)
            "loss": {"first_step": round(first_loss, 5), "last_step": round(float(loss), 5),
                     "steps_between": args.warmup + args.steps - 1},
            "roofline": roof,
            "last_frame_mode": None if el2 is None else {
                "value": round(world * B * F * args.steps / el2, 2), "unit": "frames/s",
                "ms_per_step": round(el2 / args.steps * 1e3, 3), "steps": args.steps,
                "note": "NOT the headline: same update_fn with Trainer(last_frame_only=True) -- conv tower evaluated on the "
                        "dependency cone of the last frame only (the only frame the live loss terms and frame updates read); "
                        "loss, gradients and parameter update identical to the all-frames step (bit-exact conv results), "
                        "4x fewer conv FLOPs at F=32"},
        }
        if world == 1 and not args.no_triangle:
            line["triangle"] = triangle_roofline(dev)
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(max(2, args.cpu_baseline_frames), N)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
