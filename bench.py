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
The headline `value` is the product's default step: every frame through the model, every output of every frame produced;
the conv tower of the trunk's two inner blocks -- whose output the reference's own forward multiplies by 0.0 off the last
frame -- runs on that frame's dependency cone (config.trunk_dead_code_elimination; `all_positions_mode` times the step
without it), and the backward launches skip, on the device, the frames whose incoming gradient is zero (config.zero_frame_skipping).
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
    ap.add_argument("--lr", type=float, default=1e-6,
                    help="Adam learning rate of the timed steps.  The reference trains with 1e-4 (config/train_DFOLDv2.yaml); on "
                         "random-init weights and synthetic frames that rate pushes the frame update past the reference's "
                         "`trans_loss < 100` gate (train_DFOLD_dynamics.py:1338-1340) within a few steps, after which the rot / "
                         "trans terms are zeroed and the timed steps would run a torsion-only loss.  The default keeps the gate "
                         "open for the whole run (same arithmetic per step: the rate is one scalar of the fused Adam launch); "
                         "`loss.terms_*` on the line show which regime was timed")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-frames", type=int, default=32, help="window length of the CPU leg (default: the bench's own 32 "
                    "frames: one warm-up + 2 timed iterations, about 4 minutes of host time)")
    ap.add_argument("--selftest-dist", action="store_true",
                    help="only rendezvous (nccl with GPUs, gloo without), all-reduce one number, report n_gpus")
    ap.add_argument("--no-last-frame-mode", action="store_true", help="skip the second timed region (profiling runs)")
    ap.add_argument("--no-all-positions-mode", action="store_true",
                    help="skip the third timed region (trunk without dead-code elimination; profiling runs)")
    ap.add_argument("--no-triangle", action="store_true", help="skip the triangle-operator extra object")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the config 2 / config 5 extra objects")
    ap.add_argument("--no-eval-config", action="store_true", help="skip the config 1 (eval / sampling) extra object")
    ap.add_argument("--no-neighbours", action="store_true", help="skip the pair-block / dataset-transform / per-kernel HBM extra objects")
    ap.add_argument("--same-batch", action="store_true", help="feed one batch to every step (profiling aid; default: a "
                    "fresh synthetic batch per step, generated on the device and staged in HBM before the timed region)")
    ap.add_argument("--mode", choices=("all_frames", "last_frame"), default="all_frames",
                    help="what the MAIN timed region runs (profiling aid; the contract's headline is all_frames)")
    return ap.parse_args()


def make_batch(synthetic, diffuser, B, F, N, rank, dev):
    """host-generated batch (numpy stream; the generator the golden vectors use)"""
    ws = [synthetic.synthetic_window(1000 * rank + i, F, N, t=0.5, diffuser=diffuser) for i in range(B)]
    batch = {k: torch.stack([w[k] for w in ws]).to(dev) for k in ws[0] if k != "t"}
    batch["t"] = torch.cat([w["t"] for w in ws]).to(dev)
    return batch


def make_batches(synthetic, diffuser, B, F, N, rank, dev, count, same):
    """`count` batches resident in HBM before any timed region starts: a FRESH synthetic batch per step (device Philox
    stream seeded by (rank, step): the operand sparsity the conv kernels see does not drift towards a memorised batch),
    or -- `same` -- one host-generated batch repeated."""
    if same:
        b = make_batch(synthetic, diffuser, B, F, N, rank, dev)
        return [b] * count
    from dynamicpdb_amd.rng import DeviceRNG
    out = []
    for i in range(count):
        rng = DeviceRNG(seed=0x5EED0000 + 1000003 * rank + i, device=dev)
        out.append(synthetic.device_batch(rng, diffuser, B, F, N, t=0.5))
    torch.cuda.synchronize()
    return out


def dense_conv_launches(dev, B, F, N, reps=6):
    """The three conv launch types of the tower at the bench's grid on DENSE random operands (every cell of every frame
    non-zero, nothing for the zero-frame skipping to skip), each alone, HIP events on the launch stream: forward
    (1280 -> 640, bias + ReLU), data gradient (640 -> 1280, residual add + ReLU-masked second output: the epilogue the
    backward chain uses) and the weight gradient.  Returns average ms per launch."""
    from dynamicpdb_amd import ops
    g = ops.Grid(B, F, N, dev)
    gen = torch.Generator(device=dev).manual_seed(7)
    rnd = lambda C: torch.randn(B, F, N, C, device=dev, generator=gen).to(torch.bfloat16)
    x, y, r, r2 = g.alloc(1280), g.alloc(640), g.alloc(1280), g.alloc(1280)
    g.interior(x).copy_(rnd(1280))
    g.interior(y).copy_(rnd(640))
    g.interior(r).copy_(rnd(1280))
    g.interior(r2).copy_(rnd(1280))
    wf = (torch.randn(640, 25, 1280, device=dev, generator=gen) / (25 * 1280) ** 0.5).to(torch.bfloat16)
    wd = (torch.randn(1280, 25, 640, device=dev, generator=gen) / (25 * 640) ** 0.5).to(torch.bfloat16)
    bias = torch.zeros(640, device=dev)
    o1, o2, o3 = g.alloc(640), g.alloc(1280), g.alloc(1280)
    dw = torch.zeros((1280, 25, 640), dtype=torch.float32, device=dev)
    runs = {"forward": lambda: ops.conv5x5_fwd(g, x, wf, bias, o1, relu=True),
            "dgrad": lambda: ops.conv5x5_fwd(g, y, wd, None, o2, relu=False, resid=r, C2=o3, R2=r2),
            "wgrad": lambda: ops.conv5x5_wgrad_tn(g, y, x, dw, accumulate=True)}
    out = {}
    for k, fn in runs.items():
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out[k] = e0.elapsed_time(e1) / reps
    del x, y, r, r2, o1, o2, o3, dw
    torch.cuda.empty_cache()
    return out


def collect_conv_traffic(tlog):
    """HBM-side bytes per full-size conv launch from the PMC counters, collected IN THIS RUN when rocprofv3 is on the PATH:
    two separate --pmc passes (FETCH_SIZE | WRITE_SIZE; /opt/skills/guides/MI355X_MICROARCH.md, HBM section: the two do not fit
    one pass) over scripts/conv_launches.py (the two production launches on dense operands, ten times each) in a child
    process; FETCH_SIZE (KiB) is doubled (the guide's gfx950 correction for wide coalesced reads), WRITE_SIZE taken as reported.
    Returns (bytes per launch, source string) or (None, reason).  DFOLD_BENCH_PMC=0 skips it."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if os.environ.get("DFOLD_BENCH_PMC", "1") == "0":
        return None, "DFOLD_BENCH_PMC=0"
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 not on PATH"
    vals = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="dfold_pmc_", dir="/tmp")
        try:
            env = dict(os.environ, TMPDIR="/tmp")
            r = subprocess.run([exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable,
                                os.path.join(ROOT, "scripts", "conv_launches.py")], cwd="/tmp", env=env, capture_output=True,
                               text=True, timeout=150)
            tot, n = 0.0, 0
            for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        if "dfold_conv_w4_kernel" in row["Kernel_Name"] and row["Counter_Name"] == counter:
                            tot += float(row["Counter_Value"])
                            n += 1
            if n == 0:
                return None, f"rocprofv3 --pmc {counter}: no rows for the conv kernel (rc {r.returncode})"
            vals[counter] = tot / n
        except Exception as exc:      # noqa: BLE001
            return None, f"rocprofv3 --pmc {counter} failed: {type(exc).__name__}: {exc}"[:200]
        finally:
            shutil.rmtree(d, ignore_errors=True)
    tlog("PMC passes: FETCH_SIZE %.0f KiB, WRITE_SIZE %.0f KiB per conv launch" % (vals["FETCH_SIZE"], vals["WRITE_SIZE"]))
    return int((2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024), (
        "collected in this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (two passes) over scripts/conv_launches.py "
        "(both production launches, dense operands, average per launch); bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB")


def conv_kernel_roofline(model, trainer, batch, B, F, N, tlog=None, pmc=False):
    """Durations of the 5x5 conv implicit-GEMM launches (kernel dfold_conv_w4_kernel) and of the weight-gradient launches of
    ONE extra, instrumented step, measured with HIP events on the stream the kernels are launched on.  `achieved` is quoted on
    the full-size FORWARD launches (dense activations; 16 per step with the trunk's dead-code elimination, 32 without): the
    backward launches of the same kernel meet gradients that are zero on most frames and skip them on the device (round 6), so
    their duration says how much they skipped, not how fast the kernel is -- they are listed beside it, together with all
    three launch types on dense random operands (`dense_operands`)."""
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
        events.append((e0, e1, int(a[3]), kw.get("bias") is not None))    # a[3] = M (output positions); forward launches carry a bias
        return r

    ops.gemm = timed_gemm
    # the direct weight-gradient launches: the library entry point itself is wrapped (an instance attribute on the CDLL
    # object shadows the exported function for the duration of this one step)
    from dynamicpdb_amd import _lib
    L = _lib.lib()
    wg_orig, wg_events = L.dfold_conv_wgrad_tn, []

    def timed_wgrad(*a):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = wg_orig(*a)
        e1.record()
        nf_arg = a[10]                                # (A, B, dWg, CA, CB, W, Fp, Wp, N, f0, nf, ...): frames of the launch
        wg_events.append((e0, e1, int(getattr(nf_arg, "value", nf_arg))))
        return rc

    L.dfold_conv_wgrad_tn = timed_wgrad
    legacy = None
    try:
        trainer.update_fn(batch, step_optimizer=False)
        torch.cuda.synchronize()
        if ops.CONV_NZ and os.environ.get("DFOLD_BENCH_NO_DENSE") != "1":      # (not under the kernel-trace profiler: see below)
            # the same step once more with the zero-frame skipping off: what rounds 1-5 quoted for the backward launches (every
            # tile walked; the data-gradient launches then multiply the structural zeros of the gradient grids)
            keep_ev, keep_wg = list(events), list(wg_events)
            del events[:], wg_events[:]
            ops.CONV_NZ = False
            try:
                trainer.update_fn(batch, step_optimizer=False)
                torch.cuda.synchronize()
            finally:
                ops.CONV_NZ = True
            legacy = (list(events), list(wg_events))
            events[:], wg_events[:] = keep_ev, keep_wg
    finally:
        ops.gemm = orig
        del L.dfold_conv_wgrad_tn        # the exported function is visible again
    m_full, nf_full = max(m for _, _, m, _ in events), max(n for _, _, n in wg_events) if wg_events else 0
    ms_cone = [e0.elapsed_time(e1) for e0, e1, m, _ in events if m != m_full]
    wg_cone = [e0.elapsed_time(e1) for e0, e1, n in wg_events if n != nf_full]
    wg_ms = [e0.elapsed_time(e1) for e0, e1, n in wg_events if n == nf_full]
    ms = [e0.elapsed_time(e1) for e0, e1, m, fwd in events if m == m_full and fwd]
    ms_bwd = [e0.elapsed_time(e1) for e0, e1, m, fwd in events if m == m_full and not fwd]
    avg_s = sum(ms) / len(ms) * 1e-3
    flops = 2.0 * 1280 * 640 * T(F) * T(N) * B        # algorithmic (non-padding taps), SURVEY 8(d): same for all 8 convs
    achieved = flops / avg_s / 1e12
    dev = batch["t"].device
    # (DFOLD_BENCH_NO_DENSE=1: the rocprofv3 kernel-trace runs skip the isolated dense launches, whose kernels would be averaged into
    #  the same rows as the step's own launches; the line then repeats the in-step forward time for them)
    dense = dense_conv_launches(dev, B, F, N) if os.environ.get("DFOLD_BENCH_NO_DENSE") != "1" else \
        {"forward": avg_s * 1e3, "dgrad": avg_s * 1e3, "wgrad": avg_s * 1e3}
    frac_of = lambda t_ms: round(flops / (t_ms * 1e-3) / 1e12 / BF16_MFMA_PEAK_TFLOPS, 4)
    no_skip = None
    if legacy is not None:
        lm = max(m for _, _, m, _ in legacy[0])
        lf = [e0.elapsed_time(e1) for e0, e1, m, fwd in legacy[0] if m == lm and fwd]
        lb = [e0.elapsed_time(e1) for e0, e1, m, fwd in legacy[0] if m == lm and not fwd]
        lw = [e0.elapsed_time(e1) for e0, e1, n in legacy[1] if n == nf_full]
        avg = lambda v: sum(v) / max(1, len(v))
        no_skip = {"what": "the same instrumented step with DFOLD_CONV_NZ=0 (every tile / reduction row walked, as rounds 1-5): full-size "
                           "launches in the step, on the step's own operands (ReLU-sparse activations, gradients that are structurally "
                           "zero on most frames: the power-limited kernels clock higher on them than on dense data)",
                   "forward_ms": round(avg(lf), 4), "forward_frac": frac_of(avg(lf)), "dgrad_ms": round(avg(lb), 4),
                   "dgrad_frac": frac_of(avg(lb)), "wgrad_ms": round(avg(lw), 4), "wgrad_frac": frac_of(avg(lw)),
                   "launches": [len(lf), len(lb), len(lw)]}
    traffic, traffic_source = None, None
    wt, c = None, None
    if (B, F, N) == (8, 32, 256):
        traffic, traffic_source = collect_conv_traffic(tlog or (lambda m: None)) if pmc else (None, "multi-rank run")
        for tag in ("r6b", "r6", "r5", "r4", "r3"):         # the newest committed PMC pass: the fallback, and the weight gradient's bytes
            pmc = os.path.join(ROOT, "profiles", f"{tag}_pmc_conv.json")
            if os.path.exists(pmc):
                with open(pmc) as fh:
                    c = json.load(fh)
                if traffic is None:
                    why = traffic_source
                    traffic = int((2.0 * c["FETCH_SIZE_kb"] + c["WRITE_SIZE_kb"]) * 1024)
                    traffic_source = (f"profiles/{tag}_pmc_conv.json (rocprofv3 --pmc FETCH_SIZE x 2 + WRITE_SIZE, scripts/gpu_pmc.sh); "
                                      f"not collected in this run: {why}")
                if "wgrad_tn" in c:
                    wt = int((2.0 * c["wgrad_tn"]["FETCH_SIZE_kb"] + c["wgrad_tn"]["WRITE_SIZE_kb"]) * 1024)
                break
    second = None
    if wg_ms:      # the second-largest kernel of the step: the conv weight gradient (same algorithmic FLOPs per launch)
        second = {"kernel": "conv_wgrad_tn_kernel (5x5 conv weight gradient straight from the channels-last grids)", "bound": "mfma",
                  "achieved": round(flops / (dense["wgrad"] * 1e-3) / 1e12, 1), "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                  "frac": frac_of(dense["wgrad"]), "traffic": wt, "avg_launch_ms": round(dense["wgrad"], 4), "flop_per_launch": flops,
                  "measured_on": "dense random operands, the launch alone (`dense_operands`): inside the step every full-size "
                                 "weight-gradient launch meets a gradient that is zero on most frames and skips them",
                  "in_step": {"full_size_launches": len(wg_ms), "total_ms": round(sum(wg_ms), 3),
                              "per_launch_ms": [round(v, 3) for v in wg_ms]}}
    return {"bound": "mfma", "achieved": round(achieved, 1), "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / BF16_MFMA_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_source,
            "kernel": "dfold_conv_w4_kernel (5x5 conv implicit GEMM, one wave per SIMD, 512 x 160 tile, 32-channel halo groups); "
                      "`achieved` = the full-size FORWARD launches of the step (ReLU-sparse activations, every tile computed)",
            "launches": len(ms), "avg_launch_ms": round(avg_s * 1e3, 4), "flop_per_launch": flops, "second_kernel": second,
            "dense_operands": {"what": "each launch type alone on dense random operands at the same grid (nothing to skip): ms per "
                                       "launch and fraction of the bf16 MFMA peak",
                               "forward_ms": round(dense["forward"], 4), "forward_frac": frac_of(dense["forward"]),
                               "dgrad_ms": round(dense["dgrad"], 4), "dgrad_frac": frac_of(dense["dgrad"]),
                               "wgrad_ms": round(dense["wgrad"], 4), "wgrad_frac": frac_of(dense["wgrad"])},
            "in_step_without_skipping": no_skip,
            "backward_launches": {"what": "full-size data-gradient launches of the step (same kernel, NZ instantiation): tiles whose "
                                          "input frames are zero by the frame flags skip their K walk, so a launch costs what its "
                                          "live tiles cost (DFOLD_CONV_NZ=0: every tile)",
                                  "launches": len(ms_bwd), "total_ms": round(sum(ms_bwd), 3),
                                  "per_launch_ms": [round(v, 3) for v in ms_bwd]},
            "cone_launches": {"what": "launches of the trunk's inner blocks (dependency cone of the last frame: 1 ... 15 of the "
                                      "frames per layer, split-K when thin): same kernels, not part of `achieved`",
                              "conv_fwd_dgrad": len(ms_cone), "conv_fwd_dgrad_total_ms": round(sum(ms_cone), 3),
                              "wgrad": len(wg_cone), "wgrad_total_ms": round(sum(wg_cone), 3)},
            "full_launches_total_ms": {"conv_fwd": round(sum(ms), 3), "conv_dgrad": round(sum(ms_bwd), 3), "wgrad": round(sum(wg_ms), 3)},
            "note": "peak = nominal dense bf16 MFMA rate at 2.4 GHz; the launch is power-limited on real operands: the same "
                    "binary on all-zero operands runs 2.13 PFLOP/s = 0.85 of the peak, on dense random operands 1.44-1.46 "
                    "(scripts/exp_conv_dvfs.py); hipBLASLt on the materialised GEMM of the same size 1.08 / 1.51 PFLOP/s "
                    "(scripts/bench_conv.py library); DESIGN.md section 5"}


def triangle_roofline(dev, reps=10):
    """Extra object (not the headline): forward of the north-star-named triangle operators at N_res 256 / 512 on this
    GPU: whole-call time (HIP events) vs the algorithmic bytes of SURVEY 8d (read z + write out + mask) over the 8 TB/s
    HBM peak.  Per-stage numbers and counters: scripts/bench_triangle.py, profiles/r2_triangle_*."""
    from dynamicpdb_amd.model import triangle as T_
    out = {}
    for name, ctor in (("tri_mul_out", lambda: T_.TriangleMultiplicationOutgoing(128, 128)),
                       ("tri_att_start", lambda: T_.TriangleAttentionStartingNode(128, 32, 4))):
        for n, nb in ((256, 1), (512, 1), (256, 8), (512, 8)):
            torch.manual_seed(0)
            m = ctor().to(dev)
            z = torch.randn(nb, n, n, 128, device=dev) * 1.5
            mask = (torch.rand(nb, n, n, device=dev) > 0.05).float()
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
            alg = nb * n * n * (2 * 128 * 4 + 4)
            key = f"{name}_n{n}" + ("" if nb == 1 else f"_b{nb}")
            out[key] = {"ms": round(t * 1e3, 4), "batch": nb, "algorithmic_bytes": alg, "GBps": round(alg / t / 1e9, 1),
                        "hbm_frac": round(alg / t / 8.0e12, 4)}
            if nb == 8:      # forward + backward through autograd (input and every parameter gradient), the same call shapes
                zg = z.clone().requires_grad_(True)
                gy = torch.randn_like(z)
                for _ in range(2):
                    m(zg, mask=mask).backward(gy)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(max(2, reps // 2)):
                    m(zg, mask=mask).backward(gy)
                e1.record()
                torch.cuda.synchronize()
                out[key]["fwd_bwd_ms"] = round(e0.elapsed_time(e1) / max(2, reps // 2), 4)
                del zg, gy
            del m, z, mask
    out["note"] = ("forward, fp32 pair tensor, batch 1 and batch 8, fused kernels of csrc/pair_fused.hip / triatt_*.hip; bound = hbm "
                   "(8 TB/s): algorithmic bytes = read z + write out + mask (SURVEY 8d); fwd_bwd_ms (batch 8): forward + backward "
                   "through autograd (fused three-pass backward of the multiplication, streaming backward of the attention); "
                   "counters in profiles/, DESIGN.md section 4")
    return out


def neighbours(dev, tlog, reps=5):
    """Extra object for SURVEY 8f ranks 2 / 3 (parity-green since round 2, never timed): one Evoformer pair block
    (OpenFold EvoformerBlockCore: MSA transition, outer product mean, the four triangle operators, pair transition) at N_res 256
    with 64 sequences, OmegaFold's GeometricAttention at N_res 256, and the dataset-side transforms of one 32-frame window
    (atom37 -> frames + torsions, forward noising) on the device; forward, fp32 pair tensor, HIP events."""
    from dynamicpdb_amd import synthetic
    from dynamicpdb_amd.data import data_transforms as dt
    from dynamicpdb_amd.data.se3_diffuser import SE3Diffuser
    from dynamicpdb_amd.model.geoformer import GeometricAttention
    from dynamicpdb_amd.model.pair_stack import EvoformerBlockCore

    def timed(fn):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    out = {}
    N, S = 256, 64
    torch.manual_seed(0)
    blk = EvoformerBlockCore(c_m=256, c_z=128, c_hidden_opm=32, c_hidden_mul=128, c_hidden_pair_att=32, no_heads_msa=8, no_heads_pair=4,
                             transition_n=4, pair_dropout=0.25, inf=1e9, eps=1e-8).to(dev).eval()
    m, z = torch.randn(S, N, 256, device=dev), torch.randn(N, N, 128, device=dev)
    mm, pm = torch.ones(S, N, device=dev), torch.ones(N, N, device=dev)
    with torch.no_grad():
        ms = timed(lambda: blk(m, z, mm, pm))
    pair_bytes = N * N * 128 * 4
    out["evoformer_block_core_n256_s64"] = {"ms": round(ms, 4), "pair_tensor_mb": round(pair_bytes / 1e6, 1),
                                            "pair_passes_at_8TBps": round(ms * 1e-3 * 8.0e12 / (2 * pair_bytes), 1),
                                            "note": "five pair updates + outer product mean + MSA transition, forward; "
                                                    "pair_passes_at_8TBps = how many read+write passes over the pair tensor the "
                                                    "time would buy at the HBM peak"}
    del blk, m, z
    ga = GeometricAttention(128, 32, 4, 2).to(dev).eval()
    e, em = torch.randn(N, N, 128, device=dev), torch.ones(N, device=dev)
    with torch.no_grad():
        ms = timed(lambda: ga(e, em))
    out["geometric_attention_n256"] = {"ms": round(ms, 4), "note": "OmegaFold GeometricAttention (2 gated products + 2-axis attention), forward"}
    del ga, e
    F = 32
    conf = synthetic.default_conf(F, cache_dir="/tmp/dfold_igso3_cache/")
    diffuser = SE3Diffuser(conf.diffuser)
    w = synthetic.synthetic_window(5, F, N, t=0.5, diffuser=diffuser)
    from dynamicpdb_amd.model import geometry as G
    with torch.no_grad():
        a37 = G.frames_to_atoms_hip(w["rigids_0"].to(dev).float(), w["torsion_angles_sin_cos"].to(dev).float(), w["aatype"].to(dev))[1]
    prot = {"aatype": w["aatype"].to(dev).long(), "all_atom_positions": a37.double(),
            "all_atom_mask": (a37.abs().sum(-1) > 0).double()}
    ms = timed(lambda: dt.atom37_to_torsion_angles()(dt.atom37_to_frames(dict(prot))))
    out["dataset_geometry_f32_n256"] = {"ms": round(ms, 4), "residues": F * N,
                                        "note": "atom37_to_frames + atom37_to_torsion_angles of one 32-frame window (fp64, one launch)"}
    r0 = w["rigids_0"].to(dev).float()
    ms = timed(lambda: diffuser.forward_marginal_t7(r0, 0.5))
    out["forward_marginal_f32_n256"] = {"ms": round(ms, 4), "note": "forward noising of one window incl. the numpy draws (host RNG) and their upload"}
    tlog("neighbours: " + json.dumps({k: v["ms"] for k, v in out.items()}))
    return out


def hbm_kernels(dev, tlog, reps=10):
    """Extra object: the other HBM-bound kernels of SURVEY 8d at the config-3 shapes, each called through the C ABI on its own
    (HIP events): algorithmic bytes (inputs read once + outputs written once) / time against the 8 TB/s HBM peak."""
    from ctypes import c_int32, c_int64
    from dynamicpdb_amd import _lib, ops
    from dynamicpdb_amd._lib import check, stream
    from dynamicpdb_amd.model.functional import ctypes_float
    from dynamicpdb_amd.ops import BF16, _p
    L = _lib.lib()

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e-3

    out = {}

    def row(name, t, nbytes, what):
        out[name] = {"ms": round(t * 1e3, 4), "algorithmic_bytes": int(nbytes), "GBps": round(nbytes / t / 1e9, 1),
                     "hbm_frac": round(nbytes / t / 8.0e12, 4), "what": what}

    B, F, N = 8, 32, 256
    # IPA pair-side projections: linear_b + down_z in one pass over z
    z = torch.randn(B, N, N, 128, device=dev).to(BF16)
    wb, wdz = torch.randn(8, 128, device=dev).to(BF16), torch.randn(32, 128, device=dev).to(BF16)
    bias_t = torch.empty((B, 8, N, N), dtype=torch.float32, device=dev)
    pz, pzT = torch.empty((B, N, N, 32), dtype=BF16, device=dev), torch.empty((B, N, 32, N), dtype=BF16, device=dev)
    t = timed(lambda: check(L.dfold_ipa_pair_proj(_p(z), _p(wb), _p(wdz), _p(bias_t), _p(pz), _p(pzT), c_int32(B), c_int32(N), stream()), "pair_proj"))
    row("ipa_pair_proj", t, z.numel() * 2 + bias_t.numel() * 4 + pz.numel() * 2 * 2, "z (bf16) once in; pair bias fp32 + down_z in both layouts out")
    del z, bias_t, pz, pzT
    # MyLayerNorm over a window (fp64 statistics) + apply
    P = F * N * 256
    x = torch.randn(B, P, device=dev)
    y = torch.empty((B, P), dtype=BF16, device=dev)
    st, mr = torch.empty(2 * B, dtype=torch.float64, device=dev), torch.empty(2 * B, dtype=torch.float32, device=dev)
    t = timed(lambda: check(L.dfold_gln_fwd(_p(x), _p(st), _p(y), _p(mr), c_int32(B), c_int64(P), ctypes_float(1e-4), c_int32(0), stream()), "gln"))
    row("my_layer_norm_fwd", t, x.numel() * 4 + y.numel() * 2, "fp32 pre-norm activations in, bf16 out (the statistics pass re-reads x: counted once)")
    del x, y
    # embedder input layer (k = 7 -> 256, SiLU)
    Pn = B * F * N
    xin, W, b = torch.randn(Pn, 7, device=dev), torch.randn(256, 7, device=dev), torch.randn(256, device=dev)
    o = torch.empty((Pn, 256), dtype=BF16, device=dev)
    t = timed(lambda: check(L.dfold_embed_in_fwd(_p(xin), _p(W), _p(b), _p(o), c_int64(Pn), c_int32(7), c_int32(256), stream()), "embed"))
    row("embed_in_fwd", t, xin.numel() * 4 + o.numel() * 2, "7 inputs per position in, 256 bf16 channels out")
    del xin, o
    # expand_edge: cast + 128 -> 128 linear over the pair tensor
    e = torch.randn(B * N * N, 128, device=dev)
    We = torch.randn(128, 128, device=dev).to(BF16)
    be = torch.zeros(128, device=dev)

    def expand():
        eb = ops.cast_bf16(e)
        ops.linear_fwd(eb, We, be, out_dtype=BF16)
    t = timed(expand)
    row("expand_edge", t, e.numel() * 4 + e.numel() * 2, "fp32 edge_repr in, bf16 expanded edges out (cast + MFMA linear: two launches)")
    del e
    # fused Adam (amsgrad) over the model's parameter count
    from dynamicpdb_amd.optim import FusedAdam
    ps = [torch.nn.Parameter(torch.randn(n, device=dev)) for n in (184_000_000 // 8,) * 8]
    for q in ps:
        q.grad = torch.randn_like(q)
    opt = FusedAdam(ps, lr=1e-4)                     # (always amsgrad, as the reference's Adam(amsgrad=True), train_DFOLD_dynamics.py:412)
    opt.step()
    t = timed(opt.step)
    row("adam_amsgrad", t, sum(q.numel() for q in ps) * 4 * 8, "p, g, exp_avg, exp_avg_sq, max_exp_avg_sq read; all but g written")
    tlog("hbm kernels: " + json.dumps({k: v["hbm_frac"] for k, v in out.items()}))
    return out


def cpu_baseline(F, N, seed_w=0):
    """The oracle (CPU port of the reference path; kind "port" -- the reference itself is not on the GPU box) running the
    reference's update_fn on host cores: zero_grad + forward + loss + backward + Adam(amsgrad) step
    (train_DFOLD_dynamics.py:660-667), ONE window of F frames x N_res = N (the reference has no batch axis: B windows are B
    sequential calls, so frames/s of one window is the rate).  Bounded sample: one SAME-SHAPE warm-up iteration, then two
    timed iterations whose mean is reported.  Threads = one per physical core: on the MI355X box the 256 usable hardware
    threads are 128 cores x SMT2, and 256 torch threads were measured 15x SLOWER than 128 on this workload (315.7 s vs
    21.6 s per iteration, profiles/r3_bench_cpu_threads.txt) -- oversubscribed SMT siblings in the MKLDNN conv."""
    from oracle import dfold_oracle as O
    from dynamicpdb_amd import synthetic
    from dynamicpdb_amd.data.se3_diffuser import SE3Diffuser
    try:
        cores = len(os.sched_getaffinity(0))      # cores this process may actually use (cgroup / affinity aware)
    except AttributeError:
        cores = os.cpu_count() or 1
    sd = synthetic.seeded_state_dict(seed_w)
    P = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    opt = torch.optim.Adam(list(P.values()), lr=1e-4, amsgrad=True)
    conf = synthetic.default_conf(F, cache_dir="/tmp/dfold_igso3_cache/")
    w = synthetic.synthetic_window(7, F, N, t=0.5, diffuser=SE3Diffuser(conf.diffuser))

    def one():
        t0 = time.time()
        opt.zero_grad(set_to_none=True)
        out = O.full_score_network(P, O.Schedules(), w)
        loss, _ = O.loss_fn(out, w)
        loss.backward()
        opt.step()
        return time.time() - t0

    smt = 1
    try:
        with open("/sys/devices/system/cpu/cpu0/topology/thread_siblings_list") as fh:
            sib = fh.read().strip()
        smt = max(1, len(sib.replace("-", ",").split(",")))
    except OSError:
        pass
    threads = max(1, cores // smt)
    torch.set_num_threads(threads)
    warm = {threads: one()}
    print(f"[bench cpu_baseline] warm-up, {F}-frame window, {threads} threads: {warm[threads]:.2f} s", file=sys.stderr, flush=True)
    timed = [one() for _ in range(2)]
    print(f"[bench cpu_baseline] timed, {threads} threads: {timed[0]:.2f} s, {timed[1]:.2f} s", file=sys.stderr, flush=True)
    t = sum(timed) / len(timed)
    return {"value": round(F / t, 4), "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"oracle update_fn (zero_grad+fwd+loss+bwd+Adam amsgrad), 1 window of {F} frames x N_res={N}: mean of 2 timed "
                      f"iterations ({timed[0]:.2f} s, {timed[1]:.2f} s) after same-shape warm-up iterations at "
                      + ", ".join(f"{k} threads {v:.2f} s" for k, v in sorted(warm.items())) + f"; {cores} usable hardware threads = "
                      f"{threads} physical cores x SMT{smt}, one torch CPU thread per core (all {cores}: measured 15x slower, "
                      "profiles/r3_bench_cpu_threads.txt)" + ("" if F == 32 else "; the bench's windows are 32 frames: the per-frame "
                      f"conv work T(F)/F grows from {T(F) / F:.2f} taps (F={F}) to 4.81 (F=32), so this rate OVERSTATES the CPU's "
                      "32-frame rate (conservative for any GPU/CPU ratio)"),
            "reference_probe": {"value": 0.43, "unit": "frames/s", "cores": 8, "shape": "1 window, 2 frames x N_res=256, fwd+bwd",
                                "source": "the reference's own code (FullScoreNetwork fwd+bwd) timed in the build container, "
                                          "SURVEY.md section 6 [probe]; the reference does not exist on the GPU box"}}


def config1_eval(dev, tlog, cpu=True):
    """Extra object: BASELINE config 1 is the reference's EVAL configuration (run_eval.sh:4-17, eval_DFOLD_dynamics.py:59-204 ->
    Experiment.inference_fn, train_DFOLD_dynamics.py:1425-1547): one 16-frame window at N_res 96, num_t = 10, noise_scale 0.1.
    GPU: the device-resident sampler (10 model forwards + the self-conditioning pass + 9 reverse steps as one HIP launch each)
    with device Philox draws and with host-injected numpy draws.  CPU: the oracle's restatement of ONE reverse iteration of the
    reference's loop (model forward on host cores + the host reverse step: quaternion -> rotation vector through scipy,
    so3 / r3 Euler-Maruyama updates, back to quaternions), which the reference repeats num_t times with four host round trips
    each (se3_diffuser.py:160-215, rigid_utils.py:208-227)."""
    import numpy as np
    from dynamicpdb_amd import experiment, synthetic
    from dynamicpdb_amd.data.se3_diffuser import SE3Diffuser
    from dynamicpdb_amd.model.Dfold_network_dynamic import FullScoreNetwork
    from dynamicpdb_amd.rng import DeviceRNG
    F, N, num_t, noise = 16, 96, 10, 0.1
    conf = synthetic.default_conf(F, cache_dir="/tmp/dfold_igso3_cache/")
    diffuser = SE3Diffuser(conf.diffuser)
    model = FullScoreNetwork(conf.model, diffuser)
    sd = synthetic.seeded_state_dict(31)
    model.load_state_dict(sd, strict=True)
    model.to(dev)
    w = synthetic.synthetic_window(32, F, N, t=1.0, diffuser=None)
    np.random.seed(90)
    prior = diffuser.sample_ref(n_samples=F * N, as_tensor_7=True)["rigids_t"].reshape(F, N, 7).float()
    init = {k: v.to(dev) for k, v in w.items()}
    init["rigids_t"] = prior.to(dev)
    kw = dict(num_t=num_t, min_t=0.01, center=True, aux_traj=False, self_condition=True, noise_scale=noise)

    def timed(fn, reps=3):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    ms_dev = timed(lambda: experiment.inference_fn(model, diffuser, init, rng=DeviceRNG(seed=5, device=dev), **kw))
    np.random.seed(91)
    ms_inj = timed(lambda: experiment.inference_fn(model, diffuser, init, **kw))       # numpy draws, copied to the device per step
    with torch.no_grad():
        feats = experiment.set_t_feats(diffuser, dict(init), 0.5, torch.ones(1, device=dev))
        feats["sc_ca_t"] = init["rigids_t"][..., 4:].clone()
        ms_fwd = timed(lambda: model(dict(feats)), reps=5)
        out = model(dict(feats))
        rng = DeviceRNG(seed=6, device=dev)
        ms_rev = timed(lambda: diffuser.reverse_t7(feats["rigids_t"], out["rot_score"], out["trans_score"], 0.5, 1.0 / num_t,
                                                   diffuse_mask=torch.ones(F, N, device=dev), center=True, noise_scale=noise, rng=rng),
                       reps=20)
    # eight samples of the same protein as ONE batch of windows (the engine's window axis; the reference samples them one by one)
    B8 = 8
    init8 = {k: (v[None].expand((B8,) + tuple(v.shape)).contiguous() if k != "t" else v.expand(B8).contiguous()) for k, v in init.items()}
    np.random.seed(92)
    init8["rigids_t"] = diffuser.sample_ref(n_samples=B8 * F * N, as_tensor_7=True)["rigids_t"].reshape(B8, F, N, 7).float().to(dev)
    ms_b8 = timed(lambda: experiment.inference_fn(model, diffuser, init8, rng=DeviceRNG(seed=7, device=dev), **kw), reps=2) / B8
    tlog(f"config 1 eval: {ms_dev:.1f} ms per sample (device draws), {ms_inj:.1f} ms (numpy draws), forward {ms_fwd:.2f} ms, "
         f"reverse step {ms_rev:.3f} ms, {ms_b8:.1f} ms per sample in a batch of {B8}")
    res = {"workload": "BASELINE config 1 (the reference's eval configuration): 1 window, 16 frames x N_res 96, inference_fn with "
                       "num_t = 10, noise_scale 0.1, self-conditioning pass; random-init seeded weights, synthetic window",
           "ms_per_sample_device_rng": round(ms_dev, 2), "ms_per_sample_numpy_draws": round(ms_inj, 2),
           "ms_per_model_forward": round(ms_fwd, 3), "ms_per_reverse_step_kernel": round(ms_rev, 4),
           "ms_per_sample_batch8": round(ms_b8, 2), "frames_per_s_sampled": round(F / (ms_dev * 1e-3), 1),
           "frames_per_s_sampled_batch8": round(F / (ms_b8 * 1e-3), 1), "model_forwards_per_sample": num_t + 1,
           "reverse_steps_per_sample": num_t - 1}
    del model
    torch.cuda.empty_cache()
    if cpu:
        from scipy.spatial.transform import Rotation
        from oracle import dfold_oracle as O
        try:
            cores = len(os.sched_getaffinity(0))
        except AttributeError:
            cores = os.cpu_count() or 1
        smt = 1
        try:
            with open("/sys/devices/system/cpu/cpu0/topology/thread_siblings_list") as fh:
                smt = max(1, len(fh.read().strip().replace("-", ",").split(",")))
        except OSError:
            pass
        threads = max(1, cores // smt)
        torch.set_num_threads(threads)
        P, sched = {k: v.clone() for k, v in sd.items()}, O.Schedules()
        wc = dict(w)
        wc["rigids_t"] = prior.clone()
        rs, ts = diffuser.score_scaling(0.5)
        wc["t"] = torch.tensor([0.5])
        wc["rot_score_scaling"], wc["trans_score_scaling"] = torch.tensor([float(rs)]), torch.tensor([float(ts)])
        wc["sc_ca_t"] = prior[..., 4:].clone()

        def one():
            t0 = time.time()
            with torch.no_grad():
                o = O.full_score_network(P, sched, wc)
            rt = wc["rigids_t"].double().numpy()
            q = rt[..., :4]
            rotvec = Rotation.from_quat(q.reshape(-1, 4)[:, [1, 2, 3, 0]]).as_rotvec().reshape(q.shape[:-1] + (3,))
            zr, zt = np.random.normal(size=(F, N, 3)), np.random.normal(size=(F, N, 3))
            rv1 = O.so3_reverse(sched, rotvec, o["rot_score"].double().numpy(), 0.5, 1.0 / num_t, noise * zr)
            x1 = O.r3_reverse(sched, rt[..., 4:], o["trans_score"].double().numpy(), 0.5, 1.0 / num_t, noise * zt)
            qn = Rotation.from_rotvec(rv1.reshape(-1, 3)).as_quat()[:, [3, 0, 1, 2]].reshape(F, N, 4)
            wc["rigids_t"] = torch.tensor(np.concatenate([qn, x1], -1), dtype=torch.float32)
            return time.time() - t0

        warm = one()
        timed_s = [one() for _ in range(2)]
        per = sum(timed_s) / len(timed_s)
        tlog(f"config 1 eval, CPU oracle: warm-up {warm:.2f} s, timed {timed_s[0]:.2f} s, {timed_s[1]:.2f} s per reverse iteration ({threads} threads)")
        res["cpu_baseline"] = {"value": round(per * 1e3, 1), "unit": "ms per reverse iteration (model forward + host reverse step)",
                               "cores": threads, "kind": "port",
                               "sample": f"oracle forward (the reference's aten ops: F.conv2d, F.linear, broadcast point distances) + scipy / "
                                         f"numpy reverse step, mean of 2 timed iterations ({timed_s[0]:.2f} s, {timed_s[1]:.2f} s) after a "
                                         f"warm-up; a full sample = {num_t + 1} forwards + {num_t - 1} reverse steps",
                               "ms_per_sample_extrapolated": round(per * 1e3 * (num_t + 1), 0)}
        res["speedup_vs_cpu_per_sample"] = round(per * 1e3 * (num_t + 1) / ms_dev, 1)
        res["speedup_vs_cpu_per_sample_batch8"] = round(per * 1e3 * (num_t + 1) / ms_b8, 1)
    return res


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


def timed_steps(trainer, batches, steps, sync, terms=None):
    """`steps` update_fn calls on successive pre-staged batches between two (barrier + device sync)s; returns seconds
    (this rank) and the last loss (device scalar).  terms (a list): receives the (loss, aux) device scalars of the first and
    the last timed step (read by the caller after the region)."""
    sync()
    t0 = time.perf_counter()
    loss = None
    for i in range(steps):
        loss, aux = trainer.update_fn(batches[i % len(batches)])     # device scalars: no host sync inside the timed region
        if terms is not None and i in (0, steps - 1):
            terms.append((loss, aux))
    sync()
    return time.perf_counter() - t0, loss


def other_config(tag, B, F, N, dev, steps, tlog):
    """Extra object: the same update_fn at another BASELINE configuration on ONE GPU (fresh model of that window length,
    fresh device batches), all frames through the tower."""
    from dynamicpdb_amd import experiment, synthetic
    from dynamicpdb_amd.data.se3_diffuser import SE3Diffuser
    from dynamicpdb_amd.model.Dfold_network_dynamic import FullScoreNetwork
    conf = synthetic.default_conf(F, cache_dir="/tmp/dfold_igso3_cache/")
    diffuser = SE3Diffuser(conf.diffuser)
    model = FullScoreNetwork(conf.model, diffuser)
    model.load_state_dict(synthetic.seeded_state_dict(0), strict=True)
    model.to(dev)
    tr = experiment.Trainer(model, lr=1e-4, last_frame_only=False, sync_params=False)
    batches = make_batches(synthetic, diffuser, B, F, N, 0, dev, 2 + steps, False)
    sync = torch.cuda.synchronize
    for b in batches[:2]:
        tr.update_fn(b)
    el, loss = timed_steps(tr, batches[2:], steps, sync)
    ms = el / steps * 1e3
    fl = 3.0 * synthetic.step_flops_fwd(F, N, inner_cone=bool(model.score_model.trunk_dce)) * B      # (the FLOPs the step executes)
    tlog(f"{tag}: {ms:.1f} ms/step")
    res = {"workload": f"{tag}: N_res={N}, {F}-frame windows, {B} windows on one GPU, full update_fn, all frames",
           "windows": B, "frames": F, "n_res": N, "ms_per_step": round(ms, 3), "value": round(B * F / (el / steps), 2),
           "unit": "frames/s", "steps": steps, "step_tflop": round(fl / 1e12, 2),
           "step_mfma_frac": round(fl / (el / steps) / 1e12 / BF16_MFMA_PEAK_TFLOPS, 4), "loss": round(float(loss), 5)}
    del tr, model, batches
    torch.cuda.empty_cache()
    return res


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
        # (DFOLD_BENCH_BACKEND=gloo + DFOLD_BENCH_ONE_GPU=1: N ranks sharing cuda:0, a dry run of the multi-rank path where
        # only one GPU is available; RCCL itself refuses two ranks on one device)
        dist.init_process_group(backend=os.environ.get("DFOLD_BENCH_BACKEND", "nccl"))
        assert dist.get_world_size() == args.gpus
    if os.environ.get("DFOLD_BENCH_ONE_GPU") == "1":
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import __graft_entry__
    if rank == 0:
        __graft_entry__.build()
    if world > 1:
        dist.barrier()
    from dynamicpdb_amd import experiment, synthetic
    from dynamicpdb_amd import ops as ops_mod
    from dynamicpdb_amd.data.se3_diffuser import SE3Diffuser
    from dynamicpdb_amd.model.Dfold_network_dynamic import FullScoreNetwork
    B, F, N = args.windows, args.frames, args.nres
    tlog = lambda msg: print(f"[bench rank{rank} +{time.perf_counter() - T0:.1f}s] {msg}", file=sys.stderr, flush=True)
    conf = synthetic.default_conf(F, cache_dir="/tmp/dfold_igso3_cache/")
    diffuser = SE3Diffuser(conf.diffuser)
    model = FullScoreNetwork(conf.model, diffuser)
    # like the reference, every rank initialises from its own seed (train_DFOLD_dynamics.py:419) and the trainer starts
    # all of them from rank 0's parameters (DDP's start-up broadcast, :615); rank 0 = the seeded benchmark weights
    model.load_state_dict(synthetic.seeded_state_dict(rank), strict=True)
    model.to(dev)
    # headline: every frame through the conv tower, the work the reference does (SURVEY 8d FLOP model)
    trainer = experiment.Trainer(model, lr=args.lr, last_frame_only=(args.mode == "last_frame"))
    trainer.reducer.timing = world > 1
    # the benchmark step runs one and the same autograd graph on every rank and in every step (like the reference under
    # DistributedDataParallel without find_unused_parameters): after two clean steps the reducer drops its per-step flag
    # collective and the host no longer waits for the device once per step (DFOLD_DP_STATIC_GRAPH=0 keeps the collective)
    trainer.reducer.static_graph = os.environ.get("DFOLD_DP_STATIC_GRAPH", "1") != "0"
    if world > 1:
        ck = torch.stack([p.detach().double().sum() for p in model.parameters()]).sum().reshape(1)
        lo, hi = ck.clone(), ck.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        if float(hi - lo) != 0.0:
            raise SystemExit("ranks hold different parameters after the start-up broadcast")
    tlog("model ready (%.1f MB of parameters broadcast from rank 0)" % (trainer.bytes_broadcast / 1e6))
    nb = args.warmup + args.steps + 1
    batches = make_batches(synthetic, diffuser, B, F, N, rank, dev, nb, args.same_batch)
    tlog("%d batches staged in HBM" % nb)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    first_loss = None
    for i in range(args.warmup):
        l0, _ = trainer.update_fn(batches[i])
        torch.cuda.synchronize()
        first_loss = float(l0) if first_loss is None else first_loss
        tlog("warm-up step done")
    trainer.reducer.wait_ms.clear()
    terms = []
    elapsed, loss = timed_steps(trainer, batches[args.warmup:args.warmup + args.steps], args.steps, sync, terms)
    fmt_terms = lambda la: dict({"loss": round(float(la[0]), 5)}, **{k: round(float(v), 5) for k, v in la[1].items()})
    terms = [fmt_terms(t) for t in terms]
    if first_loss is None:
        first_loss = float(loss)
    tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    elapsed = float(tt)
    tlog(f"timed region done: {elapsed / args.steps * 1e3:.1f} ms/step")
    # the instrumented extra step contains the gradient all-reduce: EVERY rank runs it (a collective issued by rank 0
    # alone would pair with the other ranks' next step and hang the job at the end); rank 0 reports its own timings
    roof = conv_kernel_roofline(model, trainer, batches[-1], B, F, N, tlog if rank == 0 else None, pmc=(world == 1)) if args.mode == "all_frames" else None
    waits, dp_info = None, None
    if world > 1:        # how long each rank's stream sat in finish() waiting for the gradient collectives, per step
        w = torch.tensor([sum(trainer.reducer.wait_ms) / max(1, len(trainer.reducer.wait_ms))], device=dev, dtype=torch.float64)
        allw = [torch.zeros_like(w) for _ in range(world)]
        dist.all_gather(allw, w)
        waits = [round(float(x), 3) for x in allw]
        # what makes the N > 1 line explain itself: who took part, how early each bucket's collective could start relative
        # to the end of backward (the time available for overlap), and what one bucket achieves on the wire in isolation
        me = "rank%d:cuda%d:%s" % (rank, local, getattr(torch.cuda.get_device_properties(dev), "gcnArchName", "?").split(":")[0])
        seen = [None] * world
        dist.all_gather_object(seen, me)
        red = trainer.reducer
        lead = list(red.bucket_lead_ms)                 # last timed step (events resolved at the instrumented step's begin)
        prof = red.profile_buckets(reps=3)              # collective: every rank calls it
        for row in prof:
            row["launched_ms_before_backward_end"] = lead[row["bucket"]] if row["bucket"] < len(lead) else None
        try:
            ver = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            ver = None
        gb = sum(r["mb"] for r in prof)
        dp_info = {"backend": dist.get_backend(), "rccl_version": ver, "ranks_seen": seen, "gradient_mb_per_step": round(gb, 1),
                   "payload_dtype": "bf16" if red.payload_dtype == torch.bfloat16 else "fp32", "static_graph": bool(red.static_graph),
                   "buckets": prof,
                   "note": "launched_ms_before_backward_end: compute-stream time between a bucket's all-reduce launch and the end "
                           "of backward (what can overlap; buckets launched in finish() show ~0); isolated_ms / bus_GBps: the same "
                           "bucket's collective alone, bus bandwidth = 2 (n-1)/n bytes / time (xGMI ring: per-link bound)"}
    # second timed region: the engine's training-step mode (Trainer default).  Loss, gradients and the optimizer update
    # are identical (tests/test_network_gpu.py::test_last_frame_only_training_mode_equals_full); the conv tower only
    # evaluates the dependency cone of the last frame, the one frame the live loss terms and frame updates read.
    el2 = None
    if not args.no_last_frame_mode and args.mode == "all_frames":
        trainer.last_frame_only = True
        trainer.reducer.reset_structure()         # another graph: re-discover (and, static_graph, re-arm the flag collective)
        trainer.update_fn(batches[0])
        trainer.update_fn(batches[0])
        el2, _ = timed_steps(trainer, batches[1:1 + args.steps], args.steps, sync)
        el2 = torch.tensor([el2], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(el2, op=dist.ReduceOp.MAX)
        el2 = float(el2)
        tlog(f"training-step mode (last-frame dependency cone): {el2 / args.steps * 1e3:.1f} ms/step")
    # third timed region: the trunk WITHOUT its dead-code elimination (every block of the conv tower on every position, the
    # reference's eager graph; the headline of rounds 1-4).  Same outputs and gradients as the headline step
    # (tests/test_network_gpu.py::test_trunk_dead_code_elimination_equals_all_positions).
    el3 = None
    dce = bool(model.score_model.trunk_dce)
    if dce and not args.no_all_positions_mode and args.mode == "all_frames":
        model.score_model.trunk_dce = False
        trainer.last_frame_only = False
        trainer.reducer.reset_structure()
        trainer.update_fn(batches[0])
        trainer.update_fn(batches[0])
        el3, _ = timed_steps(trainer, batches[1:1 + args.steps], args.steps, sync)
        el3 = torch.tensor([el3], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(el3, op=dist.ReduceOp.MAX)
        el3 = float(el3)
        model.score_model.trunk_dce = True
        tlog(f"all-positions mode (no dead-code elimination in the trunk): {el3 / args.steps * 1e3:.1f} ms/step")
    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = world * B * F * args.steps / elapsed
        step_flop = 3.0 * synthetic.step_flops_fwd(F, N, inner_cone=dce) * B
        step_flop_all = 3.0 * synthetic.step_flops_fwd(F, N) * B
        tlog("roofline: %s" % json.dumps(roof))
        line = {
            "metric": "trajectory_frames_per_sec_fwd_bwd_nres%d" % N, "value": round(value, 2), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "BASELINE config 3: synthetic N_res=%d, %d-frame windows, %d windows/GPU, full "
                                   "update_fn (fwd+loss+bwd+grad all-reduce+Adam amsgrad), random-init seeded weights, %s"
                                   % (N, F, B, "one batch repeated" if args.same_batch else
                                      "a fresh synthetic batch every step (staged in HBM before the timed region)"),
                       "windows_per_gpu": B, "frames": F, "n_res": N, "parallelism": "dp%d" % world, "mode": args.mode,
                       "trunk_dead_code_elimination": dce,
                       "zero_frame_skipping": bool(ops_mod.CONV_NZ),
                       "zero_frame_skipping_note":
                           "on (product default, DFOLD_CONV_NZ=0 turns it off): the gradient entering the conv tower's backward is "
                           "scanned for all-zero frames while it is copied into the padded grid; the data- and weight-gradient "
                           "launches skip, on the device and per tile, what is zero by those flags (no contract with the loss, no "
                           "host sync; results bit-identical).  With the reference's loss (last frame only) the full-size backward "
                           "launches of blocks 0 and 3 meet gradients that live on 1 ... 17 of the 32 frames",
                       "trunk_dead_code_elimination_note":
                           "on (product default): the node features of trunk blocks 1 and 2 feed only bb_update, whose output is "
                           "multiplied by 0.0 on every frame but the last (reference ipa_pytorch_dynamic.py:858-869), so their conv "
                           "tower runs on the last frame's dependency cone; every output (all keys, all frames), the loss and "
                           "every gradient equal the all-positions evaluation (`all_positions_mode` below times that one; "
                           "DFOLD_TRUNK_DCE=0 makes it the default)"},
            "loss": {"first_step": round(first_loss, 5), "last_step": round(float(loss), 5),
                     "steps_between": args.warmup + args.steps - 1, "lr": args.lr,
                     "terms_first_timed_step": terms[0], "terms_last_timed_step": terms[-1],
                     "gate_open_in_timed_region": bool(terms[0]["trans_loss"] > 0 and terms[-1]["trans_loss"] > 0),
                     "note": "the reference zeroes the rot / trans terms of a window whose trans_loss reaches 100 "
                             "(train_DFOLD_dynamics.py:1338-1340); gate_open: both terms were live in the first and the last "
                             "timed step (mean over the windows > 0)"},
            # whole step against the MFMA peak: algorithmic fwd+bwd FLOPs of the step / step time / 2.5 PF
            "step_tflop_per_gpu": round(step_flop / 1e12, 2),
            "step_mfma_frac": round(step_flop / (elapsed / args.steps) / 1e12 / BF16_MFMA_PEAK_TFLOPS, 4),
            "roofline": roof,
            "all_positions_mode": None if el3 is None else {
                "value": round(world * B * F * args.steps / el3, 2), "unit": "frames/s",
                "ms_per_step": round(el3 / args.steps * 1e3, 3), "steps": args.steps,
                "step_tflop_per_gpu": round(step_flop_all / 1e12, 2),
                "step_mfma_frac": round(step_flop_all / (el3 / args.steps) / 1e12 / BF16_MFMA_PEAK_TFLOPS, 4),
                "note": "the same update_fn with DFOLD_TRUNK_DCE=0: all four trunk blocks evaluate the conv tower on every position "
                        "(the reference's eager graph, incl. the positions nothing consumes) -- the quantity rounds 1-4 reported as "
                        "the headline; outputs, loss and gradients identical to the headline step"},
            "last_frame_mode": None if el2 is None else {
                "value": round(world * B * F * args.steps / el2, 2), "unit": "frames/s",
                "ms_per_step": round(el2 / args.steps * 1e3, 3), "steps": args.steps,
                "note": "NOT the headline: same update_fn with Trainer(last_frame_only=True) -- conv tower evaluated on the "
                        "dependency cone of the last frame only (the only frame the live loss terms and frame updates read); "
                        "loss, gradients and parameter update identical to the all-frames step (the same conv sums: bit-exact with DFOLD_CONV_SPLITK=0, fp32-reassociated where a thin launch splits K), "
                        "4x fewer conv FLOPs at F=32"},
        }
        if waits is not None:
            line["allreduce_wait_ms"] = waits
            line["param_broadcast_mb"] = round(trainer.bytes_broadcast / 1e6, 1)
            line["dp"] = dp_info
    del batches
    if world == 1 and rank == 0:
        del trainer, model
        torch.cuda.empty_cache()
        def extra(key, fn):      # a side object must never cost the headline line: its failure is reported in its place
            try:
                line[key] = fn()
            except Exception as exc:      # noqa: BLE001
                line[key] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
                tlog(f"extra object {key} FAILED: {line[key]['error']}")
                torch.cuda.synchronize()
                torch.cuda.empty_cache()
        if not args.no_other_configs and args.mode == "all_frames" and (B, F, N) == (8, 32, 256):
            extra("config2", lambda: other_config("BASELINE config 2", 4, 32, 128, dev, max(4, args.steps // 2), tlog))
            extra("config5_one_gpu", lambda: dict(other_config("BASELINE config 5 (per-GPU shard)", 2, 64, 512, dev, max(4, args.steps // 4), tlog),
                                                  parity_note="the reference cannot hold a 64-frame x N_res 512 window (SURVEY 8d): parity at this "
                                                              "shape is the reference's own run at 8 frames x N_res 512 (golden) plus SELF-consistency "
                                                              "at 64 frames (the two step modes agree, cosine > 0.9999: tests/test_parity_baseline_gpu.py"
                                                              "::test_step_vs_reference_golden_config5_nres512)"))
        if not args.no_triangle:
            extra("triangle", lambda: triangle_roofline(dev))
        if not args.no_eval_config:
            extra("config1_eval", lambda: config1_eval(dev, tlog, cpu=not args.no_cpu_baseline))
        if not args.no_neighbours:
            extra("neighbours", lambda: neighbours(dev, tlog))
            extra("hbm_kernels", lambda: hbm_kernels(dev, tlog))
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(max(2, args.cpu_baseline_frames), N)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
