"""Parity at BASELINE-sized configurations.

* the full step (forward, loss, backward) against golden vectors minted from the reference's own code at BASELINE config 1
  (one window, 16 frames x N_res 96: tests/golden/network_F16_N96.npz) and at the run_train.sh window on the headline
  N_res (2 frames x N_res 256: network_F2_N256.npz) -- reference lines train_DFOLD_dynamics.py:660-667,1182-1400;
* parameter gradients against the oracle with the engine's bf16 operand rounding emulated AND the engine's own ReLU masks
  fed back (oracle.RELU_MASK_FEED): the SURVEY 8c bf16 class (rel-L2 <= 3e-2);
* the production-shape conv launches (config-3 grid: 8 windows x 32 frames x N_res 256, 1280 <-> 640 channels) -- forward,
  data gradient, weight gradient (the 256x320 kernel's role 2) and the deterministic split-K of the narrow launches --
  against fp64 sums on sampled output cells / weight slices."""
import numpy as np
import pytest
import torch

from util import canon_quat, compact_window, golden_window, load_golden, max_abs, record_relu_masks, rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _build(F, seed_w, dev):
    from dynamicpdb_amd import synthetic
    from dynamicpdb_amd.data.se3_diffuser import SE3Diffuser
    from dynamicpdb_amd.model.Dfold_network_dynamic import FullScoreNetwork
    conf = synthetic.default_conf(F, cache_dir="/tmp/dfold_igso3_cache/")
    diffuser = SE3Diffuser(conf.diffuser)
    model = FullScoreNetwork(conf.model, diffuser)
    model.load_state_dict(synthetic.seeded_state_dict(seed_w), strict=True)
    return model.to(dev), diffuser


def _step_vs_golden(name, keep=None):
    from dynamicpdb_amd import experiment
    dev = torch.device(DEV)
    g = load_golden(name)
    w, (F, N, seed_w, stride) = golden_window(g)
    model, _ = _build(F, seed_w, dev)
    wd = {k: v.to(dev) for k, v in w.items()}
    out = model({k: v.clone() for k, v in wd.items()})
    if keep is not None:
        keep.update(model=model, window=wd, out=out)
    batch = {k: v[None] for k, v in wd.items()}
    batch["t"] = wd["t"].reshape(1)
    loss, aux = experiment.loss_fn({k: v[None] for k, v in out.items()}, batch)
    loss.backward()
    # forward outputs: the bf16-path class of DESIGN.md section 2
    for k in ("unorm_angles", "rigid_update"):
        assert rel_l2(out[k], g["out_" + k]) < 2e-2, (name, k, rel_l2(out[k], g["out_" + k]))
    # the NORMALISED torsions: dividing a short raw 2-vector by its length amplifies the bf16 noise of the raw output (0.9 % here),
    # so the plain rel-L2 is dominated by the ill-conditioned quarter of the torsions and moves with every change of a summation
    # order (0.018 ... 0.0225 at F16 x N96 across round-5 kernel revisions); on the well-conditioned ones (|raw| >= half the rms
    # length, 76 - 81 % of them) it sits at 0.5 - 0.8 %.  Both are bounded, the meaningful one tightly.
    raw = torch.tensor(g["out_unorm_angles"]).to(dev)
    length = raw.norm(dim=-1)
    well = length >= 0.5 * length.pow(2).mean().sqrt()
    ga = torch.tensor(g["out_angles"]).to(dev)
    assert float(well.float().mean()) > 0.6
    assert rel_l2(out["angles"][well], ga[well].cpu().numpy()) < 1.2e-2, (name, rel_l2(out["angles"][well], ga[well].cpu().numpy()))
    assert rel_l2(out["angles"], g["out_angles"]) < 3e-2, (name, rel_l2(out["angles"], g["out_angles"]))
    assert rel_l2(out["trans_score"], g["out_trans_score"]) < 1e-3
    assert rel_l2(out["rot_score"], g["out_rot_score"]) < 1e-2
    assert max_abs(out["atom14"][..., :3, :], g["out_atom14"][..., :3, :]) < 1e-2
    # side-chain atoms inherit the torsion error x lever arm; a torsion whose raw 2-vector is short is ill-conditioned
    # (normalising amplifies the bf16 noise), so single atoms may move by most of an Angstrom while the RMS stays at the
    # SURVEY 8c coordinate class.  The atom builder itself is exact: the oracle's builder on the ENGINE's frames/torsions
    # reproduces the engine's atoms (kernel check), so this distance is propagated torsion error only.
    d37 = (out["atom37"].cpu().double() - torch.tensor(g["out_atom37"]).double())
    rms = float(d37.pow(2).sum(-1).mean().sqrt())
    far = float((d37.norm(dim=-1) > 0.3).double().mean())
    print(f"[{name}] atom37 rms {rms:.4f} A, max {float(d37.abs().max()):.3f} A, fraction of atoms off by > 0.3 A: {far:.2e}")
    # RMS at the SURVEY 8c coordinate class, a robust count of outliers, and every outlier EXPLAINED: an atom off by more than
    # 0.3 A must sit on a residue with an ill-conditioned torsion -- a raw 2-vector (the reference's own `unorm_angles`) shorter
    # than half the RMS length of all raw vectors, where normalising (openfold/utils/loss.py:58, feats.py torsion frames)
    # turns the bf16-level difference of the raw vector into a rotation of tens of degrees, i.e. up to twice the lever arm
    raw = torch.tensor(g["out_unorm_angles"]).double().norm(dim=-1)                     # [F, N, 7]
    ill = raw < 0.5 * float(raw.pow(2).mean().sqrt())
    off = d37.norm(dim=-1) > 0.3                                                         # [F, N, 37]
    n_off = int(off.sum())
    explained = int((off & ill.any(-1)[..., None]).sum())
    print(f"[{name}] atoms off by > 0.3 A: {n_off}, of which on a residue with an ill-conditioned torsion: {explained}; "
          f"ill-conditioned torsions: {float(ill.double().mean()):.3f} of all")
    assert rms < 5e-2 and far < 5e-3 and float(d37.abs().max()) < 8.0, (rms, float(d37.abs().max()), far)
    assert explained >= n_off - max(1, n_off // 50), (n_off, explained)
    from oracle import dfold_oracle as O
    _, a37 = O.frames_to_atoms(out["rigids"].detach().cpu(), out["angles"].detach().cpu(), w["aatype"].long())
    assert max_abs(out["atom37"], a37) < 2e-3
    assert torch.equal(out["atom37"].cpu() == 0, torch.tensor(g["out_atom37"]) == 0)       # integer gathers bit-exact
    assert max_abs(canon_quat(out["rigids"].cpu()), canon_quat(g["out_rigids"])) < 5e-3
    assert abs(float(loss) - float(g["loss"])) < 2e-2 * abs(float(g["loss"])), (float(loss), float(g["loss"]))
    for k, v in aux.items():
        assert abs(float(v) - float(g["aux_" + k])) < 2e-2 * max(1.0, abs(float(g["aux_" + k]))), k
    P = dict(model.named_parameters())

    def grad_stats(sub, nrm):
        """norm and sampled entries of every parameter gradient against the golden keys `sub`/`nrm`"""
        stats = {}
        for k in g:
            if not k.startswith(sub):
                continue
            n = k[len(sub):]
            gr, ref_norm = P[n].grad, float(g[nrm + n])
            if ref_norm < 1e-6:
                assert gr is None or float(gr.double().norm()) < 1e-4, n
                continue
            assert gr is not None, n
            ref = torch.tensor(g[k]).double()
            mine = (gr.reshape(-1)[::stride] if gr.numel() > 70000 else gr).double().cpu().reshape(ref.shape)
            stats[n] = (abs(float(gr.double().norm()) - ref_norm) / ref_norm, float((mine - ref).norm() / (ref.norm() + 1e-30)), ref_norm)
        return stats

    stats = grad_stats("gsub_", "gnorm_")
    for k in g:
        if k.startswith("gradnone_"):
            assert P[k[9:]].grad is None, k
    if "loss_notorsion" not in g:
        return stats
    # the reference's second run with experiment.torsion_loss_weight = 0 (make_golden.py): the well-conditioned frame
    # terms alone -- these gradients are compared raw (only ReLU-branch flips separate the two sides)
    model.zero_grad(set_to_none=True)
    out0 = model({k: v.clone() for k, v in wd.items()})
    loss0, _ = experiment.loss_fn({k: v[None] for k, v in out0.items()}, batch, torsion_w=0.0)
    assert abs(float(loss0) - float(g["loss_notorsion"])) < 2e-2 * abs(float(g["loss_notorsion"]))
    loss0.backward()
    return stats, grad_stats("g0sub_", "g0norm_")


def _report(stats, tag):
    rel = sorted(v[1] for v in stats.values())
    nrm = sorted(v[0] for v in stats.values())
    worst = max(stats.items(), key=lambda kv: kv[1][1])
    print(f"[{tag}] grad rel-L2 median {rel[len(rel) // 2]:.4f} max {rel[-1]:.4f} ({worst[0]}); "
          f"norm err median {nrm[len(nrm) // 2]:.4f} max {nrm[-1]:.4f}; whole gradient {_whole(stats):.4f}")
    return rel, nrm


def _whole(stats):
    """relative L2 error of the WHOLE parameter gradient (all tensors as one vector; per-tensor errors weighted by the reference
    norms): cosine with the reference's gradient = 1 - e^2 / 2 + O(e^4)"""
    num = sum((v[1] * v[2]) ** 2 for v in stats.values())
    den = sum(v[2] ** 2 for v in stats.values())
    return (num / max(den, 1e-300)) ** 0.5


def _class_spread(ytag):
    """The per-tensor MAXIMUM of the full-loss gradient distance is a noise-dominated statistic at the BASELINE-sized windows: it
    sits on AngleResnet / ipa_0 tensors whose gradient is a heavily cancelling sum over the last frame's rows of terms ~ 1/|raw|
    (the torsion normalisation, openfold/utils/loss.py:58-59).  tests/golden/emulation_spread.json holds what the ORACLE itself
    gives when it merely rounds where the engine's storage class rounds (oracle.EMULATE_BF16_OPERANDS, no mask feed, real loss),
    over four realisations of the rounding noise (operands perturbed by 1e-6 before each bf16 rounding;
    scripts/diag_emulation_spread.py, CPU, run in the build container): e.g. at 8 x 512 medians 0.073 .. 0.081, maxima
    1.25 .. 2.25.  Returns the largest maximum of those realisations (None if the window was not measured)."""
    import json
    import os
    from util import ROOT
    with open(os.path.join(ROOT, "tests", "golden", "emulation_spread.json")) as fh:
        d = json.load(fh)
    return max(r["max"] for r in d[ytag]) if ytag in d else None


def _yardstick(ytag, which):
    """(median, max) of the per-tensor relative L2 distance between the REFERENCE's own fp32 gradients and the same reference
    code run with bf16 parameters and bf16-rounded nn.Linear / nn.Conv2d outputs (tests/golden/bf16_sensitivity*.npz, minted by
    make_golden.py bf16_sensitivity): what the declared storage class costs the reference itself at this window."""
    import os
    from util import ROOT
    for f in ("bf16_sensitivity.npz", "bf16_sensitivity_big.npz"):
        d = np.load(os.path.join(ROOT, "tests", "golden", f))
        if f"{ytag}/{which}/rel" in d.files:
            rel = np.sort(d[f"{ytag}/{which}/rel"])
            return float(rel[len(rel) // 2]), float(rel[-1])
    raise KeyError(ytag)


def _check_big(full, frames, tag, ytag):
    """Raw parameter gradients (no mask feeding, every ReLU / min() branch free to differ, the REAL loss) against the reference's
    fp32 run -- at EVERY golden, the benchmarked 32 x 256 window and 8 x 512 included (round 6; rounds 4-5 asserted the median only
    there):
      * the whole gradient: relative L2 error of all tensors as one vector <= 2e-2 (cosine >= 0.9998; measured 2e-3 .. 5e-3);
      * the typical tensor: median per-tensor rel-L2 within 1.5 x of the YARDSTICK of the storage class -- the reference's own code
        with bf16 parameters and bf16-rounded Linear / Conv outputs against its fp32 self (`_yardstick`: median 0.02 .. 0.08, max
        0.07 .. 0.46; SURVEY 8c's proposed 3e-2 is not attainable by the reference either);
      * the worst tensor: within 3 x of the yardstick's maximum, or -- where the oracle restatement of the engine's storage class
        itself lands further out -- within 1.25 x of the worst realisation of that class (`_class_spread`).  Round 6 measured
        what the maximum is made of (profiles/r6_grad_split_*.txt, r6_emulation_*.json): with the REFERENCE's output gradients
        injected into the engine's backward it barely moves (32 x 256: 1.20 -> 0.71; 8 x 512: 0.42 -> 0.21), i.e. it is not the
        forward difference amplified by the torsion normalisation but bf16 rounding inside cancelling row sums, present in every
        implementation of the class; and it moves by 2 x between realisations of the rounding noise (and did between engine
        revisions that only re-associated fp32 sums: 3.9 / 1.9 / 1.2 at 32 x 256 in rounds 4 / 5 / 6).
    The tight per-tensor statement (<= 3e-2) stays test_gradients_mask_aligned_oracle; the torsion-free run is bounded at 1.5 x /
    2 x of its yardstick at every size."""
    rel, nrm = _report(full, tag + ", full loss")
    ym, yx = _yardstick(ytag, "full")
    spread = _class_spread(ytag)
    print(f"[{tag}] yardstick (reference, bf16 storage vs fp32), full loss: median {ym:.4f} max {yx:.4f}; "
          f"storage class as restated by the oracle, worst realisation: {spread}")
    assert nrm[len(nrm) // 2] < 5e-2
    assert _whole(full) < 2e-2, _whole(full)
    bound = max(3.0 * yx, 1.25 * spread if spread is not None else 0.0)
    assert rel[len(rel) // 2] < 1.5 * ym + 5e-3 and rel[-1] < bound, (rel[len(rel) // 2], ym, rel[-1], yx, spread)
    rel0, nrm0 = _report(frames, tag + ", torsion_loss_weight = 0")
    ym0, yx0 = _yardstick(ytag, "notorsion")
    print(f"[{tag}] yardstick, torsion_loss_weight = 0: median {ym0:.4f} max {yx0:.4f}")
    assert _whole(frames) < 1e-2, _whole(frames)
    assert nrm0[-1] < 0.1 and rel0[len(rel0) // 2] < 1.5 * ym0 + 5e-3 and rel0[-1] < 2.0 * yx0, (rel0[len(rel0) // 2], ym0, rel0[-1], yx0, nrm0[-1])


def test_step_vs_reference_golden_config1():
    """BASELINE config 1 (16 frames x N_res 96, one window) against the reference's own fp32 run."""
    full, frames = _step_vs_golden("network_F16_N96.npz")
    # fp32 reference, bf16-storage engine, every ReLU / min() branch free to differ: the class of DESIGN.md section 2
    # (a flipped branch is an O(1) error on that unit).  The mask-aligned comparison below, at this same size, is the
    # tight one; the full-loss maximum sits on whichever AngleResnet tensor the ill-conditioned torsions hit (measured
    # 0.14 .. 0.33 across engine revisions whose mask-aligned error is unchanged), so it is asserted on the torsion-free run
    _check_big(full, frames, "cfg1 F16 N96", "F16_N96")


def test_step_vs_reference_golden_nres256():
    """run_train.sh window (frame_time 2) at N_res 256 against the reference's own fp32 run."""
    full, frames = _step_vs_golden("network_F2_N256.npz")
    # (2 frames: every gradient is a sum over the 256 positions of the last frame only; full loss: measured median 0.05, max
    # 0.31 .. 0.39 on an AngleResnet weight whose two ReLUs sit right in front of the ill-conditioned torsion normalisation)
    _check_big(full, frames, "F2 N256", "F2_N256")


def test_step_vs_reference_golden_config3_window():
    """One window of BASELINE config 3 / 4 -- 32 frames x N_res 256, exactly the shape bench.py times (its B = 8 windows
    are independent: test_network_gpu::test_batched_equals_independent_windows) -- against the reference's own fp32 run
    (train_DFOLD_dynamics.py:660-667,1182-1400; src/model/Dfold_network_dynamic.py:450-546): every output, the loss
    terms, every parameter gradient (norm + sampled entries)."""
    full, frames = _step_vs_golden("network_F32_N256.npz")
    _check_big(full, frames, "cfg3 F32 N256", "F32_N256")


def test_step_vs_reference_golden_config2_window():
    """One window of BASELINE config 2 (32 frames x N_res 128) against the reference's own fp32 run."""
    full, frames = _step_vs_golden("network_F32_N128.npz")
    _check_big(full, frames, "cfg2 F32 N128", "F32_N128")


def test_step_vs_reference_golden_config5_nres512():
    """BASELINE config 5 is 64 frames x N_res 512; the reference cannot hold its IPA intermediates at that size
    (SURVEY 8d), 8 frames x N_res 512 is what it runs: the engine against that run, then the engine at the full 64 frames
    on the same chain: finite, and its two step modes agree on loss and gradients (the size-independent property)."""
    from dynamicpdb_amd import experiment, synthetic
    keep = {}
    full, frames = _step_vs_golden("network_F8_N512.npz", keep=keep)
    _check_big(full, frames, "cfg5 F8 N512", "F8_N512")
    del keep
    torch.cuda.empty_cache()
    dev = torch.device(DEV)
    F, N = 64, 512
    model, diffuser = _build(F, 25, dev)
    w = synthetic.synthetic_window(26, F, N, t=0.55, diffuser=diffuser)
    batch = {k: v[None].to(dev) for k, v in w.items()}
    batch["t"] = w["t"].to(dev).reshape(1)
    res = []
    for mode in (False, True):
        model.zero_grad(set_to_none=True)
        out = model({k: v.clone() for k, v in batch.items()}, last_frame_only=mode)
        loss, _ = experiment.loss_fn(out, batch)
        loss.backward()
        gr = torch.cat([p.grad.flatten().double() for _, p in sorted(model.named_parameters()) if p.grad is not None])
        assert bool(torch.isfinite(gr).all()) and bool(torch.isfinite(out["rigids"]).all())
        res.append((float(loss), gr, out["rigids"][0, -1].clone(), out["unorm_angles"][0, -1].clone(), out["angles"][0, -1].clone()))
        del out, loss
    assert abs(res[0][0] - res[1][0]) < 5e-3 * abs(res[0][0]), (res[0][0], res[1][0])
    # raw torsion 2-vectors at the bf16 class; their normalised form amplifies the short ones (see _check_big)
    assert rel_l2(res[1][2], res[0][2]) < 2e-3 and rel_l2(res[1][3], res[0][3]) < 2e-2 and rel_l2(res[1][4], res[0][4]) < 6e-2
    cos = float(torch.nn.functional.cosine_similarity(res[0][1], res[1][1], dim=0))
    print(f"[cfg5 F64 N512] loss {res[0][0]:.4f} / {res[1][0]:.4f}, gradient cosine between step modes {cos:.5f}")
    assert cos > 0.99, cos


def test_res_mask_holes_vs_reference_golden():
    """res_mask with holes (10 % dead residues + both chain ends dead; the loader's res_mask is the CA mask,
    src/data/Dfold_data_loader_dynamic.py:248) end to end against the reference's own run: IPA key / query masks, the
    frame-update mask, score masks, the loss normalisers and the whole-tensor MyLayerNorm with dead residues -- both step
    modes."""
    from dynamicpdb_amd import experiment
    keep = {}
    stats = _step_vs_golden("network_F6_N40_holes.npz", keep=keep)
    rel, nrm = _report(stats, "holes F6 N40")
    ym, yx = _yardstick("F6_N40_holes", "full")
    assert nrm[-1] < 0.1 and rel[len(rel) // 2] < 1.5 * ym + 5e-3 and rel[-1] < 3.0 * yx, (rel[len(rel) // 2], ym, rel[-1], yx, nrm[-1])
    model, wd, out_all = keep["model"], keep["window"], keep["out"]
    assert float(wd["res_mask"].min()) == 0 and float(wd["res_mask"][:, 0].max()) == 0 and float(wd["res_mask"][:, -1].max()) == 0
    g_all = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    batch = {k: v[None] for k, v in wd.items()}
    batch["t"] = wd["t"].reshape(1)
    loss_all, _ = experiment.loss_fn({k: v[None] for k, v in out_all.items()}, batch)
    model.zero_grad(set_to_none=True)
    out = model({k: v.clone() for k, v in batch.items()}, last_frame_only=True)
    loss, _ = experiment.loss_fn(out, batch)
    loss.backward()
    assert abs(float(loss) - float(loss_all)) < 5e-3 * abs(float(loss_all))
    for k in ("angles", "unorm_angles", "rigids", "rot_score", "trans_score"):
        assert rel_l2(out[k][0, -1], out_all[k][-1]) < 2e-2, k
    a = torch.cat([g_all[n].flatten().double() for n in sorted(g_all)])
    b = torch.cat([dict(model.named_parameters())[n].grad.flatten().double() for n in sorted(g_all)])
    assert float(torch.nn.functional.cosine_similarity(a, b, dim=0)) > 0.99


@pytest.mark.parametrize("gname", ["network_F3_N16.npz", "network_F8_N16.npz", "network_F2_N256.npz", "network_F16_N96.npz",
                                   "network_F6_N40_holes.npz", "network_F32_N128.npz", "network_F32_N256.npz"])
def test_gradients_mask_aligned_oracle(gname):
    """Every parameter gradient of the full step against the oracle that (a) rounds values AND gradients to bf16 at the
    engine's storage points (operands of every dense contraction, activation gradients between kernels:
    oracle._q / _qg) and (b) takes each ReLU's branch from the ENGINE's stored mask: what is left is kernel arithmetic
    (fp32 accumulation order, roundings that straddle a bf16 tie) -- SURVEY 8c's bf16 class, rel-L2 <= 3e-2 per tensor."""
    from oracle import dfold_oracle as O
    from dynamicpdb_amd import experiment, ops, synthetic
    dev = torch.device(DEV)
    g = load_golden(gname)
    F, N, seed_w = [int(v) for v in g["meta"][:3]]
    model, _ = _build(F, seed_w, dev)
    w = {k: v.to(dev) for k, v in golden_window(g)[0].items()}   # BASELINE-sized captures: inputs from the seed
    with record_relu_masks() as masks:
        out = model({k: v.clone() for k, v in w.items()})
    assert len(masks) == 4 * 8 + 7, len(masks)       # 4 tower applications x 8 ReLUs + AngleResnet's 7
    batch = {k: v[None] for k, v in w.items()}
    batch["t"] = w["t"].reshape(1)

    def readout(loss_fn, o, b, lead):
        """translation-x0 + rotation-score terms of the reference loss (train_DFOLD_dynamics.py:1248-1340) plus a SMOOTH
        torsion read-out on the raw (un-normalised) 2-vectors of the last frame.  The reference's torsion term normalises
        them first (openfold/utils/loss.py:58-59): its gradient ~ 1/|raw| is dominated by the few torsions whose raw
        vector is short, where the forward's own bf16-level difference is amplified 10-100x -- a conditioning property
        of that read-out (measured: it alone moves whole gradient tensors by 4 % at 112 torsions, 40 % at 1792), not of
        any kernel.  The real loss is compared in test_step_vs_reference_golden_* above."""
        l, _ = loss_fn(o, b, torsion_w=0.0)
        raw, gt = o["unorm_angles"], b["torsion_angles_sin_cos"].to(o["unorm_angles"].dtype)
        m = b["torsion_angles_mask"].to(raw.dtype)[..., None]
        sel = (slice(None), -1) if lead else (-1,)
        return l + (((raw - gt) ** 2) * m)[sel].mean()

    loss = readout(experiment.loss_fn, {k: v[None] for k, v in out.items()}, batch, True)
    loss.backward()
    O.EMULATE_BF16_OPERANDS, O.RELU_MASK_FEED = True, [m.clone() for m in masks]
    try:
        Pq = {k: v.clone().requires_grad_(True) for k, v in synthetic.seeded_state_dict(seed_w).items()}
        wc = {k: v.cpu() for k, v in w.items()}
        oo = O.full_score_network(Pq, O.Schedules(), wc)
        lq = readout(O.loss_fn, oo, wc, False)
        assert not O.RELU_MASK_FEED, "the oracle consumed fewer masks than the engine recorded"
        lq.backward()
    finally:
        O.EMULATE_BF16_OPERANDS, O.RELU_MASK_FEED = False, None
    assert abs(float(loss) - float(lq)) < 1e-2 * abs(float(lq))
    for k in ("unorm_angles", "rigid_update"):
        assert rel_l2(out[k], oo[k]) < 2e-2, k
    assert rel_l2(out["angles"], oo["angles"]) < 5e-2          # normalised: short raw vectors amplify (see readout)
    errs = {}
    for name, p in model.named_parameters():
        if p.grad is None or Pq[name].grad is None or float(Pq[name].grad.norm()) < 1e-6:
            continue
        errs[name] = rel_l2(p.grad, Pq[name].grad)
    vals = sorted(errs.values())
    worst = max(errs.items(), key=lambda kv: kv[1])
    print(f"[mask-aligned {gname}] rel-L2 median {vals[len(vals) // 2]:.4f} max {worst[1]:.4f} ({worst[0]})")
    top = sorted(errs.items(), key=lambda kv: -kv[1])[:6]
    print("   worst:", ", ".join(f"{k.replace('score_model.', '')} {v:.4f}" for k, v in top))
    bad = {k: v for k, v in errs.items() if v >= 3e-2}
    assert not bad and vals[len(vals) // 2] < 2e-2, (bad, vals[len(vals) // 2])


FD_FRACS = (2e-4, 1e-3, 5e-3, 1e-2)


def test_directional_finite_difference_of_the_engine_loss():
    """An UNCONDITIONAL gradient statement: the engine's parameter gradient is the gradient of the
    engine's own loss -- no oracle, no mask feeding, the REAL loss incl. the torsion term (train_DFOLD_dynamics.py:1182-1400,
    openfold/utils/loss.py:52-76) at BASELINE config 1.  Central differences (L(theta + eps v) - L(theta - eps v)) / 2 eps of
    the loss (its terms accumulated in fp64 from the engine's fp32 outputs) against <grad L, v> for directions v = the
    claimed gradient restricted to one parameter group (and to a random half of its entries): FD / claimed = 1 + O(|e|^2 /
    |g|^2) for a gradient error e, so a ratio within 4 % bounds the relative L2 error of that group's gradient by 0.2 --
    WITHOUT any ReLU-branch alignment (a flipped branch at theta +- eps v is part of the function being differentiated).

    Step sizes (eps such that the predicted change of the loss is `frac` of it on each side), declared here, the whole sweep
    is printed: bf16 weight / activation storage makes the loss a staircase, whose noise in the difference quotient falls
    as 1 / eps -- measured (round 4, deterministic run to run): at frac 2e-4 the quotients of groups upstream of many bf16
    stages scatter by +-60 %, at 1e-3 by +-15 %, from 5e-3 on by < 3 % -- while curvature grows as eps^2: only the angle
    resnet, whose torsion term normalises the raw 2-vectors (gradient ~ 1 / |raw|, linear range ~ |raw| of the shortest
    vectors), leaves its linear range above 1e-3 (quotient 0.84 at 1e-3, 0.34 at 5e-3) and, sitting behind few bf16 stages,
    is clean at 2e-4.  Hence: angle resnet at 2e-4, every other group the mean of 5e-3 and 1e-2."""
    from dynamicpdb_amd import experiment
    dev = torch.device(DEV)
    g = load_golden("network_F16_N96.npz")
    w, (F, N, seed_w, _) = golden_window(g)
    model, _ = _build(F, seed_w, dev)
    wd = {k: v.to(dev) for k, v in w.items()}
    batch = {k: v[None] for k, v in wd.items()}
    batch["t"] = wd["t"].reshape(1)

    def loss_of(backward=False):
        with torch.set_grad_enabled(backward):
            out = model({k: v.clone() for k, v in wd.items()})
            loss, _ = experiment.loss_fn({k: v[None].double() for k, v in out.items()}, batch)      # fp64 accumulation
            if backward:
                loss.backward()
        return float(loss.detach())

    L0 = loss_of(backward=True)
    noise = abs(loss_of() - loss_of())                     # run-to-run (fp32 atomics): the floor under every difference
    named = [(n, p) for n, p in model.named_parameters() if p.grad is not None]
    grads = {n: p.grad.detach().clone() for n, p in named}
    groups = {
        "conv tower": lambda n: ".conv_0." in n,
        "IPA blocks 0-1": lambda n: ".ipa_0." in n or ".ipa_1." in n,
        "IPA blocks 2-3": lambda n: ".ipa_2." in n or ".ipa_3." in n,
        "angle resnet": lambda n: ".angle_resnet." in n,
        "embedders + expand": lambda n: "embeder" in n or n.startswith("expand_"),
        "frame update heads": lambda n: ".bb_update_" in n,
        "all parameters": lambda n: True,
    }
    gen = torch.Generator(device="cpu").manual_seed(11)
    rows = []
    for gname, sel in groups.items():
        for half in (False, True):
            v = {}
            for n, p in named:
                if not sel(n):
                    continue
                d = grads[n].double()
                if half:
                    d = d * (torch.rand(d.shape, generator=gen) < 0.5).to(d.device, d.dtype)
                v[n] = d
            claimed = sum(float((grads[n].double() * d).sum()) for n, d in v.items())        # <grad L, v>
            vnorm2 = sum(float((d * d).sum()) for d in v.values())
            if claimed <= 0 or vnorm2 == 0:
                continue
            ratios = []
            orig = {n: p.detach().clone() for n, p in named if n in v}
            for frac in FD_FRACS:
                eps = frac * L0 / claimed                     # predicted change of the loss on each side: frac * L0
                vals = []
                for sgn in (1.0, -1.0):
                    with torch.no_grad():
                        for n, p in named:
                            if n in v:
                                p.copy_((orig[n].double() + sgn * eps * v[n]).to(p.dtype))
                    vals.append(loss_of())
                ratios.append((vals[0] - vals[1]) / (2 * eps) / claimed)
            with torch.no_grad():
                for n, p in named:
                    if n in v:
                        p.copy_(orig[n])
            del orig
            rows.append((gname, half, claimed / vnorm2 ** 0.5, ratios))
    print(f"[FD cfg1] loss {L0:.5f}, run-to-run noise of the loss {noise:.2e} (relative {noise / L0:.1e})")
    for name, half, gn, r in rows:
        print(f"[FD cfg1] {name + (' (random half)' if half else ''):38s} <g,v>/|v| {gn:10.4f}   FD / claimed at {FD_FRACS} of the loss: "
              + " / ".join(f"{x:.4f}" for x in r))
    assert len(rows) >= 12
    for name, half, _, r in rows:
        if name == "angle resnet":
            # (its one usable step, 2e-4, sits on the staircase noise: single directions scatter by up to +-8 % across kernel
            #  revisions that change a summation order -- 0.925 / 1.031 for the two directions in round 5, 0.994 / 1.063 and, with
            #  the fused angle head, 0.896 / 1.055 in round 6; each is bounded at 12 %, their mean, where the noise averages, at 4 %)
            assert abs(r[0] - 1.0) < 0.12, (name, half, r)
            continue
        q = 0.5 * (r[2] + r[3])
        assert abs(q - 1.0) < 4e-2, (name, half, r)
    ar = [r[0] for name, _, _, r in rows if name == "angle resnet"]
    assert len(ar) == 2 and abs(sum(ar) / 2 - 1.0) < 4e-2, ar
    assert abs(loss_of() - L0) <= 10 * noise + 1e-6 * abs(L0)       # parameters restored


# ------------------------------------------------------------------------------------------------------------------
# production-shape conv launches vs fp64 on samples
# ------------------------------------------------------------------------------------------------------------------

def _pack(w, dev):
    from ctypes import c_int32
    from dynamicpdb_amd import _lib, ops
    CO, CI = w.shape[:2]
    wf = torch.empty((CO, 25, CI), dtype=torch.bfloat16, device=dev)
    wd = torch.empty((CI, 25, CO), dtype=torch.bfloat16, device=dev)
    _lib.check(_lib.lib().dfold_conv_weight_pack(ops._p(w), ops._p(wf), ops._p(wd), c_int32(CO), c_int32(CI), _lib.stream()), "pack")
    return wf, wd


def _patch_rows(xpad, cells):
    """xpad: padded grid [Wn,Fp,Wp,C] (border zero); cells [k,3] (window, frame, residue) -> fp64 [k, 25, C]: the 5x5
    neighbourhood of each cell, tap index = (df+2)*5 + (dn+2)."""
    w, f, n = cells[:, 0], cells[:, 1], cells[:, 2]
    rows = []
    for df in range(5):
        for dn in range(5):
            rows.append(xpad[w, f + df, n + dn].double())
    return torch.stack(rows, 1)


def _sample_cells(Wn, F, N, k, gen):
    cells = torch.stack([torch.randint(0, Wn, (k,), generator=gen), torch.randint(0, F, (k,), generator=gen),
                         torch.randint(0, N, (k,), generator=gen)], 1)
    # force the grid corners / edges in (padding taps) and the tile seams of the 256-row M tiles
    cells[0] = torch.tensor([0, 0, 0])
    cells[1] = torch.tensor([Wn - 1, F - 1, N - 1])
    cells[2] = torch.tensor([0, F - 1, 0])
    cells[3] = torch.tensor([Wn - 1, 0, N - 1])
    cells[4] = torch.tensor([3, min(1, F - 1), 255 % N])
    cells[5] = torch.tensor([3, min(2, F - 1), 0])
    return cells


@pytest.mark.parametrize("tn", [False, True])
@pytest.mark.parametrize("CI,CO", [(1280, 640), (640, 1280)])
def test_conv_fwd_dgrad_wgrad_config3_grid_vs_fp64(CI, CO, tn):
    """One conv layer of the tower on the config-3 grid (M = 8*32*256 = 65536 rows): forward (bias + ReLU epilogue),
    data gradient (tap-flipped weights, ReLU-mask epilogue) and weight / bias gradient (dfold_mfma_gemm320_kernel role 2,
    the transposed-accumulator form for 1280 -> 640) against fp64 sums over the same bf16 operands on sampled cells and
    weight slices."""
    from dynamicpdb_amd import ops
    dev = torch.device(DEV)
    Wn, F, N = 8, 32, 256
    g = ops.Grid(Wn, F, N, dev)
    gen = torch.Generator(device="cpu").manual_seed(5 + CI)
    w = (torch.randn(CO, CI, 5, 5, generator=gen) * (2.0 / (25 * CI)) ** 0.5).to(dev)
    bias = (0.1 * torch.randn(CO, generator=gen)).to(dev)
    wf, wd = _pack(w, dev)
    wq = w.to(torch.bfloat16).double()                                  # [CO,CI,5,5]
    x, gy = g.alloc(CI), g.alloc(CO)
    g.interior(x).copy_(torch.randn(Wn, F, N, CI, generator=gen).to(torch.bfloat16))
    g.interior(gy).copy_((torch.randn(Wn, F, N, CO, generator=gen) * 0.1).to(torch.bfloat16))
    cells = _sample_cells(Wn, F, N, 96, gen).to(dev)
    # ---- forward
    y = g.alloc(CO)
    ops.conv5x5_fwd(g, x, wf, bias, y, relu=True)
    patch = _patch_rows(x, cells)                                        # [k,25,CI]
    ref = torch.einsum("ktc,oct->ko", patch, wq.reshape(CO, CI, 25)) + bias.double()
    got = y[cells[:, 0], cells[:, 1] + 2, cells[:, 2] + 2].double()
    assert rel_l2(got, ref.clamp_min(0)) < 4e-3, ("fwd", rel_l2(got, ref.clamp_min(0)))
    assert float(y[:, :2].abs().max()) == 0 and float(y[:, :, -2:].abs().max()) == 0
    # ---- data gradient: dx[cell, ci] = sum_{tap,co} gy[cell - tap, co] w[co, ci, tap]  (then the ReLU mask of x)
    dx = g.alloc(CI)
    ops.conv5x5_fwd(g, gy, wd, None, dx, relu=False, relu_mask=x)
    gpatch = _patch_rows(gy, cells)                                      # gy at cell + (df-2, dn-2)
    wflip = wq.flip(2, 3).reshape(CO, CI, 25)                            # tap (df,dn) pairs with w[.., 4-df, 4-dn]
    refd = torch.einsum("kto,oct->kc", gpatch, wflip)
    mask = (x[cells[:, 0], cells[:, 1] + 2, cells[:, 2] + 2] > 0).double()
    gotd = dx[cells[:, 0], cells[:, 1] + 2, cells[:, 2] + 2].double()
    assert rel_l2(gotd, refd * mask) < 4e-3, ("dgrad", rel_l2(gotd, refd * mask))
    # ---- weight / bias gradient: dW[co,ci,df,dn] = sum_cells gy[cell,co] x[cell + (df-2,dn-2), ci]
    tower_like = ops.Workspace(dev)
    big, small = max(CI, CO), min(CI, CO)
    dwg = torch.zeros((big, 25, small), dtype=torch.float32, device=dev)
    db = torch.zeros(CO, dtype=torch.float32, device=dev)
    ops.conv5x5_wgrad(g, x, gy, dwg, tower_like, accumulate=False, bias_grad=db, tn=tn)
    ops.conv5x5_wgrad(g, x, gy, dwg, tower_like, accumulate=True, bias_grad=db, tn=tn)      # accumulates (shared tower: 4 uses)
    from ctypes import c_int32
    from dynamicpdb_amd import _lib
    gw = torch.empty((CO, CI, 5, 5), dtype=torch.float32, device=dev)
    _lib.check(_lib.lib().dfold_conv_wgrad_unpack(ops._p(dwg), ops._p(gw), c_int32(CO), c_int32(CI), c_int32(0),
                                                  c_int32(1 if CI > CO else 0), _lib.stream()), "unpack")
    gyi = g.interior(gy).double().reshape(-1, CO)                        # [M,CO]
    ci_sel = torch.tensor([0, 1, 63, 64, 319, 320, CI - 1], device=dev)
    for (df, dn) in ((0, 0), (2, 2), (4, 4), (0, 4), (3, 1)):
        xs = x[:, df:df + F, dn:dn + N][..., ci_sel].double().reshape(-1, len(ci_sel))   # x at cell + (df-2, dn-2)
        refw = 2 * gyi.t() @ xs                                           # [CO, sel]
        assert rel_l2(gw[:, ci_sel, df, dn], refw) < 2e-3, ("wgrad", df, dn, rel_l2(gw[:, ci_sel, df, dn], refw))
    assert rel_l2(db, 2 * gyi.sum(0)) < 2e-3


def test_conv_splitk_narrow_launch_config3_vs_fp64():
    """The narrow launches of the training-step mode at production shape (8 windows x N_res 256, 5 output frames of 32,
    640 -> 1280 and 1280 -> 640 channels): deterministic split-K (partial tiles in a workspace, last arriver reduces) against
    fp64 sums on sampled cells."""
    from dynamicpdb_amd import ops
    dev = torch.device(DEV)
    Wn, F, N = 8, 32, 256
    g = ops.Grid(Wn, F, N, dev)
    gen = torch.Generator(device="cpu").manual_seed(17)
    for (CI, CO) in ((1280, 640), (640, 1280)):
        w = (torch.randn(CO, CI, 5, 5, generator=gen) * (2.0 / (25 * CI)) ** 0.5).to(dev)
        bias = (0.1 * torch.randn(CO, generator=gen)).to(dev)
        wf, _ = _pack(w, dev)
        wq = w.to(torch.bfloat16).double()
        x = g.alloc(CI)
        g.interior(x).copy_(torch.randn(Wn, F, N, CI, generator=gen).to(torch.bfloat16))
        ws = ops.Workspace(dev)
        nf = 5 if CO == 640 else 1          # 80 resp. 32 output tiles for 256 CUs: the cost model splits K
        S = ops.conv_splitk(Wn * nf * N, CO, CI, dev)
        assert S > 1, (CI, CO)
        y = g.alloc(CO)
        ops.conv5x5_fwd(g, x, wf, bias, y, relu=True, f_lo=F - nf, nf=nf, ws=ws)
        cells = _sample_cells(Wn, nf, N, 64, gen)
        cells[:, 1] += F - nf
        cells = cells.to(dev)
        ref = torch.einsum("ktc,oct->ko", _patch_rows(x, cells), wq.reshape(CO, CI, 25)) + bias.double()
        got = y[cells[:, 0], cells[:, 1] + 2, cells[:, 2] + 2].double()
        assert rel_l2(got, ref.clamp_min(0)) < 4e-3, (CI, CO, S)
        assert float(y[:, : 2 + F - nf].abs().max()) == 0
