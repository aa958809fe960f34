"""The training step over several optimizer steps, the sampler loop, and the score heads, on the device.

* update_fn with the fused Adam must actually train: the bf16 weight caches follow the in-kernel parameter update
  (optim.py: the kernel writes through raw pointers) and the trajectory matches torch.optim.Adam driving the
  same model;
* a no_grad pass (sampling / self-conditioning / evaluation) must not disturb the next training step's conv-tower
  gradients (functional.py: the shared tower counts pending backward passes);
* Experiment.inference_fn (train_DFOLD_dynamics.py:1425-1547) against the reference's own run with recorded draws
  (tests/golden/sampler_F3_N16.npz);
* SE3Diffuser.calc_rot_score / calc_trans_score (src/data/se3_diffuser.py:115-125) on the committed reference outputs
  (tests/golden/diffuser.npz, fm{i}_calc_*)."""
import numpy as np
import pytest
import torch

from util import ROOT, canon_quat, load_golden, max_abs, record_relu_masks, rel_l2, window_from_golden

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


def _batch(diffuser, B, F, N, dev, seed=40):
    from dynamicpdb_amd import synthetic
    ws = [synthetic.synthetic_window(seed + i, F, N, t=0.3 + 0.1 * i, diffuser=diffuser) for i in range(B)]
    batch = {k: torch.stack([w[k] for w in ws]).to(dev) for k in ws[0]}
    batch["t"] = torch.cat([w["t"] for w in ws]).to(dev)
    return batch


def test_update_fn_trains_and_matches_torch_adam():
    """Three update_fn steps on one batch: (a) the loss moves (the forward sees the stepped weights), (b) the loss
    trajectory and the final parameters equal those of the same model driven by torch.optim.Adam(amsgrad=True)."""
    from dynamicpdb_amd import experiment
    dev = torch.device(DEV)
    F, N, B = 4, 16, 2
    lr = 2e-3
    traj, finals = [], []
    for fused in (True, False):
        model, diffuser = _build(F, 3, dev)
        batch = _batch(diffuser, B, F, N, dev)
        tr = experiment.Trainer(model, lr=lr, last_frame_only=False)
        if not fused:
            tr.opt = torch.optim.Adam(tr.params, lr=lr, amsgrad=True)
        losses = [float(tr.update_fn(batch)[0]) for _ in range(4)]
        traj.append(losses)
        finals.append({n: p.detach().clone() for n, p in model.named_parameters()})
    fused_l, torch_l = traj
    assert abs(fused_l[1] - fused_l[0]) > 1e-3 * abs(fused_l[0]), fused_l        # step 2 sees step 1's update
    assert fused_l[-1] < fused_l[0], fused_l                                      # and it is a descent direction
    for a, b in zip(fused_l, torch_l):
        assert abs(a - b) < 2e-2 * abs(b), (fused_l, torch_l)
    # parameters: same update rule on (nearly) the same gradients.  An Adam step has size ~lr whatever the gradient's
    # magnitude, so entries whose gradient is summation-order noise may move in opposite directions (<= 2 lr per step);
    # the update as a whole must point the same way.
    from dynamicpdb_amd import synthetic
    init = synthetic.seeded_state_dict(3)
    worst = max(float((finals[0][n] - finals[1][n]).abs().max()) for n in finals[0])
    assert worst <= 2 * 4 * lr * 1.01, worst
    for n in ("score_model.trunk.conv_0.conv1.0.weight", "score_model.trunk.ipa_0.linear_q.weight",
              "score_model.angle_resnet.linear_in.weight"):
        da, db = (finals[0][n].cpu() - init[n]).flatten().double(), (finals[1][n].cpu() - init[n]).flatten().double()
        assert float(da.abs().max()) > 0.5 * lr, n                                # it did get stepped
        assert float(torch.nn.functional.cosine_similarity(da, db, dim=0)) > 0.98, n


def test_no_grad_pass_does_not_disturb_next_training_step():
    """inference-style forward (no_grad), then update_fn: every conv-tower parameter gets a gradient equal to the one of
    a fresh model's step; repeated, with an aborted step in between (forward without backward)."""
    from dynamicpdb_amd import experiment
    dev = torch.device(DEV)
    F, N, B = 4, 16, 1
    model, diffuser = _build(F, 6, dev)
    batch = _batch(diffuser, B, F, N, dev, seed=50)
    tr = experiment.Trainer(model, lr=1e-3, last_frame_only=True)
    tr.update_fn(batch, step_optimizer=False)
    ref = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    assert any("conv_0" in n for n in ref)
    with torch.no_grad():
        model(batch)
        model(batch, last_frame_only=True)
    tower = model.score_model.trunk["conv_0"].tower()
    assert tower.pending == 0
    tr.update_fn(batch, step_optimizer=False)
    for n, p in model.named_parameters():
        if n in ref:
            assert p.grad is not None, n
            assert rel_l2(p.grad, ref[n]) < 1e-3, n                    # same launches (fp32 atomics reorder some sums)
    # a forward whose backward never runs (exception between loss and backward in user code) must not leak either
    out = model(batch, last_frame_only=True)
    assert tower.pending == 4
    del out
    tr.update_fn(batch, step_optimizer=False)
    assert tower.pending == 0
    for n, p in model.named_parameters():
        if "conv_0" in n:
            assert rel_l2(p.grad, ref[n]) < 1e-3, n
    # the activation grids of a tower application live in a persistent pool: a graph kept across a later step must not be
    # differentiated on the overwritten buffers -- it raises instead of returning wrong gradients
    stale = model(batch, last_frame_only=True)["rigid_update"].float().sum()
    tr.update_fn(batch, step_optimizer=False)
    with pytest.raises(RuntimeError, match="overwritten"):
        stale.backward()
    tr.update_fn(batch, step_optimizer=False)
    for n, p in model.named_parameters():
        if "conv_0" in n:
            assert rel_l2(p.grad, ref[n]) < 1e-3, n


def test_inference_fn_vs_reference_golden():
    """Device-resident sampler against the reference's Experiment.inference_fn on the same prior sample, weights and
    normal draws (3 reverse steps incl. the self-conditioning pass): all four trajectories."""
    from dynamicpdb_amd import experiment
    dev = torch.device(DEV)
    g = load_golden("sampler_F3_N16.npz")
    F, N, seed_w, _, num_t = [int(v) for v in g["meta"]]
    model, diffuser = _build(F, seed_w, dev)
    init = {k[3:]: torch.tensor(v).to(dev) for k, v in g.items() if k.startswith("in_")}
    draws = [(g[f"z_rot_{i}"], g[f"z_trans_{i}"]) for i in range(num_t - 1)]
    ret = experiment.inference_fn(model, diffuser, init, num_t=num_t, min_t=0.01, center=True, aux_traj=True,
                                  self_condition=True, noise_scale=float(g["noise_scale"][0]), z_draws=draws)
    assert ret["prot_traj"].shape == g["out_prot_traj"].shape
    # index 0 = t = min_t (last step), index -1 = first reverse step (trajectories are flipped)
    rt, rr = torch.tensor(ret["rigid_traj"].copy()), torch.tensor(g["out_rigid_traj"])
    assert max_abs(canon_quat(rt), canon_quat(rr)) < 5e-3
    assert max_abs(rt[..., 4:], rr[..., 4:]) < 1e-2                                   # Angstrom
    assert max_abs(ret["trans_traj"].copy(), g["out_trans_traj"]) < 1e-2
    bb = [0, 1, 2, 4]                                                                 # N, CA, C, O: frame-only atoms
    assert max_abs(ret["prot_traj"][..., bb[:3], :].copy(), g["out_prot_traj"][..., bb[:3], :]) < 2e-2
    assert max_abs(ret["prot_traj"].copy(), g["out_prot_traj"]) < 0.3                 # side chains: torsion error x lever arm
    assert max_abs(ret["rigid_0_traj"].copy(), g["out_rigid_0_traj"]) < 0.3
    assert np.array_equal(ret["prot_traj"] == 0, g["out_prot_traj"] == 0)             # atom masks / gathers exact
    # normalised torsions: a short raw 2-vector amplifies the bf16 noise (same conditioning as the side-chain atoms)
    assert rel_l2(ret["psi_pred"], g["out_psi_pred"]) < 5e-2


def test_inference_fn_vs_reference_golden_config1():
    """BASELINE config 1 IS the reference's eval configuration (run_eval.sh:4-17: 16-frame window, N_res ~ 96, num_t = 10,
    noise_scale = 0.1): the device-resident sampler against the reference's own Experiment.inference_fn
    (train_DFOLD_dynamics.py:1425-1547) at that size -- same prior sample, weights and normal draws (numpy stream in the
    reference's order), ten model forwards + the self-conditioning pass + nine reverse steps.  Inputs other than the prior are
    regenerated from the seed and pinned by the minted checksum."""
    from dynamicpdb_amd import experiment, synthetic
    dev = torch.device(DEV)
    g = load_golden("sampler_F16_N96.npz")
    F, N, seed_w, seed_x, num_t = [int(v) for v in g["meta"]]
    w = synthetic.synthetic_window(seed_x, F, N, t=1.0, diffuser=None)
    sums = np.array([float(np.asarray(w[k].numpy(), dtype=np.float64).sum()) for k in sorted(w)])
    assert np.array_equal(sums, g["in_checksum"]), "synthetic_window no longer reproduces the minted inputs"
    model, diffuser = _build(F, seed_w, dev)
    init = {k: v.to(dev) for k, v in w.items()}
    init["rigids_t"] = torch.tensor(g["in_rigids_t"]).float().to(dev)
    np.random.seed(int(g["seed_z"][0]))
    draws = [(np.random.normal(size=(F, N, 3)), np.random.normal(size=(F, N, 3))) for _ in range(num_t - 1)]
    ret = experiment.inference_fn(model, diffuser, init, num_t=num_t, min_t=0.01, center=True, aux_traj=True,
                                  self_condition=True, noise_scale=float(g["noise_scale"][0]), z_draws=draws)
    keep = g["traj_steps"]
    rt, rr = torch.tensor(ret["rigid_traj"].copy()), torch.tensor(g["out_rigid_traj"])
    errs = dict(quat=max_abs(canon_quat(rt), canon_quat(rr)), trans=max_abs(rt[..., 4:], rr[..., 4:]),
                trans_traj=max_abs(ret["trans_traj"].copy(), g["out_trans_traj"]),
                backbone=max_abs(ret["prot_traj"][keep][..., :3, :].copy(), g["out_prot_traj"][..., :3, :]),
                atoms_rms=float(np.sqrt(np.mean((ret["prot_traj"][keep] - g["out_prot_traj"]) ** 2))),
                psi=rel_l2(ret["psi_pred"], g["out_psi_pred"]))
    print("config-1 sampler vs the reference:", {k: round(float(v), 5) for k, v in errs.items()})
    assert errs["quat"] < 1e-2 and errs["trans"] < 2e-2 and errs["trans_traj"] < 2e-2     # frames: Angstrom
    assert errs["backbone"] < 4e-2 and errs["atoms_rms"] < 5e-2 and errs["psi"] < 5e-2
    assert np.array_equal(ret["prot_traj"][keep] == 0, g["out_prot_traj"] == 0)           # atom masks / gathers exact


def test_dropin_plain_torch_step_without_the_engine_trainer():
    """One training step of the drop-in network the way the REFERENCE drives it (train_DFOLD_dynamics.py:660-667, 1182-1400),
    with nothing of dynamicpdb_amd.experiment / dp in the loop: reference-shaped [F, N, ...] inputs, plain `model(batch)`, the
    reference's loss formulas (the oracle restatement, which runs on device tensors), `loss.backward()` through the HIP
    autograd nodes, `torch.optim.Adam(amsgrad=True).step()`.  Outputs / loss / gradients against the reference-minted golden;
    the outputs, loss and aux terms are dumped (gpurun_out/dropin_step_F3_N16.npz -> tests/golden/) for the build-container
    half of the check: tests/test_reference_experiment.py feeds them to the reference's OWN Experiment.loss_fn."""
    import os
    from oracle import dfold_oracle as O
    dev = torch.device(DEV)
    g = load_golden("network_F3_N16.npz")
    F, N, seed_w = [int(v) for v in g["meta"][:3]]
    model, _ = _build(F, seed_w, dev)
    model.train()
    w = window_from_golden(g, dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, amsgrad=True)
    before = {k: v.detach().clone() for k, v in model.named_parameters()}
    opt.zero_grad()
    out = model({k: v.clone() for k, v in w.items()})                    # [F, N, ...] in, [F, N, ...] out: no batch axis
    for k in ("angles", "rot_score", "trans_score", "rigids", "atom37"):
        assert tuple(out[k].shape) == tuple(g["out_" + k].shape), k
    loss, aux = O.loss_fn(out, w)
    loss.backward()
    assert abs(float(loss.detach()) - float(g["loss"])) < 2e-2 * abs(float(g["loss"]))
    P = dict(model.named_parameters())
    worst = 0.0
    for k in g:
        if k.startswith("gsub_") and float(g["gnorm_" + k[5:]]) > 1e-6:
            gr, ref = P[k[5:]].grad, torch.tensor(g[k]).double()
            mine = (gr.reshape(-1)[::9973] if gr.numel() > 70000 else gr).double().cpu().reshape(ref.shape)
            worst = max(worst, float((mine - ref).norm() / (ref.norm() + 1e-30)))
    assert worst < 0.25, worst                                         # the class of test_network_gpu::test_gradients
    opt.step()
    moved = [k for k, v in model.named_parameters() if v.grad is not None and not torch.equal(v.detach(), before[k])]
    assert len(moved) > 100 and all(torch.isfinite(v).all() for v in model.parameters())
    # amsgrad's first step moves every parameter with a gradient by lr * sign(g) (|update| = lr up to eps)
    k0 = "score_model.trunk.ipa_0.linear_q.weight"
    step0 = (P[k0].detach() - before[k0])[P[k0].grad.abs() > 1e-6]
    assert float((step0.abs() - 1e-4).abs().max()) < 2e-6
    dump = {f"out_{k}": v.detach().cpu().numpy() for k, v in out.items() if torch.is_tensor(v)}
    dump["loss"] = np.array(float(loss.detach()))
    for k, v in aux.items():
        if torch.is_tensor(v) and v.numel() == 1:
            dump[f"aux_{k}"] = np.array(float(v))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    np.savez_compressed(os.path.join(ROOT, "gpurun_out", "dropin_step_F3_N16.npz"), **dump)


def test_score_heads_vs_reference_golden():
    """calc_rot_score (fp32 quaternions in, float64 score out, the reference's mixed-precision series) and
    calc_trans_score on device tensors against the reference's outputs."""
    from dynamicpdb_amd import synthetic
    from dynamicpdb_amd.data.se3_diffuser import SE3Diffuser
    dev = torch.device(DEV)
    g = load_golden("diffuser.npz")
    diffuser = SE3Diffuser(synthetic.default_conf(3, cache_dir="/tmp/dfold_igso3_cache/").diffuser)
    r0 = torch.tensor(g["rigids_0"]).to(dev)
    for i, t in enumerate((0.05, 0.5, 0.9)):
        rt = torch.tensor(g[f"fm{i}_rigids_t"]).float().to(dev)
        tt = torch.tensor([t], dtype=torch.float32, device=dev)
        rs = diffuser.calc_rot_score_t7(rt[None, ..., :4], r0[None, ..., :4], tt)[0]
        ref = g[f"fm{i}_calc_rot_score"]
        assert rs.dtype == torch.float64
        assert max_abs(rs, ref) < 2e-4 * max(1.0, float(np.abs(ref).max())), (i, max_abs(rs, ref))
        ts = diffuser.calc_trans_score(rt[..., 4:], r0[..., 4:], tt[:, None, None], use_torch=True)
        reft = g[f"fm{i}_calc_trans_score"]
        assert max_abs(ts, reft) < 1e-4 * max(1.0, float(np.abs(reft).max())), i


def test_gradient_reducer_path_on_one_gpu():
    """The multi-GPU gradient path exercised on one GPU (RCCL world of 1, collectives forced): gradients live in one flat
    buffer, every bucket is launched DURING backward (conv layers from the tower's per-layer finalisation, the rest from
    autograd hooks) in the order the gradients complete, and the step equals the plain single-rank step."""
    import os
    import torch.distributed as dist
    from dynamicpdb_amd import experiment
    dev = torch.device(DEV)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29700 + os.getpid() % 2000))
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        F, N, B = 6, 16, 2
        res = {}
        for forced in (False, True):
            model, diffuser = _build(F, 3, dev)
            batch = _batch(diffuser, B, F, N, dev, seed=70)
            # lr = 0: the parameters never move, so every step of both runs sees the same weights and the hook-driven steps can
            # be compared tightly (with a real step Adam turns rounding-level noise in near-dead gradients into +-lr moves and
            # two separately trained models drift apart by several per cent within three steps)
            tr = experiment.Trainer(model, lr=0.0, last_frame_only=True, force_reduce=forced, bucket_bytes=8 << 20)
            launched_before_finish = None
            if forced:
                fin = tr.reducer.finish

                def spy():
                    nonlocal launched_before_finish
                    if tr.reducer.flat is not None:
                        launched_before_finish = list(tr.reducer._launched)
                    fin()
                tr.reducer.finish = spy
            losses = [float(tr.update_fn(batch)[0])]
            first = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
            losses += [float(tr.update_fn(batch)[0]) for _ in range(2)]
            res[forced] = (losses, {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None},
                           first)
            if forced:
                red = tr.reducer
                assert red.flat is not None and len(red.buckets) >= 8
                assert launched_before_finish is not None and all(launched_before_finish), launched_before_finish
                base = red.flat.untyped_storage().data_ptr()
                assert all(p.grad is None or p.grad.untyped_storage().data_ptr() == base for p in model.parameters())
                # the shared tower's layers complete top-down (layer 7 first), long before the embedders / expand layers
                P = dict(model.named_parameters())
                order = [red._bucket_of[id(P[f"score_model.trunk.conv_0.conv{i}.{j}.weight"])] for i in (4, 3, 2, 1) for j in (2, 0)]
                # every 82 MB conv weight is a bucket of its own, cut in the order the tower finalised them in the
                # discovery step; all of them were reduced during backward (launched_before_finish above)
                assert len(set(order)) == 8, order
                print("conv-layer buckets (layer 7..0):", order, "expand_edge:", red._bucket_of[id(P["expand_edge.weight"])])
                dead = [n for n, p in model.named_parameters() if p.grad is None]
                assert dead and all(n.startswith("embedding_layer.") or "linear_rbf" in n or n.endswith("linear_b.bias") for n in dead)
        for a, b in zip(res[False][0], res[True][0]):
            assert abs(a - b) < 2e-3 * abs(a), (res[False][0], res[True][0])
        assert set(res[False][1]) == set(res[True][1])
        # step 1 (same parameters, same activations: only the order of the fp32 atomics of the weight-gradient reductions
        # differs between two runs): the reduced flat-buffer gradients equal the plain ones tightly
        gmax = max(float(g.norm()) for g in res[False][2].values())
        for n in res[False][2]:
            a, b = res[True][2][n].double(), res[False][2][n].double()
            assert float((a - b).norm()) < 1e-3 * max(float(b.norm()), 1e-3 * gmax), n
        # step 3 (buckets launched from the hooks during backward): same gradients again
        for n in res[False][1]:
            a, b = res[True][1][n].double(), res[False][1][n].double()
            assert float((a - b).norm()) < 1e-3 * max(float(b.norm()), 1e-3 * gmax), n
    finally:
        dist.destroy_process_group()


def test_checkpoint_resume_continues_the_run_on_device(tmp_path):
    """checkpoint.save after 2 update_fn steps (FusedAdam, conv-tower bf16 caches), checkpoint.resume into a fresh model:
    the next two losses and all parameters equal the uninterrupted run (the fused Adam state is torch.optim.Adam's, so the
    same file also loads into torch.optim.Adam)."""
    from dynamicpdb_amd import checkpoint, experiment
    dev = torch.device(DEV)
    F, N, B = 4, 16, 1
    model, diffuser = _build(F, 3, dev)
    batch = _batch(diffuser, B, F, N, dev, seed=90)
    tr = experiment.Trainer(model, lr=1e-3, last_frame_only=True)
    for _ in range(2):
        tr.update_fn(batch)
    path = str(tmp_path / "step_2.pth")
    checkpoint.save(tr, path, conf=None, epoch=0, step=2)
    tail = [float(tr.update_fn(batch)[0]) for _ in range(2)]
    model2, _ = _build(F, 4, dev)                       # other weights: everything must come from the file
    tr2 = experiment.Trainer(model2, lr=1e-3, last_frame_only=True)
    assert checkpoint.resume(tr2, path)[:2] == (0, 2)
    tail2 = [float(tr2.update_fn(batch)[0]) for _ in range(2)]
    for a, b in zip(tail, tail2):
        assert abs(a - b) <= 1e-5 * abs(a), (tail, tail2)       # atomics in the weight-gradient reductions: not bitwise
    for (n, p), q in zip(model.named_parameters(), model2.parameters()):
        assert rel_l2(q, p) < 1e-4, n
    opt = torch.optim.Adam([torch.nn.Parameter(p.detach().clone()) for p in tr2.params], lr=1e-3, amsgrad=True)
    opt.load_state_dict(checkpoint.read_checkpoint(path)["optimizer"])      # interchangeable state layout


def test_angle_resnet_fused_node_row_block_skipping_and_unfused_chain(monkeypatch):
    """Round 6: the angle head as one node (functional.AngleResnetFn: ReLUs, residual adds and ReLU backward in GEMM epilogues) with
    the row-block skipping of its backward: (a) a gradient that lives on a few 256-row blocks only (what the reference's
    last-frame loss sends) -- with the flags (default) against DFOLD_ANGLE_RZ=0: identical input gradients (skipped tiles are
    zeros either way), weight gradients equal up to the order of the fp32 atomics; (b) a dense gradient: nothing skipped, same
    statement; (c) against the chain of single-layer nodes of rounds 1-5 (DFOLD_ANGLE_FUSED=0): the same function up to bf16
    rounding of the residual stream (one rounding fewer per residual add here)."""
    from dynamicpdb_amd.model import functional as F_
    from dynamicpdb_amd.model import ipa_pytorch_dynamic as ipa_mod
    dev = torch.device(DEV)
    torch.manual_seed(5)
    m = ipa_mod.AngleResnet(1280, 1280, 2, 7, 1e-12).to(dev)
    with torch.no_grad():
        for l in m.layers:
            l.linear_2.weight.normal_(0, 0.02)            # (`final` initialisation: zeros)
        for p in m.parameters():
            if p.dim() == 1:
                p.normal_(0, 0.05)
    B, F, N = 2, 4, 256
    s = torch.randn(B, F, N, 1280, device=dev).to(torch.bfloat16)
    s0 = torch.randn(B, F, N, 1280, device=dev).to(torch.bfloat16)

    def run(gu, fused=True, rz=True):
        monkeypatch.setattr(ipa_mod, "_ANGLE_FUSED", fused)
        monkeypatch.setattr(F_, "_ANGLE_RZ", rz)
        m.zero_grad(set_to_none=True)
        a, b = s.clone().requires_grad_(True), s0.clone().requires_grad_(True)
        u, ang = m(a, b)
        (u * gu).sum().backward()
        return u.detach(), a.grad, b.grad, {k: p.grad.clone() for k, p in m.named_parameters()}

    sparse = torch.zeros(B, F, N, 7, 2, device=dev)
    sparse[:, -1] = torch.randn(B, N, 7, 2, device=dev)            # the last frame of each window: 2 of 8 row blocks
    sparse[0, 1, 17, 3, 0] = 0.3                                    # + a single element somewhere else
    dense = torch.randn(B, F, N, 7, 2, device=dev)
    for gu in (sparse, dense):
        on, off = run(gu, rz=True), run(gu, rz=False)
        assert torch.equal(on[0], off[0]) and torch.equal(on[1], off[1]) and torch.equal(on[2], off[2])
        assert float(on[1].float().abs().max()) > 0
        for k in on[3]:
            assert rel_l2(on[3][k], off[3][k]) < 1e-5, k
    # rows of untouched blocks get exactly zero input gradient
    assert float(on[1].float().abs().max()) > 0
    sp = run(sparse)
    assert float(sp[1][:, 2].float().abs().max()) == 0 and float(sp[1][0, 1].float().abs().max()) > 0
    chain = run(sparse, fused=False)
    assert rel_l2(sp[0], chain[0]) < 1e-2
    # (the two paths round the residual stream at different points, so a pre-activation within bf16 rounding of zero may take the
    #  other ReLU branch: measured 5.4e-2 on the input gradients, the class of the unaligned comparisons elsewhere; the tight,
    #  mask-aligned statement about this node is test_angle_resnet_module_vs_oracle_fwd_bwd below)
    assert rel_l2(sp[1], chain[1]) < 1e-1 and rel_l2(sp[2], chain[2]) < 1e-1
    for k in sp[3]:
        assert rel_l2(sp[3][k], chain[3][k]) < 1e-1, (k, rel_l2(sp[3][k], chain[3][k]))


def test_angle_resnet_module_vs_oracle_fwd_bwd():
    """AngleResnet (openfold/model/structure_module.py:75-158) on its own: the drop-in module on the device
    against the oracle restatement -- unnormalised and normalised angles against the plain fp32 oracle; input and
    parameter gradients against the oracle with bf16 operand rounding emulated and the engine's own seven ReLU masks fed
    back (a pre-activation within bf16 rounding of zero then takes the same branch on both sides): SURVEY 8c's bf16 class."""
    from oracle import dfold_oracle as O
    from dynamicpdb_amd import ops
    from dynamicpdb_amd.model.ipa_pytorch_dynamic import AngleResnet
    dev = torch.device(DEV)
    rng = np.random.default_rng(31)
    m = AngleResnet(256, 128, 2, 7, 1e-12)
    sd = m.state_dict()
    for k, v in sd.items():
        sd[k] = torch.tensor((rng.standard_normal(tuple(v.shape)) / np.sqrt(v.shape[-1]) if v.dim() == 2
                              else 0.1 * rng.standard_normal(tuple(v.shape))).astype(np.float32))
    m.load_state_dict(sd)
    s = torch.tensor(rng.standard_normal((2, 5, 33, 256), dtype=np.float32))
    s0 = torch.tensor(rng.standard_normal((2, 5, 33, 256), dtype=np.float32))
    gu = torch.tensor(rng.standard_normal((2, 5, 33, 7, 2), dtype=np.float32))
    ga = torch.tensor(rng.standard_normal((2, 5, 33, 7, 2), dtype=np.float32))
    m.to(dev)
    sd_, s0d = s.to(dev).requires_grad_(True), s0.to(dev).requires_grad_(True)
    with record_relu_masks() as masks:
        u, a = m(sd_, s0d)
    assert len(masks) == 7
    ((u * gu.to(dev)).sum() + (a * ga.to(dev)).sum()).backward()
    P = {"ar." + k: v.clone() for k, v in sd.items()}
    with torch.no_grad():
        ur, ar = O.angle_resnet(P, "ar", s, s0)
    assert rel_l2(u, ur) < 1e-2 and rel_l2(a, ar) < 2e-2, (rel_l2(u, ur), rel_l2(a, ar))
    Pq = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    sr, s0r = s.clone().requires_grad_(True), s0.clone().requires_grad_(True)
    O.EMULATE_BF16_OPERANDS, O.RELU_MASK_FEED = True, list(masks)
    try:
        uq, aq = O.angle_resnet(Pq, "ar", sr, s0r)
        assert not O.RELU_MASK_FEED
        ((uq * gu).sum() + (aq * ga).sum()).backward()
    finally:
        O.EMULATE_BF16_OPERANDS, O.RELU_MASK_FEED = False, None
    # the input gradients cross seven bf16 storage points of the engine (dL/d(out), then dx of six dense layers); with the
    # masks aligned what is left is their rounding: 3.4e-2 .. 4.1e-2 measured (6.0e-2 against the plain fp32 oracle without the
    # feed); in the full network the same layers sit behind 1/sqrt(k)-scaled activations and meet the 3e-2 class
    assert rel_l2(sd_.grad, sr.grad) < 5e-2 and rel_l2(s0d.grad, s0r.grad) < 5e-2, (rel_l2(sd_.grad, sr.grad), rel_l2(s0d.grad, s0r.grad))
    for k, p in m.named_parameters():
        assert rel_l2(p.grad, Pq["ar." + k].grad) < 5e-2, (k, rel_l2(p.grad, Pq["ar." + k].grad))


def test_loss_fn_on_device_vs_oracle_values_and_gradients():
    """Experiment.loss_fn (train_DFOLD_dynamics.py:1182-1400) on its own: given model outputs, the batched
    device loss and its gradients w.r.t. the three outputs it reads, against the oracle's per-window loss."""
    from oracle import dfold_oracle as O
    from dynamicpdb_amd import experiment, synthetic
    from dynamicpdb_amd.data.se3_diffuser import SE3Diffuser
    dev = torch.device(DEV)
    F, N, B = 5, 21, 3
    conf = synthetic.default_conf(F, cache_dir="/tmp/dfold_igso3_cache/")
    diffuser = SE3Diffuser(conf.diffuser)
    ws = [synthetic.synthetic_window(300 + i, F, N, t=0.2 + 0.3 * i, diffuser=diffuser) for i in range(B)]
    rng = np.random.default_rng(5)
    outs = [{"angles": torch.tensor(rng.standard_normal((F, N, 7, 2), dtype=np.float32)),
             "rigids": torch.tensor(rng.standard_normal((F, N, 7), dtype=np.float32)),
             "rot_score": torch.tensor(rng.standard_normal((F, N, 3), dtype=np.float32))} for _ in range(B)]
    ref_l, ref_g = [], []
    for w, o in zip(ws, outs):
        o = {k: v.clone().requires_grad_(True) for k, v in o.items()}
        l, _ = O.loss_fn(o, w)
        l.backward()
        ref_l.append(float(l))
        ref_g.append({k: v.grad for k, v in o.items()})
    batch = {k: torch.stack([w[k] for w in ws]).to(dev) for k in ws[0] if k != "t"}
    batch["t"] = torch.cat([w["t"] for w in ws]).to(dev)
    ob = {k: torch.stack([o[k] for o in outs]).to(dev).requires_grad_(True) for k in outs[0]}
    loss, aux = experiment.loss_fn(ob, batch)
    assert abs(float(loss) - np.mean(ref_l)) < 1e-5 * abs(np.mean(ref_l)), (float(loss), ref_l)
    loss.backward()
    for k in ob:
        want = torch.stack([g[k] for g in ref_g]) / B          # the batched loss is the mean over windows
        assert rel_l2(ob[k].grad, want) < 1e-5, k
    assert set(aux) >= {"rot_loss", "trans_loss", "torsion_loss"}


@pytest.mark.parametrize("case", ["plain", "gate_closed", "t_below_threshold", "holes_and_dead_frames", "equal_alt"])
def test_loss_one_launch_equals_the_aten_graph(case, monkeypatch):
    """Round 6: loss_fn on device tensors is one HIP launch (csrc/loss.hip through experiment.LossLastFrameFn: values and the
    gradients w.r.t. angles / x0 translations / rotation scores of the last frame).  Against the aten graph of the same
    formulas (DFOLD_LOSS_FUSED=0, the path host tensors take) on the device: loss, the three aux terms and the three gradients
    -- with the trans < 100 gate open and closed, t below the rotation threshold, masked residues / frames without any residue
    (the F / live-frames normalisation), float64 rotation scores as the score head produces them, and alt torsions equal to
    the true ones (the tie of torch.minimum)."""
    from dynamicpdb_amd import experiment, synthetic
    from dynamicpdb_amd.data.se3_diffuser import SE3Diffuser
    dev = torch.device(DEV)
    F, N, B = 6, 37, 4
    conf = synthetic.default_conf(F, cache_dir="/tmp/dfold_igso3_cache/")
    diffuser = SE3Diffuser(conf.diffuser)
    ws = [synthetic.synthetic_window(400 + i, F, N, t=0.15 + 0.2 * i, diffuser=diffuser) for i in range(B)]
    batch = {k: torch.stack([w[k] for w in ws]).to(dev) for k in ws[0] if k != "t"}
    batch["t"] = torch.cat([w["t"] for w in ws]).to(dev)
    rng = np.random.default_rng(9)
    kw = {}
    rig = rng.standard_normal((B, F, N, 7), dtype=np.float32)
    if case == "gate_closed":
        rig[1, :, :, 4:] += 40.0            # window 1: translation loss far above 100 -> its rot / trans terms are gated off
    if case == "t_below_threshold":
        kw["rot_t_threshold"] = 0.4         # windows 0 and 1 lose their rotation term
    if case == "holes_and_dead_frames":
        rm = batch["res_mask"].clone()
        rm[:, :, ::5] = 0
        rm[2, 1] = 0                        # a frame without any residue: F / (live frames) != 1
        rm[2, 3] = 0
        batch["res_mask"] = rm
        fm = batch["fixed_mask"].clone().float()
        fm[:, :, 1::7] = 1
        batch["fixed_mask"] = fm
    if case == "equal_alt":
        batch["alt_torsion_angles_sin_cos"] = batch["torsion_angles_sin_cos"].clone()
    base = {"angles": torch.tensor(rng.standard_normal((B, F, N, 7, 2), dtype=np.float32)), "rigids": torch.tensor(rig),
            "rot_score": torch.tensor(rng.standard_normal((B, F, N, 3)))}          # float64, as igso3_score returns it
    res = []
    for fused in (True, False):
        monkeypatch.setattr(experiment, "_LOSS_FUSED", fused)
        ob = {k: v.clone().to(dev).requires_grad_(True) for k, v in base.items()}
        loss, aux = experiment.loss_fn(ob, batch, **kw)
        loss.backward()
        res.append((float(loss), {k: float(v) for k, v in aux.items()}, {k: ob[k].grad.clone() for k in ob}))
    (l1, a1, g1), (l0, a0, g0) = res
    assert abs(l1 - l0) <= 2e-6 * abs(l0), (l1, l0)
    for k in a0:
        assert abs(a1[k] - a0[k]) <= 2e-6 * max(abs(a0[k]), 1e-3), (k, a1[k], a0[k])
    for k in g0:
        assert g1[k].dtype == g0[k].dtype and g1[k].shape == g0[k].shape
        assert float(g1[k][:, :-1].abs().max()) == 0            # only the last frame is read
        assert rel_l2(g1[k].double(), g0[k].double()) < 5e-6, (k, rel_l2(g1[k].double(), g0[k].double()))
    if case == "gate_closed":
        assert float(g1["rigids"][1].abs().max()) == 0 and float(g1["rot_score"][1].abs().max()) == 0
        assert float(g1["angles"][1].abs().max()) > 0


def test_rot_score_head_four_launches_equals_the_aten_chain(monkeypatch):
    """Round 6: the rotation-score head (SE3Diffuser.calc_rot_score, se3_diffuser.py:119-125: quaternion inverse / product,
    quat_to_rotvec, the IGSO(3) series, sc v / (omega + eps)) on the device is four HIP launches (functional.RotScoreHeadFn, csrc/
    score.hip) instead of ~60 + ~120 aten launches.  Against the aten chain of the same formulas (DFOLD_SCORE_FUSED=0): the float64
    score and its gradient w.r.t. the predicted quaternion -- random relative rotations, un-normalised q_0 (the network's
    quaternions before normalisation), relative rotations with w < 0 (the sign flip), nearly identical frames (the small-angle
    branch of quat_to_rotvec, angle <= 1e-3) and exactly identical ones (norm subgradients)."""
    from dynamicpdb_amd.model import score_heads
    dev = torch.device(DEV)
    W, F, N = 3, 4, 50
    rng = np.random.default_rng(21)
    sigma = np.array([0.4, 1.1, 0.15])
    qt = rng.standard_normal((W, F, N, 4)).astype(np.float32)
    qt /= np.linalg.norm(qt, axis=-1, keepdims=True)
    # q_0 = q_t (x) r^-1 with a relative rotation r of 0.3 ... 2.5 sigma about a random axis: the regime the head works in (far
    # out in the tail, f(omega) ~ exp(-omega^2 / 2 sigma^2) is below the fp32 rounding noise of the series' terms and the
    # reference's score is noise / 1e-4 -- in every implementation)
    ax = rng.standard_normal((W, F, N, 3))
    ax /= np.linalg.norm(ax, axis=-1, keepdims=True)
    th = sigma[:, None, None] * rng.uniform(0.3, 2.5, size=(W, F, N))
    rinv = np.concatenate([np.cos(th / 2)[..., None], -np.sin(th / 2)[..., None] * ax], -1)
    a1, b1, c1, d1 = np.moveaxis(qt.astype(np.float64), -1, 0)
    a2, b2, c2, d2 = np.moveaxis(rinv, -1, 0)
    q0 = np.stack([a1 * a2 - b1 * b2 - c1 * c2 - d1 * d2, a1 * b2 + b1 * a2 + c1 * d2 - d1 * c2,
                   a1 * c2 - b1 * d2 + c1 * a2 + d1 * b2, a1 * d2 + b1 * c2 - c1 * b2 + d1 * a2], -1)
    q0 = (q0 * rng.uniform(0.5, 1.5, size=(W, F, N, 1))).astype(np.float32)       # un-normalised, as the network's quaternions
    q0[0, 0] = qt[0, 0] * 1.3                                               # identical rotations (zero rotation vector)
    q0[0, 1] = qt[0, 1] + 1e-4 * rng.standard_normal((N, 4)).astype(np.float32)   # small-angle branch
    q0[1, 0] = -q0[1, 0]                                                    # w of q_0^-1 q_t below zero: the sign flip
    gy = torch.tensor(rng.standard_normal((W, F, N, 3))).to(dev)
    res = []
    for fused in (True, False):
        monkeypatch.setattr(score_heads, "_HEAD_FUSED", fused)
        a = torch.tensor(q0).to(dev).requires_grad_(True)
        sc = score_heads.rot_score(torch.tensor(qt).to(dev), a, sigma)
        assert sc.dtype == torch.float64 and sc.shape == (W, F, N, 3)
        (sc * gy).sum().backward()
        res.append((sc.detach().clone(), a.grad.clone()))
    (s1, g1), (s0, g0) = res
    assert torch.isfinite(s1).all() and torch.isfinite(g1).all()
    # Where the relative rotation is tiny (angle <= 1e-3) the reference's fp32 series is rounding noise (lo dhi - hi dlo cancels
    # to O(omega^3) in fp32: so3_diffuser.py:95-105), in the aten chain as in the kernels -- last-bit differences of omega give
    # different noise.  Values and gradients are compared where the series is conditioned; the small-angle branch of the
    # rotation-vector map and its backward are checked on their own below.
    for name, sl in (("random", (slice(2, 3),)), ("flipped", (1, 0)), ("rest of windows 0 / 1", (slice(0, 2), slice(2, None)))):
        ev, eg = rel_l2(s1[sl], s0[sl]), rel_l2(g1[sl].double(), g0[sl].double())
        print(f"[rot score head] {name}: score rel-L2 vs the aten chain {ev:.2e}, gradient {eg:.2e}")
        assert ev < 1e-6 and eg < 2e-5, (name, ev, eg)
    assert float(s1[0, 0].abs().max()) == 0 and float(s0[0, 0].abs().max()) == 0          # identical frames: zero score
    # the rotation-vector map alone (no series): dfold_rot_head_pre against geometry.quat_to_rotvec, and dfold_rot_head_bwd with
    # sc = 1, dsc = 0 -- i.e. the gradient of v / (|v| + 2e-6) -- against float64 autograd of the same formulas, every region
    from ctypes import c_int64, c_void_p
    from dynamicpdb_amd import _lib
    from dynamicpdb_amd.model import geometry as G
    L = _lib.lib()
    P = W * F * N
    p = lambda t: c_void_p(t.data_ptr())
    tq, t0 = torch.tensor(qt).to(dev).contiguous(), torch.tensor(q0).to(dev).contiguous()
    vec = torch.empty(W, F, N, 3, device=dev)
    om = torch.empty(W, F, N, device=dev)
    _lib.check(L.dfold_rot_head_pre(p(tq), p(t0), p(vec), p(om), c_int64(P), _lib.stream()), "pre")
    vref = G.quat_to_rotvec(G.quat_mul(G.quat_invert(t0), tq))
    assert rel_l2(vec, vref) < 1e-6 and rel_l2(om, torch.linalg.norm(vref, dim=-1) + 1e-6) < 1e-6
    assert float((2 * torch.atan2(torch.linalg.norm(vref[0, 1], dim=-1), vref.new_ones(()))).max()) < 2e-3     # (the small-angle rows are small)
    a64 = t0.double().requires_grad_(True)
    v64 = G.quat_to_rotvec(G.quat_mul(G.quat_invert(a64), tq.double()))
    n64 = torch.linalg.norm(v64, dim=-1, keepdim=True)
    (v64 / (n64 + 2e-6) * gy).sum().backward()
    one, zero = torch.ones(W, F, N, dtype=torch.float64, device=dev), torch.zeros(W, F, N, dtype=torch.float64, device=dev)
    dq = torch.empty(W, F, N, 4, device=dev)
    _lib.check(L.dfold_rot_head_bwd(p(gy.contiguous()), p(tq), p(t0), p(vec), p(om), p(one), p(zero), p(dq), c_int64(P), _lib.stream()), "bwd")
    for name, sl in (("random", (slice(2, 3),)), ("flipped", (1, 0)), ("small angle", (0, 1))):
        e = rel_l2(dq[sl].double(), a64.grad[sl])
        print(f"[rot score head] rotation-vector map, {name}: gradient rel-L2 vs float64 autograd {e:.2e}")
        assert e < (2e-3 if name == "small angle" else 1e-4), (name, e)        # (small angles: v / |v| of fp32 differences of 1e-4)
    assert torch.isfinite(dq).all()


def test_conv_gradients_flow_through_autograd_and_ddp_wrapper():
    """Without a dp.GradReducer the shared conv tower's weight gradients leave ConvTowerFn.backward as ordinary autograd
    outputs: torch.autograd.grad() returns them, post-accumulate hooks fire, and the reference's own wrapper
    -- DistributedDataParallel(find_unused_parameters=True), train_DFOLD_dynamics.py:615 -- reduces them (world of one
    RCCL rank here: the wrapper's reducer must see every conv parameter as used and ready)."""
    import os
    import torch.distributed as dist
    from dynamicpdb_amd import experiment
    dev = torch.device(DEV)
    F, N, B = 5, 16, 1
    model, diffuser = _build(F, 8, dev)
    batch = _batch(diffuser, B, F, N, dev, seed=90)
    conv_names = [n for n, _ in model.named_parameters() if "conv_0" in n]
    P = dict(model.named_parameters())
    # (a) reference gradients: plain backward
    out = model({k: v.clone() for k, v in batch.items()})
    loss, _ = experiment.loss_fn(out, batch)
    fired = []
    hooks = [P[n].register_post_accumulate_grad_hook(lambda p, n=n: fired.append(n)) for n in conv_names]
    loss.backward()
    for h in hooks:
        h.remove()
    assert sorted(fired) == sorted(conv_names)
    ref = {n: P[n].grad.detach().clone() for n in conv_names}
    assert all(float(g.abs().max()) > 0 for g in ref.values())
    # (b) torch.autograd.grad sees them and leaves .grad alone
    model.zero_grad(set_to_none=True)
    out = model({k: v.clone() for k, v in batch.items()})
    loss, _ = experiment.loss_fn(out, batch)
    gs = torch.autograd.grad(loss, [P[n] for n in conv_names])
    assert all(P[n].grad is None for n in conv_names)
    for n, g in zip(conv_names, gs):
        assert rel_l2(g, ref[n]) < 1e-3, n
    # (c) a forward whose graph is dropped, then a kept-alive stale graph: neither blocks the next step's delivery
    _ = model({k: v.clone() for k, v in batch.items()})
    del _
    stale = model({k: v.clone() for k, v in batch.items()})
    out = model({k: v.clone() for k, v in batch.items()})
    experiment.loss_fn(out, batch)[0].backward()
    for n in conv_names:
        assert P[n].grad is not None and rel_l2(P[n].grad, ref[n]) < 1e-3, n
    del stale, out
    # (d) the reference's wrapper
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29800 + os.getpid() % 2000))
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        model.zero_grad(set_to_none=True)
        ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0], find_unused_parameters=True)
        for _ in range(2):          # second step: DDP's bucket rebuild has happened
            ddp.zero_grad(set_to_none=True)
            out = ddp({k: v.clone() for k, v in batch.items()})
            experiment.loss_fn(out, batch)[0].backward()
            for n in conv_names:
                assert P[n].grad is not None and rel_l2(P[n].grad, ref[n]) < 1e-3, n
    finally:
        dist.destroy_process_group()


def _dp_real_model_worker(rank, world, port, q):
    """one of two ranks sharing cuda:0 (gloo backend: RCCL refuses two ranks on one device)"""
    import os
    import sys
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dynamicpdb_amd import experiment
        dev = torch.device(DEV)
        F, N = 3, 16
        model, diffuser = _build(F, 100 + rank, dev)          # every rank starts from DIFFERENT weights (train:419)
        shards = [_batch(diffuser, 1, F, N, dev, seed=200 + 10 * r) for r in range(world)]
        tr = experiment.Trainer(model, lr=0.0, last_frame_only=True, bucket_bytes=64 << 20)
        tr.reducer.timing = True
        ck = float(sum(p.detach().double().sum() for p in model.parameters()))
        launched = []
        fin = tr.reducer.finish

        def spy():
            if tr.reducer.flat is not None:
                launched.append(list(tr.reducer._launched))
            fin()
        tr.reducer.finish = spy
        for _ in range(3):
            loss, _ = tr.update_fn(shards[rank])
        got = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
        # the conv tower's parameters complete ONCE per step (its own layer-by-layer hand-over; no spurious autograd hook
        # event), in the order in which the last backward application finalises the layers: 7, 6, ..., 0
        red = tr.reducer
        names = {id(p): n for n, p in model.named_parameters()}
        conv_expected = sorted({red._expected[k] for k, n in names.items() if ".conv_0." in n and k in red._expected})
        conv_order = [names[id(p)].split("conv_0.")[1] for ps in red._bucket_params for p in ps if ".conv_0." in names[id(p)]
                      and names[id(p)].endswith("weight")]
        tr.begin_step()                                      # resolves the last step's per-bucket launch events
        lead = list(red.bucket_lead_ms)
        # the mean of the per-rank gradients, computed locally on the (now common) weights without any reducer
        tr.reducer.detach()
        tr.reducer.active = False
        want = None
        for r in range(world):
            model.zero_grad(set_to_none=True)
            for m in model.modules():
                if getattr(m, "_tower", None) is not None:
                    m._tower.on_final = None
            out = model(shards[r], last_frame_only=True)
            experiment.loss_fn(out, shards[r])[0].backward()
            g = {n: p.grad.detach().double() / world for n, p in model.named_parameters() if p.grad is not None}
            want = g if want is None else {n: want[n] + g[n] for n in g}
        gmax = max(float(v.norm()) for v in want.values())
        worst = max(float((got[n].double() - want[n]).norm()) / max(float(want[n].norm()), 1e-3 * gmax) for n in want)
        q.put((rank, dict(checksum=ck, worst=worst, same_keys=set(got) == set(want), launched=launched,
                          conv_expected=conv_expected, conv_order=conv_order, lead=lead,
                          n_buckets=len(tr.reducer.buckets), wait_ms=list(tr.reducer.wait_ms),
                          bytes_broadcast=tr.bytes_broadcast, staged=bool(tr.reducer._stage_host))))
    except Exception as e:   # surface the failure in the parent instead of a queue timeout
        import traceback
        q.put((rank, dict(error=traceback.format_exc())))
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_data_parallel_real_model_two_ranks_on_one_gpu():
    """The REAL model under dp.GradReducer with a peer: two processes share cuda:0 at F3 x N16, gloo
    backend, each seeded differently.  The Trainer's start-up broadcast makes the parameters equal (same checksum); after
    the discovery step and two hook-driven steps each rank's gradients equal the mean of the per-rank gradients; every
    bucket -- the eight conv-layer buckets handed over by ConvTower.finalize_layer -> mark_ready included -- was launched
    before finish() (i.e. during backward, overlapping the peer's collective)."""
    import os
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    from util import free_port
    port = free_port()
    procs = [ctx.Process(target=_dp_real_model_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
    for r in (0, 1):
        assert "error" not in res[r], res[r].get("error")
    assert abs(res[0]["checksum"] - res[1]["checksum"]) < 1e-9 * max(1.0, abs(res[0]["checksum"])), "parameters not broadcast"
    for r in (0, 1):
        info = res[r]
        assert info["bytes_broadcast"] > 700e6 and info["same_keys"], info
        assert info["worst"] < 2e-3, info["worst"]
        assert info["n_buckets"] >= 8 and len(info["launched"]) == 2 and all(all(l) for l in info["launched"]), info["launched"]
        assert len(info["wait_ms"]) >= 1
        # one accumulation event per conv parameter and step; buckets cut in the hand-over order of the last backward
        # application (top layer first), each launched strictly earlier than the next (layer-by-layer overlap with backward)
        assert info["conv_expected"] == [1], info["conv_expected"]
        assert info["conv_order"] == ["conv4.2.weight", "conv4.0.weight", "conv3.2.weight", "conv3.0.weight", "conv2.2.weight",
                                      "conv2.0.weight", "conv1.2.weight", "conv1.0.weight"], info["conv_order"]
    print("2-rank real model on one GPU: worst gradient deviation from the mean of per-rank gradients",
          max(res[0]["worst"], res[1]["worst"]), "host-staged gloo:", res[0]["staged"], "wait_ms", res[0]["wait_ms"],
          "bucket launch lead before the end of backward (ms):", res[0]["lead"])
