"""Mint golden vectors from the REFERENCE's own code (run in the build container only;
needs /root/reference).  Usage:  python tests/golden/make_golden.py

Writes small .npz fixtures next to this file.  Weights are NOT stored: they are
regenerated bit-identically from dynamicpdb_amd.synthetic.seeded_state_dict(seed)
(numpy Generator, same image on the GPU box); loading them into the reference model
with strict=True is itself the state_dict-compatibility check.
"""
import os
import random
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle", "ref_harness"))
import ref_import  # noqa: E402

ref_import.install()
os.chdir(os.environ.get("DFOLD_GOLDEN_WORKDIR", "/tmp/work"))

from dynamicpdb_amd import synthetic  # noqa: E402  (seeded inputs / weights only: plain numpy, no device code)
# the REFERENCE's Rigid: nothing of the product sits between the seeded inputs and the reference's diffuser
from openfold.utils.rigid_utils import Rigid  # noqa: E402

torch.set_num_threads(8)


def np_(x):
    return x.detach().cpu().numpy() if torch.is_tensor(x) else np.asarray(x)


def subsample(g, stride=9973):
    flat = g.reshape(-1)
    return flat[::stride].clone()


DIFFUSER_KEYS = ("rigids_t", "rot_score", "trans_score", "rot_score_scaling", "trans_score_scaling")


def golden_network(F=3, N=16, seed_w=0, seed_x=1, t=0.5, captures=True, grad_stride=9973, compact=False, holes=0.0,
                   tag="", no_torsion_grads=False):
    """compact=True (BASELINE-sized captures): only the diffuser-dependent inputs are stored; every other input is
    regenerated bit-identically on the test side from dynamicpdb_amd.synthetic.synthetic_window(seed_x, F, N, t) and
    pinned by a float64 checksum."""
    from src.data.se3_diffuser import SE3Diffuser
    import train_DFOLD_dynamics as T
    conf = ref_import.make_conf(F, cache_dir=".cache/")
    exp = T.Experiment(conf=conf)
    model = exp.model
    sd = synthetic.seeded_state_dict(seed_w)
    model.load_state_dict(sd, strict=True)
    model.eval()
    diffuser = exp.diffuser
    win = synthetic.synthetic_window(seed_x, F, N, t=t, diffuser=diffuser, rigid_cls=Rigid, holes=holes)
    batch = {k: v.clone() for k, v in win.items()}
    # per-op captures through forward hooks on the reference modules
    cap = {}
    sm = model.score_model
    hooks = []
    def grab(name, first_only=False):
        def h(mod, inp, out):
            if first_only and name in cap:
                return
            cap[name] = out.detach().clone()
        return h
    for b in range(4):
        hooks.append(sm.trunk[f"ipa_{b}"].register_forward_hook(grab(f"ipa_{b}")))
        hooks.append(sm.trunk[f"ln_{b}"].register_forward_hook(grab(f"ipa_ln_{b}")))
    conv_calls = []
    hooks.append(sm.trunk["conv_0"].register_forward_hook(lambda m, i, o: conv_calls.append((i[0].detach().clone(), o.detach().clone()))))
    hooks.append(sm.force_embeder.register_forward_hook(grab("force_embed")))
    hooks.append(sm.rigid_embeder.register_forward_hook(grab("rigid_embed_0", first_only=True)))
    random.seed(0)   # loss_fn flips a coin for the (output-irrelevant) self-conditioning pass
    model.zero_grad()
    loss, aux = exp.loss_fn(batch)
    loss.backward()
    for h in hooks:
        h.remove()
    with torch.no_grad():
        out = model({k: v.clone() for k, v in win.items()})
    if compact:
        fix = {f"in_{k}": np_(win[k]) for k in DIFFUSER_KEYS}
        # numpy's sum: single-threaded pairwise, the same on a host with any number of cores (torch's is not)
        fix["in_checksum"] = np.array([float(np.asarray(win[k].numpy(), dtype=np.float64).sum()) for k in sorted(win)
                                       if k not in DIFFUSER_KEYS])
    else:
        fix = {f"in_{k}": np_(v) for k, v in win.items()}
    for k in ("angles", "unorm_angles", "rot_score", "trans_score", "rigids", "atom37", "atom14", "rigid_update"):
        fix[f"out_{k}"] = np_(out[k])
    if captures:
        for k, v in cap.items():
            fix[f"cap_{k}"] = np_(v)
        for i, (ci, co) in enumerate(conv_calls[:4]):
            fix[f"cap_conv_in_{i}"] = np_(ci)
            fix[f"cap_conv_out_{i}"] = np_(co)
    fix["loss"] = np_(loss)
    for k, v in aux.items():
        if k in ("rot_loss", "trans_loss", "torsion_loss", "total_loss"):
            fix[f"aux_{k}"] = np_(v)
    for name, p in model.named_parameters():
        if p.grad is None:
            fix[f"gradnone_{name}"] = np.zeros(1)
            continue
        g = p.grad
        fix[f"gnorm_{name}"] = np_(g.double().norm())
        fix[f"gsub_{name}"] = np_(subsample(g, grad_stride) if g.numel() > 70000 else g)
    if no_torsion_grads:
        # second reference run with experiment.torsion_loss_weight = 0 (the reference's own knob, train:1220): the torsion
        # term normalises the raw 2-vectors (openfold/utils/loss.py:58-59), its gradient ~ 1/|raw| is dominated by the few
        # torsions with a short raw vector, and at 1792 torsions (N_res 256) that alone moves whole gradient tensors by
        # tens of per cent under ANY change of rounding; the frame terms (translation x0, rotation score) are well
        # conditioned, so their gradients can be compared raw
        exp._exp_conf.torsion_loss_weight = 0.0
        random.seed(0)
        model.zero_grad()
        loss0, aux0 = exp.loss_fn({k: v.clone() for k, v in win.items()})
        loss0.backward()
        fix["loss_notorsion"] = np_(loss0)
        for name, p in model.named_parameters():
            if p.grad is None:
                continue
            g = p.grad
            fix[f"g0norm_{name}"] = np_(g.double().norm())
            fix[f"g0sub_{name}"] = np_(subsample(g, grad_stride) if g.numel() > 70000 else g)
        exp._exp_conf.torsion_loss_weight = 1.0
        print("   torsion-free loss", float(loss0))
    fix["meta"] = np.array([F, N, seed_w, seed_x, grad_stride], np.int64)
    fix["t"] = np.array([t])
    fix["holes"] = np.array([holes])
    np.savez_compressed(os.path.join(HERE, f"network_F{F}_N{N}{tag}.npz"), **fix)
    print("network golden: loss", float(loss), {k: float(v) for k, v in aux.items() if "batch" not in k})
    return exp


def golden_sampler(F=3, N=16, seed_w=4, seed_x=8, num_t=3, noise_scale=0.5, seed_z=77, compact=False):
    """Experiment.inference_fn of the reference (train_DFOLD_dynamics.py:1425-1547), num_t model forwards + the
    self-conditioning pass + (num_t - 1) host reverse steps, with the numpy draws of the reverse steps recorded in the
    reference's order (so3 then r3 per step, se3_diffuser.py:184-204)."""
    import train_DFOLD_dynamics as T
    conf = ref_import.make_conf(F, cache_dir=".cache/")
    conf.data.num_t = num_t
    exp = T.Experiment(conf=conf)
    exp.model.load_state_dict(synthetic.seeded_state_dict(seed_w), strict=True)
    win = synthetic.synthetic_window(seed_x, F, N, t=1.0, diffuser=exp.diffuser, rigid_cls=Rigid)
    np.random.seed(seed_z - 1)
    prior = exp.diffuser.sample_ref(n_samples=F * N, as_tensor_7=True)["rigids_t"].reshape(F, N, 7).to(torch.float32)
    init = {k: v.clone() for k, v in win.items()}
    init["rigids_t"] = prior
    np.random.seed(seed_z)
    draws = [(np.random.normal(size=(F, N, 3)), np.random.normal(size=(F, N, 3))) for _ in range(num_t - 1)]
    np.random.seed(seed_z)
    ret = exp.inference_fn({k: v.clone() for k, v in init.items()}, num_t=num_t, min_t=0.01, center=True, aux_traj=True,
                           self_condition=True, noise_scale=noise_scale)
    if compact:
        # BASELINE-sized run (config 1: 16 frames x N_res 96, num_t = 10): only the prior sample is stored, every other input is
        # regenerated on the test side from synthetic_window(seed_x, F, N, t = 1) and pinned by a float64 checksum; the draws are
        # regenerated from numpy's stream (seed_z, the reference's order); atom trajectories at the last two and the first
        # reverse step only (frames / translations / torsions: all steps)
        fix = {"in_rigids_t": np_(init["rigids_t"])}
        fix["in_checksum"] = np.array([float(np.asarray(init[k].numpy(), dtype=np.float64).sum()) for k in sorted(init)
                                       if k not in DIFFUSER_KEYS])
        fix["seed_z"] = np.array([seed_z], np.int64)
        keep = np.array([0, 1, num_t - 1])
        fix["traj_steps"] = keep
        for k in ("rigid_traj", "trans_traj", "psi_pred"):
            fix[f"out_{k}"] = np_(ret[k])
        for k in ("prot_traj", "rigid_0_traj"):
            fix[f"out_{k}"] = np_(ret[k])[keep].astype(np.float32)
    else:
        fix = {f"in_{k}": np_(v) for k, v in init.items()}
        for i, (zr, zt) in enumerate(draws):
            fix[f"z_rot_{i}"], fix[f"z_trans_{i}"] = zr, zt
        for k in ("prot_traj", "rigid_traj", "trans_traj", "rigid_0_traj", "psi_pred"):
            fix[f"out_{k}"] = np_(ret[k])
    fix["meta"] = np.array([F, N, seed_w, seed_x, num_t], np.int64)
    fix["noise_scale"] = np.array([noise_scale])
    np.savez_compressed(os.path.join(HERE, f"sampler_F{F}_N{N}.npz"), **fix)
    print("sampler golden written", {k: v.shape for k, v in fix.items() if k.startswith("out_")})


def golden_bf16_sensitivity(big=False):
    """How far does the REFERENCE move from itself under the engine's declared storage class?  For the small goldens' windows
    the reference's own loss gradients in fp32 against the same reference code with (a) every parameter rounded to bf16 and
    (b) the output of every nn.Linear / nn.Conv2d rounded to bf16 (forward AND, through the cast's own backward, the gradient
    that flows back through that point) -- nothing else changes: same modules, same aten kernels, same ReLU code, branches free
    to flip.  Per parameter tensor: relative L2 distance and relative norm difference, full loss and torsion_loss_weight = 0.
    This is the yardstick for the engine's raw gradient tolerances (tests/test_parity_baseline_gpu.py): an implementation
    that stores bf16 between its kernels cannot be closer to the fp32 reference than the reference is to its own bf16-storage
    run."""
    import train_DFOLD_dynamics as T
    fix = {}
    small = (("F3_N16", 3, 16, 0, 1, 0.5, 0.0), ("F6_N40_holes", 6, 40, 27, 28, 0.5, 0.1), ("F16_N96", 16, 96, 11, 12, 0.4, 0.0),
             ("F2_N256", 2, 256, 13, 14, 0.6, 0.0))
    large = (("F32_N128", 32, 128, 23, 24, 0.35, 0.0), ("F32_N256", 32, 256, 21, 22, 0.45, 0.0), ("F8_N512", 8, 512, 25, 26, 0.55, 0.0))
    for tag, F, N, seed_w, seed_x, t, holes in (large if big else small):
        conf = ref_import.make_conf(F, cache_dir=".cache/")
        exp = T.Experiment(conf=conf)
        model = exp.model
        win = synthetic.synthetic_window(seed_x, F, N, t=t, diffuser=exp.diffuser, rigid_cls=Rigid, holes=holes)

        def grads(torsion_w):
            exp._exp_conf.torsion_loss_weight = torsion_w
            random.seed(0)
            model.zero_grad()
            loss, _ = exp.loss_fn({k: v.clone() for k, v in win.items()})
            loss.backward()
            return float(loss), {n: p.grad.detach().double().clone() for n, p in model.named_parameters() if p.grad is not None}

        model.load_state_dict(synthetic.seeded_state_dict(seed_w), strict=True)
        model.eval()
        ref = {w: grads(w) for w in (1.0, 0.0)}
        with torch.no_grad():
            for p in model.parameters():
                p.copy_(p.to(torch.bfloat16).float())
        hooks = [m.register_forward_hook(lambda mod, inp, out: out.to(torch.bfloat16).to(out.dtype))
                 for m in model.modules() if isinstance(m, (torch.nn.Linear, torch.nn.Conv2d))]
        low = {w: grads(w) for w in (1.0, 0.0)}
        for h in hooks:
            h.remove()
        exp._exp_conf.torsion_loss_weight = 1.0
        for w, wtag in ((1.0, "full"), (0.0, "notorsion")):
            names = sorted(n for n in ref[w][1] if float(ref[w][1][n].norm()) > 1e-6)
            rel = np.array([float((low[w][1][n] - ref[w][1][n]).norm() / ref[w][1][n].norm()) for n in names])
            nrm = np.array([abs(float(low[w][1][n].norm() / ref[w][1][n].norm()) - 1.0) for n in names])
            fix[f"{tag}/{wtag}/rel"], fix[f"{tag}/{wtag}/nrm"] = rel, nrm
            fix[f"{tag}/{wtag}/names"] = np.array(names)
            fix[f"{tag}/{wtag}/loss"] = np.array([ref[w][0], low[w][0]])
            print(f"bf16-storage reference vs fp32 reference, {tag}, {wtag}: loss {ref[w][0]:.5f} / {low[w][0]:.5f}; per-tensor rel-L2 "
                  f"median {np.median(rel):.4f} max {rel.max():.4f} ({names[int(rel.argmax())]}); norm error median "
                  f"{np.median(nrm):.4f} max {nrm.max():.4f}")
    np.savez_compressed(os.path.join(HERE, "bf16_sensitivity_big.npz" if big else "bf16_sensitivity.npz"), **fix)


def golden_triangle(N=24, seed=3):
    from openfold.model.triangular_multiplicative_update import (TriangleMultiplicationIncoming,
                                                                  TriangleMultiplicationOutgoing)
    from openfold.model.triangular_attention import TriangleAttentionEndingNode, TriangleAttentionStartingNode
    rng = np.random.default_rng(seed)
    fix = {}
    z = torch.tensor(rng.standard_normal((N, N, 128), dtype=np.float32))
    mask = torch.tensor((rng.uniform(size=(N, N)) > 0.1).astype(np.float32))
    fix["z"], fix["mask"] = np_(z), np_(mask)
    mods = dict(tri_mul_out=TriangleMultiplicationOutgoing(128, 128), tri_mul_in=TriangleMultiplicationIncoming(128, 128),
                tri_att_start=TriangleAttentionStartingNode(128, 32, 4), tri_att_end=TriangleAttentionEndingNode(128, 32, 4))
    for name, m in mods.items():
        sd = m.state_dict()
        for i, (k, v) in enumerate(sd.items()):
            if v.dim() >= 2:
                w = rng.standard_normal(tuple(v.shape), dtype=np.float32) / np.sqrt(v.shape[-1])
            elif k.endswith("weight"):
                w = 1.0 + 0.1 * rng.standard_normal(tuple(v.shape), dtype=np.float32)
            else:
                w = 0.1 * rng.standard_normal(tuple(v.shape), dtype=np.float32)
            sd[k] = torch.tensor(w.astype(np.float32))
            fix[f"{name}.P.{k}"] = w.astype(np.float32)
        m.load_state_dict(sd)
        zz = z.clone().requires_grad_(True)
        y = m(zz, mask=mask)
        gy = torch.tensor(rng.standard_normal(tuple(y.shape), dtype=np.float32))
        y.backward(gy)
        fix[f"{name}.out"], fix[f"{name}.gy"], fix[f"{name}.gz"] = np_(y), np_(gy), np_(zz.grad)
        for k, p in m.named_parameters():
            fix[f"{name}.G.{k}"] = np_(p.grad)
    np.savez_compressed(os.path.join(HERE, f"triangle_N{N}.npz"), **fix)
    print("triangle golden written")


def golden_dataset_atom14():
    """make_atom14_masks + make_atom14_positions of the reference on the inputs of dataset_geom.npz."""
    from openfold.data import data_transforms
    g = np.load(os.path.join(HERE, "dataset_geom.npz"))
    prot = {"aatype": torch.tensor(g["aatype"]), "all_atom_positions": torch.tensor(g["all_atom_positions"]),
            "all_atom_mask": torch.tensor(g["all_atom_mask"])}
    prot = data_transforms.make_atom14_masks(prot)
    prot = data_transforms.make_atom14_positions(prot)
    keys = ("atom14_atom_exists", "residx_atom14_to_atom37", "residx_atom37_to_atom14", "atom37_atom_exists", "atom14_gt_exists",
            "atom14_gt_positions", "atom14_alt_gt_positions", "atom14_alt_gt_exists", "atom14_atom_is_ambiguous")
    np.savez_compressed(os.path.join(HERE, "dataset_atom14.npz"), **{k: np_(prot[k]) for k in keys})
    print("dataset_atom14 golden written", {k: (tuple(prot[k].shape), str(prot[k].dtype)) for k in keys})


def golden_pair_transition(N=24, seed=6):
    """PairTransition (openfold/model/pair_transition.py:24-99, Algorithm 15), the pair-stack neighbour of the triangle
    operators: output, input gradient and every parameter gradient of the reference module."""
    from openfold.model.pair_transition import PairTransition
    rng = np.random.default_rng(seed)
    fix = {}
    z = torch.tensor(rng.standard_normal((N, N, 128), dtype=np.float32))
    mask = torch.tensor((rng.uniform(size=(N, N)) > 0.1).astype(np.float32))
    fix["z"], fix["mask"] = np_(z), np_(mask)
    m = PairTransition(128, 4)
    sd = m.state_dict()
    for k, v in sd.items():
        if v.dim() >= 2:
            w = rng.standard_normal(tuple(v.shape), dtype=np.float32) / np.sqrt(v.shape[-1])
        elif k.endswith("weight"):
            w = 1.0 + 0.1 * rng.standard_normal(tuple(v.shape), dtype=np.float32)
        else:
            w = 0.1 * rng.standard_normal(tuple(v.shape), dtype=np.float32)
        sd[k] = torch.tensor(w.astype(np.float32))
        fix[f"P.{k}"] = w.astype(np.float32)
    m.load_state_dict(sd)
    zz = z.clone().requires_grad_(True)
    y = m(zz, mask=mask)
    gy = torch.tensor(rng.standard_normal(tuple(y.shape), dtype=np.float32))
    y.backward(gy)
    fix["out"], fix["gy"], fix["gz"] = np_(y), np_(gy), np_(zz.grad)
    for k, p_ in m.named_parameters():
        fix[f"G.{k}"] = np_(p_.grad)
    np.savez_compressed(os.path.join(HERE, f"pair_transition_N{N}.npz"), **fix)
    print("pair transition golden written", sorted(k for k in fix if k.startswith("P.")))


def _rand_sd(m, rng, fix, prefix):
    sd = m.state_dict()
    for k, v in sd.items():
        if v.dim() >= 2:
            w = rng.standard_normal(tuple(v.shape), dtype=np.float32) / np.sqrt(v.shape[-1])
        elif k.endswith("weight"):
            w = 1.0 + 0.1 * rng.standard_normal(tuple(v.shape), dtype=np.float32)
        else:
            w = 0.1 * rng.standard_normal(tuple(v.shape), dtype=np.float32)
        sd[k] = torch.tensor(w.astype(np.float32))
        fix[f"{prefix}P.{k}"] = w.astype(np.float32)
    m.load_state_dict(sd)


def golden_pair_stack(S=6, N=24, seed=17):
    """OuterProductMean (openfold/model/outer_product_mean.py:26-129) with output / input / parameter gradients, and one
    EvoformerBlockCore (openfold/model/evoformer.py:120-212) in eval mode (dropout = identity): m, z outputs and the
    gradients of both inputs.  c_m = 64 keeps the fixture small; c_z = 128, 32-channel outer product, 4 x 32 pair heads as
    openfold/config.py."""
    from openfold.model.evoformer import EvoformerBlockCore
    from openfold.model.outer_product_mean import OuterProductMean
    rng = np.random.default_rng(seed)
    fix = {}
    m = torch.tensor(rng.standard_normal((S, N, 64), dtype=np.float32))
    z = torch.tensor(rng.standard_normal((N, N, 128), dtype=np.float32))
    msa_mask = torch.tensor((rng.uniform(size=(S, N)) > 0.15).astype(np.float32))
    pair_mask = torch.tensor((rng.uniform(size=(N, N)) > 0.1).astype(np.float32))
    fix.update(m=np_(m), z=np_(z), msa_mask=np_(msa_mask), pair_mask=np_(pair_mask))
    opm = OuterProductMean(64, 128, 32)
    _rand_sd(opm, rng, fix, "opm.")
    mm = m.clone().requires_grad_(True)
    y = opm(mm, mask=msa_mask)
    gy = torch.tensor(rng.standard_normal(tuple(y.shape), dtype=np.float32))
    y.backward(gy)
    fix.update({"opm.out": np_(y), "opm.gy": np_(gy), "opm.gm": np_(mm.grad)})
    for k, p_ in opm.named_parameters():
        fix[f"opm.G.{k}"] = np_(p_.grad)
    core = EvoformerBlockCore(c_m=64, c_z=128, c_hidden_opm=32, c_hidden_mul=128, c_hidden_pair_att=32, no_heads_msa=8,
                              no_heads_pair=4, transition_n=2, pair_dropout=0.25, inf=1e9, eps=1e-10)
    _rand_sd(core, rng, fix, "core.")
    core.eval()
    mm, zz = m.clone().requires_grad_(True), z.clone().requires_grad_(True)
    mo, zo = core(mm, zz, msa_mask=msa_mask, pair_mask=pair_mask)
    gmo = torch.tensor(rng.standard_normal(tuple(mo.shape), dtype=np.float32))
    gzo = torch.tensor(rng.standard_normal(tuple(zo.shape), dtype=np.float32))
    ((mo * gmo).sum() + (zo * gzo).sum()).backward()
    fix.update({"core.m_out": np_(mo), "core.z_out": np_(zo), "core.gm_out": np_(gmo), "core.gz_out": np_(gzo),
                "core.gm": np_(mm.grad), "core.gz": np_(zz.grad)})
    fix["core.gnorm"] = np.array([float(p_.grad.norm()) for _, p_ in core.named_parameters()], np.float64)
    np.savez_compressed(os.path.join(HERE, f"pair_stack_S{S}_N{N}.npz"), **fix)
    print("pair stack golden written", len(fix), "arrays")


def golden_geoformer(S=5, N=24, seed=23):
    """OmegaFold pair-track operators as the reference vendors them (src/toolbox/OmegaFold/omegafold/modules.py):
    Node2Edge (:320-351) and GeometricAttention (:568-723) with OmegaFold's sizes (node 256, edge 128, 32-channel outer
    product, 4 heads x 32, 2 axes), forward only (the reference runs them under no_grad).  The package imports Biopython
    for PDB output at import time; it is not installed here and not needed by these modules, so it is stubbed."""
    import argparse
    import types

    class _Stub(types.ModuleType):
        def __getattr__(self, k):
            if k.startswith("__"):
                raise AttributeError(k)
            m = _Stub(self.__name__ + "." + k)
            setattr(self, k, m)
            return m
    for name in ("Bio", "Bio.Data", "Bio.Data.SCOPData", "Bio.PDB"):
        sys.modules.setdefault(name, _Stub(name))
    sys.modules["Bio.Data.SCOPData"].protein_letters_3to1 = {}
    sys.path.insert(0, os.path.join(os.environ.get("DFOLD_REFERENCE", "/root/reference"), "src", "toolbox", "OmegaFold"))
    from omegafold import modules
    rng = np.random.default_rng(seed)
    fix = {}
    node = torch.tensor(rng.standard_normal((S, N, 256), dtype=np.float32))
    edge = torch.tensor((rng.standard_normal((N, N, 128)) * 1.5 + 0.2).astype(np.float32))
    res_mask = np.ones(N, np.float32)
    res_mask[[3, N - 2]] = 0
    seq_mask = np.tile(res_mask, (S, 1))
    seq_mask[2, 7] = 0
    fix.update(node=np_(node), edge=np_(edge), res_mask=res_mask, seq_mask=seq_mask)
    n2e = modules.Node2Edge(in_dim=256, proj_dim=32, out_dim=128)
    ga = modules.GeometricAttention(d_edge=128, c=32, n_head=4, n_axis=2)
    for mod, pre, std in ((n2e, "n2e.", 0.08), (ga, "ga.", 0.09)):
        with torch.no_grad():
            for k, p_ in mod.named_parameters():
                p_.copy_(torch.tensor((rng.standard_normal(tuple(p_.shape)) * std).astype(np.float32)))
                fix[pre + "P." + k] = np_(p_)
    with torch.no_grad():
        fix["n2e.out"] = np_(n2e(node, torch.tensor(seq_mask)))
        fix["ga.out"] = np_(ga(edge, torch.tensor(res_mask), argparse.Namespace(subbatch_size=None)))
        sub7 = np_(ga(edge, torch.tensor(res_mask), argparse.Namespace(subbatch_size=7)))   # sub-batching changes nothing
    np.savez_compressed(os.path.join(HERE, f"geoformer_S{S}_N{N}.npz"), **fix)
    print("geoformer golden written", len(fix), "arrays; sub-batched vs whole:",
          float(np.abs(fix["ga.out"] - sub7).max()))


def golden_diffuser(exp, F=3, N=16):
    d = exp.diffuser
    so3, r3 = d._so3_diffuser, d._r3_diffuser
    fix = {}
    ts = np.array([0.01, 0.1, 0.25, 0.5, 0.77, 1.0])
    fix["ts"] = ts
    fix["t_to_idx"] = np.array([so3.t_to_idx(t) for t in ts])
    fix["sigma"] = np.array([so3.sigma(t) for t in ts])
    fix["so3_score_scaling"] = np.array([so3.score_scaling(t) for t in ts])
    fix["r3_score_scaling"] = np.array([r3.score_scaling(t) for t in ts])
    fix["so3_diffusion_coef"] = np.array([so3.diffusion_coef(t) for t in ts])
    sl = (slice(None, None, 37), slice(None, None, 41))
    fix["pdf_sub"], fix["cdf_sub"], fix["score_norms_sub"] = so3._pdf[sl], so3._cdf[sl], so3._score_norms[sl]
    fix["score_scaling_full"] = so3._score_scaling
    win = synthetic.synthetic_window(5, F, N)
    r0 = win["rigids_0"]
    fix["rigids_0"] = np_(r0)
    for i, t in enumerate((0.05, 0.5, 0.9)):
        np.random.seed(100 + i)
        fm = d.forward_marginal(Rigid.from_tensor_7(r0), float(t))
        for k, v in fm.items():
            fix[f"fm{i}_{k}"] = np_(v)
        # torch_score on the sampled geometry (fp32 quats in, mixed precision series)
        q_t = fm["rigids_t"][..., :4]
        rs = d.calc_rot_score(Rigid.from_tensor_7(fm["rigids_t"]).get_rots(), Rigid.from_tensor_7(r0).get_rots(),
                              torch.tensor([t], dtype=torch.float32))
        fix[f"fm{i}_calc_rot_score"] = np_(rs)
        tsq = d.calc_trans_score(fm["rigids_t"][..., 4:], r0[..., 4:], torch.tensor([t], dtype=torch.float32)[:, None, None], use_torch=True)
        fix[f"fm{i}_calc_trans_score"] = np_(tsq)
        # one reverse step with the draws recorded
        rng_state = 200 + i
        rot_score = np.asarray(fm["rot_score"]); trans_score = np.asarray(fm["trans_score"])
        np.random.seed(rng_state)
        z_rot = np.random.normal(size=rot_score.shape)      # so3.reverse draws first (se3_diffuser.py:184-190)
        z_trans = np.random.normal(size=trans_score.shape)  # then r3.reverse
        np.random.seed(rng_state)
        rig_prev = d.reverse(Rigid.from_tensor_7(fm["rigids_t"]), rot_score, trans_score, float(t), 0.1,
                             diffuse_mask=None, center=True, noise_scale=0.5)
        fix[f"rev{i}_z_rot"], fix[f"rev{i}_z_trans"] = z_rot, z_trans
        fix[f"rev{i}_rot_mats"] = np_(rig_prev.get_rots().get_rot_mats())
        fix[f"rev{i}_trans"] = np_(rig_prev.get_trans())
    np.random.seed(7)
    fix["sample_ref"] = np_(d.sample_ref(n_samples=F * N, as_tensor_7=True)["rigids_t"])
    np.savez_compressed(os.path.join(HERE, "diffuser.npz"), **fix)
    print("diffuser golden written")


def golden_dataset_geom(F=2, N=40, seed=11):
    """atom37 -> rigid-group frames / torsion angles with the reference's own dataset transforms
    (openfold/data/data_transforms.py:755-893, :923-1088), float64 as in the loader (Dfold_data_loader_dynamic.py:229-240).
    Coordinates: the oracle's frames->atoms builder on random frames / torsions (all 20 residue types + unknown), with a
    few atoms masked out so that the `exists` masks are exercised."""
    from openfold.data import data_transforms
    from oracle import dfold_oracle as O
    rng = np.random.default_rng(seed)
    aatype = torch.tensor(rng.integers(0, 21, size=(1, N))).long().expand(F, N).contiguous()
    aatype[:, :21] = torch.arange(21)[None]                       # every residue type at least once
    q = torch.tensor(rng.standard_normal((F, N, 4)))
    q = q / q.norm(dim=-1, keepdim=True)
    t7 = torch.cat([q, torch.tensor(rng.standard_normal((F, N, 3)) * 12)], -1)
    ang = torch.tensor(rng.standard_normal((F, N, 7, 2)))
    ang = ang / ang.norm(dim=-1, keepdim=True)
    _, atom37 = O.frames_to_atoms(t7, ang, torch.clamp(aatype, max=20))
    mask = O.residue_tables()["atom37_mask"][torch.clamp(aatype, max=20)].double().clone()
    drop = torch.tensor(rng.uniform(size=mask.shape) < 0.04)
    mask[drop] = 0
    atom37 = atom37.double() * mask[..., None]
    prot = {"aatype": aatype, "all_atom_positions": atom37.clone(), "all_atom_mask": mask.clone()}
    prot = data_transforms.atom37_to_frames(prot)
    prot = data_transforms.atom37_to_torsion_angles()(prot)
    fix = {"aatype": np_(aatype), "all_atom_positions": np_(atom37), "all_atom_mask": np_(mask)}
    for k in ("rigidgroups_gt_frames", "rigidgroups_gt_exists", "rigidgroups_group_exists", "rigidgroups_group_is_ambiguous",
              "rigidgroups_alt_gt_frames", "torsion_angles_sin_cos", "alt_torsion_angles_sin_cos", "torsion_angles_mask"):
        fix[k] = np_(prot[k])
    np.savez_compressed(os.path.join(HERE, "dataset_geom.npz"), **fix)
    print("dataset_geom golden written", {k: v.shape for k, v in fix.items()})


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "dataset_geom":
        golden_dataset_geom()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "dataset_atom14":
        golden_dataset_atom14()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "network_F8":
        # second network capture: 8 frames (interior frames of the 5-tap conv axis, a last-frame cone that is clipped only
        # in the first block), other seeds / diffusion time; outputs, loss, gradient norms and sparse gradient samples only
        golden_network(F=8, N=16, seed_w=2, seed_x=5, t=0.3, captures=False, grad_stride=39989)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "network_cfg1":
        # BASELINE config 1 shape: one window of 16 frames x N_res 96 (the reference's CPU-runnable configuration)
        golden_network(F=16, N=96, seed_w=11, seed_x=12, t=0.4, captures=False, grad_stride=39989, compact=True,
                       no_torsion_grads=True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "network_n256":
        # run_train.sh window (frame_time = 2) at the headline N_res = 256
        golden_network(F=2, N=256, seed_w=13, seed_x=14, t=0.6, captures=False, grad_stride=39989, compact=True,
                       no_torsion_grads=True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "network_cfg3":
        # BASELINE config 3 / 4: one 32-frame window at N_res 256 (the shape bench.py times, B = 8 independent windows)
        golden_network(F=32, N=256, seed_w=21, seed_x=22, t=0.45, captures=False, grad_stride=39989, compact=True,
                       no_torsion_grads=True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "network_cfg2":
        # BASELINE config 2: one 32-frame window at N_res 128
        golden_network(F=32, N=128, seed_w=23, seed_x=24, t=0.35, captures=False, grad_stride=39989, compact=True,
                       no_torsion_grads=True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "network_cfg5":
        # BASELINE config 5 is 64 frames x N_res 512: the reference cannot hold its IPA intermediates;
        # 8 frames at N_res 512 is what it can run
        golden_network(F=8, N=512, seed_w=25, seed_x=26, t=0.55, captures=False, grad_stride=39989, compact=True,
                       no_torsion_grads=True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "network_holes":
        # res_mask with holes (10 % dead residues + both chain ends dead): masks of IPA, frame update, score heads,
        # loss normalisers and the whole-tensor MyLayerNorm with dead residues, end to end
        golden_network(F=6, N=40, seed_w=27, seed_x=28, t=0.5, captures=False, grad_stride=39989, holes=0.1, tag="_holes")
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "sampler":
        golden_sampler()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "sampler_cfg1":
        # BASELINE config 1 IS an eval configuration (run_eval.sh:4-17, config/eval_DFOLDv2.yaml: 16-frame window, N_res ~ 96,
        # data.num_t = 10, noise_scale = 0.1): the reference's own inference_fn at that size
        golden_sampler(F=16, N=96, seed_w=31, seed_x=32, num_t=10, noise_scale=0.1, seed_z=91, compact=True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "bf16_sensitivity":
        golden_bf16_sensitivity(big=len(sys.argv) > 2 and sys.argv[2] == "big")
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "pair_transition":
        golden_pair_transition()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "pair_stack":
        golden_pair_stack()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "geoformer":
        golden_geoformer()
        sys.exit(0)
    exp = golden_network()
    golden_diffuser(exp)
    golden_triangle()
    golden_dataset_geom()
    golden_pair_transition()
    golden_pair_stack()
    golden_geoformer()
