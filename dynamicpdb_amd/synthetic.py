"""Deterministic synthetic weights and trajectory windows.

No dataset or checkpoint ships with the reference and there is no network, so
tests, smoke() and bench.py all draw from here: `seeded_state_dict` fills a
state_dict with the reference's exact key names / shapes (so it loads into the
reference model AND into this engine), `synthetic_window` makes one smooth
trajectory window with the reference dataset's feature keys
(src/data/Dfold_data_loader_dynamic.py:323-358 of the reference).
"""
import math
from collections import OrderedDict

import numpy as np
import torch

C_S, C_Z, H, C_HID, PQ, PV, NBLK = 256, 128, 8, 256, 8, 12, 4


def param_shapes():
    """state_dict key -> shape of the reference FullScoreNetwork with yaml defaults
    (verified against the reference: 184,419,962 parameters)."""
    s = OrderedDict()
    e = "embedding_layer."
    s[e + "node_timestep_proj.0.weight"] = (128, 256); s[e + "node_timestep_proj.0.bias"] = (128,)
    s[e + "node_timestep_proj.2.weight"] = (256, 128); s[e + "node_timestep_proj.2.bias"] = (256,)
    s[e + "node_ln.weight"] = (256,); s[e + "node_ln.bias"] = (256,)
    s[e + "edge_timestep_proj.0.weight"] = (64, 256); s[e + "edge_timestep_proj.0.bias"] = (64,)
    s[e + "edge_timestep_proj.2.weight"] = (128, 64); s[e + "edge_timestep_proj.2.bias"] = (128,)
    s[e + "edge_ln.weight"] = (128,); s[e + "edge_ln.bias"] = (128,)
    t = "score_model.trunk."
    for b in range(NBLK):
        p = f"{t}ipa_{b}."
        s[p + "head_weights"] = (H,)
        for n, o, i in (("linear_q", H * C_HID, C_S), ("linear_kv", 2 * H * C_HID, C_S),
                        ("linear_q_points", H * PQ * 3, C_S), ("linear_kv_points", H * (PQ + PV) * 3, C_S),
                        ("linear_b", H, C_Z), ("down_z", C_Z // 4, C_Z),
                        ("linear_out", C_S, H * (C_Z // 4 + C_HID + PV * 8)), ("linear_rbf", 1, 20)):
            s[p + n + ".weight"] = (o, i); s[p + n + ".bias"] = (o,)
        s[f"{t}bb_update_{b}.linear.weight"] = (6, 5 * C_S); s[f"{t}bb_update_{b}.linear.bias"] = (6,)
    d = 5 * C_S
    for i in (1, 2, 3, 4):
        s[f"{t}conv_0.conv{i}.0.weight"] = (d // 2, d, 5, 5); s[f"{t}conv_0.conv{i}.0.bias"] = (d // 2,)
        s[f"{t}conv_0.conv{i}.2.weight"] = (d, d // 2, 5, 5); s[f"{t}conv_0.conv{i}.2.bias"] = (d,)
    a = "score_model.angle_resnet."
    for n in ("linear_in", "linear_initial", "layers.0.linear_1", "layers.0.linear_2",
              "layers.1.linear_1", "layers.1.linear_2"):
        s[a + n + ".weight"] = (d, d); s[a + n + ".bias"] = (d,)
    s[a + "linear_out.weight"] = (14, d); s[a + "linear_out.bias"] = (14,)
    for n, k in (("force", 3), ("vel", 3), ("index", 1), ("rigid", 7), ("angle", 14)):
        p = f"score_model.{n}_embeder."
        s[p + "0.weight"] = (C_S, k); s[p + "0.bias"] = (C_S,)
        s[p + "2.weight"] = (C_S, C_S); s[p + "2.bias"] = (C_S,)
    s["expand_node.weight"] = (C_S, 256); s["expand_node.bias"] = (C_S,)
    s["expand_edge.weight"] = (C_Z, 128); s["expand_edge.bias"] = (C_Z,)
    return s


def seeded_state_dict(seed=0, dtype=torch.float32):
    """Every tensor non-degenerate: fan-in scaled normals for weights (the reference's
    zero-initialised 'final' layers get a small non-zero scale so that every path
    contributes, SURVEY section 7 'hard parts'), small normals for biases."""
    rng = np.random.default_rng(seed)
    out = OrderedDict()
    for k, shp in param_shapes().items():
        if k.endswith("head_weights"):
            v = 0.541324854612918 + 0.1 * rng.standard_normal(shp)
        elif k.endswith("_ln.weight"):
            v = 1.0 + 0.05 * rng.standard_normal(shp)
        elif k.endswith(".bias"):
            v = 0.02 * rng.standard_normal(shp)
        else:
            fan_in = int(np.prod(shp[1:]))
            scale = 1.0 / math.sqrt(fan_in)
            if ".conv" in k:
                scale *= 1.3
            if "bb_update" in k:
                scale *= 0.02            # keeps the per-block rigid update small but non-zero
            if "linear_out" in k and "ipa_" in k:
                scale *= 0.7
            if "layers" in k and k.endswith("linear_2.weight"):
                scale *= 0.5
            v = scale * rng.standard_normal(shp, dtype=np.float32)
        out[k] = torch.tensor(np.asarray(v, np.float32)).to(dtype)
    return out


def _quat_mul_np(p, q):
    a1, b1, c1, d1 = np.moveaxis(p, -1, 0)
    a2, b2, c2, d2 = np.moveaxis(q, -1, 0)
    return np.stack([a1 * a2 - b1 * b2 - c1 * c2 - d1 * d2, a1 * b2 + b1 * a2 + c1 * d2 - d1 * c2,
                     a1 * c2 - b1 * d2 + c1 * a2 + d1 * b2, a1 * d2 + b1 * c2 - c1 * b2 + d1 * a2], -1)


def synthetic_window(seed, F, N, t=0.5, diffuser=None, rigid_cls=None, holes=0.0):
    """One window of F frames x N residues as float32/int64 torch CPU tensors with the
    reference dataset's keys.  Smooth trajectory: frame f+1 = frame f composed with a small
    rigid perturbation (rotvec sigma 0.05 rad, translation sigma 0.3 A) so the reference's
    trans_loss<100 gate stays open.  If `diffuser` (an SE3Diffuser) is given, rigids_t and the
    target scores come from its forward_marginal under numpy seed `seed`; otherwise only the
    model inputs that do not need a diffuser are filled.  `rigid_cls`: the Rigid class handed to the diffuser (the
    golden-minting script passes the REFERENCE's openfold Rigid so that no product code sits between the seed and the
    reference's forward_marginal).  `holes` > 0: res_mask with that fraction of dead residues plus one dead residue at
    each end of the chain (the loader's res_mask is the CA mask, src/data/Dfold_data_loader_dynamic.py:248, and can
    have holes); drawn from a separate stream so that the other tensors do not depend on it."""
    rng = np.random.default_rng(seed)
    q0 = rng.standard_normal((N, 4))
    q0 /= np.linalg.norm(q0, axis=-1, keepdims=True)
    steps = rng.standard_normal((N, 3))
    steps *= 3.8 / np.linalg.norm(steps, axis=-1, keepdims=True)
    x0 = np.cumsum(steps, 0)
    x0 -= x0.mean(0, keepdims=True)
    quats, trans = [q0], [x0]
    for _ in range(F - 1):
        rv = 0.05 * rng.standard_normal((N, 3))
        ang = np.linalg.norm(rv, axis=-1, keepdims=True)
        dq = np.concatenate([np.cos(ang / 2), np.sin(ang / 2) * rv / np.maximum(ang, 1e-12)], -1)
        qn = _quat_mul_np(quats[-1], dq)
        quats.append(qn / np.linalg.norm(qn, axis=-1, keepdims=True))
        trans.append(trans[-1] + 0.3 * rng.standard_normal((N, 3)))
    rigids_0 = np.concatenate([np.stack(quats), np.stack(trans)], -1).astype(np.float32)
    ang = rng.uniform(-np.pi, np.pi, (F, N, 7))
    sincos = np.stack([np.sin(ang), np.cos(ang)], -1).astype(np.float32)
    aatype = np.repeat(rng.integers(0, 20, (1, N)), F, 0).astype(np.int64)
    w = dict(
        aatype=torch.tensor(aatype),
        seq_idx=torch.arange(1, N + 1, dtype=torch.int64)[None].repeat(F, 1),
        res_mask=torch.ones(F, N), fixed_mask=torch.zeros(F, N),
        node_repr=torch.tensor(rng.standard_normal((N, 256), dtype=np.float32)),
        edge_repr=torch.tensor(rng.standard_normal((N, N, 128), dtype=np.float32)),
        rigids_0=torch.tensor(rigids_0),
        force=torch.tensor(rng.standard_normal((F, N, 3), dtype=np.float32)),
        vel=torch.tensor(rng.standard_normal((F, N, 3), dtype=np.float32)),
        torsion_angles_sin_cos=torch.tensor(sincos),
        alt_torsion_angles_sin_cos=torch.tensor(-sincos),
        torsion_angles_mask=torch.ones(F, N, 7),
        t=torch.tensor([t], dtype=torch.float32),
    )
    w["sc_ca_t"] = torch.zeros(F, N, 3)
    if holes > 0:
        dead = np.random.default_rng(seed + 7919).uniform(size=N) < holes
        dead[0] = dead[-1] = True
        w["res_mask"] = torch.tensor(np.repeat((~dead)[None], F, 0).astype(np.float32))
    if diffuser is not None:
        if rigid_cls is None:
            from .rigid import Rigid
        else:
            Rigid = rigid_cls
        state = np.random.get_state()
        np.random.seed(seed)
        fm = diffuser.forward_marginal(Rigid.from_tensor_7(w["rigids_0"]), t, diffuse_mask=None, as_tensor_7=True)
        np.random.set_state(state)
        rs, ts = diffuser.score_scaling(t)
        w["rigids_t"] = fm["rigids_t"].to(torch.float32)
        w["rot_score"] = torch.tensor(np.asarray(fm["rot_score"]))
        w["trans_score"] = torch.tensor(np.asarray(fm["trans_score"]))
        w["rot_score_scaling"] = torch.tensor([rs])
        w["trans_score_scaling"] = torch.tensor([ts])
    return w


def device_batch(rng, diffuser, B, F, N, t=0.5):
    """A batch of B synthetic windows generated ON THE DEVICE (a few milliseconds: a fresh batch per training step can be
    staged in HBM ahead of a timed region): same keys, shapes, dtypes and distributions as B stacked `synthetic_window`s
    (smooth trajectories: rotvec sigma 0.05 rad, translation sigma 0.3 A per frame; 3.8 A random-walk chain), draws from
    a dynamicpdb_amd.rng.DeviceRNG (Philox, reproducible from its seed), noising by SE3Diffuser.forward_marginal_t7."""
    dev = rng.device
    f32 = torch.float32
    nrm = lambda *shape: rng.normal(shape)
    q0 = nrm(B, N, 4)
    q0 = q0 / q0.norm(dim=-1, keepdim=True)
    steps = nrm(B, N, 3)
    steps = steps * (3.8 / steps.norm(dim=-1, keepdim=True))
    x0 = torch.cumsum(steps, 1)
    x0 = x0 - x0.mean(1, keepdim=True)
    rv = 0.05 * nrm(B, max(F - 1, 1), N, 3)
    ang = rv.norm(dim=-1, keepdim=True)
    dq = torch.cat([torch.cos(ang / 2), torch.sin(ang / 2) * rv / ang.clamp_min(1e-12)], -1)
    quats = [q0]
    for f in range(F - 1):
        a1, b1, c1, d1 = quats[-1].unbind(-1)
        a2, b2, c2, d2 = dq[:, f].unbind(-1)
        qn = torch.stack([a1 * a2 - b1 * b2 - c1 * c2 - d1 * d2, a1 * b2 + b1 * a2 + c1 * d2 - d1 * c2,
                          a1 * c2 - b1 * d2 + c1 * a2 + d1 * b2, a1 * d2 + b1 * c2 - c1 * b2 + d1 * a2], -1)
        quats.append(qn / qn.norm(dim=-1, keepdim=True))
    drift = 0.3 * nrm(B, max(F - 1, 1), N, 3)
    trans = torch.cat([x0[:, None], x0[:, None] + torch.cumsum(drift, 1)[:, :F - 1]], 1)
    rigids_0 = torch.cat([torch.stack(quats, 1), trans], -1).to(f32)
    angs = (rng.uniform((B, F, N, 7)) * 2 - 1) * math.pi
    sincos = torch.stack([torch.sin(angs), torch.cos(angs)], -1).to(f32)
    aatype = (rng.uniform((B, 1, N)) * 20).long().clamp_(0, 19).expand(B, F, N).contiguous()
    w = dict(
        aatype=aatype, seq_idx=torch.arange(1, N + 1, dtype=torch.int64, device=dev)[None, None].expand(B, F, N).contiguous(),
        res_mask=torch.ones(B, F, N, device=dev), fixed_mask=torch.zeros(B, F, N, device=dev),
        node_repr=nrm(B, N, 256).to(f32), edge_repr=nrm(B, N, N, 128).to(f32), rigids_0=rigids_0,
        force=nrm(B, F, N, 3).to(f32), vel=nrm(B, F, N, 3).to(f32),
        torsion_angles_sin_cos=sincos, alt_torsion_angles_sin_cos=-sincos,
        torsion_angles_mask=torch.ones(B, F, N, 7, device=dev), sc_ca_t=torch.zeros(B, F, N, 3, device=dev),
        t=torch.full((B,), float(t), dtype=f32, device=dev))
    w["t_host"] = np.full((B,), float(t))          # the diffusion times as host numbers (no device -> host copy in the forward)
    from .model.ipa_pytorch_dynamic import stamp_t_host
    stamp_t_host(w)
    fm = diffuser.forward_marginal_t7(rigids_0, torch.full((B,), float(t), dtype=torch.float64), rng=rng)
    w["rigids_t"] = fm["rigids_t"].to(f32)
    w["rot_score"], w["trans_score"] = fm["rot_score"], fm["trans_score"]
    w["rot_score_scaling"] = torch.as_tensor(np.asarray(fm["rot_score_scaling"], np.float64)).reshape(B, 1).to(dev)
    w["trans_score_scaling"] = torch.as_tensor(np.asarray(fm["trans_score_scaling"], np.float64)).reshape(B, 1).to(dev)
    return w


def step_flops_fwd(F, N, inner_cone=False):
    """Algorithmic forward FLOPs of one window (conv: non-padding taps only).  fwd + bwd = 3 x.
    inner_cone: the trunk's two inner blocks evaluate the conv tower on the last frame's dependency cone
    (DFOLDIpaScore.trunk_dce): their conv term is scaled by the fraction of (layer, frame) pairs inside the cone."""
    T = lambda L: 5 * L - 6
    P, H, C, PQ, PV = F * N, 8, 256, 8, 12
    conv = 4 * 8 * 2 * 1280 * 640 * T(F) * T(N)
    if inner_cone:
        inside = sum(min(F, 4 * (3 - i) + 3) + min(F, 4 * (3 - i) + 1) for i in range(4))      # ops.ConvTower.cone
        conv = conv // 4 * 2 + int(conv // 4 * 2 * inside / (8.0 * F))
    ipa = 4 * (2 * P * 256 * 6816 + 2 * N * N * 128 * 40 + 4 * F * H * N * N * C + 9 * F * N * N * H * PQ
               + 6 * F * H * N * N * PV + 64 * F * H * N * N + 2 * P * 3072 * 256)
    angle = 2 * P * (1280 * 1280 * 6 + 1280 * 14)
    emb = 8 * P * (7 * 256 + 256 * 256) + 2 * P * (20 * 256 + 3 * 256 * 256)
    return conv + ipa + angle + emb + 48 * P * 1280 + 2 * N * 256 * 256 + 2 * N * N * 128 * 128


def default_conf(frame_time, cache_dir=".cache/"):
    """config/train_DFOLDv2.yaml + run_train.sh overrides of the reference, as an
    attribute dict (same tree: data / diffuser / model / experiment)."""

    class AD(dict):
        def __init__(self, d=None, **kw):
            super().__init__()
            for k, v in dict(d or {}, **kw).items():
                self[k] = AD(v) if isinstance(v, dict) and not isinstance(v, AD) else v

        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError as e:
                raise AttributeError(k) from e

        def __setattr__(self, k, v):
            self[k] = v

    Fr = frame_time
    return AD(
        diffuser=dict(dynamics=True, frame_time=Fr, diffuse_trans=True, diffuse_rot=True,
                      r3=dict(min_b=0.1, max_b=20.0, coordinate_scaling=1.0),
                      so3=dict(num_omega=1000, num_sigma=1000, min_sigma=0.1, max_sigma=1.5,
                               schedule="logarithmic", cache_dir=cache_dir, use_cached_score=False)),
        model=dict(cfg_drop_rate=0.0, cfg_drop_in_train=True, cfg_gamma=2, frame_time=Fr, dynamics=True,
                   node_embed_size=256, edge_embed_size=128, dropout=0.0,
                   embed=dict(DFOLDv2_embedder=True, index_embed_size=32, aatype_embed_size=32,
                              embed_self_conditioning=True, num_bins=22, min_bin=1e-5, max_bin=20.0,
                              skip_feature=False),
                   ipa=dict(c_s=256, c_z=128, c_hidden=256, c_skip=64, no_heads=8, no_qk_points=8,
                            no_v_points=12, seq_tfmr_num_heads=4, seq_tfmr_num_layers=2, num_blocks=4,
                            coordinate_scaling=1.0, spatial=True, temporal=False)),
        data=dict(dynamics=True, frame_time=Fr, min_t=0.01, num_t=10, is_extrapolation=False),
        experiment=dict(training=False, use_ddp=False, learning_rate=1e-4, trans_loss_weight=100.0,
                        rot_loss_weight=7.0, rot_loss_t_threshold=0.0, separate_rot_loss=False,
                        torsion_loss_weight=1.0, coordinate_scaling=1.0, noise_scale=1.0, name="synthetic"),
    )
