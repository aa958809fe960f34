"""GPU parity of the device reverse-diffusion (denoise) step against reference-minted golden vectors and the host
mirror of SE3Diffuser.reverse, and a smoke test of the device-resident sampler (inference_fn)."""
import numpy as np
import pytest
import torch

from util import load_golden, max_abs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def diffuser():
    from dynamicpdb_amd import synthetic
    from dynamicpdb_amd.data.se3_diffuser import SE3Diffuser
    return SE3Diffuser(synthetic.default_conf(3, cache_dir="/tmp/dfold_igso3_cache/").diffuser)


def _rotmats(t7):
    from dynamicpdb_amd.model import geometry as G
    q = t7[..., :4]
    return G.quat_to_rot(q / q.norm(dim=-1, keepdim=True))


def test_reverse_step_vs_reference_golden(diffuser):
    """same inputs and normal draws as the reference's own SE3Diffuser.reverse call (tests/golden/diffuser.npz)"""
    dev = torch.device("cuda:0")
    g = load_golden("diffuser.npz")
    for i, t in enumerate((0.05, 0.5, 0.9)):
        rt = torch.tensor(g[f"fm{i}_rigids_t"]).to(dev)
        out = diffuser.reverse_t7(rt, g[f"fm{i}_rot_score"], g[f"fm{i}_trans_score"], float(t), 0.1, diffuse_mask=None,
                                  center=True, noise_scale=0.5, z_rot=g[f"rev{i}_z_rot"], z_trans=g[f"rev{i}_z_trans"])
        assert max_abs(_rotmats(out), g[f"rev{i}_rot_mats"]) < 2e-5
        assert max_abs(out[..., 4:], g[f"rev{i}_trans"]) < 1e-4


def test_reverse_step_device_vs_host_mirror_with_mask(diffuser):
    from dynamicpdb_amd.rigid import Rigid
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(3)
    B, F, N = 2, 3, 24
    q = rng.standard_normal((B, F, N, 4))
    q /= np.linalg.norm(q, axis=-1, keepdims=True)
    t7 = torch.tensor(np.concatenate([q, 10 * rng.standard_normal((B, F, N, 3))], -1), dtype=torch.float32)
    rs, ts = 0.3 * rng.standard_normal((B, F, N, 3)), rng.standard_normal((B, F, N, 3))
    zr, zt = rng.standard_normal((B, F, N, 3)), rng.standard_normal((B, F, N, 3))
    mask = (rng.uniform(size=(B, F, N)) > 0.2).astype(np.float32)
    for center in (True, False):
        host = diffuser.reverse(Rigid.from_tensor_7(t7), rs, ts, 0.4, 0.1, diffuse_mask=mask, center=center, noise_scale=0.7,
                                z_rot=zr, z_trans=zt)
        devo = diffuser.reverse_t7(t7.to(dev), rs, ts, 0.4, 0.1, diffuse_mask=mask, center=center, noise_scale=0.7,
                                   z_rot=zr, z_trans=zt)
        assert max_abs(_rotmats(devo), host.get_rots().get_rot_mats()) < 2e-5
        assert max_abs(devo[..., 4:], host.get_trans()) < 1e-4
        # Rigid API on device frames routes to the same kernel
        r2 = diffuser.reverse(Rigid.from_tensor_7(t7.to(dev)), rs, ts, 0.4, 0.1, diffuse_mask=mask, center=center,
                              noise_scale=0.7, z_rot=zr, z_trans=zt)
        assert torch.equal(r2.to_tensor_7(), devo)
    with pytest.raises(ValueError):
        diffuser.reverse_t7(t7.to(dev), rs, ts, np.array([0.4, 0.5]), 0.1)


def test_device_sampler_runs_and_is_reproducible():
    from dynamicpdb_amd import experiment, synthetic
    from dynamicpdb_amd.data.se3_diffuser import SE3Diffuser
    from dynamicpdb_amd.model.Dfold_network_dynamic import FullScoreNetwork
    dev = torch.device("cuda:0")
    F, N, num_t = 3, 16, 3
    conf = synthetic.default_conf(F, cache_dir="/tmp/dfold_igso3_cache/")
    diffuser = SE3Diffuser(conf.diffuser)
    model = FullScoreNetwork(conf.model, diffuser)
    model.load_state_dict(synthetic.seeded_state_dict(0), strict=True)
    model.to(dev)
    w = synthetic.synthetic_window(2, F, N, t=1.0, diffuser=diffuser)
    np.random.seed(0)
    w["rigids_t"] = diffuser.sample_ref(n_samples=F * N, as_tensor_7=True)["rigids_t"].float()     # prior sample (eval_fn :819)
    init = {k: v[None].to(dev) for k, v in w.items()}
    init["t"] = w["t"].to(dev)
    rng = np.random.default_rng(5)
    zs = [(rng.standard_normal((1, F, N, 3)), rng.standard_normal((1, F, N, 3))) for _ in range(num_t)]
    a = experiment.inference_fn(model, diffuser, init, num_t=num_t, min_t=0.01, aux_traj=True, noise_scale=0.1, z_draws=zs)
    b = experiment.inference_fn(model, diffuser, init, num_t=num_t, min_t=0.01, aux_traj=True, noise_scale=0.1, z_draws=zs)
    assert a["prot_traj"].shape == (num_t, 1, F, N, 37, 3) and a["rigid_traj"].shape == (num_t, 1, F, N, 7)
    assert np.isfinite(a["prot_traj"]).all() and np.isfinite(a["rigid_traj"]).all()
    assert np.abs(a["prot_traj"] - b["prot_traj"]).max() < 1e-3        # same draws -> same trajectory
    assert tuple(a["psi_pred"].shape) == (1, F, N, 7, 2)
