"""CPU-side tests (no GPU): the C-ABI library loads and exports every declared symbol, the host mirror of the
reference's diffuser API reproduces reference-minted golden vectors, the drop-in modules carry the reference's
state_dict, the loss matches the oracle, the device path refuses CPU tensors (no fallback), and data-parallel
gradient averaging works across 2 gloo ranks."""
import os
import sys

import numpy as np
import pytest
import torch

from util import load_golden, max_abs, rel_l2, window_from_golden

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_cabi_library_exports_every_declared_symbol():
    import __graft_entry__
    __graft_entry__.build()                       # hipcc cross-compiles for gfx950 without a GPU
    from dynamicpdb_amd import _lib
    syms = _lib.header_symbols()
    assert len(syms) >= 25 and "dfold_gemm_bf16" in syms and "dfold_ipa_softmax_fwd" in syms
    L = _lib.lib()
    for s in syms:
        assert hasattr(L, s), s
    assert L.dfold_abi_version() == 1


def test_cabi_rejects_bad_arguments_without_a_gpu():
    """argument validation happens before any launch: EINVAL (-1) comes back on a box with no device"""
    from ctypes import byref, c_void_p
    from dynamicpdb_amd import _lib
    L = _lib.lib()
    d = _lib.GemmDesc()
    assert L.dfold_gemm_bf16(byref(d), c_void_p(0)) == -1
    assert L.dfold_gemm_bf16(None, c_void_p(0)) == -1
    with pytest.raises(ValueError):
        _lib.check(-1, "x")
    with pytest.raises(RuntimeError):
        _lib.check(-2, "x")


@pytest.fixture(scope="module")
def diffuser():
    from dynamicpdb_amd import synthetic
    from dynamicpdb_amd.data.se3_diffuser import SE3Diffuser
    return SE3Diffuser(synthetic.default_conf(3, cache_dir="/tmp/dfold_igso3_cache/").diffuser)


def test_diffuser_schedules_match_reference(diffuser):
    g = load_golden("diffuser.npz")
    so3, r3 = diffuser._so3_diffuser, diffuser._r3_diffuser
    ts = g["ts"]
    assert np.array_equal(np.array([so3.t_to_idx(t) for t in ts]), g["t_to_idx"])          # bit-exact np.digitize index
    assert np.allclose([so3.sigma(t) for t in ts], g["sigma"], rtol=1e-12)
    assert np.allclose([so3.score_scaling(t) for t in ts], g["so3_score_scaling"], rtol=1e-6)
    assert np.allclose([r3.score_scaling(t) for t in ts], g["r3_score_scaling"], rtol=1e-10)
    assert np.allclose([so3.diffusion_coef(t) for t in ts], g["so3_diffusion_coef"], rtol=1e-10)
    sl = (slice(None, None, 37), slice(None, None, 41))
    assert np.allclose(so3._pdf[sl], g["pdf_sub"], rtol=1e-6, atol=1e-12)
    assert np.allclose(so3._cdf[sl], g["cdf_sub"], rtol=1e-6, atol=1e-12)
    assert np.allclose(so3._score_norms[sl], g["score_norms_sub"], rtol=1e-5, atol=1e-8)
    with pytest.raises(ValueError):
        so3.sigma(1.5)
    with pytest.raises(ValueError):
        r3.b_t(-0.1)


def test_forward_marginal_and_reverse_match_reference(diffuser):
    from dynamicpdb_amd.rigid import Rigid
    g = load_golden("diffuser.npz")
    r0 = torch.tensor(g["rigids_0"])
    for i, t in enumerate((0.05, 0.5, 0.9)):
        np.random.seed(100 + i)                        # same numpy global-RNG stream as the reference call
        fm = diffuser.forward_marginal(Rigid.from_tensor_7(r0), float(t))
        assert max_abs(fm["rigids_t"][..., 4:], g[f"fm{i}_rigids_t"][..., 4:]) < 1e-4
        q, qr = fm["rigids_t"][..., :4], torch.tensor(g[f"fm{i}_rigids_t"][..., :4])
        assert float((1 - (q * qr).sum(-1).abs()).max()) < 1e-5           # same rotation up to the sign of q
        assert rel_l2(fm["rot_score"], g[f"fm{i}_rot_score"]) < 1e-4
        assert rel_l2(fm["trans_score"], g[f"fm{i}_trans_score"]) < 1e-5
        rig_prev = diffuser.reverse(Rigid.from_tensor_7(torch.tensor(g[f"fm{i}_rigids_t"])), g[f"fm{i}_rot_score"],
                                    g[f"fm{i}_trans_score"], float(t), 0.1, diffuse_mask=None, center=True, noise_scale=0.5,
                                    z_rot=g[f"rev{i}_z_rot"], z_trans=g[f"rev{i}_z_trans"])
        assert max_abs(rig_prev.get_rots().get_rot_mats(), g[f"rev{i}_rot_mats"]) < 2e-5
        assert max_abs(rig_prev.get_trans(), g[f"rev{i}_trans"]) < 1e-4
    np.random.seed(7)
    ref = diffuser.sample_ref(n_samples=3 * 16, as_tensor_7=True)["rigids_t"]
    assert max_abs(ref[..., 4:], g["sample_ref"][..., 4:]) < 1e-4


def test_modules_carry_reference_state_dict(diffuser):
    from dynamicpdb_amd import synthetic
    from dynamicpdb_amd.model.Dfold_network_dynamic import FullScoreNetwork
    model = FullScoreNetwork(synthetic.default_conf(3).model, diffuser)
    sd = model.state_dict()
    shapes = synthetic.param_shapes()
    assert list(sd.keys()) == list(shapes.keys()) or set(sd.keys()) == set(shapes.keys())
    for k, shp in shapes.items():
        assert tuple(sd[k].shape) == tuple(shp), k
    assert sum(p.numel() for p in model.parameters()) == 184_419_962
    # zero-initialised 'final' layers as in the reference (ipa_pytorch_dynamic.py:305,590)
    assert float(model.score_model.trunk["bb_update_0"].linear.weight.abs().max()) == 0
    assert float(model.score_model.trunk["ipa_0"].linear_out.weight.abs().max()) == 0


def test_device_path_has_no_cpu_fallback(diffuser):
    from dynamicpdb_amd import synthetic
    from dynamicpdb_amd.model.Dfold_network_dynamic import FullScoreNetwork
    from dynamicpdb_amd.model.triangle import TriangleMultiplicationOutgoing
    model = FullScoreNetwork(synthetic.default_conf(3).model, diffuser)
    w = synthetic.synthetic_window(1, 3, 16, diffuser=diffuser)
    with pytest.raises(RuntimeError):
        model(w)
    with pytest.raises(RuntimeError):
        TriangleMultiplicationOutgoing(128, 128)(torch.zeros(8, 8, 128))
    from dynamicpdb_amd.data import data_transforms
    from dynamicpdb_amd.model.triangle import PairTransition
    from dynamicpdb_amd.optim import FusedAdam
    with pytest.raises(RuntimeError):
        PairTransition(128, 4)(torch.zeros(8, 8, 128))
    prot = {"aatype": torch.zeros(2, 8, dtype=torch.long), "all_atom_positions": torch.zeros(2, 8, 37, 3, dtype=torch.float64),
            "all_atom_mask": torch.ones(2, 8, 37, dtype=torch.float64)}
    with pytest.raises(RuntimeError):
        data_transforms.atom37_to_frames(prot)
    with pytest.raises(RuntimeError):
        data_transforms.atom37_to_torsion_angles()(prot)
    with pytest.raises(RuntimeError):
        diffuser.forward_marginal_t7(torch.zeros(3, 8, 7), 0.5)
    p = torch.nn.Parameter(torch.zeros(4))
    p.grad = torch.ones(4)
    with pytest.raises(ValueError):
        FusedAdam([p], lr=1e-3).step()            # host tensors: refused, never silently stepped on the CPU


def test_batched_loss_matches_oracle():
    from oracle import dfold_oracle as O
    from dynamicpdb_amd import experiment
    g = load_golden("network_F3_N16.npz")
    w = window_from_golden(g)
    out = {k[4:]: torch.tensor(v) for k, v in g.items() if k.startswith("out_")}
    ref, aux_ref = O.loss_fn(out, w)
    batch = {k: torch.stack([v, v]) for k, v in w.items() if k != "t"}
    batch["t"] = torch.cat([w["t"], w["t"]])
    outb = {k: torch.stack([v, v]) for k, v in out.items()}
    loss, aux = experiment.loss_fn(outb, batch)
    assert abs(float(loss) - float(ref)) < 1e-6 * abs(float(ref))
    assert abs(float(loss) - float(g["loss"])) < 1e-4 * abs(float(g["loss"]))      # reference-minted value
    for k in aux:
        assert abs(float(aux[k]) - float(aux_ref[k])) < 1e-6 * max(1.0, abs(float(aux_ref[k])))


def _dp_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from dynamicpdb_amd import experiment
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Tanh(), torch.nn.Linear(16, 3))
    dead = torch.nn.Linear(4, 4)                           # never receives a gradient (like the reference's 91,540 dead params)
    holder = torch.nn.ModuleList([model, dead])
    tr = experiment.Trainer(holder, lr=1e-2, bucket_bytes=256)   # tiny buckets: several collectives
    x = torch.randn(5, 8, generator=torch.Generator().manual_seed(100 + rank))
    tr.opt.zero_grad(set_to_none=True)
    model(x).pow(2).mean().backward()
    local = [p.grad.clone() for p in model.parameters()]
    tr.allreduce_grads()
    q.put((rank, [g.numpy() for g in local], [p.grad.numpy().copy() for p in model.parameters()],
           [p.grad is None for p in dead.parameters()]))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_gradient_average_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    mean = [(a + b) / 2 for a, b in zip(res[0][1], res[1][1])]
    for r in res:
        for got, want in zip(r[2], mean):
            assert np.allclose(got, want, rtol=1e-6, atol=1e-8)
        assert all(r[3])


def test_conv_tower_cone_ranges_and_splitk_model(monkeypatch):
    """Frame ranges of the last-frame dependency cone (pure host logic): nested, each 5x5 layer adds 2 frames, clipped to
    the window; and the split-K cost model picks divisors of the chunk count that fill whole rounds of CUs."""
    from dynamicpdb_amd import ops
    for F in (1, 3, 8, 16, 17, 32, 64):
        for i in (3, 2, 1, 0):
            (l1, n1), (l2, n2) = ops.ConvTower.cone(F, i)
            r = 4 * (3 - i)
            assert n2 == min(F, r + 1) and n1 == min(F, r + 3) and l1 == F - n1 and l2 == F - n2
            assert l1 <= l2 and l1 + n1 == F and l2 + n2 == F              # suffix ranges, the inner activation wider
            if i < 3:   # what block i delivers covers what block i+1 reads: its inner range widened by one conv (2 frames)
                assert n2 == min(F, ops.ConvTower.cone(F, i + 1)[0][1] + 2)
        assert ops.ConvTower.cone(F, 3)[1] == (F - 1, 1)
    monkeypatch.setitem(ops._N_CU, "dev", 256)
    monkeypatch.delenv("DFOLD_CONV_SPLITK", raising=False)
    # (rows, CO, CI) of the narrow launches at config 3 (8 windows x nf frames x 256 residues)
    assert ops.conv_splitk(8 * 3 * 256, 640, 1280, "dev") == 5      # 48 tiles  -> 240 workgroups
    assert ops.conv_splitk(8 * 7 * 256, 640, 1280, "dev") == 2      # 112 tiles -> 224
    assert ops.conv_splitk(8 * 15 * 256, 640, 1280, "dev") == 1     # 240 tiles already fill one round
    assert ops.conv_splitk(8 * 1 * 256, 1280, 640, "dev") == 5      # 32 tiles
    assert ops.conv_splitk(8 * 32 * 256, 1280, 640, "dev") == 1     # full layer: whole rounds, no split
    assert ops.conv_splitk(8 * 3 * 256, 600, 1280, "dev") == 1      # N tile does not divide: not eligible
    monkeypatch.setenv("DFOLD_CONV_SPLITK", "0")
    assert ops.conv_splitk(8 * 3 * 256, 640, 1280, "dev") == 1
