"""GPU parity of the pair-stack drop-ins (dynamicpdb_amd/model/pair_stack.py) against golden vectors minted from the
reference's vendored OpenFold modules (tests/golden/pair_stack_S6_N24.npz): OuterProductMean with every gradient,
EvoformerBlockCore (MSA transition, outer product mean, the four triangle updates, pair transition) in eval mode with
both input gradients and all parameter-gradient norms; the shared-mask dropout layers; a batched / ragged OPM against
the oracle.  bf16 operands / fp32 accumulation: tolerances as DESIGN.md (1.5e-2 forward, 3e-2 .. 5e-2 gradients)."""
import numpy as np
import pytest
import torch

from util import load_golden, rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_outer_product_mean_vs_reference_golden():
    from dynamicpdb_amd.model.pair_stack import OuterProductMean
    dev = torch.device(DEV)
    g = load_golden("pair_stack_S6_N24.npz")
    m = OuterProductMean(64, 128, 32)
    m.load_state_dict({k[6:]: torch.tensor(v) for k, v in g.items() if k.startswith("opm.P.")}, strict=True)
    m.to(dev)
    x = torch.tensor(g["m"]).to(dev).requires_grad_(True)
    y = m(x, mask=torch.tensor(g["msa_mask"]).to(dev))
    assert rel_l2(y, g["opm.out"]) < 1.5e-2, rel_l2(y, g["opm.out"])
    y.backward(torch.tensor(g["opm.gy"]).to(dev))
    assert rel_l2(x.grad, g["opm.gm"]) < 5e-2, rel_l2(x.grad, g["opm.gm"])
    for k, p in m.named_parameters():
        assert p.grad is not None and rel_l2(p.grad, g["opm.G." + k]) < 5e-2, (k, rel_l2(p.grad, g["opm.G." + k]))


def test_evoformer_block_core_vs_reference_golden():
    from dynamicpdb_amd.model.pair_stack import EvoformerBlockCore
    dev = torch.device(DEV)
    g = load_golden("pair_stack_S6_N24.npz")
    core = EvoformerBlockCore(c_m=64, c_z=128, c_hidden_opm=32, c_hidden_mul=128, c_hidden_pair_att=32, no_heads_msa=8,
                              no_heads_pair=4, transition_n=2, pair_dropout=0.25, inf=1e9, eps=1e-10)
    core.load_state_dict({k[7:]: torch.tensor(v) for k, v in g.items() if k.startswith("core.P.")}, strict=True)
    core.to(dev).eval()
    m = torch.tensor(g["m"]).to(dev).requires_grad_(True)
    z = torch.tensor(g["z"]).to(dev).requires_grad_(True)
    mo, zo = core(m, z, msa_mask=torch.tensor(g["msa_mask"]).to(dev), pair_mask=torch.tensor(g["pair_mask"]).to(dev))
    assert rel_l2(mo, g["core.m_out"]) < 1e-2 and rel_l2(zo, g["core.z_out"]) < 1.5e-2, (rel_l2(mo, g["core.m_out"]), rel_l2(zo, g["core.z_out"]))
    ((mo * torch.tensor(g["core.gm_out"]).to(dev)).sum() + (zo * torch.tensor(g["core.gz_out"]).to(dev)).sum()).backward()
    assert rel_l2(m.grad, g["core.gm"]) < 5e-2 and rel_l2(z.grad, g["core.gz"]) < 5e-2, (rel_l2(m.grad, g["core.gm"]), rel_l2(z.grad, g["core.gz"]))
    gn = np.array([float(p.grad.norm()) for _, p in core.named_parameters()])
    assert np.allclose(gn, g["core.gnorm"], rtol=5e-2, atol=1e-4 * float(g["core.gnorm"].max())), np.abs(gn / g["core.gnorm"] - 1).max()


def test_outer_product_mean_batched_ragged_vs_oracle():
    """two batch items, N_seq = 5 (padded to 8 inside), N_res = 20, c_m = 256 as openfold/config.py"""
    from oracle import dfold_oracle as O
    from dynamicpdb_amd.model.pair_stack import OuterProductMean
    dev = torch.device(DEV)
    rng = np.random.default_rng(4)
    m = OuterProductMean(256, 128, 32)
    sd = m.state_dict()
    for k, v in sd.items():
        sd[k] = torch.tensor((rng.standard_normal(tuple(v.shape)) / np.sqrt(v.shape[-1]) if v.dim() >= 2 else
                              (1.0 if k.endswith("weight") else 0.0) + 0.1 * rng.standard_normal(tuple(v.shape))).astype(np.float32))
    m.load_state_dict(sd)
    P = {k: v.clone() for k, v in sd.items()}
    m.to(dev)
    x = torch.tensor(rng.standard_normal((2, 5, 20, 256), dtype=np.float32))
    mask = torch.tensor((rng.uniform(size=(2, 5, 20)) > 0.2).astype(np.float32))
    with torch.no_grad():
        y = m(x.to(dev), mask=mask.to(dev), chunk_size=4)
    ref = torch.stack([O.outer_product_mean(P, x[b], mask[b]) for b in range(2)])
    assert y.shape == (2, 20, 20, 128) and rel_l2(y, ref) < 1.5e-2, rel_l2(y, ref)


def test_shared_mask_dropout():
    """DropoutRowwise / DropoutColumnwise (openfold/model/dropout.py:65-78): identity in eval mode; in training mode whole
    rows / columns are kept (scaled by 1/(1-r)) or zeroed together."""
    from dynamicpdb_amd.model.pair_stack import DropoutColumnwise, DropoutRowwise
    dev = torch.device(DEV)
    x = torch.ones(2, 64, 48, 8, device=dev)
    for cls, dim in ((DropoutRowwise, 1), (DropoutColumnwise, 2)):
        d = cls(0.25)
        assert d.batch_dim == [-3 if dim == 1 else -2]
        d.eval()
        assert torch.equal(d(x), x)
        d.train()
        torch.manual_seed(0)
        y = d(x)
        vals = torch.unique(y)
        assert all(abs(v) < 1e-6 or abs(v - 1 / 0.75) < 1e-5 for v in vals.cpu().tolist()), vals
        assert torch.equal(y, y.select(dim, 0).unsqueeze(dim).expand_as(y))      # the mask is shared along that axis
        frac = float((y == 0).float().mean())
        assert 0.1 < frac < 0.4
